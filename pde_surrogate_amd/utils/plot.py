"""Output files of the training harness -- same names/formats as the reference's utils/plot.py
(`save_stats` :261-273 writes {metric}.txt via np.savetxt + {metric}.pdf; `plot_prediction_det`
:17-94 writes pred_epoch{e}_{i}.png with rows simulation / prediction / difference; `plot_prediction_bayes2`
:166-251 writes pred_epoch{e}_{i}.pdf with rows simulation / predictive mean / error / predictive variance;
`save_samples` :276-330 writes {name}_epoch{e}_idx{i}_output{c}.png, one grid of samples per output field)."""
import numpy as np

from .misc import to_numpy


def _plt():
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    return plt


def save_stats(save_dir, logger, *metrics):
    plt = _plt()
    for metric in metrics:
        values = logger[metric]
        np.savetxt(save_dir + f'/{metric}.txt', values)
        arr = np.loadtxt(save_dir + f'/{metric}.txt')
        if arr.ndim == 0:
            arr = arr[None]
        if arr.ndim == 1:
            arr = arr[:, None]
        lines = plt.plot(range(1, len(arr) + 1), arr)
        plt.legend(lines, [f'{arr[-5:, i].mean():.4f}' for i in range(arr.shape[-1])])
        plt.savefig(save_dir + f'/{metric}.pdf')
        plt.close()


def plot_prediction_det(save_dir, target, prediction, epoch, index, plot_fn='contourf', cmap='jet',
                        same_scale=False, row_labels=None, col_labels=None):
    """one figure per test input: 3 rows (simulation, prediction, difference) x n_fields columns"""
    plt = _plt()
    target, prediction = to_numpy(target), to_numpy(prediction)
    rows = row_labels or ['Simulation', 'Prediction', r'Simulation $-$ Prediction']
    cols = col_labels or ['Pressure', 'Horizontal Flux', 'Vertical Flux']
    nf = target.shape[0]
    fields = np.concatenate((target, prediction, target - prediction), axis=0)
    lo = [min(fields[i].min(), fields[i + nf].min()) for i in range(nf)]
    hi = [max(fields[i].max(), fields[i + nf].max()) for i in range(nf)]
    fig, axes = plt.subplots(3, nf, figsize=(3.75 * nf, 9))
    for j, ax in enumerate(np.atleast_1d(axes).ravel()):
        ax.set_aspect('equal')
        ax.set_xticks([])
        ax.set_yticks([])
        shared = j < 2 * nf or same_scale
        kw = dict(vmin=lo[j % nf], vmax=hi[j % nf]) if shared else {}
        if plot_fn == 'contourf':
            im = ax.contourf(fields[j], 50, cmap=cmap, **kw)
        else:
            im = ax.imshow(fields[j], cmap=cmap, origin='upper', **kw)
        cbar = fig.colorbar(im, ax=ax, fraction=0.046, pad=0.04)
        cbar.formatter.set_powerlimits((-2, 2))
        cbar.update_ticks()
    for ax, c in zip(np.atleast_2d(axes)[0], cols):
        ax.set_title(c, size='large')
    for ax, r in zip(np.atleast_2d(axes)[:, 0], rows):
        ax.set_ylabel(r, rotation=90, size='large')
    fig.tight_layout(pad=0.05, w_pad=0.05, h_pad=0.05)
    fig.savefig(save_dir + '/pred_epoch{}_{}.png'.format(epoch, index), bbox_inches='tight')
    plt.close(fig)


def _grid(ax, field, plot_fn, cmap, **kw):
    ax.set_aspect('equal')
    ax.set_xticks([])
    ax.set_yticks([])
    return ax.contourf(field, 50, cmap=cmap, **kw) if plot_fn == 'contourf' else ax.imshow(field, cmap=cmap, origin='upper', **kw)


def plot_prediction_bayes2(save_dir, target, pred_mean, pred_var, epoch, index, plot_fn='imshow', cmap='jet',
                           same_scale=False):
    """the reference's figure (utils/plot.py:181-258 there): 4 rows -- Simulation, Pred Mean, Pred Std (the SQUARE ROOT of
    the predictive variance), Sim - Pred Mean -- x n_fields columns; the first two rows of a column share a colour scale"""
    plt = _plt()
    target, pred_mean, pred_std = to_numpy(target), to_numpy(pred_mean), np.sqrt(to_numpy(pred_var))
    nf = target.shape[0]
    rows = ['Simulation', 'Pred Mean', 'Pred Std', r'Sim $-$ Pred Mean']
    fields = np.concatenate((target, pred_mean, pred_std, target - pred_mean), axis=0)
    fig, axes = plt.subplots(4, nf, figsize=(3.75 * nf, 12))
    for j, ax in enumerate(np.atleast_1d(axes).ravel()):
        kw = {}
        if j < 2 * nf or same_scale:
            c = j % nf
            kw = dict(vmin=min(fields[c].min(), fields[c + nf].min()), vmax=max(fields[c].max(), fields[c + nf].max()))
        im = _grid(ax, fields[j], plot_fn, cmap, **(kw if j < 2 * nf else {}))
        cbar = fig.colorbar(im, ax=ax, fraction=0.046, pad=0.04)
        cbar.formatter.set_powerlimits((-2, 2))
        cbar.update_ticks()
    for ax, r in zip(np.atleast_2d(axes)[:, 0], rows):
        ax.set_ylabel(r, rotation=90, size='large')
    fig.tight_layout(pad=0.05, w_pad=0.05, h_pad=0.05)
    fig.savefig(save_dir + '/pred_epoch{}_{}.pdf'.format(epoch, index), bbox_inches='tight')
    plt.close(fig)


def save_samples(save_dir, images, epoch, index, name, nrow=4, heatmap=True, cmap='jet'):
    """images: (n, C, H, W) -- the first is conventionally the target; one grid per output field"""
    plt = _plt()
    images = to_numpy(images)
    n, nc = images.shape[0], images.shape[1]
    ncol = int(np.ceil(n / nrow))
    for c in range(nc):
        fig, axes = plt.subplots(nrow, ncol, figsize=(2.5 * ncol, 2.5 * nrow))
        for k, ax in enumerate(np.atleast_1d(axes).ravel()):
            ax.set_xticks([])
            ax.set_yticks([])
            if k < n:
                ax.imshow(images[k, c], cmap=cmap if heatmap else 'gray', origin='upper')
            else:
                ax.axis('off')
        fig.tight_layout(pad=0.05)
        fig.savefig(save_dir + f'/{name}_epoch{epoch}_idx{index}_output{c}.png', bbox_inches='tight')
        plt.close(fig)
