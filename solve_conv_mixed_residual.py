"""Solve ONE Darcy-flow instance with a convolutional decoder and the mixed-residual loss on MI355X --
drop-in for the reference's solve_conv_mixed_residual.py (same flags; config 5 of BASELINE.json with
--nonlinear --alpha1 0.1 --alpha2 0.1).

    div sigma = 0,  sigma = -K grad u  (or  -K grad u = sigma + a1 sqrt(K) sigma^2 + a2 K sigma^3),
    u = 1 on the left edge, u = 0 on the right edge, no flux through top/bottom.

`Decoder(fixed latent)` -> (u, sigma1, sigma2); the loss and its gradient come from the fused HIP
kernel, the network forward/backward from the HIP convolution kernels; torch.optim.LBFGS drives the
closure on the host exactly as in the reference (lr 0.5, max_iter 20, history 50).

  --mode fused  (default)  the closure is pde_surrogate_amd.solver.ResidualClosure: forward + loss + backward through
                           the C ABI as eager launches on three streams (round 4: 1,316 closure evaluations / 45.4
                           L-BFGS epochs per second; --graph replays them as ONE serial hipGraph: 1,218 / 43.6 --
                           the graph was the faster form until the weight-gradient streams became per device and
                           the packing launch lean, tools/bench_solver.py); L-BFGS is pde_surrogate_amd.lbfgs.FlatLBFGS (torch.optim.LBFGS's
                           algorithm on the flat parameter buffer: two bandwidth-bound passes over the curvature
                           history per iteration instead of ~200 tiny kernels and host reads);
  --mode dropin            the reference's closure verbatim on the drop-in modules (autograd).

The reference validates the nonlinear case against a FEniCS solve (utils/fenics.py); dolfin is not
available here (parity unpinned for that comparison); tests/test_solver_gpu.py compares the fields this
script produces with the build's own independent fp64 finite-volume Newton solution (oracle/fd_newton.py, test
infrastructure): a self-consistency check, not FEniCS parity.
Inputs: an HDF5/.npz file with `input` (and optionally `output`), or --synthetic.
"""
import argparse
import os
import time
from pprint import pprint

import numpy as np
import torch

from pde_surrogate_amd.models.codec import Decoder
from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss
from pde_surrogate_amd.lbfgs import FlatLBFGS
from pde_surrogate_amd.solver import ResidualClosure
from pde_surrogate_amd.utils.load import read_arrays
from pde_surrogate_amd.utils.misc import mkdirs, to_numpy
from pde_surrogate_amd.utils.plot import plot_prediction_det, save_stats

_FLAGS = [
    ('--exp-dir', str, './experiments/solver'), ('--data-dir', str, './datasets'), ('--data', str, 'grf'),
    ('--kle', int, 512), ('--imsize', int, 64), ('--idx', int, 8), ('--alpha1', float, 1.0), ('--alpha2', float, 1.0),
    ('--nz', int, 1), ('--blocks', list, [8, 6]), ('--weight-bound', float, 10), ('--lr', float, 0.5),
    ('--epochs', int, 500), ('--test-freq', int, 50), ('--ckpt-freq', int, 250), ('--cmap', str, 'jet'),
    ('--cuda', int, 1),
]


def build_parser():
    p = argparse.ArgumentParser(description='CNN to solve PDE (MI355X HIP build)')
    for flag, typ, default in _FLAGS:
        kw = {'type': typ, 'default': default}
        if flag == '--data':
            kw['choices'] = ['grf', 'channelized', 'warped_grf']
        p.add_argument(flag, **kw)
    p.add_argument('--nonlinear', action='store_true', default=False, help='nonlinear corrections to Darcy law')
    p.add_argument('--same-scale', action='store_true')
    p.add_argument('--animate', action='store_true')
    p.add_argument('-v', '--verbose', action='store_true')
    p.add_argument('--synthetic', action='store_true', help='generate the permeability field instead of reading HDF5')
    p.add_argument('--mode', type=str, default='fused', choices=['fused', 'dropin'], help='closure (module docstring)')
    p.add_argument('--graph', action='store_true', help='fused mode: the closure as one serial hipGraph replay instead of eager launches')
    p.add_argument('--no-graph', action='store_true', help='(default since round 4; kept for older command lines)')
    return p


def dataset_file(args):
    if args.data == 'grf':
        assert args.kle in [512, 128, 1024, 2048]
        ntest = 1000 if args.kle == 512 else 1024
        return args.data_dir + f'/{args.imsize}x{args.imsize}/kle{args.kle}_lhs{ntest}_test.hdf5'
    if args.data == 'warped_grf':
        return args.data_dir + f'/{args.imsize}x{args.imsize}/warped_gp_ng64_n1000.hdf5'
    return args.data_dir + f'/{args.imsize}x{args.imsize}/channel_ng64_n512_test.hdf5'


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.blocks and isinstance(args.blocks[0], str):
        args.blocks = [int(c) for c in args.blocks if c.isdigit()]
    pprint(vars(args))
    if not torch.cuda.is_available():
        raise SystemExit('this build runs on an MI355X (ROCm) only -- there is no CPU fallback for the HIP kernels')
    device = torch.device('cuda', args.cuda % torch.cuda.device_count())
    torch.cuda.set_device(device)
    dataset = f'{args.data}_kle{args.kle}' if args.data == 'grf' else args.data
    hyparams = f'{dataset}_idx{args.idx}_dz{args.nz}_blocks{args.blocks}_lr{args.lr}_wb{args.weight_bound}_epochs{args.epochs}'
    exp_name = 'conv_mixed_residual_nonlinear' if args.nonlinear else 'conv_mixed_residual'
    if args.nonlinear:
        hyparams += f'_alpha1_{args.alpha1}_alpha2_{args.alpha2}'
    run_dir = args.exp_dir + '/' + exp_name + '/' + hyparams
    mkdirs(run_dir)

    assert args.idx < 1000
    output_arr = None
    if args.synthetic:
        from pde_surrogate_amd.utils.data import channelized_fields, grf_kle_fields
        fields = (grf_kle_fields(args.idx + 1, args.imsize, args.kle, cache_dir='/tmp') if args.data != 'channelized'
                  else channelized_fields(args.idx + 1, args.imsize))
        perm_arr = fields[[args.idx]]
    else:
        x, y = read_arrays(dataset_file(args), None, only_input=False)
        perm_arr = np.asarray(x[[args.idx]], np.float32)
        if not args.nonlinear:
            output_arr = np.asarray(y[args.idx], np.float32)       # linear case: FEniCS reference ships with the data
    model = Decoder(args.nz, out_channels=3, blocks=args.blocks).to(device)
    print(f'model size: {model.model_size}')
    fixed_latent = (torch.randn(1, args.nz, 16, 16) * 0.5).to(device)
    perm_tensor = torch.from_numpy(perm_arr).to(device)
    optimizer = torch.optim.LBFGS(model.parameters(), lr=args.lr, max_iter=20, history_size=50)
    b1, b2 = (args.alpha1, args.alpha2) if args.nonlinear else (0.0, 0.0)
    logger = {'loss': []}
    n_closure = [0]
    fused = None
    if args.mode == 'fused':
        model.train()
        fused = ResidualClosure(model, fixed_latent, perm_tensor, args.weight_bound, args.nonlinear, b1, b2,
                                use_graph=args.graph and not args.no_graph)
        # the same algorithm and stopping rules as torch.optim.LBFGS, in place on the flat parameter / gradient buffers
        optimizer = FlatLBFGS(model._flat, model._gscratch, lr=args.lr, max_iter=20, history_size=50)

    def train(epoch):
        model.train()

        def fused_closure():
            loss = fused()
            n_closure[0] += 1
            if args.verbose:
                t = fused.terms.tolist()
                print(f'epoch {epoch}: loss {t[0]:6f}, energy {t[1] + t[2]:.6f}, diri {t[3]:.6f}, neum {t[4]:.6f}')
            return loss

        def closure():
            optimizer.zero_grad()
            output = model(fixed_latent)
            loss, energy, l_dir, l_neu = darcy_mixed_residual_loss(perm_tensor, output, args.weight_bound,
                                                                   args.nonlinear, b1, b2)
            loss.backward()
            n_closure[0] += 1
            if args.verbose:
                print(f'epoch {epoch}: loss {loss.item():6f}, energy {energy.item():.6f}, '
                      f'diri {l_dir.item():.6f}, neum {l_neu.item():.6f}')
            return loss

        loss = optimizer.step(fused_closure if fused is not None else closure)
        value = loss.item() if not isinstance(loss, float) else loss
        logger['loss'].append(value)
        print(f'epoch {epoch}: loss {value:.6f}')
        if epoch % args.ckpt_freq == 0:
            torch.save(model.state_dict(), run_dir + "/model_epoch{}.pth".format(epoch))

    def test(epoch):
        if epoch % args.epochs == 0 or epoch % args.test_freq == 0:
            with torch.no_grad():
                output = to_numpy(model(fixed_latent))
            if output_arr is not None:
                plot_prediction_det(run_dir, output_arr, output[0], epoch, args.idx, plot_fn='imshow', cmap=args.cmap,
                                    same_scale=args.same_scale)
            np.save(run_dir + f'/epoch{epoch}.npy', output[0])

    print('start training...')
    tic = time.time()
    for epoch in range(1, args.epochs + 1):
        train(epoch)
        test(epoch)
    torch.cuda.synchronize(device)
    dt = time.time() - tic
    print(f'Finished optimization for {args.epochs} epochs using {dt / 60:.3f} minutes '
          f'({n_closure[0]} closure evaluations, {n_closure[0] / dt:.1f} per second)')
    save_stats(run_dir, logger, 'loss')
    np.save(run_dir + '/input.npy', perm_arr[0, 0])
    return logger['loss'], n_closure[0] / dt


if __name__ == '__main__':
    main()
