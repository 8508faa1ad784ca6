"""Data-driven surrogate (maximum likelihood = MSE against FEniCS targets) on MI355X -- drop-in for the reference's
train_codec_max_likelihood.py: same DenseED, same Parser flags / defaults / run-directory layout / output files, the
loop body of :197-211 (F.mse_loss, one-cycle LR, Adam) and the test() of :166-195 (eval-mode MSE, NRMSE, R^2).

  --mode fused  (default)  pde_surrogate_amd.train.MaxLikelihoodTrainer: the fused step of the mixed-residual
                           trainer with pdes_mse_loss as the loss launch (device-resident data, flat Adam kernel,
                           RCCL all-reduce under torchrun);
  --mode dropin            the reference's loop body verbatim on the drop-in modules (model(input),
                           pde_surrogate_amd.metrics.mse_loss, loss.backward(), torch.optim.Adam).

The loss needs output data: `--synthetic` is not available here; datasets are the reference's HDF5 files (or .npz
files with the same stem holding `input` and `output`).
"""
import json
import time

import numpy as np
import torch
from pde_surrogate_amd import optim          # the reference's `import torch.optim as optim`, redirected

from pde_surrogate_amd import parallel
from pde_surrogate_amd.metrics import TestMetrics, mse_launch, mse_loss
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MaxLikelihoodTrainer
from pde_surrogate_amd.utils.load import DeviceLoader, y_variation
from pde_surrogate_amd.utils.misc import mkdirs
from pde_surrogate_amd.utils.plot import plot_prediction_det, save_stats
from pde_surrogate_amd.utils.practices import OneCycleScheduler, adjust_learning_rate

import train_codec_mixed_residual as _mr

# the reference's flag table (train_codec_max_likelihood.py:25-56): the mixed-residual one without --weight-bound and
# with its own defaults for --exp-name, --epochs and --ckpt-freq
_OVERRIDES = {'--exp-name': 'codec/max_likelihood', '--epochs': 200, '--ckpt-freq': 50}
_FLAGS = [(f, t, _OVERRIDES.get(f, d), c) for f, t, d, c in _mr._REFERENCE_FLAGS if f != '--weight-bound']


class Parser(_mr.Parser):
    description = 'Learning data-driven surrogate with MLE (MI355X HIP build)'
    flags = _FLAGS

    def parse(self, argv=None, rank=0, world=1):
        args = super().parse(argv, rank, world)
        if args.synthetic:
            raise SystemExit('--synthetic has no targets: the maximum-likelihood harness needs the datasets '
                             '(reference HDF5 files, or .npz files with `input` and `output`)')
        return args


def main(argv=None):
    rank, local_rank, world = parallel.init_from_env()
    args = Parser().parse(argv, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('this build runs on an MI355X (ROCm) only -- there is no CPU fallback for the HIP kernels')
    device = (parallel.local_device(local_rank) if world > 1 else torch.device('cuda', args.cuda % torch.cuda.device_count()))
    torch.cuda.set_device(device)
    is_main = rank == 0
    say = print if is_main else (lambda *a, **k: None)

    args.train_dir = args.run_dir + '/training'
    args.pred_dir = args.train_dir + '/predictions'
    if is_main:
        mkdirs(args.train_dir, args.pred_dir)
    if world > 1:
        torch.distributed.barrier()

    model = DenseED(in_channels=1, out_channels=3, imsize=args.imsize, blocks=args.blocks,
                    growth_rate=args.growth_rate, init_features=args.init_features,
                    drop_rate=args.drop_rate, out_activation=None, upsample=args.upsample)
    if args.ckpt_epoch is not None:
        ckpt_file = args.run_dir + f'/checkpoints/model_epoch{args.ckpt_epoch}.pth'
        model.load_state_dict(torch.load(ckpt_file, map_location='cpu'))
        say(f'Loaded ckpt: {ckpt_file}')
        say(f'training from epoch {args.ckpt_epoch + 1} to {args.epochs}')
    model = model.to(device)

    x_train, x_test, y_test, y_train = _mr.make_arrays(args, only_input=False)
    y_test_variation = y_variation(y_test)
    say(f'Test output variation per channel: {y_test_variation}')
    train_loader = DeviceLoader(torch.from_numpy(x_train), torch.from_numpy(y_train), batch_size=args.batch_size,
                                device=device, seed=args.seed, rank=rank, world_size=world, order='reference')
    test_loader = DeviceLoader(torch.from_numpy(x_test), torch.from_numpy(y_test), batch_size=args.test_batch_size,
                               device=device, shuffle=True, order='reference')   # (the reference's load_data shuffles both)
    say(f'# out pixels: {y_test[0].size}')

    scheduler = OneCycleScheduler(lr_max=args.lr, div_factor=args.lr_div, pct_start=args.lr_pct)
    trainer = None
    if args.mode == 'fused':
        trainer = MaxLikelihoodTrainer(model, args.batch_size, args.imsize, lr=args.lr, weight_decay=args.weight_decay,
                                       device=device, use_graph=args.graph)
        parallel.broadcast_buffers(model)
    else:
        if world > 1:
            raise SystemExit('--mode dropin is the single-GPU reference loop; use --mode fused with torchrun')
        optimizer = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)

    logger = {'loss_train': [], 'loss_test': [], 'r2_test': [], 'nrmse_test': []}
    metrics = TestMetrics(3, device)

    def test(epoch):
        model.eval()
        metrics.reset()
        loss_accum = torch.zeros(1, device=device, dtype=torch.float64)
        nb = 0
        for input, target in test_loader:
            output = model(input)
            mse_launch(output, target, False, loss_accum)        # F.mse_loss, accumulated on the device
            metrics.update(output, target)
            nb += 1
            if (epoch % args.plot_freq == 0 or epoch == args.epochs) and nb == len(test_loader) and is_main:
                n_samples = 6 if epoch == args.epochs else 2
                idx = torch.randperm(input.size(0))[:n_samples]
                so, st = output.cpu()[idx].numpy(), target.cpu()[idx].numpy()
                for i in range(n_samples):
                    print('epoch {}: plotting prediction {}'.format(epoch, i))
                    plot_prediction_det(args.pred_dir, st[i], so[i], epoch, i, plot_fn=args.plot_fn)
        relative_l2, r2_score = metrics.result(y_test_variation)   # the one host sync of the test pass
        loss_test = float(loss_accum[0]) / nb
        say(f"Epoch: {epoch}, test r2-score:  {r2_score}, relative-l2:  {relative_l2}")
        if epoch % args.log_freq == 0:
            logger['loss_test'].append(loss_test)
            logger['r2_test'].append(r2_score)
            logger['nrmse_test'].append(relative_l2)

    say('Start training........................................................')
    start_epoch = 1 if args.ckpt_epoch is None else args.ckpt_epoch + 1
    tic = time.time()
    total_steps = args.epochs * len(train_loader)
    say(f'total steps: {total_steps}')
    train_seconds = 0.0
    for epoch in range(start_epoch, args.epochs + 1):
        model.train()
        torch.cuda.synchronize(device)
        t0 = time.time()
        if args.mode == 'fused':
            for batch_idx, (input, target) in enumerate(train_loader, start=1):
                step = (epoch - 1) * len(train_loader) + batch_idx
                lr = scheduler.step(step / total_steps)
                trainer.step(input, target, lr)
            loss_train = parallel.mean_over_ranks(trainer.epoch_means())[0]
        else:
            loss_train = 0.
            for batch_idx, (input, target) in enumerate(train_loader, start=1):
                model.zero_grad()
                output = model(input)
                loss = mse_loss(output, target)
                loss.backward()
                step = (epoch - 1) * len(train_loader) + batch_idx
                lr = scheduler.step(step / total_steps)
                adjust_learning_rate(optimizer, lr)
                optimizer.step()
                loss_train += loss.item()
            loss_train /= batch_idx
        torch.cuda.synchronize(device)
        train_seconds += time.time() - t0
        say(f'Epoch {epoch}, lr {lr:.6f}')
        say(f'Epoch {epoch}: training loss: {loss_train:.6f}')
        if epoch % args.log_freq == 0:
            logger['loss_train'].append(loss_train)
        if epoch % args.ckpt_freq == 0 and is_main:
            torch.save(model.state_dict(), args.ckpt_dir + "/model_epoch{}.pth".format(epoch))
        with torch.no_grad():
            test(epoch)

    tic2 = time.time()
    say(f'Finished training {args.epochs} epochs with {args.ntrain} data using {(tic2 - tic) / 60:.2f} mins')
    if is_main:
        save_stats(args.train_dir, logger, 'loss_train', 'loss_test', 'nrmse_test', 'r2_test')
        args.training_time = tic2 - tic
        args.train_samples_per_sec = args.ntrain * (args.epochs - start_epoch + 1) / max(train_seconds, 1e-9)
        args.n_params, args.n_layers = model.model_size
        with open(args.run_dir + "/args.txt", 'w') as args_file:
            json.dump(vars(args), args_file, indent=4)
    if world > 1:
        if trainer is not None:
            trainer.close()                  # the direct RCCL communicator goes before its process group
        torch.distributed.destroy_process_group()
    return logger


if __name__ == '__main__':
    main()
