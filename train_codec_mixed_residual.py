"""Physics-constrained surrogate training on MI355X -- drop-in for the reference's
train_codec_mixed_residual.py (same Parser flags/defaults, run-directory layout and output files).

DenseED forward/backward, the Sobel gradients and the Darcy mixed-residual loss run as hand-written
HIP kernels (pde_surrogate_amd).  Two loop bodies are available:

  --mode fused  (default)  pde_surrogate_amd.train.MixedResidualTrainer: device-resident data, fused
                           loss fwd+bwd, flat-gradient Adam kernel, optional hipGraph, one RCCL
                           all-reduce per step when launched with torchrun (one process per GPU);
  --mode dropin            the reference's loop body verbatim (model(input), constitutive /
                           continuity / boundary functions, loss.backward(), torch.optim.Adam)
                           on the drop-in modules.

Additive flags (not in the reference): --mode, --graph, --synthetic (generate GRF-KLE /
channelized inputs instead of reading the HDF5 files, which are not redistributed; R^2 / NRMSE need
the FEniCS targets of the real files and are reported as nan in that mode).

    python train_codec_mixed_residual.py --data grf_kle512 --ntrain 4096 --batch-size 32 --synthetic
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        train_codec_mixed_residual.py --ntrain 8192 --batch-size 32 --synthetic
"""
import argparse
import json
import os
import random
import time
from pprint import pprint

import numpy as np
import torch
from pde_surrogate_amd import optim          # the reference's `import torch.optim as optim`, redirected

from pde_surrogate_amd import parallel
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.models.darcy import conv_boundary_condition as boundary_condition
from pde_surrogate_amd.models.darcy import conv_constitutive_constraint as constitutive_constraint
from pde_surrogate_amd.models.darcy import conv_continuity_constraint as continuity_constraint
from pde_surrogate_amd.models.darcy import darcy_loss_launch
from pde_surrogate_amd.metrics import TestMetrics
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.image_gradient import SobelFilter
from pde_surrogate_amd.utils.load import DeviceLoader, load_data, read_arrays, y_variation
from pde_surrogate_amd.utils.misc import mkdirs, to_numpy
from pde_surrogate_amd.utils.plot import plot_prediction_det, save_stats
from pde_surrogate_amd.utils.practices import OneCycleScheduler, adjust_learning_rate


# (flag, type, default, choices) -- names, types and defaults are the reference's CLI contract
# (train_codec_mixed_residual.py:40-72 there); `--blocks` keeps its `type=list` quirk.
_REFERENCE_FLAGS = [
    ('--exp-name', str, 'codec/mixed_residual', None), ('--exp-dir', str, './experiments', None),
    ('--blocks', list, [6, 8, 6], None), ('--growth-rate', int, 16, None), ('--init-features', int, 48, None),
    ('--drop-rate', float, 0., None), ('--upsample', str, 'nearest', ['nearest', 'bilinear']),
    ('--data-dir', str, './datasets', None), ('--data', str, 'grf_kle512', ['grf_kle512', 'channelized']),
    ('--ntrain', int, 4096, None), ('--ntest', int, 512, None), ('--imsize', int, 64, None),
    ('--run', int, 1, None), ('--epochs', int, 300, None), ('--lr', float, 1e-3, None),
    ('--lr-div', float, 2., None), ('--lr-pct', float, 0.3, None), ('--weight-decay', float, 0., None),
    ('--weight-bound', float, 10, None), ('--batch-size', int, 32, None), ('--test-batch-size', int, 64, None),
    ('--seed', int, 1, None), ('--cuda', int, 1, list(range(8))),          # reference: choices 0-3; one node has 8
    ('--ckpt-epoch', int, None, None), ('--ckpt-freq', int, 100, None), ('--log-freq', int, 1, None),
    ('--plot-freq', int, 50, None), ('--plot-fn', str, 'imshow', ['contourf', 'imshow']),
]


def validate_args(args, world):
    """reject what cannot run BEFORE any directory is created: field sizes the network does not map back onto
    themselves, divisibility of the dataset by the GLOBAL batch.  The Sobel / Darcy-residual kernels take any square
    size (16 / 32 / 64: specialised kernels, everything else: csrc/darcy_loss_generic.hip); what restricts --imsize is
    DenseED itself, in the reference too: one stride-2 convolution per encoding stage (codec.py:242, :113-118) and one
    x2 upsampling per decoding stage give an output of the input's size -- which the loss needs (darcy.py:172-176
    multiplies input and output fields) -- only when imsize is a multiple of 2^(1 + encoding blocks)."""
    down = 2 ** (1 + len(args.blocks) // 2)
    if args.imsize < down or args.imsize % down:
        raise SystemExit(f'--imsize {args.imsize}: DenseED with blocks {args.blocks} halves the field {len(args.blocks) // 2 + 1} '
                         f'times and doubles it back; the output matches the {args.imsize} x {args.imsize} input only for '
                         f'multiples of {down}')
    if not (0.0 <= args.drop_rate < 1.0):
        raise SystemExit(f'--drop-rate {args.drop_rate} must be in [0, 1)')
    gb = args.batch_size * world
    if args.ntrain % gb:
        raise SystemExit(f'--ntrain {args.ntrain} is not a multiple of the global batch {args.batch_size} x {world} ranks')
    if args.ntest % args.test_batch_size:
        raise SystemExit(f'--ntest {args.ntest} is not a multiple of --test-batch-size {args.test_batch_size}')


class Parser(argparse.ArgumentParser):
    description = 'Learning surrogate with mixed residual norm loss (MI355X HIP build)'
    flags = _REFERENCE_FLAGS

    def __init__(self):
        super().__init__(description=self.description)
        for flag, typ, default, choices in self.flags:
            kw = {'type': typ, 'default': default}
            if choices is not None:
                kw['choices'] = choices
            self.add_argument(flag, **kw)
        self.add_argument('--debug', action='store_true', default=False)
        # additive flags of this build
        self.add_argument('--mode', type=str, default='fused', choices=['fused', 'dropin'],
                          help='loop body (see module docstring)')
        self.add_argument('--graph', action='store_true', default=False,
                          help='fused mode: capture the step in a hipGraph (default: eager launches, weight '
                               'gradients overlapped on a second HIP stream)')
        self.add_argument('--no-graph', action='store_true', default=False, help='(default; kept for older scripts)')
        self.add_argument('--synthetic', action='store_true', default=False,
                          help='generate inputs instead of reading HDF5 files')

    def parse(self, argv=None, rank=0, world=1):
        """rank / world: under torchrun every rank parses, only rank 0 creates directories, prints and writes args.txt"""
        args = self.parse_args(argv)
        if args.blocks and isinstance(args.blocks[0], str):       # reference quirk: type=list splits a CLI string
            args.blocks = [int(c) for c in args.blocks if c.isdigit()]
        validate_args(args, world)

        hparams = f'{args.data}_ntrain{args.ntrain}_run{args.run}_bs{args.batch_size}_lr{args.lr}_epochs{args.epochs}'
        if args.debug:
            hparams = 'debug/' + hparams
        args.run_dir = args.exp_dir + '/' + args.exp_name + '/' + hparams
        args.ckpt_dir = args.run_dir + '/checkpoints'
        if rank == 0:
            mkdirs(args.run_dir, args.ckpt_dir)

        assert args.ntrain % args.batch_size == 0 and args.ntest % args.test_batch_size == 0

        if args.seed is None:
            args.seed = random.randint(1, 10000)
        random.seed(args.seed)
        torch.manual_seed(args.seed)
        if rank == 0:
            print("Random Seed: ", args.seed)
            print('Arguments:')
            pprint(vars(args))
            with open(args.run_dir + "/args.txt", 'w') as args_file:
                json.dump(vars(args), args_file, indent=4)
        return args


def dataset_files(args):
    if args.data == 'grf_kle512':
        train = args.data_dir + f'/{args.imsize}x{args.imsize}/kle512_lhs10000_train.hdf5'
        test = args.data_dir + f'/{args.imsize}x{args.imsize}/kle512_lhs1000_val.hdf5'
        totals = (10000, 1000)
    else:
        train = args.data_dir + f'/{args.imsize}x{args.imsize}/channel_ng64_n4096_train.hdf5'
        test = args.data_dir + f'/{args.imsize}x{args.imsize}/channel_ng64_n512_test.hdf5'
        totals = (4096, 512)
    assert args.ntrain <= totals[0], f"Only {totals[0]} data available in {args.data} dataset, but needs {args.ntrain} training data."
    assert args.ntest <= totals[1], f"Only {totals[1]} data available in {args.data} dataset, but needs {args.ntest} test data."
    return train, test


def make_arrays(args, only_input=True):
    """(x_train, x_test, y_test or None[, y_train when only_input is False])"""
    if args.synthetic:
        from pde_surrogate_amd.utils.data import channelized_fields, grf_kle_fields
        if args.data == 'grf_kle512':
            x = grf_kle_fields(args.ntrain + args.ntest, args.imsize, 512, cache_dir='/tmp')
        else:
            x = channelized_fields(args.ntrain + args.ntest, args.imsize)
        return x[:args.ntrain], x[args.ntrain:], None
    train_file, test_file = dataset_files(args)
    if only_input is False:
        x_train, y_train = read_arrays(train_file, args.ntrain, only_input=False)
        x_test, y_test = read_arrays(test_file, args.ntest, only_input=False)
        return (np.asarray(x_train, np.float32), np.asarray(x_test, np.float32), np.asarray(y_test, np.float32),
                np.asarray(y_train, np.float32))
    x_train, _ = read_arrays(train_file, args.ntrain, only_input=True)
    x_test, y_test = read_arrays(test_file, args.ntest, only_input=False)
    return np.asarray(x_train, np.float32), np.asarray(x_test, np.float32), np.asarray(y_test, np.float32)


def main(argv=None):
    rank, local_rank, world = parallel.init_from_env()
    args = Parser().parse(argv, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('this build runs on an MI355X (ROCm) only -- there is no CPU fallback for the HIP kernels')
    device = (parallel.local_device(local_rank) if world > 1 else torch.device('cuda', args.cuda % torch.cuda.device_count()))
    torch.cuda.set_device(device)
    is_main = rank == 0
    say = print if is_main else (lambda *a, **k: None)
    if world > 1:                 # each rank onto its share of the cores of its GPU's NUMA node (PDES_PIN=0 disables)
        pin = parallel.pin_rank_to_gpu_numa(device, local_rank, parallel.local_world_size(world))
        say('host affinity (rank 0):', pin)

    args.train_dir = args.run_dir + '/training'
    args.pred_dir = args.train_dir + '/predictions'
    if is_main:
        mkdirs(args.train_dir, args.pred_dir)
    if world > 1:
        torch.distributed.barrier()          # the run directory exists before any rank goes on

    model = DenseED(in_channels=1, out_channels=3, imsize=args.imsize, blocks=args.blocks,
                    growth_rate=args.growth_rate, init_features=args.init_features,
                    drop_rate=args.drop_rate, out_activation=None, upsample=args.upsample)
    if args.debug and is_main:
        print(model)
    if args.ckpt_epoch is not None:
        ckpt_file = args.run_dir + f'/checkpoints/model_epoch{args.ckpt_epoch}.pth'
        model.load_state_dict(torch.load(ckpt_file, map_location='cpu'))
        say(f'Loaded ckpt: {ckpt_file}')
        say(f'Resume training from epoch {args.ckpt_epoch + 1} to {args.epochs}')
    model = model.to(device)

    x_train, x_test, y_test = make_arrays(args)
    have_targets = y_test is not None
    y_test_variation = y_variation(y_test) if have_targets else np.full(3, np.nan)
    say(f'Test output variation per channel: {y_test_variation}')
    # both loaders shuffle like the reference's (utils/load.py:34-35 there: load_data builds train AND test loader with
    # shuffle=True) and draw every epoch's permutation from torch's global generator the way its DataLoader does: with the
    # reference's seed this run trains on the reference's minibatches, in its order (tests/golden/G25)
    train_loader = DeviceLoader(torch.from_numpy(x_train), batch_size=args.batch_size, device=device,
                                seed=args.seed, rank=rank, world_size=world, order='reference')
    test_tensors = [torch.from_numpy(x_test)] + ([torch.from_numpy(y_test)] if have_targets else [])
    test_loader = DeviceLoader(*test_tensors, batch_size=args.test_batch_size, device=device, shuffle=True, order='reference')

    scheduler = OneCycleScheduler(lr_max=args.lr, div_factor=args.lr_div, pct_start=args.lr_pct)
    sobel_filter = SobelFilter(args.imsize, correct=True, device=device)
    trainer = None
    if args.mode == 'fused':
        trainer = MixedResidualTrainer(model, args.batch_size, args.imsize, lr=args.lr, weight_decay=args.weight_decay,
                                       weight_bound=args.weight_bound, device=device, use_graph=args.graph)
        parallel.broadcast_parameters(trainer.flat)
        parallel.broadcast_buffers(model)         # BatchNorm running statistics of a resumed checkpoint
    else:
        if world > 1:
            raise SystemExit('--mode dropin is the single-GPU reference loop; use --mode fused with torchrun')
        optimizer = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)

    logger = {'loss_train': [], 'loss_test': [], 'r2_test': [], 'nrmse_test': []}

    metrics = TestMetrics(3, device) if have_targets else None

    def test(epoch, loss_train):
        model.eval()
        loss_accum = torch.zeros(5, device=device, dtype=torch.float64)
        if metrics is not None:
            metrics.reset()
        nb = 0
        for batch in test_loader:
            input = batch[0]
            output = model(input)
            terms, _ = darcy_loss_launch(input, output, (1.0, 1.0, args.weight_bound, args.weight_bound), False)
            loss_accum += terms            # device accumulation: the test pass syncs once, below
            nb += 1
            if have_targets:
                target = batch[1]
                metrics.update(output, target)       # err2_sum / relative_l2 of the reference (:180-183), on device
                if (epoch % args.plot_freq == 0 or epoch == args.epochs) and nb == len(test_loader) and is_main:
                    n_samples = 6 if epoch == args.epochs else 2
                    idx = torch.randperm(input.size(0))[:n_samples]
                    so, st = output.cpu()[idx].numpy(), target.cpu()[idx].numpy()
                    for i in range(n_samples):
                        print('epoch {}: plotting prediction {}'.format(epoch, i))
                        plot_prediction_det(args.pred_dir, st[i], so[i], epoch, i, plot_fn=args.plot_fn)
        t = terms.cpu().tolist()               # last batch's terms, printed like the reference (:199-200)
        loss_test = float(loss_accum[0]) / nb
        if have_targets:
            rel, r2_score = metrics.result(y_test_variation)
        else:
            rel, r2_score = np.full(3, np.nan), np.full(3, np.nan)
        say(f"Epoch: {epoch}, test r2-score:  {r2_score}")
        say(f"Epoch: {epoch}, test relative-l2:  {rel}")
        say(f'Epoch {epoch}: test loss: {loss_test:.6f}, loss_pde: {t[1] + t[2]:.6f}, '
            f'dirichlet {t[3]:.6f}, nuemann {t[4]:.6f}')
        if epoch % args.log_freq == 0:
            logger['loss_test'].append(loss_test)
            logger['r2_test'].append(r2_score)
            logger['nrmse_test'].append(rel)

    say('Start training...................................................')
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
            for batch_idx, (input,) in enumerate(train_loader, start=1):
                step = (epoch - 1) * len(train_loader) + batch_idx
                lr = scheduler.step(step / total_steps)
                trainer.step(input, lr)
            # the only host sync of the epoch; under torchrun the five terms are averaged over the ranks (each rank
            # accumulated the means of its own shards)
            loss_train, l_const, l_cont, loss_dirichlet, loss_neumann = parallel.mean_over_ranks(trainer.epoch_means())
            loss_pde = l_const + l_cont
        else:
            loss_train = 0.
            for batch_idx, (input,) in enumerate(train_loader, start=1):
                model.zero_grad()
                output = model(input)
                loss_pde = constitutive_constraint(input, output, sobel_filter) + continuity_constraint(output, sobel_filter)
                loss_dirichlet, loss_neumann = boundary_condition(output)
                loss = loss_pde + (loss_dirichlet + loss_neumann) * args.weight_bound
                loss.backward()
                step = (epoch - 1) * len(train_loader) + batch_idx
                lr = scheduler.step(step / total_steps)
                adjust_learning_rate(optimizer, lr)
                optimizer.step()
                loss_train += loss.item()
            loss_train /= batch_idx
            loss_pde, loss_dirichlet, loss_neumann = float(loss_pde), float(loss_dirichlet), float(loss_neumann)
        torch.cuda.synchronize(device)
        train_seconds += time.time() - t0
        if is_main:
            print(f'Epoch {epoch}, lr {lr:.6f}')
            print(f'Epoch {epoch}: training loss: {loss_train:.6f}, pde: {loss_pde:.6f}, '
                  f'dirichlet {loss_dirichlet:.6f}, nuemann {loss_neumann:.6f}')
        if epoch % args.log_freq == 0:
            logger['loss_train'].append(loss_train)
        if epoch % args.ckpt_freq == 0 and is_main:
            torch.save(model.state_dict(), args.ckpt_dir + "/model_epoch{}.pth".format(epoch))
        with torch.no_grad():
            test(epoch, loss_train)

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


if __name__ == '__main__':
    main()
