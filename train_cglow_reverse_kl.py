"""Multiscale conditional Glow trained with the reverse KL divergence on MI355X -- drop-in for the reference's
train_cglow_reverse_kl.py (same Parser flags / defaults, run-directory layout, checkpoint keys and output files).
Training needs no output data: the loss is  beta * [mixed residual of the Darcy equations at y ~ p(y|x)] + E log p(y|x).

The flow, its backward pass and the Sobel / residual loss run as hand-written HIP kernels (pde_surrogate_amd).  Loop bodies:

  --mode fused  (default)  pde_surrogate_amd.train.ReverseKLTrainer: device-resident data, generate() as one descriptor
                           chain, fused loss fwd+bwd, flat-gradient Adam kernel, one RCCL all-reduce per step under
                           torchrun (one process per GPU);
  --mode dropin            the reference's loop body verbatim (model.generate, the three constraint functions,
                           loss.backward(), torch.optim.Adam) on the drop-in modules.

Additive flags: --mode, --synthetic (GRF-KLE inputs generated instead of read from HDF5: the datasets are not
redistributed; R^2 / NRMSE need the FEniCS targets of the real files and are reported as nan in that mode).
One deviation from the reference, on purpose: its test() reads the training loop's LAST `log_likeihood` (a module-level
variable, train_cglow_reverse_kl.py:179) for the test entropy; here the entropy of the test pass is the test pass's own.

    python train_cglow_reverse_kl.py --synthetic --ntrain 4096 --batch-size 32 --epochs 2 --cuda 0
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_cglow_reverse_kl.py --synthetic
"""
import argparse
import json
import math
import os
import random
import time
from pprint import pprint

import numpy as np
import torch

from pde_surrogate_amd import parallel
from pde_surrogate_amd.metrics import TestMetrics
from pde_surrogate_amd.models.darcy import conv_boundary_condition as boundary_condition
from pde_surrogate_amd.models.darcy import conv_constitutive_constraint as constitutive_constraint
from pde_surrogate_amd.models.darcy import conv_continuity_constraint as continuity_constraint
from pde_surrogate_amd.models.darcy import darcy_loss_launch
from pde_surrogate_amd.models.glow_msc import MultiScaleCondGlow
from pde_surrogate_amd.train import ReverseKLTrainer, adam_state_dict, load_adam_state, to_torch_adam_state
from pde_surrogate_amd.utils.image_gradient import SobelFilter
from pde_surrogate_amd.utils.load import DeviceLoader, read_arrays, y_variation
from pde_surrogate_amd.utils.misc import mkdirs
from pde_surrogate_amd.utils.plot import plot_prediction_bayes2, save_samples, save_stats
from pde_surrogate_amd.utils.practices import OneCycleScheduler, adjust_learning_rate

# (flag, type, default, choices) -- the reference's CLI contract (train_cglow_reverse_kl.py:29-70 there)
_REFERENCE_FLAGS = [
    ('--exp-name', str, 'cglow/reverse_kld', None), ('--exp-dir', str, './experiments', None),
    ('--enc-blocks', list, [3, 4, 4], None), ('--flow-blocks', list, [6, 6, 6], None),
    ('--data-dir', str, './datasets', None), ('--kle', int, 100, None), ('--ntrain', int, 4096, None),
    ('--ntest', int, 512, None), ('--x-channels', int, 1, None), ('--y-channels', int, 3, None), ('--imsize', int, 32, None),
    ('--epochs', int, 400, None), ('--lr', float, 1.5e-3, None), ('--lr-div', float, 2., None), ('--lr-pct', float, 0.3, None),
    ('--beta', float, 150, None), ('--weight-decay', float, 0., None), ('--weight-bound', float, 50, None),
    ('--batch-size', int, 32, None), ('--test-batch-size', int, 64, None), ('--seed', int, 1, None),
    ('--cuda', int, 1, None), ('--ckpt-epoch', int, None, None), ('--ckpt-freq', int, 25, None),
    ('--log-freq', int, 1, None), ('--plot-freq', int, 25, None), ('--plot-fn', str, 'imshow', ['contourf', 'imshow']),
]


def _int_list(v):
    """the reference declares the block lists with type=list, which splits a command-line string into characters"""
    return [int(c) for c in v if str(c).isdigit()] if v and isinstance(v[0], str) else [int(c) for c in v]


def validate_args(args, world):
    levels = len(args.flow_blocks)
    if len(args.enc_blocks) != levels:
        raise SystemExit('--enc-blocks and --flow-blocks must have the same length')
    if args.imsize % (1 << (levels - 1)) or args.imsize >> (levels - 1) < 2:
        raise SystemExit(f'--imsize {args.imsize} cannot be squeezed {levels - 1} times')
    if args.x_channels != 1 or args.y_channels != 3:
        raise SystemExit('the Darcy loss is defined for 1 input field and the 3 output fields (pressure, two fluxes)')
    # (no multiple-of-the-batch requirement: the reference's loaders drop the last partial batch, utils/load.py:34-35)
    if args.ntrain < args.batch_size * world:
        raise SystemExit(f'--ntrain {args.ntrain} is less than one global batch of {args.batch_size} x {world} ranks')
    if args.ntest < args.test_batch_size:
        raise SystemExit(f'--ntest {args.ntest} is less than --test-batch-size {args.test_batch_size}')


class Parser(argparse.ArgumentParser):
    def __init__(self):
        super().__init__(description='Training multiscale conditional Glows with reverse KLD loss (MI355X HIP build)')
        for flag, typ, default, choices in _REFERENCE_FLAGS:
            kw = {'type': typ, 'default': default}
            if choices is not None:
                kw['choices'] = choices
            self.add_argument(flag, **kw)
        self.add_argument('--no-LU-decompose', action='store_true', default=False)
        self.add_argument('--data-init', action='store_true', default=False, help='use data initialization for ActNorm')
        self.add_argument('--debug', action='store_true', default=False)
        self.add_argument('--resume', action='store_true', default=False)
        self.add_argument('--mode', type=str, default='fused', choices=['fused', 'dropin'])
        self.add_argument('--synthetic', action='store_true', default=False, help='generate inputs instead of reading HDF5 files')

    def parse(self, argv=None, rank=0, world=1):
        args = self.parse_args(argv)
        args.LU_decompose = not args.no_LU_decompose
        args.enc_blocks, args.flow_blocks = _int_list(args.enc_blocks), _int_list(args.flow_blocks)
        validate_args(args, world)
        hparams = f'kle{args.kle}_ntrain{args.ntrain}_ENC_blocks{args.enc_blocks}_FLOW_blocks{args.flow_blocks}_' \
                  f'wb{args.weight_bound}_beta{args.beta}_batch{args.batch_size}_lr{args.lr}_epochs{args.epochs}'
        if args.debug:
            hparams = 'debug/' + hparams
        if args.data_init:
            hparams = hparams + '_data_init'
        args.run_dir = args.exp_dir + '/' + args.exp_name + '/' + hparams
        args.ckpt_dir = args.run_dir + '/checkpoints'
        args.train_dir = args.run_dir + '/training'
        args.pred_dir = args.train_dir + '/predictions'
        if rank == 0:
            mkdirs(args.run_dir, args.ckpt_dir, args.train_dir, args.pred_dir)
        if args.seed is None:
            args.seed = random.randint(1, 10000)
        random.seed(args.seed)
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)              # the rotations of the invertible 1x1 convolutions come from numpy's stream
        args_file = args.run_dir + '/args.txt'
        if os.path.isfile(args_file) and args.ckpt_epoch is None and args.resume:
            with open(args_file) as f:
                args.ckpt_epoch = json.load(f).get('ckpt_epoch')
        elif rank == 0 and not os.path.isfile(args_file):
            with open(args_file, 'w') as f:
                json.dump(vars(args), f, indent=4)
        if rank == 0:
            print('Random Seed: ', args.seed)
            print('Arguments:')
            pprint(vars(args))
        return args


def make_arrays(args):
    """(x_train, x_test, y_test or None)"""
    if args.synthetic:
        from pde_surrogate_amd.utils.data import grf_kle_fields
        x = grf_kle_fields(args.ntrain + args.ntest, args.imsize, args.kle, cache_dir='/tmp')
        return x[:args.ntrain], x[args.ntrain:], None
    base = args.data_dir + f'/{args.imsize}x{args.imsize}/kle{args.kle}'
    x_train, _ = read_arrays(base + '_lhs10000_train.hdf5', args.ntrain, only_input=True)
    x_test, y_test = read_arrays(base + '_lhs1000_val.hdf5', args.ntest, only_input=False)
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
    if world > 1:
        torch.distributed.barrier()

    x_train, x_test, y_test = make_arrays(args)
    have_targets = y_test is not None
    y_test_variation = y_variation(y_test) if have_targets else np.full(3, np.nan)
    say(f'Test output variation per channel: {y_test_variation}')
    n_out_pixels = args.y_channels * args.imsize * args.imsize
    say(f'# out pixels per output: {n_out_pixels}')
    train_loader = DeviceLoader(torch.from_numpy(x_train), batch_size=args.batch_size, device=device, seed=args.seed,
                                rank=rank, world_size=world)
    test_tensors = [torch.from_numpy(x_test)] + ([torch.from_numpy(y_test)] if have_targets else [])
    test_loader = DeviceLoader(*test_tensors, batch_size=args.test_batch_size, device=device, shuffle=False)

    model = MultiScaleCondGlow(img_size=args.imsize, x_channels=args.x_channels, y_channels=args.y_channels,
                               enc_blocks=args.enc_blocks, flow_blocks=args.flow_blocks, LUdecompose=args.LU_decompose,
                               squeeze_factor=2, data_init=args.data_init)
    if args.debug and is_main:
        print(model)
    say(model.model_size)
    logger = {k: [] for k in ('loss_train', 'loss_test', 'nrmse_test', 'r2_test', 'entropy_train', 'entropy_test')}
    checkpoint = None
    if args.ckpt_epoch is not None:
        checkpoint = torch.load(args.ckpt_dir + f'/model_epoch{args.ckpt_epoch}.pth', map_location='cpu',
                                weights_only=False)        # (the logger holds numpy arrays)
        model.load_state_dict(checkpoint['model_state_dict'])
        if args.data_init:
            model.init_actnorm()
        logger = checkpoint['logger']
        say(f'Loaded checkpoint at epoch {args.ckpt_epoch}')
    model = model.to(device)
    if world > 1:
        torch.manual_seed(args.seed + 1000 * rank)      # the latents' noise differs between the ranks (the parameters do not)
    scheduler = OneCycleScheduler(lr_max=args.lr, div_factor=args.lr_div, pct_start=args.lr_pct)
    sobel_filter = SobelFilter(args.imsize, correct=True, device=device)
    trainer = None
    if args.mode == 'fused':
        trainer = ReverseKLTrainer(model, args.batch_size, args.imsize, lr=args.lr, weight_decay=args.weight_decay,
                                   weight_bound=args.weight_bound, beta=args.beta, device=device)
        parallel.broadcast_parameters(trainer.flat)
        parallel.broadcast_buffers(model)
        if checkpoint is not None:
            # a torch.optim.Adam state_dict (the reference's, --mode dropin's, --mode fused's) or a round-2 flat one
            if 'optimizer_state_dict' not in checkpoint:
                say('WARNING: the checkpoint holds no optimizer_state_dict: Adam restarts from zero moments')
            elif not load_adam_state(trainer, checkpoint['optimizer_state_dict']):
                say('WARNING: the checkpoint\'s optimizer never stepped: Adam restarts from zero moments')
    else:
        if world > 1:
            raise SystemExit('--mode dropin is the single-GPU reference loop; use --mode fused with torchrun')
        optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
        if checkpoint is not None:
            if 'optimizer_state_dict' not in checkpoint:
                say('WARNING: the checkpoint holds no optimizer_state_dict: Adam restarts from zero moments')
            else:
                model._engine(torch.zeros((args.batch_size, args.x_channels, args.imsize, args.imsize), device=device))   # flat layout
                optimizer.load_state_dict(to_torch_adam_state(checkpoint['optimizer_state_dict'], model, optimizer))
    metrics = TestMetrics(3, device) if have_targets else None

    def test(epoch):
        model.eval()
        if metrics is not None:
            metrics.reset()
        loss_accum = torch.zeros((), device=device, dtype=torch.float64)
        nb = 0
        for batch in test_loader:
            input = batch[0]
            if epoch % 10 == 0:            # every 10 epochs the predictive mean of 20 samples, otherwise one sample
                output = model.sample(input, n_samples=20, temperature=1.0).mean(0)
                logp = model.generate(input)[1]
            else:
                output, logp = model.generate(input)
            terms, _ = darcy_loss_launch(input, output, (1.0, 1.0, args.weight_bound, args.weight_bound), False)
            neg_entropy = logp.mean() / math.log(2.) / n_out_pixels
            loss_accum += terms[0].double() * args.beta + neg_entropy.double()
            nb += 1
            if have_targets:
                target = batch[1]
                metrics.update(output, target)
                if (epoch % args.plot_freq == 0 or epoch % args.epochs == 0) and nb == 1 and is_main:
                    n_samples = 6 if epoch == args.epochs else 2
                    idx = np.random.permutation(input.size(0))[:n_samples]
                    for i in range(n_samples):
                        print('epoch {}: plotting prediction {}'.format(epoch, i))
                        pred_mean, pred_var = model.predict(input[[idx[i]]])
                        plot_prediction_bayes2(args.pred_dir, target[idx[i]], pred_mean[0], pred_var[0], epoch, idx[i],
                                               plot_fn='imshow', cmap='jet', same_scale=False)
                        samples_pred = model.sample(input[[idx[i]]], n_samples=15)[:, 0]
                        save_samples(args.pred_dir, torch.cat((target[[idx[i]]], samples_pred), 0), epoch, idx[i],
                                     'samples', nrow=4, heatmap=True, cmap='jet')
        t = terms.cpu().tolist()
        loss_test = float(loss_accum) / nb
        if have_targets:
            relative_l2, r2_score = metrics.result(y_test_variation)
        else:
            relative_l2, r2_score = np.full(3, np.nan), np.full(3, np.nan)
        say(f'Epoch {epoch}: test r2-score:  {r2_score}')
        say(f'Epoch {epoch}: test relative l2:  {relative_l2}')
        say(f'Epoch {epoch}: test loss: {loss_test:.6f}, residual: {t[1] + t[2]:.6f}, boundary {t[3] + t[4]:.6f}, '
            f'neg entropy {float(neg_entropy):.6f}')
        if epoch % args.log_freq == 0:
            logger['loss_test'].append(loss_test)
            logger['r2_test'].append(r2_score)
            logger['nrmse_test'].append(relative_l2)
            logger['entropy_test'].append(-float(neg_entropy))

    say('Start training........................................................')
    tic = time.time()
    start_epoch = checkpoint['epoch'] + 1 if checkpoint is not None else 1
    initialized = start_epoch != 1
    total_steps = max((args.epochs - start_epoch) * len(train_loader), 1)
    say(f'total steps: {total_steps}')
    train_seconds = 0.0
    for epoch in range(start_epoch, args.epochs + 1):
        model.train()
        if args.data_init and not initialized:
            if not have_targets:
                raise SystemExit('--data-init needs output data (the test file): not available with --synthetic')
            batch = next(iter(test_loader))
            model(batch[1], batch[0])          # one y -> z pass initialises every ActNorm (train_cglow_reverse_kl.py:237-246)
            initialized = True
            say('Finished data initialization of Actnorm')
        torch.cuda.synchronize(device)
        t0 = time.time()
        if args.mode == 'fused':
            for batch_idx, (input,) in enumerate(train_loader):
                step = (epoch - 1) * len(train_loader) + batch_idx
                lr = scheduler.step(step / total_steps)
                trainer.step(input, lr)
            loss_train, residual, boundary, neg_entropy = parallel.mean_over_ranks(trainer.epoch_means())
        else:
            loss_train = 0.
            for batch_idx, (input,) in enumerate(train_loader):
                model.zero_grad()
                output, log_likelihood = model.generate(input)
                residual_norm = constitutive_constraint(input, output, sobel_filter) + continuity_constraint(output, sobel_filter)
                loss_dirichlet, loss_neumann = boundary_condition(output)
                loss_boundary = loss_dirichlet + loss_neumann
                loss_pde = residual_norm + loss_boundary * args.weight_bound
                neg_entropy = log_likelihood.mean() / math.log(2.) / n_out_pixels
                loss = loss_pde * args.beta + neg_entropy
                loss.backward()
                step = (epoch - 1) * len(train_loader) + batch_idx
                lr = scheduler.step(step / total_steps)
                adjust_learning_rate(optimizer, lr)
                optimizer.step()
                loss_train += loss.item()
            loss_train /= (batch_idx + 1)
            residual, boundary, neg_entropy = (float(residual_norm.detach()), float(loss_boundary.detach()),
                                               float(neg_entropy.detach()))
        torch.cuda.synchronize(device)
        train_seconds += time.time() - t0
        say(f'Epoch {epoch}: training loss: {loss_train:.6f}, residual: {residual:.6f}, boundary {boundary:.6f}, '
            f'neg entropy {neg_entropy:.6f}')
        if epoch % args.log_freq == 0:
            logger['loss_train'].append(loss_train)
            logger['entropy_train'].append(-neg_entropy)
        if epoch % args.ckpt_freq == 0 and is_main:
            # both loop bodies store what the reference stores: a torch.optim.Adam state_dict of model.parameters()
            opt_state = adam_state_dict(trainer) if args.mode == 'fused' else optimizer.state_dict()
            torch.save({'epoch': epoch, 'model_state_dict': model.state_dict(), 'optimizer_state_dict': opt_state,
                        'logger': logger}, args.ckpt_dir + f'/model_epoch{epoch}.pth')
            args.ckpt_epoch = epoch
            with open(args.run_dir + '/args.txt', 'w') as f:
                json.dump(vars(args), f, indent=4)
        with torch.no_grad():
            test(epoch)

    tic2 = time.time()
    say(f'Finished training {args.epochs} epochs with {args.ntrain} data using {(tic2 - tic) / 60:.2f} mins')
    if is_main:
        save_stats(args.train_dir, logger, 'loss_train', 'loss_test', 'nrmse_test', 'r2_test', 'entropy_test', 'entropy_train')
        args.training_time = tic2 - tic
        args.train_samples_per_sec = args.ntrain * (args.epochs - start_epoch + 1) / max(train_seconds, 1e-9)
        args.n_params, args.n_layers = model.model_size
        with open(args.run_dir + '/args.txt', 'w') as f:
            json.dump(vars(args), f, indent=4)
    if world > 1:
        if trainer is not None:
            trainer.close()                  # the direct RCCL communicator goes before its process group
        torch.distributed.destroy_process_group()
    return logger


if __name__ == '__main__':
    main()
