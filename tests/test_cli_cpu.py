"""CPU tests of the CLI surface: every reference flag/default of Parser, derived run
directories, args.txt, and the host-side schedule/metrics utilities against the golden vectors."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden

sys.path.insert(0, ROOT)

REFERENCE_DEFAULTS = {
    'exp_name': 'codec/mixed_residual', 'exp_dir': './experiments', 'blocks': [6, 8, 6], 'growth_rate': 16,
    'init_features': 48, 'drop_rate': 0.0, 'upsample': 'nearest', 'data_dir': './datasets', 'data': 'grf_kle512',
    'ntrain': 4096, 'ntest': 512, 'imsize': 64, 'run': 1, 'epochs': 300, 'lr': 1e-3, 'lr_div': 2.0, 'lr_pct': 0.3,
    'weight_decay': 0.0, 'weight_bound': 10, 'batch_size': 32, 'test_batch_size': 64, 'seed': 1, 'cuda': 1,
    'debug': False, 'ckpt_epoch': None, 'ckpt_freq': 100, 'log_freq': 1, 'plot_freq': 50, 'plot_fn': 'imshow',
}


def test_parser_flags_defaults_and_run_dir(tmp_path, capsys):
    import train_codec_mixed_residual as t
    args = t.Parser().parse(['--exp-dir', str(tmp_path)])
    for k, v in REFERENCE_DEFAULTS.items():
        if k == 'exp_dir':
            continue
        assert getattr(args, k) == v, k
    want = f'{tmp_path}/codec/mixed_residual/grf_kle512_ntrain4096_run1_bs32_lr0.001_epochs300'
    assert args.run_dir == want and args.ckpt_dir == want + '/checkpoints'
    assert os.path.isdir(args.ckpt_dir)
    saved = json.load(open(want + '/args.txt'))
    assert saved['ntrain'] == 4096 and saved['blocks'] == [6, 8, 6]
    a2 = t.Parser().parse(['--exp-dir', str(tmp_path), '--debug', '--data', 'channelized', '--ntrain', '512',
                           '--batch-size', '8', '--blocks', '343'])
    assert a2.run_dir.startswith(f'{tmp_path}/codec/mixed_residual/debug/channelized_ntrain512')
    assert a2.blocks == [3, 4, 3]                      # reference quirk: type=list splits characters
    with pytest.raises(SystemExit):
        t.Parser().parse(['--exp-dir', str(tmp_path), '--ntrain', '100', '--batch-size', '32'])
    with pytest.raises(SystemExit):
        t.Parser().parse(['--exp-dir', str(tmp_path), '--upsample', 'cubic'])


def test_parser_rejects_what_the_hip_path_cannot_run_before_creating_directories(tmp_path):
    """--imsize: any size DenseED maps back onto itself (multiples of 4 for the default blocks; 65 -> a 64 x 64 output,
    which the reference's loss cannot multiply with the 65 x 65 input either) -- the loss kernels take any square
    size; the global batch must divide the dataset under torchrun; only rank 0 writes"""
    import train_codec_mixed_residual as t
    with pytest.raises(SystemExit, match='imsize'):
        t.Parser().parse(['--exp-dir', str(tmp_path / 'a'), '--imsize', '65'])
    assert not (tmp_path / 'a').exists()
    for n in (48, 96, 128):
        assert t.Parser().parse(['--exp-dir', str(tmp_path / 'ok'), '--imsize', str(n)]).imsize == n
    with pytest.raises(SystemExit, match='imsize'):          # [3, 4, 3, 4, 3]: two encoding stages -> multiples of 8
        t.Parser().parse(['--exp-dir', str(tmp_path / 'a'), '--imsize', '20', '--blocks', '34343'])
    with pytest.raises(SystemExit, match='global batch'):
        t.Parser().parse(['--exp-dir', str(tmp_path / 'b'), '--ntrain', '4096', '--batch-size', '32'], rank=0, world=3)
    assert not (tmp_path / 'b').exists()
    args = t.Parser().parse(['--exp-dir', str(tmp_path / 'c'), '--ntrain', '8192', '--batch-size', '32'], rank=1, world=8)
    assert not os.path.exists(args.run_dir)                 # rank 1 creates nothing and writes no args.txt
    args = t.Parser().parse(['--exp-dir', str(tmp_path / 'c'), '--ntrain', '8192', '--batch-size', '32'], rank=0, world=8)
    assert os.path.exists(args.run_dir + '/args.txt')


def test_max_likelihood_parser_matches_reference_defaults(tmp_path):
    """train_codec_max_likelihood.py:25-56 of the reference: the same flags minus --weight-bound, its own defaults"""
    import train_codec_max_likelihood as m
    args = m.Parser().parse(['--exp-dir', str(tmp_path)])
    want = dict(REFERENCE_DEFAULTS, exp_name='codec/max_likelihood', epochs=200, ckpt_freq=50)
    want.pop('weight_bound')
    for k, v in want.items():
        if k != 'exp_dir':
            assert getattr(args, k) == v, k
    assert not hasattr(args, 'weight_bound')
    assert args.run_dir == f'{tmp_path}/codec/max_likelihood/grf_kle512_ntrain4096_run1_bs32_lr0.001_epochs200'
    with pytest.raises(SystemExit, match='targets'):
        m.Parser().parse(['--exp-dir', str(tmp_path), '--synthetic'])


def test_dataset_paths_match_reference_layout():
    import train_codec_mixed_residual as t
    import argparse
    a = argparse.Namespace(data='grf_kle512', data_dir='./datasets', imsize=64, ntrain=4096, ntest=512)
    assert t.dataset_files(a) == ('./datasets/64x64/kle512_lhs10000_train.hdf5', './datasets/64x64/kle512_lhs1000_val.hdf5')
    a.data = 'channelized'
    assert t.dataset_files(a) == ('./datasets/64x64/channel_ng64_n4096_train.hdf5', './datasets/64x64/channel_ng64_n512_test.hdf5')
    a.ntrain = 5000
    with pytest.raises(AssertionError):
        t.dataset_files(a)


def test_one_cycle_matches_reference_values():
    from pde_surrogate_amd.utils.practices import OneCycleScheduler
    g = golden('G8_one_cycle.npz')
    s = OneCycleScheduler(lr_max=1e-3, div_factor=2.0, pct_start=0.3)
    np.testing.assert_allclose([s.step(p) for p in g['pcts']], g['lr'], rtol=1e-12)
    s = OneCycleScheduler(lr_max=5e-4, div_factor=25.0, pct_start=0.3)
    np.testing.assert_allclose([s.step(p) for p in g['pcts']], g['lr25'], rtol=1e-12)


def test_y_variation_and_npz_reader(tmp_path):
    from pde_surrogate_amd.utils.load import load_data, y_variation
    g = golden('G9_metrics.npz')
    np.testing.assert_allclose(y_variation(g['target']), g['y_variation'], rtol=1e-6)
    f = str(tmp_path / 'd.npz')
    np.savez(f, input=np.ones((10, 1, 16, 16), np.float32), output=g['target'][:6].repeat(2, 0)[:10])
    loader, stats = load_data(f, 8, 4, only_input=False, return_stats=True)
    assert len(loader) == 2 and stats['y_variation'].shape == (3,)
    xb, yb = next(iter(loader))
    assert xb.shape == (4, 1, 16, 16) and yb.shape == (4, 3, 16, 16)


def test_save_stats_files(tmp_path):
    from pde_surrogate_amd.utils.plot import plot_prediction_det, save_stats
    logger = {'loss_train': [3.0, 2.0, 1.0], 'r2_test': [np.array([0.1, 0.2, 0.3]), np.array([0.2, 0.3, 0.4])]}
    save_stats(str(tmp_path), logger, 'loss_train', 'r2_test')
    for m in ('loss_train', 'r2_test'):
        assert os.path.exists(tmp_path / f'{m}.txt') and os.path.exists(tmp_path / f'{m}.pdf')
    np.testing.assert_allclose(np.loadtxt(tmp_path / 'loss_train.txt'), [3, 2, 1])
    g = golden('G9_metrics.npz')
    plot_prediction_det(str(tmp_path), g['target'][0], g['pred'][0], 7, 0, plot_fn='imshow')
    assert os.path.exists(tmp_path / 'pred_epoch7_0.png')


def test_synthetic_generators_shapes_and_determinism():
    from pde_surrogate_amd.utils.data import channelized_fields
    a, b = channelized_fields(3, 32, seed=1), channelized_fields(3, 32, seed=1)
    assert a.shape == (3, 1, 32, 32) and a.dtype == np.float32 and np.array_equal(a, b)
    assert set(np.unique(a)) == {1.0, 10.0}


def test_solver_flags_match_reference():
    import solve_conv_mixed_residual as s
    a = s.build_parser().parse_args([])
    ref = dict(exp_dir='./experiments/solver', nonlinear=False, data_dir='./datasets', data='grf', kle=512, imsize=64,
               idx=8, alpha1=1.0, alpha2=1.0, nz=1, blocks=[8, 6], weight_bound=10, lr=0.5, epochs=500, test_freq=50,
               ckpt_freq=250, cmap='jet', same_scale=False, animate=False, cuda=1, verbose=False)
    for k, v in ref.items():
        assert getattr(a, k) == v, k
    a.kle = 1024
    assert s.dataset_file(a) == './datasets/64x64/kle1024_lhs1024_test.hdf5'


def test_cglow_parser_contract(tmp_path):
    """train_cglow_reverse_kl.py: the reference's flags and defaults, its run-directory name, the list quirk, rejections"""
    import train_cglow_reverse_kl as cli
    a = cli.Parser().parse(['--exp-dir', str(tmp_path)])
    assert (a.enc_blocks, a.flow_blocks, a.kle, a.imsize, a.beta, a.weight_bound, a.lr, a.batch_size, a.epochs) == \
        ([3, 4, 4], [6, 6, 6], 100, 32, 150, 50, 1.5e-3, 32, 400)
    assert a.LU_decompose and not a.data_init
    assert a.run_dir.endswith('cglow/reverse_kld/kle100_ntrain4096_ENC_blocks[3, 4, 4]_FLOW_blocks[6, 6, 6]_wb50_beta150_'
                              'batch32_lr0.0015_epochs400')
    b = cli.Parser().parse(['--exp-dir', str(tmp_path), '--enc-blocks', '211', '--flow-blocks', '221', '--no-LU-decompose'])
    assert b.enc_blocks == [2, 1, 1] and b.flow_blocks == [2, 2, 1] and not b.LU_decompose
    for bad in (['--imsize', '50'], ['--enc-blocks', '34'], ['--ntrain', '20'], ['--imsize', '16', '--enc-blocks', '33333',
                                                                                  '--flow-blocks', '33333']):
        with pytest.raises(SystemExit):
            cli.Parser().parse(['--exp-dir', str(tmp_path)] + bad)
