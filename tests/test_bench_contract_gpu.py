"""The driver reads ONE JSON line from `python bench.py --gpus N --steps K --warmup W`: the keys it needs, their types and
their mutual consistency, checked on the real command (a short run without the extra legs)."""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '16', '--warmup', '8',
                        '--no-extras', '--no-cpu-baseline'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                     # nothing but the result on stdout (RCCL / library banners go to stderr)
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 16 and d['warmup'] == 8
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['unit'] == 'samples/s' and d['dtype'] == 'f32' and 'synthetic' in d['data']
    assert 'samples/sec' in d['metric'] and 'bs=32' in d['metric']
    assert d['config']['workload'].startswith('configs[1]') and d['config']['global_batch'] == 32
    assert 'model' not in d['config']
    assert d['value'] > 1000 and d['ms_per_step'] > 0
    assert math.isclose(d['value'], 32 / (d['ms_per_step'] * 1e-3), rel_tol=2e-3)       # whole-job samples/s of the timed steps
    rf = d['roofline']
    assert rf['bound'] == 'hbm' and rf['unit'] == 'GB/s' and rf['peak'] == 8000.0
    assert math.isclose(rf['frac'], rf['achieved'] / rf['peak'], rel_tol=1e-3) and 0.3 < rf['frac'] < 1.0
    assert rf['traffic'] is None or rf['traffic'] >= 0.99 * rf['algorithmic_bytes_per_launch']
    assert math.isclose(rf['achieved'], rf['algorithmic_bytes_per_launch'] / rf['us_per_launch'] / 1e3, rel_tol=2e-2)
    assert math.isfinite(d['loss_mean_over_run'])
