#!/usr/bin/env python
"""Turn the output of tools/profile_round.sh (gpurun_out/<tag>/) into the tracked files under profiles/:
    python tools/summarize_profile.py gpurun_out/r02_b r02_b
copies the kernel statistics / timeline / loss-kernel launch tables and derives
  <prefix>_loss_kernel_pmc.json      HBM bytes per launch of the fused loss kernel (FETCH_SIZE doubled: gfx950 note)
  <prefix>_dense_dgrad_traffic.json  HBM bytes of the dense-block data-gradient kernels vs their algorithmic bytes
  <prefix>_mfma_utilisation.csv      SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs) per kernel"""
import json
import os
import re
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'profiles')
for name in ('kernel_stats.csv', 'step_timeline.txt', 'loss_kernel_by_batch.csv', 'loss_kernel_sustained.csv',
             'per_layer_conv_microbench.log', 'bench.json'):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(out, f'{prefix}_{name}'))


def blocks(path):
    """[(header, {kernel: {counter: (mean, median, n)}})] of a pmc_*.txt file"""
    res, cur, kern = [], None, None
    for line in open(path):
        if line.startswith('=='):
            cur = {}
            res.append((line.strip('= \n'), cur))
        elif line.startswith('    ') and kern is not None:
            m = re.match(r'\s+(\S+)\s+mean\s+([\d.]+)(?:\s+median\s+([\d.]+))?\s+n=(\d+)', line)
            if m:
                cur.setdefault(kern, {})[m.group(1)] = (float(m.group(2)), float(m.group(3) or m.group(2)), int(m.group(4)))
        elif line.strip() and not line.startswith('--'):
            kern = line.strip()
            if cur is None:
                cur = {}
                res.append(('', cur))
    return res


# ---- loss kernel
lp = os.path.join(src, 'pmc_loss.txt')
if os.path.exists(lp):
    vals = {}
    for hdr, ks in blocks(lp):
        for k, cs in ks.items():
            for c, (mean, med, n) in cs.items():
                vals[c] = mean
    rd, wr = vals['FETCH_SIZE'] * 1024 * 2, vals['WRITE_SIZE'] * 1024
    alg = 114688 * 16384
    json.dump({'kernel': 'darcy_loss_kernel<64,true,false,false>', 'batch': 16384,
               'method': 'rocprofv3 --pmc FETCH_SIZE --kernel-trace and rocprofv3 --pmc WRITE_SIZE --kernel-trace in separate '
                         'passes (python tools/bench_loss.py pmc); values are KiB per dispatch, mean of 10 dispatches; FETCH_SIZE '
                         'doubled per the gfx950 correction for 16-B/lane coalesced reads (MI355X_MICROARCH.md)',
               'FETCH_SIZE_KiB': vals['FETCH_SIZE'], 'WRITE_SIZE_KiB': vals['WRITE_SIZE'], 'hbm_read_bytes': rd,
               'hbm_write_bytes': wr, 'hbm_bytes_per_launch': rd + wr, 'algorithmic_bytes_per_launch': alg,
               'traffic_over_algorithmic': (rd + wr) / alg}, open(os.path.join(out, f'{prefix}_loss_kernel_pmc.json'), 'w'), indent=1)

# ---- dense data-gradient kernels: layer -> (Cin, Cout, H, accumulate?)   (B = 32; tools/bench_conv.py layer numbers)
LAYERS = {24: ('DecBlock2.denselayer6 180->16 3x3 @32x32', 180, 16, 32, True),
          16: ('DecBlock1.denselayer8 184->16 3x3 @16x16', 184, 16, 16, True),
          7: ('TransDown1.conv1 144->72 1x1 @32x32', 144, 72, 32, False),
          17: ('TransUp1.conv1 200->100 1x1 @16x16', 200, 100, 16, False)}
cp = os.path.join(src, 'pmc_conv.txt')
if os.path.exists(cp):
    per = {}
    for hdr, ks in blocks(cp):
        m = re.match(r'(\w+) layer (\d+)', hdr)
        if not m:
            continue
        c, layer = m.group(1), int(m.group(2))
        # the kernel of the layer under test = the one with the most dispatches (bench_conv.py repeats it; every other
        # kernel ran once in the warm-up pass); median per dispatch
        # (bench_conv.py also runs the layer's forward and weight-gradient kernels as often: keep the data-gradient ones --
        #  conv_mfma_kernel<KS, TWG, MT, S, WK, NTW, MODE = 1, ...> / conv1x1_mfma_kernel<..., MODE = 1>)
        isdg = lambda k: re.search(r'conv_mfma_kernel<\d+, \d+, \d+, \d+, \d+, \d+, 1,', k) or re.search(r'conv1x1_mfma_kernel<[\d, ]*, 1>', k)
        cand = [(cs[c][2], k, cs[c][1]) for k, cs in ks.items() if isdg(k) and c in cs]
        if not cand:
            continue
        n, k, med = max(cand)
        per.setdefault(layer, {})[c] = med
        per[layer]['kernel'] = k[:90]
    rows = []
    B = 32
    for layer, (name, cin, cout, h, acc) in LAYERS.items():
        if layer not in per or 'FETCH_SIZE' not in per[layer] or 'WRITE_SIZE' not in per[layer]:
            continue
        plane = B * h * h * 4
        rd, wr = per[layer]['FETCH_SIZE'] * 1024 * 2, per[layer]['WRITE_SIZE'] * 1024
        own_r = plane * (cout + cin + (cin if acc else 0))        # g, x (ReLU mask / xhat), T (read-modify-write)
        own_w = plane * cin                                       # T
        rows.append({'layer': name, 'kernel': per[layer]['kernel'], 'hbm_read_bytes': rd, 'hbm_write_bytes': wr,
                     'kernel_algorithmic_read_bytes': own_r, 'kernel_algorithmic_write_bytes': own_w,
                     'traffic_over_kernel_algorithmic': (rd + wr) / (own_r + own_w),
                     'write_once_lower_bound_bytes': plane * (cout + 2 * 16) if cout == 16 else None})
    json.dump({'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) over tools/bench_conv.py '
                         '<layer>, median per dispatch of the data-gradient kernel, FETCH_SIZE doubled (gfx950); B = 32',
               'note': 'kernel_algorithmic = what THIS kernel must move (g + x + T read, T written for every input channel); '
                       'write_once_lower_bound = a scheme that writes each 16-channel group of T once per block',
               'layers': rows}, open(os.path.join(out, f'{prefix}_dense_dgrad_traffic.json'), 'w'), indent=1)

# ---- matrix-pipe occupancy
mp = os.path.join(src, 'pmc_mfma.txt')
if os.path.exists(mp):
    with open(os.path.join(out, f'{prefix}_mfma_utilisation.csv'), 'w') as f:
        f.write('kernel,dispatches,mfma_busy_cycles,gui_active_cycles_all_xcds,mfma_busy_fraction\n')
        for hdr, ks in blocks(mp):
            for k, cs in ks.items():
                if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'GRBM_GUI_ACTIVE' in cs and 'pdes' in k:
                    busy, act = cs['SQ_VALU_MFMA_BUSY_CYCLES'][1], cs['GRBM_GUI_ACTIVE'][1]
                    if busy > 0:
                        f.write(f'"{k[:100]}",{cs["GRBM_GUI_ACTIVE"][2]},{busy:.0f},{act:.0f},{busy / (act / 8 * 1024):.4f}\n')
print(sorted(p for p in os.listdir(out) if p.startswith(prefix)))
