"""The LDS swizzle of the bf16-split convolution kernels (csrc/conv_mfma_b3.hip, conv_mfma_b3_up.hip) against the bank model
of the MI355X guide: every A-fragment read must be conflict free for every tap offset (tools/lds_swizzle_check.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import lds_swizzle_check as chk


def _worst(sw):
    return max(chk.cycles(lambda l, c0=c0: ((c0 + (l & 15)) * 32 + 8 * sw(c0 + (l & 15), l >> 4)) * 2) for c0 in range(0, 35))


def test_linear_layout_has_two_way_conflicts():
    assert _worst(lambda col, o: o) == 8


def test_swizzled_layout_is_conflict_free():
    assert _worst(lambda col, o: o ^ (((col >> 2) & 1) << 1)) == 4


def test_kernels_use_the_checked_map():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in ('conv_mfma_b3.hip', 'conv_mfma_b3_up.hip'):
        src = open(os.path.join(root, 'pde_surrogate_amd', 'csrc', f)).read()
        assert 'oct ^ (((col >> 2) & 1) << 1)' in src, f
