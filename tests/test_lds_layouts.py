"""The LDS stage layouts of the hand-written K loops, replayed in integers on the CPU (tools/lds_layout_check.py): every fragment
read finds the element the MFMA expects where the staging code put it, and no ds_read_b128 / ds_write_b128 of a wave has a
bank conflict under the gfx950 group rule.  A layout slip would otherwise only show as wrong numbers (or a slow kernel) on the GPU."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import lds_layout_check as chk  # noqa: E402


@pytest.mark.parametrize("shape", chk.X_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_x_image_stage_fill_equals_reads(shape):
    pieces, stage = chk.check_tile(*shape)
    assert stage == 3 * 2 * shape[3] * 32 * (shape[0] + shape[1] * shape[2]) and pieces * 1024 == stage


@pytest.mark.parametrize("shape", chk.TN_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_weight_gradient_stage_fill_equals_reads(shape):
    nb, kb, wn, wk = shape
    assert chk.check_tn_tile(*shape) == 3072 * (nb + kb) and 32 * (nb + kb) == 64 * wn * wk


@pytest.mark.parametrize("shape", chk.STRIP_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_strip_kernel_stage_fill_equals_reads(shape):
    nb, rg, nw = shape
    assert chk.check_strip_tile(*shape) == nw * rg * 2048 + 16 * nb * 192
