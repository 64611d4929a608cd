"""GPU parity tests: the HIP path, called through the C ABI, against the oracle and the golden vectors.

Tolerances (fp32 path, model units are metres; x1000 = mm):
  * pointwise: 1e-5 on O(1) outputs after 16 blocks x T steps (measured: max 1.5e-6, mean 2.5e-7);
  * MPJPE aggregates: 1e-3 mm, and - the sharper statement - the HIP path is as close to an fp64 evaluation of
    the same function as the reference's own fp32 CPU arithmetic is (test_accuracy_equivalent_to_reference_fp32).
    The north star's 1e-4 mm sits below that floor: the reference's fp32 result itself is 2.1-2.6e-4 mm (mean
    abs, with a coherent per-part bias up to 1.2e-4 mm) away from exact arithmetic (tests/reports/error_budget.py), and
    wb_pose_from_parts / root centring turn single-joint rounding into whole-part shifts that do not average
    out in the MPJPE mean.  Measured |dMPJPE| is 1e-7 .. 3e-4 mm (tests/reports/parity_report.py).
"""
import pytest
import torch

from oracle import d3dp_oracle as orc
from tests.conftest import load_golden
from tests.golden import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda"

# |MPJPE_hip - MPJPE_oracle| in mm, per protocol AND per test case.  north_star asks 1e-4 mm; no protocol meets it at every
# step, and none can: measured |dMPJPE| is 2e-6 .. 3.7e-4 mm for J-Best / P-Best / P-Agg alike, i.e. the size of the mean
# pointwise difference (2.5e-4 mm) - the differences do not average out over the 3 618 joints of a clip because part
# centring and wb_pose_from_parts turn the rounding of ONE root / connection joint into a shift of a whole part, and both
# fp32 implementations (this one and the reference's ATen/MKL kernels) sit 2.1-2.6e-4 mm (mean) from an exact evaluation
# of the same function (profiles/r03_error_budget.json; test_accuracy_equivalent_to_reference_fp32 asserts the HIP path
# is not further from exact arithmetic than the reference is).
# What is asserted: for every named case below, each protocol's max-over-steps |dMPJPE| <= the bound FROZEN in
# tests/parity_bounds.json: for the fp32-grade modes 1.25 x the LARGER of the two values ('f32' and 'bf16x3') measured for
# that case and protocol on MI355X in round 3 (profiles/r03_parity_report.json) - for a bf16x3 case that is 1.3 - 2.0 x its own
# round-3 value; the opt-in f16x2 mode: 1.25 x its round-4 value -, never more than the old blanket 5e-4 mm.  The file is
# not regenerated and its `cases` are pinned by digest in tests/conftest.py: a new kernel set is held to the numbers the old ones
# were measured at (tests/reports/parity_report.py runs exactly the case functions of this module and REPORTS).  The per-case numbers are a
# property of both roundings; when the CPU oracle's arithmetic on this box is not the recorded one (its output hash
# differs: another BLAS / libm code path), only a host-spread bound is meaningful and only it is asserted
# (UNRECORDED_HOST_TOL_MM below; the terminal summary then says so).
# J-Agg additionally picks, per (frame, joint), the hypothesis with the smallest 2-D reprojection error: where two
# hypotheses tie to within rounding the two runs may pick differently, and one different pick moves the clip mean by
# (difference of the two hypotheses' 3-D errors) / 3618 (measured once in 20 steps x 3 618 joints: 1.1e-2 mm).  J-Agg is
# therefore compared on the joints where both runs make the same pick, and the picks that differ are bounded: few, and
# each a genuine near-tie (_j_agg_compare).
BLANKET_TOL_MM = 5e-4
# The CPU oracle's own fp32 result depends on the host it runs on (libm's sin / cos / exp in the timestep embedding, BLAS
# code paths): profiles/r03_host_variation.json measures up to 1.6e-3 mm between two x86 hosts on identical inputs.  On a host
# whose oracle output is not the recorded one (hash mismatch) the per-case numbers and the 5e-4 cap describe another
# arithmetic; what can be asserted there is that the HIP path is within that host-to-host spread of the oracle.
UNRECORDED_HOST_TOL_MM = 2e-3
PARITY_BOUNDS = "tests/parity_bounds.json"    # frozen in round 5 (1.25 x the round-3 measurements, floor 2e-5 mm, cap 5e-4 / 2e-3 mm)
PROTOCOLS = ("J-Best", "P-Best", "P-Agg", "J-Agg")
PARITY_LINES = []              # one line per asserted case, printed in the terminal summary (tests/conftest.py)


def tensor_sha256(t):
    import hashlib
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def parity_bounds(case, ref):
    """{protocol: bound in mm} for a named case (J-Agg: the same-pick comparison), and whether they are the per-case ones."""
    import json
    import os
    from tests.conftest import ROOT
    entries = json.load(open(os.path.join(ROOT, PARITY_BOUNDS)))["cases"]
    e = entries.get(case)
    if e is None:
        pytest.fail(f"{PARITY_BOUNDS} has no case '{case}'")
    if ref is not None and e["oracle_sha256"] != tensor_sha256(ref):
        import warnings
        warnings.warn(f"{case}: the CPU oracle's output on this host is not the recorded one (another BLAS / libm code path): the "
                      f"per-case MPJPE bounds do not apply; the fp64 criterion (assert_not_further_from_fp64) is asserted instead "
                      f"where the case is small enough, else only the host-spread bound {UNRECORDED_HOST_TOL_MM} mm")
        return {k: UNRECORDED_HOST_TOL_MM for k in PROTOCOLS}, False
    # (ref None: the case compares with a committed golden made on ANOTHER host - G19: the reference on the build container's
    # Xeon.  The reference's own fp32 result moves between x86 hosts by more than the HIP path differs from the oracle on
    # one host (profiles/r03_host_variation.json: up to 1.6e-3 mm at the steps t = 799 and t = 599): its frozen bounds are
    # 1.25 x the committed measurement of that very comparison under a 2e-3 mm cap, the others under the 5e-4 mm blanket)
    return {k: float(e["bound_mm"][k]) for k in PROTOCOLS}, True


def _seeded(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


# ------------------------------------------------------------------------------------------------ unit ops
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 1152, 384), (77, 384, 768), (300, 672, 224),
                                   (129, 448, 224), (64, 768, 256), (1, 512, 256), (50, 96, 64), (33, 32, 32)])
@pytest.mark.parametrize("act", [None, "gelu"])
def test_linear(M, N, K, act):
    from pafuse_amd import ops
    x, w, b = _seeded((M, K), 1), _seeded((N, K), 2, K ** -0.5), _seeded((N,), 3, 0.1)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())      # fp64 ground truth
    if act:
        ref = torch.nn.functional.gelu(ref)
    out = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act).cpu()
    # a k-ordered fp32 fma chain (what v_mfma_f32_32x32x2_f32 computes) stays within ~1e-7 * sum|a b| of fp64
    assert torch.allclose(out.double(), ref, rtol=0, atol=2.5e-7 * K ** 0.5 + 1e-6), (out - ref).abs().max()


def test_gelu_against_fp64():
    """The epilogue's GELU (one branch-free erf form, kernels.hpp::gelu_erf) against an fp64 evaluation of
    x * 0.5 * (1 + erf(x / sqrt(2))) (common/mixste.py:25,32), on x ~ N(0, 1.5) and a sweep of [-8, 8]: at least as close
    as torch's own CPU fp32 GELU (what the reference and the oracle run)."""
    from pafuse_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.cat([torch.randn(1 << 20, generator=g) * 1.5, torch.linspace(-8, 8, 1 << 16)]).reshape(-1, 64)
    eye, zero = torch.eye(64), torch.zeros(64)                     # x @ I + 0 is exact: the output is GELU(x) itself
    out = ops.linear(x.to(DEV), eye.to(DEV), zero.to(DEV), "gelu").cpu().double()
    truth = x.double() * 0.5 * (1 + torch.special.erf(x.double() / 2 ** 0.5))
    err, err_torch = (out - truth).abs(), (torch.nn.functional.gelu(x).double() - truth).abs()
    assert err.max() <= 6e-7 and err.mean() <= 3e-8, (err.max(), err.mean())
    assert err.max() <= err_torch.max() and err.mean() <= err_torch.mean(), (err.max(), err_torch.max(), err.mean(), err_torch.mean())


def test_linear_exact_integers():
    """A = I-like and asymmetric integer W: catches any row/col or k-permutation slip exactly."""
    from pafuse_amd import ops
    M, N, K = 96, 224, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float() + torch.arange(N)[:, None].float() % 5
    b = torch.arange(N).float()
    out = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV)).cpu()
    assert torch.equal(out, x @ w.t() + b)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 1152, 384), (77, 384, 768), (300, 672, 224),
                                   (129, 448, 224), (64, 768, 256), (1, 512, 256), (50, 96, 64), (33, 32, 32)])
@pytest.mark.parametrize("act", [None, "gelu"])
@pytest.mark.parametrize("layout", [0, 2])
@pytest.mark.parametrize("scheme", ["bf16x3", "bf16x3_images", "f16x2"])
def test_linear_split(M, N, K, act, layout, scheme):
    """split-precision products against fp64: the same bound as the fp32 FMA chain of test_linear, and closer to exact
    arithmetic than that chain on average.  bf16x3: the six kept terms carry every operand bit above 2^-24 relative,
    one rounding per 16-deep MFMA.  f16x2 (round 4): two fp16 slices of the activation, three of the power-of-two-scaled
    weight, three products; operands carry 22-23 bits, one rounding per MFMA.
    layout 0: the 32x32x16-MFMA tiles (mlp.fc1); layout 2: the 16x16x32-MFMA tiles of the qkv layers."""
    from pafuse_amd import ops
    from functools import partial
    if scheme == "f16x2" and (layout != 0 or not (N % 128 == 0 or N % 224 == 0)):
        pytest.skip("f16x2: one image geometry, column tiles of 128 or 224")
    if scheme == "bf16x3_images" and (layout != 0 or not (N % 128 == 0 or N % 224 == 0 or N % 96 == 0)):
        pytest.skip("bf16x3 on images: one image geometry, column tiles of 128, 224 or 96")
    ops = type("ops", (), {"linear": staticmethod(ops.linear),
                           "linear_split": staticmethod(partial(ops.linear_split, layout=layout, scheme=scheme))})
    x, w, b = _seeded((M, K), 1), _seeded((N, K), 2, K ** -0.5), _seeded((N,), 3, 0.1)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    if act:
        ref = torch.nn.functional.gelu(ref)
    out = ops.linear_split(x.to(DEV), w.to(DEV), b.to(DEV), act).cpu()
    assert torch.allclose(out.double(), ref, rtol=0, atol=2.5e-7 * K ** 0.5 + 1e-6), (out - ref).abs().max()
    if M >= 64 and act is None:        # and it is at least as close to exact arithmetic as the fp32 chain, on average
        plain = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV)).cpu()
        assert (out.double() - ref).abs().mean() <= 1.05 * (plain.double() - ref).abs().mean()


@pytest.mark.parametrize("layout,M,N,K", [(0, 96, 224, 64), (2, 96, 224, 64), (2, 200, 1152, 384), (2, 131, 672, 224),
                                          (2, 70, 768, 256), (2, 300, 96, 64), ("x", 96, 224, 64), ("x", 200, 1152, 384),
                                          ("x", 131, 672, 224), ("x", 70, 768, 256), ("x", 300, 96, 64), ("x", 513, 448, 224)])
def test_linear_split_exact_integers_and_slices(layout, M, N, K):
    """exact data: small integers (any row/col/k-permutation or sub-block rotation slip shows exactly), and operands
    that need all three bf16 slices (24-bit integers times powers of two: products exact in fp32); both image layouts /
    kernels (0: 32x32x16 tiles, 2: the qkv layers' 16x16x32 tiles at their three tile widths and a ragged M)."""
    from pafuse_amd import ops
    from functools import partial
    # layout 'x': the same products on the image pipeline (both operands as X images; one geometry, xgemm_kernel)
    lin = partial(ops.linear_split, layout=0, scheme="bf16x3_images") if layout == "x" else partial(ops.linear_split, layout=layout, scheme="bf16x3")
    ops = type("ops", (), {"linear_split": staticmethod(lin)})
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float() + torch.arange(N)[:, None].float() % 5
    b = torch.arange(N).float()
    assert torch.equal(ops.linear_split(x.to(DEV), w.to(DEV), b.to(DEV)).cpu(), x @ w.t() + b)
    # one non-zero per row of W: out[m, n] = x[m, k_n] * 2^e exactly, with 24-bit x (slices s0, s1, s2 all in play)
    x = (torch.randint(2 ** 23, 2 ** 24, (M, K), generator=g).float() * (torch.randint(0, 2, (M, K), generator=g) * 2 - 1))
    kn = torch.randint(0, K, (N,), generator=g)
    w = torch.zeros(N, K)
    w[torch.arange(N), kn] = 2.0 ** torch.randint(-3, 4, (N,), generator=g).float()
    out = ops.linear_split(x.to(DEV), w.to(DEV), torch.zeros(N, device=DEV)).cpu()
    assert torch.equal(out, x[:, kn] * w[torch.arange(N), kn])
    # and the transposed role: 24-bit weights against one-hot power-of-two activations
    w = torch.randint(2 ** 23, 2 ** 24, (N, K), generator=g).float()
    x = torch.zeros(M, K)
    km = torch.randint(0, K, (M,), generator=g)
    x[torch.arange(M), km] = 0.5
    out = ops.linear_split(x.to(DEV), w.to(DEV), torch.zeros(N, device=DEV)).cpu()
    assert torch.equal(out, (w[:, km] * 0.5).t())


@pytest.mark.parametrize("M,N,K,act", [(8997, 1152, 384, None), (25920, 768, 384, "gelu"), (20005, 672, 224, None), (9001, 864, 288, None),
                                       (5000, 512, 128, None), (73440, 448, 224, None), (127, 384, 128, "gelu"), (129, 96, 160, None)])
def test_strip_kernel_tile_streams(M, N, K, act):
    """The persistent strip kernel of the plain bf16x3 layers (csrc/sgemm.hpp, round 6) where a workgroup walks SEVERAL tiles: more
    tiles than the 2 x 256 workgroups of a launch (deferred stores of tile t inside tile t + 1's chunks, the A / W' cursors crossing
    tile boundaries), ragged last row tiles (dead rows in some waves, in all waves of a workgroup's last tile), all three tile widths
    (128, 112, 96 columns), the shortest K it takes (128 = four chunks: every chunk of a tile carries stores) and a K that is not a
    multiple of 64.  Exact data: every element must equal the integer result; with GELU: the fp64 value."""
    from pafuse_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float() + torch.arange(N)[:, None].float() % 5
    b = (torch.arange(N).float() % 17) - 8
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ref = xd.double() @ wd.double().t() + bd.double()
    if act is None:
        out = ops.linear_split(xd, wd, bd, None, layout=2)
        assert torch.equal(out.double(), ref), (out.double() - ref).abs().max()
        # 24-bit activations against one power of two per weight row: needs all three slices, still exact
        x = (torch.randint(2 ** 23, 2 ** 24, (M, K), generator=g).float() * (torch.randint(0, 2, (M, K), generator=g) * 2 - 1))
        kn = torch.randint(0, K, (N,), generator=g)
        w = torch.zeros(N, K)
        w[torch.arange(N), kn] = 2.0 ** torch.randint(-3, 4, (N,), generator=g).float()
        out = ops.linear_split(x.to(DEV), w.to(DEV), torch.zeros(N, device=DEV), None, layout=2).cpu()
        assert torch.equal(out, x[:, kn] * w[torch.arange(N), kn])
    else:
        xs, ws = xd * 0.05, wd * 0.05
        ref = torch.nn.functional.gelu(xs.double() @ ws.double().t() + bd.double() * 0.1)
        out = ops.linear_split(xs, ws, bd * 0.1, act, layout=2)
        assert torch.allclose(out.double(), ref, rtol=0, atol=2.5e-7 * K ** 0.5 + 2e-6), (out.double() - ref).abs().max()
    # twice the same launch: the same bits (no state in the ring or the cursors survives a launch)
    again = ops.linear_split(xd, wd, bd, act, layout=2)
    assert torch.equal(again, ops.linear_split(xd, wd, bd, act, layout=2))


@pytest.mark.parametrize("M,N,K", [(96, 224, 64), (200, 1152, 384), (131, 672, 224), (70, 768, 256), (300, 128, 64), (513, 448, 224)])
def test_linear_f16x2_exact_integers_and_slices(M, N, K):
    """the f16x2 scheme on exact data: small integers (row / col / k-permutation and sub-block rotation slips show exactly);
    22-bit operands that need both activation slices (hi and the 2^11-scaled lo) resp. both weight slices; activations so
    small that hi is an fp16 SUBNORMAL (the matrix cores must not flush them); weights spanning 2^-20 .. 1 inside one tensor
    (one power-of-two scale per tensor: the small ones live in fp16 subnormals of w1 / w2 and still come out exact)."""
    from pafuse_amd import ops
    from functools import partial
    lin = partial(ops.linear_split, scheme="f16x2")
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float() + torch.arange(N)[:, None].float() % 5
    b = torch.arange(N).float()
    assert torch.equal(lin(x.to(DEV), w.to(DEV), b.to(DEV)).cpu(), x @ w.t() + b)
    zero = torch.zeros(N, device=DEV)
    sign = torch.randint(0, 2, (M, K), generator=g) * 2 - 1
    x = torch.randint(2 ** 21, 2 ** 22, (M, K), generator=g).float() * sign * 2.0 ** -8          # 22 bits, |x| < 65504
    kn = torch.randint(0, K, (N,), generator=g)
    w = torch.zeros(N, K)
    w[torch.arange(N), kn] = 2.0 ** torch.randint(-3, 4, (N,), generator=g).float()
    assert torch.equal(lin(x.to(DEV), w.to(DEV), zero).cpu(), x[:, kn] * w[torch.arange(N), kn])
    w = torch.randint(2 ** 21, 2 ** 22, (N, K), generator=g).float()                             # 22-bit weights
    x = torch.zeros(M, K)
    km = torch.randint(0, K, (M,), generator=g)
    x[torch.arange(M), km] = 0.5
    assert torch.equal(lin(x.to(DEV), w.to(DEV), zero).cpu(), (w[:, km] * 0.5).t())
    x = torch.randint(1, 2 ** 10, (M, K), generator=g).float() * 2.0 ** -26                      # hi is an fp16 subnormal
    w = torch.zeros(N, K)
    w[torch.arange(N), kn] = 1.0
    assert torch.equal(lin(x.to(DEV), w.to(DEV), zero).cpu(), x[:, kn])
    w[torch.arange(N), kn] = 2.0 ** torch.randint(-20, 1, (N,), generator=g).float()             # one scale, 2^-20 .. 1
    x = torch.randint(2 ** 21, 2 ** 22, (M, K), generator=g).float() * 2.0 ** -22
    assert torch.equal(lin(x.to(DEV), w.to(DEV), zero).cpu(), x[:, kn] * w[torch.arange(N), kn])


def test_linear_f16x2_overflow_is_loud():
    """|activation| >= 65504 does not fit the fp16 hi slice: the affected outputs are inf / NaN, never a finite wrong number."""
    from pafuse_amd import ops
    x = torch.ones(64, 128)
    x[3, 5] = 1e5
    w, b = torch.eye(128), torch.zeros(128)
    out = ops.linear_split(x.to(DEV), w.to(DEV), b.to(DEV), scheme="f16x2").cpu()
    assert not torch.isfinite(out[3, 5]) and torch.isfinite(out[:3]).all() and torch.isfinite(out[4:]).all()


def test_x_images_are_the_fp32_tensor():
    """An X image (bf16x3 on the image pipeline: [rows][K/32][3 slices][32 bf16]) holds the fp32 tensor bit for bit: three
    bf16 slices are an exact split of any fp32 number (24-bit significands, both signs, 2^-60 .. 2^60, zeros), inf and NaN stay
    inf / NaN; and a producer's image output (pafuse_linear_x with out_x) is the image of exactly the fp32 rows it would have
    written - the split happens once, in the epilogue, and loses nothing."""
    from pafuse_amd import ops
    g = torch.Generator().manual_seed(9)
    R, K = 77, 224
    m = torch.randint(2 ** 23, 2 ** 24, (R, K), generator=g).float() * (torch.randint(0, 2, (R, K), generator=g) * 2 - 1)
    x = m * 2.0 ** torch.randint(-83, 37, (R, K), generator=g).float()
    x[0, :8] = 0.0
    img = ops.xsplit_rows(x.to(DEV))
    assert img.numel() == R * K * 6
    assert torch.equal(ops.xjoin_rows(img, R, K).cpu(), x)
    sl = img.view(torch.bfloat16).view(R, K // 32, 3, 32).cpu().float()
    assert torch.equal(sl[:, :, 0].reshape(R, K), x.bfloat16().float())                 # slice 0 is bf16(x), RNE
    bad = torch.ones(1, 32)
    bad[0, 3], bad[0, 9] = float("inf"), float("nan")
    back = ops.xjoin_rows(ops.xsplit_rows(bad.to(DEV)), 1, 32).cpu()
    assert not torch.isfinite(back[0, 3]) and torch.isnan(back[0, 9]) and torch.equal(back[0, :3], bad[0, :3])
    for M, N, Kk, act in ((200, 768, 384, "gelu"), (131, 448, 224, None), (70, 96, 64, None)):
        a, w, b = _seeded((M, Kk), 1), _seeded((N, Kk), 2, Kk ** -0.5), _seeded((N,), 3, 0.1)
        rows = ops.linear_split(a.to(DEV), w.to(DEV), b.to(DEV), act, scheme="bf16x3_images")
        image = ops.linear_split(a.to(DEV), w.to(DEV), b.to(DEV), act, scheme="bf16x3_images", out_image=True)
        assert torch.equal(ops.xjoin_rows(image, M, N), rows), (M, N, Kk)


@pytest.mark.parametrize("C,eps", [(384, 1e-6), (224, 1e-5), (256, 1e-6), (64, 1e-6)])
def test_layernorm(C, eps):
    from pafuse_amd import ops
    x, w, b = _seeded((37, C), 1, 2.0) + 0.5, 1 + _seeded((C,), 2, 0.1), _seeded((C,), 3, 0.1)
    truth = torch.nn.functional.layer_norm(x.double(), (C,), w.double(), b.double(), eps)      # fp64 ground truth
    ref32 = torch.nn.functional.layer_norm(x, (C,), w, b, eps)
    out = ops.layer_norm(x.to(DEV), w.to(DEV), b.to(DEV), eps).cpu()
    err, err_torch = (out.double() - truth).abs(), (ref32.double() - truth).abs()
    assert float(err.max()) <= 1.5e-6, err.max()
    assert float(err.mean()) <= 1.5 * float(err_torch.mean()) + 1e-9, (err.mean(), err_torch.mean())   # as close as torch's fp32 kernel


def _attn_ref(qkv, S, L, heads):
    """softmax(q k^T / sqrt(d)) v per (sequence, head), in the dtype of `qkv` (pass .double() for the ground truth)"""
    C = qkv.shape[-1] // 3
    d = C // heads
    q, k, v = qkv.view(S, L, 3, heads, d).permute(2, 0, 3, 1, 4)
    w = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(-1)
    return (w @ v).transpose(1, 2).reshape(S * L, C)


@pytest.mark.parametrize("L,C", [(24, 384), (27, 384), (68, 224), (27, 224), (42, 256), (27, 256), (5, 64), (80, 384)])
def test_attention_contiguous(L, C):
    from pafuse_amd import ops
    S, heads = 7, 8
    qkv = _seeded((S * L, 3 * C), 11)
    out = ops.attention(qkv.to(DEV), heads, S, L).cpu()
    truth = _attn_ref(qkv.double(), S, L, heads)                       # fp64 ground truth
    err, err_torch = (out.double() - truth).abs(), (_attn_ref(qkv, S, L, heads).double() - truth).abs()
    assert float(err.max()) <= 2e-6, err.max()
    assert float(err.mean()) <= 1.5 * float(err_torch.mean()) + 1e-9, (err.mean(), err_torch.mean())


def test_attention_temporal_strides():
    """temporal addressing: sequence (r, j) = rows (r*F + f)*J + j."""
    from pafuse_amd import ops
    R, F, J, C, heads = 3, 27, 24, 384, 8
    qkv = _seeded((R * F * J, 3 * C), 12)
    out = ops.attention(qkv.to(DEV), heads, R * J, F, group=J, group_stride=F * J, seq_stride=1, tok_stride=J).cpu()
    seqs = qkv.view(R, F, J, 3 * C).permute(0, 2, 1, 3).reshape(R * J * F, 3 * C)
    ref = _attn_ref(seqs.double(), R * J, F, heads).view(R, J, F, C).permute(0, 2, 1, 3).reshape(R * F * J, C)
    assert torch.allclose(out.double(), ref, rtol=0, atol=2e-6)


def test_g3_time_embed_golden():
    import pafuse_amd
    from pafuse_amd import ops
    z = load_golden("g3_time_mlp.npz")
    for part, C in gu.PART_WIDTH.items():
        m = pafuse_amd.MixSTE2(2, 2, 5, C, 1, 8, drop_path_rate=0.0, is_train=False)
        sd = {k: gu.seeded_tensor(f"g3.{part}.time_mlp.{k}", v.shape, 31) for k, v in m.time_mlp.state_dict().items()}
        assert gu.sha256_of(sd) == z[f"{part}.sha"].numpy().tobytes()
        m.time_mlp.load_state_dict(sd)
        out = ops.time_embed(m.to(DEV), z["t"].to(DEV)).cpu()
        assert torch.allclose(out, z[f"{part}.out"], rtol=0, atol=5e-6), (part, (out - z[f"{part}.out"]).abs().max())


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16x3_images", "f16x2"])
def test_g4_blocks_golden(precision):
    """golden G4 (one real-width block per part, spatial and temporal, outputs of the reference) through pafuse_block_forward in
    both fp32-grade product modes.  (G3, the timestep MLP, has no product mode: time_embed_kernel is fp32 VALU arithmetic.)"""
    from functools import partial
    from pafuse_amd import ops
    from pafuse_amd.mixste2 import _BlockParams
    z = load_golden("g4_blocks.npz")
    for part, C in gu.PART_WIDTH.items():
        blk = _BlockParams(C, 2.0, True, partial(torch.nn.LayerNorm, eps=1e-6))
        sd = gu.seeded_like(blk.state_dict(), seed=41, prefix=f"g4.{part}.")
        assert gu.sha256_of(sd) == z[f"{part}.sha"].numpy().tobytes()
        blk.load_state_dict(sd)
        blk = blk.to(DEV)
        for tag in ("s", "t"):
            y = ops.block_forward(blk, z[f"{part}.x{tag}"].to(DEV), precision=precision).cpu()
            ref = z[f"{part}.y{tag}"]
            assert torch.allclose(y, ref, rtol=0, atol=1e-5), (part, tag, (y - ref).abs().max())


# --------------------------------------------------------------------------------------- denoiser and loop
@pytest.fixture(scope="module")
def g5():
    from __graft_entry__ import make_model
    z = load_golden("g5_d3dp.npz")
    model, sd = make_model(2, 2, seed=51)
    assert gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    return z, model, sd


def test_g5_part_denoisers_golden(g5):
    z, model, sd = g5
    x2d, _ = gu.synthetic_inputs_2d(B=1)
    t = torch.tensor([499], device=DEV)
    for part, idx in model.parts_joint_indices.items():
        out = model.pose_estimator[part](x2d[..., idx, :].to(DEV), z["part_x3d"][..., idx, :].to(DEV), t).cpu()
        ref = z[f"part.{part}"]
        assert torch.allclose(out, ref, rtol=0, atol=1e-5), (part, (out - ref).abs().max())


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_g5_flip_loop_golden(g5, precision):
    """the reference's own output (golden G5: flip loop P=2, T=2), in both fp32-grade product modes"""
    z, model, sd = g5
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=2, n=2, seed=1)
    model.noise_fn = lambda k, shape, device: noises[k]
    before, model.precision = model.precision, precision
    try:
        out = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    finally:
        model.precision = before
    assert out.shape == (1, 2, 2, 27, 134, 3)
    assert torch.allclose(out, z["flip_out"], rtol=0, atol=1e-5), (out - z["flip_out"]).abs().max()


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x3_images", "f16x2"])
def test_split_images_follow_in_place_weight_updates(precision):
    """the pre-split weight images are a cache: an in-place change of a weight (optimizer step, load_state_dict) must
    remake them - the split-precision result after the change equals a freshly built model's, bit for bit."""
    from __graft_entry__ import make_model
    model, _ = make_model(1, 1, seed=51)
    model.precision = precision
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    n1 = gu.synthetic_noises(B=1, P=1, n=1, seed=2)
    model.noise_fn = lambda k, shape, device: n1[k]
    before = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    other, sd2 = make_model(1, 1, seed=52)
    model.load_state_dict(sd2)                                   # in place: same storages, new values
    after = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    other.precision = precision
    other.noise_fn = model.noise_fn
    assert not torch.equal(before, after)
    assert torch.equal(after, other(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)))


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x3_images", "f16x2"])
def test_folded_layernorm_is_the_same_function(precision):
    """Split-precision default: norm1 / norm2 are applied INSIDE the qkv / fc1 GEMMs (weight image of W (.) g, two vectors
    per layer, row statistics from the producing whole-row kernel; include/pafuse_hip.h pafuse_block_weights.qkv_ls).
    Against the same model with the fold off (the whole-row kernels write the normalised rows): the same function, i.e.
    rounding-level differences, both equally close to the oracle; and a changed LayerNorm weight remakes the folded
    images (they hold g)."""
    from __graft_entry__ import make_model
    model, sd = make_model(3, 2, seed=53)
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=3, n=2, seed=4)
    model.noise_fn = lambda k, shape, device: noises[k]
    model.precision = precision
    parts = list(model.denoisers().values())
    for m in parts:
        m.fold_layernorm = True
    folded = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    for m in parts:
        m.fold_layernorm = False
    plain = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    ref = orc.ddim_sample(sd, x2d, noises, 2, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    d_fold, d_plain = (folded - ref).abs(), (plain - ref).abs()
    assert float((folded - plain).abs().max()) <= 4e-6, float((folded - plain).abs().max())
    assert float(d_fold.max()) <= 1e-5 and float(d_plain.max()) <= 1e-5
    assert float(d_fold.mean()) <= 1.25 * float(d_plain.mean()) + 1e-8, (float(d_fold.mean()), float(d_plain.mean()))
    # in-place change of a LayerNorm weight: the folded images must follow
    for m in parts:
        m.fold_layernorm = True
    with torch.no_grad():
        parts[0].STEblocks[0].norm1.weight.mul_(1.5)
        parts[0].TTEblocks[1].norm2.bias.add_(0.25)
    sd2 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    changed = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    ref2 = orc.ddim_sample(sd2, x2d, noises, 2, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    assert float((changed - folded).abs().max()) > 1e-4
    assert float((changed - ref2).abs().max()) <= 1e-5, float((changed - ref2).abs().max())


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x3_images", "f16x2"])
@pytest.mark.parametrize("fold", [True, False])
def test_fused_qkv_attention_kernel_is_the_same_function(fold, precision):
    """The opt-in fused kernel (fqa_kernel: qkv projection + attention of one head per workgroup, q / k / v never written;
    MixSTE2.fuse_qkv_attention) against the default two kernels, with and without the folded LayerNorm: same products
    (six bf16 MFMA terms per pair), same attention arithmetic - rounding-level differences (the K sum rounds per 32 k in
    both), both within 1e-5 of the oracle.  Body / hands blocks and the temporal face blocks run fused; the spatial face
    blocks (68 tokens) have a fused form in the image pipelines only ('bf16x3' keeps the two kernels there) - B = 2 and
    P = 3 give ragged last tiles.  'bf16x3_images' on the image pipeline has ONLY the fused form (xfqa_kernel): it is held against
    the two kernels of 'bf16x3' - the same six products per pair."""
    from __graft_entry__ import make_model
    model, sd = make_model(3, 2, seed=57)
    model.precision = "bf16x3" if precision == "bf16x3_images" else precision
    x2d, x2f = gu.synthetic_inputs_2d(B=2)
    noises = gu.synthetic_noises(B=2, P=3, n=2, seed=6)
    model.noise_fn = lambda k, shape, device: noises[k]
    parts = list(model.denoisers().values())
    for m in parts:
        m.fold_layernorm = fold
        m.fuse_qkv_attention = False
    two = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    import ctypes
    from pafuse_amd import _lib
    count = lambda m: _lib.check(_lib.load().pafuse_mixste2_fused_blocks(ctypes.byref(m.weights_struct())))
    assert [count(m) for m in parts] == [0, 0, 0]
    for m in parts:
        m.fuse_qkv_attention = True
    model.precision = precision
    # body (24 joints) and hands (42): all 16 blocks; face: the 8 temporal blocks (27 frames) and - image pipelines only, whose
    # kernels have an 80-token form on five waves - the 8 spatial ones (68 joints: two sequences per 160-row tile)
    assert {name: count(m) for name, m in model.denoisers().items()} == {"body": 16, "face": 8 if precision == "bf16x3" else 16, "hands": 16}
    fused = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    ref = orc.ddim_sample(sd, x2d, noises, 2, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    assert float((fused - two).abs().max()) <= 4e-6, float((fused - two).abs().max())
    assert float((fused - ref).abs().max()) <= 1e-5 and float((two - ref).abs().max()) <= 1e-5
    assert float((fused - ref).abs().mean()) <= 1.25 * float((two - ref).abs().mean()) + 1e-8


def _qkv_attention_fp64(x, w, b, heads, seqs, rstd=None):
    """fp64 evaluation of what the fused kernel computes: seqs [nseq, L] = row indices of every sequence"""
    C_ = x.shape[1]
    d = C_ // heads
    y = x.double() @ w.double().t()
    if rstd is not None:
        y = y * rstd.double()[:, None]
    qkv = y + b.double()
    o = torch.zeros(x.shape[0], C_, dtype=torch.float64)
    q, k, v = (qkv[seqs][..., i * C_:(i + 1) * C_].reshape(seqs.shape[0], seqs.shape[1], heads, d).transpose(1, 2) for i in range(3))
    att = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(-1) @ v                   # [nseq, heads, L, d]
    o[seqs.reshape(-1)] = att.transpose(1, 2).reshape(-1, C_)
    return o


FQA_FORMS = [   # (name, L, C, temporal joints J or 0): every instantiation of hfqa_kernel / xfqa_kernel the loop launches
    ("body spatial <32,48>", 24, 384, 0), ("body temporal <32,48>", 27, 384, 24), ("face spatial <80,32>", 68, 224, 0),
    ("face temporal <32,32>", 27, 224, 68), ("hands spatial <48,32>", 42, 256, 0), ("hands temporal <32,32>", 27, 256, 42)]


@pytest.mark.parametrize("scheme", ["f16x2", "bf16x3_images"])
@pytest.mark.parametrize("form", FQA_FORMS, ids=[f[0] for f in FQA_FORMS])
def test_fused_qkv_attention_kernel_against_fp64(form, scheme):
    """hfqa_kernel ('f16x2') and xfqa_kernel ('bf16x3_images') on their own (pafuse_qkv_attention_image)
    against an fp64 evaluation - every form the loop launches (32-token tiles at head dim 48 and 32, the 48-token form, the
    80-token five-wave form; spatial and temporal addressing), sequence counts that end in ragged tiles, with and without the
    folded LayerNorm's row factor.  And exact data: q = k = 0 (uniform attention) with a selector v - the output is the
    sequence mean of one input channel per output channel, so a slip in the token gather, the head-major row order, the q | k | v
    tile layout or the image store shows as an O(1) error."""
    from pafuse_amd import ops
    name, L, C_, J = form
    heads, g = 8, torch.Generator().manual_seed(17 + L + C_)
    R = 3                                              # hypotheses-like outer rows: 3 x 27 frames x J (temporal) / 7 sequences + 2 (spatial)
    if J:
        M, nseq = R * 27 * J, R * J
        seqs = (torch.arange(R)[:, None, None] * 27 * J + torch.arange(J)[None, :, None] + torch.arange(27)[None, None, :] * J).reshape(nseq, 27)
        kw = dict(group=J, group_stride=27 * J, seq_stride=1, tok_stride=J)
    else:
        nseq = 23                                      # never a multiple of the sequences per tile (5, 1 or 2, 3)
        M = nseq * L
        seqs = torch.arange(M).reshape(nseq, L)
        kw = {}
    x = torch.randn(M, C_, generator=g)
    w = torch.randn(3 * C_, C_, generator=g) * C_ ** -0.5
    b = torch.randn(3 * C_, generator=g) * 0.1
    for rstd in (None, 0.5 + torch.rand(M, generator=g)):
        truth = _qkv_attention_fp64(x, w, b, heads, seqs, rstd)
        out = ops.qkv_attention_fused(x.to(DEV), w.to(DEV), b.to(DEV), heads, nseq, L, scheme,
                                      rstd=None if rstd is None else rstd.to(DEV), **kw).cpu().double()
        err = (out - truth).abs()
        # (logits reach |s| ~ 10 with the row factor: the largest errors are single sharp softmax rows; the mean is the yardstick)
        assert float(err.max()) <= 1.5e-5 and float(err.mean()) <= 4e-7, (name, scheme, rstd is not None, float(err.max()), float(err.mean()))
    xi = torch.randint(-8, 9, (M, C_), generator=g).float()
    sel = torch.randint(0, C_, (C_,), generator=g)
    wi = torch.zeros(3 * C_, C_)
    wi[2 * C_ + torch.arange(C_), sel] = 1.0           # v[:, c] = x[:, sel[c]]; q = k = 0: softmax = 1 / L everywhere
    out = ops.qkv_attention_fused(xi.to(DEV), wi.to(DEV), torch.zeros(3 * C_, device=DEV), heads, nseq, L, scheme, **kw).cpu().double()
    want = torch.zeros(M, C_, dtype=torch.float64)
    want[seqs.reshape(-1)] = xi.double()[seqs][:, :, sel].mean(1, keepdim=True).expand(-1, L, -1).reshape(-1, C_)
    assert float((out - want).abs().max()) <= 4e-6, (name, scheme, float((out - want).abs().max()))


@pytest.mark.parametrize("C_", [224, 256, 384])
def test_fused_mlp_kernel_against_fp64(C_):
    """hmlp_kernel on its own (pafuse_mlp_h): y = xc + GELU(rstd (xc W1^T) + b1) W2^T + b2 with the hidden activations in
    registers, the fc2 image in the kernel's column order - against an fp64 evaluation; M = 300 and 7344 end in ragged tiles;
    in place (the loop's form) and out of place.  The result comes back as the centred two-slice image of y: its own grid
    (2^-22 of the row's largest |value|) is inside the bound.  Small integers (every product and sum exact, GELU of +-64 k
    is the identity or zero) must come out exactly up to that grid."""
    from pafuse_amd import ops
    g = torch.Generator().manual_seed(3 + C_)
    for M in (128, 300, 7344):
        x = torch.randn(M, C_, generator=g)
        xc = x - x.mean(1, keepdim=True)
        rstd = (1.0 / torch.sqrt(xc.double().pow(2).mean(1) + 1e-6)).float()
        W1 = torch.randn(2 * C_, C_, generator=g) * C_ ** -0.5
        b1 = torch.randn(2 * C_, generator=g) * 0.1
        W2 = torch.randn(C_, 2 * C_, generator=g) * (2 * C_) ** -0.5
        b2 = torch.randn(C_, generator=g) * 0.1
        hid = torch.nn.functional.gelu(rstd.double()[:, None] * (xc.double() @ W1.double().t()) + b1.double())
        ref = xc.double() + hid @ W2.double().t() + b2.double()
        mean = ref.mean(1, keepdim=True)
        var = (ref - mean).pow(2).mean(1)
        for in_place in (False, True):
            y, st = ops.mlp_fused(xc.to(DEV), rstd.to(DEV), W1.to(DEV), b1.to(DEV), W2.to(DEV), b2.to(DEV), in_place=in_place)
            d = (y.cpu().double() - (ref - mean)).abs()
            assert float(d.max()) <= 6e-6 and float(d.mean()) <= 4e-7, (C_, M, in_place, float(d.max()), float(d.mean()))
            assert float((st[:, 0].cpu().double() - mean[:, 0]).abs().max()) <= 1e-6
            assert float((st[:, 1].cpu().double() * torch.sqrt(var + 1e-6) - 1).abs().max()) <= 1e-5
    M = 300
    xi = torch.randint(-2, 3, (M, C_), generator=g).float()
    W1 = torch.randint(-2, 3, (2 * C_, C_), generator=g).float()
    b1 = (torch.randint(0, 2, (2 * C_,), generator=g).float() * 2 - 1) * 4096.0   # GELU(v +- 4096) = v + 4096 or 0, exactly
    W2 = torch.randint(-1, 2, (C_, 2 * C_), generator=g).float() * 2.0 ** -12
    y, _ = ops.mlp_fused(xi.to(DEV), None, W1.to(DEV), b1.to(DEV), W2.to(DEV), torch.zeros(C_, device=DEV))
    ref = xi.double() + torch.nn.functional.gelu(xi.double() @ W1.double().t() + b1.double()) @ W2.double().t()
    ref = ref - ref.mean(1, keepdim=True)
    assert float((y.cpu().double() - ref).abs().max()) <= 2.0 ** -20 * float(ref.abs().max())


def test_fused_mlp_kernel_is_the_same_function():
    """The MLP of a block as ONE kernel (hmlp_kernel, MixSTE2.fuse_mlp; default on for the face and the hands) against the fc1
    and fc2 launches it replaces, in the loop: same products, the same K order between 16-deep steps (inside a step the fc2 sum
    takes the hidden units in the accumulators' order) - rounding-level differences, both within 1e-5 of the oracle and equally
    far from it.  B = 2, P = 3: ragged last tiles; every part fused, the body too (opt-in there)."""
    from __graft_entry__ import make_model
    model, sd = make_model(3, 2, seed=57)
    model.precision = "f16x2"
    x2d, x2f = gu.synthetic_inputs_2d(B=2)
    noises = gu.synthetic_noises(B=2, P=3, n=2, seed=6)
    model.noise_fn = lambda k, shape, device: noises[k]
    parts = model.denoisers()
    assert {n: bool(m.weights_struct().ste[0].fc2_hp) for n, m in parts.items()} == {"body": False, "face": True, "hands": True}
    outs = {}
    for tag, flag in (("two", False), ("default", None), ("all", True)):
        for m in parts.values():
            m.fuse_mlp = flag
        outs[tag] = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    assert all(bool(m.weights_struct().tte[7].fc2_hp) for m in parts.values())
    ref = orc.ddim_sample(sd, x2d, noises, 2, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    for tag in ("default", "all"):
        assert float((outs[tag] - outs["two"]).abs().max()) <= 4e-6, (tag, float((outs[tag] - outs["two"]).abs().max()))
        assert float((outs[tag] - ref).abs().max()) <= 1e-5
        assert float((outs[tag] - ref).abs().mean()) <= 1.25 * float((outs["two"] - ref).abs().mean()) + 1e-8
    for m in parts.values():      # with the fp32 residual rows kept, or without the fold, the block runs the two launches
        m.fuse_mlp, m.keep_f32_residual = True, True
    assert not any(bool(m.weights_struct().ste[0].fc2_hp) for m in parts.values())


@pytest.mark.parametrize("precision", ["f16x2", "bf16x3_images"])
def test_f32_residual_rows_option_is_the_same_function(precision):
    """MixSTE2.keep_f32_residual (a public attribute, a C-ABI field, bench.py --f32-residual) - the image pipelines
    with the fp32 rows of the residual stream kept beside its image (the whole-row kernels then read fp32 residuals and write
    both forms; the fused MLP is off) against their default, the image-only stream: the same function, rounding-level
    differences, both within 1e-5 of the oracle."""
    from __graft_entry__ import make_model
    model, sd = make_model(3, 2, seed=57)
    model.precision = precision
    x2d, x2f = gu.synthetic_inputs_2d(B=2)
    noises = gu.synthetic_noises(B=2, P=3, n=2, seed=6)
    model.noise_fn = lambda k, shape, device: noises[k]
    default = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    for m in model.denoisers().values():
        m.keep_f32_residual = True
    kept = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    ref = orc.ddim_sample(sd, x2d, noises, 2, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    assert not torch.equal(kept, default)
    assert float((kept - default).abs().max()) <= 4e-6, float((kept - default).abs().max())
    assert float((kept - ref).abs().max()) <= 1e-5 and float((default - ref).abs().max()) <= 1e-5


def test_g5_p1t1_both_samplers_golden(g5):
    from __graft_entry__ import make_model
    z, _, sd = g5
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    n1 = gu.synthetic_noises(B=1, P=1, n=1, seed=2)
    for flip, key in ((False, "noflip_out"), (True, "flip11_out")):
        m, _ = make_model(1, 1, seed=51, flip=flip)
        m.noise_fn = lambda k, shape, device: n1[k]
        out = m(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV) if flip else None).cpu()
        assert torch.allclose(out, z[key], rtol=0, atol=1e-5), (key, (out - z[key]).abs().max())


def _mpjpe_report(pred, target, x2d):
    """J-Best / P-Best / P-Agg / J-Agg in mm on whole-body poses (reference main_h3wb.py:327-348)."""
    # fp64 metric arithmetic: at the synthetic targets' ~1 m errors one fp32 ulp of the metric is already 1.2e-4 mm
    pred = orc.wb_pose_from_parts(pred.double())
    target = orc.wb_pose_from_parts(target.double())
    x2d = x2d.double()
    B, T, P, F = pred.shape[:4]
    cam = torch.tensor([[2.29, 2.287, 0.025, 0.029, -0.207, 0.247, -0.003, -0.0009, -0.001]], dtype=torch.float64)
    traj = torch.tensor([0.0, 0.0, 4.0], dtype=torch.float64)
    reproj = orc.project_to_2d((pred + traj).reshape(-1, 134, 3), cam.repeat(B * T * P * F, 1)).reshape(B, T, P, F, 134, 2)
    return {"J-Best": orc.j_best(pred, target) * 1000, "P-Best": orc.p_best(pred, target) * 1000,
            "P-Agg": orc.p_agg(pred, target) * 1000, "J-Agg": orc.j_agg(pred, target, reproj, x2d) * 1000}


def _j_agg_parts(pred, target, x2d):
    """per (b, t, f, j): the J-Agg pick (argmin over hypotheses of the 2-D reprojection error), the picked 3-D error
    and the margin between the best and second-best 2-D error, in fp64 (main_h3wb.py:344-348, common/loss.py:150-168)."""
    pred = orc.wb_pose_from_parts(pred.double())
    target = orc.wb_pose_from_parts(target.double())
    B, T, P, F = pred.shape[:4]
    cam = torch.tensor([[2.29, 2.287, 0.025, 0.029, -0.207, 0.247, -0.003, -0.0009, -0.001]], dtype=torch.float64)
    traj = torch.tensor([0.0, 0.0, 4.0], dtype=torch.float64)
    reproj = orc.project_to_2d((pred + traj).reshape(-1, 134, 3), cam.repeat(B * T * P * F, 1)).reshape(B, T, P, F, 134, 2)
    e2 = (reproj - x2d.double()[:, None, None]).norm(dim=-1)                       # [B,T,P,F,J]
    e3 = (pred - target[:, None, None]).norm(dim=-1)
    pick = e2.argmin(dim=2)
    top2 = e2.topk(min(2, P), dim=2, largest=False).values
    margin = (top2[:, :, -1] - top2[:, :, 0]) if P > 1 else torch.full_like(top2[:, :, 0], float("inf"))
    return pick, e3.gather(2, pick[:, :, None]).squeeze(2), margin


def _j_agg_compare(out, ref, target, x2d):
    """J-Agg of the two runs: (max over steps of |dJ-Agg| in mm restricted to the joints where both runs pick the same
    hypothesis, fraction of joints with different picks, largest 2-D margin (m) among those)."""
    pa, ea, ma = _j_agg_parts(out, target, x2d)
    pb, eb, mb = _j_agg_parts(ref, target, x2d)
    same = pa == pb
    n = same.sum(dim=(0, 2, 3)).clamp(min=1)
    d = (((ea - eb) * same).sum(dim=(0, 2, 3)) / n).abs().max().item() * 1000
    flipped = ~same
    worst = torch.maximum(ma, mb)[flipped].max().item() if bool(flipped.any()) else 0.0
    return d, flipped.double().mean().item(), worst


def _assert_mpjpe_parity(out, ref, target, x2d, case, truth_fn=None):
    """truth_fn: () -> the fp64 evaluation of the same loop; asserted INSTEAD of the recorded per-case bounds when the oracle's
    arithmetic on this host is not the recorded one (the reference-independent criterion, assert_not_further_from_fp64)."""
    got, want = _mpjpe_report(out, target, x2d), _mpjpe_report(ref, target, x2d)
    bounds, per_case = parity_bounds(case, ref)
    if not per_case and truth_fn is not None:
        assert_not_further_from_fp64(case, out, ref, truth_fn(), target, x2d)
    diffs = {k: (got[k] - want[k]).abs() for k in ("J-Best", "P-Best", "P-Agg")}
    d, frac, worst = _j_agg_compare(out, ref, target, x2d)
    met = sum(int((v <= 1e-4).sum()) for v in diffs.values())
    total = sum(v.numel() for v in diffs.values())
    PARITY_LINES.append(f"{case}: |dMPJPE| max mm " + ", ".join(f"{k} {float(v.max()):.2e} (<= {bounds[k]:.2e})" for k, v in diffs.items()) +
                        f", J-Agg same picks {d:.2e} (<= {bounds['J-Agg']:.2e}), different picks {frac:.1e} of joints; north_star "
                        f"1e-4 mm met at {met} of {total} (step, protocol) pairs; bounds: "
                        + ("1.25 x committed measurement" if per_case else "host-spread bound 2e-3 mm (the oracle's arithmetic on this host is not the recorded one)"))
    for k, v in diffs.items():
        assert v.max() <= bounds[k], (case, k, float(v.max()), bounds[k])
    # same picks: the case's bound; different picks: rare, and only where the two best hypotheses tie to within the
    # pointwise tolerance of the poses (a 1e-5 m pose difference moves a 2-D reprojection error by < 1e-4)
    assert d <= bounds["J-Agg"] and frac <= 2e-3 and worst <= 1e-4, (case, d, bounds["J-Agg"], frac, worst)


def loop_case(B, P, T, precision):
    """(name, out, ref, target, x2d) of a seeded flip-TTA loop: the HIP path and the oracle on the same inputs.  Shared by
    test_loop_vs_oracle_mpjpe and tests/reports/parity_report.py (the committed bounds are measured on these very runs)."""
    from __graft_entry__ import make_model
    model, sd = make_model(P, T, seed=77)
    model.precision = precision
    x2d, x2f = gu.synthetic_inputs_2d(B=B, seed=1234)
    noises = gu.synthetic_noises(B=B, P=P, n=T, seed=3)
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    if (B, P, T) not in _LOOP_ORACLE:       # the oracle does not depend on the HIP path's product mode
        _LOOP_ORACLE[(B, P, T)] = orc.ddim_sample(sd, x2d, noises, T, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    ref = _LOOP_ORACLE[(B, P, T)]
    return f"loop_B{B}_P{P}_T{T}_{precision}", out, ref, orc.center_pose_parts(gu.synthetic_target_3d(B)), x2d


LOOP_CASES = [(1, 5, 5), (2, 3, 2)]
_LOOP_ORACLE = {}


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16x3_images", "f16x2"])
@pytest.mark.parametrize("B,P,T", LOOP_CASES)
def test_loop_vs_oracle_mpjpe(B, P, T, precision):
    """BASELINE configs[1] shape (P=5, T=5) and a B>1 case: pointwise and MPJPE parity, for the fp32 matrix cores and
    for the split-precision (bf16x3) products alike, each against its own committed measurement."""
    if precision == "bf16x3_images" and (B, P, T) != (2, 3, 2):
        pytest.skip("the opt-in image pipeline keeps ONE loop-level case (round 6: it is slower than the default and no wider); its unit tests stay")
    case, out, ref, target, x2d = loop_case(B, P, T, precision)
    assert torch.allclose(out, ref, rtol=0, atol=1e-5), (out - ref).abs().max()
    _assert_mpjpe_parity(out, ref, target, x2d, case, truth_fn=lambda: loop_truth(B, P, T))


FP64_FACTOR = 1.1           # the HIP path may be this much further from exact arithmetic than the reference's own fp32 is
FP64_FLOOR_MM = 2e-5
_LOOP_TRUTH = {}


def fp64_truth(sd, x2d, x2f, noises, T):
    """The oracle evaluated in fp64 on the same fp32 weights, inputs and noise: the host-independent yardstick (its own
    rounding is 1e-16; libm differences between hosts vanish at that level)."""
    sd64 = {k: v.double() for k, v in sd.items()}
    return orc.ddim_sample(sd64, x2d.double(), [n.double() for n in noises], T, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT,
                           inputs_2d_flip=x2f.double())


def loop_truth(B, P, T):
    """fp64 evaluation of loop_case(B, P, T)'s loop (cached)"""
    if (B, P, T) not in _LOOP_TRUTH:
        from __graft_entry__ import make_model
        _, sd = make_model(P, T, seed=77, device="cpu")
        x2d, x2f = gu.synthetic_inputs_2d(B=B, seed=1234)
        _LOOP_TRUTH[(B, P, T)] = fp64_truth(sd, x2d, x2f, gu.synthetic_noises(B=B, P=P, n=T, seed=3), T)
    return _LOOP_TRUTH[(B, P, T)]


FP64_MAX_FACTOR = 1.5       # ... and on the largest single step (the maximum of T noisy values: a looser guard)


def assert_not_further_from_fp64(case, out, ref32, truth, target, x2d):
    """The reference-independent tolerance: through the whole DDIM loop, on every protocol, the HIP
    path is at most FP64_FACTOR x as far from the exact (fp64) evaluation as the reference's fp32 arithmetic (the oracle) is,
    + a 2e-5 mm floor.  Asserted on the MEAN over the steps of |MPJPE - MPJPE_fp64| per protocol and on the pointwise mean
    distance of the predictions; a single step's deviation of either run is one draw of rounding noise (it can be near zero
    by chance, and the larger of ten draws of two equally accurate runs differs by tens of percent), so the per-step maximum
    is held to the looser FP64_MAX_FACTOR.  Every protocol's columns go to the terminal summary next to the north star's 1e-4 mm."""
    got, o32, t64 = (_mpjpe_report(v, target, x2d) for v in (out, ref32, truth.float()))
    lines = []
    for k in ("J-Best", "P-Best", "P-Agg"):
        dh, do = (got[k] - t64[k]).abs(), (o32[k] - t64[k]).abs()
        lines.append(f"{k} hip mean {float(dh.mean()):.2e} max {float(dh.max()):.2e} / oracle32 mean {float(do.mean()):.2e} max {float(do.max()):.2e}")
        assert float(dh.mean()) <= FP64_FACTOR * float(do.mean()) + FP64_FLOOR_MM, (case, k, dh.tolist(), do.tolist())
        assert float(dh.max()) <= FP64_MAX_FACTOR * float(do.max()) + FP64_FLOOR_MM, (case, k, dh.tolist(), do.tolist())
    dj_h, fr_h, _ = _j_agg_compare(out, truth.float(), target, x2d)
    dj_o, fr_o, _ = _j_agg_compare(ref32, truth.float(), target, x2d)
    lines.append(f"J-Agg (same picks, max) hip {dj_h:.2e} / oracle32 {dj_o:.2e}")
    assert dj_h <= FP64_MAX_FACTOR * dj_o + FP64_FLOOR_MM and fr_h <= max(2e-3, 2 * fr_o), (case, dj_h, dj_o, fr_h, fr_o)
    pw_h, pw_o = (out.double() - truth).abs().mean().item(), (ref32.double() - truth).abs().mean().item()
    # (the 'f32' mode's k-ordered FMA chain - what v_mfma_f32_32x32x2_f32 computes - accumulates more rounding than the
    # blocked sums of the reference's CPU BLAS: measured 1.16 x pointwise; the split-precision modes, one rounding per 16 / 32
    # k, are CLOSER to exact arithmetic than the reference and are held to FP64_FACTOR)
    assert pw_h <= (1.25 if case.endswith("_f32") else FP64_FACTOR) * pw_o, (case, pw_h, pw_o)
    PARITY_LINES.append(f"{case} vs fp64 truth, |MPJPE - MPJPE_fp64| mm over the steps: " + "; ".join(lines) +
                        f"; pointwise mean |d| m: hip {pw_h:.2e} / oracle32 {pw_o:.2e}")


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_loop_vs_fp64_truth(precision):
    """BASELINE configs[1]'s shape (P=5, T=5, flip-TTA) through all five steps against an fp64 evaluation of the same
    loop: the tolerance that does not depend on which host ran the fp32 reference (profiles/r03_host_variation.json)."""
    B, P, T = 1, 5, 5
    case, out, ref32, target, x2d = loop_case(B, P, T, precision)
    assert_not_further_from_fp64(case, out, ref32, loop_truth(B, P, T), target, x2d)


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_accuracy_equivalent_to_reference_fp32(precision):
    """Against an fp64 evaluation of the same function (same fp32 weights and inputs), the HIP denoiser is at most
    1.5x as far as the reference's fp32 CPU arithmetic (the oracle, bit-identical to the reference here) - in both
    fp32-grade product modes."""
    from __graft_entry__ import make_model
    model, sd = make_model(2, 2, seed=77)
    model.precision = precision
    sd64 = {k: v.double() for k, v in sd.items()}
    x2d, _ = gu.synthetic_inputs_2d(B=1)
    x3d = _seeded((1, 2, 27, 134, 3), 52).clamp(-1.1, 1.1)
    t = torch.tensor([499])
    for part, idx in orc.PART_JOINTS.items():
        pre = f"pose_estimator.{part}."
        truth = orc.mixste2_eval(sd64, pre, x2d[..., idx, :].double(), x3d[..., idx, :].double(), t)
        ref32 = orc.mixste2_eval(sd, pre, x2d[..., idx, :], x3d[..., idx, :], t)
        hip = model.pose_estimator[part](x2d[..., idx, :].to(DEV), x3d[..., idx, :].to(DEV), t.to(DEV)).cpu()
        e_ref, e_hip = (ref32.double() - truth).abs(), (hip.double() - truth).abs()
        assert e_hip.mean() <= 1.5 * e_ref.mean(), (part, e_hip.mean(), e_ref.mean())
        assert e_hip.max() <= 2.0 * e_ref.max(), (part, e_hip.max(), e_ref.max())


def _trained_like(sd, gain=300.0, mean=50.0):
    """seeded weights pushed towards what trained checkpoints show: one outlier channel per part (the post-norm gains of a
    channel x 300) and residual rows far from zero mean (post-norm biases + 50: what every block hands to the next)"""
    sd = {k: v.clone() for k, v in sd.items()}
    for part in ("body", "face", "hands"):
        for norm in ("Spatial_norm", "Temporal_norm"):
            sd[f"pose_estimator.{part}.{norm}.bias"] += mean
            sd[f"pose_estimator.{part}.{norm}.weight"][7] *= gain
    return sd


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_loop_vs_fp64_truth_on_trained_like_weights(precision):
    """the loop on weights with an outlier channel (x 300) and residual rows at mean 50 - the regime where a
    LayerNorm folded into its GEMM without centring cancels large numbers and where a 22-bit residual stream would round at the
    magnitude of the mean.  Every mode stores the stream centred (or, 'f32', normalises before the GEMM): against the fp64
    evaluation of the same loop the HIP path stays within 2 x the reference's own fp32 arithmetic, pointwise (measured on MI355X:
    1.5 x in the split modes - with one channel 300 x the others every partial sum of a GEMM rides on that one product, and one
    accumulator per output rounds there at every 16-deep step where the host BLAS spreads the sum over SIMD lanes -, 1.65 x in
    f16x2, 1.3 x in f32; 1.07 x on ordinary weights) and per protocol."""
    from __graft_entry__ import make_model
    B, P, T = 1, 2, 2
    model, sd0 = make_model(P, T, seed=58)
    sd = _trained_like(sd0)
    model.load_state_dict(sd)
    model.precision = precision
    x2d, x2f = gu.synthetic_inputs_2d(B=B, seed=1234)
    noises = gu.synthetic_noises(B=B, P=P, n=T, seed=8)
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    assert bool(torch.isfinite(out).all())
    ref32 = orc.ddim_sample(sd, x2d, noises, T, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    truth = fp64_truth(sd, x2d, x2f, noises, T)
    e_hip, e_ref = (out.double() - truth).abs(), (ref32.double() - truth).abs()
    target = orc.center_pose_parts(gu.synthetic_target_3d(B))
    got, o32, t64 = (_mpjpe_report(v, target, x2d) for v in (out, ref32, truth.float()))
    cols = {k: ((got[k] - t64[k]).abs(), (o32[k] - t64[k]).abs()) for k in ("J-Best", "P-Best", "P-Agg")}
    PARITY_LINES.append(f"trained_like_B{B}_P{P}_T{T}_{precision} vs fp64 truth: pointwise mean |d| m hip {float(e_hip.mean()):.2e} / oracle32 "
                        f"{float(e_ref.mean()):.2e}, max {float(e_hip.max()):.2e} / {float(e_ref.max()):.2e}; |MPJPE - MPJPE_fp64| mm mean over steps: " +
                        "; ".join(f"{k} hip {float(dh.mean()):.2e} / oracle32 {float(do.mean()):.2e}" for k, (dh, do) in cols.items()))
    assert float(e_hip.mean()) <= 2.0 * float(e_ref.mean()), (precision, float(e_hip.mean()), float(e_ref.mean()))
    assert float(e_hip.max()) <= 3.0 * float(e_ref.max()), (precision, float(e_hip.max()), float(e_ref.max()))
    for k, (dh, do) in cols.items():
        assert float(dh.mean()) <= 2.0 * float(do.mean()) + FP64_FLOOR_MM, (precision, k, dh.tolist(), do.tolist())


def test_failure_is_loud_through_the_loop():
    """VERDICT r4 item 2.  (i) 'f16x2': a LayerNorm gain scaled until a row of the residual stream leaves the fp16 range
    (|a| >= 65504) - the loop must not return a plausible pose: the predictions of the affected rows are NaN, the output stage
    flags them and D3DP raises (PAFUSE_E_RANGE); the same weights run finite in the exact-width modes and match the oracle.
    (ii) a NaN stays a NaN: the clamps of the loop are torch.clamp's (common/diffusionpose.py:193,216-217), so with a NaN in
    one hypothesis' noise and in one clip's 2-D input the 'f32' output has NaNs exactly where the oracle's has and equals it
    elsewhere (hypotheses and clips are independent)."""
    from __graft_entry__ import make_model
    from pafuse_amd._lib import PafuseError
    B, P, T = 2, 2, 2
    model, sd0 = make_model(P, T, seed=59)
    x2d, x2f = gu.synthetic_inputs_2d(B=B, seed=1234)
    noises = gu.synthetic_noises(B=B, P=P, n=T, seed=9)
    model.noise_fn = lambda k, shape, device: noises[k]
    sd = {k: v.clone() for k, v in sd0.items()}
    sd["pose_estimator.hands.Spatial_norm.weight"] *= 3e5          # rows of the hands' residual stream at |x| ~ 1e5 .. 1e6
    model.load_state_dict(sd)
    model.precision = "f16x2"
    with pytest.raises(PafuseError, match="fp16 range"):
        model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    ref = orc.ddim_sample(sd, x2d, noises, T, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    for precision in ("bf16x3_images", "bf16x3", "f32"):               # no range limit beyond fp32's: finite, and the oracle's numbers
        model.precision = precision
        out = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
        assert bool(torch.isfinite(out).all()) and torch.allclose(out, ref, rtol=0, atol=2e-4), (precision, float((out - ref).abs().max()))
    # (ii) NaN in, NaN out - where and only where the reference has them
    model.load_state_dict(sd0)
    bad2d, bad2f = x2d.clone(), x2f.clone()
    bad2d[1, 5, 7, 0] = float("nan")                                  # clip 1: every hypothesis of it
    badn = [n.clone() for n in noises]
    badn[0][0, 1, 3, 9, 2] = float("nan")                             # clip 0, hypothesis 1: only that trajectory
    model.noise_fn = lambda k, shape, device: badn[k]
    ref = orc.ddim_sample(sd0, bad2d, badn, T, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=bad2f)
    assert bool(torch.isfinite(ref[0, :, 0]).all()) and bool(torch.isnan(ref[1]).any()) and bool(torch.isnan(ref[0, :, 1]).any())
    for precision in ("f32", "bf16x3", "bf16x3_images"):
        model.precision = precision
        out = model(bad2d.to(DEV), None, input_2d_flip=bad2f.to(DEV)).cpu()
        assert torch.equal(torch.isnan(out), torch.isnan(ref)), precision
        ok = ~torch.isnan(ref)
        assert torch.allclose(out[ok], ref[ok], rtol=0, atol=1e-5), precision


def test_noflip_multistep_vs_oracle():
    """ddim_sample (no TTA): works for any P here; the reference itself only runs it at P=1."""
    from __graft_entry__ import make_model
    model, sd = make_model(3, 3, seed=78, flip=False)
    x2d, _ = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=3, n=3, seed=4)
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(x2d.to(DEV), None).cpu()
    ref = orc.ddim_sample(sd, x2d, noises, 3)
    assert torch.allclose(out, ref, rtol=0, atol=1e-5), (out - ref).abs().max()


def test_hypothesis_axis_is_independent():
    """size-independent property used at full size: a hypothesis' trajectory does not depend on the others, so
    running P=4 equals running its two halves (this is what makes the 8-GPU sharding exact)."""
    from __graft_entry__ import make_model
    model, _ = make_model(4, 2, seed=79)
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=4, n=2, seed=5)
    model.noise_fn = lambda k, shape, device: noises[k]
    full = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    halves = []
    for lo, hi in ((0, 2), (2, 4)):
        model.proposal_shard = (lo, hi)
        halves.append(model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)))
    model.proposal_shard = None
    assert torch.equal(full, torch.cat(halves, dim=2))


def test_clip_axis_is_independent():
    """a clip's result does not depend on what else is in the batch (rows never mix across clips): the B=3 run
    equals three B=1 runs bit for bit - the size-independent property behind batching whole sequences."""
    from __graft_entry__ import make_model
    model, _ = make_model(2, 2, seed=80)
    x2d, x2f = gu.synthetic_inputs_2d(B=3)
    noises = gu.synthetic_noises(B=3, P=2, n=2, seed=6)
    model.noise_fn = lambda k, shape, device: noises[k]
    full = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    for b in range(3):
        model.noise_fn = lambda k, shape, device, b=b: noises[k][b:b + 1]
        one = model(x2d[b:b + 1].to(DEV), None, input_2d_flip=x2f[b:b + 1].to(DEV))
        assert torch.equal(one, full[b:b + 1]), b


def test_large_batches_are_cut_along_the_clip_axis():
    """B beyond max_rows_per_launch runs as several library calls; same bits as one call."""
    from __graft_entry__ import make_model
    model, _ = make_model(2, 2, seed=86)
    x2d, x2f = gu.synthetic_inputs_2d(B=5)
    noises = gu.synthetic_noises(B=5, P=2, n=2, seed=13)
    model.noise_fn = lambda k, shape, device: noises[k]
    whole = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    model.max_rows_per_launch = 8                      # 2 clips per call at P=2 with flip-TTA -> chunks 2+2+1
    cut = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    assert torch.equal(whole, cut)


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_errors_are_loud():
    import pafuse_amd
    from pafuse_amd import _lib, ops
    with pytest.raises(_lib.PafuseError):
        ops.linear(torch.zeros(4, 33, device=DEV), torch.zeros(32, 33, device=DEV), torch.zeros(32, device=DEV))
    m = pafuse_amd.MixSTE2(3, 5, 5, 16, 2, 8, drop_path_rate=0.0, is_train=False).to(DEV)
    with pytest.raises(_lib.PafuseError):       # width 16 has no kernel: refuse, never fall back
        m(torch.zeros(1, 3, 5, 2, device=DEV), torch.zeros(1, 1, 3, 5, 3, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV))
    with pytest.raises(_lib.PafuseError):       # CPU tensors: no CPU fallback
        pafuse_amd.MixSTE2(27, 24, 5, 384, 1, 8, drop_path_rate=0.0, is_train=False)(
            torch.zeros(1, 27, 24, 2), torch.zeros(1, 1, 27, 24, 3), torch.zeros(1, dtype=torch.long))


def test_g9_evaluate_accumulators_golden():
    """the aggregation kernel + reductions reproduce the reference's 14 accumulators."""
    from types import SimpleNamespace
    from pafuse_amd.evaluate import evaluate_accumulators
    z = load_golden("g9_evaluate.npz")
    ds = SimpleNamespace(parts_joint_indices=gu.DATASET_PART_JOINTS, root_indices=gu.ROOT_INDICES,
                         parts_connection_indices=dict(gu.CONNECTION_INDICES))
    got = evaluate_accumulators(z["pred_parts"].to(DEV), z["gt_parts"].to(DEV), z["x2d"].to(DEV), z["traj"].to(DEV),
                                z["cam"].to(DEV), ds)
    keys = [k for k in z if k not in ("pred_parts", "gt_parts", "x2d", "traj", "cam")]
    assert len(keys) == 14 and set(keys) == set(got)
    for k in keys:
        # fp32 means of ~1.3 m synthetic errors: one ulp is 1.2e-7; device FMA contraction in the projection and a
        # different summation order move the result by a few ulps
        assert torch.allclose(got[k].cpu(), z[k], rtol=1e-6, atol=0), (k, (got[k].cpu() - z[k]).abs().max())


def test_evaluate_sequence_vs_oracle():
    """End to end on one synthetic sequence (60 frames -> 3 clips, the last overlapping): harness + HIP loop + device
    accumulators against the oracle's loop + the oracle's accumulators (reference-pinned by G5/G9/G10)."""
    from types import SimpleNamespace
    from __graft_entry__ import make_model
    from pafuse_amd import harness
    P, T = 2, 2
    model, sd = make_model(P, T, seed=81)
    ds = SimpleNamespace(parts_joint_indices=gu.DATASET_PART_JOINTS, root_indices=gu.ROOT_INDICES,
                         parts_connection_indices=dict(gu.CONNECTION_INDICES))
    g = torch.Generator().manual_seed(82)
    seq_2d = torch.rand(60, 134, 2, generator=g) * 2 - 1
    seq_3d = torch.randn(60, 134, 3, generator=g) * 0.25 + torch.tensor([0.0, 0.0, 4.0])
    cam = torch.tensor([2.29, 2.287, 0.025, 0.029, -0.207, 0.247, -0.003, -0.0009, -0.001])
    noises = gu.synthetic_noises(B=3, P=P, n=T, seed=8)
    model.noise_fn = lambda k, shape, device: noises[k]
    sums, n = harness.evaluate_sequence(model, ds, seq_2d, seq_3d, cam, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT)
    assert n == 3 * 27 and set(sums) == set(harness.ACCUMULATORS)
    # oracle side
    x2d = harness.cut_clips(seq_2d)
    x2f = harness.cut_clips(harness.flip_2d(seq_2d, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT))
    gt = harness.cut_clips(seq_3d)
    pred = orc.ddim_sample(sd, x2d, noises, T, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    want = orc.evaluate_accumulators(pred, orc.center_pose_parts(gt), x2d, gt[:, :, :1], cam)
    for k in harness.ACCUMULATORS:
        got = sums[k].cpu() / n
        assert torch.allclose(got, want[k], rtol=2e-6, atol=0), (k, got, want[k])


def test_h3wb_files_to_protocol_numbers():
    """the whole evaluate() flow from files: synthetic H3WB npz -> loader -> fetch -> every test video through the HIP
    loop -> frame-weighted protocol means, against the oracle's loop + accumulators on the same sequences."""
    import os
    from __graft_entry__ import make_model
    from pafuse_amd import h3wb, harness
    from tests.conftest import ROOT
    ds = h3wb.Human3WBDataset(os.path.join(ROOT, "tests", "golden", "h3wb_synth", "train_h3wb.npz"))
    keypoints = h3wb.prepare_keypoints(ds)
    kps_left, kps_right = ds.keypoints_metadata["keypoints_symmetry"]
    cams, p3, p2 = h3wb.fetch(["S8"], keypoints, ds)
    P, T = 2, 1
    model, sd = make_model(P, T, seed=87)
    assert (list(kps_left), list(kps_right)) == (gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT)   # the synthetic sides
    noises = gu.synthetic_noises(B=1, P=P, n=T, seed=14)
    model.noise_fn = lambda k, shape, device: noises[k]
    rep = h3wb.evaluate(model, ds, cams, p3, p2, kps_left, kps_right, log=lambda *a: None)
    total, n = None, 0
    for cam, s3, s2 in zip(cams, p3, p2):
        s2, s3 = torch.as_tensor(s2, dtype=torch.float32), torch.as_tensor(s3, dtype=torch.float32)
        x2d, gt = harness.cut_clips(s2), harness.cut_clips(s3)
        x2f = harness.cut_clips(harness.flip_2d(s2, kps_left, kps_right))
        pred = orc.ddim_sample(sd, x2d, noises, T, model.joints_left, model.joints_right, inputs_2d_flip=x2f)
        acc = orc.evaluate_accumulators(pred, orc.center_pose_parts(gt), x2d, gt[:, :, :1],
                                        torch.as_tensor(cam, dtype=torch.float32))
        mult = x2d.shape[0] * 27
        total = {k: mult * v for k, v in acc.items()} if total is None else {k: total[k] + mult * acc[k] for k in acc}
        n += mult
    for k in harness.ACCUMULATORS:
        want = total[k] / n * 1000.0
        assert torch.allclose(torch.tensor(rep[k]), want, rtol=5e-6, atol=0), (k, rep[k], want)


def test_infer_sequence_in_the_wild():
    """n4: the in-the-wild caller (input_3d=None) returns whole-body poses equal to the oracle's."""
    from types import SimpleNamespace
    from __graft_entry__ import make_model
    from pafuse_amd import harness
    model, sd = make_model(2, 1, seed=83)
    ds = SimpleNamespace(parts_joint_indices=gu.DATASET_PART_JOINTS, root_indices=gu.ROOT_INDICES,
                         parts_connection_indices=dict(gu.CONNECTION_INDICES))
    seq_2d = _seeded((40, 134, 2), 84).clamp(-1, 1)
    noises = gu.synthetic_noises(B=2, P=2, n=1, seed=9)
    model.noise_fn = lambda k, shape, device: noises[k]
    out = harness.infer_sequence(model, ds, seq_2d, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, batch_size=2)
    x2d = harness.cut_clips(seq_2d)
    x2f = harness.cut_clips(harness.flip_2d(seq_2d, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT))
    ref = orc.wb_pose_from_parts(orc.ddim_sample(sd, x2d, noises, 1, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT,
                                                 inputs_2d_flip=x2f))
    assert out.shape == (2, 1, 2, 27, 134, 3)
    assert torch.allclose(out, ref, rtol=0, atol=1e-5), (out - ref).abs().max()


def test_graph_replay_equals_eager():
    """the whole loop captured as one hipGraph (C ABI never allocates or synchronises) replays bit-identically,
    also with the parts forked onto aux streams inside the capture, and with fresh inputs on the second replay."""
    from __graft_entry__ import make_model
    model, _ = make_model(2, 3, seed=85)
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    for aux in (None, 2):
        model.aux_streams = [torch.cuda.Stream() for _ in range(aux)] if aux else []
        for seed in (10, 11):
            noises = gu.synthetic_noises(B=1, P=2, n=3, seed=seed)
            model.noise_fn = lambda k, shape, device: noises[k]
            model.use_graph = False
            eager = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
            model.use_graph = True
            graphed = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
            torch.cuda.synchronize()
            assert torch.equal(eager, graphed), (aux, seed)
        model._graphs.clear()


def test_torch_custom_ops_equal_the_modules():
    """torch.ops.pafuse.* (pafuse_amd/torch_ops.py) are the same entry points as the nn.Modules: bit-equal outputs."""
    from __graft_entry__ import make_model
    from pafuse_amd import torch_ops as to
    ops = torch.ops.pafuse
    x, w, b = _seeded((70, 256), 1).to(DEV), _seeded((512, 256), 2).to(DEV), _seeded((512,), 3).to(DEV)
    from pafuse_amd import ops as wrap
    assert torch.equal(ops.linear(x, w, b, True), wrap.linear(x, w, b, act="gelu"))
    ref = torch.nn.functional.gelu(torch.nn.functional.linear(x.double(), w.double(), b.double()))
    assert torch.allclose(ops.linear(x, w, b, True).double(), ref, rtol=1e-5, atol=2e-5)
    g, be = _seeded((256,), 4).to(DEV), _seeded((256,), 5).to(DEV)
    assert torch.equal(ops.layer_norm(x, g, be, 1e-6), wrap.layer_norm(x, g, be, 1e-6))
    model, _ = make_model(3, 2, seed=91)
    body = model.pose_estimator["body"]
    x2d = _seeded((2, 27, 24, 2), 6).to(DEV)
    x3d = _seeded((2, 3, 27, 24, 3), 7).to(DEV)
    t = torch.tensor([999, 499], device=DEV)
    blk = body.STEblocks[0]
    xs = _seeded((4, 24, 384), 8).to(DEV)
    from pafuse_amd.ops import attention, block_forward
    i2d, i2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=3, n=2, seed=12)
    model.noise_fn = lambda k, shape, device: noises[k]
    assert model.precision == "bf16x3"                 # the model-level ops' default precision is the modules' inference default
    assert torch.equal(ops.mixste_eval(x2d, x3d, t, list(body.parameters()), body.block_depth, body.num_heads), body(x2d, x3d, t))
    for precision in ("f16x2", "bf16x3_images", "bf16x3", "f32"):   # the split-precision schemes / kernel sets and the fp32 matrix cores
        model.precision = precision
        assert torch.equal(ops.mixste_eval(x2d, x3d, t, list(body.parameters()), body.block_depth, body.num_heads, precision),
                           body(x2d, x3d, t)), precision
        assert torch.equal(ops.block(xs, list(blk.parameters()), 8, precision), block_forward(blk, xs, precision=precision)), precision
        want = model(i2d.to(DEV), None, input_2d_flip=i2f.to(DEV))
        got = ops.ddim_loop(i2d.to(DEV), i2f.to(DEV), torch.stack(noises).to(DEV), *to.ddim_loop_args(model), precision)
        assert torch.equal(got, want), precision
    # an in-place weight update must remake the ops' cached images (keyed by storage and version)
    model.precision = "bf16x3"
    before = ops.block(xs, list(blk.parameters()), 8)
    with torch.no_grad():
        blk.attn.qkv.weight.mul_(1.5)
    after = ops.block(xs, list(blk.parameters()), 8)
    assert not torch.equal(before, after) and torch.equal(after, block_forward(blk, xs, precision="bf16x3"))
    qkv = _seeded((2 * 27 * 24, 3 * 384), 9).to(DEV)
    assert torch.equal(ops.attention(qkv, 8, 24, 0), attention(qkv, 8, 2 * 27, 24))
    assert torch.equal(ops.attention(qkv, 8, 27, 24),
                       attention(qkv, 8, 2 * 24, 27, group=24, group_stride=27 * 24, seq_stride=1, tok_stride=24))


def test_replicas_share_nothing_device_bound():
    """nn.DataParallel replicates the module per device and calls forward from worker threads (main_h3wb.py:699-705):
    a replica must build its own weight table and streams from its own parameters."""
    from __graft_entry__ import make_model
    model, _ = make_model(2, 1, seed=93)
    i2d, i2f = gu.synthetic_inputs_2d(B=2)
    noises = gu.synthetic_noises(B=2, P=2, n=1, seed=13)
    model.noise_fn = lambda k, shape, device: noises[k][:shape[0]]
    want = model(i2d.to(DEV), None, input_2d_flip=i2f.to(DEV))
    dp = torch.nn.DataParallel(model, device_ids=[0])
    assert torch.equal(dp(i2d.to(DEV), None, input_2d_flip=i2f.to(DEV)), want)
    replica = torch.nn.parallel.replicate(model, [0])[0]
    assert torch.equal(replica(i2d.to(DEV), None, input_2d_flip=i2f.to(DEV)), want)


def test_replicas_follow_the_parent_weights():
    """nn.DataParallel's replicas are fresh broadcast copies on EVERY forward: version 0, and the caching allocator hands out
    the addresses of the previous forward's copies again - a weight-image cache keyed on (pointer, version) would hit with
    the images of the OLD weights.  Replicas therefore never read the cache: forward, in-place weight update
    (an optimiser step / load_state_dict of another checkpoint), forward again, through replicas on one device listed twice
    (two worker threads, two replicas), must give what the updated parent gives."""
    from __graft_entry__ import make_model
    model, _ = make_model(2, 1, seed=93)
    i2d, i2f = gu.synthetic_inputs_2d(B=2)
    noises = gu.synthetic_noises(B=2, P=2, n=1, seed=13)
    model.noise_fn = lambda k, shape, device: noises[k][:shape[0]] if shape[0] == 2 else noises[k][model._clip:model._clip + 1]
    model._clip = 0

    def replicas_forward():
        outs = []
        for b, rep in enumerate(torch.nn.parallel.replicate(model, [0, 0])):     # what DataParallel.forward does per call
            assert getattr(rep, "_is_replica", False)
            model._clip = b
            outs.append(rep(i2d[b:b + 1].to(DEV), None, input_2d_flip=i2f[b:b + 1].to(DEV)))
        return torch.cat(outs)

    first = replicas_forward()
    assert torch.equal(first, model(i2d.to(DEV), None, input_2d_flip=i2f.to(DEV)))
    _, sd2 = make_model(2, 1, seed=94)
    model.load_state_dict(sd2)                                    # in place: same parameter storages, new values
    second = replicas_forward()
    assert not torch.equal(first, second)
    assert torch.equal(second, model(i2d.to(DEV), None, input_2d_flip=i2f.to(DEV)))


def test_image_caches_do_not_outlive_their_tensors():
    """the torch ops' image cache pins the tensor an image was made from: a temporary weight freed and another allocated at
    the same address (same shape, version 0) must not be served the first one's image."""
    from pafuse_amd import torch_ops as to
    x = _seeded((64, 256), 31).to(DEV)
    outs = []
    for seed in (32, 33):
        w = _seeded((256, 256), seed, 0.05).to(DEV)               # a per-call temporary: freed at the end of the iteration
        img = to.cached_split_image(w, 0)
        from pafuse_amd import ops
        outs.append((ops.linear_split(x, w, torch.zeros(256, device=DEV), image=img).cpu(), ops.linear_split(x, w, torch.zeros(256, device=DEV)).cpu()))
        del w, img
    for cached, fresh in outs:
        assert torch.equal(cached, fresh)
    assert not torch.equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("precision,fold,factor", [("f16x2", None, 1.5), ("bf16x3_images", None, 1.5), ("bf16x3", None, 1.5), ("bf16x3", False, 2.0)])
def test_folded_layernorm_with_large_row_means(precision, fold, factor):
    """The folded LayerNorm multiplies un-normalised rows: with |mean| >> std (outlier channels of a trained residual
    stream) a three-term form rstd (acc - mean ls) + lt cancels large numbers (round 3's fold, gone since round 5).
    Every split mode now stores x - mean(row) (the stream is centred; the mean itself is dead information to a stack of
    LayerNorms and is dropped), so nothing cancels, and the fold is the default everywhere.  Against an fp64 evaluation of one
    denoiser pass whose post-norm biases (Spatial_norm, Temporal_norm: what every block hands to the next) put every row of
    the residual stream at mean 10 with std ~ 1: not further from exact arithmetic than 1.5 x the reference's own fp32
    arithmetic (whose LayerNorms see the same rows).  'bf16x3' without the fold (fp32 LayerNorm statistics of rows at mean 10,
    as the reference computes them) measures 1.1 - 1.6 x per part - the face 1.6 x against a host oracle that is itself 8.0e-7
    from exact - and is held to 2 x."""
    from __graft_entry__ import make_model
    model, sd = make_model(2, 2, seed=95)
    for part in model.pose_estimator:
        sd[f"pose_estimator.{part}.Spatial_norm.bias"] += 10.0
        sd[f"pose_estimator.{part}.Temporal_norm.bias"] += 10.0
    model.load_state_dict(sd)
    model.precision = precision
    for m in model.denoisers().values():
        m.fold_layernorm = fold
    sd64 = {k: v.double() for k, v in sd.items()}
    x2d, _ = gu.synthetic_inputs_2d(B=1)
    x3d = _seeded((1, 2, 27, 134, 3), 53).clamp(-1.1, 1.1)
    t = torch.tensor([499])
    for part, idx in orc.PART_JOINTS.items():
        pre = f"pose_estimator.{part}."
        truth = orc.mixste2_eval(sd64, pre, x2d[..., idx, :].double(), x3d[..., idx, :].double(), t)
        ref32 = orc.mixste2_eval(sd, pre, x2d[..., idx, :], x3d[..., idx, :], t)
        hip = model.pose_estimator[part](x2d[..., idx, :].to(DEV), x3d[..., idx, :].to(DEV), t.to(DEV)).cpu()
        e_ref, e_hip = (ref32.double() - truth).abs(), (hip.double() - truth).abs()
        assert e_hip.mean() <= factor * e_ref.mean(), (precision, fold, part, float(e_hip.mean()), float(e_ref.mean()))


def test_g11_scale_golden():
    """ft2d.scale != 1: clamp bounds, /scale on the way in, *scale and clamp on the way out."""
    from __graft_entry__ import make_model
    z = load_golden("g11_scale.npz")
    model, sd = make_model(2, 2, seed=111, scale=2.0)
    assert gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = [n * 1.5 for n in gu.synthetic_noises(B=1, P=2, n=2, seed=12)]
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    assert torch.allclose(out, z["out"], rtol=0, atol=2e-5), (out - z["out"]).abs().max()


def test_c_abi_from_plain_c():
    """the boundary is a C ABI: a C program with only the HIP runtime links the library and gets exact results."""
    import os
    import subprocess
    from tests.conftest import ROOT
    exe = os.path.join(ROOT, "build", "linear_smoke")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.join(ROOT, "pafuse_amd")
    subprocess.check_call(["gcc", os.path.join(ROOT, "tests", "cabi", "linear_smoke.c"), "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lpafuse_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatches" in out.stdout


def test_unit_ops_on_a_seeded_sweep_of_shapes():
    """40 seeded random shapes inside the documented limits (M tails, every N % 32 class, K up to 1152; sequence
    lengths 1..80, head dims 4..48, contiguous and strided sequences; LayerNorm widths up to 768): each against an
    fp64 evaluation of the same op."""
    import random
    from pafuse_amd import ops
    rng = random.Random(20240607)
    for case in range(16):
        M, N, K = rng.randint(1, 700), 32 * rng.randint(1, 40), 32 * rng.randint(1, 36)
        act = rng.choice([None, "gelu"])
        x, w, b = _seeded((M, K), 100 + case), _seeded((N, K), 200 + case, K ** -0.5), _seeded((N,), 300 + case, 0.1)
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        ref = torch.nn.functional.gelu(ref) if act else ref
        out = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act).cpu().double()
        assert torch.allclose(out, ref, rtol=0, atol=2.5e-7 * K ** 0.5 + 1e-6), ("linear", M, N, K, act)
    for case in range(16):
        heads = rng.choice([1, 2, 4, 8])
        d = 4 * rng.randint(1, 12)
        L, S = rng.randint(1, 80), rng.randint(1, 9)
        C = heads * d
        temporal = rng.random() < 0.5
        if temporal:                                   # S groups of J interleaved sequences, tokens J rows apart
            J = rng.randint(1, 7)
            qkv = _seeded((S * L * J, 3 * C), 400 + case)
            out = ops.attention(qkv.to(DEV), heads, S * J, L, group=J, group_stride=L * J, seq_stride=1, tok_stride=J).cpu()
            q = qkv.view(S, L, J, 3, heads, d).permute(3, 0, 2, 4, 1, 5).double()          # 3, S, J, h, L, d
            o = torch.softmax(q[0] @ q[1].transpose(-1, -2) * d ** -0.5, -1) @ q[2]         # S, J, h, L, d
            ref = o.permute(0, 3, 1, 2, 4).reshape(S * L * J, C)
        else:
            qkv = _seeded((S * L, 3 * C), 400 + case)
            out = ops.attention(qkv.to(DEV), heads, S, L).cpu()
            ref = _attn_ref(qkv.double(), S, L, heads)
        assert torch.allclose(out.double(), ref, rtol=0, atol=3e-6), ("attention", heads, d, L, S, temporal)
    for case in range(8):
        M, C = rng.randint(1, 300), rng.randint(1, 768)
        x, g, b = _seeded((M, C), 500 + case), _seeded((C,), 600 + case), _seeded((C,), 700 + case)
        eps = rng.choice([1e-5, 1e-6])
        ref = torch.nn.functional.layer_norm(x.double(), (C,), g.double(), b.double(), eps)
        out = ops.layer_norm(x.to(DEV), g.to(DEV), b.to(DEV), eps).cpu().double()
        assert torch.allclose(out, ref, rtol=0, atol=5e-6), ("layer_norm", M, C)


# ------------------------------------------------------------------------------------------ bf16-operand mode
@pytest.mark.parametrize("M,N,K", [(200, 1152, 384), (129, 448, 224), (77, 672, 224), (50, 96, 64), (33, 32, 32)])
def test_linear_bf16_operands(M, N, K):
    """opt-in mode: both operands rounded to bf16 (RNE), exact products, fp32 accumulation - checked against that
    arithmetic carried out in fp64 on bf16-rounded inputs."""
    from pafuse_amd import ops
    x, w, b = _seeded((M, K), 1), _seeded((N, K), 2, K ** -0.5), _seeded((N,), 3, 0.1)
    ref = torch.nn.functional.linear(x.bfloat16().double(), w.bfloat16().double(), b.double())
    out = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), bf16=True).cpu()
    assert torch.allclose(out.double(), ref, rtol=0, atol=2.5e-7 * K ** 0.5 + 1e-6), (out - ref).abs().max()
    plain = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV)).cpu()
    assert not torch.equal(plain, out)                                   # it really is the other arithmetic


# |MPJPE_bf16 - MPJPE_f32| bound (mm) for the opt-in bf16-operand mode at BASELINE configs[1] (P=5, T=5).  Measured on
# MI355X (profiles/r02_parity_report.json, case "5,5,1,bf16", against the oracle): J-Best 3.8, P-Best 3.5, P-Agg 3.4,
# J-Agg 3.6 mm (max over the five steps); pointwise mean 1.9e-3 m, max 1.1e-2 m.  Rounding every matrix operand to 8 significant bits costs that much on random weights - bf16 autocast
# of the reference itself sits at the same distance (hard part 1).
BF16_MPJPE_TOL_MM = 6.0


def test_bf16_precision_mode_end_to_end():
    """model.precision = 'bf16' (BASELINE configs[1]: P=5, T=5): the documented distance from the fp32 path holds per
    protocol, and switching back restores the fp32 path bit for bit."""
    from __graft_entry__ import make_model
    model, _ = make_model(5, 5, seed=61)
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=5, n=5, seed=21)
    model.noise_fn = lambda k, shape, device: noises[k]
    model.precision = "f32"
    f32 = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    model.precision = "bf16"
    assert model.precision == "bf16"
    low = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    model.precision = "f32"
    again = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    assert torch.equal(f32, again)
    err = (low - f32).abs()
    assert bool(torch.isfinite(low).all()) and 0 < float(err.mean()) < 4e-3 and float(err.max()) < 0.05, (err.max(), err.mean())
    target = orc.center_pose_parts(gu.synthetic_target_3d(1))
    got, want = _mpjpe_report(low.cpu(), target, x2d), _mpjpe_report(f32.cpu(), target, x2d)
    for k in want:
        assert (got[k] - want[k]).abs().max() <= BF16_MPJPE_TOL_MM, (k, (got[k] - want[k]).abs())


@pytest.mark.parametrize("mode", [0, 2])
def test_g18_mixste2_constructor_options_golden(mode):
    """MixSTE2(mlp_ratio=3, qkv_bias=False, qk_scale=0.3, dropout rates) in eval against the reference's output (golden
    G18), on the fp32 and the split-precision products; training such a model is refused."""
    import pafuse_amd
    from tests.test_oracle_golden import G18_KW
    z = load_golden("g18_mixste_options.npz")
    m = pafuse_amd.MixSTE2(**G18_KW)
    sd = gu.seeded_state_dict(m.state_dict(), seed=181)
    assert gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.operand_bf16 = mode
    out = m(z["x2d"].to(DEV), z["x3d"].to(DEV), z["t"].to(DEV)).cpu()
    assert (out - z["out"]).abs().max() <= 1e-5, (out - z["out"]).abs().max()
    m.is_train, m.operand_bf16 = True, 0
    with pytest.raises(NotImplementedError, match="PAFUSE configuration"):
        m(z["x2d"].to(DEV), z["x3d"][:, 0].to(DEV), z["t"].to(DEV))


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16x3_images", "f16x2"])
def test_empty_inputs_give_empty_outputs(precision):
    """torch runs the reference on an empty batch and returns an empty tensor (every reshape of diffusionpose.py / mixste.py keeps
    a zero-sized batch axis); so do the modules and the unit entry points - no launch, no error."""
    import __graft_entry__ as ge
    from pafuse_amd import ops
    model, _ = ge.make_model(3, 2, seed=51)
    model.precision = precision
    x2d = torch.zeros(0, 27, 134, 2, device=DEV)
    out = model(x2d, None, input_2d_flip=x2d.clone())
    assert tuple(out.shape) == (0, 2, 3, 27, 134, 3) and out.dtype == torch.float32
    if precision == "f32":
        assert tuple(ops.linear(torch.zeros(0, 64, device=DEV), torch.zeros(96, 64, device=DEV), torch.zeros(96, device=DEV)).shape) == (0, 96)
        assert tuple(ops.layer_norm(torch.zeros(0, 64, device=DEV), torch.ones(64, device=DEV), torch.zeros(64, device=DEV), 1e-6).shape) == (0, 64)
        assert tuple(ops.attention(torch.zeros(0, 3 * 64, device=DEV), 8, 0, 5).shape) == (0, 64)
