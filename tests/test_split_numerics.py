"""Numerics of the split-precision contractions (csrc/mfma_split.h) on the float64 pipeline model (oracle/staged_ref.py):
every per-row GEMM of the six tile kernels is replaced by its f16 hi/lo emulation -- operands split as the kernels split them
(round-to-nearest f16 high half, low half scaled by 2^11, adjoint rows scaled per row by a power of two), products
xh.wh + xl.wh + xh.wl accumulated in float32 -- while everything else stays float64, so the difference to the unmodified model
is the error the split contractions ADD.  CPU only; the hardware side is tools/split_lab.hip (T1-T4) and the GPU parity suite.

Pinned here:
  * the added error at trained-checkpoint magnitudes is an order below the float32 pipeline's own rounding error
    (and far below the north-star bars 1e-4 eV / 1e-3 eV/A),
  * without the per-row scaling of the adjoint operands the force error is several times larger -- why SCALED exists."""

from __future__ import annotations

import inspect
import textwrap

import numpy as np
import pytest

import oracle.staged_ref as sr
from chgnet_amd.pack import pack_batch, pack_weights
from conftest import load_case
from oracle.staged_ref import StagedModel, dsilu, ln_bwd, ln_fwd, sigmoid, silu

LO_SCALE = 2048.0


def _split(x, scale_rows):
    x = np.ascontiguousarray(x, np.float32)
    s = np.float32(1)
    if scale_rows:
        m = np.abs(x).max(1, keepdims=True)
        m[m == 0] = 1
        s = np.exp2(-np.floor(np.log2(m))).astype(np.float32)
    xs = x * s
    hi = xs.astype(np.float16).astype(np.float32)
    lo = ((xs - hi) * np.float32(LO_SCALE)).astype(np.float16).astype(np.float32)
    return hi, lo, s


def split_matmul(X, Wt, scaled):
    """X [rows, K] . Wt[F, K]^T like gemm_split<.., SCALED>: float32 accumulation of the three f16 products."""
    xh, xl, s = _split(X, scaled)
    wh, wl, _ = _split(Wt, False)
    acc = (xl @ wh.T + xh @ wl.T) * np.float32(1.0 / LO_SCALE) + xh @ wh.T
    return (acc / s).astype(np.float64)


class SplitModel(StagedModel):
    scale_adjoint = True

    def gated_fwd(self, z, p, hidden):
        W, D = self.W, 64
        if hidden:
            H = silu(z)
            c = split_matmul(H[:, :D], W(p + "w2c"), False) + W(p + "b2c")
            g = split_matmul(H[:, D:], W(p + "w2g"), False) + W(p + "b2g")
        else:
            H, c, g = None, z[:, :D], z[:, D:]
        n1, xh1, rs1 = ln_fwd(c, W(p + "ln1_g"), W(p + "ln1_b"))
        n2, xh2, rs2 = ln_fwd(g, W(p + "ln2_g"), W(p + "ln2_b"))
        a1, a2 = silu(n1), sigmoid(n2)
        return a1 * a2, (z, H, n1, xh1, rs1, n2, xh2, rs2, a1, a2)

    def gated_bwd(self, gy, cache, p, hidden, wg=None):
        W, D = self.W, 64
        z, H, n1, xh1, rs1, n2, xh2, rs2, a1, a2 = cache
        gc = ln_bwd(gy * a2 * dsilu(n1), W(p + "ln1_g"), xh1, rs1)
        gg = ln_bwd(gy * a1 * a2 * (1 - a2), W(p + "ln2_g"), xh2, rs2)
        if hidden:
            sc = self.scale_adjoint
            gH = np.concatenate([split_matmul(gc, W(p + "w2c").T, sc), split_matmul(gg, W(p + "w2g").T, sc)], axis=1)
            return gH * dsilu(z)
        return np.concatenate([gc, gg], axis=1)


def _patched_run():
    """StagedModel.run with the two angle-block contractions routed through split_matmul (the source is patched textually so
    that the test follows the pipeline model instead of duplicating it)."""
    src = textwrap.dedent(inspect.getsource(StagedModel.run))
    fwd, bwd = 'ang[l] @ W(p + "w_ang").T', 'Gang[:] += Gz @ W(p + "w_ang")'
    assert src.count(fwd) == 2 and src.count(bwd) == 1
    src = src.replace(fwd, 'split_matmul(ang[l], W(p + "w_ang"), False)').replace(bwd, 'Gang[:] += split_matmul(Gz, W(p + "w_ang").T, self.scale_adjoint)')
    ns = dict(sr.__dict__)
    ns["split_matmul"] = split_matmul
    exec(src, ns)  # noqa: S102
    return ns["run"]


SplitModel.run = _patched_run()


@pytest.fixture(scope="module")
def tl_case(trained_like_weights):
    pw = pack_weights(trained_like_weights)
    pb = pack_batch([load_case(n)[0] for n in ("limno2", "s16tri", "li9co7o16")])
    return pw, pb, StagedModel(pw, np.float64).run(pb)


def _errs(got, ref):
    return {k: float(np.abs(got[k] - ref[k]).max()) for k in ("e", "f", "s")}


def test_split_contractions_add_less_than_the_f32_pipeline_itself(tl_case):
    pw, pb, ref = tl_case
    f32 = _errs(StagedModel(pw, np.float32).run(pb), ref)                # everything in float32: the engine's own error class
    add = _errs(SplitModel(pw, np.float64).run(pb), ref)                 # float64 everywhere except the split contractions
    assert np.abs(ref["f"]).max() > 2.0                                   # eV/A-scale forces: the regime the bars are meant for
    assert add["e"] < 1e-6 and add["f"] < 5e-6 and add["s"] < 5e-5, add   # north star: 1e-4 eV, 1e-3 eV/A
    assert add["f"] < 0.35 * f32["f"] and add["s"] < 0.35 * f32["s"], (add, f32)


def test_row_scale_keeps_small_adjoint_rows_exact_and_is_neutral_otherwise(tl_case):
    """Adjoint rows can be arbitrarily small (far atoms, saturated gates).  With the low half carried at 2^11 the split holds
    its precision down to |x| ~ 1e-7 without any row scale (the pipeline result is the same either way), below that the f16 high
    half underflows; the power-of-two row scale (SCALED) removes the floor."""
    pw, pb, ref = tl_case
    scaled = _errs(SplitModel(pw, np.float64).run(pb), ref)
    m = SplitModel(pw, np.float64)
    m.scale_adjoint = False
    unscaled = _errs(m.run(pb), ref)
    assert abs(unscaled["e"] - scaled["e"]) < 1e-9 and unscaled["f"] < 2.0 * scaled["f"] + 1e-7
    rng = np.random.default_rng(0)
    W = rng.normal(0, 0.3, (64, 64))
    for mag, worst_unscaled in ((1e-3, 1e-6), (1e-10, None)):
        X = rng.normal(0, mag, (256, 64))
        exact = X @ W.T
        denom = np.abs(X) @ np.abs(W).T
        err_s = (np.abs(split_matmul(X, W, True) - exact) / denom).max()
        err_u = (np.abs(split_matmul(X, W, False) - exact) / denom).max()
        assert err_s < 3e-7, (mag, err_s)
        if worst_unscaled is None:
            assert err_u > 1e-3, (mag, err_u)              # 1e-10 rows: the unscaled f16 halves are all subnormal
        else:
            assert err_u < worst_unscaled, (mag, err_u)


# ---- three-piece bf16 products of the long weight-gradient contractions (csrc/kernels_train.h: cut3, k_xty3) ------------------------
def _cut3(x):
    """fp32 -> three bf16 pieces by truncation, as cut3 does it: masks and two exact fp32 subtractions."""
    x = np.ascontiguousarray(x, np.float32)
    mask = np.uint32(0xFFFF0000)
    hi = (x.view(np.uint32) & mask).view(np.float32)
    r1 = x - hi
    mid = (r1.view(np.uint32) & mask).view(np.float32)
    r2 = r1 - mid
    lo = (r2.view(np.uint32) & mask).view(np.float32)
    return hi, mid, lo


def test_three_bf16_pieces_are_exact_and_six_products_reach_fp32_accuracy():
    """Pins the arithmetic k_xty3 relies on: (i) hi + mid + lo == x bit for bit over 30 decades of magnitude (bf16 keeps fp32's
    exponent: no scaling anywhere), every piece has at most 8 significant bits (so bf16 x bf16 products are exact in fp32);
    (ii) the six products the kernel keeps reproduce A^T B over 4,096 rows to ~2^-22 of sum |a||b| -- the accuracy of an fp32
    dot product -- while the three-product (hi/mid only) form is two orders worse: why the `lo` planes exist."""
    rng = np.random.default_rng(21)
    x = (rng.normal(size=200_000) * np.exp(rng.uniform(-35, 35, 200_000))).astype(np.float32)
    x[:5] = [0.0, -0.0, 1.0, np.float32(2.0) ** -120, -np.float32(3.0e38)]
    hi, mid, lo = _cut3(x)
    assert np.array_equal((hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(hi + (mid + lo), x)                          # also in fp32: the pieces do not overlap
    for piece in (hi, mid, lo):                                        # bf16-representable: low 16 bits clear
        assert not (piece.view(np.uint32) & np.uint32(0xFFFF)).any()
    assert (np.sign(mid) * np.sign(x) >= 0).all() and (np.sign(lo) * np.sign(x) >= 0).all()   # truncation: pieces share x's sign

    rows, M, N = 4096, 64, 32
    A = (rng.normal(size=(rows, M)) * np.exp(rng.uniform(-6, 6, (rows, 1)))).astype(np.float32)   # adjoint rows: wide dynamic range
    B = rng.normal(size=(rows, N)).astype(np.float32)
    exact = A.astype(np.float64).T @ B.astype(np.float64)
    bound = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)
    a, b = _cut3(A), _cut3(B)
    prod = lambda i, j: a[i].astype(np.float64).T @ b[j].astype(np.float64)   # noqa: E731  (each bf16 product is exact; f64 stands for the f32 accumulator)
    six = prod(2, 0) + prod(0, 2) + prod(1, 1) + prod(1, 0) + prod(0, 1) + prod(0, 0)
    three = prod(1, 0) + prod(0, 1) + prod(0, 0)
    err6 = float((np.abs(six - exact) / bound).max())
    err3 = float((np.abs(three - exact) / bound).max())
    assert err6 < 2.0 ** -22, err6                                     # dropped terms: mid lo, lo mid, lo lo ~ 3 x 2^-24
    assert err3 > 30 * err6 and err3 < 2.0 ** -14, (err3, err6)


def test_sincos_reduction_model():
    """csrc/kernels_geom.h:sincos_cw (three-term Cody-Waite reduction by pi/2 + cephes minimax polynomials), the arithmetic of the
    radial / Fourier bases since round 4, restated in float32 numpy with the FMAs emulated in float64: max abs error against float64
    sin / cos below 1.2e-7 over [0, 1000] (the bases' arguments reach 31 pi ~ 100); torch.sin on fp32 (the reference) is accurate to
    ~6e-8 on the same arguments."""
    f32 = np.float32
    hi, mid, lo = f32(1.5707963705062866), f32(-4.371138828673793e-08), f32(-1.7151245100058819e-15)
    assert float(hi) + float(mid) + float(lo) == pytest.approx(np.pi / 2, abs=1e-22)

    def fma(a, b, c):
        return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)

    def sincos(x):
        n = np.rint((x * f32(0.636619772367581343)).astype(np.float32)).astype(np.float32)
        r = fma(n, -hi, x)
        r = fma(n, -mid, r)
        r = fma(n, -lo, r)
        z = (r * r).astype(np.float32)
        ps = fma(f32(-1.9515295891e-4), z, f32(8.3321608736e-3))
        ps = fma(ps, z, f32(-1.6666654611e-1))
        s0 = fma((ps * z).astype(np.float32), r, r)
        pc = fma(f32(2.443315711809948e-5), z, f32(-1.388731625493765e-3))
        pc = fma(pc, z, f32(4.166664568298827e-2))
        c0 = fma((pc * z).astype(np.float32), z, fma(z, f32(-0.5), f32(1.0)))
        q = n.astype(np.int64)
        sv, cv = np.where(q & 1, c0, s0), np.where(q & 1, s0, c0)
        return np.where(q & 2, -sv, sv), np.where((q + 1) & 2, -cv, cv)

    rng = np.random.default_rng(0)
    for top in (3.2, 100.0, 1000.0):
        x = (rng.random(400_000) * top).astype(np.float32)
        s, c = sincos(x)
        assert np.abs(s - np.sin(x.astype(np.float64))).max() < 1.2e-7 and np.abs(c - np.cos(x.astype(np.float64))).max() < 1.2e-7
    s, c = sincos(np.array([0.0], np.float32))
    assert s[0] == 0.0 and c[0] == 1.0
