// kernels_conv.h -- message-passing kernels of the CHGNet hot path, forward and reverse.
//
// Reference ops replaced (file:line relative to /root/reference/chgnet):
//   AtomConv.forward     model/layers.py:113-132   -> k_atomconv<fwd>  / k_atomconv<bwd>
//   BondConv.forward     model/layers.py:238-260   -> k_angle<hidden,fwd> / k_angle<hidden,bwd>
//   AngleUpdate.forward  model/layers.py:348-360   -> k_angle<single,fwd> / k_angle<single,bwd>
//   GatedMLP.forward     model/functions.py:177-183 (shared body: gated_forward / gated_backward)
//   aggregate            model/functions.py:10-40  -> segmented column sums (rows arrive sorted by owner)
//   nn.Linear partial products + mlp_out + residual -> k_rows_gemm
//   torch.autograd.grad  model/model.py:517-535    -> the *_bwd kernels (input gradients only)
//
// First gated-MLP layer, factorised:  W1 [x_a | x_b | x_c] = W1a x_a + W1b x_b + W1c x_c, and the
// partial products only depend on the atom / bond they come from, so they are computed once per atom /
// bond by k_rows_gemm (tables P, Q, R, S) and the per-edge / per-angle kernels gather + add them.
#pragma once

#include "mfma_tile.h"
#include "mfma_split.h"

namespace chg {

constexpr int TS = 2 * D + PAD;  // LDS tile row stride (floats) for 128-wide rows
constexpr int WS = D + PAD;      // LDS weight row stride for K = 64
constexpr int TILE_FLOATS = TILE_ROWS * TS;
constexpr int VEC_SLOTS = 6;     // b2c b2g ln1_g ln1_b ln2_g ln2_b
// Split-precision weight images (mfma_split.h), in 16-byte chunks: a 64x64 matrix is 1,024 chunks (16 KiB, both planes),
// the 128x64 angle block 2,048 (its transposed image as well).  The BondConv adjoint needs both directions of three matrices:
// it uses ONE row-major image per matrix (mode 2: 72 KiB; the two-image form would be 128 KiB and does not fit next to the tiles).
constexpr int IMG64 = 1024, IMG128 = 2048;
constexpr int angle_split(bool hidden, bool bwd) { return (hidden && bwd) ? 2 : 1; }

struct GatedW {            // global pointers into the weight blob
  const float *w2c, *b2c, *w2g, *b2g, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
};

__device__ __forceinline__ void stage_gated_vecs(float* vecs, const GatedW& gw, bool hidden, int tid) {
  if (hidden) {
    stage_vector(vecs + 0 * D, gw.b2c, D, tid);
    stage_vector(vecs + 1 * D, gw.b2g, D, tid);
  }
  stage_vector(vecs + 2 * D, gw.ln1_g, D, tid);
  stage_vector(vecs + 3 * D, gw.ln1_b, D, tid);
  stage_vector(vecs + 4 * D, gw.ln2_g, D, tid);
  stage_vector(vecs + 5 * D, gw.ln2_b, D, tid);
}

__device__ __forceinline__ V64 param64(const float* vec, int g) {
  V64 p;
  read_dl<VT>(vec, g, p.t);
  return p;
}

// Forward state of one gated MLP row kept for the backward (64 registers per lane: the affine
// core n1 = xh1*g1+b1 is one FMA to recompute, and with sigmoid(n1) kept the backward needs no
// transcendental for it)
struct GatedState {
  V64 xh1, xh2;   // LayerNorm-normalised branches (before affine)
  V64 sg1, a2;    // sigmoid(n1), sigmoid(n2);  a1 = silu(n1) = n1 * sg1
  float rstd1, rstd2;
};

// z (pre-activation of the first layer, 128 = core|gate) -> y = silu(n1) * sigmoid(n2), plus the state.
//   HIDDEN: c = W2c silu(zc) + b2c, g = W2g silu(zg) + b2g ; else c = zc, g = zg
//   HIDDEN: zc, zg are replaced by silu'(zc), silu'(zg), which is all the backward needs of them.
//   SLIM: sigmoid(n1) is not kept either (the adjoint recomputes it: 32 transcendentals for 16 registers).
// Training context of one tile (TRAIN instantiations of the adjoint kernels only):
// where the per-row quantities that the weight-gradient reductions (kernels_train.h) contract are dumped,
// and the LDS tile used for the in-tile column sums of the LayerNorm-affine gradients.
struct TrainTile {
  float* T;        // the wave's LDS tile (free while the gated MLP runs in registers)
  int nvalid, lane;
  float* hrow;     // this lane's row of the hidden-activation dump [rows][128] (null: row past the end)
  float* grow;     // this lane's row of the second-layer adjoint dump [rows][128]
  float ln[4];     // running column sums (column = lane): d ln1_g, d ln1_b, d ln2_g, d ln2_b
};

// SPLIT: 0 = W2c / W2g are padded f32 rows, 1 = split-precision images, 2 = row-major split images (mfma_split.h)
// LEAN: the hidden layer's split contractions in the two-sweep form (mfma_split.h gemm_split4: the forward kernels)
template <bool HIDDEN, bool SLIM = false, bool TRAIN = false, int SPLIT = 0, bool LEAN = false>
__device__ __forceinline__ void gated_forward(V64& zc, V64& zg, const float* W2c, const float* W2g, const float* vecs,
                                              int j, int g, GatedState& s, V64& y, TrainTile* tt = nullptr) {
  if (HIDDEN) {
    V64 hc, hg;
    CHG_EV(ft) {
      const f32x4 sc = sigmoid4(zc.t[ft]);
      hc.t[ft] = zc.t[ft] * sc;
      zc.t[ft] = sc * (1.0f + zc.t[ft] * (1.0f - sc));
    }
    CHG_EV(ft) {
      const f32x4 sg = sigmoid4(zg.t[ft]);
      hg.t[ft] = zg.t[ft] * sg;
      zg.t[ft] = sg * (1.0f + zg.t[ft] * (1.0f - sg));
    }
    if (TRAIN && tt->hrow) {   // hidden activations: the B operand of dW2 = gn'^T . h
      write_dl<VT>(tt->hrow, g, hc.t);
      write_dl<VT>(tt->hrow + D, g, hg.t);
    }
    s.xh1 = param64(vecs + 0 * D, g);
    s.xh2 = param64(vecs + 1 * D, g);
    if (SPLIT == 2) {
      gemm_rm<VT, VT, false, false>(s.xh1.t, reinterpret_cast<const _Float16*>(W2c), D, D, hc.t, j, g, j + 16 * g);
      gemm_rm<VT, VT, false, false>(s.xh2.t, reinterpret_cast<const _Float16*>(W2g), D, D, hg.t, j, g, j + 16 * g);
    } else if (SPLIT == 1) {
      gemm_split<VT, VT, false, LEAN>(s.xh1.t, reinterpret_cast<const h16x8*>(W2c), D, hc.t, j, g);
      gemm_split<VT, VT, false, LEAN>(s.xh2.t, reinterpret_cast<const h16x8*>(W2g), D, hg.t, j, g);
    } else {
      gemm_dl<VT, VT>(s.xh1.t, W2c, WS, hc.t, j, g);
      gemm_dl<VT, VT>(s.xh2.t, W2g, WS, hg.t, j, g);
    }
  } else {
    s.xh1 = zc;
    s.xh2 = zg;
  }
  s.rstd1 = ln_normalize(s.xh1);
  s.rstd2 = ln_normalize(s.xh2);
  {
    const V64 gam = param64(vecs + 4 * D, g), bet = param64(vecs + 5 * D, g);
    CHG_EV(ft) s.a2.t[ft] = sigmoid4(s.xh2.t[ft] * gam.t[ft] + bet.t[ft]);
  }
  {
    const V64 gam = param64(vecs + 2 * D, g), bet = param64(vecs + 3 * D, g);
    CHG_EV(ft) {
      const f32x4 n1 = s.xh1.t[ft] * gam.t[ft] + bet.t[ft];
      const f32x4 sg = sigmoid4(n1);
      if (!SLIM) s.sg1.t[ft] = sg;
      y.t[ft] = n1 * sg * s.a2.t[ft];
    }
  }
}

// gy = dE/dy  ->  gzc, gzg = dE/dz (128 wide);  dzc, dzg: silu'(z) as left by gated_forward<true>
// column sums of two 64-wide per-row vectors over the valid rows of the tile (column = lane)
__device__ __forceinline__ void tile_colsum2(TrainTile* tt, int j, int g, const V64& a, const V64& b, float& sa, float& sb) {
  float* Trow = tt->T + j * TS;
  __builtin_amdgcn_wave_barrier();   // earlier reads of the tile stay above these writes
  write_dl<VT>(Trow, g, a.t);
  write_dl<VT>(Trow + D, g, b.t);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int rr = 0; rr < TILE_ROWS; ++rr)
    if (rr < tt->nvalid) {
      sa += tt->T[rr * TS + tt->lane];
      sb += tt->T[rr * TS + D + tt->lane];
    }
  __builtin_amdgcn_wave_barrier();
}

// SPLIT: 1 = W2c / W2g are the split-precision images of W2c^T / W2g^T, 2 = the row-major images of W2c / W2g themselves
template <bool HIDDEN, bool SLIM = false, bool TRAIN = false, int SPLIT = 0>
__device__ __forceinline__ void gated_backward(const V64& gy, const V64& dzc, const V64& dzg, const float* W2c, const float* W2g,
                                               const float* vecs, int j, int g, const GatedState& s, V64& gzc, V64& gzg,
                                               TrainTile* tt = nullptr) {
  V64 gn1, gn2;
  const V64 gam1 = param64(vecs + 2 * D, g);
  {
    const V64 bet1 = param64(vecs + 3 * D, g);
    V64 gam2, bet2;
    if (SLIM) { gam2 = param64(vecs + 4 * D, g); bet2 = param64(vecs + 5 * D, g); }
    CHG_EV(ft) {
      const f32x4 n1 = s.xh1.t[ft] * gam1.t[ft] + bet1.t[ft];
      // SLIM: both gate activations are recomputed here (32 transcendental pairs for 32 registers over the bond-weight phase: a
      // spilled register costs the BondConv adjoints more than that)
      const f32x4 sg = SLIM ? sigmoid4(n1) : s.sg1.t[ft];
      const f32x4 a2 = SLIM ? sigmoid4(s.xh2.t[ft] * gam2.t[ft] + bet2.t[ft]) : s.a2.t[ft];
      const f32x4 ga = gy.t[ft] * a2 * sg;
      gn1.t[ft] = ga * (1.0f + n1 * (1.0f - sg));     // d silu
      gn2.t[ft] = ga * n1 * (1.0f - a2);
    }
  }
  if (TRAIN) {   // LayerNorm affine: d gamma = sum_rows gn * xhat, d beta = sum_rows gn  (gn = adjoint of the LayerNorm output)
    V64 t;
    CHG_EV(ft) t.t[ft] = gn1.t[ft] * s.xh1.t[ft];
    tile_colsum2(tt, j, g, t, gn1, tt->ln[0], tt->ln[1]);
    CHG_EV(ft) t.t[ft] = gn2.t[ft] * s.xh2.t[ft];
    tile_colsum2(tt, j, g, t, gn2, tt->ln[2], tt->ln[3]);
  }
  ln_backward(gn1, gam1, s.xh1, s.rstd1);
  ln_backward(gn2, param64(vecs + 4 * D, g), s.xh2, s.rstd2);
  if (TRAIN && tt->grow) {   // adjoint of the second-layer pre-activations (core | gate): A operand of dW2, column sums = d b2
    write_dl<VT>(tt->grow, g, gn1.t);
    write_dl<VT>(tt->grow + D, g, gn2.t);
  }
  if (HIDDEN) {
    gzc = zero64();
    gzg = zero64();
    if (SPLIT == 2) {
      gemm_rm<VT, VT, true, true>(gzc.t, reinterpret_cast<const _Float16*>(W2c), D, D, gn1.t, j, g, j + 16 * g);
      gemm_rm<VT, VT, true, true>(gzg.t, reinterpret_cast<const _Float16*>(W2g), D, D, gn2.t, j, g, j + 16 * g);
    } else if (SPLIT == 1) {
      gemm_split<VT, VT, true>(gzc.t, reinterpret_cast<const h16x8*>(W2c), D, gn1.t, j, g);
      gemm_split<VT, VT, true>(gzg.t, reinterpret_cast<const h16x8*>(W2g), D, gn2.t, j, g);
    } else {
      gemm_dl_t<VT, VT>(gzc.t, W2c, WS, gn1.t, j, g);
      gemm_dl_t<VT, VT>(gzg.t, W2g, WS, gn2.t, j, g);
    }
    CHG_EV(ft) {
      gzc.t[ft] = gzc.t[ft] * dzc.t[ft];
      gzg.t[ft] = gzg.t[ft] * dzg.t[ft];
    }
  } else {
    gzc = gn1;
    gzg = gn2;
  }
}

// =============================================================================================
// k_rows_gemm:  Y[o(r), n] (+)= sum_k X[i(r), k] * Wt[n][k] (+ bias[n]) (+ resid[o(r), n])
//   K in {64,128}, NOUT in {64,128}; i(r) / o(r) optional row index maps (null = identity).
// =============================================================================================
struct RowsGemm {
  const float* X;
  int ldx;
  const int* in_idx;
  const float* Wt;     // [NOUT][K] row-major (global)
  const float* bias;   // [NOUT] or null
  const float* resid;  // rows of ld = ldr, indexed like Y, or null
  int ldr;
  float* Y;
  int ldy;
  const int* out_idx;
  int rows;
  int accumulate;      // Y += result (rows unique -> plain read-modify-write)
  // two-part form (PARTS = 2): a second weight block Wt2 and EITHER a second output column block
  // (y2_off > 0: Y[:, y2_off:] = X . Wt2^T, same X, no bias) OR a second input column block
  // (x2_off > 0: Y (+)= X . Wt^T + X[:, x2_off:] . Wt2^T).  One launch and one pass over the rows
  // for the two halves of a 256-wide table.
  const float* Wt2;
  int x2_off, y2_off;
  // small row counts (MD, single structures): the output columns are split over gridDim.y workgroups of a narrow instance
  // (NOUT = block width) -- block y contracts columns NOUT y .. NOUT y + NOUT - 1.  PARTS = 1: weights from Wt for y < blocks1, then
  // from Wt2 (the two halves of a 256-wide table; the bias covers the Wt blocks).  PARTS = 2 (two input column blocks): the same
  // output columns of Wt and Wt2.  A few hundred rows then occupy 4-16x the CUs, each staging and contracting its share only.
  int col_blocks, blocks1;
};

template <int K, int NOUT, int PARTS = 1>
__device__ __forceinline__ void rows_gemm_body(RowsGemm p, int cb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (p.col_blocks > 1) {   // column-split form: this workgroup's NOUT output columns
    const bool first = PARTS == 2 || cb < p.blocks1;
    if (PARTS == 2) {
      p.Wt += (size_t)cb * NOUT * K;
      p.Wt2 += (size_t)cb * NOUT * K;
    } else {
      p.Wt = first ? p.Wt + (size_t)cb * NOUT * K : p.Wt2 + (size_t)(cb - p.blocks1) * NOUT * K;
    }
    p.bias = (p.bias && first) ? p.bias + cb * NOUT : nullptr;
    p.Y += cb * NOUT;
    if (p.resid) p.resid += cb * NOUT;
  }
  constexpr int KS = K + PAD, KT = K / 16, NFT = NOUT / 16;
  constexpr int XS = (K > NOUT ? K : NOUT) + PAD;  // tile stride: holds X (K wide) then Y (NOUT wide)
  // Full-width instances (NOUT a multiple of 64) contract in the split form of the tile kernels (mfma_split.h: 3 f16 MFMAs per f32
  // product on hi / lo halves, every row scaled by a power of two first -- exact, any magnitude, so gradient rows of 1e-7 and feature
  // rows of 1e3 alike keep 22 bits): the f32 MFMA kept the matrix pipe of these HBM-streaming GEMMs 45 % busy (gemm_GQ: 29 GFLOP per
  // launch = 0.18 of its 0.41 ms).  The 16-column instances of the small-batch form keep the f32 MFMA.
  constexpr bool SPLIT = NOUT % 64 == 0;
  float* W = smem;                          // [PARTS][NOUT][KS]  (SPLIT: [PARTS] split images at the same offsets)
  float* bias = W + PARTS * NOUT * KS;      // [NOUT]
  float* tiles = bias + NOUT;               // [WAVES][16][XS]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  if constexpr (SPLIT) {
    stage_split<false>(reinterpret_cast<h16x8*>(W), p.Wt, NOUT, K, tid, BLOCK);
    if (PARTS == 2) stage_split<false>(reinterpret_cast<h16x8*>(W + NOUT * KS), p.Wt2, NOUT, K, tid, BLOCK);
  } else {
    stage_weights(W, p.Wt, NOUT, K, tid);
    if (PARTS == 2) stage_weights(W + NOUT * KS, p.Wt2, NOUT, K, tid);
  }
  auto contract = [&](f32x4 (&acc)[NFT], int part, const f32x4 (&x)[KT]) {
    if constexpr (SPLIT) gemm_split<KT, NFT, true>(acc, reinterpret_cast<const h16x8*>(W + part * NOUT * KS), NOUT, x, j, g);
    else gemm_dl<KT, NFT>(acc, W + part * NOUT * KS, KS, x, j, g);
  };
  for (int idx = tid; idx < NOUT; idx += BLOCK) bias[idx] = p.bias ? p.bias[idx] : 0.f;
  __syncthreads();
  float* T = tiles + wave * TILE_ROWS * XS;
  const int ntiles = (p.rows + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range(ntiles, tb, te);
  const bool split_in = PARTS == 2 && p.x2_off > 0;   // else (PARTS == 2): split output
  if constexpr (PARTS == 1 && K == 128) {
    // 128 -> 64 form (gemm_GQ: read-dominated, Eu rows, HBM-bound): software pipeline over tiles -- the X rows of tile t+1 and
    // the old Y rows of tile t are in flight while tile t is contracted (0.46 -> 0.38 ms, 3.9 -> 4.8 TB/s).  The
    // write-dominated 64 -> 128 form (gemm_Q) measured SLOWER with the same pipeline (0.37 -> 0.44 ms) and keeps the plain loop.
    constexpr int LPR = K / 4, RPS = 64 / LPR, NV = TILE_ROWS / RPS;
    constexpr int LPO = NOUT / 4, RPO = 64 / LPO, NO = TILE_ROWS / RPO;
    const int subx = lane / LPR, tx = lane % LPR, suby = lane / LPO, ty = lane % LPO;
    auto rows_of = [&](int tile, int& nvalid, int& in_row, int& out_row) {
      const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
      nvalid = min(TILE_ROWS, p.rows - row0);
      const int rr_ = nvalid > 0 ? row0 + (j < nvalid ? j : 0) : 0;
      in_row = p.in_idx ? p.in_idx[rr_] : rr_;
      out_row = p.out_idx ? p.out_idx[rr_] : rr_;
    };
    auto issue_x = [&](int in_row, f32x4 (&v)[NV]) {
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const int r = __shfl(in_row, RPS * it + subx);
        v[it] = *reinterpret_cast<const f32x4*>(p.X + (size_t)r * p.ldx + 4 * tx);
      }
    };
    f32x4 vn[NV];
    int nvalid_n = 0, in_n = 0, out_n = 0;
    if (tb < te) {
      rows_of(tb, nvalid_n, in_n, out_n);
      if (nvalid_n > 0) issue_x(in_n, vn);
    }
    for (int tile = tb; tile < te; ++tile) {
      const int nvalid = nvalid_n, out_row = out_n;
      if (nvalid <= 0) break;                      // waves past the end of the last tile (tiles of a wave only grow)
#pragma unroll
      for (int it = 0; it < NV; ++it) *reinterpret_cast<f32x4*>(T + (RPS * it + subx) * XS + 4 * tx) = vn[it];
      __builtin_amdgcn_wave_barrier();
      f32x4 x[KT];
      read_dl<KT>(T + j * XS, g, x);
      __builtin_amdgcn_wave_barrier();
      if (tile + 1 < te) {
        rows_of(tile + 1, nvalid_n, in_n, out_n);
        if (nvalid_n > 0) issue_x(in_n, vn);
      } else {
        nvalid_n = 0;
      }
      const bool rmw = p.accumulate || p.resid;    // uniform: the plain-store form keeps no old rows in registers
      f32x4 old[NO];
      if (rmw) {
#pragma unroll
        for (int it = 0; it < NO; ++it) {
          const int r = __shfl(out_row, RPO * it + suby);
          old[it] = zero4();
          if (RPO * it + suby < nvalid) {
            if (p.accumulate) old[it] = *reinterpret_cast<const f32x4*>(p.Y + (size_t)r * p.ldy + 4 * ty);
            if (p.resid) old[it] += *reinterpret_cast<const f32x4*>(p.resid + (size_t)r * p.ldr + 4 * ty);
          }
        }
      }
      f32x4 acc[NFT];
      read_dl<NFT>(bias, g, acc);
      contract(acc, 0, x);
      write_dl<NFT>(T + j * XS, g, acc);
      __builtin_amdgcn_wave_barrier();
      // all sums, then the stores: a wait for an old row between the conditional (uncounted) stores is a wait for those stores
      f32x4 ov[NO];
      int orow[NO];
#pragma unroll
      for (int it = 0; it < NO; ++it) {
        const int rr = RPO * it + suby;
        orow[it] = __shfl(out_row, rr);
        ov[it] = *reinterpret_cast<const f32x4*>(T + rr * XS + 4 * ty);
        if (rmw) ov[it] += old[it];
      }
#pragma unroll
      for (int it = 0; it < NO; ++it) asm volatile("" : "+v"(ov[it]));
#pragma unroll
      for (int it = 0; it < NO; ++it)
        if (RPO * it + suby < nvalid) *reinterpret_cast<f32x4*>(p.Y + (size_t)orow[it] * p.ldy + 4 * ty) = ov[it];
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  for (int tile = tb; tile < te; ++tile) {
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.rows - row0);
    if (nvalid <= 0) continue;
    const int rr_ = row0 + (j < nvalid ? j : 0);
    const int in_row = p.in_idx ? p.in_idx[rr_] : rr_;
    const int out_row = p.out_idx ? p.out_idx[rr_] : rr_;
    constexpr int LPR = K / 4, RPS = 64 / LPR;   // X tile -> LDS, K/4 lanes per row
    constexpr int LPO = NOUT / 4, RPO = 64 / LPO; // Y tile -> global, NOUT/4 lanes per row
    f32x4 x[KT];
    f32x4 acc[NFT];
    read_dl<NFT>(bias, g, acc);
#pragma unroll
    for (int part = 0; part < PARTS; ++part) {
      if (part == 0 || split_in) {
        const float* X = p.X + (part ? p.x2_off : 0);
        const int sub = lane / LPR, t = lane % LPR;
        f32x4 v[TILE_ROWS / RPS];
#pragma unroll
        for (int it = 0; it < TILE_ROWS / RPS; ++it) {
          const int r = __shfl(in_row, RPS * it + sub);
          v[it] = *reinterpret_cast<const f32x4*>(X + (size_t)r * p.ldx + 4 * t);
        }
#pragma unroll
        for (int it = 0; it < TILE_ROWS / RPS; ++it) *reinterpret_cast<f32x4*>(T + (RPS * it + sub) * XS + 4 * t) = v[it];
        __builtin_amdgcn_wave_barrier();
        read_dl<KT>(T + j * XS, g, x);
        __builtin_amdgcn_wave_barrier();
      }
      if (part == 1 && !split_in) {
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) acc[ft] = zero4();
      }
      contract(acc, part, x);
      if (part == PARTS - 1 || !split_in) {
        write_dl<NFT>(T + j * XS, g, acc);
        __builtin_amdgcn_wave_barrier();
        float* Y = p.Y + (part ? p.y2_off : 0);
        const int sub = lane / LPO, t = lane % LPO;
        // every old / residual row of the tile is requested first (rows past nvalid index a valid row: harmless reads), summed, and
        // only then stored: row by row, each load was waited for in place -- behind the previous row's store
        constexpr int NOI = TILE_ROWS / RPO;
        f32x4 ov[NOI];
        f32x4* odst[NOI];
#pragma unroll
        for (int it = 0; it < NOI; ++it) {
          const int r = __shfl(out_row, RPO * it + sub);
          odst[it] = reinterpret_cast<f32x4*>(Y + (size_t)r * p.ldy + 4 * t);
          ov[it] = zero4();
          if (p.resid) ov[it] = *reinterpret_cast<const f32x4*>(p.resid + (size_t)r * p.ldr + 4 * t);
          if (p.accumulate) ov[it] += *odst[it];
        }
#pragma unroll
        for (int it = 0; it < NOI; ++it) {
          ov[it] += *reinterpret_cast<const f32x4*>(T + (RPO * it + sub) * XS + 4 * t);
          asm volatile("" : "+v"(ov[it]));
        }
#pragma unroll
        for (int it = 0; it < NOI; ++it)
          if (RPO * it + sub < nvalid) *odst[it] = ov[it];
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

template <int K, int NOUT, int PARTS = 1>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_rows_gemm(RowsGemm p) {
  rows_gemm_body<K, NOUT, PARTS>(p, blockIdx.y);
}

// Several independent column-split products in one launch (small batches): block y of the grid belongs to the problem whose
// col_blocks range holds it; gridDim.x covers the longest row count.  Each in the one- or the two-input-block form (K = 128:
// x2_off > 0 picks PARTS = 2): the S and R tables of an angle layer with the next AtomConv's P table, the table gradients of an
// angle layer (GR | GS) and of an AtomConv (GP | GQ).
struct RowsGemmN { RowsGemm p[3]; int n; };
template <int K, int NOUT>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_rows_gemm_multi(RowsGemmN g) {
  int y = blockIdx.y;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i >= g.n) return;
    if (y < g.p[i].col_blocks) {
      if (K == 128 && g.p[i].x2_off > 0) rows_gemm_body<K, NOUT, (K == 128 ? 2 : 1)>(g.p[i], y);
      else rows_gemm_body<K, NOUT, 1>(g.p[i], y);
      return;
    }
    y -= g.p[i].col_blocks;
  }
}

template <int K, int NOUT, int PARTS = 1>
constexpr size_t rows_gemm_lds() {
  return sizeof(float) * (PARTS * NOUT * (K + PAD) + NOUT + WAVES * TILE_ROWS * ((K > NOUT ? K : NOUT) + PAD));
}

// Phase timing (diagnostic builds, -DCHG_PHASE_TIMING): s_memtime deltas between the phases of a tile,
// summed per wave and added to p.phase[base + i] at the end; read back with chg_debug_fetch("phase")
// (tests/gpu_phase_probe.py).
#ifdef CHG_PHASE_TIMING
#define PH_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime(); bool ph_first = true; float ph_acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ph_acc1[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define PH_TILE(first) ph_first = (first);
#define PH(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (ph_first) ph_acc1[i] += (float)(t_ - ph_t); else ph_acc[i] += (float)(t_ - ph_t); ph_t = t_; __builtin_amdgcn_sched_barrier(0); }
// p.phase[kernel 0..3][later / first tile][slot 0..9][wave 0..PH_WAVES): every wave adds to its own floats (plain read-modify-write: the
// first version met in 20 shared addresses, and 2048 waves' same-address atomics -- ~110 ns each -- cost more than the kernel)
constexpr int PH_WAVES = 4096;
// PH_START (first statement of a kernel) / PH_FLUSH: the wave's entry and exit stamps (low 32 bits, bit pattern) of the LAST launch of a kernel
// go to the otherwise unused kernel slot 6: stamps[kernel][entry / exit][wave] -- when the waves of a launch really start and end.
#define PH_START const unsigned long long ph_t0 = __builtin_amdgcn_s_memtime();
#define PH_FLUSH(base) { const int gw_ = blockIdx.x * (blockDim.x >> 6) + wave; if (lane == 0 && gw_ < PH_WAVES) { for (int i_ = 0; i_ < 10; ++i_) { \
  p.phase[((size_t)((base) / 10 * 2 + 0) * 10 + i_) * PH_WAVES + gw_] += ph_acc[i_]; p.phase[((size_t)((base) / 10 * 2 + 1) * 10 + i_) * PH_WAVES + gw_] += ph_acc1[i_]; } \
  p.phase[((size_t)6 * 20 + (base) / 10 * 2 + 0) * PH_WAVES + gw_] = __uint_as_float((unsigned)ph_t0); \
  p.phase[((size_t)6 * 20 + (base) / 10 * 2 + 1) * PH_WAVES + gw_] = __uint_as_float((unsigned)__builtin_amdgcn_s_memtime()); } }
#else
#define PH_DECL
#define PH_START
#define PH_TILE(first)
#define PH(i)
#define PH_FLUSH(base)
#endif

// =============================================================================================
// AtomConv
// =============================================================================================
struct AtomConvArgs {
  const float* P;      // [N,256]  cols 0..127: centre partial (+b1), 128..255: neighbour partial
  const float* Q;      // [Eu,128] bond partial
  const float* wag;    // [Eu,64]  smooth bond weights (atom graph)
  const int *e_center, *e_nbr, *e_d2u;
  int n_edges;
  int interleave;      // XCD-local interleaved tile sequence (mfma_tile.h wave_tile_seq) or contiguous per-wave ranges
  GatedW gw;
  // forward: bond partial contracted in the kernel (no gemm_Q / gemm_Qnode launches):  Q[k] = hb[k] . W_bond^T (+ q_bias)
  const float *hb0, *hbc;      // [Eu,64] embedding rows; [Eb,64] layer features of the bond-graph nodes (null: every bond uses hb0)
  const int* u_bnode;          // [Eu] compact node index or -1
  const float *w_bond, *q_bias;   // [128][64]; [128] shift of the bonds outside the bond graph (0.2.0 checkpoints) or null
  float* Qout;                 // forward: [Eu,128] the partial stored as a table for the adjoint sweep, or null (energy-only tasks)
  const float* image;          // prebuilt weight block of the kernel's LDS (k_atomconv_image); the TRAIN adjoint stages from the fp32 weights
  float* agg;          // fwd out: [N,64], zeroed by the caller
  // backward only
  const float* GA;     // [N,64] dE/d agg
  float* GP;           // [N,256] zeroed: grads of the two partials
  float* GQ;           // [Eu,128] zeroed
  float* Gwag;         // [Eu,64] accumulated over layers
  int first_wag;       // this launch is the first writer of Gwag in the sweep: store, do not read (the buffer is not zeroed)
  float* phase;        // CHG_PHASE_TIMING builds only (kernel slots 6 = forward, 7 = adjoint)
  float* Gb;           // fused form of the adjoint (k_atomconv_bwd<false, NW, true>): dE/d h_bond rows [Eu,64], owned by the tile
  int gb_accumulate;   // ... += (layers below the last) or = (the sweep's first AtomConv)
  // training (k_atomconv_bwd<true>) only
  float* dumpG;        // [Ed,128] pair order: adjoint of the second-layer pre-activations (core | gate)
  float* dumpH;        // [Ed,128] pair order: hidden activations (core | gate)
  float* g_ln;         // [4][64] gradient of ln1_g, ln1_b, ln2_g, ln2_b (accumulated with atomics)
};

// LDS: split images of W2c, W2g (the TRAIN adjoint: and of their transposes), the gated-MLP vectors, one tile per wave.
// FUSEQ (inference): plus the bond block W_bond -- forward: its split image; adjoint: W2c, W2g, W_bond as row-major images (each
// serves both directions) -- and two vector slots for q_bias.
constexpr int AC_VEC_SLOTS = VEC_SLOTS + 2;
template <int NW = WAVES, bool BWD = false, bool FUSEQ = false>
constexpr size_t atomconv_lds() {
  if (FUSEQ) return 16 * (size_t)(2 * IMG64 + IMG128) + sizeof(float) * (AC_VEC_SLOTS * D + NW * TILE_FLOATS);
  return 16 * (size_t)(BWD ? 4 : 2) * IMG64 + sizeof(float) * (VEC_SLOTS * D + NW * TILE_FLOATS);
}

// The weight block at the start of the AtomConv kernels' LDS, as a function of the weights alone: staged in the kernel, or built once
// per weight upload into global memory (k_atomconv_image) and copied (stage_image).
constexpr int ac_fwd_image_floats() { return 4 * (2 * IMG64 + IMG128) + AC_VEC_SLOTS * D; }
constexpr int ac_bwd_image_floats() { return 4 * 4 * IMG64 + VEC_SLOTS * D; }
__device__ __forceinline__ void atomconv_fwd_stage(float* base, const AtomConvArgs& p, int tid, int nthreads) {
  h16x8* I2c = reinterpret_cast<h16x8*>(base);
  h16x8* I2g = I2c + IMG64;
  h16x8* Ib = I2g + IMG64;
  float* vecs = reinterpret_cast<float*>(Ib + IMG128);
  stage_split<false>(I2c, p.gw.w2c, D, D, tid, nthreads);
  stage_split<false>(I2g, p.gw.w2g, D, D, tid, nthreads);
  stage_split<false>(Ib, p.w_bond, 2 * D, D, tid, nthreads);
  stage_gated_vecs(vecs, p.gw, true, tid);
  for (int q = tid; q < 2 * D; q += nthreads) vecs[VEC_SLOTS * D + q] = p.q_bias ? p.q_bias[q] : 0.f;
}
__device__ __forceinline__ void atomconv_bwd_stage(float* base, const AtomConvArgs& p, int tid, int nthreads) {
  h16x8* I2c = reinterpret_cast<h16x8*>(base);
  h16x8* I2g = I2c + IMG64;
  h16x8* I2cT = I2g + IMG64;
  h16x8* I2gT = I2cT + IMG64;
  float* vecs = reinterpret_cast<float*>(I2gT + IMG64);
  stage_split<false>(I2c, p.gw.w2c, D, D, tid, nthreads);
  stage_split<false>(I2g, p.gw.w2g, D, D, tid, nthreads);
  stage_split<true>(I2cT, p.gw.w2c, D, D, tid, nthreads);
  stage_split<true>(I2gT, p.gw.w2g, D, D, tid, nthreads);
  stage_gated_vecs(vecs, p.gw, true, tid);
}
// Block of the fused adjoint (k_atomconv_bwd<false, true>): ONE row-major image per hidden matrix serves both directions
// (mfma_split.h gemm_rm), which leaves room for the W_bond^T image next to eight waves' tiles:
// [W2c rm | W2g rm | vectors | W_bond^T split].
constexpr int AC_RM_IMG = (int)(rm_image_bytes(D, D) / 4);                                       // floats
constexpr int ac_bwd_rm_image_floats() { return 2 * AC_RM_IMG + VEC_SLOTS * D + 4 * IMG128; }
__device__ __forceinline__ void atomconv_bwd_stage_rm(float* base, const AtomConvArgs& p, int tid, int nthreads) {
  stage_rm(reinterpret_cast<_Float16*>(base), p.gw.w2c, D, D, tid, nthreads);
  stage_rm(reinterpret_cast<_Float16*>(base + AC_RM_IMG), p.gw.w2g, D, D, tid, nthreads);
  float* vecs = base + 2 * AC_RM_IMG;
  stage_gated_vecs(vecs, p.gw, true, tid);
  stage_split<true>(reinterpret_cast<h16x8*>(vecs + VEC_SLOTS * D), p.w_bond, 2 * D, D, tid, nthreads);
}
static __global__ __launch_bounds__(BLOCK) void k_atomconv_image_rm(AtomConvArgs p, float* out) { atomconv_bwd_stage_rm(out, p, threadIdx.x, BLOCK); }

template <bool BWD>
__global__ __launch_bounds__(BLOCK) void k_atomconv_image(AtomConvArgs p, float* out) {
  if (BWD) atomconv_bwd_stage(out, p, threadIdx.x, BLOCK);
  else atomconv_fwd_stage(out, p, threadIdx.x, BLOCK);
}

// Row of h_bond that feeds bond k's partial, as an offset from hb0 (floats).  bn = bond_node(k) >= 0: the row carries layer features
// (no q_bias).  In two steps: the node index is REQUESTED two tiles ahead and only turned into an offset when the gather is issued --
// consumed at the load (compare + select) it made every tile wait for it and, the memory counter being in order, for the gathers of
// the next tile issued just before.
__device__ __forceinline__ int bond_node(const AtomConvArgs& p, int k) { return p.hbc ? p.u_bnode[k] : -1; }
__device__ __forceinline__ long bond_row_offset(const AtomConvArgs& p, int k, int bn) {
  return bn >= 0 ? (p.hbc - p.hb0) + (long)bn * D : (long)k * D;
}

// NW waves per workgroup.  The bond partial Q[k] = hb[k] . W_bond^T is contracted here (one more split contraction per tile)
// instead of being read from a table: no gemm_Q / gemm_Qnode launches, 256 B instead of 512 B read per bond.
template <int NW>
__global__ __launch_bounds__(64 * NW) CHG_TWO_WAVES void k_atomconv_fwd(AtomConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  h16x8* I2c = reinterpret_cast<h16x8*>(smem);
  h16x8* I2g = I2c + IMG64;
  h16x8* Ib = I2g + IMG64;
  float* vecs = reinterpret_cast<float*>(Ib + IMG128);
  float* tiles = vecs + AC_VEC_SLOTS * D;
  const float* W2c = reinterpret_cast<const float*>(I2c);
  const float* W2g = reinterpret_cast<const float*>(I2g);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  float* T = tiles + wave * TILE_FLOATS;
  float* Trow = T + j * TS;
  const int ntiles = (p.n_edges + TILE_ROWS - 1) / TILE_ROWS;     // wave-tiles: this wave's own sequence (mfma_tile.h wave_tile_seq)
  const TileSeq ts = wave_tile_seq(ntiles, NW, wave, p.interleave);
  // Software pipeline over tiles: the row gather of tile t+1 is issued before tile t's MFMA / VALU phase and committed to LDS
  // after it; the indices run two tiles ahead.  (SQ_WAIT_ANY was 37 % of wave time with the gather issued and awaited in place.)
  const int tstride = TILE_ROWS;
  auto row_of = [&](int v) { return max(0, min(ts.at(v) * tstride + j, p.n_edges - 1)); };   // v-th tile of this wave (clamped past its end)
  GatherPH gr;
  V64 wv_nx;
  int c_nx = 0, n_nx = 0, c_n2 = 0, n_n2 = 0;
  int bn_nx = -1, bn_n2 = -1;          // bond-graph node (or -1) of the rows gathered next / after that
  if (ts.count > 0) {
    const int r0 = row_of(0);
    c_nx = p.e_center[r0]; n_nx = p.e_nbr[r0];
    bn_nx = bond_node(p, r0 >> 1);
    gather_issue_ph(gr, p.P, c_nx, p.P + 2 * D, n_nx, 4 * D, 4 * D, p.hb0, bond_row_offset(p, r0 >> 1, bn_nx), lane);
    read_dl_g<VT>(p.wag, (unsigned)(r0 >> 1), D, g, wv_nx.t);
    const int r1 = row_of(1);
    c_n2 = p.e_center[r1]; n_n2 = p.e_nbr[r1];
    bn_n2 = bond_node(p, r1 >> 1);
  }
  // the first tile's indices and gathers are requested BEFORE the weights are staged: two dependent memory round trips land under
  // the staging (small batches -- MD -- run one or two tiles per wave, and the prologue was a tenth of the launch)
  stage_image<ac_fwd_image_floats() / 4, 64 * NW>(smem, p.image, tid);
  __syncthreads();
  for (int v = 0; v < ts.count; ++v) {
    const int row0 = ts.at(v) * tstride;
    const int nvalid = min(TILE_ROWS, p.n_edges - row0);   // even (pair order)
    // bond-pair order (rows 2k, 2k+1 = the two directions of bond k): hb[k] and w_ag[k] are fetched once per bond (the second
    // row's copy comes from L1)
    const int c = c_nx;
    const bool node = bn_nx >= 0;
    const V64 wv = wv_nx;
    gather_commit_h(gr, T, TS, lane);      // hb rows first: the tile's left half is reused by the table sums below
    __builtin_amdgcn_wave_barrier();
    V64 x;
    read_dl<VT>(Trow, g, x.t);
    __builtin_amdgcn_wave_barrier();
    gather_commit_p(gr, T, TS, lane);
    __builtin_amdgcn_wave_barrier();
    c_nx = c_n2; n_nx = n_n2; bn_nx = bn_n2;
    if (v + 1 < ts.count) {
      const int r1 = row_of(v + 1);
      gather_issue_ph(gr, p.P, c_nx, p.P + 2 * D, n_nx, 4 * D, 4 * D, p.hb0, bond_row_offset(p, r1 >> 1, bn_nx), lane);
      read_dl_g<VT>(p.wag, (unsigned)(r1 >> 1), D, g, wv_nx.t);
      const int r2 = row_of(v + 2);
      c_n2 = p.e_center[r2]; n_n2 = p.e_nbr[r2];
      bn_n2 = bond_node(p, r2 >> 1);
    }
    // (every tile of a wave_tile_seq is a real tile: nvalid >= 1.  An early `continue` here, never taken, still gave the loop latch a
    // path on which the requests above are pending -- and the waits the compiler placed there for it stood behind this tile's atomics)
    f32x4 z[2 * VT];
    if (!node) {   // bonds outside the bond graph: constant shift of the 0.2.0 checkpoints (zeros otherwise)
      read_dl<2 * VT>(vecs + VEC_SLOTS * D, g, z);
    } else {
#pragma unroll
      for (int ft = 0; ft < 2 * VT; ++ft) z[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    gemm_split<VT, 2 * VT, false, true>(z, Ib, 2 * D, x.t, j, g);
    if (p.Qout && !(j & 1) && j < nvalid) {   // the adjoint sweep gathers the partial as a table: one row per bond, from the even rows
      float* q = p.Qout + (size_t)((row0 + j) >> 1) * 2 * D + 4 * g;
#pragma unroll
      for (int ft = 0; ft < 2 * VT; ++ft) *reinterpret_cast<f32x4*>(q + 16 * ft) = z[ft];
    }
    {
      f32x4 ps[2 * VT];
      read_dl<2 * VT>(Trow, g, ps);
#pragma unroll
      for (int ft = 0; ft < 2 * VT; ++ft) z[ft] += ps[ft];
    }
    V64 zc{{z[0], z[1], z[2], z[3]}}, zg{{z[4], z[5], z[6], z[7]}};
    GatedState s;
    V64 y;
    gated_forward<true, false, false, 1, true>(zc, zg, W2c, W2g, vecs, j, g, s, y);
    // everything requested for the next tiles is taken here, before this tile's atomics (mfma_tile.h gather_take)
    gather_take(gr);
    asm volatile("" : "+v"(c_n2), "+v"(n_n2), "+v"(bn_n2));
    CHG_EV(ft) asm volatile("" : "+v"(wv_nx.t[ft]));
    __builtin_amdgcn_wave_barrier();
    V64 m;
    CHG_EV(ft) m.t[ft] = y.t[ft] * wv.t[ft];
    write_dl<VT>(Trow, g, m.t);
    __builtin_amdgcn_wave_barrier();
    {  // even rows: centre c1 is nondecreasing in k; odd rows: c2 is nondecreasing within one c1 (images of
       // one neighbour are adjacent) -> run sums on both sides, one atomic row per run
      float acc1 = 0.f, acc2 = 0.f;
      int cur1 = __builtin_amdgcn_readlane(c, 0), cur2 = __builtin_amdgcn_readlane(c, 1);
#pragma unroll
      for (int b = 0; b < TILE_ROWS / 2; ++b) {
        if (2 * b < nvalid) {
          const int c1 = __builtin_amdgcn_readlane(c, 2 * b), c2 = __builtin_amdgcn_readlane(c, 2 * b + 1);
          if (c1 != cur1) {
            tile_atomic_add(grow<float>(p.agg, (unsigned)cur1, D, lane), acc1);
            acc1 = 0.f;
            cur1 = c1;
          }
          if (c2 != cur2) {
            tile_atomic_add(grow<float>(p.agg, (unsigned)cur2, D, lane), acc2);
            acc2 = 0.f;
            cur2 = c2;
          }
          acc1 += T[(2 * b) * TS + lane];
          acc2 += T[(2 * b + 1) * TS + lane];
        }
      }
      tile_atomic_add(grow<float>(p.agg, (unsigned)cur1, D, lane), acc1);
      tile_atomic_add(grow<float>(p.agg, (unsigned)cur2, D, lane), acc2);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Adjoint of AtomConv over the PAIR-ORDERED edge list: rows 2k and 2k+1 are the two directions of
// undirected bond k (pack.py: p_center / p_nbr).  Both directions of a bond then sit in one tile, so
//   dE/dQ[k], dE/dw_ag[k]        = sum of the two rows            -> plain stores, no atomics
//   dE/dP[c1] (c1 = centre of the even row, nondecreasing in k)    -> segmented column sums
//   dE/dP[c2] (nondecreasing inside one c1: images of a neighbour)  -> run sums, one atomic row per run
// (profiles/r01 notes: row atomics were 55 % of the centre-ordered version of this kernel).
//
// Scatter of one AtomConv-adjoint tile (gz rows in LDS, pair order).  Lane owns columns
// (lane, lane+64) of the 128-wide rows.  dE/dQ[k] is a plain store (the tile owns bond k).
template <bool STORE_GQ = true>
__device__ __forceinline__ void acbwd_scatter(const float* T, int c, int nvalid, int k0, const AtomConvArgs& p, int lane) {
  // Both sides leave the tile as run sums: c1 is sorted along the pair order, and within one c1 the
  // bonds are ordered by c2 (then image), so periodic images of one neighbour are adjacent too
  // (small cells: ~3.5 distinct c2 per 8 bonds).  Fewer atomic rows matter: each is paid at the memory side.
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
  int cur1 = __builtin_amdgcn_readlane(c, 0), cur2 = __builtin_amdgcn_readlane(c, 1);
#pragma unroll
  for (int b = 0; b < TILE_ROWS / 2; ++b) {
    if (2 * b < nvalid) {
      const float e0 = T[(2 * b) * TS + lane], e1 = T[(2 * b) * TS + 64 + lane];           // gz of direction c1 -> c2
      const float o0 = T[(2 * b + 1) * TS + lane], o1 = T[(2 * b + 1) * TS + 64 + lane];   // gz of direction c2 -> c1
      if (STORE_GQ) {
        float* q = p.GQ + (size_t)(k0 + b) * 2 * D + lane;
        q[0] = e0 + o0;
        q[64] = e1 + o1;
      }
      const int c1 = __builtin_amdgcn_readlane(c, 2 * b), c2 = __builtin_amdgcn_readlane(c, 2 * b + 1);
      if (c1 != cur1) {
        float* d = p.GP + (size_t)cur1 * 4 * D + lane;
        tile_atomic_add(d, a1[0]); tile_atomic_add(d + 64, a1[1]); tile_atomic_add(d + 128, a1[2]); tile_atomic_add(d + 192, a1[3]);
        a1[0] = a1[1] = a1[2] = a1[3] = 0.f;
        cur1 = c1;
      }
      if (c2 != cur2) {
        float* d = p.GP + (size_t)cur2 * 4 * D + lane;
        tile_atomic_add(d, a2[0]); tile_atomic_add(d + 64, a2[1]); tile_atomic_add(d + 128, a2[2]); tile_atomic_add(d + 192, a2[3]);
        a2[0] = a2[1] = a2[2] = a2[3] = 0.f;
        cur2 = c2;
      }
      a1[0] += e0; a1[1] += e1; a1[2] += o0; a1[3] += o1;   // atom c1: centre part <- even row, neighbour part <- odd row
      a2[0] += o0; a2[1] += o1; a2[2] += e0; a2[3] += e1;   // atom c2: centre part <- odd row, neighbour part <- even row
    }
  }
  float* d1 = p.GP + (size_t)cur1 * 4 * D + lane;
  tile_atomic_add(d1, a1[0]); tile_atomic_add(d1 + 64, a1[1]); tile_atomic_add(d1 + 128, a1[2]); tile_atomic_add(d1 + 192, a1[3]);
  float* d2 = p.GP + (size_t)cur2 * 4 * D + lane;
  tile_atomic_add(d2, a2[0]); tile_atomic_add(d2 + 64, a2[1]); tile_atomic_add(d2 + 128, a2[2]); tile_atomic_add(d2 + 192, a2[3]);
}

// FUSE_GQ: the tile also contracts its dE/dQ rows with W_bond (128 -> 64, split form) and updates the dE/d h_bond rows of its bonds
// itself -- the rows are in registers (the pair sum is one lane swap), so the [Eu,128] table is neither written nor read back by a
// row GEMM (gemm_GQ: 1.4 ms per headline step at the HBM rate).  The 32 KB image of W_bond^T fits because the hidden layer then
// uses the row-major images (18 KB per matrix for both directions instead of 32 KB; with the split images and 7 waves per workgroup
// the kernel was 8 % slower: profiles/r04_experiments.md section 12).
template <bool TRAIN, bool FUSE_GQ = false>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_atomconv_bwd(AtomConvArgs p) {
  static_assert(!(TRAIN && FUSE_GQ), "the training sweep keeps the dE/dQ table (its weight gradients contract it)");
  PH_START
  constexpr int NW = WAVES;
  constexpr bool RM = FUSE_GQ;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int HMODE = RM ? 2 : 1;                                      // gated_forward / gated_backward SPLIT mode
  float* vecs = RM ? smem + 2 * AC_RM_IMG : smem + 4 * 4 * IMG64;
  const h16x8* IbT = reinterpret_cast<const h16x8*>(vecs + VEC_SLOTS * D);   // FUSE_GQ only
  float* tiles = vecs + VEC_SLOTS * D + (FUSE_GQ ? 4 * IMG128 : 0);
  const float* W2c = smem;
  const float* W2g = RM ? smem + AC_RM_IMG : smem + 4 * IMG64;
  const float* W2cT = RM ? W2c : smem + 2 * 4 * IMG64;
  const float* W2gT = RM ? W2g : smem + 3 * 4 * IMG64;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* T = tiles + wave * TILE_FLOATS;
  float* Trow = T + j * TS;
  const int ntiles = (p.n_edges + TILE_ROWS - 1) / TILE_ROWS;   // wave-tiles: this wave's own sequence (mfma_tile.h wave_tile_seq)
  const TileSeq ts = wave_tile_seq(ntiles, NW, wave, p.interleave);
  const int last_row = p.n_edges - 1;
  int c, n, k;
  {   // the first tile's indices and gather land under the staging of the weights (see k_atomconv_fwd)
    const int row = max(0, min(ts.at(0) * TILE_ROWS + j, last_row));
    c = p.e_center[row]; n = p.e_nbr[row]; k = row >> 1;   // pair-ordered index arrays
    GatherRegs gr;
    gather_issue128(gr, p.P, c, p.P + 2 * D, n, p.Q, k, 4 * D, 4 * D, 2 * D, lane);
    if (TRAIN) atomconv_bwd_stage(smem, p, tid, 64 * NW);   // fine-tuning: the weights change every step
    else stage_image<(RM ? ac_bwd_rm_image_floats() : ac_bwd_image_floats()) / 4, 64 * NW>(smem, p.image, tid);
    gather_commit128(gr, T, TS, lane);
  }
  __syncthreads();
  TrainTile tt{};
  tt.T = T; tt.lane = lane;
  PH_DECL
  for (int v = 0; v < ts.count; ++v) {
    PH_TILE(v == 0)
    const int row0 = ts.at(v) * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.n_edges - row0);   // even: Ed = 2 Eu and tiles are 16 rows
    if (TRAIN) {
      tt.nvalid = nvalid;
      tt.hrow = j < nvalid ? p.dumpH + (size_t)(row0 + j) * 2 * D : nullptr;
      tt.grow = j < nvalid ? p.dumpG + (size_t)(row0 + j) * 2 * D : nullptr;
    }
    int cn, nn, kn;                                         // the next tile's rows (clamped: harmless reads at the range end)
    {
      const int row = max(0, min(ts.at(v + 1) * TILE_ROWS + j, last_row));
      cn = p.e_center[row]; nn = p.e_nbr[row]; kn = row >> 1;
    }
    V64 wv, gm;
    read_dl_g<VT>(p.wag, (unsigned)k, D, g, wv.t);
    read_dl_g<VT>(p.GA, (unsigned)c, D, g, gm.t);
    const int k0 = row0 >> 1, nb = nvalid >> 1;
    float* gwag_rows = p.Gwag + (size_t)k0 * D + lane;     // this tile owns these rows of Gwag: old values read here,
    float prev[TILE_ROWS / 2];                             // under the forward recomputation
#pragma unroll
    for (int b = 0; b < TILE_ROWS / 2; ++b) prev[b] = p.first_wag ? 0.f : gwag_rows[(size_t)min(b, nb - 1) * D];
    __builtin_amdgcn_wave_barrier();
    V64 zc, zg;
    read_dl<VT>(Trow, g, zc.t);
    read_dl<VT>(Trow + D, g, zg.t);
    GatedState s;
    V64 y;
    PH(0)   // next indices, weights / aggregate adjoint rows, old Gwag rows requested; table sums read
    gated_forward<true, false, TRAIN, HMODE>(zc, zg, W2c, W2g, vecs, j, g, s, y, &tt);
    PH(1)   // forward recomputation
    asm volatile("" : "+v"(cn), "+v"(nn));   // take the index loads here (landed long ago), not behind later stores
    V64 gy, gw, gzc, gzg;
    CHG_EV(ft) {
      gw.t[ft] = gm.t[ft] * y.t[ft];   // dE/d wag[k], this direction
      gy.t[ft] = gm.t[ft] * wv.t[ft];
    }
    __builtin_amdgcn_wave_barrier();
    write_dl<VT>(Trow, g, gw.t);
    __builtin_amdgcn_wave_barrier();
    {  // Gwag[k] += gw(2b) + gw(2b+1)
      float* dst = gwag_rows;
#pragma unroll
      for (int b = 0; b < TILE_ROWS / 2; ++b) prev[b] += T[(2 * b) * TS + lane] + T[(2 * b + 1) * TS + lane];
      if (nvalid == TILE_ROWS) {
#pragma unroll
        for (int b = 0; b < TILE_ROWS / 2; ++b) dst[(size_t)b * D] = prev[b];
      } else {
#pragma unroll
        for (int b = 0; b < TILE_ROWS / 2; ++b)
          if (b < nb) dst[(size_t)b * D] = prev[b];
      }
    }
    PH(2)   // bond-weight gradient rows
    gated_backward<true, false, TRAIN, HMODE>(gy, zc, zg, W2cT, W2gT, vecs, j, g, s, gzc, gzg, &tt);
    PH(3)   // gated adjoint
    if (FUSE_GQ) {
      // dE/d h_bond[k] (+)= (gz(2b) + gz(2b+1)) . W_bond: contracted per direction (linear), the pair summed by a lane swap; the
      // even lane of a pair owns the bond's row (four 64-byte segments), plain read-modify-write: the tile owns bonds k0 .. k0 + 7
      float* gb_row = p.Gb + (size_t)(k0 + (j >> 1)) * D + 4 * g;
      const bool owner = !(j & 1) && j < nvalid;
      V64 old = zero64();
      if (p.gb_accumulate && owner) {
        CHG_EV(ft) old.t[ft] = *reinterpret_cast<const f32x4*>(gb_row + 16 * ft);
      }
      f32x4 gz[2 * VT] = {gzc.t[0], gzc.t[1], gzc.t[2], gzc.t[3], gzg.t[0], gzg.t[1], gzg.t[2], gzg.t[3]};
      V64 gq = zero64();
      gemm_split<2 * VT, VT, true>(gq.t, IbT, D, gz, j, g);
      CHG_EW(ft, r) gq.t[ft][r] += __shfl_xor(gq.t[ft][r], 1);
      CHG_EV(ft) gq.t[ft] += old.t[ft];
      CHG_EV(ft) asm volatile("" : "+v"(gq.t[ft]));   // (sums before the conditional stores)
      if (owner) {
        CHG_EV(ft) *reinterpret_cast<f32x4*>(gb_row + 16 * ft) = gq.t[ft];
      }
    }
    __builtin_amdgcn_wave_barrier();
    write_dl<VT>(Trow, g, gzc.t);
    write_dl<VT>(Trow + D, g, gzg.t);
    __builtin_amdgcn_wave_barrier();
    if (v + 1 < ts.count) {  // the next tile's gathers fly while this tile's run sums are formed and sent
      GatherRegs gr;
      gather_issue128(gr, p.P, cn, p.P + 2 * D, nn, p.Q, kn, 4 * D, 4 * D, 2 * D, lane);
      PH(4)   // next tile's gathers issued
      acbwd_scatter<!FUSE_GQ>(T, c, nvalid, k0, p, lane);
      PH(5)   // scatter: GQ rows, run sums of the two atoms
      __builtin_amdgcn_wave_barrier();
      gather_commit128(gr, T, TS, lane);
      PH(6)   // next tile's gathers landed
    } else {   // the wave's last tile (MD-size batches: its only one) does not wait for rows nobody reads: 5.6k of 47k clocks per wave
      PH(4)
      acbwd_scatter<!FUSE_GQ>(T, c, nvalid, k0, p, lane);
      PH(5)
    }
    c = cn; n = nn; k = kn;
  }
  PH_FLUSH(70)
  if (TRAIN) {
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(p.g_ln + q * D + lane, tt.ln[q]);
  }
}

// =============================================================================================
// BondConv / AngleUpdate (bond graph: rows are angles)
// =============================================================================================
struct AngleArgs {
  const float* R;      // [Eb,256] cols 0..127: bond_i partial, 128..255: bond_j partial
  const float* S;      // [N,128]  centre-atom partial (+b1)
  const float* ang;    // [A,64]   angle features (input of this layer)
  const float* wbgc;   // [Eb,64]  smooth bond weights (bond graph), compact rows (BondConv only)
  const int *a_ctr, *a_b1c, *a_b2c;
  int n_angles;
  int interleave;      // XCD-local interleaved tile sequence (mfma_tile.h wave_tile_seq) or contiguous per-wave ranges
  const float* w_ang;  // [128][64] angle block of the first layer (global)
  GatedW gw;
  const float* image;  // prebuilt weight block of the kernel's LDS (k_angle_image); the TRAIN adjoints stage from the fp32 weights
  int slot;            // host side: which layer's image (engine.hip launch_angle)
  float* out;          // BondConv fwd: agg [Eb,64] zeroed;  AngleUpdate fwd: new angle features [A,64]
  // backward only
  const float* Gagg;   // BondConv: [Eb,64] dE/d agg
  float* Gang;         // [A,64] running dE/d angle features (read as dE/d out for AngleUpdate; += W_ang^T gz)
  float* GR;           // [Eb,256] zeroed
  float* GS;           // [N,128] zeroed
  float* Gwbgc;        // [Eb,64] accumulated over layers (BondConv only)
  int first_gang;      // BondConv adjoint of the last layer: first writer of Gang in the sweep (store, do not read: not zeroed)
  float* phase;        // CHG_PHASE_TIMING builds only: per-phase shader-clock totals (40 floats)
  const int* skip_flag; // row-order kernels (not TRAIN): return at once when *skip_flag == 1 (a per-atom kernel of kernels_angle_w.h / _fa.h runs)
  float* zsave;        // [A + 16,128] or null (large batches, round 6): the forward kernel leaves the first layer's pre-activations z = W_ang x +
                       // R_i + R_j + S behind, the adjoint reads them back instead of gathering four rows per angle and contracting W_ang
                       // again -- these kernels are bound by vector issue at 0.17-0.4 of the HBM rate: bytes are what they have to spare
  // training (k_angle<.., true, .., true>) only
  float* dumpG;        // [A,128] adjoint of the second-layer pre-activations; for AngleUpdate (no hidden layer) this IS dE/dz
  float* dumpH;        // [A,128] hidden activations (BondConv)
  float* dumpZ;        // [A,128] dE/dz of BondConv (A operand of dW_ang = gz^T . angle features)
  float* g_ln;         // [4][64] gradient of ln1_g, ln1_b, ln2_g, ln2_b
};

template <bool HIDDEN, int NW = WAVES, bool BWD = false>
constexpr size_t angle_lds() {
  if (angle_split(HIDDEN, BWD) == 2) return rm_image_bytes(2 * D, D) + 2 * rm_image_bytes(D, D) + sizeof(float) * (VEC_SLOTS * D + NW * TILE_FLOATS);
  return 16 * (size_t)((BWD ? 2 : 1) * IMG128 + (HIDDEN ? 2 * IMG64 : 0)) + sizeof(float) * (VEC_SLOTS * D + NW * TILE_FLOATS);
}


// The weight block at the start of the angle kernels' LDS (see atomconv_fwd_stage).
// mode 1: split images of Wang (and, adjoint, of Wang^T), W2c, W2g.  mode 2 (BondConv adjoint): one row-major image each.
template <bool HIDDEN, bool BWD>
struct AngleLds {
  static constexpr int SPLIT = angle_split(HIDDEN, BWD);
  static constexpr int wangT = SPLIT == 2 ? 0 : 4 * IMG128;     // 4 floats per 16-byte chunk
  static constexpr int w2c = SPLIT == 2 ? (int)(rm_image_bytes(2 * D, D) / 4) : wangT + (BWD ? 4 * IMG128 : 0);
  static constexpr int w2 = HIDDEN ? (SPLIT == 2 ? (int)(rm_image_bytes(D, D) / 4) : 4 * IMG64) : 0;
  static constexpr int w2g = w2c + w2;
  static constexpr int vecs = w2g + w2;
  static constexpr int tiles = vecs + VEC_SLOTS * D;            // = floats of the prebuilt image
};
template <bool HIDDEN, bool BWD>
__device__ __forceinline__ void angle_stage(float* base, const float* __restrict__ w_ang, const GatedW& gw, int tid, int nthreads) {
  using L = AngleLds<HIDDEN, BWD>;
  float* Wang = base;
  float* WangT = base + L::wangT;
  float* W2c = base + L::w2c;
  float* W2g = base + L::w2g;
  if (L::SPLIT == 2) {
    stage_rm(reinterpret_cast<_Float16*>(Wang), w_ang, 2 * D, D, tid, nthreads);
    stage_rm(reinterpret_cast<_Float16*>(W2c), gw.w2c, D, D, tid, nthreads);
    stage_rm(reinterpret_cast<_Float16*>(W2g), gw.w2g, D, D, tid, nthreads);
  } else {
    stage_split<false>(reinterpret_cast<h16x8*>(Wang), w_ang, 2 * D, D, tid, nthreads);
    if (BWD) stage_split<true>(reinterpret_cast<h16x8*>(WangT), w_ang, 2 * D, D, tid, nthreads);
    if (HIDDEN) {
      stage_split<false>(reinterpret_cast<h16x8*>(W2c), gw.w2c, D, D, tid, nthreads);
      stage_split<false>(reinterpret_cast<h16x8*>(W2g), gw.w2g, D, D, tid, nthreads);
    }
  }
  stage_gated_vecs(base + L::vecs, gw, HIDDEN, tid);
}
template <bool HIDDEN, bool BWD>
__global__ __launch_bounds__(BLOCK) void k_angle_image(const float* w_ang, GatedW gw, float* out) {
  angle_stage<HIDDEN, BWD>(out, w_ang, gw, threadIdx.x, BLOCK);
}

// HIDDEN = true: BondConv (gated MLP with one hidden layer, weighted, aggregated over the owning bond)
// HIDDEN = false: AngleUpdate (single gated layer, residual on the angle itself)
template <bool HIDDEN, bool BWD, int NW = WAVES, bool TRAIN = false>
__global__ __launch_bounds__(64 * NW) CHG_TWO_WAVES void k_angle(AngleArgs p) {
  static_assert(!TRAIN || BWD, "TRAIN is a variant of the adjoint kernels");
  if (!TRAIN && p.skip_flag && *p.skip_flag == 1) return;   // a per-atom kernel does this launch's work (kernels_angle_w.h / kernels_angle_fa.h)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  PH_START
  PH_DECL
  constexpr int SPLIT = angle_split(HIDDEN, BWD);
  using L = AngleLds<HIDDEN, BWD>;
  float* Wang = smem;
  float* WangT = smem + L::wangT;
  float* W2c = smem + L::w2c;
  float* W2g = smem + L::w2g;
  float* vecs = smem + L::vecs;
  float* tiles = smem + L::tiles;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  float* T = tiles + wave * TILE_FLOATS;
  float* Trow = T + j * TS;
  const int ntiles = (p.n_angles + TILE_ROWS - 1) / TILE_ROWS;   // wave-tiles: this wave's own sequence (mfma_tile.h wave_tile_seq)
  const TileSeq ts = wave_tile_seq(ntiles, NW, wave, p.interleave);
  const int tstride = TILE_ROWS;
  auto row_of = [&](int v) { return max(0, min(ts.at(v) * tstride + j, p.n_angles - 1)); };   // v-th tile of this wave (clamped past its end)
  // Forward: software-pipelined gathers -- the table rows and angle rows of tile t+1 are in flight
  // (registers) while tile t is computed, its indices were loaded during tile t-1 (angleupd_fwd
  // 0.945 -> 0.871 ms).  Backward: indices one tile ahead only; with the adjoint's register load
  // anything more costs spills (measured, profiles/r01_sq_counters.md).
  constexpr bool PIPE = !BWD;
  int ctr_nx, b1_nx, b2_nx;            // indices of the tile that is gathered next
  int ctr_n2 = 0, b1_n2 = 0, b2_n2 = 0;   // forward: and of the one after it
  GatherRegs gr_p;
  V64 x_p;
  {
    const int a0 = row_of(0);
    ctr_nx = p.a_ctr[a0]; b1_nx = p.a_b1c[a0]; b2_nx = p.a_b2c[a0];
    if (PIPE) {
      gather_issue128(gr_p, p.R, b1_nx, p.R + 2 * D, b2_nx, p.S, ctr_nx, 4 * D, 4 * D, 2 * D, lane);
      read_dl_g<VT>(p.ang, (unsigned)a0, D, g, x_p.t);
      if (1 < ts.count) {
        const int a1 = row_of(1);
        ctr_n2 = p.a_ctr[a1]; b1_n2 = p.a_b1c[a1]; b2_n2 = p.a_b2c[a1];
      }
    }
  }
  // (prologue above: the first tile's indices -- forward: and its gathers -- land under the staging of the weights)
  if (TRAIN) angle_stage<HIDDEN, BWD>(smem, p.w_ang, p.gw, tid, 64 * NW);   // fine-tuning: the weights change every step
  else stage_image<L::tiles / 4, 64 * NW>(smem, p.image, tid);
  __syncthreads();
  PH(9)   // prologue: first indices (forward: first gathers issued), weight images, barrier
  TrainTile tt{};
  tt.T = T; tt.lane = lane;
  for (int v = 0; v < ts.count; ++v) {
    // BondConv adjoint (the kernel of MD-size batches, and with TRAIN of the first-order fine-tuning sweep): lane index opaque per
    // tile, so that the row pointers derived from it are formed where they are used instead of being spilled and reloaded
    // behind the tile's atomics (kernels_angle_w.h, same measure)
    int lane_t = lane;
    if (BWD && HIDDEN) asm volatile("" : "+v"(lane_t));
    PH_TILE(v == 0)
    const int row0 = ts.at(v) * tstride;
    const int nvalid = min(TILE_ROWS, p.n_angles - row0);
    const int ctr = ctr_nx, b1 = b1_nx, b2 = b2_nx;
    if (TRAIN) {
      tt.nvalid = nvalid;
      tt.hrow = (HIDDEN && j < nvalid) ? p.dumpH + (size_t)(row0 + j) * 2 * D : nullptr;
      tt.grow = j < nvalid ? p.dumpG + (size_t)(row0 + j) * 2 * D : nullptr;
    }
    if (!PIPE && v + 1 < ts.count) {
      const int a1 = row_of(v + 1);
      ctr_nx = p.a_ctr[a1]; b1_nx = p.a_b1c[a1]; b2_nx = p.a_b2c[a1];
    }
    if (nvalid <= 0) continue;   // only past the end of the last tile
    const bool valid = j < nvalid;
    const int a = row0 + (valid ? j : 0);
    V64 x;
    f32x4 z[2 * VT];
    if (PIPE) {
      gather_commit128(gr_p, T, TS, lane_t);
      x = x_p;
      __builtin_amdgcn_wave_barrier();
      read_dl<2 * VT>(Trow, g, z);
      PH(0)
      if (v + 1 < ts.count) {
        const int a1 = row_of(v + 1);
        gather_issue128(gr_p, p.R, b1_n2, p.R + 2 * D, b2_n2, p.S, ctr_n2, 4 * D, 4 * D, 2 * D, lane_t);
        // lane_t group recomputed in place (volatile: not hoisted): the loop-invariant p.ang + lane_t offset otherwise lives in a register pair
        // over the whole tile -- at 256 registers it is spilled, and its reload waits (vmcnt, in order) behind the gathers just issued
        int lane_here;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_here));
        read_dl_g<VT>(p.ang, (unsigned)a1, D, lane_here >> 4, x_p.t);
        ctr_nx = ctr_n2; b1_nx = b1_n2; b2_nx = b2_n2;
        if (v + 2 < ts.count) {   // (row index from the recomputed lane_t as well: the strength-reduced constant 2 stride + j was spilled too)
          const int a2 = max(0, min(ts.at(v + 2) * tstride + (lane_here & 15), p.n_angles - 1));
          ctr_n2 = p.a_ctr[a2]; b1_n2 = p.a_b1c[a2]; b2_n2 = p.a_b2c[a2];
        }
      }
    } else {
      // the angle rows are consumed (B operand of the first contraction) before the table sum is written
      gather_rows64(T, TS, p.ang, a, lane_t);
      __builtin_amdgcn_wave_barrier();
      read_dl<VT>(Trow, g, x.t);
      __builtin_amdgcn_wave_barrier();
      PH(0)   // indices + angle rows
      gather_sum128(T, TS, p.R, b1, p.R + 2 * D, b2, p.S, ctr, 4 * D, 4 * D, 2 * D, lane_t);
      __builtin_amdgcn_wave_barrier();
      read_dl<2 * VT>(Trow, g, z);
    }
    PH(1)   // table gather
    Rows64 gy_rows;
    if (BWD && !HIDDEN) rows64_issue(gy_rows, p.Gang, a, lane_t);   // AngleUpdate adjoint: dE/d(new angle), read under the first contraction
    if (SPLIT == 2) gemm_rm<VT, 2 * VT, false, false>(z, reinterpret_cast<const _Float16*>(Wang), 2 * D, D, x.t, j, g, lane_t);
    else gemm_split<VT, 2 * VT, false, !BWD>(z, reinterpret_cast<const h16x8*>(Wang), 2 * D, x.t, j, g);
    if (BWD && !HIDDEN) {
      __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 128, 0);
    }
    // (unconditional stores -- the rows past the end of the last tile go to 16 spare rows behind the array -- so that the compiler's count
    // of outstanding memory operations stays exact.  The stores cost this kernel 0.26 ms per headline launch (0.90 -> 1.16 ms) whichever
    // way they are issued: conditional or not, non-temporal or not; as whole 512-byte rows through the wave's LDS tile 1.28 ms.)
    if (!BWD && p.zsave) write_dl_g_nt<2 * VT>(p.zsave, (unsigned)(valid ? a : p.n_angles + j), 2 * D, g, z);
    V64 zc{{z[0], z[1], z[2], z[3]}}, zg{{z[4], z[5], z[6], z[7]}};
    PH(2)   // W_ang contraction
    GatedState s;
    V64 y;
    constexpr bool SLIM = HIDDEN && BWD;   // the BondConv adjoint is the one kernel that spills otherwise (3.41 -> 3.36 ms)
    gated_forward<HIDDEN, SLIM, TRAIN, SPLIT, !BWD>(zc, zg, W2c, W2g, vecs, j, g, s, y, &tt);
    __builtin_amdgcn_wave_barrier();
    PH(3)   // gated forward
    V64 w1, w2;   // small L2-resident tables: loaded after the MFMA phase to keep its register pressure low
    if (HIDDEN) {
      read_dl_g<VT>(p.wbgc, (unsigned)b1, D, g, w1.t);
      read_dl_g<VT>(p.wbgc, (unsigned)b2, D, g, w2.t);
    }
    if (!BWD) {
      CHG_EV(ft) {
        if (HIDDEN) y.t[ft] = y.t[ft] * (w1.t[ft] * w2.t[ft]);
        else y.t[ft] = y.t[ft] + x.t[ft];
      }
      write_dl<VT>(Trow, g, y.t);
      __builtin_amdgcn_wave_barrier();
      if (HIDDEN) seg_colsum_atomic<D>(T, TS, valid ? b1 : -1, nvalid, p.out, D, lane_t);
      else scatter_rows64<false>(T, TS, p.out, a, nvalid, lane_t);
      PH(4)   // forward output
    } else {
      V64 gy, gzc, gzg;
      if (HIDDEN) {
        V64 g1, g2, gu;
        read_dl_g<VT>(p.Gagg, (unsigned)b1, D, g, gu.t);
        CHG_EV(ft) {
          const f32x4 gyu = gu.t[ft] * y.t[ft];
          g1.t[ft] = gyu * w2.t[ft];      // dE/d wbgc[b1]
          g2.t[ft] = gyu * w1.t[ft];      // dE/d wbgc[b2]
          gy.t[ft] = gu.t[ft] * w1.t[ft] * w2.t[ft];
        }
        write_dl<VT>(Trow, g, g1.t);
        write_dl<VT>(Trow + D, g, g2.t);
        __builtin_amdgcn_wave_barrier();
        seg_colsum_atomic<D>(T, TS, valid ? b1 : -1, nvalid, p.Gwbgc, D, lane_t);
        row_atomic_add<D>(T + D, TS, valid ? b2 : -1, nvalid, p.Gwbgc, D, lane_t);
      } else {
        rows64_commit(gy_rows, T, TS, lane_t);             // dE/d(new angle) of this tile, issued above
        __builtin_amdgcn_wave_barrier();
        read_dl<VT>(Trow, g, gy.t);
      }
      __builtin_amdgcn_wave_barrier();
      PH(4)   // weight rows / Gang rows, dE/dy, (BondConv) Gwbgc scatter
      gated_backward<HIDDEN, SLIM, TRAIN, SPLIT>(gy, zc, zg, W2c, W2g, vecs, j, g, s, gzc, gzg, &tt);
      PH(5)   // gated backward
      // dE/d(angle in) += W_ang^T gz   (the residual identity is already in Gang)
      f32x4 gz[2 * VT] = {gzc.t[0], gzc.t[1], gzc.t[2], gzc.t[3], gzg.t[0], gzg.t[1], gzg.t[2], gzg.t[3]};
      if (TRAIN && HIDDEN && valid) write_dl_g<2 * VT>(p.dumpZ, (unsigned)a, 2 * D, g, gz);
      V64 ga = zero64();
      if (HIDDEN) {
        // BondConv: this tile owns rows a of Gang; their read is issued above the contraction that produces
        // the increment (the group barriers keep the 4 loads ahead of the 128 MFMAs): 3.49 -> 3.41 ms.
        // For AngleUpdate the same costs 8 % (measured), it keeps the plain read-modify-write.
        Rows64 gang_old;
        if (p.first_gang) {
#pragma unroll
          for (int it = 0; it < TILE_ROWS / 4; ++it) gang_old.v[it] = zero4();
        } else {
          rows64_issue(gang_old, p.Gang, a, lane_t);
        }
        gemm_rm<2 * VT, VT, true, true>(ga.t, reinterpret_cast<const _Float16*>(Wang), 2 * D, D, gz, j, g, lane_t);
        write_dl<VT>(Trow, g, ga.t);
        __builtin_amdgcn_wave_barrier();
        PH(6)   // W_ang^T contraction
        scatter_rows64_add(T, TS, p.Gang, a, nvalid, lane_t, gang_old);
      } else {
        gemm_split<2 * VT, VT, true>(ga.t, reinterpret_cast<const h16x8*>(WangT), D, gz, j, g);
        write_dl<VT>(Trow, g, ga.t);
        __builtin_amdgcn_wave_barrier();
        PH(6)   // W_ang^T contraction
        scatter_rows64<true>(T, TS, p.Gang, a, nvalid, lane_t);
      }
      __builtin_amdgcn_wave_barrier();
      PH(7)   // Gang update
      write_dl<2 * VT>(Trow, g, gz);
      __builtin_amdgcn_wave_barrier();
      seg_colsum_atomic<2 * D>(T, TS, valid ? b1 : -1, nvalid, p.GR, 4 * D, lane_t);
      row_atomic_add<2 * D>(T, TS, valid ? b2 : -1, nvalid, p.GR + 2 * D, 4 * D, lane_t);
      seg_colsum_atomic<2 * D>(T, TS, valid ? ctr : -1, nvalid, p.GS, 2 * D, lane_t);
      PH(8)   // GR / GS scatter
    }
    __builtin_amdgcn_wave_barrier();
  }
  PH_FLUSH((HIDDEN ? 0 : 20) + (BWD ? 10 : 0))
  if (TRAIN) {
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(p.g_ln + q * D + lane, tt.ln[q]);
  }
}

// =============================================================================================
// Readout: LayerNorm -> MLP 64-64-64(-64)-1 (silu; two or three hidden layers) -> per-structure sums; and its reverse.
//   model.py:497-509 (readout_norm, mlp, pooling); the site_wise magmom head is k_magmom
// =============================================================================================
struct ReadoutArgs {
  const float* atom;       // [N,64] features after the last AtomConv
  const int* atom_owner;   // [N]
  const int* z;            // [N] atomic numbers
  int n_atoms;
  const float *ln_g, *ln_b, *w0, *b0, *w1, *b1, *w2, *b2, *w3, *b3, *atomref;
  int has_composition;
  int n_hidden;            // hidden layers of the head: 3 (mlp_hidden_dims=[64,64,64]) or 2 ([64,64], the 0.2.0 checkpoint); w2 / b2 unused when 2
  float* site_energy;      // [N]  (includes the AtomRef shift when has_composition)
  float* site_raw;         // [N]  model part only; summed per structure (in order, fp64) by k_finalize
  float* crystal_fea;      // [B,64] zeroed
  float* Ga;               // [N,64] out: dE/d atom (null -> forward only)
  // training (k_readout<true>) only: the reverse sweep starts from d(sum_b cot[b] E_b)/d(site energy)
  int wpb;                 // waves of a workgroup that take a tile (0 = all 8).  Small batches: the six 64 x 64 contractions of a tile are
                           // 12k cycles of fp32 matrix instructions, and two of the 16 tiles of a 256-atom cell per SIMD made the kernel
                           // 20 us; one tile per workgroup spreads them over 16 CUs
  const float* cot;        // [B] cotangent of the per-structure energy sums
  float* dump;             // [9][N,64]: x0, silu(l1), silu(l2), cot*silu(l3), g1, g2, g3, gx0, gx0*xhat  (kernels_train.h contracts them)
};

enum { RO_X0 = 0, RO_S1, RO_S2, RO_S3C, RO_G1, RO_G2, RO_G3, RO_GX, RO_GXX, RO_NDUMP };

constexpr size_t readout_lds() { return sizeof(float) * (3 * D * WS + 6 * D + WAVES * TILE_FLOATS); }

template <bool TRAIN>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_readout(ReadoutArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W0 = smem;
  float* W1 = W0 + D * WS;
  float* W2 = W1 + D * WS;
  float* vecs = W2 + D * WS;  // ln_g ln_b b0 b1 b2 w3
  float* tiles = vecs + 6 * D;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  stage_weights(W0, p.w0, D, D, tid);
  stage_weights(W1, p.w1, D, D, tid);
  stage_weights(W2, p.w2, D, D, tid);
  stage_vector(vecs + 0 * D, p.ln_g, D, tid);
  stage_vector(vecs + 1 * D, p.ln_b, D, tid);
  stage_vector(vecs + 2 * D, p.b0, D, tid);
  stage_vector(vecs + 3 * D, p.b1, D, tid);
  stage_vector(vecs + 4 * D, p.b2, D, tid);
  stage_vector(vecs + 5 * D, p.w3, D, tid);
  __syncthreads();
  float* T = tiles + wave * TILE_FLOATS;
  float* Trow = T + j * TS;
  const float b3 = p.b3[0];
  const int wpb = p.wpb > 0 ? p.wpb : WAVES, block_rows = wpb * TILE_ROWS;
  const int ntiles = (p.n_atoms + block_rows - 1) / block_rows;
  int tb, te;
  tile_range(ntiles, tb, te);
  if (wave >= wpb) return;                 // (no workgroup barrier below)
  for (int tile = tb; tile < te; ++tile) {
    const int row0 = tile * block_rows + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.n_atoms - row0);
    if (nvalid <= 0) continue;
    const bool valid = j < nvalid;
    const int i = row0 + (valid ? j : 0);
    const int owner = p.atom_owner[i];
    gather_rows64(T, TS, p.atom, i, lane);
    __builtin_amdgcn_wave_barrier();
    V64 xh, x0;
    read_dl<VT>(Trow, g, xh.t);
    const float rstd = ln_normalize(xh);
    const V64 gam = param64(vecs + 0 * D, g), bet = param64(vecs + 1 * D, g);
    CHG_EW(ft, r) x0.t[ft][r] = xh.t[ft][r] * gam.t[ft][r] + bet.t[ft][r];
    __builtin_amdgcn_wave_barrier();
    write_dl<VT>(Trow, g, x0.t);
    __builtin_amdgcn_wave_barrier();
    if (!TRAIN) seg_colsum_atomic<D>(T, TS, valid ? owner : -1, nvalid, p.crystal_fea, D, lane);
    V64 l1 = param64(vecs + 2 * D, g), l2 = param64(vecs + 3 * D, g), l3 = param64(vecs + 4 * D, g), sv;
    const size_t plane = (size_t)p.n_atoms * D;                       // TRAIN: one dump plane
    float* drow = TRAIN ? p.dump + (size_t)i * D : nullptr;           // this lane's row inside a plane
    const bool dump = TRAIN && valid;
    const float cot = TRAIN ? p.cot[owner] : 1.0f;
    if (dump) write_dl<VT>(drow + RO_X0 * plane, g, x0.t);
    gemm_dl<VT, VT>(l1.t, W0, WS, x0.t, j, g);
    CHG_EW(ft, r) sv.t[ft][r] = siluf_(l1.t[ft][r]);
    if (dump) write_dl<VT>(drow + RO_S1 * plane, g, sv.t);
    gemm_dl<VT, VT>(l2.t, W1, WS, sv.t, j, g);
    CHG_EW(ft, r) sv.t[ft][r] = siluf_(l2.t[ft][r]);
    if (dump) write_dl<VT>(drow + RO_S2 * plane, g, sv.t);
    const bool three = p.n_hidden != 2;      // uniform
    if (three) gemm_dl<VT, VT>(l3.t, W2, WS, sv.t, j, g);
    else l3 = l2;                            // two hidden layers: the last Linear reads silu(l2)
    const V64 w3 = param64(vecs + 5 * D, g);
    float site = 0.f;
    CHG_EW(ft, r) site += w3.t[ft][r] * siluf_(l3.t[ft][r]);
    site = quad_sum(site) + b3;
    if (!TRAIN && valid && g == 0) {
      const float ref = p.has_composition ? p.atomref[p.z[i] - 1] : 0.f;
      p.site_energy[i] = site + ref;
      p.site_raw[i] = site;
    }
    if (dump) {
      CHG_EW(ft, r) sv.t[ft][r] = cot * siluf_(l3.t[ft][r]);
      write_dl<VT>(drow + RO_S3C * plane, g, sv.t);
    }
    if (p.Ga) {
      V64 g3, g2 = zero64(), g1 = zero64(), gx = zero64();
      CHG_EW(ft, r) g3.t[ft][r] = cot * w3.t[ft][r] * dsiluf_(l3.t[ft][r]);
      if (three) {
        gemm_dl_t<VT, VT>(g2.t, W2, WS, g3.t, j, g);
        CHG_EW(ft, r) g2.t[ft][r] *= dsiluf_(l2.t[ft][r]);
      } else {
        g2 = g3;
      }
      gemm_dl_t<VT, VT>(g1.t, W1, WS, g2.t, j, g);
      CHG_EW(ft, r) g1.t[ft][r] *= dsiluf_(l1.t[ft][r]);
      gemm_dl_t<VT, VT>(gx.t, W0, WS, g1.t, j, g);
      if (dump) {
        write_dl<VT>(drow + RO_G1 * plane, g, g1.t);
        write_dl<VT>(drow + RO_G2 * plane, g, g2.t);
        write_dl<VT>(drow + RO_G3 * plane, g, g3.t);
        write_dl<VT>(drow + RO_GX * plane, g, gx.t);
        V64 t;
        CHG_EW(ft, r) t.t[ft][r] = gx.t[ft][r] * xh.t[ft][r];
        write_dl<VT>(drow + RO_GXX * plane, g, t.t);
      }
      ln_backward(gx, gam, xh, rstd);
      __builtin_amdgcn_wave_barrier();
      write_dl<VT>(Trow, g, gx.t);
      __builtin_amdgcn_wave_barrier();
      scatter_rows64<false>(T, TS, p.Ga, i, nvalid, lane);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace chg
