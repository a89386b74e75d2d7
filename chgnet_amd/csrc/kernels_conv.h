// kernels_conv.h -- message-passing kernels of the CHGNet hot path, forward and reverse.
//
// Reference ops replaced (file:line relative to /root/reference/chgnet):
//   AtomConv.forward     model/layers.py:113-132   -> k_atomconv_fwd  / k_atomconv_bwd
//   BondConv.forward     model/layers.py:238-260   -> k_bondconv_fwd  / k_bondconv_bwd
//   AngleUpdate.forward  model/layers.py:348-360   -> k_angleupd_fwd  / k_angleupd_bwd
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

namespace chg {

constexpr int TS = 2 * D + PAD;  // LDS tile row stride (floats) for 128-wide rows
constexpr int WS = D + PAD;      // LDS weight row stride for K = 64
constexpr int TILE_FLOATS = TILE_ROWS * TS;
constexpr int VEC_SLOTS = 6;     // b2c b2g ln1_g ln1_b ln2_g ln2_b

struct GatedW {            // global pointers into the weight blob
  const float *w2c, *b2c, *w2g, *b2g, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
};

__device__ __forceinline__ void stage_gated_vecs(float* vecs, const GatedW& g, bool hidden, int tid) {
  if (hidden) {
    stage_vector(vecs + 0 * D, g.b2c, D, tid);
    stage_vector(vecs + 1 * D, g.b2g, D, tid);
  }
  stage_vector(vecs + 2 * D, g.ln1_g, D, tid);
  stage_vector(vecs + 3 * D, g.ln1_b, D, tid);
  stage_vector(vecs + 4 * D, g.ln2_g, D, tid);
  stage_vector(vecs + 5 * D, g.ln2_b, D, tid);
}

// z (pre-activation of the first layer, 128 = core|gate) -> normalised branches and activations.
//   HIDDEN: c = W2c silu(zc) + b2c, g = W2g silu(zg) + b2g ; else c = zc, g = zg
//   xh* = LayerNorm-normalised (before affine), n1 = affine core branch, a1 = silu(n1), a2 = sigmoid(n2)
template <bool HIDDEN>
__device__ __forceinline__ void gated_forward(const f32x16 (&zc)[2], const f32x16 (&zg)[2], const float* W2c, const float* W2g,
                                              const float* vecs, int j, int h, f32x16 (&xh1)[2], f32x16 (&xh2)[2], float& rstd1,
                                              float& rstd2, f32x16 (&n1)[2], f32x16 (&a1)[2], f32x16 (&a2)[2]) {
  if (HIDDEN) {
    f32x16 hc[2], hg[2], b[2];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hc[ft][r] = siluf_(zc[ft][r]);
        hg[ft][r] = siluf_(zg[ft][r]);
      }
    param_read_dl<2>(vecs + 0 * D, h, b);
    xh1[0] = b[0];
    xh1[1] = b[1];
    gemm_dl<2, 2>(xh1, W2c, WS, hc, j, h);
    param_read_dl<2>(vecs + 1 * D, h, b);
    xh2[0] = b[0];
    xh2[1] = b[1];
    gemm_dl<2, 2>(xh2, W2g, WS, hg, j, h);
  } else {
    xh1[0] = zc[0];
    xh1[1] = zc[1];
    xh2[0] = zg[0];
    xh2[1] = zg[1];
  }
  rstd1 = ln_normalize(xh1);
  rstd2 = ln_normalize(xh2);
  f32x16 g[2], b[2];
  param_read_dl<2>(vecs + 2 * D, h, g);
  param_read_dl<2>(vecs + 3 * D, h, b);
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      n1[ft][r] = xh1[ft][r] * g[ft][r] + b[ft][r];
      a1[ft][r] = siluf_(n1[ft][r]);
    }
  param_read_dl<2>(vecs + 4 * D, h, g);
  param_read_dl<2>(vecs + 5 * D, h, b);
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int r = 0; r < 16; ++r) a2[ft][r] = sigmoidf_(xh2[ft][r] * g[ft][r] + b[ft][r]);
}

// gy = dE/d(a1*a2)  ->  gzc, gzg = dE/dz (128 wide)
template <bool HIDDEN>
__device__ __forceinline__ void gated_backward(const f32x16 (&gy)[2], const f32x16 (&zc)[2], const f32x16 (&zg)[2], const float* W2c,
                                               const float* W2g, const float* vecs, int j, int h, const f32x16 (&xh1)[2],
                                               const f32x16 (&xh2)[2], float rstd1, float rstd2, const f32x16 (&n1)[2],
                                               const f32x16 (&a1)[2], const f32x16 (&a2)[2], f32x16 (&gzc)[2], f32x16 (&gzg)[2]) {
  f32x16 gn1[2], gn2[2], gam[2];
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gn1[ft][r] = gy[ft][r] * a2[ft][r] * dsiluf_(n1[ft][r]);
      gn2[ft][r] = gy[ft][r] * a1[ft][r] * a2[ft][r] * (1.0f - a2[ft][r]);
    }
  param_read_dl<2>(vecs + 2 * D, h, gam);
  ln_backward(gn1, gam, xh1, rstd1);
  param_read_dl<2>(vecs + 4 * D, h, gam);
  ln_backward(gn2, gam, xh2, rstd2);
  if (HIDDEN) {
    gzc[0] = zero16();
    gzc[1] = zero16();
    gzg[0] = zero16();
    gzg[1] = zero16();
    gemm_dl_t<2, 2>(gzc, W2c, WS, gn1, j, h);
    gemm_dl_t<2, 2>(gzg, W2g, WS, gn2, j, h);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        gzc[ft][r] *= dsiluf_(zc[ft][r]);
        gzg[ft][r] *= dsiluf_(zg[ft][r]);
      }
  } else {
    gzc[0] = gn1[0];
    gzc[1] = gn1[1];
    gzg[0] = gn2[0];
    gzg[1] = gn2[1];
  }
}

// =============================================================================================
// k_rows_gemm:  Y[o(r), yoff + n] (+)= sum_k X[i(r), xoff + k] * Wt[n][k] (+ bias[n]) (+ resid[o(r), n])
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
};

template <int K, int NOUT>
__global__ __launch_bounds__(BLOCK) void k_rows_gemm(RowsGemm p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KS = K + PAD, KT = K / 32, NFT = NOUT / 32;
  constexpr int XS = (K > NOUT ? K : NOUT) + PAD;  // tile stride: holds X (K wide) then Y (NOUT wide)
  float* W = smem;                          // [NOUT][KS]
  float* bias = W + NOUT * KS;              // [NOUT]
  float* tiles = bias + NOUT;               // [WAVES][32][XS]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
  stage_weights(W, p.Wt, NOUT, K, tid);
  for (int idx = tid; idx < NOUT; idx += BLOCK) bias[idx] = p.bias ? p.bias[idx] : 0.f;
  __syncthreads();
  float* T = tiles + wave * TILE_ROWS * XS;
  const int ntiles = (p.rows + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range(ntiles, tb, te);
  for (int tile = tb; tile < te; ++tile) {
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.rows - row0);
    if (nvalid <= 0) continue;
    const int rr_ = row0 + (j < nvalid ? j : 0);
    const int in_row = p.in_idx ? p.in_idx[rr_] : rr_;
    const int out_row = p.out_idx ? p.out_idx[rr_] : rr_;
    // X tile -> LDS, K/4 lanes per row
    constexpr int LPR = K / 4, RPS = 64 / LPR;
    {
      const int sub = lane / LPR, t = lane % LPR;
#pragma unroll 4
      for (int it = 0; it < TILE_ROWS / RPS; ++it) {
        const int rr = RPS * it + sub;
        const int r = __shfl(in_row, rr);
        *reinterpret_cast<f32x4*>(T + rr * XS + 4 * t) = *reinterpret_cast<const f32x4*>(p.X + (size_t)r * p.ldx + 4 * t);
      }
    }
    __builtin_amdgcn_wave_barrier();
    f32x16 x[KT];
    lds_read_dl<KT>(T, XS, j, h, 0, x);
    f32x16 acc[NFT];
    param_read_dl<NFT>(bias, h, acc);
    gemm_dl<KT, NFT>(acc, W, KS, x, j, h);
    __builtin_amdgcn_wave_barrier();
    lds_write_dl<NFT>(T, XS, j, h, 0, acc);
    __builtin_amdgcn_wave_barrier();
    // Y tile -> global, NOUT/4 lanes per row
    constexpr int LPO = NOUT / 4, RPO = 64 / LPO;
    {
      const int sub = lane / LPO, t = lane % LPO;
#pragma unroll 4
      for (int it = 0; it < TILE_ROWS / RPO; ++it) {
        const int rr = RPO * it + sub;
        const int r = __shfl(out_row, rr);
        if (rr < nvalid) {
          f32x4 v = *reinterpret_cast<const f32x4*>(T + rr * XS + 4 * t);
          if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + (size_t)r * p.ldr + 4 * t);
          f32x4* dst = reinterpret_cast<f32x4*>(p.Y + (size_t)r * p.ldy + 4 * t);
          if (p.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int K, int NOUT>
constexpr size_t rows_gemm_lds() {
  return sizeof(float) * (NOUT * (K + PAD) + NOUT + WAVES * TILE_ROWS * ((K > NOUT ? K : NOUT) + PAD));
}

// =============================================================================================
// AtomConv
// =============================================================================================
struct AtomConvArgs {
  const float* P;      // [N,256]  cols 0..127: centre partial (+b1), 128..255: neighbour partial
  const float* Q;      // [Eu,128] bond partial
  const float* wag;    // [Eu,64]  smooth bond weights (atom graph)
  const int *e_center, *e_nbr, *e_d2u;
  int n_edges;
  GatedW gw;
  float* agg;          // fwd out: [N,64], zeroed by the caller
  // backward only
  const float* GA;     // [N,64] dE/d agg
  float* GP;           // [N,256] zeroed: grads of the two partials
  float* GQ;           // [Eu,128] zeroed
  float* Gwag;         // [Eu,64] accumulated over layers
};

constexpr size_t atomconv_lds() { return sizeof(float) * (2 * D * WS + VEC_SLOTS * D + WAVES * TILE_FLOATS); }

template <bool BWD>
__global__ __launch_bounds__(BLOCK) void k_atomconv(AtomConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W2c = smem;
  float* W2g = W2c + D * WS;
  float* vecs = W2g + D * WS;
  float* tiles = vecs + VEC_SLOTS * D;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
  stage_weights(W2c, p.gw.w2c, D, D, tid);
  stage_weights(W2g, p.gw.w2g, D, D, tid);
  stage_gated_vecs(vecs, p.gw, true, tid);
  __syncthreads();
  float* T = tiles + wave * TILE_FLOATS;
  const int ntiles = (p.n_edges + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range(ntiles, tb, te);
  for (int tile = tb; tile < te; ++tile) {
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.n_edges - row0);
    if (nvalid <= 0) continue;
    const bool valid = j < nvalid;
    const int e = row0 + (valid ? j : 0);
    const int c = p.e_center[e], n = p.e_nbr[e], k = p.e_d2u[e];
    gather_sum128(T, TS, p.P, c, p.P + 2 * D, n, p.Q, k, 4 * D, 4 * D, 2 * D, lane);
    __builtin_amdgcn_wave_barrier();
    f32x16 zc[2], zg[2];
    lds_read_dl<2>(T, TS, j, h, 0, zc);
    lds_read_dl<2>(T, TS, j, h, D, zg);
    f32x16 xh1[2], xh2[2], n1[2], a1[2], a2[2], wv[2];
    float rstd1, rstd2;
    gated_forward<true>(zc, zg, W2c, W2g, vecs, j, h, xh1, xh2, rstd1, rstd2, n1, a1, a2);
    glb_read_dl<2>(p.wag + (size_t)k * D, h, wv);
    __builtin_amdgcn_wave_barrier();
    if (!BWD) {
      f32x16 m[2];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) m[ft][r] = a1[ft][r] * a2[ft][r] * wv[ft][r];
      lds_write_dl<2>(T, TS, j, h, 0, m);
      __builtin_amdgcn_wave_barrier();
      seg_colsum_atomic<D>(T, TS, valid ? c : -1, nvalid, p.agg, D, lane);
    } else {
      f32x16 gm[2], gy[2], gw[2], gzc[2], gzg[2];
      glb_read_dl<2>(p.GA + (size_t)c * D, h, gm);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          gw[ft][r] = gm[ft][r] * a1[ft][r] * a2[ft][r];   // dE/d wag[k]
          gy[ft][r] = gm[ft][r] * wv[ft][r];
        }
      lds_write_dl<2>(T, TS, j, h, 0, gw);
      __builtin_amdgcn_wave_barrier();
      row_atomic_add<D>(T, TS, valid ? k : -1, nvalid, p.Gwag, D, lane);
      gated_backward<true>(gy, zc, zg, W2c, W2g, vecs, j, h, xh1, xh2, rstd1, rstd2, n1, a1, a2, gzc, gzg);
      __builtin_amdgcn_wave_barrier();
      lds_write_dl<2>(T, TS, j, h, 0, gzc);
      lds_write_dl<2>(T, TS, j, h, D, gzg);
      __builtin_amdgcn_wave_barrier();
      seg_colsum_atomic<2 * D>(T, TS, valid ? c : -1, nvalid, p.GP, 4 * D, lane);
      row_atomic_add<2 * D>(T, TS, valid ? n : -1, nvalid, p.GP + 2 * D, 4 * D, lane);
      row_atomic_add<2 * D>(T, TS, valid ? k : -1, nvalid, p.GQ, 2 * D, lane);
    }
    __builtin_amdgcn_wave_barrier();
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
  const float* w_ang;  // [128][64] angle block of the first layer (global)
  GatedW gw;
  float* out;          // BondConv fwd: agg [Eb,64] zeroed;  AngleUpdate fwd: new angle features [A,64]
  // backward only
  const float* Gagg;   // BondConv: [Eb,64] dE/d agg
  float* Gang;         // [A,64] running dE/d angle features (read as dE/d out for AngleUpdate; += W_ang^T gz)
  float* GR;           // [Eb,256] zeroed
  float* GS;           // [N,128] zeroed
  float* Gwbgc;        // [Eb,64] accumulated over layers (BondConv only)
};

template <bool HIDDEN>
constexpr size_t angle_lds() {
  return sizeof(float) * (2 * D * WS + (HIDDEN ? 2 * D * WS : 0) + VEC_SLOTS * D + WAVES * TILE_FLOATS);
}

// HIDDEN = true: BondConv (gated MLP with one hidden layer, weighted, aggregated over the owning bond)
// HIDDEN = false: AngleUpdate (single gated layer, residual on the angle itself)
template <bool HIDDEN, bool BWD>
__global__ __launch_bounds__(BLOCK) void k_angle(AngleArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wang = smem;                                  // [128][WS]
  float* W2c = Wang + 2 * D * WS;
  float* W2g = W2c + (HIDDEN ? D * WS : 0);
  float* vecs = W2g + (HIDDEN ? D * WS : 0);
  float* tiles = vecs + VEC_SLOTS * D;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
  stage_weights(Wang, p.w_ang, 2 * D, D, tid);
  if (HIDDEN) {
    stage_weights(W2c, p.gw.w2c, D, D, tid);
    stage_weights(W2g, p.gw.w2g, D, D, tid);
  }
  stage_gated_vecs(vecs, p.gw, HIDDEN, tid);
  __syncthreads();
  float* T = tiles + wave * TILE_FLOATS;
  const int ntiles = (p.n_angles + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range(ntiles, tb, te);
  for (int tile = tb; tile < te; ++tile) {
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.n_angles - row0);
    if (nvalid <= 0) continue;
    const bool valid = j < nvalid;
    const int a = row0 + (valid ? j : 0);
    const int ctr = p.a_ctr[a], b1 = p.a_b1c[a], b2 = p.a_b2c[a];
    // angle features of the tile -> D layout (B operand of the first contraction)
    gather_rows64(T, TS, p.ang, a, lane);
    __builtin_amdgcn_wave_barrier();
    f32x16 x[2];
    lds_read_dl<2>(T, TS, j, h, 0, x);
    __builtin_amdgcn_wave_barrier();
    gather_sum128(T, TS, p.R, b1, p.R + 2 * D, b2, p.S, ctr, 4 * D, 4 * D, 2 * D, lane);
    __builtin_amdgcn_wave_barrier();
    f32x16 z[4];
    lds_read_dl<4>(T, TS, j, h, 0, z);
    gemm_dl<2, 4>(z, Wang, WS, x, j, h);
    f32x16 zc[2] = {z[0], z[1]}, zg[2] = {z[2], z[3]};
    f32x16 xh1[2], xh2[2], n1[2], a1[2], a2[2];
    float rstd1, rstd2;
    gated_forward<HIDDEN>(zc, zg, W2c, W2g, vecs, j, h, xh1, xh2, rstd1, rstd2, n1, a1, a2);
    __builtin_amdgcn_wave_barrier();
    f32x16 w1[2], w2[2];
    if (HIDDEN) {
      glb_read_dl<2>(p.wbgc + (size_t)b1 * D, h, w1);
      glb_read_dl<2>(p.wbgc + (size_t)b2 * D, h, w2);
    }
    if (!BWD) {
      f32x16 y[2];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          y[ft][r] = a1[ft][r] * a2[ft][r];
          if (HIDDEN) y[ft][r] *= w1[ft][r] * w2[ft][r];
          else y[ft][r] += x[ft][r];
        }
      lds_write_dl<2>(T, TS, j, h, 0, y);
      __builtin_amdgcn_wave_barrier();
      if (HIDDEN) seg_colsum_atomic<D>(T, TS, valid ? b1 : -1, nvalid, p.out, D, lane);
      else scatter_rows64<false>(T, TS, p.out, a, nvalid, lane);
    } else {
      f32x16 gy[2], gzc[2], gzg[2];
      if (HIDDEN) {
        f32x16 gu[2], g1[2], g2[2];
        glb_read_dl<2>(p.Gagg + (size_t)b1 * D, h, gu);
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float y = a1[ft][r] * a2[ft][r];
            g1[ft][r] = gu[ft][r] * y * w2[ft][r];      // dE/d wbgc[b1]
            g2[ft][r] = gu[ft][r] * y * w1[ft][r];      // dE/d wbgc[b2]
            gy[ft][r] = gu[ft][r] * w1[ft][r] * w2[ft][r];
          }
        lds_write_dl<2>(T, TS, j, h, 0, g1);
        lds_write_dl<2>(T, TS, j, h, D, g2);
        __builtin_amdgcn_wave_barrier();
        seg_colsum_atomic<D>(T, TS, valid ? b1 : -1, nvalid, p.Gwbgc, D, lane);
        row_atomic_add<D>(T + D, TS, valid ? b2 : -1, nvalid, p.Gwbgc, D, lane);
      } else {
        gather_rows64(T, TS, p.Gang, a, lane);          // dE/d(new angle) of this tile
        __builtin_amdgcn_wave_barrier();
        lds_read_dl<2>(T, TS, j, h, 0, gy);
      }
      __builtin_amdgcn_wave_barrier();
      gated_backward<HIDDEN>(gy, zc, zg, W2c, W2g, vecs, j, h, xh1, xh2, rstd1, rstd2, n1, a1, a2, gzc, gzg);
      // dE/d(angle in) += W_ang^T gz   (the residual identity is already in Gang)
      f32x16 gz[4] = {gzc[0], gzc[1], gzg[0], gzg[1]};
      f32x16 ga[2] = {zero16(), zero16()};
      gemm_dl_t<4, 2>(ga, Wang, WS, gz, j, h);
      lds_write_dl<2>(T, TS, j, h, 0, ga);
      __builtin_amdgcn_wave_barrier();
      scatter_rows64<true>(T, TS, p.Gang, a, nvalid, lane);
      __builtin_amdgcn_wave_barrier();
      lds_write_dl<4>(T, TS, j, h, 0, gz);
      __builtin_amdgcn_wave_barrier();
      seg_colsum_atomic<2 * D>(T, TS, valid ? b1 : -1, nvalid, p.GR, 4 * D, lane);
      row_atomic_add<2 * D>(T, TS, valid ? b2 : -1, nvalid, p.GR + 2 * D, 4 * D, lane);
      seg_colsum_atomic<2 * D>(T, TS, valid ? ctr : -1, nvalid, p.GS, 2 * D, lane);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// =============================================================================================
// Readout: LayerNorm -> MLP 64-64-64-64-1 (silu) -> per-structure sums; and its reverse.
//   model.py:497-509 (readout_norm, mlp, pooling), 477-487 (site_wise magmom lives in k_magmom)
// =============================================================================================
struct ReadoutArgs {
  const float* atom;       // [N,64] features after the last AtomConv
  const int* atom_owner;   // [N]
  const int* z;            // [N] atomic numbers
  int n_atoms;
  const float *ln_g, *ln_b, *w0, *b0, *w1, *b1, *w2, *b2, *w3, *b3, *atomref;
  int has_composition;
  float* site_energy;      // [N]  (includes the AtomRef shift when has_composition)
  float* energy;           // [B]  zeroed: sum of site energies (without AtomRef)
  float* comp_energy;      // [B]  zeroed: sum of AtomRef site shifts
  float* crystal_fea;      // [B,64] zeroed
  float* Ga;               // [N,64] out: dE/d atom (null -> forward only)
};

constexpr size_t readout_lds() { return sizeof(float) * (3 * D * WS + 9 * D + WAVES * TILE_FLOATS); }

__global__ __launch_bounds__(BLOCK) void k_readout(ReadoutArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W0 = smem;
  float* W1 = W0 + D * WS;
  float* W2 = W1 + D * WS;
  float* vecs = W2 + D * WS;  // ln_g ln_b b0 b1 b2 w3
  float* tiles = vecs + 9 * D;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
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
  const float b3 = p.b3[0];
  const int ntiles = (p.n_atoms + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range(ntiles, tb, te);
  for (int tile = tb; tile < te; ++tile) {
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.n_atoms - row0);
    if (nvalid <= 0) continue;
    const bool valid = j < nvalid;
    const int i = row0 + (valid ? j : 0);
    const int owner = p.atom_owner[i];
    gather_rows64(T, TS, p.atom, i, lane);
    __builtin_amdgcn_wave_barrier();
    f32x16 xh[2], x0[2], g[2], b[2];
    lds_read_dl<2>(T, TS, j, h, 0, xh);
    const float rstd = ln_normalize(xh);
    param_read_dl<2>(vecs + 0 * D, h, g);
    param_read_dl<2>(vecs + 1 * D, h, b);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) x0[ft][r] = xh[ft][r] * g[ft][r] + b[ft][r];
    __builtin_amdgcn_wave_barrier();
    lds_write_dl<2>(T, TS, j, h, 0, x0);
    __builtin_amdgcn_wave_barrier();
    seg_colsum_atomic<D>(T, TS, valid ? owner : -1, nvalid, p.crystal_fea, D, lane);
    f32x16 l1[2], l2[2], l3[2], s[2];
    param_read_dl<2>(vecs + 2 * D, h, l1);
    gemm_dl<2, 2>(l1, W0, WS, x0, j, h);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[ft][r] = siluf_(l1[ft][r]);
    param_read_dl<2>(vecs + 3 * D, h, l2);
    gemm_dl<2, 2>(l2, W1, WS, s, j, h);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[ft][r] = siluf_(l2[ft][r]);
    param_read_dl<2>(vecs + 4 * D, h, l3);
    gemm_dl<2, 2>(l3, W2, WS, s, j, h);
    f32x16 w3[2];
    param_read_dl<2>(vecs + 5 * D, h, w3);
    float site = 0.f;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) site += w3[ft][r] * siluf_(l3[ft][r]);
    site = pair_sum(site) + b3;
    if (valid && h == 0) {
      const float ref = p.has_composition ? p.atomref[p.z[i] - 1] : 0.f;
      p.site_energy[i] = site + ref;
      atomicAdd(p.energy + owner, site);
      if (p.has_composition) atomicAdd(p.comp_energy + owner, ref);
    }
    if (p.Ga) {
      f32x16 g3[2], g2[2] = {zero16(), zero16()}, g1[2] = {zero16(), zero16()}, gx[2] = {zero16(), zero16()};
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) g3[ft][r] = w3[ft][r] * dsiluf_(l3[ft][r]);
      gemm_dl_t<2, 2>(g2, W2, WS, g3, j, h);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) g2[ft][r] *= dsiluf_(l2[ft][r]);
      gemm_dl_t<2, 2>(g1, W1, WS, g2, j, h);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[ft][r] *= dsiluf_(l1[ft][r]);
      gemm_dl_t<2, 2>(gx, W0, WS, g1, j, h);
      ln_backward(gx, g, xh, rstd);
      __builtin_amdgcn_wave_barrier();
      lds_write_dl<2>(T, TS, j, h, 0, gx);
      __builtin_amdgcn_wave_barrier();
      scatter_rows64<false>(T, TS, p.Ga, i, nvalid, lane);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace chg
