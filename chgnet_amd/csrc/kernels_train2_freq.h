// kernels_train2_freq.h -- frequency gradients of the second-order fine-tuning sweep on 16-row tiles (round 5).
//
//   d f_j += sum_k [ bar(basis)_kj  d basis_j / d f  +  G(basis)_kj  d2 basis_j / (d x d f)  xdot_k ]
//   bar(basis) = bar(h) W_emb (+ bar(w) W_w),   G(basis) likewise        (x = bond length r / angle theta)
//
// kernels_train2.h did this one ROW per wave with lane = basis index (31 of 64 lanes busy, a 64-step loop of four lane broadcasts and
// two LDS reads per row: 640 vector instructions per row, 6.5 ms of a 1024-structure training step).  Here a wave owns 16 rows like
// every tile kernel of the engine: the adjoint rows are contracted with W^T on the matrix pipe (`embed_adjoint`: split-precision,
// rows scaled per row by a power of two -- gradients of any magnitude), which leaves lane (row j, g) with the 8 basis indices
// k = 16 kt + 4 g + r it also evaluates the basis derivatives for -- the layout of kernels_embed.h.  Sums over the rows of a wave by
// lane shuffles, one atomic per frequency and wave.
#pragma once

#include "kernels_embed.h"
#include "kernels_train2.h"

namespace chg {

struct FreqGradTArgs {
  int rows;                       // bonds (atom-graph cutoff: all Eu; bond-graph cutoff: the Eb node bonds)
  const int* row_und;             // null: row k is undirected bond k; else undirected index of row
  const f32x4 *ev, *vd4;
  const int* u_u2d;
  const float* freq;              // [31]
  float rc;
  Envelope env;
  const float *barA, *gA, *WA;    // adjoint rows [rows,64] and their [64][31] weight
  const float *barB, *gB, *WB;    // optional second pair (null)
  float* g_freq;                  // [31]
};

constexpr size_t freq_grad_lds() { return sizeof(float) * (2 * D * WSB + WAVES * TILE_ROWS * ETS); }

static __global__ __launch_bounds__(BLOCK) void k2_freq_grad_t(FreqGradTArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wa = smem;
  float* Wb = Wa + D * WSB;
  float* tiles = Wb + D * WSB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  stage_embed_split_t(reinterpret_cast<h16x8*>(Wa), p.WA, tid);
  if (p.WB) stage_embed_split_t(reinterpret_cast<h16x8*>(Wb), p.WB, tid);
  float fq[2][4], acc[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * kt + 4 * g + r;
      fq[kt][r] = k < NRAD ? p.freq[k] : 0.f;
      acc[kt][r] = 0.f;
    }
  __syncthreads();
  float* T = tiles + wave * TILE_ROWS * ETS;
  float* Trow = T + j * ETS;
  const int ntiles = (p.rows + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range(ntiles, tb, te);
  for (int tile = tb; tile < te; ++tile) {
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.rows - row0);
    if (nvalid <= 0) continue;
    const bool valid = j < nvalid;
    const int row = row0 + (valid ? j : 0);
    const int e = p.u_u2d[p.row_und ? p.row_und[row] : row];
    const float rr = p.ev[e][3], rd = p.vd4[e][3];
    f32x4 tb_[2] = {zero4(), zero4()}, tg_[2] = {zero4(), zero4()};
    V64 gin;
    auto contract = [&](const float* rows, const float* img, f32x4 (&t)[2]) {
      gather_rows64(T, ETS, rows, row, lane);
      __builtin_amdgcn_wave_barrier();
      read_dl<VT>(Trow, g, gin.t);
      embed_adjoint(t, img, gin, j, g);
      __builtin_amdgcn_wave_barrier();
    };
    contract(p.barA, Wa, tb_);
    contract(p.gA, Wa, tg_);
    if (p.WB) {
      contract(p.barB, Wb, tb_);
      contract(p.gB, Wb, tg_);
    }
    if (valid) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v, dr, df, drdf;
          rbf_all(rr, p.rc, fq[kt][r], p.env, v, dr, df, drdf);
          acc[kt][r] += tb_[kt][r] * df + tg_[kt][r] * drdf * rd;
        }
    }
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = acc[kt][r];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) a += __shfl_xor(a, off);      // over the 16 rows held by the lanes that share g
      const int kf = 16 * kt + 4 * g + r;
      if (j == 0 && kf < NRAD) atomicAdd(p.g_freq + kf, a);
    }
}

// Fourier frequencies: column k of the 31-wide expansion is 1/sqrt(2) (k = 0), sin(f_{k-1} t) (k = 1..15), cos(f_{k-16} t) (k = 16..30)
constexpr size_t angle_freq_grad_lds() { return sizeof(float) * (D * WSB + WAVES * TILE_ROWS * ETS); }

static __global__ __launch_bounds__(BLOCK) void k2_angle_freq_grad_t(const float* __restrict__ bar_ang, const float* __restrict__ g_ang,
                                                                     const float* __restrict__ Wae, const float* __restrict__ th2,
                                                                     const float* __restrict__ freq, float* __restrict__ g_freq, int n_angles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* We = smem;
  float* tiles = We + D * WSB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  stage_embed_split_t(reinterpret_cast<h16x8*>(We), Wae, tid);
  float fq[2][4], acc[2][4];
  int kind[2][4];      // 0 nothing, 1 sine column, 2 cosine column
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * kt + 4 * g + r;
      kind[kt][r] = (k >= 1 && k <= NFREQ) ? 1 : ((k > NFREQ && k < NANG) ? 2 : 0);
      fq[kt][r] = kind[kt][r] == 1 ? freq[k - 1] : (kind[kt][r] == 2 ? freq[k - 1 - NFREQ] : 0.f);
      acc[kt][r] = 0.f;
    }
  __syncthreads();
  float* T = tiles + wave * TILE_ROWS * ETS;
  float* Trow = T + j * ETS;
  const int ntiles = (n_angles + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range(ntiles, tb, te);
  for (int tile = tb; tile < te; ++tile) {
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, n_angles - row0);
    if (nvalid <= 0) continue;
    const bool valid = j < nvalid;
    const int a = row0 + (valid ? j : 0);
    const float theta = th2[2 * (size_t)a], thd = th2[2 * (size_t)a + 1];
    f32x4 tb_[2] = {zero4(), zero4()}, tg_[2] = {zero4(), zero4()};
    V64 gin;
    gather_rows64(T, ETS, bar_ang, a, lane);
    __builtin_amdgcn_wave_barrier();
    read_dl<VT>(Trow, g, gin.t);
    embed_adjoint(tb_, We, gin, j, g);
    __builtin_amdgcn_wave_barrier();
    gather_rows64(T, ETS, g_ang, a, lane);
    __builtin_amdgcn_wave_barrier();
    read_dl<VT>(Trow, g, gin.t);
    embed_adjoint(tg_, We, gin, j, g);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sn, cs;
          sincos_cw(fq[kt][r] * theta, sn, cs);
          const float gq = fq[kt][r];
          if (kind[kt][r] == 1) acc[kt][r] += (tb_[kt][r] * theta * cs + tg_[kt][r] * (cs - gq * theta * sn) * thd) * INV_SQRT_PI;
          if (kind[kt][r] == 2) acc[kt][r] += (-tb_[kt][r] * theta * sn + tg_[kt][r] * (-sn - gq * theta * cs) * thd) * INV_SQRT_PI;
        }
    }
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[kt][r];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off);
      const int k = 16 * kt + 4 * g + r;
      if (j == 0 && kind[kt][r] == 1) atomicAdd(g_freq + (k - 1), v);
      if (j == 0 && kind[kt][r] == 2) atomicAdd(g_freq + (k - 1 - NFREQ), v);
    }
}

}  // namespace chg
