// mfma_tile.h -- wave-level building blocks shared by every gfx950 kernel of the engine.
//
// One wave (64 lanes) owns a tile of 32 rows (edges / angles / atoms / bonds) and keeps
// feature vectors in the "D layout" of v_mfma_f32_32x32x2_f32:
//
//     lane = j + 32*h      j in [0,32) = tile row,  h in {0,1}
//     x[ft][r]  (ft = 32-feature tile, r in [0,16))  holds feature
//         F(ft,r,h) = 32*ft + 8*(r>>2) + 4*h + (r&3)      of row j
//
// i.e. a 64-wide vector is split between lane j and lane j+32, each owning 8 groups of 4
// consecutive floats (one float4 per (ft, r>>2)).  With the contraction index k enumerated
// in that same order, the accumulator of one MFMA GEMM is directly the B operand of the
// next one ("swapped" form D[f][row] = sum_k W[f][k] * X[row][k]: weights are the A operand,
// rows are the columns of D), so chained layers never leave registers, and LayerNorm over
// the 64 features of a row is an in-lane sum plus one exchange with lane^32.
//
// fp32 MFMA is exact f32 (fmaf chain) at 64 FLOP/clk/SIMD; there is no xf32 on CDNA4.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 64;            // feature width (atom = bond = angle)
constexpr int TILE_ROWS = 32;    // rows per wave tile
constexpr int WAVES = 4;         // waves per workgroup
constexpr int BLOCK = 64 * WAVES;
constexpr int BLOCK_ROWS = TILE_ROWS * WAVES;
constexpr int PAD = 4;           // floats of padding per LDS row: keeps 16-B alignment and makes
                                 // ds_read_b128 over 32 consecutive rows conflict-free (row stride = 4 mod 64)
constexpr float LN_EPS = 1e-5f;

__device__ __forceinline__ int dfeat(int ft, int r, int h) { return 32 * ft + 8 * (r >> 2) + 4 * h + (r & 3); }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// ---- activations -------------------------------------------------------------------------------
// sigmoid(x) = 1 / (1 + 2^(-x log2 e)) on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp
// each) -- 4 VALU instructions instead of ~22 for expf + IEEE division, which made the conv kernels
// VALU-bound (profiles/r01 notes).  Saturates correctly: x -> -inf gives rcp(inf) = 0, x -> +inf gives 1.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
// d/dx silu(x) = s (1 + x (1 - s))
__device__ __forceinline__ float dsiluf_(float x) {
  const float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}

// ---- D-layout loads / stores ---------------------------------------------------------------
// LDS tile [32][stride] (or any row-major buffer): row j, features f0 + [0, 32*NT)
template <int NT>
__device__ __forceinline__ void lds_read_dl(const float* tile, int stride, int j, int h, int f0, f32x16 (&x)[NT]) {
#pragma unroll
  for (int ft = 0; ft < NT; ++ft)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + j * stride + f0 + 32 * ft + 8 * q + 4 * h);
      x[ft][4 * q + 0] = v[0];
      x[ft][4 * q + 1] = v[1];
      x[ft][4 * q + 2] = v[2];
      x[ft][4 * q + 3] = v[3];
    }
}

template <int NT>
__device__ __forceinline__ void lds_write_dl(float* tile, int stride, int j, int h, int f0, const f32x16 (&x)[NT]) {
#pragma unroll
  for (int ft = 0; ft < NT; ++ft)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
      v[0] = x[ft][4 * q + 0];
      v[1] = x[ft][4 * q + 1];
      v[2] = x[ft][4 * q + 2];
      v[3] = x[ft][4 * q + 3];
      *reinterpret_cast<f32x4*>(tile + j * stride + f0 + 32 * ft + 8 * q + 4 * h) = v;
    }
}

// per-lane global row pointer (each lane reads its own row): features [0, 32*NT)
template <int NT>
__device__ __forceinline__ void glb_read_dl(const float* __restrict__ row, int h, f32x16 (&x)[NT]) {
#pragma unroll
  for (int ft = 0; ft < NT; ++ft)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + 32 * ft + 8 * q + 4 * h);
      x[ft][4 * q + 0] = v[0];
      x[ft][4 * q + 1] = v[1];
      x[ft][4 * q + 2] = v[2];
      x[ft][4 * q + 3] = v[3];
    }
}

// a 64*NT/2-float parameter vector (bias / gamma / beta), same for every row
template <int NT>
__device__ __forceinline__ void param_read_dl(const float* vec, int h, f32x16 (&x)[NT]) {
  lds_read_dl<NT>(vec, 0, 0, h, 0, x);
}

// ---- MFMA GEMMs in swapped form ---------------------------------------------------------------
// acc[fo] (fo < NFT) += W[32*fo + i][k] * x[k]   for k over 32*KT inputs held in D layout.
// W: LDS, row-major [out][ws] (ws = K + PAD): the A operand of lane (i,h) is read as float4.
template <int KT, int NFT>
__device__ __forceinline__ void gemm_dl(f32x16 (&acc)[NFT], const float* W, int ws, const f32x16 (&x)[KT], int i, int h) {
#pragma unroll
  for (int fo = 0; fo < NFT; ++fo) {
    const float* wrow = W + (32 * fo + i) * ws + 4 * h;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + 32 * kt + 8 * q);
        acc[fo] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], x[kt][4 * q + 0], acc[fo], 0, 0, 0);
        acc[fo] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], x[kt][4 * q + 1], acc[fo], 0, 0, 0);
        acc[fo] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], x[kt][4 * q + 2], acc[fo], 0, 0, 0);
        acc[fo] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], x[kt][4 * q + 3], acc[fo], 0, 0, 0);
      }
  }
}

// Transposed use of the same LDS copy: acc[fo] += W[k][32*fo + i] * x[k]  (W row-major [k][ws]).
// Lanes i are consecutive in memory -> ds_read_b32, conflict-free.
template <int KT, int NFT>
__device__ __forceinline__ void gemm_dl_t(f32x16 (&acc)[NFT], const float* W, int ws, const f32x16 (&x)[KT], int i, int h) {
#pragma unroll
  for (int fo = 0; fo < NFT; ++fo)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float w = W[dfeat(kt, r, h) * ws + 32 * fo + i];
        acc[fo] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x[kt][r], acc[fo], 0, 0, 0);
      }
}

// ---- LayerNorm over the 64 features of a row (2 tiles in D layout) ---------------------------
__device__ __forceinline__ float pair_sum(float v) { return v + __shfl_xor(v, 32); }

// in: c (pre-norm).  out: c <- xhat, returns rstd.
__device__ __forceinline__ float ln_normalize(f32x16 (&c)[2]) {
  float s = 0.f;
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c[ft][r];
  const float mu = pair_sum(s) * (1.0f / 64.0f);
  float v = 0.f;
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      c[ft][r] -= mu;
      v += c[ft][r] * c[ft][r];
    }
  const float rstd = __builtin_amdgcn_rsqf(pair_sum(v) * (1.0f / 64.0f) + LN_EPS);
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[ft][r] *= rstd;
  return rstd;
}

// LayerNorm backward: gy <- rstd * (gx - mean(gx) - xhat * mean(gx*xhat)),  gx = gy * gamma
__device__ __forceinline__ void ln_backward(f32x16 (&gy)[2], const f32x16 (&gamma)[2], const f32x16 (&xhat)[2], float rstd) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gy[ft][r] *= gamma[ft][r];
      s1 += gy[ft][r];
      s2 += gy[ft][r] * xhat[ft][r];
    }
  const float m1 = pair_sum(s1) * (1.0f / 64.0f);
  const float m2 = pair_sum(s2) * (1.0f / 64.0f);
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int r = 0; r < 16; ++r) gy[ft][r] = rstd * (gy[ft][r] - m1 - xhat[ft][r] * m2);
}

// ---- LDS staging of a row-major weight matrix [rows][K] -> [rows][K+PAD] ------------------------
__device__ __forceinline__ void stage_weights(float* dst, const float* __restrict__ src, int rows, int K, int tid) {
  const int ws = K + PAD;
  const int k4 = K / 4;
  for (int idx = tid; idx < rows * k4; idx += BLOCK) {
    const int rr = idx / k4, c4 = idx - rr * k4;
    *reinterpret_cast<f32x4*>(dst + rr * ws + 4 * c4) = *reinterpret_cast<const f32x4*>(src + rr * K + 4 * c4);
  }
}
__device__ __forceinline__ void stage_vector(float* dst, const float* __restrict__ src, int n, int tid) {
  for (int idx = tid; idx < n; idx += BLOCK) dst[idx] = src[idx];
}

// ---- tile scheduling: contiguous tile ranges per workgroup, neighbouring ranges on one XCD -------
// Workgroup b is dispatched to XCD b % 8 (observed, speed only); giving XCD x the logical blocks
// [x*G/8, (x+1)*G/8) keeps one structure's tables inside one XCD's L2.
__device__ __forceinline__ void tile_range(int ntiles, int& begin, int& end) {
  const int G = gridDim.x, b = blockIdx.x;
  int lb = b;
  if ((G & 7) == 0) lb = (b & 7) * (G >> 3) + (b >> 3);
  const int per = ntiles / G, rem = ntiles - per * G;
  begin = lb * per + (lb < rem ? lb : rem);
  end = begin + per + (lb < rem ? 1 : 0);
}

// ---- segmented reductions over the rows of a wave tile -----------------------------------------
// tile: [32][stride] LDS, W columns; lane `rr` holds the (sorted-run) key of row rr in `key`
// (key < 0: skip).  Runs of equal keys are summed per column and flushed with one fp32 atomic
// per (run, column); runs that continue in another tile meet in memory.
template <int W>
__device__ __forceinline__ void seg_colsum_atomic(const float* tile, int stride, int key, int nvalid, float* __restrict__ dst,
                                                  int ldd, int lane) {
#ifdef CHG_EXP_NO_SEG_ATOMICS   // timing experiment only: wrong results
  return;
#endif
  constexpr int NC = W / 64;
  float acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.f;
  int cur = __builtin_amdgcn_readlane(key, 0);
#pragma unroll
  for (int rr = 0; rr < TILE_ROWS; ++rr) {
    if (rr < nvalid) {
      const int kk = __builtin_amdgcn_readlane(key, rr);
      if (kk != cur) {
        if (cur >= 0) {
#pragma unroll
          for (int c = 0; c < NC; ++c) atomicAdd(dst + (size_t)cur * ldd + 64 * c + lane, acc[c]);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = 0.f;
        cur = kk;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] += tile[rr * stride + 64 * c + lane];
    }
  }
  if (cur >= 0 && nvalid > 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) atomicAdd(dst + (size_t)cur * ldd + 64 * c + lane, acc[c]);
  }
}

// every row goes to its own (unsorted) destination row: one 256-B coalesced atomic per 64 columns
template <int W>
__device__ __forceinline__ void row_atomic_add(const float* tile, int stride, int key, int nvalid, float* __restrict__ dst,
                                               int ldd, int lane) {
#ifdef CHG_EXP_NO_ROW_ATOMICS   // timing experiment only (tests/gpu_experiments.sh): wrong results
  return;
#endif
#pragma unroll
  for (int rr = 0; rr < TILE_ROWS; ++rr) {
    if (rr < nvalid) {
      const int kk = __builtin_amdgcn_readlane(key, rr);
      if (kk >= 0) {
#pragma unroll
        for (int c = 0; c < W / 64; ++c) atomicAdd(dst + (size_t)kk * ldd + 64 * c + lane, tile[rr * stride + 64 * c + lane]);
      }
    }
  }
}

// ---- row-spread gather: sum of up to three table rows, 128 floats wide, into an LDS tile ---------
// Half-wave `hw` (32 lanes) handles one row per step; lane t loads float4 #t of the 512-B row, so
// every load instruction covers two full rows.  Lane rr (< 32) holds the three row indices of tile
// row rr; they must be valid (clamped) even for rows past the end of the problem.
__device__ __forceinline__ void gather_sum128(float* tile, int stride, const float* __restrict__ t0, int i0,
                                              const float* __restrict__ t1, int i1, const float* __restrict__ t2, int i2,
                                              int ld0, int ld1, int ld2, int lane) {
  const int hw = lane >> 5, t = lane & 31;
#pragma unroll 4
  for (int it = 0; it < TILE_ROWS / 2; ++it) {
    const int rr = 2 * it + hw;
#ifdef CHG_EXP_NO_GATHER        // timing experiment only: every row reads table row 0 (cache-resident)
    const int r0 = 0, r1 = 0, r2 = 0;
#else
    const int r0 = __shfl(i0, rr), r1 = __shfl(i1, rr), r2 = __shfl(i2, rr);
#endif
    f32x4 a = *reinterpret_cast<const f32x4*>(t0 + (size_t)r0 * ld0 + 4 * t);
    const f32x4 b = *reinterpret_cast<const f32x4*>(t1 + (size_t)r1 * ld1 + 4 * t);
    const f32x4 c = *reinterpret_cast<const f32x4*>(t2 + (size_t)r2 * ld2 + 4 * t);
    a += b;
    a += c;
    *reinterpret_cast<f32x4*>(tile + rr * stride + 4 * t) = a;
  }
}

// contiguous or gathered 64-wide rows into an LDS tile (half-wave per row pair: 16 lanes per row)
__device__ __forceinline__ void gather_rows64(float* tile, int stride, const float* __restrict__ src, int idx, int lane) {
  const int sub = lane >> 4, t = lane & 15;   // 4 rows per step
#pragma unroll 4
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    const int rr = 4 * it + sub;
    const int r = __shfl(idx, rr);
    *reinterpret_cast<f32x4*>(tile + rr * stride + 4 * t) = *reinterpret_cast<const f32x4*>(src + (size_t)r * D + 4 * t);
  }
}

// LDS tile (64 wide) -> global rows (coalesced 256-B rows), plain store or read-modify-write add
template <bool ACCUM>
__device__ __forceinline__ void scatter_rows64(const float* tile, int stride, float* __restrict__ dst, int idx, int nvalid,
                                               int lane) {
  const int sub = lane >> 4, t = lane & 15;
#pragma unroll 4
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    const int rr = 4 * it + sub;
    const int r = __shfl(idx, rr);
    if (rr < nvalid) {
      f32x4 v = *reinterpret_cast<const f32x4*>(tile + rr * stride + 4 * t);
      f32x4* p = reinterpret_cast<f32x4*>(dst + (size_t)r * D + 4 * t);
      if (ACCUM) v += *p;
      *p = v;
    }
  }
}

}  // namespace chg
