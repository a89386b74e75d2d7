// kernels_train.h -- reductions that turn the adjoints of the reverse sweep into WEIGHT gradients
// (fine-tuning backward; reference: loss.backward() in chgnet/trainer/trainer.py:399-411
// through CHGNet._compute, model/model.py:427-542).
//
// With the first gated-MLP layer factorised into per-atom / per-bond tables (kernels_conv.h), the
// gradient of a first-layer weight block is a contraction over the TABLE rows, not over edges / angles:
//     dW_centre = GP[:, 0:128]^T . h_atom      dW_bond = GQ^T . h_bond      dW_ctr = GS^T . h_atom  ...
// and the only per-edge / per-angle contractions left are the second Linear of the gated MLP
// (dW2 = gn'^T . hidden) and the angle block (dW_ang = gz^T . angle features).  All of them are the same
// operation, out[m][n] += alpha * sum_rows A[row][m] * B[row][n], done here on MFMA tiles with the
// accumulator kept in registers over a workgroup's whole row range (k_xty).  Bias / LayerNorm-affine
// gradients are column sums (k_colsum, or in-tile sums inside the adjoint kernels).
#pragma once

#include "mfma_tile.h"

namespace chg {

struct XtyArgs {
  const float* A;      // [rows or gathered][>= 16*MT] row stride lda
  int lda;
  const int* a_idx;    // optional row map (null = identity)
  const float* B;      // [rows or gathered][>= 16*NT] row stride ldb
  int ldb;
  const int* b_idx;
  int rows;
  float alpha;
  float* out;          // [16*MT][ldo], accumulated with fp32 atomics (zeroed by the caller)
  int ldo;
  int n_cols;          // columns of the result that exist (<= 16*NT): the 31-wide embedding weights are padded to 32
  float* a_colsum;     // optional [16*MT]: += alpha * column sums of A (bias gradients ride along with the weight gradient)
};

template <int MT, int NT>
constexpr size_t xty_lds() {
  return sizeof(float) * (16 * MT * 16 * NT + WAVES * TILE_ROWS * ((16 * MT + PAD) + (16 * NT + PAD)));
}

// v_mfma_f32_16x16x4_f32 with the ROW index as the contraction index: lane (i, kk) supplies
// A_op[i][kk] = A[row 4s+kk][16mt + i] and B_op[kk][i] = B[row 4s+kk][16nt + i] (both read from the LDS copy of
// the tile, ds_read_b32, consecutive lanes = consecutive addresses); the accumulator element (lane (i, kk), r)
// is out[16mt + 4kk + r][16nt + i].
template <int MT, int NT>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_xty(XtyArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int M = 16 * MT, N = 16 * NT, SA = M + PAD, SB = N + PAD;
  float* red = smem;                         // [M][N] workgroup partial
  float* tiles = red + M * N;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
  for (int idx = tid; idx < M * N; idx += BLOCK) red[idx] = 0.f;
  __syncthreads();
  float* TA = tiles + wave * TILE_ROWS * (SA + SB);
  float* TB = TA + TILE_ROWS * SA;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero4();
  const int ntiles = (p.rows + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range(ntiles, tb, te);
  float asum[(M + 63) / 64];
#pragma unroll
  for (int c = 0; c < (M + 63) / 64; ++c) asum[c] = 0.f;
  constexpr int LA = M / 4, RA = 64 / LA;    // lanes per A row (float4 each), rows per load step
  constexpr int LB = N / 4, RB = 64 / LB;
  constexpr int NA = TILE_ROWS / RA, NB = TILE_ROWS / RB;
  // Software pipeline over tiles: the rows of tile t+1 are requested before tile t's MFMAs and committed to LDS after them (the
  // HBM-bound contractions of the second-order sweep ran at 2.7 TB/s with load -> wait -> contract per tile).
  f32x4 va[NA], vb[NB];
  int nvalid_n = 0;
  auto issue = [&](int tile) {
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    nvalid_n = min(TILE_ROWS, p.rows - row0);
    if (nvalid_n <= 0) return;
    const int row = row0 + (i < nvalid_n ? i : 0);
    const int ra = p.a_idx ? p.a_idx[row] : row;
    const int rb = p.b_idx ? p.b_idx[row] : row;
    {
      const int sub = lane / LA, t = lane % LA;
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        const int rr = RA * it + sub;
        const int r = __shfl(ra, rr);
        va[it] = rr < nvalid_n ? *reinterpret_cast<const f32x4*>(p.A + (size_t)r * p.lda + 4 * t) : zero4();   // rows past the end add nothing
      }
    }
    {
      const int sub = lane / LB, t = lane % LB;
#pragma unroll
      for (int it = 0; it < NB; ++it) {
        const int rr = RB * it + sub;
        const int r = __shfl(rb, rr);
        vb[it] = rr < nvalid_n ? *reinterpret_cast<const f32x4*>(p.B + (size_t)r * p.ldb + 4 * t) : zero4();
      }
    }
  };
  if (tb < te) issue(tb);
  for (int tile = tb; tile < te; ++tile) {
    if (nvalid_n <= 0) break;                  // waves past the end of the last tile (a wave's tiles only move up)
    {
      const int sub = lane / LA, t = lane % LA;
#pragma unroll
      for (int it = 0; it < NA; ++it) *reinterpret_cast<f32x4*>(TA + (RA * it + sub) * SA + 4 * t) = va[it];
    }
    {
      const int sub = lane / LB, t = lane % LB;
#pragma unroll
      for (int it = 0; it < NB; ++it) *reinterpret_cast<f32x4*>(TB + (RB * it + sub) * SB + 4 * t) = vb[it];
    }
    __builtin_amdgcn_wave_barrier();
    if (tile + 1 < te) issue(tile + 1); else nvalid_n = 0;
    if (p.a_colsum) {   // rows past the end were zero-filled: all 16 rows may be summed
#pragma unroll
      for (int c = 0; c < (M + 63) / 64; ++c)
        if (64 * c + lane < M) {
#pragma unroll
          for (int rr = 0; rr < TILE_ROWS; ++rr) asum[c] += TA[rr * SA + 64 * c + lane];
        }
    }
#pragma unroll
    for (int s = 0; s < TILE_ROWS / 4; ++s) {
      float a[MT], b[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = TA[(4 * s + kk) * SA + 16 * mt + i];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = TB[(4 * s + kk) * SB + 16 * nt + i];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the workgroup's waves meet in LDS (ds_add_f32), then one global atomic per element and workgroup
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&red[(16 * mt + 4 * kk + r) * N + 16 * nt + i], acc[mt][nt][r]);
  __syncthreads();
  for (int idx = tid; idx < M * N; idx += BLOCK) {
    const int m = idx / N, n = idx - m * N;
    if (n < p.n_cols) atomicAdd(p.out + (size_t)m * p.ldo + n, p.alpha * red[idx]);
  }
  if (p.a_colsum) {
#pragma unroll
    for (int c = 0; c < (M + 63) / 64; ++c)
      if (64 * c + lane < M) atomicAdd(p.a_colsum + 64 * c + lane, p.alpha * asum[c]);
  }
}

// out[c] += alpha * sum_rows A[row][c] (* Bm[row][c]),  width in {64, 128, 256}
struct ColsumArgs {
  const float* A;
  int lda;
  const float* Bm;     // optional elementwise factor (null = 1)
  int ldb;
  int rows, width;
  float alpha;
  float* out;
};

__global__ __launch_bounds__(256) void k_colsum(ColsumArgs p) {
  __shared__ float part[256];
  const int tid = threadIdx.x;
  const int c = tid % p.width, grp = tid / p.width, ngrp = 256 / p.width;
  float acc = 0.f;
  for (int r = blockIdx.x * ngrp + grp; r < p.rows; r += gridDim.x * ngrp) {
    const float a = p.A[(size_t)r * p.lda + c];
    acc += p.Bm ? a * p.Bm[(size_t)r * p.ldb + c] : a;
  }
  part[tid] = acc;
  __syncthreads();
  if (grp == 0) {
    for (int g2 = 1; g2 < ngrp; ++g2) acc += part[g2 * p.width + c];
    atomicAdd(p.out + c, p.alpha * acc);
  }
}

// d emb[z-1] += dE/d atom[0]   (embedding lookup, model.py:432-434)
__global__ void k_embed_grad(const float* __restrict__ Ga, const int* __restrict__ z, float* __restrict__ gemb, int n_atoms) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_atoms * D) return;
  const int i = idx / D, c = idx - i * D;
  atomicAdd(gemb + (size_t)(z[i] - 1) * D + c, Ga[idx]);
}

// magmom head m_i = |h_i . w + b| (model.py:484-487): given gm_i = d loss / d m_i,
//   dE/d h_i += gm_i sign(h_i . w + b) w,   d w += sum_i gm_i sign h_i,   d b += sum_i gm_i sign
// one wave per atom (grid-stride), lane = feature; the weight sums stay in registers until the end.
__global__ __launch_bounds__(256) void k_magmom_bwd(const float* __restrict__ atom, const float* __restrict__ w, const float* __restrict__ b,
                                                    const float* __restrict__ gm, float* __restrict__ Ga, float* __restrict__ g_w,
                                                    float* __restrict__ g_b, int n_atoms) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const float wl = w[lane], bb = b[0];
  float acc_w = 0.f, acc_b = 0.f;
  for (int i = wave; i < n_atoms; i += nwaves) {
    const float h = atom[(size_t)i * D + lane];
    float s = h * wl;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    s += bb;
    const float coef = gm[i] * (s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f));   // torch.abs: subgradient 0 at 0
    Ga[(size_t)i * D + lane] += coef * wl;
    acc_w += coef * h;
    acc_b += coef;
  }
  atomicAdd(g_w + lane, acc_w);
  if (lane == 0) atomicAdd(g_b, acc_b);
}

}  // namespace chg
