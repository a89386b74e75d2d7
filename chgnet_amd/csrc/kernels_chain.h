// kernels_chain.h -- row programs for MD-size batches: what used to be two to four dependent small row-GEMM launches between two tile
// kernels as ONE launch.
//
// Why.  A 256-atom prediction is ~60 dependent launches; a launch costs ~4.5 us before its first instruction plus two cold memory
// round trips (every kernel starts with the L2 invalidated: ~2 us each), and 27 of the 60 were row GEMMs over 256 atom rows or ~2,000
// bond-node rows -- 170 us of a 900 us prediction for a few MFLOP.  Between two tile kernels these GEMMs form ROW-LOCAL chains:
//   forward, atoms      atom[l+1] = agg . Wout^T + b + atom[l]  ->  S_bc = atom[l+1] . Wctr^T + b1,  P(l+1) = atom[l+1] . [Wc;Wn]^T,  S_au = ...
//   forward, bond nodes hbc[l+1]  = aggB . Wout^T + b + hbc[l]  ->  R_au = hbc[l+1] . [Wi;Wj]^T,  R_bc(l+1) = ...
//   reverse, bond nodes Gb[bn]   += GR . [Wi;Wj]                 ->  Gagg = Gb[bn] . Wout
//   reverse, atoms      Ga       += GS . Wctr + GP . [Wc;Wn]     ->  GA   = Ga . Wout
// (reference: the Linear layers of AtomConv / BondConv / AngleUpdate, chgnet/model/layers.py:113-132, 238-260, 348-360, and their
// adjoints).  A problem here is: stage 1, T[r] = sum_t X_t[i_t(r)] . W_t^T (+ bias) (+ add[i(r)]), 64 wide, optionally stored; stage 2,
// any number of 64-column blocks Y_o[r][64 c ..] = T[r] . W_o[64 c ..]^T (+ bias).  The grid is (row blocks, column blocks): every
// workgroup recomputes stage 1 for its 128 rows (a 64 x 64..384 contraction per row: cheaper than waiting for another launch to
// have written it) and contracts ONE column block of stage 2; the block with index 0 stores T.  Both stages in the split form of
// the full-width k_rows_gemm instances (16-column blocks were tried first: 544 workgroups of 119 KB LDS = three rounds on 256 CUs).
#pragma once

#include "kernels_conv.h"

namespace chg {

constexpr int CHAIN_TERMS = 3, CHAIN_OUTS = 4, CHAIN_PROBS = 2;
struct ChainTerm {
  const float* X;      // rows of ldx floats; columns 0 .. K-1 are contracted
  const int* idx;      // row map (null: identity)
  const float* W;      // [64][K] row-major
  int ldx, K;          // K = 64 or 128
};
struct ChainOut {
  const float* W;      // [ncols][64] row-major
  const float* bias;   // [ncols] or null
  float* Y;            // rows of ldy floats (identity row map)
  int ldy, ncols;      // ncols a multiple of 64
};
struct ChainProb {
  int rows, nterms, nouts, col_blocks;
  int serial_outs;     // 1 (Y1 aliases `add`): the problem has at most ONE 64-column output, i.e. one workgroup per row block -- the in-place
                       // update of stage 1 must not be seen by a second workgroup that still has to read the old rows
  ChainTerm t[CHAIN_TERMS];
  const float* bias1;  // [64] or null
  const float* add;    // rows of lda floats added to stage 1 (residual / the running value that is accumulated onto), or null
  const int* add_idx;
  int lda;
  float* Y1;           // stage 1 stored here (rows y1_idx[r] or r), or null
  const int* y1_idx;
  int ldy1;
  ChainOut o[CHAIN_OUTS];
};
struct ChainArgs {
  ChainProb p[CHAIN_PROBS];
  int n;
};

__host__ __device__ constexpr int chain_tiles_per_wg(int nterms) { return nterms >= 2 ? 4 : WAVES; }
constexpr int CH_TS = D + PAD;       // row stride of the wave tiles (64 wide: a 128-wide stage-1 input goes through in two halves)
constexpr int CH_W1 = D * (2 * D + PAD);   // floats of one term's weight image
// All terms' weight images are resident at once (three buffers: 101 KB of the 154 KB; the 128-wide wave tiles they displaced were only
// ever used as a transposition buffer): one workgroup barrier per launch instead of two per term.  (Simpler, not faster: the three-term
// in-place chains of the reverse sweep stay at ~18 us -- timing-only builds: every further 128-wide term costs ~3 us of contraction, the
// operand split of eight waves on four SIMDs, and ~1.7 us of weight load + image build; the barriers cost nothing measurable.)
constexpr size_t chain_lds() {
  return sizeof(float) * ((size_t)CHAIN_TERMS * CH_W1 + (size_t)D * (D + PAD) + D + D + (size_t)WAVES * TILE_ROWS * CH_TS);
}

// Everything a workgroup reads is REQUESTED before anything is waited for: a launch of this size is one or two memory round trips
// (cold: every kernel starts behind an L2 invalidate, ~2 us each) plus a few hundred matrix instructions, so the first version --
// stage a term's weights, wait, gather its rows, wait, next term -- spent 17-28 us in five to eight dependent round trips.  A
// thread's share of a 64 x 128 weight block is four 16-byte loads, a lane's share of a 16 x 128 row tile eight.
// Stage 1 (and the 64-column stage 2 of the in-place problems) contracts in the split form of the tile kernels (mfma_split.h: three f16
// matrix instructions per f32 product on hi / lo halves, every row scaled by a power of two first -- exact, any magnitude), like the
// full-width k_rows_gemm instances: with the f32 matrix instruction a 16-row tile x three 128-wide terms is 384 x 32 cycles per wave,
// 12 us on a SIMD that two waves share -- the whole launch.  The split image is built from the prefetched registers: chunk c of the
// image (output f, lane group g, K-step mk) is the two 16-byte pieces W[f][32 mk + 4 g ..] and W[f][32 mk + 16 + 4 g ..].
struct ChainW { f32x4 v[4]; };
struct ChainX { f32x4 v[8]; };
__device__ __forceinline__ void chain_ws_issue(ChainW& r, const float* __restrict__ W, int K, int tid) {      // W: [64][K]
  const int nchunks = (K / 32) * 4 * D;          // 512 (K = 64) or 1,024 (K = 128): one or two per thread
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = tid + q * BLOCK;
    if (c < nchunks) {
      const int f = c % D, g = (c / D) & 3, mk = c / (4 * D);
      const float* src = W + (size_t)f * K + 32 * mk + 4 * g;
      r.v[2 * q] = *reinterpret_cast<const f32x4*>(src);
      r.v[2 * q + 1] = *reinterpret_cast<const f32x4*>(src + 16);
    }
  }
}
__device__ __forceinline__ void chain_ws_commit(const ChainW& r, h16x8* img, int K, int tid) {
  const int nchunks = (K / 32) * 4 * D;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = tid + q * BLOCK;
    if (c < nchunks) {
      h16x8 hi, lo;
      split8(r.v[2 * q], r.v[2 * q + 1], hi, lo);
      img[c] = hi;
      img[nchunks + c] = lo;
    }
  }
}
template <int K>
__device__ __forceinline__ void chain_x_issue(ChainX& r, const ChainTerm& tm, int in_row, int lane) {
  constexpr int LPR = K / 4, RPS = 64 / LPR, NV = TILE_ROWS / RPS;
  const int sub = lane / LPR, t = lane % LPR;
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int rw = __shfl(in_row, RPS * it + sub);
    r.v[it] = *reinterpret_cast<const f32x4*>(tm.X + (size_t)rw * tm.ldx + 4 * t);
  }
}
template <int K>
__device__ __forceinline__ void chain_x_contract(f32x4 (&acc)[VT], const ChainX& r, const float* W1, float* T, int lane, int j, int g) {
  constexpr int LPR = K / 4, RPS = 64 / LPR, NV = TILE_ROWS / RPS, KT = K / 16;
  const int sub = lane / LPR, t = lane % LPR;
  f32x4 x[KT];
  // the row tile through the wave's 64-wide LDS tile, one 64-column half at a time (registers: 16 lanes per row -> accumulator layout)
#pragma unroll
  for (int h = 0; h < K / D; ++h) {
    if ((t >> 4) == h) {
#pragma unroll
      for (int it = 0; it < NV; ++it) *reinterpret_cast<f32x4*>(T + (RPS * it + sub) * CH_TS + 4 * (t & 15)) = r.v[it];
    }
    __builtin_amdgcn_wave_barrier();
    f32x4 xh[VT];
    read_dl<VT>(T + j * CH_TS, g, xh);
#pragma unroll
    for (int f = 0; f < VT; ++f) x[VT * h + f] = xh[f];
    __builtin_amdgcn_wave_barrier();
  }
  gemm_split<KT, VT, true>(acc, reinterpret_cast<const h16x8*>(W1), D, x, j, g);
}

static __global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_rows_chain(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W1 = smem;                              // [CHAIN_TERMS][64][K + PAD]: every term's image
  float* W2 = W1 + CHAIN_TERMS * CH_W1;          // [16 or 64][64 + PAD]
  float* b1s = W2 + D * (D + PAD);               // [64]
  float* b2s = b1s + D;                          // [64]
  float* tiles = b2s + D;                        // [WAVES][16][CH_TS]
  int y = blockIdx.y, pi = 0;
  for (; pi < a.n; ++pi) {
    if (y < a.p[pi].col_blocks) break;
    y -= a.p[pi].col_blocks;
  }
  if (pi >= a.n) return;
  const ChainProb& p = a.p[pi];
  const int cb = y;
  // Row tiles per workgroup: eight, or FOUR for problems with several terms -- the waves 4-7 then only help to build the weight images
  // and leave, so that each of the others has a SIMD to itself for its two or three 128-wide contractions (~3 us each with two waves
  // per SIMD: these launches are a few dozen workgroups on 256 CUs) and twice as many CUs take part.
  const int tpw = chain_tiles_per_wg(p.nterms);
  if ((int)blockIdx.x * tpw * TILE_ROWS >= p.rows) return;      // (uniform: the grid covers the longest problem)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int row0 = (blockIdx.x * tpw + wave) * TILE_ROWS;
  const int nvalid = wave < tpw ? min(TILE_ROWS, p.rows - row0) : 0;
  const bool active = nvalid > 0;
  const int rr_ = active ? row0 + min(j, nvalid - 1) : 0;
  float* T = tiles + wave * TILE_ROWS * CH_TS;
  // ---- requests: row maps; weights of every term and of this workgroup's stage-2 block(s); biases ----
  int in_row[CHAIN_TERMS];
#pragma unroll
  for (int t = 0; t < CHAIN_TERMS; ++t) in_row[t] = (t < p.nterms && p.t[t].idx) ? p.t[t].idx[rr_] : rr_;
  const int add_row = (p.add && p.add_idx) ? p.add_idx[rr_] : rr_;
  const int y1_row = (p.Y1 && p.y1_idx) ? p.y1_idx[rr_] : rr_;
  ChainW wr[CHAIN_TERMS], w2r;
#pragma unroll
  for (int t = 0; t < CHAIN_TERMS; ++t)
    if (t < p.nterms) chain_ws_issue(wr[t], p.t[t].W, p.t[t].K, tid);
  // stage 2: 64-column block `cb` of the outputs laid end to end
  int lb = cb, oi = 0;
  for (; oi < p.nouts - 1; ++oi) {
    const int nb = p.o[oi].ncols >> 6;
    if (lb < nb) break;
    lb -= nb;
  }
  if (p.nouts) chain_ws_issue(w2r, p.o[oi].W + (size_t)lb * D * D, D, tid);
  float bias1 = 0.f, bias2 = 0.f;
  if (tid < D && p.bias1) bias1 = p.bias1[tid];
  if (p.nouts && tid < D && p.o[oi].bias) bias2 = p.o[oi].bias[lb * D + tid];
  // ---- requests: the rows of every term, the rows added to stage 1 ----
  ChainX xr[CHAIN_TERMS];
  const int sub4 = lane >> 4, t4 = lane & 15;    // 16 lanes per 64-wide row, 4 rows per pass
  f32x4 av[TILE_ROWS / 4];
  int orow[TILE_ROWS / 4];
  if (active) {
#pragma unroll
    for (int t = 0; t < CHAIN_TERMS; ++t)
      if (t < p.nterms) {
        if (p.t[t].K == D) chain_x_issue<D>(xr[t], p.t[t], in_row[t], lane);
        else chain_x_issue<2 * D>(xr[t], p.t[t], in_row[t], lane);
      }
#pragma unroll
    for (int it = 0; it < TILE_ROWS / 4; ++it) {
      const int rr = 4 * it + sub4;
      const int ra = __shfl(add_row, rr);
      orow[it] = __shfl(y1_row, rr);
      av[it] = zero4();
      if (p.add) av[it] = *reinterpret_cast<const f32x4*>(p.add + (size_t)ra * p.lda + 4 * t4);
    }
  }
  if (tid < D) { b1s[tid] = bias1; b2s[tid] = bias2; }
  if (p.nouts) chain_ws_commit(w2r, reinterpret_cast<h16x8*>(W2), D, tid);
#pragma unroll
  for (int t = 0; t < CHAIN_TERMS; ++t)
    if (t < p.nterms) chain_ws_commit(wr[t], reinterpret_cast<h16x8*>(W1 + t * CH_W1), p.t[t].K, tid);
  __syncthreads();                               // the only workgroup barrier
  if (!active) return;
  f32x4 acc[VT];
  read_dl<VT>(b1s, g, acc);
  // ---- stage 1 ----
#pragma unroll
  for (int t = 0; t < CHAIN_TERMS; ++t)
    if (t < p.nterms) {
      if (p.t[t].K == D) chain_x_contract<D>(acc, xr[t], W1 + t * CH_W1, T, lane, j, g);
      else chain_x_contract<2 * D>(acc, xr[t], W1 + t * CH_W1, T, lane, j, g);
    }
  write_dl<VT>(T + j * CH_TS, g, acc);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    av[it] += *reinterpret_cast<const f32x4*>(T + (4 * it + sub4) * CH_TS + 4 * t4);
    asm volatile("" : "+v"(av[it]));             // all sums before the conditional stores
  }
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    const int rr = 4 * it + sub4;
    *reinterpret_cast<f32x4*>(T + rr * CH_TS + 4 * t4) = av[it];
    if (p.Y1 && cb == 0 && rr < nvalid) *reinterpret_cast<f32x4*>(p.Y1 + (size_t)orow[it] * p.ldy1 + 4 * t4) = av[it];
  }
  if (p.nouts == 0) return;
  __builtin_amdgcn_wave_barrier();
  f32x4 x2[VT];
  read_dl<VT>(T + j * CH_TS, g, x2);
  __builtin_amdgcn_wave_barrier();
  // ---- stage 2 ----
  const ChainOut& o = p.o[oi];
  f32x4 acc2[VT];
  read_dl<VT>(b2s, g, acc2);
  gemm_split<VT, VT, true>(acc2, reinterpret_cast<const h16x8*>(W2), D, x2, j, g);
  write_dl<VT>(T + j * CH_TS, g, acc2);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    const int rr = 4 * it + sub4;
    if (rr < nvalid)
      *reinterpret_cast<f32x4*>(o.Y + (size_t)(row0 + rr) * o.ldy + lb * D + 4 * t4) = *reinterpret_cast<const f32x4*>(T + rr * CH_TS + 4 * t4);
  }
}

}  // namespace chg
