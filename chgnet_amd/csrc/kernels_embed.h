// kernels_embed.h -- basis functions + 31->64 embeddings on MFMA tiles, forward and adjoint.
//
// Reference ops replaced (file:line relative to /root/reference/chgnet):
//   RadialBessel x CutoffPolynomial for both cutoffs   model/basis.py:108-116, 197-206
//   bond_embedding / bond_weights_ag / bond_weights_bg  model/model.py:435-437  (3 x Linear(31->64))
//   AngleEncoder (acos, Fourier) + angle_embedding      model/encoders.py:144-146, basis.py:33-40, model.py:439
//   their adjoints w.r.t. the bond length / the two unit vectors (model.py:517-535 via autograd)
//
// One wave = 16 bonds (or angles).  The 31 basis functions (padded to K = 32) are evaluated directly
// in the B-operand layout of v_mfma_f32_16x16x4_f32 -- lane (row j, g) owns basis indices
// k = 16*kt + 4*g + r -- so every lane evaluates 8 sin/cos per row instead of one lane per basis
// function with 33 lanes idle, and the 31x64 contraction runs on the matrix core.  Bases are never
// written to memory.
#pragma once

#include "kernels_geom.h"
#include "mfma_tile.h"
#include "mfma_split.h"

namespace chg {

#ifndef CHG_EMBED_WAVES
#define CHG_EMBED_WAVES CHG_TWO_WAVES
#endif

constexpr int KB = 32;            // basis count padded to a multiple of 16
constexpr int WSB = KB + PAD;     // LDS row stride of a [64][32] embedding weight
constexpr int ETS = D + PAD;      // LDS tile row stride (64-wide rows)

// a [64][31] embedding weight as a split-precision image (mfma_split.h, K = 32 = one k-step, column 31 zero): the forward kernels contract the 31 -> 64 embeddings
// as three f16 MFMAs per product (rows scaled by a power of two) instead of f32 MFMAs at the vector rate -- 96 x 32 matrix-pipe cycles
// per tile of 16 bonds were a third of the bond embedding kernel once its sin / cos were cheap
__device__ __forceinline__ void stage_embed_split(h16x8* img, const float* __restrict__ src, int tid) {
  constexpr int NCH = 4 * D;                        // chunks per plane: [g][f]
  for (int c = tid; c < NCH; c += BLOCK) {
    const int f = c % D, g = c / D;
    h16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * (e >> 2) + 4 * g + (e & 3);
      const float w = k < NRAD ? src[f * NRAD + k] : 0.f;
      hi[e] = (_Float16)w;
      lo[e] = (_Float16)((w - (float)hi[e]) * LO_SCALE);
    }
    img[c] = hi;
    img[NCH + c] = lo;
  }
}

// ... and its transpose [32 outputs k][64 contraction f] as a split image (K = 64: two k-steps) for the adjoint kernels:
// t[k] = sum_f W[f][k] g[f], the adjoint rows g scaled per row by a power of two (gradients of any magnitude)
constexpr int EMB_T_CHUNKS = 2 * 4 * KB;            // chunks per plane
__device__ __forceinline__ void stage_embed_split_t(h16x8* img, const float* __restrict__ src, int tid) {
  for (int c = tid; c < EMB_T_CHUNKS; c += BLOCK) {
    const int k = c % KB, g = (c / KB) & 3, mk = c / (4 * KB);
    h16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = 32 * mk + 16 * (e >> 2) + 4 * g + (e & 3);
      const float w = k < NRAD ? src[f * NRAD + k] : 0.f;
      hi[e] = (_Float16)w;
      lo[e] = (_Float16)((w - (float)hi[e]) * LO_SCALE);
    }
    img[c] = hi;
    img[EMB_T_CHUNKS + c] = lo;
  }
}
// t (2 tiles = 32 values per row) += W^T g
__device__ __forceinline__ void embed_adjoint(f32x4 (&t)[2], const float* img, const V64& gin, int j, int g) {
  SplitRow<2> sr;
  split_row<VT, true>(sr, gin.t);
  gemm_split2<2, true>(t, reinterpret_cast<const h16x8*>(img), KB, sr, j, g);
}

struct BondEmbedTArgs {
  const f32x4* ev;            // [Ed] (v, r)
  const int* u_u2d;           // [Eu]
  const int* u_bnode;         // [Eu] compact bond-node index or -1
  const int* bn_und;          // [Eb] undirected bond of every bond-graph node (PART 2)
  int n_und, n_nodes;
  const float *freq_ag, *freq_bg;   // [31]
  const float *w_emb, *w_ag, *w_bg; // [64][31]
  float rc_ag, rc_bg;
  Envelope env;
  float *hb0, *wag, *wbgc;    // fwd out: [Eu,64], [Eu,64], [Eb,64]
  float* hbc0;                // fwd out (optional): [Eb,64] the bond-graph nodes' copy of their hb0 rows (else a separate gather kernel)
  const float *Gb, *Gwag, *Gwbgc;   // bwd in
  float* Grk;                 // bwd out [Eu] dE/d r_k
  // training (k_bond_embed_t<true, true>) only
  float* Xb;                  // [Eu,64] out: radial bases of every bond, cols 0..31 cutoff r_atom (col 31 = 0), 32..63 cutoff r_bond
  float *g_freq_ag, *g_freq_bg;     // [31] gradients of the learnable frequencies (atomics)
};

constexpr size_t bond_embed_lds() { return sizeof(float) * (3 * D * WSB + WAVES * TILE_ROWS * ETS); }

// PART: 0 = both radial expansions for every bond (the training variant); 1 = the atom-graph part, all Eu bonds (hb0, wag; adjoint: Grk
// written); 2 = the bond-graph part over the Eb bonds that are bond-graph nodes only (wbgc; adjoint: Grk[k] += ...; launched after
// part 1).  Only 12 % of the bonds of a 6 A / 3 A graph are nodes, and the kernel is bound by its 62 sin / cos per bond: evaluating the
// 31 bond-graph functions for the nodes only takes 44 % of the transcendentals out (forward 0.34 -> 0.2x ms, section 6 of DESIGN.md).
// MERGED (small batches): parts 1 and 2 -- and the angle expansion -- run as bodies of ONE launch (k_embed_all), i.e. concurrently: the
// adjoint's two contributions to Grk then meet as atomic adds on a cleared array instead of store + read-modify-write.
template <bool BWD, bool TRAIN = false, int PART = 0, bool MERGED = false>
__device__ __forceinline__ void bond_embed_body(const BondEmbedTArgs& p, int vG, int vb) {
  static_assert(!TRAIN || BWD, "TRAIN is a variant of the adjoint kernel");
  static_assert(!TRAIN || PART == 0, "the training variant keeps both expansions in one pass (it dumps them side by side)");
  constexpr bool AG = PART != 2, BG = PART != 1;     // which expansion(s) this instantiation evaluates
  const int n_rows = PART == 2 ? p.n_nodes : p.n_und;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* We = smem;
  float* Wa = We + D * WSB;
  float* Wb = Wa + D * WSB;
  float* tiles = Wb + D * WSB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  if (BWD) {   // adjoint: split images of the transposes (2,048 of the 2,304 floats of a slot)
    if (AG) stage_embed_split_t(reinterpret_cast<h16x8*>(We), p.w_emb, tid);
    if (AG) stage_embed_split_t(reinterpret_cast<h16x8*>(Wa), p.w_ag, tid);
    if (BG) stage_embed_split_t(reinterpret_cast<h16x8*>(Wb), p.w_bg, tid);
  } else {   // forward: split images in the same slots (2,048 of the 2,304 floats)
    if (AG) stage_embed_split(reinterpret_cast<h16x8*>(We), p.w_emb, tid);
    if (AG) stage_embed_split(reinterpret_cast<h16x8*>(Wa), p.w_ag, tid);
    if (BG) stage_embed_split(reinterpret_cast<h16x8*>(Wb), p.w_bg, tid);
  }
  // this lane's 8 basis indices k = 16*kt + 4*g + r and their frequencies (k = 31 is padding)
  float f6[2][4], f3[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * kt + 4 * g + r;
      f6[kt][r] = k < NRAD ? p.freq_ag[k] : 0.f;
      f3[kt][r] = k < NRAD ? p.freq_bg[k] : 0.f;
    }
  __syncthreads();
  float* T = tiles + wave * TILE_ROWS * ETS;
  float* Trow = T + j * ETS;
  const int ntiles = (n_rows + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range_of(ntiles, vG, vb, tb, te);
  float fa6[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, fa3[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // TRAIN: d freq, this lane's rows
  // Inputs run ahead of the tiles: the bond length of tile t+1 (a load through the index requested during tile t-1) and the indices of
  // tile t+2 are requested at the top of tile t and taken (an empty asm: a wait placed by hand) after the basis functions, BEFORE the
  // tile's stores.  Read in place, every tile began with two dependent round trips, waited for behind the previous tile's stores
  // (conditional, so the compiler's wait covered them all): SQ_WAIT_ANY 61 % of the wave cycles of the forward kernel.
  // (PART 2 walks the node list: row n -> bond bn_und[n]; the bond index rides in `node_n*` there, one more dependent load per request)
  auto row_k = [&](int tile) { return max(0, min(tile * BLOCK_ROWS + wave * TILE_ROWS + j, n_rows - 1)); };
  auto bond_of = [&](int row) { return PART == 2 ? p.bn_und[row] : row; };
  int d_n = 0, node_n = -1, d_n2 = 0, node_n2 = -1;
  float rlen_n = 1.f;
  if (tb < te) {
    const int k0 = bond_of(row_k(tb)), k1 = bond_of(row_k(tb + 1));
    d_n = p.u_u2d[k0]; node_n = PART == 2 ? k0 : p.u_bnode[k0];
    d_n2 = p.u_u2d[k1]; node_n2 = PART == 2 ? k1 : p.u_bnode[k1];
    rlen_n = p.ev[d_n][3];
  }
  for (int tile = tb; tile < te; ++tile) {
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, n_rows - row0);
    const float rlen = rlen_n;
    const int node = node_n;      // PART 0 / 1: compact node index of this bond (or -1); PART 2: the bond of this node row
    {   // requests for the next two tiles (clamped rows: harmless reads past the end)
      node_n = node_n2;
      rlen_n = p.ev[d_n2][3];
      const int k2 = bond_of(row_k(tile + 2));
      d_n2 = p.u_u2d[k2]; node_n2 = PART == 2 ? k2 : p.u_bnode[k2];
    }
    if (nvalid <= 0) continue;
    const bool valid = j < nvalid;
    const int k = row0 + (valid ? j : 0);      // row of this instantiation's list: bond (PART 0 / 1) or node (PART 2)
    f32x4 x6[2], x3[2], d6[2], d3[2], q6[2], q3[2];
    const EnvAt e6 = env_at(rlen, p.rc_ag, p.env), e3 = env_at(rlen, p.rc_bg, p.env);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool pad = 16 * kt + 4 * g + r >= NRAD;
        float v = 0.f, dv = 0.f, df = 0.f;
        if (AG) rbf_eval(rlen, p.rc_ag, f6[kt][r], e6, v, dv, df);
        x6[kt][r] = (pad || !AG) ? 0.f : v;
        d6[kt][r] = (pad || !AG) ? 0.f : dv;
        q6[kt][r] = (pad || !AG) ? 0.f : df;
        if (BG) rbf_eval(rlen, p.rc_bg, f3[kt][r], e3, v, dv, df);
        x3[kt][r] = (pad || !BG) ? 0.f : v;
        d3[kt][r] = (pad || !BG) ? 0.f : dv;
        q3[kt][r] = (pad || !BG) ? 0.f : df;
      }
    asm volatile("" : "+v"(rlen_n), "+v"(d_n2), "+v"(node_n2));   // the requests above have landed by now; nothing is stored before here
    if (TRAIN && valid) {   // bases of this bond: B operand of the embedding-weight gradients (kernels_train.h)
      write_dl<2>(p.Xb + (size_t)k * D, g, x6);
      write_dl<2>(p.Xb + (size_t)k * D + KB, g, x3);
    }
    if (!BWD && PART == 2) {     // bond-graph weights of the node rows: contiguous rows of wbgc
      V64 h = zero64();
      gemm_split<2, VT, true>(h.t, reinterpret_cast<const h16x8*>(Wb), D, x3, j, g);
      write_dl<VT>(Trow, g, h.t);
      __builtin_amdgcn_wave_barrier();
      scatter_rows64<false>(T, ETS, p.wbgc, k, nvalid, lane);
      __builtin_amdgcn_wave_barrier();
    } else if (!BWD) {
      V64 h = zero64();
      gemm_split<2, VT, true>(h.t, reinterpret_cast<const h16x8*>(We), D, x6, j, g);
      write_dl<VT>(Trow, g, h.t);
      __builtin_amdgcn_wave_barrier();
      scatter_rows64<false>(T, ETS, p.hb0, k, nvalid, lane);
      if (p.hbc0 && __any(valid && node >= 0)) {   // bond-graph nodes keep a compact copy of their embedding rows
        const int sub = lane >> 4, t = lane & 15;
#pragma unroll
        for (int it = 0; it < TILE_ROWS / 4; ++it) {
          const int rr = 4 * it + sub;
          const int nd = __shfl(node, rr);
          if (rr < nvalid && nd >= 0)
            *reinterpret_cast<f32x4*>(p.hbc0 + (size_t)nd * D + 4 * t) = *reinterpret_cast<const f32x4*>(T + rr * ETS + 4 * t);
        }
      }
      __builtin_amdgcn_wave_barrier();
      h = zero64();
      gemm_split<2, VT, true>(h.t, reinterpret_cast<const h16x8*>(Wa), D, x6, j, g);
      write_dl<VT>(Trow, g, h.t);
      __builtin_amdgcn_wave_barrier();
      scatter_rows64<false>(T, ETS, p.wag, k, nvalid, lane);
      __builtin_amdgcn_wave_barrier();
      if (BG && __any(valid && node >= 0)) {
        h = zero64();
        gemm_split<2, VT, true>(h.t, reinterpret_cast<const h16x8*>(Wb), D, x3, j, g);
        write_dl<VT>(Trow, g, h.t);
        __builtin_amdgcn_wave_barrier();
        // rows that are bond-graph nodes go to their compact slot; others are dropped
        const int sub = lane >> 4, t = lane & 15;
#pragma unroll
        for (int it = 0; it < TILE_ROWS / 4; ++it) {
          const int rr = 4 * it + sub;
          const int nd = __shfl(node, rr);
          if (rr < nvalid && nd >= 0)
            *reinterpret_cast<f32x4*>(p.wbgc + (size_t)nd * D + 4 * t) = *reinterpret_cast<const f32x4*>(T + rr * ETS + 4 * t);
        }
      }
      __builtin_amdgcn_wave_barrier();
    } else if (PART == 2) {      // dE/dr of the node bonds through the bond-graph weights: Grk[bond] += (Wb^T Gwbgc[node]) . d(basis)/dr
      f32x4 t3[2] = {zero4(), zero4()};
      V64 gin;
      const float old = (valid && !MERGED) ? p.Grk[node] : 0.f;      // (node = this row's bond here; part 1 has written Grk)
      read_dl<VT>(p.Gwbgc + (size_t)k * D, g, gin.t);
      embed_adjoint(t3, Wb, gin, j, g);
      float acc = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc += t3[kt][r] * d3[kt][r];
      acc = quad_sum(acc);
      if (valid && g == 0) {
        if (MERGED) atomicAdd(p.Grk + node, acc); else p.Grk[node] = old + acc;
      }
    } else {
      // t6 = We^T Gb[k] + Wa^T Gwag[k],  t3 = Wb^T Gwbgc[node]   (64 -> 32 each), then dot with d(basis)/dr
      f32x4 t6[2] = {zero4(), zero4()}, t3[2] = {zero4(), zero4()};
      V64 gin;
      gather_rows64(T, ETS, p.Gb, k, lane);
      __builtin_amdgcn_wave_barrier();
      read_dl<VT>(Trow, g, gin.t);
      embed_adjoint(t6, We, gin, j, g);
      __builtin_amdgcn_wave_barrier();
      gather_rows64(T, ETS, p.Gwag, k, lane);
      __builtin_amdgcn_wave_barrier();
      read_dl<VT>(Trow, g, gin.t);
      embed_adjoint(t6, Wa, gin, j, g);
      __builtin_amdgcn_wave_barrier();
      if (BG && __any(valid && node >= 0)) {
        read_dl<VT>(p.Gwbgc + (size_t)(node >= 0 ? node : 0) * D, g, gin.t);
        if (node < 0) gin = zero64();
        embed_adjoint(t3, Wb, gin, j, g);
      }
      float acc = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc += t6[kt][r] * d6[kt][r] + t3[kt][r] * d3[kt][r];
      acc = quad_sum(acc);
      if (valid && g == 0) {
        if (MERGED) atomicAdd(p.Grk + k, acc); else p.Grk[k] = acc;
      }
      if (TRAIN && valid) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            fa6[kt][r] += t6[kt][r] * q6[kt][r];
            fa3[kt][r] += t3[kt][r] * q3[kt][r];
          }
      }
    }
  }
  if (TRAIN) {   // sum over the 16 rows held by the lanes that share g, then one atomic per frequency and wave
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = fa6[kt][r], b = fa3[kt][r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          a += __shfl_xor(a, off);
          b += __shfl_xor(b, off);
        }
        const int kf = 16 * kt + 4 * g + r;
        if (j == 0 && kf < NRAD) {
          atomicAdd(p.g_freq_ag + kf, a);
          atomicAdd(p.g_freq_bg + kf, b);
        }
      }
  }
}

template <bool BWD, bool TRAIN = false, int PART = 0>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_bond_embed_t(BondEmbedTArgs p) {
  bond_embed_body<BWD, TRAIN, PART>(p, gridDim.x, blockIdx.x);
}

// The forward expansions at FOUR waves per SIMD (100 / 82 registers: two workgroups of 62 / 44 KB LDS per CU): they wait on the
// acknowledgements of their own stores (gfx950: loads and stores share ONE in-order counter; SQ_WAIT_ANY 70 %), and twice the waves
// per CU is twice the stores in flight.
#define CHG_FOUR_WAVES __attribute__((amdgpu_waves_per_eu(4, 4)))
template <int PART>
__global__ __launch_bounds__(BLOCK) CHG_FOUR_WAVES void k_bond_embed_fwd_o4(BondEmbedTArgs p) {
  bond_embed_body<false, false, PART>(p, gridDim.x, blockIdx.x);
}

struct AngleEmbedTArgs {
  const f32x4* eu;            // [Ed] unit vectors
  const int *a_d1, *a_d2;     // [A] directed edges of the two bonds
  int n_angles;
  const float* freq;          // [15]
  const float* w_emb;         // [64][31]
  float* ang0;                // fwd out [A,64]
  const float* Gang;          // bwd in  [A,64]
  float* Gu;                  // bwd out [Ed,4] zeroed, dE/d unit vectors
  // training (k_angle_embed_t<true, true>) only
  float* Xa;                  // [A,32] out: Fourier basis of every angle (col 31 = 0)
  float* g_freq;              // [15] gradient of the learnable frequencies (atomics)
};

constexpr size_t angle_embed_lds() { return sizeof(float) * (D * WSB + WAVES * TILE_ROWS * ETS); }

template <bool BWD, bool TRAIN = false>
__device__ __forceinline__ void angle_embed_body(const AngleEmbedTArgs& p, int vG, int vb) {
  static_assert(!TRAIN || BWD, "TRAIN is a variant of the adjoint kernel");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* We = smem;
  float* tiles = We + D * WSB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  if (BWD) stage_embed_split_t(reinterpret_cast<h16x8*>(We), p.w_emb, tid);
  else stage_embed_split(reinterpret_cast<h16x8*>(We), p.w_emb, tid);
  // basis index k = 16*kt + 4*g + r:  k = 0 const, 1..15 sin(f_{k-1} t), 16..30 cos(f_{k-16} t), 31 padding
  float fs[4], fc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ks = 4 * g + r;            // kt = 0 : sin of freq[ks - 1]
    fs[r] = (ks >= 1) ? p.freq[ks - 1] : 0.f;
    fc[r] = (ks < NFREQ) ? p.freq[ks] : 0.f;   // kt = 1 : cos of freq[ks]   (ks = 15 is the padding column)
  }
  __syncthreads();
  float* T = tiles + wave * TILE_ROWS * ETS;
  float* Trow = T + j * ETS;
  const int ntiles = (p.n_angles + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range_of(ntiles, vG, vb, tb, te);
  float fas[4] = {0.f, 0.f, 0.f, 0.f}, fac[4] = {0.f, 0.f, 0.f, 0.f};   // TRAIN: d freq through the sin / cos columns of this lane
  // Software pipeline over tiles (three dependent memory round trips per tile -- indices, unit vectors, adjoint rows -- left the waves
  // waiting 55-73 % of their cycles): indices two tiles ahead, unit vectors and the adjoint rows one tile ahead.
  const int last_angle = p.n_angles - 1;
  auto angle_of = [&](int tile) { return min(tile * BLOCK_ROWS + wave * TILE_ROWS + j, last_angle); };   // clamped: loads stay valid
  int a_n = 0, d1_n = 0, d2_n = 0, a_n2 = 0, d1_n2 = 0, d2_n2 = 0;
  f32x4 u1_n = zero4(), u2_n = zero4();
  Rows64 gin_n;
  if (tb < te) {
    a_n = angle_of(tb); d1_n = p.a_d1[a_n]; d2_n = p.a_d2[a_n];
    a_n2 = angle_of(tb + 1); d1_n2 = p.a_d1[a_n2]; d2_n2 = p.a_d2[a_n2];
    u1_n = p.eu[d1_n]; u2_n = p.eu[d2_n];
    if (BWD) rows64_issue(gin_n, p.Gang, a_n, lane);
  }
  for (int tile = tb; tile < te; ++tile) {
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.n_angles - row0);
    const bool valid = j < nvalid;
    const int a = row0 + (valid ? j : 0);
    const int d1 = d1_n, d2 = d2_n;
    const f32x4 u1 = u1_n, u2 = u2_n;
    Rows64 gin_rows;
    if (BWD) gin_rows = gin_n;
    a_n = a_n2; d1_n = d1_n2; d2_n = d2_n2;
    if (tile + 1 < te) {
      u1_n = p.eu[d1_n]; u2_n = p.eu[d2_n];
      if (BWD) rows64_issue(gin_n, p.Gang, a_n, lane);
      a_n2 = angle_of(tile + 2); d1_n2 = p.a_d1[a_n2]; d2_n2 = p.a_d2[a_n2];
    }
    if (nvalid <= 0) continue;
    const float cosv = (u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2]) * KAPPA;   // encoders.py:144
    const float theta = acosf(cosv);
    f32x4 x[2], dx[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ks = 4 * g + r;
      float sn, cs;
      sincos_cw(fs[r] * theta, sn, cs);
      x[0][r] = ks == 0 ? INV_SQRT_2 * INV_SQRT_PI : sn * INV_SQRT_PI;
      dx[0][r] = ks == 0 ? 0.f : fs[r] * cs * INV_SQRT_PI;
      sincos_cw(fc[r] * theta, sn, cs);
      x[1][r] = ks < NFREQ ? cs * INV_SQRT_PI : 0.f;
      dx[1][r] = ks < NFREQ ? -fc[r] * sn * INV_SQRT_PI : 0.f;
    }
    if (!BWD) {
      V64 h = zero64();
      gemm_split<2, VT, true>(h.t, reinterpret_cast<const h16x8*>(We), D, x, j, g);
      write_dl<VT>(Trow, g, h.t);
      __builtin_amdgcn_wave_barrier();
      scatter_rows64<false>(T, ETS, p.ang0, a, nvalid, lane);
      __builtin_amdgcn_wave_barrier();
    } else {
      f32x4 t[2] = {zero4(), zero4()};
      V64 gin;
      rows64_commit(gin_rows, T, ETS, lane);
      __builtin_amdgcn_wave_barrier();
      read_dl<VT>(Trow, g, gin.t);
      embed_adjoint(t, We, gin, j, g);
      __builtin_amdgcn_wave_barrier();
      if (TRAIN && valid) {
        write_dl<2>(p.Xa + (size_t)a * KB, g, x);
        // d/df sin(f t) = t cos(f t) = t * dx_sin / f,  d/df cos(f t) = -t sin(f t) = t * dx_cos / f: reuse dx (zero where padded)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          fas[r] += fs[r] != 0.f ? t[0][r] * dx[0][r] * theta / fs[r] : 0.f;
          fac[r] += fc[r] != 0.f ? t[1][r] * dx[1][r] * theta / fc[r] : 0.f;
        }
      }
      float gtheta = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) gtheta += t[kt][r] * dx[kt][r];
      gtheta = quad_sum(gtheta);
      const float gcos = -gtheta / sqrtf(1.0f - cosv * cosv) * KAPPA;
      // lane g handles cartesian component g of this row.  The rows arrive sorted by their first bond (graph.py:283-327: runs of
      // n - 1 angles share d1): the first-bond terms leave as ONE atomic per run and tile, not one per angle -- these scattered
      // 4-byte atomics execute at the memory side (~19 G requests/s chip-wide) and were what bounded this kernel (12.8 M requests in
      // 0.79 ms); the second bonds of a run are all different and keep their per-angle atomics.
      __builtin_amdgcn_wave_barrier();
      if (g < 3) T[4 * j + g] = valid ? gcos * u2[g] : 0.f;
      if (g == 3) reinterpret_cast<int*>(T)[4 * j + 3] = valid ? d1 : -1;
      __builtin_amdgcn_wave_barrier();
      if (valid && g < 3) {
        const int* keys = reinterpret_cast<const int*>(T);
        if (j == 0 || keys[4 * (j - 1) + 3] != d1) {        // head of a run: sum it
          float run = 0.f;
          for (int r = j; r < nvalid && keys[4 * r + 3] == d1; ++r) run += T[4 * r + g];
          atomicAdd(p.Gu + 4 * (size_t)d1 + g, run);
        }
        atomicAdd(p.Gu + 4 * (size_t)d2 + g, gcos * u1[g]);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (TRAIN) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sa = fas[r], ca = fac[r];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        sa += __shfl_xor(sa, off);
        ca += __shfl_xor(ca, off);
      }
      const int ks = 4 * g + r;                 // sin column ks uses freq[ks - 1], cos column uses freq[ks]
      if (j == 0) {
        if (ks >= 1) atomicAdd(p.g_freq + ks - 1, sa);
        if (ks < NFREQ) atomicAdd(p.g_freq + ks, ca);
      }
    }
  }
}

template <bool BWD, bool TRAIN = false>
__global__ __launch_bounds__(BLOCK) CHG_EMBED_WAVES void k_angle_embed_t(AngleEmbedTArgs p) {
  angle_embed_body<BWD, TRAIN>(p, gridDim.x, blockIdx.x);
}

template <int UNUSED = 0>
__global__ __launch_bounds__(BLOCK) CHG_FOUR_WAVES void k_angle_embed_fwd_o4(AngleEmbedTArgs p) {
  angle_embed_body<false, false>(p, gridDim.x, blockIdx.x);
}

// Small batches: the three basis-expansion launches of a direction as ONE (blocks [0, g_bond1) are part 1 of the bond expansion,
// the next g_bond2 part 2, the rest the angle expansion; a count of zero leaves a body out).  They only share their inputs (bond
// vectors) -- three dependent launches of 6-14 us each become one of the longest's length.
struct EmbedAllArgs {
  BondEmbedTArgs b;
  AngleEmbedTArgs a;
  int g_bond1, g_bond2, g_angle;
};
template <bool BWD>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_embed_all(EmbedAllArgs p) {
  int b = blockIdx.x;
  if (b < p.g_bond1) { bond_embed_body<BWD, false, 1, true>(p.b, p.g_bond1, b); return; }
  b -= p.g_bond1;
  if (b < p.g_bond2) { bond_embed_body<BWD, false, 2, true>(p.b, p.g_bond2, b); return; }
  b -= p.g_bond2;
  angle_embed_body<BWD, false>(p.a, p.g_angle, b);
}

}  // namespace chg
