// mfma_tile.h -- wave-level building blocks shared by every gfx950 kernel of the engine.
//
// One wave (64 lanes) owns a tile of 16 rows (edges / angles / atoms / bonds) and keeps feature
// vectors in the accumulator ("D") layout of v_mfma_f32_16x16x4_f32:
//
//     lane = j + 16*g      j in [0,16) = tile row,  g in [0,4)
//     x[ft][r]  (ft = 16-feature tile, r in [0,4))  holds feature  F(ft,r,g) = 16*ft + 4*g + r  of row j
//
// i.e. a 64-wide vector is split over the 4 lanes {j, j+16, j+32, j+48}, each owning one float4 per
// 16-feature tile (16 registers per lane).  With the contraction index k enumerated in that same order
// the accumulator of one MFMA GEMM is directly the B operand of the next one ("swapped" form
// D[f][row] = sum_k W[f][k] * X[row][k]: weights are the A operand, rows are the columns of D), so
// chained layers never leave registers, and LayerNorm over the 64 features of a row is an in-lane sum
// plus two cross-lane exchanges (lane^16, lane^32).
//
// Why 16-row tiles: the conv kernels were latency-bound at one 32-row wave per SIMD (478-512 VGPRs,
// profiles/r01).  Halving the rows halves every per-lane vector, which fits the backward kernels in 256
// VGPRs -> 8 waves per workgroup = 2 waves per SIMD, so one wave's gathers / VALU run under the other's
// MFMAs, and a whole tile's gather (24 row loads) is in flight at once.
//
// fp32 MFMA is exact f32 (fmaf chain) at 64 FLOP/clk/SIMD; there is no xf32 on CDNA4.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// The one diagnostic build switch left: -DCHG_PHASE_TIMING (s_memtime stamps between the phases of the angle kernels, right results,
// slower).  It needs -DCHG_EXPERIMENTS next to it (chgnet_amd/build.py:build_variant adds it; tests/test_abi.py checks that the product
// flags define neither).  The wrong-result timing switches of rounds 1-3 (dropped atomics, a quarter of the MFMAs, ...) are gone from the
// sources; what they measured is in profiles/r01_sq_counters.md, r02_experiments.md, r03_experiments.md.
#if defined(CHG_PHASE_TIMING) && !defined(CHG_EXPERIMENTS)
#error "CHG_PHASE_TIMING is a diagnostic build: define CHG_EXPERIMENTS as well (never in a product build)"
#endif

namespace chg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 64;            // feature width (atom = bond = angle)
constexpr int TILE_ROWS = 16;    // rows per wave tile
constexpr int WAVES = 8;         // waves per workgroup (2 per SIMD)
constexpr int BLOCK = 64 * WAVES;
constexpr int BLOCK_ROWS = TILE_ROWS * WAVES;
constexpr int PAD = 4;           // floats of padding per LDS row: keeps 16-B alignment and spreads
                                 // ds_read_b128 of consecutive rows over the banks (row stride = 4 mod 64)
constexpr int VT = 4;            // 16-feature tiles per 64 features
constexpr float LN_EPS = 1e-5f;
constexpr int MFMA_R = 4;

struct V64 { f32x4 t[VT]; };     // 64 features of one row, this lane's share (16 floats)

__device__ __forceinline__ int dfeat(int ft, int r, int g) { return 16 * ft + 4 * g + r; }

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ V64 zero64() { return V64{{zero4(), zero4(), zero4(), zero4()}}; }

// ---- activations -------------------------------------------------------------------------------
// sigmoid(x) = 1 / (1 + 2^(-x log2 e)) on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp
// each) -- 4 VALU instructions instead of ~22 for expf + IEEE division, which made the conv kernels
// VALU-bound (profiles/r01 notes).  Saturates correctly: x -> -inf gives rcp(inf) = 0, x -> +inf gives 1.
#if defined(CHG_EXPERIMENTS) && defined(CHG_EXP_NO_TRANS)
// TIMING-ONLY variant build (wrong results; profiles/r05_experiments.md): the two quarter-rate transcendentals of every sigmoid replaced
// by full-rate arithmetic, to measure what share of a tile kernel's time they are
__device__ __forceinline__ float exp_rcp_stub(float t) { return 0.5f + 0.1f * t; }
#define CHG_EXP2(t) (1.0f + 0.05f * (t))
#define CHG_RCP(t) (2.0f - (t))
#else
#define CHG_EXP2(t) __builtin_amdgcn_exp2f(t)
#define CHG_RCP(t) __builtin_amdgcn_rcpf(t)
#endif
__device__ __forceinline__ float sigmoidf_(float x) {
  return CHG_RCP(1.0f + CHG_EXP2(-1.4426950408889634f * x));
}
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
// d/dx silu(x) = s (1 + x (1 - s))
__device__ __forceinline__ float dsiluf_(float x) {
  const float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}

// The same on four values: written on vectors so that the multiplies / adds lower to v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (two
// floats per VALU issue slot; the transcendentals stay one per slot).  The tile kernels are bound by vector-ALU issue.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 sigmoid2(f32x2 x) {
  f32x2 t = x * -1.4426950408889634f;
  t[0] = CHG_EXP2(t[0]); t[1] = CHG_EXP2(t[1]);
  t = t + 1.0f;
  t[0] = CHG_RCP(t[0]); t[1] = CHG_RCP(t[1]);
  return t;
}
__device__ __forceinline__ f32x4 sigmoid4(f32x4 x) {   // pair by pair: four values at once cost the forward kernels ~10 spilled registers
  const f32x2 lo = sigmoid2(f32x2{x[0], x[1]}), hi = sigmoid2(f32x2{x[2], x[3]});
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ float hsum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
#define CHG_EV(ft) _Pragma("unroll") for (int ft = 0; ft < VT; ++ft)

// elementwise loop over this lane's 16 floats of a 64-wide vector
#define CHG_EW(ft, r) _Pragma("unroll") for (int ft = 0; ft < VT; ++ft) _Pragma("unroll") for (int r = 0; r < 4; ++r)

// ---- global row addressing ------------------------------------------------------------------------------------------------
// Row `r` of a row-major float table with `ld` floats per row, at float column `c`.  CHG_ADDR32=1 forms the byte offset in 32 bits
// against the uniform table base (`global_load ... v_off, s[base:base+1]`: one 32-bit multiply-add and one VGPR per address instead
// of three 64-bit vector instructions and a register pair).  Measured (profiles/r05_experiments.md section 7): 231 -> 99 64-bit
// address instructions in the BondConv adjoint, -1.5 % of its vector instructions, -1 % of the step -- not worth the 4 GiB-per-table
// limit it brings (a 4096-structure batch has 4.2 GB of angle rows): the product keeps 64-bit offsets.
#ifndef CHG_ADDR32
#define CHG_ADDR32 0
#endif
template <class T>
__device__ __forceinline__ const T* grow(const float* __restrict__ base, unsigned r, int ld, int c) {
#if CHG_ADDR32
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (r * (unsigned)(ld * 4) + (unsigned)(c * 4)));
#else
  return reinterpret_cast<const T*>(base + (size_t)r * ld + c);
#endif
}
template <class T>
__device__ __forceinline__ T* grow(float* __restrict__ base, unsigned r, int ld, int c) {
#if CHG_ADDR32
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (r * (unsigned)(ld * 4) + (unsigned)(c * 4)));
#else
  return reinterpret_cast<T*>(base + (size_t)r * ld + c);
#endif
}

// ---- D-layout loads / stores ---------------------------------------------------------------
// `row` points at feature 0 of the lane's row in a row-major buffer (LDS tile or global table)
template <int NT>
__device__ __forceinline__ void read_dl(const float* row, int g, f32x4 (&x)[NT]) {
#pragma unroll
  for (int ft = 0; ft < NT; ++ft) x[ft] = *reinterpret_cast<const f32x4*>(row + 16 * ft + 4 * g);
}
template <int NT>
__device__ __forceinline__ void write_dl(float* row, int g, const f32x4 (&x)[NT]) {
#pragma unroll
  for (int ft = 0; ft < NT; ++ft) *reinterpret_cast<f32x4*>(row + 16 * ft + 4 * g) = x[ft];
}

// the same for row `r` of a global table (32-bit offsets: grow above)
template <int NT>
__device__ __forceinline__ void read_dl_g(const float* __restrict__ base, unsigned r, int ld, int g, f32x4 (&x)[NT]) {
#pragma unroll
  for (int ft = 0; ft < NT; ++ft) x[ft] = *grow<f32x4>(base, r, ld, 16 * ft + 4 * g);
}
template <int NT>
__device__ __forceinline__ void write_dl_g(float* __restrict__ base, unsigned r, int ld, int g, const f32x4 (&x)[NT]) {
#pragma unroll
  for (int ft = 0; ft < NT; ++ft) *grow<f32x4>(base, r, ld, 16 * ft + 4 * g) = x[ft];
}

// streaming forms (non-temporal: the lines are not kept in L2, which the table gathers of the same kernel live on)
template <int NT>
__device__ __forceinline__ void read_dl_g_nt(const float* __restrict__ base, unsigned r, int ld, int g, f32x4 (&x)[NT]) {
#pragma unroll
  for (int ft = 0; ft < NT; ++ft) x[ft] = __builtin_nontemporal_load(grow<f32x4>(base, r, ld, 16 * ft + 4 * g));
}
template <int NT>
__device__ __forceinline__ void write_dl_g_nt(float* __restrict__ base, unsigned r, int ld, int g, const f32x4 (&x)[NT]) {
#pragma unroll
  for (int ft = 0; ft < NT; ++ft) __builtin_nontemporal_store(x[ft], grow<f32x4>(base, r, ld, 16 * ft + 4 * g));
}

// ---- MFMA GEMMs in swapped form ---------------------------------------------------------------
// acc[fo] (fo < NFT) += W[16*fo + i][k] * x[k]   for k over 16*KT inputs held in D layout.
// W: LDS, row-major [out][ws] (ws = K + PAD): the A operand of lane (i,g) is read as float4.
// Consecutive MFMAs go to different accumulators (16x16x4: 32-cycle issue, 40-cycle dependent latency).
template <int KT, int NFT>
__device__ __forceinline__ void gemm_dl(f32x4 (&acc)[NFT], const float* W, int ws, const f32x4 (&x)[KT], int i, int g) {
  const float* wbase = W + i * ws + 4 * g;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    f32x4 w[NFT];
#pragma unroll
    for (int fo = 0; fo < NFT; ++fo) w[fo] = *reinterpret_cast<const f32x4*>(wbase + 16 * fo * ws + 16 * kt);
#pragma unroll
    for (int r = 0; r < MFMA_R; ++r)
#pragma unroll
      for (int fo = 0; fo < NFT; ++fo) acc[fo] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[fo][r], x[kt][r], acc[fo], 0, 0, 0);
  }
}

// Transposed use of the same LDS copy: acc[fo] += W[k][16*fo + i] * x[k]  (W row-major [k][ws]).
// Lanes i are consecutive in memory -> ds_read_b32, conflict-free.
// The 4 x NFT operand reads of a k-group are issued together and one group ahead of the MFMAs that
// consume them.  Left to itself the scheduler sinks each read next to its MFMA (read -> wait -> 2
// MFMAs -> read ...), which exposes the LDS latency on every pair: 1.9x slower than the forward form.
// The empty asm is a use of the whole group: it pins the (counted) wait in front of the group's MFMAs.
template <int NFT>
__device__ __forceinline__ void gemm_t_load(float (&a)[4][NFT], const float* W, int ws, int kt, int i, int g) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float* wrow = W + dfeat(kt, r, g) * ws + i;
#pragma unroll
    for (int fo = 0; fo < NFT; ++fo) a[r][fo] = wrow[16 * fo];
  }
}
template <int NFT>
__device__ __forceinline__ void gemm_t_touch(float (&a)[4][NFT]) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int fo = 0; fo < NFT; ++fo) asm volatile("" : "+v"(a[r][fo]));
}
template <int KT, int NFT>
__device__ __forceinline__ void gemm_dl_t(f32x4 (&acc)[NFT], const float* W, int ws, const f32x4 (&x)[KT], int i, int g) {
  float a[2][4][NFT];
  gemm_t_load<NFT>(a[0], W, ws, 0, i, g);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) gemm_t_load<NFT>(a[(kt + 1) & 1], W, ws, kt + 1, i, g);
    gemm_t_touch<NFT>(a[kt & 1]);
#pragma unroll
    for (int r = 0; r < MFMA_R; ++r)
#pragma unroll
      for (int fo = 0; fo < NFT; ++fo) acc[fo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt & 1][r][fo], x[kt][r], acc[fo], 0, 0, 0);
  }
}

// ---- LayerNorm over the 64 features of a row (spread over 4 lanes) ------------------------------
// Sum over the four 16-lane rows of a wave (the four lanes that share a tile row), result in all of
// them.  gfx950's row swaps do it in the VALU: v_permlane16_swap exchanges the odd rows of one operand
// with the even rows of the other (both copies of v: a + b = row0+row1 | row2+row3), v_permlane32_swap
// the upper half of one with the lower half of the other.  Two ds_bpermute round trips through the LDS
// crossbar before.
__device__ __forceinline__ float quad_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const unsigned w = __float_as_uint(s);
  const auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// in: c (pre-norm).  out: c <- xhat, returns rstd.
__device__ __forceinline__ float ln_normalize(V64& c) {
  f32x4 s4 = (c.t[0] + c.t[1]) + (c.t[2] + c.t[3]);
  const float mu = quad_sum(hsum4(s4)) * (1.0f / 64.0f);
  f32x4 v4 = zero4();
  CHG_EV(ft) {
    c.t[ft] = c.t[ft] - mu;
    v4 += c.t[ft] * c.t[ft];
  }
  const float rstd = __builtin_amdgcn_rsqf(quad_sum(hsum4(v4)) * (1.0f / 64.0f) + LN_EPS);
  CHG_EV(ft) c.t[ft] = c.t[ft] * rstd;
  return rstd;
}

// LayerNorm backward: gy <- rstd * (gx - mean(gx) - xhat * mean(gx*xhat)),  gx = gy * gamma
__device__ __forceinline__ void ln_backward(V64& gy, const V64& gamma, const V64& xhat, float rstd) {
  f32x4 s1 = zero4(), s2 = zero4();
  CHG_EV(ft) {
    gy.t[ft] = gy.t[ft] * gamma.t[ft];
    s1 += gy.t[ft];
    s2 += gy.t[ft] * xhat.t[ft];
  }
  const float m1 = quad_sum(hsum4(s1)) * (1.0f / 64.0f);
  const float m2 = quad_sum(hsum4(s2)) * (1.0f / 64.0f);
  CHG_EV(ft) gy.t[ft] = (gy.t[ft] - m1 - xhat.t[ft] * m2) * rstd;
}

// ---- LDS staging of a row-major weight matrix [rows][K] -> [rows][K+PAD] ------------------------
__device__ __forceinline__ void stage_weights(float* dst, const float* __restrict__ src, int rows, int K, int tid) {
  const int ws = K + PAD;
  const int k4 = K / 4;
  for (int idx = tid; idx < rows * k4; idx += (int)blockDim.x) {
    const int rr = idx / k4, c4 = idx - rr * k4;
    *reinterpret_cast<f32x4*>(dst + rr * ws + 4 * c4) = *reinterpret_cast<const f32x4*>(src + rr * K + 4 * c4);
  }
}
__device__ __forceinline__ void stage_vector(float* dst, const float* __restrict__ src, int n, int tid) {
  for (int idx = tid; idx < n; idx += (int)blockDim.x) dst[idx] = src[idx];
}

// ---- prologue from a PREBUILT image: the weight block of a tile kernel's LDS (split / row-major f16 images + parameter vectors) as
// the kernel lays it out, built once per weight upload by k_*_image (kernels_conv.h) -- the prologue is then N16 16-byte copies, every
// load of a thread in flight before its first store, instead of strided fp32 reads + the hi / lo split in every workgroup of every
// launch (measured on a 256-atom MD cell, two wave-tiles per wave: 12-17k of a launch's 50-125k clocks per wave were the prologue).
template <int N16, int NT, int CHUNK = 16>   // CHUNK: loads of a thread in flight together (registers: 4 each)
__device__ __forceinline__ void stage_image(float* dst, const float* __restrict__ src, int tid) {
  constexpr int IT = (N16 + NT - 1) / NT;
#pragma unroll
  for (int u0 = 0; u0 < IT; u0 += CHUNK) {
    f32x4 v[CHUNK];
#pragma unroll
    for (int u = 0; u < CHUNK; ++u) {
      const int idx = tid + (u0 + u) * NT;
      if (u0 + u < IT && idx < N16) v[u] = reinterpret_cast<const f32x4*>(src)[idx];
    }
#pragma unroll
    for (int u = 0; u < CHUNK; ++u) {
      const int idx = tid + (u0 + u) * NT;
      if (u0 + u < IT && idx < N16) reinterpret_cast<f32x4*>(dst)[idx] = v[u];
    }
  }
}

// ---- tile scheduling at WAVE granularity: wave w of logical workgroup lb takes a contiguous range of 16-row wave-tiles; the
// surplus tiles (n mod waves) go one each to different waves.  With workgroup-granular ranges a batch of 4,221 wave-tiles on
// 2,048 wave slots (a 256-atom MD cell) made 16 workgroups run a third round with all eight waves; now 125 waves do, alone on
// their SIMDs.  Large batches: every wave a long contiguous row range, neighbouring ranges on one XCD as before.
__device__ __forceinline__ void wave_tile_range(int n_wave_tiles, int waves_per_block, int wave, int& begin, int& end) {
  const int G = gridDim.x, b = blockIdx.x;
  int lb = b;
  if ((G & 7) == 0) lb = (b & 7) * (G >> 3) + (b >> 3);
  const long W = (long)G * waves_per_block, lw = (long)lb * waves_per_block + wave;
  begin = (int)(lw * n_wave_tiles / W);            // evenly spaced boundaries: the surplus tiles land on every (W / surplus)-th wave,
  end = (int)((lw + 1) * n_wave_tiles / W);        // not on the first workgroups
}

// ---- tile scheduling of the large-batch tile kernels: XCD-local interleaved sweeps ------------------------------------------
// With contiguous per-wave ranges the 256 waves of an XCD sit in 256 different places of the batch, i.e. in the tables of ~128
// different structures (30 MB against a 4 MiB L2: hit rates of 24-49 %, fabric traffic 1.5-2.9x the compulsory bytes,
// profiles/r03_l2_counters.csv).  Here XCD x (= blockIdx & 7, the observed dispatch: speed only) owns the x-th eighth of the tiles and
// its waves walk it side by side: wave lw takes tiles xb + lw, xb + lw + WX, ... (WX = waves of the XCD), the eight waves of a
// workgroup on eight consecutive tiles.  At any moment an XCD then works inside one or two structures; every table row is fetched
// from the fabric once.  Small batches (a few tiles per wave: MD) keep the evenly spaced contiguous ranges of wave_tile_range.
#ifndef CHG_TILE_INTERLEAVE
#define CHG_TILE_INTERLEAVE 1
#endif
constexpr int INTERLEAVE_MIN_TILES = 6;      // per wave
struct TileSeq {
  int first, stride, count;
  __device__ __forceinline__ int at(int v) const { return first + v * stride; }
};
__device__ __forceinline__ TileSeq wave_tile_seq(int n_wave_tiles, int waves_per_block, int wave, int interleave = 1) {
  const int G = gridDim.x, b = blockIdx.x;
  TileSeq s;
  if (CHG_TILE_INTERLEAVE && interleave && (G & 7) == 0 && (long)n_wave_tiles >= (long)INTERLEAVE_MIN_TILES * G * waves_per_block) {
    const int x = b & 7, WX = (G >> 3) * waves_per_block, lw = (b >> 3) * waves_per_block + wave;
    const int xb = (int)((long)n_wave_tiles * x / 8), xe = (int)((long)n_wave_tiles * (x + 1) / 8);
    s.first = xb + lw;
    s.stride = WX;
    s.count = xe - xb > lw ? (xe - xb - lw + WX - 1) / WX : 0;
    return s;
  }
  int tb, te;
  wave_tile_range(n_wave_tiles, waves_per_block, wave, tb, te);
  s.first = tb; s.stride = 1; s.count = te - tb;
  return s;
}

// XCD of the executing wave: hwreg(HW_REG_XCC_ID, 0, 4) (speed only: which of the per-XCD work queues a wave starts with)
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u); }

// ---- tile scheduling: contiguous tile ranges per workgroup, neighbouring ranges on one XCD -------
// Workgroup b is dispatched to XCD b % 8 (observed, speed only); giving XCD x the logical blocks
// [x*G/8, (x+1)*G/8) keeps one structure's tables inside one XCD's L2.
// (tile_range_of: the same for a kernel that runs as one of several bodies of a merged launch -- its own block count and index)
__device__ __forceinline__ void tile_range_of(int ntiles, int G, int b, int& begin, int& end) {
  int lb = b;
  if ((G & 7) == 0) lb = (b & 7) * (G >> 3) + (b >> 3);
  const int per = ntiles / G, rem = ntiles - per * G;
  begin = lb * per + (lb < rem ? lb : rem);
  end = begin + per + (lb < rem ? 1 : 0);
}
__device__ __forceinline__ void tile_range(int ntiles, int& begin, int& end) { tile_range_of(ntiles, gridDim.x, blockIdx.x, begin, end); }

// The tile kernels run 8 waves per CU (LDS-limited), i.e. two per SIMD, whatever their register count:
// telling the compiler lets its scheduler spend the 256-register budget on overlap instead of
// trading instruction-level parallelism for an occupancy it cannot have.
#define CHG_TWO_WAVES __attribute__((amdgpu_waves_per_eu(2, 2)))

// fp32 add to global memory (global_atomic_add_f32, no return: -munsafe-fp-atomics, no CAS loop)
__device__ __forceinline__ void tile_atomic_add(float* p, float v) {
  atomicAdd(p, v);
}

// ---- segmented reductions over the rows of a wave tile -----------------------------------------
// tile: [16][stride] LDS, W columns; lane `rr` holds the (sorted-run) key of row rr in `key`
// (key < 0: skip).  Runs of equal keys are summed per column and flushed with one fp32 atomic
// per (run, column); runs that continue in another tile meet in memory.
template <int W>
__device__ __forceinline__ void seg_colsum_atomic(const float* tile, int stride, int key, int nvalid, float* __restrict__ dst,
                                                  int ldd, int lane) {
  constexpr int NC = W / 64;
  float acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.f;
  int cur = __builtin_amdgcn_readlane(key, 0);   // row 0 is always valid (nvalid >= 1)
#pragma unroll
  for (int rr = 0; rr < TILE_ROWS; ++rr) {
    const int kk = __builtin_amdgcn_readlane(key, rr);   // rows past the end carry key -1
    if (kk != cur) {
      if (cur >= 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) tile_atomic_add(grow<float>(dst, (unsigned)cur, ldd, 64 * c + lane), acc[c]);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = 0.f;
      cur = kk;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] += tile[rr * stride + 64 * c + lane];
  }
  if (cur >= 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) tile_atomic_add(grow<float>(dst, (unsigned)cur, ldd, 64 * c + lane), acc[c]);
  }
}

// every row goes to its own (unsorted) destination row: one 256-B coalesced atomic per 64 columns
template <int W>
__device__ __forceinline__ void row_atomic_add(const float* tile, int stride, int key, int nvalid, float* __restrict__ dst,
                                               int ldd, int lane) {
#pragma unroll
  for (int rr = 0; rr < TILE_ROWS; ++rr) {
    const int kk = __builtin_amdgcn_readlane(key, rr);   // rows past the end carry key -1
    if (kk >= 0) {
#pragma unroll
      for (int c = 0; c < W / 64; ++c) tile_atomic_add(grow<float>(dst, (unsigned)kk, ldd, 64 * c + lane), tile[rr * stride + 64 * c + lane]);
    }
  }
}

// ---- row-spread gather: sum of three table rows, 128 floats wide, into an LDS tile ---------------
// Half-wave `hw` (32 lanes) handles one row per step; lane t loads float4 #t of the 512-B row, so
// every load instruction covers two full rows and all 24 loads of a tile are in flight together.
// Lane rr (< 16) holds the three row indices of tile row rr; they must be valid (clamped) even for
// rows past the end of the problem.
__device__ __forceinline__ void gather_sum128(float* tile, int stride, const float* __restrict__ t0, int i0,
                                              const float* __restrict__ t1, int i1, const float* __restrict__ t2, int i2,
                                              int ld0, int ld1, int ld2, int lane) {
  const int hw = lane >> 5, t = lane & 31;
  // all index shuffles first (independent ds_bpermutes, one wait), then all 24 row loads back to back:
  // interleaving shuffle -> wait -> load serialised the issue of the gather (~100 cycles per row)
  int r0[TILE_ROWS / 2], r1[TILE_ROWS / 2], r2[TILE_ROWS / 2];
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it) {
    const int rr = 2 * it + hw;
    r0[it] = __shfl(i0, rr); r1[it] = __shfl(i1, rr); r2[it] = __shfl(i2, rr);
  }
  f32x4 a[TILE_ROWS / 2], b[TILE_ROWS / 2], c[TILE_ROWS / 2];
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it) {
    a[it] = *grow<f32x4>(t0, (unsigned)r0[it], ld0, 4 * t);
    b[it] = *grow<f32x4>(t1, (unsigned)r1[it], ld1, 4 * t);
    c[it] = *grow<f32x4>(t2, (unsigned)r2[it], ld2, 4 * t);
  }
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it) {
    const int rr = 2 * it + hw;
    *reinterpret_cast<f32x4*>(tile + rr * stride + 4 * t) = (a[it] + b[it]) + c[it];
  }
}

// The same gather split in two so that the loads of tile t+1 can be in flight while tile t computes:
// issue (registers only) ... commit (sum + LDS write).
struct GatherRegs { f32x4 a[TILE_ROWS / 2], b[TILE_ROWS / 2], c[TILE_ROWS / 2]; };
__device__ __forceinline__ void gather_take(GatherRegs& gr);   // (defined below, next to GatherPH's)

__device__ __forceinline__ void gather_issue128(GatherRegs& gr, const float* __restrict__ t0, int i0, const float* __restrict__ t1,
                                                int i1, const float* __restrict__ t2, int i2, int ld0, int ld1, int ld2, int lane) {
  const int hw = lane >> 5, t = lane & 31;
  int r0[TILE_ROWS / 2], r1[TILE_ROWS / 2], r2[TILE_ROWS / 2];
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it) {
    const int rr = 2 * it + hw;
    r0[it] = __shfl(i0, rr); r1[it] = __shfl(i1, rr); r2[it] = __shfl(i2, rr);
  }
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it) {
    gr.a[it] = *grow<f32x4>(t0, (unsigned)r0[it], ld0, 4 * t);
    gr.b[it] = *grow<f32x4>(t1, (unsigned)r1[it], ld1, 4 * t);
    gr.c[it] = *grow<f32x4>(t2, (unsigned)r2[it], ld2, 4 * t);
  }
}
__device__ __forceinline__ void gather_commit128(const GatherRegs& gr, float* tile, int stride, int lane) {
  const int hw = lane >> 5, t = lane & 31;
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it)
    *reinterpret_cast<f32x4*>(tile + (2 * it + hw) * stride + 4 * t) = (gr.a[it] + gr.b[it]) + gr.c[it];
}

// Two 128-wide table rows summed + one 64-wide row per tile row (its address as an offset from `hbase`, in floats): the AtomConv
// gather when the bond partial is contracted in the kernel (kernels_conv.h FUSEQ) instead of read from a table.
struct GatherPH { f32x4 a[TILE_ROWS / 2], b[TILE_ROWS / 2], h[TILE_ROWS / 4]; };
// "Take" requested rows: an empty asm on every register, i.e. a wait placed by hand.  Software-pipelined kernels request tile t+1's
// rows early in tile t and commit them at the top of t+1; left to the compiler, the wait sits at the loop latch, right behind the
// tile's closing atomics -- and the counter being in order, it waits for those.  Taken before the tile's first atomic, the rows
// (requested thousands of cycles earlier) cost nothing and the latch waits for nothing.
__device__ __forceinline__ void gather_take(GatherPH& gr) {
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it) asm volatile("" : "+v"(gr.a[it]), "+v"(gr.b[it]));
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) asm volatile("" : "+v"(gr.h[it]));
}
__device__ __forceinline__ void gather_take(GatherRegs& gr) {
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it) asm volatile("" : "+v"(gr.a[it]), "+v"(gr.b[it]), "+v"(gr.c[it]));
}
__device__ __forceinline__ void gather_issue_ph(GatherPH& gr, const float* __restrict__ t0, int i0, const float* __restrict__ t1, int i1, int ld0,
                                                int ld1, const float* __restrict__ hbase, long hoff, int lane) {
  const int hw = lane >> 5, t = lane & 31, sub = lane >> 4, t16 = lane & 15;
  int r0[TILE_ROWS / 2], r1[TILE_ROWS / 2];
  long ho[TILE_ROWS / 4];
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it) {
    const int rr = 2 * it + hw;
    r0[it] = __shfl(i0, rr); r1[it] = __shfl(i1, rr);
  }
  const int hlo = (int)(hoff & 0xffffffffL), hhi = (int)(hoff >> 32);
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    const int rr = 4 * it + sub;
    ho[it] = ((long)__shfl(hhi, rr) << 32) | (unsigned)__shfl(hlo, rr);
  }
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it) {
    gr.a[it] = *grow<f32x4>(t0, (unsigned)r0[it], ld0, 4 * t);
    gr.b[it] = *grow<f32x4>(t1, (unsigned)r1[it], ld1, 4 * t);
  }
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) gr.h[it] = *reinterpret_cast<const f32x4*>(hbase + ho[it] + 4 * t16);
}
__device__ __forceinline__ void gather_commit_p(const GatherPH& gr, float* tile, int stride, int lane) {
  const int hw = lane >> 5, t = lane & 31;
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 2; ++it) *reinterpret_cast<f32x4*>(tile + (2 * it + hw) * stride + 4 * t) = gr.a[it] + gr.b[it];
}
__device__ __forceinline__ void gather_commit_h(const GatherPH& gr, float* tile, int stride, int lane) {
  const int sub = lane >> 4, t = lane & 15;
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) *reinterpret_cast<f32x4*>(tile + (4 * it + sub) * stride + 4 * t) = gr.h[it];
}

// The read half of a row-wise read-modify-write, issued early (its round trip then runs under the
// contraction that produces the increment); scatter_rows64_add finishes it.
struct Rows64 { f32x4 v[TILE_ROWS / 4]; };
__device__ __forceinline__ void rows64_issue(Rows64& rr, const float* __restrict__ src, int idx, int lane) {
  const int sub = lane >> 4, t = lane & 15;
  int r[TILE_ROWS / 4];
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) r[it] = __shfl(idx, 4 * it + sub);
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) rr.v[it] = *grow<f32x4>(src, (unsigned)r[it], D, 4 * t);
}
__device__ __forceinline__ void rows64_commit(const Rows64& rr, float* tile, int stride, int lane) {
  const int sub = lane >> 4, t = lane & 15;
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) *reinterpret_cast<f32x4*>(tile + (4 * it + sub) * stride + 4 * t) = rr.v[it];
}
template <bool OPAQUE = false>
__device__ __forceinline__ void scatter_rows64_add(const float* tile, int stride, float* __restrict__ dst, int idx, int nvalid, int lane,
                                                   const Rows64& old) {
  const int sub = lane >> 4, t = lane & 15;
  // all sums first, then the stores: a wait for an old row placed between the (conditional, hence uncounted) stores is a wait for the
  // stores before it
  f32x4 v[TILE_ROWS / 4];
  int r[TILE_ROWS / 4];
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    const int rr = 4 * it + sub;
    r[it] = __shfl(idx, rr);
    v[it] = old.v[it] + *reinterpret_cast<const f32x4*>(tile + rr * stride + 4 * t);
  }
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) asm volatile("" : "+v"(v[it]));   // (the scheduler re-interleaves the two loops otherwise)
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    // OPAQUE (per-atom BondConv adjoint): compared as sub < nvalid - 4 it with the right side hidden from the optimiser.  Written as
    // rr < nvalid the compiler kept 4 it + sub for every `it` in registers across the kernel, spilled them and reloaded each BETWEEN
    // these stores: the wait for a scratch reload is a wait for every store before it -- 3 store round trips per tile
    int lim = nvalid - 4 * it;
    if (OPAQUE) asm volatile("" : "+v"(lim));
    if (sub < lim) *grow<f32x4>(dst, (unsigned)r[it], D, 4 * t) = v[it];
  }
}

// contiguous or gathered 64-wide rows into an LDS tile (16 lanes per row, 4 rows per step)
__device__ __forceinline__ void gather_rows64(float* tile, int stride, const float* __restrict__ src, int idx, int lane) {
  const int sub = lane >> 4, t = lane & 15;
  f32x4 v[TILE_ROWS / 4];
  int r[TILE_ROWS / 4];
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) r[it] = __shfl(idx, 4 * it + sub);
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) v[it] = *grow<f32x4>(src, (unsigned)r[it], D, 4 * t);
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) *reinterpret_cast<f32x4*>(tile + (4 * it + sub) * stride + 4 * t) = v[it];
}

// LDS tile (64 wide) -> global rows (coalesced 256-B rows), plain store or read-modify-write add
template <bool ACCUM>
__device__ __forceinline__ void scatter_rows64(const float* tile, int stride, float* __restrict__ dst, int idx, int nvalid,
                                               int lane) {
  const int sub = lane >> 4, t = lane & 15;
  unsigned r[TILE_ROWS / 4];
  f32x4 v[TILE_ROWS / 4];
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) r[it] = (unsigned)__shfl(idx, 4 * it + sub);
  if (ACCUM) {   // all loads first: one memory round trip for the tile, not one per step (idx of rows past nvalid is a valid row)
#pragma unroll
    for (int it = 0; it < TILE_ROWS / 4; ++it) v[it] = *grow<f32x4>(dst, r[it], D, 4 * t);
#pragma unroll
    for (int it = 0; it < TILE_ROWS / 4; ++it) v[it] += *reinterpret_cast<const f32x4*>(tile + (4 * it + sub) * stride + 4 * t);
  } else {
#pragma unroll
    for (int it = 0; it < TILE_ROWS / 4; ++it) v[it] = *reinterpret_cast<const f32x4*>(tile + (4 * it + sub) * stride + 4 * t);
  }
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it)
    if (sub < nvalid - 4 * it) *grow<f32x4>(dst, r[it], D, 4 * t) = v[it];
}

}  // namespace chg
