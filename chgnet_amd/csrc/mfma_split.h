// mfma_split.h -- fp32-accurate 64-wide contractions on the f16 matrix pipe (gfx950).
//
// v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate (32 cycles per 2,048 flop per SIMD); the six
// message-passing kernels spend 16k of a tile's 30k cycles in it (profiles/r02_experiments.md: a four
// times faster matrix pipe is +24 % on the step, +80 % once the adjoints' atomics are out of the way).
// v_mfma_f32_16x16x32_f16 does 16,384 flop in 16 cycles, so three of them on split operands
//
//     x = xh + xl,  w = wh + wl      (xh = f16(x) round-to-nearest, xl = f16(x - xh): 22 significand bits)
//     x.w ~= xh.wh + xl.wh + xh.wl   (the dropped xl.wl term is 2^-22 relative), f32 accumulation
//
// cost 3 x 2 x 16 = 96 cycles per 16x16x64 block against 512 for the f32 form: 5.3x less matrix-pipe
// time at a per-product error of ~2^-22 (f32: 2^-24).  Measured on the float64 pipeline model
// (tests/test_split_numerics.py): the split contractions add 2e-8 eV/atom, 1.5e-6 eV/A, 1.2e-5 GPa at
// trained-checkpoint magnitudes -- a tenth of the f32 pipeline's own rounding error (1.4e-5 eV/A).
//
// f16 range: forward operands (activations O(1), weights) are used as they are; the adjoint operands
// (gradient rows, 1e-3 .. 1e-7) are scaled per row by a power of two first (exact), else their low
// halves fall into the f16 subnormals (measured: 10x the error).  The low planes of both operands are
// carried scaled by 2^11 so that they stay normal f16 numbers: the matrix pipe does keep f16 subnormals
// (tools/split_lab.hip T2), but their fixed 2^-24 spacing costs bits -- with unscaled low planes the force
// error at trained-checkpoint magnitudes was 2.4x the f32 engine's (5.5e-5 vs 2.3e-5 eV/A on the 32-atom golden
// case), with scaled ones it is the f32 engine's to two digits (tools/gpu_precision_probe.py).
//
// Layouts.  Rows keep the accumulator ("D") layout of mfma_tile.h: lane = j + 16 g holds features
// 16 ft + 4 g + r of row j.  One K = 32 MFMA takes from lane (j, g) the 8 values
//     k(mk, g, e) = 32 mk + 16 (e >> 2) + 4 g + (e & 3),   e = 0..7
// i.e. exactly this lane's registers of feature tiles 2 mk and 2 mk + 1: chained layers still never
// leave registers.  The hardware pairs element e of lane group g of A with the same (g, e) of B, so any
// such enumeration of k is a valid contraction order as long as the weights are stored to match:
//     LDS image  [plane hi|lo][mk][g][f][8 x f16]     (16-byte chunk per (mk, g, f))
// A operand of output tile fo for lane (i, g): chunk ((plane * MK + mk) * 4 + g) * F + 16 fo + i --
// one ds_read_b128, conflict-free (the 16 lanes of every b128 group hold 16 different i, and the g
// stride F * 16 B is a multiple of the 256-B bank row).  The adjoint contractions (sum over the OUTPUT
// index of W) use a second image built from W^T.
#pragma once

#include "mfma_tile.h"

namespace chg {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

#ifndef CHG_SPLIT_LO_SEPARATE
#define CHG_SPLIT_LO_SEPARATE 1
#endif
// CHG_WIDE_RANGE (engine_predict_wide.hip): EVERY operand row is scaled by a power of two before the split, not only the adjoint rows
#ifdef CHG_WIDE_RANGE
constexpr bool WIDE_RANGE = true;
#else
constexpr bool WIDE_RANGE = false;
#endif
constexpr bool LO_SEPARATE = CHG_SPLIT_LO_SEPARATE;   // low planes scaled by 2^11 in their own accumulator (see above)
constexpr float LO_SCALE = LO_SEPARATE ? 2048.0f : 1.0f, LO_UNSCALE = 1.0f / LO_SCALE;

// bytes of one split image of a [F][K] matrix (both planes)
constexpr size_t split_image_bytes(int F, int K) { return (size_t)F * K * 4; }

// ---- staging: global fp32 W[F][K] (row-major, ld = ldw) -> LDS split image ------------------------
// TRANSPOSE: the image of W^T, i.e. the contraction then runs over the rows of W (F_img = K, K_img = F).
template <bool TRANSPOSE>
__device__ __forceinline__ void stage_split(h16x8* img, const float* __restrict__ W, int F, int K, int tid, int nthreads) {
  const int Fi = TRANSPOSE ? K : F, Ki = TRANSPOSE ? F : K;      // image dims: Fi outputs, Ki contraction
  const int MK = Ki / 32;
  const int nchunks = MK * 4 * Fi;
  for (int c = tid; c < nchunks; c += nthreads) {
    const int f = c % Fi, g = (c / Fi) & 3, mk = c / (4 * Fi);
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * mk + 16 * (e >> 2) + 4 * g + (e & 3);
      w[e] = TRANSPOSE ? W[(size_t)k * K + f] : W[(size_t)f * K + k];
    }
    h16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      hi[e] = (_Float16)w[e];
      lo[e] = (_Float16)((w[e] - (float)hi[e]) * LO_SCALE);
    }
    img[c] = hi;
    img[nchunks + c] = lo;
  }
}

// ---- operand split of 32 contraction values held by this lane (feature tiles 2 mk, 2 mk + 1) -------
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, h16x8& hi, h16x8& lo) {
  f32x4 ha, hb;      // the high halves back in f32; the residuals on vectors (packed subtract / multiply)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (_Float16)a[e];
    hi[4 + e] = (_Float16)b[e];
    ha[e] = (float)hi[e];
    hb[e] = (float)hi[4 + e];
  }
  const f32x4 la = (a - ha) * LO_SCALE, lb = (b - hb) * LO_SCALE;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    lo[e] = (_Float16)la[e];
    lo[4 + e] = (_Float16)lb[e];
  }
}

// max over the four lanes that share a tile row (cf. quad_sum)
__device__ __forceinline__ float quad_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const unsigned w = __float_as_uint(s);
  const auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

// power-of-two row scale that brings the largest |x| of the row into [1, 2): returns the exponent e with
// max|x| in [2^e, 2^(e+1)); rows of zeros (and non-finite rows) get e = 0
template <int KT>
__device__ __forceinline__ int row_exponent(const f32x4 (&x)[KT]) {
  float m = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) m = fmaxf(fmaxf(m, fmaxf(fabsf(x[kt][0]), fabsf(x[kt][1]))), fmaxf(fabsf(x[kt][2]), fabsf(x[kt][3])));
  m = quad_max(m);
  const int e = __builtin_amdgcn_frexp_expf(m) - 1;          // m = f * 2^(e+1), f in [0.5, 1)
  return (m > 0.f && m < 3.0e38f) ? max(-125, min(125, e)) : 0;   // clamped: 2^e and 2^-e are both normal numbers
}

// ---- the contraction ------------------------------------------------------------------------------
// acc[fo] += sum_k W[16 fo + i][k] x[k],  k over the 16 KT values this row holds in D layout.
// img: split image with F outputs (as staged above).  SCALED: x is an adjoint row -- scaled per row by a
// power of two before the split, the product scaled back.  Output tiles are produced four at a time so
// that the operands of one K = 32 step (4 x (hi, lo) weights = 32 registers) and the accumulators stay small.
template <int MK>
struct SplitRow { h16x8 hi[MK], lo[MK]; float up; };   // up = 2^e of a SCALED row (the product is multiplied back by it: exact)

template <int KT, bool SCALED>
__device__ __forceinline__ void split_row(SplitRow<KT / 2>& s, const f32x4 (&x)[KT]) {
  static_assert(KT % 2 == 0, "K must be a multiple of 32");
  s.up = 1.0f;
  float down = 1.0f;
  if (SCALED) {
    const int ex = row_exponent<KT>(x);
    s.up = __builtin_ldexpf(1.0f, ex);
    down = __builtin_ldexpf(1.0f, -ex);
  }
#pragma unroll
  for (int mk = 0; mk < KT / 2; ++mk) {
    f32x4 a = x[2 * mk], b = x[2 * mk + 1];
    if (SCALED) { a = a * down; b = b * down; }     // powers of two: exact (two values per v_pk_mul_f32 instead of one v_ldexp_f32 each)
    split8(a, b, s.hi[mk], s.lo[mk]);
  }
}

// TIMING-ONLY (wrong results; -DCHG_EXPERIMENTS -DCHG_EXP_NO_OPERAND_READS, profiles/r05_experiments.md section 5): the weight operands
// come out of a register instead of LDS -- what the LDS operand supply of the contractions costs
#if defined(CHG_EXPERIMENTS) && defined(CHG_EXP_NO_OPERAND_READS)
#define CHG_OPERAND(expr, salt) fake_operand(salt)
__device__ __forceinline__ h16x8 fake_operand(int salt) {
  const _Float16 v = (_Float16)(0.001f * (float)(salt & 7));
  return h16x8{v, v, v, v, v, v, v, v};
}
#else
#define CHG_OPERAND(expr, salt) (expr)
#endif

// four output tiles fo0 .. fo0+3 from an already split row.  LO_SEPARATE: the cross products (low planes scaled by 2^11) and the
// main product go to separate accumulators, the former are scaled back -- a power of two, exact -- and added once; every operand is
// read from LDS once.  LEAN: the same two tiles at a time (half the accumulator / operand registers in flight): the forward kernels,
// which carry the next tile's gathered rows in registers and spill with the wide form (same-box A/B: atomconv_fwd +8 %, bondconv_fwd
// +11 % with it; the adjoint kernels -1 .. -5 %).
template <int MK, bool SCALED, bool LEAN = false>
__device__ __forceinline__ void gemm_split4(f32x4* acc, const h16x8* img, int F, const SplitRow<MK>& s, int fo0, int i, int g) {
  const int nchunks = MK * 4 * F;
  const h16x8* base0 = img + g * F + 16 * fo0 + i;
  // LO_SEPARATE: the cross products (hi x lo, lo x hi, carried at 2^11) and the main product in separate accumulators, so that
  // every operand is read from LDS ONCE (a second sweep over the high plane cost a third more LDS reads than the contraction needs;
  // the tile kernels' weight reads are a large share of their LDS traffic)
  // LEAN (the forward kernels, which sit at the register limit): TWO output tiles at a time with both accumulators (cross products at
  // 2^11 | main product) -- every operand read from LDS once, like the wide form below, at 16 accumulator + 16 operand registers.
  // (Rounds 3-4 used a two-sweep form here -- four tiles, one accumulator set, the high plane read twice: a third more LDS operand
  // reads; same-box A/B of round 5: atomconv_fwd -2 %, bondconv_fwd -2.5 %, profiles/r05_experiments.md section 6.)
  if constexpr (LEAN && LO_SEPARATE) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 t[2], u[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) { t[q] = zero4(); u[q] = SCALED ? zero4() : acc[2 * h + q]; }
#pragma unroll
      for (int mk = 0; mk < MK; ++mk) {
        const h16x8* base = base0 + mk * 4 * F + 32 * h;
        h16x8 wh[2], wl[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { wh[q] = CHG_OPERAND(base[16 * q], q + mk); wl[q] = CHG_OPERAND(base[nchunks + 16 * q], q + mk + 1); }
#pragma unroll
        for (int q = 0; q < 2; ++q) t[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q], s.lo[mk], t[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) u[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q], s.hi[mk], u[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) t[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[q], s.hi[mk], t[q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 r = u[q] + t[q] * LO_UNSCALE;
        if (SCALED) acc[2 * h + q] += r * s.up;
        else acc[2 * h + q] = r;
      }
    }
    return;
  }
  f32x4 t[4], u[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { t[q] = (SCALED || LO_SEPARATE) ? zero4() : acc[q]; u[q] = SCALED ? zero4() : acc[q]; }
#pragma unroll
  for (int mk = 0; mk < MK; ++mk) {
    const h16x8* base = base0 + mk * 4 * F;
    h16x8 wh[4], wl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { wh[q] = CHG_OPERAND(base[16 * q], q + mk); wl[q] = CHG_OPERAND(base[nchunks + 16 * q], q + mk + 1); }
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q], s.lo[mk], t[q], 0, 0, 0);
    if (LO_SEPARATE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) u[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q], s.hi[mk], u[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[q], s.hi[mk], t[q], 0, 0, 0);
    if (!LO_SEPARATE) {   // the matrix pipe keeps f16 subnormals (tools/split_lab.hip T2): one pass for all three products
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q], s.hi[mk], t[q], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 r = LO_SEPARATE ? u[q] + t[q] * LO_UNSCALE : t[q];
    if (SCALED) {
      acc[q] += r * s.up;
    } else {
      acc[q] = r;
    }
  }
}

// two output tiles (32 outputs): the adjoint of the 31 -> 64 embedding contractions (kernels_embed.h).  Same arithmetic as gemm_split4.
template <int MK, bool SCALED>
__device__ __forceinline__ void gemm_split2(f32x4* acc, const h16x8* img, int F, const SplitRow<MK>& s, int i, int g) {
  const int nchunks = MK * 4 * F;
  const h16x8* base0 = img + g * F + i;
  f32x4 t[2], u[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) { t[q] = zero4(); u[q] = SCALED ? zero4() : acc[q]; }
#pragma unroll
  for (int mk = 0; mk < MK; ++mk) {
    const h16x8* base = base0 + mk * 4 * F;
    h16x8 wh[2], wl[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { wh[q] = base[16 * q]; wl[q] = base[nchunks + 16 * q]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) t[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q], s.lo[mk], t[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 2; ++q) u[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q], s.hi[mk], u[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 2; ++q) t[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[q], s.hi[mk], t[q], 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const f32x4 r = u[q] + t[q] * LO_UNSCALE;
    if (SCALED) acc[q] += r * s.up;
    else acc[q] = r;
  }
}

template <int KT, int NFT, bool SCALED_, bool LEAN = false>
__device__ __forceinline__ void gemm_split(f32x4 (&acc)[NFT], const h16x8* img, int F, const f32x4 (&x)[KT], int i, int g) {
  static_assert(NFT % 4 == 0, "output width must be a multiple of 64");
  constexpr bool SCALED = SCALED_ || WIDE_RANGE;    // acc += W x either way: the scaled form adds the product, the plain one seeds with acc
  SplitRow<KT / 2> s;
  split_row<KT, SCALED>(s, x);
#pragma unroll
  for (int fo0 = 0; fo0 < NFT; fo0 += 4) gemm_split4<KT / 2, SCALED, LEAN>(acc + fo0, img, F, s, fo0, i, g);
}

// ---- row-major images: ONE LDS copy for both contraction directions -------------------------------------------------
// The images above need a second copy of W^T for the adjoint contractions (8 bytes per weight for both directions) -- too much
// for the BondConv adjoint (16,384 weights).  A plain row-major f16 matrix [F][K + 8] per plane serves both:
//   forward  (sum over k):  lane (i, g) reads W[16 fo + i][32 mk + 4 g ..+3] and [.. + 16 ..]       -- two ds_read_b64
//   adjoint  (sum over f):  ds_read_b64_tr_b16 -- in every 16-lane group lanes 4 r + q point at row f0 + r, columns 16 ko + 4 q ..+3
//                           and lane c RECEIVES W[f0 + 0..3][16 ko + c] (tools/split_lab.hip T5) -- two of them per operand
// 4 bytes per weight for both directions; the row stride K + 8 halves (= 4 banks mod 64 for K = 64) keeps both patterns
// conflict-free.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int rm_stride(int K) { return K + 8; }                                   // halves per row
constexpr size_t rm_image_bytes(int F, int K) { return (size_t)2 * F * rm_stride(K) * 2; }   // both planes

__device__ __forceinline__ void stage_rm(_Float16* img, const float* __restrict__ W, int F, int K, int tid, int nthreads) {
  const int S = rm_stride(K), k4 = K / 4;
  _Float16* lo_plane = img + F * S;
  for (int idx = tid; idx < F * k4; idx += nthreads) {
    const int f = idx / k4, c = 4 * (idx - f * k4);
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)f * K + c);
    h16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (_Float16)w[e];
      lo[e] = (_Float16)((w[e] - (float)hi[e]) * LO_SCALE);
    }
    *reinterpret_cast<h16x4*>(img + f * S + c) = hi;
    *reinterpret_cast<h16x4*>(lo_plane + f * S + c) = lo;
  }
}

__device__ __forceinline__ h16x8 join8(h16x4 a, h16x4 b) { return h16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
__device__ __forceinline__ h16x4 tr_read4(const _Float16* p) {
  const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p);
  return __builtin_bit_cast(h16x4, v);
}

// operand of output tile `o` and contraction step `m` for this lane; plane = 0 hi, 1 lo
template <bool ADJOINT>
__device__ __forceinline__ h16x8 rm_operand(const _Float16* img, int F, int K, int plane, int o, int m, int i, int g, int lane) {
  const int S = rm_stride(K);
  const _Float16* base = img + plane * F * S;
  if (!ADJOINT) {          // forward: output feature 16 o + i, k = 32 m + 4 g + (0..3 | 16..19)
    const _Float16* p = base + (16 * o + i) * S + 32 * m + 4 * g;
    return join8(*reinterpret_cast<const h16x4*>(p), *reinterpret_cast<const h16x4*>(p + 16));
  }
  // adjoint: output column 16 o + i, f = 32 m + 4 g + (0..3 | 16..19); this lane ADDRESSES row f0 + (q >> 2), columns 4 (q & 3)
  const int q = lane & 15;
  const _Float16* p = base + (32 * m + 4 * g + (q >> 2)) * S + 16 * o + 4 * (q & 3);
  return join8(tr_read4(p), tr_read4(p + 16 * S));
}

// four output tiles o0 .. o0 + 3 from an already split row; contraction over MS steps of 32
template <int MS, bool SCALED, bool ADJOINT>
__device__ __forceinline__ void gemm_rm4(f32x4* acc, const _Float16* img, int F, int K, const SplitRow<MS>& s, int o0, int i, int g, int lane) {
  // the cross products (hi x lo, lo x hi: carried at 2^11) and the main product in separate accumulators: every operand is read
  // from LDS once (re-reading the high plane for a second sweep cost a third more LDS reads than the contraction needs)
  f32x4 t[4], u[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { t[q] = zero4(); u[q] = SCALED ? zero4() : acc[q]; }
#pragma unroll
  for (int m = 0; m < MS; ++m) {
    h16x8 wh[4], wl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { wh[q] = CHG_OPERAND(rm_operand<ADJOINT>(img, F, K, 0, o0 + q, m, i, g, lane), q + m); wl[q] = CHG_OPERAND(rm_operand<ADJOINT>(img, F, K, 1, o0 + q, m, i, g, lane), q + m + 1); }
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q], s.lo[m], t[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q], s.hi[m], u[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[q], s.hi[m], t[q], 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 r = u[q] + t[q] * LO_UNSCALE;
    if (SCALED) {
      acc[q] += r * s.up;
    } else {
      acc[q] = r;
    }
  }
}

// acc[o] += W x (forward: W [F][K], x has K values, F / 16 output tiles) or W^T x (adjoint: x has F values, K / 16 output tiles)
template <int KT, int NOT, bool SCALED_, bool ADJOINT>
__device__ __forceinline__ void gemm_rm(f32x4 (&acc)[NOT], const _Float16* img, int F, int K, const f32x4 (&x)[KT], int i, int g, int lane) {
  static_assert(NOT % 4 == 0 && KT % 2 == 0, "widths are multiples of 64 / 32");
  static_assert(LO_SEPARATE, "row-major images carry the low plane scaled");
  constexpr bool SCALED = SCALED_ || WIDE_RANGE;
  SplitRow<KT / 2> s;
  split_row<KT, SCALED>(s, x);
#pragma unroll
  for (int o0 = 0; o0 < NOT; o0 += 4) gemm_rm4<KT / 2, SCALED, ADJOINT>(acc + o0, img, F, K, s, o0, i, g, lane);
}

}  // namespace chg
