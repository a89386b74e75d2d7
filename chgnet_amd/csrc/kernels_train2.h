// kernels_train2.h -- second-order part of the fine-tuning backward (stage B): the parameter
// gradient of a loss that depends on FORCES and STRESS (reference: loss.backward() through the
// create_graph=True force / stress of chgnet/model/model.py:517-535, trainer.py:399-411).
//
// Derivation (checked in float64 against torch double-backward by the tests' pipeline model):
//   dL/d theta = d/d theta [ sum_b ce_b E_b + D E ],   D E = d/d tau E(v + tau vdot),
//   vdot_e = ux[c_e] - ux[n_e] + v_e W_b,  ux = -dL/dF,  W_b = (kappa / V_b) dL/d sigma_b.
// One tangent (forward-mode) sweep gives the tangent of every activation; the reverse sweep then carries TWO
// adjoints per activation: G(y) = dE/dy (seed 1) and bar(y) = d Phi / dy (seed ce), with
//   bar(x) = J^T bar(y) + d/dx [G(y) . J(x) xdot],    bar(W) += bar(y) x^T + G(y) xdot^T.
//
// This first device version keeps the sweep UNFUSED: every Linear runs through the engine's row GEMMs
// (k_rows_gemm) on primal, tangent, bar and G rows alike, every weight gradient is a k_xty contraction, and the
// kernels here are the row-local nonlinear pieces in between -- one wave per row, lane f = feature f of the core
// branch and of the gate branch ([rows][128] arrays: columns 0..63 core, 64..127 gate).  It is HBM-bound by
// construction (a dozen [rows,128] arrays per layer); fusing it like the first-order kernels is later work.
#pragma once

#include "kernels_geom.h"
#include "mfma_tile.h"

namespace chg {

__device__ __forceinline__ float wmean64(float v) { return wave_sum(v) * (1.0f / 64.0f); }

// silu''(x) and sigmoid''(x)
__device__ __forceinline__ float ddsiluf_(float x) {
  const float s = sigmoidf_(x);
  return s * (1.0f - s) * (2.0f + x * (1.0f - 2.0f * s));
}

struct LnRow { float xh, rstd; };
__device__ __forceinline__ LnRow ln_row(float x) {   // LayerNorm statistics of a 64-vector held one element per lane
  const float mu = wmean64(x);
  const float xc = x - mu;
  const float rstd = __builtin_amdgcn_rsqf(wmean64(xc * xc) + LN_EPS);
  return LnRow{xc * rstd, rstd};
}
// P(a) = a - mean(a) - xhat mean(a xhat);  m_ax returns mean(a xhat)
__device__ __forceinline__ float ln_proj(float a, float xh, float& m_ax) {
  const float ma = wmean64(a);
  m_ax = wmean64(a * xh);
  return a - ma - xh * m_ax;
}

// ---------------------------------------------------------------------------------------------------------
// gather of the first-layer pre-activation z (and its tangent) from the per-layer tables
// ---------------------------------------------------------------------------------------------------------
struct GatherZArgs {
  int rows;
  // three table gathers:  z = T0[i0][off0 + .] + T1[i1][off1 + .] + T2[i2][off2 + .]  (+ add[row][.])
  const float *t0, *t1, *t2;     // primal tables
  const float *d0, *d1, *d2;     // tangent tables (same shapes)
  int ld0, ld1, ld2, off0, off1, off2;
  const int *i0, *i1, *i2;
  const float *add, *addd;       // optional [rows,128] addends (W_ang . angle features), primal and tangent
  int hidden;                    // 1: also write H = silu(z), Hd = silu'(z) zd
  float *Z, *Zd, *H, *Hd;        // [rows,128]
};

static __global__ __launch_bounds__(256) void k2_gather_z(GatherZArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < p.rows; r += nwaves) {
    const size_t a0 = (size_t)p.i0[r] * p.ld0 + p.off0, a1 = (size_t)p.i1[r] * p.ld1 + p.off1, a2 = (size_t)p.i2[r] * p.ld2 + p.off2;
    const size_t o = (size_t)r * 2 * D;
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // core, gate
      const int f = 64 * h + lane;
      float z = p.t0[a0 + f] + p.t1[a1 + f] + p.t2[a2 + f];
      float zd = p.d0[a0 + f] + p.d1[a1 + f] + p.d2[a2 + f];
      if (p.add) { z += p.add[o + f]; zd += p.addd[o + f]; }
      p.Z[o + f] = z;
      p.Zd[o + f] = zd;
      if (p.hidden) {
        p.H[o + f] = siluf_(z);
        p.Hd[o + f] = dsiluf_(z) * zd;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// gated MLP tail:  (c, g) -> y = silu(LN1 c) * sigmoid(LN2 g), forward with tangent and the two-adjoint backward
// ---------------------------------------------------------------------------------------------------------
struct GatedRow {       // everything the backward needs of one row, this lane's feature
  float xh1, r1, xh2, r2, xh1d, xh2d, n1d, n2d, a1, a2, a1d, a2d, da1, da2, s1, n1;
  float pt1, pt2, mt1, mt2;   // P(cd), P(gd), mean(cd xh1), mean(gd xh2)
  float y, yd;
};

__device__ __forceinline__ GatedRow gated_row_fwd(float c, float g, float cd, float gd, float g1, float b1, float g2, float b2) {
  GatedRow s;
  const LnRow l1 = ln_row(c), l2 = ln_row(g);
  s.xh1 = l1.xh; s.r1 = l1.rstd; s.xh2 = l2.xh; s.r2 = l2.rstd;
  s.pt1 = ln_proj(cd, s.xh1, s.mt1);
  s.pt2 = ln_proj(gd, s.xh2, s.mt2);
  s.xh1d = s.r1 * s.pt1;
  s.xh2d = s.r2 * s.pt2;
  s.n1 = g1 * s.xh1 + b1;
  const float n2 = g2 * s.xh2 + b2;
  s.n1d = g1 * s.xh1d;
  s.n2d = g2 * s.xh2d;
  s.s1 = sigmoidf_(s.n1);
  s.a1 = s.n1 * s.s1;
  s.da1 = s.s1 * (1.0f + s.n1 * (1.0f - s.s1));
  s.a2 = sigmoidf_(n2);
  s.da2 = s.a2 * (1.0f - s.a2);
  s.a1d = s.da1 * s.n1d;
  s.a2d = s.da2 * s.n2d;
  s.y = s.a1 * s.a2;
  s.yd = s.a1d * s.a2 + s.a1 * s.a2d;
  return s;
}

// bar(y), G(y) -> bar(c), bar(g), G(c), G(g); lnacc[8] += LayerNorm-affine gradients of this lane's feature
__device__ __forceinline__ void gated_row_bwd(const GatedRow& s, float bar_y, float g_y, float g1, float g2, float (&lnacc)[4],
                                              float& bar_c, float& bar_g, float& g_c, float& g_g) {
  const float bar_a1 = s.a2 * bar_y + s.a2d * g_y, bar_a2 = s.a1 * bar_y + s.a1d * g_y;
  const float g_a1 = s.a2 * g_y, g_a2 = s.a1 * g_y;
  const float dda1 = s.s1 * (1.0f - s.s1) * (2.0f + s.n1 * (1.0f - 2.0f * s.s1));
  const float dda2 = s.da2 * (1.0f - 2.0f * s.a2);
  const float bar_n1 = s.da1 * bar_a1 + dda1 * s.n1d * g_a1, g_n1 = s.da1 * g_a1;
  const float bar_n2 = s.da2 * bar_a2 + dda2 * s.n2d * g_a2, g_n2 = s.da2 * g_a2;
  lnacc[0] += bar_n1 * s.xh1 + g_n1 * s.xh1d;
  lnacc[1] += bar_n1;
  lnacc[2] += bar_n2 * s.xh2 + g_n2 * s.xh2d;
  lnacc[3] += bar_n2;
  {   // LayerNorm 1
    const float h = g1 * g_n1;
    float m_hx, m_bx;
    const float ph = ln_proj(h, s.xh1, m_hx);
    const float pb = ln_proj(g1 * bar_n1, s.xh1, m_bx);
    const float m_hpt = wmean64(h * s.pt1);
    bar_c = s.r1 * pb - s.r1 * s.r1 * (s.xh1 * m_hpt + ph * s.mt1 + s.pt1 * m_hx);
    g_c = s.r1 * ph;
  }
  {   // LayerNorm 2
    const float h = g2 * g_n2;
    float m_hx, m_bx;
    const float ph = ln_proj(h, s.xh2, m_hx);
    const float pb = ln_proj(g2 * bar_n2, s.xh2, m_bx);
    const float m_hpt = wmean64(h * s.pt2);
    bar_g = s.r2 * pb - s.r2 * s.r2 * (s.xh2 * m_hpt + ph * s.mt2 + s.pt2 * m_hx);
    g_g = s.r2 * ph;
  }
}

enum { T2_ATOM = 0, T2_BOND = 1, T2_ANGLE = 2 };

struct GatedTArgs {
  int rows, mode;
  const float *CG, *CGd;                 // [rows,128] second-layer pre-activations (c | g) and tangents
  const float *ln;                       // [4][64] ln1_g, ln1_b, ln2_g, ln2_b
  // ATOM: m = y * wag[k] scattered by centre;  BOND: u = y * wbg[b1] * wbg[b2] scattered by b1;  ANGLE: ang' = ang + y
  const int *i_dst, *i_w1, *i_w2;        // ATOM: centre, d2u, -;  BOND: b1c, b1c, b2c;  ANGLE: -, -, -
  const float *w, *wd;                   // ATOM: wag / wagd [Eu,64];  BOND: wbgc / wbgcd [Eb,64]
  float* aggd;                           // ATOM / BOND: tangent of the aggregate (zeroed), atomics
  const float* angd_in;                  // ANGLE: [rows,64]
  float* angd_out;
};

// Rows are centre-major (edges) / sorted by owning bond (angles): each wave takes a CONTIGUOUS block of rows, keeps what
// depends only on the run's key in registers (weights of the owning bond) and sends one atomic row per run instead of one
// per row.
static __global__ __launch_bounds__(256) void k2_gated_t(GatedTArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const float g1 = p.ln[lane], b1 = p.ln[D + lane], g2 = p.ln[2 * D + lane], b2 = p.ln[3 * D + lane];
  const int per = (p.rows + nwaves - 1) / nwaves;
  const int rb = min(p.rows, wave * per), re = min(p.rows, rb + per);
  int cur = -1;
  float acc = 0.f, w1 = 0.f, w1d = 0.f;
  float in[4] = {0.f, 0.f, 0.f, 0.f};   // the next row's inputs are requested before this row's results are stored
  if (rb < re) { const size_t o = (size_t)rb * 2 * D; in[0] = p.CG[o + lane]; in[1] = p.CG[o + D + lane]; in[2] = p.CGd[o + lane]; in[3] = p.CGd[o + D + lane]; }
  for (int r = rb; r < re; ++r) {
    const float c0 = in[0], c1 = in[1], c2 = in[2], c3 = in[3];
    if (r + 1 < re) { const size_t o = (size_t)(r + 1) * 2 * D; in[0] = p.CG[o + lane]; in[1] = p.CG[o + D + lane]; in[2] = p.CGd[o + lane]; in[3] = p.CGd[o + D + lane]; }
    const GatedRow s = gated_row_fwd(c0, c1, c2, c3, g1, b1, g2, b2);
    if (p.mode == T2_ANGLE) {
      p.angd_out[(size_t)r * D + lane] = p.angd_in[(size_t)r * D + lane] + s.yd;
      continue;
    }
    const int dst = p.i_dst[r];
    if (dst != cur) {
      if (cur >= 0) atomicAdd(p.aggd + (size_t)cur * D + lane, acc);
      cur = dst;
      acc = 0.f;
      if (p.mode == T2_BOND) {          // i_w1 == i_dst: the owning bond's weight row
        w1 = p.w[(size_t)dst * D + lane];
        w1d = p.wd[(size_t)dst * D + lane];
      }
    }
    if (p.mode == T2_ATOM) {
      const size_t k = (size_t)p.i_w1[r] * D + lane;
      acc += s.yd * p.w[k] + s.y * p.wd[k];
    } else {
      const size_t k2 = (size_t)p.i_w2[r] * D + lane;
      const float w2 = p.w[k2];
      acc += s.yd * w1 * w2 + s.y * (w1d * w2 + w1 * p.wd[k2]);
    }
  }
  if (cur >= 0) atomicAdd(p.aggd + (size_t)cur * D + lane, acc);
}

struct GatedBArgs {
  int rows, mode;
  const float *CG, *CGd, *ln;
  const int *i_dst, *i_w1, *i_w2;
  const float *w, *wd;
  const float *bar_agg, *g_agg;          // ATOM: [N,64] adjoints of the aggregate (gathered by centre); BOND: [Eb,64] by b1; ANGLE: bar_ang / g_ang [rows,64]
  float *bar_w, *g_w;                    // ATOM: bar / G of wag [Eu,64]; BOND: of wbgc [Eb,64]  (atomics)
  float *BCG, *GCG;                      // out [rows,128]: bar(c|g), G(c|g)
  float* g_ln;                           // [4][64] LayerNorm-affine gradients (atomics)
};

static __global__ __launch_bounds__(256) void k2_gated_b(GatedBArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const float g1 = p.ln[lane], b1 = p.ln[D + lane], g2 = p.ln[2 * D + lane], b2 = p.ln[3 * D + lane];
  float lnacc[4] = {0.f, 0.f, 0.f, 0.f};
  const int per = (p.rows + nwaves - 1) / nwaves;          // contiguous rows per wave: see k2_gated_t
  const int rb = min(p.rows, wave * per), re = min(p.rows, rb + per);
  int cur = -1;
  float bar_a = 0.f, g_a = 0.f;                            // adjoints of the run's aggregate row
  float w1 = 0.f, w1d = 0.f, acc_bw = 0.f, acc_gw = 0.f;   // BOND: the owning bond's weight row and its gradient sums
  float in[4] = {0.f, 0.f, 0.f, 0.f};                      // next row's inputs in flight over this row's work (see k2_gated_t)
  if (rb < re) { const size_t o = (size_t)rb * 2 * D; in[0] = p.CG[o + lane]; in[1] = p.CG[o + D + lane]; in[2] = p.CGd[o + lane]; in[3] = p.CGd[o + D + lane]; }
  for (int r = rb; r < re; ++r) {
    const size_t o = (size_t)r * 2 * D;
    const float c0 = in[0], c1 = in[1], c2 = in[2], c3 = in[3];
    if (r + 1 < re) { const size_t o1 = o + 2 * D; in[0] = p.CG[o1 + lane]; in[1] = p.CG[o1 + D + lane]; in[2] = p.CGd[o1 + lane]; in[3] = p.CGd[o1 + D + lane]; }
    const GatedRow s = gated_row_fwd(c0, c1, c2, c3, g1, b1, g2, b2);
    float bar_y, g_y;
    if (p.mode == T2_ANGLE) {
      bar_y = p.bar_agg[(size_t)r * D + lane];
      g_y = p.g_agg[(size_t)r * D + lane];
    } else {
      const int dst = p.i_dst[r];
      if (dst != cur) {
        if (p.mode == T2_BOND && cur >= 0) {
          atomicAdd(p.bar_w + (size_t)cur * D + lane, acc_bw);
          atomicAdd(p.g_w + (size_t)cur * D + lane, acc_gw);
        }
        cur = dst;
        bar_a = p.bar_agg[(size_t)dst * D + lane];
        g_a = p.g_agg[(size_t)dst * D + lane];
        if (p.mode == T2_BOND) {        // i_w1 == i_dst
          w1 = p.w[(size_t)dst * D + lane];
          w1d = p.wd[(size_t)dst * D + lane];
          acc_bw = acc_gw = 0.f;
        }
      }
      if (p.mode == T2_ATOM) {
        const size_t k = (size_t)p.i_w1[r] * D + lane;
        const float w = p.w[k], wd = p.wd[k];
        atomicAdd(p.bar_w + k, s.y * bar_a + s.yd * g_a);
        atomicAdd(p.g_w + k, s.y * g_a);
        bar_y = w * bar_a + wd * g_a;
        g_y = w * g_a;
      } else {
        const size_t k2 = (size_t)p.i_w2[r] * D + lane;
        const float w2 = p.w[k2], w2d = p.wd[k2];
        acc_bw += s.y * w2 * bar_a + (s.yd * w2 + s.y * w2d) * g_a;
        acc_gw += s.y * w2 * g_a;
        atomicAdd(p.bar_w + k2, s.y * w1 * bar_a + (s.yd * w1 + s.y * w1d) * g_a);
        atomicAdd(p.g_w + k2, s.y * w1 * g_a);
        bar_y = w1 * w2 * bar_a + (w1d * w2 + w1 * w2d) * g_a;
        g_y = w1 * w2 * g_a;
      }
    }
    float bar_c, bar_g, g_c, g_g;
    gated_row_bwd(s, bar_y, g_y, g1, g2, lnacc, bar_c, bar_g, g_c, g_g);
    p.BCG[o + lane] = bar_c;
    p.BCG[o + D + lane] = bar_g;
    p.GCG[o + lane] = g_c;
    p.GCG[o + D + lane] = g_g;
  }
  if (p.mode == T2_BOND && cur >= 0) {
    atomicAdd(p.bar_w + (size_t)cur * D + lane, acc_bw);
    atomicAdd(p.g_w + (size_t)cur * D + lane, acc_gw);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) atomicAdd(p.g_ln + q * D + lane, lnacc[q]);
}

// hidden layer:  bar(z) = silu'(z) bar(H) + silu''(z) zd G(H),   G(z) = silu'(z) G(H)      (elementwise over [rows,128])
static __global__ void k2_hidden_b(const float* __restrict__ Z, const float* __restrict__ Zd, const float* __restrict__ BH,
                            const float* __restrict__ GH, float* __restrict__ BZ, float* __restrict__ GZ, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float z = Z[i], d1 = dsiluf_(z), gh = GH[i];
  BZ[i] = d1 * BH[i] + ddsiluf_(z) * Zd[i] * gh;
  GZ[i] = d1 * gh;
}

// scatter of the first-layer adjoints back to the tables: three destinations per row, bar and G together
struct ScatterZArgs {
  int rows;
  const float *BZ, *GZ;           // [rows,128]
  float *b0, *b1, *b2;            // bar table gradients (zeroed)
  float *g0, *g1, *g2;            // G table gradients
  int ld0, ld1, ld2, off0, off1, off2;
  const int *i0, *i1, *i2;
};

// One wave = 16 consecutive rows staged in LDS; each destination leaves the tile as run sums over equal adjacent keys
// (mfma_tile.h:seg_colsum_atomic): the rows are centre-major (edges) / sorted by owning bond (angles), so the first key is one
// long run, and keys that are not sorted simply make 16 runs of one row -- one 256-B atomic row per run and 64 columns.
constexpr int SZ_TS = 2 * D + PAD;
constexpr size_t scatter_z_lds() { return sizeof(float) * 4 * 2 * TILE_ROWS * SZ_TS; }

static __global__ __launch_bounds__(256) void k2_scatter_z(ScatterZArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* TB = smem + wv * 2 * TILE_ROWS * SZ_TS;
  float* TG = TB + TILE_ROWS * SZ_TS;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int ntiles = (p.rows + TILE_ROWS - 1) / TILE_ROWS;
  const int hw = lane >> 5, t4 = lane & 31;     // half-wave per row, 32 lanes x float4 = one 512-B row
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int row0 = tile * TILE_ROWS, nvalid = min(TILE_ROWS, p.rows - row0);
    const int j = lane & 15;
    const int r = row0 + (j < nvalid ? j : 0);
    const int k0 = j < nvalid ? p.i0[r] : -1, k1 = j < nvalid ? p.i1[r] : -1, k2 = j < nvalid ? p.i2[r] : -1;
#pragma unroll
    for (int it = 0; it < TILE_ROWS / 2; ++it) {
      const int rr = 2 * it + hw;
      if (rr < nvalid) {
        *reinterpret_cast<f32x4*>(TB + rr * SZ_TS + 4 * t4) = *reinterpret_cast<const f32x4*>(p.BZ + (size_t)(row0 + rr) * 2 * D + 4 * t4);
        *reinterpret_cast<f32x4*>(TG + rr * SZ_TS + 4 * t4) = *reinterpret_cast<const f32x4*>(p.GZ + (size_t)(row0 + rr) * 2 * D + 4 * t4);
      }
    }
    __builtin_amdgcn_wave_barrier();
    seg_colsum_atomic<2 * D>(TB, SZ_TS, k0, nvalid, p.b0 + p.off0, p.ld0, lane);
    seg_colsum_atomic<2 * D>(TB, SZ_TS, k1, nvalid, p.b1 + p.off1, p.ld1, lane);
    seg_colsum_atomic<2 * D>(TB, SZ_TS, k2, nvalid, p.b2 + p.off2, p.ld2, lane);
    seg_colsum_atomic<2 * D>(TG, SZ_TS, k0, nvalid, p.g0 + p.off0, p.ld0, lane);
    seg_colsum_atomic<2 * D>(TG, SZ_TS, k1, nvalid, p.g1 + p.off1, p.ld1, lane);
    seg_colsum_atomic<2 * D>(TG, SZ_TS, k2, nvalid, p.g2 + p.off2, p.ld2, lane);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------
// geometry: tangent of the bond vectors
// ---------------------------------------------------------------------------------------------------------
// vd_e = ux[c] - ux[n] + v_e W_b;  rd = u . vd;  ud = (vd - u rd) / r          out: vd4 = (vd, rd), ud4 = (ud, 0)
static __global__ void k2_geom_t(const f32x4* __restrict__ ev, const f32x4* __restrict__ eu, const int* __restrict__ e_center,
                          const int* __restrict__ e_nbr, const int* __restrict__ e_owner, const float* __restrict__ ux,
                          const float* __restrict__ Wst, f32x4* __restrict__ vd4, f32x4* __restrict__ ud4, int n_edges) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const f32x4 v = ev[e], u = eu[e];
  const float* W = Wst + 9 * (size_t)e_owner[e];
  const int c = e_center[e], n = e_nbr[e];
  float vd[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) vd[k] = ux[3 * c + k] - ux[3 * n + k] + v[0] * W[k] + v[1] * W[3 + k] + v[2] * W[6 + k];
  const float rd = u[0] * vd[0] + u[1] * vd[1] + u[2] * vd[2];
  const float inv_r = 1.0f / v[3];
  vd4[e] = f32x4{vd[0], vd[1], vd[2], rd};
  ud4[e] = f32x4{(vd[0] - u[0] * rd) * inv_r, (vd[1] - u[1] * rd) * inv_r, (vd[2] - u[2] * rd) * inv_r, 0.f};
}

// ---------------------------------------------------------------------------------------------------------
// radial / Fourier bases with tangents, the 31 -> 64 embedding linears, and the frequency gradients
// ---------------------------------------------------------------------------------------------------------
// rbf value and its r-, f- and mixed derivatives (basis.py:108-116, 197-206)
__device__ __forceinline__ void rbf_all(float r, float rc, float freq, Envelope env, float& val, float& dr, float& df, float& drdf) {
  const float inv_rc = 1.0f / rc, w = freq * inv_rc, cn = sqrtf(2.0f * inv_rc);
  float sn, cs;
  sincos_cw(w * r, sn, cs);
  const float s = r * inv_rc;
  float e = 0.f, de = 0.f;
  if (s < 1.0f) {
    const float sp1 = ipow(s, env.p - 1), sp = sp1 * s;
    e = 1.0f + env.a * sp + env.b * sp * s + env.c * sp * s * s;
    de = (env.a * env.p * sp1 + env.b * (env.p + 1) * sp + env.c * (env.p + 2) * sp * s) * inv_rc;
  }
  val = e * cn * sn / r;
  dr = de * cn * sn / r + e * cn * (w * cs / r - sn / (r * r));
  df = e * cn * cs * inv_rc;
  drdf = cn * inv_rc * (de * cs - e * w * sn);
}

constexpr int KB2 = 32;   // basis count padded

struct BondBasisArgs {
  int n_und;
  const f32x4 *ev, *vd4;
  const int* u_u2d;
  const float *freq_ag, *freq_bg;
  float rc_ag, rc_bg;
  Envelope env;
  float *X6, *X6d, *X3, *X3d;     // [Eu,32]: basis and tangent (column 31 = 0)
};

static __global__ void k2_bond_basis(BondBasisArgs p) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = t / KB2, j = t % KB2;
  if (k >= p.n_und) return;
  const int e = p.u_u2d[k];
  const float r = p.ev[e][3], rd = p.vd4[e][3];
  float v6 = 0.f, d6 = 0.f, v3 = 0.f, d3 = 0.f, df, drdf;
  if (j < NRAD) {
    rbf_all(r, p.rc_ag, p.freq_ag[j], p.env, v6, d6, df, drdf);
    rbf_all(r, p.rc_bg, p.freq_bg[j], p.env, v3, d3, df, drdf);
  }
  p.X6[t] = v6; p.X6d[t] = d6 * rd; p.X3[t] = v3; p.X3d[t] = d3 * rd;
}

// out[row][f] = sum_j W[f][j] X[row][j]  (W [64][31] row-major, X [rows][32]); optional output row map
static __global__ __launch_bounds__(256) void k2_embed_lin(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ out,
                                                    const int* __restrict__ in_rows, int rows) {
  __shared__ float Ws[D * NRAD];
  for (int i = threadIdx.x; i < D * NRAD; i += blockDim.x) Ws[i] = W[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < rows; r += nwaves) {
    const size_t src = (size_t)(in_rows ? in_rows[r] : r) * KB2;
    const float x = lane < KB2 ? X[src + lane] : 0.f;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NRAD; ++j) acc += Ws[lane * NRAD + j] * bcast(x, j);
    out[(size_t)r * D + lane] = acc;
  }
}

// frequency gradients of the radial bases:  d f_j += sum_k [ bar(rbf)_kj d rbf/df + G(rbf)_kj d2 rbf/(dr df) rdot_k ]
// with bar(rbf) = bar(hb0) Wbe + bar(wag) Wag (cutoff r_atom) or bar(wbg) Wbg (cutoff r_bond): lane j = basis index
struct FreqGradArgs {
  int rows;                       // bonds (atom-graph cutoff: all Eu; bond-graph cutoff: the Eb node bonds)
  const int* row_und;             // null: row k is undirected bond k; else undirected index of row
  const f32x4 *ev, *vd4;
  const int* u_u2d;
  const float* freq;
  float rc;
  Envelope env;
  const float *barA, *gA, *WA;    // adjoint rows [rows,64] and their [64][31] weight
  const float *barB, *gB, *WB;    // optional second pair (null)
  float* g_freq;                  // [31]
};

static __global__ __launch_bounds__(256) void k2_freq_grad(FreqGradArgs p) {
  __shared__ float WAs[D * NRAD], WBs[D * NRAD];
  for (int i = threadIdx.x; i < D * NRAD; i += blockDim.x) {
    WAs[i] = p.WA[i];
    WBs[i] = p.WB ? p.WB[i] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const float fj = lane < NRAD ? p.freq[lane] : 0.f;
  float acc = 0.f;
  for (int r = wave; r < p.rows; r += nwaves) {
    const int k = p.row_und ? p.row_und[r] : r;
    const int e = p.u_u2d[k];
    const float rr = p.ev[e][3], rd = p.vd4[e][3];
    const size_t o = (size_t)r * D + lane;
    const float ba = p.barA[o], ga = p.gA[o], bb = p.barB ? p.barB[o] : 0.f, gb = p.gB ? p.gB[o] : 0.f;
    float bar_x = 0.f, g_x = 0.f;      // this lane's basis index j = lane: sum over the 64 features (broadcast from every lane)
#pragma unroll 8
    for (int f = 0; f < D; ++f) {
      const float wa = lane < NRAD ? WAs[f * NRAD + lane] : 0.f, wb = lane < NRAD ? WBs[f * NRAD + lane] : 0.f;
      bar_x += bcast(ba, f) * wa + bcast(bb, f) * wb;
      g_x += bcast(ga, f) * wa + bcast(gb, f) * wb;
    }
    if (lane < NRAD) {
      float v, dr, df, drdf;
      rbf_all(rr, p.rc, fj, p.env, v, dr, df, drdf);
      acc += bar_x * df + g_x * drdf * rd;
    }
  }
  if (lane < NRAD) atomicAdd(p.g_freq + lane, acc);
}

// Fourier basis of every angle with tangent:  X [A,32], Xd [A,32];  also theta and thetadot (for the frequency gradient)
static __global__ void k2_angle_basis(const f32x4* __restrict__ eu, const f32x4* __restrict__ ud4, const int* __restrict__ a_d1,
                               const int* __restrict__ a_d2, const float* __restrict__ freq, float* __restrict__ X, float* __restrict__ Xd,
                               float* __restrict__ th2, int n_angles) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int a = t / KB2, j = t % KB2;
  if (a >= n_angles) return;
  const f32x4 u1 = eu[a_d1[a]], u2 = eu[a_d2[a]], v1 = ud4[a_d1[a]], v2 = ud4[a_d2[a]];
  const float cosv = (u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2]) * KAPPA;
  const float cosd = (v1[0] * u2[0] + v1[1] * u2[1] + v1[2] * u2[2] + u1[0] * v2[0] + u1[1] * v2[1] + u1[2] * v2[2]) * KAPPA;
  const float theta = acosf(cosv), thd = -cosd / sqrtf(1.0f - cosv * cosv);
  float x = 0.f, dx = 0.f;
  if (j == 0) {
    x = INV_SQRT_2 * INV_SQRT_PI;
  } else if (j <= NFREQ) {
    float sn, cs;
    sincos_cw(freq[j - 1] * theta, sn, cs);
    x = sn * INV_SQRT_PI; dx = freq[j - 1] * cs * INV_SQRT_PI;
  } else if (j < NANG) {
    float sn, cs;
    sincos_cw(freq[j - 1 - NFREQ] * theta, sn, cs);
    x = cs * INV_SQRT_PI; dx = -freq[j - 1 - NFREQ] * sn * INV_SQRT_PI;
  }
  X[t] = x;
  Xd[t] = dx * thd;
  if (j == 0) { th2[2 * a] = theta; th2[2 * a + 1] = thd; }
}

// d g_q += sum_a [ bar(four) d four/dg + G(four) d2 four/(d theta dg) thetadot ],  bar(four) = bar(ang0) Wae
static __global__ __launch_bounds__(256) void k2_angle_freq_grad(const float* __restrict__ bar_ang, const float* __restrict__ g_ang,
                                                          const float* __restrict__ Wae, const float* __restrict__ th2,
                                                          const float* __restrict__ freq, float* __restrict__ g_freq, int n_angles) {
  __shared__ float Ws[D * NANG];
  for (int i = threadIdx.x; i < D * NANG; i += blockDim.x) Ws[i] = Wae[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  // lane j in 1..15: sin column of frequency j-1; lane j in 16..30: cos column of frequency j-16
  const bool is_sin = lane >= 1 && lane <= NFREQ, is_cos = lane > NFREQ && lane < NANG;
  const int q = is_sin ? lane - 1 : (is_cos ? lane - 1 - NFREQ : 0);
  const float gq = freq[q];
  float acc = 0.f;
  for (int a = wave; a < n_angles; a += nwaves) {
    const float ba = bar_ang[(size_t)a * D + lane], ga = g_ang[(size_t)a * D + lane];
    float bar_x = 0.f, g_x = 0.f;
#pragma unroll 8
    for (int f = 0; f < D; ++f) {
      const float w = lane < NANG ? Ws[f * NANG + lane] : 0.f;
      bar_x += bcast(ba, f) * w;
      g_x += bcast(ga, f) * w;
    }
    const float theta = th2[2 * a], thd = th2[2 * a + 1];
    float sn, cs;
    sincos_cw(gq * theta, sn, cs);
    if (is_sin) acc += (bar_x * theta * cs + g_x * (cs - gq * theta * sn) * thd) * INV_SQRT_PI;
    if (is_cos) acc += (-bar_x * theta * sn + g_x * (-sn - gq * theta * cs) * thd) * INV_SQRT_PI;
  }
  if (is_sin || is_cos) atomicAdd(g_freq + q, acc);
}

// ---------------------------------------------------------------------------------------------------------
// readout: LayerNorm and the three silu layers, tangent forward and two-adjoint backward (rows = atoms, width 64)
// ---------------------------------------------------------------------------------------------------------
// LayerNorm forward with tangent:  y = gamma xhat + beta,  yd = gamma xhatd;  keeps xhat, xhatd, rstd (per row in R[3*row..])
static __global__ __launch_bounds__(256) void k2_ln_t(const float* __restrict__ x, const float* __restrict__ xd, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ yd,
                                               float* __restrict__ xh, float* __restrict__ xhd, int rows) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const float g = gamma[lane], b = beta[lane];
  for (int r = wave; r < rows; r += nwaves) {
    const size_t o = (size_t)r * D + lane;
    const LnRow l = ln_row(x[o]);
    float m;
    const float hd = l.rstd * ln_proj(xd[o], l.xh, m);
    y[o] = g * l.xh + b; yd[o] = g * hd; xh[o] = l.xh; xhd[o] = hd;
  }
}

// bar(y), G(y) -> bar(x), G(x) through the LayerNorm; dgam / dbet rows are written for a later column sum
static __global__ __launch_bounds__(256) void k2_ln_b(const float* __restrict__ x, const float* __restrict__ xd, const float* __restrict__ gamma,
                                               const float* __restrict__ bar_y, const float* __restrict__ g_y, float* __restrict__ bar_x,
                                               float* __restrict__ g_x, float* __restrict__ dgam, float* __restrict__ dbet, int rows) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const float g = gamma[lane];
  for (int r = wave; r < rows; r += nwaves) {
    const size_t o = (size_t)r * D + lane;
    const LnRow l = ln_row(x[o]);
    float mt, m_hx, m_bx;
    const float pt = ln_proj(xd[o], l.xh, mt);
    const float by = bar_y[o], gy = g_y[o];
    const float h = g * gy;
    const float ph = ln_proj(h, l.xh, m_hx);
    const float pb = ln_proj(g * by, l.xh, m_bx);
    const float m_hpt = wmean64(h * pt);
    bar_x[o] = l.rstd * pb - l.rstd * l.rstd * (l.xh * m_hpt + ph * mt + pt * m_hx);
    g_x[o] = l.rstd * ph;
    dgam[o] = by * l.xh + gy * l.rstd * pt;
    dbet[o] = by;
  }
}

// s = silu(l), sd = silu'(l) ld
static __global__ void k2_silu_t(const float* __restrict__ l, const float* __restrict__ ld, float* __restrict__ s, float* __restrict__ sd, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s[i] = siluf_(l[i]);
  sd[i] = dsiluf_(l[i]) * ld[i];
}

// seeds of the reverse sweep at the site energies:  bar(s3) = cot[owner] w3,  G(s3) = w3;  also d w3 rows = cot s3 + s3d
static __global__ void k2_readout_seed(const float* __restrict__ w3, const float* __restrict__ cot, const int* __restrict__ owner,
                                const float* __restrict__ s3, const float* __restrict__ s3d, float* __restrict__ bar_s,
                                float* __restrict__ g_s, float* __restrict__ dw3_rows, int n_atoms) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_atoms * D) return;
  const int i = t / D, f = t % D;
  const float c = cot[owner[i]];
  bar_s[t] = c * w3[f];
  g_s[t] = w3[f];
  dw3_rows[t] = c * s3[t] + s3d[t];
}

// out[i] = a[i] + b[i]   /   magmom head handled by k_magmom_bwd (kernels_train.h)
static __global__ void k2_add(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

}  // namespace chg
