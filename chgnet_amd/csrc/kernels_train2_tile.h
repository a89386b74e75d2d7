// kernels_train2_tile.h -- the second-order fine-tuning sweep (kernels_train2.h: derivation, reference lines) FUSED per layer:
// one tile kernel per layer kind and direction instead of the chain  gather_z -> 4 row GEMMs -> gated_t | gated_b -> 4 row GEMMs
// -> hidden_b -> scatter_z  over [rows,128] arrays in HBM (a dozen array passes per layer visit; the sweep was HBM-bound at
// ~3.5 TB/s for 215 ms per 1024-structure step).
//
// One wave = 16 rows, as in the first-order kernels (mfma_tile.h).  Linear pieces run on the matrix pipe in the accumulator
// layout (split-precision contractions against row-major weight images in LDS, mfma_split.h: one image serves W and W^T); the
// row-local nonlinear piece -- LayerNorm, activations, their tangents and the two-adjoint backward -- reuses gated_row_fwd /
// gated_row_bwd of the unfused sweep verbatim in their layout (one row at a time, lane = feature): the tile goes through the
// wave's LDS tile eight rows at a time.  Only what the weight-gradient contractions (k_xty) need leaves the kernel as rows:
// H, Hd, bar(c|g), G(c|g).  Nothing of the tangent forward is kept: the reverse kernel recomputes it from the tables.
#pragma once

#include <type_traits>

#include "kernels_conv.h"
#include "kernels_train2.h"

#ifndef CHG_T2_SB
#define CHG_T2_SB 0
#endif

namespace chg {

constexpr int T2_AS = 4 * D + 4;     // row stride of the 8-row staging area inside the wave's tile: c | g | cd | gd
static_assert(8 * T2_AS <= TILE_FLOATS, "the staging area reuses the wave's gather tile");

constexpr size_t t2_atom_lds() { return 2 * rm_image_bytes(D, D) + sizeof(float) * (VEC_SLOTS * D + WAVES * TILE_FLOATS); }

// silu, silu' and silu'' zd of a pre-activation vector with tangent: h, hd = silu'(z) zd, d1 = silu'(z), e = silu''(z) zd
__device__ __forceinline__ void hidden_t(const V64& z, const V64& zd, V64& h, V64& hd, V64& d1, V64& e) {
  CHG_EW(ft, r) {
    const float x = z.t[ft][r], s = sigmoidf_(x);
    h.t[ft][r] = x * s;
    const float d = s * (1.0f + x * (1.0f - s));
    d1.t[ft][r] = d;
    hd.t[ft][r] = d * zd.t[ft][r];
    e.t[ft][r] = s * (1.0f - s) * (2.0f + x * (1.0f - 2.0f * s)) * zd.t[ft][r];
  }
}

// ---- the gated-MLP tail with tangent and two adjoints IN THE ACCUMULATOR LAYOUT (lane (j, g) = row j, 16 of its 64 features) ----
// Same formulas as gated_row_fwd / gated_row_bwd (kernels_train2.h: derivation), sixteen rows at once: a row reduction is 12 vector
// adds and one quad_sum for ALL rows of the tile instead of a 64-lane reduction per row -- the one-row-at-a-time form spent 5,500 of
// the reverse kernels' 8,000 vector instructions per tile there (profiles/r05_experiments.md section 13).
__device__ __forceinline__ float row_mean64(const f32x4& s) { return quad_sum(hsum4(s)) * (1.0f / 64.0f); }
__device__ __forceinline__ f32x4 vec4(const float* vec, int ft, int g) { return *reinterpret_cast<const f32x4*>(vec + 16 * ft + 4 * g); }   // LDS parameter vector
__device__ __forceinline__ f32x4 rd4(const float* base, int row, int ft, int g, int ld = D) { return *grow<f32x4>(base, (unsigned)row, ld, 16 * ft + 4 * g); }

// c <- xhat, cd <- P(cd) = cd - mean(cd) - xhat mean(cd xhat);  rstd and mt = mean(cd xhat) of this lane's row
__device__ __forceinline__ void ln2_forward(V64& c, V64& cd, float& rstd, float& mt) {
  rstd = ln_normalize(c);
  f32x4 s1 = zero4(), s2 = zero4();
  CHG_EV(ft) { s1 += cd.t[ft]; s2 += cd.t[ft] * c.t[ft]; }
  const float m = row_mean64(s1);
  mt = row_mean64(s2);
  CHG_EV(ft) cd.t[ft] = cd.t[ft] - m - c.t[ft] * mt;
}

// The gate arithmetic ELEMENT BY ELEMENT (two at a time, a scheduling barrier in between): written on f32x4 slices the compiler kept
// all four elements' temporaries in flight -- ~130 registers for a slice -- and spilled 350-440 registers in the reverse kernels; a
// dependent chain of plain vector instructions costs nothing on this pipe, and the other wave of the SIMD covers the transcendentals.
struct Gate1 { float n1, n1d, n2d, s1, a1, da1, a1d, a2, da2, a2d, y, yd; };
__device__ __forceinline__ Gate1 gate1(float xh1, float pt1, float xh2, float pt2, float r1, float r2, float ga1, float be1, float ga2, float be2) {
  Gate1 e;
  e.n1 = ga1 * xh1 + be1;
  e.n1d = ga1 * (pt1 * r1);
  const float n2 = ga2 * xh2 + be2;
  e.n2d = ga2 * (pt2 * r2);
  e.s1 = sigmoidf_(e.n1);
  e.a1 = e.n1 * e.s1;
  e.da1 = e.s1 * (1.0f + e.n1 * (1.0f - e.s1));
  e.a2 = sigmoidf_(n2);
  e.da2 = e.a2 * (1.0f - e.a2);
  e.a1d = e.da1 * e.n1d;
  e.a2d = e.da2 * e.n2d;
  e.y = e.a1 * e.a2;
  e.yd = e.a1d * e.a2 + e.a1 * e.a2d;
  return e;
}
// bar(y), G(y) -> adjoints of the two LayerNorm outputs
__device__ __forceinline__ void gate1_bwd(const Gate1& e, float bar_y, float g_y, float& bn1, float& gn1, float& bn2, float& gn2) {
  const float bar_a1 = e.a2 * bar_y + e.a2d * g_y, bar_a2 = e.a1 * bar_y + e.a1d * g_y;
  const float g_a1 = e.a2 * g_y, g_a2 = e.a1 * g_y;
  const float dda1 = e.s1 * (1.0f - e.s1) * (2.0f + e.n1 * (1.0f - 2.0f * e.s1));
  const float dda2 = e.da2 * (1.0f - 2.0f * e.a2);
  bn1 = e.da1 * bar_a1 + dda1 * e.n1d * g_a1;
  gn1 = e.da1 * g_a1;
  bn2 = e.da2 * bar_a2 + dda2 * e.n2d * g_a2;
  gn2 = e.da2 * g_a2;
}
// LayerNorm-affine gradients of one branch: column sums over the tile's valid rows of  bn xhat + gn xhat_d  and of  bn  (column = lane)
__device__ __forceinline__ void ln2_affine_sums(float* T, int j, int g, int lane, int nvalid, const V64& bn, const V64& gn, const V64& xh, const V64& pt,
                                                float r, float& d_gamma, float& d_beta) {
  float* Trow = T + j * TS;
  __builtin_amdgcn_wave_barrier();
  CHG_EV(ft) {
    *reinterpret_cast<f32x4*>(Trow + 16 * ft + 4 * g) = bn.t[ft] * xh.t[ft] + gn.t[ft] * (pt.t[ft] * r);
    *reinterpret_cast<f32x4*>(Trow + D + 16 * ft + 4 * g) = bn.t[ft];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int rr = 0; rr < TILE_ROWS; ++rr)
    if (rr < nvalid) {
      d_gamma += T[rr * TS + lane];
      d_beta += T[rr * TS + D + lane];
    }
  __builtin_amdgcn_wave_barrier();
}
// bn, gn (adjoints of the LayerNorm output: bar, G) -> bar(c), G(c) in place;  xh, pt, r, mt as ln2_forward left them
__device__ __forceinline__ void ln2_backward(V64& bn, V64& gn, const float* gamma, int g, const V64& xh, const V64& pt, float r, float mt) {
  f32x4 s_h = zero4(), s_hx = zero4(), s_hpt = zero4(), s_b = zero4(), s_bx = zero4();
  CHG_EV(ft) {
    const f32x4 ga = vec4(gamma, ft, g);      // (LDS)
    gn.t[ft] = gn.t[ft] * ga;
    bn.t[ft] = bn.t[ft] * ga;
    s_h += gn.t[ft];
    s_hx += gn.t[ft] * xh.t[ft];
    s_hpt += gn.t[ft] * pt.t[ft];
    s_b += bn.t[ft];
    s_bx += bn.t[ft] * xh.t[ft];
  }
  const float m_h = row_mean64(s_h), m_hx = row_mean64(s_hx), m_hpt = row_mean64(s_hpt), m_b = row_mean64(s_b), m_bx = row_mean64(s_bx);
  const float rr2 = r * r;
  CHG_EV(ft) {
    const f32x4 ph = gn.t[ft] - m_h - xh.t[ft] * m_hx;
    const f32x4 pb = bn.t[ft] - m_b - xh.t[ft] * m_bx;
    bn.t[ft] = pb * r - (xh.t[ft] * m_hpt + ph * mt + pt.t[ft] * m_hx) * rr2;
    gn.t[ft] = ph * r;
  }
}

// k2_atom<reverse> keeps the one-row-at-a-time form of its row-local part (t2_rows below: why); CHG_T2_ROWS=1: the tangent forward as well
#ifndef CHG_T2_ROWS
#define CHG_T2_ROWS 0
#endif
constexpr bool t2_rows_atom(bool reverse) { return CHG_T2_ROWS != 0 || reverse; }

struct Atom2Args {
  int n_edges;
  const int *e_center, *e_nbr;        // pair order (rows 2k, 2k+1 = the two directions of bond k)
  const float *P, *Q, *Pd, *Qd;       // first-layer tables [N,256], [Eu,128] and their tangents
  GatedW gw;
  const float *wag, *wagd;            // [Eu,64] smooth bond weights and tangents
  float* aggd;                        // tangent forward: [N,64] tangent of the aggregate (zeroed)
  // reverse sweep
  const float *bar_agg, *g_agg;       // [N,64] the two adjoints of the aggregate
  float* bar_w;                       // [Eu,64] bar adjoint of wag, accumulated over the layers (zeroed once per sweep)
  float *H, *Hd, *BCG, *GCG;          // [Ed,128] pair order: operands of the second-layer weight gradients
  float *park0, *park1;               // [Ed,128] x 2 scratch rows (the angle kernels' BZ / GZ dumps, idle here): silu'(z) | silu''(z) zd parked over the row loop
  float* barP;                        // [N,256] (zeroed): bar adjoint of the P table
  float *barQ, *gQ;                   // [Eu,128]: adjoints of the Q table, plain stores (the tile owns its bonds)
  // The G adjoints of everything that only LEAVES the sweep -- G(wag), G(P) -- are the first-order adjoints with seed 1: the
  // force sweep of chg_predict has left them in Gwag / GP_l[l]; only G(Q), which that sweep keeps for one layer at a time, is formed here.
  float* g_ln;                        // [4][64] LayerNorm-affine gradients
};

// AtomConv l: tangent forward (REVERSE = false: aggd only) or reverse sweep with the two adjoints.
template <bool REVERSE>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k2_atom(Atom2Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* I2c = reinterpret_cast<_Float16*>(smem);
  _Float16* I2g = I2c + rm_image_bytes(D, D) / 2;
  float* vecs = reinterpret_cast<float*>(I2g + rm_image_bytes(D, D) / 2);
  float* tiles = vecs + VEC_SLOTS * D;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  stage_rm(I2c, p.gw.w2c, D, D, tid, BLOCK);
  stage_rm(I2g, p.gw.w2g, D, D, tid, BLOCK);
  stage_gated_vecs(vecs, p.gw, true, tid);
  __syncthreads();
  float* T = tiles + wave * TILE_FLOATS;
  float* Trow = T + j * TS;
  [[maybe_unused]] float* Arow = T + (j & 7) * T2_AS;
  [[maybe_unused]] const float ln_g1 = vecs[2 * D + lane], ln_b1 = vecs[3 * D + lane], ln_g2 = vecs[4 * D + lane], ln_b2 = vecs[5 * D + lane];
  float lnacc[4] = {0.f, 0.f, 0.f, 0.f};
  const int ntiles = (p.n_edges + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range(ntiles, tb, te);
  const int last_row = p.n_edges - 1;
  for (int tile = tb; tile < te; ++tile) {
    // lane index opaque per tile: the 64-bit row pointers (base + 4 lane, one pair per buffer: a dozen of them) and row constants
    // derived from it are recomputed where used instead of living -- spilled -- across the kernel; a spilled value reloaded behind
    // stores or atomics waits for their round trip (kernels_angle_w.h, same measure)
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.n_edges - row0);   // even (pair order)
    if (nvalid <= 0) break;
    const int row = min(row0 + j, last_row);
    const int c = p.e_center[row], n = p.e_nbr[row], k = row >> 1, k0 = row0 >> 1;
    // ---- first layer from the tables (primal and tangent), hidden activation, second layer: core branch, then gate branch ----
    // (branch by branch: with both in flight the reverse kernel, which keeps silu' and silu'' zd of every row for the way back,
    // spilled 180 registers)
    V64 cc, cg, cdc, cdg, d1c, d1g, ec, eg;
    {
      GatherRegs gr;
      gather_issue128(gr, p.P, c, p.P + 2 * D, n, p.Q, k, 4 * D, 4 * D, 2 * D, lane_t);
      GatherRegs gd;
      gather_issue128(gd, p.Pd, c, p.Pd + 2 * D, n, p.Qd, k, 4 * D, 4 * D, 2 * D, lane_t);
      __builtin_amdgcn_wave_barrier();
      gather_commit128(gr, T, TS, lane_t);
      __builtin_amdgcn_wave_barrier();
      V64 zc, zg;
      read_dl<VT>(Trow, g, zc.t);
      read_dl<VT>(Trow + D, g, zg.t);
      __builtin_amdgcn_wave_barrier();
      gather_commit128(gd, T, TS, lane_t);
      __builtin_amdgcn_wave_barrier();
      float* hrow = p.H + (size_t)(row0 + j) * 2 * D;      // B operands of dW2 = bar(c|g)^T H + G(c|g)^T Hd
      float* hdrow = p.Hd + (size_t)(row0 + j) * 2 * D;
      {
        V64 zd, h, hd;
        read_dl<VT>(Trow, g, zd.t);
        hidden_t(zc, zd, h, hd, d1c, ec);
        if (REVERSE && j < nvalid) { write_dl<VT>(hrow, g, h.t); write_dl<VT>(hdrow, g, hd.t); }
        // silu'(z), silu''(z) zd come back on the way back through the hidden layer: parked in scratch rows (L2) instead of 64 registers
        // across the row loop (the kernel spilled 90 registers around them; a reload behind the loop's stores waits for their round trip)
        if (REVERSE && j < nvalid) { write_dl<VT>(p.park0 + (size_t)(row0 + j) * 2 * D, g, d1c.t); write_dl<VT>(p.park1 + (size_t)(row0 + j) * 2 * D, g, ec.t); }
        cc = param64(vecs + 0 * D, g);
        cdc = zero64();
        gemm_rm<VT, VT, false, false>(cc.t, I2c, D, D, h.t, j, g, lane_t);
        gemm_rm<VT, VT, true, false>(cdc.t, I2c, D, D, hd.t, j, g, lane_t);
      }
      {
        V64 zd, h, hd;
        read_dl<VT>(Trow + D, g, zd.t);
        hidden_t(zg, zd, h, hd, d1g, eg);
        if (REVERSE && j < nvalid) { write_dl<VT>(hrow + D, g, h.t); write_dl<VT>(hdrow + D, g, hd.t); }
        if (REVERSE && j < nvalid) { write_dl<VT>(p.park0 + (size_t)(row0 + j) * 2 * D + D, g, d1g.t); write_dl<VT>(p.park1 + (size_t)(row0 + j) * 2 * D + D, g, eg.t); }
        cg = param64(vecs + 1 * D, g);
        cdg = zero64();
        gemm_rm<VT, VT, false, false>(cg.t, I2g, D, D, h.t, j, g, lane_t);
        gemm_rm<VT, VT, true, false>(cdg.t, I2g, D, D, hd.t, j, g, lane_t);
      }
    }
    if constexpr (t2_rows_atom(REVERSE)) {
    // ---- row-local part, eight rows at a time through the tile (lane_t = feature) ----
    // Rolled loops: unrolled, the sixteen copies of the row math cost 175 spilled registers.  One bond (two rows) per step; what the
    // next bond needs from memory (adjoints of its two atoms' aggregates, its weight row, the old rows of the weight adjoints) is
    // requested a step ahead.
    float acc1 = 0.f, acc2 = 0.f;                                   // tangent forward: run sums of the aggregate tangent
    int cur1 = __builtin_amdgcn_readlane(c, 0), cur2 = __builtin_amdgcn_readlane(c, 1);
    const int nb = nvalid >> 1;
    struct BondIn { float ba0, ga0, ba1, ga1, w, wd, obw; };
    auto fetch = [&](int b) {                                        // b: bond of the tile (clamped: harmless reads past the end)
      BondIn v{};
      const int bb = min(b, nb - 1);
      const size_t kb = (size_t)(k0 + bb) * D + lane_t;
      v.w = p.wag[kb]; v.wd = p.wagd[kb];
      if (REVERSE) {
        const int c0 = __builtin_amdgcn_readlane(c, 2 * bb), c1 = __builtin_amdgcn_readlane(c, 2 * bb + 1);
        v.ba0 = p.bar_agg[(size_t)c0 * D + lane_t]; v.ga0 = p.g_agg[(size_t)c0 * D + lane_t];
        v.ba1 = p.bar_agg[(size_t)c1 * D + lane_t]; v.ga1 = p.g_agg[(size_t)c1 * D + lane_t];
        v.obw = p.bar_w[kb];
      }
      return v;
    };
    BondIn nx = fetch(0);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      __builtin_amdgcn_wave_barrier();
      if ((j >> 3) == half) {
        write_dl<VT>(Arow, g, cc.t); write_dl<VT>(Arow + D, g, cg.t);
        write_dl<VT>(Arow + 2 * D, g, cdc.t); write_dl<VT>(Arow + 3 * D, g, cdg.t);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll 1
      for (int pb = 0; pb < 4; ++pb) {
        const int b = 4 * half + pb;                     // bond of the tile
        if (b >= nb) break;
        const BondIn in = nx;
        nx = fetch(b + 1);
        float pair_bw = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d) {                    // the two directions
          float* a = T + (2 * pb + d) * T2_AS + lane_t;
          const GatedRow s = gated_row_fwd(a[0], a[D], a[2 * D], a[3 * D], ln_g1, ln_b1, ln_g2, ln_b2);
          if (!REVERSE) {
            const float m = s.yd * in.w + s.y * in.wd;
            const int ci = __builtin_amdgcn_readlane(c, 2 * b + d);
            if (d) {
              if (ci != cur2) { tile_atomic_add(p.aggd + (size_t)cur2 * D + lane_t, acc2); acc2 = 0.f; cur2 = ci; }
              acc2 += m;
            } else {
              if (ci != cur1) { tile_atomic_add(p.aggd + (size_t)cur1 * D + lane_t, acc1); acc1 = 0.f; cur1 = ci; }
              acc1 += m;
            }
          } else {
            const float bar_a = d ? in.ba1 : in.ba0, g_a = d ? in.ga1 : in.ga0;
            pair_bw += s.y * bar_a + s.yd * g_a;
            float bar_c, bar_g, g_c, g_g;
            gated_row_bwd(s, in.w * bar_a + in.wd * g_a, in.w * g_a, ln_g1, ln_g2, lnacc, bar_c, bar_g, g_c, g_g);
            a[0] = bar_c; a[D] = bar_g; a[2 * D] = g_c; a[3 * D] = g_g;
          }
        }
        if (REVERSE) {                                   // the tile owns bond k0 + b: plain update of the weight adjoints
          const size_t kb = (size_t)(k0 + b) * D + lane_t;
          p.bar_w[kb] = in.obw + pair_bw;
        }
      }
      if (REVERSE) {
        __builtin_amdgcn_wave_barrier();
        if ((j >> 3) == half) {
          read_dl<VT>(Arow, g, cc.t); read_dl<VT>(Arow + D, g, cg.t);
          read_dl<VT>(Arow + 2 * D, g, cdc.t); read_dl<VT>(Arow + 3 * D, g, cdg.t);
        }
      }
    }
    if (!REVERSE) {
      tile_atomic_add(p.aggd + (size_t)cur1 * D + lane_t, acc1);
      tile_atomic_add(p.aggd + (size_t)cur2 * D + lane_t, acc2);
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    } else {
    // ---- row-local part in the accumulator layout, sixteen rows at once (tangent forward only: t2_rows_atom) ----
    static_assert(!REVERSE, "the AtomConv reverse kernel keeps the one-row-at-a-time form (t2_rows_atom)");
    float r1, r2, mt1, mt2;
    ln2_forward(cc, cdc, r1, mt1);            // cc = xhat1, cdc = P(cd)
    ln2_forward(cg, cdg, r2, mt2);
    __builtin_amdgcn_wave_barrier();
    {
      f32x4 w = rd4(p.wag, k, 0, g), wd = rd4(p.wagd, k, 0, g);     // the bond's weight row and its tangent, one slice ahead
      CHG_EV(ft) {
        const f32x4 wn = rd4(p.wag, k, ft + 1 < VT ? ft + 1 : ft, g), wdn = rd4(p.wagd, k, ft + 1 < VT ? ft + 1 : ft, g);
        const f32x4 ga1 = vec4(vecs + 2 * D, ft, g), be1 = vec4(vecs + 3 * D, ft, g), ga2 = vec4(vecs + 4 * D, ft, g), be2 = vec4(vecs + 5 * D, ft, g);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const Gate1 e = gate1(cc.t[ft][r], cdc.t[ft][r], cg.t[ft][r], cdg.t[ft][r], r1, r2, ga1[r], be1[r], ga2[r], be2[r]);
          o[r] = e.yd * w[r] + e.y * wd[r];     // tangent of the message  y wag[k]
        }
        *reinterpret_cast<f32x4*>(Trow + 16 * ft + 4 * g) = o;
        w = wn; wd = wdn;
      }
    }
    __builtin_amdgcn_wave_barrier();
    {
      // tangent of the aggregates: run sums over the centres of the even rows (direction c1 -> c2, sorted) and of the odd rows
      float acc1 = 0.f, acc2 = 0.f;
      int cur1 = __builtin_amdgcn_readlane(c, 0), cur2 = __builtin_amdgcn_readlane(c, 1);
#pragma unroll
      for (int b = 0; b < TILE_ROWS / 2; ++b)
        if (2 * b < nvalid) {
          const float e0 = T[(2 * b) * TS + lane_t], o0 = T[(2 * b + 1) * TS + lane_t];
          const int c1 = __builtin_amdgcn_readlane(c, 2 * b), c2 = __builtin_amdgcn_readlane(c, 2 * b + 1);
          if (c1 != cur1) { tile_atomic_add(p.aggd + (size_t)cur1 * D + lane_t, acc1); acc1 = 0.f; cur1 = c1; }
          if (c2 != cur2) { tile_atomic_add(p.aggd + (size_t)cur2 * D + lane_t, acc2); acc2 = 0.f; cur2 = c2; }
          acc1 += e0; acc2 += o0;
        }
      tile_atomic_add(p.aggd + (size_t)cur1 * D + lane_t, acc1);
      tile_atomic_add(p.aggd + (size_t)cur2 * D + lane_t, acc2);
    }
    __builtin_amdgcn_wave_barrier();
    continue;
    }
    // cc | cg = bar(c | g), cdc | cdg = G(c | g)
    if (j < nvalid) {               // A operands of dW2 (column sums of bar(c|g) = d b2)
      float* brow = p.BCG + (size_t)(row0 + j) * 2 * D;
      float* grow = p.GCG + (size_t)(row0 + j) * 2 * D;
      write_dl<VT>(brow, g, cc.t); write_dl<VT>(brow + D, g, cg.t);
      write_dl<VT>(grow, g, cdc.t); write_dl<VT>(grow + D, g, cdg.t);
    }
    // ---- back through the second layer and the hidden activation, branch by branch:
    //      bar(z) = silu'(z) bar(H) + silu''(z) zd G(H),  G(z) = silu'(z) G(H);  bar(z) goes to the tile at once ----
    __builtin_amdgcn_wave_barrier();
    {
      const size_t po = (size_t)min(row0 + j, last_row) * 2 * D;
      read_dl<VT>(p.park0 + po, g, d1c.t); read_dl<VT>(p.park1 + po, g, ec.t);
      V64 bh = zero64(), gh = zero64();
      gemm_rm<VT, VT, true, true>(bh.t, I2c, D, D, cc.t, j, g, lane_t);
      gemm_rm<VT, VT, true, true>(gh.t, I2c, D, D, cdc.t, j, g, lane_t);
      CHG_EW(ft, r) {
        bh.t[ft][r] = d1c.t[ft][r] * bh.t[ft][r] + ec.t[ft][r] * gh.t[ft][r];
        cdc.t[ft][r] = d1c.t[ft][r] * gh.t[ft][r];
      }
      write_dl<VT>(Trow, g, bh.t);
    }
    {
      const size_t po = (size_t)min(row0 + j, last_row) * 2 * D + D;
      read_dl<VT>(p.park0 + po, g, d1g.t); read_dl<VT>(p.park1 + po, g, eg.t);
      V64 bh = zero64(), gh = zero64();
      gemm_rm<VT, VT, true, true>(bh.t, I2g, D, D, cg.t, j, g, lane_t);
      gemm_rm<VT, VT, true, true>(gh.t, I2g, D, D, cdg.t, j, g, lane_t);
      CHG_EW(ft, r) {
        bh.t[ft][r] = d1g.t[ft][r] * bh.t[ft][r] + eg.t[ft][r] * gh.t[ft][r];
        cdg.t[ft][r] = d1g.t[ft][r] * gh.t[ft][r];
      }
      write_dl<VT>(Trow + D, g, bh.t);
    }
    // ---- first-layer adjoints back to the tables (pair order: Q rows owned, both atoms as run sums) ----
    AtomConvArgs sc{};
    __builtin_amdgcn_wave_barrier();
    sc.GP = p.barP; sc.GQ = p.barQ;
    acbwd_scatter(T, c, nvalid, k0, sc, lane_t);
    __builtin_amdgcn_wave_barrier();
    // G side: G(P) is the first-order adjoint the force sweep of chg_predict left in GP_l[l] -- only the Q rows are formed here
    write_dl<VT>(Trow, g, cdc.t);
    write_dl<VT>(Trow + D, g, cdg.t);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < TILE_ROWS / 2; ++b)
      if (2 * b < nvalid) {
        float* q = p.gQ + (size_t)(k0 + b) * 2 * D + lane_t;
        q[0] = T[(2 * b) * TS + lane_t] + T[(2 * b + 1) * TS + lane_t];
        q[64] = T[(2 * b) * TS + 64 + lane_t] + T[(2 * b + 1) * TS + 64 + lane_t];
      }
    __builtin_amdgcn_wave_barrier();
  }
  if (REVERSE) {
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(p.g_ln + q * D + lane, lnacc[q]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// BondConv (HIDDEN) / AngleUpdate (single layer): rows are angles in the batch's order (owning bond b1 and centre atom sorted,
// second bond b2 not).  z = R[b1] + R[b2]' + S[ctr] + W_ang . ang: the angle block is contracted here (no [A,128] product arrays).
// ---------------------------------------------------------------------------------------------------------
struct Angle2Args {
  int n_angles;
  const int *a_ctr, *a_b1c, *a_b2c;
  const float *R, *S, *Rd, *Sd;       // tables [Eb,256], [N,128] and tangents
  const float *ang, *angd;            // [A,64] angle features of the layer and tangents
  const float* w_ang;                 // [128][64]
  GatedW gw;
  // BondConv: u = y * wbg[b1] * wbg[b2] summed over the angles of bond b1
  const float *w, *wd;                // [Eb,64] smooth bond weights (bond graph) and tangents
  float* aggd;                        // tangent forward: [Eb,64] (zeroed)
  const float *bar_agg, *g_agg;       // reverse: [Eb,64]
  float* bar_w;                       // reverse: [Eb,64] bar adjoint of wbg, accumulated over the layers (atomics); G(wbg), like
                                      // G(R) and G(S), is the first-order adjoint chg_predict's force sweep left behind
  // AngleUpdate: ang' = ang + y
  float* angd_out;                    // tangent forward: [A,64]
  // reverse, both kinds
  float *bar_ang, *g_ang;             // [A,64] adjoints of the angle features: read (AngleUpdate: adjoint of ang'), updated in place
  float *H, *Hd, *BCG, *GCG;          // [A,128] HIDDEN: operands of the second-layer weight gradients
  float *BZ, *GZ;                     // [A,128] first-layer adjoints: operands of the W_ang gradient
  float *barR, *barS;                 // [Eb,256], [N,128] (zeroed)
  float* g_ln;
};

template <bool HIDDEN>
constexpr size_t t2_angle_lds() {
  return rm_image_bytes(2 * D, D) + (HIDDEN ? 2 * rm_image_bytes(D, D) : 0) + sizeof(float) * (VEC_SLOTS * D + WAVES * TILE_FLOATS);
}

// Which instantiations keep the one-row-at-a-time form of the row-local part (lane = feature, kernels_train2.h): the reverse kernels
// WITH a hidden layer (BondConv here, AtomConv above).  Next to the hidden layer's state and six gathered rows per angle the two passes
// of the accumulator-layout form do not fit 256 registers (225-270 spilled registers: 40 ms instead of 29 per step), and a form that
// parks the row state in the dump rows and streams it back slice by slice waits for its own loads behind the scatter's atomics (57 ms);
// profiles/r05_experiments.md section 13.  CHG_T2_ROWS=1: every instantiation (A/B builds).
// CHG_T2_BOND_ACC=1 (A/B builds): the BondConv reverse kernel in the accumulator layout as well -- three passes over the slices, the
// first of which forms bar(y), G(y) from the six bond rows and keeps them in 32 registers, so that the other two read nothing.  Round 6,
// same box: 84 spilled registers (56 scratch stores, 77 loads per tile against 25 / 25), t2_bond_b 31.2 ms against 29.7 in the
// one-row-at-a-time form.  The passes themselves are the spill source: the same kernel cut off behind the BCG / GCG dump (the
// "recompute + adjoint of the tail" half of a two-kernel split) spills 169.
#ifndef CHG_T2_BOND_ACC
#define CHG_T2_BOND_ACC 0
#endif
constexpr bool t2_rows(bool hidden, bool reverse) { return CHG_T2_ROWS != 0 || (hidden && reverse && !CHG_T2_BOND_ACC); }

template <bool HIDDEN, bool REVERSE>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k2_angle(Angle2Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* Iang = reinterpret_cast<_Float16*>(smem);
  _Float16* I2c = Iang + rm_image_bytes(2 * D, D) / 2;
  _Float16* I2g = I2c + rm_image_bytes(D, D) / 2;
  float* vecs = reinterpret_cast<float*>(HIDDEN ? I2g + rm_image_bytes(D, D) / 2 : I2c);
  float* tiles = vecs + VEC_SLOTS * D;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  stage_rm(Iang, p.w_ang, 2 * D, D, tid, BLOCK);
  if (HIDDEN) {
    stage_rm(I2c, p.gw.w2c, D, D, tid, BLOCK);
    stage_rm(I2g, p.gw.w2g, D, D, tid, BLOCK);
  }
  stage_gated_vecs(vecs, p.gw, HIDDEN, tid);
  __syncthreads();
  float* T = tiles + wave * TILE_FLOATS;
  float* Trow = T + j * TS;
  [[maybe_unused]] float* Arow = T + (j & 7) * T2_AS;
  [[maybe_unused]] const float ln_g1 = vecs[2 * D + lane], ln_b1 = vecs[3 * D + lane], ln_g2 = vecs[4 * D + lane], ln_b2 = vecs[5 * D + lane];
  float lnacc[4] = {0.f, 0.f, 0.f, 0.f};
  const int ntiles = (p.n_angles + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int tb, te;
  tile_range(ntiles, tb, te);
  const int last_row = p.n_angles - 1;
  for (int tile = tb; tile < te; ++tile) {
    // lane index opaque per tile: the 64-bit row pointers (base + 4 lane, one pair per buffer: a dozen of them) and row constants
    // derived from it are recomputed where used instead of living -- spilled -- across the kernel; a spilled value reloaded behind
    // stores or atomics waits for their round trip (kernels_angle_w.h, same measure)
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    const int row0 = tile * BLOCK_ROWS + wave * TILE_ROWS;
    const int nvalid = min(TILE_ROWS, p.n_angles - row0);
    if (nvalid <= 0) break;
    const int row = min(row0 + j, last_row);
    const int b1 = p.a_b1c[row], b2 = p.a_b2c[row], ct = p.a_ctr[row];
    // ---- first layer: table sums + the angle block, primal and tangent; hidden activation and second layer branch by branch ----
    V64 cc, cg, cdc, cdg, d1c, d1g, ec, eg;
    {
      GatherRegs gr;
      gather_issue128(gr, p.R, b1, p.R + 2 * D, b2, p.S, ct, 4 * D, 4 * D, 2 * D, lane_t);
      GatherRegs gd;
      gather_issue128(gd, p.Rd, b1, p.Rd + 2 * D, b2, p.Sd, ct, 4 * D, 4 * D, 2 * D, lane_t);
      V64 x, xd;
      read_dl<VT>(p.ang + (size_t)row * D, g, x.t);
      read_dl<VT>(p.angd + (size_t)row * D, g, xd.t);
      __builtin_amdgcn_wave_barrier();
      gather_commit128(gr, T, TS, lane_t);
      __builtin_amdgcn_wave_barrier();
      f32x4 z[2 * VT], zd[2 * VT];
      read_dl<2 * VT>(Trow, g, z);
      __builtin_amdgcn_wave_barrier();
      gather_commit128(gd, T, TS, lane_t);
      __builtin_amdgcn_wave_barrier();
      read_dl<2 * VT>(Trow, g, zd);
      __builtin_amdgcn_wave_barrier();
      gemm_rm<VT, 2 * VT, false, false>(z, Iang, 2 * D, D, x.t, j, g, lane_t);
      gemm_rm<VT, 2 * VT, true, false>(zd, Iang, 2 * D, D, xd.t, j, g, lane_t);
#pragma unroll
      for (int ft = 0; ft < VT; ++ft) { cc.t[ft] = z[ft]; cg.t[ft] = z[VT + ft]; cdc.t[ft] = zd[ft]; cdg.t[ft] = zd[VT + ft]; }
      if (HIDDEN) {
        float* hrow = p.H + (size_t)(row0 + j) * 2 * D;      // B operands of dW2 = bar(c|g)^T H + G(c|g)^T Hd
        float* hdrow = p.Hd + (size_t)(row0 + j) * 2 * D;
        {
          V64 h, hd;
          hidden_t(cc, cdc, h, hd, d1c, ec);
          if (REVERSE && j < nvalid) { write_dl<VT>(hrow, g, h.t); write_dl<VT>(hdrow, g, hd.t); }
          // silu'(z), silu''(z) zd: parked in this row's BZ / GZ dump (written for good only after they are used again) instead of 64
          // registers across the row loop
          if (REVERSE && j < nvalid) { write_dl<VT>(p.BZ + (size_t)(row0 + j) * 2 * D, g, d1c.t); write_dl<VT>(p.GZ + (size_t)(row0 + j) * 2 * D, g, ec.t); }
          cc = param64(vecs + 0 * D, g);
          cdc = zero64();
          gemm_rm<VT, VT, false, false>(cc.t, I2c, D, D, h.t, j, g, lane_t);
          gemm_rm<VT, VT, true, false>(cdc.t, I2c, D, D, hd.t, j, g, lane_t);
        }
        {
          V64 h, hd;
          hidden_t(cg, cdg, h, hd, d1g, eg);
          if (REVERSE && j < nvalid) { write_dl<VT>(hrow + D, g, h.t); write_dl<VT>(hdrow + D, g, hd.t); }
          if (REVERSE && j < nvalid) { write_dl<VT>(p.BZ + (size_t)(row0 + j) * 2 * D + D, g, d1g.t); write_dl<VT>(p.GZ + (size_t)(row0 + j) * 2 * D + D, g, eg.t); }
          cg = param64(vecs + 1 * D, g);
          cdg = zero64();
          gemm_rm<VT, VT, false, false>(cg.t, I2g, D, D, h.t, j, g, lane_t);
          gemm_rm<VT, VT, true, false>(cdg.t, I2g, D, D, hd.t, j, g, lane_t);
        }
      }
    }
    if constexpr (t2_rows(HIDDEN, REVERSE)) {
    // ---- row-local part, eight rows at a time through the tile (lane_t = feature); what the next row needs from memory is
    //      requested a row ahead ----
    float acc = 0.f, acc_bw = 0.f;                    // BondConv: run sums over the owning bond
    float w1 = 0.f, w1d = 0.f, bar_a = 0.f, g_a = 0.f;
    int cur = -1;
    struct RowIn { float w2, w2d, by, gy; };
    auto fetch = [&](int rt) {                         // rt: row of the tile (clamped)
      RowIn v{};
      const int rc = min(rt, nvalid - 1);
      if (HIDDEN) {
        const size_t k2 = (size_t)__builtin_amdgcn_readlane(b2, rc) * D + lane_t;
        v.w2 = p.w[k2]; v.w2d = p.wd[k2];
      } else if (REVERSE) {
        const size_t o = (size_t)(row0 + rc) * D + lane_t;
        v.by = p.bar_ang[o]; v.gy = p.g_ang[o];
      } else {
        v.by = p.angd[(size_t)(row0 + rc) * D + lane_t];
      }
      return v;
    };
    RowIn nx = fetch(0);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      __builtin_amdgcn_wave_barrier();
      if ((j >> 3) == half) {
        write_dl<VT>(Arow, g, cc.t); write_dl<VT>(Arow + D, g, cg.t);
        write_dl<VT>(Arow + 2 * D, g, cdc.t); write_dl<VT>(Arow + 3 * D, g, cdg.t);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll 1
      for (int rr = 0; rr < 8; ++rr) {
        const int rt = 8 * half + rr;
        if (rt >= nvalid) break;
        const RowIn in = nx;
        nx = fetch(rt + 1);
        float* a = T + rr * T2_AS + lane_t;
        const GatedRow s = gated_row_fwd(a[0], a[D], a[2 * D], a[3 * D], ln_g1, ln_b1, ln_g2, ln_b2);
        float bar_y = 0.f, g_y = 0.f;
        if (HIDDEN) {
          const int dst = __builtin_amdgcn_readlane(b1, rt);
          if (dst != cur) {                              // a new owning bond: flush the run, fetch its rows
            if (cur >= 0) {
              if (!REVERSE) {
                tile_atomic_add(p.aggd + (size_t)cur * D + lane_t, acc);
              } else {
                atomicAdd(p.bar_w + (size_t)cur * D + lane_t, acc_bw);
              }
            }
            cur = dst;
            acc = acc_bw = 0.f;
            w1 = p.w[(size_t)dst * D + lane_t];
            w1d = p.wd[(size_t)dst * D + lane_t];
            if (REVERSE) {
              bar_a = p.bar_agg[(size_t)dst * D + lane_t];
              g_a = p.g_agg[(size_t)dst * D + lane_t];
            }
          }
          if (!REVERSE) {
            acc += s.yd * w1 * in.w2 + s.y * (w1d * in.w2 + w1 * in.w2d);
          } else {
            acc_bw += s.y * in.w2 * bar_a + (s.yd * in.w2 + s.y * in.w2d) * g_a;
            const size_t k2 = (size_t)__builtin_amdgcn_readlane(b2, rt) * D + lane_t;
            atomicAdd(p.bar_w + k2, s.y * w1 * bar_a + (s.yd * w1 + s.y * w1d) * g_a);
            bar_y = w1 * in.w2 * bar_a + (w1d * in.w2 + w1 * in.w2d) * g_a;
            g_y = w1 * in.w2 * g_a;
          }
        } else if (!REVERSE) {
          p.angd_out[(size_t)(row0 + rt) * D + lane_t] = in.by + s.yd;
        } else {
          bar_y = in.by;
          g_y = in.gy;
        }
        if (REVERSE) {
          float bar_c, bar_g, g_c, g_g;
          gated_row_bwd(s, bar_y, g_y, ln_g1, ln_g2, lnacc, bar_c, bar_g, g_c, g_g);
          a[0] = bar_c; a[D] = bar_g; a[2 * D] = g_c; a[3 * D] = g_g;
        }
      }
      if (REVERSE) {
        __builtin_amdgcn_wave_barrier();
        if ((j >> 3) == half) {
          read_dl<VT>(Arow, g, cc.t); read_dl<VT>(Arow + D, g, cg.t);
          read_dl<VT>(Arow + 2 * D, g, cdc.t); read_dl<VT>(Arow + 3 * D, g, cdg.t);
        }
      }
    }
    if (HIDDEN && cur >= 0) {
      if (!REVERSE) {
        tile_atomic_add(p.aggd + (size_t)cur * D + lane_t, acc);
      } else {
        atomicAdd(p.bar_w + (size_t)cur * D + lane_t, acc_bw);
      }
    }
    if (!REVERSE) {
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    } else {
    // ---- row-local part in the accumulator layout: sixteen rows at once (the helpers above) ----
    const bool valid = j < nvalid;
    const int kb1 = valid ? b1 : -1;
    [[maybe_unused]] const int kb2 = valid ? b2 : -1;
    float r1, r2, mt1, mt2;
    ln2_forward(cc, cdc, r1, mt1);            // cc = xhat1, cdc = P(cd): what the way back needs of the first LayerNorm
    ln2_forward(cg, cdg, r2, mt2);
    float sum[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // reverse, pass 1: this lane's share of the ten row sums
    float m[10];                              // reverse, pass 2: the row means  m_h, m_hx, m_hpt, m_b, m_bx  of branch 1, then of branch 2
    int sb1 = b1, sb2 = b2, srow = row, sg = g;   // the slices' row indices and column group (made opaque between the two reverse passes)
    // what a slice reads from memory, requested one slice ahead
    struct SliceIn { f32x4 w1, w1d, w2, w2d, a, b; };
    // BondConv reverse: bar(y), G(y) of the tile, formed once (pass 3, which reads the six bond rows anyway) and kept for passes 1 and 2,
    // which then read nothing from memory: 32 registers instead of 2 x 24 of slice inputs in flight and 2 x 24 loads per pass
    [[maybe_unused]] V64 by, gy;
    auto slice_in = [&](int ft, auto pass_c) {
      constexpr int PASS = decltype(pass_c)::value;
      SliceIn in;
      in.w1 = in.w1d = in.w2 = in.w2d = in.a = in.b = zero4();
      if (HIDDEN) {                          // BondConv: u = y wbg[b1] wbg[b2] summed over the angles of bond b1
        if (PASS == 0 || PASS == 3) {
          in.w1 = rd4(p.w, sb1, ft, sg); in.w1d = rd4(p.wd, sb1, ft, sg); in.w2 = rd4(p.w, sb2, ft, sg); in.w2d = rd4(p.wd, sb2, ft, sg);
        }
        if (PASS == 3) { in.a = rd4(p.bar_agg, sb1, ft, sg); in.b = rd4(p.g_agg, sb1, ft, sg); }     // the two adjoints of the owning bond's aggregate
      } else if (!REVERSE) {                 // AngleUpdate: ang' = ang + y
        in.a = rd4(p.angd, srow, ft, sg);
      } else {
        in.a = rd4(p.bar_ang, srow, ft, sg);   // bar(y), G(y) of the row
        in.b = rd4(p.g_ang, srow, ft, sg);
      }
      return in;
    };
    // One 4-feature slice, element by element.  PASS 0: tangent forward.  Reverse (AngleUpdate), two passes over the slices: PASS 1 forms
    // the ten row sums the two LayerNorm adjoints need (and sends the LayerNorm-affine terms of the second branch to the tile); PASS 2
    // RECOMPUTES the slice (2 x 16 transcendentals per lane more), finishes in place and sends the first branch's affine terms.  Keeping
    // the four adjoint arrays of the first pass instead costs 64 registers next to the 64 of xhat / P(cd) (spills).
    auto slice = [&](int ft, const SliceIn& in, auto pass_c) {
      constexpr int PASS = decltype(pass_c)::value;
      const f32x4 ga1 = vec4(vecs + 2 * D, ft, sg), be1 = vec4(vecs + 3 * D, ft, sg), ga2 = vec4(vecs + 4 * D, ft, sg), be2 = vec4(vecs + 5 * D, ft, sg);
      f32x4 o0 = in.a, o1 = zero4();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xh1 = cc.t[ft][r], pt1 = cdc.t[ft][r], xh2 = cg.t[ft][r], pt2 = cdg.t[ft][r];
        const Gate1 e = gate1(xh1, pt1, xh2, pt2, r1, r2, ga1[r], be1[r], ga2[r], be2[r]);
        if (PASS == 0) {
          if (HIDDEN) o0[r] = e.yd * in.w1[r] * in.w2[r] + e.y * (in.w1d[r] * in.w2[r] + in.w1[r] * in.w2d[r]);
          else o0[r] += e.yd;
        } else if (PASS == 3) {
          // BondConv reverse: bar adjoints of the two bond weights (first bond: a run sum over the tile's rows, second bond: a row each)
          o0[r] = e.y * in.w2[r] * in.a[r] + (e.yd * in.w2[r] + e.y * in.w2d[r]) * in.b[r];
          o1[r] = e.y * in.w1[r] * in.a[r] + (e.yd * in.w1[r] + e.y * in.w1d[r]) * in.b[r];
          // y enters u = y w1 w2: bar(y) = w1 w2 bar(u) + (w1 w2)_d G(u), G(y) = w1 w2 G(u)
          const float w12 = in.w1[r] * in.w2[r], w12d = in.w1d[r] * in.w2[r] + in.w1[r] * in.w2d[r];
          by.t[ft][r] = w12 * in.a[r] + w12d * in.b[r];
          gy.t[ft][r] = w12 * in.b[r];
        } else {
          float bar_y = in.a[r], g_y = in.b[r];
          if (HIDDEN) { bar_y = by.t[ft][r]; g_y = gy.t[ft][r]; }
          float bn1, gn1, bn2, gn2;
          gate1_bwd(e, bar_y, g_y, bn1, gn1, bn2, gn2);
          const float h1 = gn1 * ga1[r], p1 = bn1 * ga1[r], h2 = gn2 * ga2[r], p2 = bn2 * ga2[r];
          if (PASS == 1) {
            sum[0] += h1; sum[1] += h1 * xh1; sum[2] += h1 * pt1; sum[3] += p1; sum[4] += p1 * xh1;
            sum[5] += h2; sum[6] += h2 * xh2; sum[7] += h2 * pt2; sum[8] += p2; sum[9] += p2 * xh2;
            o0[r] = bn2 * xh2 + gn2 * (pt2 * r2);     // LayerNorm-affine terms (d gamma | d beta) of the second branch -> tile
            o1[r] = bn2;
          } else {
            o0[r] = bn1 * xh1 + gn1 * (pt1 * r1);     // ... of the first
            o1[r] = bn1;
            const float ph1 = h1 - m[0] - xh1 * m[1], pb1 = p1 - m[3] - xh1 * m[4];
            const float ph2 = h2 - m[5] - xh2 * m[6], pb2 = p2 - m[8] - xh2 * m[9];
            cc.t[ft][r] = pb1 * r1 - (xh1 * m[2] + ph1 * mt1 + pt1 * m[1]) * (r1 * r1);     // bar(c)
            cdc.t[ft][r] = ph1 * r1;                                                            // G(c)
            cg.t[ft][r] = pb2 * r2 - (xh2 * m[7] + ph2 * mt2 + pt2 * m[6]) * (r2 * r2);      // bar(g)
            cdg.t[ft][r] = ph2 * r2;                                                            // G(g)
          }
        }
      }
      if (HIDDEN || PASS != 0) *reinterpret_cast<f32x4*>(Trow + 16 * ft + 4 * g) = o0;
      if (PASS != 0) *reinterpret_cast<f32x4*>(Trow + D + 16 * ft + 4 * g) = o1;
      if (!HIDDEN && PASS == 0 && valid) *grow<f32x4>(p.angd_out, (unsigned)row, D, 16 * ft + 4 * g) = o0;
    };
    auto sweep = [&](auto pass_c) {
      __builtin_amdgcn_wave_barrier();
      SliceIn cur = slice_in(0, pass_c);
      CHG_EV(ft) {
        const SliceIn nxt = slice_in(ft + 1 < VT ? ft + 1 : ft, pass_c);
        slice(ft, cur, pass_c);
        cur = nxt;
      }
      __builtin_amdgcn_wave_barrier();
    };
    // column sums of the tile's two 64-wide halves over its valid rows (column = lane).  Into locals first: the four running sums live
    // across the whole kernel, and where the allocator keeps them in scratch a read-modify-write per row is sixteen dependent round trips
    auto tile_colsums = [&](float& d0, float& d1) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int rr = 0; rr < TILE_ROWS; ++rr) {
        const float a = T[rr * TS + lane_t], b = T[rr * TS + D + lane_t];
        s0 += rr < nvalid ? a : 0.f;
        s1 += rr < nvalid ? b : 0.f;
      }
      d0 += s0; d1 += s1;
    };
    if (!REVERSE) {
      sweep(std::integral_constant<int, 0>{});
      if (HIDDEN) seg_colsum_atomic<D>(T, TS, kb1, nvalid, p.aggd, D, lane_t);
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    if (HIDDEN) {                               // (before pass 2 overwrites xhat / P(cd))
      sweep(std::integral_constant<int, 3>{});
      seg_colsum_atomic<D>(T, TS, kb1, nvalid, p.bar_w, D, lane_t);
      row_atomic_add<D>(T + D, TS, kb2, nvalid, p.bar_w, D, lane_t);
      __builtin_amdgcn_wave_barrier();
    }
    sweep(std::integral_constant<int, 1>{});
    tile_colsums(lnacc[2], lnacc[3]);
#pragma unroll
    for (int q = 0; q < 10; ++q) m[q] = quad_sum(sum[q]) * (1.0f / 64.0f);
    // (the inputs made opaque: otherwise the compiler keeps pass 1's loads and arithmetic alive for pass 2 instead of recomputing)
    CHG_EW(ft, r) asm volatile("" : "+v"(cc.t[ft][r]), "+v"(cdc.t[ft][r]), "+v"(cg.t[ft][r]), "+v"(cdg.t[ft][r]));
    asm volatile("" : "+v"(sb1), "+v"(sb2), "+v"(srow), "+v"(sg), "+v"(r1), "+v"(r2));
    sweep(std::integral_constant<int, 2>{});
    tile_colsums(lnacc[0], lnacc[1]);
    __builtin_amdgcn_wave_barrier();
    }
    // cc | cg = bar(c | g), cdc | cdg = G(c | g)
    if (HIDDEN) {
      if (j < nvalid) {               // A operands of dW2 (column sums of bar(c|g) = d b2)
        float* brow = p.BCG + (size_t)(row0 + j) * 2 * D;
        float* grow = p.GCG + (size_t)(row0 + j) * 2 * D;
        write_dl<VT>(brow, g, cc.t); write_dl<VT>(brow + D, g, cg.t);
        write_dl<VT>(grow, g, cdc.t); write_dl<VT>(grow + D, g, cdg.t);
      }
      // back through the second layer and the hidden activation, branch by branch
      {
        const size_t po = (size_t)row * 2 * D;     // (rows past the end read the last row's: never stored)
        read_dl<VT>(p.BZ + po, g, d1c.t); read_dl<VT>(p.GZ + po, g, ec.t);
        V64 bh = zero64(), gh = zero64();
        gemm_rm<VT, VT, true, true>(bh.t, I2c, D, D, cc.t, j, g, lane_t);
        gemm_rm<VT, VT, true, true>(gh.t, I2c, D, D, cdc.t, j, g, lane_t);
        CHG_EW(ft, r) {
          cc.t[ft][r] = d1c.t[ft][r] * bh.t[ft][r] + ec.t[ft][r] * gh.t[ft][r];
          cdc.t[ft][r] = d1c.t[ft][r] * gh.t[ft][r];
        }
      }
      {
        const size_t po = (size_t)row * 2 * D + D;
        read_dl<VT>(p.BZ + po, g, d1g.t); read_dl<VT>(p.GZ + po, g, eg.t);
        V64 bh = zero64(), gh = zero64();
        gemm_rm<VT, VT, true, true>(bh.t, I2g, D, D, cg.t, j, g, lane_t);
        gemm_rm<VT, VT, true, true>(gh.t, I2g, D, D, cdg.t, j, g, lane_t);
        CHG_EW(ft, r) {
          cg.t[ft][r] = d1g.t[ft][r] * bh.t[ft][r] + eg.t[ft][r] * gh.t[ft][r];
          cdg.t[ft][r] = d1g.t[ft][r] * gh.t[ft][r];
        }
      }
    }
    // cc | cg = bar(z), cdc | cdg = G(z)
    if (j < nvalid) {                 // A operands of the W_ang gradient
      float* brow = p.BZ + (size_t)(row0 + j) * 2 * D;
      float* grow = p.GZ + (size_t)(row0 + j) * 2 * D;
      write_dl<VT>(brow, g, cc.t); write_dl<VT>(brow + D, g, cg.t);
      write_dl<VT>(grow, g, cdc.t); write_dl<VT>(grow + D, g, cdg.t);
    }
    const int k1 = j < nvalid ? b1 : -1, k2 = j < nvalid ? b2 : -1, k3 = j < nvalid ? ct : -1;
    // ---- bar: angle features (rows owned: plain update), then the tables ----
    {
      V64 old, up = zero64();
      float* arow = p.bar_ang + (size_t)row * D;
      read_dl<VT>(arow, g, old.t);
      f32x4 bz[2 * VT];
#pragma unroll
      for (int ft = 0; ft < VT; ++ft) { bz[ft] = cc.t[ft]; bz[VT + ft] = cg.t[ft]; }
      gemm_rm<2 * VT, VT, true, true>(up.t, Iang, 2 * D, D, bz, j, g, lane_t);
      CHG_EW(ft, r) up.t[ft][r] += old.t[ft][r];
      if (j < nvalid) write_dl<VT>(arow, g, up.t);
    }
    __builtin_amdgcn_wave_barrier();
    write_dl<VT>(Trow, g, cc.t);
    write_dl<VT>(Trow + D, g, cg.t);
    __builtin_amdgcn_wave_barrier();
    seg_colsum_atomic<2 * D>(T, TS, k1, nvalid, p.barR, 4 * D, lane_t);
    row_atomic_add<2 * D>(T, TS, k2, nvalid, p.barR + 2 * D, 4 * D, lane_t);
    seg_colsum_atomic<2 * D>(T, TS, k3, nvalid, p.barS, 2 * D, lane_t);
    __builtin_amdgcn_wave_barrier();
    // ---- G: only the angle features (an input of the earlier layers); G(R), G(S) are the first-order table adjoints ----
    {
      V64 old, up = zero64();
      float* arow = p.g_ang + (size_t)row * D;
      read_dl<VT>(arow, g, old.t);
      f32x4 gz[2 * VT];
#pragma unroll
      for (int ft = 0; ft < VT; ++ft) { gz[ft] = cdc.t[ft]; gz[VT + ft] = cdg.t[ft]; }
      gemm_rm<2 * VT, VT, true, true>(up.t, Iang, 2 * D, D, gz, j, g, lane_t);
      CHG_EW(ft, r) up.t[ft][r] += old.t[ft][r];
      if (j < nvalid) write_dl<VT>(arow, g, up.t);
    }
  }
  if (REVERSE) {
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(p.g_ln + q * D + lane, lnacc[q]);
  }
}

}  // namespace chg
