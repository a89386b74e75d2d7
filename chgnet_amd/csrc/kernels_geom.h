// kernels_geom.h -- geometry, basis functions, embeddings and the final force / virial step.
//
// Reference ops replaced (file:line relative to /root/reference/chgnet):
//   frac @ lattice, bond vectors under PBC      model/model.py:840, model/encoders.py:98-102
//   RadialBessel x CutoffPolynomial (both cutoffs) model/basis.py:108-116, 197-206
//   bond_embedding / bond_weights_ag / _bg      model/model.py:435-437
//   AngleEncoder (acos, Fourier) + angle_embedding model/encoders.py:144-146, basis.py:33-40, model.py:439
//   AtomEmbedding                               model/model.py:432-434
//   site_wise magmom head                       model/model.py:484-487
//   autograd of all of the above w.r.t. positions / strain  model/model.py:517-535
// (basis functions + 31->64 embeddings live in kernels_embed.h)
#pragma once

#include "mfma_tile.h"

namespace chg {

constexpr int NRAD = 31;
constexpr int NANG = 31;
constexpr int NFREQ = 15;
constexpr float KAPPA = 0.999999f;          // fp32(1 - 1e-6), encoders.py:144
constexpr float INV_SQRT_PI = 0.56418958354775628f;
constexpr float INV_SQRT_2 = 0.70710678118654752f;
constexpr float EV_A3_TO_GPA = 160.21766208f;  // model.py:532

// Sum over the 64 lanes, result in every lane, entirely in the VALU: four DPP adds inside each row of 16 lanes (quad
// swaps, then the half-row and row mirrors, which act as lane^4 / lane^8 once the smaller groups are uniform) and the
// two gfx950 row swaps of quad_sum across the four rows.  The ds_bpermute form (__shfl_xor) made the one-wave-per-row
// kernels wait on the LDS crossbar six times per reduction.
__device__ __forceinline__ float wave_sum(float v) {
#define CHG_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))
  CHG_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  CHG_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  CHG_DPP_ADD(0x141);   // row_half_mirror
  CHG_DPP_ADD(0x140);   // row_mirror
#undef CHG_DPP_ADD
  return quad_sum(v);
}
__device__ __forceinline__ float bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// ---- cartesian coordinates: x = frac @ lattice ------------------------------------------------
static __global__ void k_cart(const float* __restrict__ frac, const float* __restrict__ lattice, const int* __restrict__ owner,
                       float* __restrict__ cart, int n_atoms) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_atoms) return;
  const float* L = lattice + 9 * owner[i];
  const float f0 = frac[3 * i], f1 = frac[3 * i + 1], f2 = frac[3 * i + 2];
#pragma unroll
  for (int k = 0; k < 3; ++k) cart[3 * i + k] = fmaf(f2, L[6 + k], fmaf(f1, L[3 + k], f0 * L[k]));
}

// ---- directed bond vectors, lengths and unit vectors ---------------------------------------------
// ev[e] = (vx, vy, vz, r),  eu[e] = (ux, uy, uz, 0)
static __global__ void k_edge_geom(const float* __restrict__ cart, const float* __restrict__ lattice, const int* __restrict__ e_center,
                            const int* __restrict__ e_nbr, const float* __restrict__ e_image, const int* __restrict__ e_owner,
                            f32x4* __restrict__ ev, f32x4* __restrict__ eu, int n_edges) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const float* L = lattice + 9 * e_owner[e];
  const int c = e_center[e], n = e_nbr[e];
  const float i0 = e_image[3 * e], i1 = e_image[3 * e + 1], i2 = e_image[3 * e + 2];
  float v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float off = fmaf(i2, L[6 + k], fmaf(i1, L[3 + k], i0 * L[k]));  // image @ lattice
    const float nb = cart[3 * n + k] + off;                               // encoders.py:98
    v[k] = cart[3 * c + k] - nb;                                          // encoders.py:99
  }
  const float r = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  f32x4 a = {v[0], v[1], v[2], r};
  f32x4 b = {v[0] / r, v[1] / r, v[2] / r, 0.f};                           // zero-length bond -> NaN, as the reference
  ev[e] = a;
  eu[e] = b;
}

// ---- small batches: everything that precedes the basis expansions in ONE launch ------------------------------------------------
// An MD-size prediction is ~60 dependent launches of 4-70 us, and a launch costs ~4.5 us before its first instruction: k_cart,
// k_edge_geom, k_atom_embed, the memset of the scatter targets and the first AtomConv's P table (atom[0] = emb[z], so its table is a
// row of a per-ELEMENT table contracted once per weight upload: engine_predict.hip build_images) were five of them.  Same arithmetic
// per value as the kernels above (the edge's end points are transformed by the same fmaf chain k_cart uses).
struct PrologueArgs {
  const float *frac, *lattice;
  const int *atom_owner, *z;
  int n_atoms;
  const int *e_center, *e_nbr, *e_owner;
  const float* e_image;
  int n_edges;
  float* cart;
  f32x4 *ev, *eu;
  const float* emb;        // [94][64]
  float* atom0;            // [N][64]
  const float* p_elem;     // [94][256] P table of the first AtomConv per element (null: the table is contracted by its own launch)
  float* P0;               // [N][256]
  f32x4* zero_begin;       // range cleared for the sweeps (16-byte units)
  size_t zero_n;
};
static __global__ __launch_bounds__(256) void k_prologue(PrologueArgs p) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = tid; e < (size_t)p.n_edges; e += stride) {
    const float* L = p.lattice + 9 * p.e_owner[e];
    const int c = p.e_center[e], n = p.e_nbr[e];
    const float i0 = p.e_image[3 * e], i1 = p.e_image[3 * e + 1], i2 = p.e_image[3 * e + 2];
    const float c0 = p.frac[3 * c], c1 = p.frac[3 * c + 1], c2 = p.frac[3 * c + 2];
    const float n0 = p.frac[3 * n], n1 = p.frac[3 * n + 1], n2 = p.frac[3 * n + 2];
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float off = fmaf(i2, L[6 + k], fmaf(i1, L[3 + k], i0 * L[k]));
      const float xc = fmaf(c2, L[6 + k], fmaf(c1, L[3 + k], c0 * L[k]));      // k_cart
      const float xn = fmaf(n2, L[6 + k], fmaf(n1, L[3 + k], n0 * L[k]));
      v[k] = xc - (xn + off);
    }
    const float r = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    p.ev[e] = f32x4{v[0], v[1], v[2], r};
    p.eu[e] = f32x4{v[0] / r, v[1] / r, v[2] / r, 0.f};
  }
  for (size_t i = tid; i < (size_t)p.n_atoms; i += stride) {
    const float* L = p.lattice + 9 * p.atom_owner[i];
    const float f0 = p.frac[3 * i], f1 = p.frac[3 * i + 1], f2 = p.frac[3 * i + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) p.cart[3 * i + k] = fmaf(f2, L[6 + k], fmaf(f1, L[3 + k], f0 * L[k]));
  }
  for (size_t idx = tid; idx < (size_t)p.n_atoms * (D / 4); idx += stride) {
    const size_t i = idx / (D / 4), q = idx % (D / 4);
    reinterpret_cast<f32x4*>(p.atom0)[idx] = reinterpret_cast<const f32x4*>(p.emb + (size_t)(p.z[i] - 1) * D)[q];
  }
  if (p.p_elem)
    for (size_t idx = tid; idx < (size_t)p.n_atoms * D; idx += stride) {       // 4 D floats per atom = D 16-byte units
      const size_t i = idx / D, q = idx % D;
      reinterpret_cast<f32x4*>(p.P0)[idx] = reinterpret_cast<const f32x4*>(p.p_elem + (size_t)(p.z[i] - 1) * 4 * D)[q];
    }
  const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
  for (size_t idx = tid; idx < p.zero_n; idx += stride) p.zero_begin[idx] = zz;
}

// ---- radial basis helpers ----------------------------------------------------------------------
struct Envelope { float a, b, c; int p; };
__device__ __forceinline__ float ipow(float x, int n) {
  float r = 1.f, b = x;
  while (n) {
    if (n & 1) r *= b;
    b *= b;
    n >>= 1;
  }
  return r;
}
// sin and cos of x for |x| up to ~1e3 (the basis arguments reach 31 pi): three-term Cody-Waite reduction by pi/2 with FMAs (the
// products are exact, so the cancellation is), minimax polynomials on [-pi/4, pi/4] (cephes sinf / cosf coefficients), quadrant by
// bit tests: branch-free, ~22 vector instructions for BOTH values, max abs error 9.2e-8 on [0, 1000] (tests/test_split_numerics.py
// carries the float64 model of exactly this arithmetic).  OCML's sincosf (large-argument path and its branches, inlined 16 times per
// row) was most of the embedding kernels' time.
__device__ __forceinline__ void sincos_cw(float x, float& sn, float& cs) {
  const float n = __builtin_rintf(x * 0.636619772367581343f);         // 2 / pi
  float r = __builtin_fmaf(n, -1.5707963705062866f, x);               // pi/2 = HI + MID + LO
  r = __builtin_fmaf(n, 4.371138828673793e-08f, r);
  r = __builtin_fmaf(n, 1.7151245100058819e-15f, r);
  const float z = r * r;
  float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  const float s0 = __builtin_fmaf(ps * z, r, r);
  float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  const float c0 = __builtin_fmaf(pc * z, z, __builtin_fmaf(z, -0.5f, 1.0f));
  const int q = (int)n;
  const float sv = (q & 1) ? c0 : s0, cv = (q & 1) ? s0 : c0;
  sn = (q & 2) ? -sv : sv;
  cs = ((q + 1) & 2) ? -cv : cv;
}

// polynomial envelope u(s) = 1 + a s^p + b s^(p+1) + c s^(p+2) for s = r / rc < 1, else 0, and du/dr (basis.py:184-206): depends on
// (r, rc) only -- evaluated ONCE per row and cutoff, not once per basis function
struct EnvAt { float e, de; };
__device__ __forceinline__ EnvAt env_at(float r, float rc, Envelope env) {
  const float inv_rc = 1.0f / rc, s = r * inv_rc;
  EnvAt o{0.f, 0.f};
  if (s < 1.0f) {
    const float sp1 = ipow(s, env.p - 1), sp = sp1 * s;
    o.e = 1.0f + env.a * sp + env.b * sp * s + env.c * sp * s * s;
    o.de = (env.a * env.p * sp1 + env.b * (env.p + 1) * sp + env.c * (env.p + 2) * sp * s) * inv_rc;
  }
  return o;
}

// basis value and d/dr for one frequency (basis.py:108-116, 197-206)
// (dfreq: d val / d freq, needed by the weight-gradient path only)
__device__ __forceinline__ void rbf_eval(float r, float rc, float freq, EnvAt ea, float& val, float& dval, float& dfreq) {
  const float inv_rc = 1.0f / rc;
  const float ds = r * inv_rc;
  const float arg = freq * ds;
  const float norm = sqrtf(2.0f * inv_rc);
  float sn, cs;
  sincos_cw(arg, sn, cs);
  const float inv_r = 1.0f / r;
  const float base = norm * sn * inv_r;
  const float dbase = norm * (freq * inv_rc * cs - sn * inv_r) * inv_r;
  val = ea.e * base;
  dval = ea.de * base + ea.e * dbase;
  dfreq = ea.e * norm * cs * inv_rc;      // d/df [sin(f r / rc) / r] = cos(f r / rc) / rc
}

// ---- atom embedding ------------------------------------------------------------------------------
static __global__ void k_atom_embed(const int* __restrict__ z, const float* __restrict__ emb, float* __restrict__ out, int n_atoms) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_atoms * (D / 4)) return;
  const int i = idx / (D / 4), q = idx % (D / 4);
  reinterpret_cast<f32x4*>(out)[idx] = reinterpret_cast<const f32x4*>(emb + (size_t)(z[i] - 1) * D)[q];
}

// ---- magmom head: |h . w + b| ---------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_magmom(const float* __restrict__ atom, const float* __restrict__ w, const float* __restrict__ b,
                                                float* __restrict__ out, int n_atoms) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const float wl = w[lane], bb = b[0];
  for (int i = wave; i < n_atoms; i += nwaves) {
    const float s = wave_sum(atom[(size_t)i * D + lane] * wl);
    if (lane == 0) out[i] = fabsf(s + bb);
  }
}

// ---- dE/dv_e -> forces and per-structure virial (SURVEY App. B "Geometry") -----------------------
struct ForceArgs {
  const f32x4 *ev, *eu;
  const float* Gu;            // [Ed,4]
  const float* Grk;           // [Eu]
  const int *e_center, *e_d2u, *e_owner, *e_rev, *u_u2d;
  int n_edges;
  float* force;               // [N,3] zeroed
  float* virial;              // [B,9] zeroed
};

// F_i = -sum_{c_e = i} gv_e + sum_{n_e = i} gv_e with gv_e = dE/dv_e, and every edge with n_e = i is the reverse of an edge
// with c_e = i, so each thread forms gv_rev(e) - gv_e for its own edge and the force is a segmented sum over runs of equal
// centre (edges are centre-major): a wave-level segmented scan and one atomic per run end replace 6 same-address atomics
// per edge.  Runs are delimited by head flags (lane 0 or a key change), not by key equality alone, so hand-built graphs whose
// edges are not centre-major ([A,B,A]) stay correct: they only make the runs shorter.
//
// The reverse edge shares everything but the angular gradient with its partner (v_rev = -v, u_rev = -u, same length, same
// undirected bond, and exactly one of the two is the bond's representative u2d[k] that carries dE/dr), so ONE random 16-byte
// gather (Gu[e_rev]) replaces the three geometry rows and three index hops the first version re-read for it: 88 B per edge
// instead of ~140, one gather instead of six.  The virial is reduced per workgroup through LDS (one atomic set per 256 edges
// of a structure instead of one per wave).
//
// Each wave takes EF_IT consecutive 64-edge chunks with all their loads in flight together (one edge per thread left the wave
// slots waiting 95 % of their cycles: two dependent memory round trips and nothing to overlap them with), and the nine virial
// sums are formed once per wave over the chunks when they belong to one structure.
#ifndef CHG_EF_IT
#define CHG_EF_IT 4
#endif
#ifdef CHG_EF_WAVES
#define CHG_EF_ATTR __attribute__((amdgpu_waves_per_eu(CHG_EF_WAVES, CHG_EF_WAVES)))
#else
#define CHG_EF_ATTR
#endif
constexpr int EF_IT = CHG_EF_IT;    // measured: 2 -> 0.157 ms, 4 -> 0.145, 8 -> 0.190 (one edge per thread: 0.180)
constexpr int EF_EDGES_PER_BLOCK = 4 * 64 * EF_IT;
static __global__ __launch_bounds__(256) CHG_EF_ATTR void k_edge_force(ForceArgs p) {
  __shared__ float vir[4][9];
  __shared__ int vown[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = (blockIdx.x * 4 + wave) * (64 * EF_IT);
  f32x4 vr[EF_IT], u[EF_IT], gu[EF_IT], gw[EF_IT];
  int kk[EF_IT], er[EF_IT], owner[EF_IT], key[EF_IT];
  bool valid[EF_IT];
#pragma unroll
  for (int it = 0; it < EF_IT; ++it) {
    const int e = base + 64 * it + lane;
    valid[it] = e < p.n_edges;
    const int ec = valid[it] ? e : 0;
    vr[it] = p.ev[ec]; u[it] = p.eu[ec];
    kk[it] = p.e_d2u[ec]; er[it] = p.e_rev[ec];
    gu[it] = *reinterpret_cast<const f32x4*>(p.Gu + 4 * (size_t)ec);
    owner[it] = valid[it] ? p.e_owner[ec] : -1;
    key[it] = valid[it] ? p.e_center[ec] : -1;
  }
  float grk[EF_IT];
  int rep_e[EF_IT];
#pragma unroll
  for (int it = 0; it < EF_IT; ++it) {
    gw[it] = *reinterpret_cast<const f32x4*>(p.Gu + 4 * (size_t)er[it]);
    grk[it] = p.Grk[kk[it]];
    rep_e[it] = p.u_u2d[kk[it]];
  }
  float tv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // this lane's virial products summed over the chunks
  const int first = __builtin_amdgcn_readfirstlane(owner[0]);
  bool uniform = first >= 0;            // all chunks of the wave in one structure: one set of wave sums
#pragma unroll
  for (int it = 0; it < EF_IT; ++it) uniform = uniform && __all(owner[it] == first || owner[it] < 0) != 0;   // else chunk by chunk
#pragma unroll
  for (int it = 0; it < EF_IT; ++it) {
    const int e = base + 64 * it + lane;
    float gv[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
    if (valid[it]) {
      const bool rep = rep_e[it] == e;                      // lengths enter only via the representative edge of the bond
      const float gr = rep ? grk[it] : 0.f, gr_rev = rep ? 0.f : grk[it];
      const f32x4 uu = u[it], g0 = gu[it], g1 = gw[it];
      const float dotp = g0[0] * uu[0] + g0[1] * uu[1] + g0[2] * uu[2];
      const float dotr = g1[0] * uu[0] + g1[1] * uu[1] + g1[2] * uu[2];
      const float inv_r = 1.0f / vr[it][3];
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) {
        gv[k3] = gr * uu[k3] + (g0[k3] - dotp * uu[k3]) * inv_r;
        const float gvr = -gr_rev * uu[k3] + (g1[k3] - dotr * uu[k3]) * inv_r;   // u_rev = -u: (gw - (gw.u_rev) u_rev) = gw - (gw.u) u
        d[k3] = gvr - gv[k3];
        v[k3] = vr[it][k3];
      }
    }
    // segmented inclusive scan over runs of equal key: `start` = first lane of this lane's run
    // (DPP moves -- row_shr inside the 16-lane rows, then row_bcast:15 / row_bcast:31 carry the row totals -- instead of 20
    // ds_bpermute per chunk: the kernel waited on instruction ISSUE 43 % of its wave cycles, the LDS crossbar queue.)
    const int kcur = key[it];
    const int kprev = __builtin_amdgcn_update_dpp(-2, kcur, 0x138, 0xF, 0xF, false);    // wave_shr:1 (lane 0 keeps -2: a head anyway)
    const unsigned long long heads = __ballot(lane == 0 || kprev != kcur);
    const int start = 63 - __builtin_clzll(heads & (~0ull >> (63 - lane)));
#define CHG_SEG_STEP(ctrl, rmask, cond)                                                                                  \
    {                                                                                                                      \
      const float t0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d[0]), (ctrl), (rmask), 0xF, true)); \
      const float t1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d[1]), (ctrl), (rmask), 0xF, true)); \
      const float t2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d[2]), (ctrl), (rmask), 0xF, true)); \
      if (cond) {                                                                                                          \
        d[0] += t0;                                                                                                        \
        d[1] += t1;                                                                                                        \
        d[2] += t2;                                                                                                        \
      }                                                                                                                    \
    }
    const int row_lane = lane & 15, row_base = lane & ~15;
    CHG_SEG_STEP(0x111, 0xF, row_lane >= 1 && lane - 1 >= start)     // row_shr:1
    CHG_SEG_STEP(0x112, 0xF, row_lane >= 2 && lane - 2 >= start)     // row_shr:2
    CHG_SEG_STEP(0x114, 0xF, row_lane >= 4 && lane - 4 >= start)     // row_shr:4
    CHG_SEG_STEP(0x118, 0xF, row_lane >= 8 && lane - 8 >= start)     // row_shr:8
    CHG_SEG_STEP(0x142, 0xA, (row_base & 16) && start < row_base)    // row_bcast:15 -> rows 1 and 3: the run reaches into the row before
    CHG_SEG_STEP(0x143, 0xC, row_base >= 32 && start < 32)           // row_bcast:31 -> rows 2 and 3
#undef CHG_SEG_STEP
    const int knext = __builtin_amdgcn_update_dpp(-2, kcur, 0x130, 0xF, 0xF, false);    // wave_shl:1
    if (valid[it] && (lane == 63 || knext != kcur)) {
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) atomicAdd(p.force + 3 * (size_t)kcur + k3, d[k3]);
    }
    // virial: dE/d eps[a][b] = sum_e v_e[a] * gv_e[b]
    float t9[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) t9[3 * a + b] = v[a] * gv[b];
    if (uniform) {
#pragma unroll
      for (int c = 0; c < 9; ++c) tv[c] += t9[c];
    } else {
      // the wave spans structures: one set of wave sums per structure present in the chunk (one or two).  Per-edge atomics on the
      // chunks that straddle a boundary -- 576 same-address atomics each -- were 60 % of this kernel at 1,024 structures a batch.
      unsigned long long rem = __ballot(valid[it]);
      while (rem) {
        const int o = __builtin_amdgcn_readlane(owner[it], __builtin_ctzll(rem));
        const bool mine = valid[it] && owner[it] == o;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          const float sck = wave_sum(mine ? t9[c] : 0.f);
          if (lane == 0) atomicAdd(p.virial + 9 * (size_t)o + c, sck);
        }
        rem &= ~__ballot(mine);
      }
    }
  }
  // wave sums when the wave's edges sit in one structure, combined per workgroup
  if (lane == 0) vown[wave] = uniform ? first : -2;
  if (uniform) {
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const float s = wave_sum(tv[c]);
      if (lane == 0) vir[wave][c] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < 9) {   // waves of one structure leave as one atomic per component
    const int c = threadIdx.x;
    int cur = -2;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int o = vown[w];
      if (o != cur) {
        if (cur >= 0) atomicAdd(p.virial + 9 * (size_t)cur + c, s);
        s = 0.f;
        cur = o;
      }
      if (o >= 0) s += vir[w][c];
    }
    if (cur >= 0) atomicAdd(p.virial + 9 * (size_t)cur + c, s);
  }
}

// ---- per-structure finalisation: energy normalisation, AtomRef, stress scaling ---------------------
struct FinalizeArgs {
  const float* lattice;       // [B,9]
  const int* atom_off;        // [B+1]
  int n_struct;
  int is_intensive, has_composition, want_stress;
  const float* site_raw;      // [N] model site energies
  const int* z;               // [N]
  const float* atomref;       // [94]
  float* energy_out;          // [B]
  float* virial;              // [B,9] in: dE/d eps, out: stress in GPa
  float* volume;              // [B]
};

// One wave per structure: lane l sums atoms l, l + 64, ... in that order, the lanes meet in a fixed butterfly -- fp64, and the
// same operation order wherever the structure sits in the batch (one THREAD per structure walked a 256-atom MD cell in 41 us:
// 256 dependent loads).
static __global__ __launch_bounds__(256) void k_finalize(FinalizeArgs p) {
  const int lane = threadIdx.x & 63;
  const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (b >= p.n_struct) return;
  const float* L = p.lattice + 9 * b;
  const float cx = L[4] * L[8] - L[5] * L[7], cy = L[5] * L[6] - L[3] * L[8], cz = L[3] * L[7] - L[4] * L[6];
  const float vol = L[0] * cx + L[1] * cy + L[2] * cz;                 // model.py:834-836
  // per-structure sums in fp64 and a fixed order: the result does not depend on where the structure
  // sits in the batch (fp32 atomics in arrival order cost several ulp of the ~300 eV total)
  const int a0 = p.atom_off[b], a1 = p.atom_off[b + 1];
  double es = 0.0, cs = 0.0;
  for (int i = a0 + lane; i < a1; i += 64) {
    es += (double)p.site_raw[i];
    if (p.has_composition) cs += (double)p.atomref[p.z[i] - 1];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    es += __shfl_xor(es, off);
    cs += __shfl_xor(cs, off);
  }
  if (lane == 0) {
    p.volume[b] = vol;
    const double n = (double)(a1 - a0);
    double e = p.is_intensive ? es / n : es;                              // model.py:538-540
    if (p.has_composition) e += p.is_intensive ? cs / n : cs;             // model.py:378
    p.energy_out[b] = (float)e;
  }
  if (p.want_stress && lane < 9) p.virial[9 * b + lane] *= 1.0f / vol * EV_A3_TO_GPA;   // model.py:532
}

}  // namespace chg
