// kernels_angle_fa.h -- AngleUpdate FORWARD, one wave per atom, the atom's table rows in wave-private LDS.
//
// Reference op: AngleUpdate.forward (model/layers.py:348-360): new angle = angle + GatedMLP([bond_i | bond_j | atom | angle]).
//
// The row-order kernel (k_angle<false, false>) gathers, per angle, two 512-byte rows of the bond table R and one of the atom table S
// through L1 / L2: 24 load instructions, 24 lane shuffles, 64-bit address arithmetic and an LDS round trip per tile of 16 angles --
// 130 of its 505 vector instructions -- for 1.5 KB of gathered bytes per 256-byte angle row.  All n (n - 1) angles around an atom
// touch only its n short bonds (kernels_angle_w.h builds the centre-major order and deals the atoms to the waves): here a wave copies
// the atom's rows [R_i + S | R_j] ONCE into its own slice of LDS (1 KB per bond, FA_NSL = 15 rows; AngleUpdate has the room: its
// weights take 34 KB) and a tile adds two LDS rows per angle, addressed by the bonds' RANKS at the atom: 16 ds_read_b128 + 32 packed
// adds, no table loads, no shuffles, 32-bit addresses.  The rows of the NEXT atom are requested into registers while the current one
// is processed.  Exact f32 (the same additions in a different order).
//
// The first try of a per-atom forward (round 4, profiles/r04_rejected) kept the rows in REGISTERS and selected them with a one-hot
// contraction on the f32 matrix pipe: exact, but 1,536-2,048 matrix cycles per tile -- slower than the gathers it replaced.
//
// Atoms with more than 15 short bonds (dense oxides: up to ~20 within 3 A): the tiles that touch a rank past the LDS rows gather
// that tile's rows from the tables (a uniform branch per tile).  Batches without the canonical angle structure or too small for one
// atom per wave never launch this kernel's path (device flag, like the per-atom adjoints): the row-order kernel runs.
#pragma once

#include "kernels_angle_w.h"

namespace chg {

constexpr int FA_NSL = 15;                       // table rows per wave
constexpr int FA_RS = 4 * D + 4;                 // floats per row: [R_i + S (128) | R_j (128)] + pad (rows of a tile on different banks)
constexpr size_t angle_fa_lds() { return sizeof(float) * (size_t)(AngleLds<false, false>::tiles + WAVES * FA_NSL * FA_RS); }

struct FaRows { f32x4 r[FA_NSL]; f32x4 s; };     // one atom's table rows in flight: lane t holds floats 4 t .. 4 t + 3 of every row

// request the rows of the atom (c, n short bonds, bond list `bkv` in lanes 0 .. n - 1)
__device__ __forceinline__ void fa_request(FaRows& q, const AngleArgs& p, int c, int n, int bkv, int lane) {
#pragma unroll
  for (int k = 0; k < FA_NSL; ++k)
    if (k < n) q.r[k] = *reinterpret_cast<const f32x4*>(p.R + (size_t)(unsigned)__builtin_amdgcn_readlane(bkv, k) * 4 * D + 4 * lane);
  q.s = lane < 32 ? *reinterpret_cast<const f32x4*>(p.S + (size_t)c * 2 * D + 4 * lane) : zero4();
}
__device__ __forceinline__ void fa_commit(const FaRows& q, float* tab, int n, int lane) {
#pragma unroll
  for (int k = 0; k < FA_NSL; ++k)
    if (k < n) *reinterpret_cast<f32x4*>(tab + k * FA_RS + 4 * lane) = q.r[k] + q.s;      // q.s is zero in the R_j half
}

__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_angleupd_fwd_a(AngleWArgs pw) {
  const AngleArgs& p = pw.a;
  const WinIndex& w = pw.w;
  if (w.flag[0] != 1) return;                   // this batch runs the row-order forward (k_angle<false, false>)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using L = AngleLds<false, false>;
  const h16x8* Wang = reinterpret_cast<const h16x8*>(smem);
  const float* vecs = smem + L::vecs;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* tab = smem + L::tiles + wave * (FA_NSL * FA_RS);

  // ---- the wave's atoms: a list (k_win_schedule); headers run two atoms ahead ----
  int c_cur = w.wave_head[blockIdx.x * WAVES + wave];
  int n_raw = 0, ab_raw = 0, r_raw = 0, c_nx = -1;
  if (c_cur >= 0) { n_raw = w.na[c_cur]; ab_raw = w.boff[c_cur]; r_raw = w.aoff[c_cur]; c_nx = w.next_atom[c_cur]; }
  stage_image<L::tiles / 4, BLOCK>(smem, p.image, tid);
  __syncthreads();
  if (c_cur < 0) return;
  int n_nraw = 0, ab_nraw = 0, r_nraw = 0, c_n2 = -1;
  if (c_nx >= 0) { n_nraw = w.na[c_nx]; ab_nraw = w.boff[c_nx]; r_nraw = w.aoff[c_nx]; c_n2 = w.next_atom[c_nx]; }
  FaRows rows;                                   // rows of the atom that becomes current next (first: of the first atom)
  // Indices travel RAW (ranks are q - ab0 at the point of use): a subtraction right after the load would wait for it -- and, the
  // counter being in order, for every load issued before it.  Loads and stores of a tile are unconditional (clamped addresses,
  // masked lanes) so that the compiler's count of outstanding operations is the same on every path: its waits stay exact.
  int aA, q1A, q2A;                              // first tile of the atom that becomes current next: angle row, (atom, bond) pairs
  V64 xn;                                        // angle rows of the tile processed next
  {
    const int c = __builtin_amdgcn_readfirstlane(c_cur), n = __builtin_amdgcn_readfirstlane(n_raw);
    const int ab0 = __builtin_amdgcn_readfirstlane(ab_raw), r0 = __builtin_amdgcn_readfirstlane(r_raw);
    const int bkv = w.abbond[ab0 + min(lane, n - 1)];
    const int ra = min(r0 + j, r0 + n * (n - 1) - 1);
    aA = w.q_a[ra]; q1A = w.q_ab1[ra]; q2A = w.q_ab2[ra];
    fa_request(rows, p, c, min(n, FA_NSL), bkv, lane);
    read_dl_g<VT>(p.ang, (unsigned)aA, D, g, xn.t);
  }
  V64 y_prev;                                    // the previous tile's result: stored one tile late, behind this tile's requests
  CHG_EV(ft) y_prev.t[ft] = zero4();
  int a_prev = 0, nvalid_prev = 0;

  for (int c = __builtin_amdgcn_readfirstlane(c_cur); c >= 0; c = __builtin_amdgcn_readfirstlane(c_cur)) {
    const int n = __builtin_amdgcn_readfirstlane(n_raw), ab0 = __builtin_amdgcn_readfirstlane(ab_raw);
    const int r_begin = __builtin_amdgcn_readfirstlane(r_raw), r_end = r_begin + n * (n - 1);
    // ---- this atom's rows into the wave's LDS slice; the next atom's bond list, first indices and the header after it are requested ----
    __builtin_amdgcn_wave_barrier();
    fa_commit(rows, tab, min(n, FA_NSL), lane);
    __builtin_amdgcn_wave_barrier();
    const int c_next = __builtin_amdgcn_readfirstlane(c_nx);
    int bkv_n = 0, aA_n = 0, q1A_n = 0, q2A_n = -1;
    int n1 = 0, ab1 = 0, rb1 = 0;
    int n_n2raw = 0, ab_n2raw = 0, r_n2raw = 0, c_n3 = -1;
    if (c_next >= 0) {
      n1 = __builtin_amdgcn_readfirstlane(n_nraw); ab1 = __builtin_amdgcn_readfirstlane(ab_nraw); rb1 = __builtin_amdgcn_readfirstlane(r_nraw);
      bkv_n = w.abbond[ab1 + min(lane, n1 - 1)];
      const int ra = min(rb1 + j, rb1 + n1 * (n1 - 1) - 1);
      aA_n = w.q_a[ra]; q1A_n = w.q_ab1[ra]; q2A_n = w.q_ab2[ra];
      if (c_n2 >= 0) { n_n2raw = w.na[c_n2]; ab_n2raw = w.boff[c_n2]; r_n2raw = w.aoff[c_n2]; c_n3 = w.next_atom[c_n2]; }
    }
    int a_t = aA, q1_t = q1A, q2_t = q2A;
    int a_n, q1_n, q2_n;
    {                                            // indices of the second tile (clamped: of the last row)
      const int rr = min(r_begin + TILE_ROWS + j, r_end - 1);
      a_n = w.q_a[rr]; q1_n = w.q_ab1[rr]; q2_n = w.q_ab2[rr];
    }
    for (int row0 = r_begin; row0 < r_end; row0 += TILE_ROWS) {
      const int nvalid = min(TILE_ROWS, r_end - row0);
      asm volatile("" ::: "memory");   // the weight operands are re-read from LDS in every tile (hoisted out of the loop they are spilled)
      const V64 x = xn;
      const int a = a_t, r1 = q1_t - ab0, r2 = q2_t < 0 ? -1 : q2_t - ab0;
      const bool last = row0 + TILE_ROWS >= r_end;
      // requests for later tiles: angle rows of the next tile (after this atom's last tile: of the next atom's first), indices two ahead
      const int a_ld = last ? (c_next >= 0 ? aA_n : a) : a_n;
      read_dl_g<VT>(p.ang, (unsigned)a_ld, D, g, xn.t);
      const int rr2 = min(row0 + 2 * TILE_ROWS + j, r_end - 1);
      const int a_n2 = w.q_a[rr2], q1_n2 = w.q_ab1[rr2], q2_n2 = w.q_ab2[rr2];
      // ---- z = W_ang x + (R_i + S)[first bond] + R_j[second bond] ----
      f32x4 z[2 * VT];
#pragma unroll
      for (int fo = 0; fo < 2 * VT; ++fo) z[fo] = zero4();
      gemm_split<VT, 2 * VT, false, true>(z, Wang, 2 * D, x.t, j, g);
      // the previous tile's rows leave now: behind this tile's requests, so that no wait for those covers the stores
      asm volatile("" ::: "memory");
      if (j < nvalid_prev) write_dl_g<VT>(p.out, (unsigned)a_prev, D, g, y_prev.t);
      asm volatile("" ::: "memory");
      const bool in_lds = r1 < FA_NSL && r2 >= 0 && r2 < FA_NSL;        // (r1 >= 0 always: the row's own first bond)
      if (__builtin_amdgcn_ballot_w64(!in_lds) == 0) {
        const float* t1 = tab + r1 * FA_RS + 4 * g;
        const float* t2 = tab + r2 * FA_RS + 2 * D + 4 * g;
#pragma unroll
        for (int fo = 0; fo < 2 * VT; ++fo)
          z[fo] += *reinterpret_cast<const f32x4*>(t1 + 16 * fo) + *reinterpret_cast<const f32x4*>(t2 + 16 * fo);
      } else {                                                           // a rank past the LDS rows: this tile gathers from the tables
        const int row = min(row0 + j, r_end - 1);
        const float* g1 = p.R + (size_t)w.q_b1c[row] * 4 * D + 4 * g;
        const float* g2 = p.R + (size_t)w.q_b2c[row] * 4 * D + 2 * D + 4 * g;
        const float* gs = p.S + (size_t)c * 2 * D + 4 * g;
#pragma unroll
        for (int fo = 0; fo < 2 * VT; ++fo)
          z[fo] += (*reinterpret_cast<const f32x4*>(g1 + 16 * fo) + *reinterpret_cast<const f32x4*>(gs + 16 * fo)) +
                   *reinterpret_cast<const f32x4*>(g2 + 16 * fo);
      }
      // the next atom's rows are requested during this atom's first tile (its bond list, asked for at the top of the atom, has landed)
      if (row0 == r_begin && c_next >= 0) fa_request(rows, p, c_next, min(n1, FA_NSL), bkv_n, lane);
      V64 zc{{z[0], z[1], z[2], z[3]}}, zg{{z[4], z[5], z[6], z[7]}};
      GatedState st;
      V64 y;
      gated_forward<false, false, false, 1>(zc, zg, nullptr, nullptr, vecs, j, g, st, y);
      // new angle features = y + x, rows written from the accumulator layout (four 64-byte segments per row)
      CHG_EV(ft) y_prev.t[ft] = y.t[ft] + x.t[ft];
      a_prev = a; nvalid_prev = nvalid;
      a_t = a_n; q1_t = q1_n; q2_t = q2_n; a_n = a_n2; q1_n = q1_n2; q2_n = q2_n2;
    }
    // ---- rotate the atom pipeline ----
    c_cur = c_next; n_raw = n_nraw; ab_raw = ab_nraw; r_raw = r_nraw;
    c_nx = c_n2; n_nraw = n_n2raw; ab_nraw = ab_n2raw; r_nraw = r_n2raw; c_n2 = c_n3;
    aA = aA_n; q1A = q1A_n; q2A = q2A_n;
  }
  if (j < nvalid_prev) write_dl_g<VT>(p.out, (unsigned)a_prev, D, g, y_prev.t);
}

}  // namespace chg
