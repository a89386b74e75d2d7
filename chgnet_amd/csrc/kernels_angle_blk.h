// kernels_angle_blk.h -- adjoints of BondConv / AngleUpdate for MD-size batches over BLOCKED angle tiles.
//
// Why (profiles/r06_experiments.md section 14): at a few hundred atoms the row-order adjoints (k_angle<.., true>) are bound by the
// fp32 atomics of their scatter -- the L2 executes them at one lane per clock and channel, ~1.3 TB/s chip-wide whatever the scope or
// the sharing (tools/lab/atomic_scope_lab.hip) -- and nearly all of those are the SECOND bond's rows: a 16-row tile of the reference's
// order is 16 angles with one first bond and 16 different second bonds (939 B of atomics per angle for BondConv).  The per-atom and
// TEAM kernels (kernels_angle_w.h) take them off the chip's atomic units by summing an atom's n (n - 1) rows in LDS, but an atom is a
// coarse unit of work when the whole batch is a few thousand tiles: the launch lasts as long as its largest atom / pays two
// workgroup barriers per atom segment.
//
// Here a tile is a P x Q BLOCK of the atom's n x (n - 1) matrix (first bond, position of the second bond among the other n - 1 bonds):
// slot il Q + jl holds first bond i0 + il and position j'0 + jl.  P x Q = 4 x 4, 2 x 8 or 8 x 2 per atom, whichever takes the fewest tiles
// (blk_shape_of, kernels_graph.h: 90 % of the slots hold an angle on a thermalised cell; 4 x 4 blocks over the n x n matrix with its
// diagonal: 78 %).  The Q rows of a first bond are summed in registers; so are the rows of a second bond -- position j' is bond j' below the
// diagonal and bond j' + 1 on / above it, so a column leaves as a "below" sum and an "on / above" sum and neighbouring columns share a
// bond: Q + 1 second-bond rows per tile.  A tile sends P + Q + 1 <= 11 rows per scatter target instead of 1-2 + 16 -- ~550 B of atomics
// per angle for BondConv -- and every tile is self-contained: any wave takes any tile, no barrier, no private rows, no schedule.  Empty
// slots (the ragged edges) read row 0 of the tables and their upstream gradient is set to zero, so they add zeros.
//
// Index: the device graph builder writes slot -> angle while it emits the angles (k_angle_fill, from the ranks it counts anyway) and
// k_multi_copy adds the compact bond indices: nothing is launched for it.  Uploaded graphs get the same index from the centre-major
// order of kernels_angle_w.h (k_win_*: ranks of the bonds at their atom) and k_blk_from_q below, once per upload; a graph without the
// canonical angle structure leaves WinIndex::flag at 0 and the row-order adjoint, launched behind this one, does the work.
#pragma once

#include "kernels_angle_w.h"

namespace chg {

struct BlkIndex {
  const int* tiles;                       // [1] number of 16-slot tiles (device quantity: sum of blk_shape_of(n) over the atoms)
  const int *a, *b1c, *b2c, *ctr;         // [16 tiles] angle (-1: empty slot), compact bond indices, centre atom
  const int* desc;                        // [tiles] log2 P | log2 Q << 4 | i0 << 8 | j'0 << 16
  const int* flag;                        // uploaded graphs: WinIndex::flag (1: the graph has the canonical angle structure, the index is valid); else null
};
struct AngleBlkArgs {
  AngleArgs a;
  BlkIndex x;
};

template <bool HIDDEN>
constexpr size_t angle_blk_lds() {
  return sizeof(float) * ((size_t)AngleLds<HIDDEN, true>::tiles + WAVES * TILE64_FLOATS);
}

// The tile's scatter for one 64-wide array held column-wise (lane = column, c.v[slot]; empty slots hold zeros): rows of dst1 keyed by the
// first bond (ROWS: slots il Q .. il Q + Q - 1 share it) and rows of dst2 keyed by the second bond (COLS: position jl is bond j'0 + jl for
// the slots below the diagonal, j'0 + jl + 1 on / above it -- `ge`, bit per slot -- so output k of 0 .. Q sums column k's "below" part and
// column k - 1's "on / above" part).  vmask: bit per slot that holds an angle; keys: per-lane values of lanes 0-15.
template <int PS, int QS, bool ROWS, bool COLS>
__device__ __forceinline__ void blk_scatter(const Cols64& c, unsigned vmask, unsigned ge, int key1, int key2, float* __restrict__ dst1,
                                            float* __restrict__ dst2, int ld, int lane) {
  constexpr int P = 1 << PS, Q = 1 << QS;
  if (ROWS) {
#pragma unroll
    for (int il = 0; il < P; ++il) {
      const unsigned gm = (vmask >> (il * Q)) & ((1u << Q) - 1u);
      if (gm) {
        float s = 0.f;
#pragma unroll
        for (int jl = 0; jl < Q; ++jl) s += c.v[il * Q + jl];
        atomicAdd(grow<float>(dst1, (unsigned)__builtin_amdgcn_readlane(key1, il * Q + __builtin_ctz(gm)), ld, lane), s);
      }
    }
  }
  if (COLS) {
    float lo[Q], hi[Q];
    unsigned col = 0;                        // slots of column 0
#pragma unroll
    for (int il = 0; il < P; ++il) col |= 1u << (il * Q);
    if ((ge & vmask) == 0u || ((~ge) & vmask) == 0u) {          // the tile does not cross the diagonal: plain column sums
      const int up = (ge & vmask) ? 1 : 0;
#pragma unroll
      for (int jl = 0; jl < Q; ++jl) {
        const unsigned cm = vmask & (col << jl);
        if (cm) {
          float s = 0.f;
#pragma unroll
          for (int il = 0; il < P; ++il) s += c.v[il * Q + jl];
          atomicAdd(grow<float>(dst2, (unsigned)__builtin_amdgcn_readlane(key2, __builtin_ctz(cm)), ld, lane), s);
        }
      }
      (void)up;
      return;
    }
#pragma unroll
    for (int jl = 0; jl < Q; ++jl) {
      lo[jl] = 0.f; hi[jl] = 0.f;
#pragma unroll
      for (int il = 0; il < P; ++il) {
        const int sl = il * Q + jl;
        const float v = c.v[sl];
        const bool up = (ge >> sl) & 1u;       // uniform
        lo[jl] += up ? 0.f : v;
        hi[jl] += up ? v : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k <= Q; ++k) {
      const unsigned ml = k < Q ? (vmask & ~ge & (col << (k < Q ? k : 0))) : 0u;
      const unsigned mh = k > 0 ? (vmask & ge & (col << (k > 0 ? k - 1 : 0))) : 0u;
      if (ml | mh) {
        const float s = (k < Q ? lo[k < Q ? k : 0] : 0.f) + (k > 0 ? hi[k > 0 ? k - 1 : 0] : 0.f);
        atomicAdd(grow<float>(dst2, (unsigned)__builtin_amdgcn_readlane(key2, __builtin_ctz(ml | mh)), ld, lane), s);
      }
    }
  }
}
// ... dispatched on the tile's shape (uniform)
template <bool ROWS, bool COLS>
__device__ __forceinline__ void blk_scatter_any(int ps, const Cols64& c, unsigned vmask, unsigned ge, int key1, int key2, float* __restrict__ dst1,
                                                float* __restrict__ dst2, int ld, int lane) {
  if (ps == 2) blk_scatter<2, 2, ROWS, COLS>(c, vmask, ge, key1, key2, dst1, dst2, ld, lane);
  else if (ps == 1) blk_scatter<1, 3, ROWS, COLS>(c, vmask, ge, key1, key2, dst1, dst2, ld, lane);
  else blk_scatter<3, 1, ROWS, COLS>(c, vmask, ge, key1, key2, dst1, dst2, ld, lane);
}

// rows idx (< 0: none) of dst = old + tile: the read-modify-write of rows this tile owns
__device__ __forceinline__ void scatter_rows64_add_masked(const float* tile, int stride, float* __restrict__ dst, int idx, int lane, const Rows64& old) {
  const int sub = lane >> 4, t = lane & 15;
  f32x4 v[TILE_ROWS / 4];
  int r[TILE_ROWS / 4];
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    const int rr = 4 * it + sub;
    r[it] = __shfl(idx, rr);
    v[it] = old.v[it] + *reinterpret_cast<const f32x4*>(tile + rr * stride + 4 * t);
  }
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) asm volatile("" : "+v"(v[it]));
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it)
    if (r[it] >= 0) *grow<f32x4>(dst, (unsigned)r[it], D, 4 * t) = v[it];
}

// The index of an UPLOADED graph from its centre-major order (WinIndex: q_a .. q_ab2 after k_win_rows, toff4 after k_win_scan2): row -> slot.
// blk_a was set to -1, the other slot arrays to 0 (memsets).
static __global__ void k_blk_from_q(int A, int N, WinIndex w, int* __restrict__ blk_a, int* __restrict__ blk_b1c, int* __restrict__ blk_b2c,
                                    int* __restrict__ blk_ctr, int* __restrict__ blk_desc, int* __restrict__ blk_tiles, int cap_tiles) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row == 0) *blk_tiles = min(w.toff4[N], cap_tiles);
  if (row >= A || w.flag[0] != 1) return;
  const int c = w.q_ctr[row], n = w.na[c], ab0 = w.boff[c];
  const int i = w.q_ab1[row] - ab0, r2 = w.q_ab2[row] < 0 ? -1 : w.q_ab2[row] - ab0;
  if (i < 0 || i >= n || r2 < 0 || r2 >= n || r2 == i) { w.flag[0] = 0; return; }    // the second bond is not one of the atom's first bonds
  const int jp = r2 - (r2 > i ? 1 : 0);
  int ps, qs;
  blk_shape_of(n, ps, qs);
  const int nq = (n - 1 + (1 << qs) - 1) >> qs;
  const long tile = (long)w.toff4[c] + (long)(i >> ps) * nq + (jp >> qs);
  if (tile >= cap_tiles) { w.flag[0] = 0; return; }
  const size_t sl = (size_t)tile * 16 + ((i & ((1 << ps) - 1)) << qs) + (jp & ((1 << qs) - 1));
  blk_a[sl] = w.q_a[row]; blk_b1c[sl] = w.q_b1c[row]; blk_b2c[sl] = w.q_b2c[row]; blk_ctr[sl] = c;
  blk_desc[tile] = ps | (qs << 4) | ((i >> ps << ps) << 8) | ((jp >> qs << qs) << 16);
}

template <bool HIDDEN>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_angle_bwd_blk(AngleBlkArgs pb) {
  const AngleArgs& p = pb.a;
  const BlkIndex& x = pb.x;
  if (x.flag && *x.flag != 1) return;           // not a canonical graph: the row-order adjoint launched behind this one runs
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the weight block of k_angle<HIDDEN, true> / k_angle_bwd_w (prebuilt image: AngleLds<HIDDEN, true>)
  constexpr int MODE = HIDDEN ? 2 : 1;
  float* Wang = smem;
  float* WangT = HIDDEN ? Wang : Wang + 4 * IMG128;
  float* W2c = HIDDEN ? Wang + WIN_RM_ANG / 4 : WangT + 4 * IMG128;
  float* W2g = W2c + (HIDDEN ? WIN_RM_W2 / 4 : 0);
  float* vecs = W2g + (HIDDEN ? WIN_RM_W2 / 4 : 0);
  float* tiles = vecs + VEC_SLOTS * D;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this wave's tiles: t0, t0 + nw, ... (neighbouring waves on neighbouring tiles: one atom's rows of the tables stay in one L2)
  const int nw = (int)gridDim.x * WAVES, t0 = (int)blockIdx.x * WAVES + wave;
  const int ntiles = __builtin_amdgcn_readfirstlane(*x.tiles);
  int a_n = -1, b1_n = 0, b2_n = 0, c_n = 0, d_n = 0x22;
  if (t0 < ntiles) {
    const size_t sl = (size_t)t0 * TILE_ROWS + j;
    a_n = x.a[sl]; b1_n = x.b1c[sl]; b2_n = x.b2c[sl]; c_n = x.ctr[sl]; d_n = x.desc[t0];
  }
  stage_image<AngleLds<HIDDEN, true>::tiles / 4, BLOCK>(smem, p.image, tid);
  __syncthreads();
  float* T = tiles + wave * TILE64_FLOATS;
  float* Trow = T + j * TS64;
  for (int t = t0; t < ntiles; t += nw) {
    int lane_t = lane;
    if (HIDDEN) asm volatile("" : "+v"(lane_t));     // as in k_angle_bwd_w: row pointers formed where they are used
    const int a_raw = a_n, b1 = b1_n, b2 = b2_n, c = c_n;
    const int desc = __builtin_amdgcn_readfirstlane(d_n);
    if (t + nw < ntiles) {
      const size_t sl = (size_t)(t + nw) * TILE_ROWS + j;
      a_n = x.a[sl]; b1_n = x.b1c[sl]; b2_n = x.b2c[sl]; c_n = x.ctr[sl]; d_n = x.desc[t + nw];
    }
    const bool valid = a_raw >= 0;
    const int a = valid ? a_raw : 0;
    const unsigned vmask = (unsigned)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(valid) & 0xFFFFull));
    // the tile's shape and origin; ge: bit per slot whose position j' lies on / above the diagonal (its second bond is j' + 1)
    const int ps = desc & 15, qs = (desc >> 4) & 15, i0 = (desc >> 8) & 255, j0 = (desc >> 16) & 255;
    const unsigned ge = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(__builtin_amdgcn_ballot_w64(j0 + (j & ((1 << qs) - 1)) >= i0 + (j >> qs)) & 0xFFFFull));
    // ---- gathers: angle rows, the two halves of the table sum ----
    f32x4 z[2 * VT];
    Rows64 gy_rows;
    V64 w1, w2, gu;
    {
      Gather64 gc, gg;
      gather64_issue(gc, p.R, b1, 4 * D, p.R + 2 * D, b2, 4 * D, p.S, c, 2 * D, lane_t);
      gather64_issue(gg, p.R + D, b1, 4 * D, p.R + 3 * D, b2, 4 * D, p.S + D, c, 2 * D, lane_t);
      gather_rows64(T, TS64, p.ang, a, lane_t);
      __builtin_amdgcn_wave_barrier();
      V64 xr;
      read_dl<VT>(Trow, g, xr.t);
      __builtin_amdgcn_wave_barrier();
      gather64_commit(gc, T, lane_t);
      __builtin_amdgcn_wave_barrier();
      read_dl<VT>(Trow, g, *reinterpret_cast<f32x4(*)[VT]>(&z[0]));
      __builtin_amdgcn_wave_barrier();
      gather64_commit(gg, T, lane_t);
      __builtin_amdgcn_wave_barrier();
      read_dl<VT>(Trow, g, *reinterpret_cast<f32x4(*)[VT]>(&z[VT]));
      __builtin_amdgcn_wave_barrier();
      if (!HIDDEN) rows64_issue(gy_rows, p.Gang, a, lane_t);
      if (HIDDEN) {
        read_dl_g<VT>(p.wbgc, (unsigned)b1, D, g, w1.t);
        read_dl_g<VT>(p.wbgc, (unsigned)b2, D, g, w2.t);
        read_dl_g<VT>(p.Gagg, (unsigned)b1, D, g, gu.t);
      }
      if (HIDDEN) gemm_rm<VT, 2 * VT, false, false>(z, reinterpret_cast<const _Float16*>(Wang), 2 * D, D, xr.t, j, g, lane_t);
      else gemm_split<VT, 2 * VT, false>(z, reinterpret_cast<const h16x8*>(Wang), 2 * D, xr.t, j, g);
    }
    V64 zc{{z[0], z[1], z[2], z[3]}}, zg{{z[4], z[5], z[6], z[7]}};
    GatedState s;
    V64 y;
    constexpr bool SLIM = HIDDEN;
    gated_forward<HIDDEN, SLIM, false, MODE>(zc, zg, W2c, W2g, vecs, j, g, s, y);
    // ---- upstream gradient; an empty slot's is zero, and with it everything the slot scatters ----
    V64 gy;
    if (HIDDEN) {
      V64 g1, g2;
      CHG_EV(ft) {
        const f32x4 guv = valid ? gu.t[ft] : zero4();
        const f32x4 gyu = guv * y.t[ft];
        g1.t[ft] = gyu * w2.t[ft];      // dE/d wbgc[b1]
        g2.t[ft] = gyu * w1.t[ft];      // dE/d wbgc[b2]
        gy.t[ft] = guv * w1.t[ft] * w2.t[ft];
      }
      Cols64 c1, c2;
      to_columns(g1, T, Trow, g, lane_t, c1);
      to_columns(g2, T, Trow, g, lane_t, c2);
      blk_scatter_any<true, false>(ps, c1, vmask, ge, b1, b2, p.Gwbgc, p.Gwbgc, D, lane_t);      // first-bond sums of g1
      blk_scatter_any<false, true>(ps, c2, vmask, ge, b1, b2, p.Gwbgc, p.Gwbgc, D, lane_t);      // second-bond sums of g2
    } else {
      rows64_commit(gy_rows, T, TS64, lane_t);
      __builtin_amdgcn_wave_barrier();
      read_dl<VT>(Trow, g, gy.t);
      __builtin_amdgcn_wave_barrier();
      CHG_EV(ft) gy.t[ft] = valid ? gy.t[ft] : zero4();
    }
    V64 gzc, gzg;
    gated_backward<HIDDEN, SLIM, false, MODE>(gy, zc, zg, W2c, W2g, vecs, j, g, s, gzc, gzg);
    // ---- dE/d(angle in) += W_ang^T gz ----
    {
      f32x4 gz[2 * VT] = {gzc.t[0], gzc.t[1], gzc.t[2], gzc.t[3], gzg.t[0], gzg.t[1], gzg.t[2], gzg.t[3]};
      V64 ga = zero64();
      Rows64 gang_old;
      const bool fresh = HIDDEN && p.first_gang;
      if (fresh) {
#pragma unroll
        for (int it = 0; it < TILE_ROWS / 4; ++it) gang_old.v[it] = zero4();
      } else {
        rows64_issue(gang_old, p.Gang, a, lane_t);
      }
      if (HIDDEN) gemm_rm<2 * VT, VT, true, true>(ga.t, reinterpret_cast<const _Float16*>(Wang), 2 * D, D, gz, j, g, lane_t);
      else gemm_split<2 * VT, VT, true>(ga.t, reinterpret_cast<const h16x8*>(WangT), D, gz, j, g);
      write_dl<VT>(Trow, g, ga.t);
      __builtin_amdgcn_wave_barrier();
      scatter_rows64_add_masked(T, TS64, p.Gang, a_raw, lane_t, gang_old);
      __builtin_amdgcn_wave_barrier();
    }
    // ---- scatter: 4 first-bond rows, 4 second-bond rows, the centre ----
    {
      Cols64 cc[2];
      to_columns(gzc, T, Trow, g, lane_t, cc[0]);
      to_columns(gzg, T, Trow, g, lane_t, cc[1]);
      blk_scatter_any<true, true>(ps, cc[0], vmask, ge, b1, b2, p.GR, p.GR + 2 * D, 4 * D, lane_t);
      blk_scatter_any<true, true>(ps, cc[1], vmask, ge, b1, b2, p.GR + D, p.GR + 3 * D, 4 * D, lane_t);
      if (vmask) {
        const int ck = __builtin_amdgcn_readlane(c, __builtin_ctz(vmask));
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int rr = 0; rr < TILE_ROWS; ++rr) { s0 += cc[0].v[rr]; s1 += cc[1].v[rr]; }
        atomicAdd(grow<float>(p.GS, (unsigned)ck, 2 * D, lane_t), s0);
        atomicAdd(grow<float>(p.GS, (unsigned)ck, 2 * D, D + lane_t), s1);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace chg
