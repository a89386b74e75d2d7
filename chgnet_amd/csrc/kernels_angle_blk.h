// kernels_angle_blk.h -- adjoints of BondConv / AngleUpdate for MD-size batches over 4 x 4 BLOCKED angle tiles.
//
// Why (profiles/r06_experiments.md section 12): at a few hundred atoms the row-order adjoints (k_angle<.., true>) are bound by the
// fp32 atomics of their scatter -- the L2 executes them at one lane per clock and channel, ~1.3 TB/s chip-wide whatever the scope or
// the sharing (tools/lab/atomic_scope_lab.hip) -- and nearly all of those are the SECOND bond's rows: a 16-row tile of the reference's
// order is 16 angles with one first bond and 16 different second bonds (939 B of atomics per angle for BondConv).  The per-atom and
// TEAM kernels (kernels_angle_w.h) take them off the chip's atomic units by summing an atom's n (n - 1) rows in LDS, but an atom is a
// coarse unit of work when the whole batch is a few thousand tiles: the launch lasts as long as its largest atom / pays two
// workgroup barriers per atom segment.
//
// Here a tile is a 4 x 4 BLOCK of the atom's (first bond, second bond) matrix: rows 4 i + j hold first bond 4 I + i and second bond
// 4 J + j.  The four rows of a first bond and the four rows of a second bond are summed in registers, so a tile sends 4 + 4 rows per
// scatter target instead of 1-2 + 16 -- 505 B of atomics per angle for BondConv -- and every tile is self-contained: any wave takes any
// tile, no barrier, no private rows, no schedule.  The price is the empty slots (the diagonal, and the ragged edge when n is not a
// multiple of 4: 210 angles of an atom with 15 short bonds occupy 16 tiles instead of 14); they read row 0 of the tables and their
// upstream gradient is set to zero, so they add zeros.
//
// Index: the device graph builder writes slot -> angle while it emits the angles (k_angle_fill, from the ranks it counts anyway) and
// k_multi_copy adds the compact bond indices: nothing is launched for it.  Hand-made / uploaded graphs keep the row-order adjoints.
#pragma once

#include "kernels_angle_w.h"

namespace chg {

struct BlkIndex {
  const int* tiles;                       // [1] number of 16-slot tiles (device quantity: sum of ceil(n / 4)^2 over the atoms)
  const int *a, *b1c, *b2c, *ctr;         // [16 tiles] angle (-1: empty slot), compact bond indices, centre atom
};
struct AngleBlkArgs {
  AngleArgs a;
  BlkIndex x;
};

template <bool HIDDEN>
constexpr size_t angle_blk_lds() {
  return sizeof(float) * ((size_t)AngleLds<HIDDEN, true>::tiles + WAVES * TILE64_FLOATS);
}

// dst rows `key1` (first bond: rows 4 i .. 4 i + 3) and `key2` (second bond: rows j, j + 4, j + 8, j + 12) += the column sums of a
// 64-wide tile held column-wise; vmask: bit r = slot r holds an angle (empty slots hold zeros).  Keys are per-lane values of lanes 0-15.
__device__ __forceinline__ void block_scatter64(const Cols64& c, unsigned vmask, int key1, int key2, float* __restrict__ dst1, float* __restrict__ dst2,
                                                int ld, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned gm = (vmask >> (4 * i)) & 0xFu;
    if (gm) {
      const int k = __builtin_amdgcn_readlane(key1, 4 * i + __builtin_ctz(gm));
      atomicAdd(grow<float>(dst1, (unsigned)k, ld, lane), (c.v[4 * i] + c.v[4 * i + 1]) + (c.v[4 * i + 2] + c.v[4 * i + 3]));
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned cm = vmask & (0x1111u << j);
    if (cm) {
      const int k = __builtin_amdgcn_readlane(key2, __builtin_ctz(cm));
      atomicAdd(grow<float>(dst2, (unsigned)k, ld, lane), (c.v[j] + c.v[j + 4]) + (c.v[j + 8] + c.v[j + 12]));
    }
  }
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

template <bool HIDDEN>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_angle_bwd_blk(AngleBlkArgs pb) {
  const AngleArgs& p = pb.a;
  const BlkIndex& x = pb.x;
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
  int a_n = -1, b1_n = 0, b2_n = 0, c_n = 0;
  if (t0 < ntiles) {
    const size_t sl = (size_t)t0 * TILE_ROWS + j;
    a_n = x.a[sl]; b1_n = x.b1c[sl]; b2_n = x.b2c[sl]; c_n = x.ctr[sl];
  }
  stage_image<AngleLds<HIDDEN, true>::tiles / 4, BLOCK>(smem, p.image, tid);
  __syncthreads();
  float* T = tiles + wave * TILE64_FLOATS;
  float* Trow = T + j * TS64;
  for (int t = t0; t < ntiles; t += nw) {
    int lane_t = lane;
    if (HIDDEN) asm volatile("" : "+v"(lane_t));     // as in k_angle_bwd_w: row pointers formed where they are used
    const int a_raw = a_n, b1 = b1_n, b2 = b2_n, c = c_n;
    if (t + nw < ntiles) {
      const size_t sl = (size_t)(t + nw) * TILE_ROWS + j;
      a_n = x.a[sl]; b1_n = x.b1c[sl]; b2_n = x.b2c[sl]; c_n = x.ctr[sl];
    }
    const bool valid = a_raw >= 0;
    const int a = valid ? a_raw : 0;
    const unsigned vmask = (unsigned)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(valid) & 0xFFFFull));
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
      // (two calls: first-bond sums of g1, second-bond sums of g2 -- the other four sums of each call are skipped through an empty mask)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned gm = (vmask >> (4 * i)) & 0xFu;
        if (gm) {
          const int k = __builtin_amdgcn_readlane(b1, 4 * i + __builtin_ctz(gm));
          atomicAdd(grow<float>(p.Gwbgc, (unsigned)k, D, lane_t), (c1.v[4 * i] + c1.v[4 * i + 1]) + (c1.v[4 * i + 2] + c1.v[4 * i + 3]));
        }
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const unsigned cm = vmask & (0x1111u << jj);
        if (cm) {
          const int k = __builtin_amdgcn_readlane(b2, __builtin_ctz(cm));
          atomicAdd(grow<float>(p.Gwbgc, (unsigned)k, D, lane_t), (c2.v[jj] + c2.v[jj + 4]) + (c2.v[jj + 8] + c2.v[jj + 12]));
        }
      }
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
      block_scatter64(cc[0], vmask, b1, b2, p.GR, p.GR + 2 * D, 4 * D, lane_t);
      block_scatter64(cc[1], vmask, b1, b2, p.GR + D, p.GR + 3 * D, 4 * D, lane_t);
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
