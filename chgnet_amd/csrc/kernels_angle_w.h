// kernels_angle_w.h -- adjoints of BondConv / AngleUpdate with the second-bond scatter aggregated per ATOM in LDS.
//
// Why (profiles/r03_experiments.md): the per-row fp32 atomics of the angle adjoints (the second bond of every angle:
// 939 B per angle for BondConv) run at ~1.2-1.3 TB/s chip-wide -- one lane per clock and L2 channel, whatever their scope
// and however the target rows are shared between the XCDs (round 6: tools/lab/atomic_scope_lab.hip; round 3 read it as
// "executed at the memory side") -- and that, not the matrix pipe or the gathers, is what bounded both kernels
// (no atomics: -28 % / -13 %; no atomics AND a 4x faster matrix pipe: -50 % / -55 %).
//
// Order.  The reference emits angles sorted by their first bond (graph.py:283-327).  All n (n - 1) angles around one atom
// only touch that atom's n short bonds: they form the off-diagonal of an n x n matrix (first bond, second bond).  k_win_*
// build, once per batch topology and entirely on the device, the CENTRE-ATOM-major order
//   q_a[row]            the angle (index into the reference-ordered per-angle arrays) processed as row `row`
//   q_ctr, q_b1c, q_b2c its centre atom and compact bond indices
//   q_ab1, q_ab2        the (atom, bond) pair index  boff[centre] + rank of the bond among the centre's short bonds
//   abbond[ab]          the compact bond index behind a pair;  aoff[c] the first row of atom c
// Inside one atom: by directed-edge index of the first bond, then the reference's order -- deterministic.
// Graphs without this structure (hand-made bond graphs: incomplete angle sets, groups that are not contiguous) and batches too
// small to give every wave a few atoms clear flag[0]; both adjoint kernels are always launched and the one that does not
// apply returns at once, so there is no host round trip and a captured hipGraph stays valid.
//
// Scatter.  One WAVE owns whole atoms (contiguous ranges, balanced by row count).  While it walks the rows of an atom
//   * the first-bond sums and the centre sum are run sums in registers (rows arrive sorted by first bond): one 256-byte
//     atomic per 64 columns when a run ends,
//   * the second-bond sums -- the 15.8-distinct-rows-per-tile scatter of the row-order kernels -- accumulate in a
//     wave-private LDS array [rank of the bond at this atom][128] by plain read-modify-write (no other wave ever touches it;
//     LDS float ATOMICS were tried first for workgroup-shared windows and are serialised per lane: 5x slower), flushed with one
//     atomic row per bond when the atom is done: n instead of n (n - 1) second-bond rows leave the workgroup.
// No barrier, no flag, no LDS atomic.  Bonds of rank >= NS (atoms with more short bonds than the array has rows) and rows whose
// second bond is not a first bond at the same atom fall back to direct row atomics.
#pragma once

#include "blk_shape.h"
#include "kernels_conv.h"

namespace chg {

constexpr int TS64 = D + PAD;                  // row stride of the 64-wide wave tiles of this kernel
constexpr int TILE64_FLOATS = TILE_ROWS * TS64;
constexpr int WIN_LIST = 32;                   // short bonds per atom the fast path handles
constexpr int WIN_MAX_GRID = 1024;             // workgroups of the per-atom kernels (one per CU; capacity of the schedule's arrays)
constexpr int WIN_MIN_ATOMS_PER_WAVE = 3;      // below this the atom-per-wave order leaves most of the chip idle: plain adjoints

struct WinIndex {             // built by k_win_*; all in the batch arena
  int* flag;                  // [4]  flag[0] = 1: the centre-major order is valid for this batch (the per-atom adjoints run);
                              //      flag[3] = workgroups the atom schedule was built for
  int *wave_head, *next_atom; // [win_grid * WAVES], [N]: the atoms of every wave as a linked list (k_win_schedule)
  int* xatom;                 // [win_grid / 8 + 1] atoms [xatom[i], xatom[i + 1]) are dealt to the 64 waves of group i (k_win_groups)
  int *na, *boff, *aoff;      // [N+1] short bonds per atom, exclusive scans of na and na (na - 1)
  int* toff;                  // [N+1] exclusive scan of the atoms' 16-row tile counts ceil(na (na - 1) / 16) (TEAM kernels: tiles dealt evenly)
  int* toff4;                 // [N+1] exclusive scan of the atoms' blocked-tile counts (blk_shape_of; kernels_angle_blk.h, uploaded graphs)
  int *head, *rank;           // [Ed] first row of the group whose first bond is this directed edge (-1: none); its rank at the centre
  int* list;                  // [N][WIN_LIST] directed edges of the groups of an atom (unordered)
  int *q_a, *q_ctr, *q_b1c, *q_b2c, *q_ab1, *q_ab2;   // [A]
  int* abbond;                // [2 Eb]
};

// ---- index construction ---------------------------------------------------------------------------------
static __global__ void k_win_init(WinIndex w, int grid) { w.flag[0] = 1; w.flag[3] = grid; }

static __global__ void k_win_heads(const int* __restrict__ a_d1, const int* __restrict__ a_ctr, int A, int Ed, int N, WinIndex w) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  const int d1 = a_d1[a];
  if (d1 < 0 || d1 >= Ed) { w.flag[0] = 0; return; }
  if (a > 0 && a_d1[a - 1] == d1) return;
  const int c = a_ctr[a];
  if (atomicCAS(w.head + d1, -1, a) != -1 || c < 0 || c >= N) { w.flag[0] = 0; return; }   // a second group with the same first bond
  const int pos = atomicAdd(w.na + c, 1);
  if (pos < WIN_LIST) w.list[(size_t)c * WIN_LIST + pos] = d1;
  else w.flag[0] = 0;
}

static __global__ void k_win_counts(int N, WinIndex w, int* __restrict__ nang) {   // nang[c] = na (na - 1): rows of atom c
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > N) return;
  const int n = c < N ? w.na[c] : 0;
  nang[c] = n * (n - 1);
}

static __global__ void k_win_ranks(const int* __restrict__ a_d1, const int* __restrict__ a_ctr, const int* __restrict__ a_b1c, int A, int N,
                            WinIndex w) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A || w.flag[0] == 0) return;
  const int d1 = a_d1[a];
  if (w.head[d1] != a) return;                      // group heads only
  const int c = a_ctr[a], n = w.na[c];
  int r = 0;
  for (int q = 0; q < n; ++q) r += w.list[(size_t)c * WIN_LIST + q] < d1;
  w.rank[d1] = r;
  w.abbond[w.boff[c] + r] = a_b1c[a];
  // the group must be exactly the n - 1 rows a .. a + n - 2
  const int last = a + n - 2;
  if (n < 2 || last >= A || a_d1[last] != d1 || (last + 1 < A && a_d1[last + 1] == d1)) w.flag[0] = 0;
}

static __global__ void k_win_rows(const int* __restrict__ a_ctr, const int* __restrict__ a_b1c, const int* __restrict__ a_b2c,
                           const int* __restrict__ a_d1, const int* __restrict__ a_d2, int A, int N, int Ed, WinIndex w) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a == 0 && w.aoff[N] != A) w.flag[0] = 0;     // every atom's angle set must be the complete n (n - 1) block
  if (a >= A || w.flag[0] == 0) return;
  const int d1 = a_d1[a], d2 = a_d2[a], c = a_ctr[a];
  const int h = w.head[d1];
  if (h < 0 || a_ctr[h] != c) { w.flag[0] = 0; return; }
  const int n = w.na[c], q = a - h, r = w.rank[d1];
  const long row = (long)w.aoff[c] + (long)r * (n - 1) + q;
  if (q < 0 || q >= n - 1 || row < 0 || row >= A) { w.flag[0] = 0; return; }
  int r2 = -1;
  if (d2 >= 0 && d2 < Ed) {
    const int h2 = w.head[d2];
    if (h2 >= 0 && a_ctr[h2] == c) r2 = w.rank[d2];
  }
  w.q_a[row] = a; w.q_ctr[row] = c; w.q_b1c[row] = a_b1c[a]; w.q_b2c[row] = a_b2c[a];
  w.q_ab1[row] = w.boff[c] + r;
  w.q_ab2[row] = r2 >= 0 ? w.boff[c] + r2 : -1;
}

// ---- small batches (team mode, below): the index without the schedule, in five launches -----------------------------------------
// every array k_win_heads accumulates into, and the flags, in one launch (instead of four memsets and k_win_init)
static __global__ void k_win_clear(WinIndex w, int N, int Ed, int grid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (int k = i; k <= N; k += stride) w.na[k] = 0;
  for (int k = i; k < Ed; k += stride) { w.head[k] = -1; w.rank[k] = -1; }
  if (i == 0) { w.flag[0] = 1; w.flag[1] = 0; w.flag[2] = 0; w.flag[3] = grid; }
}

// boff = exclusive scan of na, aoff = exclusive scan of na (na - 1), toff = exclusive scan of the tile counts, all over the N + 1 entries, by ONE workgroup (N + 1 <= 8192:
// a few thousand atoms is all team mode is for): each thread sums a contiguous run, the 1024 run sums are scanned through LDS
static __global__ __launch_bounds__(1024) void k_win_scan2(int N, WinIndex w) {
  __shared__ int tot_b[16], tot_a[16], tot_t[16], tot_4[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = N + 1, per = (n + 1023) / 1024;
  const int b = min(tid * per, n), e = min(b + per, n);
  int sb = 0, sa = 0, st = 0, s4 = 0;
  for (int k = b; k < e; ++k) {
    const int v = k < N ? w.na[k] : 0;
    int ps, qs;
    sb += v; sa += v * (v - 1); st += (v * (v - 1) + TILE_ROWS - 1) / TILE_ROWS; s4 += blk_shape_of(v, ps, qs);
  }
  int ib = sb, ia = sa, it = st, i4 = s4;            // inclusive scans over the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int ub = __shfl_up(ib, off), ua = __shfl_up(ia, off), ut = __shfl_up(it, off), u4 = __shfl_up(i4, off);
    if (lane >= off) { ib += ub; ia += ua; it += ut; i4 += u4; }
  }
  if (lane == 63) { tot_b[wave] = ib; tot_a[wave] = ia; tot_t[wave] = it; tot_4[wave] = i4; }
  __syncthreads();
  int rb = ib - sb, ra = ia - sa, rt = it - st, r4 = i4 - s4;
  for (int q = 0; q < wave; ++q) { rb += tot_b[q]; ra += tot_a[q]; rt += tot_t[q]; r4 += tot_4[q]; }
  for (int k = b; k < e; ++k) {
    const int v = k < N ? w.na[k] : 0;
    int ps, qs;
    w.boff[k] = rb; w.aoff[k] = ra; w.toff[k] = rt; w.toff4[k] = r4;
    rb += v; ra += v * (v - 1); rt += (v * (v - 1) + TILE_ROWS - 1) / TILE_ROWS; r4 += blk_shape_of(v, ps, qs);
  }
}

// ---- which wave works on which atom: a static schedule with the locality of a dynamic one ----------------------------------
// With one contiguous atom range per wave (round 3) the 256 waves of an XCD sat in ~128 different structures at any moment and their
// tables (R: 1 KB per bond node) missed the 4 MiB L2: hit rates of 24-37 %, fabric traffic 1.5-2.3x the compulsory bytes
// (profiles/r03_l2_counters.csv).  Now the rows are cut into grid / 8 equal ranges and GROUPS of 64 waves -- eight workgroups of one
// XCD (blockIdx & 7 = XCD is the observed dispatch: speed only), the groups of an XCD on neighbouring ranges -- walk one range
// TOGETHER: its atoms are dealt, in order, always to the wave of the group with the least work so far (greedy list scheduling on the
// tile count of an atom: what a shared work queue would do, but decided once per batch topology, deterministic, and without a
// returning atomic per atom).  Waves advance at the same pace, so at any moment an XCD works inside a handful of neighbouring
// structures, and the loads of a group differ by at most one atom at the end.  The result is a linked list per wave:
// wave_head[blockIdx * WAVES + wave] -> next_atom[c] -> ... -> -1, so a wave knows its next atom one atom ahead (index prefetch).
constexpr int WIN_GROUP_WAVES = 64;

static __global__ void k_win_groups(int N, int A, int ngroups, WinIndex w) {   // xatom[i] = first atom c with aoff[c + 1] > A i / ngroups
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > ngroups) return;
  if (i == ngroups) { w.xatom[i] = N; return; }
  const long target = (long)A * i / ngroups;
  int lo = 0, hi = N;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (w.aoff[mid + 1] > target) hi = mid; else lo = mid + 1;
  }
  w.xatom[i] = lo;
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {   // minimum over the 64 lanes, in every lane (cf. quad_sum / wave_sum)
#define CHG_DPP_MIN(ctrl) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, (ctrl), 0xF, 0xF, false))
  CHG_DPP_MIN(0xB1);    // quad_perm [1,0,3,2]
  CHG_DPP_MIN(0x4E);    // quad_perm [2,3,0,1]
  CHG_DPP_MIN(0x141);   // row_half_mirror
  CHG_DPP_MIN(0x140);   // row_mirror
#undef CHG_DPP_MIN
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  const unsigned s = min((unsigned)r[0], (unsigned)r[1]);
  const auto q = __builtin_amdgcn_permlane32_swap(s, s, false, false);
  return min((unsigned)q[0], (unsigned)q[1]);
}

// one wave per group; lane = slot of the group = ((blockIdx >> 3) & 7) * WAVES + wave of the kernels that consume the lists
static __global__ __launch_bounds__(64) void k_win_schedule(int grid, WinIndex w) {
  const int gi = blockIdx.x, lane = threadIdx.x;
  const int spx = grid >> 6;                   // groups per XCD
  const int x = gi / spx, sub = gi - x * spx;
  const int head_slot = ((((sub << 3) + (lane / WAVES)) << 3) + x) * WAVES + (lane % WAVES);
  if (w.flag[0] != 1) { w.wave_head[head_slot] = -1; return; }
  const int c0 = w.xatom[gi], c1 = w.xatom[gi + 1];
  unsigned load = 0;
  int head = -1, tail = -1;
  for (int cb = c0; cb < c1; cb += 64) {
    const int nn = w.na[min(cb + lane, c1 - 1)];
    const int m = min(64, c1 - cb);
    for (int i = 0; i < m; ++i) {
      const int n = __builtin_amdgcn_readlane(nn, i);
      if (n < 2) continue;                    // no angles around this atom
      const unsigned cost = (unsigned)(n * (n - 1) + TILE_ROWS - 1) / TILE_ROWS + 1;   // its tiles + the per-atom prologue
      const unsigned best = wave_min_u32((load << 6) | (unsigned)lane);              // least loaded wave, lowest slot among equals
      if ((int)(best & 63u) == lane) {
        const int c = cb + i;
        load += cost;
        w.next_atom[c] = -1;
        if (tail < 0) head = c; else w.next_atom[tail] = c;
        tail = c;
      }
    }
  }
  w.wave_head[head_slot] = head;
}

// ---- the adjoint kernel ---------------------------------------------------------------------------------
struct AngleWArgs {
  AngleArgs a;                // tables, weights, gradient buffers as for k_angle
  WinIndex w;                 // incl. the per-wave atom lists (k_win_schedule); the grid is the one the schedule was built for
  int n_atoms;                // TEAM kernels: atoms of the batch
};

// Private second-bond rows per wave: dE/dR_j (128 wide).  AngleUpdate: the angle block as two split images (64 KiB), 14 rows.
// BondConv: ALL contractions in split precision from ONE row-major image per matrix (mfma_split.h: 72 KiB for the angle block and
// the hidden layer, both directions -- the two-image form would need 128 KiB), 13 rows; the bond-weight gradients of the second
// bond leave as one atomic row per angle (no LDS left for private rows of theirs).
template <bool HIDDEN> constexpr int win_ns() { return HIDDEN ? 13 : 14; }
template <bool HIDDEN> constexpr int win_pst() { return 2 * D; }   // stride of a private row (floats)
constexpr size_t WIN_RM_ANG = rm_image_bytes(2 * D, D), WIN_RM_W2 = rm_image_bytes(D, D);

template <bool HIDDEN>
constexpr size_t angle_w_lds() {
  const size_t weights = HIDDEN ? WIN_RM_ANG + 2 * WIN_RM_W2 : 16 * (size_t)(2 * IMG128);
  return weights + sizeof(float) * (VEC_SLOTS * D + WAVES * TILE64_FLOATS + WAVES * win_ns<HIDDEN>() * win_pst<HIDDEN>());
}

// sum of three 64-wide table row halves into a 64-wide tile: 16 lanes per row, 4 rows per load instruction
struct Gather64 { f32x4 a[TILE_ROWS / 4], b[TILE_ROWS / 4], c[TILE_ROWS / 4]; };
__device__ __forceinline__ void gather64_issue(Gather64& gr, const float* __restrict__ t0, int i0, int ld0, const float* __restrict__ t1, int i1,
                                               int ld1, const float* __restrict__ t2, int i2, int ld2, int lane) {
  const int sub = lane >> 4, t = lane & 15;
  int r0[TILE_ROWS / 4], r1[TILE_ROWS / 4], r2[TILE_ROWS / 4];
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    const int rr = 4 * it + sub;
    r0[it] = __shfl(i0, rr); r1[it] = __shfl(i1, rr); r2[it] = __shfl(i2, rr);
  }
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it) {
    gr.a[it] = *grow<f32x4>(t0, (unsigned)r0[it], ld0, 4 * t);
    gr.b[it] = *grow<f32x4>(t1, (unsigned)r1[it], ld1, 4 * t);
    gr.c[it] = *grow<f32x4>(t2, (unsigned)r2[it], ld2, 4 * t);
  }
}
__device__ __forceinline__ void gather64_commit(const Gather64& gr, float* tile, int lane) {
  const int sub = lane >> 4, t = lane & 15;
#pragma unroll
  for (int it = 0; it < TILE_ROWS / 4; ++it)
    *reinterpret_cast<f32x4*>(tile + (4 * it + sub) * TS64 + 4 * t) = (gr.a[it] + gr.b[it]) + gr.c[it];
}

// A 64-wide tile held column-wise: lane = column, v[rr] = row rr.  Transposed through the wave's LDS tile; all 16 row
// reads are issued together (one LDS round trip) and every scatter below then works on registers -- reading the rows one by
// one inside data-dependent branches serialised the LDS latency 16 times per pass (measured: half of the kernel).
struct Cols64 { float v[TILE_ROWS]; };
__device__ __forceinline__ void to_columns(const V64& x, float* T, float* Trow, int g, int lane, Cols64& c) {
  write_dl<VT>(Trow, g, x.t);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int rr = 0; rr < TILE_ROWS; ++rr) c.v[rr] = T[rr * TS64 + lane];
  __builtin_amdgcn_wave_barrier();
}
// Run sum for a key that is sorted along the rows, carried across tiles (`sum` / `cur` live in the caller); a finished run
// leaves with one 256-byte atomic.
__device__ __forceinline__ void run_sum64(const Cols64& c, int nvalid, int key, float& sum, int& cur, float* __restrict__ dst, int ld, int lane) {
#pragma unroll
  for (int rr = 0; rr < TILE_ROWS; ++rr) {
    if (rr < nvalid) {
      const int k = __builtin_amdgcn_readlane(key, rr);
      if (k != cur) {
        if (cur >= 0) atomicAdd(grow<float>(dst, (unsigned)cur, ld, lane), sum);
        sum = 0.f;
        cur = k;
      }
      sum += c.v[rr];
    }
  }
}
__device__ __forceinline__ void run_flush64(float& sum, int& cur, float* __restrict__ dst, int ld, int lane) {
  if (cur >= 0) atomicAdd(grow<float>(dst, (unsigned)cur, ld, lane), sum);
  sum = 0.f;
  cur = -1;
}
__device__ __forceinline__ void row_add64(const Cols64& c, int nvalid, float* __restrict__ base, int row, int ld, int lane) {
#pragma unroll
  for (int rr = 0; rr < TILE_ROWS; ++rr)
    if (rr < nvalid) atomicAdd(grow<float>(base, (unsigned)__builtin_amdgcn_readlane(row, rr), ld, lane), c.v[rr]);
}
// Second-bond rows (NB column blocks of 64) into the wave-private LDS rows: plain read-modify-write, two rows per step
// (neighbouring rows of the centre-major order never share a second bond; rows further apart may, and the LDS executes a
// wave's accesses in order).  Rows past the end, and rows without a private row (slot < 0), go to `dump` (the idle wave tile) --
// the latter are then sent as direct atomics by the caller's slow path.
template <int NB>
__device__ __forceinline__ void private_add(const Cols64 (&c)[NB], int nvalid, int slot, float* pacc, float* dump, int lane) {
#pragma unroll
  for (int rr = 0; rr < TILE_ROWS; rr += 2) {
    const int sa = __builtin_amdgcn_readlane(slot, rr), sb = __builtin_amdgcn_readlane(slot, rr + 1);
    float* pa = (rr < nvalid && sa >= 0) ? pacc + sa * (NB * D) : dump;
    float* pb = (rr + 1 < nvalid && sb >= 0) ? pacc + sb * (NB * D) : dump + NB * D;
    float a[NB], b[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) { a[q] = pa[q * D + lane]; b[q] = pb[q * D + lane]; }
#pragma unroll
    for (int q = 0; q < NB; ++q) { pa[q * D + lane] = a[q] + c[q].v[rr]; pb[q * D + lane] = b[q] + c[q].v[rr + 1]; }
  }
}

// TEAM (round 6; MD-size batches): the 16-row tiles of all atoms, in centre-major order, are cut into EQUAL ranges, one per workgroup
// (toff: exclusive scan of the atoms' tile counts), and a workgroup walks the atoms its range touches: the eight waves deal that
// atom's tiles among themselves (tile k to wave k mod 8), each accumulates the second-bond sums of ITS tiles in its own private rows as
// before, and when the atom (or the workgroup's part of it) is done the workgroup adds its eight copies of every row together in LDS
// (two workgroup barriers per atom) and sends n rows out -- not n (n - 1) as the row-order adjoints do, which at this size are bound
// by exactly those memory-side atomics (939 B per angle at ~1.2 TB/s: 53 of the 72 us of a 67,536-angle launch).  A 256-atom cell
// then occupies all 2,048 waves, and because the cut is by TILES an atom with 24 short bonds (35 tiles) is shared by two or three
// workgroups instead of holding one for five rounds: the first version gave every atom to one team of waves and took 75 us per launch
// on a thermalised MD cell against 50 us on the perfect crystal (same angle count) -- the launch lasted as long as its largest atom.
// Run sums of a wave's non-adjacent tiles simply end at the tile's last row.
// ZS (BondConv, large batches): the forward kernel kept z (AngleArgs::zsave) -- the tile starts from ONE 512-byte row per angle instead of
// four gathered rows and the W_ang contraction.
template <bool HIDDEN, bool TEAM = false, bool ZS = false>
__global__ __launch_bounds__(BLOCK) CHG_TWO_WAVES void k_angle_bwd_w(AngleWArgs pw) {
  const AngleArgs& p = pw.a;
  const WinIndex& w = pw.w;
  if (w.flag[0] != 1) return;                   // this batch runs the plain adjoint (k_angle<.., true>)
  PH_START
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NS = win_ns<HIDDEN>(), PST = win_pst<HIDDEN>();
  // AngleUpdate: split images of W_ang and W_ang^T.  BondConv: row-major images of W_ang, W2c, W2g (each serves both directions).
  constexpr int MODE = HIDDEN ? 2 : 1;
  float* Wang = smem;
  float* WangT = HIDDEN ? Wang : Wang + 4 * IMG128;
  float* W2c = HIDDEN ? Wang + WIN_RM_ANG / 4 : WangT + 4 * IMG128;
  float* W2g = W2c + (HIDDEN ? WIN_RM_W2 / 4 : 0);
  float* vecs = W2g + (HIDDEN ? WIN_RM_W2 / 4 : 0);
  float* tiles = vecs + VEC_SLOTS * D;
  float* paccs = tiles + WAVES * TILE64_FLOATS;   // [WAVES][NS][PST]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the weight block is the row-order adjoint's (AngleLds<HIDDEN, true>: same images at the same offsets), prebuilt once per weight
  // upload (k_angle_image): a 16-byte-per-lane copy.  Building it here from the fp32 weights -- strided reads + the hi / lo split in
  // every workgroup of every launch -- is nothing next to a large batch's hundreds of tiles per wave, but was a quarter of a TEAM
  // launch's two tiles per wave (~30k of 118k clocks per wave on a 256-atom cell).
  static_assert((size_t)AngleLds<HIDDEN, true>::tiles * sizeof(float) == (HIDDEN ? WIN_RM_ANG + 2 * WIN_RM_W2 : 16 * (size_t)(2 * IMG128)) + sizeof(float) * VEC_SLOTS * D,
                "the per-atom adjoints share the weight image of k_angle<HIDDEN, true>");
  if (p.image) {
    stage_image<AngleLds<HIDDEN, true>::tiles / 4, BLOCK>(smem, p.image, tid);
  } else if (HIDDEN) {
    stage_rm(reinterpret_cast<_Float16*>(Wang), p.w_ang, 2 * D, D, tid, BLOCK);
    stage_rm(reinterpret_cast<_Float16*>(W2c), p.gw.w2c, D, D, tid, BLOCK);
    stage_rm(reinterpret_cast<_Float16*>(W2g), p.gw.w2g, D, D, tid, BLOCK);
    stage_gated_vecs(vecs, p.gw, HIDDEN, tid);
  } else {
    stage_split<false>(reinterpret_cast<h16x8*>(Wang), p.w_ang, 2 * D, D, tid, BLOCK);
    stage_split<true>(reinterpret_cast<h16x8*>(WangT), p.w_ang, 2 * D, D, tid, BLOCK);
    stage_gated_vecs(vecs, p.gw, HIDDEN, tid);
  }
  for (int q = tid; q < WAVES * NS * PST; q += BLOCK) paccs[q] = 0.f;
  __syncthreads();
  float* T = tiles + wave * TILE64_FLOATS;
  float* Trow = T + j * TS64;
  float* pacc = paccs + wave * NS * PST;
  PH_DECL
  // TEAM: this workgroup's tiles [t_lo, t_hi) of the centre-major tile sequence; c_team: the atom that holds tile t_lo
  int t_lo = 0, t_hi = 0, c_team = 0;
  if (TEAM) {
    const int total = __builtin_amdgcn_readfirstlane(w.toff[pw.n_atoms]);
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    t_lo = min(total, (int)blockIdx.x * per);
    t_hi = min(total, t_lo + per);
    int lo = 0, hi = pw.n_atoms;              // the first atom whose tiles end after t_lo
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (__builtin_amdgcn_readfirstlane(w.toff[mid + 1]) > t_lo) hi = mid; else lo = mid + 1;
    }
    c_team = lo;
  }
  const int wt = TEAM ? wave : 0;
  constexpr int tstep = TEAM ? WAVES * TILE_ROWS : TILE_ROWS;
  int c_next = TEAM ? 0 : w.wave_head[blockIdx.x * WAVES + wave];     // this wave's atoms (k_win_schedule): a list, one atom ahead
  for (;;) {
    int c, n = 0, ab0 = 0, k_lo = 0, k_hi = 0x7fffffff;
    bool active = true;
    if (TEAM) {
      if (t_lo >= t_hi) break;
      c = c_team++;
      const int a0 = __builtin_amdgcn_readfirstlane(w.toff[c]), a1 = __builtin_amdgcn_readfirstlane(w.toff[c + 1]);
      if (a1 <= t_lo) continue;               // an atom without angles
      n = __builtin_amdgcn_readfirstlane(w.na[c]);
      k_lo = t_lo - a0; k_hi = min(t_hi, a1) - a0;     // this workgroup's tiles of atom c
      t_lo = min(t_hi, a1);
    } else {
      c = __builtin_amdgcn_readfirstlane(c_next);
      if (c < 0) break;
      c_next = w.next_atom[c];
      n = __builtin_amdgcn_readfirstlane(w.na[c]);
    }
    if (active) {
    const int r_begin = __builtin_amdgcn_readfirstlane(w.aoff[c]), r_end = r_begin + n * (n - 1);
    const int seg_end = TEAM ? min(r_end, r_begin + k_hi * TILE_ROWS) : r_end;    // (TEAM: the workgroup's part of the atom)
    const int row_first = r_begin + (TEAM ? (k_lo + wt) * TILE_ROWS : 0);
    ab0 = __builtin_amdgcn_readfirstlane(w.boff[c]);
    // run sums carried over the tiles of this atom (lane = column): first bond (core | gate halves), centre, bond weights
    float ri0 = 0.f, ri1 = 0.f, rs0 = 0.f, rs1 = 0.f, rg = 0.f;
    int cur0 = -1, cur1 = -1, curg = -1;
    int a_n, b1_n, b2_n, ab2_n;
    {
      const int row = min(row_first + j, r_end - 1);
      a_n = w.q_a[row]; b1_n = w.q_b1c[row]; b2_n = w.q_b2c[row]; ab2_n = w.q_ab2[row];
    }
    for (int row0 = row_first; row0 < seg_end; row0 += tstep) {
      // lane index made opaque once per tile: everything derived from it (64-bit row pointers base + 4 lane for every buffer, row
      // constants) is recomputed where it is used instead of living -- and being spilled -- across the whole kernel; a spilled
      // value reloaded between stores or atomics costs their full round trip (the reload's wait is in order behind them)
      // (BondConv only: the AngleUpdate adjoint does not spill, and recomputing cost it 3 %)
      int lane_t = lane;
      if (HIDDEN) asm volatile("" : "+v"(lane_t));
      const int nvalid = min(TILE_ROWS, r_end - row0);
      const int a = a_n, b1 = b1_n, b2 = b2_n, ab2 = ab2_n;
      if (row0 + tstep < seg_end) {
        const int row = min(row0 + tstep + j, r_end - 1);
        a_n = w.q_a[row]; b1_n = w.q_b1c[row]; b2_n = w.q_b2c[row]; ab2_n = w.q_ab2[row];
      }
      int s2 = ab2 - ab0;                         // rank of the second bond at this atom = private row
      if (ab2 < 0 || s2 < 0 || s2 >= NS) s2 = -1;
      // ---- gathers: angle rows, then the two halves of the table sum.  (Requesting tile t+1's rows ahead -- before tile t's scatter
      // phase -- was tried: 112 loop-carried registers for the table rows cost 225 spills (2x slower), 16 for the angle rows alone 51
      // spills (+14 %): this kernel has no registers left to pipeline with.) ----
      f32x4 z[2 * VT];
      Rows64 gy_rows;
      V64 w1, w2, gu;        // BondConv: bond weights and the aggregate's adjoint, requested ahead of the forward recomputation
      if (ZS) {
        // (requesting the NEXT tile's rows a tile ahead -- 32 loop-carried registers -- spilled 11-13 and cost 2.06 -> 2.24 ms, like every
        // other attempt to pipeline this kernel)
        read_dl_g_nt<2 * VT>(p.zsave, (unsigned)a, 2 * D, g, z);
        PH(0)
        if (!HIDDEN) rows64_issue(gy_rows, p.Gang, a, lane_t);
        if (HIDDEN) {
          read_dl_g<VT>(p.wbgc, (unsigned)b1, D, g, w1.t);
          read_dl_g<VT>(p.wbgc, (unsigned)b2, D, g, w2.t);
          read_dl_g<VT>(p.Gagg, (unsigned)b1, D, g, gu.t);
        }
      } else {
      Gather64 gc, gg;
      gather64_issue(gc, p.R, b1, 4 * D, p.R + 2 * D, b2, 4 * D, p.S, c, 2 * D, lane_t);
      gather64_issue(gg, p.R + D, b1, 4 * D, p.R + 3 * D, b2, 4 * D, p.S + D, c, 2 * D, lane_t);
      gather_rows64(T, TS64, p.ang, a, lane_t);
      __builtin_amdgcn_wave_barrier();
      V64 x;
      read_dl<VT>(Trow, g, x.t);
      __builtin_amdgcn_wave_barrier();
      gather64_commit(gc, T, lane_t);
      __builtin_amdgcn_wave_barrier();
      read_dl<VT>(Trow, g, *reinterpret_cast<f32x4(*)[VT]>(&z[0]));
      __builtin_amdgcn_wave_barrier();
      gather64_commit(gg, T, lane_t);
      __builtin_amdgcn_wave_barrier();
      read_dl<VT>(Trow, g, *reinterpret_cast<f32x4(*)[VT]>(&z[VT]));
      __builtin_amdgcn_wave_barrier();
      PH(0)   // indices + gathers
      if (!HIDDEN) rows64_issue(gy_rows, p.Gang, a, lane_t);
      if (HIDDEN) {
        read_dl_g<VT>(p.wbgc, (unsigned)b1, D, g, w1.t);
        read_dl_g<VT>(p.wbgc, (unsigned)b2, D, g, w2.t);
        read_dl_g<VT>(p.Gagg, (unsigned)b1, D, g, gu.t);
      }
      if (HIDDEN) gemm_rm<VT, 2 * VT, false, false>(z, reinterpret_cast<const _Float16*>(Wang), 2 * D, D, x.t, j, g, lane_t);
      else gemm_split<VT, 2 * VT, false>(z, reinterpret_cast<const h16x8*>(Wang), 2 * D, x.t, j, g);
      }
      V64 zc{{z[0], z[1], z[2], z[3]}}, zg{{z[4], z[5], z[6], z[7]}};
      GatedState s;
      V64 y;
      constexpr bool SLIM = HIDDEN;
      gated_forward<HIDDEN, SLIM, false, MODE>(zc, zg, W2c, W2g, vecs, j, g, s, y);
      PH(7)   // forward recomputation
      V64 gy;
      if (HIDDEN) {
        V64 g1, g2;
        CHG_EV(ft) {
          const f32x4 gyu = gu.t[ft] * y.t[ft];
          g1.t[ft] = gyu * w2.t[ft];      // dE/d wbgc[b1]
          g2.t[ft] = gyu * w1.t[ft];      // dE/d wbgc[b2]
          gy.t[ft] = gu.t[ft] * w1.t[ft] * w2.t[ft];
        }
        // the bond-weight gradients leave now (two vectors less to carry through the adjoint of the gated MLP):
        // first bond as a run sum, second bond one atomic row per angle
        {
          Cols64 c1, c2[1];
          to_columns(g1, T, Trow, g, lane_t, c1);
          to_columns(g2, T, Trow, g, lane_t, c2[0]);
          run_sum64(c1, nvalid, b1, rg, curg, p.Gwbgc, D, lane_t);
#if defined(CHG_EXPERIMENTS) && defined(CHG_EXP_NO_G2_ATOMICS)
          // TIMING-ONLY (wrong results; profiles/r06_experiments.md): the second bond's weight gradient is dropped -- the upper bound of
          // what ANY scheme that takes these per-angle atomic rows out of the kernel (private rows, a deferred reduction) can gain
          asm volatile("" :: "v"(c2[0].v[0]));
#else
          row_add64(c2[0], nvalid, p.Gwbgc, b2, D, lane_t);
#endif
        }
        PH(6)   // bond-weight gradient scatter
      } else {
        rows64_commit(gy_rows, T, TS64, lane_t);
        __builtin_amdgcn_wave_barrier();
        read_dl<VT>(Trow, g, gy.t);
        __builtin_amdgcn_wave_barrier();
      }
      V64 gzc, gzg;
      gated_backward<HIDDEN, SLIM, false, MODE>(gy, zc, zg, W2c, W2g, vecs, j, g, s, gzc, gzg);
      PH(1)   // contractions + gated MLP, forward and adjoint
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
        PH(5)   // W_ang^T contraction
        write_dl<VT>(Trow, g, ga.t);
        __builtin_amdgcn_wave_barrier();
        scatter_rows64_add<HIDDEN>(T, TS64, p.Gang, a, nvalid, lane_t, gang_old);
        __builtin_amdgcn_wave_barrier();
      }
      PH(2)   // W_ang^T contraction + Gang update
      // ---- scatter: first bond and centre as carried run sums, second bond into the private rows ----
      {
        Cols64 cc[2];
        to_columns(gzc, T, Trow, g, lane_t, cc[0]);
        to_columns(gzg, T, Trow, g, lane_t, cc[1]);
#pragma unroll
        for (int rr = 0; rr < TILE_ROWS; ++rr)
          if (rr < nvalid) { rs0 += cc[0].v[rr]; rs1 += cc[1].v[rr]; }
        run_sum64(cc[0], nvalid, b1, ri0, cur0, p.GR, 4 * D, lane_t);
        run_sum64(cc[1], nvalid, b1, ri1, cur1, p.GR + D, 4 * D, lane_t);
        private_add<2>(cc, nvalid, s2, pacc, T, lane_t);
        if (__builtin_amdgcn_ballot_w64(j < nvalid && lane_t < TILE_ROWS && s2 < 0)) {   // rare: no private row for this second bond
#pragma unroll
          for (int rr = 0; rr < TILE_ROWS; ++rr)
            if (rr < nvalid && __builtin_amdgcn_readlane(s2, rr) < 0) {
              float* d = p.GR + (size_t)__builtin_amdgcn_readlane(b2, rr) * 4 * D + 2 * D + lane_t;
              atomicAdd(d, cc[0].v[rr]);
              atomicAdd(d + D, cc[1].v[rr]);
            }
        }
        __builtin_amdgcn_wave_barrier();
      }
      PH(3)   // scatter
    }
    // ---- the atom is done: runs, centre sum, private second-bond rows ----
    int lane_f = lane;
    if (HIDDEN) asm volatile("" : "+v"(lane_f));   // as lane_t above
    run_flush64(ri0, cur0, p.GR, 4 * D, lane_f);
    run_flush64(ri1, cur1, p.GR + D, 4 * D, lane_f);
    if (HIDDEN) run_flush64(rg, curg, p.Gwbgc, D, lane_f);
    atomicAdd(grow<float>(p.GS, (unsigned)c, 2 * D, lane_f), rs0);
    atomicAdd(grow<float>(p.GS, (unsigned)c, 2 * D, D + lane_f), rs1);
    if (!TEAM) {
      const int nrows = min(n, NS);
      const int bond_of = w.abbond[ab0 + min(lane_f, nrows - 1)];
      for (int sl = 0; sl < nrows; ++sl) {
        const int bond = __builtin_amdgcn_readlane(bond_of, sl);
        float* src = pacc + sl * PST;
        const float v0 = src[lane_f], v1 = src[D + lane_f];
        src[lane_f] = 0.f; src[D + lane_f] = 0.f;
        atomicAdd(grow<float>(p.GR, (unsigned)bond, 4 * D, 2 * D + lane_f), v0);
        atomicAdd(grow<float>(p.GR, (unsigned)bond, 4 * D, 3 * D + lane_f), v1);
      }
    }
    PH(4)   // per-atom flush
    }   // active
    if (TEAM) {
      // the workgroup's eight copies of the private rows, summed and sent out: row sl by wave sl mod 8
      __syncthreads();
      if (active) {
        const int nrows = min(n, NS);
        const int bond_of = w.abbond[ab0 + min(lane, nrows - 1)];
        for (int sl = wt; sl < nrows; sl += WAVES) {
          const int bond = __builtin_amdgcn_readlane(bond_of, sl);
          float v0 = 0.f, v1 = 0.f;
#pragma unroll
          for (int k = 0; k < WAVES; ++k) {
            float* src = paccs + (k * NS + sl) * PST;
            v0 += src[lane]; v1 += src[D + lane];
            src[lane] = 0.f; src[D + lane] = 0.f;
          }
          atomicAdd(grow<float>(p.GR, (unsigned)bond, 4 * D, 2 * D + lane), v0);
          atomicAdd(grow<float>(p.GR, (unsigned)bond, 4 * D, 3 * D + lane), v1);
        }
      }
      __syncthreads();
    }
  }
  PH_FLUSH(HIDDEN ? 40 : 50)
}

}  // namespace chg
