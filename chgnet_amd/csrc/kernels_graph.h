// kernels_graph.h -- crystal-graph construction on the device.
//
// Replaces, for structures that are already on their way to the GPU, the host path
//   Structure.get_neighbor_list        chgnet/graph/converter.py:132-134   (pymatgen, un-vendored)
//   create_graph / Graph.add_edge      fast_converter_libraries/create_graph.c:135-203, graph/graph.py:132-224
//   Graph.adjacency_list               graph/graph.py:226-247
//   Graph.line_graph_adjacency_list    graph/graph.py:249-328
// and the index offsetting of BatchedGraph.from_graphs (model/model.py:856-857, 873-877), writing the
// packed batch arrays (pack.py) directly in HBM.
//
// Contract: bit-for-bit the arrays chgnet_amd/csrc/host_graph.cpp + pack.py produce.  The neighbour
// search therefore repeats the host's float64 arithmetic operation by operation (explicit
// round-to-nearest mul/add, no FMA contraction), lists neighbours centre-major in (neighbour, image)
// lexicographic order, numbers undirected bonds by first appearance and enumerates angles in the
// reference's order (owning bond, end 0 then end 1, the centre's edges in row order).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "blk_shape.h"

namespace chg {

struct NlArgs {
  const double* cart;        // [N,3] float64 cartesian coordinates (computed on the host like host_graph.cpp)
  const double* frac;        // [N,3]
  const double* lattice;     // [B,9]
  const double* reach;       // [B,3]  r / plane spacing per axis
  const int* atom_owner;     // [N]
  const int* atom_off;       // [B+1]
  int n_atoms;
  double r2, tol;
  // count pass
  int* center_cnt;           // [N]
  // fill pass
  const int* center_off;     // [N+1]
  int* e_center;
  int* e_nbr;
  int* e_img;                // [Ed,3] int
  float* e_image;            // [Ed,3] float (engine input)
  double* e_dist;            // [Ed]
  int* e_owner;
  int cap_edges;             // fill pass: capacity of the e_* arrays (speculative sizing: rows beyond it are dropped and flagged)
  int* overflow;             // set to 1 when a capacity was exceeded
  // cell list (structures the host binned; host_graph.cpp neighbor_list_cells): null / -1 = all pairs
  const int* cell_off;       // [B] offset of the structure's bins in bin_start, or -1
  const int* cell_nb;        // [B,3] bins per axis
  const int* cell_reach;     // [B,3] bin offsets to visit per axis
  const int* bin_start;      // per cell structure n_bins + 1 positions into bin_atoms
  const int* bin_atoms;      // [N] atom ids grouped by bin
  const int* a_bin3;         // [N,3] the atom's bin
  const int* a_shift;        // [N,3] floor of the fractional coordinate
  int* cell_flag;            // set to 1 when a centre has more rows than the in-LDS sort holds (the build repeats with all pairs)
};

constexpr int CELL_SORT_MAX = 1024;   // rows per centre the wave sorts in LDS

// A count that is either known on the host (exact, two-pass build) or still on the device (single-pass build with
// speculative capacities: the host reads all counts once, at the end).
struct DevCount {
  int host;
  const int* dev;            // null: use host
  int div;                   // the count is *dev / div (undirected = directed / 2)
  __device__ __forceinline__ int get() const { return dev ? *dev / div : host; }
};

__device__ __forceinline__ double sq_dist(const double* __restrict__ cart, const double* __restrict__ L, int i, int j, int ia, int ib,
                                          int ic) {
  double d2 = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // host: dx = cart[j][k] + ia*a[k] + ib*b[k] + ic*c[k] - cart[i][k]   (left to right, every op rounded)
    double t = __dadd_rn(cart[3 * j + k], __dmul_rn((double)ia, L[k]));
    t = __dadd_rn(t, __dmul_rn((double)ib, L[3 + k]));
    t = __dadd_rn(t, __dmul_rn((double)ic, L[6 + k]));
    const double dx = __dsub_rn(t, cart[3 * i + k]);
    d2 = __dadd_rn(d2, __dmul_rn(dx, dx));
  }
  return d2;
}

// image window of the pair (i, j): host_graph.cpp neighbor_list
__device__ __forceinline__ void image_window(const NlArgs& p, const double* reach, int i, int j, int (&lo)[3], int (&hi)[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double df = __dsub_rn(p.frac[3 * j + k], p.frac[3 * i + k]);
    lo[k] = (int)ceil(__dsub_rn(__dsub_rn(-df, reach[k]), 1e-9));
    hi[k] = (int)floor(__dadd_rn(__dadd_rn(-df, reach[k]), 1e-9));
  }
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int& total) {
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  total = __shfl(incl, 63);
  return incl - v;
}

// Cell-list search for one centre (one wave): candidates come from the bins within reach, every candidate is tested
// with the expression of the all-pairs path (sq_dist), and the rows are brought into (neighbour, image) order by a
// bitonic sort of packed keys in the wave's LDS slice -- so the rows are the all-pairs rows, bit for bit.
__device__ __forceinline__ unsigned long long cell_key(int j_local, int ia, int ib, int ic) {
  return ((unsigned long long)j_local << 42) | ((unsigned long long)(ia + 8192) << 28) | ((unsigned long long)(ib + 8192) << 14) |
         (unsigned long long)(ic + 8192);
}

template <bool FILL>
__device__ void neighbors_cells(const NlArgs& p, int i, int b, int lane, unsigned long long* keys) {
  const int a0 = p.atom_off[b];
  const double* L = p.lattice + 9 * b;
  const int* bs = p.bin_start + p.cell_off[b];
  const int nb0 = p.cell_nb[3 * b], nb1 = p.cell_nb[3 * b + 1], nb2 = p.cell_nb[3 * b + 2];
  const int r0 = p.cell_reach[3 * b], r1 = p.cell_reach[3 * b + 1], r2b = p.cell_reach[3 * b + 2];
  const int b0 = p.a_bin3[3 * i], b1 = p.a_bin3[3 * i + 1], b2 = p.a_bin3[3 * i + 2];
  const int s0 = p.a_shift[3 * i], s1 = p.a_shift[3 * i + 1], s2 = p.a_shift[3 * i + 2];
  auto wrap_of = [](int x, int nb) { return x >= 0 ? x / nb : -((-x + nb - 1) / nb); };
  int running = 0;
  for (int oa = -r0; oa <= r0; ++oa) {
    const int w0 = wrap_of(b0 + oa, nb0), t0 = b0 + oa - w0 * nb0;
    for (int ob = -r1; ob <= r1; ++ob) {
      const int w1 = wrap_of(b1 + ob, nb1), t1 = b1 + ob - w1 * nb1;
      for (int oc = -r2b; oc <= r2b; ++oc) {
        const int w2 = wrap_of(b2 + oc, nb2), t2 = b2 + oc - w2 * nb2;
        const int q = (t0 * nb1 + t1) * nb2 + t2;
        const int beg = bs[q], end = bs[q + 1];
        for (int s = beg; s < end; s += 64) {
          const int ss = s + lane;
          bool hit = false;
          unsigned long long key = 0;
          if (ss < end) {
            const int j = p.bin_atoms[ss];
            const int ia = w0 - p.a_shift[3 * j] + s0, ib = w1 - p.a_shift[3 * j + 1] + s1, ic = w2 - p.a_shift[3 * j + 2] + s2;
            const double d2 = sq_dist(p.cart, L, i, j, ia, ib, ic);
            hit = d2 < p.r2 && sqrt(d2) > p.tol;
            key = cell_key(j - a0, ia, ib, ic);
          }
          const unsigned long long mask = __ballot(hit);
          if (FILL && hit) {
            const int pos = running + __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < CELL_SORT_MAX) keys[pos] = key;
          }
          running += __popcll(mask);
        }
      }
    }
  }
  if (!FILL) {
    if (lane == 0) p.center_cnt[i] = running;
    return;
  }
  if (running > CELL_SORT_MAX) {   // the later kernels stand down on the overflow flag; the host repeats the build with all pairs
    if (lane == 0) { *p.cell_flag = 1; *p.overflow = 1; }
    return;
  }
  int P = 64;
  while (P < running) P <<= 1;
  for (int idx = running + lane; idx < P; idx += 64) keys[idx] = ~0ull;
  __builtin_amdgcn_wave_barrier();
  for (int k = 2; k <= P; k <<= 1)
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int t = lane; t < P / 2; t += 64) {
        const int idx = (t / jj) * 2 * jj + (t % jj), partner = idx + jj;
        const unsigned long long x = keys[idx], y = keys[partner];
        const bool up = (idx & k) == 0;
        if ((x > y) == up) { keys[idx] = y; keys[partner] = x; }
      }
      __builtin_amdgcn_wave_barrier();
    }
  const int base = p.center_off[i];
  for (int idx = lane; idx < running; idx += 64) {
    const unsigned long long key = keys[idx];
    const int j = a0 + (int)(key >> 42);
    const int ia = (int)((key >> 28) & 16383) - 8192, ib = (int)((key >> 14) & 16383) - 8192, ic = (int)(key & 16383) - 8192;
    const int w = base + idx;
    if (w >= p.cap_edges) { *p.overflow = 1; continue; }
    p.e_center[w] = i;
    p.e_nbr[w] = j;
    p.e_img[3 * w] = ia; p.e_img[3 * w + 1] = ib; p.e_img[3 * w + 2] = ic;
    p.e_image[3 * w] = (float)ia; p.e_image[3 * w + 1] = (float)ib; p.e_image[3 * w + 2] = (float)ic;
    p.e_dist[w] = sqrt(sq_dist(p.cart, L, i, j, ia, ib, ic));
    p.e_owner[w] = b;
  }
}

// one wave per centre atom; lanes stride over the neighbour atoms j of the same structure
template <bool FILL>
__global__ __launch_bounds__(256) void k_neighbors(NlArgs p) {
  __shared__ unsigned long long sort_keys[FILL ? 4 * CELL_SORT_MAX : 1];
  const int lane = threadIdx.x & 63;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= p.n_atoms) return;
  const int b = p.atom_owner[i];
  if (p.cell_off && p.cell_off[b] >= 0) {
    neighbors_cells<FILL>(p, i, b, lane, sort_keys + (FILL ? (threadIdx.x >> 6) * CELL_SORT_MAX : 0));
    return;
  }
  const int a0 = p.atom_off[b], n = p.atom_off[b + 1] - a0;
  const double* L = p.lattice + 9 * b;
  const double* reach = p.reach + 3 * b;
  int running = FILL ? p.center_off[i] : 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int jj = j0 + lane;
    const int j = a0 + jj;
    int cnt = 0;
    int lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1};
    if (jj < n) {
      image_window(p, reach, i, j, lo, hi);
      for (int ia = lo[0]; ia <= hi[0]; ++ia)
        for (int ib = lo[1]; ib <= hi[1]; ++ib)
          for (int ic = lo[2]; ic <= hi[2]; ++ic) {
            const double d2 = sq_dist(p.cart, L, i, j, ia, ib, ic);
            if (d2 < p.r2 && sqrt(d2) > p.tol) ++cnt;
          }
    }
    int total;
    const int excl = wave_excl_scan(cnt, lane, total);
    if (FILL && cnt > 0) {
      int w = running + excl;
      for (int ia = lo[0]; ia <= hi[0]; ++ia)
        for (int ib = lo[1]; ib <= hi[1]; ++ib)
          for (int ic = lo[2]; ic <= hi[2]; ++ic) {
            const double d2 = sq_dist(p.cart, L, i, j, ia, ib, ic);
            if (d2 < p.r2) {
              const double d = sqrt(d2);
              if (d > p.tol) {
                if (w >= p.cap_edges) { *p.overflow = 1; ++w; continue; }
                p.e_center[w] = i;
                p.e_nbr[w] = j;
                p.e_img[3 * w] = ia; p.e_img[3 * w + 1] = ib; p.e_img[3 * w + 2] = ic;
                p.e_image[3 * w] = (float)ia; p.e_image[3 * w + 1] = (float)ib; p.e_image[3 * w + 2] = (float)ic;
                p.e_dist[w] = d;
                p.e_owner[w] = b;
                ++w;
              }
            }
          }
    }
    running += total;
  }
  if (!FILL && lane == 0) p.center_cnt[i] = running;
}

// reverse edge of every directed edge by binary search in the neighbour's (nbr, image)-sorted range
static __global__ void k_reverse(const int* __restrict__ e_center, const int* __restrict__ e_nbr, const int* __restrict__ e_img,
                          const int* __restrict__ center_off, DevCount n_edges, int* __restrict__ e_rev, int* __restrict__ is_first,
                          int* __restrict__ err) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges.get() || err[2]) return;     // err[2]: a speculative capacity was exceeded, the edge list is truncated
  const int i = e_center[e], j = e_nbr[e];
  const int t0 = -e_img[3 * e], t1 = -e_img[3 * e + 1], t2 = -e_img[3 * e + 2];
  int lo = center_off[j], hi = center_off[j + 1] - 1, found = -1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int n = e_nbr[mid], m0 = e_img[3 * mid], m1 = e_img[3 * mid + 1], m2 = e_img[3 * mid + 2];
    int cmp = n < i ? -1 : (n > i ? 1 : 0);
    if (cmp == 0) cmp = m0 < t0 ? -1 : (m0 > t0 ? 1 : 0);
    if (cmp == 0) cmp = m1 < t1 ? -1 : (m1 > t1 ? 1 : 0);
    if (cmp == 0) cmp = m2 < t2 ? -1 : (m2 > t2 ? 1 : 0);
    if (cmp == 0) { found = mid; break; }
    if (cmp < 0) lo = mid + 1; else hi = mid - 1;
  }
  if (found < 0) { atomicExch(err, 1); found = e; }   // unpaired directed edge: graph.py:273-278 raises
  e_rev[e] = found;
  is_first[e] = e < found ? 1 : 0;   // undirected index = order of first appearance (create_graph.c:152-189)
}

static __global__ void k_undirected(const int* __restrict__ e_center, const int* __restrict__ e_nbr, const int* __restrict__ e_rev,
                             const int* __restrict__ is_first, const int* __restrict__ first_scan, DevCount n_edges, int* __restrict__ e_d2u,
                             int* __restrict__ u_u2d, int* __restrict__ p_center, int* __restrict__ p_nbr, const int* __restrict__ overflow) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges.get() || *overflow || !is_first[e]) return;
  const int k = first_scan[e], s = e_rev[e];
  u_u2d[k] = e;
  e_d2u[e] = k;
  e_d2u[s] = k;
  p_center[2 * k] = e_center[e]; p_nbr[2 * k] = e_nbr[e];
  p_center[2 * k + 1] = e_center[s]; p_nbr[2 * k + 1] = e_nbr[s];
}

// per centre: number of edges strictly shorter than the bond-graph cutoff (graph.py:313 uses '<')
// One wave per atom (a thread per atom walked ~110 edges of dependent loads: 38 us for a 256-atom cell).
// (noncanon: raised when an atom has more short bonds than the per-atom angle adjoints index, kernels_angle_w.h WIN_LIST)
static __global__ __launch_bounds__(256) void k_short_count(const double* __restrict__ e_dist, const int* __restrict__ center_off, int n_atoms, double r_bond,
                                                     int* __restrict__ short_cnt, int* __restrict__ n_isolated, const int* __restrict__ overflow,
                                                     int max_short, int* __restrict__ noncanon) {
  const int lane = threadIdx.x & 63;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_atoms || *overflow) return;
  const int b = center_off[i], e = center_off[i + 1];
  int c = 0;
  for (int k = b + lane; k < e; k += 64) c += e_dist[k] < r_bond ? 1 : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
  if (lane == 0) {
    short_cnt[i] = c;
    if (e == b) atomicAdd(n_isolated, 1);
    if (c > max_short) *noncanon = 1;
  }
}

// angles owned by undirected bond k (graph.py:283-327): both ends, the end's other short edges
static __global__ void k_angle_count(const int* __restrict__ u_u2d, const int* __restrict__ e_rev, const int* __restrict__ e_center,
                              const double* __restrict__ e_dist, const int* __restrict__ short_cnt, DevCount n_und, double r_bond,
                              int* __restrict__ ang_cnt, const int* __restrict__ overflow, int* __restrict__ noncanon, int n_atoms,
                              int* __restrict__ boff, int* __restrict__ aoff, int* __restrict__ toff, int* __restrict__ toff4) {
  // Workgroup 0 first writes the per-atom offsets of the centre-major angle order (the index of the per-atom / team angle adjoints,
  // kernels_angle_w.h): boff = exclusive scan of the short-bond counts n, aoff = exclusive scan of n (n - 1), toff = exclusive scan of
  // the 16-row tile counts ceil(n (n - 1) / 16), N + 1 entries each -- and / or toff4 = exclusive scan of the blocked-tile counts of
  // the MD-size adjoints (blk_shape_of above, kernels_angle_blk.h).
  // Riding in this launch they cost nothing; as k_win_* launches after the build they were five more of an MD step's ~60.
  if (blockIdx.x == 0 && (boff || toff4)) {
    __shared__ int wtot[4][4];
    const int tid = threadIdx.x, n = n_atoms + 1, per = (n + 255) / 256;
    const int b = min(tid * per, n), e = min(b + per, n);
    int tb = 0, ta = 0, tt = 0, t4 = 0;
    // (an atom with ONE short bond has no angles and owns no (atom, bond) pair: like k_win_heads, which counts group heads)
    for (int q = b; q < e; ++q) {
      const int v = q < n_atoms ? short_cnt[q] : 0;
      int ps, qs;
      tb += v >= 2 ? v : 0; ta += v * (v - 1); tt += (v * (v - 1) + 15) >> 4; t4 += blk_shape_of(v, ps, qs);
    }
    // exclusive prefixes over the 256 threads: inclusive scans inside the four waves, their totals through LDS (a serial walk over the
    // earlier threads' sums -- up to 255 x 4 LDS reads -- made workgroup 0 the longest of the launch)
    const int lane = tid & 63, wave = tid >> 6;
    int ib = tb, ia = ta, it = tt, i4 = t4;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int ub = __shfl_up(ib, off), ua = __shfl_up(ia, off), ut = __shfl_up(it, off), u4 = __shfl_up(i4, off);
      if (lane >= off) { ib += ub; ia += ua; it += ut; i4 += u4; }
    }
    if (lane == 63) { wtot[wave][0] = ib; wtot[wave][1] = ia; wtot[wave][2] = it; wtot[wave][3] = i4; }
    __syncthreads();
    int rb = ib - tb, ra = ia - ta, rt = it - tt, r4 = i4 - t4;
    for (int q = 0; q < wave; ++q) { rb += wtot[q][0]; ra += wtot[q][1]; rt += wtot[q][2]; r4 += wtot[q][3]; }
    for (int q = b; q < e; ++q) {
      const int v = q < n_atoms ? short_cnt[q] : 0;
      if (boff) { boff[q] = rb; aoff[q] = ra; toff[q] = rt; }
      if (toff4) toff4[q] = r4;
      int ps, qs;
      rb += v >= 2 ? v : 0; ra += v * (v - 1); rt += (v * (v - 1) + 15) >> 4; r4 += blk_shape_of(v, ps, qs);
    }
  }
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_und.get() || *overflow) return;
  const int f = u_u2d[k], s = e_rev[f];
  // a bond EXACTLY at the cutoff (or one whose two directions disagree about it) owns angles without being anybody's short bond: the
  // angle set of its end atoms is then not the complete n (n - 1) block the per-atom adjoints assume
  if (!(e_dist[f] > r_bond) != (e_dist[f] < r_bond) || (e_dist[f] < r_bond) != (e_dist[s] < r_bond)) *noncanon = 1;
  int c = 0;
  if (!(e_dist[f] > r_bond)) {   // note '>' on the first directed edge's distance (graph.py:289)
    c += short_cnt[e_center[f]] - (e_dist[f] < r_bond ? 1 : 0);
    c += short_cnt[e_center[s]] - (e_dist[s] < r_bond ? 1 : 0);
  }
  ang_cnt[k] = c;
}

// One wave per undirected bond: the other edges of each end are tested 64 at a time and written in edge order (ballot prefix) --
// the order the serial loop of the reference produces (graph.py:283-327).  (A thread per bond walked up to 2 x ~110 edges: 72 us for
// a 256-atom cell.)
static __global__ __launch_bounds__(256) void k_angle_fill(const int* __restrict__ u_u2d, const int* __restrict__ e_rev, const int* __restrict__ e_center,
                                                    const int* __restrict__ e_d2u, const double* __restrict__ e_dist, const int* __restrict__ center_off,
                                                    const int* __restrict__ ang_off, DevCount n_und, double r_bond, int* __restrict__ a_ctr,
                                                    int* __restrict__ a_b1, int* __restrict__ a_d1, int* __restrict__ a_b2, int* __restrict__ a_d2,
                                                    int* __restrict__ is_node, int cap_angles, int* __restrict__ overflow,
                                                    const int* __restrict__ short_cnt, const int* __restrict__ boff, const int* __restrict__ aoff,
                                                    int* __restrict__ q_a, int* __restrict__ q_ctr, int* __restrict__ q_ab1, int* __restrict__ q_ab2,
                                                    const int* __restrict__ toff4, int* __restrict__ blk_a, int* __restrict__ blk_desc, int cap_tiles4) {
  const int lane = threadIdx.x & 63;
  const int k = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (k >= n_und.get() || *overflow) return;
  if (ang_off[k + 1] == ang_off[k]) return;
  if (ang_off[k + 1] > cap_angles) { if (lane == 0) *overflow = 1; return; }   // speculative capacity exceeded: the build is repeated exactly
  const int f = u_u2d[k];
  const int des[2] = {f, e_rev[f]};
  int w = ang_off[k];
  for (int end = 0; end < 2; ++end) {
    const int de = des[end], ctr = e_center[de];
    const int b = center_off[ctr], e = center_off[ctr + 1];
    // centre-major position of this group (kernels_angle_w.h: rows of atom c at aoff[c] + rank of the first bond x (n - 1) + position
    // in the group; ranks count the centre's SHORT edges in edge order -- the order k_win_ranks / k_win_rows establish)
    int rank1 = 0;
    if (q_a || blk_a) {
      for (int base = b; base < e; base += 64) {
        const int other = base + lane;
        rank1 += __popcll(__ballot(other < e && other < de && e_dist[other] < r_bond));
      }
    }
    const int n_c = (q_a || blk_a) ? short_cnt[ctr] : 0;
    // blocked tiles (kernels_angle_blk.h): the angle (first bond of rank i, second bond at position j' among the other n - 1) of this
    // atom is slot (i mod P) Q + (j' mod Q) of tile toff4[ctr] + (i / P) ceil((n - 1) / Q) + (j' / Q); blk_a holds angle + 1 (0: empty
    // slot), blk_desc the tile's shape and origin: log2 P | log2 Q << 4 | i0 << 8 | j'0 << 16
    int ps4 = 2, qs4 = 2;
    if (blk_a) blk_shape_of(n_c, ps4, qs4);
    const int nq4 = (n_c - 1 + (1 << qs4) - 1) >> qs4, t4_0 = blk_a ? toff4[ctr] : 0;
    const long row0 = q_a ? (long)aoff[ctr] + (long)rank1 * (n_c - 1) : 0;
    const int ab0 = q_a ? boff[ctr] : 0;
    const int w_group = w;
    int shorts_before = 0;
    for (int base = b; base < e; base += 64) {
      const int other = base + lane;
      const bool is_short = other < e && e_dist[other] < r_bond;
      const bool take = is_short && other != de;
      const unsigned long long m = __ballot(take), ms = __ballot(is_short), lower = (1ull << lane) - 1ull;
      if (take) {
        const int at = w + __popcll(m & lower);
        const int b2 = e_d2u[other];
        a_ctr[at] = ctr; a_b1[at] = k; a_d1[at] = de; a_b2[at] = b2; a_d2[at] = other;
        is_node[b2] = 1;
        if (q_a) {
          const long row = row0 + (at - w_group);
          if (row >= 0 && row < cap_angles) {
            q_a[row] = at; q_ctr[row] = ctr; q_ab1[row] = ab0 + rank1; q_ab2[row] = ab0 + shorts_before + __popcll(ms & lower);
          }
        }
        if (blk_a) {
          const int jp = at - w_group;            // position in the group = rank of the second bond among the others
          const long tile = (long)t4_0 + (long)(rank1 >> ps4) * nq4 + (jp >> qs4);
          if (rank1 < n_c && jp < n_c - 1 && tile < cap_tiles4) {   // (else: not a canonical graph, the index is not used)
            blk_a[tile * 16 + ((rank1 & ((1 << ps4) - 1)) << qs4) + (jp & ((1 << qs4) - 1))] = at + 1;
            blk_desc[tile] = ps4 | (qs4 << 4) | ((rank1 >> ps4 << ps4) << 8) | ((jp >> qs4 << qs4) << 16);   // (every angle of the tile writes the same value)
          }
        }
      }
      w += __popcll(m);
      shorts_before += __popcll(ms);
    }
  }
  if (lane == 0) is_node[k] = 1;
}

// (single-pass builds: the LAST workgroup to finish also gathers the counts of the build for its one device-to-host copy -- `collect`
// non-null; ticket = a zeroed counter -- instead of a launch of its own, 5.6 us of an MD step, behind this one.)
struct CollectCounts { const int *ed, *a, *eb, *flags; int* out; int* ticket; };
static __global__ __launch_bounds__(256) void k_bond_nodes(const int* __restrict__ is_node, const int* __restrict__ node_scan, DevCount n_und,
                                                    int* __restrict__ u_bnode, int* __restrict__ bn_und, int cap_nodes, int* __restrict__ overflow,
                                                    CollectCounts collect) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_und.get() && !*overflow) {
    if (is_node[k]) {
      u_bnode[k] = node_scan[k];
      if (node_scan[k] >= cap_nodes) *overflow = 1;
      else bn_und[node_scan[k]] = k;
    } else {
      u_bnode[k] = -1;
    }
  }
  if (!collect.out) return;
  __threadfence();                     // this thread's flag write is visible before the workgroup takes its ticket
  __syncthreads();
  __shared__ int last;
  if (threadIdx.x == 0) last = atomicAdd(collect.ticket, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    const volatile int* f = collect.flags;
    // {Ed, A, Eb, unpaired-edge flag, isolated atoms, overflow, cell-sort overflow, non-canonical angle sets}
    collect.out[0] = *collect.ed; collect.out[1] = *collect.a; collect.out[2] = *collect.eb;
    collect.out[3] = f[0]; collect.out[4] = f[1]; collect.out[5] = f[2]; collect.out[6] = f[3]; collect.out[7] = f[4];
  }
}

static __global__ void k_angle_compact(const int* __restrict__ a_b1, const int* __restrict__ a_b2, const int* __restrict__ u_bnode, int n_ang,
                                int* __restrict__ a_b1c, int* __restrict__ a_b2c) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_ang) return;
  a_b1c[a] = u_bnode[a_b1[a]];
  a_b2c[a] = u_bnode[a_b2[a]];
}

// counts of a single-pass build, gathered for one device-to-host copy:
// {Ed, A, Eb, unpaired-edge flag, isolated atoms, overflow, cell-sort overflow}
static __global__ void k_collect_counts(const int* ed, const int* a, const int* eb, const int* flags, int* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = *ed; out[1] = *a; out[2] = *eb; out[3] = flags[0]; out[4] = flags[1]; out[5] = flags[2]; out[6] = flags[3]; out[7] = flags[4];
  }
}

// Exclusive prefix sum of up to a few hundred thousand ints by ONE workgroup (one launch instead of the three of a
// device-wide scan, which is what counts for the ~1e4-element arrays of a single-structure build): each thread sums a
// contiguous chunk, the 1024 chunk sums are scanned through LDS, each thread writes its chunk.
static __global__ __launch_bounds__(1024) void k_small_scan(const int* __restrict__ in, int* __restrict__ out, int n) {
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + 1023) / 1024;
  const int b = min(tid * per, n), e = min(b + per, n);
  int s = 0;
  for (int k = b; k < e; ++k) s += in[k];
  int total;
  int excl = wave_excl_scan(s, lane, total);
  if (lane == 63) wave_tot[wave] = total;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  int run = base + excl;
  for (int k = b; k < e; ++k) {
    const int v = in[k];
    out[k] = run;
    run += v;
  }
}

// Exclusive prefix sum of a few chunks in ONE launch (chained scan): workgroups take chunk numbers from a ticket counter (so a chunk's
// predecessor is always running), scan their chunk, wait for the predecessor's running total and publish their own.  The arrays of an
// MD-size build (18-35k entries: 3-5 chunks) took the three launches below, 17.5 us against ~6 for this one -- a launch costs more than
// the scan.  state: SCAN_STATE_INTS zeroed ints per scan ([0] ticket, [16 + c] chunk c's inclusive total | 0x80000000 once published).
constexpr int SCAN_CHAIN_MAX = 64;              // chunks (every chunk reads all its predecessors' totals: longer arrays take the three launches below)
constexpr int SCAN_STATE_INTS = 16 + SCAN_CHAIN_MAX;
static __global__ __launch_bounds__(1024) void k_scan_chained(const int* __restrict__ in, int* __restrict__ out, int n, int* __restrict__ state) {
  __shared__ int wave_tot[16];
  __shared__ int s_chunk, s_prefix;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_chunk = atomicAdd(state, 1);
  __syncthreads();
  const int chunk = s_chunk;
  constexpr int PER = 8;                        // 8,192 entries per chunk
  const int b = min(chunk * 8192 + tid * PER, n), e = min(b + PER, n);
  int v[PER], s = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { v[k] = b + k < e ? in[b + k] : 0; s += v[k]; }
  int incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int run = incl - s;
  for (int w = 0; w < wave; ++w) run += wave_tot[w];
  // every chunk publishes its OWN total at once and sums its predecessors' totals (lanes of wave 0 in parallel): one memory round trip
  // whatever the chunk count -- waiting for the predecessor's running total instead chained 3-5 round trips (10.5 us per scan)
  if (wave == 0) {
    int total = 0;
    for (int w = 0; w < 16; ++w) total += wave_tot[w];
    if (lane == 0) __hip_atomic_store(state + 16 + chunk, total | (int)0x80000000, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int prefix = 0;
    for (int q = lane; q < chunk; q += 64) {
      int f;
      do { f = __hip_atomic_load(state + 16 + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while (f == 0);
      prefix += f & 0x7fffffff;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) prefix += __shfl_xor(prefix, off);
    if (lane == 0) s_prefix = prefix;
  }
  __syncthreads();
  run += s_prefix;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (b + k < e) out[b + k] = run;
    run += v[k];
  }
}

// Exclusive prefix sum of longer arrays: chunks of SCAN_CHUNK ints per workgroup -- chunk totals (k_scan_totals), one
// k_small_scan over the totals, then every workgroup rescans its chunk from its offset (k_scan_apply).  Three launches,
// 2 reads + 1 write per element; replaces the library scan the first rounds used.
constexpr int SCAN_CHUNK = 8192;
__device__ __forceinline__ int block_sum_1024(int v, int* wave_tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if (lane == 0) wave_tot[wave] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < 16; ++w) t += wave_tot[w];
  return t;
}
static __global__ __launch_bounds__(1024) void k_scan_totals(const int* __restrict__ in, int* __restrict__ totals, int n) {
  __shared__ int wave_tot[16];
  const int b0 = blockIdx.x * SCAN_CHUNK;
  int s = 0;
  for (int k = b0 + threadIdx.x; k < min(b0 + SCAN_CHUNK, n); k += 1024) s += in[k];
  const int t = block_sum_1024(s, wave_tot);
  if (threadIdx.x == 0) totals[blockIdx.x] = t;
}
static __global__ __launch_bounds__(1024) void k_scan_apply(const int* __restrict__ in, int* __restrict__ out, const int* __restrict__ chunk_off, int n) {
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PER = SCAN_CHUNK / 1024;
  const int b = min(blockIdx.x * SCAN_CHUNK + tid * PER, n), e = min(b + PER, n);
  int v[PER], s = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { v[k] = b + k < e ? in[b + k] : 0; s += v[k]; }
  int total;
  const int excl = wave_excl_scan(s, lane, total);
  if (lane == 63) wave_tot[wave] = total;
  __syncthreads();
  int run = chunk_off[blockIdx.x] + excl;
  for (int w = 0; w < wave; ++w) run += wave_tot[w];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (b + k < e) out[b + k] = run;
    run += v[k];
  }
}

// Several device-to-device copies in one launch (the index arrays of a freshly built graph into the batch arena):
// blockIdx.y picks the segment, the blocks of a row stride over its 4-byte words.
constexpr int MULTI_COPY_MAX = 28;
struct MultiCopy {
  void* dst[MULTI_COPY_MAX];
  const void* src[MULTI_COPY_MAX];
  unsigned long long words[MULTI_COPY_MAX];
  // rows of the grid past the n_copy plain copies (the small kernels that used to follow: a launch costs more than they do):
  int n_copy;
  const double* cvt_src[2];   // rows n_copy, n_copy + 1: float64 -> float32 (fractional coordinates, lattices)
  float* cvt_dst[2];
  int cvt_n[2];
  const int *a_b1, *a_b2, *u_bnode_new;   // row n_copy + 2: compact bond-node indices of the angles (reads the builder's u_bnode, not the copy)
  int *a_b1c, *a_b2c;
  int n_ang;
  // ... and, when the builder emitted the centre-major order (k_angle_fill), the rest of the per-atom adjoints' index: the compact bond
  // indices in that order, the bond behind every (atom, rank) pair, the flags
  const int *q_a_new, *q_ab1_new;
  int *q_b1c, *q_b2c, *abbond, *win_flag;
  int win_grid;
  // ... or the blocked tiles of the MD-size adjoints (kernels_angle_blk.h; row n_copy + 3): slot -> angle (-1: empty), compact
  // bond indices, centre; [0 .. 16 toff4[N]) of the arena arrays (capacity cap_tiles4 tiles), and the tile count itself
  const int *blk_a_new, *blk_desc_new, *toff4_new, *a_ctr_new;
  int *blk_a, *blk_b1c, *blk_b2c, *blk_ctr, *blk_desc, *blk_tiles;
  int n_atoms, cap_tiles4;
};
static __global__ __launch_bounds__(256) void k_multi_copy(MultiCopy m) {
  const int seg = blockIdx.y;
  if (seg >= m.n_copy) {
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x, tstride = gridDim.x * blockDim.x;
    const int q = seg - m.n_copy;
    if (q < 2) {
      for (int t = t0; t < m.cvt_n[q]; t += tstride) m.cvt_dst[q][t] = (float)m.cvt_src[q][t];
    } else if (q == 3) {
      if (!m.blk_a_new) return;
      const int tiles = min(m.toff4_new[m.n_atoms], m.cap_tiles4);
      if (t0 == 0) *m.blk_tiles = tiles;
      for (int t = t0; t < tiles; t += tstride) m.blk_desc[t] = m.blk_desc_new[t];
      for (int sl = t0; sl < 16 * tiles; sl += tstride) {
        const int a = m.blk_a_new[sl] - 1;
        m.blk_a[sl] = a;
        m.blk_b1c[sl] = a >= 0 ? m.u_bnode_new[m.a_b1[a]] : 0;      // (an empty slot reads row 0 of the tables and contributes zeros)
        m.blk_b2c[sl] = a >= 0 ? m.u_bnode_new[m.a_b2[a]] : 0;
        m.blk_ctr[sl] = a >= 0 ? m.a_ctr_new[a] : 0;
      }
    } else {
      for (int a = t0; a < m.n_ang; a += tstride) {
        m.a_b1c[a] = m.u_bnode_new[m.a_b1[a]];
        m.a_b2c[a] = m.u_bnode_new[m.a_b2[a]];
        if (m.q_a_new) {
          const int ar = m.q_a_new[a];               // the angle that is row a of the centre-major order
          const int c1 = m.u_bnode_new[m.a_b1[ar]];
          m.q_b1c[a] = c1;
          m.q_b2c[a] = m.u_bnode_new[m.a_b2[ar]];
          m.abbond[m.q_ab1_new[a]] = c1;             // (every row of a group writes the same value)
        }
      }
      if (m.q_a_new && t0 == 0) { m.win_flag[0] = 1; m.win_flag[1] = 0; m.win_flag[2] = 0; m.win_flag[3] = m.win_grid; }
    }
    return;
  }
  const unsigned long long n = m.words[seg];
  const unsigned* __restrict__ src = static_cast<const unsigned*>(m.src[seg]);
  unsigned* __restrict__ dst = static_cast<unsigned*>(m.dst[seg]);
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const unsigned long long n4 = n / 4;
    const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src);
    uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst);
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) d4[i] = s4[i];
    for (unsigned long long i = 4 * n4 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
  } else {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
  }
}

static __global__ void k_f64_to_f32(const double* __restrict__ src, float* __restrict__ dst, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = (float)src[t];
}

}  // namespace chg
