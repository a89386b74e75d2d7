// engine_graph.hip -- device-side graph construction: chg_batch_build replaces pymatgen's get_neighbor_list +
// create_graph.c + Graph.line_graph_adjacency_list (reference graph/converter.py:132-159, graph/graph.py:249-328).
#include "engine_internal.h"

#include "kernels_geom.h"
#include "kernels_graph.h"

namespace chgh {

size_t scan_scratch_ints(int n) { return n <= SCAN_CHUNK ? 1 : 2 * ((size_t)n / SCAN_CHUNK + 2); }

int exclusive_scan_with(chg_engine* eng, int* scratch, const int* in, int* out, int n) {
  if (n <= 0) return CHG_OK;
  if (n <= SCAN_CHUNK) {   // one workgroup, one launch; beyond a chunk its strided per-thread runs get slow (28k elements: 29 us)
    hipLaunchKernelGGL(k_small_scan, dim3(1), dim3(1024), 0, eng->stream, in, out, n);
    return CHG_OK;
  }
  const int nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;          // <= 2^18 for n < 2^31
  int *totals = scratch, *offs = scratch + nchunks + 1;
  hipLaunchKernelGGL(k_scan_totals, dim3(nchunks), dim3(1024), 0, eng->stream, in, totals, n);
  hipLaunchKernelGGL(k_small_scan, dim3(1), dim3(1024), 0, eng->stream, totals, offs, nchunks);   // nchunks <= 65536 up to n = 5e8: one level is enough
  hipLaunchKernelGGL(k_scan_apply, dim3(nchunks), dim3(1024), 0, eng->stream, in, out, offs, n);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}


// ---- device-side graph construction ------------------------------------------------------------------
struct TmpPool {   // scratch device memory of one chg_batch_build call: bump allocation from the engine's
                   // grow-only scratch buffer; requests that do not fit fall back to hipMalloc and make the
                   // buffer grow before the next call
  chg_engine* eng;
  size_t pos = 0, overflow = 0;
  std::vector<void*> extra;
  explicit TmpPool(chg_engine* e) : eng(e) {
    if (eng->scratch_wanted > eng->scratch_bytes) {
      if (eng->scratch) hipFree(eng->scratch);
      eng->scratch = nullptr;
      eng->scratch_bytes = 0;
      const size_t want = eng->scratch_wanted + eng->scratch_wanted / 4;
      if (hipMalloc(&eng->scratch, want) == hipSuccess) eng->scratch_bytes = want; else eng->scratch = nullptr;
    }
  }
  template <class T>
  T* get(size_t n) {
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~size_t(255);
    if (pos + bytes <= eng->scratch_bytes) {
      T* p = reinterpret_cast<T*>(eng->scratch + pos);
      pos += bytes;
      return p;
    }
    overflow += bytes;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    extra.push_back(p);
    return static_cast<T*>(p);
  }
  ~TmpPool() {
    for (void* p : extra) hipFree(p);
    if (overflow) eng->scratch_wanted = std::max(eng->scratch_wanted, pos + overflow);
  }
};

// `state`: SCAN_STATE_INTS zeroed ints (graph_pass clears them with its counters): mid-size arrays are scanned by ONE chained launch
int exclusive_scan(chg_engine* eng, TmpPool& tmp, const int* in, int* out, int n, int* state = nullptr) {
  if (n <= 0) return CHG_OK;
  if (state && n > SCAN_CHUNK && (n + 8191) / 8192 <= SCAN_CHAIN_MAX) {
    hipLaunchKernelGGL(k_scan_chained, dim3((n + 8191) / 8192), dim3(1024), 0, eng->stream, in, out, n, state);
    return CHG_OK;
  }
  if ((size_t)n / SCAN_CHUNK + 1 > (1u << 16)) { eng->err = "graph build: array too long for the two-level scan"; return CHG_EINVAL; }
  int* scratch = tmp.get<int>(scan_scratch_ints(n));
  if (!scratch) { eng->err = "graph build: scratch allocation failed"; return CHG_ENOMEM; }
  return exclusive_scan_with(eng, scratch, in, out, n);
}

int acquire_arena(chg_engine* eng, chg_batch* b, size_t total) {
  if (eng->memory_limit && total > eng->memory_limit) {
    eng->err = "batch needs " + std::to_string(total) + " bytes of device memory, the engine's limit is " + std::to_string(eng->memory_limit);
    return CHG_ENOMEM;
  }
  std::lock_guard<std::mutex> lk(eng->pool_mu);
  int best = -1;
  for (int i = 0; i < (int)eng->arena_pool.size(); ++i)
    if (eng->arena_pool[i].second >= total && (best < 0 || eng->arena_pool[i].second < eng->arena_pool[best].second)) best = i;
  if (best >= 0) {
    b->arena = eng->arena_pool[best].first;
    b->arena_bytes = eng->arena_pool[best].second;
    eng->arena_pool.erase(eng->arena_pool.begin() + best);
    return CHG_OK;
  }
  for (auto& a : eng->arena_pool) hipFree(a.first);   // nothing fits: drop the cache before growing
  eng->arena_pool.clear();
  // a little headroom (3 %, 16 MiB granules): the batches of an epoch / the chunks of a sweep differ slightly in size and
  // should reuse one arena instead of paying a multi-GB hipFree + hipMalloc each
  const size_t roomy = ((total + total / 32) + (size_t(16) << 20) - 1) & ~((size_t(16) << 20) - 1);
  if ((!eng->memory_limit || roomy <= eng->memory_limit) && hipMalloc(&b->arena, roomy) == hipSuccess) {
    b->arena_bytes = roomy;
    return CHG_OK;
  }
  (void)hipGetLastError();
  if (hipMalloc(&b->arena, total) != hipSuccess) {
    (void)hipGetLastError();
    for (auto& a : eng->work_pool) hipFree(a.first);   // pooled training workspaces (tens of GB) go before giving up
    eng->work_pool.clear();
    eng->work_kind.clear();
    if (hipMalloc(&b->arena, total) != hipSuccess) {
      (void)hipGetLastError();
      eng->err = "hipMalloc of " + std::to_string(total) + " bytes failed";
      return CHG_ENOMEM;
    }
  }
  b->arena_bytes = total;
  return CHG_OK;
}

template <class T>
int d2d(chg_engine* eng, T* dst, const T* src, size_t n) {
  if (n == 0) return CHG_OK;
  HIP_TRY(eng, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToDevice, eng->stream));
  return CHG_OK;
}

// Two ways through the same kernels:
//  * exact (first build of a shape): count pass -> host reads Ed -> fill -> host reads A -> fill -> host reads Eb:
//    three blocking round trips;
//  * single pass (every later build: MD rebuilds the graph of the same cell every step, a sweep builds chunk after
//    chunk of similar structures): the scratch arrays are sized from the PREVIOUS build's per-atom counts plus
//    headroom, every kernel takes its counts from device memory, and the host reads {Ed, A, Eb, flags} once at the
//    end.  A capacity that turns out too small raises a device-side flag and the build is repeated exactly.
struct GraphCounts { int Ed = 0, A = 0, Eb = 0, unpaired = 0, isolated = 0; bool cell_overflow = false, noncanonical = false; };

// device copies of the host-side binning (one entry per structure; all-pairs structures have off = -1)
struct CellLists { const int *off, *nb, *reach, *bin_start, *bin_atoms, *bin3, *shift; };

// Bins for the structures large enough to profit (the arithmetic of host_graph.cpp neighbor_list_cells: bins at least
// one cutoff wide along every axis, wrapped coordinates, floor shifts).  Any binning yields the same rows -- the bins
// only propose candidates -- so the bin counts are free to differ from the host's.
struct HostCells {
  std::vector<int> off, nb, reach, bin_start, bin_atoms, bin3, shift;
  bool any = false;
};

void bin_structures(const chg_structs_host* h, const std::vector<double>& hk, double r, int min_atoms, HostCells& hc) {
  const int B = h->n_struct, N = h->n_atoms;
  hc.off.assign(B, -1); hc.nb.assign(3 * (size_t)B, 1); hc.reach.assign(3 * (size_t)B, 0);
  hc.bin_atoms.assign(std::max(N, 1), 0); hc.bin3.assign(3 * (size_t)std::max(N, 1), 0); hc.shift.assign(3 * (size_t)std::max(N, 1), 0);
  hc.bin_start.clear();
  for (int b = 0; b < B; ++b) {
    const int a0 = h->atom_off[b], n = h->atom_off[b + 1] - a0;
    if (n < min_atoms || n >= (1 << 21)) continue;
    int nb[3];
    for (int k = 0; k < 3; ++k) nb[k] = std::max(1, std::min(1024, (int)std::floor(hk[3 * b + k] / r)));
    while ((int64_t)nb[0] * nb[1] * nb[2] > 4 * (int64_t)n + 64) {   // keep the table O(atoms)
      const int k = nb[0] >= nb[1] && nb[0] >= nb[2] ? 0 : (nb[1] >= nb[2] ? 1 : 2);
      nb[k] = (nb[k] + 1) / 2;
    }
    bool ok = true;
    std::vector<int> bin_of(n);
    for (int i = 0; i < n && ok; ++i)
      for (int k = 0; k < 3; ++k) {
        const double f = h->frac[3 * (size_t)(a0 + i) + k];
        double fl = std::floor(f), w = f - fl;
        if (w >= 1.0) { w -= 1.0; fl += 1.0; }
        if (!(std::fabs(fl) < 4000.0)) { ok = false; break; }        // images must fit the sort key (and NaN lands here)
        hc.shift[3 * (size_t)(a0 + i) + k] = (int)fl;
        hc.bin3[3 * (size_t)(a0 + i) + k] = std::min(nb[k] - 1, (int)(w * nb[k]));
      }
    if (!ok) continue;
    const int n_bins = nb[0] * nb[1] * nb[2];
    const int base = (int)hc.bin_start.size();
    hc.bin_start.resize(base + n_bins + 1, 0);
    int* bs = hc.bin_start.data() + base;
    for (int i = 0; i < n; ++i) {
      const int* q = hc.bin3.data() + 3 * (size_t)(a0 + i);
      bin_of[i] = (q[0] * nb[1] + q[1]) * nb[2] + q[2];
      ++bs[bin_of[i] + 1];
    }
    bs[0] = a0;                                                       // positions index the batch-wide bin_atoms array
    for (int q = 0; q < n_bins; ++q) bs[q + 1] += bs[q];
    std::vector<int> fill(bs, bs + n_bins);
    for (int i = 0; i < n; ++i) hc.bin_atoms[fill[bin_of[i]]++] = a0 + i;
    for (int k = 0; k < 3; ++k) {
      hc.nb[3 * b + k] = nb[k];
      // |x_j + I nb - x_i| <= r nb / h in bin units: the offset is at most floor(r nb / h) + 1
      hc.reach[3 * b + k] = (int)std::floor(r * nb[k] / hk[3 * b + k] + 1e-9) + 1;
    }
    hc.off[b] = base;
    hc.any = true;
  }
  if (hc.bin_start.empty()) hc.bin_start.push_back(0);
}

int graph_pass(chg_engine* eng, TmpPool& tmp, const chg_structs_host* h, const double* d_cart, const double* d_frac, const double* d_lat,
               const double* d_reach, const int* d_owner, const int* d_aoff, const CellLists* cells, double r_atom, double r_bond, double tol,
               bool speculative, int capE, int capA, int capEb, GraphCounts& gc, bool& overflowed, int*& e_center, int*& e_nbr, float*& e_image,
               int*& e_owner, int*& e_rev, int*& e_d2u, int*& p_center, int*& p_nbr, int*& u_u2d, int*& u_bnode, int*& bn_und, int*& a_ctr,
               int*& a_b1, int*& a_d1, int*& a_b2, int*& a_d2, int*& short_cnt_out, int*& boff, int*& aoff, int*& toff, int*& q_a, int*& q_ctr,
               int*& q_ab1, int*& q_ab2, int*& toff4, int*& blk_a, int*& blk_desc, int& blk_cap) {
  const int N = h->n_atoms;
  hipStream_t st = eng->stream;
  overflowed = false;
  // Everything that starts at zero sits in ONE block cleared by one memset (they were six, ~3.4 us each on the device): the per-centre
  // counts, the flags, the state of the chained scans and -- single-pass builds know the capacities up front -- the per-bond /
  // per-edge counters below.  (The exact build learns Ed only after its first round trip: those get a second block.)
  const size_t z_ccnt = 0, z_flags = z_ccnt + (size_t)N + 1, z_scan = z_flags + 8, z_head = z_scan + 3 * (size_t)SCAN_STATE_INTS;
  const int capU0 = capE / 2;
  auto tail_ints = [&](int cE, int cU) { return 2 * ((size_t)cU + 1) + ((size_t)cE + 1) + (size_t)std::max(N, 1); };
  // blocked tiles of the MD-size angle adjoints (kernels_angle_blk.h): slot -> angle + 1, written by k_angle_fill; a single-pass
  // build knows the capacities up front and clears the slots with everything else, an exact one after it has learnt A
  const long blk_max = blk_max_angles();
  const bool blk_wanted = blk_max > 0 && N + 1 <= 8192;
  toff4 = nullptr; blk_a = nullptr; blk_desc = nullptr; blk_cap = 0;
  if (speculative && blk_wanted && capA > 0 && ((double)capA - 4096.0) / 1.25 <= 1.1 * (double)blk_max) blk_cap = (int)blk_tile_bound(capA, capEb, N);
  const size_t z_total = z_head + (speculative ? tail_ints(capE, capU0) + (size_t)blk_cap * 16 : 0);
  int* zblock = tmp.get<int>(z_total);
  int* d_coff = tmp.get<int>(N + 1);
  int* d_counts = tmp.get<int>(8);
  if (!zblock || !d_coff || !d_counts) { eng->err = "graph build: scratch allocation failed"; return CHG_ENOMEM; }
  int* d_ccnt = zblock + z_ccnt;
  int* d_flags = zblock + z_flags;   // [0] unpaired directed edge, [1] isolated atoms, [2] speculative capacity exceeded, [3] cell list overflow,
                                     // [4] angle sets not the canonical n (n - 1) blocks (k_short_count / k_angle_count)
  int* scan_state = zblock + z_scan;
  HIP_TRY(eng, hipMemsetAsync(zblock, 0, sizeof(int) * z_total, st));
  NlArgs nl{};
  nl.cart = d_cart; nl.frac = d_frac; nl.lattice = d_lat; nl.reach = d_reach; nl.atom_owner = d_owner; nl.atom_off = d_aoff;
  nl.n_atoms = N; nl.r2 = r_atom * r_atom; nl.tol = tol; nl.center_cnt = d_ccnt; nl.overflow = d_flags + 2; nl.cell_flag = d_flags + 3;
  if (cells) {
    nl.cell_off = cells->off; nl.cell_nb = cells->nb; nl.cell_reach = cells->reach; nl.bin_start = cells->bin_start;
    nl.bin_atoms = cells->bin_atoms; nl.a_bin3 = cells->bin3; nl.a_shift = cells->shift;
  }
  const dim3 wave_per_atom((unsigned)((N + 3) / 4));
  hipLaunchKernelGGL((k_neighbors<false>), wave_per_atom, dim3(256), 0, st, nl);
  TRY(exclusive_scan(eng, tmp, d_ccnt, d_coff, N + 1));
  int Ed = 0;
  if (!speculative) {
    HIP_TRY(eng, hipMemcpyAsync(&Ed, d_coff + N, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(eng, hipStreamSynchronize(st));
    if (Ed & 1) { eng->err = "graph build: odd number of directed edges"; return CHG_EINVAL; }
    capE = Ed;
  }
  const int capU = capE / 2;
  const DevCount nE{Ed, speculative ? d_coff + N : nullptr, 1}, nU{Ed / 2, speculative ? d_coff + N : nullptr, 2};
  e_center = tmp.get<int>(capE); e_nbr = tmp.get<int>(capE);
  int* e_img = tmp.get<int>(3 * (size_t)capE);
  e_image = tmp.get<float>(3 * (size_t)capE);
  double* e_dist = tmp.get<double>(capE);
  e_owner = tmp.get<int>(capE); e_rev = tmp.get<int>(capE); e_d2u = tmp.get<int>(capE);
  p_center = tmp.get<int>(capE); p_nbr = tmp.get<int>(capE);
  u_u2d = tmp.get<int>(capU);
  int* zblock2 = speculative ? zblock + z_head : tmp.get<int>(tail_ints(capE, capU));
  int* ang_off = tmp.get<int>(capU + 1);
  int* node_scan = tmp.get<int>(capU + 1);
  u_bnode = tmp.get<int>(capU);
  int* ang_cnt = zblock2; int* is_node = zblock2 ? ang_cnt + capU + 1 : nullptr; int* is_first = zblock2 ? is_node + capU + 1 : nullptr;
  int* short_cnt = zblock2 ? is_first + capE + 1 : nullptr;
  int* first_scan = tmp.get<int>(capE + 1);
  if (!e_center || !e_nbr || !e_img || !e_image || !e_dist || !e_owner || !e_rev || !e_d2u || !is_first || !first_scan || !p_center ||
      !p_nbr || !u_u2d || !short_cnt || !ang_cnt || !ang_off || !is_node || !node_scan || !u_bnode) {
    eng->err = "graph build: scratch allocation failed";
    return CHG_ENOMEM;
  }
  if (!speculative) HIP_TRY(eng, hipMemsetAsync(zblock2, 0, sizeof(int) * tail_ints(capE, capU), st));
  if (capE > 0) {
    nl.center_off = d_coff; nl.e_center = e_center; nl.e_nbr = e_nbr; nl.e_img = e_img; nl.e_image = e_image; nl.e_dist = e_dist;
    nl.e_owner = e_owner; nl.cap_edges = capE;
    hipLaunchKernelGGL((k_neighbors<true>), wave_per_atom, dim3(256), 0, st, nl);
    hipLaunchKernelGGL(k_reverse, g1(capE), dim3(256), 0, st, e_center, e_nbr, e_img, d_coff, nE, e_rev, is_first, d_flags);
    TRY(exclusive_scan(eng, tmp, is_first, first_scan, capE + 1, scan_state));
    hipLaunchKernelGGL(k_undirected, g1(capE), dim3(256), 0, st, e_center, e_nbr, e_rev, is_first, first_scan, nE, e_d2u, u_u2d, p_center, p_nbr,
                       d_flags + 2);
  }
  hipLaunchKernelGGL(k_short_count, g1((int64_t)N * 64), dim3(256), 0, st, (const double*)e_dist, (const int*)d_coff, N, r_bond, short_cnt, d_flags + 1,
                     d_flags + 2, WIN_LIST, d_flags + 4);
  int A = 0, Eb = 0;
  if (capU > 0) {
    // (the in-launch scan is one workgroup: batches of a few thousand atoms; and only when the batch is likely to use the index -- a
    // single-pass build knows the previous build's angle count, an exact one emits it in any case)
    const long tmin = team_min_angles();
    const bool index_likely = tmin >= 0 && (!speculative || ((double)capA - 4096.0) / 1.25 >= 0.8 * (double)tmin);   // capA = previous A x 1.25 + 4096
    boff = (N + 1 <= 8192 && index_likely) ? tmp.get<int>(N + 1) : nullptr;   // (not emitted but needed after all: prepare_windows builds it, k_win_*)
    aoff = boff ? tmp.get<int>(N + 1) : nullptr;
    toff = boff ? tmp.get<int>(N + 1) : nullptr;
    if (boff && (!aoff || !toff)) boff = nullptr;
    if (blk_wanted && (blk_cap > 0 || !speculative)) toff4 = tmp.get<int>(N + 1);
    hipLaunchKernelGGL(k_angle_count, g1(capU), dim3(256), 0, st, u_u2d, e_rev, e_center, e_dist, short_cnt, nU, r_bond, ang_cnt, d_flags + 2, d_flags + 4,
                       N, boff, aoff, toff, toff4);
    TRY(exclusive_scan(eng, tmp, ang_cnt, ang_off, capU + 1, scan_state + SCAN_STATE_INTS));   // entries past Eu are zero: the total sits at ang_off[capU]
  } else {
    HIP_TRY(eng, hipMemsetAsync(ang_off, 0, sizeof(int) * (capU + 1), st));
  }
  int flags[5] = {0, 0, 0, 0, 0};
  if (!speculative) {
    HIP_TRY(eng, hipMemcpyAsync(&A, ang_off + capU, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(eng, hipMemcpyAsync(flags, d_flags, sizeof(int) * 5, hipMemcpyDeviceToHost, st));
    HIP_TRY(eng, hipStreamSynchronize(st));
    if (flags[3]) { gc.cell_overflow = true; overflowed = true; return CHG_OK; }
    if (flags[0]) { eng->err = "graph build: number of directed edges != 2 * number of undirected edges (directed edges are not complete)"; return CHG_EINVAL; }
    capA = A;
  }
  a_ctr = tmp.get<int>(capA); a_b1 = tmp.get<int>(capA); a_d1 = tmp.get<int>(capA); a_b2 = tmp.get<int>(capA); a_d2 = tmp.get<int>(capA);
  if (!a_ctr || !a_b1 || !a_d1 || !a_b2 || !a_d2) { eng->err = "graph build: scratch allocation failed"; return CHG_ENOMEM; }
  // the centre-major order of the per-atom / team angle adjoints, emitted with the angles (small batches: the index launches that used
  // to follow the build were 5 of an MD step's ~60)
  short_cnt_out = short_cnt;
  q_a = q_ctr = q_ab1 = q_ab2 = nullptr;
  if (boff && capA > 0) {
    q_a = tmp.get<int>(capA); q_ctr = tmp.get<int>(capA); q_ab1 = tmp.get<int>(capA); q_ab2 = tmp.get<int>(capA);
    if (!q_a || !q_ctr || !q_ab1 || !q_ab2) q_a = q_ctr = q_ab1 = q_ab2 = nullptr;
  }
  if (toff4 && capA > 0) {
    if (speculative) {
      blk_a = zblock + z_head + tail_ints(capE, capU0);
    } else if (capA <= blk_max) {
      blk_cap = (int)blk_tile_bound(capA, capU, N);     // (Eb is not known yet: every bond could be a node)
      blk_a = tmp.get<int>((size_t)blk_cap * 16);
      if (!blk_a) blk_cap = 0;
      else HIP_TRY(eng, hipMemsetAsync(blk_a, 0, sizeof(int) * (size_t)blk_cap * 16, st));
    }
  }
  if (blk_a) blk_desc = tmp.get<int>(blk_cap);
  if (!blk_a || !blk_desc) { toff4 = nullptr; blk_a = nullptr; blk_cap = 0; }
  if (capA > 0 && capU > 0) {
    hipLaunchKernelGGL(k_angle_fill, g1((int64_t)capU * 64), dim3(256), 0, st, u_u2d, e_rev, e_center, e_d2u, e_dist, d_coff, ang_off, nU, r_bond, a_ctr, a_b1,
                       a_d1, a_b2, a_d2, is_node, capA, d_flags + 2, short_cnt, boff, aoff, q_a, q_ctr, q_ab1, q_ab2, toff4, blk_a, blk_desc, blk_cap);
    TRY(exclusive_scan(eng, tmp, is_node, node_scan, capU + 1, scan_state + 2 * SCAN_STATE_INTS));
  } else {
    HIP_TRY(eng, hipMemsetAsync(node_scan, 0, sizeof(int) * (capU + 1), st));
  }
  if (!speculative) {
    if (A > 0) {
      HIP_TRY(eng, hipMemcpyAsync(&Eb, node_scan + capU, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(eng, hipStreamSynchronize(st));
    }
    capEb = Eb;
  }
  bn_und = tmp.get<int>(capEb);
  if (!bn_und) { eng->err = "graph build: scratch allocation failed"; return CHG_ENOMEM; }
  // (single-pass builds: the kernel's last workgroup also gathers the counts for the one device-to-host copy below; d_flags[6] is its ticket)
  const CollectCounts collect{d_coff + N, ang_off + capU, node_scan + capU, d_flags, (speculative && capU > 0) ? d_counts : nullptr, d_flags + 6};
  if (capU > 0) hipLaunchKernelGGL(k_bond_nodes, g1(capU), dim3(256), 0, st, is_node, node_scan, nU, u_bnode, bn_und, capEb, d_flags + 2, collect);
  HIP_TRY(eng, hipGetLastError());
  if (speculative) {   // the one round trip of this path
    int hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (capU <= 0)     // (no bond at all: k_bond_nodes was not launched)
      hipLaunchKernelGGL(k_collect_counts, dim3(1), dim3(64), 0, st, (const int*)(d_coff + N), (const int*)(ang_off + capU),
                         (const int*)(node_scan + capU), (const int*)d_flags, d_counts);
    HIP_TRY(eng, hipMemcpyAsync(hc, d_counts, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(eng, hipStreamSynchronize(st));
    Ed = hc[0]; A = hc[1]; Eb = hc[2]; flags[0] = hc[3]; flags[1] = hc[4]; flags[2] = hc[5]; flags[4] = hc[7];
    if (hc[6]) { gc.cell_overflow = true; overflowed = true; return CHG_OK; }
    if (flags[2] || Ed > capE || A > capA || Eb > capEb) { overflowed = true; return CHG_OK; }
    if (Ed & 1) { eng->err = "graph build: odd number of directed edges"; return CHG_EINVAL; }
    if (flags[0]) { eng->err = "graph build: number of directed edges != 2 * number of undirected edges (directed edges are not complete)"; return CHG_EINVAL; }
  }
  gc.Ed = Ed; gc.A = A; gc.Eb = Eb; gc.unpaired = flags[0]; gc.isolated = flags[1]; gc.noncanonical = flags[4] != 0;
  return CHG_OK;
}

int build_batch_on_device(chg_engine* eng, const chg_structs_host* h, double r_atom, double r_bond, double tol, chg_batch** out,
                          int32_t* counts_out) {
  const int B = h->n_struct, N = h->n_atoms;
  hipStream_t st = eng->stream;
  // per-structure constants and float64 cartesian coordinates, computed exactly as host_graph.cpp does
  std::vector<double> reach(3 * (size_t)B), spacing(3 * (size_t)B), cart(3 * (size_t)N);
  std::vector<int> owner(N);
  for (int b = 0; b < B; ++b) {
    const double* L = h->lattice + 9 * b;
    const double *a = L, *bb = L + 3, *c = L + 6;
    const double bc[3] = {bb[1] * c[2] - bb[2] * c[1], bb[2] * c[0] - bb[0] * c[2], bb[0] * c[1] - bb[1] * c[0]};
    const double ca[3] = {c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0]};
    const double ab[3] = {a[1] * bb[2] - a[2] * bb[1], a[2] * bb[0] - a[0] * bb[2], a[0] * bb[1] - a[1] * bb[0]};
    const double vol = a[0] * bc[0] + a[1] * bc[1] + a[2] * bc[2];
    if (!(std::fabs(vol) > 1e-12)) { eng->err = "graph build: singular lattice"; return CHG_EINVAL; }
    const double hk[3] = {std::fabs(vol) / std::sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]),
                          std::fabs(vol) / std::sqrt(ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2]),
                          std::fabs(vol) / std::sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2])};
    for (int k = 0; k < 3; ++k) { reach[3 * b + k] = r_atom / hk[k]; spacing[3 * b + k] = hk[k]; }
    for (int i = h->atom_off[b]; i < h->atom_off[b + 1]; ++i) {
      owner[i] = b;
      for (int k = 0; k < 3; ++k)
        cart[3 * i + k] = h->frac[3 * i] * a[k] + h->frac[3 * i + 1] * bb[k] + h->frac[3 * i + 2] * c[k];
    }
  }
  // cell lists for the large structures (chg_engine_set_graph_search; the rows do not depend on the choice)
  HostCells hcells;
  bool use_cells = false;
  if (eng->graph_search != 1) {
    bin_structures(h, spacing, r_atom, eng->graph_search == 2 ? 0 : eng->cell_min_atoms, hcells);
    use_cells = hcells.any;
  }
  // speculative capacities from the previous build (same cutoffs): per-atom counts + 25 % + a constant
  const bool speculate = eng->spec_builds && eng->last_N > 0 && eng->last_r_atom == r_atom && eng->last_r_bond == r_bond;
  GraphCounts gc;
  int *e_center = nullptr, *e_nbr = nullptr, *e_owner = nullptr, *e_rev = nullptr, *e_d2u = nullptr, *p_center = nullptr, *p_nbr = nullptr,
      *u_u2d = nullptr, *u_bnode = nullptr, *bn_und = nullptr, *a_ctr = nullptr, *a_b1 = nullptr, *a_d1 = nullptr, *a_b2 = nullptr, *a_d2 = nullptr;
  float* e_image = nullptr;
  int *w_na = nullptr, *w_boff = nullptr, *w_aoff = nullptr, *w_toff = nullptr, *w_qa = nullptr, *w_qctr = nullptr, *w_qab1 = nullptr, *w_qab2 = nullptr;
  int *w_toff4 = nullptr, *w_blk_a = nullptr, *w_blk_desc = nullptr;
  int w_blk_cap = 0;
  double *d_cart = nullptr, *d_frac = nullptr, *d_lat = nullptr;
  int *d_owner = nullptr, *d_aoff = nullptr;
  for (int attempt = speculate ? 0 : 1; attempt < 2; ++attempt) {
    TmpPool tmp(eng);
    // one staged upload: the seven input arrays are laid out back to back in a pinned host buffer (doubles first) and
    // travel in a single asynchronous copy; the buffer is free again at the round trip that ends every pass
    const size_t n_f64 = 6 * (size_t)N + 12 * (size_t)B, n_i32 = 2 * (size_t)N + (size_t)B + 1;
    const size_t in_bytes = n_f64 * sizeof(double) + n_i32 * sizeof(int);
    if (in_bytes > eng->h_stage_bytes) {
      if (eng->h_stage) hipHostFree(eng->h_stage);
      eng->h_stage = nullptr; eng->h_stage_bytes = 0;
      const size_t want = in_bytes + in_bytes / 4 + 4096;
      if (hipHostMalloc(&eng->h_stage, want, hipHostMallocDefault) != hipSuccess) { eng->h_stage = nullptr; eng->err = "graph build: pinned staging allocation failed"; return CHG_ENOMEM; }
      eng->h_stage_bytes = want;
    }
    char* d_in = tmp.get<char>(in_bytes);
    if (!d_in) { eng->err = "graph build: scratch allocation failed"; return CHG_ENOMEM; }
    {
      double* hd = reinterpret_cast<double*>(eng->h_stage);
      std::memcpy(hd, cart.data(), sizeof(double) * 3 * N);
      std::memcpy(hd + 3 * (size_t)N, h->frac, sizeof(double) * 3 * N);
      std::memcpy(hd + 6 * (size_t)N, h->lattice, sizeof(double) * 9 * B);
      std::memcpy(hd + 6 * (size_t)N + 9 * (size_t)B, reach.data(), sizeof(double) * 3 * B);
      int* hi = reinterpret_cast<int*>(hd + n_f64);
      std::memcpy(hi, owner.data(), sizeof(int) * N);
      std::memcpy(hi + N, h->atom_off, sizeof(int) * ((size_t)B + 1));
      std::memcpy(hi + N + B + 1, h->z, sizeof(int) * N);
    }
    HIP_TRY(eng, hipMemcpyAsync(d_in, eng->h_stage, in_bytes, hipMemcpyHostToDevice, st));
    d_cart = reinterpret_cast<double*>(d_in);
    d_frac = d_cart + 3 * (size_t)N;
    d_lat = d_cart + 6 * (size_t)N;
    double* d_reach = d_lat + 9 * (size_t)B;
    d_owner = reinterpret_cast<int*>(d_cart + n_f64);
    d_aoff = d_owner + N;
    int* d_z = d_aoff + B + 1;      // every host buffer is consumed before the pass's round trip: nothing of the caller's is read after it
    CellLists cells{};
    if (use_cells) {
      auto up = [&](const std::vector<int>& v) -> const int* {
        int* d = tmp.get<int>(v.size());
        if (d && hipMemcpyAsync(d, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice, st) != hipSuccess) d = nullptr;
        return d;
      };
      cells.off = up(hcells.off); cells.nb = up(hcells.nb); cells.reach = up(hcells.reach); cells.bin_start = up(hcells.bin_start);
      cells.bin_atoms = up(hcells.bin_atoms); cells.bin3 = up(hcells.bin3); cells.shift = up(hcells.shift);
      if (!cells.off || !cells.nb || !cells.reach || !cells.bin_start || !cells.bin_atoms || !cells.bin3 || !cells.shift) {
        eng->err = "graph build: scratch allocation failed";
        return CHG_ENOMEM;
      }
    }
    const bool spec = attempt == 0;
    auto cap = [&](double per_atom) { return (int)std::min<double>(2.0e9, per_atom * N * 1.25 + 4096.0); };
    int capE = spec ? (cap(eng->last_Ed / (double)eng->last_N) & ~1) : 0, capA = spec ? cap(eng->last_A / (double)eng->last_N) : 0,
        capEb = spec ? cap(eng->last_Eb / (double)eng->last_N) : 0;
    bool overflowed = false;
    gc = GraphCounts();
    TRY(graph_pass(eng, tmp, h, d_cart, d_frac, d_lat, d_reach, d_owner, d_aoff, use_cells ? &cells : nullptr, r_atom, r_bond, tol, spec, capE,
                   capA, capEb, gc, overflowed, e_center, e_nbr, e_image, e_owner, e_rev, e_d2u, p_center, p_nbr, u_u2d, u_bnode, bn_und, a_ctr,
                   a_b1, a_d1, a_b2, a_d2, w_na, w_boff, w_aoff, w_toff, w_qa, w_qctr, w_qab1, w_qab2, w_toff4, w_blk_a, w_blk_desc, w_blk_cap));
    if (overflowed && gc.cell_overflow) {   // a centre with more rows than the in-LDS sort holds: same attempt again, all pairs
      use_cells = false;
      eng->n_cell_fallbacks++;
      --attempt;
      continue;
    }
    if (overflowed) { eng->n_spec_overflows++; continue; }   // capacities too small: repeat with the exact, three-round-trip pass
    if (spec) eng->n_spec_builds++;
    if (use_cells) eng->n_cell_builds++;
    const int Ed = gc.Ed, Eu = gc.Ed / 2, A = gc.A, Eb = gc.Eb;
    eng->last_N = N; eng->last_Ed = Ed; eng->last_A = A; eng->last_Eb = Eb; eng->last_r_atom = r_atom; eng->last_r_bond = r_bond;

    // the batch itself: same arena layout as an uploaded batch, filled by device-to-device copies (stream order: no sync)
    chg_batch* b = new (std::nothrow) chg_batch();
    if (!b) return CHG_ENOMEM;
    b->B = B; b->N = N; b->Ed = Ed; b->Eu = Eu; b->A = A; b->Eb = Eb; b->L = eng->desc.n_conv;
    b->canonical = !gc.noncanonical;   // built here: the angle sets are complete n (n - 1) blocks unless the builder saw one of the two exceptions
    // the blocked tiles of the MD-size adjoints were emitted with the angles: the batch uses them (capacity from the exact counts:
    // never more than the builder's own)
    const bool use_blk = w_blk_a && b->canonical && A > 0 && (long)A <= blk_max_angles();
    b->blk_cap = use_blk ? (int)std::min<size_t>(blk_tile_bound(A, Eb, N), (size_t)w_blk_cap) : 0;
    b->blk_ready = use_blk;
    size_t total = 0;
    carve(b, nullptr, total);
    int s = acquire_arena(eng, b, total);
    if (s != CHG_OK) { delete b; return s; }
    carve(b, b->arena, total);
    b->h_atom_off.assign(h->atom_off, h->atom_off + B + 1);
    b->h_volume.resize(B);
    for (int q = 0; q < B; ++q) {   // float32 lattice like k_finalize (model.py:834-836)
      float Lf[9];
      for (int k = 0; k < 9; ++k) Lf[k] = (float)h->lattice[9 * q + k];
      b->h_volume[q] = Lf[0] * (Lf[4] * Lf[8] - Lf[5] * Lf[7]) + Lf[1] * (Lf[5] * Lf[6] - Lf[3] * Lf[8]) + Lf[2] * (Lf[3] * Lf[7] - Lf[4] * Lf[6]);
    }
    {   // every array of the new graph goes into the arena with ONE copy kernel
      MultiCopy mc{};
      int nseg = 0;
      unsigned long long most = 0;
      auto add = [&](void* dst, const void* src, size_t words) {
        if (words == 0) return;
        mc.dst[nseg] = dst; mc.src[nseg] = src; mc.words[nseg] = words;
        most = std::max<unsigned long long>(most, words);
        ++nseg;
      };
      add(b->z, d_z, N); add(b->atom_owner, d_owner, N); add(b->atom_off, d_aoff, (size_t)B + 1);
      add(b->e_center, e_center, Ed); add(b->e_nbr, e_nbr, Ed); add(b->e_d2u, e_d2u, Ed); add(b->e_owner, e_owner, Ed);
      add(b->e_rev, e_rev, Ed); add(b->p_center, p_center, Ed); add(b->p_nbr, p_nbr, Ed); add(b->e_image, e_image, 3 * (size_t)Ed);
      add(b->u_u2d, u_u2d, Eu); add(b->u_bnode, u_bnode, Eu); add(b->bn_und, bn_und, Eb);
      add(b->a_ctr, a_ctr, A); add(b->a_d1, a_d1, A); add(b->a_d2, a_d2, A);
      // the index of the per-atom / team angle adjoints, when the builder emitted it and the batch will use it (decide_windows)
      const bool index_ready = !use_blk && b->canonical && w_qa && A > 0 && decide_windows(eng, b);
      if (index_ready) {
        add(b->win.na, w_na, N); add(b->win.boff, w_boff, (size_t)N + 1); add(b->win.aoff, w_aoff, (size_t)N + 1); add(b->win.toff, w_toff, (size_t)N + 1);
        add(b->win.q_a, w_qa, A); add(b->win.q_ctr, w_qctr, A); add(b->win.q_ab1, w_qab1, A); add(b->win.q_ab2, w_qab2, A);
        mc.q_a_new = w_qa; mc.q_ab1_new = w_qab1; mc.q_b1c = b->win.q_b1c; mc.q_b2c = b->win.q_b2c; mc.abbond = b->win.abbond;
        mc.win_flag = b->win.flag; mc.win_grid = b->win_grid;
      }
      b->win_index_ready = index_ready;
      static_assert(MULTI_COPY_MAX >= 25, "one slot per array");
      // ... and the float32 copies of the coordinates / lattices and the angles' compact bond indices ride in the same launch
      mc.n_copy = nseg;
      mc.cvt_src[0] = d_frac; mc.cvt_dst[0] = b->frac; mc.cvt_n[0] = 3 * N;
      mc.cvt_src[1] = d_lat; mc.cvt_dst[1] = b->lattice; mc.cvt_n[1] = 9 * B;
      mc.a_b1 = a_b1; mc.a_b2 = a_b2; mc.u_bnode_new = u_bnode; mc.a_b1c = b->a_b1c; mc.a_b2c = b->a_b2c; mc.n_ang = A;
      if (use_blk) {
        mc.blk_a_new = w_blk_a; mc.blk_desc_new = w_blk_desc; mc.blk_desc = b->blk_desc; mc.toff4_new = w_toff4; mc.a_ctr_new = a_ctr; mc.n_atoms = N; mc.cap_tiles4 = b->blk_cap;
        mc.blk_a = b->blk_a; mc.blk_b1c = b->blk_b1c; mc.blk_b2c = b->blk_b2c; mc.blk_ctr = b->blk_ctr; mc.blk_tiles = b->blk_tiles;
        most = std::max<unsigned long long>(most, (unsigned long long)b->blk_cap * 64);
      }
      most = std::max<unsigned long long>(most, std::max<unsigned long long>((unsigned long long)A * 4, (unsigned long long)12 * N));
      const unsigned gx = (unsigned)std::min<unsigned long long>((most / 4 + 255) / 256 + 1, (unsigned long long)4 * eng->num_cus);
      hipLaunchKernelGGL(k_multi_copy, dim3(gx, (unsigned)nseg + 4), dim3(256), 0, st, mc);
    }
    if (s == CHG_OK) s = prepare_windows(eng, b);
    // the scratch (TmpPool) is reused by the next build on this same stream, so stream order protects it; overflow
    // allocations of the pool are freed by its destructor and need the copies to have finished
    if (s == CHG_OK && !tmp.extra.empty() && hipStreamSynchronize(st) != hipSuccess) { eng->err = "graph build: synchronisation failed"; s = CHG_EHIP; }
    if (s == CHG_OK && hipGetLastError() != hipSuccess) { eng->err = "graph build: launch failed"; s = CHG_EHIP; }
    if (s != CHG_OK) { hipStreamSynchronize(st); hipFree(b->arena); delete b; return s; }
    if (counts_out) {
      counts_out[0] = Ed; counts_out[1] = Eu; counts_out[2] = A; counts_out[3] = Eb; counts_out[4] = gc.isolated; counts_out[5] = spec ? 1 : 0;
    }
    *out = b;
    return CHG_OK;
  }
  eng->err = "graph build: internal error";
  return CHG_EINVAL;
}


}  // namespace chgh

extern "C" {

int chg_batch_build(chg_engine* eng, const chg_structs_host* h, double r_atom, double r_bond, double numerical_tol, chg_batch** out,
                    int32_t* counts_out) {
  if (!eng || !h || !out || !h->z || !h->frac || !h->lattice || !h->atom_off || h->n_struct <= 0 || h->n_atoms <= 0 || !(r_atom > 0))
    return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  return build_batch_on_device(eng, h, r_atom, r_bond, numerical_tol, out, counts_out);
}

int chg_batch_build_predict(chg_engine* eng, const chg_structs_host* h, double r_atom, double r_bond, double numerical_tol, uint32_t task_mask,
                            chg_batch** out, int32_t* counts_out) {
  int s = chg_batch_build(eng, h, r_atom, r_bond, numerical_tol, out, counts_out);
  if (s != CHG_OK) return s;
  s = chg_predict(eng, *out, task_mask);
  if (s != CHG_OK) { chg_batch_free(eng, *out); *out = nullptr; }
  return s;
}

}  // extern "C"
