// engine.hip -- C-ABI (include/chgnet_hip.h) and launch orchestration of the gfx950 CHGNet engine.
//
// One chg_engine per GPU owns the weight blob and a HIP stream; one chg_batch owns a packed
// batch of structures plus every activation / gradient buffer the forward and reverse sweeps
// need, carved from a single device arena.  chg_predict enqueues the whole E/F/S/M computation
// on the engine's stream without any host synchronisation.
//
// Order of layers follows CHGNet._compute (reference chgnet/model/model.py:442-503); the reverse
// sweep is the hand-derived adjoint of it, producing dE/dv_e once and both
// forces and the virial from it, instead of the reference's two autograd.grad passes
// (model.py:517-535).

#include "engine_internal.h"
#include <cstring>

namespace chgh {

template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {   // UNROLL 16-byte loads in flight per lane
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// Second shape of the same copy: every workgroup owns CONTIGUOUS 16 KiB segments (256 lanes x 4 x 16 B, the four loads of a lane
// 4 KiB apart inside the segment, all in flight before the first store) and walks them grid-stride -- fewer DRAM pages open at a time
// than the fully interleaved form above.  NT: non-temporal accesses (streaming data that nothing reads again) or plain ones.
template <bool NT>
__global__ __launch_bounds__(256) void k_stream_copy_seg(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
  constexpr size_t SEG = 1024;   // f32x4 per segment
  const size_t nseg = n / SEG;
  for (size_t sgm = blockIdx.x; sgm < nseg; sgm += gridDim.x) {
    const f32x4* s = src + sgm * SEG + threadIdx.x;
    f32x4* d = dst + sgm * SEG + threadIdx.x;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(s + u * 256) : s[u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + u * 256); else d[u * 256] = v[u]; }
  }
  for (size_t i = nseg * SEG + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// The tile kernels carry their weights as f16 hi / lo images (mfma_split.h): a weight of magnitude >= 65504 would become inf there.
// No trained checkpoint comes near (|w| < 10); a blob that does is refused instead of producing NaN where the reference is finite.
int check_weight_range(chg_engine* eng, const float* blob, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    if (!(std::fabs(blob[i]) < F16_OPERAND_LIMIT)) {   // also catches NaN
      eng->err = "weights: entry " + std::to_string(i) + " of the blob is " + std::to_string(blob[i]) +
                 ": outside the operand range of the split-precision contractions (|w| < 65504)";
      return CHG_ERANGE;
    }
  }
  return CHG_OK;
}

// ---- self-test of the split-precision contractions (chg_test_split_gemm): the exact device functions of the tile kernels, W [F][64]
// MODE 0 / 1: split images (forward / adjoint of W^T);  MODE 2 / 3: one row-major image, forward / adjoint (mfma_split.h)
template <int MODE, int F>
__global__ __launch_bounds__(BLOCK) void k_test_split(const float* __restrict__ x, const float* __restrict__ W, float* __restrict__ y, int rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool ADJ = (MODE & 1) != 0, RM = MODE >= 2;
  constexpr int KIN = ADJ ? F : D, NOUT = ADJ ? D : F;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
  if (RM) stage_rm(reinterpret_cast<_Float16*>(smem), W, F, D, tid, BLOCK);
  else stage_split<ADJ>(reinterpret_cast<h16x8*>(smem), W, F, D, tid, BLOCK);
  __syncthreads();
  for (int row0 = (blockIdx.x * WAVES + wave) * TILE_ROWS; row0 < rows; row0 += gridDim.x * BLOCK_ROWS) {
    const int row = min(row0 + j, rows - 1);
    f32x4 xin[KIN / 16], acc[NOUT / 16];
    read_dl<KIN / 16>(x + (size_t)row * KIN, g, xin);
#pragma unroll
    for (int q = 0; q < NOUT / 16; ++q) acc[q] = zero4();
    if (RM) gemm_rm<KIN / 16, NOUT / 16, ADJ, ADJ>(acc, reinterpret_cast<const _Float16*>(smem), F, D, xin, j, g, lane);
    else gemm_split<KIN / 16, NOUT / 16, ADJ>(acc, reinterpret_cast<const h16x8*>(smem), NOUT, xin, j, g);
    if (row0 + j < rows) write_dl<NOUT / 16>(y + (size_t)row * NOUT, g, acc);
  }
}

template <int MODE, int F>
void launch_test_split(hipStream_t st, const float* x, const float* W, float* y, int rows) {
  const size_t lds = MODE >= 2 ? rm_image_bytes(F, D) : split_image_bytes(F, D);
  hipLaunchKernelGGL((k_test_split<MODE, F>), dim3(std::max(1, std::min(256, (rows + BLOCK_ROWS - 1) / BLOCK_ROWS))), dim3(BLOCK), lds, st, x, W, y, rows);
}

}  // namespace chgh

// =====================================================================================================
// C-ABI
// =====================================================================================================
extern "C" {

int chg_abi_version(void) { return CHG_ABI_VERSION; }

int chg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int64_t chg_weights_required(int32_t n_conv) {
  if (n_conv < 2 || n_conv > MAX_CONV) return -1;
  Weights probe{};
  return (int64_t)layout_weights(nullptr, n_conv, probe);
}

int chg_engine_create(const chg_model_desc* desc, const float* weights_blob, int device, chg_engine** out) {
  if (!desc || !weights_blob || !out) return CHG_EINVAL;
  if (desc->n_conv < 2 || desc->n_conv > MAX_CONV) return CHG_EUNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CHG_ENODEV;
  chg_engine* eng = new (std::nothrow) chg_engine();
  if (!eng) return CHG_ENOMEM;
  *out = eng;  // returned even on failure so that chg_last_error is readable; destroy it either way
  eng->err = "";   // (a recycled address: this thread's slot may hold an earlier engine's text, engine_internal.h ErrText)
  eng->device = device;
  eng->desc = *desc;
  if (eng->desc.n_mlp_hidden == 0) eng->desc.n_mlp_hidden = 3;
  if (eng->desc.n_mlp_hidden != 2 && eng->desc.n_mlp_hidden != 3) {
    eng->err = "n_mlp_hidden = " + std::to_string(desc->n_mlp_hidden) + ": the energy head has two or three hidden layers";
    return CHG_EUNSUPPORTED;
  }
  if (const char* g = std::getenv("CHGNET_HIP_GRAPHS")) eng->use_graphs = std::string(g) != "0";
  if (const char* g = std::getenv("CHGNET_WIDE_RANGE")) eng->force_wide = std::string(g) == "1";
  if (const char* g = std::getenv("CHGNET_SPEC_BUILD")) eng->spec_builds = std::string(g) != "0";
  HIP_TRY(eng, hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(eng, hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
    eng->err = std::string("device is ") + prop.gcnArchName + ", this library contains gfx950 code only";
    return CHG_ENODEV;
  }
  eng->num_cus = prop.multiProcessorCount;
  HIP_TRY(eng, hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking));
  HIP_TRY(eng, hipStreamCreateWithFlags(&eng->copy_stream, hipStreamNonBlocking));
  HIP_TRY(eng, hipEventCreate(&eng->t0));
  HIP_TRY(eng, hipEventCreate(&eng->t1));
  Weights probe{};
  const size_t need = layout_weights(nullptr, desc->n_conv, probe);
  if ((int64_t)need != desc->n_weights) {
    eng->err = "weight blob has " + std::to_string(desc->n_weights) + " floats, layout needs " + std::to_string(need);
    return CHG_EINVAL;
  }
  HIP_TRY(eng, hipMalloc(&eng->d_weights, need * sizeof(float)));
  if (int rs = check_weight_range(eng, weights_blob, need); rs != CHG_OK) { eng->err = "chg_engine_create: " + eng->err; *out = eng; return rs; }
  HIP_TRY(eng, hipMemcpy(eng->d_weights, weights_blob, need * sizeof(float), hipMemcpyHostToDevice));
  layout_weights(eng->d_weights, desc->n_conv, eng->w);
  { const int si = build_images(eng); if (si) return si; }
  // kernels that need more than the default 64 KiB of dynamic LDS: every unit sets the attributes of the kernels it launches
  { int s; if ((s = predict_set_lds(eng)) || (s = chgh_wide::predict_set_lds(eng)) || (s = train_set_lds(eng)) || (s = chgh_wide::train_set_lds(eng))) return s; }
  return CHG_OK;
}

int chg_engine_destroy(chg_engine* eng) {
  if (!eng) return CHG_OK;
  hipSetDevice(eng->device);
  if (eng->stream) hipStreamSynchronize(eng->stream);
  for (auto& pe : eng->pending) { hipEventDestroy(pe.start); hipEventDestroy(pe.stop); }
  for (auto e : eng->event_pool) hipEventDestroy(e);
  if (eng->t0) hipEventDestroy(eng->t0);
  if (eng->t1) hipEventDestroy(eng->t1);
  for (auto& a : eng->arena_pool) hipFree(a.first);
  for (auto& a : eng->work_pool) hipFree(a.first);
  if (eng->scratch) hipFree(eng->scratch);
  if (eng->h_stage) hipHostFree(eng->h_stage);
  if (eng->h_out) hipHostFree(eng->h_out);
  if (eng->d_weights) hipFree(eng->d_weights);
  if (eng->d_images) hipFree(eng->d_images);
  if (eng->copy_stream) hipStreamDestroy(eng->copy_stream);
  if (eng->stream) hipStreamDestroy(eng->stream);
  delete eng;
  return CHG_OK;
}

const char* chg_last_error(const chg_engine* eng) { return eng ? eng->err.c_str() : "null engine"; }

int chg_engine_build_stats(chg_engine* eng, int64_t* single_pass_builds, int64_t* capacity_overflows) {
  if (!eng) return CHG_EINVAL;
  if (single_pass_builds) *single_pass_builds = eng->n_spec_builds;
  if (capacity_overflows) *capacity_overflows = eng->n_spec_overflows;
  return CHG_OK;
}

int chg_engine_set_graph_search(chg_engine* eng, int32_t search, int32_t cell_min_atoms) {
  if (!eng || search < 0 || search > 2 || cell_min_atoms < 0) return CHG_EINVAL;
  eng->graph_search = search;
  if (cell_min_atoms > 0) eng->cell_min_atoms = cell_min_atoms;
  return CHG_OK;
}

int chg_engine_cell_stats(chg_engine* eng, int64_t* cell_builds, int64_t* all_pairs_fallbacks) {
  if (!eng) return CHG_EINVAL;
  if (cell_builds) *cell_builds = eng->n_cell_builds;
  if (all_pairs_fallbacks) *all_pairs_fallbacks = eng->n_cell_fallbacks;
  return CHG_OK;
}

int chg_engine_set_memory_limit(chg_engine* eng, int64_t bytes) {
  if (!eng || bytes < 0) return CHG_EINVAL;
  eng->memory_limit = (size_t)bytes;
  return CHG_OK;
}

int chg_engine_memory_info(chg_engine* eng, int64_t* free_bytes, int64_t* total_bytes) {
  if (!eng) return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  size_t f = 0, t = 0;
  HIP_TRY(eng, hipMemGetInfo(&f, &t));
  std::lock_guard<std::mutex> lk(eng->pool_mu);
  for (auto& a : eng->arena_pool) f += a.second;   // pooled arenas are reusable
  for (auto& a : eng->work_pool) f += a.second;
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return CHG_OK;
}

int64_t chg_batch_bytes_required(int32_t n_conv, int32_t n_struct, int32_t n_atoms, int32_t n_directed, int32_t n_angles, int32_t n_bnodes) {
  if (n_conv < 2 || n_conv > MAX_CONV || n_struct < 0 || n_atoms < 0 || n_directed < 0 || (n_directed & 1) || n_angles < 0 || n_bnodes < 0) return -1;
  chg_batch probe{};
  probe.B = n_struct; probe.N = n_atoms; probe.Ed = n_directed; probe.Eu = n_directed / 2; probe.A = n_angles; probe.Eb = n_bnodes; probe.L = n_conv;
  probe.blk_cap = upload_blk_cap(probe.N, probe.A, probe.Eb);
  size_t total = 0;
  carve(&probe, nullptr, total);
  return (int64_t)total;
}

int chg_batch_upload(chg_engine* eng, const chg_batch_host* h, chg_batch** out) {
  if (!eng || !h || !out) return CHG_EINVAL;
  if (h->n_struct <= 0 || h->n_atoms <= 0 || h->n_directed < 0 || h->n_directed != 2 * h->n_undirected || h->n_angles < 0 ||
      h->n_bnodes < 0 || (h->n_angles > 0 && h->n_bnodes == 0)) {
    eng->err = "chg_batch_upload: inconsistent counts";
    return CHG_EINVAL;
  }
  HIP_TRY(eng, hipSetDevice(eng->device));
  chg_batch* b = new (std::nothrow) chg_batch();
  if (!b) return CHG_ENOMEM;
  b->B = h->n_struct; b->N = h->n_atoms; b->Ed = h->n_directed; b->Eu = h->n_undirected; b->A = h->n_angles; b->Eb = h->n_bnodes;
  b->L = eng->desc.n_conv;
  b->blk_cap = upload_blk_cap(b->N, b->A, b->Eb);      // small batches: the angle adjoints over blocked tiles (index: prepare_windows)
  size_t total = 0;
  carve(b, nullptr, total);
  {
    const int sa = acquire_arena(eng, b, total);
    if (sa != CHG_OK) { delete b; return sa; }
  }
  carve(b, b->arena, total);
  // (debug names: registered on the first chg_debug_fetch -- ~70 map insertions per batch are 10 us of an MD step)
  if (h->atom_off) b->h_atom_off.assign(h->atom_off, h->atom_off + h->n_struct + 1);
  b->h_volume.resize(h->n_struct);
  for (int q = 0; q < h->n_struct; ++q) {
    const float* Lf = h->lattice + 9 * q;
    b->h_volume[q] = Lf[0] * (Lf[4] * Lf[8] - Lf[5] * Lf[7]) + Lf[1] * (Lf[5] * Lf[6] - Lf[3] * Lf[8]) + Lf[2] * (Lf[3] * Lf[7] - Lf[4] * Lf[6]);
  }
  int s = CHG_OK;
  const size_t B = b->B, N = b->N, Ed = b->Ed, Eu = b->Eu, A = b->A, Eb = b->Eb;
#define UP(field, n) if (s == CHG_OK) s = h2d(eng, b->field, h->field, (n), eng->copy_stream)
  UP(z, N); UP(atom_owner, N); UP(atom_off, B + 1); UP(frac, 3 * N); UP(lattice, 9 * B);
  UP(e_center, Ed); UP(e_nbr, Ed); UP(e_d2u, Ed); UP(e_owner, Ed); UP(e_image, 3 * Ed); UP(e_rev, Ed); UP(p_center, Ed); UP(p_nbr, Ed);
  UP(u_u2d, Eu); UP(u_bnode, Eu); UP(bn_und, Eb);
  UP(a_ctr, A); UP(a_b1c, A); UP(a_b2c, A); UP(a_d1, A); UP(a_d2, A);
#undef UP
  // (copy stream only: the caller may be a loader thread next to a running sweep -- nothing is launched on the compute stream here)
  b->win_pending = true;
  if (s == CHG_OK && hipStreamSynchronize(eng->copy_stream) != hipSuccess) { eng->err = "chg_batch_upload: sync failed"; s = CHG_EHIP; }
  if (s != CHG_OK) { hipFree(b->arena); delete b; return s; }
  *out = b;
  return CHG_OK;
}

int chg_debug_fetch_i32(chg_engine* eng, chg_batch* b, const char* name, int32_t* dst, int64_t capacity, int64_t* n_written) {
  if (!eng || !b || !name || !dst) return CHG_EINVAL;
  if (std::strcmp(name, "wide_range") == 0) {   // host-side state: 1 = the batch runs on the wide-range sweeps (engine_*_wide.hip)
    if (capacity >= 1) dst[0] = b->wide_range ? 1 : 0;
    if (n_written) *n_written = capacity >= 1 ? 1 : 0;
    return CHG_OK;
  }
  if (b->named_i32.empty()) register_names(b);
  auto it = b->named_i32.find(name);
  if (it == b->named_i32.end()) { eng->err = std::string("chg_debug_fetch_i32: unknown buffer ") + name; return CHG_EINVAL; }
  const size_t n = std::min<size_t>(it->second.second, (size_t)std::max<int64_t>(capacity, 0));
  TRY(ensure_windows(eng, b));
  HIP_TRY(eng, hipStreamSynchronize(eng->stream));
  if (n) HIP_TRY(eng, hipMemcpy(dst, it->second.first, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (n_written) *n_written = (int64_t)n;
  return CHG_OK;
}

int chg_batch_update_geometry(chg_engine* eng, chg_batch* b, const float* frac, const float* lattice) {
  if (!eng || !b) return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  int s = CHG_OK;
  if (frac) s = h2d(eng, b->frac, frac, (size_t)3 * b->N);
  if (s == CHG_OK && lattice) {
    s = h2d(eng, b->lattice, lattice, (size_t)9 * b->B);
    b->h_volume.resize(b->B);      // chg_backward scales the stress cotangent with the host copy of the cell volumes
    for (int q = 0; q < b->B; ++q) {
      const float* Lf = lattice + 9 * (size_t)q;
      b->h_volume[q] = Lf[0] * (Lf[4] * Lf[8] - Lf[5] * Lf[7]) + Lf[1] * (Lf[5] * Lf[6] - Lf[3] * Lf[8]) + Lf[2] * (Lf[3] * Lf[7] - Lf[4] * Lf[6]);
    }
  }
  b->q_tables = false;             // tables of the previous geometry
  if (s == CHG_OK) HIP_TRY(eng, hipStreamSynchronize(eng->stream));
  return s;
}

int chg_batch_free(chg_engine* eng, chg_batch* b) {
  if (!b) return CHG_OK;
  if (eng) { hipSetDevice(eng->device); hipStreamSynchronize(eng->stream); }
  if (b->graph_exec) hipGraphExecDestroy(b->graph_exec);
  release_workspace(eng, b->train_arena, b->train_bytes, 0);
  release_workspace(eng, b->t2_arena, b->t2_bytes, 1);
  free_train2(b);
  if (b->arena) {
    bool pooled = false;
    if (eng) {
      std::lock_guard<std::mutex> lk(eng->pool_mu);
      if (eng->arena_pool.size() < 2) { eng->arena_pool.emplace_back(b->arena, b->arena_bytes); pooled = true; }
    }
    if (!pooled) hipFree(b->arena);
  }
  delete b;
  return CHG_OK;
}

int64_t chg_batch_device_bytes(const chg_batch* b) { return b ? (int64_t)b->arena_bytes : 0; }

int chg_predict(chg_engine* eng, chg_batch* b, uint32_t task_mask) {
  if (!eng || !b) return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  const uint32_t task = task_mask | CHG_TASK_E;
  TRY(ensure_windows(eng, b));
  if (eng->force_wide) b->wide_range = true;
  if (b->wide_range) return chgh_wide::run_predict(eng, b, task);   // an earlier prediction of this batch overflowed the f16 operands
  if (eng->profiling || !eng->use_graphs) return run_predict(eng, b, task);   // per-kernel events need eager launches
  if (b->graph_task != task) {
    if (b->graph_exec) hipGraphExecDestroy(b->graph_exec);
    b->graph_exec = nullptr;
    b->graph_task = task;
    b->eager_calls = 0;
  }
  if (!b->graph_exec && b->eager_calls++ == 0) return run_predict(eng, b, task);
  if (!b->graph_exec) {
    hipGraph_t graph = nullptr;
    HIP_TRY(eng, hipStreamBeginCapture(eng->stream, hipStreamCaptureModeThreadLocal));
    const int s = run_predict(eng, b, task);
    const hipError_t e = hipStreamEndCapture(eng->stream, &graph);
    if (s != CHG_OK) { if (graph) hipGraphDestroy(graph); return s; }
    if (e != hipSuccess) { eng->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); return CHG_EHIP; }
    const hipError_t ei = hipGraphInstantiate(&b->graph_exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (ei != hipSuccess) { b->graph_exec = nullptr; eng->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(ei); return CHG_EHIP; }
    b->graph_task = task;
  }
  HIP_TRY(eng, hipGraphLaunch(b->graph_exec, eng->stream));
  b->last_task = task;
  b->seed1_adjoints = (task & (CHG_TASK_F | CHG_TASK_S)) != 0;
  b->q_tables = b->seed1_adjoints;
  return CHG_OK;
}

int chg_engine_update_weights(chg_engine* eng, const float* weights_blob) {
  if (!eng || !weights_blob) return CHG_EINVAL;
  TRY(check_weight_range(eng, weights_blob, (size_t)eng->desc.n_weights));
  HIP_TRY(eng, hipSetDevice(eng->device));
  HIP_TRY(eng, hipMemcpyAsync(eng->d_weights, weights_blob, sizeof(float) * (size_t)eng->desc.n_weights, hipMemcpyHostToDevice, eng->stream));
  TRY(build_images(eng));
  HIP_TRY(eng, hipStreamSynchronize(eng->stream));
  return CHG_OK;
}

int chg_batch_all_gather_energy(chg_engine* eng, chg_batch* b, chg_comm* comm, int64_t width, float* table) {
  if (!eng || !b || !comm || !table || width < b->B) return CHG_EINVAL;
  if (b->last_task == 0) { eng->err = "chg_batch_all_gather_energy: run chg_predict on this batch first"; return CHG_EINVAL; }
  HIP_TRY(eng, hipSetDevice(eng->device));
  int32_t nranks = 1, comm_dev = -1;
  float* stage = nullptr;
  if (chg_comm_info(comm, nullptr, &nranks, &comm_dev) == CHG_OK && comm_dev != eng->device) {
    eng->err = "chg_batch_all_gather_energy: the communicator belongs to device " + std::to_string(comm_dev) + ", the engine to device " + std::to_string(eng->device);
    return CHG_EINVAL;
  }
  if (chg_comm_info(comm, nullptr, &nranks, nullptr) != CHG_OK || chg_comm_reserve(comm, width * (int64_t)(nranks + 1), &stage) != CHG_OK) {
    eng->err = std::string("chg_batch_all_gather_energy: ") + chg_comm_last_error(comm);
    return CHG_EHIP;
  }
  hipStream_t st = eng->stream;
  HIP_TRY(eng, hipMemsetAsync(stage, 0, sizeof(float) * (size_t)width, st));
  HIP_TRY(eng, hipMemcpyAsync(stage, b->energy, sizeof(float) * (size_t)b->B, hipMemcpyDeviceToDevice, st));
  if (chg_comm_all_gather_f32_device(comm, stage, width, stage + width, st) != CHG_OK) {
    eng->err = std::string("chg_batch_all_gather_energy: ") + chg_comm_last_error(comm);
    return CHG_EHIP;
  }
  HIP_TRY(eng, hipMemcpyAsync(table, stage + width, sizeof(float) * (size_t)width * nranks, hipMemcpyDeviceToHost, st));
  HIP_TRY(eng, hipStreamSynchronize(st));
  return CHG_OK;
}

void* chg_engine_stream(chg_engine* eng) { return eng ? static_cast<void*>(eng->stream) : nullptr; }
int chg_engine_device(chg_engine* eng) { return eng ? eng->device : -1; }

int chg_synchronize(chg_engine* eng) {
  if (!eng) return CHG_EINVAL;
  HIP_TRY(eng, hipStreamSynchronize(eng->stream));
  if (eng->profiling) return collect_profile(eng);
  return CHG_OK;
}

int chg_batch_download(chg_engine* eng, chg_batch* b, const chg_out_host* o) {
  if (!eng || !b || !o) return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  const size_t B = b->B, N = b->N;
  struct Piece { float* dst; const float* src; size_t n; };
  const Piece pieces[] = {
      {o->energy, b->energy, B},
      {(b->last_task & CHG_TASK_F) ? o->force : nullptr, b->force, 3 * N},
      {(b->last_task & CHG_TASK_S) ? o->stress : nullptr, b->virial, 9 * B},
      {(b->last_task & CHG_TASK_M) ? o->magmom : nullptr, b->magmom, N},
      {o->site_energy, b->site_energy, N},
      {o->atom_fea, b->atom[b->L - 1], N * D},
      {o->crystal_fea, b->crystal_fea, B * D}};
  size_t total = 0;
  for (const Piece& pc : pieces) if (pc.dst) total += pc.n;
  auto copy_out = [&]() -> int {
    // Small results (MD-size batches: 4-7 pieces of a few KB) go through pinned staging: a copy to pageable memory is staged and waited
    // for piece by piece (~20 us of idle GPU each, 6 % of a 256-atom MD step); into pinned memory the copies queue back to back.
    constexpr size_t PINNED_MAX = (size_t)4 << 20;   // floats (16 MB)
    if (total > 0 && total <= PINNED_MAX) {
      if ((total + 1024) * sizeof(float) > eng->h_out_bytes) {
        if (eng->h_out) hipHostFree(eng->h_out);
        eng->h_out = nullptr; eng->h_out_bytes = 0;
        const size_t want = std::max((total + 1024) * sizeof(float) * 2, (size_t)1 << 16);
        if (hipHostMalloc(&eng->h_out, want, hipHostMallocDefault) != hipSuccess) { eng->h_out = nullptr; eng->err = "chg_batch_download: pinned staging allocation failed"; return CHG_ENOMEM; }
        eng->h_out_bytes = want;
      }
      float* stage = reinterpret_cast<float*>(eng->h_out);
      // pieces that are neighbours in the arena (engine_predict.hip carve: crystal_fea | force | virial | energy | magmom) travel as ONE
      // copy of their span when the gaps are small
      const float *lo = nullptr, *hi = nullptr;
      for (const Piece& pc : pieces)
        if (pc.dst && pc.n) { lo = (!lo || pc.src < lo) ? pc.src : lo; hi = (!hi || pc.src + pc.n > hi) ? pc.src + pc.n : hi; }
      const size_t span = lo ? (size_t)(hi - lo) : 0;
      if (span <= total + 1024 && span * sizeof(float) <= eng->h_out_bytes) {
        HIP_TRY(eng, hipMemcpyAsync(stage, lo, span * sizeof(float), hipMemcpyDeviceToHost, eng->stream));
        TRY(chg_synchronize(eng));
        for (const Piece& pc : pieces)
          if (pc.dst && pc.n) std::memcpy(pc.dst, stage + (pc.src - lo), pc.n * sizeof(float));
      } else {
        size_t off = 0;
        for (const Piece& pc : pieces)
          if (pc.dst && pc.n) { HIP_TRY(eng, hipMemcpyAsync(stage + off, pc.src, pc.n * sizeof(float), hipMemcpyDeviceToHost, eng->stream)); off += pc.n; }
        TRY(chg_synchronize(eng));
        off = 0;
        for (const Piece& pc : pieces)
          if (pc.dst && pc.n) { std::memcpy(pc.dst, stage + off, pc.n * sizeof(float)); off += pc.n; }
      }
    } else {
      int s = CHG_OK;
      for (const Piece& pc : pieces)
        if (s == CHG_OK) s = d2h(eng, pc.dst, pc.src, pc.n);
      if (s != CHG_OK) return s;
      TRY(chg_synchronize(eng));
    }
    return CHG_OK;
  };
  auto all_finite = [&]() {
    for (int q = 0; q < 4; ++q)    // energy, forces, stress, magmoms as requested
      if (pieces[q].dst)
        for (size_t i = 0; i < pieces[q].n; ++i)
          if (!std::isfinite(pieces[q].dst[i])) return false;
    return true;
  };
  TRY(copy_out());
  // Non-finite results.  The reference gives them too for coincident atoms (1 / r of a zero-length bond): passed through.  But an
  // activation past the f16 operand range of the split contractions (|x| >= 65504, mfma_split.h) ALSO ends as inf / NaN here where the
  // reference's fp32 path (crystalgraph.py:12 TORCH_DTYPE) stays finite -- an overflow cannot stay silent: an inf operand makes the
  // accumulator inf / NaN, LayerNorm spreads it over the row, the sums carry it to the energy.  Such a batch is COMPUTED AGAIN by the
  // wide-range sweep (engine_predict_wide.hip: every operand row scaled by a power of two before the split, exact, any fp32 magnitude)
  // and stays on it; what is still non-finite then is non-finite in the reference as well.  The check costs nothing until it triggers.
  if (!b->wide_range && b->last_task != 0 && !all_finite()) {
    b->wide_range = true;
    if (b->graph_exec) { hipGraphExecDestroy(b->graph_exec); b->graph_exec = nullptr; }
    TRY(chgh_wide::run_predict(eng, b, b->last_task));
    TRY(copy_out());
  }
  return CHG_OK;
}

// Page-locked host memory for the caller's packed batches: chg_batch_upload's copies from it are true asynchronous DMA at the link
// rate (~250 MB per 1024-structure batch: 5 ms instead of the 22 ms of a staged copy from pageable memory).
int chg_host_alloc(int64_t bytes, void** out) {
  if (bytes <= 0 || !out) return CHG_EINVAL;
  *out = nullptr;
  // portable: the caller may be a loader thread whose current device is not the engine's (rank k > 0 of a multi-GPU node: a new
  // thread starts on device 0), and the block is then read by that engine's copy stream
  if (hipHostMalloc(out, (size_t)bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return CHG_ENOMEM; }
  return CHG_OK;
}
int chg_host_free(void* p) {
  if (p && hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return CHG_EHIP; }
  return CHG_OK;
}

int chg_timer_start(chg_engine* eng) {
  if (!eng) return CHG_EINVAL;
  HIP_TRY(eng, hipEventRecord(eng->t0, eng->stream));
  return CHG_OK;
}

int chg_timer_stop_ms(chg_engine* eng, float* elapsed_ms) {
  if (!eng || !elapsed_ms) return CHG_EINVAL;
  HIP_TRY(eng, hipEventRecord(eng->t1, eng->stream));
  HIP_TRY(eng, hipEventSynchronize(eng->t1));
  HIP_TRY(eng, hipEventElapsedTime(elapsed_ms, eng->t0, eng->t1));
  return CHG_OK;
}

int chg_profile_enable(chg_engine* eng, int on) {
  if (!eng) return CHG_EINVAL;
  HIP_TRY(eng, hipStreamSynchronize(eng->stream));
  if (eng->profiling) collect_profile(eng);
  eng->profiling = on != 0;
  return CHG_OK;
}

int chg_profile_reset(chg_engine* eng) {
  if (!eng) return CHG_EINVAL;
  for (auto& p : eng->prof) { p.launches = 0; p.ms = 0.0; }
  return CHG_OK;
}

int chg_profile_count(chg_engine* eng) { return eng ? (int)eng->prof.size() : 0; }

int chg_profile_read(chg_engine* eng, int i, char* label, int label_cap, int64_t* launches, double* total_ms) {
  if (!eng || i < 0 || i >= (int)eng->prof.size()) return CHG_EINVAL;
  const ProfEntry& p = eng->prof[i];
  if (label && label_cap > 0) { std::strncpy(label, p.label.c_str(), label_cap - 1); label[label_cap - 1] = 0; }
  if (launches) *launches = p.launches;
  if (total_ms) *total_ms = p.ms;
  return CHG_OK;
}

int chg_debug_fetch(chg_engine* eng, chg_batch* b, const char* name, float* dst, int64_t capacity, int64_t* n_written) {
  if (!eng || !b || !name || !dst) return CHG_EINVAL;
  if (b->named.empty()) register_names(b);
  auto it = b->named.find(name);
  if (it == b->named.end()) { eng->err = std::string("chg_debug_fetch: unknown buffer ") + name; return CHG_EINVAL; }
  const size_t n = std::min<size_t>(it->second.second, (size_t)std::max<int64_t>(capacity, 0));
  HIP_TRY(eng, hipStreamSynchronize(eng->stream));
  if (n) HIP_TRY(eng, hipMemcpy(dst, it->second.first, n * sizeof(float), hipMemcpyDeviceToHost));
  if (n_written) *n_written = (int64_t)n;
  return CHG_OK;
}

// STREAM-like copy (read + write of `bytes` each) on the engine's stream: the measured HBM ceiling that the
// HBM-bound kernels of the path are reported against (bench.py roofline_hbm).  Several kernel shapes are tried and
// the fastest is reported, so that the ceiling is not an artefact of one launch geometry.
int chg_stream_copy(chg_engine* eng, int64_t bytes, int iters, float* ms_per_iter) {
  if (!eng || bytes < 4096 || iters <= 0 || !ms_per_iter) return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  char *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, (size_t)bytes) != hipSuccess) { eng->err = "chg_stream_copy: allocation failed"; return CHG_ENOMEM; }
  if (hipMalloc(&b, (size_t)bytes) != hipSuccess) { hipFree(a); eng->err = "chg_stream_copy: allocation failed"; return CHG_ENOMEM; }
  const size_t n = (size_t)bytes / sizeof(f32x4);
  hipMemsetAsync(a, 1, (size_t)bytes, eng->stream);
  hipMemsetAsync(b, 2, (size_t)bytes, eng->stream);
  hipEvent_t e0 = get_event(eng), e1 = get_event(eng);
  int s = CHG_OK;
  float best = 1e30f;
  for (int variant = 0; variant < 12 && s == CHG_OK; ++variant) {
    const int shape = variant / 3;   // 0: interleaved x1, 1: interleaved x4, 2: 16 KiB segments non-temporal, 3: segments, plain accesses
    const unsigned blocks = (unsigned)((variant % 3 == 0 ? 8 : (variant % 3 == 1 ? 32 : 128)) * eng->num_cus);
    auto launch = [&]() {
      if (shape == 0) hipLaunchKernelGGL(k_stream_copy<1>, dim3(blocks), dim3(256), 0, eng->stream, (const f32x4*)a, (f32x4*)b, n);
      else if (shape == 1) hipLaunchKernelGGL(k_stream_copy<4>, dim3(blocks), dim3(256), 0, eng->stream, (const f32x4*)a, (f32x4*)b, n);
      else if (shape == 2) hipLaunchKernelGGL(k_stream_copy_seg<true>, dim3(blocks), dim3(256), 0, eng->stream, (const f32x4*)a, (f32x4*)b, n);
      else hipLaunchKernelGGL(k_stream_copy_seg<false>, dim3(blocks), dim3(256), 0, eng->stream, (const f32x4*)a, (f32x4*)b, n);
    };
    launch();   // warm-up (page faults, clocks)
    hipEventRecord(e0, eng->stream);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1, eng->stream);
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { eng->err = "chg_stream_copy: timing failed"; s = CHG_EHIP; }
    best = std::min(best, ms / iters);
  }
  eng->event_pool.push_back(e0); eng->event_pool.push_back(e1);
  hipFree(a); hipFree(b);
  *ms_per_iter = best;
  return s;
}

int chg_test_rows_gemm(chg_engine* eng, const float* x, const float* wt, const float* bias, float* y, int rows, int k, int nout) {
  if (!eng || !x || !wt || !y || rows <= 0) return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  float *dx = nullptr, *dw = nullptr, *db = nullptr, *dy = nullptr;
  HIP_TRY(eng, hipMalloc(&dx, sizeof(float) * (size_t)rows * k));
  HIP_TRY(eng, hipMalloc(&dw, sizeof(float) * (size_t)nout * k));
  HIP_TRY(eng, hipMalloc(&db, sizeof(float) * (size_t)nout));
  HIP_TRY(eng, hipMalloc(&dy, sizeof(float) * (size_t)rows * nout));
  HIP_TRY(eng, hipMemcpy(dx, x, sizeof(float) * (size_t)rows * k, hipMemcpyHostToDevice));
  HIP_TRY(eng, hipMemcpy(dw, wt, sizeof(float) * (size_t)nout * k, hipMemcpyHostToDevice));
  if (bias) HIP_TRY(eng, hipMemcpy(db, bias, sizeof(float) * (size_t)nout, hipMemcpyHostToDevice));
  int s = rows_gemm(eng, "test_gemm", k, nout, dx, k, nullptr, dw, bias ? db : nullptr, nullptr, 0, dy, nout, nullptr, rows, 0);
  if (s == CHG_OK && hipStreamSynchronize(eng->stream) != hipSuccess) { eng->err = "test gemm failed"; s = CHG_EHIP; }
  if (s == CHG_OK) hipMemcpy(y, dy, sizeof(float) * (size_t)rows * nout, hipMemcpyDeviceToHost);
  hipFree(dx); hipFree(dw); hipFree(db); hipFree(dy);
  return s;
}

int chg_test_split_gemm(chg_engine* eng, const float* x, const float* w, float* y, int rows, int f, int mode) {
  if (!eng || !x || !w || !y || rows <= 0 || (f != D && f != 2 * D) || mode < 0 || mode > 3) return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  const int kin = (mode & 1) ? f : D, nout = (mode & 1) ? D : f;
  float *dx = nullptr, *dw = nullptr, *dy = nullptr;
  HIP_TRY(eng, hipMalloc(&dx, sizeof(float) * (size_t)rows * kin));
  HIP_TRY(eng, hipMalloc(&dw, sizeof(float) * (size_t)f * D));
  HIP_TRY(eng, hipMalloc(&dy, sizeof(float) * (size_t)rows * nout));
  HIP_TRY(eng, hipMemcpy(dx, x, sizeof(float) * (size_t)rows * kin, hipMemcpyHostToDevice));
  HIP_TRY(eng, hipMemcpy(dw, w, sizeof(float) * (size_t)f * D, hipMemcpyHostToDevice));
  hipStream_t st = eng->stream;
  switch (mode * 2 + (f == D ? 0 : 1)) {
    case 0: launch_test_split<0, D>(st, dx, dw, dy, rows); break;
    case 1: launch_test_split<0, 2 * D>(st, dx, dw, dy, rows); break;
    case 2: launch_test_split<1, D>(st, dx, dw, dy, rows); break;
    case 3: launch_test_split<1, 2 * D>(st, dx, dw, dy, rows); break;
    case 4: launch_test_split<2, D>(st, dx, dw, dy, rows); break;
    case 5: launch_test_split<2, 2 * D>(st, dx, dw, dy, rows); break;
    case 6: launch_test_split<3, D>(st, dx, dw, dy, rows); break;
    default: launch_test_split<3, 2 * D>(st, dx, dw, dy, rows); break;
  }
  int s = CHG_OK;
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { eng->err = "chg_test_split_gemm failed"; s = CHG_EHIP; }
  if (s == CHG_OK) hipMemcpy(y, dy, sizeof(float) * (size_t)rows * nout, hipMemcpyDeviceToHost);
  hipFree(dx); hipFree(dw); hipFree(dy);
  return s;
}

}  // extern "C"
