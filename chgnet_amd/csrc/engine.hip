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

#include "chgnet_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "kernels_conv.h"
#include "kernels_angle_w.h"
#include "kernels_embed.h"
#include "kernels_geom.h"
#include "kernels_graph.h"
#include "kernels_train.h"
#include "kernels_train2.h"
#include "kernels_train2_tile.h"


using namespace chg;

namespace {

constexpr int MAX_CONV = 8;
constexpr float F16_OPERAND_LIMIT = 65504.0f;   // largest finite f16: forward operands of the split contractions are not rescaled
#ifndef CHG_FWD_WAVES
#define CHG_FWD_WAVES 8
#endif
#ifdef CHG_PHASE_TIMING
constexpr size_t PHASE_FLOATS = (size_t)4 * 2 * 10 * PH_WAVES;   // kernels_conv.h PH_FLUSH
#else
constexpr size_t PHASE_FLOATS = 64;
#endif
constexpr int FWD_WAVES = CHG_FWD_WAVES;   // waves per workgroup of the light forward kernels (12 = 3 per SIMD measured no better: profiles notes)

struct ACW { const float *w_cn, *w_bond, *b1, *q_bias; GatedW g; const float *w2c_t, *w2g_t, *w_out, *b_out, *w_out_t, *w_cn_t, *w_bond_t; };
struct BCW { const float *w_bij, *w_ang, *w_ctr, *b1; GatedW g; const float *w2c_t, *w2g_t, *w_out, *b_out, *w_out_t, *w_bij_t, *w_ang_t, *w_ctr_t; };
struct AUW { const float *w_bij, *w_ang, *w_ctr, *b1; GatedW g; const float *w_bij_t, *w_ang_t, *w_ctr_t; };

struct Weights {
  const float *atomref, *emb, *freq_ag, *freq_bg, *freq_ang, *w_bond_emb, *w_wag, *w_wbg, *w_ang_emb;
  ACW ac[MAX_CONV];
  BCW bc[MAX_CONV];
  AUW au[MAX_CONV];
  const float *site_w, *site_b, *ro_ln_g, *ro_ln_b, *mlp_w0, *mlp_b0, *mlp_w1, *mlp_b1, *mlp_w2, *mlp_b2, *mlp_w3, *mlp_b3;
  const float *mlp_w0_t, *mlp_w1_t, *mlp_w2_t;
};

struct ProfEntry { std::string label; int64_t launches = 0; double ms = 0.0; };
struct PendingEvent { int entry; hipEvent_t start, stop; };

}  // namespace

struct chg_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  chg_model_desc desc{};
  float* d_weights = nullptr;
  Weights w{};
  // prebuilt LDS weight blocks of the inference tile kernels (kernels_conv.h k_*_image), rebuilt by every weight upload
  float* d_images = nullptr;
  const float* img_ac_fwd[2][MAX_CONV] = {};   // [without / with q_bias][layer]
  const float* img_ac_bwd[MAX_CONV] = {};
  const float* img_angle[2][2 * MAX_CONV] = {};   // [fwd / bwd][slot: BondConv l | L + AngleUpdate l]
  std::string err;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool profiling = false;
  std::vector<ProfEntry> prof;
  std::map<std::string, int> prof_index;
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
  std::vector<std::pair<char*, size_t>> arena_pool;   // released batch arenas, reused by later uploads
  std::vector<std::pair<char*, size_t>> work_pool;    // released training workspaces (tens of GB: a hipMalloc per step would dominate it)
  std::vector<int> work_kind;                         // 0: first-order workspace, 1: second-order workspace
  bool use_graphs = true;   // CHGNET_HIP_GRAPHS=0 forces eager launches
  char* scratch = nullptr;  // grow-only scratch of chg_batch_build (MD rebuilds the graph every step)
  size_t scratch_bytes = 0, scratch_wanted = 0;
  char* h_stage = nullptr;   // pinned staging for the inputs of chg_batch_build
  size_t h_stage_bytes = 0;
  int num_cus = 256;
  // single-pass graph builds (chg_batch_build): counts of the previous build size the next one's scratch speculatively
  bool spec_builds = true;    // CHGNET_SPEC_BUILD=0 forces the exact three-round-trip pass
  int last_N = 0, last_Ed = 0, last_A = 0, last_Eb = 0;
  double last_r_atom = 0.0, last_r_bond = 0.0;
  long n_spec_builds = 0, n_spec_overflows = 0, n_cell_builds = 0, n_cell_fallbacks = 0;
  int graph_search = 0;       // chg_engine_set_graph_search: 0 by size, 1 all pairs, 2 cell list
  int cell_min_atoms = 512;   // structures at least this large are binned (by size)
  size_t memory_limit = 0;  // chg_engine_set_memory_limit: arenas larger than this are refused with CHG_ENOMEM (0 = no limit)
};

struct Train2;   // stage-B buffers (defined with run_backward2)

struct chg_batch {
  int B = 0, N = 0, Ed = 0, Eu = 0, A = 0, Eb = 0, L = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  // inputs
  int *z, *atom_owner, *atom_off, *e_center, *e_nbr, *e_d2u, *e_owner, *e_rev, *p_center, *p_nbr, *u_u2d, *u_bnode, *bn_und, *a_ctr, *a_b1c, *a_b2c, *a_d1, *a_d2;
  float *frac, *lattice, *e_image;
  // geometry / features
  float* cart;
  f32x4 *ev, *eu;
  float *hb0, *wag, *wbgc;
  float* atom[MAX_CONV + 1];
  float* hbc[MAX_CONV + 1];
  float* ang[MAX_CONV];
  // first-layer partial-product tables, one set per layer so the reverse sweep reuses the forward's
  float* Pl[MAX_CONV];          // AtomConv l: [N,256]
  float* Ql[MAX_CONV];          // AtomConv l: [Eu,128]
  float* Rl[2 * MAX_CONV];      // BondConv l (slot l) / AngleUpdate l (slot L+l): [Eb,256]
  float* Sl[2 * MAX_CONV];      // same slots: [N,128]
  // scatter targets, one per layer so that each direction of the sweep needs ONE memset (not one per layer)
  float* agg_l[MAX_CONV];       // AtomConv l: [N,64]
  float* aggB_l[MAX_CONV];      // BondConv l: [Eb,64]
  float* GP_l[MAX_CONV];        // AtomConv l adjoint: [N,256]
  float* GR_l[2 * MAX_CONV];    // BondConv / AngleUpdate adjoint (slots like Rl): [Eb,256]
  float* GS_l[2 * MAX_CONV];    // same slots: [N,128]
  // outputs
  float *energy, *site_energy, *site_raw, *magmom, *crystal_fea, *force, *virial, *volume;
  // reverse sweep
  float *Ga, *GA, *Gb, *Gwag, *Gwbgc, *Gang, *GQ, *Gagg, *Grk, *Gu;
  float* phase = nullptr;   // CHG_PHASE_TIMING builds: per-phase shader-clock totals of the angle kernels
  WinIndex win{};           // centre-major row order + window slots of the angle adjoints (kernels_angle_w.h), built by prepare_windows
  int *win_tmp = nullptr, *win_scan = nullptr;
  int win_grid = 64;        // workgroups of the per-atom kernels (a multiple of 64: the atom schedule is built for it, k_win_schedule)
  bool win_built = false;   // the index exists (batches too small to give every wave a few atoms never build it)
  float *zero1, *zero1_end, *zero2, *zero2_end;   // contiguous ranges cleared by one memset each
  uint32_t last_task = 0;
  bool seed1_adjoints = false;   // the first-order adjoints (seed 1) of the last force / stress sweep are still in the batch (GP_l, GR_l, GS_l, Gwag, Gwbgc)
  // the whole launch sequence of one chg_predict, captured once per (batch, task) and replayed:
  // ~170 launches per call make small batches (MD: one structure) launch-bound otherwise
  hipGraphExec_t graph_exec = nullptr;
  uint32_t graph_task = 0;
  int eager_calls = 0;        // the first call of a (batch, task) runs eagerly: one-shot batches never pay a capture
  std::map<std::string, std::pair<const float*, size_t>> named;
  std::map<std::string, std::pair<const int*, size_t>> named_i32;
  // fine-tuning backward (chg_backward): allocated on first use, freed with the batch
  char* train_arena = nullptr;
  size_t train_bytes = 0;
  std::vector<int> h_atom_off;   // host copy (chg_backward: atoms per structure)
  std::vector<double> h_volume;  // host copy of the cell volumes (chg_backward: stress cotangent -> strain direction)
  // stage B (second-order) workspace: one more arena, carved by layout_train2
  char* t2_arena = nullptr;
  size_t t2_bytes = 0;
  struct Train2* t2 = nullptr;
  float* t_mcot = nullptr;   // [N] magmom cotangent
  float h_g_b3 = 0.f;        // host-side gradient of the readout's last bias (copied into the blob on the device)
  bool t_has_mcot = false;
  float *t_grad = nullptr, *t_cot = nullptr, *t_dumpG = nullptr, *t_dumpH = nullptr, *t_dumpZ = nullptr, *t_Xb = nullptr, *t_Xa = nullptr,
        *t_ro = nullptr;
};

namespace {

#define HIP_TRY(eng, expr)                                                                          \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      (eng)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                               \
      return CHG_EHIP;                                                                              \
    }                                                                                               \
  } while (0)

// ---- weight blob layout: MUST mirror chgnet_amd/pack.py:weight_layout -------------------------------
struct Cursor {
  const float* base;
  size_t pos = 0;
  const float* take(size_t n) {
    pos += (4 - pos % 4) % 4;  // 16-byte alignment of every tensor
    const float* p = base + pos;
    pos += n;
    return p;
  }
};

void take_gated_tail(Cursor& c, GatedW& g, const float*& w2c_t, const float*& w2g_t) {
  g.w2c = c.take(D * D); g.b2c = c.take(D); g.w2g = c.take(D * D); g.b2g = c.take(D);
  w2c_t = c.take(D * D); w2g_t = c.take(D * D);
}
void take_ln(Cursor& c, GatedW& g) {
  g.ln1_g = c.take(D); g.ln1_b = c.take(D); g.ln2_g = c.take(D); g.ln2_b = c.take(D);
}

size_t layout_weights(const float* base, int L, Weights& w) {
  Cursor c{base};
  w.atomref = c.take(94); w.emb = c.take(94 * D);
  w.freq_ag = c.take(NRAD); w.freq_bg = c.take(NRAD); w.freq_ang = c.take(NFREQ);
  w.w_bond_emb = c.take(D * NRAD); w.w_wag = c.take(D * NRAD); w.w_wbg = c.take(D * NRAD); w.w_ang_emb = c.take(D * NANG);
  for (int l = 0; l < L; ++l) {
    ACW& a = w.ac[l];
    a.w_cn = c.take(4 * D * D); a.w_bond = c.take(2 * D * D); a.b1 = c.take(2 * D); a.q_bias = c.take(2 * D);
    take_gated_tail(c, a.g, a.w2c_t, a.w2g_t);
    take_ln(c, a.g);
    a.w_out = c.take(D * D); a.b_out = c.take(D); a.w_out_t = c.take(D * D);
    a.w_cn_t = c.take(4 * D * D); a.w_bond_t = c.take(2 * D * D);
  }
  for (int l = 0; l < L - 1; ++l) {
    BCW& b = w.bc[l];
    b.w_bij = c.take(4 * D * D); b.w_ang = c.take(2 * D * D); b.w_ctr = c.take(2 * D * D); b.b1 = c.take(2 * D);
    take_gated_tail(c, b.g, b.w2c_t, b.w2g_t);
    take_ln(c, b.g);
    b.w_out = c.take(D * D); b.b_out = c.take(D); b.w_out_t = c.take(D * D);
    b.w_bij_t = c.take(4 * D * D); b.w_ang_t = c.take(2 * D * D); b.w_ctr_t = c.take(2 * D * D);
  }
  for (int l = 0; l < L - 1; ++l) {
    AUW& u = w.au[l];
    u.w_bij = c.take(4 * D * D); u.w_ang = c.take(2 * D * D); u.w_ctr = c.take(2 * D * D); u.b1 = c.take(2 * D);
    u.g = GatedW{};
    take_ln(c, u.g);
    u.w_bij_t = c.take(4 * D * D); u.w_ang_t = c.take(2 * D * D); u.w_ctr_t = c.take(2 * D * D);
  }
  w.site_w = c.take(D); w.site_b = c.take(1); w.ro_ln_g = c.take(D); w.ro_ln_b = c.take(D);
  w.mlp_w0 = c.take(D * D); w.mlp_b0 = c.take(D); w.mlp_w1 = c.take(D * D); w.mlp_b1 = c.take(D);
  w.mlp_w2 = c.take(D * D); w.mlp_b2 = c.take(D); w.mlp_w3 = c.take(D); w.mlp_b3 = c.take(1);
  w.mlp_w0_t = c.take(D * D); w.mlp_w1_t = c.take(D * D); w.mlp_w2_t = c.take(D * D);
  return c.pos;
}

// ---- launch helpers --------------------------------------------------------------------------------
int prof_entry(chg_engine* eng, const char* label) {
  auto it = eng->prof_index.find(label);
  if (it != eng->prof_index.end()) return it->second;
  eng->prof.push_back(ProfEntry{label});
  const int idx = (int)eng->prof.size() - 1;
  eng->prof_index[label] = idx;
  return idx;
}

hipEvent_t get_event(chg_engine* eng) {
  if (!eng->event_pool.empty()) {
    hipEvent_t e = eng->event_pool.back();
    eng->event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

struct LaunchScope {
  chg_engine* eng;
  PendingEvent pe{};
  bool on;
  LaunchScope(chg_engine* e, const char* label) : eng(e), on(e->profiling) {
    if (on) {
      pe.entry = prof_entry(eng, label);
      pe.start = get_event(eng);
      pe.stop = get_event(eng);
      hipEventRecord(pe.start, eng->stream);
    }
  }
  ~LaunchScope() {
    if (on) {
      hipEventRecord(pe.stop, eng->stream);
      eng->pending.push_back(pe);
    }
  }
};

int collect_profile(chg_engine* eng) {
  for (auto& pe : eng->pending) {
    float ms = 0.f;
    HIP_TRY(eng, hipEventSynchronize(pe.stop));
    HIP_TRY(eng, hipEventElapsedTime(&ms, pe.start, pe.stop));
    eng->prof[pe.entry].launches += 1;
    eng->prof[pe.entry].ms += ms;
    eng->event_pool.push_back(pe.start);
    eng->event_pool.push_back(pe.stop);
  }
  eng->pending.clear();
  return CHG_OK;
}

// workgroups per CU launched for the tile kernels (CHGNET_GRID_MULT, timing experiments; default 2)
static int tile_grid_mult() {
  static const int m = [] { const char* e = std::getenv("CHGNET_GRID_MULT"); const int v = e ? std::atoi(e) : 2; return v > 0 ? v : 2; }();
  return m;
}

int grid_for(int rows, int max_blocks, int block_rows = BLOCK_ROWS) {
  int ntiles = (rows + block_rows - 1) / block_rows;
  int g = std::min(ntiles, max_blocks);
  if (g >= 8) g &= ~7;  // multiple of 8: tile_range keeps neighbouring ranges on one XCD
  return std::max(g, 1);
}

// grid of a tile kernel: CHGNET_GRID_MULT workgroups per CU -- one per CU when that already gives every workgroup no more than a few
// tiles (small batches: the second workgroup of a CU would stage the weights again for one or two tiles; MD replay 1.478 -> 1.437 ms)
int tile_grid(chg_engine* eng, int rows, int block_rows = BLOCK_ROWS) {
  const int ntiles = (rows + block_rows - 1) / block_rows;
  const int mult = ntiles <= 4 * eng->num_cus ? 1 : tile_grid_mult();
  // rounded UP to a multiple of 8 (tile_range's XCD mapping): rounding 221 blocks down to 216 left 42 waves of a 256-atom cell's
  // AtomConv kernels with a second tile, i.e. doubled the kernel's time; a workgroup without tiles costs nothing
  return std::max(1, std::min((ntiles + 7) & ~7, mult * eng->num_cus));
}

template <int K, int NOUT, int PARTS = 1>
int launch_rows_gemm(chg_engine* eng, const char* label, const RowsGemm& p) {
  if (p.rows <= 0) return CHG_OK;
  LaunchScope ls(eng, label);
  const size_t lds = rows_gemm_lds<K, NOUT, PARTS>();
  hipLaunchKernelGGL((k_rows_gemm<K, NOUT, PARTS>), dim3(grid_for(p.rows, 4 * eng->num_cus)), dim3(BLOCK), lds, eng->stream, p);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

// few rows (MD-size batches): every row GEMM as column blocks of a 16-wide instance (RowsGemm::col_blocks)
constexpr int SMALL_GEMM_ROWS = 32768;
constexpr int SMALL_GEMM_COLS = 16;
template <int K, int PARTS>
int launch_rows_gemm_cols(chg_engine* eng, const char* label, RowsGemm p, int n_out, int n_out_first) {
  if (p.rows <= 0) return CHG_OK;
  p.col_blocks = n_out / SMALL_GEMM_COLS; p.blocks1 = n_out_first / SMALL_GEMM_COLS;
  LaunchScope ls(eng, label);
  hipLaunchKernelGGL((k_rows_gemm<K, SMALL_GEMM_COLS, PARTS>), dim3(grid_for(p.rows, 4 * eng->num_cus), p.col_blocks), dim3(BLOCK),
                     (rows_gemm_lds<K, SMALL_GEMM_COLS, PARTS>()), eng->stream, p);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

// Y[out] (+)= X[in] . Wt^T, dispatch on (K, NOUT)
int rows_gemm(chg_engine* eng, const char* label, int K, int NOUT, const float* X, int ldx, const int* in_idx, const float* Wt,
              const float* bias, const float* resid, int ldr, float* Y, int ldy, const int* out_idx, int rows, int accumulate) {
  RowsGemm p{X, ldx, in_idx, Wt, bias, resid, ldr, Y, ldy, out_idx, rows, accumulate, nullptr, 0, 0, 0, 0};
  if (rows <= SMALL_GEMM_ROWS && K == 64 && (NOUT == 64 || NOUT == 128)) return launch_rows_gemm_cols<64, 1>(eng, label, p, NOUT, NOUT);
  if (rows <= SMALL_GEMM_ROWS && K == 128 && NOUT == 64) return launch_rows_gemm_cols<128, 1>(eng, label, p, NOUT, NOUT);
  if (K == 64 && NOUT == 64) return launch_rows_gemm<64, 64>(eng, label, p);
  if (K == 64 && NOUT == 128) return launch_rows_gemm<64, 128>(eng, label, p);
  if (K == 128 && NOUT == 64) return launch_rows_gemm<128, 64>(eng, label, p);
  eng->err = "rows_gemm: unsupported shape";
  return CHG_EINVAL;
}

// both halves of a 256-wide table in one launch:  Y[:, 0:128 | 128:256] = X . [Wt ; Wt2]^T  (64 -> 2 x 128)
int rows_gemm_out2(chg_engine* eng, const char* label, const float* X, const int* in_idx, const float* Wt, const float* Wt2,
                   const float* bias, float* Y, int ldy, int rows) {
  RowsGemm p{X, D, in_idx, Wt, bias, nullptr, 0, Y, ldy, nullptr, rows, 0, Wt2, 0, 2 * D, 0, 0};
  if (rows <= SMALL_GEMM_ROWS) return launch_rows_gemm_cols<64, 1>(eng, label, p, 4 * D, 2 * D);
  return launch_rows_gemm<64, 128, 2>(eng, label, p);
}
// ... and its adjoint:  Y (+)= X[:, 0:128] . Wt^T + X[:, 128:256] . Wt2^T   (2 x 128 -> 64)
int rows_gemm_in2(chg_engine* eng, const char* label, const float* X, int ldx, const float* Wt, const float* Wt2, float* Y,
                  const int* out_idx, int rows, int accumulate) {
  RowsGemm p{X, ldx, nullptr, Wt, nullptr, nullptr, 0, Y, D, out_idx, rows, accumulate, Wt2, 2 * D, 0, 0, 0};
  if (rows <= SMALL_GEMM_ROWS) return launch_rows_gemm_cols<128, 2>(eng, label, p, D, D);
  return launch_rows_gemm<128, 64, 2>(eng, label, p);
}

int zero(chg_engine* eng, void* p, size_t bytes) {
  if (bytes == 0) return CHG_OK;
  LaunchScope ls(eng, "memset");
  HIP_TRY(eng, hipMemsetAsync(p, 0, bytes, eng->stream));
  return CHG_OK;
}

__global__ void k_gather_rows(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int rows) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * (D / 4)) return;
  const int r = t / (D / 4), q = t % (D / 4);
  reinterpret_cast<f32x4*>(dst)[t] = reinterpret_cast<const f32x4*>(src + (size_t)idx[r] * D)[q];
}

#define TRY(x)                   \
  do {                           \
    int _s = (x);                \
    if (_s != CHG_OK) return _s; \
  } while (0)

// scratch ints exclusive_scan_with needs for n elements
inline size_t scan_scratch_ints(int n) { return n <= SCAN_CHUNK ? 1 : 2 * ((size_t)n / SCAN_CHUNK + 2); }

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

inline dim3 g1(int64_t n, int b = 256) { return dim3((unsigned)std::max<int64_t>(1, (n + b - 1) / b)); }
inline int wave_grid(chg_engine* eng, int64_t items) {   // one wave per item, 4 waves per block, grid-stride
  return (int)std::max<int64_t>(1, std::min<int64_t>((items + 3) / 4, 16 * (int64_t)eng->num_cus));
}

// ---- AtomConv ----------------------------------------------------------------------------------------
// tables of layer l:  P = atom[l] . [Wc;Wn]^T (+b1 on the centre half).  The bond partial Q = h_bond^l . Wb^T is contracted inside
// k_atomconv_fwd, which leaves it behind as a table when a reverse sweep follows; chg_backward after an energy-only predict builds
// the tables itself (atomconv_q_table).
// Prebuilt weight blocks (see stage_image, mfma_tile.h): one per tile kernel and layer, laid out by the kernels' own staging code.
int build_images(chg_engine* eng) {
  const int L = eng->desc.n_conv;
  constexpr size_t AF = ac_fwd_image_floats(), AB = ac_bwd_image_floats();
  constexpr size_t BF = AngleLds<true, false>::tiles, BB = AngleLds<true, true>::tiles, UF = AngleLds<false, false>::tiles,
                   UB = AngleLds<false, true>::tiles;
  static_assert(AF % 4 == 0 && AB % 4 == 0 && BF % 4 == 0 && BB % 4 == 0 && UF % 4 == 0 && UB % 4 == 0, "images are copied in 16-byte units");
  const size_t total = (size_t)L * (2 * AF + AB) + (size_t)(L - 1) * (BF + BB + UF + UB);
  if (!eng->d_images) {
    HIP_TRY(eng, hipMalloc(&eng->d_images, total * sizeof(float)));
    HIP_TRY(eng, hipMemsetAsync(eng->d_images, 0, total * sizeof(float), eng->stream));   // slots no staging writes (unused vectors)
  }
  float* at = eng->d_images;
  auto take = [&](size_t n) { float* p = at; at += n; return p; };
  for (int l = 0; l < L; ++l) {
    AtomConvArgs a{};
    a.gw = eng->w.ac[l].g; a.w_bond = eng->w.ac[l].w_bond;
    for (int qb = 0; qb < 2; ++qb) {
      a.q_bias = qb ? eng->w.ac[l].q_bias : nullptr;
      float* img = take(AF);
      eng->img_ac_fwd[qb][l] = img;
      hipLaunchKernelGGL(k_atomconv_image<false>, dim3(1), dim3(BLOCK), 0, eng->stream, a, img);
    }
    float* img = take(AB);
    eng->img_ac_bwd[l] = img;
    hipLaunchKernelGGL(k_atomconv_image<true>, dim3(1), dim3(BLOCK), 0, eng->stream, a, img);
  }
  for (int l = 0; l + 1 < L; ++l) {
    const BCW& bc = eng->w.bc[l];
    const AUW& au = eng->w.au[l];
    float* img;
    eng->img_angle[0][l] = img = take(BF);
    hipLaunchKernelGGL((k_angle_image<true, false>), dim3(1), dim3(BLOCK), 0, eng->stream, bc.w_ang, bc.g, img);
    eng->img_angle[1][l] = img = take(BB);
    hipLaunchKernelGGL((k_angle_image<true, true>), dim3(1), dim3(BLOCK), 0, eng->stream, bc.w_ang, bc.g, img);
    eng->img_angle[0][L + l] = img = take(UF);
    hipLaunchKernelGGL((k_angle_image<false, false>), dim3(1), dim3(BLOCK), 0, eng->stream, au.w_ang, au.g, img);
    eng->img_angle[1][L + l] = img = take(UB);
    hipLaunchKernelGGL((k_angle_image<false, true>), dim3(1), dim3(BLOCK), 0, eng->stream, au.w_ang, au.g, img);
  }
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

int atomconv_tables(chg_engine* eng, chg_batch* b, int l) {
  const ACW& w = eng->w.ac[l];
  return rows_gemm_out2(eng, "gemm_P", b->atom[l], nullptr, w.w_cn, w.w_cn + 2 * D * D, w.b1, b->Pl[l], 4 * D, b->N);
}

int atomconv_q_table(chg_engine* eng, chg_batch* b, int l) {
  const ACW& w = eng->w.ac[l];
  float* Q = b->Ql[l];
  if (b->Eu == 0) return CHG_OK;
  // q_bias: constant shift of the bonds outside the bond graph when mlp_out has a bias (0.2.0 only; zero otherwise);
  // the reference runs BondConv only when the batch has angles (model.py:460)
  TRY(rows_gemm(eng, "gemm_Q", 64, 128, b->hb0, D, nullptr, w.w_bond, b->A > 0 ? w.q_bias : nullptr, nullptr, 0, Q, 2 * D, nullptr, b->Eu, 0));
  if (b->Eb > 0 && b->hbc[l] != b->hbc[0])   // bond-graph nodes carry layer-l features
    TRY(rows_gemm(eng, "gemm_Qnode", 64, 128, b->hbc[l], D, nullptr, w.w_bond, nullptr, nullptr, 0, Q, 2 * D, b->bn_und, b->Eb, 0));
  return CHG_OK;
}

// Tile order per kernel: bit k of CHGNET_TILE_INTERLEAVE (default 31 = every kernel; A/B switch) -- 1 atomconv_fwd, 2 atomconv_bwd,
// 4 bondconv_fwd, 8 angleupd_fwd, 16 row-order angle adjoints.  Same-box A/B (profiles/r04_experiments.md): the interleaved sweep cuts
// the fabric traffic of every kernel by 20-30 %; their times move by 0-3 % (they are bound by vector-ALU issue, not by bytes).
static int interleave_mask() {
  static const int m = [] { const char* e = std::getenv("CHGNET_TILE_INTERLEAVE"); return e ? std::atoi(e) : 31; }();
  return m;
}

AtomConvArgs atomconv_args(chg_engine* eng, chg_batch* b, int l) {
  AtomConvArgs a{};
  a.P = b->Pl[l]; a.Q = b->Ql[l]; a.wag = b->wag;
  a.e_center = b->e_center; a.e_nbr = b->e_nbr; a.e_d2u = b->e_d2u; a.n_edges = b->Ed;
  a.gw = eng->w.ac[l].g;
  a.agg = b->agg_l[l]; a.GA = b->GA; a.GP = b->GP_l[l]; a.GQ = b->GQ; a.Gwag = b->Gwag;
  a.first_wag = l == b->L - 1;   // the reverse sweep starts with the last AtomConv
  a.hb0 = b->hb0;
  a.hbc = (b->Eb > 0 && b->hbc[l] != b->hbc[0]) ? b->hbc[l] : nullptr;   // bond-graph nodes carry layer-l features
  a.u_bnode = b->u_bnode;
  a.w_bond = eng->w.ac[l].w_bond;
  a.q_bias = b->A > 0 ? eng->w.ac[l].q_bias : nullptr;
  return a;
}

int atomconv_fwd(chg_engine* eng, chg_batch* b, int l, bool keep_q) {
  const ACW& w = eng->w.ac[l];
  if (b->Ed > 0) {
    TRY(atomconv_tables(eng, b, l));
    LaunchScope ls(eng, "atomconv_fwd");
    const size_t lds = atomconv_lds<FWD_WAVES, false, true>();
    AtomConvArgs a = atomconv_args(eng, b, l);
    a.e_center = b->p_center;   // bond-pair order
    a.image = eng->img_ac_fwd[a.q_bias ? 1 : 0][l];
    a.e_nbr = b->p_nbr;
    a.Qout = keep_q ? b->Ql[l] : nullptr;   // the reverse sweep gathers the bond partial as a table
    a.interleave = interleave_mask() & 1;
    hipLaunchKernelGGL((k_atomconv_fwd<FWD_WAVES>), dim3(tile_grid(eng, b->Ed, TILE_ROWS * FWD_WAVES)), dim3(64 * FWD_WAVES), lds, eng->stream, a);
    HIP_TRY(eng, hipGetLastError());
  }
  // atom[l+1] = agg . Wout^T + b_out + atom[l]       (layers.py:127-132)
  return rows_gemm(eng, "gemm_out", 64, 64, b->agg_l[l], D, nullptr, w.w_out, w.b_out, b->atom[l], D, b->atom[l + 1], D, nullptr, b->N, 0);
}

int atomconv_bwd(chg_engine* eng, chg_batch* b, int l) {
  const ACW& w = eng->w.ac[l];
  if (b->Ed == 0) return CHG_OK;  // agg == 0: only the residual path, already in Ga
  TRY(rows_gemm(eng, "gemm_Gagg", 64, 64, b->Ga, D, nullptr, w.w_out_t, nullptr, nullptr, 0, b->GA, D, nullptr, b->N, 0));
  {  // pair-ordered edge list: GQ and Gwag rows are owned by one tile each (no zeroing, no atomics)
    AtomConvArgs a = atomconv_args(eng, b, l);
    a.e_center = b->p_center;
    a.e_nbr = b->p_nbr;
    a.image = eng->img_ac_bwd[l];
    a.interleave = (interleave_mask() >> 1) & 1;
    LaunchScope ls(eng, "atomconv_bwd");
    hipLaunchKernelGGL(k_atomconv_bwd<false>, dim3(tile_grid(eng, b->Ed)), dim3(BLOCK), (atomconv_lds<WAVES, true>()), eng->stream, a);
    HIP_TRY(eng, hipGetLastError());
  }
  if (l > 0) {  // dE/d atom[l] += GPc . Wc + GPn . Wn   (atom[0] is an embedding: no position dependence)
    TRY(rows_gemm_in2(eng, "gemm_GP", b->GP_l[l], 4 * D, w.w_cn_t, w.w_cn_t + 2 * D * D, b->Ga, nullptr, b->N, 1));
  }
  return rows_gemm(eng, "gemm_GQ", 128, 64, b->GQ, 2 * D, nullptr, w.w_bond_t, nullptr, nullptr, 0, b->Gb, D, nullptr, b->Eu, l == b->L - 1 ? 0 : 1);
}

// ---- BondConv / AngleUpdate ----------------------------------------------------------------------------
// tables: S = atom . Wctr^T + b1 (per atom),  R = hbc . [Wi;Wj]^T (per bond-graph node)
int angle_tables(chg_engine* eng, chg_batch* b, int slot, const float* atom, const float* hbc, const float* w_bij, const float* w_ctr, const float* b1) {
  float *S = b->Sl[slot], *R = b->Rl[slot];
  if (b->N <= SMALL_GEMM_ROWS && b->Eb <= SMALL_GEMM_ROWS && b->N > 0 && b->Eb > 0) {   // small batch: both tables in one launch
    RowsGemm2 g{};
    g.a = RowsGemm{atom, D, nullptr, w_ctr, b1, nullptr, 0, S, 2 * D, nullptr, b->N, 0, nullptr, 0, 0, 2 * D / SMALL_GEMM_COLS, 2 * D / SMALL_GEMM_COLS};
    g.b = RowsGemm{hbc, D, nullptr, w_bij, nullptr, nullptr, 0, R, 4 * D, nullptr, b->Eb, 0, w_bij + 2 * D * D, 0, 2 * D, 4 * D / SMALL_GEMM_COLS,
                   2 * D / SMALL_GEMM_COLS};
    LaunchScope ls(eng, "gemm_SR");
    hipLaunchKernelGGL((k_rows_gemm_pair<64, SMALL_GEMM_COLS>), dim3(grid_for(std::max(b->N, b->Eb), 4 * eng->num_cus), g.a.col_blocks + g.b.col_blocks),
                       dim3(BLOCK), (rows_gemm_lds<64, SMALL_GEMM_COLS, 1>()), eng->stream, g);
    HIP_TRY(eng, hipGetLastError());
    return CHG_OK;
  }
  TRY(rows_gemm(eng, "gemm_S", 64, 128, atom, D, nullptr, w_ctr, b1, nullptr, 0, S, 2 * D, nullptr, b->N, 0));
  return rows_gemm_out2(eng, "gemm_R", hbc, nullptr, w_bij, w_bij + 2 * D * D, nullptr, R, 4 * D, b->Eb);
}

AngleArgs angle_args(chg_batch* b, int slot, const float* ang, const float* w_ang, const GatedW& g, float* out) {
  AngleArgs a{};
  a.R = b->Rl[slot]; a.S = b->Sl[slot]; a.ang = ang; a.wbgc = b->wbgc;
  a.a_ctr = b->a_ctr; a.a_b1c = b->a_b1c; a.a_b2c = b->a_b2c; a.n_angles = b->A;
  a.w_ang = w_ang; a.gw = g; a.out = out; a.slot = slot;
  a.Gagg = b->Gagg; a.Gang = b->Gang; a.GR = b->GR_l[slot]; a.GS = b->GS_l[slot]; a.Gwbgc = b->Gwbgc; a.phase = b->phase;
  a.first_gang = slot == b->L - 2;   // slot l < L is BondConv l; the sweep's first angle kernel is BondConv L-2
  a.skip_flag = b->win.flag;
  return a;
}

// Which adjoints run per atom (kernels_angle_w.h): both.  AngleUpdate 2.23 -> 1.60 ms; BondConv 3.44 -> 3.03 ms once all of its
// contractions run in split precision from row-major images.  CHGNET_PER_ATOM_BONDCONV=0 / CHGNET_PER_ATOM_ANGLEUPD=0 switch back to
// the plain kernels for A/B timing.
static bool per_atom_adjoint(bool hidden) {
  static const bool bc = [] { const char* e = std::getenv("CHGNET_PER_ATOM_BONDCONV"); return !e || std::atoi(e) != 0; }();
  static const bool au = [] { const char* e = std::getenv("CHGNET_PER_ATOM_ANGLEUPD"); return !e || std::atoi(e) != 0; }();
  return hidden ? bc : au;
}

template <bool HIDDEN, bool BWD, int NW = WAVES>
int launch_angle(chg_engine* eng, const char* label, chg_batch* b, const AngleArgs& a) {
  LaunchScope ls(eng, label);
  AngleArgs plain = a;
  if (BWD && b->win_built && per_atom_adjoint(HIDDEN)) {
    // per-atom adjoint (kernels_angle_w.h) when the batch has the canonical angle structure, else the row-order one: both are
    // launched, the device flag picks (no host round trip, and a captured hipGraph stays valid across rebuilt graphs)
    AngleWArgs w{};
    w.a = a; w.w = b->win;
    hipLaunchKernelGGL((k_angle_bwd_w<HIDDEN>), dim3(b->win_grid), dim3(BLOCK), angle_w_lds<HIDDEN>(), eng->stream, w);
    HIP_TRY(eng, hipGetLastError());
  } else {
    plain.skip_flag = nullptr;
  }
  plain.image = eng->img_angle[BWD ? 1 : 0][a.slot];
  plain.interleave = (interleave_mask() >> (BWD ? 4 : HIDDEN ? 2 : 3)) & 1;
  const size_t lds = angle_lds<HIDDEN, NW, BWD>();
  hipLaunchKernelGGL((k_angle<HIDDEN, BWD, NW>), dim3(tile_grid(eng, b->A, TILE_ROWS * NW)), dim3(64 * NW), lds, eng->stream, plain);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

int bondconv_fwd(chg_engine* eng, chg_batch* b, int l) {
  const BCW& w = eng->w.bc[l];
  TRY(angle_tables(eng, b, l, b->atom[l + 1], b->hbc[l], w.w_bij, w.w_ctr, w.b1));
  TRY((launch_angle<true, false>(eng, "bondconv_fwd", b, angle_args(b, l, b->ang[l], w.w_ang, w.g, b->aggB_l[l]))));
  // hbc[l+1] = agg . Wout^T + b_out + hbc[l]          (layers.py:255-260)
  return rows_gemm(eng, "gemm_out", 64, 64, b->aggB_l[l], D, nullptr, w.w_out, w.b_out, b->hbc[l], D, b->hbc[l + 1], D, nullptr, b->Eb, 0);
}

int angleupd_fwd(chg_engine* eng, chg_batch* b, int l) {
  const AUW& w = eng->w.au[l];
  TRY(angle_tables(eng, b, b->L + l, b->atom[l + 1], b->hbc[l + 1], w.w_bij, w.w_ctr, w.b1));
  return launch_angle<false, false, FWD_WAVES>(eng, "angleupd_fwd", b, angle_args(b, b->L + l, b->ang[l], w.w_ang, w.g, b->ang[l + 1]));
}

// scatter of the table gradients back to atoms / bond nodes
int angle_table_grads(chg_engine* eng, chg_batch* b, int slot, const float* w_bij_t, const float* w_ctr_t) {
  TRY(rows_gemm_in2(eng, "gemm_GR", b->GR_l[slot], 4 * D, w_bij_t, w_bij_t + 2 * D * D, b->Gb, b->bn_und, b->Eb, 1));
  return rows_gemm(eng, "gemm_GS", 128, 64, b->GS_l[slot], 2 * D, nullptr, w_ctr_t, nullptr, nullptr, 0, b->Ga, D, nullptr, b->N, 1);
}

int bondconv_bwd(chg_engine* eng, chg_batch* b, int l) {
  const BCW& w = eng->w.bc[l];
  TRY(rows_gemm(eng, "gemm_Gagg", 64, 64, b->Gb, D, b->bn_und, w.w_out_t, nullptr, nullptr, 0, b->Gagg, D, nullptr, b->Eb, 0));
  TRY((launch_angle<true, true>(eng, "bondconv_bwd", b, angle_args(b, l, b->ang[l], w.w_ang, w.g, nullptr))));
  return angle_table_grads(eng, b, l, w.w_bij_t, w.w_ctr_t);
}

int angleupd_bwd(chg_engine* eng, chg_batch* b, int l) {
  const AUW& w = eng->w.au[l];
  TRY((launch_angle<false, true>(eng, "angleupd_bwd", b, angle_args(b, b->L + l, b->ang[l], w.w_ang, w.g, nullptr))));
  return angle_table_grads(eng, b, b->L + l, w.w_bij_t, w.w_ctr_t);
}

BondEmbedTArgs bond_embed_args(chg_engine* eng, chg_batch* b) {
  const Weights& w = eng->w;
  BondEmbedTArgs a{};
  a.ev = b->ev; a.u_u2d = b->u_u2d; a.u_bnode = b->u_bnode; a.n_und = b->Eu;
  a.freq_ag = w.freq_ag; a.freq_bg = w.freq_bg; a.w_emb = w.w_bond_emb; a.w_ag = w.w_wag; a.w_bg = w.w_wbg;
  a.rc_ag = eng->desc.atom_graph_cutoff; a.rc_bg = eng->desc.bond_graph_cutoff;
  const double p = eng->desc.cutoff_coeff;   // basis.py:184-186
  a.env = Envelope{(float)(-(p + 1) * (p + 2) / 2), (float)(p * (p + 2)), (float)(-p * (p + 1) / 2), eng->desc.cutoff_coeff};
  a.hb0 = b->hb0; a.wag = b->wag; a.wbgc = b->wbgc;
  a.hbc0 = b->Eb > 0 ? b->hbc[0] : nullptr;
  a.Gb = b->Gb; a.Gwag = b->Gwag; a.Gwbgc = b->Gwbgc; a.Grk = b->Grk;
  return a;
}

AngleEmbedTArgs angle_embed_args(chg_engine* eng, chg_batch* b) {
  AngleEmbedTArgs a{};
  a.eu = b->eu; a.a_d1 = b->a_d1; a.a_d2 = b->a_d2; a.n_angles = b->A;
  a.freq = eng->w.freq_ang; a.w_emb = eng->w.w_ang_emb;
  a.ang0 = b->ang[0]; a.Gang = b->Gang; a.Gu = b->Gu;
  return a;
}

int run_predict(chg_engine* eng, chg_batch* b, uint32_t task) {
  const Weights& w = eng->w;
  const int L = b->L;
  const bool want_f = task & CHG_TASK_F, want_s = task & CHG_TASK_S, want_m = task & CHG_TASK_M;
  const bool want_grad = want_f || want_s;
  hipStream_t st = eng->stream;
#ifdef CHG_PHASE_TIMING
  HIP_TRY(eng, hipMemsetAsync(b->phase, 0, sizeof(float) * PHASE_FLOATS, st));
#endif

  // ---- geometry, bases, embeddings (model.py:826-871, 432-439) ----
  { LaunchScope ls(eng, "cart");
    hipLaunchKernelGGL(k_cart, g1(b->N), dim3(256), 0, st, b->frac, b->lattice, b->atom_owner, b->cart, b->N); }
  if (b->Ed > 0) {
    { LaunchScope ls(eng, "edge_geom");
      hipLaunchKernelGGL(k_edge_geom, g1(b->Ed), dim3(256), 0, st, b->cart, b->lattice, b->e_center, b->e_nbr, b->e_image, b->e_owner, b->ev, b->eu, b->Ed); }
    { LaunchScope ls(eng, "bond_embed_fwd");
      hipLaunchKernelGGL((k_bond_embed_t<false>), dim3(grid_for(b->Eu, 2 * eng->num_cus)), dim3(BLOCK), bond_embed_lds(), st, bond_embed_args(eng, b)); }
  }
  if (b->A > 0) {
    LaunchScope ls(eng, "angle_embed_fwd");
    hipLaunchKernelGGL((k_angle_embed_t<false>), dim3(grid_for(b->A, 2 * eng->num_cus)), dim3(BLOCK), angle_embed_lds(), st, angle_embed_args(eng, b));
  }
  { LaunchScope ls(eng, "atom_embed");
    hipLaunchKernelGGL(k_atom_embed, g1((int64_t)b->N * (D / 4)), dim3(256), 0, st, b->z, w.emb, b->atom[0], b->N); }
  HIP_TRY(eng, hipGetLastError());   // (hbc[0], the nodes' copy of their embedding rows, is written by k_bond_embed_t)

  // ---- message passing (model.py:442-496) ----
  // every forward scatter target + crystal_fea -- and, when a reverse sweep follows, its accumulators too (the two ranges are
  // adjacent in the arena: one memset instead of two)
  TRY(zero(eng, b->zero1, (size_t)((char*)(want_grad ? b->zero2_end : b->zero1_end) - (char*)b->zero1)));
  for (int l = 0; l < L - 1; ++l) {
    TRY(atomconv_fwd(eng, b, l, want_grad));
    if (b->A > 0) {
      TRY(bondconv_fwd(eng, b, l));
      if (l < L - 2) TRY(angleupd_fwd(eng, b, l));   // the last AngleUpdate's output is never consumed
    }
  }
  if (want_m) {
    LaunchScope ls(eng, "magmom");
    hipLaunchKernelGGL(k_magmom, dim3(wave_grid(eng, b->N)), dim3(256), 0, st, b->atom[L - 1], w.site_w, w.site_b, b->magmom, b->N);
  }
  TRY(atomconv_fwd(eng, b, L - 1, want_grad));

  // ---- readout (model.py:497-509) and its adjoint ----
  {
    ReadoutArgs r{};
    r.atom = b->atom[L]; r.atom_owner = b->atom_owner; r.z = b->z; r.n_atoms = b->N;
    r.ln_g = w.ro_ln_g; r.ln_b = w.ro_ln_b; r.w0 = w.mlp_w0; r.b0 = w.mlp_b0; r.w1 = w.mlp_w1; r.b1 = w.mlp_b1;
    r.w2 = w.mlp_w2; r.b2 = w.mlp_b2; r.w3 = w.mlp_w3; r.b3 = w.mlp_b3; r.atomref = w.atomref;
    r.has_composition = eng->desc.has_composition;
    r.site_energy = b->site_energy; r.site_raw = b->site_raw; r.crystal_fea = b->crystal_fea;
    r.Ga = want_grad ? b->Ga : nullptr;
    LaunchScope ls(eng, "readout");
    hipLaunchKernelGGL(k_readout<false>, dim3(grid_for(b->N, eng->num_cus)), dim3(BLOCK), readout_lds(), st, r);
    HIP_TRY(eng, hipGetLastError());
  }

  // ---- reverse sweep: dE/dv_e ----
  if (want_grad) {
    TRY(atomconv_bwd(eng, b, L - 1));
    for (int l = L - 2; l >= 0; --l) {
      if (b->A > 0) {
        if (l < L - 2) TRY(angleupd_bwd(eng, b, l));
        TRY(bondconv_bwd(eng, b, l));
      }
      TRY(atomconv_bwd(eng, b, l));
    }
    if (b->Ed > 0) {
      { LaunchScope ls(eng, "bond_embed_bwd");
        hipLaunchKernelGGL((k_bond_embed_t<true>), dim3(grid_for(b->Eu, 2 * eng->num_cus)), dim3(BLOCK), bond_embed_lds(), st, bond_embed_args(eng, b)); }
      if (b->A > 0) {
        LaunchScope ls(eng, "angle_embed_bwd");
        hipLaunchKernelGGL((k_angle_embed_t<true>), dim3(grid_for(b->A, 2 * eng->num_cus)), dim3(BLOCK), angle_embed_lds(), st, angle_embed_args(eng, b));
      }
      ForceArgs f{};
      f.ev = b->ev; f.eu = b->eu; f.Gu = b->Gu; f.Grk = b->Grk;
      f.e_center = b->e_center; f.e_d2u = b->e_d2u; f.e_owner = b->e_owner; f.e_rev = b->e_rev; f.u_u2d = b->u_u2d;
      f.n_edges = b->Ed; f.force = b->force; f.virial = b->virial;
      LaunchScope ls(eng, "edge_force");
      hipLaunchKernelGGL(k_edge_force, g1(b->Ed, EF_EDGES_PER_BLOCK), dim3(256), 0, st, f);
    }
    HIP_TRY(eng, hipGetLastError());
  }
  {
    FinalizeArgs f{};
    f.lattice = b->lattice; f.atom_off = b->atom_off; f.n_struct = b->B;
    f.is_intensive = eng->desc.is_intensive; f.has_composition = eng->desc.has_composition; f.want_stress = want_s;
    f.site_raw = b->site_raw; f.z = b->z; f.atomref = eng->w.atomref; f.energy_out = b->energy; f.virial = b->virial; f.volume = b->volume;
    LaunchScope ls(eng, "finalize");
    hipLaunchKernelGGL(k_finalize, g1((int64_t)b->B * 64), dim3(256), 0, st, f);   // one wave per structure
    HIP_TRY(eng, hipGetLastError());
  }
  b->last_task = task;
  b->seed1_adjoints = want_grad;
  return CHG_OK;
}

// ---- arena ---------------------------------------------------------------------------------------------
struct Carver {
  char* base;
  size_t pos = 0;
  template <class T>
  T* take(size_t n) {
    pos = (pos + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + pos) : nullptr;
    pos += std::max<size_t>(n, 1) * sizeof(T);
    return p;
  }
};

void carve(chg_batch* b, char* base, size_t& total) {
  Carver c{base};
  const size_t B = b->B, N = b->N, Ed = b->Ed, Eu = b->Eu, A = b->A, Eb = b->Eb;
  const int L = b->L;
  b->z = c.take<int>(N); b->atom_owner = c.take<int>(N); b->atom_off = c.take<int>(B + 1);
  b->e_center = c.take<int>(Ed); b->e_nbr = c.take<int>(Ed); b->e_d2u = c.take<int>(Ed); b->e_owner = c.take<int>(Ed);
  b->e_rev = c.take<int>(Ed); b->p_center = c.take<int>(Ed); b->p_nbr = c.take<int>(Ed);
  b->u_u2d = c.take<int>(Eu); b->u_bnode = c.take<int>(Eu); b->bn_und = c.take<int>(Eb);
  b->a_ctr = c.take<int>(A); b->a_b1c = c.take<int>(A); b->a_b2c = c.take<int>(A); b->a_d1 = c.take<int>(A); b->a_d2 = c.take<int>(A);
  b->frac = c.take<float>(3 * N); b->lattice = c.take<float>(9 * B); b->e_image = c.take<float>(3 * Ed);
  b->cart = c.take<float>(3 * N); b->ev = c.take<f32x4>(Ed); b->eu = c.take<f32x4>(Ed);
  b->hb0 = c.take<float>(Eu * D); b->wag = c.take<float>(Eu * D); b->wbgc = c.take<float>(Eb * D);
  for (int l = 0; l <= L; ++l) b->atom[l] = c.take<float>(N * D);
  for (int l = 0; l < L; ++l) b->hbc[l] = (A > 0 || l == 0) ? c.take<float>(Eb * D) : nullptr;
  for (int l = 0; l < L - 1; ++l) b->ang[l] = c.take<float>(A * D);
  for (int l = 0; l < L; ++l) { b->Pl[l] = c.take<float>(N * 4 * D); b->Ql[l] = c.take<float>(Eu * 2 * D); }
  for (int t = 0; t < 2 * L; ++t) { b->Rl[t] = c.take<float>(Eb * 4 * D); b->Sl[t] = c.take<float>(N * 2 * D); }
  b->energy = c.take<float>(B); b->site_energy = c.take<float>(N); b->site_raw = c.take<float>(N); b->magmom = c.take<float>(N); b->volume = c.take<float>(B);
  // zero group 1 (cleared with one memset before the readout)
  b->zero1 = c.take<float>(0);
  b->crystal_fea = c.take<float>(B * D);
  for (int l = 0; l < L; ++l) b->agg_l[l] = c.take<float>(N * D);
  for (int l = 0; l < L - 1; ++l) b->aggB_l[l] = c.take<float>(Eb * D);
  b->zero1_end = c.take<float>(0);
  // zero group 2 (cleared with one memset before the reverse sweep)
  b->zero2 = c.take<float>(0);
  b->Gwbgc = c.take<float>(Eb * D);
  b->Gu = c.take<float>(4 * Ed); b->force = c.take<float>(3 * N); b->virial = c.take<float>(9 * B);
  for (int l = 0; l < L; ++l) b->GP_l[l] = c.take<float>(N * 4 * D);
  for (int t = 0; t < 2 * L; ++t) {
    const bool used = (t < L - 1) || (t >= L && t < 2 * L - 2);   // BondConv 0..L-2, AngleUpdate L..2L-3
    b->GR_l[t] = used ? c.take<float>(Eb * 4 * D) : nullptr;
    b->GS_l[t] = used ? c.take<float>(N * 2 * D) : nullptr;
  }
  b->zero2_end = c.take<float>(0);
  // first written by a plain store in every sweep (AtomConv L-1: Gwag, its gemm_GQ: Gb; BondConv L-2: Gang): never zeroed
  b->Gb = c.take<float>(Eu * D); b->Gwag = c.take<float>(Eu * D); b->Gang = c.take<float>(A * D);
  b->Ga = c.take<float>(N * D); b->GA = c.take<float>(N * D);
  b->GQ = c.take<float>(Eu * 2 * D);
  b->Gagg = c.take<float>(Eb * D);
  b->Grk = c.take<float>(Eu);
  b->phase = c.take<float>(PHASE_FLOATS);
  {   // windowed angle adjoints (kernels_angle_w.h)
    WinIndex& w = b->win;
    w.flag = c.take<int>(4); w.na = c.take<int>(N + 1); w.boff = c.take<int>(N + 1); w.aoff = c.take<int>(N + 1);
    w.head = c.take<int>(Ed); w.rank = c.take<int>(Ed); w.list = c.take<int>(A ? N * WIN_LIST : 0);
    w.q_a = c.take<int>(A); w.q_ctr = c.take<int>(A); w.q_b1c = c.take<int>(A); w.q_b2c = c.take<int>(A); w.q_ab1 = c.take<int>(A); w.q_ab2 = c.take<int>(A);
    w.abbond = c.take<int>(2 * Eb);
    w.wave_head = c.take<int>(A ? WIN_MAX_GRID * WAVES : 0); w.next_atom = c.take<int>(A ? N : 0); w.xatom = c.take<int>(WIN_MAX_GRID / 8 + 1);
    b->win_tmp = c.take<int>(N + 1);
    b->win_scan = c.take<int>(scan_scratch_ints((int)N + 1));
  }
  if (A == 0) for (int l = 1; l < L; ++l) b->hbc[l] = b->hbc[0];   // no BondConv: bond features never change
  total = (c.pos + 255) & ~size_t(255);
}

// Centre-major row order and (atom, bond) pair indices of the angle adjoints (kernels_angle_w.h): once per batch topology,
// stream-ordered, no host round trip; a graph without the canonical structure leaves win.flag[0] = 0.
int prepare_windows(chg_engine* eng, chg_batch* b) {
  hipStream_t st = eng->stream;
  WinIndex& w = b->win;
  b->win_built = false;
  // one workgroup per CU (their LDS admits no second one), in whole groups of 64 waves = 8 workgroups per XCD (k_win_schedule)
  b->win_grid = std::max(64, std::min(eng->num_cus / 64 * 64, WIN_MAX_GRID));
  // MD-size batches: fewer than a few atoms per wave would leave most of the chip idle -- plain adjoints, and nothing to build
  // (CHGNET_WIN_MIN_ATOMS_PER_WAVE=0 sends small batches through the per-atom kernels too: parity tests on the golden cases)
  const char* min_env = std::getenv("CHGNET_WIN_MIN_ATOMS_PER_WAVE");
  const long min_atoms = min_env ? std::atol(min_env) : WIN_MIN_ATOMS_PER_WAVE;
  if (b->A == 0 || (long)b->N < min_atoms * b->win_grid * WAVES) return CHG_OK;
  if ((size_t)b->N / SCAN_CHUNK + 1 > (1u << 16)) return CHG_OK;      // beyond the two-level scan: plain adjoints
  b->win_built = true;
  HIP_TRY(eng, hipMemsetAsync(w.flag, 0, sizeof(int) * 4, st));
  HIP_TRY(eng, hipMemsetAsync(w.na, 0, sizeof(int) * ((size_t)b->N + 1), st));
  HIP_TRY(eng, hipMemsetAsync(w.head, 0xFF, sizeof(int) * (size_t)b->Ed, st));
  HIP_TRY(eng, hipMemsetAsync(w.rank, 0xFF, sizeof(int) * (size_t)b->Ed, st));
  hipLaunchKernelGGL(k_win_init, dim3(1), dim3(1), 0, st, w, b->win_grid);
  hipLaunchKernelGGL(k_win_heads, g1(b->A), dim3(256), 0, st, b->a_d1, b->a_ctr, b->A, b->Ed, b->N, w);
  TRY(exclusive_scan_with(eng, b->win_scan, w.na, w.boff, b->N + 1));
  hipLaunchKernelGGL(k_win_counts, g1((int64_t)b->N + 1), dim3(256), 0, st, b->N, w, b->win_tmp);
  TRY(exclusive_scan_with(eng, b->win_scan, b->win_tmp, w.aoff, b->N + 1));
  hipLaunchKernelGGL(k_win_ranks, g1(b->A), dim3(256), 0, st, b->a_d1, b->a_ctr, b->a_b1c, b->A, b->N, w);
  hipLaunchKernelGGL(k_win_rows, g1(b->A), dim3(256), 0, st, b->a_ctr, b->a_b1c, b->a_b2c, b->a_d1, b->a_d2, b->A, b->N, b->Ed, w);
  hipLaunchKernelGGL(k_win_groups, g1(b->win_grid / 8 + 1), dim3(256), 0, st, b->N, b->A, b->win_grid / 8, w);
  hipLaunchKernelGGL(k_win_schedule, dim3(b->win_grid / 8), dim3(64), 0, st, b->win_grid, w);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

void register_names(chg_batch* b) {
  auto& m = b->named;
  m.clear();
  const size_t N = b->N, Ed = b->Ed, Eu = b->Eu, A = b->A, Eb = b->Eb, B = b->B;
  m["cart"] = {b->cart, 3 * N};
  m["ev"] = {reinterpret_cast<const float*>(b->ev), 4 * Ed};
  m["eu"] = {reinterpret_cast<const float*>(b->eu), 4 * Ed};
  m["hb0"] = {b->hb0, Eu * D}; m["wag"] = {b->wag, Eu * D}; m["wbgc"] = {b->wbgc, Eb * D};
  for (int l = 0; l <= b->L; ++l) m["atom" + std::to_string(l)] = {b->atom[l], N * D};
  for (int l = 0; l < b->L; ++l) m["hbc" + std::to_string(l)] = {b->hbc[l], Eb * D};
  for (int l = 0; l < b->L - 1; ++l) m["ang" + std::to_string(l)] = {b->ang[l], A * D};
  for (int l = 0; l < b->L; ++l) { m["P" + std::to_string(l)] = {b->Pl[l], N * 4 * D}; m["Q" + std::to_string(l)] = {b->Ql[l], Eu * 2 * D}; }
  m["agg"] = {b->agg_l[b->L - 1], N * D}; m["aggB"] = {b->aggB_l[0], Eb * D};
  m["Ga"] = {b->Ga, N * D}; m["GA"] = {b->GA, N * D}; m["Gb"] = {b->Gb, Eu * D}; m["Gwag"] = {b->Gwag, Eu * D};
  m["Gwbgc"] = {b->Gwbgc, Eb * D}; m["Gang"] = {b->Gang, A * D}; m["GP"] = {b->GP_l[0], N * 4 * D}; m["GQ"] = {b->GQ, Eu * 2 * D};
  m["GR"] = {b->GR_l[0], Eb * 4 * D}; m["GS"] = {b->GS_l[0], N * 2 * D}; m["Grk"] = {b->Grk, Eu}; m["Gu"] = {b->Gu, 4 * Ed};
  m["virial"] = {b->virial, 9 * B}; m["volume"] = {b->volume, B}; m["phase"] = {b->phase, PHASE_FLOATS};
  m["frac"] = {b->frac, 3 * N}; m["lattice"] = {b->lattice, 9 * B}; m["e_image"] = {b->e_image, 3 * Ed};
  auto& mi = b->named_i32;
  mi.clear();
  mi["z"] = {b->z, N}; mi["atom_owner"] = {b->atom_owner, N}; mi["atom_off"] = {b->atom_off, B + 1};
  mi["e_center"] = {b->e_center, Ed}; mi["e_nbr"] = {b->e_nbr, Ed}; mi["e_d2u"] = {b->e_d2u, Ed}; mi["e_owner"] = {b->e_owner, Ed};
  mi["e_rev"] = {b->e_rev, Ed}; mi["p_center"] = {b->p_center, Ed}; mi["p_nbr"] = {b->p_nbr, Ed};
  mi["u_u2d"] = {b->u_u2d, Eu}; mi["u_bnode"] = {b->u_bnode, Eu}; mi["bn_und"] = {b->bn_und, Eb};
  mi["a_ctr"] = {b->a_ctr, A}; mi["a_b1c"] = {b->a_b1c, A}; mi["a_b2c"] = {b->a_b2c, A}; mi["a_d1"] = {b->a_d1, A}; mi["a_d2"] = {b->a_d2, A};
  mi["win_flag"] = {b->win.flag, 4}; mi["win_q_a"] = {b->win.q_a, A}; mi["win_q_ctr"] = {b->win.q_ctr, A}; mi["win_na"] = {b->win.na, N + 1};
  mi["win_aoff"] = {b->win.aoff, N + 1}; mi["win_q_ab1"] = {b->win.q_ab1, A}; mi["win_q_ab2"] = {b->win.q_ab2, A};
  mi["win_next_atom"] = {b->win.next_atom, A ? N : 0};
  mi["win_wave_head"] = {b->win.wave_head, A ? (size_t)WIN_MAX_GRID * WAVES : 0}; mi["win_xatom"] = {b->win.xatom, (size_t)WIN_MAX_GRID / 8 + 1};   // win_flag[3] = workgroups
}

template <class T>
int h2d(chg_engine* eng, T* dst, const T* src, size_t n) {
  if (n == 0) return CHG_OK;
  if (!src) { eng->err = "chg_batch_upload: null host array"; return CHG_EINVAL; }
  HIP_TRY(eng, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, eng->stream));
  return CHG_OK;
}
template <class T>
int d2h(chg_engine* eng, T* dst, const T* src, size_t n) {
  if (n == 0 || !dst) return CHG_OK;
  HIP_TRY(eng, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, eng->stream));
  return CHG_OK;
}

// =====================================================================================================
// Fine-tuning backward, stage A: gradient of  sum_b cot[b] * E_b  w.r.t. every weight.
// The reverse sweep is the one of run_predict started from the cotangent; the TRAIN instantiations of the
// adjoint kernels additionally leave what the weight-gradient reductions need (kernels_train.h).
// =====================================================================================================
template <int MT, int NT>
int xty(chg_engine* eng, const char* label, const float* A, int lda, const int* a_idx, const float* B, int ldb, const int* b_idx, int rows,
        float alpha, float* out, int ldo, int n_cols, float* a_colsum = nullptr) {
  if (rows <= 0) return CHG_OK;
  LaunchScope ls(eng, label);
  XtyArgs p{A, lda, a_idx, B, ldb, b_idx, rows, alpha, out, ldo, n_cols, a_colsum};
  // long identity-mapped operands (the angle / edge rows of the fine-tuning sweeps): three-piece bf16 form (kernels_train.h k_xty3);
  // CHGNET_XTY3=0: the f32-MFMA kernel everywhere (A/B timing)
  static const bool x3 = [] { const char* e = std::getenv("CHGNET_XTY3"); return !e || std::atoi(e) != 0; }();
  if (x3 && !a_idx && !b_idx && rows >= 65536 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0) {
    const int nstages = (rows + X3_ROWS - 1) / X3_ROWS;
    const int g3 = std::max(8, std::min(nstages / 8, 2 * eng->num_cus) & ~7);   // two workgroups (2 x 78 KB of LDS, <= 128 registers) per CU
    hipLaunchKernelGGL((k_xty3<MT, NT>), dim3(g3), dim3(BLOCK), (xty3_lds<MT, NT>()), eng->stream, p);
    HIP_TRY(eng, hipGetLastError());
    return CHG_OK;
  }
  // one workgroup per CU is resident (LDS), and every workgroup ends with one global atomic per output element:
  // no more workgroups than CUs, and at least four row tiles each
  const int ntiles = (rows + BLOCK_ROWS - 1) / BLOCK_ROWS;
  int grid = std::max(1, std::min((ntiles + 3) / 4, eng->num_cus));
  if (grid >= 8) grid &= ~7;
  hipLaunchKernelGGL((k_xty<MT, NT>), dim3(grid), dim3(BLOCK), (xty_lds<MT, NT>()), eng->stream, p);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

// core^T core and gate^T gate of two [rows,128] = [core | gate] arrays (the second layer of a gated MLP): one pass over full 512-byte
// rows (k_xty3<8, 8, true>) when the operands are long, else the two half-row contractions
int xty_halves(chg_engine* eng, const char* label, const float* A, const float* B, int rows, float* out_c, float* out_g,
               float* colsum_c = nullptr, float* colsum_g = nullptr) {
  if (rows <= 0) return CHG_OK;
  static const bool x3 = [] { const char* e = std::getenv("CHGNET_XTY3"); return !e || std::atoi(e) != 0; }();
  if (x3 && rows >= 65536 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0) {
    LaunchScope ls(eng, label);
    XtyArgs p{A, 2 * D, nullptr, B, 2 * D, nullptr, rows, 1.0f, out_c, D, D, colsum_c, out_g, colsum_g};
    const int nstages = (rows + X3_ROWS - 1) / X3_ROWS;
    const int g3 = std::max(8, std::min(nstages / 8, 2 * eng->num_cus) & ~7);   // 52 KB of LDS (single buffer): two workgroups per CU
    hipLaunchKernelGGL((k_xty3<8, 8, true>), dim3(g3), dim3(BLOCK), (xty3_lds<8, 8, true>()), eng->stream, p);
    HIP_TRY(eng, hipGetLastError());
    return CHG_OK;
  }
  TRY((xty<4, 4>(eng, label, A, 2 * D, nullptr, B, 2 * D, nullptr, rows, 1.0f, out_c, D, D, colsum_c)));
  return xty<4, 4>(eng, label, A + D, 2 * D, nullptr, B + D, 2 * D, nullptr, rows, 1.0f, out_g, D, D, colsum_g);
}

int colsum(chg_engine* eng, const float* A, int lda, const float* Bm, int ldb, int rows, int width, float* out) {
  if (rows <= 0) return CHG_OK;
  LaunchScope ls(eng, "wgrad_colsum");
  ColsumArgs p{A, lda, Bm, ldb, rows, width, 1.0f, out};
  const int ngrp = 256 / width;
  const int grid = std::max(1, std::min((rows + ngrp * 64 - 1) / (ngrp * 64), 4 * eng->num_cus));
  hipLaunchKernelGGL(k_colsum, dim3(grid), dim3(256), 0, eng->stream, p);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

// Training workspaces are taken from / returned to the engine: a train step makes a new batch every iteration, and a hipMalloc
// of tens of GB per step would dominate it.  One slot per kind (0: first-order workspace, 1: second-order workspace); a request
// is rounded up by 8 % so that the slightly different batches of an epoch reuse the same block.
char* acquire_workspace(chg_engine* eng, size_t total, size_t& got, int kind) {
  for (int i = 0; i < (int)eng->work_pool.size(); ++i)
    if (eng->work_kind[i] == kind && eng->work_pool[i].second >= total) {
      char* p = eng->work_pool[i].first;
      got = eng->work_pool[i].second;
      eng->work_pool.erase(eng->work_pool.begin() + i);
      eng->work_kind.erase(eng->work_kind.begin() + i);
      return p;
    }
  for (int i = (int)eng->work_pool.size() - 1; i >= 0; --i)     // a pooled block of this kind that is too small is of no use any more
    if (eng->work_kind[i] == kind) {
      hipFree(eng->work_pool[i].first);
      eng->work_pool.erase(eng->work_pool.begin() + i);
      eng->work_kind.erase(eng->work_kind.begin() + i);
    }
  const size_t want = ((total + total / 12) + (size_t(64) << 20) - 1) & ~((size_t(64) << 20) - 1);
  char* p = nullptr;
  if (hipMalloc(&p, want) == hipSuccess) { got = want; return p; }
  (void)hipGetLastError();
  for (auto& a : eng->work_pool) hipFree(a.first);   // make room (both pools) and ask for the exact size
  eng->work_pool.clear();
  eng->work_kind.clear();
  for (auto& a : eng->arena_pool) hipFree(a.first);
  eng->arena_pool.clear();
  if (hipMalloc(&p, total) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  got = total;
  return p;
}
void release_workspace(chg_engine* eng, char* p, size_t bytes, int kind) {
  if (!p) return;
  bool have = false;
  if (eng) for (int k : eng->work_kind) have = have || k == kind;
  if (eng && !have) { eng->work_pool.emplace_back(p, bytes); eng->work_kind.push_back(kind); }
  else hipFree(p);
}

int ensure_train_buffers(chg_engine* eng, chg_batch* b) {
  if (b->train_arena) return CHG_OK;
  const size_t N = b->N, Ed = b->Ed, Eu = b->Eu, A = b->A, rows = std::max(Ed, A);
  Carver c{nullptr};
  auto lay = [&](Carver& cv) {
    b->t_grad = cv.take<float>((size_t)eng->desc.n_weights);
    b->t_cot = cv.take<float>(b->B);
    b->t_mcot = cv.take<float>(N);
    b->t_dumpG = cv.take<float>(rows * 2 * D);
    b->t_dumpH = cv.take<float>(rows * 2 * D);
    b->t_dumpZ = cv.take<float>(A * 2 * D);
    b->t_Xb = cv.take<float>(Eu * D);
    b->t_Xa = cv.take<float>(A * KB);
    b->t_ro = cv.take<float>((size_t)RO_NDUMP * N * D);
  };
  lay(c);
  const size_t total = (c.pos + 255) & ~size_t(255);
  if (eng->memory_limit && total + b->arena_bytes > eng->memory_limit) {
    eng->err = "chg_backward: training workspace of " + std::to_string(total) + " bytes exceeds the engine's memory limit";
    return CHG_ENOMEM;
  }
  size_t got = 0;
  char* base = acquire_workspace(eng, total, got, 0);
  if (!base) {
    eng->err = "hipMalloc of " + std::to_string(total) + " bytes (training workspace) failed";
    return CHG_ENOMEM;
  }
  Carver c2{base};
  lay(c2);
  b->train_arena = base;
  b->train_bytes = got;
  return CHG_OK;
}

float* grad_of(chg_engine* eng, chg_batch* b, const float* w) { return b->t_grad + (w - eng->d_weights); }

// gated-MLP internals of one layer: dW2c, dW2g, db2c, db2g from the (adjoint, hidden activation) dumps
int gated_tail_grads(chg_engine* eng, chg_batch* b, const GatedW& g, int rows, float* (*G)(chg_engine*, chg_batch*, const float*)) {
  return xty_halves(eng, "wgrad_w2", b->t_dumpG, b->t_dumpH, rows, G(eng, b, g.w2c), G(eng, b, g.w2g), G(eng, b, g.b2c), G(eng, b, g.b2g));
}

int run_backward(chg_engine* eng, chg_batch* b) {
  const Weights& w = eng->w;
  b->seed1_adjoints = false;   // this sweep reuses the force sweep's buffers with the loss cotangents as seeds
  const int L = b->L;
  hipStream_t st = eng->stream;
  auto G = [&](const float* wp) { return grad_of(eng, b, wp); };
  const int N = b->N, Ed = b->Ed, Eu = b->Eu, A = b->A, Eb = b->Eb;
  TRY(zero(eng, b->t_grad, sizeof(float) * (size_t)eng->desc.n_weights));
  TRY(zero(eng, b->zero2, (size_t)((char*)b->zero2_end - (char*)b->zero2)));

  // ---- readout: dE/d atom[L] from the cotangent; per-atom operands of the MLP / LayerNorm gradients ----
  {
    ReadoutArgs r{};
    r.atom = b->atom[L]; r.atom_owner = b->atom_owner; r.z = b->z; r.n_atoms = N;
    r.ln_g = w.ro_ln_g; r.ln_b = w.ro_ln_b; r.w0 = w.mlp_w0; r.b0 = w.mlp_b0; r.w1 = w.mlp_w1; r.b1 = w.mlp_b1;
    r.w2 = w.mlp_w2; r.b2 = w.mlp_b2; r.w3 = w.mlp_w3; r.b3 = w.mlp_b3; r.atomref = w.atomref;
    r.has_composition = eng->desc.has_composition;
    r.site_energy = b->site_energy; r.site_raw = b->site_raw; r.crystal_fea = b->crystal_fea;
    r.Ga = b->Ga; r.cot = b->t_cot; r.dump = b->t_ro;
    LaunchScope ls(eng, "readout_train");
    hipLaunchKernelGGL(k_readout<true>, dim3(grid_for(N, eng->num_cus)), dim3(BLOCK), readout_lds(), st, r);
    HIP_TRY(eng, hipGetLastError());
  }
  {
    const size_t pl = (size_t)N * D;
    const float* ro = b->t_ro;
    TRY((xty<4, 4>(eng, "wgrad_readout", ro + RO_G1 * pl, D, nullptr, ro + RO_X0 * pl, D, nullptr, N, 1.0f, G(w.mlp_w0), D, D, G(w.mlp_b0))));
    TRY((xty<4, 4>(eng, "wgrad_readout", ro + RO_G2 * pl, D, nullptr, ro + RO_S1 * pl, D, nullptr, N, 1.0f, G(w.mlp_w1), D, D, G(w.mlp_b1))));
    TRY((xty<4, 4>(eng, "wgrad_readout", ro + RO_G3 * pl, D, nullptr, ro + RO_S2 * pl, D, nullptr, N, 1.0f, G(w.mlp_w2), D, D, G(w.mlp_b2))));
    TRY(colsum(eng, ro + RO_S3C * pl, D, nullptr, 0, N, D, G(w.mlp_w3)));
    TRY(colsum(eng, ro + RO_GXX * pl, D, nullptr, 0, N, D, G(w.ro_ln_g)));
    TRY(colsum(eng, ro + RO_GX * pl, D, nullptr, 0, N, D, G(w.ro_ln_b)));
    // d mlp_b3 = sum_b cot[b] * n_atoms[b] is formed on the host (chg_backward)
  }

  auto atomconv_train = [&](int l) -> int {
    const ACW& aw = w.ac[l];
    // atom[l+1] = agg . Wout^T + b_out + atom[l]
    TRY((xty<4, 4>(eng, "wgrad_out", b->Ga, D, nullptr, b->agg_l[l], D, nullptr, N, 1.0f, G(aw.w_out), D, D, G(aw.b_out))));
    if (Ed == 0) return CHG_OK;
    TRY(rows_gemm(eng, "gemm_Gagg", 64, 64, b->Ga, D, nullptr, aw.w_out_t, nullptr, nullptr, 0, b->GA, D, nullptr, N, 0));
    {
      AtomConvArgs a = atomconv_args(eng, b, l);
      a.e_center = b->p_center; a.e_nbr = b->p_nbr;
      a.dumpG = b->t_dumpG; a.dumpH = b->t_dumpH; a.g_ln = G(aw.g.ln1_g);
      LaunchScope ls(eng, "atomconv_bwd_train");
      hipLaunchKernelGGL(k_atomconv_bwd<true>, dim3(tile_grid(eng, Ed)), dim3(BLOCK), (atomconv_lds<WAVES, true>()), st, a);
      HIP_TRY(eng, hipGetLastError());
    }
    TRY(gated_tail_grads(eng, b, aw.g, Ed, grad_of));
    // first layer, factorised: table gradients contract with the rows the tables were made from
    TRY((xty<8, 4>(eng, "wgrad_tab", b->GP_l[l], 4 * D, nullptr, b->atom[l], D, nullptr, N, 1.0f, G(aw.w_cn), D, D, G(aw.b1))));   // b1 sits in the centre half
    TRY((xty<8, 4>(eng, "wgrad_tab", b->GP_l[l] + 2 * D, 4 * D, nullptr, b->atom[l], D, nullptr, N, 1.0f, G(aw.w_cn) + 2 * D * D, D, D)));
    TRY((xty<8, 4>(eng, "wgrad_tab", b->GQ, 2 * D, nullptr, b->hb0, D, nullptr, Eu, 1.0f, G(aw.w_bond), D, D)));
    if (Eb > 0 && b->hbc[l] != b->hbc[0]) {   // bond-graph nodes carry layer-l features instead of the embedding
      TRY((xty<8, 4>(eng, "wgrad_tab", b->GQ, 2 * D, b->bn_und, b->hbc[l], D, nullptr, Eb, 1.0f, G(aw.w_bond), D, D)));
      TRY((xty<8, 4>(eng, "wgrad_tab", b->GQ, 2 * D, b->bn_und, b->hb0, D, b->bn_und, Eb, -1.0f, G(aw.w_bond), D, D)));
    }
    TRY(rows_gemm_in2(eng, "gemm_GP", b->GP_l[l], 4 * D, aw.w_cn_t, aw.w_cn_t + 2 * D * D, b->Ga, nullptr, N, 1));   // l == 0 too: d emb needs dE/d atom[0]
    return rows_gemm(eng, "gemm_GQ", 128, 64, b->GQ, 2 * D, nullptr, aw.w_bond_t, nullptr, nullptr, 0, b->Gb, D, nullptr, Eu, l == L - 1 ? 0 : 1);
  };

  // shared tail of BondConv / AngleUpdate: table gradients of slot -> weights, then back to atoms / bonds
  auto angle_tables_train = [&](int slot, const float* hbc_rows, const float* atom_rows, const float* ang_rows, const float* gz_dump,
                                const float* w_bij, const float* w_ctr, const float* b1, const float* w_ang, const float* w_bij_t,
                                const float* w_ctr_t) -> int {
    TRY((xty<8, 4>(eng, "wgrad_tab", b->GR_l[slot], 4 * D, nullptr, hbc_rows, D, nullptr, Eb, 1.0f, G(w_bij), D, D)));
    TRY((xty<8, 4>(eng, "wgrad_tab", b->GR_l[slot] + 2 * D, 4 * D, nullptr, hbc_rows, D, nullptr, Eb, 1.0f, G(w_bij) + 2 * D * D, D, D)));
    TRY((xty<8, 4>(eng, "wgrad_tab", b->GS_l[slot], 2 * D, nullptr, atom_rows, D, nullptr, N, 1.0f, G(w_ctr), D, D, G(b1))));
    TRY((xty<8, 4>(eng, "wgrad_ang", gz_dump, 2 * D, nullptr, ang_rows, D, nullptr, A, 1.0f, G(w_ang), D, D)));
    return angle_table_grads(eng, b, slot, w_bij_t, w_ctr_t);
  };

  TRY(atomconv_train(L - 1));
  if (b->t_has_mcot) {   // Ga is dE/d atom[L-1] now: the features the magmom head reads (model.py:477-487)
    LaunchScope ls(eng, "magmom_bwd");
    hipLaunchKernelGGL(k_magmom_bwd, dim3(wave_grid(eng, N)), dim3(256), 0, st, b->atom[L - 1], w.site_w, w.site_b, b->t_mcot, b->Ga, G(w.site_w),
                       G(w.site_b), N);
    HIP_TRY(eng, hipGetLastError());
  }
  for (int l = L - 2; l >= 0; --l) {
    if (A > 0) {
      if (l < L - 2) {
        const AUW& uw = w.au[l];
        AngleArgs a = angle_args(b, L + l, b->ang[l], uw.w_ang, uw.g, nullptr);
        a.dumpG = b->t_dumpG; a.dumpH = nullptr; a.dumpZ = nullptr; a.g_ln = G(uw.g.ln1_g);
        {
          LaunchScope ls(eng, "angleupd_bwd_train");
          hipLaunchKernelGGL((k_angle<false, true, WAVES, true>), dim3(tile_grid(eng, A, TILE_ROWS * WAVES)), dim3(BLOCK),
                             (angle_lds<false, WAVES, true>()), st, a);
          HIP_TRY(eng, hipGetLastError());
        }
        TRY(angle_tables_train(L + l, b->hbc[l + 1], b->atom[l + 1], b->ang[l], b->t_dumpG, uw.w_bij, uw.w_ctr, uw.b1, uw.w_ang, uw.w_bij_t,
                               uw.w_ctr_t));
      }
      const BCW& bw = w.bc[l];
      // hbc[l+1] = aggB . Wout^T + b_out + hbc[l]; dE/d hbc[l+1] lives in the node rows of Gb
      TRY((xty<4, 4>(eng, "wgrad_out", b->Gb, D, b->bn_und, b->aggB_l[l], D, nullptr, Eb, 1.0f, G(bw.w_out), D, D)));
      TRY(rows_gemm(eng, "gemm_Gagg", 64, 64, b->Gb, D, b->bn_und, bw.w_out_t, nullptr, nullptr, 0, b->Gagg, D, nullptr, Eb, 0));
      AngleArgs a = angle_args(b, l, b->ang[l], bw.w_ang, bw.g, nullptr);
      a.dumpG = b->t_dumpG; a.dumpH = b->t_dumpH; a.dumpZ = b->t_dumpZ; a.g_ln = G(bw.g.ln1_g);
      {
        LaunchScope ls(eng, "bondconv_bwd_train");
        hipLaunchKernelGGL((k_angle<true, true, WAVES, true>), dim3(tile_grid(eng, A, TILE_ROWS * WAVES)), dim3(BLOCK),
                           (angle_lds<true, WAVES, true>()), st, a);
        HIP_TRY(eng, hipGetLastError());
      }
      TRY(gated_tail_grads(eng, b, bw.g, A, grad_of));
      TRY(angle_tables_train(l, b->hbc[l], b->atom[l + 1], b->ang[l], b->t_dumpZ, bw.w_bij, bw.w_ctr, bw.b1, bw.w_ang, bw.w_bij_t, bw.w_ctr_t));
    }
    TRY(atomconv_train(l));
  }

  // ---- embeddings: 31 -> 64 linears, learnable frequencies, atom embedding table ----
  if (Ed > 0) {
    {
      BondEmbedTArgs a = bond_embed_args(eng, b);
      a.Xb = b->t_Xb; a.g_freq_ag = G(w.freq_ag); a.g_freq_bg = G(w.freq_bg);
      LaunchScope ls(eng, "bond_embed_bwd_train");
      hipLaunchKernelGGL((k_bond_embed_t<true, true>), dim3(grid_for(Eu, 2 * eng->num_cus)), dim3(BLOCK), bond_embed_lds(), st, a);
      HIP_TRY(eng, hipGetLastError());
    }
    TRY((xty<4, 2>(eng, "wgrad_embed", b->Gb, D, nullptr, b->t_Xb, D, nullptr, Eu, 1.0f, G(w.w_bond_emb), NRAD, NRAD)));
    TRY((xty<4, 2>(eng, "wgrad_embed", b->Gwag, D, nullptr, b->t_Xb, D, nullptr, Eu, 1.0f, G(w.w_wag), NRAD, NRAD)));
    TRY((xty<4, 2>(eng, "wgrad_embed", b->Gwbgc, D, nullptr, b->t_Xb + KB, D, b->bn_und, Eb, 1.0f, G(w.w_wbg), NRAD, NRAD)));
    if (A > 0) {
      AngleEmbedTArgs a = angle_embed_args(eng, b);
      a.Xa = b->t_Xa; a.g_freq = G(w.freq_ang);
      {
        LaunchScope ls(eng, "angle_embed_bwd_train");
        hipLaunchKernelGGL((k_angle_embed_t<true, true>), dim3(grid_for(A, 2 * eng->num_cus)), dim3(BLOCK), angle_embed_lds(), st, a);
        HIP_TRY(eng, hipGetLastError());
      }
      TRY((xty<4, 2>(eng, "wgrad_embed", b->Gang, D, nullptr, b->t_Xa, KB, nullptr, A, 1.0f, G(w.w_ang_emb), NANG, NANG)));
    }
  }
  {
    LaunchScope ls(eng, "wgrad_atom_embed");
    hipLaunchKernelGGL(k_embed_grad, g1((int64_t)N * D), dim3(256), 0, st, b->Ga, b->z, G(w.emb), N);
    HIP_TRY(eng, hipGetLastError());
  }
  return CHG_OK;
}

// =====================================================================================================
// Stage B: gradient of  sum_b ce_b e_b + sum_i gm_i m_i + sum_i gF_i . F_i + sum_b gS_b : sigma_b  w.r.t. every weight
// (kernels_train2.h has the derivation).  Unfused first version: row GEMMs + row-local kernels + k_xty.
// =====================================================================================================
}  // namespace

struct Train2 {
  float *ux, *Wst;                                   // direction: [N,3], [B,9]
  f32x4 *vd4, *ud4;                                  // [Ed]
  float *X6, *X6d, *X3, *X3d, *X4, *X4d, *th2;       // bases + tangents [Eu,32] x4, [A,32] x2, [A,2]
  float *hb0d, *wagd, *wbgcd;                        // tangent embeddings [Eu,64] x2, [Eb,64]
  float *atomd[MAX_CONV + 1], *hbcd[MAX_CONV + 1], *angd[MAX_CONV], *aggd[MAX_CONV], *aggBd[MAX_CONV];
  float *Pd, *Qd, *Rd, *Sd, *ZA, *ZAd;               // tangent tables [N,256] [Eu,128] [Eb,256] [N,128]; W_ang . ang [A,128] x2
  float *Z, *Zd, *H, *Hd, *CG, *CGd, *BCG, *GCG, *BH, *GH, *BZ, *GZ;   // [R,128]  (Z..CGd: the CURRENT layer's rows, see cache)
  float* scratch6[6];                                // one shared set of Z, Zd, H, Hd, CG, CGd (recompute mode)
  // per-layer rows kept from the tangent forward for the reverse sweep when device memory allows (layer ids: AtomConv l -> l,
  // BondConv l -> L + l, AngleUpdate l -> 2L + l); otherwise the reverse sweep recomputes them into scratch6
  float* cache[3 * MAX_CONV][6];
  bool cached = false;
  float *bar_a, *g_a, *bar_b, *g_b, *bar_wag, *g_wag, *bar_wbg, *g_wbg, *bar_ang, *g_ang, *bar_agg, *g_agg;
  float *barP, *gP, *barQ, *gQ, *barR, *gR, *barS, *gS;
  float *gP0, *gR0, *gS0;                            // this workspace's own G(P), G(R), G(S) (gP / gR / gS may point into the batch)
  float* ro[26];                                     // readout planes [N,64]
  float *zero_lo, *zero_hi;                          // range cleared at the start of every call
};

namespace {

void free_train2(chg_batch* b) { delete b->t2; b->t2 = nullptr; }

// fused tile kernels (kernels_train2_tile.h); CHGNET_T2_UNFUSED=1 keeps the row-array pipeline of kernels_train2.h (A/B, debugging)
bool t2_fused() {
  static const bool fused = !std::getenv("CHGNET_T2_UNFUSED");
  return fused;
}

void layout_train2(chg_batch* b, Train2& t, Carver& c) {
  const size_t B = b->B, N = b->N, Ed = b->Ed, Eu = b->Eu, A = b->A, Eb = b->Eb, R = std::max(Ed, A);
  const int L = b->L;
  t.ux = c.take<float>(3 * N); t.Wst = c.take<float>(9 * B);
  t.vd4 = c.take<f32x4>(Ed); t.ud4 = c.take<f32x4>(Ed);
  t.X6 = c.take<float>(Eu * KB2); t.X6d = c.take<float>(Eu * KB2); t.X3 = c.take<float>(Eu * KB2); t.X3d = c.take<float>(Eu * KB2);
  t.X4 = c.take<float>(A * KB2); t.X4d = c.take<float>(A * KB2); t.th2 = c.take<float>(2 * A);
  t.hb0d = c.take<float>(Eu * D); t.wagd = c.take<float>(Eu * D); t.wbgcd = c.take<float>(Eb * D);
  for (int l = 0; l <= L; ++l) t.atomd[l] = c.take<float>(N * D);
  for (int l = 0; l < L; ++l) t.hbcd[l] = c.take<float>(Eb * D);
  for (int l = 0; l < L - 1; ++l) t.angd[l] = c.take<float>(A * D);
  t.Pd = c.take<float>(N * 4 * D); t.Qd = c.take<float>(Eu * 2 * D); t.Rd = c.take<float>(Eb * 4 * D); t.Sd = c.take<float>(N * 2 * D);
  // the fused sweep keeps six [rows,128] arrays (the operands of the weight-gradient contractions); the row-array pipeline sixteen
  // here and, when memory allows, a cache of six per layer
  const bool fused = t2_fused();
  t.ZA = t.ZAd = t.BH = t.GH = nullptr;
  if (!fused) { t.ZA = c.take<float>(A * 2 * D); t.ZAd = c.take<float>(A * 2 * D); }
  for (int q = 0; q < 6; ++q) t.scratch6[q] = (!fused || q == 2 || q == 3) ? c.take<float>(R * 2 * D) : nullptr;   // fused: H, Hd dumps
  float** rows[] = {&t.BCG, &t.GCG, &t.BZ, &t.GZ};
  for (float** r : rows) *r = c.take<float>(R * 2 * D);
  if (!fused) { t.BH = c.take<float>(R * 2 * D); t.GH = c.take<float>(R * 2 * D); }
  if (fused) t.cached = false;
  for (int id = 0; id < 3 * MAX_CONV; ++id)
    for (int q = 0; q < 6; ++q) t.cache[id][q] = nullptr;
  if (t.cached) {
    for (int l = 0; l < L; ++l)
      for (int q = 0; q < 6; ++q) t.cache[l][q] = c.take<float>(Ed * 2 * D);
    if (A > 0) {
      for (int l = 0; l < L - 1; ++l)
        for (int q = 0; q < 6; ++q) t.cache[L + l][q] = c.take<float>(A * 2 * D);
      for (int l = 0; l < L - 2; ++l)
        for (int q = 4; q < 6; ++q) t.cache[2 * L + l][q] = c.take<float>(A * 2 * D);   // single layer: only c|g (= z) and its tangent
    }
  }
  t.bar_agg = c.take<float>(std::max(N, Eb) * D); t.g_agg = c.take<float>(std::max(N, Eb) * D);
  t.bar_a = c.take<float>(N * D); t.g_a = c.take<float>(N * D);
  for (int i = 0; i < 26; ++i) t.ro[i] = c.take<float>(N * D);
  // everything below is accumulated into (atomics / += GEMMs): cleared at the start of a call
  t.zero_lo = c.take<float>(0);
  for (int l = 0; l < L; ++l) t.aggd[l] = c.take<float>(N * D);
  for (int l = 0; l < L - 1; ++l) t.aggBd[l] = c.take<float>(Eb * D);
  t.bar_b = c.take<float>(Eu * D); t.g_b = c.take<float>(Eu * D); t.bar_wag = c.take<float>(Eu * D); t.g_wag = c.take<float>(Eu * D);
  t.bar_wbg = c.take<float>(Eb * D); t.g_wbg = c.take<float>(Eb * D); t.bar_ang = c.take<float>(A * D); t.g_ang = c.take<float>(A * D);
  t.zero_hi = c.take<float>(0);
  // table gradients: cleared before every layer
  t.barP = c.take<float>(N * 4 * D); t.gP = c.take<float>(N * 4 * D); t.barQ = c.take<float>(Eu * 2 * D); t.gQ = c.take<float>(Eu * 2 * D);
  t.barR = c.take<float>(Eb * 4 * D); t.gR = c.take<float>(Eb * 4 * D); t.barS = c.take<float>(N * 2 * D); t.gS = c.take<float>(N * 2 * D);
  t.gP0 = t.gP; t.gR0 = t.gR; t.gS0 = t.gS;
}

int ensure_train2_buffers(chg_engine* eng, chg_batch* b) {
  if (b->t2) return CHG_OK;
  Train2* t = new (std::nothrow) Train2();
  if (!t) return CHG_ENOMEM;
  // keep the per-layer rows of the tangent forward for the reverse sweep if that still leaves a quarter of the free memory
  size_t total = 0;
  {
    size_t free_b = 0, total_b = 0;
    hipMemGetInfo(&free_b, &total_b);
    for (auto& a : eng->work_pool) free_b += a.second;
    t->cached = true;
    Carver cc{nullptr};
    layout_train2(b, *t, cc);
    const size_t want = (cc.pos + 255) & ~size_t(255);
    const size_t budget = eng->memory_limit ? std::min(free_b, eng->memory_limit) : free_b;
    if (std::getenv("CHGNET_TRAIN_NO_CACHE") || want > budget - budget / 4) t->cached = false;
  }
  Carver c{nullptr};
  layout_train2(b, *t, c);
  total = (c.pos + 255) & ~size_t(255);
  if (eng->memory_limit && total + b->arena_bytes + b->train_bytes > eng->memory_limit) {
    delete t;
    eng->err = "chg_backward: second-order training workspace of " + std::to_string(total) + " bytes exceeds the engine's memory limit";
    return CHG_ENOMEM;
  }
  size_t got = 0;
  char* base = acquire_workspace(eng, total, got, 1);
  if (!base) {
    delete t;
    eng->err = "hipMalloc of " + std::to_string(total) + " bytes (second-order training workspace) failed";
    return CHG_ENOMEM;
  }
  Carver c2{base};
  layout_train2(b, *t, c2);
  b->t2 = t;
  b->t2_arena = base;
  b->t2_bytes = got;
  return CHG_OK;
}

inline dim3 wave_rows_grid(chg_engine* eng, int64_t rows) { return dim3((unsigned)wave_grid(eng, rows)); }

int run_backward2(chg_engine* eng, chg_batch* b) {
  const Weights& w = eng->w;
  Train2& t = *b->t2;
  const int L = b->L;
  hipStream_t st = eng->stream;
  auto G = [&](const float* wp) { return grad_of(eng, b, wp); };
  const int N = b->N, Ed = b->Ed, Eu = b->Eu, A = b->A, Eb = b->Eb;
  const bool angles = A > 0;
  TRY(zero(eng, b->t_grad, sizeof(float) * (size_t)eng->desc.n_weights));
  TRY(zero(eng, t.zero_lo, (size_t)((char*)t.zero_hi - (char*)t.zero_lo)));
  auto gemm = [&](const char* label, int K, int NOUT, const float* X, int ldx, const int* in_idx, const float* Wt, const float* bias,
                  const float* resid, int ldr, float* Y, int ldy, const int* out_idx, int rows, int acc) {
    return rows_gemm(eng, label, K, NOUT, X, ldx, in_idx, Wt, bias, resid, ldr, Y, ldy, out_idx, rows, acc);
  };
  // Y[:, 0:64 | 64:128] = X[:, 0:64 | 64:128] . [Wc ; Wg]^T   (the two 64 x 64 second-layer blocks of a gated MLP)
  auto gemm_pair = [&](const float* X, const float* Wc, const float* Wg, const float* bc, const float* bg, float* Y, int rows) -> int {
    TRY(gemm("t2_gemm_w2", 64, 64, X, 2 * D, nullptr, Wc, bc, nullptr, 0, Y, 2 * D, nullptr, rows, 0));
    return gemm("t2_gemm_w2", 64, 64, X + D, 2 * D, nullptr, Wg, bg, nullptr, 0, Y + D, 2 * D, nullptr, rows, 0);
  };
  auto check = [&]() -> int { HIP_TRY(eng, hipGetLastError()); return CHG_OK; };
  // The fused sweep (t2_fused) does not re-form the G adjoints (seed 1) of quantities that only leave it: those are the first-order adjoints the
  // force sweep of chg_predict left in the batch (chg_backward makes sure that sweep has run): Gwag, Gwbgc and, per layer, GP / GR / GS.
  const bool fused = t2_fused();
  const float* g_wag = fused ? b->Gwag : t.g_wag;
  const float* g_wbg = fused ? b->Gwbgc : t.g_wbg;
  auto table_adjoints_of = [&](int atom_layer, int angle_slot) {   // where G(P) / G(R), G(S) of the layer being swept live
    t.gP = (fused && atom_layer >= 0) ? b->GP_l[atom_layer] : t.gP0;
    t.gR = (fused && angle_slot >= 0) ? b->GR_l[angle_slot] : t.gR0;
    t.gS = (fused && angle_slot >= 0) ? b->GS_l[angle_slot] : t.gS0;
  };

  // ---- direction -> tangent of geometry, bases, embeddings ---------------------------------------------
  if (Ed > 0) {
    LaunchScope ls(eng, "t2_geom");
    hipLaunchKernelGGL(k2_geom_t, g1(Ed), dim3(256), 0, st, b->ev, b->eu, b->e_center, b->e_nbr, b->e_owner, t.ux, t.Wst, t.vd4, t.ud4, Ed);
  }
  const double pc = eng->desc.cutoff_coeff;
  const Envelope env{(float)(-(pc + 1) * (pc + 2) / 2), (float)(pc * (pc + 2)), (float)(-pc * (pc + 1) / 2), eng->desc.cutoff_coeff};
  if (Eu > 0) {
    BondBasisArgs a{Eu, b->ev, t.vd4, b->u_u2d, w.freq_ag, w.freq_bg, eng->desc.atom_graph_cutoff, eng->desc.bond_graph_cutoff, env,
                    t.X6, t.X6d, t.X3, t.X3d};
    { LaunchScope ls(eng, "t2_basis");
      hipLaunchKernelGGL(k2_bond_basis, g1((int64_t)Eu * KB2), dim3(256), 0, st, a); }
    LaunchScope ls(eng, "t2_embed_lin");
    hipLaunchKernelGGL(k2_embed_lin, wave_rows_grid(eng, Eu), dim3(256), 0, st, t.X6d, w.w_bond_emb, t.hb0d, nullptr, Eu);
    hipLaunchKernelGGL(k2_embed_lin, wave_rows_grid(eng, Eu), dim3(256), 0, st, t.X6d, w.w_wag, t.wagd, nullptr, Eu);
    if (Eb > 0) hipLaunchKernelGGL(k2_embed_lin, wave_rows_grid(eng, Eb), dim3(256), 0, st, t.X3d, w.w_wbg, t.wbgcd, b->bn_und, Eb);
  }
  if (angles) {
    { LaunchScope ls(eng, "t2_basis");
      hipLaunchKernelGGL(k2_angle_basis, g1((int64_t)A * KB2), dim3(256), 0, st, b->eu, t.ud4, b->a_d1, b->a_d2, w.freq_ang, t.X4, t.X4d, t.th2, A); }
    LaunchScope ls(eng, "t2_embed_lin");
    hipLaunchKernelGGL(k2_embed_lin, wave_rows_grid(eng, A), dim3(256), 0, st, t.X4d, w.w_ang_emb, t.angd[0], nullptr, A);
  }
  TRY(zero(eng, t.atomd[0], sizeof(float) * (size_t)N * D));     // the atom embedding does not depend on the geometry
  if (Eb > 0) {
    LaunchScope ls(eng, "t2_gather");
    hipLaunchKernelGGL(k_gather_rows, g1((int64_t)Eb * (D / 4)), dim3(256), 0, st, t.hb0d, b->bn_und, t.hbcd[0], Eb);
  }
  TRY(check());

  // rows of the layer being worked on: its cache slot (filled by the tangent forward, reused by the reverse sweep) or the scratch set
  bool reverse = false;
  auto select_rows = [&](int id) -> bool {      // returns true when the rows are already there (reverse sweep, cached)
    float** dst[6] = {&t.Z, &t.Zd, &t.H, &t.Hd, &t.CG, &t.CGd};
    for (int q = 0; q < 6; ++q) *dst[q] = (t.cached && t.cache[id][q]) ? t.cache[id][q] : t.scratch6[q];
    return t.cached && reverse;
  };
  // ---- per-layer pieces ------------------------------------------------------------------------------------
  // tangent tables of AtomConv l:  Pd = atomd . [Wc;Wn]^T,  Qd = hbd . Wb^T  (node rows from hbcd[l])
  auto atom_tables_t = [&](int l) -> int {
    const ACW& aw = w.ac[l];
    TRY(rows_gemm_out2(eng, "t2_gemm_tab", t.atomd[l], nullptr, aw.w_cn, aw.w_cn + 2 * D * D, nullptr, t.Pd, 4 * D, N));
    TRY(gemm("t2_gemm_tab", 64, 128, t.hb0d, D, nullptr, aw.w_bond, nullptr, nullptr, 0, t.Qd, 2 * D, nullptr, Eu, 0));
    if (Eb > 0 && b->hbc[l] != b->hbc[0])
      TRY(gemm("t2_gemm_tab", 64, 128, t.hbcd[l], D, nullptr, aw.w_bond, nullptr, nullptr, 0, t.Qd, 2 * D, b->bn_und, Eb, 0));
    return CHG_OK;
  };
  // z, zd (and the hidden activations) of AtomConv l for every directed edge (centre-major order), then c|g and tangents
  auto atom_rows = [&](int l) -> int {
    const ACW& aw = w.ac[l];
    if (select_rows(l)) return CHG_OK;
    TRY(atom_tables_t(l));
    GatherZArgs a{};
    a.rows = Ed; a.t0 = b->Pl[l]; a.t1 = b->Pl[l]; a.t2 = b->Ql[l]; a.d0 = t.Pd; a.d1 = t.Pd; a.d2 = t.Qd;
    a.ld0 = 4 * D; a.ld1 = 4 * D; a.ld2 = 2 * D; a.off0 = 0; a.off1 = 2 * D; a.off2 = 0;
    a.i0 = b->e_center; a.i1 = b->e_nbr; a.i2 = b->e_d2u; a.hidden = 1; a.Z = t.Z; a.Zd = t.Zd; a.H = t.H; a.Hd = t.Hd;
    { LaunchScope ls(eng, "t2_gather_z");
      hipLaunchKernelGGL(k2_gather_z, wave_rows_grid(eng, Ed), dim3(256), 0, st, a); }
    TRY(gemm_pair(t.H, aw.g.w2c, aw.g.w2g, aw.g.b2c, aw.g.b2g, t.CG, Ed));
    return gemm_pair(t.Hd, aw.g.w2c, aw.g.w2g, nullptr, nullptr, t.CGd, Ed);
  };
  // the same for BondConv (hidden) / AngleUpdate (single layer) of slot; hrows / atoms / angs are the layer's inputs
  auto angle_rows = [&](int slot, bool hidden, const float* w_bij, const float* w_ctr, const float* w_ang, const GatedW& g, const float* hrowsd,
                        const float* atomsd, const float* angs, const float* angsd) -> int {
    if (select_rows(slot < L ? L + slot : 2 * L + (slot - L))) return CHG_OK;
    TRY(rows_gemm_out2(eng, "t2_gemm_tab", hrowsd, nullptr, w_bij, w_bij + 2 * D * D, nullptr, t.Rd, 4 * D, Eb));
    TRY(gemm("t2_gemm_tab", 64, 128, atomsd, D, nullptr, w_ctr, nullptr, nullptr, 0, t.Sd, 2 * D, nullptr, N, 0));
    TRY(gemm("t2_gemm_ang", 64, 128, angs, D, nullptr, w_ang, nullptr, nullptr, 0, t.ZA, 2 * D, nullptr, A, 0));
    TRY(gemm("t2_gemm_ang", 64, 128, angsd, D, nullptr, w_ang, nullptr, nullptr, 0, t.ZAd, 2 * D, nullptr, A, 0));
    GatherZArgs a{};
    a.rows = A; a.t0 = b->Rl[slot]; a.t1 = b->Rl[slot]; a.t2 = b->Sl[slot]; a.d0 = t.Rd; a.d1 = t.Rd; a.d2 = t.Sd;
    a.ld0 = 4 * D; a.ld1 = 4 * D; a.ld2 = 2 * D; a.off0 = 0; a.off1 = 2 * D; a.off2 = 0;
    a.i0 = b->a_b1c; a.i1 = b->a_b2c; a.i2 = b->a_ctr; a.add = t.ZA; a.addd = t.ZAd; a.hidden = hidden ? 1 : 0;
    a.Z = hidden ? t.Z : t.CG; a.Zd = hidden ? t.Zd : t.CGd; a.H = t.H; a.Hd = t.Hd;   // single layer: c|g IS z
    { LaunchScope ls(eng, "t2_gather_z");
      hipLaunchKernelGGL(k2_gather_z, wave_rows_grid(eng, A), dim3(256), 0, st, a); }
    if (!hidden) return CHG_OK;
    TRY(gemm_pair(t.H, g.w2c, g.w2g, g.b2c, g.b2g, t.CG, A));
    return gemm_pair(t.Hd, g.w2c, g.w2g, nullptr, nullptr, t.CGd, A);
  };

  // ---- tangent forward ---------------------------------------------------------------------------------------
  auto atom2_args = [&](int l) {
    Atom2Args a{};
    a.n_edges = Ed; a.e_center = b->p_center; a.e_nbr = b->p_nbr;
    a.P = b->Pl[l]; a.Q = b->Ql[l]; a.Pd = t.Pd; a.Qd = t.Qd; a.gw = w.ac[l].g; a.wag = b->wag; a.wagd = t.wagd;
    a.aggd = t.aggd[l]; a.bar_agg = t.bar_agg; a.g_agg = t.g_agg; a.bar_w = t.bar_wag;
    a.H = t.scratch6[2]; a.Hd = t.scratch6[3]; a.BCG = t.BCG; a.GCG = t.GCG;
    a.barP = t.barP; a.barQ = t.barQ; a.gQ = t.gQ; a.g_ln = G(w.ac[l].g.ln1_g);
    return a;
  };
  // tangent tables of an angle layer:  Rd = hrowsd . [Wi;Wj]^T,  Sd = atomsd . Wctr^T
  auto angle_tables_t = [&](const float* w_bij, const float* w_ctr, const float* hrowsd, const float* atomsd) -> int {
    TRY(rows_gemm_out2(eng, "t2_gemm_tab", hrowsd, nullptr, w_bij, w_bij + 2 * D * D, nullptr, t.Rd, 4 * D, Eb));
    return gemm("t2_gemm_tab", 64, 128, atomsd, D, nullptr, w_ctr, nullptr, nullptr, 0, t.Sd, 2 * D, nullptr, N, 0);
  };
  auto angle2_args = [&](int slot, const float* w_ang, const GatedW& g, const float* angs, const float* angsd) {
    Angle2Args a{};
    a.n_angles = A; a.a_ctr = b->a_ctr; a.a_b1c = b->a_b1c; a.a_b2c = b->a_b2c;
    a.R = b->Rl[slot]; a.S = b->Sl[slot]; a.Rd = t.Rd; a.Sd = t.Sd; a.ang = angs; a.angd = angsd; a.w_ang = w_ang; a.gw = g;
    a.w = b->wbgc; a.wd = t.wbgcd; a.bar_agg = t.bar_agg; a.g_agg = t.g_agg; a.bar_w = t.bar_wbg;
    a.bar_ang = t.bar_ang; a.g_ang = t.g_ang;
    a.H = t.scratch6[2]; a.Hd = t.scratch6[3]; a.BCG = t.BCG; a.GCG = t.GCG; a.BZ = t.BZ; a.GZ = t.GZ;
    a.barR = t.barR; a.barS = t.barS; a.g_ln = G(g.ln1_g);
    return a;
  };
  const dim3 angle_grid(tile_grid(eng, std::max(A, 1)));
  auto atomconv_t = [&](int l) -> int {
    const ACW& aw = w.ac[l];
    if (Ed > 0 && fused) {
      TRY(atom_tables_t(l));
      LaunchScope ls(eng, "t2_atom_t");
      hipLaunchKernelGGL(k2_atom<false>, dim3(tile_grid(eng, Ed)), dim3(BLOCK), t2_atom_lds(), st, atom2_args(l));
      HIP_TRY(eng, hipGetLastError());
    } else if (Ed > 0) {
      TRY(atom_rows(l));
      GatedTArgs a{};
      a.rows = Ed; a.mode = T2_ATOM; a.CG = t.CG; a.CGd = t.CGd; a.ln = aw.g.ln1_g; a.i_dst = b->e_center; a.i_w1 = b->e_d2u;
      a.w = b->wag; a.wd = t.wagd; a.aggd = t.aggd[l];
      LaunchScope ls(eng, "t2_gated_t");
      hipLaunchKernelGGL(k2_gated_t, wave_rows_grid(eng, Ed), dim3(256), 0, st, a);
    }
    return gemm("t2_gemm_out", 64, 64, t.aggd[l], D, nullptr, aw.w_out, nullptr, t.atomd[l], D, t.atomd[l + 1], D, nullptr, N, 0);
  };
  for (int l = 0; l < L - 1; ++l) {
    TRY(atomconv_t(l));
    if (angles) {
      const BCW& bw = w.bc[l];
      if (fused) {
        TRY(angle_tables_t(bw.w_bij, bw.w_ctr, t.hbcd[l], t.atomd[l + 1]));
        Angle2Args a = angle2_args(l, bw.w_ang, bw.g, b->ang[l], t.angd[l]);
        a.aggd = t.aggBd[l];
        LaunchScope ls(eng, "t2_bond_t");
        hipLaunchKernelGGL((k2_angle<true, false>), angle_grid, dim3(BLOCK), t2_angle_lds<true>(), st, a);
        HIP_TRY(eng, hipGetLastError());
      } else {
      TRY(angle_rows(l, true, bw.w_bij, bw.w_ctr, bw.w_ang, bw.g, t.hbcd[l], t.atomd[l + 1], b->ang[l], t.angd[l]));
      {
        GatedTArgs a{};
        a.rows = A; a.mode = T2_BOND; a.CG = t.CG; a.CGd = t.CGd; a.ln = bw.g.ln1_g; a.i_dst = b->a_b1c; a.i_w1 = b->a_b1c; a.i_w2 = b->a_b2c;
        a.w = b->wbgc; a.wd = t.wbgcd; a.aggd = t.aggBd[l];
        LaunchScope ls(eng, "t2_gated_t");
        hipLaunchKernelGGL(k2_gated_t, wave_rows_grid(eng, A), dim3(256), 0, st, a);
      }
      }
      TRY(gemm("t2_gemm_out", 64, 64, t.aggBd[l], D, nullptr, bw.w_out, nullptr, t.hbcd[l], D, t.hbcd[l + 1], D, nullptr, Eb, 0));
      if (l < L - 2 && fused) {
        const AUW& uw = w.au[l];
        TRY(angle_tables_t(uw.w_bij, uw.w_ctr, t.hbcd[l + 1], t.atomd[l + 1]));
        Angle2Args a = angle2_args(L + l, uw.w_ang, uw.g, b->ang[l], t.angd[l]);
        a.angd_out = t.angd[l + 1];
        LaunchScope ls(eng, "t2_angle_t");
        hipLaunchKernelGGL((k2_angle<false, false>), angle_grid, dim3(BLOCK), t2_angle_lds<false>(), st, a);
        HIP_TRY(eng, hipGetLastError());
      } else if (l < L - 2) {
        const AUW& uw = w.au[l];
        TRY(angle_rows(L + l, false, uw.w_bij, uw.w_ctr, uw.w_ang, uw.g, t.hbcd[l + 1], t.atomd[l + 1], b->ang[l], t.angd[l]));
        GatedTArgs a{};
        a.rows = A; a.mode = T2_ANGLE; a.CG = t.CG; a.CGd = t.CGd; a.ln = uw.g.ln1_g; a.angd_in = t.angd[l]; a.angd_out = t.angd[l + 1];
        LaunchScope ls(eng, "t2_gated_t");
        hipLaunchKernelGGL(k2_gated_t, wave_rows_grid(eng, A), dim3(256), 0, st, a);
      }
    } else if (Eb > 0) {
      HIP_TRY(eng, hipMemcpyAsync(t.hbcd[l + 1], t.hbcd[l], sizeof(float) * (size_t)Eb * D, hipMemcpyDeviceToDevice, st));
    }
  }
  TRY(atomconv_t(L - 1));
  TRY(check());

  // ---- readout: tangent forward, seeds, two-adjoint backward --------------------------------------------------
  enum { X0 = 0, X0D, XH, XHD, L0, L0D, L1, L1D, L2, L2D, S1, S1D, S2, S2D, S3, S3D, BS, GS, BL, GLr, DW3, DGAM, DBET, TMP0, TMP1, TMP2 };
  const size_t nd = (size_t)N * D;
  {
    { LaunchScope ls(eng, "t2_readout");
      hipLaunchKernelGGL(k2_ln_t, wave_rows_grid(eng, N), dim3(256), 0, st, b->atom[L], t.atomd[L], w.ro_ln_g, w.ro_ln_b, t.ro[X0], t.ro[X0D],
                         t.ro[XH], t.ro[XHD], N); }
    const float* Wm[3] = {w.mlp_w0, w.mlp_w1, w.mlp_w2};
    const float* Wt[3] = {w.mlp_w0_t, w.mlp_w1_t, w.mlp_w2_t};
    const float* bm[3] = {w.mlp_b0, w.mlp_b1, w.mlp_b2};
    const int sidx[4] = {X0, S1, S2, S3}, sdidx[4] = {X0D, S1D, S2D, S3D}, lidx[3] = {L0, L1, L2}, ldidx[3] = {L0D, L1D, L2D};
    for (int i = 0; i < 3; ++i) {
      TRY(gemm("t2_readout", 64, 64, t.ro[sidx[i]], D, nullptr, Wm[i], bm[i], nullptr, 0, t.ro[lidx[i]], D, nullptr, N, 0));
      TRY(gemm("t2_readout", 64, 64, t.ro[sdidx[i]], D, nullptr, Wm[i], nullptr, nullptr, 0, t.ro[ldidx[i]], D, nullptr, N, 0));
      LaunchScope ls(eng, "t2_readout");
      hipLaunchKernelGGL(k2_silu_t, g1((int64_t)nd), dim3(256), 0, st, t.ro[lidx[i]], t.ro[ldidx[i]], t.ro[sidx[i + 1]], t.ro[sdidx[i + 1]], nd);
    }
    { LaunchScope ls(eng, "t2_readout");
      hipLaunchKernelGGL(k2_readout_seed, g1((int64_t)nd), dim3(256), 0, st, w.mlp_w3, b->t_cot, b->atom_owner, t.ro[S3], t.ro[S3D], t.ro[BS],
                         t.ro[GS], t.ro[DW3], N); }
    TRY(colsum(eng, t.ro[DW3], D, nullptr, 0, N, D, G(w.mlp_w3)));
    const float* gW[3] = {G(w.mlp_w0), G(w.mlp_w1), G(w.mlp_w2)};
    const float* gb[3] = {G(w.mlp_b0), G(w.mlp_b1), G(w.mlp_b2)};
    for (int i = 2; i >= 0; --i) {
      { LaunchScope ls(eng, "t2_readout");
        hipLaunchKernelGGL(k2_hidden_b, g1((int64_t)nd), dim3(256), 0, st, t.ro[lidx[i]], t.ro[ldidx[i]], t.ro[BS], t.ro[GS], t.ro[BL], t.ro[GLr], nd); }
      TRY((xty<4, 4>(eng, "t2_wgrad", t.ro[BL], D, nullptr, t.ro[sidx[i]], D, nullptr, N, 1.0f, (float*)gW[i], D, D, (float*)gb[i])));
      TRY((xty<4, 4>(eng, "t2_wgrad", t.ro[GLr], D, nullptr, t.ro[sdidx[i]], D, nullptr, N, 1.0f, (float*)gW[i], D, D)));
      TRY(gemm("t2_readout", 64, 64, t.ro[BL], D, nullptr, Wt[i], nullptr, nullptr, 0, t.ro[BS], D, nullptr, N, 0));
      TRY(gemm("t2_readout", 64, 64, t.ro[GLr], D, nullptr, Wt[i], nullptr, nullptr, 0, t.ro[GS], D, nullptr, N, 0));
    }
    { LaunchScope ls(eng, "t2_readout");
      hipLaunchKernelGGL(k2_ln_b, wave_rows_grid(eng, N), dim3(256), 0, st, b->atom[L], t.atomd[L], w.ro_ln_g, t.ro[BS], t.ro[GS], t.bar_a, t.g_a,
                         t.ro[DGAM], t.ro[DBET], N); }
    TRY(colsum(eng, t.ro[DGAM], D, nullptr, 0, N, D, G(w.ro_ln_g)));
    TRY(colsum(eng, t.ro[DBET], D, nullptr, 0, N, D, G(w.ro_ln_b)));
  }
  TRY(check());

  // ---- reverse sweep with two adjoints -----------------------------------------------------------------------
  reverse = true;
  // gated-MLP internals common to the three layer kinds: BCG / GCG -> weight gradients of the second layer, BZ / GZ
  auto hidden_back = [&](const GatedW& g, const float* w2c_t, const float* w2g_t, int rows) -> int {
    TRY(xty_halves(eng, "t2_wgrad", t.BCG, t.H, rows, G(g.w2c), G(g.w2g), G(g.b2c), G(g.b2g)));
    TRY(xty_halves(eng, "t2_wgrad", t.GCG, t.Hd, rows, G(g.w2c), G(g.w2g)));
    TRY(gemm_pair(t.BCG, w2c_t, w2g_t, nullptr, nullptr, t.BH, rows));
    TRY(gemm_pair(t.GCG, w2c_t, w2g_t, nullptr, nullptr, t.GH, rows));
    LaunchScope ls(eng, "t2_hidden_b");
    hipLaunchKernelGGL(k2_hidden_b, g1((int64_t)rows * 2 * D), dim3(256), 0, st, t.Z, t.Zd, t.BH, t.GH, t.BZ, t.GZ, (size_t)rows * 2 * D);
    return check();
  };

  auto atomconv_b = [&](int l) -> int {
    const ACW& aw = w.ac[l];
    // atom[l+1] = agg . Wout^T + b_out + atom[l]
    TRY((xty<4, 4>(eng, "t2_wgrad", t.bar_a, D, nullptr, b->agg_l[l], D, nullptr, N, 1.0f, G(aw.w_out), D, D, G(aw.b_out))));
    TRY((xty<4, 4>(eng, "t2_wgrad", t.g_a, D, nullptr, t.aggd[l], D, nullptr, N, 1.0f, G(aw.w_out), D, D)));
    if (Ed == 0) return CHG_OK;
    TRY(gemm("t2_gemm_out", 64, 64, t.bar_a, D, nullptr, aw.w_out_t, nullptr, nullptr, 0, t.bar_agg, D, nullptr, N, 0));
    TRY(gemm("t2_gemm_out", 64, 64, t.g_a, D, nullptr, aw.w_out_t, nullptr, nullptr, 0, t.g_agg, D, nullptr, N, 0));
    table_adjoints_of(l, -1);
    if (fused) {
      TRY(atom_tables_t(l));
      TRY(zero(eng, t.barP, sizeof(float) * (size_t)N * 4 * D));
      const Atom2Args a = atom2_args(l);
      { LaunchScope ls(eng, "t2_atom_b");
        hipLaunchKernelGGL(k2_atom<true>, dim3(tile_grid(eng, Ed)), dim3(BLOCK), t2_atom_lds(), st, a);
        HIP_TRY(eng, hipGetLastError()); }
      TRY(xty_halves(eng, "t2_wgrad", a.BCG, a.H, Ed, G(aw.g.w2c), G(aw.g.w2g), G(aw.g.b2c), G(aw.g.b2g)));
      TRY(xty_halves(eng, "t2_wgrad", a.GCG, a.Hd, Ed, G(aw.g.w2c), G(aw.g.w2g)));
    } else {
    TRY(atom_rows(l));
    {
      GatedBArgs a{};
      a.rows = Ed; a.mode = T2_ATOM; a.CG = t.CG; a.CGd = t.CGd; a.ln = aw.g.ln1_g; a.i_dst = b->e_center; a.i_w1 = b->e_d2u;
      a.w = b->wag; a.wd = t.wagd; a.bar_agg = t.bar_agg; a.g_agg = t.g_agg; a.bar_w = t.bar_wag; a.g_w = t.g_wag;
      a.BCG = t.BCG; a.GCG = t.GCG; a.g_ln = G(aw.g.ln1_g);
      LaunchScope ls(eng, "t2_gated_b");
      hipLaunchKernelGGL(k2_gated_b, wave_rows_grid(eng, Ed), dim3(256), 0, st, a);
    }
    TRY(hidden_back(aw.g, aw.w2c_t, aw.w2g_t, Ed));
    TRY(zero(eng, t.barP, sizeof(float) * (size_t)N * 4 * D)); TRY(zero(eng, t.gP, sizeof(float) * (size_t)N * 4 * D));
    TRY(zero(eng, t.barQ, sizeof(float) * (size_t)Eu * 2 * D)); TRY(zero(eng, t.gQ, sizeof(float) * (size_t)Eu * 2 * D));
    {
      ScatterZArgs a{Ed, t.BZ, t.GZ, t.barP, t.barP, t.barQ, t.gP, t.gP, t.gQ, 4 * D, 4 * D, 2 * D, 0, 2 * D, 0, b->e_center, b->e_nbr, b->e_d2u};
      LaunchScope ls(eng, "t2_scatter_z");
      hipLaunchKernelGGL(k2_scatter_z, wave_rows_grid(eng, (Ed + TILE_ROWS - 1) / TILE_ROWS), dim3(256), scatter_z_lds(), st, a);
    }
    }
    // first layer (factorised): table gradients contract with the rows the tables were made from, bar with primal and G with tangent
    for (int half = 0; half < 2; ++half) {
      TRY((xty<8, 4>(eng, "t2_wgrad", t.barP + half * 2 * D, 4 * D, nullptr, b->atom[l], D, nullptr, N, 1.0f, G(aw.w_cn) + half * 2 * D * D, D, D,
                     half == 0 ? G(aw.b1) : nullptr)));
      TRY((xty<8, 4>(eng, "t2_wgrad", t.gP + half * 2 * D, 4 * D, nullptr, t.atomd[l], D, nullptr, N, 1.0f, G(aw.w_cn) + half * 2 * D * D, D, D)));
    }
    TRY((xty<8, 4>(eng, "t2_wgrad", t.barQ, 2 * D, nullptr, b->hb0, D, nullptr, Eu, 1.0f, G(aw.w_bond), D, D)));
    TRY((xty<8, 4>(eng, "t2_wgrad", t.gQ, 2 * D, nullptr, t.hb0d, D, nullptr, Eu, 1.0f, G(aw.w_bond), D, D)));
    if (Eb > 0 && b->hbc[l] != b->hbc[0]) {
      TRY((xty<8, 4>(eng, "t2_wgrad", t.barQ, 2 * D, b->bn_und, b->hbc[l], D, nullptr, Eb, 1.0f, G(aw.w_bond), D, D)));
      TRY((xty<8, 4>(eng, "t2_wgrad", t.barQ, 2 * D, b->bn_und, b->hb0, D, b->bn_und, Eb, -1.0f, G(aw.w_bond), D, D)));
      TRY((xty<8, 4>(eng, "t2_wgrad", t.gQ, 2 * D, b->bn_und, t.hbcd[l], D, nullptr, Eb, 1.0f, G(aw.w_bond), D, D)));
      TRY((xty<8, 4>(eng, "t2_wgrad", t.gQ, 2 * D, b->bn_und, t.hb0d, D, b->bn_und, Eb, -1.0f, G(aw.w_bond), D, D)));
    }
    TRY(rows_gemm_in2(eng, "t2_gemm_tab", t.barP, 4 * D, aw.w_cn_t, aw.w_cn_t + 2 * D * D, t.bar_a, nullptr, N, 1));
    TRY(rows_gemm_in2(eng, "t2_gemm_tab", t.gP, 4 * D, aw.w_cn_t, aw.w_cn_t + 2 * D * D, t.g_a, nullptr, N, 1));
    TRY(gemm("t2_gemm_tab", 128, 64, t.barQ, 2 * D, nullptr, aw.w_bond_t, nullptr, nullptr, 0, t.bar_b, D, nullptr, Eu, 1));
    return gemm("t2_gemm_tab", 128, 64, t.gQ, 2 * D, nullptr, aw.w_bond_t, nullptr, nullptr, 0, t.g_b, D, nullptr, Eu, 1);
  };

  // tail shared by BondConv / AngleUpdate: BZ / GZ [A,128] -> table gradients, weight gradients, adjoints of the inputs
  auto angle_back = [&](const float* w_bij, const float* w_ctr, const float* b1, const float* w_ang, const float* w_bij_t, const float* w_ctr_t,
                        const float* w_ang_t, const float* hrows, const float* hrowsd, const float* atoms, const float* atomsd,
                        const float* angs, const float* angsd, const std::function<int()>& fused_kernel) -> int {
    TRY(zero(eng, t.barR, sizeof(float) * (size_t)Eb * 4 * D));
    TRY(zero(eng, t.barS, sizeof(float) * (size_t)N * 2 * D));
    if (!fused) { TRY(zero(eng, t.gR, sizeof(float) * (size_t)Eb * 4 * D)); TRY(zero(eng, t.gS, sizeof(float) * (size_t)N * 2 * D)); }
    if (fused) {
      TRY(fused_kernel());     // first-layer adjoints scattered to the tables and contracted back to the angle features in the kernel
    } else {
      ScatterZArgs a{A, t.BZ, t.GZ, t.barR, t.barR, t.barS, t.gR, t.gR, t.gS, 4 * D, 4 * D, 2 * D, 0, 2 * D, 0, b->a_b1c, b->a_b2c, b->a_ctr};
      LaunchScope ls(eng, "t2_scatter_z");
      hipLaunchKernelGGL(k2_scatter_z, wave_rows_grid(eng, (A + TILE_ROWS - 1) / TILE_ROWS), dim3(256), scatter_z_lds(), st, a);
    }
    for (int half = 0; half < 2; ++half) {
      TRY((xty<8, 4>(eng, "t2_wgrad", t.barR + half * 2 * D, 4 * D, nullptr, hrows, D, nullptr, Eb, 1.0f, G(w_bij) + half * 2 * D * D, D, D)));
      TRY((xty<8, 4>(eng, "t2_wgrad", t.gR + half * 2 * D, 4 * D, nullptr, hrowsd, D, nullptr, Eb, 1.0f, G(w_bij) + half * 2 * D * D, D, D)));
    }
    TRY((xty<8, 4>(eng, "t2_wgrad", t.barS, 2 * D, nullptr, atoms, D, nullptr, N, 1.0f, G(w_ctr), D, D, G(b1))));
    TRY((xty<8, 4>(eng, "t2_wgrad", t.gS, 2 * D, nullptr, atomsd, D, nullptr, N, 1.0f, G(w_ctr), D, D)));
    TRY((xty<8, 4>(eng, "t2_wgrad", t.BZ, 2 * D, nullptr, angs, D, nullptr, A, 1.0f, G(w_ang), D, D)));
    TRY((xty<8, 4>(eng, "t2_wgrad", t.GZ, 2 * D, nullptr, angsd, D, nullptr, A, 1.0f, G(w_ang), D, D)));
    TRY(rows_gemm_in2(eng, "t2_gemm_tab", t.barR, 4 * D, w_bij_t, w_bij_t + 2 * D * D, t.bar_b, b->bn_und, Eb, 1));
    TRY(rows_gemm_in2(eng, "t2_gemm_tab", t.gR, 4 * D, w_bij_t, w_bij_t + 2 * D * D, t.g_b, b->bn_und, Eb, 1));
    TRY(gemm("t2_gemm_tab", 128, 64, t.barS, 2 * D, nullptr, w_ctr_t, nullptr, nullptr, 0, t.bar_a, D, nullptr, N, 1));
    TRY(gemm("t2_gemm_tab", 128, 64, t.gS, 2 * D, nullptr, w_ctr_t, nullptr, nullptr, 0, t.g_a, D, nullptr, N, 1));
    if (fused) return CHG_OK;
    TRY(gemm("t2_gemm_ang", 128, 64, t.BZ, 2 * D, nullptr, w_ang_t, nullptr, nullptr, 0, t.bar_ang, D, nullptr, A, 1));
    return gemm("t2_gemm_ang", 128, 64, t.GZ, 2 * D, nullptr, w_ang_t, nullptr, nullptr, 0, t.g_ang, D, nullptr, A, 1);
  };

  TRY(atomconv_b(L - 1));
  if (b->t_has_mcot) {   // magmom head reads atom[L-1]: first-order term, joins bar(atom[L-1])
    LaunchScope ls(eng, "magmom_bwd");
    hipLaunchKernelGGL(k_magmom_bwd, dim3(wave_grid(eng, N)), dim3(256), 0, st, b->atom[L - 1], w.site_w, w.site_b, b->t_mcot, t.bar_a, G(w.site_w),
                       G(w.site_b), N);
  }
  for (int l = L - 2; l >= 0; --l) {
    if (angles) {
      if (l < L - 2) {
        const AUW& uw = w.au[l];
        if (fused) {
          TRY(angle_tables_t(uw.w_bij, uw.w_ctr, t.hbcd[l + 1], t.atomd[l + 1]));
        } else {
        TRY(angle_rows(L + l, false, uw.w_bij, uw.w_ctr, uw.w_ang, uw.g, t.hbcd[l + 1], t.atomd[l + 1], b->ang[l], t.angd[l]));
        GatedBArgs a{};
        a.rows = A; a.mode = T2_ANGLE; a.CG = t.CG; a.CGd = t.CGd; a.ln = uw.g.ln1_g; a.bar_agg = t.bar_ang; a.g_agg = t.g_ang;
        a.BCG = t.BZ; a.GCG = t.GZ; a.g_ln = G(uw.g.ln1_g);      // single layer: bar(c|g) IS bar(z)
        { LaunchScope ls(eng, "t2_gated_b");
          hipLaunchKernelGGL(k2_gated_b, wave_rows_grid(eng, A), dim3(256), 0, st, a); }
        }
        table_adjoints_of(-1, L + l);
        TRY(angle_back(uw.w_bij, uw.w_ctr, uw.b1, uw.w_ang, uw.w_bij_t, uw.w_ctr_t, uw.w_ang_t, b->hbc[l + 1], t.hbcd[l + 1], b->atom[l + 1],
                       t.atomd[l + 1], b->ang[l], t.angd[l], [&]() -> int {
                         LaunchScope ls(eng, "t2_angle_b");
                         hipLaunchKernelGGL((k2_angle<false, true>), angle_grid, dim3(BLOCK), t2_angle_lds<false>(), st,
                                            angle2_args(L + l, uw.w_ang, uw.g, b->ang[l], t.angd[l]));
                         HIP_TRY(eng, hipGetLastError());
                         return CHG_OK;
                       }));
      }
      const BCW& bw = w.bc[l];
      // hbc[l+1] = aggB . Wout^T + hbc[l]; its adjoints live in the node rows of bar_b / g_b
      TRY((xty<4, 4>(eng, "t2_wgrad", t.bar_b, D, b->bn_und, b->aggB_l[l], D, nullptr, Eb, 1.0f, G(bw.w_out), D, D)));
      TRY((xty<4, 4>(eng, "t2_wgrad", t.g_b, D, b->bn_und, t.aggBd[l], D, nullptr, Eb, 1.0f, G(bw.w_out), D, D)));
      TRY(gemm("t2_gemm_out", 64, 64, t.bar_b, D, b->bn_und, bw.w_out_t, nullptr, nullptr, 0, t.bar_agg, D, nullptr, Eb, 0));
      TRY(gemm("t2_gemm_out", 64, 64, t.g_b, D, b->bn_und, bw.w_out_t, nullptr, nullptr, 0, t.g_agg, D, nullptr, Eb, 0));
      if (fused) {
        TRY(angle_tables_t(bw.w_bij, bw.w_ctr, t.hbcd[l], t.atomd[l + 1]));
      } else {
      TRY(angle_rows(l, true, bw.w_bij, bw.w_ctr, bw.w_ang, bw.g, t.hbcd[l], t.atomd[l + 1], b->ang[l], t.angd[l]));
      {
        GatedBArgs a{};
        a.rows = A; a.mode = T2_BOND; a.CG = t.CG; a.CGd = t.CGd; a.ln = bw.g.ln1_g; a.i_dst = b->a_b1c; a.i_w1 = b->a_b1c; a.i_w2 = b->a_b2c;
        a.w = b->wbgc; a.wd = t.wbgcd; a.bar_agg = t.bar_agg; a.g_agg = t.g_agg; a.bar_w = t.bar_wbg; a.g_w = t.g_wbg;
        a.BCG = t.BCG; a.GCG = t.GCG; a.g_ln = G(bw.g.ln1_g);
        LaunchScope ls(eng, "t2_gated_b");
        hipLaunchKernelGGL(k2_gated_b, wave_rows_grid(eng, A), dim3(256), 0, st, a);
      }
      TRY(hidden_back(bw.g, bw.w2c_t, bw.w2g_t, A));
      }
      table_adjoints_of(-1, l);
      TRY(angle_back(bw.w_bij, bw.w_ctr, bw.b1, bw.w_ang, bw.w_bij_t, bw.w_ctr_t, bw.w_ang_t, b->hbc[l], t.hbcd[l], b->atom[l + 1], t.atomd[l + 1],
                     b->ang[l], t.angd[l], [&]() -> int {
                       const Angle2Args a = angle2_args(l, bw.w_ang, bw.g, b->ang[l], t.angd[l]);
                       { LaunchScope ls(eng, "t2_bond_b");
                         hipLaunchKernelGGL((k2_angle<true, true>), angle_grid, dim3(BLOCK), t2_angle_lds<true>(), st, a);
                         HIP_TRY(eng, hipGetLastError()); }
                       TRY(xty_halves(eng, "t2_wgrad", a.BCG, a.H, A, G(bw.g.w2c), G(bw.g.w2g), G(bw.g.b2c), G(bw.g.b2g)));
                       return xty_halves(eng, "t2_wgrad", a.GCG, a.Hd, A, G(bw.g.w2c), G(bw.g.w2g));
                     }));
    }
    TRY(atomconv_b(l));
  }
  TRY(check());

  // ---- embeddings: 31 -> 64 linears (bar with basis, G with basis tangent), frequencies, atom embedding table ----
  if (Eu > 0) {
    TRY((xty<4, 2>(eng, "t2_wgrad", t.bar_b, D, nullptr, t.X6, KB2, nullptr, Eu, 1.0f, G(w.w_bond_emb), NRAD, NRAD)));
    TRY((xty<4, 2>(eng, "t2_wgrad", t.g_b, D, nullptr, t.X6d, KB2, nullptr, Eu, 1.0f, G(w.w_bond_emb), NRAD, NRAD)));
    TRY((xty<4, 2>(eng, "t2_wgrad", t.bar_wag, D, nullptr, t.X6, KB2, nullptr, Eu, 1.0f, G(w.w_wag), NRAD, NRAD)));
    TRY((xty<4, 2>(eng, "t2_wgrad", g_wag, D, nullptr, t.X6d, KB2, nullptr, Eu, 1.0f, G(w.w_wag), NRAD, NRAD)));
    TRY((xty<4, 2>(eng, "t2_wgrad", t.bar_wbg, D, nullptr, t.X3, KB2, b->bn_und, Eb, 1.0f, G(w.w_wbg), NRAD, NRAD)));
    TRY((xty<4, 2>(eng, "t2_wgrad", g_wbg, D, nullptr, t.X3d, KB2, b->bn_und, Eb, 1.0f, G(w.w_wbg), NRAD, NRAD)));
    {
      FreqGradArgs a{Eu, nullptr, b->ev, t.vd4, b->u_u2d, w.freq_ag, eng->desc.atom_graph_cutoff, env, t.bar_b, t.g_b, w.w_bond_emb,
                     t.bar_wag, g_wag, w.w_wag, G(w.freq_ag)};
      LaunchScope ls(eng, "t2_freq");
      hipLaunchKernelGGL(k2_freq_grad, wave_rows_grid(eng, Eu), dim3(256), 0, st, a);
    }
    if (Eb > 0) {
      FreqGradArgs a{Eb, b->bn_und, b->ev, t.vd4, b->u_u2d, w.freq_bg, eng->desc.bond_graph_cutoff, env, t.bar_wbg, g_wbg, w.w_wbg,
                     nullptr, nullptr, nullptr, G(w.freq_bg)};
      LaunchScope ls(eng, "t2_freq");
      hipLaunchKernelGGL(k2_freq_grad, wave_rows_grid(eng, Eb), dim3(256), 0, st, a);
    }
  }
  if (angles) {
    TRY((xty<4, 2>(eng, "t2_wgrad", t.bar_ang, D, nullptr, t.X4, KB2, nullptr, A, 1.0f, G(w.w_ang_emb), NANG, NANG)));
    TRY((xty<4, 2>(eng, "t2_wgrad", t.g_ang, D, nullptr, t.X4d, KB2, nullptr, A, 1.0f, G(w.w_ang_emb), NANG, NANG)));
    LaunchScope ls(eng, "t2_freq");
    hipLaunchKernelGGL(k2_angle_freq_grad, wave_rows_grid(eng, A), dim3(256), 0, st, t.bar_ang, t.g_ang, w.w_ang_emb, t.th2, w.freq_ang,
                       G(w.freq_ang), A);
  }
  {
    LaunchScope ls(eng, "wgrad_atom_embed");
    hipLaunchKernelGGL(k_embed_grad, g1((int64_t)N * D), dim3(256), 0, st, t.bar_a, b->z, G(w.emb), N);
  }
  return check();
}

template <class K>
int set_lds(chg_engine* eng, K kernel, size_t bytes) {
  HIP_TRY(eng, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return CHG_OK;
}

}  // namespace


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

int exclusive_scan(chg_engine* eng, TmpPool& tmp, const int* in, int* out, int n) {
  if (n <= 0) return CHG_OK;
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
struct GraphCounts { int Ed = 0, A = 0, Eb = 0, unpaired = 0, isolated = 0; bool cell_overflow = false; };

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
               int*& a_b1, int*& a_d1, int*& a_b2, int*& a_d2) {
  const int N = h->n_atoms;
  hipStream_t st = eng->stream;
  overflowed = false;
  int* d_ccnt = tmp.get<int>(N + 1);
  int* d_coff = tmp.get<int>(N + 1);
  int* d_flags = tmp.get<int>(4);   // [0] unpaired directed edge, [1] isolated atoms, [2] speculative capacity exceeded
  int* d_counts = tmp.get<int>(8);
  if (!d_ccnt || !d_coff || !d_flags || !d_counts) { eng->err = "graph build: scratch allocation failed"; return CHG_ENOMEM; }
  HIP_TRY(eng, hipMemsetAsync(d_ccnt, 0, sizeof(int) * (N + 1), st));
  HIP_TRY(eng, hipMemsetAsync(d_flags, 0, sizeof(int) * 4, st));
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
  int* is_first = tmp.get<int>(capE + 1); int* first_scan = tmp.get<int>(capE + 1);
  p_center = tmp.get<int>(capE); p_nbr = tmp.get<int>(capE);
  u_u2d = tmp.get<int>(capU);
  int* short_cnt = tmp.get<int>(N); int* ang_cnt = tmp.get<int>(capU + 1); int* ang_off = tmp.get<int>(capU + 1);
  int* is_node = tmp.get<int>(capU + 1); int* node_scan = tmp.get<int>(capU + 1);
  u_bnode = tmp.get<int>(capU);
  if (!e_center || !e_nbr || !e_img || !e_image || !e_dist || !e_owner || !e_rev || !e_d2u || !is_first || !first_scan || !p_center ||
      !p_nbr || !u_u2d || !short_cnt || !ang_cnt || !ang_off || !is_node || !node_scan || !u_bnode) {
    eng->err = "graph build: scratch allocation failed";
    return CHG_ENOMEM;
  }
  HIP_TRY(eng, hipMemsetAsync(ang_cnt, 0, sizeof(int) * (capU + 1), st));
  HIP_TRY(eng, hipMemsetAsync(is_node, 0, sizeof(int) * (capU + 1), st));
  HIP_TRY(eng, hipMemsetAsync(is_first, 0, sizeof(int) * (capE + 1), st));
  HIP_TRY(eng, hipMemsetAsync(short_cnt, 0, sizeof(int) * std::max(N, 1), st));
  if (capE > 0) {
    nl.center_off = d_coff; nl.e_center = e_center; nl.e_nbr = e_nbr; nl.e_img = e_img; nl.e_image = e_image; nl.e_dist = e_dist;
    nl.e_owner = e_owner; nl.cap_edges = capE;
    hipLaunchKernelGGL((k_neighbors<true>), wave_per_atom, dim3(256), 0, st, nl);
    hipLaunchKernelGGL(k_reverse, g1(capE), dim3(256), 0, st, e_center, e_nbr, e_img, d_coff, nE, e_rev, is_first, d_flags);
    TRY(exclusive_scan(eng, tmp, is_first, first_scan, capE + 1));
    hipLaunchKernelGGL(k_undirected, g1(capE), dim3(256), 0, st, e_center, e_nbr, e_rev, is_first, first_scan, nE, e_d2u, u_u2d, p_center, p_nbr,
                       d_flags + 2);
  }
  hipLaunchKernelGGL(k_short_count, g1((int64_t)N * 64), dim3(256), 0, st, (const double*)e_dist, (const int*)d_coff, N, r_bond, short_cnt, d_flags + 1,
                     d_flags + 2);
  int A = 0, Eb = 0;
  if (capU > 0) {
    hipLaunchKernelGGL(k_angle_count, g1(capU), dim3(256), 0, st, u_u2d, e_rev, e_center, e_dist, short_cnt, nU, r_bond, ang_cnt, d_flags + 2);
    TRY(exclusive_scan(eng, tmp, ang_cnt, ang_off, capU + 1));   // entries past Eu are zero: the total sits at ang_off[capU]
  } else {
    HIP_TRY(eng, hipMemsetAsync(ang_off, 0, sizeof(int) * (capU + 1), st));
  }
  int flags[4] = {0, 0, 0, 0};
  if (!speculative) {
    HIP_TRY(eng, hipMemcpyAsync(&A, ang_off + capU, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(eng, hipMemcpyAsync(flags, d_flags, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(eng, hipStreamSynchronize(st));
    if (flags[3]) { gc.cell_overflow = true; overflowed = true; return CHG_OK; }
    if (flags[0]) { eng->err = "graph build: number of directed edges != 2 * number of undirected edges (directed edges are not complete)"; return CHG_EINVAL; }
    capA = A;
  }
  a_ctr = tmp.get<int>(capA); a_b1 = tmp.get<int>(capA); a_d1 = tmp.get<int>(capA); a_b2 = tmp.get<int>(capA); a_d2 = tmp.get<int>(capA);
  if (!a_ctr || !a_b1 || !a_d1 || !a_b2 || !a_d2) { eng->err = "graph build: scratch allocation failed"; return CHG_ENOMEM; }
  if (capA > 0 && capU > 0) {
    hipLaunchKernelGGL(k_angle_fill, g1((int64_t)capU * 64), dim3(256), 0, st, u_u2d, e_rev, e_center, e_d2u, e_dist, d_coff, ang_off, nU, r_bond, a_ctr, a_b1,
                       a_d1, a_b2, a_d2, is_node, capA, d_flags + 2);
    TRY(exclusive_scan(eng, tmp, is_node, node_scan, capU + 1));
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
  if (capU > 0) hipLaunchKernelGGL(k_bond_nodes, g1(capU), dim3(256), 0, st, is_node, node_scan, nU, u_bnode, bn_und, capEb, d_flags + 2);
  HIP_TRY(eng, hipGetLastError());
  if (speculative) {   // the one round trip of this path
    int hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipLaunchKernelGGL(k_collect_counts, dim3(1), dim3(64), 0, st, (const int*)(d_coff + N), (const int*)(ang_off + capU),
                       (const int*)(node_scan + capU), (const int*)d_flags, d_counts);
    HIP_TRY(eng, hipMemcpyAsync(hc, d_counts, sizeof(int) * 7, hipMemcpyDeviceToHost, st));
    HIP_TRY(eng, hipStreamSynchronize(st));
    Ed = hc[0]; A = hc[1]; Eb = hc[2]; flags[0] = hc[3]; flags[1] = hc[4]; flags[2] = hc[5];
    if (hc[6]) { gc.cell_overflow = true; overflowed = true; return CHG_OK; }
    if (flags[2] || Ed > capE || A > capA || Eb > capEb) { overflowed = true; return CHG_OK; }
    if (Ed & 1) { eng->err = "graph build: odd number of directed edges"; return CHG_EINVAL; }
    if (flags[0]) { eng->err = "graph build: number of directed edges != 2 * number of undirected edges (directed edges are not complete)"; return CHG_EINVAL; }
  }
  gc.Ed = Ed; gc.A = A; gc.Eb = Eb; gc.unpaired = flags[0]; gc.isolated = flags[1];
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
                   a_b1, a_d1, a_b2, a_d2));
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
    size_t total = 0;
    carve(b, nullptr, total);
    int s = acquire_arena(eng, b, total);
    if (s != CHG_OK) { delete b; return s; }
    carve(b, b->arena, total);
    register_names(b);
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
      static_assert(MULTI_COPY_MAX >= 17, "one slot per array");
      if (nseg > 0) {
        const unsigned gx = (unsigned)std::min<unsigned long long>((most / 4 + 255) / 256 + 1, (unsigned long long)4 * eng->num_cus);
        hipLaunchKernelGGL(k_multi_copy, dim3(gx, (unsigned)nseg), dim3(256), 0, st, mc);
      }
      hipLaunchKernelGGL(k_f64_to_f32, g1(3 * (int64_t)N), dim3(256), 0, st, d_frac, b->frac, 3 * N);
      hipLaunchKernelGGL(k_f64_to_f32, g1(9 * (int64_t)B), dim3(256), 0, st, d_lat, b->lattice, 9 * B);
    }
    if (s == CHG_OK && A > 0) hipLaunchKernelGGL(k_angle_compact, g1(A), dim3(256), 0, st, a_b1, a_b2, b->u_bnode, A, b->a_b1c, b->a_b2c);
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

// ---- range diagnostic (only after a prediction returned non-finite energies) --------------------------------------------
// max |finite value| and the number of non-finite entries of a buffer: out[0] = max as float bits (non-negative floats order like
// unsigned integers), out[1] = count
__global__ void k_range_scan(const float* __restrict__ x, size_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  unsigned bad = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (v == v && fabsf(v) < 3.0e38f) m = fmaxf(m, fabsf(v)); else ++bad;
  }
  atomicMax(out, __float_as_uint(m));
  if (bad) atomicAdd(out + 1, bad);
}

int diagnose_non_finite(chg_engine* eng, chg_batch* b) {
  unsigned* d_out = nullptr;
  HIP_TRY(eng, hipMalloc(&d_out, 4 * sizeof(unsigned)));
  auto scan = [&](const float* p, size_t n, float& mx, unsigned& bad) -> int {
    unsigned h[2] = {0, 0};
    mx = 0.f; bad = 0;
    if (!p || n == 0) return CHG_OK;
    HIP_TRY(eng, hipMemsetAsync(d_out, 0, 2 * sizeof(unsigned), eng->stream));
    hipLaunchKernelGGL(k_range_scan, dim3(1024), dim3(256), 0, eng->stream, p, n, d_out);
    HIP_TRY(eng, hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, eng->stream));
    HIP_TRY(eng, hipStreamSynchronize(eng->stream));
    std::memcpy(&mx, &h[0], sizeof(float));
    bad = h[1];
    return CHG_OK;
  };
  int status = CHG_OK;
  float mx = 0.f;
  unsigned bad = 0;
  // 1. the geometry: a zero-length or non-finite bond vector makes the bases NaN in the reference as well (basis.py: sin(w r) / r)
  status = scan(reinterpret_cast<const float*>(b->ev), (size_t)4 * b->Ed, mx, bad);
  bool geometry_nan = bad > 0;
  if (status == CHG_OK && !geometry_nan && b->Ed > 0) {   // any r == 0 ?  (ev = (v, r): scan the embedding rows built from 1 / r instead)
    status = scan(b->hb0, (size_t)b->Eu * D, mx, bad);
    geometry_nan = bad > 0;
  }
  // 2. the operands of the split contractions: feature rows and first-layer tables of every layer
  std::string where;
  float worst = 0.f;
  if (status == CHG_OK && !geometry_nan) {
    auto look = [&](const char* name, int l, const float* p, size_t n) {
      if (status != CHG_OK || !p) return;
      float m1; unsigned b1;
      status = scan(p, n, m1, b1);
      if (m1 > worst) { worst = m1; where = std::string(name) + "[" + std::to_string(l) + "]"; }
    };
    const size_t N = b->N, Eu = b->Eu, Eb = b->Eb, A = b->A;
    for (int l = 0; l <= b->L; ++l) look("atom features", l, b->atom[l], N * D);
    for (int l = 0; l < b->L; ++l) look("bond features", l, b->hbc[l], Eb * D);
    for (int l = 0; l < b->L - 1; ++l) look("angle features", l, b->ang[l], A * D);
    for (int l = 0; l < b->L; ++l) { look("AtomConv atom table", l, b->Pl[l], N * 4 * D); look("AtomConv bond table", l, b->Ql[l], Eu * 2 * D); }
    for (int t = 0; t < 2 * b->L; ++t) { look("angle-layer bond table", t, b->Rl[t], Eb * 4 * D); look("angle-layer atom table", t, b->Sl[t], N * 2 * D); }
  }
  hipFree(d_out);
  if (status != CHG_OK) return status;
  if (!geometry_nan && worst >= 0.25f * F16_OPERAND_LIMIT) {
    eng->err = "non-finite results: " + where + " reaches |x| = " + std::to_string(worst) + ", at or beyond the f16 operand range of the "
               "split-precision contractions (65504; csrc/mfma_split.h) -- the reference's fp32 path does not overflow here.  Weights this far "
               "from any trained checkpoint are outside the engine's domain";
    return CHG_ERANGE;
  }
  return CHG_OK;   // coincident atoms / non-finite inputs: NaN like the reference
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

// =====================================================================================================
// C-ABI
// =====================================================================================================
extern "C" {

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
  eng->device = device;
  eng->desc = *desc;
  if (const char* g = std::getenv("CHGNET_HIP_GRAPHS")) eng->use_graphs = std::string(g) != "0";
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
  // kernels that need more than the default 64 KiB of dynamic LDS
  int s;
  if ((s = set_lds(eng, k_rows_gemm<64, 64>, rows_gemm_lds<64, 64>()))) return s;
  if ((s = set_lds(eng, k_rows_gemm<64, 128>, rows_gemm_lds<64, 128>()))) return s;
  if ((s = set_lds(eng, k_rows_gemm<128, 64>, rows_gemm_lds<128, 64>()))) return s;
  if ((s = set_lds(eng, k_rows_gemm<64, SMALL_GEMM_COLS, 1>, (rows_gemm_lds<64, SMALL_GEMM_COLS, 1>())))) return s;
  if ((s = set_lds(eng, k_rows_gemm_pair<64, SMALL_GEMM_COLS>, (rows_gemm_lds<64, SMALL_GEMM_COLS, 1>())))) return s;
  if ((s = set_lds(eng, k_rows_gemm<128, SMALL_GEMM_COLS, 1>, (rows_gemm_lds<128, SMALL_GEMM_COLS, 1>())))) return s;
  if ((s = set_lds(eng, k_rows_gemm<128, SMALL_GEMM_COLS, 2>, (rows_gemm_lds<128, SMALL_GEMM_COLS, 2>())))) return s;
  if ((s = set_lds(eng, (k_rows_gemm<64, 128, 2>), (rows_gemm_lds<64, 128, 2>())))) return s;
  if ((s = set_lds(eng, (k_rows_gemm<128, 64, 2>), (rows_gemm_lds<128, 64, 2>())))) return s;
  if ((s = set_lds(eng, k_atomconv_fwd<FWD_WAVES>, (atomconv_lds<FWD_WAVES, false, true>())))) return s;
  if ((s = set_lds(eng, k_atomconv_bwd<false>, (atomconv_lds<WAVES, true>())))) return s;
  if ((s = set_lds(eng, k_atomconv_bwd<true>, (atomconv_lds<WAVES, true>())))) return s;
  if ((s = set_lds(eng, k2_atom<false>, t2_atom_lds()))) return s;
  if ((s = set_lds(eng, k2_atom<true>, t2_atom_lds()))) return s;
  if ((s = set_lds(eng, k2_angle<true, false>, t2_angle_lds<true>()))) return s;
  if ((s = set_lds(eng, k2_angle<true, true>, t2_angle_lds<true>()))) return s;
  if ((s = set_lds(eng, k2_angle<false, false>, t2_angle_lds<false>()))) return s;
  if ((s = set_lds(eng, k2_angle<false, true>, t2_angle_lds<false>()))) return s;
  if ((s = set_lds(eng, (k_angle<true, true, WAVES, true>), (angle_lds<true, WAVES, true>())))) return s;
  if ((s = set_lds(eng, (k_angle<false, true, WAVES, true>), (angle_lds<false, WAVES, true>())))) return s;
  if ((s = set_lds(eng, k_readout<true>, readout_lds()))) return s;
  if ((s = set_lds(eng, (k_bond_embed_t<true, true>), bond_embed_lds()))) return s;
  if ((s = set_lds(eng, (k_angle_embed_t<true, true>), angle_embed_lds()))) return s;
  if ((s = set_lds(eng, (k_xty<8, 4>), (xty_lds<8, 4>())))) return s;
  if ((s = set_lds(eng, (k_xty3<8, 4>), (xty3_lds<8, 4>())))) return s;
  if ((s = set_lds(eng, (k_xty3<4, 4>), (xty3_lds<4, 4>())))) return s;
  if ((s = set_lds(eng, (k_xty3<4, 2>), (xty3_lds<4, 2>())))) return s;
  if ((s = set_lds(eng, (k_xty3<8, 8, true>), (xty3_lds<8, 8, true>())))) return s;
  if ((s = set_lds(eng, (k_xty<4, 4>), (xty_lds<4, 4>())))) return s;
  if ((s = set_lds(eng, (k_xty<4, 2>), (xty_lds<4, 2>())))) return s;
  if ((s = set_lds(eng, k2_scatter_z, scatter_z_lds()))) return s;
  if ((s = set_lds(eng, k_angle_bwd_w<true>, angle_w_lds<true>()))) return s;
  if ((s = set_lds(eng, k_angle_bwd_w<false>, angle_w_lds<false>()))) return s;
  if ((s = set_lds(eng, k_angle<true, false>, angle_lds<true>()))) return s;
  if ((s = set_lds(eng, k_angle<true, true>, (angle_lds<true, WAVES, true>())))) return s;
  if ((s = set_lds(eng, k_angle<false, true>, (angle_lds<false, WAVES, true>())))) return s;
  if ((s = set_lds(eng, k_angle<false, false, FWD_WAVES>, (angle_lds<false, FWD_WAVES>())))) return s;
  if ((s = set_lds(eng, k_readout<false>, readout_lds()))) return s;
  if ((s = set_lds(eng, k_bond_embed_t<false>, bond_embed_lds()))) return s;
  if ((s = set_lds(eng, k_bond_embed_t<true>, bond_embed_lds()))) return s;
  if ((s = set_lds(eng, k_angle_embed_t<false>, angle_embed_lds()))) return s;
  if ((s = set_lds(eng, k_angle_embed_t<true>, angle_embed_lds()))) return s;
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
  if (eng->d_weights) hipFree(eng->d_weights);
  if (eng->d_images) hipFree(eng->d_images);
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
  size_t total = 0;
  carve(b, nullptr, total);
  {
    const int sa = acquire_arena(eng, b, total);
    if (sa != CHG_OK) { delete b; return sa; }
  }
  carve(b, b->arena, total);
  register_names(b);
  if (h->atom_off) b->h_atom_off.assign(h->atom_off, h->atom_off + h->n_struct + 1);
  b->h_volume.resize(h->n_struct);
  for (int q = 0; q < h->n_struct; ++q) {
    const float* Lf = h->lattice + 9 * q;
    b->h_volume[q] = Lf[0] * (Lf[4] * Lf[8] - Lf[5] * Lf[7]) + Lf[1] * (Lf[5] * Lf[6] - Lf[3] * Lf[8]) + Lf[2] * (Lf[3] * Lf[7] - Lf[4] * Lf[6]);
  }
  int s = CHG_OK;
  const size_t B = b->B, N = b->N, Ed = b->Ed, Eu = b->Eu, A = b->A, Eb = b->Eb;
#define UP(field, n) if (s == CHG_OK) s = h2d(eng, b->field, h->field, (n))
  UP(z, N); UP(atom_owner, N); UP(atom_off, B + 1); UP(frac, 3 * N); UP(lattice, 9 * B);
  UP(e_center, Ed); UP(e_nbr, Ed); UP(e_d2u, Ed); UP(e_owner, Ed); UP(e_image, 3 * Ed); UP(e_rev, Ed); UP(p_center, Ed); UP(p_nbr, Ed);
  UP(u_u2d, Eu); UP(u_bnode, Eu); UP(bn_und, Eb);
  UP(a_ctr, A); UP(a_b1c, A); UP(a_b2c, A); UP(a_d1, A); UP(a_d2, A);
#undef UP
  if (s == CHG_OK) s = prepare_windows(eng, b);
  if (s == CHG_OK && hipStreamSynchronize(eng->stream) != hipSuccess) { eng->err = "chg_batch_upload: sync failed"; s = CHG_EHIP; }
  if (s != CHG_OK) { hipFree(b->arena); delete b; return s; }
  *out = b;
  return CHG_OK;
}

int chg_batch_build(chg_engine* eng, const chg_structs_host* h, double r_atom, double r_bond, double numerical_tol, chg_batch** out,
                    int32_t* counts_out) {
  if (!eng || !h || !out || !h->z || !h->frac || !h->lattice || !h->atom_off || h->n_struct <= 0 || h->n_atoms <= 0 || !(r_atom > 0))
    return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  return build_batch_on_device(eng, h, r_atom, r_bond, numerical_tol, out, counts_out);
}

int chg_debug_fetch_i32(chg_engine* eng, chg_batch* b, const char* name, int32_t* dst, int64_t capacity, int64_t* n_written) {
  if (!eng || !b || !name || !dst) return CHG_EINVAL;
  auto it = b->named_i32.find(name);
  if (it == b->named_i32.end()) { eng->err = std::string("chg_debug_fetch_i32: unknown buffer ") + name; return CHG_EINVAL; }
  const size_t n = std::min<size_t>(it->second.second, (size_t)std::max<int64_t>(capacity, 0));
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
    if (eng && eng->arena_pool.size() < 2) eng->arena_pool.emplace_back(b->arena, b->arena_bytes);
    else hipFree(b->arena);
  }
  delete b;
  return CHG_OK;
}

int64_t chg_batch_device_bytes(const chg_batch* b) { return b ? (int64_t)b->arena_bytes : 0; }

int chg_predict(chg_engine* eng, chg_batch* b, uint32_t task_mask) {
  if (!eng || !b) return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  const uint32_t task = task_mask | CHG_TASK_E;
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

// the local part of chg_backward: everything up to the gradient blob in HBM (b->t_grad), nothing leaves the device
static int backward_compute(chg_engine* eng, chg_batch* b, const float* energy_cotangent, const float* magmom_cotangent,
                            const float* force_cotangent, const float* stress_cotangent) {
  if (b->last_task == 0) { eng->err = "chg_backward: run chg_predict on this batch first (the reverse sweep reuses its activations)"; return CHG_EINVAL; }
  if ((int)b->h_atom_off.size() != b->B + 1) { eng->err = "chg_backward: batch has no host atom offsets"; return CHG_EINVAL; }
  TRY(ensure_train_buffers(eng, b));
  // cotangent of the per-structure energy SUMS: e_b = E_b / n_b for an intensive model (model.py:538-540); AtomRef is frozen
  std::vector<float> cot(b->B);
  double g_b3 = 0.0;
  for (int i = 0; i < b->B; ++i) {
    const double n = b->h_atom_off[i + 1] - b->h_atom_off[i];
    const double ce = energy_cotangent ? energy_cotangent[i] : 1.0;
    cot[i] = (float)(eng->desc.is_intensive ? ce / n : ce);
    g_b3 += (double)cot[i] * n;
  }
  b->h_g_b3 = (float)g_b3;   // gradient of the readout's last bias: sum_b cot_b n_b, known on the host
  HIP_TRY(eng, hipMemcpyAsync(b->t_cot, cot.data(), sizeof(float) * b->B, hipMemcpyHostToDevice, eng->stream));
  b->t_has_mcot = magmom_cotangent != nullptr;
  if (magmom_cotangent) HIP_TRY(eng, hipMemcpyAsync(b->t_mcot, magmom_cotangent, sizeof(float) * b->N, hipMemcpyHostToDevice, eng->stream));
  const bool second_order = force_cotangent || stress_cotangent;
  std::vector<float> ux, wst;
  if (second_order) {
    // direction of the one tangent sweep: ux = -dL/dF,  W_b = (160.21766208 / V_b) dL/d sigma_b   (kernels_train2.h)
    // the sweep reuses the first-order adjoints (seed 1) that the force / stress sweep of chg_predict leaves in the batch: run it
    // if the last prediction was energy-only or a first-order chg_backward has overwritten them since
    if (!b->seed1_adjoints) TRY(run_predict(eng, b, b->last_task | CHG_TASK_F));
    TRY(ensure_train2_buffers(eng, b));
    ux.assign((size_t)3 * b->N, 0.f);
    wst.assign((size_t)9 * b->B, 0.f);
    if (force_cotangent) for (size_t q = 0; q < ux.size(); ++q) ux[q] = -force_cotangent[q];
    if (stress_cotangent)
      for (int q = 0; q < b->B; ++q)
        for (int k = 0; k < 9; ++k) wst[9 * q + k] = (float)(EV_A3_TO_GPA / b->h_volume[q]) * stress_cotangent[9 * q + k];
    HIP_TRY(eng, hipMemcpyAsync(b->t2->ux, ux.data(), sizeof(float) * ux.size(), hipMemcpyHostToDevice, eng->stream));
    HIP_TRY(eng, hipMemcpyAsync(b->t2->Wst, wst.data(), sizeof(float) * wst.size(), hipMemcpyHostToDevice, eng->stream));
  }
  HIP_TRY(eng, hipStreamSynchronize(eng->stream));   // cot is a stack-lifetime host buffer
  // the forward kernels contract the bond partials in place (and store them only for force / stress tasks); the training sweeps gather them as tables
  if (b->Ed > 0) for (int l = 0; l < b->L; ++l) TRY(atomconv_q_table(eng, b, l));
  TRY(second_order ? run_backward2(eng, b) : run_backward(eng, b));
  // the b3 slot joins the blob ON THE DEVICE, so that a following all-reduce sums it over the ranks like every other entry (it used
  // to be written into the host copy after the collective: every rank then applied its LOCAL value / world -- ADVICE r03)
  HIP_TRY(eng, hipMemcpyAsync(b->t_grad + (eng->w.mlp_b3 - eng->d_weights), &b->h_g_b3, sizeof(float), hipMemcpyHostToDevice, eng->stream));
  return CHG_OK;
}

static int backward_impl(chg_engine* eng, chg_batch* b, const float* energy_cotangent, const float* magmom_cotangent,
                         const float* force_cotangent, const float* stress_cotangent, chg_comm* comm, float* grad_blob) {
  if (!eng || !b || !grad_blob) return CHG_EINVAL;
  HIP_TRY(eng, hipSetDevice(eng->device));
  const int64_t n_w = (int64_t)eng->desc.n_weights;
  if (comm) {   // the collective is enqueued on the engine's stream: the communicator must live on the engine's device
    int32_t comm_dev = -1;
    if (chg_comm_info(comm, nullptr, nullptr, &comm_dev) != CHG_OK || comm_dev != eng->device) {
      eng->err = "chg_backward_allreduce: the communicator belongs to device " + std::to_string(comm_dev) + ", the engine to device " + std::to_string(eng->device);
      return CHG_EINVAL;
    }
  }
  const int status = backward_compute(eng, b, energy_cotangent, magmom_cotangent, force_cotangent, stress_cotangent);
  if (!comm) {
    if (status != CHG_OK) return status;
    HIP_TRY(eng, hipMemcpyAsync(grad_blob, b->t_grad, sizeof(float) * (size_t)n_w, hipMemcpyDeviceToHost, eng->stream));
    return chg_synchronize(eng);
  }
  // Data-parallel step: sum of the blob over the ranks, in HBM, on this stream.  A rank whose local sweep failed (an arena that did
  // not fit, a bad argument) STILL enters the collective -- with zeros -- and reports its error afterwards: returning early would
  // leave the other ranks blocked in ncclAllReduce for ever.
  float* send = b->t_grad;
  const std::string local_err = eng->err;
  if (status != CHG_OK) {
    if (chg_comm_reserve(comm, n_w, &send) != CHG_OK) { eng->err = local_err + " (and no staging for the collective: " + chg_comm_last_error(comm) + ")"; return status; }
    if (hipMemsetAsync(send, 0, sizeof(float) * (size_t)n_w, eng->stream) != hipSuccess) return status;
  }
  if (chg_comm_all_reduce_sum_f32_device(comm, send, n_w, eng->stream) != CHG_OK) {
    eng->err = std::string("chg_backward_allreduce: ") + chg_comm_last_error(comm);
    return status != CHG_OK ? status : CHG_EHIP;
  }
  if (status != CHG_OK) {
    hipStreamSynchronize(eng->stream);
    eng->err = local_err;
    return status;
  }
  HIP_TRY(eng, hipMemcpyAsync(grad_blob, b->t_grad, sizeof(float) * (size_t)n_w, hipMemcpyDeviceToHost, eng->stream));
  return chg_synchronize(eng);
}

int chg_backward(chg_engine* eng, chg_batch* b, const float* energy_cotangent, const float* magmom_cotangent,
                 const float* force_cotangent, const float* stress_cotangent, float* grad_blob) {
  return backward_impl(eng, b, energy_cotangent, magmom_cotangent, force_cotangent, stress_cotangent, nullptr, grad_blob);
}

int chg_backward_allreduce(chg_engine* eng, chg_batch* b, const float* energy_cotangent, const float* magmom_cotangent,
                           const float* force_cotangent, const float* stress_cotangent, chg_comm* comm, float* grad_blob) {
  return backward_impl(eng, b, energy_cotangent, magmom_cotangent, force_cotangent, stress_cotangent, comm, grad_blob);
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
  int s = d2h(eng, o->energy, b->energy, B);
  if (s == CHG_OK && (b->last_task & CHG_TASK_F)) s = d2h(eng, o->force, b->force, 3 * N);
  if (s == CHG_OK && (b->last_task & CHG_TASK_S)) s = d2h(eng, o->stress, b->virial, 9 * B);
  if (s == CHG_OK && (b->last_task & CHG_TASK_M)) s = d2h(eng, o->magmom, b->magmom, N);
  if (s == CHG_OK) s = d2h(eng, o->site_energy, b->site_energy, N);
  if (s == CHG_OK) s = d2h(eng, o->atom_fea, b->atom[b->L - 1], N * D);
  if (s == CHG_OK) s = d2h(eng, o->crystal_fea, b->crystal_fea, B * D);
  if (s != CHG_OK) return s;
  TRY(chg_synchronize(eng));
  // Non-finite energies: the reference gives them too for coincident atoms (1 / r of a zero-length bond) -- passed through.  But an
  // activation past the f16 operand range of the split contractions (mfma_split.h) ALSO ends as NaN here where the reference's fp32
  // path stays finite: that case is an error, not a result.  An overflow cannot stay silent (an inf operand makes the accumulator
  // inf / NaN, LayerNorm spreads it over the row, the sums carry it to the energy), so the check costs nothing until it triggers.
  bool finite = true;
  if (o->energy) for (size_t i = 0; i < B && finite; ++i) finite = std::isfinite(o->energy[i]);
  if (!finite) TRY(diagnose_non_finite(eng, b));
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
  for (int variant = 0; variant < 6 && s == CHG_OK; ++variant) {
    const int unroll = variant < 3 ? 1 : 4;
    const unsigned blocks = (unsigned)((variant % 3 == 0 ? 8 : (variant % 3 == 1 ? 32 : 128)) * eng->num_cus);
    auto launch = [&]() {
      if (unroll == 1) hipLaunchKernelGGL(k_stream_copy<1>, dim3(blocks), dim3(256), 0, eng->stream, (const f32x4*)a, (f32x4*)b, n);
      else hipLaunchKernelGGL(k_stream_copy<4>, dim3(blocks), dim3(256), 0, eng->stream, (const f32x4*)a, (f32x4*)b, n);
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
