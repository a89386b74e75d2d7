// engine_internal.h -- host-side types and helpers shared by the translation units of libchgnet_hip.so (not an API):
//   engine.hip          C-ABI: engine / batch lifetime, upload / download, timers, profiling, debug fetch, self-tests
//   engine_predict.hip  launch schedule of chg_predict (forward + force / stress sweep), arena layout, per-atom schedule
//   engine_train.hip    fine-tuning backward: first- and second-order sweeps (chg_backward*)
//   engine_graph.hip    device-side graph construction (chg_batch_build)
// A kernel edit recompiles the unit that launches it, not 150 KB of host code; every kernel instantiation is launched from ONE unit.
#pragma once

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
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "mfma_tile.h"
#include "mfma_split.h"
#include "kernels_conv.h"     // argument structs (GatedW, AtomConvArgs, AngleArgs, RowsGemm) and the tile constants
#include "kernels_angle_w.h"  // WinIndex
#include "kernels_angle_blk.h"
#include "kernels_embed.h"    // BondEmbedTArgs / AngleEmbedTArgs

using namespace chg;

constexpr int MAX_CONV = 8;
constexpr float F16_OPERAND_LIMIT = 65504.0f;   // largest finite f16: forward operands of the split contractions are not rescaled
#ifndef CHG_FWD_WAVES
#define CHG_FWD_WAVES 8
#endif
#ifdef CHG_PHASE_TIMING
constexpr size_t PHASE_FLOATS = (size_t)8 * 2 * 10 * PH_WAVES;   // kernels_conv.h PH_FLUSH
#else
constexpr size_t PHASE_FLOATS = 64;
#endif
constexpr int FWD_WAVES = CHG_FWD_WAVES;   // waves per workgroup of the light forward kernels (12 = 3 per SIMD measured no better: profiles notes)

struct ACW { const float *w_cn, *w_bond, *b1, *q_bias, *q_shift; GatedW g; const float *w2c_t, *w2g_t, *w_out, *b_out, *w_out_t, *w_cn_t, *w_bond_t; };
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

// Text of the last failure, one slot PER CALLING THREAD: chg_batch_upload may run on a loader thread while the engine's own thread
// computes (INTEGRATION.md "Threads"), and both may fail.  Every thread reads -- chg_last_error -- the text of the status IT
// received; no std::string is shared between threads.
struct ErrText {
  static std::unordered_map<const ErrText*, std::string>& texts() {
    thread_local std::unordered_map<const ErrText*, std::string> t;
    return t;
  }
  std::string& slot() const { return texts()[this]; }
  ErrText& operator=(const std::string& v) { slot() = v; return *this; }
  ErrText& operator=(const char* v) { slot() = v; return *this; }
  operator std::string() const { return slot(); }
  const char* c_str() const { return slot().c_str(); }
  // The destroying thread's entry is erased.  Entries OTHER threads made for this address stay in their maps: chg_engine_create
  // clears the creating thread's slot, and any other thread reads its slot only after a call of its own on the engine has failed and
  // written it -- so a recycled address never shows an earlier engine's text.
  ~ErrText() { texts().erase(this); }
};
inline std::string operator+(const char* a, const ErrText& b) { return std::string(a) + std::string(b); }
inline std::string operator+(const std::string& a, const ErrText& b) { return a + std::string(b); }
inline std::string operator+(const ErrText& a, const char* b) { return std::string(a) + b; }
inline std::string operator+(const ErrText& a, const std::string& b) { return std::string(a) + b; }

struct ProfEntry { std::string label; int64_t launches = 0; double ms = 0.0; };
struct PendingEvent { int entry; hipEvent_t start, stop; };

struct chg_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  chg_model_desc desc{};
  float* d_weights = nullptr;
  Weights w{};
  // prebuilt LDS weight blocks of the inference tile kernels (kernels_conv.h k_*_image), rebuilt by every weight upload
  float* d_images = nullptr;
  const float* p_elem = nullptr;               // [94][256] P table of the first AtomConv per element (k_prologue; rebuilt with the images)
  const float* img_ac_fwd[2][MAX_CONV] = {};   // [without / with q_bias][layer]
  const float* img_ac_bwd[MAX_CONV] = {};
  const float* img_ac_bwd_rm[MAX_CONV] = {};   // row-major block of the fused adjoint (k_atomconv_image_rm)
  const float* img_angle[2][2 * MAX_CONV] = {};   // [fwd / bwd][slot: BondConv l | L + AngleUpdate l]
  ErrText err;                // last failure text of the calling thread
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool profiling = false;
  std::vector<ProfEntry> prof;
  std::map<std::string, int> prof_index;
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
  // chg_batch_upload may run on a second host thread while this engine computes (a data loader uploading the next batch under the
  // current step's sweeps): its copies go through copy_stream, BOTH pools below (arena_pool, work_pool / work_kind) are only touched
  // under pool_mu, the failure text is per thread (ErrText), and everything it would launch on the
  // compute stream (the per-atom index, prepare_windows) waits for the batch's first use (chg_batch::win_pending)
  hipStream_t copy_stream = nullptr;
  std::mutex pool_mu;
  std::vector<std::pair<char*, size_t>> arena_pool;   // released batch arenas, reused by later uploads
  std::vector<std::pair<char*, size_t>> work_pool;    // released training workspaces (tens of GB: a hipMalloc per step would dominate it)
  std::vector<int> work_kind;                         // 0: first-order workspace, 1: second-order workspace
  bool use_graphs = true;   // CHGNET_HIP_GRAPHS=0 forces eager launches
  bool force_wide = false;  // CHGNET_WIDE_RANGE=1: every batch runs on the wide-range sweeps from its first prediction (tests: the goldens through them)
  char* scratch = nullptr;  // grow-only scratch of chg_batch_build (MD rebuilds the graph every step)
  size_t scratch_bytes = 0, scratch_wanted = 0;
  char* h_stage = nullptr;   // pinned staging for the inputs of chg_batch_build
  size_t h_stage_bytes = 0;
  char* h_out = nullptr;     // pinned staging for the outputs of chg_batch_download (small batches: one queue of async copies, one sync)
  size_t h_out_bytes = 0;
  int num_cus = 256;
  // single-pass graph builds (chg_batch_build): counts of the previous build size the next one's scratch speculatively
  bool spec_builds = true;    // CHGNET_SPEC_BUILD=0 forces the exact three-round-trip pass
  int last_N = 0, last_Ed = 0, last_A = 0, last_Eb = 0;
  double last_r_atom = 0.0, last_r_bond = 0.0;
  long n_spec_builds = 0, n_spec_overflows = 0, n_cell_builds = 0, n_cell_fallbacks = 0;
  int graph_search = 0;       // chg_engine_set_graph_search: 0 by size, 1 all pairs, 2 cell list
  int cell_min_atoms = 2048;  // structures at least this large are binned (by size).  Round 6, same box: all pairs beats the cell list at 512 (build 210 vs 273 us)
                              // and 1,024 atoms (245 vs 311 us): one wave per centre walks the structure's atoms 64 at a time -- 8-16 iterations --
                              // while the cell path pays a bin walk and an in-LDS sort per centre
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
  float* zsave_l[2 * MAX_CONV] = {};   // same slots: [A,128] first-layer pre-activations kept by the forward angle kernels for their adjoints
                                       // (large batches only: AngleArgs::zsave), else null
  // outputs
  float *energy, *site_energy, *site_raw, *magmom, *crystal_fea, *force, *virial, *volume;
  // reverse sweep
  float *Ga, *GA, *Gb, *Gwag, *Gwbgc, *Gang, *GQ, *Gagg, *Grk, *Gu;
  float* phase = nullptr;   // CHG_PHASE_TIMING builds: per-phase shader-clock totals of the angle kernels
  WinIndex win{};           // centre-major row order + window slots of the angle adjoints (kernels_angle_w.h), built by prepare_windows
  int *win_tmp = nullptr, *win_scan = nullptr;
  int win_grid = 64;        // workgroups of the per-atom kernels (a multiple of 64: the atom schedule is built for it, k_win_schedule)
  bool win_built = false;   // the index exists (batches too small to give every wave a few atoms never build it)
  bool canonical = false;   // built by chg_batch_build and known to have the canonical angle structure (uploaded graphs: unknown -> false):
                            // the per-atom / team adjoints then need no row-order launch behind them
  bool win_index_ready = false;   // chg_batch_build emitted the centre-major index with the graph (prepare_windows only schedules)
  int win_team = 0;         // > 0: small batch in TEAM mode (kernels_angle_w.h) -- the index without the schedule exists and the angle
                            // adjoints give every atom to a team of this many waves; win_grid is their workgroup count
  // MD-size batches built on the device: the angle adjoints over blocked tiles (kernels_angle_blk.h); blk_cap = capacity of the
  // index in tiles (0: none), the tile count itself is a device quantity (blk_tiles)
  int blk_cap = 0;
  bool blk_ready = false;   // the index is in place (device-built: written by the graph builder; uploaded: prepare_windows launches it)
  int *blk_a = nullptr, *blk_b1c = nullptr, *blk_b2c = nullptr, *blk_ctr = nullptr, *blk_desc = nullptr, *blk_tiles = nullptr;
  bool win_pending = false; // uploaded, prepare_windows not launched yet (ensure_windows: first predict / debug fetch)
  bool zsave_now = false;   // this prediction has a reverse sweep: its forward angle kernels keep z (zsave_l)
  int p_table_done = -1;    // forward sweep, small batches: the AtomConv layer whose P table an angle layer's launch has contracted already
  float *zero1, *zero1_end, *zero2, *zero2_end;   // contiguous ranges cleared by one memset each
  float* zero2_keep_end = nullptr;                // group 2 holds [energy, magmom) up to here: results of the prediction a later chg_backward keeps
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
  bool q_tables = false;       // Ql[l] hold the bond partials of the LAST prediction (its forward kernels store them when a reverse sweep follows)
  bool wide_range = false;     // a prediction of this batch left the f16 operand range: it runs through chgh_wide::run_predict from then on
  bool t_has_mcot = false;
  float *t_grad = nullptr, *t_cot = nullptr, *t_tmp = nullptr /* 256 floats of scratch */, *t_dumpG = nullptr, *t_dumpH = nullptr, *t_dumpZ = nullptr, *t_Xb = nullptr, *t_Xa = nullptr,
        *t_ro = nullptr;
};

#define HIP_TRY(eng, expr)                                                                          \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      (eng)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                               \
      return CHG_EHIP;                                                                              \
    }                                                                                               \
  } while (0)

#define TRY(x)                   \
  do {                           \
    int _s = (x);                \
    if (_s != CHG_OK) return _s; \
  } while (0)

namespace chgh {   // host-side internals (one namespace: the library exports only the chg_* C symbols by name)

int prof_entry(chg_engine* eng, const char* label);
hipEvent_t get_event(chg_engine* eng);
int collect_profile(chg_engine* eng);

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

inline dim3 g1(int64_t n, int b = 256) { return dim3((unsigned)std::max<int64_t>(1, (n + b - 1) / b)); }
inline int wave_grid(chg_engine* eng, int64_t items) {   // one wave per item, 4 waves per block, grid-stride
  return (int)std::max<int64_t>(1, std::min<int64_t>((items + 3) / 4, 16 * (int64_t)eng->num_cus));
}

template <class T>
int h2d(chg_engine* eng, T* dst, const T* src, size_t n, hipStream_t stream = nullptr) {
  if (n == 0) return CHG_OK;
  if (!src) { eng->err = "chg_batch_upload: null host array"; return CHG_EINVAL; }
  HIP_TRY(eng, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, stream ? stream : eng->stream));
  return CHG_OK;
}
template <class T>
int d2h(chg_engine* eng, T* dst, const T* src, size_t n) {
  if (n == 0 || !dst) return CHG_OK;
  HIP_TRY(eng, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, eng->stream));
  return CHG_OK;
}

template <class K>
int set_lds(chg_engine* eng, K kernel, size_t bytes) {
  HIP_TRY(eng, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return CHG_OK;
}

// ---- engine_predict.hip
size_t layout_weights(const float* base, int L, Weights& w);
int grid_for(int rows, int max_blocks, int block_rows = BLOCK_ROWS);
int tile_grid(chg_engine* eng, int rows, int block_rows = BLOCK_ROWS);
int rows_gemm(chg_engine* eng, const char* label, int K, int NOUT, const float* X, int ldx, const int* in_idx, const float* Wt,
              const float* bias, const float* resid, int ldr, float* Y, int ldy, const int* out_idx, int rows, int accumulate);
int rows_gemm_out2(chg_engine* eng, const char* label, const float* X, const int* in_idx, const float* Wt, const float* Wt2,
                   const float* bias, float* Y, int ldy, int rows);
int rows_gemm_in2(chg_engine* eng, const char* label, const float* X, int ldx, const float* Wt, const float* Wt2, float* Y,
                  const int* out_idx, int rows, int accumulate);
int zero(chg_engine* eng, void* p, size_t bytes);
int build_images(chg_engine* eng);
int predict_set_lds(chg_engine* eng);      // dynamic-LDS attributes of the kernels this unit launches
int atomconv_q_table(chg_engine* eng, chg_batch* b, int l);
AtomConvArgs atomconv_args(chg_engine* eng, chg_batch* b, int l);
AngleArgs angle_args(chg_batch* b, int slot, const float* ang, const float* w_ang, const GatedW& g, float* out);
int angle_table_grads(chg_engine* eng, chg_batch* b, int slot, const float* w_bij_t, const float* w_ctr_t);
BondEmbedTArgs bond_embed_args(chg_engine* eng, chg_batch* b);
AngleEmbedTArgs angle_embed_args(chg_engine* eng, chg_batch* b);
int run_predict(chg_engine* eng, chg_batch* b, uint32_t task);
void carve(chg_batch* b, char* base, size_t& total);
int prepare_windows(chg_engine* eng, chg_batch* b);
inline long team_min_angles() {   // TEAM-mode threshold of the angle adjoints (engine_predict.hip decide_windows; engine_graph.hip: is the index worth emitting)
  static const long v = [] { const char* e = std::getenv("CHGNET_TEAM_MIN_ANGLES"); return e ? std::atol(e) : 131072L; }();
  return v;
}
// ... and device-built batches of up to 8,191 atoms (below ~6,000 the per-atom adjoints leave waves idle) and this many angles use the
// blocked tiles (kernels_angle_blk.h) instead of either; 0: never.  Same box, thermalised Li9Co7O16 cells, BondConv / AngleUpdate
// adjoint per launch, blocked / TEAM / row order: 256 atoms 46 / -- / 58 and 27 / -- / 43 us, 512 atoms 82 / -- / 103 and 50 / -- / 74,
// 1,024 atoms 153 / 175 / 200 and ~90 / 115 / 140 (profiles/r06_experiments.md section 14).
inline long blk_max_angles() {
  static const long v = [] { const char* e = std::getenv("CHGNET_BLK_MAX_ANGLES"); return e ? std::atol(e) : (1L << 22); }();
  return v;
}
inline size_t blk_tile_bound(size_t A, size_t Eb, size_t N) { return (A + 14 * Eb + 9 * N) / 16 + 1; }
inline int upload_blk_cap(size_t N, size_t A, size_t Eb) {    // capacity of the blocked-tile index of an uploaded batch (0: none)
  return (A > 0 && N + 1 <= 8192 && (long)A <= blk_max_angles()) ? (int)blk_tile_bound(A, Eb, N) : 0;
}   // 16 ceil(n/4)^2 <= n (n - 1) + 7 n + 9, sum of n <= 2 Eb
bool decide_windows(chg_engine* eng, chg_batch* b);   // sets win_built / win_team / win_grid; true when the batch uses the per-atom index
int ensure_windows(chg_engine* eng, chg_batch* b);   // launches a pending prepare_windows (compute stream)
void register_names(chg_batch* b);

// ---- engine_train.hip
int train_set_lds(chg_engine* eng);
void free_train2(chg_batch* b);
void release_workspace(chg_engine* eng, char* p, size_t bytes, int kind);
int backward_impl(chg_engine* eng, chg_batch* b, const float* energy_cotangent, const float* magmom_cotangent, const float* force_cotangent,
                  const float* stress_cotangent, chg_comm* comm, float* grad_blob);

// ---- engine_graph.hip
size_t scan_scratch_ints(int n);           // scratch ints exclusive_scan_with needs for n elements
int exclusive_scan_with(chg_engine* eng, int* scratch, const int* in, int* out, int n);
int acquire_arena(chg_engine* eng, chg_batch* b, size_t total);
int build_batch_on_device(chg_engine* eng, const chg_structs_host* h, double r_atom, double r_bond, double numerical_tol, chg_batch** out);

}  // namespace chgh

// engine_predict_wide.hip: the prediction sweep with every operand row of the split contractions scaled (any fp32 magnitude)
// engine_train_wide.hip: the fine-tuning sweeps likewise
namespace chgh_wide {
int run_predict(chg_engine* eng, chg_batch* b, uint32_t task);
int predict_set_lds(chg_engine* eng);
int backward_compute(chg_engine* eng, chg_batch* b, const float* energy_cotangent, const float* magmom_cotangent, const float* force_cotangent,
                     const float* stress_cotangent);
int train_set_lds(chg_engine* eng);
}  // namespace chgh_wide

using namespace chgh;

