/* chgnet_hip.h -- C-ABI of the MI355X (gfx950) CHGNet engine (libchgnet_hip.so).
 *
 * The reference has no FFI for this path: its boundary is the Python class API
 *   CHGNet.predict_graph / forward      chgnet/model/model.py:593-665, 330-387
 *   BatchedGraph.from_graphs            chgnet/model/model.py:792-913
 *   CHGNet._compute                     chgnet/model/model.py:389-542
 * This header is what a ctypes/cffi binding inside that class binds instead of running the
 * torch modules.  Entry points:
 *
 *   chg_engine_create   <- CHGNet.__init__/load_state_dict (weights re-laid by pack.py)
 *   chg_batch_upload    <- [g.to(device) for g in graphs] + index offsetting of from_graphs
 *                          (model.py:640-644, 856-857, 873-877)
 *   chg_predict         <- from_graphs geometry/bases + _compute + the two autograd.grad sweeps
 *                          (model.py:826-871, 427-540, 517-535) + AtomRef (model.py:356-358,378)
 *   chg_batch_download  <- tensor.cpu().detach().numpy() per key (model.py:651-663)
 *
 * Conventions: plain C types only; every function returns 0 or a negative chg_status and
 * never throws; chg_last_error() gives the text of the last failure on that engine.  One
 * engine per GPU; calls on one engine must be serialised by the caller; engines on different
 * devices are independent (one process or thread per GPU).  Host buffers belong to the
 * caller, device memory to the library.  All floating point is fp32, all indices int32.
 */
#ifndef CHGNET_HIP_H
#define CHGNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  CHG_OK = 0,
  CHG_EINVAL = -1,     /* bad argument */
  CHG_EHIP = -2,       /* HIP runtime error (text in chg_last_error) */
  CHG_ENOMEM = -3,     /* device or host allocation failed */
  CHG_ENODEV = -4,     /* no usable gfx950 device */
  CHG_EUNSUPPORTED = -5,
  CHG_ERANGE = -6      /* a WEIGHT left the operand range of the split-precision contractions (|w| >= 65504: the tile kernels hold
                          their weights as f16 hi / lo images) -- chg_engine_create / chg_engine_update_weights.  ACTIVATIONS of any fp32
                          magnitude are computed: a batch whose product sweep overflows the f16 operands is detected at
                          chg_batch_download (non-finite results) and run again on the wide-range sweep (csrc/engine_predict_wide.hip),
                          like the reference's fp32 path (crystalgraph.py:12 TORCH_DTYPE); chg_backward follows it there
                          (csrc/engine_train_wide.hip) */
} chg_status;

/* task bits (reference task strings "e","ef","em","efs","efsm": chgnet/__init__.py:15) */
enum {
  CHG_TASK_E = 1u,
  CHG_TASK_F = 2u,
  CHG_TASK_S = 4u,
  CHG_TASK_M = 8u
};

typedef struct chg_engine chg_engine;
typedef struct chg_batch chg_batch;

typedef struct chg_model_desc {
  int32_t n_conv;              /* 4 for every released checkpoint */
  int32_t cutoff_coeff;        /* envelope exponent p (8 for 0.3.0) */
  int32_t is_intensive;
  int32_t has_composition;     /* AtomRef present */
  float atom_graph_cutoff;     /* 6 A */
  float bond_graph_cutoff;     /* 3 A */
  int64_t n_weights;           /* length of the blob in floats; layout = chgnet_amd/pack.py:weight_layout */
  int32_t n_mlp_hidden;        /* hidden layers of the energy head: 3 (0.3.0 / r2scan, mlp_hidden_dims=[64,64,64]) or 2 (0.2.0, [64,64]);
                                  0 = 3.  The blob keeps the third layer's slots either way (unused, zero, when 2) */
  int32_t mlp_out_bias;        /* 1: the mlp_out Linears of AtomConv / BondConv carry a bias (0.2.0 checkpoint, model.py:734) -- chg_backward then
                                  also forms their gradients and W_bond's term through the bonds outside the bond graph */
} chg_model_desc;

/* Packed batch of B structures in global (batch-wide) numbering: chgnet_amd/pack.py:pack_batch. */
typedef struct chg_batch_host {
  int32_t n_struct, n_atoms, n_directed, n_undirected, n_angles, n_bnodes;
  const int32_t* z;            /* [N]     atomic numbers                          */
  const float* frac;           /* [N,3]   fractional coordinates                  */
  const float* lattice;        /* [B,3,3] rows a,b,c                              */
  const int32_t* atom_owner;   /* [N]     structure index                         */
  const int32_t* atom_off;     /* [B+1]                                           */
  const int32_t* e_center;     /* [Ed]    atom_graph[:,0]                         */
  const int32_t* e_nbr;        /* [Ed]    atom_graph[:,1]                         */
  const float* e_image;        /* [Ed,3]  neighbor_image                          */
  const int32_t* e_d2u;        /* [Ed]    directed2undirected                     */
  const int32_t* e_owner;      /* [Ed]    structure index                         */
  const int32_t* e_rev;        /* [Ed]    index of the opposite directed edge     */
  const int32_t* p_center;     /* [Ed]    centre atom, bond-pair order (rows 2k,2k+1 = bond k) */
  const int32_t* p_nbr;        /* [Ed]    neighbour atom, bond-pair order         */
  const int32_t* u_u2d;        /* [Eu]    undirected2directed                     */
  const int32_t* u_bnode;      /* [Eu]    compact bond-graph node id or -1        */
  const int32_t* bn_und;       /* [Eb]    undirected index of each node           */
  const int32_t* a_ctr;        /* [A]     bond_graph[:,0]                         */
  const int32_t* a_b1c;        /* [A]     node id of bond_graph[:,1]              */
  const int32_t* a_b2c;        /* [A]     node id of bond_graph[:,3]              */
  const int32_t* a_d1;         /* [A]     bond_graph[:,2]                         */
  const int32_t* a_d2;         /* [A]     bond_graph[:,4]                         */
} chg_batch_host;

/* Host destinations for chg_batch_download; null pointers are skipped. */
typedef struct chg_out_host {
  float* energy;        /* [B]    eV/atom if is_intensive else eV (incl. AtomRef) */
  float* force;         /* [N,3]  eV/A            (needs CHG_TASK_F)              */
  float* stress;        /* [B,9]  GPa             (needs CHG_TASK_S)              */
  float* magmom;        /* [N]    mu_B            (needs CHG_TASK_M)              */
  float* site_energy;   /* [N]    eV, incl. AtomRef site shift                    */
  float* atom_fea;      /* [N,64] atom features before the last AtomConv          */
  float* crystal_fea;   /* [B,64]                                                 */
} chg_out_host;

/* Version of this interface: bumped whenever a struct of this header grows or an entry point changes meaning (chg_model_desc gained
 * n_mlp_hidden / mlp_out_bias at 2; chg_batch_build_predict arrived at 3).  A binding compiled against another value must refuse the
 * library: chg_engine_create COPIES *desc, so an older, shorter chg_model_desc would be read past its end. */
#define CHG_ABI_VERSION 3
int chg_abi_version(void);
int chg_device_count(void);
/* Length in floats of the weight blob for an n_conv-block model (same table as pack.py:weight_layout). */
int64_t chg_weights_required(int32_t n_conv);
int chg_engine_create(const chg_model_desc* desc, const float* weights_blob, int device, chg_engine** out);
int chg_engine_destroy(chg_engine* eng);
const char* chg_last_error(const chg_engine* eng);

/* Device-memory contract.  A batch lives in ONE arena (inputs + every activation / gradient buffer of the
 * forward and reverse sweeps); chg_batch_bytes_required gives its exact size from the counts alone, so a
 * caller can size chunks before uploading (the reference's only memory knob is batch_size,
 * chgnet/model/model.py:639-650).  chg_batch_upload / chg_batch_build return CHG_ENOMEM -- and leave the
 * engine usable -- when the arena cannot be allocated or exceeds the limit set here (0 = no limit); the
 * host side then splits the chunk and retries (chgnet_amd/model.py). */
int chg_engine_build_stats(chg_engine* eng, int64_t* single_pass_builds, int64_t* capacity_overflows);
int chg_engine_set_memory_limit(chg_engine* eng, int64_t bytes);
int chg_engine_memory_info(chg_engine* eng, int64_t* free_bytes, int64_t* total_bytes);
int64_t chg_batch_bytes_required(int32_t n_conv, int32_t n_struct, int32_t n_atoms, int32_t n_directed, int32_t n_angles, int32_t n_bnodes);

int chg_batch_upload(chg_engine* eng, const chg_batch_host* host, chg_batch** out);
/* Page-locked host buffers for the arrays of a chg_batch_host (optional): uploads from them run as asynchronous DMA at the link rate
 * instead of staged copies from pageable memory (the reference's counterpart is DataLoader(pin_memory=True), data/dataset.py). */
int chg_host_alloc(int64_t bytes, void** out);
int chg_host_free(void* p);

/* Structures only (no graph): the periodic neighbour list, bond numbering and bond graph are built on
 * the device, bit-for-bit the arrays of chg_graph_build (include/chgnet_graph.h) + pack.py.  Replaces
 * CrystalGraphConverter.forward (chgnet/graph/converter.py:102-190) for structures headed to the GPU. */
typedef struct chg_structs_host {
  int32_t n_struct, n_atoms;
  const int32_t* z;           /* [N]                                  */
  const double* frac;         /* [N,3]   float64 fractional coordinates (unwrapped is fine) */
  const double* lattice;      /* [B,3,3] float64, rows a,b,c          */
  const int32_t* atom_off;    /* [B+1]                                */
} chg_structs_host;
/* counts_out[6] = { n_directed, n_undirected, n_angles, n_bnodes, n_isolated_atoms, single_pass }.
 * The first build on an engine takes three blocking count round trips; later builds size their scratch from the previous
 * build's per-atom counts (+25 %), take every count from device memory and read them once at the end (single_pass = 1);
 * a capacity that proves too small is caught by a device-side flag and the build repeats on the exact path.
 * chg_engine_build_stats reports how many builds went each way. */
int chg_batch_build(chg_engine* eng, const chg_structs_host* host, double r_atom, double r_bond, double numerical_tol,
                    chg_batch** out, int32_t* counts_out);
/* chg_batch_build followed at once by chg_predict(task_mask) on the new batch, in ONE call: between the two the device of a
 * single-structure caller (an MD / relaxation step through CHGNetCalculator.calculate, reference chgnet/model/dynamics.py:129-181) sat
 * idle for the trip back into the host language (~15 us of a ~1 ms step).  Same results and errors as the two calls. */
int chg_batch_build_predict(chg_engine* eng, const chg_structs_host* host, double r_atom, double r_bond, double numerical_tol,
                            uint32_t task_mask, chg_batch** out, int32_t* counts_out);
/* Neighbour search of chg_batch_build, the device-side twin of chg_graph_build_with's `search` (chgnet_graph.h):
 * 0 = by size (structures with at least cell_min_atoms atoms -- default 2048, 0 keeps the current value -- are binned on
 * the host and searched through a cell list, one wave per centre, rows sorted in LDS), 1 = all pairs, 2 = cell list for
 * every structure.  The rows are the same bit for bit either way; a centre with more than 1024 rows makes the build
 * repeat with all pairs (counted by chg_engine_cell_stats). */
int chg_engine_set_graph_search(chg_engine* eng, int32_t search, int32_t cell_min_atoms);
int chg_engine_cell_stats(chg_engine* eng, int64_t* cell_builds, int64_t* all_pairs_fallbacks);
/* int32 index array of a batch by pack.py name (e_center, e_nbr, e_d2u, u_u2d, a_ctr, ...) -- tests only; "wide_range": one int,
 * 1 when chg_batch_download has moved the batch to the wide-range sweeps (an activation beyond the f16 operand range) */
int chg_debug_fetch_i32(chg_engine* eng, chg_batch* batch, const char* name, int32_t* dst, int64_t capacity, int64_t* n_written);
/* new positions / cells on an unchanged graph topology (finite differences, strain scans, a relaxation step that keeps its neighbours) */
int chg_batch_update_geometry(chg_engine* eng, chg_batch* batch, const float* frac, const float* lattice);
int chg_batch_free(chg_engine* eng, chg_batch* batch);
int64_t chg_batch_device_bytes(const chg_batch* batch);

/* Asynchronous on the engine's stream; results stay in HBM until downloaded. */
int chg_predict(chg_engine* eng, chg_batch* batch, uint32_t task_mask);
int chg_synchronize(chg_engine* eng);

/* Fine-tuning backward (reference: loss.backward() through CHGNet.forward, chgnet/trainer/trainer.py:399-411, model.py:427-542,
 * and through the create_graph=True forces / stress of model.py:517-535).  After chg_predict on `batch`:
 *   grad_blob[i] = d( sum_b ce[b] energy[b] + sum_i gm[i] magmom[i] + sum_i gF[i] . force[i] + sum_b gS[b] : stress[b] ) / d weights_blob[i]
 * in the layout of the weight blob (chgnet_amd/pack.py:weight_layout; derived entries -- transposed copies, q_bias -- and the
 * frozen AtomRef stay 0; pack.py:unpack_weight_grads maps it back to state_dict names).  All cotangents are host arrays in the
 * units chg_batch_download returns the quantities in: energy_cotangent [B] (null = ones), magmom_cotangent [N] or null,
 * force_cotangent [N,3] or null, stress_cotangent [B,9] or null; grad_blob: host [n_weights].  Synchronous.
 * Without force / stress terms this is a first-order reverse sweep (fused kernels).  With them it is ONE tangent sweep along
 * (ux = -gF, strain direction (160.2 / V) gS) followed by a reverse sweep with two adjoints per activation, one fused tile kernel
 * per layer and direction (kernels_train2_tile.h; CHGNET_T2_UNFUSED=1 selects the row-array pipeline of kernels_train2.h).  That
 * sweep reuses the first-order adjoints the force / stress sweep of chg_predict leaves in the batch: if the last prediction on
 * `batch` was energy-only, chg_backward runs the prediction with forces itself first.  Overwrites the batch's gradient workspace:
 * download forces / stress before calling. */
int chg_backward(chg_engine* eng, chg_batch* batch, const float* energy_cotangent, const float* magmom_cotangent,
                 const float* force_cotangent, const float* stress_cotangent, float* grad_blob);
/* The same, for a data-parallel step: the gradient blob is summed over the ranks of `comm` (ncclAllReduce on the engine's
 * stream, in HBM) before it is copied to the host -- every rank receives the summed blob.  comm = NULL: chg_backward. */
struct chg_comm;
int chg_backward_allreduce(chg_engine* eng, chg_batch* batch, const float* energy_cotangent, const float* magmom_cotangent,
                           const float* force_cotangent, const float* stress_cotangent, struct chg_comm* comm, float* grad_blob);
/* All-gather of the batch's per-structure energies (after chg_predict) from HBM on the engine's stream: every rank
 * contributes `width` floats (its n_struct energies, zero-padded), table: host [nranks * width] in rank order. */
int chg_batch_all_gather_energy(chg_engine* eng, chg_batch* batch, struct chg_comm* comm, int64_t width, float* table);
/* The engine's HIP stream (a hipStream_t) and device ordinal, for callers that enqueue their own device work behind it. */
void* chg_engine_stream(chg_engine* eng);
int chg_engine_device(chg_engine* eng);
/* New parameter values for an existing engine (optimizer step): same blob layout and length as at creation.  Also rebuilds the
 * prebuilt LDS weight images the inference kernels read (captured hipGraphs stay valid: they hold the image buffer, not its contents). */
int chg_engine_update_weights(chg_engine* eng, const float* weights_blob);
int chg_batch_download(chg_engine* eng, chg_batch* batch, const chg_out_host* out);

/* Wall time of the stream between two marks, from HIP events recorded on the engine's stream. */
int chg_timer_start(chg_engine* eng);
int chg_timer_stop_ms(chg_engine* eng, float* elapsed_ms);

/* STREAM-like device copy of `bytes` (read + write) repeated `iters` times on the engine's stream; the average
 * time of one copy.  2 * bytes / time is the measured HBM ceiling quoted next to the HBM-bound kernels. */
int chg_stream_copy(chg_engine* eng, int64_t bytes, int iters, float* ms_per_iter);

/* Per-kernel profile: when enabled every launch is bracketed by HIP events on the engine's
 * stream.  chg_profile_read returns, for entry i, the kernel label, launch count and total ms. */
int chg_profile_enable(chg_engine* eng, int on);
int chg_profile_reset(chg_engine* eng);
int chg_profile_count(chg_engine* eng);
int chg_profile_read(chg_engine* eng, int i, char* label, int label_cap, int64_t* launches, double* total_ms);

/* Copy a named intermediate device buffer of the last chg_predict to the host (tests only).
 * Returns the number of floats written in *n_written; CHG_EINVAL if the name is unknown. */
int chg_debug_fetch(chg_engine* eng, chg_batch* batch, const char* name, float* dst, int64_t capacity, int64_t* n_written);

/* Self-test of the MFMA tile primitives: Y[rows,nout] = X[rows,k] . Wt[nout,k]^T + bias (k, nout in {64,128}). */
int chg_test_rows_gemm(chg_engine* eng, const float* x, const float* wt, const float* bias, float* y, int rows, int k, int nout);
/* Self-test of the split-precision contractions every tile kernel runs on (csrc/mfma_split.h: three f16 MFMAs per f32 product,
 * f32 accumulation), W [f][64] row-major with f in {64, 128}:
 *   mode 0  Y[rows,f]  = X[rows,64] . W^T   forward operands, split image [plane][k/32][g][f][8]
 *   mode 1  Y[rows,64] = X[rows,f]  . W     adjoint operands (rows scaled by a power of two), split image of W^T
 *   mode 2 / 3  the same two products from ONE row-major image (forward ds_read_b64, adjoint ds_read_b64_tr_b16) */
int chg_test_split_gemm(chg_engine* eng, const float* x, const float* w, float* y, int rows, int f, int mode);

/* ---- exchange steps of the multi-GPU path, straight on RCCL (one communicator per process = per GPU) ----------
 * The reference is single-device; these carry what SURVEY 8e needs and nothing else: the all-gather of per-structure
 * energies after a sweep sharded over independent structures, and the sum of the 412,525-float parameter gradient of a
 * data-parallel train step (the slot is loss.backward() -> optimizer.step(), chgnet/trainer/trainer.py:399-411).
 * librccl is opened at run time (CHG_EUNSUPPORTED when it is missing).  Rank 0 calls chg_comm_unique_id and hands the
 * CHG_COMM_ID_BYTES bytes to the other ranks by any means (chgnet_amd/distributed.py uses a TCP socket on MASTER_ADDR);
 * every rank then calls chg_comm_create.  Buffers are HOST pointers; counts are per rank; all calls block until done.
 * all_gather: recv holds world * count floats in rank order. */
#define CHG_COMM_ID_BYTES 128
typedef struct chg_comm chg_comm;
int chg_comm_unique_id(uint8_t* id_out);
int chg_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device, chg_comm** out);
int chg_comm_all_gather_f32(chg_comm* comm, const float* send, int64_t count, float* recv);
int chg_comm_all_reduce_sum_f32(chg_comm* comm, float* data, int64_t count);
/* Device-pointer forms, enqueued on `stream` (a hipStream_t, e.g. chg_engine_stream) without synchronisation: the two
 * exchange steps of the path act on buffers that already live in HBM (per-structure energies, the gradient blob). */
int chg_comm_all_gather_f32_device(chg_comm* comm, const float* d_send, int64_t count, float* d_recv, void* stream);
int chg_comm_all_reduce_sum_f32_device(chg_comm* comm, float* d_data, int64_t count, void* stream);
int chg_comm_reserve(chg_comm* comm, int64_t floats, float** device_ptr);   /* grow-only device staging of the communicator */
int chg_comm_info(chg_comm* comm, int32_t* rank, int32_t* nranks, int32_t* device);   /* nranks = ncclCommCount */
int chg_comm_barrier(chg_comm* comm);
int chg_comm_destroy(chg_comm* comm);
const char* chg_comm_last_error(const chg_comm* comm);   /* NULL: the error of the last failed call without a communicator */

#ifdef __cplusplus
}
#endif
#endif /* CHGNET_HIP_H */
