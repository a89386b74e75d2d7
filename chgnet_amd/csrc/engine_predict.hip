// engine_predict.hip -- launch schedule of chg_predict: weight layout, launch helpers, the forward sweep and the force / stress
// sweep (order of layers: CHGNet._compute, reference chgnet/model/model.py:442-503; the reverse sweep is its hand-derived adjoint,
// SURVEY Appendix B), the batch arena, the centre-major atom schedule of the per-atom adjoints.
#include "engine_internal.h"

#include "kernels_geom.h"
#include "kernels_angle_fa.h"
#include "kernels_chain.h"

namespace chgh {


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
    a.w_cn = c.take(4 * D * D); a.w_bond = c.take(2 * D * D); a.b1 = c.take(2 * D); a.q_bias = c.take(2 * D); a.q_shift = c.take(D);
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

int grid_for(int rows, int max_blocks, int block_rows) {
  int ntiles = (rows + block_rows - 1) / block_rows;
  int g = std::min(ntiles, max_blocks);
  if (g >= 8) g &= ~7;  // multiple of 8: tile_range keeps neighbouring ranges on one XCD
  return std::max(g, 1);
}

// grid of a tile kernel: CHGNET_GRID_MULT workgroups per CU -- one per CU when that already gives every workgroup no more than a few
// tiles (small batches: the second workgroup of a CU would stage the weights again for one or two tiles; MD replay 1.478 -> 1.437 ms)
int tile_grid(chg_engine* eng, int rows, int block_rows) {
  const int ntiles = (rows + block_rows - 1) / block_rows;
  const int mult = ntiles <= 4 * eng->num_cus ? 1 : tile_grid_mult();
  // rounded UP to a multiple of 8 (tile_range's XCD mapping): rounding 221 blocks down to 216 left 42 waves of a 256-atom cell's
  // AtomConv kernels with a second tile, i.e. doubled the kernel's time; a workgroup without tiles costs nothing
  return std::max(1, std::min((ntiles + 7) & ~7, mult * eng->num_cus));
}

#ifdef CHG_EXPERIMENTS
static int exp_nw() {
  static const int v = [] { const char* e = std::getenv("CHGNET_EXP_NW"); return e ? std::atoi(e) : 0; }();
  return v;
}
#endif

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

// small batches: several independent row GEMMs of one K in a single launch (k_rows_gemm_multi); n_out / n_out_first as in
// launch_rows_gemm_cols.  The caller has checked that every row count is in (0, SMALL_GEMM_ROWS].
struct MultiGemm {
  RowsGemmN g{};
  int max_rows = 0;
  void add(RowsGemm p, int n_out, int n_out_first) {
    p.col_blocks = n_out / SMALL_GEMM_COLS; p.blocks1 = n_out_first / SMALL_GEMM_COLS;
    g.p[g.n++] = p;
    max_rows = std::max(max_rows, p.rows);
  }
};
template <int K>
int launch_rows_gemm_multi(chg_engine* eng, const char* label, const MultiGemm& m) {
  int blocks = 0;
  for (int i = 0; i < m.g.n; ++i) blocks += m.g.p[i].col_blocks;
  LaunchScope ls(eng, label);
  hipLaunchKernelGGL((k_rows_gemm_multi<K, SMALL_GEMM_COLS>), dim3(grid_for(m.max_rows, 4 * eng->num_cus), blocks), dim3(BLOCK),
                     (rows_gemm_lds<K, SMALL_GEMM_COLS, (K == 128 ? 2 : 1)>()), eng->stream, m.g);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}
static bool small_rows(int rows) { return rows > 0 && rows <= SMALL_GEMM_ROWS; }

int zero(chg_engine* eng, void* p, size_t bytes) {
  if (bytes == 0) return CHG_OK;
  LaunchScope ls(eng, "memset");
  HIP_TRY(eng, hipMemsetAsync(p, 0, bytes, eng->stream));
  return CHG_OK;
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
  constexpr size_t PE = (size_t)94 * 4 * D;     // per-element P table of AtomConv 0 (k_prologue)
  const size_t total = (size_t)L * (2 * AF + AB + ac_bwd_rm_image_floats()) + (size_t)(L - 1) * (BF + BB + UF + UB) + PE;
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
    img = take(ac_bwd_rm_image_floats());
    eng->img_ac_bwd_rm[l] = img;
    hipLaunchKernelGGL(k_atomconv_image_rm, dim3(1), dim3(BLOCK), 0, eng->stream, a, img);
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
  {   // P table of the first AtomConv for every element: atom[0] = emb[z], so P(0)[i] = p_elem[z_i] (same contraction as atomconv_tables)
    float* tab = take(PE);
    eng->p_elem = tab;
    const ACW& w0 = eng->w.ac[0];
    TRY(rows_gemm_out2(eng, "gemm_Pelem", eng->w.emb, nullptr, w0.w_cn, w0.w_cn + 2 * D * D, w0.b1, tab, 4 * D, 94));
  }
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

// MD-size batches: launches are merged where the data allows (k_prologue, the merged embedding launches, the chained row GEMMs) --
// a launch costs ~4.5 us before its first instruction and such a prediction is ~60 dependent ones.  CHGNET_TINY_FUSE=0: the
// launch sequence of the large batches (A/B timing, parity tests).
static bool tiny_batch(const chg_batch* b) {
  static const bool on = [] { const char* e = std::getenv("CHGNET_TINY_FUSE"); return !e || std::atoi(e) != 0; }();
  return on && b->N <= SMALL_GEMM_ROWS && b->Ed <= (1 << 18) && b->A <= (1 << 19);
}

static RowsGemm atomconv_p_problem(chg_engine* eng, chg_batch* b, int l) {
  const ACW& w = eng->w.ac[l];
  return RowsGemm{b->atom[l], D, nullptr, w.w_cn, w.b1, nullptr, 0, b->Pl[l], 4 * D, nullptr, b->N, 0, w.w_cn + 2 * D * D, 0, 2 * D, 0, 0};
}

int atomconv_tables(chg_engine* eng, chg_batch* b, int l) {
  const ACW& w = eng->w.ac[l];
  if (b->p_table_done == l) return CHG_OK;   // contracted next to the S and R tables of BondConv l - 1 (angle_tables, small batches)
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
  a.phase = b->phase;
  a.hb0 = b->hb0;
  a.hbc = (b->Eb > 0 && b->hbc[l] != b->hbc[0]) ? b->hbc[l] : nullptr;   // bond-graph nodes carry layer-l features
  a.u_bnode = b->u_bnode;
  a.w_bond = eng->w.ac[l].w_bond;
  a.q_bias = b->A > 0 ? eng->w.ac[l].q_bias : nullptr;
  return a;
}

int atomconv_fwd_kernel(chg_engine* eng, chg_batch* b, int l, bool keep_q) {
  {
    LaunchScope ls(eng, "atomconv_fwd");
    const size_t lds = atomconv_lds<FWD_WAVES, false, true>();
    AtomConvArgs a = atomconv_args(eng, b, l);
    a.e_center = b->p_center;   // bond-pair order
    a.image = eng->img_ac_fwd[a.q_bias ? 1 : 0][l];
    a.e_nbr = b->p_nbr;
    a.Qout = keep_q ? b->Ql[l] : nullptr;   // the reverse sweep gathers the bond partial as a table
    a.interleave = interleave_mask() & 1;
#ifdef CHG_EXPERIMENTS
    if (exp_nw() == 4) {
      const size_t lds4 = std::max(atomconv_lds<4, false, true>(), (size_t)100 << 10);
      hipLaunchKernelGGL((k_atomconv_fwd<4>), dim3(tile_grid(eng, b->Ed, TILE_ROWS * 4)), dim3(64 * 4), lds4, eng->stream, a);
    } else
#endif
    hipLaunchKernelGGL((k_atomconv_fwd<FWD_WAVES>), dim3(tile_grid(eng, b->Ed, TILE_ROWS * FWD_WAVES)), dim3(64 * FWD_WAVES), lds, eng->stream, a);
    HIP_TRY(eng, hipGetLastError());
  }
  return CHG_OK;
}

int atomconv_fwd(chg_engine* eng, chg_batch* b, int l, bool keep_q) {
  const ACW& w = eng->w.ac[l];
  if (b->Ed > 0) {
    TRY(atomconv_tables(eng, b, l));
    TRY(atomconv_fwd_kernel(eng, b, l, keep_q));
  }
  // atom[l+1] = agg . Wout^T + b_out + atom[l]       (layers.py:127-132)
  return rows_gemm(eng, "gemm_out", 64, 64, b->agg_l[l], D, nullptr, w.w_out, w.b_out, b->atom[l], D, b->atom[l + 1], D, nullptr, b->N, 0);
}

// The adjoint with the dE/d h_bond update in its tiles (k_atomconv_bwd<false, true>: no dE/dQ table, no gemm_GQ).
// Same box, ms per headline step: table + gemm_GQ 4.52 + 1.38; fused 5.23: -0.6 ms, and one launch fewer per layer for MD-size
// batches.  CHGNET_FUSE_GQ=0 switches back for A/B timing.
constexpr size_t acb_fused_lds() { return sizeof(float) * ((size_t)ac_bwd_rm_image_floats() + WAVES * TILE_FLOATS); }
static bool fuse_gq() {
  static const bool on = [] { const char* e = std::getenv("CHGNET_FUSE_GQ"); return !e || std::atoi(e) != 0; }();
  return on;
}

int atomconv_bwd_kernel(chg_engine* eng, chg_batch* b, int l, bool fused) {
  {  // pair-ordered edge list: GQ (or, fused, Gb) and Gwag rows are owned by one tile each (no zeroing, no atomics)
    AtomConvArgs a = atomconv_args(eng, b, l);
    a.e_center = b->p_center;
    a.e_nbr = b->p_nbr;
    a.image = eng->img_ac_bwd[l];
    a.interleave = (interleave_mask() >> 1) & 1;
    a.Gb = b->Gb;
    a.gb_accumulate = l == b->L - 1 ? 0 : 1;
    LaunchScope ls(eng, "atomconv_bwd");
    if (fused) {
      a.image = eng->img_ac_bwd_rm[l];
      hipLaunchKernelGGL((k_atomconv_bwd<false, true>), dim3(tile_grid(eng, b->Ed)), dim3(BLOCK), acb_fused_lds(), eng->stream, a);
    } else
      hipLaunchKernelGGL(k_atomconv_bwd<false>, dim3(tile_grid(eng, b->Ed)), dim3(BLOCK), (atomconv_lds<WAVES, true>()), eng->stream, a);
    HIP_TRY(eng, hipGetLastError());
  }
  return CHG_OK;
}

int atomconv_bwd(chg_engine* eng, chg_batch* b, int l) {
  const ACW& w = eng->w.ac[l];
  if (b->Ed == 0) return CHG_OK;  // agg == 0: only the residual path, already in Ga
  TRY(rows_gemm(eng, "gemm_Gagg", 64, 64, b->Ga, D, nullptr, w.w_out_t, nullptr, nullptr, 0, b->GA, D, nullptr, b->N, 0));
  const bool fused = fuse_gq();
  TRY(atomconv_bwd_kernel(eng, b, l, fused));
  if (fused) {
    if (l > 0) TRY(rows_gemm_in2(eng, "gemm_GP", b->GP_l[l], 4 * D, w.w_cn_t, w.w_cn_t + 2 * D * D, b->Ga, nullptr, b->N, 1));
    return CHG_OK;
  }
  if (l > 0 && small_rows(b->N) && small_rows(b->Eu)) {   // small batch: both in one launch (targets: atom rows, bond rows)
    MultiGemm m;
    m.add(RowsGemm{b->GP_l[l], 4 * D, nullptr, w.w_cn_t, nullptr, nullptr, 0, b->Ga, D, nullptr, b->N, 1, w.w_cn_t + 2 * D * D, 2 * D, 0, 0, 0}, D, D);
    m.add(RowsGemm{b->GQ, 2 * D, nullptr, w.w_bond_t, nullptr, nullptr, 0, b->Gb, D, nullptr, b->Eu, l == b->L - 1 ? 0 : 1, nullptr, 0, 0, 0, 0}, D, D);
    return launch_rows_gemm_multi<128>(eng, "gemm_GPQ", m);
  }
  if (l > 0) {  // dE/d atom[l] += GPc . Wc + GPn . Wn   (atom[0] is an embedding: no position dependence)
    TRY(rows_gemm_in2(eng, "gemm_GP", b->GP_l[l], 4 * D, w.w_cn_t, w.w_cn_t + 2 * D * D, b->Ga, nullptr, b->N, 1));
  }
  return rows_gemm(eng, "gemm_GQ", 128, 64, b->GQ, 2 * D, nullptr, w.w_bond_t, nullptr, nullptr, 0, b->Gb, D, nullptr, b->Eu, l == b->L - 1 ? 0 : 1);
}

// ---- BondConv / AngleUpdate ----------------------------------------------------------------------------
// tables: S = atom . Wctr^T + b1 (per atom),  R = hbc . [Wi;Wj]^T (per bond-graph node)
// p_layer >= 0 (BondConv l: atom = atom[l + 1]): the P table of AtomConv p_layer = l + 1, which reads the same atom rows, joins a
// small batch's launch
int angle_tables(chg_engine* eng, chg_batch* b, int slot, const float* atom, const float* hbc, const float* w_bij, const float* w_ctr, const float* b1,
                 int p_layer = -1) {
  float *S = b->Sl[slot], *R = b->Rl[slot];
  if (small_rows(b->N) && small_rows(b->Eb)) {   // small batch: the tables in one launch
    MultiGemm m;
    m.add(RowsGemm{atom, D, nullptr, w_ctr, b1, nullptr, 0, S, 2 * D, nullptr, b->N, 0, nullptr, 0, 0, 0, 0}, 2 * D, 2 * D);
    m.add(RowsGemm{hbc, D, nullptr, w_bij, nullptr, nullptr, 0, R, 4 * D, nullptr, b->Eb, 0, w_bij + 2 * D * D, 0, 2 * D, 0, 0}, 4 * D, 2 * D);
    if (p_layer >= 0 && b->Ed > 0) {
      m.add(atomconv_p_problem(eng, b, p_layer), 4 * D, 2 * D);
      b->p_table_done = p_layer;
    }
    return launch_rows_gemm_multi<64>(eng, m.g.n == 3 ? "gemm_SRP" : "gemm_SR", m);
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
  // the forward keeps z for the adjoint only when an adjoint that reads it follows (per-atom / TEAM kernels, kernels_angle_w.h)
  a.zsave = (b->zsave_now && (b->win_built || b->win_team > 0)) ? b->zsave_l[slot] : nullptr;
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

// AngleUpdate forward per atom, table rows in LDS (kernels_angle_fa.h).  CHGNET_PER_ATOM_FWD=0 switches back to the row-order kernel.
static bool per_atom_forward() {
  static const bool on = [] { const char* e = std::getenv("CHGNET_PER_ATOM_FWD"); return !e || std::atoi(e) != 0; }();
  return on;
}

template <bool HIDDEN, bool BWD, int NW = WAVES>
int launch_angle(chg_engine* eng, const char* label, chg_batch* b, const AngleArgs& a) {
  LaunchScope ls(eng, label);
  AngleArgs plain = a;
  if (!BWD && !HIDDEN && b->win_built && per_atom_forward()) {
    AngleWArgs w{};
    w.a = a; w.w = b->win;
    w.a.image = eng->img_angle[0][a.slot];
    hipLaunchKernelGGL(k_angleupd_fwd_a, dim3(b->win_grid), dim3(BLOCK), angle_fa_lds(), eng->stream, w);
    HIP_TRY(eng, hipGetLastError());
    if (b->canonical) return CHG_OK;
  } else if (BWD && b->blk_cap > 0) {
    // small batch: self-contained blocked tiles (kernels_angle_blk.h; built on the device: the index is valid by construction)
    AngleBlkArgs w{};
    w.a = a; w.a.image = eng->img_angle[1][a.slot];
    w.x.tiles = b->blk_tiles; w.x.a = b->blk_a; w.x.b1c = b->blk_b1c; w.x.b2c = b->blk_b2c; w.x.ctr = b->blk_ctr; w.x.desc = b->blk_desc;
    w.x.flag = b->canonical ? nullptr : b->win.flag;      // uploaded: valid only if the graph has the canonical angle structure
    const int grid = std::max(1, std::min(eng->num_cus, (b->blk_cap + WAVES - 1) / WAVES));
    hipLaunchKernelGGL((k_angle_bwd_blk<HIDDEN>), dim3(grid), dim3(BLOCK), angle_blk_lds<HIDDEN>(), eng->stream, w);
    HIP_TRY(eng, hipGetLastError());
    if (b->canonical) return CHG_OK;                      // else the row-order kernel below: it returns at once when the flag says 1
  } else if (BWD && b->win_team > 0) {
    // MD-size batch: an atom per team of waves (kernels_angle_w.h TEAM); the row-order kernel below returns at once unless the graph
    // turned out not to have the canonical angle structure
    AngleWArgs w{};
    w.a = a; w.w = b->win; w.n_atoms = b->N;
    w.a.image = eng->img_angle[1][a.slot];
    if (HIDDEN && w.a.zsave) hipLaunchKernelGGL((k_angle_bwd_w<HIDDEN, true, HIDDEN>), dim3(b->win_grid), dim3(BLOCK), angle_w_lds<HIDDEN>(), eng->stream, w);
    else
    hipLaunchKernelGGL((k_angle_bwd_w<HIDDEN, true>), dim3(b->win_grid), dim3(BLOCK), angle_w_lds<HIDDEN>(), eng->stream, w);
    HIP_TRY(eng, hipGetLastError());
    if (b->canonical) return CHG_OK;    // built on the device: the index is valid by construction (a launch less per layer: ~4.5 us each)
  } else if (BWD && b->win_built && per_atom_adjoint(HIDDEN)) {
    // per-atom adjoint (kernels_angle_w.h) when the batch has the canonical angle structure, else the row-order one: both are
    // launched, the device flag picks (no host round trip, and a captured hipGraph stays valid across rebuilt graphs)
    AngleWArgs w{};
    w.a = a; w.w = b->win;
    w.a.image = eng->img_angle[1][a.slot];
    if (HIDDEN && w.a.zsave) hipLaunchKernelGGL((k_angle_bwd_w<HIDDEN, false, HIDDEN>), dim3(b->win_grid), dim3(BLOCK), angle_w_lds<HIDDEN>(), eng->stream, w);
    else
    hipLaunchKernelGGL((k_angle_bwd_w<HIDDEN>), dim3(b->win_grid), dim3(BLOCK), angle_w_lds<HIDDEN>(), eng->stream, w);
    HIP_TRY(eng, hipGetLastError());
    if (b->canonical) return CHG_OK;
  } else {
    plain.skip_flag = nullptr;
  }
  plain.image = eng->img_angle[BWD ? 1 : 0][a.slot];
  plain.interleave = (interleave_mask() >> (BWD ? 4 : HIDDEN ? 2 : 3)) & 1;
#ifdef CHG_EXPERIMENTS
  // occupancy probe (profiles/r05_experiments.md): the same kernel with FOUR waves per workgroup and the LDS request padded so that
  // one workgroup fits per CU -> one wave per SIMD.  t(1 wave) / t(2 waves) says how much of a wave's time the SIMD is free.
  if (exp_nw() == 4 && !BWD && NW == WAVES) {
    const size_t lds4 = std::max(angle_lds<HIDDEN, 4, BWD>(), (size_t)100 << 10);
    hipLaunchKernelGGL((k_angle<HIDDEN, BWD, 4>), dim3(tile_grid(eng, b->A, TILE_ROWS * 4)), dim3(64 * 4), lds4, eng->stream, plain);
    HIP_TRY(eng, hipGetLastError());
    return CHG_OK;
  }
#endif
  const size_t lds = angle_lds<HIDDEN, NW, BWD>();
  hipLaunchKernelGGL((k_angle<HIDDEN, BWD, NW>), dim3(tile_grid(eng, b->A, TILE_ROWS * NW)), dim3(64 * NW), lds, eng->stream, plain);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

int bondconv_fwd(chg_engine* eng, chg_batch* b, int l) {
  const BCW& w = eng->w.bc[l];
  TRY(angle_tables(eng, b, l, b->atom[l + 1], b->hbc[l], w.w_bij, w.w_ctr, w.b1, l + 1));
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
  if (small_rows(b->N) && small_rows(b->Eb)) {   // small batch: both in one launch (different targets: bond rows, atom rows)
    MultiGemm m;
    m.add(RowsGemm{b->GR_l[slot], 4 * D, nullptr, w_bij_t, nullptr, nullptr, 0, b->Gb, D, b->bn_und, b->Eb, 1, w_bij_t + 2 * D * D, 2 * D, 0, 0, 0}, D, D);
    m.add(RowsGemm{b->GS_l[slot], 2 * D, nullptr, w_ctr_t, nullptr, nullptr, 0, b->Ga, D, nullptr, b->N, 1, nullptr, 0, 0, 0, 0}, D, D);
    return launch_rows_gemm_multi<128>(eng, "gemm_GRS", m);
  }
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
  a.ev = b->ev; a.u_u2d = b->u_u2d; a.u_bnode = b->u_bnode; a.n_und = b->Eu; a.bn_und = b->bn_und; a.n_nodes = b->Eb;
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

static int embed_grid_mult() {
  static const int m = [] { const char* e = std::getenv("CHGNET_EMBED_GRID_MULT"); return e ? std::atoi(e) : 2; }();
  return m;
}

// ---- MD-size batches: the row GEMMs between two tile kernels as ONE chained launch (kernels_chain.h) -------------------------------
struct ChainBuilder {
  ChainArgs a{};
  int max_rows = 0;
  ChainProb& add(int rows) {
    ChainProb& p = a.p[a.n++];
    p = ChainProb{};
    p.rows = rows;
    max_rows = std::max(max_rows, rows);
    return p;
  }
  static void term(ChainProb& p, const float* X, int ldx, const int* idx, const float* W, int K) { p.t[p.nterms++] = ChainTerm{X, idx, W, ldx, K}; }
  static void out(ChainProb& p, const float* W, const float* bias, float* Y, int ldy, int ncols) { p.o[p.nouts++] = ChainOut{W, bias, Y, ldy, ncols}; }
};
int launch_chain(chg_engine* eng, const char* label, ChainBuilder& cb) {
  int blocks = 0;
  for (int i = 0; i < cb.a.n; ++i) {
    ChainProb& p = cb.a.p[i];
    p.col_blocks = 0;
    for (int o = 0; o < p.nouts; ++o) p.col_blocks += p.o[o].ncols / D;
    // in-place stage 1 (Y1 is the array `add` is read from): one workgroup per row block does everything (kernels_chain.h)
    p.serial_outs = (p.Y1 && p.Y1 == p.add) ? 1 : 0;
    if (p.serial_outs) {
      if (p.nouts > 1 || (p.nouts == 1 && p.o[0].ncols != D)) { eng->err = "launch_chain: an in-place problem takes one 64-column output"; return CHG_EINVAL; }
      p.col_blocks = 1;
    }
    p.col_blocks = std::max(1, p.col_blocks);
    blocks += p.col_blocks;
  }
  if (cb.max_rows <= 0 || blocks == 0) return CHG_OK;
  int row_blocks = 1;                  // the grid covers the problem with the most row blocks (kernels_chain.h: 4 or 8 row tiles per workgroup)
  for (int i = 0; i < cb.a.n; ++i) {
    const int rows_per_wg = chain_tiles_per_wg(cb.a.p[i].nterms) * TILE_ROWS;
    row_blocks = std::max(row_blocks, (cb.a.p[i].rows + rows_per_wg - 1) / rows_per_wg);
  }
  LaunchScope ls(eng, label);
  hipLaunchKernelGGL(k_rows_chain, dim3(row_blocks, blocks), dim3(BLOCK), chain_lds(), eng->stream, cb.a);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

// Forward message passing of a small batch with bonds and angles (model.py:442-496): per layer
//   AtomConv | chain A: atom[l+1] -> S_bc, P(l+1), S_au (+ R_bc(0)) | BondConv | chain B: hbc[l+1] -> R_au, R_bc(l+1) | AngleUpdate
// -- five launches where the large-batch sequence (atomconv_fwd / bondconv_fwd / angleupd_fwd above) has seven.
int forward_tiny(chg_engine* eng, chg_batch* b, bool want_grad, bool want_m) {
  const Weights& w = eng->w;
  const int L = b->L;
  for (int l = 0; l < L - 1; ++l) {
    TRY(atomconv_fwd_kernel(eng, b, l, want_grad));
    {
      ChainBuilder cb;
      ChainProb& p = cb.add(b->N);                       // atom[l+1] = agg . Wout^T + b_out + atom[l]   (layers.py:127-132)
      cb.term(p, b->agg_l[l], D, nullptr, w.ac[l].w_out, D);
      p.bias1 = w.ac[l].b_out; p.add = b->atom[l]; p.lda = D; p.Y1 = b->atom[l + 1]; p.ldy1 = D;
      cb.out(p, w.bc[l].w_ctr, w.bc[l].b1, b->Sl[l], 2 * D, 2 * D);                                        // S of BondConv l
      cb.out(p, w.ac[l + 1].w_cn, w.ac[l + 1].b1, b->Pl[l + 1], 4 * D, 2 * D);                              // P of AtomConv l + 1: centre half
      cb.out(p, w.ac[l + 1].w_cn + 2 * D * D, nullptr, b->Pl[l + 1] + 2 * D, 4 * D, 2 * D);                 //   ... neighbour half
      if (l < L - 2) cb.out(p, w.au[l].w_ctr, w.au[l].b1, b->Sl[L + l], 2 * D, 2 * D);                      // S of AngleUpdate l
      if (l == 0) {                                      // R of BondConv 0 from the embedded bond features (later layers: chain B)
        ChainProb& q = cb.add(b->Eb);
        q.add = b->hbc[0]; q.lda = D;
        cb.out(q, w.bc[0].w_bij, nullptr, b->Rl[0], 4 * D, 2 * D);
        cb.out(q, w.bc[0].w_bij + 2 * D * D, nullptr, b->Rl[0] + 2 * D, 4 * D, 2 * D);
      }
      TRY(launch_chain(eng, "chain_A", cb));
    }
    TRY((launch_angle<true, false>(eng, "bondconv_fwd", b, angle_args(b, l, b->ang[l], w.bc[l].w_ang, w.bc[l].g, b->aggB_l[l]))));
    {
      ChainBuilder cb;
      ChainProb& p = cb.add(b->Eb);                      // hbc[l+1] = agg . Wout^T + b_out + hbc[l]     (layers.py:255-260)
      cb.term(p, b->aggB_l[l], D, nullptr, w.bc[l].w_out, D);
      p.bias1 = w.bc[l].b_out; p.add = b->hbc[l]; p.lda = D; p.Y1 = b->hbc[l + 1]; p.ldy1 = D;
      if (l < L - 2) {
        cb.out(p, w.au[l].w_bij, nullptr, b->Rl[L + l], 4 * D, 2 * D);                                      // R of AngleUpdate l
        cb.out(p, w.au[l].w_bij + 2 * D * D, nullptr, b->Rl[L + l] + 2 * D, 4 * D, 2 * D);
        cb.out(p, w.bc[l + 1].w_bij, nullptr, b->Rl[l + 1], 4 * D, 2 * D);                                  // R of BondConv l + 1
        cb.out(p, w.bc[l + 1].w_bij + 2 * D * D, nullptr, b->Rl[l + 1] + 2 * D, 4 * D, 2 * D);
      }
      TRY(launch_chain(eng, "chain_B", cb));
    }
    if (l < L - 2)
      TRY((launch_angle<false, false, FWD_WAVES>(eng, "angleupd_fwd", b, angle_args(b, L + l, b->ang[l], w.au[l].w_ang, w.au[l].g, b->ang[l + 1]))));
  }
  if (want_m) {
    LaunchScope ls(eng, "magmom");
    hipLaunchKernelGGL(k_magmom, dim3(wave_grid(eng, b->N)), dim3(256), 0, eng->stream, b->atom[L - 1], w.site_w, w.site_b, b->magmom, b->N);
  }
  b->p_table_done = L - 1;                               // chain A of layer L - 2 contracted it
  return atomconv_fwd(eng, b, L - 1, want_grad);
}

// ... and its reverse sweep: per layer
//   AngleUpdate adj. | chain B': Gb[bn] += GR_au . W -> Gagg  (+ Ga += GS_au . W) | BondConv adj. |
//   chain A': Ga += GS_bc . W + GP(l+1) . W -> GA  (+ Gb[bn] += GR_bc . W) | AtomConv adj.
// -- five launches for the eight of atomconv_bwd / bondconv_bwd / angleupd_bwd.  The P-table gradient of AtomConv l + 1 joins the atom
// rows one launch later than in the large-batch order; nothing reads Ga in between.
int reverse_tiny(chg_engine* eng, chg_batch* b) {
  const Weights& w = eng->w;
  const int L = b->L;
  TRY(rows_gemm(eng, "gemm_Gagg", 64, 64, b->Ga, D, nullptr, w.ac[L - 1].w_out_t, nullptr, nullptr, 0, b->GA, D, nullptr, b->N, 0));
  TRY(atomconv_bwd_kernel(eng, b, L - 1, true));
  for (int l = L - 2; l >= 0; --l) {
    const bool au = l < L - 2;
    if (au) TRY((launch_angle<false, true>(eng, "angleupd_bwd", b, angle_args(b, L + l, b->ang[l], w.au[l].w_ang, w.au[l].g, nullptr))));
    {
      ChainBuilder cb;
      ChainProb& p = cb.add(b->Eb);                      // dE/d hbc[l+1] complete -> dE/d agg of BondConv l
      if (au) {
        cb.term(p, b->GR_l[L + l], 4 * D, nullptr, w.au[l].w_bij_t, 2 * D);
        cb.term(p, b->GR_l[L + l] + 2 * D, 4 * D, nullptr, w.au[l].w_bij_t + 2 * D * D, 2 * D);
        p.Y1 = b->Gb; p.y1_idx = b->bn_und; p.ldy1 = D;
      }
      p.add = b->Gb; p.add_idx = b->bn_und; p.lda = D;
      cb.out(p, w.bc[l].w_out_t, nullptr, b->Gagg, D, D);
      if (au) {
        ChainProb& q = cb.add(b->N);                     // Ga += GS_au . Wctr
        cb.term(q, b->GS_l[L + l], 2 * D, nullptr, w.au[l].w_ctr_t, 2 * D);
        q.add = b->Ga; q.lda = D; q.Y1 = b->Ga; q.ldy1 = D;
      }
      TRY(launch_chain(eng, "chain_Bt", cb));
    }
    TRY((launch_angle<true, true>(eng, "bondconv_bwd", b, angle_args(b, l, b->ang[l], w.bc[l].w_ang, w.bc[l].g, nullptr))));
    {
      ChainBuilder cb;
      ChainProb& p = cb.add(b->N);                       // dE/d atom[l+1] complete -> dE/d agg of AtomConv l
      cb.term(p, b->GS_l[l], 2 * D, nullptr, w.bc[l].w_ctr_t, 2 * D);
      cb.term(p, b->GP_l[l + 1], 4 * D, nullptr, w.ac[l + 1].w_cn_t, 2 * D);
      cb.term(p, b->GP_l[l + 1] + 2 * D, 4 * D, nullptr, w.ac[l + 1].w_cn_t + 2 * D * D, 2 * D);
      p.add = b->Ga; p.lda = D; p.Y1 = b->Ga; p.ldy1 = D;
      cb.out(p, w.ac[l].w_out_t, nullptr, b->GA, D, D);
      ChainProb& q = cb.add(b->Eb);                      // Gb[bn] += GR_bc . [Wi;Wj]
      cb.term(q, b->GR_l[l], 4 * D, nullptr, w.bc[l].w_bij_t, 2 * D);
      cb.term(q, b->GR_l[l] + 2 * D, 4 * D, nullptr, w.bc[l].w_bij_t + 2 * D * D, 2 * D);
      q.add = b->Gb; q.add_idx = b->bn_und; q.lda = D; q.Y1 = b->Gb; q.y1_idx = b->bn_und; q.ldy1 = D;
      TRY(launch_chain(eng, "chain_At", cb));
    }
    TRY(atomconv_bwd_kernel(eng, b, l, true));
  }
  return CHG_OK;
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
  const bool tiny = tiny_batch(b);
  b->zsave_now = want_grad;
  const size_t zero_bytes = (size_t)((char*)(want_grad ? b->zero2_end : b->zero1_end) - (char*)b->zero1);
  b->p_table_done = -1;
  if (tiny) {   // coordinates, bond vectors, atom embedding, the first P table and the cleared scatter targets: one launch
    PrologueArgs a{};
    a.frac = b->frac; a.lattice = b->lattice; a.atom_owner = b->atom_owner; a.z = b->z; a.n_atoms = b->N;
    a.e_center = b->e_center; a.e_nbr = b->e_nbr; a.e_owner = b->e_owner; a.e_image = b->e_image; a.n_edges = b->Ed;
    a.cart = b->cart; a.ev = b->ev; a.eu = b->eu; a.emb = w.emb; a.atom0 = b->atom[0];
    a.p_elem = b->Ed > 0 ? eng->p_elem : nullptr; a.P0 = b->Pl[0];
    a.zero_begin = reinterpret_cast<f32x4*>(b->zero1); a.zero_n = zero_bytes / sizeof(f32x4);
    LaunchScope ls(eng, "prologue");
    hipLaunchKernelGGL(k_prologue, dim3(std::min<unsigned>(4 * eng->num_cus, g1(std::max<int64_t>(b->Ed, (int64_t)b->N * D)).x)), dim3(256), 0, st, a);
    if (a.p_elem) b->p_table_done = 0;
  } else {
    LaunchScope ls(eng, "cart");
    hipLaunchKernelGGL(k_cart, g1(b->N), dim3(256), 0, st, b->frac, b->lattice, b->atom_owner, b->cart, b->N);
  }
  if (b->Ed > 0) {
    if (!tiny) { LaunchScope ls(eng, "edge_geom");
      hipLaunchKernelGGL(k_edge_geom, g1(b->Ed), dim3(256), 0, st, b->cart, b->lattice, b->e_center, b->e_nbr, b->e_image, b->e_owner, b->ev, b->eu, b->Ed); }
    if (tiny) {   // both parts of the bond expansion and the angle expansion: one launch
      EmbedAllArgs ea{bond_embed_args(eng, b), angle_embed_args(eng, b), grid_for(b->Eu, 2 * eng->num_cus),
                      b->Eb > 0 ? grid_for(b->Eb, 2 * eng->num_cus) : 0, b->A > 0 ? grid_for(b->A, embed_grid_mult() * eng->num_cus) : 0};
      LaunchScope ls(eng, "embed_fwd");
      hipLaunchKernelGGL(k_embed_all<false>, dim3(ea.g_bond1 + ea.g_bond2 + ea.g_angle), dim3(BLOCK), bond_embed_lds(), st, ea);
    } else { LaunchScope ls(eng, "bond_embed_fwd");   // atom-graph expansion for every bond, bond-graph expansion for the bond-graph nodes only
      static const int o4 = [] { const char* e = std::getenv("CHGNET_EMBED_O4"); return e ? std::atoi(e) : 1; }();
      if (o4) hipLaunchKernelGGL((k_bond_embed_fwd_o4<1>), dim3(grid_for(b->Eu, o4 * 2 * eng->num_cus)), dim3(BLOCK), bond_embed_lds(), st, bond_embed_args(eng, b));
      else
      hipLaunchKernelGGL((k_bond_embed_t<false, false, 1>), dim3(grid_for(b->Eu, 2 * eng->num_cus)), dim3(BLOCK), bond_embed_lds(), st, bond_embed_args(eng, b));
      if (b->Eb > 0)
        hipLaunchKernelGGL((k_bond_embed_t<false, false, 2>), dim3(grid_for(b->Eb, 2 * eng->num_cus)), dim3(BLOCK), bond_embed_lds(), st, bond_embed_args(eng, b)); }
  }
  if (b->A > 0 && !(tiny && b->Ed > 0)) {
    LaunchScope ls(eng, "angle_embed_fwd");
    static const int o4 = [] { const char* e = std::getenv("CHGNET_EMBED_O4"); return e ? std::atoi(e) : 1; }();
    if (o4) hipLaunchKernelGGL((k_angle_embed_fwd_o4<0>), dim3(grid_for(b->A, o4 * embed_grid_mult() * eng->num_cus)), dim3(BLOCK), angle_embed_lds(), st, angle_embed_args(eng, b));
    else
    hipLaunchKernelGGL((k_angle_embed_t<false>), dim3(grid_for(b->A, embed_grid_mult() * eng->num_cus)), dim3(BLOCK), angle_embed_lds(), st, angle_embed_args(eng, b));
  }
  if (!tiny) { LaunchScope ls(eng, "atom_embed");
    hipLaunchKernelGGL(k_atom_embed, g1((int64_t)b->N * (D / 4)), dim3(256), 0, st, b->z, w.emb, b->atom[0], b->N); }
  HIP_TRY(eng, hipGetLastError());   // (hbc[0], the nodes' copy of their embedding rows, is written by k_bond_embed_t)

  // ---- message passing (model.py:442-496) ----
  // every forward scatter target + crystal_fea -- and, when a reverse sweep follows, its accumulators too (the two ranges are
  // adjacent in the arena: one memset instead of two)
  if (!tiny) TRY(zero(eng, b->zero1, zero_bytes));
  // small batches with bonds and angles: the chained schedule (forward_tiny / reverse_tiny); CHGNET_TINY_CHAIN=0 keeps the launch
  // sequence of the large batches
  static const bool chain_on = [] { const char* e = std::getenv("CHGNET_TINY_CHAIN"); return !e || std::atoi(e) != 0; }();
  const bool chained = tiny && chain_on && fuse_gq() && L >= 2 && b->Ed > 0 && b->A > 0 && b->Eb > 0 && b->p_table_done == 0;
  if (chained) {
    TRY(forward_tiny(eng, b, want_grad, want_m));
  } else {
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
  }

  // ---- readout (model.py:497-509) and its adjoint ----
  {
    ReadoutArgs r{};
    r.atom = b->atom[L]; r.atom_owner = b->atom_owner; r.z = b->z; r.n_atoms = b->N;
    r.ln_g = w.ro_ln_g; r.ln_b = w.ro_ln_b; r.w0 = w.mlp_w0; r.b0 = w.mlp_b0; r.w1 = w.mlp_w1; r.b1 = w.mlp_b1;
    r.w2 = w.mlp_w2; r.b2 = w.mlp_b2; r.w3 = w.mlp_w3; r.b3 = w.mlp_b3; r.atomref = w.atomref;
    r.has_composition = eng->desc.has_composition; r.n_hidden = eng->desc.n_mlp_hidden;
    r.site_energy = b->site_energy; r.site_raw = b->site_raw; r.crystal_fea = b->crystal_fea;
    r.Ga = want_grad ? b->Ga : nullptr;
    // few atoms: fewer working waves per workgroup, more workgroups (ReadoutArgs::wpb)
    const int tiles16 = (b->N + TILE_ROWS - 1) / TILE_ROWS;
    r.wpb = std::max(1, std::min(WAVES, (tiles16 + eng->num_cus - 1) / eng->num_cus));
    LaunchScope ls(eng, "readout");
    hipLaunchKernelGGL(k_readout<false>, dim3(grid_for(b->N, eng->num_cus, r.wpb * TILE_ROWS)), dim3(BLOCK), readout_lds(), st, r);
    HIP_TRY(eng, hipGetLastError());
  }

  // ---- reverse sweep: dE/dv_e ----
  if (want_grad) {
    if (chained) {
      TRY(reverse_tiny(eng, b));
    } else {
    TRY(atomconv_bwd(eng, b, L - 1));
    for (int l = L - 2; l >= 0; --l) {
      if (b->A > 0) {
        if (l < L - 2) TRY(angleupd_bwd(eng, b, l));
        TRY(bondconv_bwd(eng, b, l));
      }
      TRY(atomconv_bwd(eng, b, l));
    }
    }
    if (b->Ed > 0) {
      if (tiny) {
        EmbedAllArgs ea{bond_embed_args(eng, b), angle_embed_args(eng, b), grid_for(b->Eu, 2 * eng->num_cus),
                        b->Eb > 0 ? grid_for(b->Eb, 2 * eng->num_cus) : 0, b->A > 0 ? grid_for(b->A, embed_grid_mult() * eng->num_cus) : 0};
        LaunchScope ls(eng, "embed_bwd");
        hipLaunchKernelGGL(k_embed_all<true>, dim3(ea.g_bond1 + ea.g_bond2 + ea.g_angle), dim3(BLOCK), bond_embed_lds(), st, ea);
      } else { LaunchScope ls(eng, "bond_embed_bwd");
        hipLaunchKernelGGL((k_bond_embed_t<true, false, 1>), dim3(grid_for(b->Eu, 2 * eng->num_cus)), dim3(BLOCK), bond_embed_lds(), st, bond_embed_args(eng, b));
        if (b->Eb > 0)
          hipLaunchKernelGGL((k_bond_embed_t<true, false, 2>), dim3(grid_for(b->Eb, 2 * eng->num_cus)), dim3(BLOCK), bond_embed_lds(), st, bond_embed_args(eng, b)); }
      if (b->A > 0 && !tiny) {
        LaunchScope ls(eng, "angle_embed_bwd");
        hipLaunchKernelGGL((k_angle_embed_t<true>), dim3(grid_for(b->A, embed_grid_mult() * eng->num_cus)), dim3(BLOCK), angle_embed_lds(), st, angle_embed_args(eng, b));
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
  b->q_tables = want_grad;      // atomconv_fwd(keep_q = want_grad) left the bond partials behind as tables
  return CHG_OK;
}

// ---- arena ---------------------------------------------------------------------------------------------

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
  // z rows of the BondConv layers for their adjoints (large batches: AngleArgs::zsave; + 16 spare rows for the unconditional stores of
  // the last tile).  Same box, headline batch, per launch: bondconv_bwd 2.40 -> 2.06 ms (no gathers of three table rows and the angle
  // row, no W_ang contraction), bondconv_fwd 0.90 -> 1.16 ms (2.1 GB of stores next to its gathers), step 25.97 -> 25.75 ms; 6.3 GB of
  // a 1024-structure batch's 21 GB.  CHGNET_ZSAVE=0: off.  (The AngleUpdate layers would need the store in the per-atom forward
  // kernel, whose tiles are scheduled around ONE delayed store stream: not done.)
  static const bool zsave_on = [] { const char* e = std::getenv("CHGNET_ZSAVE"); return !e || std::atoi(e) != 0; }();
  for (int t = 0; t < 2 * L; ++t)
    b->zsave_l[t] = (zsave_on && t < L - 1 && A > ((size_t)1 << 19)) ? c.take<float>((A + TILE_ROWS) * 2 * D) : nullptr;
  b->site_energy = c.take<float>(N); b->site_raw = c.take<float>(N); b->volume = c.take<float>(B);
  // zero group 1 (cleared with one memset before the readout)
  b->zero1 = c.take<float>(0);
  for (int l = 0; l < L; ++l) b->agg_l[l] = c.take<float>(N * D);
  for (int l = 0; l < L - 1; ++l) b->aggB_l[l] = c.take<float>(Eb * D);
  // the results a download copies sit side by side across the border of the two groups -- crystal_fea | force | virial | energy | magmom --
  // so that an MD-size download is ONE device-to-host copy (chg_batch_download: every copy is ~5 us of dependent device time); energy
  // and magmom are plain stores, clearing them with group 2 is harmless
  b->crystal_fea = c.take<float>(B * D);
  b->zero1_end = c.take<float>(0);
  // zero group 2 (cleared with one memset before the reverse sweep)
  b->zero2 = c.take<float>(0);
  b->force = c.take<float>(3 * N); b->virial = c.take<float>(9 * B); b->energy = c.take<float>(B); b->magmom = c.take<float>(N);
  b->zero2_keep_end = c.take<float>(0);   // (chg_backward clears group 2 again AFTER the prediction: energy / magmom are skipped there)
  b->Gwbgc = c.take<float>(Eb * D);
  b->Gu = c.take<float>(4 * Ed);
  b->Grk = c.take<float>(Eu);   // (cleared for the merged embedding adjoint of small batches, whose two parts add into it concurrently)
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
  b->phase = c.take<float>(PHASE_FLOATS);
  {   // windowed angle adjoints (kernels_angle_w.h)
    WinIndex& w = b->win;
    w.flag = c.take<int>(4); w.na = c.take<int>(N + 1); w.boff = c.take<int>(N + 1); w.aoff = c.take<int>(N + 1); w.toff = c.take<int>(N + 1); w.toff4 = c.take<int>(N + 1);
    w.head = c.take<int>(Ed); w.rank = c.take<int>(Ed); w.list = c.take<int>(A ? N * WIN_LIST : 0);
    w.q_a = c.take<int>(A); w.q_ctr = c.take<int>(A); w.q_b1c = c.take<int>(A); w.q_b2c = c.take<int>(A); w.q_ab1 = c.take<int>(A); w.q_ab2 = c.take<int>(A);
    w.abbond = c.take<int>(2 * Eb);
    w.wave_head = c.take<int>(A ? WIN_MAX_GRID * WAVES : 0); w.next_atom = c.take<int>(A ? N : 0); w.xatom = c.take<int>(WIN_MAX_GRID / 8 + 1);
    b->win_tmp = c.take<int>(N + 1);
    b->win_scan = c.take<int>(scan_scratch_ints((int)N + 1));
  }
  {   // blocked tiles of the MD-size adjoints (kernels_angle_blk.h)
    const size_t slots = (size_t)b->blk_cap * TILE_ROWS;
    b->blk_a = c.take<int>(slots); b->blk_b1c = c.take<int>(slots); b->blk_b2c = c.take<int>(slots); b->blk_ctr = c.take<int>(slots);
    b->blk_desc = c.take<int>(b->blk_cap);
    b->blk_tiles = c.take<int>(b->blk_cap ? 4 : 0);
  }
  if (A == 0) for (int l = 1; l < L; ++l) b->hbc[l] = b->hbc[0];   // no BondConv: bond features never change
  total = (c.pos + 255) & ~size_t(255);
}

// Centre-major row order and (atom, bond) pair indices of the angle adjoints (kernels_angle_w.h): once per batch topology,
// stream-ordered, no host round trip; a graph without the canonical structure leaves win.flag[0] = 0.
int ensure_windows(chg_engine* eng, chg_batch* b) {
  if (!b->win_pending) return CHG_OK;
  b->win_pending = false;
  return prepare_windows(eng, b);
}

// Which angle adjoints a batch runs: the per-atom kernels (large batches), the TEAM kernels (MD-size batches), or the row-order ones.
bool decide_windows(chg_engine* eng, chg_batch* b) {
  b->win_built = false;
  b->win_team = 0;
  // one workgroup per CU (their LDS admits no second one), in whole groups of 64 waves = 8 workgroups per XCD (k_win_schedule)
  b->win_grid = std::max(64, std::min(eng->num_cus / 64 * 64, WIN_MAX_GRID));
  // MD-size batches: fewer than a few atoms per wave would leave most of the chip idle with an atom per wave
  // (CHGNET_WIN_MIN_ATOMS_PER_WAVE=0 sends small batches through the per-atom kernels too: parity tests on the golden cases)
  const char* min_env = std::getenv("CHGNET_WIN_MIN_ATOMS_PER_WAVE");
  const long min_atoms = min_env ? std::atol(min_env) : WIN_MIN_ATOMS_PER_WAVE;
  if (b->A == 0) return false;
  if ((long)b->N < min_atoms * b->win_grid * WAVES) {
    // TEAM mode (round 6): below a few atoms per wave the tiles of the atoms are dealt evenly to the workgroups and a workgroup's eight
    // waves share an atom.  Same box, thermalised Li9Co7O16 cells (displacements of 0.15 A), BondConv adjoint per launch, TEAM /
    // row-order: 256 atoms (56k angles) 60 / 57 us, 512 atoms 95 / 102, 1,024 atoms 168 / 196, 2,048 atoms 322 / 368 -- with ~1.7
    // tiles per wave the row-order kernel's 2-deep walk beats the per-atom segments (3 deep) although it sends 16x the atomic rows;
    // from ~600 atoms on the segments amortise.  Hence the default threshold of 131,072 angles (CHGNET_TEAM_MIN_ANGLES; 0 in the
    // parity tests sends every batch through it, -1 none).
    const long team_min = team_min_angles();
    if (team_min < 0 || b->A < team_min || b->N + 1 > 8192) return false;
    // one workgroup per CU, each with an equal share of the tiles (kernels_angle_w.h TEAM); fewer when that share would be below a
    // tile per wave (the tile count is a device quantity: A / 16 full tiles + at most one partial tile per atom)
    const int cus = std::max(1, std::min(eng->num_cus, WIN_MAX_GRID));
    const long tiles_max = (long)b->A / TILE_ROWS + b->N;
    b->win_team = WAVES;
    b->win_grid = (int)std::max<long>(1, std::min<long>(cus, (tiles_max + WAVES - 1) / WAVES));
    return true;
  }
  if (scan_scratch_ints(b->N + 1) > (size_t)(1u << 17)) return false;      // beyond the two-level scan (65,536 chunks): plain adjoints
  b->win_built = true;
  return true;
}

int prepare_windows(chg_engine* eng, chg_batch* b) {
  hipStream_t st = eng->stream;
  WinIndex& w = b->win;
  if (b->blk_cap > 0) {   // blocked tiles (kernels_angle_blk.h)
    b->win_built = false; b->win_team = 0;
    if (b->blk_ready) return CHG_OK;             // the graph builder wrote their index
    // an uploaded graph: the centre-major order first (ranks of the bonds at their atom; flag[0] = 0 unless the graph has the canonical
    // angle structure), then row -> slot
    b->win_grid = std::max(1, std::min(eng->num_cus, WIN_MAX_GRID));
    hipLaunchKernelGGL(k_win_clear, g1(std::max(b->Ed, b->N + 1)), dim3(256), 0, st, w, b->N, b->Ed, b->win_grid);
    hipLaunchKernelGGL(k_win_heads, g1(b->A), dim3(256), 0, st, b->a_d1, b->a_ctr, b->A, b->Ed, b->N, w);
    hipLaunchKernelGGL(k_win_scan2, dim3(1), dim3(1024), 0, st, b->N, w);
    hipLaunchKernelGGL(k_win_ranks, g1(b->A), dim3(256), 0, st, b->a_d1, b->a_ctr, b->a_b1c, b->A, b->N, w);
    hipLaunchKernelGGL(k_win_rows, g1(b->A), dim3(256), 0, st, b->a_ctr, b->a_b1c, b->a_b2c, b->a_d1, b->a_d2, b->A, b->N, b->Ed, w);
    const size_t slots = (size_t)b->blk_cap * TILE_ROWS;
    HIP_TRY(eng, hipMemsetAsync(b->blk_a, 0xFF, sizeof(int) * slots, st));
    HIP_TRY(eng, hipMemsetAsync(b->blk_b1c, 0, (size_t)((char*)(b->blk_ctr + slots) - (char*)b->blk_b1c), st));     // blk_b1c | blk_b2c | blk_ctr are neighbours (carve)
    hipLaunchKernelGGL(k_blk_from_q, g1(b->A), dim3(256), 0, st, b->A, b->N, w, b->blk_a, b->blk_b1c, b->blk_b2c, b->blk_ctr, b->blk_desc, b->blk_tiles, b->blk_cap);
    HIP_TRY(eng, hipGetLastError());
    b->blk_ready = true;
    return CHG_OK;
  }
  const bool ready = b->win_index_ready;        // chg_batch_build wrote the index with the graph (and has called decide_windows)
  if (!ready && !decide_windows(eng, b)) return CHG_OK;
  if (b->win_team > 0) {
    if (ready) return CHG_OK;
    hipLaunchKernelGGL(k_win_clear, g1(std::max(b->Ed, b->N + 1)), dim3(256), 0, st, w, b->N, b->Ed, b->win_grid);
    hipLaunchKernelGGL(k_win_heads, g1(b->A), dim3(256), 0, st, b->a_d1, b->a_ctr, b->A, b->Ed, b->N, w);
    hipLaunchKernelGGL(k_win_scan2, dim3(1), dim3(1024), 0, st, b->N, w);
    hipLaunchKernelGGL(k_win_ranks, g1(b->A), dim3(256), 0, st, b->a_d1, b->a_ctr, b->a_b1c, b->A, b->N, w);
    hipLaunchKernelGGL(k_win_rows, g1(b->A), dim3(256), 0, st, b->a_ctr, b->a_b1c, b->a_b2c, b->a_d1, b->a_d2, b->A, b->N, b->Ed, w);
    HIP_TRY(eng, hipGetLastError());
    return CHG_OK;
  }
  if (!ready) {
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
  }
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
  mi["blk_tiles"] = {b->blk_tiles, b->blk_cap ? (size_t)1 : 0}; mi["blk_a"] = {b->blk_a, (size_t)b->blk_cap * TILE_ROWS};
  mi["blk_b1c"] = {b->blk_b1c, (size_t)b->blk_cap * TILE_ROWS}; mi["blk_b2c"] = {b->blk_b2c, (size_t)b->blk_cap * TILE_ROWS};
  mi["blk_ctr"] = {b->blk_ctr, (size_t)b->blk_cap * TILE_ROWS}; mi["blk_desc"] = {b->blk_desc, (size_t)b->blk_cap};
  mi["win_wave_head"] = {b->win.wave_head, A ? (size_t)WIN_MAX_GRID * WAVES : 0}; mi["win_xatom"] = {b->win.xatom, (size_t)WIN_MAX_GRID / 8 + 1};   // win_flag[3] = workgroups
}


// dynamic-LDS attributes (more than the default 64 KiB) of the kernels this unit launches
int predict_set_lds(chg_engine* eng) {
  int s;
  if ((s = set_lds(eng, k_rows_gemm<64, 64>, rows_gemm_lds<64, 64>()))) return s;
  if ((s = set_lds(eng, k_rows_gemm<64, 128>, rows_gemm_lds<64, 128>()))) return s;
  if ((s = set_lds(eng, k_rows_gemm<128, 64>, rows_gemm_lds<128, 64>()))) return s;
  if ((s = set_lds(eng, k_rows_gemm<64, SMALL_GEMM_COLS, 1>, (rows_gemm_lds<64, SMALL_GEMM_COLS, 1>())))) return s;
  if ((s = set_lds(eng, k_rows_gemm_multi<64, SMALL_GEMM_COLS>, (rows_gemm_lds<64, SMALL_GEMM_COLS, 1>())))) return s;
  if ((s = set_lds(eng, k_rows_gemm_multi<128, SMALL_GEMM_COLS>, (rows_gemm_lds<128, SMALL_GEMM_COLS, 2>())))) return s;
  if ((s = set_lds(eng, k_rows_gemm<128, SMALL_GEMM_COLS, 1>, (rows_gemm_lds<128, SMALL_GEMM_COLS, 1>())))) return s;
  if ((s = set_lds(eng, k_rows_gemm<128, SMALL_GEMM_COLS, 2>, (rows_gemm_lds<128, SMALL_GEMM_COLS, 2>())))) return s;
  if ((s = set_lds(eng, (k_rows_gemm<64, 128, 2>), (rows_gemm_lds<64, 128, 2>())))) return s;
  if ((s = set_lds(eng, (k_rows_gemm<128, 64, 2>), (rows_gemm_lds<128, 64, 2>())))) return s;
  if ((s = set_lds(eng, k_atomconv_fwd<FWD_WAVES>, (atomconv_lds<FWD_WAVES, false, true>())))) return s;
  if ((s = set_lds(eng, k_atomconv_bwd<false>, (atomconv_lds<WAVES, true>())))) return s;
  if ((s = set_lds(eng, (k_atomconv_bwd<false, true>), acb_fused_lds()))) return s;
  if ((s = set_lds(eng, k_angleupd_fwd_a, angle_fa_lds()))) return s;
  if ((s = set_lds(eng, k_angle_bwd_blk<true>, angle_blk_lds<true>()))) return s;
  if ((s = set_lds(eng, k_angle_bwd_blk<false>, angle_blk_lds<false>()))) return s;
  if ((s = set_lds(eng, k_angle_bwd_w<true>, angle_w_lds<true>()))) return s;
  if ((s = set_lds(eng, k_angle_bwd_w<false>, angle_w_lds<false>()))) return s;
  if ((s = set_lds(eng, (k_angle_bwd_w<true, true>), angle_w_lds<true>()))) return s;
  if ((s = set_lds(eng, (k_angle_bwd_w<true, false, true>), angle_w_lds<true>()))) return s;
  if ((s = set_lds(eng, (k_angle_bwd_w<true, true, true>), angle_w_lds<true>()))) return s;
  if ((s = set_lds(eng, (k_angle_bwd_w<false, true>), angle_w_lds<false>()))) return s;
  if ((s = set_lds(eng, k_angle<true, false>, angle_lds<true>()))) return s;
  if ((s = set_lds(eng, k_angle<true, true>, (angle_lds<true, WAVES, true>())))) return s;
  if ((s = set_lds(eng, k_angle<false, true>, (angle_lds<false, WAVES, true>())))) return s;
  if ((s = set_lds(eng, k_angle<false, false, FWD_WAVES>, (angle_lds<false, FWD_WAVES>())))) return s;
#ifdef CHG_EXPERIMENTS
  if ((s = set_lds(eng, (k_angle<true, false, 4>), (size_t)100 << 10))) return s;
  if ((s = set_lds(eng, (k_angle<false, false, 4>), (size_t)100 << 10))) return s;
  if ((s = set_lds(eng, k_atomconv_fwd<4>, (size_t)100 << 10))) return s;
#endif
  if ((s = set_lds(eng, k_readout<false>, readout_lds()))) return s;
  if ((s = set_lds(eng, (k_bond_embed_t<false, false, 1>), bond_embed_lds()))) return s;
  if ((s = set_lds(eng, (k_bond_embed_t<false, false, 2>), bond_embed_lds()))) return s;
  if ((s = set_lds(eng, (k_bond_embed_t<true, false, 1>), bond_embed_lds()))) return s;
  if ((s = set_lds(eng, (k_bond_embed_t<true, false, 2>), bond_embed_lds()))) return s;
  if ((s = set_lds(eng, k_angle_embed_t<false>, angle_embed_lds()))) return s;
  if ((s = set_lds(eng, k_angle_embed_fwd_o4<0>, angle_embed_lds()))) return s;
  if ((s = set_lds(eng, k_bond_embed_fwd_o4<1>, bond_embed_lds()))) return s;
  if ((s = set_lds(eng, k_angle_embed_t<true>, angle_embed_lds()))) return s;
  if ((s = set_lds(eng, k_rows_chain, chain_lds()))) return s;
  if ((s = set_lds(eng, k_embed_all<false>, bond_embed_lds()))) return s;
  if ((s = set_lds(eng, k_embed_all<true>, bond_embed_lds()))) return s;
  return CHG_OK;
}


}  // namespace chgh
