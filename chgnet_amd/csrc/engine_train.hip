// engine_train.hip -- fine-tuning backward (reference: loss.backward() through CHGNet.forward, trainer/trainer.py:399-411,
// model.py:427-542 and the create_graph=True forces / stress of model.py:517-535): stage A (first order), stage B (second order),
// chg_backward / chg_backward_allreduce.
#include "engine_internal.h"

#include "kernels_geom.h"
#include "kernels_train.h"
#include "kernels_train2.h"
#include "kernels_train2_tile.h"
#include "kernels_train2_freq.h"

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

namespace chgh {

__global__ void k_gather_rows(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int rows) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * (D / 4)) return;
  const int r = t / (D / 4), q = t % (D / 4);
  reinterpret_cast<f32x4*>(dst)[t] = reinterpret_cast<const f32x4*>(src + (size_t)idx[r] * D)[q];
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

// mlp_out biases (0.2.0): bonds outside the bond graph enter AtomConv l as embedding + q_shift (the earlier BondConv biases), so
// dL/dW_bond += (sum over those bonds of dQ) x q_shift.  dQ: [Eu,128] table adjoint (dE/dQ, or bar(Q) of the second-order sweep).
int bond_shift_wgrad(chg_engine* eng, chg_batch* b, const float* dQ, const float* q_shift, float* g_w_bond) {
  if (b->Eu <= 0) return CHG_OK;
  TRY(zero(eng, b->t_tmp, sizeof(float) * 128));
  LaunchScope ls(eng, "wgrad_shift");
  hipLaunchKernelGGL(k_colsum_nonnode, dim3(std::max(1, std::min((b->Eu + 127) / 128, 2 * eng->num_cus))), dim3(256), 0, eng->stream, dQ, 2 * D,
                     b->u_bnode, b->Eu, b->t_tmp);
  hipLaunchKernelGGL(k_outer_add, dim3(32), dim3(256), 0, eng->stream, g_w_bond, b->t_tmp, q_shift);
  HIP_TRY(eng, hipGetLastError());
  return CHG_OK;
}

// Training workspaces are taken from / returned to the engine: a train step makes a new batch every iteration, and a hipMalloc
// of tens of GB per step would dominate it.  One slot per kind (0: first-order workspace, 1: second-order workspace); a request
// is rounded up by 8 % so that the slightly different batches of an epoch reuse the same block.
char* acquire_workspace(chg_engine* eng, size_t total, size_t& got, int kind) {
  std::lock_guard<std::mutex> lk(eng->pool_mu);   // acquire_arena (loader thread) frees work_pool under the same lock when memory is short
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
  if (!eng) { hipFree(p); return; }
  bool have = false;
  {
    std::lock_guard<std::mutex> lk(eng->pool_mu);
    for (int k : eng->work_kind) have = have || k == kind;
    if (!have) { eng->work_pool.emplace_back(p, bytes); eng->work_kind.push_back(kind); }
  }
  if (have) hipFree(p);   // outside the lock: a multi-GB hipFree takes milliseconds
}

int ensure_train_buffers(chg_engine* eng, chg_batch* b) {
  if (b->train_arena) return CHG_OK;
  const size_t N = b->N, Ed = b->Ed, Eu = b->Eu, A = b->A, rows = std::max(Ed, A);
  Carver c{nullptr};
  auto lay = [&](Carver& cv) {
    b->t_grad = cv.take<float>((size_t)eng->desc.n_weights);
    b->t_cot = cv.take<float>(b->B);
    b->t_tmp = cv.take<float>(256);
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
  const bool bias = eng->desc.mlp_out_bias != 0;   // 0.2.0: mlp_out biases are parameters
  TRY(zero(eng, b->t_grad, sizeof(float) * (size_t)eng->desc.n_weights));
  // group 2 without the energies / magmoms of the prediction in its middle (they stay readable: chg_batch_all_gather_energy, download)
  TRY(zero(eng, b->zero2, (size_t)((char*)b->energy - (char*)b->zero2)));
  TRY(zero(eng, b->zero2_keep_end, (size_t)((char*)b->zero2_end - (char*)b->zero2_keep_end)));

  // ---- readout: dE/d atom[L] from the cotangent; per-atom operands of the MLP / LayerNorm gradients ----
  {
    ReadoutArgs r{};
    r.atom = b->atom[L]; r.atom_owner = b->atom_owner; r.z = b->z; r.n_atoms = N;
    r.ln_g = w.ro_ln_g; r.ln_b = w.ro_ln_b; r.w0 = w.mlp_w0; r.b0 = w.mlp_b0; r.w1 = w.mlp_w1; r.b1 = w.mlp_b1;
    r.w2 = w.mlp_w2; r.b2 = w.mlp_b2; r.w3 = w.mlp_w3; r.b3 = w.mlp_b3; r.atomref = w.atomref;
    r.has_composition = eng->desc.has_composition; r.n_hidden = eng->desc.n_mlp_hidden;
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
    if (eng->desc.n_mlp_hidden == 3)
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
    if (bias && A > 0 && l > 0) TRY(bond_shift_wgrad(eng, b, b->GQ, aw.q_shift, G(aw.w_bond)));
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
      // (the reference adds b_out to EVERY bond, layers.py:252-258: its gradient is the column sum of the running dE/d bond over all Eu rows)
      if (bias) TRY(colsum(eng, b->Gb, D, nullptr, 0, Eu, D, G(bw.b_out)));
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
    {
      std::lock_guard<std::mutex> lk(eng->pool_mu);
      for (auto& a : eng->work_pool) free_b += a.second;
    }
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
    a.H = t.scratch6[2]; a.Hd = t.scratch6[3]; a.BCG = t.BCG; a.GCG = t.GCG; a.park0 = t.BZ; a.park1 = t.GZ;
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
    const int nh = eng->desc.n_mlp_hidden;   // 2 or 3 hidden layers
    for (int i = 0; i < nh; ++i) {
      TRY(gemm("t2_readout", 64, 64, t.ro[sidx[i]], D, nullptr, Wm[i], bm[i], nullptr, 0, t.ro[lidx[i]], D, nullptr, N, 0));
      TRY(gemm("t2_readout", 64, 64, t.ro[sdidx[i]], D, nullptr, Wm[i], nullptr, nullptr, 0, t.ro[ldidx[i]], D, nullptr, N, 0));
      LaunchScope ls(eng, "t2_readout");
      hipLaunchKernelGGL(k2_silu_t, g1((int64_t)nd), dim3(256), 0, st, t.ro[lidx[i]], t.ro[ldidx[i]], t.ro[sidx[i + 1]], t.ro[sdidx[i + 1]], nd);
    }
    { LaunchScope ls(eng, "t2_readout");
      hipLaunchKernelGGL(k2_readout_seed, g1((int64_t)nd), dim3(256), 0, st, w.mlp_w3, b->t_cot, b->atom_owner, t.ro[sidx[nh]], t.ro[sdidx[nh]], t.ro[BS],
                         t.ro[GS], t.ro[DW3], N); }
    TRY(colsum(eng, t.ro[DW3], D, nullptr, 0, N, D, G(w.mlp_w3)));
    const float* gW[3] = {G(w.mlp_w0), G(w.mlp_w1), G(w.mlp_w2)};
    const float* gb[3] = {G(w.mlp_b0), G(w.mlp_b1), G(w.mlp_b2)};
    for (int i = nh - 1; i >= 0; --i) {
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
    if (eng->desc.mlp_out_bias && angles && l > 0) {   // constant shift of the bonds outside the bond graph: no tangent, bar(Q) only
      TRY(bond_shift_wgrad(eng, b, t.barQ, aw.q_shift, G(aw.w_bond)));
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
      // hbc[l+1] = aggB . Wout^T + b_out + hbc[l]; its adjoints live in the node rows of bar_b / g_b; b_out reaches every bond (above)
      if (eng->desc.mlp_out_bias) TRY(colsum(eng, t.bar_b, D, nullptr, 0, Eu, D, G(bw.b_out)));
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
    // frequency gradients on 16-row tiles (kernels_train2_freq.h; CHGNET_T2_FREQ_ROWS=1 keeps the one-row-per-wave kernels for A/B runs)
    static const bool freq_rows = [] { const char* e = std::getenv("CHGNET_T2_FREQ_ROWS"); return e && std::atoi(e) != 0; }();
    {
      LaunchScope ls(eng, "t2_freq");
      if (freq_rows) {
        FreqGradArgs a{Eu, nullptr, b->ev, t.vd4, b->u_u2d, w.freq_ag, eng->desc.atom_graph_cutoff, env, t.bar_b, t.g_b, w.w_bond_emb,
                       t.bar_wag, g_wag, w.w_wag, G(w.freq_ag)};
        hipLaunchKernelGGL(k2_freq_grad, wave_rows_grid(eng, Eu), dim3(256), 0, st, a);
      } else {
        FreqGradTArgs a{Eu, nullptr, b->ev, t.vd4, b->u_u2d, w.freq_ag, eng->desc.atom_graph_cutoff, env, t.bar_b, t.g_b, w.w_bond_emb,
                        t.bar_wag, g_wag, w.w_wag, G(w.freq_ag)};
        hipLaunchKernelGGL(k2_freq_grad_t, dim3(grid_for(Eu, 2 * eng->num_cus)), dim3(BLOCK), freq_grad_lds(), st, a);
      }
    }
    if (Eb > 0) {
      LaunchScope ls(eng, "t2_freq");
      if (freq_rows) {
        FreqGradArgs a{Eb, b->bn_und, b->ev, t.vd4, b->u_u2d, w.freq_bg, eng->desc.bond_graph_cutoff, env, t.bar_wbg, g_wbg, w.w_wbg,
                       nullptr, nullptr, nullptr, G(w.freq_bg)};
        hipLaunchKernelGGL(k2_freq_grad, wave_rows_grid(eng, Eb), dim3(256), 0, st, a);
      } else {
        FreqGradTArgs a{Eb, b->bn_und, b->ev, t.vd4, b->u_u2d, w.freq_bg, eng->desc.bond_graph_cutoff, env, t.bar_wbg, g_wbg, w.w_wbg,
                        nullptr, nullptr, nullptr, G(w.freq_bg)};
        hipLaunchKernelGGL(k2_freq_grad_t, dim3(grid_for(Eb, 2 * eng->num_cus)), dim3(BLOCK), freq_grad_lds(), st, a);
      }
    }
  }
  if (angles) {
    TRY((xty<4, 2>(eng, "t2_wgrad", t.bar_ang, D, nullptr, t.X4, KB2, nullptr, A, 1.0f, G(w.w_ang_emb), NANG, NANG)));
    TRY((xty<4, 2>(eng, "t2_wgrad", t.g_ang, D, nullptr, t.X4d, KB2, nullptr, A, 1.0f, G(w.w_ang_emb), NANG, NANG)));
    LaunchScope ls(eng, "t2_freq");
    static const bool freq_rows_a = [] { const char* e = std::getenv("CHGNET_T2_FREQ_ROWS"); return e && std::atoi(e) != 0; }();
    if (freq_rows_a)
      hipLaunchKernelGGL(k2_angle_freq_grad, wave_rows_grid(eng, A), dim3(256), 0, st, t.bar_ang, t.g_ang, w.w_ang_emb, t.th2, w.freq_ang,
                         G(w.freq_ang), A);
    else
      hipLaunchKernelGGL(k2_angle_freq_grad_t, dim3(grid_for(A, 2 * eng->num_cus)), dim3(BLOCK), angle_freq_grad_lds(), st, t.bar_ang, t.g_ang,
                         w.w_ang_emb, t.th2, w.freq_ang, G(w.freq_ang), A);
  }
  {
    LaunchScope ls(eng, "wgrad_atom_embed");
    hipLaunchKernelGGL(k_embed_grad, g1((int64_t)N * D), dim3(256), 0, st, t.bar_a, b->z, G(w.emb), N);
  }
  return check();
}


int train_set_lds(chg_engine* eng) {
  int s;
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
  if ((s = set_lds(eng, k2_freq_grad_t, freq_grad_lds()))) return s;
  if ((s = set_lds(eng, k2_angle_freq_grad_t, angle_freq_grad_lds()))) return s;
  return CHG_OK;
}

// the local part of chg_backward: everything up to the gradient blob in HBM (b->t_grad), nothing leaves the device
int backward_compute(chg_engine* eng, chg_batch* b, const float* energy_cotangent, const float* magmom_cotangent,
                            const float* force_cotangent, const float* stress_cotangent) {
  if (b->last_task == 0) { eng->err = "chg_backward: run chg_predict on this batch first (the reverse sweep reuses its activations)"; return CHG_EINVAL; }
  if ((int)b->h_atom_off.size() != b->B + 1) { eng->err = "chg_backward: batch has no host atom offsets"; return CHG_EINVAL; }
#ifndef CHG_WIDE_RANGE
  // a batch that left the f16 operand range of the split contractions (chg_batch_download moved it to the wide-range prediction
  // sweep) gets its gradients from the same sweeps compiled with row-scaled operands: engine_train_wide.hip
  if (b->wide_range) return chgh_wide::backward_compute(eng, b, energy_cotangent, magmom_cotangent, force_cotangent, stress_cotangent);
#endif
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
  // (an energy-only prediction does not: build them then; 4 row GEMMs over all bonds, 1.6 ms per 1024 structures)
  if (b->Ed > 0 && !b->q_tables) {
    for (int l = 0; l < b->L; ++l) TRY(atomconv_q_table(eng, b, l));
    b->q_tables = true;
  }
  TRY(second_order ? run_backward2(eng, b) : run_backward(eng, b));
  // the b3 slot joins the blob ON THE DEVICE, so that a following all-reduce sums it over the ranks like every other entry (it used
  // to be written into the host copy after the collective: every rank then applied its LOCAL value / world -- ADVICE r03)
  HIP_TRY(eng, hipMemcpyAsync(b->t_grad + (eng->w.mlp_b3 - eng->d_weights), &b->h_g_b3, sizeof(float), hipMemcpyHostToDevice, eng->stream));
  return CHG_OK;
}

int backward_impl(chg_engine* eng, chg_batch* b, const float* energy_cotangent, const float* magmom_cotangent,
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

}  // namespace chgh

#ifndef CHG_WIDE_RANGE   // (engine_train_wide.hip compiles this unit a second time for the sweeps only)
extern "C" {

int chg_backward(chg_engine* eng, chg_batch* b, const float* energy_cotangent, const float* magmom_cotangent,
                 const float* force_cotangent, const float* stress_cotangent, float* grad_blob) {
  return backward_impl(eng, b, energy_cotangent, magmom_cotangent, force_cotangent, stress_cotangent, nullptr, grad_blob);
}

int chg_backward_allreduce(chg_engine* eng, chg_batch* b, const float* energy_cotangent, const float* magmom_cotangent,
                           const float* force_cotangent, const float* stress_cotangent, chg_comm* comm, float* grad_blob) {
  return backward_impl(eng, b, energy_cotangent, magmom_cotangent, force_cotangent, stress_cotangent, comm, grad_blob);
}

}  // extern "C"
#endif
