// engine_predict_wide.hip -- the prediction sweep a SECOND time, with every forward operand of the split-precision contractions
// scaled per row by a power of two before the f16 split (mfma_split.h CHG_WIDE_RANGE: exact, any fp32 magnitude), in its own
// namespaces (chg_wide:: kernels, chgh_wide:: host code).  chg_batch_download re-runs a batch through chgh_wide::run_predict when the
// product sweep returns a non-finite energy / force / stress: where the reference's fp32 path (crystalgraph.py:12 TORCH_DTYPE) is
// finite -- an activation beyond the f16 range, |x| >= 65504 -- the re-run is finite too; a degenerate geometry stays NaN like the
// reference.  The product path keeps the unscaled operands: the row scaling costs its forward kernels ~10 % (vector instructions).
//
// The unit is engine_predict.hip itself, compiled under two renames; it shares the engine's weight images, arenas and streams.
#define CHG_WIDE_RANGE 1
#define chg chg_wide
#define chgh chgh_wide
#include "engine_predict.hip"

// The two helpers of engine_graph.hip this unit calls (prepare_windows' scans) were declared under the rename: forward them to the
// one definition in the real namespace.
#undef chgh
namespace chgh {
size_t scan_scratch_ints(int n);
int exclusive_scan_with(chg_engine* eng, int* scratch, const int* in, int* out, int n);
}  // namespace chgh
namespace chgh_wide {
size_t scan_scratch_ints(int n) { return chgh::scan_scratch_ints(n); }
int exclusive_scan_with(chg_engine* eng, int* scratch, const int* in, int* out, int n) { return chgh::exclusive_scan_with(eng, scratch, in, out, n); }
}  // namespace chgh_wide
