// engine_train_wide.hip -- the fine-tuning sweeps (first order and second order) a SECOND time with every operand row of the split
// contractions scaled by a power of two before the f16 split (mfma_split.h CHG_WIDE_RANGE), next to engine_predict_wide.hip: a batch
// whose activations leave the f16 range is moved to the wide-range prediction sweep by chg_batch_download, and chg_backward then takes
// its parameter gradients from here (chgh::backward_compute forwards) -- the reference's fp32 autograd (trainer.py:399-411) has no
// such range either.  The unit is engine_train.hip itself under the two renames, without its C entry points.
#define CHG_WIDE_RANGE 1
#define chg chg_wide
#define chgh chgh_wide
#include "engine_train.hip"
