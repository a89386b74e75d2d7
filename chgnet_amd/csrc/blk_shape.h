// blk_shape.h -- the block shape of an atom's tiles in the blocked angle adjoints (kernels_angle_blk.h), shared by the graph builder
// (kernels_graph.h), the index of uploaded graphs (kernels_angle_w.h k_win_scan2, kernels_angle_blk.h k_blk_from_q) and the host.
#pragma once
#include <hip/hip_runtime.h>

namespace chg {

// The n x (n - 1) matrix (first bond rank, position of the second bond among the OTHER n - 1 bonds) of an atom is cut into 16-slot tiles
// of P x Q = 4 x 4, 2 x 8 or 8 x 2, whichever takes fewest (4 x 4 on a tie: it sends the fewest atomic rows).  ps = log2 P, qs = log2 Q;
// returns the tile count.
__host__ __device__ inline int blk_shape_of(int n, int& ps, int& qs) {
  if (n < 2) { ps = 2; qs = 2; return 0; }
  const int t44 = ((n + 3) >> 2) * ((n + 2) >> 2), t28 = ((n + 1) >> 1) * ((n + 6) >> 3), t82 = ((n + 7) >> 3) * (n >> 1);   // ceil(n / P) ceil((n - 1) / Q)
  if (t44 <= t28 && t44 <= t82) { ps = 2; qs = 2; return t44; }
  if (t28 <= t82) { ps = 1; qs = 3; return t28; }
  ps = 3; qs = 1; return t82;
}

}  // namespace chg
