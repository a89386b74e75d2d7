// atomic_scope_lab.hip -- where do fp32 atomic adds run, and how fast?  (experiment, not product)
//   mode 0: agent-scope atomics, ONE copy of the target rows (what the adjoint kernels do)
//   mode 1: workgroup-scope atomics into a copy of the rows PRIVATE TO THE XCD the wave runs on (HW_REG_XCC_ID): every wave that
//           touches a copy shares its L2, so the L2 may keep the line; the copies are summed afterwards
//   mode 2: agent-scope atomics into the per-XCD copies (separates "fewer writers per line" from "scope")
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lab/atomic_scope_lab.hip -o gpurun_out/atomic_scope_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }   // HW_REG_XCC_ID[3:0]

template <int MODE>
__global__ __launch_bounds__(512) void k_add(float* __restrict__ rows, int n_rows, int row_floats, int iters, int* xcc_seen) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int xcc = xcc_id();
  if (threadIdx.x == 0) xcc_seen[blockIdx.x] = xcc;
  float* base = rows + (MODE == 0 ? 0 : (size_t)xcc * n_rows * row_floats);
  unsigned s = 0x9E3779B9u * (wave + 1);
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const int r = (s >> 8) % n_rows;
    float* p = base + (size_t)r * row_floats + lane;
    for (int q = 0; q < row_floats / 64; ++q) {
      if (MODE == 1) __hip_atomic_fetch_add(p + 64 * q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(p + 64 * q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void k_sum(const float* rows, size_t n, int copies, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n * copies; i += (size_t)gridDim.x * blockDim.x) s += rows[i];
  atomicAdd(out, s);
}

int main(int argc, char** argv) {
  const int n_rows = argc > 1 ? atoi(argv[1]) : 2048, row_floats = 256, iters = argc > 2 ? atoi(argv[2]) : 64, grid = 256;
  float* rows; int* seen; double* total;
  const size_t n = (size_t)n_rows * row_floats;
  CK(hipMalloc(&rows, 8 * n * sizeof(float))); CK(hipMalloc(&seen, grid * sizeof(int))); CK(hipMalloc(&total, sizeof(double)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    double sum = 0;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemset(rows, 0, 8 * n * sizeof(float))); CK(hipMemset(total, 0, sizeof(double)));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_add<0>, dim3(grid), dim3(512), 0, 0, rows, n_rows, row_floats, iters, seen);
      if (mode == 1) hipLaunchKernelGGL(k_add<1>, dim3(grid), dim3(512), 0, 0, rows, n_rows, row_floats, iters, seen);
      if (mode == 2) hipLaunchKernelGGL(k_add<2>, dim3(grid), dim3(512), 0, 0, rows, n_rows, row_floats, iters, seen);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      hipLaunchKernelGGL(k_sum, dim3(256), dim3(256), 0, 0, rows, n, 8, total);
      CK(hipMemcpy(&sum, total, sizeof(double), hipMemcpyDeviceToHost));
    }
    const double adds = (double)grid * 8 * iters * row_floats;
    std::vector<int> h(grid); CK(hipMemcpy(h.data(), seen, grid * sizeof(int), hipMemcpyDeviceToHost));
    int hist[8] = {0}; bool rr = true;
    for (int i = 0; i < grid; ++i) { hist[h[i] & 7]++; rr = rr && h[i] == i % 8; }
    printf("mode %d: rows %d x %d floats, %d row adds per wave: %.1f us, %.0f GB/s of atomic payload; sum %.0f (expected %.0f) %s; xcc hist %d %d %d %d %d %d %d %d, blockIdx%%8 == xcc: %s\n",
           mode, n_rows, row_floats, iters, 1e3 * best, adds * 4 / (best * 1e-3) / 1e9, sum, adds, sum == adds ? "EXACT" : "MISMATCH",
           hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], rr ? "yes" : "no");
  }
  return 0;
}
