// cu_mask_lab.hip -- can two streams own disjoint sets of CUs (hipExtStreamCreateWithCUMask) so that an HBM-bound kernel runs NEXT TO a
// register-/LDS-bound one?  (experiment, not product)
//   k_spin: 1 workgroup per CU (160 KB LDS), spins for a fixed number of clocks, records (xcc, cu) it ran on
//   k_copy: float4 copy of 1 GiB
// hipcc --offload-arch=gfx950 -O3 tools/lab/cu_mask_lab.hip -o tools/lab/cu_mask_lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void k_spin(long clocks, int* where) {
  extern __shared__ float smem[];
  smem[threadIdx.x] = 0.f;
  const long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF;          // HW_REG_XCC_ID
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);                 // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    where[blockIdx.x] = (int)((xcc << 16) | (hw & 0xFFFF));
  }
  while (wall_clock64() - t0 < clocks) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("CUs %d\n", ncu);
  const size_t n = (size_t)1 << 26;   // float4: 1 GiB
  float4 *a, *b; int* where;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&where, 1024 * sizeof(int)));
  CK(hipMemset(a, 1, n * 16));
  CK(hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int reserve_per_word = 0; reserve_per_word <= 8; reserve_per_word += 4) {
    // mask words of 32 CUs: the copy stream gets the low `reserve_per_word` bits of every word, the spin stream the rest
    std::vector<uint32_t> m_spin(ncu / 32), m_copy(ncu / 32);
    for (int w = 0; w < ncu / 32; ++w) {
      const uint32_t low = reserve_per_word ? ((1u << reserve_per_word) - 1u) : 0u;
      m_copy[w] = reserve_per_word ? low : 0xFFFFFFFFu;
      m_spin[w] = reserve_per_word ? ~low : 0xFFFFFFFFu;
    }
    hipStream_t s_spin, s_copy;
    CK(hipExtStreamCreateWithCUMask(&s_spin, ncu / 32, m_spin.data()));
    CK(hipExtStreamCreateWithCUMask(&s_copy, ncu / 32, m_copy.data()));
    hipEvent_t e0, e1, c0, c1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
    const int spin_cus = ncu - reserve_per_word * (ncu / 32);
    // copy alone on its stream
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(c0, s_copy));
      hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s_copy, a, b, n);
      CK(hipEventRecord(c1, s_copy)); CK(hipEventSynchronize(c1));
    }
    float ms_alone; CK(hipEventElapsedTime(&ms_alone, c0, c1));
    // spin (2 ms) on its CUs with the copy next to it
    CK(hipMemset(where, 0xFF, 1024 * sizeof(int)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s_spin));
    hipLaunchKernelGGL(k_spin, dim3(spin_cus), dim3(512), 160 * 1024, s_spin, (long)200000, where);   // 100 MHz wall clock: 2 ms
    CK(hipEventRecord(e1, s_spin));
    CK(hipEventRecord(c0, s_copy));
    hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s_copy, a, b, n);
    CK(hipEventRecord(c1, s_copy));
    CK(hipDeviceSynchronize());
    float ms_spin, ms_copy; CK(hipEventElapsedTime(&ms_spin, e0, e1)); CK(hipEventElapsedTime(&ms_copy, c0, c1));
    std::vector<int> h(1024); CK(hipMemcpy(h.data(), where, 1024 * sizeof(int), hipMemcpyDeviceToHost));
    std::set<int> places; for (int i = 0; i < spin_cus; ++i) places.insert(h[i]);
    printf("copy stream: %d CUs per 32: copy alone %.3f ms (%.0f GB/s); spin on %d workgroups: %.3f ms on %zu distinct (xcc, se, sh, cu); copy next to it %.3f ms (%.0f GB/s)\n",
           reserve_per_word ? reserve_per_word : 32, ms_alone, 2.0 * n * 16 / ms_alone / 1e6, spin_cus, ms_spin, places.size(), ms_copy, 2.0 * n * 16 / ms_copy / 1e6);
    CK(hipStreamDestroy(s_spin)); CK(hipStreamDestroy(s_copy));
  }
  return 0;
}
