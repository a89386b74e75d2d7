// split_lab.hip -- hardware checks behind csrc/mfma_split.h (build: tools/build_split_lab.sh, run on the GPU box).
//  T1  v_mfma_f32_16x16x32_f16 pairs element e of lane group g of A with the same (g, e) of B
//  T2  what the matrix pipe does with f16 subnormal inputs
//  T3  gemm_split (normal and transposed image, plain and row-scaled) against float64 on the host
//  T4  matrix-pipe time of the split contraction against the f32 form (same tile, same LDS traffic pattern)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mfma_split.h"

using namespace chg;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_pair(const _Float16* a, const _Float16* b, float* d) {   // a, b: [64 lanes][8]
  const int l = threadIdx.x;
  h16x8 av, bv;
  for (int e = 0; e < 8; ++e) { av[e] = a[l * 8 + e]; bv[e] = b[l * 8 + e]; }
  f32x4 acc = zero4();
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

// X [rows][K] row-major, W [F][K]; Y [rows][F] (normal) or, TRANSPOSE, W [K][F]... see host
template <int K, int F, bool TRANSPOSE, bool SCALED, bool SPLIT>
__global__ __launch_bounds__(512) void k_gemm(const float* X, const float* W, float* Y, int rows, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  h16x8* img = reinterpret_cast<h16x8*>(smem_raw);
  float* Wf = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  // W is given as [Fw][Kw] = TRANSPOSE ? [K][F] : [F][K]
  if (SPLIT) stage_split<TRANSPOSE>(img, W, TRANSPOSE ? K : F, TRANSPOSE ? F : K, tid, 512);
  else stage_weights(Wf, W, TRANSPOSE ? K : F, TRANSPOSE ? F : K, tid);
  __syncthreads();
  constexpr int KT = K / 16, NFT = F / 16;
  for (int row0 = (blockIdx.x * 8 + wave) * 16; row0 < rows; row0 += gridDim.x * 128) {
    f32x4 x[KT];
    read_dl<KT>(X + (size_t)(row0 + j) * K, g, x);
    f32x4 acc[NFT];
    for (int fo = 0; fo < NFT; ++fo) acc[fo] = zero4();
    for (int rep = 0; rep < reps; ++rep) {
      if (SPLIT) gemm_split<KT, NFT, SCALED>(acc, img, F, x, j, g);
      else if (TRANSPOSE) gemm_dl_t<KT, NFT>(acc, Wf, F + PAD, x, j, g);
      else gemm_dl<KT, NFT>(acc, Wf, K + PAD, x, j, g);
      if (reps > 1) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) x[kt][0] += 1e-9f * acc[kt % NFT][0];   // keep the repetitions dependent
      }
    }
    write_dl<NFT>(Y + (size_t)(row0 + j) * F, g, acc);
  }
}

// T6: gemm_rm (one row-major image, both directions) against float64
template <int K, int F, bool ADJOINT, bool SCALED>
__global__ __launch_bounds__(512) void k_gemm_rm(const float* X, const float* W, float* Y, int rows, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* img = reinterpret_cast<_Float16*>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  stage_rm(img, W, F, K, tid, 512);                   // W is always [F][K]
  __syncthreads();
  constexpr int NIN = ADJOINT ? F : K, NOUT = ADJOINT ? K : F;
  constexpr int KT = NIN / 16, NOT = NOUT / 16;
  for (int row0 = (blockIdx.x * 8 + wave) * 16; row0 < rows; row0 += gridDim.x * 128) {
    f32x4 x[KT];
    read_dl<KT>(X + (size_t)(row0 + j) * NIN, g, x);
    f32x4 acc[NOT];
    for (int o = 0; o < NOT; ++o) acc[o] = zero4();
    for (int rep = 0; rep < reps; ++rep) {
      gemm_rm<KT, NOT, SCALED, ADJOINT>(acc, img, F, K, x, j, g, lane);
      if (reps > 1) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) x[kt][0] += 1e-9f * acc[kt % NOT][0];
      }
    }
    write_dl<NOT>(Y + (size_t)(row0 + j) * NOUT, g, acc);
  }
}

template <int K, int F, bool ADJOINT, bool SCALED>
static void check_rm(const char* label, float xscale) {
  const int rows = 256;
  constexpr int NIN = ADJOINT ? F : K, NOUT = ADJOINT ? K : F;
  std::vector<float> X(rows * NIN), W(F * K), Y(rows * NOUT);
  for (auto& v : X) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * xscale * (rand() % 7 == 0 ? 1e-3f : 1.f);
  for (auto& v : W) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.3f;
  float *dX, *dW, *dY;
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dY, Y.size() * 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  k_gemm_rm<K, F, ADJOINT, SCALED><<<2, 512, rm_image_bytes(F, K)>>>(dX, dW, dY, rows, 1);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
  double worst_rel = 0;
  for (int r = 0; r < rows; ++r) {
    double rowmax = 0;
    std::vector<double> ref(NOUT);
    for (int o = 0; o < NOUT; ++o) {
      double sdot = 0, sa = 0;
      for (int c = 0; c < NIN; ++c) {
        const double w = ADJOINT ? W[c * K + o] : W[o * K + c];
        sdot += w * X[r * NIN + c]; sa += std::fabs(w * X[r * NIN + c]);
      }
      ref[o] = sdot; rowmax = std::fmax(rowmax, sa);
    }
    for (int o = 0; o < NOUT; ++o) worst_rel = std::fmax(worst_rel, std::fabs(Y[r * NOUT + o] - ref[o]) / (rowmax + 1e-300));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int trows = 256 * 128 * 4, reps = 64;
  float *tX, *tY;
  CK(hipMalloc(&tX, (size_t)trows * NIN * 4)); CK(hipMalloc(&tY, (size_t)trows * NOUT * 4)); CK(hipMemset(tX, 0, (size_t)trows * NIN * 4));
  const size_t lds = rm_image_bytes(F, K) + 64 * 1024;
  CK(hipFuncSetAttribute((const void*)k_gemm_rm<K, F, ADJOINT, SCALED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float best = 1e9;
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0));
    k_gemm_rm<K, F, ADJOINT, SCALED><<<256, 512, lds>>>(tX, dW, tY, trows, reps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::fmin(best, ms);
  }
  printf("T6 row-major image %-26s max err / sum|w x| %.3e   %.1f TFLOP/s (f32-equivalent)\n", label, worst_rel, 2.0 * trows * K * F * reps / best * 1e-9);
  hipFree(dX); hipFree(dW); hipFree(dY); hipFree(tX); hipFree(tY);
}

// T5: ds_read_b64_tr_b16.  LDS holds a row-major f16 matrix M[64][72]; in every 16-lane group, lanes 4 r + q point at the four
// columns 16 + 4 q .. + 3 of row f0 + r (f0 = 4 * group).  Hypothesis: lane c of the group receives M[f0 + j][16 + c], j = 0..3.
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k_tr(float* out) {
  __shared__ _Float16 M[64 * 72];
  for (int i = threadIdx.x; i < 64 * 72; i += 64) M[i] = (_Float16)(float)((i / 72) * 32 + (i % 72) % 32);
  __syncthreads();
  const int l = threadIdx.x, q = l & 15;
  const _Float16* p = M + (4 * (l >> 4) + (q >> 2)) * 72 + 16 + 4 * (q & 3);
  const auto v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

template <int K, int F, bool TRANSPOSE, bool SCALED>
static void check(const char* label, float xscale) {
  const int rows = 256;
  std::vector<float> X(rows * K), W(F * K), Y(rows * F);
  for (auto& v : X) v = frand() * xscale * (rand() % 7 == 0 ? 1e-3f : 1.f);
  for (auto& v : W) v = frand() * 0.3f;
  float *dX, *dW, *dY;
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dY, Y.size() * 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  for (int split = 0; split < 2; ++split) {
    size_t lds = split ? split_image_bytes(F, K) : sizeof(float) * (TRANSPOSE ? K * (F + PAD) : F * (K + PAD));
    if (split) k_gemm<K, F, TRANSPOSE, SCALED, true><<<2, 512, lds>>>(dX, dW, dY, rows, 1);
    else k_gemm<K, F, TRANSPOSE, SCALED, false><<<2, 512, lds>>>(dX, dW, dY, rows, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst_rel = 0;
    for (int r = 0; r < rows; ++r) {
      double rowmax = 0;
      std::vector<double> ref(F);
      for (int f = 0; f < F; ++f) {
        double s = 0, sa = 0;
        for (int k = 0; k < K; ++k) {
          const double w = TRANSPOSE ? W[k * F + f] : W[f * K + k];
          s += w * X[r * K + k]; sa += std::fabs(w * X[r * K + k]);
        }
        ref[f] = s; rowmax = std::fmax(rowmax, sa);
      }
      for (int f = 0; f < F; ++f) {
        const double err = std::fabs(Y[r * F + f] - ref[f]);
        worst = std::fmax(worst, err); worst_rel = std::fmax(worst_rel, err / (rowmax + 1e-300));
      }
    }
    printf("T3 %-28s %s  max|err| %.3e   max err / sum|w x| %.3e\n", label, split ? "split f16x3" : "f32 mfma   ", worst, worst_rel);
  }
  hipFree(dX); hipFree(dW); hipFree(dY);
}

template <int K, int F, bool TRANSPOSE, bool SCALED>
static void timeit(const char* label) {
  const int rows = 256 * 128 * 4, reps = 64;
  float *dX, *dW, *dY;
  CK(hipMalloc(&dX, (size_t)rows * K * 4)); CK(hipMalloc(&dW, F * K * 4)); CK(hipMalloc(&dY, (size_t)rows * F * 4));
  CK(hipMemset(dX, 0, (size_t)rows * K * 4)); CK(hipMemset(dW, 0, F * K * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int split = 0; split < 2; ++split) {
    size_t lds = split ? split_image_bytes(F, K) : sizeof(float) * (TRANSPOSE ? K * (F + PAD) : F * (K + PAD));
    lds += 64 * 1024;    // limit occupancy to one 8-wave workgroup per CU like the tile kernels
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(e0));
      if (split) { CK(hipFuncSetAttribute((const void*)k_gemm<K, F, TRANSPOSE, SCALED, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_gemm<K, F, TRANSPOSE, SCALED, true><<<256, 512, lds>>>(dX, dW, dY, rows, reps); }
      else { CK(hipFuncSetAttribute((const void*)k_gemm<K, F, TRANSPOSE, SCALED, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_gemm<K, F, TRANSPOSE, SCALED, false><<<256, 512, lds>>>(dX, dW, dY, rows, reps); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::fmin(best, ms);
    }
    const double flop = 2.0 * rows * K * F * reps;
    printf("T4 %-28s %s  %.3f ms  %.1f TFLOP/s (f32-equivalent)\n", label, split ? "split f16x3" : "f32 mfma   ", best, flop / best * 1e-9);
  }
  hipFree(dX); hipFree(dW); hipFree(dY);
}

int main() {
  srand(1);
  {  // T1 + T2
    std::vector<_Float16> a(512), b(512);
    std::vector<float> d(256);
    _Float16 *da, *db; float* dd;
    CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dd, 1024));
    for (auto& v : a) v = (_Float16)frand();
    for (auto& v : b) v = (_Float16)frand();
    CK(hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice));
    k_pair<<<1, 64>>>(da, db, dd); CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {       // D[row = 4 (l>>4) + r][col = l & 15]
      const int i = 4 * (l >> 4) + r, jj = l & 15;
      double s = 0;
      for (int g = 0; g < 4; ++g) for (int e = 0; e < 8; ++e) s += (double)a[(i + 16 * g) * 8 + e] * (double)b[(jj + 16 * g) * 8 + e];
      worst = std::fmax(worst, std::fabs(s - d[l * 4 + r]));
    }
    printf("T1 pairing (g, e) of A with (g, e) of B: max|err| %.3e  %s\n", worst, worst < 1e-5 ? "OK" : "MISMATCH");
    // subnormal inputs: A = 2^-20 everywhere (f16 subnormal), B = 1 -> 32 * 2^-20 if kept, 0 if flushed
    for (auto& v : a) v = (_Float16)9.5367431640625e-07f;
    for (auto& v : b) v = (_Float16)1.0f;
    CK(hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice));
    k_pair<<<1, 64>>>(da, db, dd); CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    printf("T2 f16 subnormal A (2^-20) x 1.0, K = 32: D = %.6e (kept: %.6e, flushed: 0)\n", d[0], 32 * 9.5367431640625e-07);
    for (auto& v : a) v = (_Float16)9.5367431640625e-07f;
    for (auto& v : b) v = (_Float16)9.5367431640625e-07f;
    CK(hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice));
    k_pair<<<1, 64>>>(da, db, dd); CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    printf("T2 subnormal x subnormal: D = %.6e (exact: %.6e)\n", d[0], 32 * 9.5367431640625e-07 * 9.5367431640625e-07);
  }
  {
    float* dd; std::vector<float> d(256);
    CK(hipMalloc(&dd, 1024));
    k_tr<<<1, 64>>>(dd); CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      const int row = 4 * (l >> 4) + j, col = 16 + (l & 15);
      if (d[l * 4 + j] != (float)(row * 32 + col % 32)) ++bad;
    }
    printf("T5 ds_read_b64_tr_b16: lane c of a 16-lane group gets M[f0 + j][c0 + c] (rows from lanes 4 r + q): %s (%d mismatches; lane 5 got %g %g %g %g)\n",
           bad ? "NO" : "yes", bad, d[20], d[21], d[22], d[23]);
    hipFree(dd);
  }
  check_rm<64, 64, false, false>("64->64 forward", 1.f);
  check_rm<64, 128, false, false>("64->128 forward", 3.f);
  check_rm<64, 64, true, true>("64<-64 adjoint, scaled", 1e-4f);
  check_rm<64, 128, true, true>("64<-128 adjoint, scaled", 1e-6f);
  check<64, 64, false, false>("64->64 forward", 1.f);
  check<64, 128, false, false>("64->128 forward", 3.f);
  check<64, 64, true, true>("64->64 transposed, scaled", 1e-4f);
  check<128, 64, true, true>("128->64 transposed, scaled", 1e-6f);
  check<64, 64, true, false>("64->64 transposed, unscaled", 1e-4f);
  timeit<64, 64, false, false>("64->64 forward");
  timeit<64, 128, false, false>("64->128 forward");
  timeit<64, 64, true, true>("64->64 transposed, scaled");
  timeit<128, 64, true, true>("128->64 transposed, scaled");
  return 0;
}
