// v_mfma_f64_16x16x4_f64 issue rate: independent and dependent chains, 1..4 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
using f64x4 = __attribute__((ext_vector_type(4))) double;
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double a = threadIdx.x * 0.001, b = 0.5 + threadIdx.x * 0.002;
  f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // one dependent chain of 4
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    } else {  // four independent accumulators
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int MODE>
void run(const char* name, int blocks) {
  double* out;
  (void)hipMalloc(&out, 4096 * 256 * 8);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100);
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = blocks / 256.0;
  printf("%-28s waves/SIMD %.0f  %8.3f ms  %.1f ns per MFMA per SIMD (64 cycles @2.4GHz = 26.7 ns)\n", name, per_simd,
         ms, ms * 1e6 / (iters * 4.0 * per_simd));
  (void)hipFree(out);
}
int main() {
  run<0>("dependent chain", 256);
  run<1>("independent x4", 256);
  run<0>("dependent chain", 512);
  run<1>("independent x4", 512);
  run<0>("dependent chain", 1024);
  run<1>("independent x4", 1024);
  return 0;
}
