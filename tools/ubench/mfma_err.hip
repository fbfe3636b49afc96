// Accuracy of the f16 MFMA accumulation the prefilter's error budget relies on: c + sum_k a_k b_k over
// K = 32 f16 products (2 chained v_mfma_f32_32x32x16_f16) against the same sum in f64, relative to
// |c| + sum_k |a_k b_k|.  The budget assumes <= 1 ulp (2^-23) per addition, i.e. <= 33 * 2^-23 = 3.9e-6.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
__device__ unsigned long long rng(unsigned long long x) {
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31);
}
__global__ void k(int iters, unsigned long long seed, double* out_max) {
  __shared__ _Float16 A[32][32], B[32][32];   // A[row][k], B[col][k]
  __shared__ float C[32];                      // per column
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  double worst = 0.0;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) {
      const unsigned long long r = rng(seed + ((unsigned long long)blockIdx.x * iters + it) * 4096 + i);
      // magnitudes spread over several binades, both signs (like residual terms that cancel)
      const float a = ldexpf((float)((r & 0xffff) / 65536.0 - 0.5), (int)((r >> 16) & 7) - 3);
      const float b = ldexpf((float)(((r >> 24) & 0xffff) / 65536.0 - 0.5), (int)((r >> 40) & 7) - 3);
      A[i >> 5][i & 31] = (_Float16)a; B[i >> 5][i & 31] = (_Float16)b;
      if (i < 32) C[i] = ldexpf((float)(((r >> 44) & 0xffff) / 65536.0 - 0.5), 2);
    }
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[col];
    for (int m = 0; m < 2; ++m) {
      f16x8 a, b;
      for (int j = 0; j < 8; ++j) { a[j] = A[col][16 * m + 8 * half + j]; b[j] = B[col][16 * m + 8 * half + j]; }
      // A operand: row = lane & 31 supplies A[row][k]; B operand: col = lane & 31 supplies B[k][col]
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      double ref = (double)C[col], mag = fabs((double)C[col]);
      for (int kk = 0; kk < 32; ++kk) {
        const double p = (double)A[row][kk] * (double)B[col][kk];
        ref += p; mag += fabs(p);
      }
      const double e = fabs((double)acc[r] - ref) / mag;
      worst = fmax(worst, e);
    }
  }
  for (int off = 32; off >= 1; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off));
  if (lane == 0) out_max[blockIdx.x] = worst;
}
int main() {
  const int blocks = 2048, iters = 200;
  double* d; (void)hipMalloc(&d, blocks * 8);
  k<<<blocks, 64>>>(iters, 12345ULL, d);
  std::vector<double> h(blocks);
  (void)hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  double w = 0; for (double v : h) w = fmax(w, v);
  printf("results checked: %.3g; worst |mfma - f64| / (|c| + sum |a b|) = %.3e = %.2f x 2^-23 (budget: 33 x 2^-23 = %.3e)\n",
         (double)blocks * iters * 1024, w, w / 1.1920929e-7, 33 * 1.1920929e-7);
  return 0;
}
