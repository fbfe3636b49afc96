// v_mfma_f32_32x32x2_f32 issue rate as the sparse convolution uses it: dependent accumulator chains (1 or 2
// interleaved), operands from registers or re-read from LDS per MFMA, with or without a workgroup barrier every
// 16 MFMAs, 1..4 waves per SIMD.  `hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form mfma_f32.hip`
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
// MODE bit0: two chains; bit1: operands from LDS; bit2: barrier per 16 MFMAs
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float a = threadIdx.x * 0.001f, b = 0.5f + threadIdx.x * 0.002f;
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE & 2) {
        a = lds[(it & 7) * 1024 + j * 64 + lane];
        b = lds[((it + 3) & 7) * 1024 + j * 64 + lane];
      }
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      if (MODE & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
    }
    if (MODE & 4) __syncthreads();
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
}
template <int MODE>
void run(const char* name, int blocks) {
  float* out;
  (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 50);
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = blocks / 256.0;
  const double n = iters * 16.0 * ((MODE & 1) ? 2 : 1) * per_simd;
  printf("%-44s waves/SIMD %.0f  %8.3f ms  %6.1f ns per MFMA per SIMD  -> %6.1f TF\n", name, per_simd, ms,
         ms * 1e6 / n, n * 1024 * 4096.0 / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 1024}) {
    run<0>("1 chain, regs", blocks);
    run<1>("2 chains, regs", blocks);
    run<2>("1 chain, LDS operands", blocks);
    run<3>("2 chains, LDS operands", blocks);
    run<6>("1 chain, LDS operands, barrier/16", blocks);
    run<7>("2 chains, LDS operands, barrier/32", blocks);
  }
  return 0;
}
