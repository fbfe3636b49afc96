// How much VALU work hides under v_mfma_f32_32x32x2_f32?  One dependent MFMA chain per wave with V extra
// independent VALU instructions (v_cndmask / v_add_u32 / 64-bit v_mad_u64_u32) per MFMA, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int V, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a = threadIdx.x * 0.001f, b = 0.5f + threadIdx.x * 0.002f;
  f32x16 c0;
  for (int i = 0; i < 16; ++i) c0[i] = 0.f;
  unsigned x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * (i + 3);
  unsigned long long y = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(x[(v + 1) & 7]));
        if (KIND == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[v & 7]) : "v"(x[(v + 1) & 7]));
        if (KIND == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y) : "v"(x[v & 7]), "v"(x[(v + 1) & 7]) : "vcc");
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7] + (float)y;
}
template <int V, int KIND>
void run(const char* name, int blocks) {
  float* out;
  (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<V, KIND><<<blocks, 256>>>(out, 50);
  (void)hipEventRecord(e0);
  k<V, KIND><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = blocks / 256.0;
  const double n = iters * 16.0 * per_simd;
  printf("%-16s V=%2d waves/SIMD %.0f  %6.1f ns per MFMA per SIMD (26.7 = peak) -> %6.1f TF\n", name, V, per_simd,
         ms * 1e6 / n, n * 1024 * 4096.0 / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 1024}) {
    run<0, 0>("none", blocks);
    run<2, 0>("v_add_u32", blocks);
    run<4, 0>("v_add_u32", blocks);
    run<8, 0>("v_add_u32", blocks);
    run<16, 0>("v_add_u32", blocks);
    run<4, 1>("v_cndmask", blocks);
    run<8, 1>("v_cndmask", blocks);
    run<4, 2>("v_mad_u64_u32", blocks);
    run<8, 2>("v_mad_u64_u32", blocks);
  }
  return 0;
}
