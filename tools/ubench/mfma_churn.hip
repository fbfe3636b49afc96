// Does the matrix pipe stay busy when the grid is MANY SHORT workgroups (the sparse convolution: 37 k waves of ~300
// dependent v_mfma_f32_32x32x2_f32 each, then 16 row stores per lane)?  Sweeps MFMAs per wave, threads per workgroup,
// LDS per workgroup (occupancy) and whether the epilogue stores.  `hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form`
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int THREADS, int LDS_KB, int STORE, int SALU, int PRO = 0, int PRIO = 0>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters, unsigned mask, const int* __restrict__ tab = nullptr, float abias = 0.f, unsigned long long* clk = nullptr) {
  __shared__ float lds[LDS_KB * 256];
  if (iters < 0) lds[threadIdx.x] = 1.f;
  const int lane = threadIdx.x & 63;
  float a = threadIdx.x * 0.001f + abias, b = 0.5f + threadIdx.x * 0.002f;
  const unsigned long long c_0 = __builtin_amdgcn_s_memtime(), r_0 = __builtin_amdgcn_s_memrealtime();
  f32x16 c0;
  for (int i = 0; i < 16; ++i) c0[i] = 0.f;
  if (PRIO == 1) {   // a pseudo-random priority per workgroup: the waves sharing a SIMD stop advancing in lock step
    const unsigned h = (blockIdx.x * 2654435761u) >> 30;
    if (h == 1) __builtin_amdgcn_s_setprio(1);
    if (h == 2) __builtin_amdgcn_s_setprio(2);
    if (h == 3) __builtin_amdgcn_s_setprio(3);
  }
  if (PRIO == 2) {   // two levels
    if ((blockIdx.x * 2654435761u) >> 31) __builtin_amdgcn_s_setprio(2);
  }
  int k = 0;
  if (PRO) {  // the convolution's prologue: the wave's trip count comes from a table (scalar load), one vector load is waited for
    const size_t w = ((size_t)blockIdx.x * THREADS + threadIdx.x) >> 6;
    iters = __builtin_amdgcn_readfirstlane(tab[w]);
    a += (float)tab[w * 0 + (threadIdx.x & 63) + 64];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    if (SALU) {  // the cursor arithmetic of the convolution's chunk loop (scalar)
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const unsigned rest = k < 31 ? mask >> (k + 1) : 0u;
        k = rest ? k + 1 + __builtin_ctz(rest) : (int)(mask & 3);
        asm volatile("" : "+s"(k));
      }
    }
  }
  const size_t wave = ((size_t)blockIdx.x * THREADS + threadIdx.x) >> 6;
  if (clk && (threadIdx.x & 63) == 0 && (wave & 63) == 0) {
    atomicAdd(clk, __builtin_amdgcn_s_memtime() - c_0);
    atomicAdd(clk + 1, __builtin_amdgcn_s_memrealtime() - r_0);
  }
  if (STORE) {
#pragma unroll
    for (int i = 0; i < 16; ++i) out[(wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c0[i] + k;
  } else if (c0[0] == 123.f) {
    out[0] = c0[1] + k;
  }
}
template <int THREADS, int LDS_KB, int STORE, int SALU, int PRO = 0, int PRIO = 0>
void run(const char* name, int waves, int iters, int spread = 0, float abias = 0.f) {
  unsigned long long* clk;
  (void)hipMalloc(&clk, 16);
  int* tab = nullptr;
  if (PRO) {
    (void)hipMalloc(&tab, (size_t)(waves + 128) * 4);
    int* h = new int[waves + 128];
    // spread: trip counts iters -+ spread, ascending with the wave index (the Gray order: heavy groups last) or, negative, descending
    for (int i = 0; i < waves + 128; ++i) {
      const double f = (double)i / waves - 0.5;
      h[i] = iters + (int)(2.0 * f * spread);
    }
    (void)hipMemcpy(tab, h, (size_t)(waves + 128) * 4, hipMemcpyHostToDevice);
    delete[] h;
  }
  float* out;
  (void)hipMalloc(&out, (size_t)waves * 32 * 32 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int blocks = waves / (THREADS / 64);
  k<THREADS, LDS_KB, STORE, SALU, PRO, PRIO><<<blocks, THREADS>>>(out, iters, 0x7ffffffu, tab, abias, nullptr);
  (void)hipMemset(clk, 0, 16);
  (void)hipEventRecord(e0);
  k<THREADS, LDS_KB, STORE, SALU, PRO, PRIO><<<blocks, THREADS>>>(out, iters, 0x7ffffffu, tab, abias, clk);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)waves * iters * 16.0;
  unsigned long long hc[2];
  (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
  printf("%-52s waves %6d x %4d MFMAs  %8.1f us  -> %6.1f TF   clock %.2f GHz\n", name, waves, iters * 16, ms * 1e3, n * 4096.0 / (ms * 1e-3) / 1e12, hc[1] ? hc[0] / (hc[1] * 10.0) : 0.0);
  (void)hipFree(out);
}
int main() {
  for (int iters : {18, 72}) {
    const int waves = 36880 * 18 / iters;
    run<64, 8, 0, 0>("64 thr, 8 KB LDS, no store", waves, iters);
    run<64, 8, 1, 0>("64 thr, 8 KB LDS, store", waves, iters);
    run<64, 8, 1, 1>("64 thr, 8 KB LDS, store, scalar cursor", waves, iters);
    run<64, 16, 1, 0>("64 thr, 16 KB LDS, store", waves, iters);
    run<64, 32, 1, 0>("64 thr, 32 KB LDS, store", waves, iters);
    run<256, 32, 0, 0>("256 thr, 32 KB LDS, no store", waves, iters);
    run<256, 32, 1, 0>("256 thr, 32 KB LDS, store", waves, iters);
    run<256, 64, 1, 0>("256 thr, 64 KB LDS, store", waves, iters);
    run<256, 32, 1, 1, 0>("256 thr, 32 KB, store, cursor", waves, iters);
    run<256, 32, 1, 0, 0>("256 thr, 32 KB, store, a += 18", waves, iters, 0, 18.f);
    run<256, 32, 1, 0, 0>("256 thr, 32 KB, store, a += 1e-3", waves, iters, 0, 1e-3f);
    run<256, 32, 1, 0, 0>("256 thr, 32 KB, store, a += 1.2345", waves, iters, 0, 1.2345f);
    run<256, 32, 1, 0, 1>("256 thr, 32 KB, store, table prologue", waves, iters);
    run<256, 32, 1, 1, 1>("256 thr, 32 KB, store, cursor, table prologue", waves, iters);
    run<256, 32, 1, 1, 1, 1>("  + s_setprio hash(workgroup) & 3", waves, iters);
    run<256, 32, 1, 1, 1, 2>("  + s_setprio 2 levels", waves, iters);
    run<256, 32, 1, 0, 0, 1>("256 thr, 32 KB, store + s_setprio hash & 3", waves, iters);
    run<256, 32, 1, 1, 1>("  ... trip counts -+50 % ascending", waves, iters, iters / 2);
    run<256, 32, 1, 1, 1>("  ... trip counts -+50 % descending", waves, iters, -iters / 2);
    run<256, 32, 1, 1, 1>("  ... trip counts -+90 % ascending", waves, iters, iters * 9 / 10);
    run<256, 32, 1, 1, 1>("  ... trip counts -+90 % descending", waves, iters, -iters * 9 / 10);
  }
  return 0;
}
