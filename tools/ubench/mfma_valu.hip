// Does VALU work overlap with f16 MFMA work on one SIMD?  (probe for k_ransac_prefilter)
// modes: 0 = 3 dependent MFMAs per unit only; 1 = + 16 dependent v_alignbit interleaved (6/6/4);
//        2 = 16 alignbit only; 3 = 3 MFMAs + 16 v_add_u32; 4 = 3 INDEPENDENT MFMAs (3 accumulators) + 16 alignbit
//        5 = MFMAs with constant C (no chain: each overwrites), + alignbit
#include <hip/hip_runtime.h>
#include <stdio.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
#define AB6 "v_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\t"
#define AB4 "v_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\t"
#define AD6 "v_add_u32 %1, %1, %6\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %1, %1, %6\n\t"
#define AD4 "v_add_u32 %1, %1, %6\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %1, %1, %6\n\t"
#define UNIT(DST, SRC)                                                                              \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\t"                                         \
               "v_alignbit_b32 %1, %1, %5, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %7, 31\n\t"   \
               "v_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %1, %1, %9, 31\n\tv_alignbit_b32 %1, %1, %10, 31\n\t" \
               "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t"                                         \
               "v_alignbit_b32 %1, %1, %11, 31\n\tv_alignbit_b32 %1, %1, %12, 31\n\tv_alignbit_b32 %1, %1, %13, 31\n\t" \
               "v_alignbit_b32 %1, %1, %14, 31\n\tv_alignbit_b32 %1, %1, %15, 31\n\tv_alignbit_b32 %1, %1, %16, 31\n\t" \
               "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t"                                         \
               "v_alignbit_b32 %1, %1, %17, 31\n\tv_alignbit_b32 %1, %1, %18, 31\n\tv_alignbit_b32 %1, %1, %19, 31\n\t" \
               "v_alignbit_b32 %1, %1, %20, 31"                                                      \
               : "=&v"(DST), "+v"(bits)                                                               \
               : "v"(a), "v"(b), "v"(c), "v"(SRC[0]), "v"(SRC[1]), "v"(SRC[2]), "v"(SRC[3]), "v"(SRC[4]),   \
                 "v"(SRC[5]), "v"(SRC[6]), "v"(SRC[7]), "v"(SRC[8]), "v"(SRC[9]), "v"(SRC[10]), "v"(SRC[11]), \
                 "v"(SRC[12]), "v"(SRC[13]), "v"(SRC[14]), "v"(SRC[15]));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f * i); }
  f32x16 c, d, d2, d3;
  for (int i = 0; i < 16; ++i) { c[i] = 1.0f; d[i] = 0.f; d2[i] = 0.f; d3[i] = 0.f; }
  unsigned bits = threadIdx.x, x = threadIdx.x * 7u;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, %0"
                   : "=&v"(d), "+v"(bits) : "v"(a), "v"(b), "v"(c), "v"(c), "v"(x));
    else if (MODE == 1)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\t" AB6 "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t" AB6 "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t" AB4
                   : "=&v"(d), "+v"(bits) : "v"(a), "v"(b), "v"(c), "v"(c), "v"(x));
    else if (MODE == 2)
      asm volatile(AB6 AB6 AB4 : "=&v"(d), "+v"(bits) : "v"(a), "v"(b), "v"(c), "v"(c), "v"(x));
    else if (MODE == 3)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\t" AD6 "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t" AD6 "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t" AD4
                   : "=&v"(d), "+v"(bits) : "v"(a), "v"(b), "v"(c), "v"(c), "v"(x));
    else if (MODE == 6) {
      UNIT(d, d2) UNIT(d2, d3) UNIT(d3, d)
    } else if (MODE == 4)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\t" AB6 "v_mfma_f32_32x32x16_f16 %7, %2, %3, %4\n\t" AB6 "v_mfma_f32_32x32x16_f16 %8, %2, %3, %4\n\t" AB4
                   : "=&v"(d), "+v"(bits) : "v"(a), "v"(b), "v"(c), "v"(c), "v"(x), "v"(d2), "v"(d3));
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += d[i] + d2[i] + d3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)bits;
}
template <int MODE>
void run(const char* name, int blocks) {
  float* out;
  (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 40000;
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
  const double waves_per_simd = blocks / 256.0;
  printf("%-44s waves/SIMD %.0f  %8.3f ms  %.1f ns per unit per SIMD\n", name, waves_per_simd, ms,
         ms * 1e6 / (iters * waves_per_simd));
  (void)hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 1024}) {
    if (blocks == 256) { run<0>("3 dep MFMA", 256); run<1>("3 dep MFMA + 16 alignbit", 256); run<2>("16 alignbit", 256); run<3>("3 dep MFMA + 16 add", 256); run<4>("3 indep MFMA + 16 alignbit", 256); run<6>("kernel-like rotating sets (x3 units)", 256); }
    if (blocks == 512) { run<0>("3 dep MFMA", 512); run<1>("3 dep MFMA + 16 alignbit", 512); run<2>("16 alignbit", 512); run<3>("3 dep MFMA + 16 add", 512); run<4>("3 indep MFMA + 16 alignbit", 512); run<6>("kernel-like rotating sets (x3 units)", 512); }
    if (blocks == 1024) { run<0>("3 dep MFMA", 1024); run<1>("3 dep MFMA + 16 alignbit", 1024); run<2>("16 alignbit", 1024); run<3>("3 dep MFMA + 16 add", 1024); run<4>("3 indep MFMA + 16 alignbit", 1024); run<6>("kernel-like rotating sets (x3 units)", 1024); }
  }
  return 0;
}
