// What would a ONE-MFMA prefilter unit cost?  (K = 16: a_hi . b_hi only, the dropped a_hi . b_lo bounded per PAIR and folded
// into the pair's constant term.)  Unit = 1 x v_mfma_f32_32x32x16_f16 + 16 x v_alignbit per 32 x 32 tile, three rotating
// accumulator sets as in k_ransac_prefilter; ns per unit per SIMD at 1 - 4 waves per SIMD, next to the shipped K = 32 unit.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
#define SRCS(S) "v"(S[0]), "v"(S[1]), "v"(S[2]), "v"(S[3]), "v"(S[4]), "v"(S[5]), "v"(S[6]), "v"(S[7]), \
                "v"(S[8]), "v"(S[9]), "v"(S[10]), "v"(S[11]), "v"(S[12]), "v"(S[13]), "v"(S[14]), "v"(S[15])
#define AB8(o) "v_alignbit_b32 %1, %1, %" #o ", 31\n\t"
#define UNIT_K32(DST, S)                                                                               \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\t"                                         \
               "v_alignbit_b32 %1, %1, %5, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %7, 31\n\t"   \
               "v_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %1, %1, %9, 31\n\tv_alignbit_b32 %1, %1, %10, 31\n\t" \
               "v_alignbit_b32 %1, %1, %11, 31\n\tv_alignbit_b32 %1, %1, %12, 31\n\t"                  \
               "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t"                                         \
               "v_alignbit_b32 %1, %1, %13, 31\n\tv_alignbit_b32 %1, %1, %14, 31\n\tv_alignbit_b32 %1, %1, %15, 31\n\t" \
               "v_alignbit_b32 %1, %1, %16, 31\n\tv_alignbit_b32 %1, %1, %17, 31\n\tv_alignbit_b32 %1, %1, %18, 31\n\t" \
               "v_alignbit_b32 %1, %1, %19, 31\n\tv_alignbit_b32 %1, %1, %20, 31"                      \
               : "=&v"(DST), "+v"(bits) : "v"(a), "v"(b), "v"(c), SRCS(S));
#define UNIT_K16(DST, S)                                                                               \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\t"                                         \
               "v_alignbit_b32 %1, %1, %5, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %7, 31\n\t"   \
               "v_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %1, %1, %9, 31\n\tv_alignbit_b32 %1, %1, %10, 31\n\t" \
               "v_alignbit_b32 %1, %1, %11, 31\n\tv_alignbit_b32 %1, %1, %12, 31\n\t"                  \
               "v_alignbit_b32 %1, %1, %13, 31\n\tv_alignbit_b32 %1, %1, %14, 31\n\tv_alignbit_b32 %1, %1, %15, 31\n\t" \
               "v_alignbit_b32 %1, %1, %16, 31\n\tv_alignbit_b32 %1, %1, %17, 31\n\tv_alignbit_b32 %1, %1, %18, 31\n\t" \
               "v_alignbit_b32 %1, %1, %19, 31\n\tv_alignbit_b32 %1, %1, %20, 31"                      \
               : "=&v"(DST), "+v"(bits) : "v"(a), "v"(b), "v"(c), SRCS(S));
// two history words (no dependent chain between consecutive alignbits)
#define UNIT_K16B(DST, S)                                                                              \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %5\n\t"                                         \
               "v_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %2, %2, %7, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\t"   \
               "v_alignbit_b32 %2, %2, %9, 31\n\tv_alignbit_b32 %1, %1, %10, 31\n\tv_alignbit_b32 %2, %2, %11, 31\n\t" \
               "v_alignbit_b32 %1, %1, %12, 31\n\tv_alignbit_b32 %2, %2, %13, 31\n\t"                  \
               "v_alignbit_b32 %1, %1, %14, 31\n\tv_alignbit_b32 %2, %2, %15, 31\n\tv_alignbit_b32 %1, %1, %16, 31\n\t" \
               "v_alignbit_b32 %2, %2, %17, 31\n\tv_alignbit_b32 %1, %1, %18, 31\n\tv_alignbit_b32 %2, %2, %19, 31\n\t" \
               "v_alignbit_b32 %1, %1, %20, 31\n\tv_alignbit_b32 %2, %2, %21, 31"                      \
               : "=&v"(DST), "+v"(bits), "+v"(bits2) : "v"(a), "v"(b), "v"(c), SRCS(S));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f * i); }
  f32x16 c, d, d2, d3;
  for (int i = 0; i < 16; ++i) { c[i] = 1.0f; d[i] = 0.f; d2[i] = 0.f; d3[i] = 0.f; }
  unsigned bits = threadIdx.x, bits2 = threadIdx.x * 3u;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { UNIT_K32(d, d2) UNIT_K32(d2, d3) UNIT_K32(d3, d) }
    else if (MODE == 1) { UNIT_K16(d, d2) UNIT_K16(d2, d3) UNIT_K16(d3, d) }
    else { UNIT_K16B(d, d2) UNIT_K16B(d2, d3) UNIT_K16B(d3, d) }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += d[i] + d2[i] + d3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)bits + (float)bits2;
}
template <int MODE>
void run(const char* name, int blocks) {
  float* out;
  (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 2000);
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double w = blocks / 256.0;
  printf("%-52s waves/SIMD %.0f  %8.3f ms  %6.1f ns per unit per SIMD\n", name, w, ms, ms * 1e6 / ((double)iters * 3 * w));
  (void)hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 768, 1024, 1280, 1536}) {
    run<0>("K32 unit: 2 MFMA + 16 v_alignbit (shipped)", blocks);
    run<1>("K16 unit: 1 MFMA + 16 v_alignbit", blocks);
    run<2>("K16 unit: 1 MFMA + 16 v_alignbit, two history words", blocks);
  }
  return 0;
}
