// K = 16 prefilter unit: does the add-based sign count run at the VALU's issue rate or at the latency of its dependency
// chains?  16 v_add_f32 per unit spread over 2 (shipped) / 4 / 8 / 16 independent counters.  ns per unit per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
#define SRCS(S) "v"(S[0]), "v"(S[1]), "v"(S[2]), "v"(S[3]), "v"(S[4]), "v"(S[5]), "v"(S[6]), "v"(S[7]), \
                "v"(S[8]), "v"(S[9]), "v"(S[10]), "v"(S[11]), "v"(S[12]), "v"(S[13]), "v"(S[14]), "v"(S[15])
#define CNTS "+v"(cn[0]), "+v"(cn[1]), "+v"(cn[2]), "+v"(cn[3]), "+v"(cn[4]), "+v"(cn[5]), "+v"(cn[6]), "+v"(cn[7]), \
             "+v"(cn[8]), "+v"(cn[9]), "+v"(cn[10]), "+v"(cn[11]), "+v"(cn[12]), "+v"(cn[13]), "+v"(cn[14]), "+v"(cn[15])
#define UNIT_C2(DST, S) asm volatile("v_mfma_f32_32x32x16_f16 %0, %17, %18, %19\n\t" "v_add_f32 %1, %20, %1\n\t" "v_add_f32 %2, %21, %2\n\t" "v_add_f32 %1, %22, %1\n\t" "v_add_f32 %2, %23, %2\n\t" "v_add_f32 %1, %24, %1\n\t" "v_add_f32 %2, %25, %2\n\t" "v_add_f32 %1, %26, %1\n\t" "v_add_f32 %2, %27, %2\n\t" "v_add_f32 %1, %28, %1\n\t" "v_add_f32 %2, %29, %2\n\t" "v_add_f32 %1, %30, %1\n\t" "v_add_f32 %2, %31, %2\n\t" "v_add_f32 %1, %32, %1\n\t" "v_add_f32 %2, %33, %2\n\t" "v_add_f32 %1, %34, %1\n\t" "v_add_f32 %2, %35, %2" : "=&v"(DST), CNTS : "v"(a), "v"(b), "v"(c), SRCS(S));
#define UNIT_C4(DST, S) asm volatile("v_mfma_f32_32x32x16_f16 %0, %17, %18, %19\n\t" "v_add_f32 %1, %20, %1\n\t" "v_add_f32 %2, %21, %2\n\t" "v_add_f32 %3, %22, %3\n\t" "v_add_f32 %4, %23, %4\n\t" "v_add_f32 %1, %24, %1\n\t" "v_add_f32 %2, %25, %2\n\t" "v_add_f32 %3, %26, %3\n\t" "v_add_f32 %4, %27, %4\n\t" "v_add_f32 %1, %28, %1\n\t" "v_add_f32 %2, %29, %2\n\t" "v_add_f32 %3, %30, %3\n\t" "v_add_f32 %4, %31, %4\n\t" "v_add_f32 %1, %32, %1\n\t" "v_add_f32 %2, %33, %2\n\t" "v_add_f32 %3, %34, %3\n\t" "v_add_f32 %4, %35, %4" : "=&v"(DST), CNTS : "v"(a), "v"(b), "v"(c), SRCS(S));
#define UNIT_C8(DST, S) asm volatile("v_mfma_f32_32x32x16_f16 %0, %17, %18, %19\n\t" "v_add_f32 %1, %20, %1\n\t" "v_add_f32 %2, %21, %2\n\t" "v_add_f32 %3, %22, %3\n\t" "v_add_f32 %4, %23, %4\n\t" "v_add_f32 %5, %24, %5\n\t" "v_add_f32 %6, %25, %6\n\t" "v_add_f32 %7, %26, %7\n\t" "v_add_f32 %8, %27, %8\n\t" "v_add_f32 %1, %28, %1\n\t" "v_add_f32 %2, %29, %2\n\t" "v_add_f32 %3, %30, %3\n\t" "v_add_f32 %4, %31, %4\n\t" "v_add_f32 %5, %32, %5\n\t" "v_add_f32 %6, %33, %6\n\t" "v_add_f32 %7, %34, %7\n\t" "v_add_f32 %8, %35, %8" : "=&v"(DST), CNTS : "v"(a), "v"(b), "v"(c), SRCS(S));
#define UNIT_C16(DST, S) asm volatile("v_mfma_f32_32x32x16_f16 %0, %17, %18, %19\n\t" "v_add_f32 %1, %20, %1\n\t" "v_add_f32 %2, %21, %2\n\t" "v_add_f32 %3, %22, %3\n\t" "v_add_f32 %4, %23, %4\n\t" "v_add_f32 %5, %24, %5\n\t" "v_add_f32 %6, %25, %6\n\t" "v_add_f32 %7, %26, %7\n\t" "v_add_f32 %8, %27, %8\n\t" "v_add_f32 %9, %28, %9\n\t" "v_add_f32 %10, %29, %10\n\t" "v_add_f32 %11, %30, %11\n\t" "v_add_f32 %12, %31, %12\n\t" "v_add_f32 %13, %32, %13\n\t" "v_add_f32 %14, %33, %14\n\t" "v_add_f32 %15, %34, %15\n\t" "v_add_f32 %16, %35, %16" : "=&v"(DST), CNTS : "v"(a), "v"(b), "v"(c), SRCS(S));
#define UNIT_NOMFMA(DST, S) asm volatile("v_add_f32 %1, %20, %1\n\t" "v_add_f32 %2, %21, %2\n\t" "v_add_f32 %1, %22, %1\n\t" "v_add_f32 %2, %23, %2\n\t" "v_add_f32 %1, %24, %1\n\t" "v_add_f32 %2, %25, %2\n\t" "v_add_f32 %1, %26, %1\n\t" "v_add_f32 %2, %27, %2\n\t" "v_add_f32 %1, %28, %1\n\t" "v_add_f32 %2, %29, %2\n\t" "v_add_f32 %1, %30, %1\n\t" "v_add_f32 %2, %31, %2\n\t" "v_add_f32 %1, %32, %1\n\t" "v_add_f32 %2, %33, %2\n\t" "v_add_f32 %1, %34, %1\n\t" "v_add_f32 %2, %35, %2" : "+v"(DST), CNTS : "v"(a), "v"(b), "v"(c), SRCS(S));
#define UNIT_NOMFMA8(DST, S) asm volatile("v_add_f32 %1, %20, %1\n\t" "v_add_f32 %2, %21, %2\n\t" "v_add_f32 %3, %22, %3\n\t" "v_add_f32 %4, %23, %4\n\t" "v_add_f32 %5, %24, %5\n\t" "v_add_f32 %6, %25, %6\n\t" "v_add_f32 %7, %26, %7\n\t" "v_add_f32 %8, %27, %8\n\t" "v_add_f32 %1, %28, %1\n\t" "v_add_f32 %2, %29, %2\n\t" "v_add_f32 %3, %30, %3\n\t" "v_add_f32 %4, %31, %4\n\t" "v_add_f32 %5, %32, %5\n\t" "v_add_f32 %6, %33, %6\n\t" "v_add_f32 %7, %34, %7\n\t" "v_add_f32 %8, %35, %8" : "+v"(DST), CNTS : "v"(a), "v"(b), "v"(c), SRCS(S));

template <int X>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.0001f + 0.01f * i); b[i] = (_Float16)(0.05f * i); }
  f32x16 c, d, d2, d3;
  for (int i = 0; i < 16; ++i) { c[i] = -0.25f; d[i] = 0.f; d2[i] = 0.f; d3[i] = 0.f; }
  float cn[16];
  for (int i = 0; i < 16; ++i) cn[i] = 16777215.0f;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2");
  for (int i = 0; i < iters; ++i) {
    if (X == 0) { UNIT_C2(d, d2) UNIT_C2(d2, d3) UNIT_C2(d3, d) }
    if (X == 1) { UNIT_C4(d, d2) UNIT_C4(d2, d3) UNIT_C4(d3, d) }
    if (X == 2) { UNIT_C8(d, d2) UNIT_C8(d2, d3) UNIT_C8(d3, d) }
    if (X == 3) { UNIT_C16(d, d2) UNIT_C16(d2, d3) UNIT_C16(d3, d) }
    if (X == 4) { UNIT_NOMFMA(d, d2) UNIT_NOMFMA(d2, d3) UNIT_NOMFMA(d3, d) }
    if (X == 5) { UNIT_NOMFMA8(d, d2) UNIT_NOMFMA8(d2, d3) UNIT_NOMFMA8(d3, d) }
    if ((i & 1023) == 1023) for (int j = 0; j < 16; ++j) cn[j] = 16777215.0f;
  }
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += d[i] + d2[i] + d3[i] + cn[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
static const char* NAMES[] = {"1 MFMA + 16 v_add_f32 in  2 chains (shipped)", "1 MFMA + 16 v_add_f32 in  4 chains", "1 MFMA + 16 v_add_f32 in  8 chains",
                              "1 MFMA + 16 v_add_f32 in 16 chains", "         16 v_add_f32 in  2 chains, no MFMA", "         16 v_add_f32 in  8 chains, no MFMA"};
template <int X>
void run(int blocks) {
  float* out;
  (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<X><<<blocks, 256>>>(out, 2000);
  (void)hipEventRecord(e0);
  k<X><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double w = blocks / 256.0;
  printf("%-48s waves/SIMD %.0f  %8.3f ms  %6.1f ns per unit per SIMD\n", NAMES[X], w, ms, ms * 1e6 / ((double)iters * 3 * w));
  (void)hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 1024, 1536}) {
    run<0>(blocks); run<1>(blocks); run<2>(blocks); run<3>(blocks); run<4>(blocks); run<5>(blocks);
  }
  return 0;
}
