// Counting negatives with ONE full-rate VALU op per value: under round-toward-minus-infinity
// (MODE.fp_round f32 = 2) and with sum in [2^23, 2^24) (ulp 1), sum + x for |x| < 1 is sum - 1 iff x < 0.
// Part 1: correctness on edge values.  Part 2: the K = 32 prefilter unit with 16 v_add_f32 instead of
// 16 v_alignbit (ns per unit per SIMD at 1-4 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void k_check(const float* x, int n, float* out) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2");
  float sum = 16777215.0f;  // 2^24 - 1
  for (int i = 0; i < n; ++i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(sum) : "v"(x[i]));
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
  out[threadIdx.x] = sum;
}

#define SRCS "v"(SRC[0]), "v"(SRC[1]), "v"(SRC[2]), "v"(SRC[3]), "v"(SRC[4]), "v"(SRC[5]), "v"(SRC[6]), "v"(SRC[7]), \
             "v"(SRC[8]), "v"(SRC[9]), "v"(SRC[10]), "v"(SRC[11]), "v"(SRC[12]), "v"(SRC[13]), "v"(SRC[14]), "v"(SRC[15])
#define UNIT_F(DST, SRC)                                                                             \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %5\n\t"                                         \
               "v_add_f32 %1, %6, %1\n\tv_add_f32 %2, %7, %2\n\tv_add_f32 %1, %8, %1\n\t"   \
               "v_add_f32 %2, %9, %2\n\tv_add_f32 %1, %10, %1\n\tv_add_f32 %2, %11, %2\n\t" \
               "v_add_f32 %1, %12, %1\n\tv_add_f32 %2, %13, %2\n\t"                  \
               "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\t"                                         \
               "v_add_f32 %1, %14, %1\n\tv_add_f32 %2, %15, %2\n\tv_add_f32 %1, %16, %1\n\t" \
               "v_add_f32 %2, %17, %2\n\tv_add_f32 %1, %18, %1\n\tv_add_f32 %2, %19, %2\n\t" \
               "v_add_f32 %1, %20, %1\n\tv_add_f32 %2, %21, %2"                      \
               : "=&v"(DST), "+v"(c1), "+v"(c2) : "v"(a), "v"(b), "v"(c), SRCS);
#define UNIT_1(DST, SRC)                                                                             \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %5\n\t"                                         \
               "v_add_f32 %1, %6, %1\n\tv_add_f32 %1, %7, %1\n\tv_add_f32 %1, %8, %1\n\t"   \
               "v_add_f32 %1, %9, %1\n\tv_add_f32 %1, %10, %1\n\tv_add_f32 %1, %11, %1\n\t" \
               "v_add_f32 %1, %12, %1\n\tv_add_f32 %1, %13, %1\n\t"                  \
               "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\t"                                         \
               "v_add_f32 %1, %14, %1\n\tv_add_f32 %1, %15, %1\n\tv_add_f32 %1, %16, %1\n\t" \
               "v_add_f32 %1, %17, %1\n\tv_add_f32 %1, %18, %1\n\tv_add_f32 %1, %19, %1\n\t" \
               "v_add_f32 %1, %20, %1\n\tv_add_f32 %1, %21, %1"                      \
               : "=&v"(DST), "+v"(c1), "+v"(c2) : "v"(a), "v"(b), "v"(c), SRCS);
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.0001f + 0.01f * i); b[i] = (_Float16)(0.05f * i); }
  f32x16 c, d, d2, d3;
  for (int i = 0; i < 16; ++i) { c[i] = -0.25f; d[i] = 0.f; d2[i] = 0.f; d3[i] = 0.f; }
  float c1 = 16777215.0f, c2 = 16777215.0f;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2");
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#define SRC d2
      UNIT_F(d, d2)
#undef SRC
#define SRC d3
      UNIT_F(d2, d3)
#undef SRC
#define SRC d
      UNIT_F(d3, d)
#undef SRC
    } else {
#define SRC d2
      UNIT_1(d, d2)
#undef SRC
#define SRC d3
      UNIT_1(d2, d3)
#undef SRC
#define SRC d
      UNIT_1(d3, d)
#undef SRC
    }
    if ((i & 1023) == 1023) { c1 = 16777215.0f; c2 = 16777215.0f; }
  }
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += d[i] + d2[i] + d3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + c1 + c2;
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
  printf("%-48s waves/SIMD %.0f  %8.3f ms  %6.1f ns per unit per SIMD\n", name, w, ms, ms * 1e6 / ((double)iters * 3 * w));
  (void)hipFree(out);
}
int main() {
  const float xs[] = {-0.5f, 0.5f, -1e-30f, 1e-30f, -0.0f, 0.0f, -0.999999f, 0.999999f, -5.9604645e-08f /* -2^-24 */,
                      5.9604645e-08f, -1.0e-38f, -1.4e-45f /* denormal */, 1.4e-45f, -3.5527137e-15f /* -2^-48 */};
  const int n = sizeof(xs) / sizeof(xs[0]);
  float *dx, *dout, hout[64];
  (void)hipMalloc(&dx, sizeof(xs));
  (void)hipMalloc(&dout, 64 * 4);
  int expect = 0;
  for (int i = 0; i < n; ++i) expect += xs[i] < 0.0f;
  (void)hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
  k_check<<<1, 64>>>(dx, n, dout);
  (void)hipMemcpy(hout, dout, 64 * 4, hipMemcpyDeviceToHost);
  printf("negatives counted %.0f, expected %d (of %d values incl. denormals)\n", 16777215.0 - hout[0], expect, n);
  for (int i = 0; i < n; ++i) {
    (void)hipMemcpy(dx, xs + i, 4, hipMemcpyHostToDevice);
    k_check<<<1, 64>>>(dx, 1, dout);
    (void)hipMemcpy(hout, dout, 4, hipMemcpyDeviceToHost);
    printf("  x = %-14g -> counted %.0f\n", xs[i], 16777215.0 - hout[0]);
  }
  for (int blocks : {256, 512, 768, 1024}) {
    run<0>("K32 unit, 16 v_add_f32 (two counters)", blocks);
    run<1>("K32 unit, 16 v_add_f32 (one counter)", blocks);
  }
  return 0;
}
