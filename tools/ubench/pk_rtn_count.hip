// Round 4 probe for the prefilter's sign-count floor (DESIGN "The prefilter's vector-ALU floor"): every way of counting the
// sign of an MFMA result found so far costs ONE slow-class VALU op (4.5 cycles) per result register = 72 cycles per
// 32 x 32 tile against 64 MFMA cycles.  A VOP3P op reads 64-bit operands, i.e. TWO accumulator registers at once:
// under round-toward-minus-infinity `v_pk_add_f32 cnt[0:1], acc[2j:2j+1], cnt[0:1]` with both counters in [2^23, 2^24)
// subtracts 1 from counter h iff acc[2j+h] < 0 (tools/ubench/rtn_count.hip has the scalar form and its edge cases).
// 8 ops per tile instead of 16.  Question: do they overlap with the f16 MFMAs, unlike v_add_f32 (39.3 ns per unit)?
//   0: 2 MFMA + 16 v_alignbit (the shipped unit)      1: 2 MFMA + 8 v_pk_add_f32, one counter pair
//   2: 2 MFMA + 8 v_pk_add_f32, two counter pairs     3: 2 MFMA + 4 v_pk_add_f32 + 8 v_alignbit
//   4: 8 v_pk_add_f32 alone                           5: 2 MFMA alone
#include <hip/hip_runtime.h>
#include <stdio.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
#define P(S, j) "v"(__builtin_shufflevector(S, S, 2 * (j), 2 * (j) + 1))
#define PAIRS(S) P(S, 0), P(S, 1), P(S, 2), P(S, 3), P(S, 4), P(S, 5), P(S, 6), P(S, 7)
#define SRCS(S) "v"(S[0]), "v"(S[1]), "v"(S[2]), "v"(S[3]), "v"(S[4]), "v"(S[5]), "v"(S[6]), "v"(S[7]), \
                "v"(S[8]), "v"(S[9]), "v"(S[10]), "v"(S[11]), "v"(S[12]), "v"(S[13]), "v"(S[14]), "v"(S[15])
// operands: 0 dst, 1 cnt (pair), 2 cnt2 (pair), 3 a, 4 b, 5 c, 6..13 source pairs
#define UNIT_P1(DST, S)                                                                              \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %5\n\t"                                         \
               "v_pk_add_f32 %1, %6, %1\n\tv_pk_add_f32 %1, %7, %1\n\tv_pk_add_f32 %1, %8, %1\n\tv_pk_add_f32 %1, %9, %1\n\t" \
               "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\t"                                         \
               "v_pk_add_f32 %1, %10, %1\n\tv_pk_add_f32 %1, %11, %1\n\tv_pk_add_f32 %1, %12, %1\n\tv_pk_add_f32 %1, %13, %1" \
               : "=&v"(DST), "+v"(c1), "+v"(c2) : "v"(a), "v"(b), "v"(c), PAIRS(S));
#define UNIT_P2(DST, S)                                                                              \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %5\n\t"                                         \
               "v_pk_add_f32 %1, %6, %1\n\tv_pk_add_f32 %2, %7, %2\n\tv_pk_add_f32 %1, %8, %1\n\tv_pk_add_f32 %2, %9, %2\n\t" \
               "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\t"                                         \
               "v_pk_add_f32 %1, %10, %1\n\tv_pk_add_f32 %2, %11, %2\n\tv_pk_add_f32 %1, %12, %1\n\tv_pk_add_f32 %2, %13, %2" \
               : "=&v"(DST), "+v"(c1), "+v"(c2) : "v"(a), "v"(b), "v"(c), PAIRS(S));
// 4 pk adds (results 0..7) + 8 alignbit (results 8..15): operands 0 dst, 1 cnt, 2 bits, 3 a, 4 b, 5 c, 6..9 pairs, 10..17 regs
#define UNIT_MIX(DST, S)                                                                             \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %5\n\t"                                         \
               "v_pk_add_f32 %1, %6, %1\n\tv_alignbit_b32 %2, %2, %10, 31\n\tv_alignbit_b32 %2, %2, %11, 31\n\t" \
               "v_pk_add_f32 %1, %7, %1\n\tv_alignbit_b32 %2, %2, %12, 31\n\tv_alignbit_b32 %2, %2, %13, 31\n\t" \
               "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\t"                                         \
               "v_pk_add_f32 %1, %8, %1\n\tv_alignbit_b32 %2, %2, %14, 31\n\tv_alignbit_b32 %2, %2, %15, 31\n\t" \
               "v_pk_add_f32 %1, %9, %1\n\tv_alignbit_b32 %2, %2, %16, 31\n\tv_alignbit_b32 %2, %2, %17, 31" \
               : "=&v"(DST), "+v"(c1), "+v"(bits) : "v"(a), "v"(b), "v"(c), P(S, 0), P(S, 1), P(S, 2), P(S, 3),       \
                 "v"(S[8]), "v"(S[9]), "v"(S[10]), "v"(S[11]), "v"(S[12]), "v"(S[13]), "v"(S[14]), "v"(S[15]));
#define UNIT_A(DST, S)                                                                               \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\t"                                         \
               "v_alignbit_b32 %1, %1, %5, 31\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %7, 31\n\t"   \
               "v_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %1, %1, %9, 31\n\tv_alignbit_b32 %1, %1, %10, 31\n\t" \
               "v_alignbit_b32 %1, %1, %11, 31\n\tv_alignbit_b32 %1, %1, %12, 31\n\t"                  \
               "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t"                                         \
               "v_alignbit_b32 %1, %1, %13, 31\n\tv_alignbit_b32 %1, %1, %14, 31\n\tv_alignbit_b32 %1, %1, %15, 31\n\t" \
               "v_alignbit_b32 %1, %1, %16, 31\n\tv_alignbit_b32 %1, %1, %17, 31\n\tv_alignbit_b32 %1, %1, %18, 31\n\t" \
               "v_alignbit_b32 %1, %1, %19, 31\n\tv_alignbit_b32 %1, %1, %20, 31"                      \
               : "=&v"(DST), "+v"(bits) : "v"(a), "v"(b), "v"(c), SRCS(S));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.0001f + 0.01f * i); b[i] = (_Float16)(0.05f * i); }
  f32x16 c, d, d2, d3;
  for (int i = 0; i < 16; ++i) { c[i] = -0.25f; d[i] = 0.f; d2[i] = 0.f; d3[i] = 0.f; }
  f32x2 c1 = {16777215.0f, 16777215.0f}, c2 = {16777215.0f, 16777215.0f};
  unsigned bits = threadIdx.x;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2");
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { UNIT_A(d, d2) UNIT_A(d2, d3) UNIT_A(d3, d) }
    else if (MODE == 1) { UNIT_P1(d, d2) UNIT_P1(d2, d3) UNIT_P1(d3, d) }
    else if (MODE == 2) { UNIT_P2(d, d2) UNIT_P2(d2, d3) UNIT_P2(d3, d) }
    else if (MODE == 3) { UNIT_MIX(d, d2) UNIT_MIX(d2, d3) UNIT_MIX(d3, d) }
    else if (MODE == 4) {
      for (int r = 0; r < 3; ++r)
        asm volatile("v_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %3, %1\n\tv_pk_add_f32 %0, %4, %0\n\tv_pk_add_f32 %1, %5, %1\n\t"
                     "v_pk_add_f32 %0, %6, %0\n\tv_pk_add_f32 %1, %7, %1\n\tv_pk_add_f32 %0, %8, %0\n\tv_pk_add_f32 %1, %9, %1"
                     : "+v"(c1), "+v"(c2) : PAIRS(d2));
    } else {
      for (int r = 0; r < 3; ++r)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    }
    if ((i & 1023) == 1023) { c1[0] = c1[1] = c2[0] = c2[1] = 16777215.0f; }
  }
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += d[i] + d2[i] + d3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + c1[0] + c1[1] + c2[0] + c2[1] + (float)bits;
}
__global__ void k_check(const float* x, int n, float* out) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2");
  f32x2 cnt = {16777215.0f, 16777215.0f};
  for (int i = 0; i + 1 < n; i += 2) {
    f32x2 v = {x[i], x[i + 1]};
    asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(cnt) : "v"(v));
  }
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
  out[0] = cnt[0];
  out[1] = cnt[1];
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
  const float xs[] = {-0.5f, 0.5f, -1e-30f, 1e-30f, -0.0f, 0.0f, -0.999999f, 0.999999f, -5.9604645e-08f, 5.9604645e-08f,
                      -1.0e-38f, -1.4e-45f, 1.4e-45f, -3.5527137e-15f, -0.75f, 0.25f};
  const int n = sizeof(xs) / sizeof(xs[0]);
  float *dx, *dout, hout[2];
  (void)hipMalloc(&dx, sizeof(xs));
  (void)hipMalloc(&dout, 8);
  int e0 = 0, e1 = 0;
  for (int i = 0; i < n; i += 2) { e0 += xs[i] < 0.0f; e1 += xs[i + 1] < 0.0f; }
  (void)hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
  k_check<<<1, 1>>>(dx, n, dout);
  (void)hipMemcpy(hout, dout, 8, hipMemcpyDeviceToHost);
  printf("v_pk_add_f32 under RTN: negatives counted (%.0f, %.0f), expected (%d, %d)\n", 16777215.0 - hout[0],
         16777215.0 - hout[1], e0, e1);
  for (int blocks : {256, 512, 768, 1024}) {
    run<5>("2 dep MFMA alone", blocks);
    run<4>("8 v_pk_add_f32 alone", blocks);
    run<0>("K32 unit: 2 MFMA + 16 v_alignbit (shipped)", blocks);
    run<1>("K32 unit: 2 MFMA + 8 v_pk_add_f32 (one pair)", blocks);
    run<2>("K32 unit: 2 MFMA + 8 v_pk_add_f32 (two pairs)", blocks);
    run<3>("K32 unit: 2 MFMA + 4 v_pk_add_f32 + 8 v_alignbit", blocks);
  }
  return 0;
}
