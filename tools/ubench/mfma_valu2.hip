// Follow-up probe for the K = 32 prefilter unit (2 dependent f16 MFMAs + 16 sign extractions per 32 x 32 tile):
// what does one unit cost per SIMD with 1 / 2 / 3 / 4 waves resident, and which part does not overlap?
//   0: 2 dependent MFMAs only                      1: 16 v_alignbit only (one history word)
//   2: unit with rotating accumulator sets (kernel-like), one history word
//   3: same, two history words (no dependent VALU chain)
//   4: same as 2, first MFMA takes an inline-constant C (no 16-register C read)
//   5: same as 2 with 8 alignbit only (how much is VALU-issue bound)
//   6: same as 2 with v_cmp_lt_f32 + s_bcnt1 on half of the results (SALU takes half the counting)
#include <hip/hip_runtime.h>
#include <stdio.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
#define SRCS "v"(SRC[0]), "v"(SRC[1]), "v"(SRC[2]), "v"(SRC[3]), "v"(SRC[4]), "v"(SRC[5]), "v"(SRC[6]), "v"(SRC[7]), \
             "v"(SRC[8]), "v"(SRC[9]), "v"(SRC[10]), "v"(SRC[11]), "v"(SRC[12]), "v"(SRC[13]), "v"(SRC[14]), "v"(SRC[15])
// operands: 0 dst, 1 bits, 2 bits2, 3 a, 4 b, 5 c, 6.. src
#define UNIT_A(DST, SRC, C0)                                                                         \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, " C0 "\n\t"                                      \
               "v_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %7, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\t"   \
               "v_alignbit_b32 %1, %1, %9, 31\n\tv_alignbit_b32 %1, %1, %10, 31\n\tv_alignbit_b32 %1, %1, %11, 31\n\t" \
               "v_alignbit_b32 %1, %1, %12, 31\n\tv_alignbit_b32 %1, %1, %13, 31\n\t"                  \
               "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\t"                                         \
               "v_alignbit_b32 %1, %1, %14, 31\n\tv_alignbit_b32 %1, %1, %15, 31\n\tv_alignbit_b32 %1, %1, %16, 31\n\t" \
               "v_alignbit_b32 %1, %1, %17, 31\n\tv_alignbit_b32 %1, %1, %18, 31\n\tv_alignbit_b32 %1, %1, %19, 31\n\t" \
               "v_alignbit_b32 %1, %1, %20, 31\n\tv_alignbit_b32 %1, %1, %21, 31"                      \
               : "=&v"(DST), "+v"(bits), "+v"(bits2) : "v"(a), "v"(b), "v"(c), SRCS);
#define UNIT_B(DST, SRC)                                                                             \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %5\n\t"                                         \
               "v_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %2, %2, %7, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\t"   \
               "v_alignbit_b32 %2, %2, %9, 31\n\tv_alignbit_b32 %1, %1, %10, 31\n\tv_alignbit_b32 %2, %2, %11, 31\n\t" \
               "v_alignbit_b32 %1, %1, %12, 31\n\tv_alignbit_b32 %2, %2, %13, 31\n\t"                  \
               "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\t"                                         \
               "v_alignbit_b32 %1, %1, %14, 31\n\tv_alignbit_b32 %2, %2, %15, 31\n\tv_alignbit_b32 %1, %1, %16, 31\n\t" \
               "v_alignbit_b32 %2, %2, %17, 31\n\tv_alignbit_b32 %1, %1, %18, 31\n\tv_alignbit_b32 %2, %2, %19, 31\n\t" \
               "v_alignbit_b32 %1, %1, %20, 31\n\tv_alignbit_b32 %2, %2, %21, 31"                      \
               : "=&v"(DST), "+v"(bits), "+v"(bits2) : "v"(a), "v"(b), "v"(c), SRCS);
#define UNIT_H(DST, SRC)                                                                             \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %5\n\t"                                         \
               "v_alignbit_b32 %1, %1, %6, 31\n\tv_alignbit_b32 %1, %1, %7, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\t"   \
               "v_alignbit_b32 %1, %1, %9, 31\n\t"                                                    \
               "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\t"                                         \
               "v_alignbit_b32 %1, %1, %14, 31\n\tv_alignbit_b32 %1, %1, %15, 31\n\tv_alignbit_b32 %1, %1, %16, 31\n\t" \
               "v_alignbit_b32 %1, %1, %17, 31"                                                      \
               : "=&v"(DST), "+v"(bits), "+v"(bits2) : "v"(a), "v"(b), "v"(c), SRCS);
#define UNIT_S(DST, SRC)                                                                             \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %6\n\t"                                         \
               "v_alignbit_b32 %1, %1, %7, 31\n\tv_cmp_gt_f32 vcc, 0, %8\n\tv_alignbit_b32 %1, %1, %9, 31\n\t"   \
               "s_bcnt1_i32_b64 s20, vcc\n\tv_cmp_gt_f32 vcc, 0, %10\n\ts_add_u32 %3, %3, s20\n\tv_alignbit_b32 %1, %1, %11, 31\n\t" \
               "s_bcnt1_i32_b64 s20, vcc\n\tv_cmp_gt_f32 vcc, 0, %12\n\ts_add_u32 %3, %3, s20\n\t" \
               "v_alignbit_b32 %1, %1, %13, 31\n\ts_bcnt1_i32_b64 s20, vcc\n\tv_cmp_gt_f32 vcc, 0, %14\n\ts_add_u32 %3, %3, s20\n\t"                  \
               "v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\t"                                         \
               "v_alignbit_b32 %1, %1, %15, 31\n\ts_bcnt1_i32_b64 s20, vcc\n\tv_cmp_gt_f32 vcc, 0, %16\n\ts_add_u32 %3, %3, s20\n\tv_alignbit_b32 %1, %1, %17, 31\n\t" \
               "s_bcnt1_i32_b64 s20, vcc\n\tv_cmp_gt_f32 vcc, 0, %18\n\ts_add_u32 %3, %3, s20\n\tv_alignbit_b32 %1, %1, %19, 31\n\t" \
               "s_bcnt1_i32_b64 s20, vcc\n\tv_cmp_gt_f32 vcc, 0, %20\n\ts_add_u32 %3, %3, s20\n\t" \
               "v_alignbit_b32 %1, %1, %21, 31\n\ts_bcnt1_i32_b64 s20, vcc\n\tv_cmp_gt_f32 vcc, 0, %22\n\ts_add_u32 %3, %3, s20\n\t" \
               "s_bcnt1_i32_b64 s20, vcc\n\ts_add_u32 %3, %3, s20"                                   \
               : "=&v"(DST), "+v"(bits), "+v"(bits2), "+s"(scnt) : "v"(a), "v"(b), "v"(c), SRCS : "vcc", "s20", "scc");
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f * i); }
  f32x16 c, d, d2, d3;
  for (int i = 0; i < 16; ++i) { c[i] = 1.0f; d[i] = 0.f; d2[i] = 0.f; d3[i] = 0.f; }
  unsigned bits = threadIdx.x, bits2 = threadIdx.x * 3u;
  int scnt = 0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, %0"
                   : "=&v"(d), "+v"(bits) : "v"(a), "v"(b), "v"(c));
    else if (MODE == 1) {
#define SRC d2
      asm volatile("v_alignbit_b32 %0, %0, %1, 31\n\tv_alignbit_b32 %0, %0, %2, 31\n\tv_alignbit_b32 %0, %0, %3, 31\n\tv_alignbit_b32 %0, %0, %4, 31\n\t"
                   "v_alignbit_b32 %0, %0, %5, 31\n\tv_alignbit_b32 %0, %0, %6, 31\n\tv_alignbit_b32 %0, %0, %7, 31\n\tv_alignbit_b32 %0, %0, %8, 31\n\t"
                   "v_alignbit_b32 %0, %0, %9, 31\n\tv_alignbit_b32 %0, %0, %10, 31\n\tv_alignbit_b32 %0, %0, %11, 31\n\tv_alignbit_b32 %0, %0, %12, 31\n\t"
                   "v_alignbit_b32 %0, %0, %13, 31\n\tv_alignbit_b32 %0, %0, %14, 31\n\tv_alignbit_b32 %0, %0, %15, 31\n\tv_alignbit_b32 %0, %0, %16, 31"
                   : "+v"(bits) : SRCS);
#undef SRC
    } else if (MODE == 2) {
#define SRC d2
      UNIT_A(d, d2, "%5")
#undef SRC
#define SRC d3
      UNIT_A(d2, d3, "%5")
#undef SRC
#define SRC d
      UNIT_A(d3, d, "%5")
#undef SRC
    } else if (MODE == 3) {
#define SRC d2
      UNIT_B(d, d2)
#undef SRC
#define SRC d3
      UNIT_B(d2, d3)
#undef SRC
#define SRC d
      UNIT_B(d3, d)
#undef SRC
    } else if (MODE == 4) {
#define SRC d2
      UNIT_A(d, d2, "0")
#undef SRC
#define SRC d3
      UNIT_A(d2, d3, "0")
#undef SRC
#define SRC d
      UNIT_A(d3, d, "0")
#undef SRC
    } else if (MODE == 5) {
#define SRC d2
      UNIT_H(d, d2)
#undef SRC
#define SRC d3
      UNIT_H(d2, d3)
#undef SRC
#define SRC d
      UNIT_H(d3, d)
#undef SRC
    } else if (MODE == 6) {
#define SRC d2
      UNIT_S(d, d2)
#undef SRC
#define SRC d3
      UNIT_S(d2, d3)
#undef SRC
#define SRC d
      UNIT_S(d3, d)
#undef SRC
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += d[i] + d2[i] + d3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)bits + (float)bits2 + (float)scnt;
}
template <int MODE>
void run(const char* name, int blocks, int per_iter) {
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
  const double waves_per_simd = blocks / 256.0;
  printf("%-52s waves/SIMD %.0f  %8.3f ms  %6.1f ns per unit per SIMD\n", name, waves_per_simd, ms,
         ms * 1e6 / ((double)iters * per_iter * waves_per_simd));
  (void)hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 768, 1024}) {
    run<0>("2 dep MFMA", blocks, 1);
    run<1>("16 alignbit", blocks, 1);
    run<2>("K32 unit, rotating sets", blocks, 3);
    run<3>("K32 unit, two history words", blocks, 3);
    run<4>("K32 unit, first MFMA with C = 0", blocks, 3);
    run<5>("K32 unit with 8 alignbit", blocks, 3);
    run<6>("K32 unit, 8 alignbit + 8 cmp/s_bcnt1", blocks, 3);
  }
  return 0;
}
