// Sign counting of the K = 16 prefilter unit with the gfx950 block-scale conversions (microbenchmark).
// Shipped unit: 1 v_mfma_f32_32x32x16_f16 + 16 v_add_f32 (one VALU op per result; on gfx950 those ops do not overlap the
// matrix pipe).  Here the sign bits of the results are PACKED by a conversion whose magnitudes underflow to zero
// (scale 2^127: every |v| < 2^40 becomes +-0), so the packed word holds nothing but sign bits and v_bcnt_u32_b32 counts them:
//   fp4:  v_cvt_scalef32_pk_fp4_f32 packs 2 results per op into one byte -> 8 cvt + 2 bcnt per 16 results (0.625 op / result)
//   fp6:  v_cvt_scalef32_2xpk16_fp6_f32 packs the 32 results of TWO tiles in one op -> 1 cvt + 6 bcnt per 32 results
// Part 1 checks the bit patterns on edge values (-0, denormals, tiny and huge negatives, +0, positives); part 2 times
// the units.  build: hipcc --offload-arch=gfx950 -O3 -o pf_k16_cvtsign pf_k16_cvtsign.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x6 = __attribute__((ext_vector_type(6))) unsigned;

// ---- part 1: semantics -------------------------------------------------------------------------
__global__ void k_check(const float* in /* [64][32] */, unsigned scale_bits, unsigned* out6 /* [64][6] */, unsigned* out4 /* [64][4] */) {
  const int l = threadIdx.x;
  f32x16 a, b;
  for (int i = 0; i < 16; ++i) { a[i] = in[l * 32 + i]; b[i] = in[l * 32 + 16 + i]; }
  float sc = __uint_as_float(scale_bits);
  u32x6 d;
  asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(sc));
  for (int i = 0; i < 6; ++i) out6[l * 6 + i] = d[i];
  unsigned p[4] = {0, 0, 0, 0};
  // fp4: byte j of dword w <- results (8 w + 2 j, 8 w + 2 j + 1)
#define CVT4(W, J, X, Y) asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0," #J "]" : "+v"(p[W]) : "v"(X), "v"(Y), "v"(sc));
  // op_sel[3:2] selects the destination byte: written as the two high op_sel bits
  asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "+v"(p[0]) : "v"(a[0]), "v"(a[1]), "v"(sc));
  asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,1,0]" : "+v"(p[0]) : "v"(a[2]), "v"(a[3]), "v"(sc));
  asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(p[0]) : "v"(a[4]), "v"(a[5]), "v"(sc));
  asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,1,1]" : "+v"(p[0]) : "v"(a[6]), "v"(a[7]), "v"(sc));
  asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "+v"(p[1]) : "v"(a[8]), "v"(a[9]), "v"(sc));
  asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,1,0]" : "+v"(p[1]) : "v"(a[10]), "v"(a[11]), "v"(sc));
  asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(p[1]) : "v"(a[12]), "v"(a[13]), "v"(sc));
  asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,1,1]" : "+v"(p[1]) : "v"(a[14]), "v"(a[15]), "v"(sc));
  for (int i = 0; i < 4; ++i) out4[l * 4 + i] = p[i];
}

// ---- part 2: timing ------------------------------------------------------------------------------
// variant 0: shipped unit x 2 (2 MFMA + 32 v_add_f32 under round-toward-minus-infinity)
// variant 1: 2 MFMA + 1 v_cvt_scalef32_2xpk16_fp6_f32 + 6 v_bcnt_u32_b32
// variant 2: 2 MFMA + 16 v_cvt_scalef32_pk_fp4_f32 + 4 v_bcnt_u32_b32
// variant 3: 2 MFMA only;  4: the fp6 cvt + 6 bcnt only;  5: 16 fp4 cvt + 4 bcnt only; 6: 32 v_add only
template <int V>
__global__ __launch_bounds__(256) void k_time(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.0001f + 0.01f * i); b[i] = (_Float16)(0.05f * i); }
  f32x16 c, d0, d1, e0, e1;
  for (int i = 0; i < 16; ++i) { c[i] = -0.25f; d0[i] = 0.f; d1[i] = 0.f; e0[i] = -1.f; e1[i] = 1.f; }
  float cntf = 16777215.0f;
  unsigned cnt = 0;
  u32x6 pk;
  unsigned p4[4] = {0, 0, 0, 0};
  float sc = __uint_as_float(0x7f000000u);
  if (V == 0 || V == 6) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2");
  // software pipeline as in the kernel: the MFMAs of a step write (W0, W1) while the VALU consumes (R0, R1), the tiles of
  // the step before; two steps per iteration with the roles swapped, so no register is ever copied
#define STEP(W0, W1, R0, R1)                                                                                                      \
  {                                                                                                                               \
    if (V <= 3)                                                                                                                   \
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %4"                             \
                   : "=&v"(W0), "=&v"(W1) : "v"(a), "v"(b), "v"(c));                                                              \
    if (V == 0 || V == 6) {                                                                                                       \
      _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                                              \
          asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0" : "+v"(cntf) : "v"(R0[i]), "v"(R1[i]));                     \
    }                                                                                                                             \
    if (V == 1 || V == 4) {                                                                                                       \
      asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=v"(pk) : "v"(R0), "v"(R1), "v"(sc));                        \
      asm volatile("v_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %0, %3, %0\n\t"                      \
                   "v_bcnt_u32_b32 %0, %4, %0\n\tv_bcnt_u32_b32 %0, %5, %0\n\tv_bcnt_u32_b32 %0, %6, %0"                          \
                   : "+v"(cnt) : "v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]), "v"(pk[4]), "v"(pk[5]));                         \
    }                                                                                                                             \
    if (V == 2 || V == 5) {                                                                                                       \
      _Pragma("unroll") for (int w = 0; w < 2; ++w) {                                                                             \
        asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "+v"(p4[w]) : "v"(R0[8 * w + 0]), "v"(R0[8 * w + 1]), "v"(sc)); \
        asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,1,0]" : "+v"(p4[w]) : "v"(R0[8 * w + 2]), "v"(R0[8 * w + 3]), "v"(sc)); \
        asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(p4[w]) : "v"(R0[8 * w + 4]), "v"(R0[8 * w + 5]), "v"(sc)); \
        asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,1,1]" : "+v"(p4[w]) : "v"(R0[8 * w + 6]), "v"(R0[8 * w + 7]), "v"(sc)); \
        asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "+v"(p4[2 + w]) : "v"(R1[8 * w + 0]), "v"(R1[8 * w + 1]), "v"(sc)); \
        asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,1,0]" : "+v"(p4[2 + w]) : "v"(R1[8 * w + 2]), "v"(R1[8 * w + 3]), "v"(sc)); \
        asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(p4[2 + w]) : "v"(R1[8 * w + 4]), "v"(R1[8 * w + 5]), "v"(sc)); \
        asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,1,1]" : "+v"(p4[2 + w]) : "v"(R1[8 * w + 6]), "v"(R1[8 * w + 7]), "v"(sc)); \
      }                                                                                                                           \
      asm volatile("v_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %0, %3, %0\n\tv_bcnt_u32_b32 %0, %4, %0" \
                   : "+v"(cnt) : "v"(p4[0]), "v"(p4[1]), "v"(p4[2]), "v"(p4[3]));                                                 \
    }                                                                                                                             \
  }
  for (int it = 0; it < iters; it += 2) {
    STEP(d0, d1, e0, e1)
    STEP(e0, e1, d0, d1)
    if ((it & 1023) == 1022) cntf = 16777215.0f;
  }
#undef STEP
  if (V == 0 || V == 6) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += d0[i] + d1[i] + e0[i] + e1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + cntf + (float)cnt;
}
template <int V>
void run(int blocks, const char* what) {
  float* out;
  (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k_time<V><<<blocks, 256>>>(out, 2000);
  (void)hipEventRecord(e0);
  k_time<V><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double w = blocks / 256.0;   // waves per SIMD (256 CUs x 4 SIMDs, 4 waves per workgroup)
  printf("%-66s waves/SIMD %.0f  %8.3f ms  %6.1f ns per UNIT (1 MFMA tile = 16 results per lane) per SIMD\n", what, w, ms,
         ms * 1e6 / ((double)iters * 2 * w));
  (void)hipFree(out);
}
int main() {
  // ---- semantics ----
  float h_in[64 * 32];
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 32; ++i) {
      const int sel = (l * 7 + i * 3) % 12;
      const float vals[12] = {-0.0f, 0.0f, -1e-45f, 1e-45f, -1.17549435e-38f, 1.0f, -1.0f, -1.0995e12f /* -2^40 */, 1.0995e12f,
                              -3.0e-39f, 5.0e11f, -7.25f};
      h_in[l * 32 + i] = vals[sel];
    }
  float* d_in;
  unsigned *d6, *d4;
  (void)hipMalloc(&d_in, sizeof(h_in));
  (void)hipMalloc(&d6, 64 * 6 * 4);
  (void)hipMalloc(&d4, 64 * 4 * 4);
  (void)hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  for (unsigned scale_bits : {0x7f000000u, 0x7e800000u, 0x00800000u, 0x3f800000u}) {
    k_check<<<1, 64>>>(d_in, scale_bits, d6, d4);
    unsigned h6[64 * 6], h4[64 * 4];
    (void)hipMemcpy(h6, d6, sizeof(h6), hipMemcpyDeviceToHost);
    (void)hipMemcpy(h4, d4, sizeof(h4), hipMemcpyDeviceToHost);
    int bad6 = 0, bad4 = 0, nonsign6 = 0, nonsign4 = 0;
    for (int l = 0; l < 64; ++l) {
      int neg32 = 0, neg16 = 0;
      for (int i = 0; i < 32; ++i) {
        unsigned u;
        memcpy(&u, &h_in[l * 32 + i], 4);
        neg32 += u >> 31;
        if (i < 16) neg16 += u >> 31;
      }
      int pc6 = 0, pc4 = 0;
      for (int i = 0; i < 6; ++i) pc6 += __builtin_popcount(h6[l * 6 + i]);
      for (int i = 0; i < 2; ++i) pc4 += __builtin_popcount(h4[l * 4 + i]);
      bad6 += pc6 != neg32;
      bad4 += pc4 != neg16;
      // are all set bits at sign positions?  fp6: bit 5 of every 6-bit field; fp4: bit 3 of every nibble
      unsigned long long lo = (unsigned long long)h6[l * 6] | ((unsigned long long)h6[l * 6 + 1] << 32);
      (void)lo;
      for (int f = 0; f < 32; ++f) {
        const int bit0 = f * 6;
        unsigned field = 0;
        for (int bb = 0; bb < 6; ++bb) field |= ((h6[l * 6 + (bit0 + bb) / 32] >> ((bit0 + bb) % 32)) & 1u) << bb;
        nonsign6 += (field & 0x1f) != 0;
      }
      for (int i = 0; i < 2; ++i) nonsign4 += (h4[l * 4 + i] & 0x77777777u) != 0;
    }
    printf("scale bits %08x: fp6 lanes whose popcount != number of negative inputs: %d of 64 (fields with magnitude bits: %d); "
           "fp4: %d of 64 (words with magnitude bits: %d)\n", scale_bits, bad6, nonsign6, bad4, nonsign4);
    if (scale_bits == 0x7f000000u)
      printf("  lane 0 fp6 words %08x %08x %08x %08x %08x %08x, fp4 words %08x %08x\n", h6[0], h6[1], h6[2], h6[3], h6[4], h6[5], h4[0], h4[1]);
  }
  // ---- timing ----
  for (int blocks : {1024, 1280}) {
    run<0>(blocks, "shipped: 1 MFMA + 16 v_add_f32 (RTN)");
    run<1>(blocks, "fp6: 2 MFMA + 1 v_cvt_scalef32_2xpk16_fp6_f32 + 6 v_bcnt");
    run<2>(blocks, "fp4: 2 MFMA + 16 v_cvt_scalef32_pk_fp4_f32 + 4 v_bcnt");
    run<3>(blocks, "MFMA alone");
    run<4>(blocks, "1 v_cvt_scalef32_2xpk16_fp6_f32 + 6 v_bcnt alone (per 2 units)");
    run<5>(blocks, "16 v_cvt_scalef32_pk_fp4_f32 + 4 v_bcnt alone (per 2 units)");
    run<6>(blocks, "32 v_add_f32 alone (per 2 units)");
  }
  return 0;
}
