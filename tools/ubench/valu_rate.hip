// VALU issue-rate probe for the sign-accumulation idioms considered for k_ransac_prefilter.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; prints cycles per wave-instruction per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, float thr) {
  unsigned a = threadIdx.x, b = threadIdx.x * 3u, c = 5u, d = 7u;
  float f0 = (float)threadIdx.x, f1 = f0 * 0.5f;
  int cnt = 0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // dependent alignbit chain
      REP16(asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a) : "v"(b));)
    } else if (MODE == 1) {  // 4 independent alignbit chains
      REP16(asm volatile("v_alignbit_b32 %0, %0, %4, 31\n\tv_alignbit_b32 %1, %1, %4, 31\n\t"
                         "v_alignbit_b32 %2, %2, %4, 31\n\tv_alignbit_b32 %3, %3, %4, 31"
                         : "+v"(a), "+v"(c), "+v"(d), "+v"(cnt) : "v"(b));)
    } else if (MODE == 2) {  // dependent add chain
      REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
    } else if (MODE == 3) {  // cmp + addc
      REP16(asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(f0), "v"(thr) : "vcc");)
    } else if (MODE == 4) {  // lshl_or with pre-shifted sign: v_lshrrev + v_lshl_or
      REP16(asm volatile("v_lshrrev_b32 %1, 31, %2\n\tv_lshl_or_b32 %0, %0, 1, %1" : "+v"(a), "+v"(c) : "v"(b));)
    } else if (MODE == 5) {  // v_cmp to sgpr pair + s_bcnt1 x2 + s_add x2
      REP16(asm volatile("v_cmp_lt_f32 s[20:21], %1, %2\n\ts_bcnt1_i32_b32 s22, s20\n\ts_bcnt1_i32_b32 s23, s21\n\t"
                         "s_add_u32 s24, s24, s22\n\ts_add_u32 s25, s25, s23" : "+v"(cnt) : "v"(f0), "v"(thr)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "scc");)
    } else if (MODE == 6) {  // v_bfi dependent chain
      REP16(asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));)
    } else if (MODE == 7) {  // v_and_or dependent chain
      REP16(asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));)
    } else if (MODE == 8) {  // v_lshl_add_u32 dependent
      REP16(asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b));)
    } else if (MODE == 9) {  // v_alignbit independent outputs (no chain), 16 distinct
      REP16(asm volatile("v_alignbit_b32 %0, %1, %2, 31" : "=v"(a) : "v"(b), "v"(c));)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + c + d + cnt + (unsigned)f1;
}
template <int MODE>
void run(const char* name, int per_rep) {
  unsigned* out;
  hipMalloc(&out, 1024 * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<1024, 256>>>(out, 100, 1.0f);
  hipEventRecord(e0);
  k<MODE><<<1024, 256>>>(out, iters, 1.0f);  // 1024 blocks of 4 waves: 4 blocks per CU = 4 waves per SIMD
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x iters x 16 x per_rep instructions
  double inst = 4.0 * iters * 16.0 * per_rep;
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-34s %8.3f ms  %.2f cycles per wave-instruction (4 waves/SIMD, 2.4 GHz assumed)\n", name, ms, cyc / inst);
  hipFree(out);
}
int main() {
  run<0>("alignbit dependent", 1);
  run<1>("alignbit 4 chains", 4);
  run<9>("alignbit no chain", 1);
  run<2>("v_add_u32 dependent", 1);
  run<3>("v_cmp + v_addc", 2);
  run<4>("v_lshrrev + v_lshl_or", 2);
  run<5>("v_cmp sgpr + 2 bcnt + 2 s_add", 5);
  run<6>("v_bfi dependent", 1);
  run<7>("v_and_or dependent", 1);
  run<8>("v_lshl_add dependent", 1);
  return 0;
}
