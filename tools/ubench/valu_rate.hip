// VALU issue-rate probe for the sign-accumulation idioms considered for k_ransac_prefilter.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; prints cycles per wave-instruction per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, float thr, unsigned long long* clk) {
  unsigned a = threadIdx.x, b = threadIdx.x * 3u, c = 5u, d = 7u;
  float f0 = (float)threadIdx.x, f1 = f0 * 0.5f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 p0 = {f0, f1}, p1 = {f1, f0}, p2 = {thr, thr};
  int cnt = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
    } else if (MODE == 10) {  // fma dependent
      REP16(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(f1));)
    } else if (MODE == 11) {  // v_perm_b32 dependent, literal selector
      REP16(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(0x0b070c0cu));)
    } else if (MODE == 12) {  // v_bcnt accumulate
      REP16(asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b));)
    } else if (MODE == 13) {  // v_xor VOP2 dependent
      REP16(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));)
    } else if (MODE == 14) {  // alignbit with SGPR shift operand
      REP16(asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(31));)
    } else if (MODE == 15) {  // packed f32 multiply with clamp (2 results per instruction)
      REP16(asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(p0) : "v"(p1), "v"(p2));)
    } else if (MODE == 16) {  // packed f32 add, dependent
      REP16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(p1));)
    } else if (MODE == 17) {  // v_med3_f32
      REP16(asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(f0) : "v"(f1));)
    } else if (MODE == 18) {  // v_dot4c_i32_i8 accumulate
      REP16(asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(cnt) : "v"(b), "v"(c));)
    } else if (MODE == 19) {  // v_mul_f32 clamp (VOP3) independent
      REP16(asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(f0) : "v"(f1), "v"(thr));)
    } else if (MODE == 20) {  // v_pk_fma_f32 dependent
      REP16(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p2));)
    } else if (MODE == 21) {  // v_add_f32 dependent
      REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));)
    } else if (MODE == 22) {  // v_pk_add_u16
      REP16(asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));)
    } else if (MODE == 23) {  // v_subrev/ v_sub_u32 with ashr: two fast ops
      REP16(asm volatile("v_ashrrev_i32 %1, 31, %2\n\tv_sub_u32 %0, %0, %1" : "+v"(a), "+v"(c) : "v"(b));)
    } else if (MODE == 9) {  // v_alignbit independent outputs (no chain), 16 distinct
      REP16(asm volatile("v_alignbit_b32 %0, %1, %2, 31" : "=v"(a) : "v"(b), "v"(c));)
    }
  }
  asm volatile("" : "+v"(a), "+v"(c), "+v"(d), "+v"(cnt));
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    clk[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = c1 - c0;
    clk[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + c + d + cnt + (unsigned)f1 + (unsigned)f0 + (unsigned)p0[0] + (unsigned)p0[1];
}
template <int MODE>
void run(const char* name, int per_rep) {
  unsigned* out;
  hipMalloc(&out, 1024 * 256 * 4);
  unsigned long long* clk;
  hipMalloc(&clk, 1024 * 4 * 2 * 8);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<1024, 256>>>(out, 100, 1.0f, clk);
  hipEventRecord(e0);
  k<MODE><<<1024, 256>>>(out, iters, 1.0f, clk);  // 1024 blocks of 4 waves: 4 blocks per CU = 4 waves per SIMD
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x iters x 16 x per_rep instructions
  double inst = 4.0 * iters * 16.0 * per_rep;
  static unsigned long long h[1024 * 4 * 2];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double cs = 0, rs = 0;
  for (int i = 0; i < 1024 * 4; ++i) { cs += (double)h[2 * i]; rs += (double)h[2 * i + 1]; }
  const double ghz = cs / rs * 0.1;  // s_memrealtime ticks at 100 MHz
  double cyc = ms * 1e-3 * ghz * 1e9;
  printf("%-34s %8.3f ms  clock %.3f GHz  %.2f cycles per wave-instruction per SIMD (4 waves/SIMD)\n", name, ms, ghz, cyc / inst);
  hipFree(clk);
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
  run<10>("v_fma_f32 dependent", 1);
  run<11>("v_perm_b32 dependent", 1);
  run<12>("v_bcnt_u32_b32 accumulate", 1);
  run<13>("v_xor_b32 dependent", 1);
  run<14>("alignbit, sgpr shift", 1);
  run<15>("v_pk_mul_f32 clamp", 1);
  run<16>("v_pk_add_f32 dependent", 1);
  run<20>("v_pk_fma_f32 dependent", 1);
  run<17>("v_med3_f32 dependent", 1);
  run<18>("v_dot4c_i32_i8 accumulate", 1);
  run<19>("v_mul_f32 clamp (VOP3)", 1);
  run<21>("v_add_f32 dependent", 1);
  run<22>("v_pk_add_u16 dependent", 1);
  run<23>("v_ashrrev + v_sub_u32", 2);
  return 0;
}
