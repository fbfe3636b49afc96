// Does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` write zeros to LDS (like a VGPR
// destination would receive zeros) or leave the LDS bytes untouched?  Decides how k_conv_dma stages absent rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int int4v __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* out, int nbytes) {
  __shared__ __attribute__((aligned(128))) float lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -7.0f;
  __syncthreads();
  int4v srd;
  srd.x = __builtin_amdgcn_readfirstlane((int)(uintptr_t)in);
  srd.y = __builtin_amdgcn_readfirstlane((int)((uintptr_t)in >> 32));
  srd.z = __builtin_amdgcn_readfirstlane(nbytes);
  srd.w = __builtin_amdgcn_readfirstlane(0x00020000);
  // even lanes read their 16 bytes, odd lanes are 2 GiB out of range
  unsigned voff = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16;
  int soff = 0;
  unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds);
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(l) : "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  float *in, *out, h[256], hin[256];
  for (int i = 0; i < 256; ++i) hin[i] = i + 1;
  hipMalloc(&in, 1024); hipMalloc(&out, 1024);
  hipMemcpy(in, hin, 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(in, out, 1024);
  hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 6; ++l) printf("lane %d: %g %g %g %g\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  int zeros = 0, stale = 0, ok = 0;
  for (int l = 0; l < 64; ++l) {
    if (l & 1) { zeros += h[4 * l] == 0.f; stale += h[4 * l] == -7.f; } else ok += h[4 * l] == 4 * l + 1;
  }
  printf("in-range lanes correct %d/32, out-of-range lanes: zero %d, untouched %d\n", ok, zeros, stale);
  return 0;
}
