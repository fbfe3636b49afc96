// Internal helpers shared by the translation units of libcorsair_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/corsair_hip.h"

namespace cs {

void set_error(const char* fmt, ...);

#define CS_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      cs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                    __LINE__);                                                          \
      return CS_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

#define CS_REQUIRE(cond, code, ...)   \
  do {                                \
    if (!(cond)) {                    \
      cs::set_error(__VA_ARGS__);     \
      return (code);                  \
    }                                 \
  } while (0)

#define CS_LAUNCH_CHECK() CS_HIP_CHECK(hipGetLastError())

// Size-class caching device allocator for library-owned scratch and map storage.
void* pool_alloc(size_t bytes);
void pool_free(void* p);
void pool_stats(unsigned long long out[3]);
void pool_use_stream(hipStream_t s);
// device -> host through the thread's page-locked staging buffer (runtime.hip); sync hands the bytes out
hipError_t download_async(void* host_dst, const void* dev_src, size_t bytes, hipStream_t s);
hipError_t download_sync(hipStream_t s);  // see runtime.hip: scratch is cached per (thread, stream)

template <typename T>
struct PoolBuf {
  T* p = nullptr;
  size_t n = 0;
  PoolBuf() = default;
  explicit PoolBuf(size_t count) { alloc(count); }
  PoolBuf(const PoolBuf&) = delete;
  PoolBuf& operator=(const PoolBuf&) = delete;
  ~PoolBuf() { release(); }
  bool alloc(size_t count) {
    release();
    n = count;
    p = static_cast<T*>(pool_alloc((count ? count : 1) * sizeof(T)));
    return p != nullptr;
  }
  void release() {
    if (p) pool_free(p);
    p = nullptr;
    n = 0;
  }
  T* take() {
    T* r = p;
    p = nullptr;
    n = 0;
    return r;
  }
};

// Profiling (cs_prof_*).
struct ProfScope {
  int id;
  hipStream_t stream;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // units: algorithmic work of the bracketed launch (FLOP for conv / ransac_eval / knn / ...)
  ProfScope(const char* name, hipStream_t s, double units = 0.0);
  ~ProfScope();
};

// adds work units to a family without bracketing a launch (the units of a launch enqueued earlier)
void prof_add_units(const char* name, double units);
// the same for units whose launch was bracketed in profile epoch `epoch` (cs_prof_reset starts a new one): dropped when the
// profile has been reset since, accepted also after cs_prof_enable(0) (the region they belong to has only been CLOSED)
void prof_add_units_epoch(const char* name, double units, uint64_t epoch);
uint64_t prof_epoch();
// kernel maps that still hold deferred convolution work units hand them over (cs_prof_enable(0), cs_prof_get_units)
void kernelmap_flush_prof();
// A second, lowest-priority stream of the calling thread (created on first use, lives as long as the
// process) for work that may overlap the caller's stream; nullptr if it cannot be created.
hipStream_t side_stream(int which = 0);
void pool_defer_begin();
void pool_defer_end();

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- coordinate packing ---------------------------------------------------------------
// key = batch(16) | x+32768 (16) | y+32768 (16) | z+32768 (16)
__host__ __device__ static inline uint64_t pack_key(int b, int x, int y, int z) {
  return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)(x + 32768) << 32) |
         ((uint64_t)(uint16_t)(y + 32768) << 16) | (uint64_t)(uint16_t)(z + 32768);
}
__host__ __device__ static inline bool coord_in_range(int b, int x, int y, int z) {
  return b >= 0 && b < 65536 && x > -32768 && x < 32768 && y > -32768 && y < 32768 &&
         z > -32768 && z < 32768;
}
__host__ __device__ static inline uint64_t hash64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}
static constexpr uint64_t kEmptyKey = ~0ULL;

// LDS-DMA of 64 x 16 B (global_load_lds_dwordx4): lane l's 16 bytes at g land at LDS byte address
// lds_addr + 16 l (lds_addr wave-uniform, in an SGPR).  Written as inline asm ON PURPOSE: behind the
// __builtin_amdgcn_global_load_lds form hipcc (ROCm 7.2) treats the DMA as a pending LDS write that may
// alias every later ds_read and puts `s_waitcnt vmcnt(0)` in front of the first one -- a stage issued
// "one chunk ahead" is then waited for before the current chunk is computed.  The asm form is invisible
// to that pass: the caller orders the data itself (s_waitcnt vmcnt(0) by every issuing wave, then a
// workgroup barrier, before anyone reads the staged bytes).  No "memory" clobber: ordinary LDS reads of
// OTHER bytes may be scheduled across the DMA (that is the point of issuing it early); volatile asm keeps
// its order against the barrier / s_waitcnt statements, which is all the protocol needs.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);   // wave-uniform by contract; the "s" operand needs an SGPR
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr)
               : "m0");
#endif
}
#pragma clang diagnostic pop
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
#else
  return 0;
#endif
}

}  // namespace cs

struct cs_coordmap {
  int64_t n = 0;
  int tensor_stride = 1;
  int32_t* d_coords = nullptr;   // [n,4]
  uint64_t* d_keys = nullptr;    // [capacity]
  int32_t* d_vals = nullptr;     // [capacity]
  uint64_t capacity = 0;         // power of two
  // per-sample row segments (rows grouped by batch index), filled lazily for the LDS kernel-map path
  int32_t* d_seg = nullptr;      // [n_batch + 1]
  int n_batch = 0;
  int max_seg = 0;
  int seg_state = 0;             // 0 unknown, 1 available, -1 rows are not grouped by sample
};

struct cs_kernelmap {
  int64_t n_out = 0, n_in = 0;
  int kvol = 27;
  int transposed = 0;
  int32_t* d_nbr = nullptr;      // [n_out, kvol]
  int32_t* d_rowlist = nullptr;  // [n_out] output rows ordered by neighbour-presence mask (tiling order)
  // the offsets present in every group of 32 consecutive rows of the tiling order (padded with zeros to a multiple of 8
  // groups); the convolution reads `d_nbr` through `d_rowlist` (round 5: no copy of the table in tiling order)
  uint32_t* d_gmask = nullptr;       // [ceil(n_out / 32) rounded up to 8]
  // Pair count: written by the build kernels, copied to a page-locked slot behind them; resolved on
  // first use (kernelmap_pairs) so that building a map does not stall the host.
  int64_t num_pairs = -1;
  hipEvent_t cnt_ready = nullptr;
  unsigned long long* h_cnt = nullptr;
  int cnt_slot = -1;
  // profiling only: FLOP per pair of the convolutions that ran on this map while its pair count was still on the way
  // (2 Cin Cout each); turned into work units when the count is known (kernelmap_pairs / cs_kernelmap_free) instead of
  // stalling the host in cs_conv_fwd
  double prof_flop_per_pair = 0.0;
  uint64_t prof_epoch = 0;   // the profile epoch those deferred units were collected in (stale ones are dropped)
};
namespace cs {
int64_t kernelmap_pairs(const cs_kernelmap* km);
void kernelmap_defer_prof(cs_kernelmap* km, double per_pair);
bool prof_enabled();
}
