// GPU voxel quantisation: replaces ME.utils.sparse_quantize(floor(xyz / voxel),
// return_index=True, return_maps_only=True) + the batch-index prepend of ME.utils.sparse_collate
// (utils/Info/CADLib.py:106-121,148-178; datasets/CategoryDataset.py:179-197).
// Same hash-insert machinery as the strided coordinate map: first point of every voxel wins,
// kept indices ascending.  Grid index = floor(x / voxel) evaluated in the cloud's OWN type, as NumPy
// does: f32 clouds (the catalog side, utils/Info/CADLib.py:106-121: `f32 array / python float` is an f32
// division by f32(voxel)) through cs_voxelize, f64 clouds (the query side: datasets/CategoryDataset.py:
// 179-197 floors the f64 output of apply_transform, evaluation-shapenet.py:97-119 the f64 posed cloud;
// the cast to f32 comes AFTER the selection there) through cs_voxelize_f64 with an IEEE f64 division.
#include <hipcub/hipcub.hpp>

#include <vector>

#include "common.h"

namespace cs {

// floor(x / vs) as an int, saturated so that a huge or non-finite quotient fails the range check
// instead of wrapping in the conversion.
__device__ __forceinline__ int vox_cell(float x, float vs) {
  const float q = floorf(x / vs);
  return q >= -65536.f && q <= 65536.f ? (int)q : 0x7fffffff;
}
__device__ __forceinline__ int vox_cell(double x, double vs) {
  const double q = floor(x / vs);
  return q >= -65536.0 && q <= 65536.0 ? (int)q : 0x7fffffff;
}

template <typename T>
__device__ __forceinline__ bool vox_key(const T* __restrict__ xyz, int64_t i, int seg, T vs,
                                        uint64_t* key, int* g) {
  g[0] = vox_cell(xyz[3 * i + 0], vs);
  g[1] = vox_cell(xyz[3 * i + 1], vs);
  g[2] = vox_cell(xyz[3 * i + 2], vs);
  if (g[0] == 0x7fffffff || g[1] == 0x7fffffff || g[2] == 0x7fffffff) return false;
  if (!coord_in_range(seg, g[0], g[1], g[2])) return false;
  *key = pack_key(seg, g[0], g[1], g[2]);
  return true;
}

__global__ void k_vox_fill(uint64_t* keys, int32_t* vals, uint64_t cap) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < cap; i += stride) {
    keys[i] = kEmptyKey;
    vals[i] = 0x7fffffff;
  }
}

template <typename T>
__global__ void k_vox_insert(const T* __restrict__ xyz, const int32_t* __restrict__ seg_of,
                             int64_t n, T vs, uint64_t* keys, int32_t* vals, uint64_t mask,
                             int* status) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key;
  int g[3];
  if (!vox_key(xyz, i, seg_of[i], vs, &key, g)) {
    atomicOr(status, 1);
    return;
  }
  uint64_t slot = hash64(key) & mask;
  while (true) {
    unsigned long long old = atomicCAS((unsigned long long*)&keys[slot],
                                       (unsigned long long)kEmptyKey, (unsigned long long)key);
    if (old == kEmptyKey || old == key) {
      atomicMin(&vals[slot], (int32_t)i);
      return;
    }
    slot = (slot + 1) & mask;
  }
}

template <typename T>
__global__ void k_vox_flag(const T* __restrict__ xyz, const int32_t* __restrict__ seg_of,
                           int64_t n, T vs, const uint64_t* __restrict__ keys,
                           const int32_t* __restrict__ vals, uint64_t mask, int32_t* flag) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key;
  int g[3];
  int f = 0;
  if (vox_key(xyz, i, seg_of[i], vs, &key, g)) {
    uint64_t slot = hash64(key) & mask;
    while (keys[slot] != key) slot = (slot + 1) & mask;
    f = vals[slot] == (int32_t)i;
  }
  flag[i] = f;
}

template <typename T>
__global__ void k_vox_emit(const T* __restrict__ xyz, const int32_t* __restrict__ seg_of,
                           int64_t n, T vs, const int32_t* __restrict__ flag,
                           const int32_t* __restrict__ pos, int64_t* keep, int32_t* grid) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const int32_t o = pos[i];
  keep[o] = i;
  grid[4 * o + 0] = seg_of[i];
  grid[4 * o + 1] = vox_cell(xyz[3 * i + 0], vs);
  grid[4 * o + 2] = vox_cell(xyz[3 * i + 1], vs);
  grid[4 * o + 3] = vox_cell(xyz[3 * i + 2], vs);
}

__global__ void k_vox_segment_ids(const int64_t* __restrict__ off, int n_seg, int32_t* seg_of) {
  const int s = blockIdx.y;
  const int64_t b = off[s], e = off[s + 1];
  for (int64_t i = b + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < e;
       i += (int64_t)gridDim.x * blockDim.x)
    seg_of[i] = s;
}

__global__ void k_vox_offsets(const int64_t* __restrict__ off, int n_seg, int64_t n,
                              const int32_t* __restrict__ pos, const int32_t* __restrict__ flag,
                              int64_t* out_off) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > n_seg) return;
  const int64_t i = off[s];
  out_off[s] = i < n ? (int64_t)pos[i] : (n > 0 ? (int64_t)pos[n - 1] + flag[n - 1] : 0);
}

template <typename T>
static int voxelize_impl(const T* d_xyz, const int64_t* h_offsets, int n_seg, double voxel_size,
                         int64_t* d_keep_idx, int32_t* d_grid, int64_t* h_out_offsets, void* stream) {
  CS_REQUIRE(d_xyz && h_offsets && d_keep_idx && d_grid && h_out_offsets, CS_ERR_INVALID,
             "cs_voxelize: NULL argument");
  CS_REQUIRE(n_seg >= 1 && n_seg < 65536, CS_ERR_INVALID, "cs_voxelize: bad segment count");
  CS_REQUIRE(voxel_size > 0.0, CS_ERR_INVALID, "cs_voxelize: voxel size must be positive");
  CS_REQUIRE(h_offsets[0] == 0, CS_ERR_INVALID, "cs_voxelize: offsets must start at 0");
  const int64_t n = h_offsets[n_seg];
  int64_t seg_max = 0;
  for (int s = 0; s < n_seg; ++s) {
    CS_REQUIRE(h_offsets[s + 1] >= h_offsets[s], CS_ERR_INVALID, "cs_voxelize: bad offsets");
    if (h_offsets[s + 1] - h_offsets[s] > seg_max) seg_max = h_offsets[s + 1] - h_offsets[s];
  }
  CS_REQUIRE(n < (1LL << 30), CS_ERR_UNSUPPORTED, "cs_voxelize: too many points");
  for (int s = 0; s <= n_seg; ++s) h_out_offsets[s] = 0;
  if (n == 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  const T vs = (T)voxel_size;
  uint64_t cap = 1024;
  while (cap < (uint64_t)(2 * n)) cap <<= 1;
  PoolBuf<uint64_t> keys(cap);
  PoolBuf<int32_t> vals(cap), seg_of(n), flag(n), pos(n);
  PoolBuf<int64_t> d_off(n_seg + 1), d_out_off(n_seg + 1);
  PoolBuf<int> status(1);
  CS_REQUIRE(keys.p && vals.p && seg_of.p && flag.p && pos.p && d_off.p && d_out_off.p && status.p,
             CS_ERR_HIP, "cs_voxelize: scratch allocation failed");
  CS_HIP_CHECK(hipMemcpyAsync(d_off.p, h_offsets, sizeof(int64_t) * (n_seg + 1),
                              hipMemcpyHostToDevice, s));
  CS_HIP_CHECK(hipMemsetAsync(status.p, 0, sizeof(int), s));
  const unsigned g = (unsigned)ceil_div(n, 256);
  hipLaunchKernelGGL(k_vox_fill, dim3((unsigned)(cap / 256 < 2048 ? cap / 256 : 2048)), dim3(256),
                     0, s, keys.p, vals.p, cap);
  hipLaunchKernelGGL(k_vox_segment_ids,
                     dim3((unsigned)(ceil_div(seg_max > 0 ? seg_max : 1, 256)), (unsigned)n_seg),
                     dim3(256), 0, s, d_off.p, n_seg, seg_of.p);
  hipLaunchKernelGGL(k_vox_insert<T>, dim3(g), dim3(256), 0, s, d_xyz, seg_of.p, n, vs, keys.p,
                     vals.p, cap - 1, status.p);
  hipLaunchKernelGGL(k_vox_flag<T>, dim3(g), dim3(256), 0, s, d_xyz, seg_of.p, n, vs, keys.p, vals.p,
                     cap - 1, flag.p);
  CS_LAUNCH_CHECK();
  size_t tmp_bytes = 0;
  CS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, flag.p, pos.p, (int)n, s));
  PoolBuf<char> tmp(tmp_bytes);
  CS_REQUIRE(tmp.p, CS_ERR_HIP, "cs_voxelize: scan scratch allocation failed");
  CS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, flag.p, pos.p, (int)n, s));
  hipLaunchKernelGGL(k_vox_emit<T>, dim3(g), dim3(256), 0, s, d_xyz, seg_of.p, n, vs, flag.p, pos.p,
                     d_keep_idx, d_grid);
  hipLaunchKernelGGL(k_vox_offsets, dim3((unsigned)ceil_div(n_seg + 1, 256)), dim3(256), 0, s,
                     d_off.p, n_seg, n, pos.p, flag.p, d_out_off.p);
  CS_LAUNCH_CHECK();
  int h_status = 0;
  CS_HIP_CHECK(download_async(&h_status, status.p, sizeof(int), s));
  CS_HIP_CHECK(download_async(h_out_offsets, d_out_off.p, sizeof(int64_t) * (n_seg + 1), s));
  CS_HIP_CHECK(download_sync(s));
  CS_REQUIRE(!h_status, CS_ERR_RANGE,
             "cs_voxelize: voxel index out of the supported range (|index| < 32768)");
  return CS_OK;
}

}  // namespace cs

using namespace cs;

extern "C" {

int cs_voxelize(const float* d_xyz, const int64_t* h_offsets, int n_seg, double voxel_size,
                int64_t* d_keep_idx, int32_t* d_grid, int64_t* h_out_offsets, void* stream) {
  return voxelize_impl<float>(d_xyz, h_offsets, n_seg, voxel_size, d_keep_idx, d_grid, h_out_offsets,
                              stream);
}

int cs_voxelize_f64(const double* d_xyz, const int64_t* h_offsets, int n_seg, double voxel_size,
                    int64_t* d_keep_idx, int32_t* d_grid, int64_t* h_out_offsets, void* stream) {
  return voxelize_impl<double>(d_xyz, h_offsets, n_seg, voxel_size, d_keep_idx, d_grid, h_out_offsets,
                               stream);
}

}  // extern "C"
