// Coordinate maps (hash-indexed COO coordinates) and output-stationary kernel maps.
//
// Replaces the coordinate manager of MinkowskiEngine as used by the reference
// (evaluation.py:215-218; model/resunet.py:49-193).  Data layout in HBM:
//   coords  int32 [n,4]  (batch, x, y, z), row order = tensor row order
//   keys    uint64 [cap] open-addressing table (linear probing), cap = pow2 >= 2n, empty = ~0
//   vals    int32  [cap] row index of the key
//   nbr     int32  [n_out, 27] neighbour table: in-row feeding out-row o through offset k, or -1
// All of it is integer work bound by HBM/L2 latency, not MFMA; the tables of a batch-32 eval
// batch (145k voxels -> 4.7 MB of table) stay L2/Infinity-Cache resident between the insert
// and the 27 probes per output row.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace cs {

__device__ __forceinline__ int32_t hash_lookup(const uint64_t* __restrict__ keys,
                                               const int32_t* __restrict__ vals, uint64_t mask,
                                               uint64_t key) {
  uint64_t slot = hash64(key) & mask;
  while (true) {
    uint64_t k = keys[slot];
    if (k == key) return vals[slot];
    if (k == kEmptyKey) return -1;
    slot = (slot + 1) & mask;
  }
}

// floor division for possibly negative a, b > 0
__device__ __forceinline__ int floor_div(int a, int b) {
  int q = a / b;
  int r = a % b;
  return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

__global__ void k_fill_table(uint64_t* keys, int32_t* vals, uint64_t cap) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < cap; i += stride) {
    keys[i] = kEmptyKey;
    vals[i] = 0x7fffffff;
  }
}

// Insert key(row) with value = min row index.  stride_mul > 0: key of the coarse cell.
// status[0] |= 1 on out-of-range coordinate; status[1] counts duplicate keys.
__global__ void k_insert(const int32_t* __restrict__ coords, int64_t n, int cell, uint64_t* keys,
                         int32_t* vals, uint64_t mask, int* status) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b = coords[4 * i + 0], x = coords[4 * i + 1], y = coords[4 * i + 2], z = coords[4 * i + 3];
  if (cell > 1) {
    x = floor_div(x, cell) * cell;
    y = floor_div(y, cell) * cell;
    z = floor_div(z, cell) * cell;
  }
  if (!coord_in_range(b, x, y, z)) {
    atomicOr(&status[0], 1);
    return;
  }
  uint64_t key = pack_key(b, x, y, z);
  uint64_t slot = hash64(key) & mask;
  while (true) {
    unsigned long long old = atomicCAS((unsigned long long*)&keys[slot],
                                       (unsigned long long)kEmptyKey, (unsigned long long)key);
    if (old == kEmptyKey || old == key) {
      if (old == key) atomicAdd(&status[1], 1);
      atomicMin(&vals[slot], (int32_t)i);
      return;
    }
    slot = (slot + 1) & mask;
  }
}

// flag[i] = 1 iff row i is the first (minimum) row of its coarse cell.
__global__ void k_flag_first(const int32_t* __restrict__ coords, int64_t n, int cell,
                             const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                             uint64_t mask, int32_t* flag) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b = coords[4 * i + 0];
  int x = floor_div(coords[4 * i + 1], cell) * cell;
  int y = floor_div(coords[4 * i + 2], cell) * cell;
  int z = floor_div(coords[4 * i + 3], cell) * cell;
  int32_t v = hash_lookup(keys, vals, mask, pack_key(b, x, y, z));
  flag[i] = (v == (int32_t)i) ? 1 : 0;
}

// Write coarse coordinates in first-occurrence order and re-point the table at the new rows.
__global__ void k_emit_strided(const int32_t* __restrict__ coords, int64_t n, int cell,
                               const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                               uint64_t* keys, int32_t* vals, uint64_t mask, int32_t* out_coords) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  int b = coords[4 * i + 0];
  int x = floor_div(coords[4 * i + 1], cell) * cell;
  int y = floor_div(coords[4 * i + 2], cell) * cell;
  int z = floor_div(coords[4 * i + 3], cell) * cell;
  int32_t o = pos[i];
  out_coords[4 * o + 0] = b;
  out_coords[4 * o + 1] = x;
  out_coords[4 * o + 2] = y;
  out_coords[4 * o + 3] = z;
  uint64_t key = pack_key(b, x, y, z);
  uint64_t slot = hash64(key) & mask;
  while (keys[slot] != key) slot = (slot + 1) & mask;
  vals[slot] = o;  // exactly one row per key reaches here
}

// One thread per (out row, k): probe the in-map.  Offsets: k = (dx+1) + 3(dy+1) + 9(dz+1).
__global__ void k_build_nbr(const int32_t* __restrict__ out_coords, int64_t n_out, int kvol,
                            int step, int sign, const uint64_t* __restrict__ keys,
                            const int32_t* __restrict__ vals, uint64_t mask, int32_t* nbr,
                            unsigned long long* pair_count) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int found = 0;
  if (t < n_out * kvol) {
    int64_t o = t / kvol;
    int k = (int)(t - o * kvol);
    int dx = 0, dy = 0, dz = 0;
    if (kvol == 27) {
      dx = k % 3 - 1;
      dy = (k / 3) % 3 - 1;
      dz = k / 9 - 1;
    }
    int b = out_coords[4 * o + 0];
    int x = out_coords[4 * o + 1] + sign * dx * step;
    int y = out_coords[4 * o + 2] + sign * dy * step;
    int z = out_coords[4 * o + 3] + sign * dz * step;
    int32_t v = -1;
    if (coord_in_range(b, x, y, z)) v = hash_lookup(keys, vals, mask, pack_key(b, x, y, z));
    nbr[t] = v;
    found = v >= 0;
  }
  // wave-level count then one atomic per wave
  unsigned long long m = __ballot(found);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(pair_count, (unsigned long long)__popcll(m));
}

// 27-bit neighbour-presence mask of every output row + identity row ids (sorted by mask afterwards)
__global__ void k_row_masks(const int32_t* __restrict__ nbr, int64_t n_out, int kvol,
                            uint32_t* __restrict__ mask, int32_t* __restrict__ rows) {
  int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  uint32_t m = 0;
  for (int k = 0; k < kvol; ++k) m |= (nbr[o * kvol + k] >= 0 ? 1u : 0u) << k;
  mask[o] = m;
  rows[o] = (int32_t)o;
}

// export helpers: element t = k * n_out + o of the k-major view
__global__ void k_export_flag(const int32_t* __restrict__ nbr, int64_t n_out, int kvol,
                              int32_t* flag) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n_out * kvol) return;
  int64_t k = t / n_out, o = t - k * n_out;
  flag[t] = nbr[o * kvol + k] >= 0;
}
__global__ void k_export_emit(const int32_t* __restrict__ nbr, int64_t n_out, int kvol,
                              const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                              int64_t cap, int32_t* ok, int32_t* oin, int32_t* oout) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n_out * kvol || !flag[t]) return;
  int64_t k = t / n_out, o = t - k * n_out;
  int64_t p = pos[t];
  if (p >= cap) return;
  ok[p] = (int32_t)k;
  oin[p] = nbr[o * kvol + k];
  oout[p] = (int32_t)o;
}

static int exclusive_scan_i32(const int32_t* d_in, int32_t* d_out, int64_t n, hipStream_t s) {
  size_t tmp_bytes = 0;
  CS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_in, d_out, (int)n, s));
  PoolBuf<char> tmp(tmp_bytes);
  CS_REQUIRE(tmp.p, CS_ERR_HIP, "scan scratch allocation failed");
  CS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, d_in, d_out, (int)n, s));
  return CS_OK;
}

static uint64_t table_capacity(int64_t n) {
  uint64_t c = 1024;
  while (c < (uint64_t)(2 * n)) c <<= 1;
  return c;
}

static int alloc_table(cs_coordmap* m, int64_t n_keys, hipStream_t s) {
  m->capacity = table_capacity(n_keys);
  m->d_keys = (uint64_t*)pool_alloc(m->capacity * sizeof(uint64_t));
  m->d_vals = (int32_t*)pool_alloc(m->capacity * sizeof(int32_t));
  CS_REQUIRE(m->d_keys && m->d_vals, CS_ERR_HIP, "hash table allocation failed");
  int blocks = (int)(m->capacity / 256 < 2048 ? m->capacity / 256 : 2048);
  hipLaunchKernelGGL(k_fill_table, dim3(blocks), dim3(256), 0, s, m->d_keys, m->d_vals,
                     m->capacity);
  CS_LAUNCH_CHECK();
  return CS_OK;
}

}  // namespace cs

using namespace cs;

extern "C" {

int cs_coordmap_create(const int32_t* d_coords, int64_t n, int tensor_stride, void* stream,
                       cs_coordmap** out) {
  CS_REQUIRE(out, CS_ERR_INVALID, "cs_coordmap_create: out is NULL");
  *out = nullptr;
  CS_REQUIRE(n >= 0 && n < (1LL << 30), CS_ERR_INVALID, "cs_coordmap_create: bad row count %lld",
             (long long)n);
  CS_REQUIRE(n == 0 || d_coords, CS_ERR_INVALID, "cs_coordmap_create: coords is NULL");
  CS_REQUIRE(tensor_stride >= 1, CS_ERR_INVALID, "cs_coordmap_create: bad tensor stride");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof("kmap", s);
  cs_coordmap* m = new cs_coordmap();
  m->n = n;
  m->tensor_stride = tensor_stride;
  int rc = alloc_table(m, n, s);
  if (rc) {
    cs_coordmap_free(m);
    return rc;
  }
  m->d_coords = (int32_t*)pool_alloc((n ? n : 1) * 4 * sizeof(int32_t));
  PoolBuf<int> status(2);
  if (!m->d_coords || !status.p) {
    cs_coordmap_free(m);
    set_error("coordinate allocation failed");
    return CS_ERR_HIP;
  }
  int h_status[2] = {0, 0};
  if (n > 0) {
    hipError_t e = hipMemcpyAsync(m->d_coords, d_coords, n * 4 * sizeof(int32_t),
                                  hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(status.p, 0, 2 * sizeof(int), s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_insert, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, m->d_coords,
                         n, 1, m->d_keys, m->d_vals, m->capacity - 1, status.p);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipMemcpyAsync(h_status, status.p, sizeof(h_status), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      cs_coordmap_free(m);
      set_error("cs_coordmap_create: %s", hipGetErrorString(e));
      return CS_ERR_HIP;
    }
  }
  if (h_status[0]) {
    cs_coordmap_free(m);
    set_error("cs_coordmap_create: coordinate out of the supported range "
              "(|x|,|y|,|z| < 32768, 0 <= batch < 65536)");
    return CS_ERR_RANGE;
  }
  if (h_status[1]) {
    cs_coordmap_free(m);
    set_error("cs_coordmap_create: %d duplicate coordinate rows (quantise the cloud first)",
              h_status[1]);
    return CS_ERR_DUPLICATE;
  }
  *out = m;
  return CS_OK;
}

int cs_coordmap_stride(const cs_coordmap* in, int stride, void* stream, cs_coordmap** out) {
  CS_REQUIRE(in && out, CS_ERR_INVALID, "cs_coordmap_stride: NULL argument");
  *out = nullptr;
  CS_REQUIRE(stride >= 2, CS_ERR_INVALID, "cs_coordmap_stride: stride must be >= 2");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof("kmap", s);
  const int64_t n = in->n;
  const int cell = in->tensor_stride * stride;
  cs_coordmap* m = new cs_coordmap();
  m->tensor_stride = cell;
  int rc = alloc_table(m, n, s);
  if (rc) {
    cs_coordmap_free(m);
    return rc;
  }
  PoolBuf<int> status(2);
  PoolBuf<int32_t> flag(n + 1), pos(n + 1);
  if (!status.p || !flag.p || !pos.p) {
    cs_coordmap_free(m);
    set_error("cs_coordmap_stride: scratch allocation failed");
    return CS_ERR_HIP;
  }
  int32_t h_last[2] = {0, 0};
  int h_status[2] = {0, 0};
  if (n > 0) {
    const unsigned g = (unsigned)ceil_div(n, 256);
    hipError_t e = hipMemsetAsync(status.p, 0, 2 * sizeof(int), s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_insert, dim3(g), dim3(256), 0, s, in->d_coords, n, cell, m->d_keys,
                         m->d_vals, m->capacity - 1, status.p);
      hipLaunchKernelGGL(k_flag_first, dim3(g), dim3(256), 0, s, in->d_coords, n, cell,
                         m->d_keys, m->d_vals, m->capacity - 1, flag.p);
      e = hipGetLastError();
    }
    if (e != hipSuccess) {
      cs_coordmap_free(m);
      set_error("cs_coordmap_stride: %s", hipGetErrorString(e));
      return CS_ERR_HIP;
    }
    rc = exclusive_scan_i32(flag.p, pos.p, n, s);
    if (rc) {
      cs_coordmap_free(m);
      return rc;
    }
    e = hipMemcpyAsync(&h_last[0], pos.p + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
      e = hipMemcpyAsync(&h_last[1], flag.p + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
      e = hipMemcpyAsync(h_status, status.p, sizeof(h_status), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      cs_coordmap_free(m);
      set_error("cs_coordmap_stride: %s", hipGetErrorString(e));
      return CS_ERR_HIP;
    }
    if (h_status[0]) {
      cs_coordmap_free(m);
      set_error("cs_coordmap_stride: coordinate out of the supported range");
      return CS_ERR_RANGE;
    }
  }
  m->n = (int64_t)h_last[0] + h_last[1];
  m->d_coords = (int32_t*)pool_alloc((m->n ? m->n : 1) * 4 * sizeof(int32_t));
  if (!m->d_coords) {
    cs_coordmap_free(m);
    set_error("cs_coordmap_stride: coordinate allocation failed");
    return CS_ERR_HIP;
  }
  if (n > 0) {
    hipLaunchKernelGGL(k_emit_strided, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s,
                       in->d_coords, n, cell, flag.p, pos.p, m->d_keys, m->d_vals,
                       m->capacity - 1, m->d_coords);
    hipError_t e = hipGetLastError();
    // flag/pos go back to the pool when this function returns; make sure the kernel is done
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      cs_coordmap_free(m);
      set_error("cs_coordmap_stride: %s", hipGetErrorString(e));
      return CS_ERR_HIP;
    }
  }
  *out = m;
  return CS_OK;
}

int64_t cs_coordmap_size(const cs_coordmap* m) { return m ? m->n : -1; }
int cs_coordmap_tensor_stride(const cs_coordmap* m) { return m ? m->tensor_stride : -1; }
const int32_t* cs_coordmap_coords(const cs_coordmap* m) { return m ? m->d_coords : nullptr; }

void cs_coordmap_free(cs_coordmap* m) {
  if (!m) return;
  pool_free(m->d_coords);
  pool_free(m->d_keys);
  pool_free(m->d_vals);
  delete m;
}

int cs_kernelmap_build(const cs_coordmap* in, const cs_coordmap* out, int kernel_size,
                       int transposed, void* stream, cs_kernelmap** km_out) {
  CS_REQUIRE(in && out && km_out, CS_ERR_INVALID, "cs_kernelmap_build: NULL argument");
  *km_out = nullptr;
  CS_REQUIRE(kernel_size == 3 || kernel_size == 1, CS_ERR_UNSUPPORTED,
             "cs_kernelmap_build: kernel_size %d not supported (1 or 3)", kernel_size);
  int step, sign;
  if (!transposed) {
    CS_REQUIRE(out->tensor_stride == in->tensor_stride ||
                   out->tensor_stride == 2 * in->tensor_stride,
               CS_ERR_INVALID, "cs_kernelmap_build: tensor strides %d -> %d not supported",
               in->tensor_stride, out->tensor_stride);
    step = in->tensor_stride;
    sign = +1;
  } else {
    CS_REQUIRE(in->tensor_stride == 2 * out->tensor_stride, CS_ERR_INVALID,
               "cs_kernelmap_build: transposed map needs in stride == 2 * out stride (%d, %d)",
               in->tensor_stride, out->tensor_stride);
    step = out->tensor_stride;
    sign = -1;
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof("kmap", s);
  cs_kernelmap* km = new cs_kernelmap();
  km->n_out = out->n;
  km->n_in = in->n;
  km->kvol = kernel_size == 3 ? 27 : 1;
  km->transposed = transposed;
  const int64_t total = km->n_out * km->kvol;
  km->d_nbr = (int32_t*)pool_alloc((total ? total : 1) * sizeof(int32_t));
  PoolBuf<unsigned long long> cnt(1);
  if (!km->d_nbr || !cnt.p) {
    cs_kernelmap_free(km);
    set_error("cs_kernelmap_build: allocation failed");
    return CS_ERR_HIP;
  }
  unsigned long long h_cnt = 0;
  hipError_t e = hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), s);
  if (e == hipSuccess && total > 0) {
    hipLaunchKernelGGL(k_build_nbr, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s,
                       out->d_coords, km->n_out, km->kvol, step, sign, in->d_keys, in->d_vals,
                       in->capacity - 1, km->d_nbr, cnt.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&h_cnt, cnt.p, sizeof(h_cnt), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    cs_kernelmap_free(km);
    set_error("cs_kernelmap_build: %s", hipGetErrorString(e));
    return CS_ERR_HIP;
  }
  km->num_pairs = (int64_t)h_cnt;
  // tiling order for the convolution kernels: rows sorted by presence mask (stable radix sort)
  if (km->kvol == 27 && km->n_out > 0) {
    const int64_t n = km->n_out;
    km->d_rowlist = (int32_t*)pool_alloc(n * sizeof(int32_t));
    PoolBuf<uint32_t> mask(n), mask_sorted(n);
    PoolBuf<int32_t> rows(n);
    if (!km->d_rowlist || !mask.p || !mask_sorted.p || !rows.p) {
      cs_kernelmap_free(km);
      set_error("cs_kernelmap_build: row list allocation failed");
      return CS_ERR_HIP;
    }
    hipLaunchKernelGGL(k_row_masks, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, km->d_nbr, n,
                       km->kvol, mask.p, rows.p);
    size_t tmp_bytes = 0;
    hipError_t e2 = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, mask.p, mask_sorted.p, rows.p,
                                                       km->d_rowlist, (int)n, 0, 27, s);
    PoolBuf<char> tmp(tmp_bytes);
    if (e2 == hipSuccess && tmp.p)
      e2 = hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, mask.p, mask_sorted.p, rows.p,
                                              km->d_rowlist, (int)n, 0, 27, s);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(s);
    if (e2 != hipSuccess || !tmp.p) {
      cs_kernelmap_free(km);
      set_error("cs_kernelmap_build: row sort failed: %s", hipGetErrorString(e2));
      return CS_ERR_HIP;
    }
  }
  *km_out = km;
  return CS_OK;
}

int64_t cs_kernelmap_num_pairs(const cs_kernelmap* km) { return km ? km->num_pairs : -1; }
int64_t cs_kernelmap_rows(const cs_kernelmap* km) { return km ? km->n_out : -1; }
const int32_t* cs_kernelmap_table(const cs_kernelmap* km) { return km ? km->d_nbr : nullptr; }

int64_t cs_kernelmap_export(const cs_kernelmap* km, int32_t* d_k, int32_t* d_in, int32_t* d_out,
                            int64_t capacity, void* stream) {
  CS_REQUIRE(km && d_k && d_in && d_out, CS_ERR_INVALID, "cs_kernelmap_export: NULL argument");
  CS_REQUIRE(capacity >= km->num_pairs, CS_ERR_INVALID,
             "cs_kernelmap_export: capacity %lld < %lld pairs", (long long)capacity,
             (long long)km->num_pairs);
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = km->n_out * km->kvol;
  if (total == 0) return 0;
  CS_REQUIRE(total < (1LL << 31), CS_ERR_UNSUPPORTED, "cs_kernelmap_export: table too large");
  PoolBuf<int32_t> flag(total), pos(total);
  CS_REQUIRE(flag.p && pos.p, CS_ERR_HIP, "cs_kernelmap_export: scratch allocation failed");
  const unsigned g = (unsigned)ceil_div(total, 256);
  hipLaunchKernelGGL(k_export_flag, dim3(g), dim3(256), 0, s, km->d_nbr, km->n_out, km->kvol,
                     flag.p);
  CS_LAUNCH_CHECK();
  int rc = exclusive_scan_i32(flag.p, pos.p, total, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_export_emit, dim3(g), dim3(256), 0, s, km->d_nbr, km->n_out, km->kvol,
                     flag.p, pos.p, capacity, d_k, d_in, d_out);
  CS_LAUNCH_CHECK();
  CS_HIP_CHECK(hipStreamSynchronize(s));
  return km->num_pairs;
}

void cs_kernelmap_free(cs_kernelmap* km) {
  if (!km) return;
  pool_free(km->d_nbr);
  pool_free(km->d_rowlist);
  delete km;
}

}  // extern "C"
