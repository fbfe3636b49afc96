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
#include <stdlib.h>

#include <mutex>
#include <algorithm>
#include <vector>

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

// floor(a / cell) * cell; cell a power of two (every tensor stride of the ResUNet): a mask instead of the
// ~35-instruction run-time division
__device__ __forceinline__ int floor_to_cell(int a, int cell) {
  return (cell & (cell - 1)) == 0 ? (a & ~(cell - 1)) : floor_div(a, cell) * cell;
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
    x = floor_to_cell(x, cell);
    y = floor_to_cell(y, cell);
    z = floor_to_cell(z, cell);
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
  int x = floor_to_cell(coords[4 * i + 1], cell);
  int y = floor_to_cell(coords[4 * i + 2], cell);
  int z = floor_to_cell(coords[4 * i + 3], cell);
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
  int x = floor_to_cell(coords[4 * i + 1], cell);
  int y = floor_to_cell(coords[4 * i + 2], cell);
  int z = floor_to_cell(coords[4 * i + 3], cell);
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
                            unsigned long long* pair_count,
                            const int* __restrict__ sample_mask) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int found = 0;
  if (t < n_out * kvol && (!sample_mask || sample_mask[out_coords[4 * (t / kvol)]])) {
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

// The same probes for the samples the LDS kernel flagged (too many voxels / too wide), launched behind
// it without a host decision: grid (slices, samples), a workgroup of an unflagged sample leaves at once.
__global__ __launch_bounds__(256) void k_build_nbr_flagged(
    const int32_t* __restrict__ out_coords, const int32_t* __restrict__ out_seg, int step, int sign,
    const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals, uint64_t mask,
    int32_t* __restrict__ nbr, unsigned long long* pair_count, const int* __restrict__ flagged) {
  const int b = blockIdx.y;
  if (!flagged[b]) return;
  const int64_t t0 = (int64_t)out_seg[b] * 27, t1 = (int64_t)out_seg[b + 1] * 27;
  for (int64_t base = t0 + (int64_t)blockIdx.x * 256; base < t1; base += (int64_t)gridDim.x * 256) {
    const int64_t t = base + threadIdx.x;
    int found = 0;
    if (t < t1) {
      const int64_t o = t / 27;
      const int k = (int)(t - o * 27);
      const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
      const int bb = out_coords[4 * o + 0];
      const int x = out_coords[4 * o + 1] + sign * dx * step;
      const int y = out_coords[4 * o + 2] + sign * dy * step;
      const int z = out_coords[4 * o + 3] + sign * dz * step;
      int32_t v = -1;
      if (coord_in_range(bb, x, y, z)) v = hash_lookup(keys, vals, mask, pack_key(bb, x, y, z));
      nbr[t] = v;
      found = v >= 0;
    }
    const unsigned long long m = __ballot(found);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(pair_count, (unsigned long long)__popcll(m));
  }
}

// ------------------------------------------------------------------------------------------------
// Kernel maps built in LDS.  Coordinates of a batch are grouped by sample (collate order), and a
// kernel-map probe never leaves its sample, so one workgroup loads the in-map coordinates of ONE
// sample into an LDS hash table (30-bit keys relative to the sample's bounding box, 16-bit local row)
// and answers all 27 probes of its slice of output rows from LDS -- no random global access at all.
// Samples with more than LDS_MAX_ROWS voxels or bounding boxes wider than 1023 cells are flagged and
// handled by the global-table kernel (k_build_nbr, restricted to the flagged samples); batches that
// are not grouped by sample use the global kernel for everything.
// ------------------------------------------------------------------------------------------------
constexpr int LDS_SLOTS_MAX = 24576;  // 4-B keys + 2-B local rows = 144 KB of the CU's 160 KB LDS (load factor <= 0.625)
constexpr uint32_t LDS_EMPTY = 0xffffffffu;

// seg[b] = first row whose batch index is >= b (rows must be grouped by ascending batch index).
__global__ void k_segments(const int32_t* __restrict__ coords, int64_t n, int n_batch,
                           int32_t* __restrict__ seg, int* __restrict__ flags) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = coords[4 * i];
  const int prev = i > 0 ? coords[4 * (i - 1)] : -1;
  if (b < prev || b >= n_batch || b < 0) {
    atomicOr(&flags[0], 1);  // not grouped by sample
    return;
  }
  for (int bb = prev + 1; bb <= b; ++bb) seg[bb] = (int32_t)i;
  if (i == n - 1)
    for (int bb = b + 1; bb <= n_batch; ++bb) seg[bb] = (int32_t)n;
}
__global__ void k_segment_max(const int32_t* __restrict__ seg, int n_batch, int* __restrict__ flags) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n_batch) atomicMax(&flags[1], seg[b + 1] - seg[b]);
}

// grid: x = slice of the sample's output rows, y = sample.  block = NT threads.
// SLOTS = table size: 24576 (144 KB, one workgroup per CU: the stride-1 maps of 15 k-voxel samples), 8192
// (48 KB, three per CU) or 2048 (12 KB) for the coarser levels, picked on the host from the mean sample size;
// a sample that does not fit its table is flagged and goes through the global table like any other overflow.
// ushift: log2(unit) when the in-map's tensor stride is a power of two (1, 2, 4, 8: every level of the ResUNet),
// else -1.  The probe loop divides by `unit` six times per probe; as run-time integer divisions (~35 VALU
// instructions each) they WERE the kernel: 174 -> see DESIGN 7c us per stride-1 map of the stress batch.
// SYM (in map == out map, a submanifold convolution): row o has row i at offset k exactly when row i has row o at the
// opposite offset 26 - k, and offset 13 is the row itself.  Only offsets 0..12 are probed; a hit writes both entries
// (every entry has one writer), misses keep the -1 the table was filled with before the launch.
template <int SLOTS, int NT, bool SYM>
__global__ __launch_bounds__(NT) void k_build_nbr_lds(
    const int32_t* __restrict__ in_coords, const int32_t* __restrict__ in_seg,
    const int32_t* __restrict__ out_coords, const int32_t* __restrict__ out_seg, int unit, int ushift, int step,
    int sign, int32_t* __restrict__ nbr, unsigned long long* __restrict__ pair_count,
    int* __restrict__ fallback) {
  auto udiv = [&](int v) { return ushift >= 0 ? v >> ushift : v / unit; };          // v >= 0
  auto umult = [&](int v) { return ushift >= 0 ? (v & (unit - 1)) == 0 : v % unit == 0; };
  constexpr int LDS_SLOTS = SLOTS;
  constexpr int LDS_MAX_ROWS = SLOTS / 8 * 5;   // load factor <= 0.625
  __shared__ uint32_t keys[LDS_SLOTS];
  __shared__ uint16_t vals[LDS_SLOTS];
  __shared__ int bmin[3], bmax[3];
  const int b = blockIdx.y;
  const int i0 = in_seg[b], i1 = in_seg[b + 1];
  const int o0 = out_seg[b], o1 = out_seg[b + 1];
  const int tid = threadIdx.x;
  if (o1 <= o0) return;
  const int per = (o1 - o0 + gridDim.x - 1) / gridDim.x;
  const int s0 = o0 + blockIdx.x * per, s1 = min(o1, s0 + per);
  if (s0 >= s1) return;
  if (tid < 3) {
    bmin[tid] = 0x7fffffff;
    bmax[tid] = -0x7fffffff;
  }
  for (int i = tid; i < LDS_SLOTS; i += NT) keys[i] = LDS_EMPTY;
  __syncthreads();
  {
    // bounding box: per-thread, then per-wave (shuffles), then one LDS atomic per wave and axis
    // (one atomic per coordinate serialises ~27 k updates on six addresses: 85 us -> measured below)
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
    for (int i = i0 + tid; i < i1; i += NT) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int v = in_coords[4 * i + 1 + a];
        lo[a] = min(lo[a], v);
        hi[a] = max(hi[a], v);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        lo[a] = min(lo[a], __shfl_xor(lo[a], off));
        hi[a] = max(hi[a], __shfl_xor(hi[a], off));
      }
      if ((tid & 63) == 0) {
        atomicMin(&bmin[a], lo[a]);
        atomicMax(&bmax[a], hi[a]);
      }
    }
  }
  __syncthreads();
  const int mx = bmin[0], my = bmin[1], mz = bmin[2];
  const int ex = i1 > i0 ? udiv(bmax[0] - mx) : 0, ey = i1 > i0 ? udiv(bmax[1] - my) : 0,
            ez = i1 > i0 ? udiv(bmax[2] - mz) : 0;
  if (ex > 1023 || ey > 1023 || ez > 1023 || i1 - i0 > LDS_MAX_ROWS) {
    if (tid == 0) fallback[b] = 1;
    return;
  }
  for (int i = i0 + tid; i < i1; i += NT) {
    const uint32_t key = (uint32_t)udiv(in_coords[4 * i + 1] - mx) |
                         ((uint32_t)udiv(in_coords[4 * i + 2] - my) << 10) |
                         ((uint32_t)udiv(in_coords[4 * i + 3] - mz) << 20);
    uint32_t slot = ((key * 2654435761u) >> 8) % LDS_SLOTS;
    while (true) {
      const uint32_t old = atomicCAS(&keys[slot], LDS_EMPTY, key);
      if (old == LDS_EMPTY || old == key) {
        vals[slot] = (uint16_t)(i - i0);  // coordinates are unique inside a map
        break;
      }
      slot = slot + 1 == LDS_SLOTS ? 0 : slot + 1;
    }
  }
  __syncthreads();
  int found_total = 0;
  constexpr int KP = SYM ? 14 : 27;   // offsets handled per row
  const int total = (s1 - s0) * KP;
  for (int t = tid; t < total; t += NT) {
    const int o = s0 + t / KP;
    const int k = t - (t / KP) * KP;
    if (SYM && k == 13) {
      nbr[(int64_t)o * 27 + 13] = o;
      found_total += 1;
      continue;
    }
    const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
    const int x = out_coords[4 * o + 1] + sign * dx * step - mx;
    const int y = out_coords[4 * o + 2] + sign * dy * step - my;
    const int z = out_coords[4 * o + 3] + sign * dz * step - mz;
    int32_t v = -1;
    if (x >= 0 && y >= 0 && z >= 0 && umult(x) && umult(y) && umult(z)) {
      const int cx = udiv(x), cy = udiv(y), cz = udiv(z);
      if (cx <= ex && cy <= ey && cz <= ez) {
        const uint32_t key = (uint32_t)cx | ((uint32_t)cy << 10) | ((uint32_t)cz << 20);
        uint32_t slot = ((key * 2654435761u) >> 8) % LDS_SLOTS;
        while (true) {
          const uint32_t kk = keys[slot];
          if (kk == key) {
            v = i0 + (int32_t)vals[slot];
            break;
          }
          if (kk == LDS_EMPTY) break;
          slot = slot + 1 == LDS_SLOTS ? 0 : slot + 1;
        }
      }
    }
    if (SYM) {
      if (v >= 0) {
        nbr[(int64_t)o * 27 + k] = v;
        nbr[(int64_t)v * 27 + (26 - k)] = o;
        found_total += 2;
      }
    } else {
      nbr[(int64_t)o * 27 + k] = v;
      found_total += v >= 0;
    }
  }
  // block-level pair count: wave reduce, one atomic per wave
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) found_total += __shfl_xor(found_total, off);
  if ((tid & 63) == 0 && found_total) atomicAdd(pair_count, (unsigned long long)found_total);
}

// Per-sample segment table of a coordinate map (computed once, cached on the map).
static int ensure_segments(cs_coordmap* m, hipStream_t s) {
  if (m->seg_state != 0) return CS_OK;
  m->seg_state = -1;  // unavailable unless everything below succeeds
  if (m->n == 0) return CS_OK;
  int32_t last_b = -1;
  CS_HIP_CHECK(download_async(&last_b, m->d_coords + 4 * (m->n - 1), sizeof(int32_t), s));
  CS_HIP_CHECK(download_sync(s));
  if (last_b < 0 || last_b >= 65536) return CS_OK;
  const int nb = last_b + 1;
  int32_t* seg = (int32_t*)pool_alloc((size_t)(nb + 1) * sizeof(int32_t));
  PoolBuf<int> flags(2);
  if (!seg || !flags.p) {
    pool_free(seg);
    return CS_OK;
  }
  int h_flags[2] = {0, 0};
  CS_HIP_CHECK(hipMemsetAsync(flags.p, 0, 2 * sizeof(int), s));
  CS_HIP_CHECK(hipMemsetAsync(seg, 0, (size_t)(nb + 1) * sizeof(int32_t), s));
  hipLaunchKernelGGL(k_segments, dim3((unsigned)ceil_div(m->n, 256)), dim3(256), 0, s, m->d_coords,
                     m->n, nb, seg, flags.p);
  hipLaunchKernelGGL(k_segment_max, dim3((unsigned)ceil_div(nb, 256)), dim3(256), 0, s, seg, nb,
                     flags.p);
  CS_HIP_CHECK(download_async(h_flags, flags.p, sizeof(h_flags), s));
  CS_HIP_CHECK(download_sync(s));
  if (h_flags[0]) {
    pool_free(seg);
    return CS_OK;  // not grouped by sample: global path
  }
  m->d_seg = seg;
  m->n_batch = nb;
  m->max_seg = h_flags[1];
  m->seg_state = 1;
  return CS_OK;
}

// The same for several maps at once: TWO host round trips in total (last batch index of every map; the grouping
// flags of every map) instead of two per map -- the four coordinate maps of a ResUNet batch cost eight otherwise.
static int ensure_segments_many(cs_coordmap* const* maps, int n, hipStream_t s) {
  std::vector<cs_coordmap*> todo;
  for (int i = 0; i < n; ++i) {
    cs_coordmap* m = maps[i];
    if (!m || m->seg_state != 0) continue;
    if (std::find(todo.begin(), todo.end(), m) != todo.end()) continue;
    m->seg_state = -1;  // unavailable unless everything below succeeds
    if (m->n > 0) todo.push_back(m);
  }
  if (todo.empty()) return CS_OK;
  const size_t k = todo.size();
  std::vector<int32_t> last_b(k, -1);
  std::vector<int32_t*> seg(k, nullptr);
  std::vector<int> h_flags(2 * k, 0);
  // every HIP error is COLLECTED (no return in the middle of the loops): on failure the blocks allocated so far go
  // back to the pool and the maps return to "not computed" (seg_state 0), so a later call can try again
  hipError_t e = hipSuccess;
  auto keep = [&e](hipError_t r) {
    if (e == hipSuccess && r != hipSuccess) e = r;
  };
  for (size_t i = 0; i < k && e == hipSuccess; ++i)
    keep(download_async(&last_b[i], todo[i]->d_coords + 4 * (todo[i]->n - 1), sizeof(int32_t), s));
  if (e == hipSuccess) keep(download_sync(s));
  PoolBuf<int> flags(2 * k);
  if (e == hipSuccess && !flags.p) e = hipErrorOutOfMemory;
  if (e == hipSuccess) keep(hipMemsetAsync(flags.p, 0, 2 * k * sizeof(int), s));
  for (size_t i = 0; i < k && e == hipSuccess; ++i) {
    cs_coordmap* m = todo[i];
    if (last_b[i] < 0 || last_b[i] >= 65536) continue;   // stays -1: the global-table path serves this map
    const int nb = last_b[i] + 1;
    seg[i] = (int32_t*)pool_alloc((size_t)(nb + 1) * sizeof(int32_t));
    if (!seg[i]) {
      e = hipErrorOutOfMemory;
      break;
    }
    keep(hipMemsetAsync(seg[i], 0, (size_t)(nb + 1) * sizeof(int32_t), s));
    if (e != hipSuccess) break;
    hipLaunchKernelGGL(k_segments, dim3((unsigned)ceil_div(m->n, 256)), dim3(256), 0, s, m->d_coords, m->n, nb, seg[i],
                       flags.p + 2 * i);
    hipLaunchKernelGGL(k_segment_max, dim3((unsigned)ceil_div(nb, 256)), dim3(256), 0, s, seg[i], nb, flags.p + 2 * i);
    keep(hipGetLastError());
  }
  if (e == hipSuccess) keep(download_async(h_flags.data(), flags.p, 2 * k * sizeof(int), s));
  if (e == hipSuccess) keep(download_sync(s));
  for (size_t i = 0; i < k; ++i) {
    cs_coordmap* m = todo[i];
    if (e != hipSuccess) {
      pool_free(seg[i]);          // (nullptr is fine)
      m->seg_state = 0;           // not computed: nothing about this map was learnt
      continue;
    }
    if (!seg[i]) continue;
    if (h_flags[2 * i]) {         // not grouped by sample: global path, seg_state stays -1
      pool_free(seg[i]);
      continue;
    }
    m->d_seg = seg[i];
    m->n_batch = last_b[i] + 1;
    m->max_seg = h_flags[2 * i + 1];
    m->seg_state = 1;
  }
  if (e != hipSuccess) {
    set_error("ensure_segments_many: %s", hipGetErrorString(e));
    return CS_ERR_HIP;
  }
  return CS_OK;
}

// Page-locked slots for the pair counts of kernel maps (one per live map, recycled on free).
namespace {
std::mutex g_slot_mu;
std::vector<unsigned long long*> g_slot_slabs;
std::vector<int> g_slot_free;
constexpr int SLOTS_PER_SLAB = 1024;
}  // namespace

static unsigned long long* count_slot_acquire(int* slot) {
  std::lock_guard<std::mutex> lk(g_slot_mu);
  if (g_slot_free.empty()) {
    unsigned long long* slab = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&slab), sizeof(unsigned long long) * SLOTS_PER_SLAB, 0) !=
        hipSuccess)
      return nullptr;
    const int first = (int)g_slot_slabs.size() * SLOTS_PER_SLAB;
    g_slot_slabs.push_back(slab);
    for (int i = SLOTS_PER_SLAB - 1; i >= 0; --i) g_slot_free.push_back(first + i);
  }
  *slot = g_slot_free.back();
  g_slot_free.pop_back();
  return g_slot_slabs[*slot / SLOTS_PER_SLAB] + *slot % SLOTS_PER_SLAB;
}

static void count_slot_release(int slot) {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_slot_mu);
  g_slot_free.push_back(slot);
}

// maps holding deferred profile units (cs_conv_fwd registers, delivery / cs_kernelmap_free remove)
static std::mutex g_prof_maps_mu;
static std::vector<cs_kernelmap*> g_prof_maps;

static void prof_maps_remove(cs_kernelmap* km) {
  std::lock_guard<std::mutex> lk(g_prof_maps_mu);
  for (size_t i = 0; i < g_prof_maps.size(); ++i)
    if (g_prof_maps[i] == km) {
      g_prof_maps[i] = g_prof_maps.back();
      g_prof_maps.pop_back();
      return;
    }
}

void kernelmap_defer_prof(cs_kernelmap* km, double per_pair) {
  const uint64_t ep = prof_epoch();
  std::lock_guard<std::mutex> lk(g_prof_maps_mu);
  const bool registered = km->prof_flop_per_pair > 0.0;
  if (km->prof_epoch != ep) {      // what it still holds belongs to a region that has been reset away
    km->prof_flop_per_pair = 0.0;
    km->prof_epoch = ep;
  }
  if (!registered) g_prof_maps.push_back(km);
  km->prof_flop_per_pair += per_pair;
}

int64_t kernelmap_pairs(const cs_kernelmap* km_c) {
  cs_kernelmap* km = const_cast<cs_kernelmap*>(km_c);  // cached on first use
  if (!km) return -1;
  if (km->num_pairs < 0 && km->cnt_ready && km->h_cnt) {
    if (hipEventSynchronize(km->cnt_ready) != hipSuccess) return -1;
    km->num_pairs = (int64_t)*km->h_cnt;
  }
  if (km->num_pairs >= 0 && km->prof_flop_per_pair > 0.0) {   // convolutions profiled before the count had arrived
    double units = 0.0;
    uint64_t ep = 0;
    {
      std::lock_guard<std::mutex> lk(g_prof_maps_mu);
      units = km->prof_flop_per_pair * (double)km->num_pairs;
      ep = km->prof_epoch;
      km->prof_flop_per_pair = 0.0;
    }
    prof_maps_remove(km);
    if (units > 0.0) prof_add_units_epoch("conv", units, ep);
  }
  return km->num_pairs;
}

void kernelmap_flush_prof() {
  std::vector<cs_kernelmap*> maps;
  {
    std::lock_guard<std::mutex> lk(g_prof_maps_mu);
    maps = g_prof_maps;
  }
  for (cs_kernelmap* km : maps) (void)kernelmap_pairs(km);   // waits for the count (a few microseconds behind the build)
}

// Tiling order of the convolution kernels: output rows sorted by the Gray-code RANK of their 27-bit
// neighbour-presence mask (rows with the same mask are adjacent; neighbours in the order differ in few
// offsets).  A 32-row group of k_conv_dma then executes only 1.2 - 1.4x the MFMAs its rows need, because
// it skips every offset none of its rows has (measured on the stress clouds, executed / useful per 32-row
// group at strides 1 / 2 / 4: Gray rank 1.41 / 1.19 / 1.28, plain numeric order 1.48 / 1.20 / 1.29, the
// 12-bit grouping of rounds 1-2 2.05 / 1.59 / 1.52).  The order inside equal keys is the radix sort's
// (stable: row order); every output row is computed independently, results do not depend on the order.
__global__ __launch_bounds__(256) void k_row_keys(const int32_t* __restrict__ nbr, int64_t n_out, int kvol,
                                                  uint32_t* __restrict__ key, int32_t* __restrict__ row,
                                                  const unsigned long long* __restrict__ d_cnt,
                                                  unsigned long long* host_cnt) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // the pair count of the build kernels in front of this one goes to its page-locked slot from here (a
  // device -> host copy of 8 bytes was one blit-kernel launch per map)
  if (o == 0 && host_cnt) __hip_atomic_store(host_cnt, *d_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (o >= n_out) return;
  uint32_t m = 0;
  for (int k = 0; k < kvol; ++k) m |= (nbr[o * kvol + k] >= 0 ? 1u : 0u) << k;
  // inverse Gray code: r with r ^ (r >> 1) == m
  m ^= m >> 1;
  m ^= m >> 2;
  m ^= m >> 4;
  m ^= m >> 8;
  m ^= m >> 16;
  key[o] = m;
  row[o] = (int32_t)o;
}

// Row order of a SMALL map (<= SORT_SMALL_MAX rows: the coarsest level of a batch) in one launch: bitonic sort
// of (key << 32 | row) in LDS by one workgroup.  hipcub's radix sort takes a block-sort + merge path of 5 - 8
// launches for such sizes (55 launches per chair step over the ten maps of a batch).  Any order of equal keys
// is fine for the convolution; (key, row) pairs are distinct, so the result is also deterministic.
constexpr int SORT_SMALL_MAX = 4096;   // (16 384 rows take one workgroup 150 us: hipcub wins from ~8 k rows on)
__global__ __launch_bounds__(1024) void k_sort_small(const uint32_t* __restrict__ key, int n, int npow2,
                                                     uint32_t* __restrict__ key_sorted, int32_t* __restrict__ rowlist) {
  extern __shared__ unsigned long long sk[];
  for (int i = threadIdx.x; i < npow2; i += 1024)
    sk[i] = i < n ? ((unsigned long long)key[i] << 32) | (unsigned)i : ~0ULL;
  __syncthreads();
  for (int size = 2; size <= npow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (npow2 >> 1); t += 1024) {
        const int lo = 2 * t - (t & (stride - 1));      // index of the lower element of pair t
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = sk[lo], b = sk[hi];
        if ((a > b) == up) {
          sk[lo] = b;
          sk[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += 1024) {
    key_sorted[i] = (uint32_t)(sk[i] >> 32);
    rowlist[i] = (int32_t)(sk[i] & 0xffffffffu);
  }
}

// the neighbour table in tiling order + the offsets every 32-row group of that order uses (one thread per
// table element; the group masks come from the sorted keys: mask = Gray code of the rank)
__global__ __launch_bounds__(256) void k_sorted_tables(const int32_t* __restrict__ nbr, const int32_t* __restrict__ rowlist,
                                                       const uint32_t* __restrict__ key_sorted, int64_t n_out, int kvol,
                                                       int32_t* __restrict__ nbr_sorted, uint32_t* __restrict__ gmask,
                                                       int64_t n_groups_padded, int32_t absent) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n_out * kvol) {
    const int64_t t = e / kvol;
    const int k = (int)(e - t * kvol);
    // an absent neighbour is stored as row `absent` = n_in, ONE PAST the input tensor: the convolution turns entries
    // into byte offsets with one multiply-add and the buffer descriptor's range check returns zeros for that row
    const int32_t v = nbr[(int64_t)rowlist[t] * kvol + k];
    nbr_sorted[e] = v < 0 ? absent : v;
  }
  if (e < n_groups_padded) {
    uint32_t m = 0;
    for (int r = 0; r < 32; ++r) {
      const int64_t t = e * 32 + r;
      if (t < n_out) {
        const uint32_t rank = key_sorted[t];
        m |= rank ^ (rank >> 1);
      }
    }
    gmask[e] = m;
  }
}

// export helpers: element t = k * n_out + o of the k-major view
// ---- the tiling order of ALL kernel maps of a batch in one pass (round 4) ---------------------------------------------
// Ten maps of a batch used to mean ten row-key launches, ten device sorts (5 - 20 launches each: hipcub takes a block-sort +
// merge path for the mid-size maps) and ten table launches: 115 launches per stress batch, 60 per chair batch, bound by
// launch latency.  Here the keys of all maps go into ONE array -- key = map index << 27 | Gray rank, so one radix sort
// leaves every map's rows contiguous and ordered -- and one launch writes all sorted tables.  The descriptors travel as a
// kernel argument (no upload).
constexpr int ORDER_MAX_MAPS = 16;
struct OrderMap {
  const int32_t* nbr;
  int32_t* nbr_sorted;
  uint32_t* gmask;
  int32_t* rowlist;
  const unsigned long long* d_cnt;
  unsigned long long* host_cnt;
  int64_t n_out, base, n_groups;
  int32_t absent;
  uint32_t kblk0, tblk0;   // first workgroup of this map in the key / table launch
};
struct OrderTab {
  int n;
  OrderMap m[ORDER_MAX_MAPS];
};
__global__ __launch_bounds__(256) void k_row_keys_all(const OrderTab tab, uint32_t* __restrict__ key,
                                                      int32_t* __restrict__ row) {
  int i = 0;
  while (i + 1 < tab.n && blockIdx.x >= tab.m[i + 1].kblk0) ++i;
  const OrderMap& mp = tab.m[i];
  const int64_t o = (int64_t)(blockIdx.x - mp.kblk0) * 256 + threadIdx.x;
  if (o == 0 && mp.host_cnt) __hip_atomic_store(mp.host_cnt, *mp.d_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (o >= mp.n_out) return;
  uint32_t m = 0;
  for (int k = 0; k < 27; ++k) m |= (mp.nbr[o * 27 + k] >= 0 ? 1u : 0u) << k;
  m ^= m >> 1;   // inverse Gray code (see k_row_keys)
  m ^= m >> 2;
  m ^= m >> 4;
  m ^= m >> 8;
  m ^= m >> 16;
  key[mp.base + o] = ((uint32_t)i << 27) | m;
  row[mp.base + o] = (int32_t)o;
}
__global__ __launch_bounds__(256) void k_sorted_tables_all(const OrderTab tab, const uint32_t* __restrict__ key_sorted,
                                                           const int32_t* __restrict__ row_sorted) {
  int i = 0;
  while (i + 1 < tab.n && blockIdx.x >= tab.m[i + 1].tblk0) ++i;
  const OrderMap& mp = tab.m[i];
  const int64_t e = (int64_t)(blockIdx.x - mp.tblk0) * 256 + threadIdx.x;
  const int32_t* rl = row_sorted + mp.base;
  const uint32_t* ks = key_sorted + mp.base;
  if (e < mp.n_out * 27) {
    const int64_t t = e / 27;
    const int k = (int)(e - t * 27);
    const int32_t v = mp.nbr[(int64_t)rl[t] * 27 + k];
    mp.nbr_sorted[e] = v < 0 ? mp.absent : v;
  }
  if (e < mp.n_out) mp.rowlist[e] = rl[e];
  if (e < mp.n_groups) {
    uint32_t m = 0;
    for (int r = 0; r < 32; ++r) {
      const int64_t t = e * 32 + r;
      if (t < mp.n_out) {
        const uint32_t rank = ks[t] & 0x7ffffffu;
        m |= rank ^ (rank >> 1);
      }
    }
    mp.gmask[e] = m;
  }
}

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
  pool_use_stream(s);
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
      e = download_async(h_status, status.p, sizeof(h_status), s);
    if (e == hipSuccess) e = download_sync(s);
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
  pool_use_stream(s);
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
    e = download_async(&h_last[0], pos.p + (n - 1), sizeof(int32_t), s);
    if (e == hipSuccess)
      e = download_async(&h_last[1], flag.p + (n - 1), sizeof(int32_t), s);
    if (e == hipSuccess)
      e = download_async(h_status, status.p, sizeof(h_status), s);
    if (e == hipSuccess) e = download_sync(s);
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
    // (no synchronisation: flag / pos go back to this thread's stream-ordered cache, whose next user is enqueued
    // behind this kernel)
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
  pool_free(m->d_seg);
  delete m;
}

}  // extern "C"

// one kernel map on stream `s` (the caller has announced its stream to the pool)
// defer_order: leave the tiling order (row list, sorted table, group masks) and the pair-count hand-over to
// order_many(), which does them for all maps of a batch at once; *d_cnt_out then receives the device counter
// (scratch of this call: valid until the caller ends its pool deferral)
static int kernelmap_build_on(const cs_coordmap* in, const cs_coordmap* out, int kernel_size, int transposed,
                              hipStream_t s, cs_kernelmap** km_out, bool defer_order = false,
                              const unsigned long long** d_cnt_out = nullptr) {
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
  hipError_t e = hipSuccess;
  // LDS path: per-sample hash tables (see k_build_nbr_lds), the flagged samples through the global
  // table right behind it; otherwise the global table for everything.  No host decision in between:
  // the pair count travels to a page-locked slot and is read when somebody asks (kernelmap_pairs).
  bool used_lds = false;
  PoolBuf<int> fb;
  const unsigned long long* d_cnt = cnt.p;
  if (e == hipSuccess && total > 0 && km->kvol == 27 && getenv("CS_KMAP_GLOBAL") == nullptr) {
    cs_coordmap* in_m = const_cast<cs_coordmap*>(in);
    cs_coordmap* out_m = const_cast<cs_coordmap*>(out);
    if (ensure_segments(in_m, s) == CS_OK && ensure_segments(out_m, s) == CS_OK &&
        in_m->seg_state == 1 && out_m->seg_state == 1 && in_m->n_batch == out_m->n_batch) {
      const int nb = in_m->n_batch;
      // pair counter and per-sample fallback flags in one block: one memset
      fb.alloc(2 + nb);
      if (fb.p) {
        unsigned long long* const cnt_p = reinterpret_cast<unsigned long long*>(fb.p);
        int* const fb_p = fb.p + 2;
        const int ts_in = in->tensor_stride;
        int ushift = -1;
        if (ts_in > 0 && (ts_in & (ts_in - 1)) == 0) {
          ushift = 0;
          while ((1 << ushift) < ts_in) ++ushift;
        }
        // workgroups per map: one per CU.  Every slice of a sample rebuilds the sample's table, so 512 workgroups (two
        // rounds on 256 CUs with the 144-KB tables) build every table twice as often as 256 do for the same probes:
        // stress 6 330 -> 6 430 clouds/s (CS_KMAP_WGS=128 / 256 / 512 / 1024: 6 235 / 6 430 / 6 330 / 6 312)
        static const int slice_wgs = getenv("CS_KMAP_WGS") ? std::max(atoi(getenv("CS_KMAP_WGS")), 1) : 256;
        int slices = slice_wgs / (nb > 0 ? nb : 1);
        if (slices < 1) slices = 1;
        if (slices > 8) slices = 8;
        e = hipMemsetAsync(fb.p, 0, sizeof(int) * (2 + nb), s);
        if (e == hipSuccess) {
          // table size from the mean in-sample size (2.5x headroom; larger samples take the flagged path)
          const int64_t need = (in->n / (nb > 0 ? nb : 1)) * 5 / 2;
          static const bool small_tables = !(getenv("CS_KMAP_SMALL") && getenv("CS_KMAP_SMALL")[0] == '0');
          // submanifold map (same coordinate map on both sides, plain convolution): half of the probes
          // (maps of >= 200 000 rows: below that the table fill and the scattered mirror writes cost what the probes
          // save -- stress 7 217 -> 7 317 clouds/s, the 160 000-row maps of the chair batch unchanged either way)
          // CS_KMAP_SYM=0 / 1: never / whatever the size (tests)
          const char* env_sym = getenv("CS_KMAP_SYM");
          const bool sym = in == out && !transposed && sign > 0 &&
                           (env_sym ? env_sym[0] == '1' : km->n_out >= 200000);
          if (sym) e = hipMemsetAsync(km->d_nbr, 0xff, (size_t)total * sizeof(int32_t), s);
#define CS_NBR_LDS(SLOTS_, NT_)                                                                                          \
  do {                                                                                                                    \
    if (sym)                                                                                                              \
      hipLaunchKernelGGL((k_build_nbr_lds<SLOTS_, NT_, true>), dim3((unsigned)slices, (unsigned)nb), dim3(NT_), 0, s,      \
                         in->d_coords, in_m->d_seg, out->d_coords, out_m->d_seg, in->tensor_stride, ushift, step, sign,   \
                         km->d_nbr, cnt_p, fb_p);                                                                         \
    else                                                                                                                  \
      hipLaunchKernelGGL((k_build_nbr_lds<SLOTS_, NT_, false>), dim3((unsigned)slices, (unsigned)nb), dim3(NT_), 0, s,     \
                         in->d_coords, in_m->d_seg, out->d_coords, out_m->d_seg, in->tensor_stride, ushift, step, sign,   \
                         km->d_nbr, cnt_p, fb_p);                                                                         \
  } while (0)
          if (e != hipSuccess) {
          } else if (small_tables && need <= 2048 / 8 * 5)
            CS_NBR_LDS(2048, 256);
          else if (small_tables && need <= 8192 / 8 * 5)
            CS_NBR_LDS(8192, 512);
          else
            CS_NBR_LDS(LDS_SLOTS_MAX, 1024);
#undef CS_NBR_LDS
          hipLaunchKernelGGL(k_build_nbr_flagged, dim3(64, (unsigned)nb), dim3(256), 0, s, out->d_coords,
                             out_m->d_seg, step, sign, in->d_keys, in->d_vals, in->capacity - 1,
                             km->d_nbr, cnt_p, fb_p);
          e = hipGetLastError();
        }
        if (e == hipSuccess) {
          used_lds = true;
          d_cnt = cnt_p;
        }
      }
    }
  }
  if (e == hipSuccess && total > 0 && !used_lds) {
    e = hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), s);   // (the LDS path keeps its counter in `fb`)
    hipLaunchKernelGGL(k_build_nbr, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s,
                       out->d_coords, km->n_out, km->kvol, step, sign, in->d_keys, in->d_vals,
                       in->capacity - 1, km->d_nbr, cnt.p, (const int*)nullptr);
    e = hipGetLastError();
  }
  unsigned long long* cnt_host_dev = nullptr;
  if (e == hipSuccess) {
    if (total == 0) {
      km->num_pairs = 0;
    } else {
      km->h_cnt = count_slot_acquire(&km->cnt_slot);
      if (!km->h_cnt || hipEventCreateWithFlags(&km->cnt_ready, hipEventDisableTiming) != hipSuccess) {
        e = hipErrorOutOfMemory;
      } else if (km->kvol == 27) {
        // the row-key kernel (below, or order_many's) writes the count into the page-locked slot itself (no copy launch)
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&cnt_host_dev), km->h_cnt, 0) != hipSuccess)
          cnt_host_dev = nullptr;
      }
      if (e == hipSuccess && !cnt_host_dev) {
        e = hipMemcpyAsync(km->h_cnt, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipEventRecord(km->cnt_ready, s);
      }
    }
  }
  if (e != hipSuccess) {
    cs_kernelmap_free(km);
    set_error("cs_kernelmap_build: %s", hipGetErrorString(e));
    return CS_ERR_HIP;
  }
  if (defer_order && km->kvol == 27 && km->n_out > 0 && cnt_host_dev) {
    *d_cnt_out = d_cnt;
    *km_out = km;
    return CS_OK;
  }
  // tiling order for the convolution kernels: rows sorted by the Gray rank of their presence mask
  if (km->kvol == 27 && km->n_out > 0) {
    const int64_t n = km->n_out;
    km->d_rowlist = (int32_t*)pool_alloc(n * sizeof(int32_t));
    PoolBuf<uint32_t> key(n), key_sorted(n);
    PoolBuf<int32_t> row(n);
    const bool small = n <= SORT_SMALL_MAX;
    size_t tmp_bytes = 0;
    hipError_t e2 = small ? hipSuccess
                          : hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key.p, key_sorted.p, row.p,
                                                               km->d_rowlist, (int)n, 0, 27, s);
    PoolBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    if (!km->d_rowlist || !key.p || !key_sorted.p || !row.p || !tmp.p) {
      cs_kernelmap_free(km);
      set_error("cs_kernelmap_build: row list allocation failed");
      return CS_ERR_HIP;
    }
    hipLaunchKernelGGL(k_row_keys, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, km->d_nbr, n, km->kvol, key.p,
                       row.p, d_cnt, cnt_host_dev);
    if (cnt_host_dev && hipEventRecord(km->cnt_ready, s) != hipSuccess) e2 = hipErrorUnknown;
    if (small) {
      int npow2 = 2;
      while (npow2 < n) npow2 <<= 1;
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_small),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         SORT_SMALL_MAX * (int)sizeof(unsigned long long));
      if (e2 == hipSuccess) e2 = attr;   // (keeps an earlier error, e.g. of the cnt_ready record)
      if (e2 == hipSuccess)
        hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(1024), (size_t)npow2 * sizeof(unsigned long long), s, key.p, (int)n,
                           npow2, key_sorted.p, km->d_rowlist);
    } else if (e2 == hipSuccess) {
      e2 = hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key.p, key_sorted.p, row.p, km->d_rowlist, (int)n, 0,
                                              27, s);
    }
    const int64_t n_groups = ceil_div(ceil_div(n, 32), 8) * 8;
    km->d_nbr_sorted = (int32_t*)pool_alloc((size_t)n * km->kvol * sizeof(int32_t));
    km->d_gmask = (uint32_t*)pool_alloc((size_t)n_groups * sizeof(uint32_t));
    if (!km->d_nbr_sorted || !km->d_gmask) {
      cs_kernelmap_free(km);
      set_error("cs_kernelmap_build: sorted table allocation failed");
      return CS_ERR_HIP;
    }
    hipLaunchKernelGGL(k_sorted_tables, dim3((unsigned)ceil_div(n * km->kvol, 256)), dim3(256), 0, s, km->d_nbr,
                       km->d_rowlist, key_sorted.p, n, km->kvol, km->d_nbr_sorted, km->d_gmask, n_groups, (int32_t)km->n_in);
    if (e2 == hipSuccess) e2 = hipGetLastError();
    // no synchronisation: the scratch returns to this thread's stream-ordered cache
    if (e2 != hipSuccess) {
      cs_kernelmap_free(km);
      set_error("cs_kernelmap_build: row grouping failed: %s", hipGetErrorString(e2));
      return CS_ERR_HIP;
    }
  }
  *km_out = km;
  return CS_OK;
}

extern "C" {

int cs_kernelmap_build(const cs_coordmap* in, const cs_coordmap* out, int kernel_size,
                       int transposed, void* stream, cs_kernelmap** km_out) {
  pool_use_stream((hipStream_t)stream);
  return kernelmap_build_on(in, out, kernel_size, transposed, (hipStream_t)stream, km_out);
}

// Tiling order of the maps kms[idx[0..m)] (all kvol 27, n_out > 0, built with defer_order) on stream s, which is
// behind every stream that built them: one key launch, one radix sort, one table launch.
static int order_many(cs_kernelmap* const* kms, const unsigned long long* const* d_cnts, const int* idx, int m,
                      hipStream_t s) {
  if (m == 0) return CS_OK;
  OrderTab tab;
  tab.n = m;
  int64_t total = 0;
  uint64_t kblk = 0, tblk = 0;
  for (int j = 0; j < m; ++j) {
    cs_kernelmap* km = kms[idx[j]];
    OrderMap& o = tab.m[j];
    const int64_t n = km->n_out;
    o.n_out = n;
    o.base = total;
    o.n_groups = ceil_div(ceil_div(n, 32), 8) * 8;
    o.absent = (int32_t)km->n_in;
    o.nbr = km->d_nbr;
    km->d_rowlist = (int32_t*)pool_alloc(n * sizeof(int32_t));
    km->d_nbr_sorted = (int32_t*)pool_alloc((size_t)n * 27 * sizeof(int32_t));
    km->d_gmask = (uint32_t*)pool_alloc((size_t)o.n_groups * sizeof(uint32_t));
    CS_REQUIRE(km->d_rowlist && km->d_nbr_sorted && km->d_gmask, CS_ERR_HIP, "cs_kernelmap_build_many: sorted table allocation failed");
    o.rowlist = km->d_rowlist;
    o.nbr_sorted = km->d_nbr_sorted;
    o.gmask = km->d_gmask;
    o.d_cnt = d_cnts[idx[j]];
    o.host_cnt = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&o.host_cnt), km->h_cnt, 0) != hipSuccess) o.host_cnt = nullptr;
    CS_REQUIRE(o.host_cnt, CS_ERR_HIP, "cs_kernelmap_build_many: no device view of the pair-count slot");
    o.kblk0 = (uint32_t)kblk;
    o.tblk0 = (uint32_t)tblk;
    kblk += (uint64_t)ceil_div(n, 256);
    tblk += (uint64_t)ceil_div(n * 27, 256);
    total += n;
  }
  CS_REQUIRE(total < (1LL << 31) && tblk < (1ULL << 31), CS_ERR_UNSUPPORTED, "cs_kernelmap_build_many: batch too large");
  int map_bits = 0;
  while ((1 << map_bits) < m) ++map_bits;
  PoolBuf<uint32_t> key(total), key_sorted(total);
  PoolBuf<int32_t> row(total), row_sorted(total);
  size_t tmp_bytes = 0;
  CS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key.p, key_sorted.p, row.p, row_sorted.p, (int)total, 0,
                                                  27 + map_bits, s));
  PoolBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
  CS_REQUIRE(key.p && key_sorted.p && row.p && row_sorted.p && tmp.p, CS_ERR_HIP,
             "cs_kernelmap_build_many: row order scratch allocation failed");
  hipLaunchKernelGGL(k_row_keys_all, dim3((unsigned)kblk), dim3(256), 0, s, tab, key.p, row.p);
  CS_LAUNCH_CHECK();
  for (int j = 0; j < m; ++j) CS_HIP_CHECK(hipEventRecord(kms[idx[j]]->cnt_ready, s));
  CS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key.p, key_sorted.p, row.p, row_sorted.p, (int)total, 0,
                                                  27 + map_bits, s));
  hipLaunchKernelGGL(k_sorted_tables_all, dim3((unsigned)tblk), dim3(256), 0, s, tab, key_sorted.p, row_sorted.p);
  CS_LAUNCH_CHECK();
  return CS_OK;
}

// The kernel maps of a batch are independent chains of ~10-20 small dependent launches each (table, row keys, sort,
// tiling-order tables): enqueued one behind the other they are bound by launch latency, not by the GPU.  Here map i
// goes to stream i % N (the caller's, and N - 1 of the thread's side streams): every side stream starts behind the caller's
// stream and the caller's stream continues behind all of them; scratch freed meanwhile is handed back to the pool
// only after that join.  CS_KMAP_STREAMS=1: everything on the caller's stream.
int cs_kernelmap_build_many(int n, const cs_coordmap* const* in, const cs_coordmap* const* out, const int* kernel_size,
                            const int* transposed, void* stream, cs_kernelmap** km_out) {
  CS_REQUIRE(n >= 0 && (n == 0 || (in && out && kernel_size && transposed && km_out)), CS_ERR_INVALID,
             "cs_kernelmap_build_many: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  for (int i = 0; i < n; ++i) km_out[i] = nullptr;
  static const int n_streams = [] {
    const char* e = getenv("CS_KMAP_STREAMS");
    int v = e ? atoi(e) : 4;   // 2 / 3 / 4 / 5 streams: stress 6016 / 6201 / 6235 / 5608 clouds/s (the runtime maps streams to 4 queues)
    return v < 1 ? 1 : (v > 5 ? 5 : v);
  }();
  hipStream_t st[5] = {s, nullptr, nullptr, nullptr, nullptr};
  for (int k = 1; k < n_streams; ++k) st[k] = side_stream(k - 1);
  int ns = 1;
  while (ns < n_streams && st[ns]) ++ns;
  if (ns == 1 || n < 2) {
    for (int i = 0; i < n; ++i) {
      const int rc = kernelmap_build_on(in[i], out[i], kernel_size[i], transposed[i], s, &km_out[i]);
      if (rc != CS_OK) {
        for (int j = 0; j < i; ++j) { cs_kernelmap_free(km_out[j]); km_out[j] = nullptr; }
        return rc;
      }
    }
    return CS_OK;
  }
  // per-sample segments of the coordinate maps are made lazily by the first map that needs them: make them on the
  // caller's stream before the streams fork
  {
    std::vector<cs_coordmap*> need;
    for (int i = 0; i < n; ++i) {
      CS_REQUIRE(in[i] && out[i], CS_ERR_INVALID, "cs_kernelmap_build_many: NULL coordinate map");
      if (kernel_size[i] == 3 && getenv("CS_KMAP_GLOBAL") == nullptr) {
        need.push_back(const_cast<cs_coordmap*>(in[i]));
        need.push_back(const_cast<cs_coordmap*>(out[i]));
      }
    }
    const int rc_seg = ensure_segments_many(need.data(), (int)need.size(), s);
    if (rc_seg != CS_OK) return rc_seg;
  }
  struct Ev {
    hipEvent_t e = nullptr;
    ~Ev() { if (e) (void)hipEventDestroy(e); }
  } fork, join[5];
  CS_HIP_CHECK(hipEventCreateWithFlags(&fork.e, hipEventDisableTiming));
  CS_HIP_CHECK(hipEventRecord(fork.e, s));
  for (int k = 1; k < ns; ++k) {
    CS_HIP_CHECK(hipEventCreateWithFlags(&join[k].e, hipEventDisableTiming));
    CS_HIP_CHECK(hipStreamWaitEvent(st[k], fork.e, 0));
  }
  // CS_KMAP_ORDER_MANY=0: every map orders its own rows on its own stream (the round-3 path)
  static const bool order_all = !(getenv("CS_KMAP_ORDER_MANY") && getenv("CS_KMAP_ORDER_MANY")[0] == '0');
  const bool defer = order_all && n <= ORDER_MAX_MAPS;
  std::vector<const unsigned long long*> d_cnts(n, nullptr);
  pool_defer_begin();
  int rc = CS_OK;
  for (int i = 0; i < n && rc == CS_OK; ++i)
    rc = kernelmap_build_on(in[i], out[i], kernel_size[i], transposed[i], st[i % ns], &km_out[i], defer, &d_cnts[i]);
  // join (also on the error path: the side streams may hold work that reads scratch of this call)
  hipError_t je = hipSuccess;
  for (int k = 1; k < ns; ++k) {
    hipError_t e1 = hipEventRecord(join[k].e, st[k]);
    if (e1 == hipSuccess) e1 = hipStreamWaitEvent(s, join[k].e, 0);
    if (e1 != hipSuccess) {
      (void)hipStreamSynchronize(st[k]);
      je = e1;
    }
  }
  // the tiling order of all deferred maps in one pass on the caller's stream, behind the join and BEFORE the deferred
  // scratch (the maps' device counters) goes back to the pool
  if (rc == CS_OK && je == hipSuccess && defer) {
    std::vector<int> idx;
    for (int i = 0; i < n; ++i)
      if (km_out[i] && d_cnts[i]) idx.push_back(i);
    ProfScope prof("kmap", s);
    rc = order_many(km_out, d_cnts.data(), idx.data(), (int)idx.size(), s);
  }
  if (rc != CS_OK || je != hipSuccess) {
    for (int i = 0; i < n; ++i) {
      if (km_out[i]) cs_kernelmap_free(km_out[i]);
      km_out[i] = nullptr;
    }
    pool_defer_end();
    if (rc == CS_OK) {
      set_error("cs_kernelmap_build_many: joining the streams failed: %s", hipGetErrorString(je));
      rc = CS_ERR_HIP;
    }
    return rc;
  }
  pool_defer_end();
  return CS_OK;
}

int64_t cs_kernelmap_num_pairs(const cs_kernelmap* km) { return cs::kernelmap_pairs(km); }
int64_t cs_kernelmap_rows(const cs_kernelmap* km) { return km ? km->n_out : -1; }
const int32_t* cs_kernelmap_table(const cs_kernelmap* km) { return km ? km->d_nbr : nullptr; }

int64_t cs_kernelmap_export(const cs_kernelmap* km, int32_t* d_k, int32_t* d_in, int32_t* d_out,
                            int64_t capacity, void* stream) {
  CS_REQUIRE(km && d_k && d_in && d_out, CS_ERR_INVALID, "cs_kernelmap_export: NULL argument");
  const int64_t pairs = cs::kernelmap_pairs(km);
  CS_REQUIRE(pairs >= 0 && capacity >= pairs, CS_ERR_INVALID,
             "cs_kernelmap_export: capacity %lld < %lld pairs", (long long)capacity, (long long)pairs);
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
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
  CS_HIP_CHECK(download_sync(s));
  return pairs;
}

void cs_kernelmap_free(cs_kernelmap* km) {
  if (!km) return;
  pool_free(km->d_nbr);
  pool_free(km->d_rowlist);
  pool_free(km->d_nbr_sorted);
  pool_free(km->d_gmask);
  if (km->cnt_ready) {
    (void)hipEventSynchronize(km->cnt_ready);  // the slot must not be recycled under a pending copy
    if (km->prof_flop_per_pair > 0.0) (void)cs::kernelmap_pairs(km);   // hands the deferred work units to the profile
    (void)hipEventDestroy(km->cnt_ready);
  }
  if (km->prof_flop_per_pair > 0.0) cs::prof_maps_remove(km);
  count_slot_release(km->cnt_slot);
  delete km;
}

}  // extern "C"
