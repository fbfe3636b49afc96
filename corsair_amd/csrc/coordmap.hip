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

// ---- all coordinate levels of a batch in one pass (round 5: cs_coordmap_pyramid) -------------------------------------------
// The strided maps of a ResUNet batch (tensor strides 2, 4, 8) used to be a chain c1 -> c2 -> c4 -> c8 of
// insert / flag / scan / host round trip / emit, one link per level and one host wait per link, plus two more waits for the
// per-sample segments: six GPU-idle round trips and ~25 launches per forward.  The unique cells of stride 2^l in
// first-occurrence order are the same whether they are taken from the level below or from the stride-1 rows (the first
// stride-1 row of a coarse cell is also the first row of its finer cell, and every level keeps first-occurrence order), so
// all levels come from the stride-1 coordinates at once: one insert kernel (a thread reads its row once and has its four
// atomics in flight together), one flag kernel, ONE scan (the three flags packed as 21-bit fields of a 64-bit word), one
// emit kernel, the segment tables of all levels, and ONE host wait for the sizes.
constexpr int PYR_MAX_LEVELS = 4;
constexpr int PYR_FIELD = 21;                       // rows < 2^21: three counters in one 64-bit scan element
struct PyrLevel {
  uint64_t* keys;
  int32_t* vals;
  uint64_t mask;
  int32_t* coords;                                  // out (level >= 1)
  int32_t* seg;                                     // [n_batch + 1] or nullptr
  int cell;
};
struct PyrArgs {
  int n_levels;
  PyrLevel lv[PYR_MAX_LEVELS];
  int n_batch;
  int* status;                                      // [0] out of range, [1] duplicates, [2 + 2 l] not grouped, [3 + 2 l] unused
  int32_t* counts;                                  // rows of level l (l >= 1)
};

// (the pyramid kernels are only launched for power-of-two cells -- cs_coordmap_pyramid takes the chained path otherwise --:
// a mask, without the division arm of floor_to_cell and the uniform branch around it)
__device__ __forceinline__ int floor_to_cell2(int a, int cell) { return a & ~(cell - 1); }
__device__ __forceinline__ uint64_t pyr_key(int b, int x, int y, int z, int cell) {
  return pack_key(b, floor_to_cell2(x, cell), floor_to_cell2(y, cell), floor_to_cell2(z, cell));
}

__global__ __launch_bounds__(256) void k_pyr_insert(const int32_t* __restrict__ coords, int64_t n, const PyrArgs a) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coords)[i];
  // (the floor cells of an in-range coordinate are in range: |floor(x)| <= 32767 < 32768 needs x > -32768, checked here)
  if (!coord_in_range(c.x, c.y, c.z, c.w)) {
    atomicOr(&a.status[0], 1);
    return;
  }
  uint64_t key[PYR_MAX_LEVELS], slot[PYR_MAX_LEVELS];
  unsigned long long old[PYR_MAX_LEVELS];
#pragma unroll
  for (int l = 0; l < PYR_MAX_LEVELS; ++l)
    if (l < a.n_levels) {
      const int cell = a.lv[l].cell;
      // (a floor cell can fall on -32768 when the coordinate itself is in range: refused like cs_coordmap_stride does)
      if (l > 0 && !coord_in_range(c.x, floor_to_cell2(c.y, cell), floor_to_cell2(c.z, cell), floor_to_cell2(c.w, cell)))
        atomicOr(&a.status[0], 1);
      key[l] = pyr_key(c.x, c.y, c.z, c.w, cell);
      slot[l] = hash64(key[l]) & a.lv[l].mask;
      old[l] = atomicCAS((unsigned long long*)&a.lv[l].keys[slot[l]], (unsigned long long)kEmptyKey, (unsigned long long)key[l]);
    }
#pragma unroll
  for (int l = 0; l < PYR_MAX_LEVELS; ++l)
    if (l < a.n_levels) {
      while (old[l] != kEmptyKey && old[l] != key[l]) {
        slot[l] = (slot[l] + 1) & a.lv[l].mask;
        old[l] = atomicCAS((unsigned long long*)&a.lv[l].keys[slot[l]], (unsigned long long)kEmptyKey, (unsigned long long)key[l]);
      }
      if (l == 0 && old[l] == key[l]) atomicAdd(&a.status[1], 1);   // level 0 keeps every row: a second one is a duplicate
      atomicMin(&a.lv[l].vals[slot[l]], (int32_t)i);
    }
}

// packed flags: bit field l - 1 = 1 iff row i is the first (minimum) row of its level-l cell
__global__ __launch_bounds__(256) void k_pyr_flag(const int32_t* __restrict__ coords, int64_t n, const PyrArgs a,
                                                  unsigned long long* __restrict__ flag) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coords)[i];
  unsigned long long f = 0;
#pragma unroll
  for (int l = 1; l < PYR_MAX_LEVELS; ++l)
    if (l < a.n_levels) {
      const int32_t v = hash_lookup(a.lv[l].keys, a.lv[l].vals, a.lv[l].mask, pyr_key(c.x, c.y, c.z, c.w, a.lv[l].cell));
      if (v == (int32_t)i) f |= 1ULL << (PYR_FIELD * (l - 1));
    }
  flag[i] = f;
}

__global__ __launch_bounds__(256) void k_pyr_emit(const int32_t* __restrict__ coords, int64_t n, const PyrArgs a,
                                                  const unsigned long long* __restrict__ flag,
                                                  const unsigned long long* __restrict__ pos) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long f = flag[i], p = pos[i];
  if (i == n - 1) {
#pragma unroll
    for (int l = 1; l < PYR_MAX_LEVELS; ++l)
      if (l < a.n_levels)
        a.counts[l] = (int32_t)(((p + f) >> (PYR_FIELD * (l - 1))) & ((1ULL << PYR_FIELD) - 1));
  }
  if (!f) return;
  const int4 c = reinterpret_cast<const int4*>(coords)[i];
#pragma unroll
  for (int l = 1; l < PYR_MAX_LEVELS; ++l)
    if (l < a.n_levels && ((f >> (PYR_FIELD * (l - 1))) & 1ULL)) {
      const int cell = a.lv[l].cell;
      const int x = floor_to_cell2(c.y, cell), y = floor_to_cell2(c.z, cell), z = floor_to_cell2(c.w, cell);
      const int32_t o = (int32_t)((p >> (PYR_FIELD * (l - 1))) & ((1ULL << PYR_FIELD) - 1));
      reinterpret_cast<int4*>(a.lv[l].coords)[o] = make_int4(c.x, x, y, z);
      const uint64_t key = pack_key(c.x, x, y, z);
      uint64_t slot = hash64(key) & a.lv[l].mask;
      while (a.lv[l].keys[slot] != key) slot = (slot + 1) & a.lv[l].mask;
      a.lv[l].vals[slot] = o;   // exactly one row per key reaches here
    }
}

// per-sample segments of every level in one launch: grid.y = level; rows of level l >= 1 are counted on the device
__global__ __launch_bounds__(256) void k_pyr_segments(const int32_t* __restrict__ coords0, int64_t n0, const PyrArgs a) {
  const int l = blockIdx.y;
  if (l >= a.n_levels || !a.lv[l].seg) return;
  const int32_t* coords = l == 0 ? coords0 : a.lv[l].coords;
  const int64_t n = l == 0 ? n0 : (int64_t)a.counts[l];
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= n) return;
  int32_t* seg = a.lv[l].seg;
  const int b = coords[4 * i];
  const int prev = i > 0 ? coords[4 * (i - 1)] : -1;
  if (b < prev || b >= a.n_batch || b < 0) {
    atomicOr(&a.status[2 + 2 * l], 1);  // not grouped by sample (or more samples than announced): the global-table path
    return;
  }
  for (int bb = prev + 1; bb <= b; ++bb) seg[bb] = (int32_t)i;
  if (i == n - 1)
    for (int bb = b + 1; bb <= a.n_batch; ++bb) seg[bb] = (int32_t)n;
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

// ------------------------------------------------------------------------------------------------
// Kernel maps built in LDS (k_level_maps, below).  Coordinates of a batch are grouped by sample (collate order), and a
// kernel-map probe never leaves its sample, so one workgroup loads the in-map coordinates of ONE sample into an LDS hash
// table (30-bit keys relative to the sample's bounding box, 16-bit local row) and answers the probes of its slice of output
// rows from LDS -- no random global access at all.  Samples with more rows than the table holds or bounding boxes wider
// than 1023 cells are probed in the level's global table by the same workgroup; batches that are not grouped by sample use
// the global kernel (k_build_nbr) for everything.  (The per-map kernels of rounds 1-4, k_build_nbr_lds / _flagged, were
// removed in round 5 after the A/B against k_level_maps: profiles/r5d_kmap_fused_ab.txt, HISTORY.md.)
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

// Per-sample segment tables of coordinate maps (computed once, cached on the map): for several maps at once --
// TWO host round trips in total (last batch index of every map; the grouping
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
// ---- the tiling order of ALL kernel maps of a batch in one pass (round 4) ---------------------------------------------
// Ten maps of a batch used to mean ten row-key launches, ten device sorts (5 - 20 launches each: hipcub takes a block-sort +
// merge path for the mid-size maps) and ten table launches: 115 launches per stress batch, 60 per chair batch, bound by
// launch latency.  Here the keys of all maps go into ONE array -- key = map index << 27 | Gray rank, so one radix sort
// leaves every map's rows contiguous and ordered -- and one launch writes all sorted tables.  The descriptors travel as a
// kernel argument (no upload).
constexpr int ORDER_MAX_MAPS = 16;
struct OrderMap {
  const int32_t* nbr;
  uint32_t* gmask;
  int32_t* rowlist;
  const unsigned long long* d_cnt;
  unsigned long long* host_cnt;
  int64_t n_out, base, n_groups;
  uint32_t kblk0, tblk0;   // first workgroup of this map in the key / finish launch
  uint32_t tag;            // map index << 27: the sort keeps every map's rows together, in the order of `base`
  int need_keys;           // 1: k_row_keys_all computes this map's keys from its table (maps the level kernel did not build)
};
struct OrderTab {
  int n;
  OrderMap m[ORDER_MAX_MAPS];
};
// Bit j of a row's tiling mask stands for kernel offset KORDER[j]: the centre, the six face neighbours, the twelve edge
// neighbours, the eight corner neighbours (by |d|_1, then by k).  The frequent offsets sit in the low bits, the rare corner
// offsets decide the coarse order of the Gray ranks -- rows that need a rare offset end up in the same 32-row groups.  Against
// the Gray rank of the mask in plain k order the executed / useful matrix work of the convolutions drops 1.190 -> 1.160 on the
// 64-cloud stress batch (stride-1 maps 1.31 -> 1.22), 1.270 -> 1.231 on a 24-cloud batch (tools/exec_ratio_cpu.py).
__device__ const int8_t KORDER[32] = {13, 4, 10, 12, 14, 16, 22, 1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25, 0, 2, 6, 8, 18, 20, 24, 26, 0, 0, 0, 0, 0};
__device__ __forceinline__ uint32_t unpermute_mask(uint32_t m) {   // tiling mask (bit j = offset KORDER[j]) -> bit k = offset k
  uint32_t out = 0;
#pragma unroll
  for (int j = 0; j < 27; ++j) out |= ((m >> j) & 1u) << KORDER[j];
  return out;
}
__device__ __forceinline__ uint32_t gray_rank(uint32_t m) {   // r with r ^ (r >> 1) == m
  m ^= m >> 1;
  m ^= m >> 2;
  m ^= m >> 4;
  m ^= m >> 8;
  m ^= m >> 16;
  return m;
}
__global__ __launch_bounds__(256) void k_row_keys_all(const OrderTab tab, uint32_t* __restrict__ key,
                                                      int32_t* __restrict__ row) {
  int i = 0;
  while (i + 1 < tab.n && blockIdx.x >= tab.m[i + 1].kblk0) ++i;
  const OrderMap& mp = tab.m[i];
  if (!mp.need_keys) return;
  const int64_t o = (int64_t)(blockIdx.x - mp.kblk0) * 256 + threadIdx.x;
  if (o >= mp.n_out) return;
  uint32_t m = 0;
  for (int j = 0; j < 27; ++j) m |= (mp.nbr[o * 27 + KORDER[j]] >= 0 ? 1u : 0u) << j;
  key[mp.base + o] = mp.tag | gray_rank(m);
  row[mp.base + o] = (int32_t)o;
}
// behind the sort: every map's row list (its slice of the sorted payload), the group masks, and the pair count of the
// build kernels handed to its page-locked slot (a device -> host copy of 8 bytes was one blit-kernel launch per map)
__global__ __launch_bounds__(256) void k_order_finish(const OrderTab tab, const uint32_t* __restrict__ key_sorted,
                                                      const int32_t* __restrict__ row_sorted) {
  int i = 0;
  while (i + 1 < tab.n && blockIdx.x >= tab.m[i + 1].tblk0) ++i;
  const OrderMap& mp = tab.m[i];
  const int64_t e = (int64_t)(blockIdx.x - mp.tblk0) * 256 + threadIdx.x;
  if (e == 0 && mp.host_cnt) __hip_atomic_store(mp.host_cnt, *mp.d_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int32_t* rl = row_sorted + mp.base;
  const uint32_t* ks = key_sorted + mp.base;
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
    mp.gmask[e] = unpermute_mask(m);
  }
}

// ---- every kernel map that probes ONE coordinate level, in one launch (round 5) ---------------------------------------
// The ten maps of a ResUNet batch probe four tables: level L serves L -> L (submanifold), L -> 2L (strided: output rows of
// level 2L look for level-L rows) and the transposed 2L' -> L' map whose INPUT is level L (output rows of the finer level
// look for level-L rows).  Round 4 launched one kernel per map, every slice of every launch re-inserting its sample into LDS
// (2.2 ms of kernel-map work per batch-64 stress forward, 40 - 48 % of the convolutions' time).  Here a
// workgroup (slice x, sample b) builds the sample's table ONCE and answers the probes of all jobs from it:
//   * the sample's coordinates are read once (16-byte loads, held in registers for the bounding box AND the insert);
//   * a thread owns one (row, offset) with 32 lanes per row (27 used): the 27 results of a row are one contiguous 108-byte
//     store, the row's presence mask is one ballot -- so the tiling key (Gray rank of the mask) and the pair count come out
//     of the builder and nothing has to read the table again (k_row_keys_all re-read 108 B per row);
//   * coordinates of the next four rows of a thread are requested before the first is probed (the probe loop was a chain
//     of dependent global load -> LDS probe -> store per entry, with four waves per SIMD to hide it);
//   * a sample that does not fit the LDS table (too many rows, bounding box wider than 1023 cells) is probed in the level's
//     GLOBAL table by the same workgroup -- no flag array, no second launch that usually does nothing.
constexpr int LEVEL_MAX_JOBS = 4;
struct LevelJob {
  const int32_t* out_coords;   // [n_out,4] rows grouped by sample
  const int32_t* out_seg;      // [n_batch + 1]
  int32_t* nbr;                // [n_out,27]
  uint32_t* key;               // tiling keys / row payload of this map inside the batch's sort arrays (nullptr: not wanted)
  int32_t* row;
  unsigned long long* cnt;     // pair counter
  int step, sign;
  uint32_t tag;                // map index << 27
};
struct LevelArgs {
  int n_jobs;
  LevelJob job[LEVEL_MAX_JOBS];
  const int32_t* in_coords;
  const int32_t* in_seg;
  int unit, ushift;
  const uint64_t* gkeys;       // the level's global table (cs_coordmap): oversized samples
  const int32_t* gvals;
  uint64_t gmask;
  unsigned long long* trace;   // CS_KMAP_TRACE=1: [workgroup][8] phase stamps (100 MHz), else nullptr
};

template <int SLOTS, int NT>
__global__ __launch_bounds__(NT) void k_level_maps(const LevelArgs a) {
  // The table: NB buckets of four 4-byte keys (one ds_read_b128 fetches a bucket) + four 2-byte local rows.  A key goes into
  // the first free slot of its bucket, slots fill in order, a full bucket spills into the next one.  A probe reads ONE
  // bucket and is done unless that bucket is full and does not hold the key (6 % of the probes at the usual load of 1.5
  // keys per bucket) -- so the first reads of several probes can be in flight together: the round-4 kernels chased one
  // linear-probing chain per lane with four waves per SIMD to hide it, and were bound by that latency (trace: 142 us of
  // probes against 22 us of table build per stride-1 level of a stress batch).
  constexpr int NB = SLOTS / 4;
  constexpr int MAX_ROWS = SLOTS / 8 * 5;            // load factor <= 0.625 (2.5 keys per bucket)
  constexpr int RPT = (MAX_ROWS + NT - 1) / NT;      // in-rows a thread holds in registers
  constexpr int ROWS_PER_PASS = NT / 32;             // output rows a workgroup probes at a time
  constexpr int U = 8;                               // probes a lane keeps in flight
  __shared__ __attribute__((aligned(16))) uint32_t keys[SLOTS];
  __shared__ __attribute__((aligned(8))) uint16_t vals[SLOTS];
  __shared__ int bmin[3], bmax[3];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int i0 = a.in_seg[b], i1 = a.in_seg[b + 1];
  const int n_in = i1 - i0;
  // rows of this slice per job; a workgroup without any leaves before the first barrier
  auto slice = [&](int j, int& lo_, int& hi_) {
    const int o0 = a.job[j].out_seg[b], o1 = a.job[j].out_seg[b + 1];
    const int per = (o1 - o0 + (int)gridDim.x - 1) / (int)gridDim.x;
    lo_ = o0 + (int)blockIdx.x * per;
    hi_ = min(o1, lo_ + per);
  };
  bool any = false;
  for (int j = 0; j < a.n_jobs; ++j) {
    int lo_, hi_;
    slice(j, lo_, hi_);
    any = any || lo_ < hi_;
  }
  if (!any) return;
  unsigned long long* tr = a.trace ? a.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
  if (tr && tid == 0) tr[0] = __builtin_amdgcn_s_memrealtime();
  // cells are counted in units of the tensor stride, a power of two for every level the network makes: shifts and masks.
  // (A stride that is not one goes through the global table below.  With the general `v / unit` kept as the other arm of a
  // select, every probe carried six uniform branches around six 30-instruction integer divisions.)
  const int ushift = a.ushift, unit = a.unit;
  auto udiv = [&](int v) { return v >> ushift; };                    // v >= 0
  auto umult = [&](int v) { return (v & (unit - 1)) == 0; };
  auto bucket_of = [&](uint32_t key) { return __umulhi(key * 2654435761u, (uint32_t)NB); };   // [0, NB)

  bool use_lds = n_in <= MAX_ROWS && ushift >= 0;    // workgroup-uniform
  int4 c[RPT];
  if (use_lds) {
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int i = i0 + tid + j * NT;
      c[j] = make_int4(0, 0, 0, 0);
      if (i < i1) c[j] = reinterpret_cast<const int4*>(a.in_coords)[i];
    }
    if (tid < 3) {
      bmin[tid] = 0x7fffffff;
      bmax[tid] = -0x7fffffff;
    }
    for (int i = tid; i < NB; i += NT)
      reinterpret_cast<uint4*>(keys)[i] = make_uint4(LDS_EMPTY, LDS_EMPTY, LDS_EMPTY, LDS_EMPTY);
    __syncthreads();
    // bounding box: per thread (registers), per wave (shuffles), one LDS atomic per wave and axis
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      if (i0 + tid + j * NT < i1) {
        lo[0] = min(lo[0], c[j].y);
        hi[0] = max(hi[0], c[j].y);
        lo[1] = min(lo[1], c[j].z);
        hi[1] = max(hi[1], c[j].z);
        lo[2] = min(lo[2], c[j].w);
        hi[2] = max(hi[2], c[j].w);
      }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        lo[ax] = min(lo[ax], __shfl_xor(lo[ax], off));
        hi[ax] = max(hi[ax], __shfl_xor(hi[ax], off));
      }
      if ((tid & 63) == 0) {
        atomicMin(&bmin[ax], lo[ax]);
        atomicMax(&bmax[ax], hi[ax]);
      }
    }
    __syncthreads();
  }
  int mx = 0, my = 0, mz = 0, ex = -1, ey = -1, ez = -1;   // an empty sample: no cell is inside the box
  if (use_lds && n_in > 0) {
    mx = bmin[0];
    my = bmin[1];
    mz = bmin[2];
    ex = udiv(bmax[0] - mx);
    ey = udiv(bmax[1] - my);
    ez = udiv(bmax[2] - mz);
    if (ex > 1023 || ey > 1023 || ez > 1023) use_lds = false;   // (uniform: every thread reads the same box)
  }
  if (tr && tid == 0) tr[1] = __builtin_amdgcn_s_memrealtime();
  if (use_lds) {
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int i = i0 + tid + j * NT;
      if (i < i1) {
        const uint32_t key = (uint32_t)udiv(c[j].y - mx) | ((uint32_t)udiv(c[j].z - my) << 10) |
                             ((uint32_t)udiv(c[j].w - mz) << 20);
        uint32_t slot = bucket_of(key) * 4u;   // coordinates are unique inside a map: every key claims its own slot
        while (atomicCAS(&keys[slot], LDS_EMPTY, key) != LDS_EMPTY) slot = slot + 1 == SLOTS ? 0 : slot + 1;
        vals[slot] = (uint16_t)(i - i0);
      }
    }
    __syncthreads();
  }
  if (tr && tid == 0) tr[2] = __builtin_amdgcn_s_memrealtime();

  const int kj = tid & 31;                            // lane j of a row handles offset KORDER[j]; lanes 27..31 idle
  const int k = kj < 27 ? (int)KORDER[kj] : 27;       // (the ballot bit of lane j is then bit j of the tiling mask)
  const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
  const int sh = tid & 32;                            // this row's half of the wave ballot
#pragma unroll 1
  for (int j = 0; j < a.n_jobs; ++j) {
    const LevelJob& J = a.job[j];
    const int ox = J.sign * dx * J.step - mx, oy = J.sign * dy * J.step - my, oz = J.sign * dz * J.step - mz;
    const int4* oc = reinterpret_cast<const int4*>(J.out_coords);
    int32_t* const nbr = J.nbr;
    uint32_t* const jkey = J.key;
    int32_t* const jrow = J.row;
    const uint32_t tag = J.tag;
    int found = 0;
    int s0j, s1j;
    slice(j, s0j, s1j);
    // the row's mask is a ballot, its tiling key and row payload come from the lane of offset 0
    auto emit = [&](int o, int32_t v) {
      if (k < 27) nbr[(int64_t)o * 27 + k] = v;
      const uint32_t m = (uint32_t)(__ballot(v >= 0) >> sh) & 0x7ffffffu;
      if (kj == 0) {
        found += __popc(m);
        if (jkey) {
          jkey[o] = tag | gray_rank(m);
          jrow[o] = o;
        }
      }
    };
    if (use_lds) {
      for (int ob = s0j + (tid >> 5); ob < s1j; ob += U * ROWS_PER_PASS) {
        int4 p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int o = ob + u * ROWS_PER_PASS;
          p[u] = make_int4(0, 0, 0, 0);
          if (o < s1j) p[u] = oc[o];
        }
        uint32_t key[U], bk[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int x = p[u].y + ox, y = p[u].z + oy, z = p[u].w + oz;      // relative to the box corner
          const int cx = udiv(max(x, 0)), cy = udiv(max(y, 0)), cz = udiv(max(z, 0));
          live[u] = k < 27 && ob + u * ROWS_PER_PASS < s1j && (x | y | z) >= 0 && umult(x) && umult(y) && umult(z) &&
                    cx <= ex && cy <= ey && cz <= ez;
          key[u] = (uint32_t)cx | ((uint32_t)cy << 10) | ((uint32_t)cz << 20);
          bk[u] = live[u] ? bucket_of(key[u]) : 0u;
        }
        uint4 K[U];
#pragma unroll
        for (int u = 0; u < U; ++u) K[u] = reinterpret_cast<const uint4*>(keys)[bk[u]];   // U reads in flight
        int32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int sl = K[u].x == key[u] ? 0 : K[u].y == key[u] ? 1 : K[u].z == key[u] ? 2 : K[u].w == key[u] ? 3 : -1;
          uint32_t bb = bk[u];
          if (live[u] && sl < 0 && K[u].w != LDS_EMPTY) {     // full bucket without the key: it may have spilled on
            while (true) {
              bb = bb + 1 == NB ? 0 : bb + 1;
              const uint4 Kn = reinterpret_cast<const uint4*>(keys)[bb];
              sl = Kn.x == key[u] ? 0 : Kn.y == key[u] ? 1 : Kn.z == key[u] ? 2 : Kn.w == key[u] ? 3 : -1;
              if (sl >= 0 || Kn.w == LDS_EMPTY) break;
            }
          }
          v[u] = (live[u] && sl >= 0) ? i0 + (int32_t)vals[bb * 4 + sl] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int o = ob + u * ROWS_PER_PASS;
          if (o < s1j) emit(o, v[u]);
        }
      }
    } else {
      // a sample too large or too wide for the LDS table: the same probes in the level's global table
      for (int o = s0j + (tid >> 5); o < s1j; o += ROWS_PER_PASS) {
        const int4 pp = oc[o];
        int32_t v = -1;
        if (k < 27) {
          const int X = pp.y + ox + mx, Y = pp.z + oy + my, Z = pp.w + oz + mz;
          if (coord_in_range(pp.x, X, Y, Z)) v = hash_lookup(a.gkeys, a.gvals, a.gmask, pack_key(pp.x, X, Y, Z));
        }
        emit(o, v);
      }
    }
    // pair count: wave reduce, one atomic per wave
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) found += __shfl_xor(found, off);
    if ((tid & 63) == 0 && found) atomicAdd(J.cnt, (unsigned long long)found);
    if (tr && tid == 0) tr[3 + j] = __builtin_amdgcn_s_memrealtime();
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

/* All coordinate levels of a batch: out[0] = the map cs_coordmap_create(d_coords, n, tensor_stride) returns, out[l] =
 * cs_coordmap_stride(out[l - 1], 2) -- the same coordinates, row order and tables -- made from the stride-1 rows in one pass
 * with ONE host wait (see k_pyr_insert).  n_batch > 0: rows are announced to be grouped by sample with batch indices
 * < n_batch, and the per-sample segments of every level are made in the same pass (a violated announcement only costs the
 * LDS kernel-map path, never correctness).  Batches of 2^21 rows or more take the chained calls. */
int cs_coordmap_pyramid(const int32_t* d_coords, int64_t n, int tensor_stride, int n_levels, int n_batch, void* stream,
                        cs_coordmap** out) {
  CS_REQUIRE(out && n_levels >= 1 && n_levels <= PYR_MAX_LEVELS, CS_ERR_INVALID, "cs_coordmap_pyramid: bad arguments");
  for (int l = 0; l < n_levels; ++l) out[l] = nullptr;
  CS_REQUIRE(n >= 0 && n < (1LL << 30), CS_ERR_INVALID, "cs_coordmap_pyramid: bad row count %lld", (long long)n);
  CS_REQUIRE(n == 0 || d_coords, CS_ERR_INVALID, "cs_coordmap_pyramid: coords is NULL");
  CS_REQUIRE(tensor_stride >= 1, CS_ERR_INVALID, "cs_coordmap_pyramid: bad tensor stride");
  static const bool chained = getenv("CS_PYRAMID") && getenv("CS_PYRAMID")[0] == '0';
  const bool pow2 = (tensor_stride & (tensor_stride - 1)) == 0;   // cells of every level are tensor_stride << l
  if (n == 0 || n >= (1LL << PYR_FIELD) || chained || !pow2) {
    int rc = cs_coordmap_create(d_coords, n, tensor_stride, stream, &out[0]);
    for (int l = 1; l < n_levels && rc == CS_OK; ++l) rc = cs_coordmap_stride(out[l - 1], 2, stream, &out[l]);
    if (rc != CS_OK)
      for (int l = 0; l < n_levels; ++l) {
        cs_coordmap_free(out[l]);
        out[l] = nullptr;
      }
    return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  ProfScope prof("kmap", s);
  if (n_batch > 65536) n_batch = 0;
  cs_coordmap* m[PYR_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
  auto fail = [&](int code) {
    for (int l = 0; l < n_levels; ++l) cs_coordmap_free(m[l]);
    return code;
  };
  PyrArgs a;
  a.n_levels = n_levels;
  a.n_batch = n_batch;
  PoolBuf<int> status(2 + 2 * PYR_MAX_LEVELS);
  PoolBuf<int32_t> counts(PYR_MAX_LEVELS);
  PoolBuf<unsigned long long> flag(n), pos(n);
  if (!status.p || !counts.p || !flag.p || !pos.p) {
    set_error("cs_coordmap_pyramid: scratch allocation failed");
    return CS_ERR_HIP;
  }
  a.status = status.p;
  a.counts = counts.p;
  for (int l = 0; l < n_levels; ++l) {
    m[l] = new cs_coordmap();
    m[l]->tensor_stride = tensor_stride << l;
    // (the coarse levels have at most n rows: tables and coordinate arrays are sized for that)
    const int rc = alloc_table(m[l], n, s);
    if (rc != CS_OK) return fail(rc);
    m[l]->d_coords = (int32_t*)pool_alloc((size_t)n * 4 * sizeof(int32_t));
    if (n_batch > 0) m[l]->d_seg = (int32_t*)pool_alloc((size_t)(n_batch + 1) * sizeof(int32_t));
    if (!m[l]->d_coords || (n_batch > 0 && !m[l]->d_seg)) {
      set_error("cs_coordmap_pyramid: coordinate allocation failed");
      return fail(CS_ERR_HIP);
    }
    a.lv[l].keys = m[l]->d_keys;
    a.lv[l].vals = m[l]->d_vals;
    a.lv[l].mask = m[l]->capacity - 1;
    a.lv[l].coords = m[l]->d_coords;
    a.lv[l].seg = m[l]->d_seg;
    a.lv[l].cell = tensor_stride << l;
  }
  hipError_t e = hipMemcpyAsync(m[0]->d_coords, d_coords, (size_t)n * 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemsetAsync(status.p, 0, sizeof(int) * (2 + 2 * PYR_MAX_LEVELS), s);
  if (e == hipSuccess) e = hipMemsetAsync(counts.p, 0, sizeof(int32_t) * PYR_MAX_LEVELS, s);
  for (int l = 0; l < n_levels && e == hipSuccess && n_batch > 0; ++l)
    e = hipMemsetAsync(m[l]->d_seg, 0, (size_t)(n_batch + 1) * sizeof(int32_t), s);
  if (e != hipSuccess) {
    set_error("cs_coordmap_pyramid: %s", hipGetErrorString(e));
    return fail(CS_ERR_HIP);
  }
  const unsigned g = (unsigned)ceil_div(n, 256);
  // level 0's cell is 1 in units of its own stride: its key is the coordinate itself
  PyrArgs a_ins = a;
  a_ins.lv[0].cell = 1;
  hipLaunchKernelGGL(k_pyr_insert, dim3(g), dim3(256), 0, s, m[0]->d_coords, n, a_ins);
  if (n_levels > 1) {
    hipLaunchKernelGGL(k_pyr_flag, dim3(g), dim3(256), 0, s, m[0]->d_coords, n, a, flag.p);
    size_t tmp_bytes = 0;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, flag.p, pos.p, (int)n, s);
    PoolBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    if (e == hipSuccess && !tmp.p) e = hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, flag.p, pos.p, (int)n, s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_pyr_emit, dim3(g), dim3(256), 0, s, m[0]->d_coords, n, a, flag.p, pos.p);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess && n_batch > 0) {
    hipLaunchKernelGGL(k_pyr_segments, dim3(g, (unsigned)n_levels), dim3(256), 0, s, m[0]->d_coords, n, a);
    e = hipGetLastError();
  }
  int h_status[2 + 2 * PYR_MAX_LEVELS] = {0};
  int32_t h_counts[PYR_MAX_LEVELS] = {0};
  if (e == hipSuccess) e = download_async(h_status, status.p, sizeof(h_status), s);
  if (e == hipSuccess) e = download_async(h_counts, counts.p, sizeof(h_counts), s);
  if (e == hipSuccess) e = download_sync(s);
  if (e != hipSuccess) {
    set_error("cs_coordmap_pyramid: %s", hipGetErrorString(e));
    return fail(CS_ERR_HIP);
  }
  if (h_status[0]) {
    set_error("cs_coordmap_create: coordinate out of the supported range (|x|,|y|,|z| < 32768, 0 <= batch < 65536)");
    return fail(CS_ERR_RANGE);
  }
  if (h_status[1]) {
    set_error("cs_coordmap_create: %d duplicate coordinate rows (quantise the cloud first)", h_status[1]);
    return fail(CS_ERR_DUPLICATE);
  }
  for (int l = 0; l < n_levels; ++l) {
    m[l]->n = l == 0 ? n : (int64_t)h_counts[l];
    if (n_batch > 0) {
      if (h_status[2 + 2 * l] == 0) {
        m[l]->n_batch = n_batch;
        m[l]->seg_state = 1;
      } else {                      // not grouped by sample: the global-table path serves this map
        pool_free(m[l]->d_seg);
        m[l]->d_seg = nullptr;
        m[l]->seg_state = -1;
      }
    }
    out[l] = m[l];
  }
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

// ---- building the kernel maps of a batch -------------------------------------------------------------------------------
// Geometry of one map: which way the probes go.  Plain: out row at p looks for in rows at p + d * step (step = the in
// map's tensor stride); transposed (in = the coarser map): p - d * step with step = the out map's stride.
static int map_geometry(const cs_coordmap* in, const cs_coordmap* out, int kernel_size, int transposed, int* step, int* sign) {
  CS_REQUIRE(in && out, CS_ERR_INVALID, "cs_kernelmap_build: NULL argument");
  CS_REQUIRE(kernel_size == 3 || kernel_size == 1, CS_ERR_UNSUPPORTED,
             "cs_kernelmap_build: kernel_size %d not supported (1 or 3)", kernel_size);
  if (!transposed) {
    CS_REQUIRE(out->tensor_stride == in->tensor_stride || out->tensor_stride == 2 * in->tensor_stride, CS_ERR_INVALID,
               "cs_kernelmap_build: tensor strides %d -> %d not supported", in->tensor_stride, out->tensor_stride);
    *step = in->tensor_stride;
    *sign = +1;
  } else {
    CS_REQUIRE(in->tensor_stride == 2 * out->tensor_stride, CS_ERR_INVALID,
               "cs_kernelmap_build: transposed map needs in stride == 2 * out stride (%d, %d)", in->tensor_stride,
               out->tensor_stride);
    *step = out->tensor_stride;
    *sign = -1;
  }
  return CS_OK;
}

static int ushift_of(int ts) {
  if (ts <= 0 || (ts & (ts - 1)) != 0) return -1;
  int sh = 0;
  while ((1 << sh) < ts) ++sh;
  return sh;
}

struct MapPlan {
  cs_kernelmap* km = nullptr;
  const cs_coordmap* in = nullptr;
  const cs_coordmap* out = nullptr;
  int step = 0, sign = 0;
  unsigned long long* d_cnt = nullptr;   // device pair counter (zeroed before the streams fork)
  bool fused = false;                    // built by k_level_maps (keys come out of the builder)
  bool ordered = false;                  // gets a tiling order (kvol 27, n_out > 0)
  int64_t base = 0;                      // first element of this map in the batch's sort arrays
};

// CS_KMAP_TRACE=1: phase stamps of one k_level_maps launch, averaged over its workgroups (synchronises; diagnostics)
static void level_trace_report(unsigned long long* d_trace, int wgs, int n_jobs, int slots, int nb, int slices,
                               int64_t n_in, hipStream_t s) {
  std::vector<unsigned long long> h((size_t)wgs * 8);
  if (hipStreamSynchronize(s) != hipSuccess) return;
  if (hipMemcpy(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
  double ph[8] = {0};
  int live = 0;
  unsigned long long t_min = ~0ULL, t_max = 0;
  for (int w = 0; w < wgs; ++w) {
    const unsigned long long* t = &h[(size_t)w * 8];
    if (!t[0]) continue;
    ++live;
    for (int i = 1; i < 3 + n_jobs; ++i) ph[i] += (double)(t[i] - t[i - 1]) * 0.01;   // 100 MHz -> us
    t_min = std::min(t_min, t[0]);
    t_max = std::max(t_max, t[2 + n_jobs]);
  }
  if (!live) return;
  fprintf(stderr, "[kmap trace] level n_in %lld, %d samples x %d slices, %d slots, %d jobs: box+fill %.1f us, insert %.1f us",
          (long long)n_in, nb, slices, slots, n_jobs, ph[1] / live, ph[2] / live);
  for (int j = 0; j < n_jobs; ++j) fprintf(stderr, ", job %d %.1f us", j, ph[3 + j] / live);
  fprintf(stderr, "; %d workgroups, launch span %.1f us\n", live, (double)(t_max - t_min) * 0.01);
}

// all fused maps that probe the table of `in`, on stream s: jobs[0..nj) index into plans
static int launch_level(const cs_coordmap* in, const MapPlan* plans, const int* jobs, int nj, uint32_t* key, int32_t* row,
                        hipStream_t s) {
  cs_coordmap* in_m = const_cast<cs_coordmap*>(in);
  const int nb = in_m->n_batch;
  LevelArgs a;
  a.n_jobs = nj;
  for (int j = 0; j < nj; ++j) {
    const MapPlan& p = plans[jobs[j]];
    LevelJob& J = a.job[j];
    J.out_coords = p.out->d_coords;
    J.out_seg = p.out->d_seg;
    J.nbr = p.km->d_nbr;
    J.key = p.ordered ? key + p.base : nullptr;
    J.row = p.ordered ? row + p.base : nullptr;
    J.cnt = p.d_cnt;
    J.step = p.step;
    J.sign = p.sign;
    J.tag = (uint32_t)(jobs[j] % ORDER_MAX_MAPS) << 27;
  }
  a.in_coords = in->d_coords;
  a.in_seg = in_m->d_seg;
  a.unit = in->tensor_stride;
  a.ushift = ushift_of(in->tensor_stride);
  a.gkeys = in->d_keys;
  a.gvals = in->d_vals;
  a.gmask = in->capacity - 1;
  a.trace = nullptr;
  // workgroups per level: one per CU and more (a slice re-inserts its sample: 512 workgroups build every table twice as
  // often as 256 for the same probes); CS_KMAP_WGS overrides
  static const int slice_wgs = getenv("CS_KMAP_WGS") ? std::max(atoi(getenv("CS_KMAP_WGS")), 1) : 256;
  int slices = slice_wgs / (nb > 0 ? nb : 1);
  slices = std::min(std::max(slices, 1), 8);
  // table size from the mean in-sample size (2.5x headroom; a larger sample is probed in the global table)
  const int64_t need = (in->n / (nb > 0 ? nb : 1)) * 5 / 2;
  static const bool small_tables = !(getenv("CS_KMAP_SMALL") && getenv("CS_KMAP_SMALL")[0] == '0');
  const int slots = (small_tables && need <= 2048 / 8 * 5) ? 2048 : (small_tables && need <= 8192 / 8 * 5) ? 8192 : LDS_SLOTS_MAX;
  static const bool trace_on = getenv("CS_KMAP_TRACE") && getenv("CS_KMAP_TRACE")[0] == '1';
  const int wgs = slices * nb;
  if (trace_on) {
    if (hipMalloc(reinterpret_cast<void**>(&a.trace), (size_t)wgs * 64) != hipSuccess) a.trace = nullptr;
    if (a.trace) (void)hipMemsetAsync(a.trace, 0, (size_t)wgs * 64, s);
  }
  const dim3 grid((unsigned)slices, (unsigned)nb);
  if (slots == 2048)
    hipLaunchKernelGGL((k_level_maps<2048, 256>), grid, dim3(256), 0, s, a);
  else if (slots == 8192)
    hipLaunchKernelGGL((k_level_maps<8192, 512>), grid, dim3(512), 0, s, a);
  else
    hipLaunchKernelGGL((k_level_maps<LDS_SLOTS_MAX, 1024>), grid, dim3(1024), 0, s, a);
  CS_LAUNCH_CHECK();
  if (a.trace) {
    level_trace_report(a.trace, wgs, nj, slots, nb, slices, in->n, s);
    (void)hipFree(a.trace);
  }
  return CS_OK;
}

// one map through the GLOBAL table (1x1 "maps", rows not grouped by sample, CS_KMAP_GLOBAL)
static int launch_single(const MapPlan& p, hipStream_t s) {
  cs_kernelmap* km = p.km;
  const int64_t total = km->n_out * km->kvol;
  if (total == 0) return CS_OK;
  hipLaunchKernelGGL(k_build_nbr, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, p.out->d_coords, km->n_out, km->kvol,
                     p.step, p.sign, p.in->d_keys, p.in->d_vals, p.in->capacity - 1, km->d_nbr, p.d_cnt, (const int*)nullptr);
  CS_LAUNCH_CHECK();
  return CS_OK;
}

// Tiling order of the maps plans[idx[0..m)] (m <= ORDER_MAX_MAPS, all `ordered`) on stream s, which is behind every stream
// that built them: keys of the maps the level kernel did not build, ONE radix sort, one finish launch (row lists, group
// masks, pair counts to the host slots).
static int order_many(const MapPlan* plans, const int* idx, int m, uint32_t* key, int32_t* row, uint32_t* key_sorted,
                      int32_t* row_sorted, int64_t first, int64_t total, hipStream_t s) {
  if (m == 0) return CS_OK;
  OrderTab tab;
  tab.n = m;
  uint64_t kblk = 0, tblk = 0;
  bool any_keys = false;
  for (int j = 0; j < m; ++j) {
    const MapPlan& p = plans[idx[j]];
    cs_kernelmap* km = p.km;
    OrderMap& o = tab.m[j];
    const int64_t n = km->n_out;
    o.n_out = n;
    o.base = p.base - first;
    o.n_groups = ceil_div(ceil_div(n, 32), 8) * 8;
    o.nbr = km->d_nbr;
    km->d_rowlist = (int32_t*)pool_alloc(n * sizeof(int32_t));
    km->d_gmask = (uint32_t*)pool_alloc((size_t)o.n_groups * sizeof(uint32_t));
    CS_REQUIRE(km->d_rowlist && km->d_gmask, CS_ERR_HIP, "cs_kernelmap_build_many: tiling order allocation failed");
    o.rowlist = km->d_rowlist;
    o.gmask = km->d_gmask;
    o.d_cnt = p.d_cnt;
    o.host_cnt = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&o.host_cnt), km->h_cnt, 0) != hipSuccess) o.host_cnt = nullptr;
    CS_REQUIRE(o.host_cnt, CS_ERR_HIP, "cs_kernelmap_build_many: no device view of the pair-count slot");
    o.tag = (uint32_t)(idx[j] % ORDER_MAX_MAPS) << 27;
    o.need_keys = p.fused ? 0 : 1;
    any_keys = any_keys || !p.fused;
    o.kblk0 = (uint32_t)kblk;
    o.tblk0 = (uint32_t)tblk;
    kblk += (uint64_t)ceil_div(n, 256);
    tblk += (uint64_t)std::max<int64_t>(ceil_div(n, 256), 1);
  }
  CS_REQUIRE(total < (1LL << 31) && kblk < (1ULL << 31), CS_ERR_UNSUPPORTED, "cs_kernelmap_build_many: batch too large");
  int map_bits = 0;
  while ((1 << map_bits) < ORDER_MAX_MAPS) ++map_bits;
  size_t tmp_bytes = 0;
  CS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key + first, key_sorted + first, row + first,
                                                  row_sorted + first, (int)total, 0, 27 + map_bits, s));
  PoolBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
  CS_REQUIRE(tmp.p, CS_ERR_HIP, "cs_kernelmap_build_many: row order scratch allocation failed");
  if (any_keys) {
    hipLaunchKernelGGL(k_row_keys_all, dim3((unsigned)kblk), dim3(256), 0, s, tab, key + first, row + first);
    CS_LAUNCH_CHECK();
  }
  CS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key + first, key_sorted + first, row + first,
                                                  row_sorted + first, (int)total, 0, 27 + map_bits, s));
  hipLaunchKernelGGL(k_order_finish, dim3((unsigned)tblk), dim3(256), 0, s, tab, key_sorted + first, row_sorted + first);
  CS_LAUNCH_CHECK();
  for (int j = 0; j < m; ++j) CS_HIP_CHECK(hipEventRecord(plans[idx[j]].km->cnt_ready, s));
  return CS_OK;
}

extern "C" {

// The kernel maps of a batch.  Maps are grouped by the coordinate level they PROBE (their `in` map): one k_level_maps launch
// per level serves every map of the group from one LDS table per sample, and the levels run on up to four streams (the
// caller's and the thread's side streams: every side stream starts behind the caller's stream and the caller's stream
// continues behind all of them; scratch freed meanwhile is handed back to the pool only after that join).  Then ONE radix
// sort gives every map its tiling order.  CS_KMAP_STREAMS=1: everything on the caller's stream; CS_KMAP_GLOBAL=1: the
// global-table kernel for everything.
int cs_kernelmap_build_many(int n, const cs_coordmap* const* in, const cs_coordmap* const* out, const int* kernel_size,
                            const int* transposed, void* stream, cs_kernelmap** km_out) {
  CS_REQUIRE(n >= 0 && (n == 0 || (in && out && kernel_size && transposed && km_out)), CS_ERR_INVALID,
             "cs_kernelmap_build_many: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  for (int i = 0; i < n; ++i) km_out[i] = nullptr;
  if (n == 0) return CS_OK;
  std::vector<MapPlan> plans(n);
  for (int i = 0; i < n; ++i) {
    const int rc = map_geometry(in[i], out[i], kernel_size[i], transposed[i], &plans[i].step, &plans[i].sign);
    if (rc != CS_OK) return rc;
    plans[i].in = in[i];
    plans[i].out = out[i];
  }
  const bool global_only = getenv("CS_KMAP_GLOBAL") != nullptr;
  // per-sample segments of the coordinate maps (lazily, cached on the map): on the caller's stream before the streams fork
  if (!global_only) {
    std::vector<cs_coordmap*> need;
    for (int i = 0; i < n; ++i)
      if (kernel_size[i] == 3) {
        need.push_back(const_cast<cs_coordmap*>(in[i]));
        need.push_back(const_cast<cs_coordmap*>(out[i]));
      }
    const int rc_seg = ensure_segments_many(need.data(), (int)need.size(), s);
    if (rc_seg != CS_OK) return rc_seg;
  }
  static const int n_streams = [] {
    const char* e = getenv("CS_KMAP_STREAMS");
    int v = e ? atoi(e) : 4;   // 2 / 3 / 4 / 5 streams: stress 6016 / 6201 / 6235 / 5608 clouds/s (the runtime maps streams to 4 queues)
    return v < 1 ? 1 : (v > 5 ? 5 : v);
  }();
  hipStream_t st[5] = {s, nullptr, nullptr, nullptr, nullptr};
  for (int k = 1; k < n_streams; ++k) st[k] = side_stream(k - 1);
  int ns = 1;
  while (ns < n_streams && st[ns]) ++ns;

  int rc = CS_OK;
  hipError_t je = hipSuccess;
  auto fail = [&](int code) {
    for (int i = 0; i < n; ++i) {
      if (plans[i].km) cs_kernelmap_free(plans[i].km);
      km_out[i] = nullptr;
    }
    pool_defer_end();
    return code;
  };
  pool_defer_begin();
  // ---- map objects, counters, the batch's sort arrays ----
  PoolBuf<unsigned long long> cnts(n);
  if (!cnts.p) {
    set_error("cs_kernelmap_build_many: allocation failed");
    return fail(CS_ERR_HIP);
  }
  int64_t total_rows = 0;
  for (int i = 0; i < n; ++i) {
    MapPlan& p = plans[i];
    cs_kernelmap* km = new cs_kernelmap();
    p.km = km;
    km->n_out = out[i]->n;
    km->n_in = in[i]->n;
    km->kvol = kernel_size[i] == 3 ? 27 : 1;
    km->transposed = transposed[i];
    const int64_t total = km->n_out * km->kvol;
    km->d_nbr = (int32_t*)pool_alloc((total ? total : 1) * sizeof(int32_t));
    p.d_cnt = cnts.p + i;
    if (!km->d_nbr) {
      set_error("cs_kernelmap_build_many: allocation failed");
      return fail(CS_ERR_HIP);
    }
    if (total == 0) {
      km->num_pairs = 0;
      continue;
    }
    km->h_cnt = count_slot_acquire(&km->cnt_slot);
    if (!km->h_cnt || hipEventCreateWithFlags(&km->cnt_ready, hipEventDisableTiming) != hipSuccess) {
      set_error("cs_kernelmap_build_many: pair-count slot allocation failed");
      return fail(CS_ERR_HIP);
    }
    cs_coordmap* in_m = const_cast<cs_coordmap*>(in[i]);
    cs_coordmap* out_m = const_cast<cs_coordmap*>(out[i]);
    p.ordered = km->kvol == 27;
    p.fused = !global_only && km->kvol == 27 && in_m->seg_state == 1 && out_m->seg_state == 1 &&
              in_m->n_batch == out_m->n_batch;
    if (p.ordered) {
      p.base = total_rows;
      total_rows += km->n_out;
    }
  }
  if (total_rows >= (1LL << 31)) {
    set_error("cs_kernelmap_build_many: batch too large");
    return fail(CS_ERR_UNSUPPORTED);
  }
  PoolBuf<uint32_t> key(total_rows), key_sorted(total_rows);
  PoolBuf<int32_t> row(total_rows), row_sorted(total_rows);
  if (!key.p || !key_sorted.p || !row.p || !row_sorted.p) {
    set_error("cs_kernelmap_build_many: row order scratch allocation failed");
    return fail(CS_ERR_HIP);
  }
  if (hipMemsetAsync(cnts.p, 0, sizeof(unsigned long long) * n, s) != hipSuccess) {
    set_error("cs_kernelmap_build_many: memset failed");
    return fail(CS_ERR_HIP);
  }
  // ---- fork ----
  struct Ev {
    hipEvent_t e = nullptr;
    ~Ev() { if (e) (void)hipEventDestroy(e); }
  } fork, join[5];
  if (ns > 1) {
    hipError_t e = hipEventCreateWithFlags(&fork.e, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(fork.e, s);
    for (int k = 1; k < ns && e == hipSuccess; ++k) {
      e = hipEventCreateWithFlags(&join[k].e, hipEventDisableTiming);
      if (e == hipSuccess) e = hipStreamWaitEvent(st[k], fork.e, 0);
    }
    if (e != hipSuccess) {
      set_error("cs_kernelmap_build_many: %s", hipGetErrorString(e));
      return fail(CS_ERR_HIP);
    }
  }
  // ---- builders: one launch per probed level (fused maps), one per remaining map ----
  {
    ProfScope prof("kmap", s);
    int lane = 0;
    std::vector<char> done(n, 0);
    for (int i = 0; i < n && rc == CS_OK; ++i) {
      if (done[i] || !plans[i].fused) continue;
      int jobs[LEVEL_MAX_JOBS], nj = 0;
      for (int j = i; j < n && nj < LEVEL_MAX_JOBS; ++j)
        if (!done[j] && plans[j].fused && in[j] == in[i] && (j / ORDER_MAX_MAPS) == (i / ORDER_MAX_MAPS)) {
          jobs[nj++] = j;
          done[j] = 1;
        }
      rc = launch_level(in[i], plans.data(), jobs, nj, key.p, row.p, st[lane++ % ns]);
    }
    for (int i = 0; i < n && rc == CS_OK; ++i) {
      if (done[i]) continue;
      rc = launch_single(plans[i], st[lane++ % ns]);
    }
    // join (also on the error path: the side streams may hold work that reads scratch of this call)
    for (int k = 1; k < ns; ++k) {
      hipError_t e1 = hipEventRecord(join[k].e, st[k]);
      if (e1 == hipSuccess) e1 = hipStreamWaitEvent(s, join[k].e, 0);
      if (e1 != hipSuccess) {
        (void)hipStreamSynchronize(st[k]);
        je = e1;
      }
    }
  }
  // ---- tiling order of all ordered maps (ONE sort per ORDER_MAX_MAPS maps) on the caller's stream, behind the join ----
  if (rc == CS_OK && je == hipSuccess) {
    ProfScope prof("kmap", s);
    for (int c0 = 0; c0 < n && rc == CS_OK; c0 += ORDER_MAX_MAPS) {
      int idx[ORDER_MAX_MAPS], m = 0;
      int64_t first = -1, rows = 0;
      for (int i = c0; i < std::min(n, c0 + ORDER_MAX_MAPS); ++i)
        if (plans[i].ordered) {
          if (first < 0) first = plans[i].base;
          rows += plans[i].km->n_out;
          idx[m++] = i;
        }
      if (m) rc = order_many(plans.data(), idx, m, key.p, row.p, key_sorted.p, row_sorted.p, first, rows, s);
    }
    // maps without a tiling order (1x1 "maps"): the count goes to the host slot by a copy
    for (int i = 0; i < n && rc == CS_OK; ++i) {
      cs_kernelmap* km = plans[i].km;
      if (plans[i].ordered || km->num_pairs == 0) continue;
      hipError_t e = hipMemcpyAsync(km->h_cnt, plans[i].d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipEventRecord(km->cnt_ready, s);
      if (e != hipSuccess) {
        set_error("cs_kernelmap_build_many: %s", hipGetErrorString(e));
        rc = CS_ERR_HIP;
      }
    }
  }
  if (rc != CS_OK || je != hipSuccess) {
    if (rc == CS_OK) {
      set_error("cs_kernelmap_build_many: joining the streams failed: %s", hipGetErrorString(je));
      rc = CS_ERR_HIP;
    }
    return fail(rc);
  }
  for (int i = 0; i < n; ++i) km_out[i] = plans[i].km;
  pool_defer_end();
  return CS_OK;
}

int cs_kernelmap_build(const cs_coordmap* in, const cs_coordmap* out, int kernel_size, int transposed, void* stream,
                       cs_kernelmap** km_out) {
  CS_REQUIRE(km_out, CS_ERR_INVALID, "cs_kernelmap_build: NULL argument");
  *km_out = nullptr;
  const cs_coordmap* ins[1] = {in};
  const cs_coordmap* outs[1] = {out};
  return cs_kernelmap_build_many(1, ins, outs, &kernel_size, &transposed, stream, km_out);
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
