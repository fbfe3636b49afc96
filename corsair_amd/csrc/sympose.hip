// Correspondence assembly of sym_pose (utils/symmetry.py:145-179, 262-358) on the device: the stable
// partition of a query cloud's voxels by part label (split_corr concatenates the parts in order, rows in
// original order inside a part), the test "did a part run out of neighbours" per part configuration, and
// the gather of the (query xyz, CAD xyz) correspondence lists the RANSAC call takes.  Rounds 1-2 did these
// with ~25 torch ops per batch (a stable sort = ~75 rocprim launches, repeat_interleave / cumsum / index
// chains); here each is one launch.
#include "common.h"

namespace cs {

// One workgroup per cloud: rows [off[c], off[c+1]) in order, stable by label (labels outside 0..7 count as
// 0..7 clamped, as the host code of rounds 1-2 did).  order[off[c] + i] = global row of the i-th row of the
// partitioned cloud.
__global__ __launch_bounds__(256) void k_partition_by_label(const int32_t* __restrict__ label,
                                                            const int64_t* __restrict__ off,
                                                            int64_t* __restrict__ order) {
  __shared__ int hist[8];
  __shared__ int base[8];
  __shared__ int wcnt[4][8];
  const int c = blockIdx.x;
  const int64_t r0 = off[c], r1 = off[c + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 8) hist[tid] = 0;
  __syncthreads();
  for (int64_t r = r0 + tid; r < r1; r += 256) {
    int l = label[r];
    l = l < 0 ? 0 : (l > 7 ? 7 : l);
    atomicAdd(&hist[l], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int l = 0; l < 8; ++l) {
      base[l] = run;
      run += hist[l];
    }
  }
  __syncthreads();
  for (int64_t rb = r0; rb < r1; rb += 256) {
    const int64_t r = rb + tid;
    int l = -1;
    if (r < r1) {
      l = label[r];
      l = l < 0 ? 0 : (l > 7 ? 7 : l);
    }
    int my_rank = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned long long m = __ballot(l == q);
      if (l == q) my_rank = __popcll(m & ((1ULL << lane) - 1ULL));
      if (lane == 0) wcnt[wave][q] = __popcll(m);
    }
    __syncthreads();
    if (l >= 0) {
      int pos = base[l] + my_rank;
      for (int w = 0; w < wave; ++w) pos += wcnt[w][l];
      order[r0 + pos] = r;
    }
    __syncthreads();
    if (tid < 8) base[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
    __syncthreads();
  }
}

// desc[j] = {q_first, n_first, t_first, len, out_first}: configuration j takes query rows
// rows[q_first .. q_first + len) (rows == nullptr: the rows themselves), their neighbour lists
// nn[n_first + i][0 .. k) (row index local to the CAD cloud that starts at t_first) and writes its
// len * k correspondences from out_first * k on, query point repeated k times, neighbours in list order
// (find_kcorr: inds0 = repeat(arange(N0), k), nn_inds.flatten()).
struct CorrDesc {
  int64_t q_first, n_first, t_first, len, out_first;
};
__global__ __launch_bounds__(256) void k_corr_assemble(const float* __restrict__ xyz0, const float* __restrict__ xyz1,
                                                       const int64_t* __restrict__ rows, const int32_t* __restrict__ nn,
                                                       int k, const CorrDesc* __restrict__ desc,
                                                       float* __restrict__ src, float* __restrict__ tgt) {
  const CorrDesc d = desc[blockIdx.y];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < d.len * k; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / k;
    const int nb = (int)(e - i * k);
    const int64_t q = rows ? rows[d.q_first + i] : d.q_first + i;
    // a missing neighbour (-1: fewer than k targets) never becomes an address: the callers reject such
    // configurations (registration.py raises for the vanilla list, cs_cfg_bad drops part configurations); should
    // one get here anyway it reads the cloud's first row instead of the bytes in front of the buffer
    const int32_t nbv = nn[(d.n_first + i) * k + nb];
    const int64_t t = d.t_first + (nbv < 0 ? 0 : nbv);
    const int64_t o = (d.out_first * k + e) * 3;
    src[o] = xyz0[q * 3];
    src[o + 1] = xyz0[q * 3 + 1];
    src[o + 2] = xyz0[q * 3 + 2];
    tgt[o] = xyz1[t * 3];
    tgt[o + 1] = xyz1[t * 3 + 1];
    tgt[o + 2] = xyz1[t * 3 + 2];
  }
}

// bad[j] = 1 when any neighbour entry of rows [first[j], first[j+1]) is negative (a CAD part with fewer than
// k voxels: the reference cannot build that configuration)
__global__ __launch_bounds__(256) void k_cfg_bad(const int32_t* __restrict__ nn, int k, const int64_t* __restrict__ first,
                                                 int32_t* __restrict__ bad) {
  const int j = blockIdx.y;
  const int64_t a = first[j] * k, b = first[j + 1] * k;
  int any = 0;
  for (int64_t e = a + (int64_t)blockIdx.x * 256 + threadIdx.x; e < b; e += (int64_t)gridDim.x * 256) any |= nn[e] < 0;
  if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(&bad[j], 1);
}
__global__ void k_zero_i32(int32_t* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

}  // namespace cs

using namespace cs;

extern "C" {

int cs_partition_by_label(const int32_t* d_label, const int64_t* d_off, int n_cloud, int64_t* d_order, void* stream) {
  CS_REQUIRE(n_cloud >= 0 && (n_cloud == 0 || (d_label && d_off && d_order)), CS_ERR_INVALID,
             "cs_partition_by_label: bad argument");
  if (n_cloud == 0) return CS_OK;
  hipLaunchKernelGGL(k_partition_by_label, dim3((unsigned)n_cloud), dim3(256), 0, (hipStream_t)stream, d_label, d_off,
                     d_order);
  CS_LAUNCH_CHECK();
  return CS_OK;
}

int cs_corr_assemble(const float* d_xyz0, const float* d_xyz1, const int64_t* d_rows, const int32_t* d_nn, int k,
                     const int64_t* d_desc, int n_cfg, int64_t max_len, float* d_src, float* d_tgt, void* stream) {
  CS_REQUIRE(k >= 1 && n_cfg >= 0 && max_len >= 0, CS_ERR_INVALID, "cs_corr_assemble: bad argument");
  if (n_cfg == 0 || max_len == 0) return CS_OK;
  CS_REQUIRE(d_xyz0 && d_xyz1 && d_nn && d_desc && d_src && d_tgt, CS_ERR_INVALID, "cs_corr_assemble: NULL tensor");
  unsigned gx = (unsigned)ceil_div(max_len * k, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_corr_assemble, dim3(gx, (unsigned)n_cfg), dim3(256), 0, (hipStream_t)stream, d_xyz0, d_xyz1, d_rows,
                     d_nn, k, reinterpret_cast<const CorrDesc*>(d_desc), d_src, d_tgt);
  CS_LAUNCH_CHECK();
  return CS_OK;
}

int cs_cfg_bad(const int32_t* d_nn, int k, const int64_t* d_first, int n_cfg, int32_t* d_bad, void* stream) {
  CS_REQUIRE(k >= 1 && n_cfg >= 0, CS_ERR_INVALID, "cs_cfg_bad: bad argument");
  if (n_cfg == 0) return CS_OK;
  CS_REQUIRE(d_nn && d_first && d_bad, CS_ERR_INVALID, "cs_cfg_bad: NULL tensor");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_zero_i32, dim3((unsigned)ceil_div(n_cfg, 256)), dim3(256), 0, s, d_bad, n_cfg);
  hipLaunchKernelGGL(k_cfg_bad, dim3(16, (unsigned)n_cfg), dim3(256), 0, s, d_nn, k, d_first, d_bad);
  CS_LAUNCH_CHECK();
  return CS_OK;
}

}  // extern "C"
