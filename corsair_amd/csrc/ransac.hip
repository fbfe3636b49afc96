// Batched correspondence RANSAC on gfx950.
//
// Replaces registration_based_on_corr -> Open3D registration_ransac_based_on_correspondence
// (utils/eval_pose.py:82-100 of the reference; ransac_n = 10, 100 000 iterations, confidence 0.999).
// Semantics: Open3D's loop as executed by ONE thread (iteration order = index order), with the
// global Mersenne twister replaced by a counter-based generator so that iteration i of every
// problem is reproducible anywhere.  Iterations are processed in growing chunks; within a chunk
//   k_ransac_hyp   one lane per hypothesis: sample ransac_n pairs, closed-form rigid fit
//                  (Horn quaternion, 4x4 Jacobi eigen-solver, f64), emit R|t as f32
//   k_ransac_eval  one lane per hypothesis, correspondences streamed through the scalar cache
//                  (wave-uniform s_load), 9 fma + 3 sub + 3 fma + compare per pair in f32;
//                  inlier count (int) and fixed-point squared error (u64) are exact integers, so
//                  any split of the correspondence range across waves gives identical sums
//   k_ransac_scan  per problem, replays the chunk in iteration order: best-so-far update and
//                  the early-exit bound est_k, exactly as the sequential loop would
// The correspondence set of a problem (M x 32 B ~ 0.7 MB at eval size) is L2 resident, so the
// evaluation is VALU-bound: ~20 f32 VALU ops per (hypothesis, pair).
#include <math.h>

#include <vector>

#include "common.h"

namespace cs {

struct RansacProb {
  int64_t off;
  int32_t m;
  int32_t est_k;
  int32_t best_cnt;
  int32_t best_itr;
  unsigned long long best_err;
  int32_t done;
  int32_t iters;
  float best_T[12];
};

__host__ __device__ static inline uint64_t rng_u64(uint64_t seed, uint64_t itr, uint64_t j) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ULL * (itr * 64ULL + j + 1ULL);
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  x = x ^ (x >> 31);
  return x;
}
__host__ __device__ static inline uint32_t rng_index(uint64_t seed, uint64_t itr, uint64_t j,
                                                     uint32_t m) {
  return (uint32_t)(((rng_u64(seed, itr, j) >> 32) * (uint64_t)m) >> 32);
}

__global__ void k_ransac_pack(const float* __restrict__ src, const float* __restrict__ tgt,
                              int64_t n, float4* __restrict__ pack) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  pack[2 * i + 0] = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], 0.f);
  pack[2 * i + 1] = make_float4(tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2], 0.f);
}

// Cyclic Jacobi on a symmetric 4x4 (fixed 8 sweeps), eigenvectors in v (columns).
__device__ __forceinline__ void jacobi4(double a[4][4], double v[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 8; ++sweep) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        const double apq = a[p][q];
        if (apq != 0.0) {
          const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0);
          const double s = t * c;
          a[p][p] = a[p][p] - t * apq;
          a[q][q] = a[q][q] + t * apq;
          a[p][q] = 0.0;
          a[q][p] = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (r != p && r != q) {
              const double arp = a[r][p], arq = a[r][q];
              const double nrp = c * arp - s * arq;
              const double nrq = s * arp + c * arq;
              a[r][p] = nrp;
              a[p][r] = nrp;
              a[r][q] = nrq;
              a[q][r] = nrq;
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double vrp = v[r][p], vrq = v[r][q];
            v[r][p] = c * vrp - s * vrq;
            v[r][q] = s * vrp + c * vrq;
          }
        }
      }
    }
  }
}

// hyp layout: [prob][12][bmax] (structure of arrays so the evaluating lanes read coalesced)
__global__ __launch_bounds__(256) void k_ransac_hyp(const RansacProb* __restrict__ probs,
                                                    const float4* __restrict__ pack, int it0,
                                                    int bcount, int bmax, int ransac_n,
                                                    uint64_t seed, float* __restrict__ hyp) {
  const int p = blockIdx.y;
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= bcount) return;
  const RansacProb pr = probs[p];
  const int itr = it0 + h;
  if (pr.done || itr >= pr.est_k) return;
  const uint32_t m = (uint32_t)pr.m;
  // centroids
  double cs_[3] = {0, 0, 0}, ct_[3] = {0, 0, 0};
  for (int j = 0; j < ransac_n; ++j) {
    const int64_t i = pr.off + rng_index(seed, (uint64_t)itr, (uint64_t)j, m);
    const float4 s = pack[2 * i], t = pack[2 * i + 1];
    cs_[0] += (double)s.x;
    cs_[1] += (double)s.y;
    cs_[2] += (double)s.z;
    ct_[0] += (double)t.x;
    ct_[1] += (double)t.y;
    ct_[2] += (double)t.z;
  }
  const double inv_n = (double)ransac_n;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    cs_[a] = cs_[a] / inv_n;
    ct_[a] = ct_[a] / inv_n;
  }
  // cross-covariance S[a][b] = sum (s_a - cs_a)(t_b - ct_b)
  double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int j = 0; j < ransac_n; ++j) {
    const int64_t i = pr.off + rng_index(seed, (uint64_t)itr, (uint64_t)j, m);
    const float4 s = pack[2 * i], t = pack[2 * i + 1];
    const double ds[3] = {(double)s.x - cs_[0], (double)s.y - cs_[1], (double)s.z - cs_[2]};
    const double dt[3] = {(double)t.x - ct_[0], (double)t.y - ct_[1], (double)t.z - ct_[2]};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) S[a][b] = fma(ds[a], dt[b], S[a][b]);
  }
  double N[4][4], V[4][4];
  N[0][0] = S[0][0] + S[1][1] + S[2][2];
  N[0][1] = S[1][2] - S[2][1];
  N[0][2] = S[2][0] - S[0][2];
  N[0][3] = S[0][1] - S[1][0];
  N[1][1] = S[0][0] - S[1][1] - S[2][2];
  N[1][2] = S[0][1] + S[1][0];
  N[1][3] = S[2][0] + S[0][2];
  N[2][2] = -S[0][0] + S[1][1] - S[2][2];
  N[2][3] = S[1][2] + S[2][1];
  N[3][3] = -S[0][0] - S[1][1] + S[2][2];
  N[1][0] = N[0][1];
  N[2][0] = N[0][2];
  N[3][0] = N[0][3];
  N[2][1] = N[1][2];
  N[3][1] = N[1][3];
  N[3][2] = N[2][3];
  jacobi4(N, V);
  // eigenvector of the largest eigenvalue (ties -> lowest index), selected without dynamic indexing
  double best = N[0][0];
  double qw = V[0][0], qx = V[1][0], qy = V[2][0], qz = V[3][0];
#pragma unroll
  for (int c = 1; c < 4; ++c) {
    if (N[c][c] > best) {
      best = N[c][c];
      qw = V[0][c];
      qx = V[1][c];
      qy = V[2][c];
      qz = V[3][c];
    }
  }
  const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  qw = qw / qn;
  qx = qx / qn;
  qy = qy / qn;
  qz = qz / qn;
  double R[3][3];
  R[0][0] = 1.0 - 2.0 * (qy * qy + qz * qz);
  R[0][1] = 2.0 * (qx * qy - qw * qz);
  R[0][2] = 2.0 * (qx * qz + qw * qy);
  R[1][0] = 2.0 * (qx * qy + qw * qz);
  R[1][1] = 1.0 - 2.0 * (qx * qx + qz * qz);
  R[1][2] = 2.0 * (qy * qz - qw * qx);
  R[2][0] = 2.0 * (qx * qz - qw * qy);
  R[2][1] = 2.0 * (qy * qz + qw * qx);
  R[2][2] = 1.0 - 2.0 * (qx * qx + qy * qy);
  float* o = hyp + ((int64_t)p * 12) * bmax + h;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double t = ct_[a] - (R[a][0] * cs_[0] + R[a][1] * cs_[1] + R[a][2] * cs_[2]);
    o[(int64_t)(4 * a + 0) * bmax] = (float)R[a][0];
    o[(int64_t)(4 * a + 1) * bmax] = (float)R[a][1];
    o[(int64_t)(4 * a + 2) * bmax] = (float)R[a][2];
    o[(int64_t)(4 * a + 3) * bmax] = (float)t;
  }
}

// grid: x = hypothesis tile (256) * splits, y = problem
__global__ __launch_bounds__(256) void k_ransac_eval(const RansacProb* __restrict__ probs,
                                                     const float4* __restrict__ pack,
                                                     const float* __restrict__ hyp, int it0,
                                                     int bcount, int bmax, int splits, float thr2,
                                                     float scale, int32_t* __restrict__ res_cnt,
                                                     unsigned long long* __restrict__ res_err) {
  const int p = blockIdx.y;
  const int tile = blockIdx.x / splits;
  const int split = blockIdx.x - tile * splits;
  const int h = tile * 256 + threadIdx.x;
  const RansacProb pr = probs[p];
  if (pr.done) return;
  if (it0 + tile * 256 >= pr.est_k) return;  // whole tile beyond the bound
  const bool valid = h < bcount && it0 + h < pr.est_k;
  const int hh = valid ? h : 0;
  const float* hp = hyp + ((int64_t)p * 12) * bmax + hh;
  const float r00 = hp[0 * (int64_t)bmax], r01 = hp[1 * (int64_t)bmax],
              r02 = hp[2 * (int64_t)bmax], tx = hp[3 * (int64_t)bmax];
  const float r10 = hp[4 * (int64_t)bmax], r11 = hp[5 * (int64_t)bmax],
              r12 = hp[6 * (int64_t)bmax], ty = hp[7 * (int64_t)bmax];
  const float r20 = hp[8 * (int64_t)bmax], r21 = hp[9 * (int64_t)bmax],
              r22 = hp[10 * (int64_t)bmax], tz = hp[11 * (int64_t)bmax];
  const int per = (pr.m + splits - 1) / splits;
  const int beg = split * per;
  const int end = min(pr.m, beg + per);
  const float4* __restrict__ pk = pack + 2 * pr.off;
  int cnt = 0;
  unsigned long long err = 0;
#pragma unroll 4
  for (int i = beg; i < end; ++i) {
    const float4 s = pk[2 * i];
    const float4 q = pk[2 * i + 1];
    const float px = __fmaf_rn(r00, s.x, __fmaf_rn(r01, s.y, __fmaf_rn(r02, s.z, tx)));
    const float py = __fmaf_rn(r10, s.x, __fmaf_rn(r11, s.y, __fmaf_rn(r12, s.z, ty)));
    const float pz = __fmaf_rn(r20, s.x, __fmaf_rn(r21, s.y, __fmaf_rn(r22, s.z, tz)));
    const float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
    const float d2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
    const bool in = d2 < thr2;
    cnt += in ? 1 : 0;
    err += in ? (unsigned long long)(uint32_t)(d2 * scale) : 0ULL;
  }
  if (valid) {
    if (splits == 1) {
      res_cnt[(int64_t)p * bmax + h] = cnt;
      res_err[(int64_t)p * bmax + h] = err;
    } else {
      atomicAdd(&res_cnt[(int64_t)p * bmax + h], cnt);
      atomicAdd(&res_err[(int64_t)p * bmax + h], err);
    }
  }
}

__global__ void k_ransac_scan(RansacProb* probs, int n_prob, const float* __restrict__ hyp,
                              const int32_t* __restrict__ res_cnt,
                              const unsigned long long* __restrict__ res_err, int it0, int bcount,
                              int bmax, int ransac_n, int max_iter, double log_1mc,
                              int* n_active) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_prob) return;
  RansacProb pr = probs[p];
  if (pr.done) return;
  int h = 0;
  for (; h < bcount; ++h) {
    const int itr = it0 + h;
    if (itr >= pr.est_k) break;
    const int cnt = res_cnt[(int64_t)p * bmax + h];
    const unsigned long long err = res_err[(int64_t)p * bmax + h];
    if (cnt > pr.best_cnt || (cnt == pr.best_cnt && cnt > 0 && err < pr.best_err)) {
      pr.best_cnt = cnt;
      pr.best_err = err;
      pr.best_itr = itr;
      for (int c = 0; c < 12; ++c) pr.best_T[c] = hyp[((int64_t)p * 12 + c) * bmax + h];
      const double ratio = fmin(1.0, (double)cnt / (double)pr.m);
      double pw = 1.0;
      for (int j = 0; j < ransac_n; ++j) pw = pw * ratio;
      const double den = log(1.0 - pw);
      if (den < 0.0) {  // den == 0: (inliers/M)^n below 2^-53, no finite bound
        const double est = log_1mc / den;
        if (est < (double)pr.est_k) pr.est_k = (int)ceil(est);
      }
    }
  }
  const int consumed = it0 + h;
  if (consumed >= pr.est_k || consumed >= max_iter) {
    pr.done = 1;
    pr.iters = consumed < max_iter ? consumed : max_iter;
  } else {
    atomicAdd(n_active, 1);
  }
  probs[p] = pr;
}

__global__ void k_ransac_finish(const RansacProb* __restrict__ probs, int n_prob, double scale,
                                float* __restrict__ T, int32_t* __restrict__ inliers,
                                double* __restrict__ rmse, int32_t* __restrict__ iters) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_prob) return;
  const RansacProb pr = probs[p];
  float* o = T + (int64_t)p * 16;
  if (pr.best_cnt > 0) {
    for (int c = 0; c < 12; ++c) o[c] = pr.best_T[c];
  } else {
    for (int c = 0; c < 12; ++c) o[c] = (c % 5 == 0) ? 1.f : 0.f;
  }
  o[12] = 0.f;
  o[13] = 0.f;
  o[14] = 0.f;
  o[15] = 1.f;
  if (inliers) inliers[p] = pr.best_cnt;
  if (rmse)
    rmse[p] = pr.best_cnt > 0 ? sqrt(((double)pr.best_err / scale) / (double)pr.best_cnt) : 0.0;
  if (iters) iters[p] = pr.iters;
}

}  // namespace cs

using namespace cs;

extern "C" {

int cs_ransac_batch(const float* d_src, const float* d_tgt, const int64_t* h_off, int n_prob,
                    float max_corr, int ransac_n, int max_iter, double confidence, uint64_t seed,
                    float* d_T, int32_t* d_inliers, double* d_rmse, int32_t* d_iters,
                    void* stream) {
  CS_REQUIRE(d_src && d_tgt && h_off && d_T, CS_ERR_INVALID, "cs_ransac_batch: NULL argument");
  CS_REQUIRE(ransac_n >= 3 && ransac_n <= 64, CS_ERR_INVALID,
             "cs_ransac_batch: ransac_n %d not in [3, 64]", ransac_n);
  CS_REQUIRE(max_corr > 0.f && max_iter >= 1, CS_ERR_INVALID,
             "cs_ransac_batch: need max_corr > 0 and max_iter >= 1");
  CS_REQUIRE(confidence > 0.0 && confidence <= 1.0, CS_ERR_INVALID,
             "cs_ransac_batch: confidence must be in (0, 1]");
  if (n_prob <= 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = h_off[n_prob] - h_off[0];
  CS_REQUIRE(h_off[0] == 0 && total >= 0, CS_ERR_INVALID, "cs_ransac_batch: bad offsets");

  std::vector<RansacProb> hp(n_prob);
  int m_max = 0;
  for (int p = 0; p < n_prob; ++p) {
    int64_t m = h_off[p + 1] - h_off[p];
    CS_REQUIRE(m >= 0 && m < (1LL << 31), CS_ERR_INVALID, "cs_ransac_batch: bad segment %d", p);
    RansacProb& pr = hp[p];
    memset(&pr, 0, sizeof(pr));
    pr.off = h_off[p];
    pr.m = (int32_t)m;
    pr.est_k = max_iter;
    pr.best_cnt = 0;
    pr.best_itr = -1;
    pr.best_err = 0;
    // Open3D returns the default (identity) result when there are fewer pairs than ransac_n
    pr.done = m < ransac_n ? 1 : 0;
    pr.iters = 0;
    if (m > m_max) m_max = (int)m;
  }
  const int bmax = 4096;
  PoolBuf<RansacProb> probs(n_prob);
  PoolBuf<float4> pack((size_t)(total ? total : 1) * 2);
  PoolBuf<float> hyp((size_t)n_prob * 12 * bmax);
  PoolBuf<int32_t> res_cnt((size_t)n_prob * bmax);
  PoolBuf<unsigned long long> res_err((size_t)n_prob * bmax);
  PoolBuf<int> n_active(1);
  CS_REQUIRE(probs.p && pack.p && hyp.p && res_cnt.p && res_err.p && n_active.p, CS_ERR_HIP,
             "cs_ransac_batch: scratch allocation failed");
  CS_HIP_CHECK(hipMemcpyAsync(probs.p, hp.data(), sizeof(RansacProb) * n_prob,
                              hipMemcpyHostToDevice, s));
  if (total > 0) {
    hipLaunchKernelGGL(k_ransac_pack, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s,
                       d_src, d_tgt, total, pack.p);
    CS_LAUNCH_CHECK();
  }
  // squared threshold and power-of-two fixed-point scale (thr2 * scale <= 2^31)
  const float thr2 = max_corr * max_corr;
  int ex = 0;
  (void)frexpf(thr2, &ex);
  const float scale = ldexpf(1.0f, 31 - ex);
  const double log_1mc = log(1.0 - confidence);  // -inf when confidence == 1: never exits early

  int it0 = 0;
  while (it0 < max_iter) {
    int b = it0 < 256 ? 256 : (it0 < bmax ? it0 : bmax);
    if (b > max_iter - it0) b = max_iter - it0;
    const int tiles = (b + 255) / 256;
    int splits = (int)(8192 / ((int64_t)n_prob * tiles * 4 > 0 ? (int64_t)n_prob * tiles * 4 : 1));
    if (splits < 1) splits = 1;
    if (splits > 16) splits = 16;
    if (m_max < 2048) splits = 1;
    {
      ProfScope prof("ransac_hyp", s);
      hipLaunchKernelGGL(k_ransac_hyp, dim3((unsigned)tiles, (unsigned)n_prob), dim3(256), 0, s,
                         probs.p, pack.p, it0, b, bmax, ransac_n, seed, hyp.p);
    }
    if (splits > 1) {  // partial sums of the splits are combined with integer atomics
      CS_HIP_CHECK(hipMemset2DAsync(res_cnt.p, sizeof(int32_t) * bmax, 0, sizeof(int32_t) * b,
                                    n_prob, s));
      CS_HIP_CHECK(hipMemset2DAsync(res_err.p, sizeof(unsigned long long) * bmax, 0,
                                    sizeof(unsigned long long) * b, n_prob, s));
    }
    CS_HIP_CHECK(hipMemsetAsync(n_active.p, 0, sizeof(int), s));
    // algorithmic work of this chunk: 30 FLOP per (evaluated hypothesis, correspondence)
    // (transform 18 + squared distance 8 + compare/accumulate)
    double eval_flop = 0.0;
    for (int p = 0; p < n_prob; ++p) {
      if (hp[p].done) continue;
      int nh = hp[p].est_k - it0;
      if (nh > b) nh = b;
      if (nh > 0) eval_flop += 30.0 * (double)nh * (double)hp[p].m;
    }
    {
      ProfScope prof("ransac_eval", s, eval_flop);
      hipLaunchKernelGGL(k_ransac_eval, dim3((unsigned)(tiles * splits), (unsigned)n_prob),
                         dim3(256), 0, s, probs.p, pack.p, hyp.p, it0, b, bmax, splits, thr2,
                         scale, res_cnt.p, res_err.p);
    }
    hipLaunchKernelGGL(k_ransac_scan, dim3((unsigned)ceil_div(n_prob, 64)), dim3(64), 0, s,
                       probs.p, n_prob, hyp.p, res_cnt.p, res_err.p, it0, b, bmax, ransac_n,
                       max_iter, log_1mc, n_active.p);
    CS_LAUNCH_CHECK();
    int h_active = 0;
    CS_HIP_CHECK(hipMemcpyAsync(&h_active, n_active.p, sizeof(int), hipMemcpyDeviceToHost, s));
    // the per-problem state (est_k, done) comes back with the activity counter: it sizes the next
    // chunk's work accounting and costs one small copy behind a synchronisation we need anyway
    CS_HIP_CHECK(hipMemcpyAsync(hp.data(), probs.p, sizeof(RansacProb) * n_prob,
                                hipMemcpyDeviceToHost, s));
    CS_HIP_CHECK(hipStreamSynchronize(s));
    it0 += b;
    if (h_active == 0) break;
  }
  hipLaunchKernelGGL(k_ransac_finish, dim3((unsigned)ceil_div(n_prob, 64)), dim3(64), 0, s,
                     probs.p, n_prob, (double)scale, d_T, d_inliers, d_rmse, d_iters);
  CS_LAUNCH_CHECK();
  CS_HIP_CHECK(hipStreamSynchronize(s));
  return CS_OK;
}

}  // extern "C"
