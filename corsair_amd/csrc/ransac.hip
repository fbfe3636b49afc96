// Batched correspondence RANSAC on gfx950.
//
// Replaces registration_based_on_corr -> Open3D registration_ransac_based_on_correspondence
// (utils/eval_pose.py:82-100 of the reference; ransac_n = 10, 100 000 iterations, confidence 0.999).
// Semantics: Open3D's loop as executed by ONE thread (iteration order = index order), with the
// global Mersenne twister replaced by a counter-based generator so that iteration i of every
// problem is reproducible anywhere.  Iterations are processed in growing chunks (256, 256, 512, ...,
// 16 384); within a chunk
//   k_ransac_hyp        one lane per hypothesis: sample ransac_n pairs (packed 32-B rows), closed-form
//                       rigid fit (Horn quaternion; largest eigenpair of the 4x4 matrix from its characteristic
//                       polynomial, horn_qcp, with the Jacobi eigen-solver as per-lane fallback; f64), emit R|t as f64
//                       (Open3D keeps the Matrix4d; the f32 cast happens at the very end, where the
//                       reference casts the result: utils/symmetry.py:274); from iteration 256 on it also emits the
//                       hypothesis' prefilter row (pf_emit_row: 16 f16 coefficients + c_h)
//   k_ransac_prefilter  (from iteration 256 on) an UPPER bound of every hypothesis' inlier count on the
//                       f16 matrix cores; hypotheses whose bound is below the carried best cannot
//                       matter and get count 0.  Round 4: <1, true> = one MFMA per tile (K = 16: a_hi . b_hi',
//                       the dropped term bounded per pair), signs counted by v_add_f32 under
//                       round-toward-minus-infinity; 0.2 % survive, and the survivors go through the K = 32
//                       bound (<2, false> in list mode) before the exact count.  See the block comments above
//                       k_ransac_pack16_b0 / the kernel and DESIGN.md ("RANSAC prefilter") for the bound.
//   k_ransac_count      the exact count in Open3D's arithmetic: the reference hands Open3D f64 points
//                       (utils/eval_pose.py:83-86) and Eigen transforms and compares in double, so the
//                       inlier test is evaluated in f64: a lane owns one hypothesis (R|t in 12 f64
//                       registers), the pairs of a 256-row stage are converted to f64 once and read
//                       from LDS as broadcasts; p = fma(r2,sz, fma(r1,sy, fma(r0,sx, t))), d = p - q,
//                       |d|^2 = fma(dz,dz, fma(dy,dy, dx dx)) < max_corr^2 (the canonical chain, the
//                       oracle's).  Used for all hypotheses of the first 256 iterations and (LIST) for
//                       long survivor lists; k_ransac_count_few handles the usual handful of survivors
//                       (same chain, count and fixed-point error in one pass).  These kernels see
//                       ~0.1 % of the (hypothesis, pair) work; the f16 prefilter carries the rest.
//   k_ransac_scan1      one wave per problem replays the chunk in iteration order (prefix max of the
//                       inlier counts -> early-exit bound est_k -> stop position) and lists the
//                       hypotheses that tie for the best count
//   k_ransac_err        fixed-point squared error (exact integer sums) of those few candidates (only
//                       when k_ransac_count_few has not produced it already)
//   k_ransac_scan2      best = max count, then min error, then first -- the final state of the
//                       sequential rule "better = more inliers, or equal inliers and smaller rmse"
//                       (inside k_ransac_scan1 whenever k_ransac_count_few has left the errors: the usual round is
//                       hyp -> prefilter -> survivors -> second-stage prefilter -> count_few -> scan1, six launches)
// The host loop synchronises once per chunk (one pinned copy of the per-problem state); the next
// chunk's hypotheses are already enqueued at that point.
// Inlier counts and fixed-point errors are integers, so any split of the correspondence range across
// workgroups gives identical sums.
#include <math.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "common.h"

namespace cs {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

struct RansacProb {
  int64_t off;
  int32_t m;
  int32_t est_k;
  int32_t best_cnt;
  int32_t best_itr;
  unsigned long long best_err;
  int32_t done;
  int32_t iters;
  // per-chunk scratch written by scan1, read by err / scan2
  int32_t n_cand;
  int32_t chunk_max;
  double best_T[12];
};

// The kernels of a round's front half (hypotheses, f16 rows, prefilter) may run while the previous
// round's scan kernels update est_k / done (cs_ransac_batch): they read the two fields with relaxed
// atomic loads and only use them to skip work -- either value is safe (est_k only shrinks, done only
// rises), the scan kernels decide with the final state.
__device__ __forceinline__ RansacProb prob_view(const RansacProb* probs, int p) {
  RansacProb v = {};
  v.off = probs[p].off;
  v.m = probs[p].m;
  v.est_k = __atomic_load_n(&probs[p].est_k, __ATOMIC_RELAXED);
  v.done = __atomic_load_n(&probs[p].done, __ATOMIC_RELAXED);
  return v;
}

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

// structure-of-arrays copy of the correspondences: pk[c * total + i], c = sx,sy,sz,qx,qy,qz
// + pair32[i] = (sx, sy, sz, qx | qy, qz, 0, 0): one aligned 32-B sector per pair for the random
// sampling of k_ransac_hyp (two 12-B rows of the caller's arrays would touch two to four lines)
__global__ void k_ransac_pack(const float* __restrict__ src, const float* __restrict__ tgt,
                              int64_t n, float* __restrict__ pk, float4* __restrict__ pair32) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  pair32[2 * i + 0] = make_float4(src[3 * i + 0], src[3 * i + 1], src[3 * i + 2], tgt[3 * i + 0]);
  pair32[2 * i + 1] = make_float4(tgt[3 * i + 1], tgt[3 * i + 2], 0.f, 0.f);
  pk[0 * n + i] = src[3 * i + 0];
  pk[1 * n + i] = src[3 * i + 1];
  pk[2 * n + i] = src[3 * i + 2];
  pk[3 * n + i] = tgt[3 * i + 0];
  pk[4 * n + i] = tgt[3 * i + 1];
  pk[5 * n + i] = tgt[3 * i + 2];
}

// Cyclic Jacobi on a symmetric 4x4, eigenvectors in v (columns).  Fixed 5 sweeps: Horn matrices of
// 10-point samples have a relative off-diagonal of at most 2.5e-12 after 5 (round-off after 6), five
// orders below the f32 rounding of the hypothesis that is stored.  Rotation from h = (aqq - app) / 2 and
// g = apq as t = sgn(h) g / (|h| + sqrt(h^2 + g^2)) -- the textbook sgn(theta) / (|theta| + sqrt(theta^2 + 1))
// with theta = h / g, without that division (one f64 divide less per rotation; h = 0 gives t = +1).
__device__ __forceinline__ void jacobi4(double a[4][4], double v[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 5; ++sweep) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        const double apq = a[p][q];
        if (apq != 0.0) {
          const double h = 0.5 * (a[q][q] - a[p][p]);
          const double den = fabs(h) + sqrt(h * h + apq * apq);
          const double sg = (h == 0.0 || ((h > 0.0) == (apq > 0.0))) ? 1.0 : -1.0;
          const double t = den > 0.0 ? sg * fabs(apq) / den : sg;
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

// Largest eigenpair of the Horn matrix from its characteristic polynomial (round 4; oracle/corsair_oracle.c
// oc_horn_qcp is the same operation sequence, so the hypotheses stay bit-identical).  N is symmetric and
// traceless: P(l) = l^4 + c2 l^2 + c1 l + c0, c2 = -2 |S|_F^2, c1 = -8 det S, c0 = det N.  All roots are real,
// so Halley's iteration from the upper bound sqrt(3) |S|_F descends monotonically onto the largest one with
// cubic order (3-6 steps, one f64 divide each); the eigenvector is the row of adj(N - l I) with the largest
// diagonal entry.  Against the 5-sweep Jacobi (30 rotations x 2 IEEE sqrt + 2 IEEE divides, ~3 000 f64
// instructions) this is ~400.  Accepted only when the iteration converged and P'(l) >= 0.02 l^3 (the largest
// eigenvalue is well separated); the caller falls back to jacobi4 otherwise (3 in 10^5 samples on the bench).
__device__ __forceinline__ bool horn_qcp(const double (&S)[3][3], const double (&N)[4][4], double (&q)[4]) {
  double f2 = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) f2 = fma(S[a][b], S[a][b], f2);
  const double c2 = -2.0 * f2;
  const double detS = S[0][0] * (S[1][1] * S[2][2] - S[1][2] * S[2][1]) -
                      S[0][1] * (S[1][0] * S[2][2] - S[1][2] * S[2][0]) +
                      S[0][2] * (S[1][0] * S[2][1] - S[1][1] * S[2][0]);
  const double c1 = -8.0 * detS;
  const double u5 = N[0][2] * N[1][3] - N[0][3] * N[1][2];
  const double w0 = N[2][0] * N[3][1] - N[2][1] * N[3][0];
  double c0;
  {
    const double u0 = N[0][0] * N[1][1] - N[0][1] * N[1][0];
    const double u1 = N[0][0] * N[1][2] - N[0][2] * N[1][0];
    const double u2 = N[0][0] * N[1][3] - N[0][3] * N[1][0];
    const double u3 = N[0][1] * N[1][2] - N[0][2] * N[1][1];
    const double u4 = N[0][1] * N[1][3] - N[0][3] * N[1][1];
    const double w1 = N[2][0] * N[3][2] - N[2][2] * N[3][0];
    const double w2 = N[2][0] * N[3][3] - N[2][3] * N[3][0];
    const double w3 = N[2][1] * N[3][2] - N[2][2] * N[3][1];
    const double w4 = N[2][1] * N[3][3] - N[2][3] * N[3][1];
    const double w5 = N[2][2] * N[3][3] - N[2][3] * N[3][2];
    c0 = u0 * w5 - u1 * w4 + u2 * w3 + u3 * w2 - u4 * w1 + u5 * w0;
  }
  double lam = sqrt(3.0 * f2);
  bool conv = false;
  for (int it = 0; it < 8 && !conv; ++it) {
    const double l2 = lam * lam;
    const double P = fma(fma(l2 + c2, lam, c1), lam, c0);
    const double dP = fma(fma(4.0, l2, 2.0 * c2), lam, c1);
    const double ddP = fma(12.0, l2, 2.0 * c2);
    const double d = (2.0 * P * dP) / fma(2.0 * dP, dP, -(P * ddP));
    lam = lam - d;
    conv = fabs(d) <= 1e-6 * lam;  // false for NaN
  }
  {
    const double l2 = lam * lam;
    const double dP = fma(fma(4.0, l2, 2.0 * c2), lam, c1);
    if (!(conv && dP >= 0.02 * (l2 * lam))) return false;
  }
  const double m00 = N[0][0] - lam, m11 = N[1][1] - lam, m22 = N[2][2] - lam, m33 = N[3][3] - lam;
  const double m01 = N[0][1], m02 = N[0][2], m03 = N[0][3], m12 = N[1][2], m13 = N[1][3], m23 = N[2][3];
  const double u0 = m00 * m11 - m01 * m01;
  const double u1 = m00 * m12 - m02 * m01;
  const double u2 = m00 * m13 - m03 * m01;
  const double u3 = m01 * m12 - m02 * m11;
  const double u4 = m01 * m13 - m03 * m11;
  const double w1 = m02 * m23 - m22 * m03;
  const double w2 = m02 * m33 - m23 * m03;
  const double w3 = m12 * m23 - m22 * m13;
  const double w4 = m12 * m33 - m23 * m13;
  const double w5 = m22 * m33 - m23 * m23;
  const double a00 = m11 * w5 - m12 * w4 + m13 * w3;
  const double a01 = -m01 * w5 + m02 * w4 - m03 * w3;
  const double a02 = m13 * u5 - m23 * u4 + m33 * u3;
  const double a03 = -m12 * u5 + m22 * u4 - m23 * u3;
  const double a11 = m00 * w5 - m02 * w2 + m03 * w1;
  const double a12 = -m03 * u5 + m23 * u2 - m33 * u1;
  const double a13 = m02 * u5 - m22 * u2 + m23 * u1;
  const double a22 = m03 * u4 - m13 * u2 + m33 * u0;
  const double a23 = -m02 * u4 + m12 * u2 - m23 * u0;
  const double a33 = m02 * u3 - m12 * u1 + m22 * u0;
  (void)w0;
  // row of the largest |diagonal| (first one on ties), selected without dynamic indexing
  double best = fabs(a00);
  q[0] = a00; q[1] = a01; q[2] = a02; q[3] = a03;
  if (fabs(a11) > best) { best = fabs(a11); q[0] = a01; q[1] = a11; q[2] = a12; q[3] = a13; }
  if (fabs(a22) > best) { best = fabs(a22); q[0] = a02; q[1] = a12; q[2] = a22; q[3] = a23; }
  if (fabs(a33) > best) { best = fabs(a33); q[0] = a03; q[1] = a13; q[2] = a23; q[3] = a33; }
  return best > 0.0;
}

// hyp layout: [prob][12][bmax] (structure of arrays), element 4a+b = R[a][b], 4a+3 = t[a]
// RN = ransac_n when it is known at compile time (10: the reference's value; the sampled pairs then stay
// in registers between the centroid and the covariance pass), 0 = read it from the argument
// Placement table of a round (problem of XCD x, slot i) as a kernel ARGUMENT: the host builds it per round, a
// device copy of it was one hipMemcpyAsync (a blit-kernel launch) per round.  Rounds with more than XCD_SLOTS
// problems per XCD fall back to the device table (xcd_ptr != nullptr).
constexpr int XCD_SLOTS = 64;
struct XcdTab {
  int32_t v[8 * XCD_SLOTS];
};
__device__ __forceinline__ int xcd_problem(const int32_t* __restrict__ xcd_ptr, const XcdTab& tab, int i) {
  return xcd_ptr ? xcd_ptr[i] : tab.v[i];
}

// (defined with the prefilter's operand kernels below)
__device__ __forceinline__ void pf_emit_row(const RansacProb& pr, int p, int h, int bmax, const double (&R)[3][3], double (&t)[3],
                                            const unsigned* __restrict__ stat, const double* __restrict__ sums, double thr2,
                                            double tcap, _Float16* __restrict__ A16, float* __restrict__ c_h);

template <int RN>
__global__ __launch_bounds__(256) void k_ransac_hyp(const RansacProb* probs,
                                                    const float4* __restrict__ pair32, int it0,
                                                    int bcount, int bmax, int ransac_n,
                                                    uint64_t seed,
                                                    const int32_t* __restrict__ xcd_prob, const XcdTab xcd_tab,
                                                    int slots, int tiles, int force_jacobi,
                                                    double* __restrict__ hyp,
                                                    // fused prefilter rows (A16 != nullptr): what k_ransac_hyp16 computes
                                                    const unsigned* __restrict__ pf_stat, const double* __restrict__ pf_sums,
                                                    double thr2, double tcap, _Float16* __restrict__ A16,
                                                    float* __restrict__ c_h, int32_t* __restrict__ cnt_zero) {
  // 1-D grid dealt round-robin to the XCDs: XCD x samples only the problems xcd_prob[x][.], whose
  // correspondences then stay in that XCD's L2 (the sampling is a random gather of 24-B rows)
  const int xcd = blockIdx.x & 7;
  const int item = blockIdx.x >> 3;
  const int slot = item / tiles;
  const int p = xcd_problem(xcd_prob, xcd_tab, xcd * slots + slot);
  if (p < 0) return;
  const int h = (item - slot * tiles) * blockDim.x + threadIdx.x;
  if (h >= bcount) return;
  // the prefilter behind this kernel adds the partial counts of its pair-range splits with atomics: cleared here
  if (cnt_zero) cnt_zero[(int64_t)p * bmax + h] = 0;
  const RansacProb pr = prob_view(probs, p);
  const int itr = it0 + h;
  if (pr.done || itr >= pr.est_k) return;
  const uint32_t m = (uint32_t)pr.m;
  double cs_[3] = {0, 0, 0}, ct_[3] = {0, 0, 0};
  double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  if (RN > 0) {
    float4 pa[RN > 0 ? RN : 1];
    float2 pb[RN > 0 ? RN : 1];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int64_t i = pr.off + rng_index(seed, (uint64_t)itr, (uint64_t)j, m);
      const float4 a = pair32[2 * i], b = pair32[2 * i + 1];  // one 32-B sector
      pa[j] = a;
      pb[j] = make_float2(b.x, b.y);
      cs_[0] += (double)a.x;
      cs_[1] += (double)a.y;
      cs_[2] += (double)a.z;
      ct_[0] += (double)a.w;
      ct_[1] += (double)b.x;
      ct_[2] += (double)b.y;
    }
    const double dn = (double)RN;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      cs_[a] = cs_[a] / dn;
      ct_[a] = ct_[a] / dn;
    }
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const double ds[3] = {(double)pa[j].x - cs_[0], (double)pa[j].y - cs_[1], (double)pa[j].z - cs_[2]};
      const double dt[3] = {(double)pa[j].w - ct_[0], (double)pb[j].x - ct_[1], (double)pb[j].y - ct_[2]};
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) S[a][b] = fma(ds[a], dt[b], S[a][b]);
    }
  } else {
    for (int j = 0; j < ransac_n; ++j) {
      const int64_t i = pr.off + rng_index(seed, (uint64_t)itr, (uint64_t)j, m);
      const float4 a = pair32[2 * i], b = pair32[2 * i + 1];  // one 32-B sector
      cs_[0] += (double)a.x;
      cs_[1] += (double)a.y;
      cs_[2] += (double)a.z;
      ct_[0] += (double)a.w;
      ct_[1] += (double)b.x;
      ct_[2] += (double)b.y;
    }
    const double dn = (double)ransac_n;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      cs_[a] = cs_[a] / dn;
      ct_[a] = ct_[a] / dn;
    }
    for (int j = 0; j < ransac_n; ++j) {
      const int64_t i = pr.off + rng_index(seed, (uint64_t)itr, (uint64_t)j, m);
      const float4 a = pair32[2 * i], b = pair32[2 * i + 1];
      const double ds[3] = {(double)a.x - cs_[0], (double)a.y - cs_[1], (double)a.z - cs_[2]};
      const double dt[3] = {(double)a.w - ct_[0], (double)b.x - ct_[1], (double)b.y - ct_[2]};
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) S[a][b] = fma(ds[a], dt[b], S[a][b]);
    }
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
  double qv[4];
  if (force_jacobi || !horn_qcp(S, N, qv)) {
    // rare (ill-separated largest eigenvalue): the lanes that need it run the Jacobi solver
    jacobi4(N, V);
    // eigenvector of the largest eigenvalue (ties -> lowest index), selected without dynamic indexing
    double best = N[0][0];
    qv[0] = V[0][0]; qv[1] = V[1][0]; qv[2] = V[2][0]; qv[3] = V[3][0];
#pragma unroll
    for (int c = 1; c < 4; ++c) {
      if (N[c][c] > best) {
        best = N[c][c];
        qv[0] = V[0][c];
        qv[1] = V[1][c];
        qv[2] = V[2][c];
        qv[3] = V[3][c];
      }
    }
  }
  double qw = qv[0], qx = qv[1], qy = qv[2], qz = qv[3];
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
  double* o = hyp + ((int64_t)p * 12) * bmax + h;
  double tv[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double t = ct_[a] - (R[a][0] * cs_[0] + R[a][1] * cs_[1] + R[a][2] * cs_[2]);
    tv[a] = t;
    o[(int64_t)(4 * a + 0) * bmax] = R[a][0];
    o[(int64_t)(4 * a + 1) * bmax] = R[a][1];
    o[(int64_t)(4 * a + 2) * bmax] = R[a][2];
    o[(int64_t)(4 * a + 3) * bmax] = t;
  }
  if (A16) pf_emit_row(pr, p, h, bmax, R, tv, pf_stat, pf_sums, thr2, tcap, A16, c_h);
}

// ------------------------------------------------------------------------------------------------
// Exact inlier counts, f64 (Open3D evaluates Matrix4d * Vector4d and squaredNorm in double).
// grid: x = (hypothesis tile of 256) * splits + split, y = problem; block = 256 lanes = 256 hypotheses.
// A lane keeps its hypothesis in 12 f64 registers and walks the pair range of its split; pairs are
// staged 256 at a time: each thread loads one pair (6 coalesced f32 loads from the SoA copy), converts
// it to f64 once and stores it as one 48-B LDS row, which all lanes then read as broadcasts.
// Canonical chain (oracle/corsair_oracle.c oc_ransac):
//   p_c = fma(r_c2, sz, fma(r_c1, sy, fma(r_c0, sx, t_c))),  d_c = p_c - q_c,
//   |d|^2 = fma(dz, dz, fma(dy, dy, dx dx)),  inlier iff |d|^2 < max_corr^2 (all f64).
// ------------------------------------------------------------------------------------------------
constexpr int RC_CHUNK = 256;   // pairs per LDS stage = threads per workgroup
constexpr int RC_HYP = 256;     // hypotheses per workgroup

__device__ __forceinline__ double residual2_f64(const double (&R)[12], double sx, double sy, double sz,
                                                double qx, double qy, double qz) {
  const double dx = fma(R[2], sz, fma(R[1], sy, fma(R[0], sx, R[3]))) - qx;
  const double dy = fma(R[6], sz, fma(R[5], sy, fma(R[4], sx, R[7]))) - qy;
  const double dz = fma(R[10], sz, fma(R[9], sy, fma(R[8], sx, R[11]))) - qz;
  return fma(dz, dz, fma(dy, dy, dx * dx));
}

// LIST: the hypotheses are the survivors of the prefilter, hlist[p][0 .. n_surv[p]) (any order).
// HPW = hypotheses per workgroup: 256 (one per lane) or 64 (round 5: the FIRST chunk of a call, 64 iterations counted
// exactly before there is a best count to prune against -- lane = hypothesis + 64 x quarter, every quarter (= wave) takes
// every fourth staged pair and the four partial counts meet in the integer atomics the pair-range splits use anyway).
template <bool LIST, int HPW>
__device__ __forceinline__ void ransac_count_tile(double (*lds)[RC_CHUNK][6], const int p, const int tile,
                                                  const int split,
                                                  const RansacProb* __restrict__ probs,
                                                  const float* __restrict__ pk, int64_t total,
                                                  const double* __restrict__ hyp, int it0,
                                                  int bcount, int bmax, int splits, double thr2,
                                                  int32_t* __restrict__ res_cnt,
                                                  const int32_t* __restrict__ hlist,
                                                  const int32_t* __restrict__ n_surv) {
  const RansacProb pr = probs[p];
  if (pr.done) return;
  static_assert(HPW == RC_HYP || (!LIST && HPW == 64), "hypotheses per workgroup");
  constexpr int NPART = RC_HYP / HPW;                     // lanes that share a hypothesis (pair-interleaved)
  const int nlist = LIST ? n_surv[p] : 0;
  if (LIST) {
    if (tile * HPW >= nlist) return;
  } else {
    if (it0 + tile * HPW >= pr.est_k || tile * HPW >= bcount) return;  // whole block beyond the bound
  }
  const int tid = threadIdx.x;
  const int part = tid / HPW;
  const int h = tile * HPW + (tid - part * HPW);          // hypothesis slot of this lane
  const bool mine = LIST ? h < nlist : (h < bcount && it0 + h < pr.est_k);
  const int hsel = LIST ? hlist[(int64_t)p * bmax + min(h, nlist - 1)] : min(h, bmax - 1);
  double R[12];
  {
    const double* hp = hyp + ((int64_t)p * 12) * bmax + hsel;
#pragma unroll
    for (int e = 0; e < 12; ++e) R[e] = hp[(int64_t)e * bmax];
  }
  const int per = ((pr.m + splits - 1) / splits + RC_CHUNK - 1) / RC_CHUNK * RC_CHUNK;
  const int beg = split * per;
  const int end = min(pr.m, beg + per);
  int cnt = 0;
  // staging registers: the next stage's pair of this thread is in flight while the current stage is
  // evaluated; rows past the range become far-away targets (never inliers)
  float stg[6];
  auto stage_load = [&](int base) {
    const int i = base + tid;
    const int64_t g = pr.off + (i < end ? i : 0);
#pragma unroll
    for (int c = 0; c < 6; ++c) stg[c] = pk[(int64_t)c * total + g];
  };
  auto stage_store = [&](int b, int base) {
    const bool ok = base + tid < end;
#pragma unroll
    for (int c = 0; c < 6; ++c) lds[b][tid][c] = ok ? (double)stg[c] : (c >= 3 ? 1.0e30 : 0.0);
  };
  if (beg < end) {
    stage_load(beg);
    stage_store(0, beg);
  }
  int buf = 0;
  for (int base = beg; base < end; base += RC_CHUNK) {
    __syncthreads();
    const bool more = base + RC_CHUNK < end;
    if (more) stage_load(base + RC_CHUNK);
    const int nrow = min(RC_CHUNK, end - base);
    if (nrow == RC_CHUNK) {
#pragma unroll 4
      for (int j = part; j < RC_CHUNK; j += NPART) {
        const double* q = lds[buf][j];
        cnt += residual2_f64(R, q[0], q[1], q[2], q[3], q[4], q[5]) < thr2 ? 1 : 0;
      }
    } else {
      for (int j = part; j < nrow; j += NPART) {
        const double* q = lds[buf][j];
        cnt += residual2_f64(R, q[0], q[1], q[2], q[3], q[4], q[5]) < thr2 ? 1 : 0;
      }
    }
    if (more) stage_store(buf ^ 1, base + RC_CHUNK);
    buf ^= 1;
  }
  if (mine) {
    if (splits == 1 && NPART == 1)
      res_cnt[(int64_t)p * bmax + hsel] = cnt;
    else
      atomicAdd(&res_cnt[(int64_t)p * bmax + hsel], cnt);   // (res_cnt of the chunk is zero on entry)
  }
}

// grid: x = (hypothesis tile of 256) * splits + split, y = problem.  LIST: the survivor count is only
// known on the device, so a fixed number of tile slots (gridDim.x / splits) strides over the list.
template <bool LIST, int HPW = RC_HYP>
__global__ __launch_bounds__(256) void k_ransac_count(const RansacProb* __restrict__ probs,
                                                      const float* __restrict__ pk, int64_t total,
                                                      const double* __restrict__ hyp, int it0,
                                                      int bcount, int bmax, int splits, double thr2,
                                                      int32_t* __restrict__ res_cnt,
                                                      const int32_t* __restrict__ hlist,
                                                      const int32_t* __restrict__ n_surv) {
  // [buf][j][c]: c = 0..2 source xyz, c = 3..5 target xyz of pair j, f64 (one 48-B row per pair)
  __shared__ __attribute__((aligned(16))) double lds[2][RC_CHUNK][6];
  const int p = blockIdx.y;
  const int tile0 = blockIdx.x / splits;
  const int split = blockIdx.x - tile0 * splits;
  if (LIST) {
    const int nlist = n_surv[p];
    const int tstride = gridDim.x / splits;
    for (int tile = tile0; tile * RC_HYP < nlist; tile += tstride) {
      ransac_count_tile<LIST, HPW>(lds, p, tile, split, probs, pk, total, hyp, it0, bcount, bmax, splits, thr2,
                                   res_cnt, hlist, n_surv);
      __syncthreads();  // the next tile restages LDS
    }
  } else {
    ransac_count_tile<LIST, HPW>(lds, p, tile0, split, probs, pk, total, hyp, it0, bcount, bmax, splits, thr2,
                                 res_cnt, hlist, n_surv);
  }
}

// ------------------------------------------------------------------------------------------------
// Exactness-preserving f16 prefilter.
// Once a problem has a best inlier count, a hypothesis matters only if its own count can reach it
// (otherwise it changes neither the best, nor the early-exit bound, nor the tie set).  The squared
// residual is bilinear in hypothesis and pair quantities,
//   |R s + t - q|^2 = |t|^2 + a . b,   a = [1, 2 R^T t, -2 R, -2 t],  b = [|s|^2 + |q|^2, s, q (x) s, q]   (16 terms)
// The pair side is split into f16 hi + lo, the hypothesis side is rounded to f16 (a_hi): a_hi . b_hi +
// a_hi . b_lo (K = 32) is two v_mfma_f32_32x32x16_f16 per 32 x 32 tile (16x the f32 matrix rate) whose
// accumulator INPUT holds |t|^2 - (thr^2 + eps_h): the sign of the result says whether the pair is
// within the INFLATED threshold.  eps_h bounds |d~^2 - d^2| (pf_emit_row) -- including the dropped
// (a - a_hi) . b, bounded per hypothesis with the per-problem maxima of |b_k| -- so the sign count is an
// UPPER bound of the exact inlier count.  Hypotheses whose bound is below the carried best get count
// 0, the few survivors go through the exact f64 kernels: results are unchanged bit for bit.
// (The K = 48 form with a_lo . b_hi has a ~2.5x tighter eps_h but 3 MFMAs per tile: measured slower
// end to end, DESIGN.md "What was tried".)
// ------------------------------------------------------------------------------------------------
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int PF_K = 16;        // halfs per hypothesis row (32 B): a_hi, used by both MFMAs
constexpr int PF_PITCH = 40;    // halfs per pair row, K = 32 form (80 B = 5 slots of 16 B: conflict-free ds_read_b128)
constexpr int PF_PITCH1 = 24;   // halfs per pair row, K = 16 form (48 B = 3 slots: rows 0..15 start in 16 different slots of 4 banks)
__host__ __device__ constexpr int pf_pitch(int nm) { return nm == 2 ? PF_PITCH : PF_PITCH1; }
constexpr int PF_ROWS = 192;    // pairs per LDS stage (6 MFMA row tiles)
constexpr int PF_NG = 2;        // 32-hypothesis groups per wave (LDS fragments are reused NG times)
constexpr int PF_HYP = 4 * 32 * PF_NG;  // hypotheses per workgroup
constexpr int PF_STAT = 17;     // per-problem statistics of the pair image: smax, max |b_k| (k = 0..15)
constexpr float PF_SMAX = 128.0f;       // point norm above which a problem bypasses the prefilter
                                        // (f16 range: |s|^2 + |q|^2 and q (x) s must stay below 65504)

// f64 -> f16 through f32 (v_cvt_f32_f64 + v_cvt_f16_f32).  gfx950 has no direct conversion: `(_Float16)double` is a
// ~25-instruction integer sequence, and the prefilter's operand kernels make 16 - 32 of them per hypothesis and per pair (a sixth
// of k_ransac_hyp's instructions).  The two roundings can differ from the single one by one f16 ulp in rare ties; nothing
// below assumes a correctly rounded value -- every bound is computed from the value this function RETURNS (|x - f16_of(x)|),
// and its relative error 2^-11 + 2^-24 sits inside the constants' slack (2.002 for 2 sqrt(1.001), 1.0005).
__device__ __forceinline__ _Float16 f16_of(double v) { return (_Float16)(float)v; }

__device__ __forceinline__ void split16(double v, _Float16* hi, _Float16* lo) {
  const _Float16 h = f16_of(v);
  *hi = h;
  *lo = f16_of(v - (double)h);
}

// rows of problem p in the f16 pair image: m rounded up to whole LDS stages
__host__ __device__ static inline int64_t pf_padded(int64_t m) { return (m + PF_ROWS - 1) / PF_ROWS * PF_ROWS; }

// pair side: 80-B rows [bh(0..15) | bl(0..15) | 8 x 0] in exactly the layout the prefilter keeps in LDS
// (a stage is one contiguous 15-KiB copy); every problem is padded to whole stages with rows whose d~^2
// is +60000 (never counted).  stat[p] = {largest point norm, max |b_k| (k = 0..15)} of problem p as
// float bit patterns (non-negative floats order like their bits), rounded up.
// grid: x = blocks over the rows of a problem (grid-stride), y = problem; off16[p] = first row.
// Per-problem sums of the source and target points (mu = sum / m is evaluated with the same expression by every consumer).  The prefilter works in coordinates CENTRED per problem, s' = s - mu_s, q' = q - mu_q: the residual is the same,
// R s' + t' - q' = R s + t - q with t' = t + R mu_s - mu_q (pf_emit_row), but every magnitude the error bounds scale with
// -- smax, W, |t'| = |c'_t - R c'_s| with c' the centroids of the ten sampled points in centred coordinates -- shrinks to the
// spread of the problem's points.  The part-to-part problems of split_corr (utils/symmetry.py:145-179: a leg against a leg)
// sit far from the origin; without the centring 30 % of their hypotheses exceeded the |t| cap of the K = 16 form.
__global__ __launch_bounds__(256) void k_ransac_pair_sums(const RansacProb* __restrict__ probs, const float* __restrict__ src,
                                                          const float* __restrict__ tgt, double* __restrict__ sums) {
  // ONE workgroup per problem and a fixed reduction order: the means -- and with them the prefilter's survivor sets -- are
  // the same in every run (an atomic accumulation made the survivor counts of otherwise identical runs differ by 1e-4)
  __shared__ double red[4][6];
  const RansacProb pr = probs[blockIdx.x];
  double a[6] = {0, 0, 0, 0, 0, 0};
  for (int j = threadIdx.x; j < pr.m; j += 256) {
    const int64_t i = pr.off + j;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a[c] += (double)src[3 * i + c];
      a[3 + c] += (double)tgt[3 * i + c];
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a[c] += __shfl_xor(a[c], off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = a[c];
  }
  __syncthreads();
  // what is stored is the MEAN (the six f64 divisions were made by every hypothesis and every pair row that read the sums)
  if (threadIdx.x < 6) {
    const double sum = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    const double v = sum / (double)(pr.m > 0 ? pr.m : 1);
    sums[blockIdx.x * 6 + threadIdx.x] = (v == v && fabs(v) < 1.0e30) ? v : 0.0;   // non-finite input: no centring (the rows are rejected by their norm)
  }
}
__device__ __forceinline__ void pf_centre(const double* __restrict__ sums, int p, int m, double (&mu)[6]) {
  (void)m;
#pragma unroll
  for (int c = 0; c < 6; ++c) mu[c] = sums[p * 6 + c];
}

// NM = 2: rows [bh | bl | pad] of the K = 32 form.  NM = 1 (round 4): rows [bh | pad] of the K = 16 form -- the matrix pipe
// then evaluates a_hi . b_hi only, and what it drops, a_hi . b_lo, is bounded PER PAIR and taken out of the pair's constant
// term b_0 (a_0 = 1 exactly) by k_ransac_pack16_b0 below, so the sign test stays an upper bound (see there).
template <int NM>
__global__ __launch_bounds__(256) void k_ransac_pack16(const RansacProb* __restrict__ probs,
                                                       const int64_t* __restrict__ off16,
                                                       const float* __restrict__ src,
                                                       const float* __restrict__ tgt,
                                                       const double* __restrict__ sums,
                                                       _Float16* __restrict__ B16,
                                                       unsigned* __restrict__ stat) {
  __shared__ float red[4][PF_STAT];
  const RansacProb pr = probs[blockIdx.y];
  double mu[6];
  pf_centre(sums, blockIdx.y, pr.m, mu);
  const int mpad = (int)pf_padded(pr.m);
  float mx[PF_STAT];
#pragma unroll
  for (int k = 0; k < PF_STAT; ++k) mx[k] = 0.f;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < mpad; j += gridDim.x * blockDim.x) {
    constexpr int NV = pf_pitch(NM) / 8;   // 16-B pieces per row
    union {
      _Float16 h[PF_PITCH];
      uint4 v[5];
    } row;
#pragma unroll
    for (int k = 0; k < 5; ++k) row.v[k] = make_uint4(0u, 0u, 0u, 0u);
    if (j < pr.m) {
      const int64_t i = pr.off + j;
      const double sx = src[3 * i] - mu[0], sy = src[3 * i + 1] - mu[1], sz = src[3 * i + 2] - mu[2];
      const double qx = tgt[3 * i] - mu[3], qy = tgt[3 * i + 1] - mu[4], qz = tgt[3 * i + 2] - mu[5];
      const double ss = sx * sx + sy * sy + sz * sz, qq = qx * qx + qy * qy + qz * qz;
      double b[16];
      b[0] = ss + qq;
      b[1] = sx; b[2] = sy; b[3] = sz;
      b[4] = qx * sx; b[5] = qx * sy; b[6] = qx * sz;
      b[7] = qy * sx; b[8] = qy * sy; b[9] = qy * sz;
      b[10] = qz * sx; b[11] = qz * sy; b[12] = qz * sz;
      b[13] = qx; b[14] = qy; b[15] = qz;
      // 1.0000002: the f32 norm may round down
      float mag = 1.0000002f * (float)sqrt(fmax(ss, qq));
      // out of f16 range (or NaN): a finite (zero) row; smax then marks the problem's hypotheses unusable
      const bool ok = mag <= PF_SMAX;
      if (!(mag == mag)) mag = INFINITY;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        _Float16 hi = (_Float16)0.0f, lo = (_Float16)0.0f;
        if (ok) {
          split16(b[k], &hi, &lo);
          mx[1 + k] = fmaxf(mx[1 + k], __double2float_ru(fabs(b[k])));
        }
        row.h[k] = hi;
        if (NM == 2) row.h[16 + k] = lo;
      }
      mx[0] = fmaxf(mx[0], mag);
    } else {
      row.h[0] = (_Float16)60000.0f;  // pairs with a_0 = 1
    }
    uint4* dst = reinterpret_cast<uint4*>(B16 + (off16[blockIdx.y] + j) * pf_pitch(NM));
#pragma unroll
    for (int k = 0; k < NV; ++k) dst[k] = row.v[k];
  }
#pragma unroll
  for (int k = 0; k < PF_STAT; ++k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = mx[k];
  }
  __syncthreads();
  if (threadIdx.x < PF_STAT) {
    const int k = threadIdx.x;
    const float m = fmaxf(fmaxf(red[0][k], red[1][k]), fmaxf(red[2][k], red[3][k]));
    if (m > 0.f) atomicMax(&stat[blockIdx.y * PF_STAT + k], __float_as_uint(m));
  }
}

// K = 16 form, second pass over the pairs (needs the problem's smax, which the first pass produces): the constant term of
// every pair becomes   b_0' = round_down_f16( b_0 - E_p ),   E_p = (1 + 2^-10) sum_{k=1..15} A_k |b_k - hi(b_k)|,
// with A_k an upper bound of |a_hi_k| over all USABLE hypotheses of the problem:
//   k = 4..12  (a = -2 R):          |a| <= 2 sqrt(1 + max|E|) <= 2.002   (pf_emit_row requires max|E| < 1e-3)
//   k = 1..3, 13..15 (2 R^T t, -2 t): |a| <= 2 |t| sqrt(1 + max|E|) with |t| <= tcap * smax: pf_emit_row CHECKS that and
//     marks the other hypotheses unusable (they survive to the exact kernels).  |t| = |c_t - R c_s| can reach 2 smax, but
//     both centroids are means of ten points of a centred object: on the bench clouds |t| / smax has median 0.2 and
//     99.99 % of the hypotheses are below 0.8, so tcap = 0.75 (CS_RANSAC_PF_TCAP) costs 1e-4 of them and shrinks E_p 2.7x
// The constant term is also CENTRED: b_0 - beta with beta = smax^2 (b_0 = |s|^2 + |q|^2 lies in [0, 2 smax^2]); the
// hypothesis side adds beta to its accumulator input.  f16 is finer near zero: the round-down costs ~6e-5 instead of 2.4e-4.
// and |a_hi| <= |a| (1 + 2^-11).  Then  sum_k a_hi_k b'_k  <=  sum_k a_hi_k (b_hi_k + b_lo_k)  for every usable hypothesis:
// the one-MFMA value is never above what the K = 32 form computes exactly, i.e. every pair the K = 32 form counts is
// counted -- the count stays an UPPER bound (pf_emit_row's eps_h covers the rest as before).  The price is a looser
// bound: E_p is ~1e-3 for unit-sized objects (2.5 % of thr^2 = 0.04), the rounding of b_0 another ~2.4e-4 on average.
__global__ __launch_bounds__(256) void k_ransac_pack16_b0(const RansacProb* __restrict__ probs,
                                                          const int64_t* __restrict__ off16,
                                                          const float* __restrict__ src, const float* __restrict__ tgt,
                                                          const double* __restrict__ sums,
                                                          const unsigned* __restrict__ stat, double tcap,
                                                          _Float16* __restrict__ B16) {
  const RansacProb pr = probs[blockIdx.y];
  double mu[6];
  pf_centre(sums, blockIdx.y, pr.m, mu);
  const double smax = (double)__uint_as_float(stat[blockIdx.y * PF_STAT]);
  if (!(smax <= (double)PF_SMAX)) return;   // the problem bypasses the prefilter (every hypothesis unusable)
  const double beta = smax * smax;
  const double a_rot = 2.002 * (1.0 + 0x1p-11), a_t = 2.0 * tcap * smax * 1.0005 * (1.0 + 0x1p-11);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < pr.m; j += gridDim.x * blockDim.x) {
    const int64_t i = pr.off + j;
    const double sx = src[3 * i] - mu[0], sy = src[3 * i + 1] - mu[1], sz = src[3 * i + 2] - mu[2];
    const double qx = tgt[3 * i] - mu[3], qy = tgt[3 * i + 1] - mu[4], qz = tgt[3 * i + 2] - mu[5];
    const double ss = sx * sx + sy * sy + sz * sz, qq = qx * qx + qy * qy + qz * qz;
    const double b[16] = {ss + qq, sx, sy, sz, qx * sx, qx * sy, qx * sz, qy * sx, qy * sy, qy * sz,
                          qz * sx, qz * sy, qz * sz, qx, qy, qz};
    double e_t = 0.0, e_rot = 0.0;
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const double lo = fabs(b[k] - (double)f16_of(b[k]));
      if (k >= 4 && k <= 12) e_rot += lo; else e_t += lo;
    }
    const double ep = (1.0 + 0x1p-10) * (a_rot * e_rot + a_t * e_t);
    // round toward -inf into f16: RNE first, one ulp down when that landed above
    const double v = (b[0] - beta) - ep - 0x1p-40 * (fabs(b[0]) + beta + ep);   // (the f64 roundings of the terms themselves)
    _Float16 h = f16_of(v);
    if ((double)h > v) {
      unsigned short u = __builtin_bit_cast(unsigned short, h);
      // next representable value below: magnitude down for positive values, up for negative ones (+0 -> -min subnormal)
      u = (u & 0x8000u) ? (unsigned short)(u + 1) : (u == 0 ? (unsigned short)0x8001u : (unsigned short)(u - 1));
      h = __builtin_bit_cast(_Float16, u);
    }
    B16[(off16[blockIdx.y] + j) * PF_PITCH1] = h;
  }
}

// hypothesis side: row = a_hi(0..15) (f16 roundings of a) and the accumulator input
//   c_h = |t|^2 - (thr^2 + eps_h).
// eps_h >= |d~^2 - d^2| where d^2 is what the exact (f64) kernels compute and d~^2 the f16 pipeline
// c_h + sum_k a_hi_k (b_hi_k + b_lo_k):
//   * 32 products, exact in f32; their accumulation rounds (or truncates) at most 33 times relative
//     to sum_k |a_k b_k| <= sqrt(3) (|s| + |q| + |t|)^2 =: sqrt(3) W            <= 33 * 2^-23 * sqrt(3) W
//   * the residuals of the hi+lo splits of b                            <= 2^-22 * sqrt(3) W + 2^-25 (2 W + 59)
//   * (the exact kernels evaluated d^2 in f32 when this budget was set:   <= 2^-20 W; they are f64 now and
//      the term is kept as slack)
//   => < 8.3e-6 W + 1.8e-6; charged 2.5e-5 W + 6e-6 (3x margin).  The accumulation term assumes one ulp per
//   addition; measured, the two chained MFMAs are within 2.3 ulp in total (tools/ubench/mfma_err.hip,
//   4e8 results), so the charge is ~30x the observed error.  CS_RANSAC_CHECK runs validate the bound.
//   * the dropped (a - a_hi) . b                      <= sum_k |a_k - a_hi_k| max_pairs |b_k|   (stat[p])
//   * |R s|^2 = |s|^2 only up to the orthonormality defect E = R^T R - I:    <= 3 max|E| smax^2
// with W <= (2 smax + |t|)^2.  A hypothesis outside the f16 range (or not finite) gets c_h = -inf and
// a zero row: every pair counts, it always survives to the exact kernel.
// Prefilter row of one hypothesis (R, t): 16 f16 coefficients + the f32 constant c_h (see k_ransac_prefilter).  Called by
// k_ransac_hyp16 (hypotheses read back from the table) and, fused, by k_ransac_hyp itself (round 4: one launch and one
// 96-byte read per hypothesis less).
__device__ __forceinline__ void pf_emit_row(const RansacProb& pr, int p, int h, int bmax, const double (&R)[3][3], double (&t)[3],
                                            const unsigned* __restrict__ stat, const double* __restrict__ sums, double thr2,
                                            double tcap, _Float16* __restrict__ A16, float* __restrict__ c_h) {
  // the pair image is in centred coordinates: t' = t + R mu_s - mu_q (see k_ransac_pair_sums); the f64 rounding of these
  // nine operations (<= 1e-15 (|t| + |mu|)) sits far inside the 6e-6 of eps
  {
    double mu[6];
    pf_centre(sums, p, pr.m, mu);
#pragma unroll
    for (int a = 0; a < 3; ++a) t[a] = t[a] + (R[a][0] * mu[0] + R[a][1] * mu[1] + R[a][2] * mu[2]) - mu[3 + a];
  }
  const double smax = (double)__uint_as_float(stat[p * PF_STAT]);
  const double tt = t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
  const double tn = sqrt(tt);
  double a[16];
  a[0] = 1.0;
#pragma unroll
  for (int b = 0; b < 3; ++b) a[1 + b] = 2.0 * (R[0][b] * t[0] + R[1][b] * t[1] + R[2][b] * t[2]);
#pragma unroll
  for (int x = 0; x < 3; ++x)
#pragma unroll
    for (int b = 0; b < 3; ++b) a[4 + 3 * x + b] = -2.0 * R[x][b];
#pragma unroll
  for (int x = 0; x < 3; ++x) a[13 + x] = -2.0 * t[x];
  double dev = 0.0;
#pragma unroll
  for (int x = 0; x < 3; ++x)
#pragma unroll
    for (int y = 0; y < 3; ++y) {
      const double e = R[0][x] * R[0][y] + R[1][x] * R[1][y] + R[2][x] * R[2][y] - (x == y ? 1.0 : 0.0);
      dev = fmax(dev, fabs(e));
    }
  bool usable = smax <= (double)PF_SMAX && tn <= 4.0 * (double)PF_SMAX && dev < 1.0e-3;
  // K = 16 form: the per-pair bound of the dropped a_hi . b_lo (k_ransac_pack16_b0) assumes |t| <= 2.002 smax
  // (tcap > 0) and its constant term is centred by beta = smax^2, which comes back through c_h
  if (tcap > 0.0) usable = usable && tn <= tcap * smax;
  usable = usable && thr2 < 3.0e4;   // the padding rows (b_0 = 60000) must stay positive: c_h > -60000
  const double beta = tcap > 0.0 ? smax * smax : 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) usable = usable && fabs(a[k]) < 6.0e4;  // false for NaN
  union {
    _Float16 h[PF_K];
    uint4 v[2];
  } row;
  double drop = 0.0;  // sum_k |a_k - a_hi_k| max |b_k|
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const _Float16 hi = usable ? f16_of(a[k]) : (_Float16)0.0f;
    row.h[k] = hi;
    if (usable) drop += fabs(a[k] - (double)hi) * (double)__uint_as_float(stat[p * PF_STAT + 1 + k]);
  }
  uint4* dst = reinterpret_cast<uint4*>(A16 + ((int64_t)p * bmax + h) * PF_K);
  dst[0] = row.v[0];
  dst[1] = row.v[1];
  const double w = 2.0 * smax + tn;
  const double eps = 2.5e-5 * w * w + 6.0e-6 + 3.0 * dev * smax * smax + 1.000001 * drop;
  // rounded towards -inf so that the f32 value never tightens the test
  // unusable: a zero row and c_h = -1, so every row (padding included) counts and the hypothesis survives.  (FINITE: the
  // round-toward-minus-infinity counters of k_ransac_prefilter<1, true> add the results themselves.)
  c_h[(int64_t)p * bmax + h] = usable ? __double2float_rd((tt + beta) - (thr2 + eps)) : -1.0f;
}

__global__ void k_ransac_hyp16(const RansacProb* probs, const double* __restrict__ hyp,
                               const unsigned* __restrict__ stat, const double* __restrict__ sums, int it0, int bcount,
                               int bmax, double thr2, _Float16* __restrict__ A16, float* __restrict__ c_h,
                               int32_t* __restrict__ cnt_zero, double tcap) {
  const int p = blockIdx.y;
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= bcount) return;
  // the prefilter behind this kernel adds the partial counts of its pair-range splits with atomics: the
  // counters of this round are cleared here (a hipMemset2DAsync per round before)
  if (cnt_zero) cnt_zero[(int64_t)p * bmax + h] = 0;
  const RansacProb pr = prob_view(probs, p);
  if (pr.done || it0 + h >= pr.est_k) return;
  const double* hp = hyp + ((int64_t)p * 12) * bmax + h;
  double R[3][3], t[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int b = 0; b < 3; ++b) R[a][b] = hp[(int64_t)(4 * a + b) * bmax];
    t[a] = hp[(int64_t)(4 * a + 3) * bmax];
  }
  pf_emit_row(pr, p, h, bmax, R, t, stat, sums, thr2, tcap, A16, c_h);
}

// Upper bounds of the inlier counts.
// grid: 1-D, 8 * slots * tiles * splits workgroups.  Workgroups are dealt round-robin to the 8 XCDs, so
// XCD x = id % 8 is given the problems xcd_prob[x][0..slots) (host: longest-first balancing): all
// workgroups that stream one problem's pair image run on ONE XCD at about the same time and share it
// through that XCD's 4-MiB L2 instead of each pulling it from HBM / Infinity Cache.
// MFMA operand maps (v_mfma_f32_32x32x16_f16): lane l supplies A[row l&31][k = 8(l>>5) .. +8) and
// B[k = 8(l>>5) .. +8)][col l&31]; D as for the f32 shape.  rows = pairs (LDS, shared by the four
// waves), cols = hypotheses (registers, PF_NG groups of 32 per wave).
//
// Staging: a stage is PF_ROWS rows = 15 KiB, contiguous in the pair image, copied global -> LDS by
// 15 LDS-DMA instructions of 1 KiB (global_load_lds_dwordx4: no staging registers, no ds_write); the
// copy of stage s+1 is in flight while stage s is computed.
//
// Inner loop: units k = (row tile t, hypothesis group g), 12 per stage.  The two MFMAs of unit k are
// issued interleaved with the sign extraction (16 x v_alignbit into a per-lane history word, one VALU
// op per pair) of unit k-2, held in another of three rotating accumulator sets: a result is first
// read a whole unit (>= 64 cycles) after the MFMA that wrote it, beyond the 11 wait states the
// hardware requires.  The VALU side is the longer one (v_alignbit_b32 issues every ~4.5 cycles per
// SIMD, tools/ubench/valu_rate.hip: 16 x 4.5 = 72 cycles against 64 for the MFMAs).  The unit is one asm block: the compiler's scheduler does not keep this order
// (it hoists the dependent VALU ops and pays s_nop 10 per unit).
// RTN (with NM = 1): the signs are counted by the results THEMSELVES -- under round-toward-minus-infinity (MODE.fp_round)
// and with a counter in [2^63, 2^64), whose ulp is 2^40, `v_add_f32 cnt, acc, cnt` subtracts exactly 2^40 iff acc < 0 for
// any |acc| < 2^40 (tools/ubench/rtn_count.hip: edge cases incl. -0 and denormals) -- one FULL-RATE VALU op per result
// instead of a 4.5-cycle v_alignbit.  Full-rate ops do not overlap with the matrix pipe, but the K = 16 unit has only one
// MFMA: 32 + 16 x 2.3 = 69 cycles against 72+ (tools/ubench/pf_k16_mix.hip: 29.2 vs 32.8 ns per unit per SIMD).
template <int NM, bool RTN>   // NM = MFMAs per unit: 2 = K 32 (a_hi . (b_hi + b_lo)), 1 = K 16 (a_hi . b_hi', k_ransac_pack16<1> + _b0)
__global__ __launch_bounds__(256) void k_ransac_prefilter(const RansacProb* probs,
                                                          const int64_t* __restrict__ off16,
                                                          const _Float16* __restrict__ B16,
                                                          const _Float16* __restrict__ A16,
                                                          const float* __restrict__ c_h, int it0,
                                                          int bcount, int bmax, int splits,
                                                          const int32_t* __restrict__ xcd_prob,
                                                          const XcdTab xcd_tab, int slots, int tiles,
                                                          int32_t* __restrict__ cnt_up,
                                                          unsigned long long* __restrict__ trace,
                                                          const int32_t* __restrict__ n_list) {
  const unsigned long long t_start = trace ? wall_clock64() : 0ULL;
  const unsigned long long c_start = trace ? __builtin_amdgcn_s_memtime() : 0ULL;
  constexpr int PITCH = pf_pitch(NM);
  constexpr int STAGE_BYTES = PF_ROWS * PITCH * 2;  // 15360 (K 32) / 9216 (K 16)
  constexpr int STAGE_KIB = STAGE_BYTES / 1024;        // 15 LDS-DMA instructions
  static_assert(STAGE_BYTES % 1024 == 0, "a stage must be whole 1-KiB LDS-DMA instructions");
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_BYTES];
  const int xcd = blockIdx.x & 7;
  const int item = blockIdx.x >> 3;
  const int slot = item / (tiles * splits);
  const int inner = item - slot * (tiles * splits);
  const int p = xcd_problem(xcd_prob, xcd_tab, xcd * slots + slot);
  if (p < 0) return;
  const int tile = inner / splits;
  const int split = inner - tile * splits;
  const RansacProb pr = prob_view(probs, p);
  if (pr.done) return;
  // n_list: the hypotheses are a COMPACT per-problem list of n_list[p] rows (second stage over the survivors: rows
  // gathered by k_ransac_stage2_gather, bmax = its row capacity) instead of the iterations it0 .. it0 + bcount of a chunk
  if (n_list) bcount = min(n_list[p], bmax);
  const int ek_rel = n_list ? 0x7fffffff : pr.est_k - it0;   // hypotheses at or beyond it are past the iteration bound
  if (tile * PF_HYP >= ek_rel || tile * PF_HYP >= bcount) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int col = lane & 31;
  const int h0 = tile * PF_HYP + wave * 32 * PF_NG;
  const bool wave_live = h0 < bcount && h0 < ek_rel;
  f16x8 bop[PF_NG];
  f32x16 cin[PF_NG];
#pragma unroll
  for (int g = 0; g < PF_NG; ++g) {
    // hypotheses past the chunk / bound read a valid row; their result is not stored
    int hh = h0 + 32 * g + col;
    if (hh >= bcount || hh >= ek_rel) hh = wave_live ? h0 : 0;
    const _Float16* row = A16 + ((int64_t)p * bmax + hh) * PF_K + 8 * half;
    bop[g] = *reinterpret_cast<const f16x8*>(row);
    const float c = c_h[(int64_t)p * bmax + hh];
#pragma unroll
    for (int r = 0; r < 16; ++r) cin[g][r] = c;
    asm volatile("" : "+v"(cin[g]));  // keep the 16 copies resident instead of re-splatting per tile
  }
  const int mpad = (int)pf_padded(pr.m);
  const int per = ((mpad / PF_ROWS + splits - 1) / splits) * PF_ROWS;
  const int beg = split * per;
  const int end = min(mpad, beg + per);
  static_assert(!RTN || NM == 1, "the add-based sign count is written for the one-MFMA unit");
  unsigned bits[PF_NG];
  int cnt[PF_NG];
  constexpr float RTN_C0 = 0x1p64f - 0x1p40f;   // 2^64 - 2^40: all 24 significand bits set, ulp 2^40
  float fc[PF_NG][2];
#pragma unroll
  for (int g = 0; g < PF_NG; ++g) {
    bits[g] = 0u;
    cnt[g] = 0;
    fc[g][0] = RTN_C0;
    fc[g][1] = RTN_C0;
  }
  const char* gsrc = reinterpret_cast<const char*>(B16 + off16[p] * PITCH) + lane * 16;
  auto issue_stage = [&](int b, int base) {
    const char* g = gsrc + (int64_t)base * (PITCH * 2);
#pragma unroll
    for (int i = 0; i < (STAGE_KIB + 3) / 4; ++i) {
      const int piece = wave + 4 * i;  // wave-uniform
      if (piece < STAGE_KIB)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(g + piece * 1024),
            (__attribute__((address_space(3))) void*)(lds + b * STAGE_BYTES + piece * 1024), 16, 0, 0);
    }
  };
  static_assert(PF_NG == 2 && PF_ROWS == 192, "the unrolled schedule below is written for 12 units per stage");
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 S0, S1 = zero16, S2 = zero16;  // +0: the first two (dummy) extractions shift in zeros
#define PF_UNIT1(DST, SRC, G, A, COUNT) \
  asm volatile( \
      "v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\t" \
      "v_alignbit_b32 %1, %1, %5, 31\n\t" \
      "v_alignbit_b32 %1, %1, %6, 31\n\t" \
      "v_alignbit_b32 %1, %1, %7, 31\n\t" \
      "v_alignbit_b32 %1, %1, %8, 31\n\t" \
      "v_alignbit_b32 %1, %1, %9, 31\n\t" \
      "v_alignbit_b32 %1, %1, %10, 31\n\t" \
      "v_alignbit_b32 %1, %1, %11, 31\n\t" \
      "v_alignbit_b32 %1, %1, %12, 31\n\t" \
      "v_alignbit_b32 %1, %1, %13, 31\n\t" \
      "v_alignbit_b32 %1, %1, %14, 31\n\t" \
      "v_alignbit_b32 %1, %1, %15, 31\n\t" \
      "v_alignbit_b32 %1, %1, %16, 31\n\t" \
      "v_alignbit_b32 %1, %1, %17, 31\n\t" \
      "v_alignbit_b32 %1, %1, %18, 31\n\t" \
      "v_alignbit_b32 %1, %1, %19, 31\n\t" \
      "v_alignbit_b32 %1, %1, %20, 31" \
      : "=&v"(DST), "+v"(bits[G]) \
      : "v"(A[0]), "v"(bop[G]), \
        "v"(cin[G]), "v"(SRC[0]), "v"(SRC[1]), "v"(SRC[2]), "v"(SRC[3]), "v"(SRC[4]), "v"(SRC[5]), \
        "v"(SRC[6]), "v"(SRC[7]), "v"(SRC[8]), "v"(SRC[9]), "v"(SRC[10]), "v"(SRC[11]), \
        "v"(SRC[12]), "v"(SRC[13]), "v"(SRC[14]), "v"(SRC[15])); \
  if (COUNT) cnt[G] += __popc(bits[G]);
#define PF_UNIT2(DST, SRC, G, A, COUNT) \
  asm volatile( \
      "v_mfma_f32_32x32x16_f16 %0, %2, %4, %6\n\t" \
      "v_alignbit_b32 %1, %1, %7, 31\n\t" \
      "v_alignbit_b32 %1, %1, %8, 31\n\t" \
      "v_alignbit_b32 %1, %1, %9, 31\n\t" \
      "v_alignbit_b32 %1, %1, %10, 31\n\t" \
      "v_alignbit_b32 %1, %1, %11, 31\n\t" \
      "v_alignbit_b32 %1, %1, %12, 31\n\t" \
      "v_alignbit_b32 %1, %1, %13, 31\n\t" \
      "v_alignbit_b32 %1, %1, %14, 31\n\t" \
      "v_mfma_f32_32x32x16_f16 %0, %3, %5, %0\n\t" \
      "v_alignbit_b32 %1, %1, %15, 31\n\t" \
      "v_alignbit_b32 %1, %1, %16, 31\n\t" \
      "v_alignbit_b32 %1, %1, %17, 31\n\t" \
      "v_alignbit_b32 %1, %1, %18, 31\n\t" \
      "v_alignbit_b32 %1, %1, %19, 31\n\t" \
      "v_alignbit_b32 %1, %1, %20, 31\n\t" \
      "v_alignbit_b32 %1, %1, %21, 31\n\t" \
      "v_alignbit_b32 %1, %1, %22, 31" \
      : "=&v"(DST), "+v"(bits[G]) \
      : "v"(A[0]), "v"(A[1]), "v"(bop[G]), "v"(bop[G]), \
        "v"(cin[G]), "v"(SRC[0]), "v"(SRC[1]), "v"(SRC[2]), "v"(SRC[3]), "v"(SRC[4]), "v"(SRC[5]), \
        "v"(SRC[6]), "v"(SRC[7]), "v"(SRC[8]), "v"(SRC[9]), "v"(SRC[10]), "v"(SRC[11]), \
        "v"(SRC[12]), "v"(SRC[13]), "v"(SRC[14]), "v"(SRC[15])); \
  if (COUNT) cnt[G] += __popc(bits[G]);
#define PF_UNITR(DST, SRC, G, GC, A) \
  asm volatile( \
      "v_mfma_f32_32x32x16_f16 %0, %3, %4, %5\n\t" \
      "v_add_f32 %1, %6, %1\n\t" \
      "v_add_f32 %2, %7, %2\n\t" \
      "v_add_f32 %1, %8, %1\n\t" \
      "v_add_f32 %2, %9, %2\n\t" \
      "v_add_f32 %1, %10, %1\n\t" \
      "v_add_f32 %2, %11, %2\n\t" \
      "v_add_f32 %1, %12, %1\n\t" \
      "v_add_f32 %2, %13, %2\n\t" \
      "v_add_f32 %1, %14, %1\n\t" \
      "v_add_f32 %2, %15, %2\n\t" \
      "v_add_f32 %1, %16, %1\n\t" \
      "v_add_f32 %2, %17, %2\n\t" \
      "v_add_f32 %1, %18, %1\n\t" \
      "v_add_f32 %2, %19, %2\n\t" \
      "v_add_f32 %1, %20, %1\n\t" \
      "v_add_f32 %2, %21, %2" \
      : "=&v"(DST), "+v"(fc[GC][0]), "+v"(fc[GC][1]) \
      : "v"(A[0]), "v"(bop[G]), \
        "v"(cin[G]), "v"(SRC[0]), "v"(SRC[1]), "v"(SRC[2]), "v"(SRC[3]), "v"(SRC[4]), "v"(SRC[5]), \
        "v"(SRC[6]), "v"(SRC[7]), "v"(SRC[8]), "v"(SRC[9]), "v"(SRC[10]), "v"(SRC[11]), \
        "v"(SRC[12]), "v"(SRC[13]), "v"(SRC[14]), "v"(SRC[15]));
#define PF_UNIT(DST, SRC, G, A, COUNT)            \
  if constexpr (NM == 2) {                        \
    PF_UNIT2(DST, SRC, G, A, COUNT)               \
  } else if constexpr (RTN) {                     \
  } else {                                        \
    PF_UNIT1(DST, SRC, G, A, COUNT)               \
  }
#define PF_LOAD(A, TILE)                                                                          \
  {                                                                                               \
    const _Float16* arow_ = reinterpret_cast<const _Float16*>(lds + buf * STAGE_BYTES) +          \
                            ((TILE) * 32 + col) * PITCH + 8 * half;                               \
    _Pragma("unroll") for (int m = 0; m < NM; ++m) A[m] =                                         \
        *reinterpret_cast<const f16x8*>(arow_ + 16 * m);                                          \
  }
  if (beg < end) issue_stage(0, beg);
  // f32 rounding toward -inf from here on (MODE[1:0]); nothing below depends on round-to-nearest: the MFMA results may
  // come out one ulp lower, which can only add to the count
  if constexpr (RTN) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2");
  const unsigned long long t_loop = trace ? wall_clock64() : 0ULL;
  unsigned long long t_wait_dma = 0, t_wait_bar = 0;
  int buf = 0;
  for (int base = beg; base < end; base += PF_ROWS) {
    // this wave's pieces of the stage have landed; after the barrier everybody's have, and everybody
    // has finished reading the other buffer, which the next copy overwrites
    const unsigned long long tw0 = trace ? wall_clock64() : 0ULL;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tw1 = trace ? wall_clock64() : 0ULL;
    __builtin_amdgcn_s_barrier();
    if (trace) {
      const unsigned long long tw2 = wall_clock64();
      t_wait_dma += tw1 - tw0;
      t_wait_bar += tw2 - tw1;
    }
    if (base + PF_ROWS < end) issue_stage(buf ^ 1, base + PF_ROWS);
    if constexpr (RTN) {
      if (wave_live) {
        // TWO accumulator sets: S0 always holds group 0, S1 group 1; unit k = (tile k / 2, group k % 2) writes its group's
        // set and adds the OTHER set = the results of unit k - 1 into that group's counters.  Between the MFMA of unit
        // k - 1 and the first read of its results lie its own 16 adds and the MFMA of unit k: 17 instructions, beyond
        // the 11 wait states an 8-pass MFMA needs.  (A third set as in the v_alignbit schedule costs 16 VGPRs: 142, three
        // waves per SIMD.)
        f16x8 aX[2], aY[2];
        PF_LOAD(aX, 0)
        PF_LOAD(aY, 1)
        PF_UNITR(S0, S1, 0, 1, aX)
        PF_UNITR(S1, S0, 1, 0, aX)
        PF_LOAD(aX, 2)
        PF_UNITR(S0, S1, 0, 1, aY)
        PF_UNITR(S1, S0, 1, 0, aY)
        PF_LOAD(aY, 3)
        PF_UNITR(S0, S1, 0, 1, aX)
        PF_UNITR(S1, S0, 1, 0, aX)
        PF_LOAD(aX, 4)
        PF_UNITR(S0, S1, 0, 1, aY)
        PF_UNITR(S1, S0, 1, 0, aY)
        PF_LOAD(aY, 5)
        PF_UNITR(S0, S1, 0, 1, aX)
        PF_UNITR(S1, S0, 1, 0, aX)
        PF_UNITR(S0, S1, 0, 1, aY)
        PF_UNITR(S1, S0, 1, 0, aY)
      }
    } else if (wave_live) {
      // unit k writes set k % 3 and extracts set (k + 1) % 3 = unit k-2 = (tile t-1, same group);
      // 32 fresh sign bits are counted whenever tile t-1 is odd
      f16x8 aX[2], aY[2];
      PF_LOAD(aX, 0)
      PF_LOAD(aY, 1)
      PF_UNIT(S0, S1, 0, aX, true)
      PF_UNIT(S1, S2, 1, aX, true)
      PF_LOAD(aX, 2)
      PF_UNIT(S2, S0, 0, aY, false)
      PF_UNIT(S0, S1, 1, aY, false)
      PF_LOAD(aY, 3)
      PF_UNIT(S1, S2, 0, aX, true)
      PF_UNIT(S2, S0, 1, aX, true)
      PF_LOAD(aX, 4)
      PF_UNIT(S0, S1, 0, aY, false)
      PF_UNIT(S1, S2, 1, aY, false)
      PF_LOAD(aY, 5)
      PF_UNIT(S2, S0, 0, aX, true)
      PF_UNIT(S0, S1, 1, aX, true)
      PF_UNIT(S1, S2, 0, aY, false)
      PF_UNIT(S2, S0, 1, aY, false)
    }
    buf ^= 1;
  }
#undef PF_UNIT
#undef PF_UNITR
#undef PF_UNIT1
#undef PF_UNIT2
#undef PF_LOAD
  if (trace && lane == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = trace + ((size_t)blockIdx.x * 4 + wave) * 4;
    o[0] = t_start;
    o[1] = (t_wait_dma << 32) | t_wait_bar;
    o[2] = wall_clock64();
    // shader-clock cycles of this wave's lifetime in the top bits (with o[2] - o[0] at 100 MHz: the
    // in-kernel clock), placement in the low bits
    o[3] = ((__builtin_amdgcn_s_memtime() - c_start) << 24) | ((unsigned long long)(xcc & 0xf) << 20) |
           (hwid & 0xfffff);
    (void)t_loop;
  }
  if (!wave_live || beg >= end) return;
  // drain: the asm blocks hide their MFMAs from the compiler's hazard recognizer
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S1), "+v"(S2));
  if constexpr (RTN) {
    // the last unit (group 1) is the only one not counted yet
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("v_add_f32 %0, %1, %0" : "+v"(fc[1][r & 1]) : "v"(S1[r]));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" ::: "memory");
#pragma unroll
    for (int g = 0; g < PF_NG; ++g)   // (C0 - fc) is count * 2^40 exactly: both operands are multiples of 2^40 below 2^64
      cnt[g] = (int)((RTN_C0 - fc[g][0]) * 0x1p-40f) + (int)((RTN_C0 - fc[g][1]) * 0x1p-40f);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) bits[0] = __builtin_amdgcn_alignbit(bits[0], __float_as_uint(S1[r]), 31);
    cnt[0] += __popc(bits[0]);
#pragma unroll
    for (int r = 0; r < 16; ++r) bits[1] = __builtin_amdgcn_alignbit(bits[1], __float_as_uint(S2[r]), 31);
    cnt[1] += __popc(bits[1]);
  }
#pragma unroll
  for (int g = 0; g < PF_NG; ++g) {
    const int c = cnt[g] + __shfl_xor(cnt[g], 32);
    const int h = h0 + 32 * g + col;
    if (half == 0 && h < bcount && h < ek_rel) {
      if (splits == 1)
        cnt_up[(int64_t)p * bmax + h] = c;
      else
        atomicAdd(&cnt_up[(int64_t)p * bmax + h], c);
    }
  }
}

constexpr int PF_S2_CAP = 1024;   // rows per problem of the second stage's compact list (= the largest list the few-survivor kernel takes)
// What the second stage needs of a survivor, written by k_ransac_survivors itself when the stage runs (s2.A16s != nullptr):
// A16s[p][slot] = A16[p][h], c_hs = c_h - beta (the K = 32 image is not centred by beta: (c - beta) in double is exact, and
// narrowing toward -inf never tightens the test; an unusable hypothesis -- c_h = -1, zero row -- stays negative: every row
// counts again), its counter cleared.
struct Stage2Rows {
  const _Float16* A16;
  const float* c_h;
  const unsigned* stat;
  double tcap;
  _Float16* A16s;
  float* c_hs;
  int32_t* cnt2;
  int32_t* n_surv2;
};
// Survivors: hypotheses whose upper bound reaches the carried best count.  The others get count 0.
__global__ void k_ransac_survivors(const RansacProb* __restrict__ probs, const int32_t* __restrict__ cnt_up,
                                   int it0, int bcount, int bmax, int32_t* __restrict__ res_cnt,
                                   unsigned long long* __restrict__ err_by_h,
                                   int32_t* __restrict__ hlist, int32_t* __restrict__ n_surv, const Stage2Rows s2) {
  const int p = blockIdx.y;
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h == 0 && s2.A16s) s2.n_surv2[p] = 0;
  if (h >= bcount) return;
  const RansacProb pr = probs[p];
  if (pr.done || it0 + h >= pr.est_k) return;
  res_cnt[(int64_t)p * bmax + h] = 0;
  if (cnt_up[(int64_t)p * bmax + h] >= pr.best_cnt) {
    const int slot = atomicAdd(&n_surv[p], 1);
    hlist[(int64_t)p * bmax + slot] = h;
    err_by_h[(int64_t)p * bmax + h] = 0;  // accumulated by k_ransac_count_few
    if (s2.A16s && slot < PF_S2_CAP) {
      const uint4* src = reinterpret_cast<const uint4*>(s2.A16 + ((int64_t)p * bmax + h) * PF_K);
      uint4* dst = reinterpret_cast<uint4*>(s2.A16s + ((int64_t)p * PF_S2_CAP + slot) * PF_K);
      dst[0] = src[0];
      dst[1] = src[1];
      const double smax = (double)__uint_as_float(s2.stat[p * PF_STAT]);
      const double beta = s2.tcap > 0.0 ? smax * smax : 0.0;   // the double pf_emit_row added
      s2.c_hs[(int64_t)p * PF_S2_CAP + slot] = __double2float_rd((double)s2.c_h[(int64_t)p * bmax + h] - beta);
      s2.cnt2[(int64_t)p * PF_S2_CAP + slot] = 0;
    }
  }
}

// ---- second stage (round 4): the K = 16 bound leaves 3 - 4x the survivors of the K = 32 bound; the survivors of a round --
// a compact list of ~16 hypotheses per problem -- go through the K = 32 form (a_hi . (b_hi + b_lo), same a_hi rows, same
// eps_h) before they are counted exactly.  1.5 % of the matrix work of a first-stage launch.
// survivors of the second stage: hlist2 = the entries of hlist whose K = 32 bound still reaches the carried best count
__global__ void k_ransac_stage2_survivors(const RansacProb* __restrict__ probs, const int32_t* __restrict__ hlist,
                                          const int32_t* __restrict__ n_surv, int bmax, const int32_t* __restrict__ cnt2,
                                          int32_t* __restrict__ hlist2, int32_t* __restrict__ n_surv2,
                                          unsigned long long* __restrict__ total2) {
  const int p = blockIdx.y;
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  const RansacProb pr = probs[p];
  if (pr.done) return;
  if (slot >= n_surv[p]) return;
  // entries beyond the capacity of the compact list were not looked at by the second stage: they pass unfiltered
  if (slot >= PF_S2_CAP || cnt2[(int64_t)p * PF_S2_CAP + slot] >= pr.best_cnt) {
    const int o = atomicAdd(&n_surv2[p], 1);
    hlist2[(int64_t)p * bmax + o] = hlist[(int64_t)p * bmax + slot];
    atomicAdd(total2, 1ULL);
  }
}

// Debug check (CS_RANSAC_CHECK=1): the bound must dominate the exact count of every hypothesis.
__global__ void k_ransac_check_bound(const RansacProb* __restrict__ probs, const int32_t* __restrict__ exact,
                                     const int32_t* __restrict__ cnt_up, int it0, int bcount, int bmax,
                                     unsigned long long* __restrict__ stats) {
  const int p = blockIdx.y;
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= bcount) return;
  const RansacProb pr = probs[p];
  if (pr.done || it0 + h >= pr.est_k) return;
  const int e = exact[(int64_t)p * bmax + h], u = cnt_up[(int64_t)p * bmax + h];
  if (e > u) atomicAdd(&stats[0], 1ULL);
  atomicAdd(&stats[1], 1ULL);
  atomicAdd(&stats[2], (unsigned long long)(u - e > 0 ? u - e : 0));
}

// Debug check of the SECOND stage (CS_RANSAC_CHECK=1): the K = 32 bound of every compacted survivor must dominate its exact
// count too (violations and comparisons go to the same counters as the first stage's).
__global__ void k_ransac_check_bound2(const RansacProb* __restrict__ probs, const int32_t* __restrict__ exact,
                                      const int32_t* __restrict__ hlist, const int32_t* __restrict__ n_surv, int bmax,
                                      const int32_t* __restrict__ cnt2, unsigned long long* __restrict__ stats) {
  const int p = blockIdx.y;
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (probs[p].done || slot >= min(n_surv[p], PF_S2_CAP)) return;
  const int e = exact[(int64_t)p * bmax + hlist[(int64_t)p * bmax + slot]], u = cnt2[(int64_t)p * PF_S2_CAP + slot];
  if (e > u) atomicAdd(&stats[0], 1ULL);
  atomicAdd(&stats[1], 1ULL);
  atomicAdd(&stats[2], (unsigned long long)(u - e > 0 ? u - e : 0));
}

// Exact counts (and fixed-point errors) when only a handful of hypotheses survive the prefilter (the
// normal case: ~2 per problem and round).  The MFMA list kernel above needs a 128-hypothesis tile per
// workgroup and costs ~110 us per round even for two survivors.  Here the pair range of a problem is
// split over gridDim.x workgroups, each walks its slice once per survivor with the canonical f64
// chain; integer partial sums are combined with atomics (exact, order-free).
// grid: x = pair slice, y = problem, z = survivor slot (strided).  res_cnt / err_by_h of the survivors
// are zero on entry.
__global__ __launch_bounds__(256) void k_ransac_count_few(const RansacProb* __restrict__ probs,
                                                          const float* __restrict__ pk, int64_t total,
                                                          const double* __restrict__ hyp, int bmax,
                                                          double thr2, double scale,
                                                          int32_t* __restrict__ res_cnt,
                                                          unsigned long long* __restrict__ err_by_h,
                                                          const int32_t* __restrict__ hlist,
                                                          const int32_t* __restrict__ n_surv, int list_stride,
                                                          // fused k_ransac_stage2_survivors (cnt2 != nullptr): entry c of the
                                                          // first-stage list is skipped when its second-stage bound is below the best
                                                          const int32_t* __restrict__ cnt2) {
  const int p = blockIdx.y;
  const RansacProb pr = probs[p];
  if (pr.done) return;
  const int nlist = n_surv[p];
  // entries beyond the capacity of the compact list were not looked at by the second stage: they pass unfiltered
  auto next_entry = [&](int c) {
    if (cnt2)
      while (c < nlist && c < PF_S2_CAP && cnt2[(int64_t)p * PF_S2_CAP + c] < pr.best_cnt) c += gridDim.z;
    return c;
  };
  int c = next_entry(blockIdx.z);
  if (c >= nlist) return;
  const int tid = threadIdx.x;
  const int per = (pr.m + gridDim.x - 1) / gridDim.x;
  const int i0 = blockIdx.x * per, i1 = min(pr.m, i0 + per);
  if (i0 >= i1) return;
  // this thread's pairs stay in registers across the survivors (slices are short: m / gridDim.x / 256)
  constexpr int MAXP = 8;
  const bool in_regs = per <= 256 * MAXP;
  float ps[MAXP][6];
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      const int i = i0 + tid + 256 * j;
      const int64_t g = pr.off + (i < i1 ? i : i0);
#pragma unroll
      for (int c = 0; c < 6; ++c) ps[j][c] = pk[(int64_t)c * total + g];
      if (i >= i1) ps[j][3] = 1.0e30f;  // far-away target: never an inlier
    }
  }
  // the next survivor's hypothesis is requested before the current one is evaluated: list entry -> twelve strided f64
  // loads are two dependent trips to L2 (~2 us), as long as the 8 x 22 f64 operations per thread they feed
  int hn = hlist[(int64_t)p * list_stride + c];
  double Rn[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) Rn[e] = hyp[((int64_t)p * 12 + e) * bmax + hn];
  for (; c < nlist;) {
    const int h = hn;
    double R[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) R[e] = Rn[e];
    c = next_entry(c + gridDim.z);
    if (c < nlist) {
      hn = hlist[(int64_t)p * list_stride + c];
#pragma unroll
      for (int e = 0; e < 12; ++e) Rn[e] = hyp[((int64_t)p * 12 + e) * bmax + hn];
    }
    int cnt = 0;
    unsigned long long err = 0;  // fixed-point squared error of the inliers (exact integer sum, as k_ransac_err)
    auto one = [&](float sx, float sy, float sz, float qx, float qy, float qz) {
      const double d2 = residual2_f64(R, (double)sx, (double)sy, (double)sz, (double)qx, (double)qy, (double)qz);
      if (d2 < thr2) {
        ++cnt;
        err += (unsigned long long)(d2 * scale);
      }
    };
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < MAXP; ++j) one(ps[j][0], ps[j][1], ps[j][2], ps[j][3], ps[j][4], ps[j][5]);
    } else {
      for (int i = i0 + tid; i < i1; i += 256) {
        const int64_t g = pr.off + i;
        one(pk[0 * total + g], pk[1 * total + g], pk[2 * total + g], pk[3 * total + g], pk[4 * total + g],
            pk[5 * total + g]);
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      cnt += __shfl_xor(cnt, off);
      err += __shfl_xor(err, off);
    }
    if ((tid & 63) == 0 && cnt) {
      atomicAdd(&res_cnt[(int64_t)p * bmax + h], cnt);
      atomicAdd(&err_by_h[(int64_t)p * bmax + h], err);
    }
  }
}

// est_k implied by a best inlier count c (Open3D: log(1 - confidence) / log(1 - ratio^n))
__device__ __forceinline__ int est_bound(int c, int m, int ransac_n, double log_1mc, int est_k0) {
  const double ratio = fmin(1.0, (double)c / (double)m);
  double pw = 1.0;
  for (int j = 0; j < ransac_n; ++j) pw = pw * ratio;
  const double den = log(1.0 - pw);
  if (den < 0.0) {  // den == 0: (inliers/M)^n below 2^-53 -> no finite bound
    const double est = log_1mc / den;
    if (est < (double)est_k0) return (int)ceil(est);
  }
  return est_k0;
}

// One wave per problem: sequential-semantics replay of the chunk from the inlier counts alone.
__global__ __launch_bounds__(256) void k_ransac_scan1(RansacProb* probs, int n_prob,
                                                     const int32_t* __restrict__ res_cnt, int it0,
                                                     int bcount, int bmax, int ransac_n,
                                                     int max_iter, double log_1mc,
                                                     int32_t* __restrict__ cand, int* n_active,
                                                     // fused k_ransac_scan2 (err_by_h != nullptr: k_ransac_count_few left the
                                                     // error of every survivor, so no k_ransac_err has to run in between)
                                                     const unsigned long long* __restrict__ err_by_h,
                                                     const double* __restrict__ hyp, int32_t* __restrict__ next_nsurv,
                                                     int* __restrict__ next_nactive) {
  // The chunk's counts are staged in LDS by all four waves (coalesced), then wave 0 replays them: lane l
  // owns the contiguous segment [l seg, (l+1) seg).  Element h of lane l sits at h + l, which spreads
  // the lanes' same-offset reads over the banks.
  extern __shared__ int32_t lcnt[];
  const int p = blockIdx.x;
  if (err_by_h && threadIdx.x == 0) {   // what k_ransac_scan2 clears for the next round (the other parity's counters)
    next_nsurv[p] = 0;
    if (p == 0) *next_nactive = 0;
  }
  RansacProb pr = probs[p];
  if (pr.done) return;
  const int seg = (bcount + 63) / 64;
  {
    const int32_t* g = res_cnt + (int64_t)p * bmax;
    for (int h = threadIdx.x; h < bcount; h += 256) lcnt[h + h / seg] = g[h];
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  const int32_t* cnt = lcnt + lane;  // cnt[h] for h in this lane's segment
  const int s0 = lane * seg, s1 = min(bcount, s0 + seg);
  // 1. exclusive prefix max of the counts over lanes, seeded with the carried best
  int lmax = 0;
  for (int h = s0; h < s1; ++h)
    if (it0 + h < pr.est_k) lmax = max(lmax, cnt[h]);
  int incl = lmax;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl = max(incl, o);
  }
  int pre = __shfl_up(incl, 1);
  if (lane == 0) pre = 0;
  pre = max(pre, pr.best_cnt);
  // 2. first position where the loop of the reference would stop
  int stop = 0x7fffffff;
  {
    int cur = pre;
    int ek = cur > pr.best_cnt ? est_bound(cur, pr.m, ransac_n, log_1mc, pr.est_k) : pr.est_k;
    for (int h = s0; h < s1; ++h) {
      if (it0 + h >= ek) {
        stop = h;
        break;
      }
      const int c = cnt[h];
      if (c > cur) {
        cur = c;
        ek = est_bound(cur, pr.m, ransac_n, log_1mc, pr.est_k);
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) stop = min(stop, __shfl_xor(stop, off));
  if (stop > bcount) stop = bcount;
  // 3. best count over the evaluated prefix, new bound
  int cmax = 0;
  for (int h = s0; h < min(s1, stop); ++h) cmax = max(cmax, cnt[h]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
  int new_ek = pr.est_k;
  if (cmax > pr.best_cnt) new_ek = est_bound(cmax, pr.m, ransac_n, log_1mc, pr.est_k);
  // 4. candidates: evaluated hypotheses that tie for the best count (ascending order)
  int ncand = 0;
  // fused scan2: the sequential rule "take candidate c if the count rose or its error is strictly below the best so far" ends
  // at the FIRST candidate (ascending hypothesis) of the smallest error, provided the count rose or that error is below
  // the carried one -- a lexicographic (error, hypothesis) minimum over the lanes' segments
  unsigned long long e_min = ~0ULL;
  int h_min = 0x7fffffff;
  if (cmax > 0 && cmax >= pr.best_cnt) {
    if (err_by_h) {
      for (int h = s0; h < min(s1, stop); ++h)
        if (cnt[h] == cmax) {
          const unsigned long long e = err_by_h[(int64_t)p * bmax + h];
          if (e < e_min) {   // strict: the earlier hypothesis keeps a tie
            e_min = e;
            h_min = h;
          }
        }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long eo = __shfl_xor(e_min, off);
        const int ho = __shfl_xor(h_min, off);
        if (eo < e_min || (eo == e_min && ho < h_min)) {
          e_min = eo;
          h_min = ho;
        }
      }
    } else {
      int mine = 0;
      for (int h = s0; h < min(s1, stop); ++h) mine += cnt[h] == cmax;
      int incl2 = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl2, off);
        if (lane >= off) incl2 += o;
      }
      int pos = incl2 - mine;
      ncand = __shfl(incl2, 63);
      for (int h = s0; h < min(s1, stop); ++h)
        if (cnt[h] == cmax) cand[(int64_t)p * bmax + pos++] = h;
    }
  }
  if (lane == 0) {
    const int consumed = it0 + stop;
    pr.est_k = new_ek;
    pr.n_cand = ncand;
    pr.chunk_max = cmax;
    if (consumed >= pr.est_k || consumed >= max_iter) {
      pr.done = 1;
      pr.iters = consumed < max_iter ? consumed : max_iter;
    } else {
      atomicAdd(n_active, 1);
    }
    if (err_by_h && h_min != 0x7fffffff && (cmax > pr.best_cnt || e_min < pr.best_err)) {
      pr.best_cnt = cmax;
      pr.best_err = e_min;
      pr.best_itr = it0 + h_min;
      for (int c = 0; c < 12; ++c) pr.best_T[c] = hyp[((int64_t)p * 12 + c) * bmax + h_min];
    }
    // (unfused: best_* are updated by scan2; keep everything else)
    probs[p] = pr;
  }
}

// Fixed-point squared error of the candidate hypotheses: grid (slots, problems).
__global__ __launch_bounds__(256) void k_ransac_err(const RansacProb* __restrict__ probs,
                                                    const float* __restrict__ pk, int64_t total,
                                                    const double* __restrict__ hyp, int bmax,
                                                    const int32_t* __restrict__ cand, double thr2,
                                                    double scale,
                                                    unsigned long long* __restrict__ cand_err) {
  __shared__ unsigned long long red[256];
  const int p = blockIdx.y;
  const RansacProb pr = probs[p];
  const int tid = threadIdx.x;
  for (int c = blockIdx.x; c < pr.n_cand; c += gridDim.x) {
    const int h = cand[(int64_t)p * bmax + c];
    const double* hp = hyp + ((int64_t)p * 12) * bmax + h;
    double R[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) R[e] = hp[(int64_t)e * bmax];
    unsigned long long err = 0;
    for (int i = tid; i < pr.m; i += 256) {
      const int64_t g = pr.off + i;
      const double d2 = residual2_f64(R, (double)pk[0 * total + g], (double)pk[1 * total + g],
                                      (double)pk[2 * total + g], (double)pk[3 * total + g],
                                      (double)pk[4 * total + g], (double)pk[5 * total + g]);
      if (d2 < thr2) err += (unsigned long long)(d2 * scale);
    }
    red[tid] = err;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    if (tid == 0) cand_err[(int64_t)p * bmax + c] = red[0];
    __syncthreads();
  }
}

// cand_err is indexed by candidate slot (k_ransac_err) or, when by_h is set, by hypothesis
// (k_ransac_count_few computed the error of every survivor along with its count)
__global__ void k_ransac_scan2(RansacProb* probs, int n_prob, const double* __restrict__ hyp,
                               const int32_t* __restrict__ cand,
                               const unsigned long long* __restrict__ cand_err, int by_h, int it0,
                               int bmax, int32_t* __restrict__ next_nsurv, int* __restrict__ next_nactive) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_prob) return;
  // last kernel of a round: clear the survivor / activity counters the NEXT round accumulates into (the other
  // parity's: this round's are still to be copied to the host) -- a hipMemsetAsync per round before
  next_nsurv[p] = 0;
  if (p == 0) *next_nactive = 0;
  RansacProb pr = probs[p];
  if (pr.n_cand <= 0) return;
  bool changed = false;
  int best_h = -1;
  for (int c = 0; c < pr.n_cand; ++c) {
    const unsigned long long e = cand_err[(int64_t)p * bmax + (by_h ? cand[(int64_t)p * bmax + c] : c)];
    if (pr.chunk_max > pr.best_cnt || e < pr.best_err) {
      pr.best_cnt = pr.chunk_max;
      pr.best_err = e;
      best_h = cand[(int64_t)p * bmax + c];
      changed = true;
    }
  }
  pr.n_cand = 0;
  if (changed) {
    pr.best_itr = it0 + best_h;
    for (int c = 0; c < 12; ++c) pr.best_T[c] = hyp[((int64_t)p * 12 + c) * bmax + best_h];
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
    for (int c = 0; c < 12; ++c) o[c] = (float)pr.best_T[c];  // the reference's T_est.astype(np.float32)
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

// page-locked host staging buffer of the calling thread (grown on demand, freed at thread exit)
namespace {
struct PinnedScratch {
  char* p = nullptr;
  size_t n = 0;
  ~PinnedScratch() {
    if (p) (void)hipHostFree(p);
  }
};
thread_local PinnedScratch t_pinned;
char* pinned_scratch(size_t bytes) {
  if (t_pinned.n < bytes) {
    if (t_pinned.p) (void)hipHostFree(t_pinned.p);
    t_pinned.p = nullptr;
    t_pinned.n = 0;
    void* q = nullptr;
    if (hipHostMalloc(&q, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    t_pinned.p = static_cast<char*>(q);
    t_pinned.n = bytes;
  }
  return t_pinned.p;
}
}  // namespace

// prefilter diagnostics: {bound violations, hypotheses checked, sum of (bound - exact)} from
// CS_RANSAC_CHECK runs, {survivors, hypotheses evaluated} always
static std::atomic<unsigned long long> g_pf_stats[5];  // zero-initialised (static storage)

extern "C" {

void cs_ransac_prefilter_stats(uint64_t out[5], int reset) {
  for (int i = 0; i < 5; ++i) {
    if (out) out[i] = g_pf_stats[i].load();
    if (reset) g_pf_stats[i].store(0);
  }
}

int cs_ransac_batch(const float* d_src, const float* d_tgt, const int64_t* h_off, int n_prob,
                    double max_corr, int ransac_n, int max_iter, double confidence, uint64_t seed,
                    float* d_T, int32_t* d_inliers, double* d_rmse, int32_t* d_iters,
                    void* stream) {
  CS_REQUIRE(h_off && d_T, CS_ERR_INVALID, "cs_ransac_batch: NULL argument");
  CS_REQUIRE(ransac_n >= 3 && ransac_n <= 64, CS_ERR_INVALID,
             "cs_ransac_batch: ransac_n %d not in [3, 64]", ransac_n);
  CS_REQUIRE(max_corr > 0.0 && max_iter >= 1, CS_ERR_INVALID,
             "cs_ransac_batch: need max_corr > 0 and max_iter >= 1");
  CS_REQUIRE(confidence > 0.0 && confidence <= 1.0, CS_ERR_INVALID,
             "cs_ransac_batch: confidence must be in (0, 1]");
  if (n_prob <= 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  const int64_t total = h_off[n_prob] - h_off[0];
  CS_REQUIRE(h_off[0] == 0 && total >= 0, CS_ERR_INVALID, "cs_ransac_batch: bad offsets");
  CS_REQUIRE(total == 0 || (d_src && d_tgt), CS_ERR_INVALID, "cs_ransac_batch: NULL correspondences");

  std::vector<RansacProb> hp(n_prob);
  int m_max = 0;
  for (int p = 0; p < n_prob; ++p) {
    int64_t m = h_off[p + 1] - h_off[p];
    CS_REQUIRE(m >= 0 && m < (1LL << 24), CS_ERR_INVALID,
               "cs_ransac_batch: bad segment %d (a problem holds fewer than 2^24 correspondences)", p);
    RansacProb& pr = hp[p];
    memset(&pr, 0, sizeof(pr));
    pr.off = h_off[p];
    pr.m = (int32_t)m;
    pr.est_k = max_iter;
    pr.best_itr = -1;
    // Open3D returns the default (identity) result when there are fewer pairs than ransac_n
    pr.done = m < ransac_n ? 1 : 0;
    if (m > m_max) m_max = (int)m;
  }
  // Largest chunk of iterations per round.  One workgroup = 128 hypotheses x all pairs of a problem
  // (~100 us), 1536 workgroups are resident: chunks of 16384 give >= 8 "waves" of workgroups for a
  // 32-query batch, so the partially filled last wave costs ~10 % instead of ~33 % at 4096.
  const int bmax = 16384;
  const int64_t tot1 = total ? total : 1;
  // f16 prefilter (see k_ransac_prefilter): on from the third chunk, when every live problem normally
  // carries a best count; CS_RANSAC_PREFILTER=0 disables it, CS_RANSAC_CHECK=1 verifies the bound
  // against the exact count of EVERY hypothesis (slow; tests).
  const char* env_pf = getenv("CS_RANSAC_PREFILTER");
  const char* env_ck = getenv("CS_RANSAC_CHECK");
  const bool use_pf = !(env_pf && env_pf[0] == '0') && total > 0;
  // CS_RANSAC_PF_K = 32: the two-MFMA form a_hi . (b_hi + b_lo); 16 (default, round 4): one MFMA, a_hi . b_hi with the
  // dropped term bounded per pair (k_ransac_pack16_b0).  Same results either way (the exact kernels decide).
  const int pf_nm = (getenv("CS_RANSAC_PF_K") && atoi(getenv("CS_RANSAC_PF_K")) == 32) ? 2 : 1;
  // K = 16: largest |t| / smax a hypothesis may have to go through the prefilter (the others are counted exactly)
  double pf_tcap = getenv("CS_RANSAC_PF_TCAP") ? atof(getenv("CS_RANSAC_PF_TCAP")) : 0.75;
  if (!(pf_tcap > 0.01 && pf_tcap <= 2.002)) pf_tcap = 2.002;
  if (pf_nm == 2) pf_tcap = 0.0;
  // CS_RANSAC_PF_COUNT=alignbit: the sign history of rounds 1-3 instead of the add-based count (K = 16 only)
  const bool pf_rtn = pf_nm == 1 && !(getenv("CS_RANSAC_PF_COUNT") && getenv("CS_RANSAC_PF_COUNT")[0] == 'a');
  const bool check = use_pf && env_ck && env_ck[0] == '1';
  // first chunk (all hypotheses counted exactly: there is no best count to prune against yet) and first
  // prefiltered iteration.  Round 1 (f32 matrix-pipe exact kernel): 512 = 256, no difference; with the f64
  // exact kernel of round 2 the unfiltered rounds are the expensive ones: 256 instead of 512 is +2.4 % queries/s
  // (experiment knobs: clamped -- a first chunk above bmax would index the bmax-sized scratch out of range, 0 would
  // never advance the chunk loop)
  // Round 5: 64.  The unfiltered chunk costs its hypotheses x ALL pairs in f64 (0.31 ms per 48-problem call at 256, twice per
  // chair step); with 64 the second chunk, [64, 512), is already prefiltered -- against the best of 64 hypotheses instead of 256,
  // which lets a few more of its hypotheses through to the exact kernels.  Results are the sequential loop's either way.
  int first_chunk = getenv("CS_RANSAC_FIRST") ? atoi(getenv("CS_RANSAC_FIRST")) : 64;
  first_chunk = ((std::min(std::max(first_chunk, 64), bmax) + 63) / 64) * 64;
  int pf_from = getenv("CS_RANSAC_PF_FROM") ? atoi(getenv("CS_RANSAC_PF_FROM")) : first_chunk;
  pf_from = std::max(pf_from, first_chunk);
  // per-round state in ONE block, so a round ends with one device->host copy (into pinned memory):
  // [RansacProb x n_prob | 2 x { n_surv int32 x n_prob (padded to 8 B) | n_active int32 (8 B) }]: the counters
  // exist once per round parity, k_ransac_scan2 clears the set of the next round
  const size_t st_probs = sizeof(RansacProb) * (size_t)n_prob;
  const size_t st_surv = ((sizeof(int32_t) * (size_t)n_prob + 7) / 8) * 8;
  const size_t st_cnt = st_surv + 8;
  const size_t st_bytes = st_probs + 2 * st_cnt;
  PoolBuf<char> state(st_bytes);
  RansacProb* const d_probs = reinterpret_cast<RansacProb*>(state.p);
  auto nsurv_of = [&](int par) { return reinterpret_cast<int32_t*>(state.p + st_probs + par * st_cnt); };
  auto nactive_of = [&](int par) { return reinterpret_cast<int*>(state.p + st_probs + par * st_cnt + st_surv); };
  char* const h_state = pinned_scratch(st_bytes);
  CS_REQUIRE(state.p && h_state, CS_ERR_HIP, "cs_ransac_batch: scratch allocation failed");
  PoolBuf<float> pk((size_t)tot1 * 6);
  PoolBuf<float4> pair32((size_t)tot1 * 2);
  PoolBuf<double> hyp((size_t)2 * n_prob * 12 * bmax);  // hypotheses (f64 R|t), one set per round parity
  PoolBuf<int32_t> res_cnt((size_t)n_prob * bmax), cand((size_t)n_prob * bmax);
  PoolBuf<unsigned long long> cand_err((size_t)n_prob * bmax);
  CS_REQUIRE(pk.p && pair32.p && hyp.p && res_cnt.p && cand.p && cand_err.p,
             CS_ERR_HIP, "cs_ransac_batch: scratch allocation failed");
  const bool pf_alloc = use_pf && max_iter > pf_from;
  // f16 pair image: every problem padded to whole LDS stages
  std::vector<int64_t> h_off16(n_prob + 1, 0);
  for (int p = 0; p < n_prob; ++p) h_off16[p + 1] = h_off16[p] + pf_padded(hp[p].m);
  const int64_t rows16 = h_off16[n_prob] ? h_off16[n_prob] : 1;
  PoolBuf<_Float16> B16(pf_alloc ? (size_t)rows16 * pf_pitch(pf_nm) : 8), A16(pf_alloc ? (size_t)2 * n_prob * bmax * PF_K : 8);
  PoolBuf<int64_t> off16(n_prob + 1);
  PoolBuf<float> c_h(pf_alloc ? (size_t)2 * n_prob * bmax : 1);
  PoolBuf<int32_t> cnt_up(pf_alloc ? (size_t)2 * n_prob * bmax : 1), hlist(pf_alloc ? (size_t)n_prob * bmax : 1);
  PoolBuf<unsigned> pf_stat((size_t)n_prob * PF_STAT);
  PoolBuf<double> pf_sums((size_t)n_prob * 6);
  // second stage (K = 32 on the compacted survivors of the K = 16 stage; CS_RANSAC_STAGE2=0 switches it off)
  const bool stage2 = pf_alloc && pf_nm == 1 && !(getenv("CS_RANSAC_STAGE2") && getenv("CS_RANSAC_STAGE2")[0] == '0');
  PoolBuf<_Float16> B32(stage2 ? (size_t)rows16 * PF_PITCH : 8), A16s(stage2 ? (size_t)n_prob * PF_S2_CAP * PF_K : 8);
  PoolBuf<float> c_hs(stage2 ? (size_t)n_prob * PF_S2_CAP : 1);
  PoolBuf<int32_t> cnt2(stage2 ? (size_t)n_prob * PF_S2_CAP : 1), hlist2(stage2 ? (size_t)n_prob * bmax : 1),
      n_surv2((size_t)n_prob);
  PoolBuf<unsigned> pf_stat2((size_t)n_prob * PF_STAT);   // (the K = 32 pack writes the same statistics again: scratch)
  PoolBuf<unsigned long long> s2_total(1);
  CS_REQUIRE(B32.p && A16s.p && c_hs.p && cnt2.p && hlist2.p && n_surv2.p && pf_stat2.p && s2_total.p, CS_ERR_HIP,
             "cs_ransac_batch: scratch allocation failed");
  PoolBuf<int32_t> exact_dbg(check ? (size_t)n_prob * bmax : 1);
  PoolBuf<unsigned long long> chk_stats(4);
  CS_REQUIRE(off16.p && B16.p && A16.p && c_h.p && cnt_up.p && hlist.p && pf_stat.p && pf_sums.p &&
                 exact_dbg.p && chk_stats.p,
             CS_ERR_HIP, "cs_ransac_batch: scratch allocation failed");
  std::vector<int32_t> h_surv(n_prob), h_xcd[2];
  // placement tables, one per round parity (8 x n_prob entries each)
  PoolBuf<int32_t> xcd_buf((size_t)16 * n_prob);
  CS_REQUIRE(xcd_buf.p, CS_ERR_HIP, "cs_ransac_batch: scratch allocation failed");
  {
    // problems + zeroed counters in one upload (h_state is page-locked and not read before the first round ends)
    memset(h_state, 0, st_bytes);
    memcpy(h_state, hp.data(), st_probs);
    CS_HIP_CHECK(hipMemcpyAsync(state.p, h_state, st_bytes, hipMemcpyHostToDevice, s));
  }
  if (total > 0) {
    hipLaunchKernelGGL(k_ransac_pack, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s,
                       d_src, d_tgt, total, pk.p, pair32.p);
    CS_LAUNCH_CHECK();
  }
  if (pf_alloc) {
    CS_HIP_CHECK(hipMemsetAsync(pf_stat.p, 0, sizeof(unsigned) * n_prob * PF_STAT, s));
    CS_HIP_CHECK(hipMemsetAsync(chk_stats.p, 0, sizeof(unsigned long long) * 4, s));
    int pblocks = (int)ceil_div(m_max > 0 ? m_max : 1, 256);
    if (pblocks > 64) pblocks = 64;
    CS_HIP_CHECK(hipMemcpyAsync(off16.p, h_off16.data(), sizeof(int64_t) * (n_prob + 1),
                                hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_ransac_pair_sums, dim3((unsigned)n_prob), dim3(256), 0, s, d_probs, d_src, d_tgt, pf_sums.p);
    if (pf_nm == 2) {
      hipLaunchKernelGGL(k_ransac_pack16<2>, dim3((unsigned)pblocks, (unsigned)n_prob), dim3(256), 0, s,
                         d_probs, off16.p, d_src, d_tgt, pf_sums.p, B16.p, pf_stat.p);
    } else {
      hipLaunchKernelGGL(k_ransac_pack16<1>, dim3((unsigned)pblocks, (unsigned)n_prob), dim3(256), 0, s,
                         d_probs, off16.p, d_src, d_tgt, pf_sums.p, B16.p, pf_stat.p);
      hipLaunchKernelGGL(k_ransac_pack16_b0, dim3((unsigned)pblocks, (unsigned)n_prob), dim3(256), 0, s,
                         d_probs, off16.p, d_src, d_tgt, pf_sums.p, pf_stat.p, pf_tcap, B16.p);
    }
    if (stage2) {
      CS_HIP_CHECK(hipMemsetAsync(pf_stat2.p, 0, sizeof(unsigned) * n_prob * PF_STAT, s));
      CS_HIP_CHECK(hipMemsetAsync(s2_total.p, 0, sizeof(unsigned long long), s));
      hipLaunchKernelGGL(k_ransac_pack16<2>, dim3((unsigned)pblocks, (unsigned)n_prob), dim3(256), 0, s,
                         d_probs, off16.p, d_src, d_tgt, pf_sums.p, B32.p, pf_stat2.p);
    }
    CS_LAUNCH_CHECK();
  }
  // squared threshold (Open3D: max_correspondence_distance * max_correspondence_distance in double) and
  // the power-of-two fixed-point scale of the inlier error: terms d^2 * scale < 2^38, so the u64 sum over
  // a problem's (< 2^24) pairs cannot overflow and is exact in any order
  const double thr2 = max_corr * max_corr;
  int ex = 0;
  (void)frexp(thr2, &ex);
  const double scale = ldexp(1.0, 38 - ex);
  const double log_1mc = log(1.0 - confidence);  // -inf when confidence == 1: never exits early
  unsigned long long tot_surv = 0, tot_eval = 0;
  const int trace_it0 = getenv("CS_PF_TRACE") ? atoi(getenv("CS_PF_TRACE")) : -1;
  PoolBuf<unsigned long long> trace(trace_it0 >= 0 ? (size_t)8 * n_prob * 64 * 16 * 16 : 1);
  size_t trace_n = 0;
  if (trace_it0 >= 0) CS_HIP_CHECK(hipMemsetAsync(trace.p, 0, sizeof(unsigned long long) * (size_t)8 * n_prob * 64 * 16 * 16, s));

  struct Event {
    hipEvent_t e = nullptr;
    ~Event() {
      if (e) (void)hipEventDestroy(e);
    }
  } round_done, front_done[2], hyp_done[2];
  CS_HIP_CHECK(hipEventCreateWithFlags(&round_done.e, hipEventDisableTiming));
  CS_HIP_CHECK(hipEventCreateWithFlags(&front_done[0].e, hipEventDisableTiming));
  CS_HIP_CHECK(hipEventCreateWithFlags(&front_done[1].e, hipEventDisableTiming));
  CS_HIP_CHECK(hipEventCreateWithFlags(&hyp_done[0].e, hipEventDisableTiming));
  CS_HIP_CHECK(hipEventCreateWithFlags(&hyp_done[1].e, hipEventDisableTiming));
  // A round = front half (hypotheses, f16 rows, prefilter: independent of the carried best) + back
  // half (survivors, exact counts, scans: the sequential RANSAC state).  The front half of round i+1
  // is enqueued on a second, low-priority stream before the host waits for round i, so it fills the
  // GPU while the back half of round i (many small dependent kernels) and the host turnaround run.
  // Its skip tests read est_k / done while they may be updated (prob_view); its placement table is
  // built from the state one round earlier: finished problems cost a few empty workgroups.
  // CS_RANSAC_JACOBI=1: every hypothesis through the Jacobi eigen-solver (the fallback of horn_qcp; the oracle's
  // oc_rigid_fit_force_jacobi is its counterpart) -- tests and A/B timing
  const int force_jacobi = getenv("CS_RANSAC_JACOBI") && atoi(getenv("CS_RANSAC_JACOBI")) != 0;
  const char* env_ov = getenv("CS_RANSAC_OVERLAP");
  hipStream_t side = (env_ov && env_ov[0] == '0') ? nullptr : side_stream();
  // ... and the hypotheses of a side-stream front half go to a THIRD stream: those of round i+2 are enqueued when round i
  // is known (their buffers, one set per parity, are free then) and run UNDER the prefilter of round i+1 instead of
  // behind it on the same stream (f64 vector work beside f16 matrix work).  CS_RANSAC_HYP_STREAM=0: one side stream.
  const char* env_hs = getenv("CS_RANSAC_HYP_STREAM");
  hipStream_t side_hyp = (side && !(env_hs && env_hs[0] == '0')) ? side_stream(1) : nullptr;
  // an error return must not hand the scratch buffers back to the pool while the side stream still uses
  // them; on the normal path the final wait on the main stream is already ordered behind its events
  struct SideDrain {
    hipStream_t st;
    bool clean = false;
    ~SideDrain() {
      if (st && !clean) (void)hipStreamSynchronize(st);
    }
  } side_drain{side}, side_hyp_drain{side_hyp};
  struct Front {
    int it0 = 0, b = 0, par = 0;
    bool pf = false, on_side = false;
    // placement of the round's prefilter launch (the second stage of the back half uses the same)
    XcdTab xtab;
    const int32_t* xcd_prob = nullptr;
    int pslots = 1;
  };
  // chunks: [0, first) counted exactly, [first, 512) in one piece, then doubling ([512, 1024), [1024, 2048), ...) up to bmax
  auto chunk_of = [&](int it0) {
    int b = it0 < first_chunk ? first_chunk : it0 < 512 ? 512 - it0 : (it0 < bmax ? it0 : bmax);
    return b > max_iter - it0 ? max_iter - it0 : b;
  };
  auto enqueue_front = [&](int it0, int par, hipStream_t st) -> Front {
    Front f;
    f.it0 = it0;
    f.b = chunk_of(it0);
    f.par = par;
    f.pf = pf_alloc && it0 >= pf_from;
    f.on_side = st != s;
    const int b = f.b;
    // deal the live problems to the 8 XCDs, longest first onto the least loaded XCD
    std::vector<int> order;
    for (int p = 0; p < n_prob; ++p)
      if (!hp[p].done && hp[p].est_k > it0) order.push_back(p);
    const int live = (int)order.size();
    std::sort(order.begin(), order.end(), [&](int a, int c) { return hp[a].m > hp[c].m; });
    std::vector<std::vector<int>> lists(8);
    int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p : order) {
      int best = 0;
      for (int x = 1; x < 8; ++x)
        if (load[x] < load[best]) best = x;
      lists[best].push_back(p);
      load[best] += pf_padded(hp[p].m);
    }
    int pslots = 1;
    for (int x = 0; x < 8; ++x) pslots = std::max(pslots, (int)lists[x].size());
    std::vector<int32_t>& tab = h_xcd[par];
    tab.assign((size_t)8 * pslots, -1);
    for (int x = 0; x < 8; ++x)
      for (size_t i = 0; i < lists[x].size(); ++i) tab[(size_t)x * pslots + i] = lists[x][i];
    int32_t* xcd_prob = nullptr;
    XcdTab xtab;
    // stream of the hypothesis kernels: the third stream for a prefiltered front half on the side stream
    hipStream_t sh = (f.on_side && f.pf && side_hyp) ? side_hyp : st;
    if (pslots <= XCD_SLOTS) {
      memcpy(xtab.v, tab.data(), sizeof(int32_t) * 8 * pslots);
    } else {
      xcd_prob = xcd_buf.p + (size_t)par * 8 * n_prob;
      (void)hipMemcpyAsync(xcd_prob, tab.data(), sizeof(int32_t) * 8 * pslots, hipMemcpyHostToDevice, sh);
    }
    double* hyp_r = hyp.p + (size_t)par * n_prob * 12 * bmax;
    _Float16* A16_r = f.pf ? A16.p + (size_t)par * n_prob * bmax * PF_K : nullptr;
    float* c_h_r = f.pf ? c_h.p + (size_t)par * n_prob * bmax : nullptr;
    int32_t* cnt_up_r = f.pf ? cnt_up.p + (size_t)par * n_prob * bmax : nullptr;
    const int ptiles = (b + PF_HYP - 1) / PF_HYP;
    // 1024 workgroups are resident (4 per CU): split the pair range until there are >= 8 rounds of
    // workgroups, as long as a workgroup keeps >= 8 stages
    int psplits = (int)((8 * 1024 + (int64_t)live * ptiles - 1) / std::max<int64_t>((int64_t)live * ptiles, 1));
    if (psplits < 1) psplits = 1;
    if (psplits > 16) psplits = 16;
    while (psplits > 1 && m_max / psplits < 8 * PF_ROWS) --psplits;
    // the prefilter rows of the hypotheses come out of the hypothesis kernel itself (CS_RANSAC_FUSE_HYP16=0: a second
    // kernel reads the table back, the arrangement of rounds 1 - 3)
    static const bool fuse16 = !(getenv("CS_RANSAC_FUSE_HYP16") && getenv("CS_RANSAC_FUSE_HYP16")[0] == '0');
    const bool fused = f.pf && fuse16;
    {
      ProfScope prof("ransac_hyp", sh);
      const int htiles = (b + 255) / 256;
      _Float16* fa = fused ? A16_r : nullptr;
      int32_t* fz = fused && psplits > 1 ? cnt_up_r : nullptr;
      if (ransac_n == 10)
        hipLaunchKernelGGL(k_ransac_hyp<10>, dim3((unsigned)(8 * pslots * htiles)), dim3(256), 0, sh, d_probs,
                           pair32.p, it0, b, bmax, ransac_n, seed, xcd_prob, xtab, pslots, htiles, force_jacobi, hyp_r,
                           pf_stat.p, pf_sums.p, thr2, pf_tcap, fa, c_h_r, fz);
      else
        hipLaunchKernelGGL(k_ransac_hyp<0>, dim3((unsigned)(8 * pslots * htiles)), dim3(256), 0, sh, d_probs,
                           pair32.p, it0, b, bmax, ransac_n, seed, xcd_prob, xtab, pslots, htiles, force_jacobi, hyp_r,
                           pf_stat.p, pf_sums.p, thr2, pf_tcap, fa, c_h_r, fz);
    }
    if (f.pf) {
      if (!fused)
        hipLaunchKernelGGL(k_ransac_hyp16, dim3((unsigned)((b + 255) / 256), (unsigned)n_prob), dim3(256),
                           0, sh, d_probs, hyp_r, pf_stat.p, pf_sums.p, it0, b, bmax, thr2, A16_r, c_h_r,
                           psplits > 1 ? cnt_up_r : (int32_t*)nullptr, pf_tcap);
      if (sh != st) {   // the prefilter (side stream) follows the hypotheses (third stream)
        (void)hipEventRecord(hyp_done[par].e, sh);
        (void)hipStreamWaitEvent(st, hyp_done[par].e, 0);
      }
      {
        ProfScope prof("ransac_pre", st);  // work units are added by the back half (state known there)
        const unsigned nblk = (unsigned)(8 * pslots * ptiles * psplits);
        if (pf_nm == 2)
          hipLaunchKernelGGL((k_ransac_prefilter<2, false>), dim3(nblk), dim3(256), 0, st, d_probs, off16.p, B16.p,
                             A16_r, c_h_r, it0, b, bmax, psplits, xcd_prob, xtab, pslots, ptiles, cnt_up_r,
                             (trace_it0 == it0) ? trace.p : nullptr, (const int32_t*)nullptr);
        else if (pf_rtn)
          hipLaunchKernelGGL((k_ransac_prefilter<1, true>), dim3(nblk), dim3(256), 0, st, d_probs, off16.p, B16.p,
                             A16_r, c_h_r, it0, b, bmax, psplits, xcd_prob, xtab, pslots, ptiles, cnt_up_r,
                             (trace_it0 == it0) ? trace.p : nullptr, (const int32_t*)nullptr);
        else
          hipLaunchKernelGGL((k_ransac_prefilter<1, false>), dim3(nblk), dim3(256), 0, st, d_probs, off16.p, B16.p,
                             A16_r, c_h_r, it0, b, bmax, psplits, xcd_prob, xtab, pslots, ptiles, cnt_up_r,
                             (trace_it0 == it0) ? trace.p : nullptr, (const int32_t*)nullptr);
        if (trace_it0 == it0) trace_n = (size_t)nblk * 16;
      }
    }
    f.xtab = xtab;
    f.xcd_prob = xcd_prob;
    f.pslots = pslots;
    if (f.on_side) (void)hipEventRecord(front_done[par].e, st);
    return f;
  };

  int max_surv_prev = 1 << 30;  // survivors per problem in the previous prefiltered round (unknown: many)
  Front cur = enqueue_front(0, 0, s);
  CS_LAUNCH_CHECK();
  bool side_pending = false;  // the side stream holds work the main stream has not waited for
  int side_par = 0;
  while (true) {
    const int it0 = cur.it0, b = cur.b;
    const bool pf = cur.pf;
    const double* hyp_r = hyp.p + (size_t)cur.par * n_prob * 12 * bmax;
    const int32_t* cnt_up_r = cnt_up.p + (size_t)cur.par * n_prob * bmax;
    if (cur.on_side) {
      CS_HIP_CHECK(hipStreamWaitEvent(s, front_done[cur.par].e, 0));
      side_pending = false;
    }
    bool err_known = false;  // the fixed-point errors of all candidates are already in cand_err (by hypothesis)
    const int hpw = (!pf && b <= 64) ? 64 : RC_HYP;      // hypotheses per workgroup of the unfiltered exact count
    const int tiles = (b + hpw - 1) / hpw;
    // enough workgroups for 256 CUs x several waves; the correspondence range is split when the
    // chunk is small (integer partial sums combine exactly)
    int splits = (int)(4096 / ((int64_t)n_prob * tiles > 0 ? (int64_t)n_prob * tiles : 1));
    if (splits < 1) splits = 1;
    if (splits > 16) splits = 16;
    while (splits > 1 && m_max / splits < 4 * RC_CHUNK) --splits;
    int32_t* const d_nsurv = nsurv_of(cur.par);   // cleared by the previous round's k_ransac_scan2 (or the upload)
    int* const d_nactive = nactive_of(cur.par);
    // algorithmic work of this chunk (hp is the state before it): 30 FLOP per (evaluated hypothesis,
    // correspondence) (transform 18 + squared distance 8 + compare/accumulate) -- for the
    // exact count and for the prefilter alike (its matrix pipe executes 64 per pair: the 32
    // multiply-adds of the a_hi (b_hi + b_lo) expansion)
    double eval_pairs = 0.0;
    for (int p = 0; p < n_prob; ++p) {
      if (hp[p].done) continue;
      int nh = hp[p].est_k - it0;
      if (nh > b) nh = b;
      if (nh > 0) {
        eval_pairs += (double)nh * (double)hp[p].m;
        tot_eval += (unsigned long long)nh;
      }
    }
    if (!pf) {
      if (splits > 1 || hpw != RC_HYP)  // partial counts (pair-range splits, lane quarters) are combined with integer atomics
        CS_HIP_CHECK(hipMemset2DAsync(res_cnt.p, sizeof(int32_t) * bmax, 0, sizeof(int32_t) * b,
                                      n_prob, s));
      ProfScope prof("ransac_eval", s, 30.0 * eval_pairs);
      if (hpw == 64)
        hipLaunchKernelGGL((k_ransac_count<false, 64>), dim3((unsigned)(tiles * splits), (unsigned)n_prob),
                           dim3(256), 0, s, d_probs, pk.p, tot1, hyp_r, it0, b, bmax, splits, thr2,
                           res_cnt.p, (const int32_t*)nullptr, (const int32_t*)nullptr);
      else
        hipLaunchKernelGGL(k_ransac_count<false>, dim3((unsigned)(tiles * splits), (unsigned)n_prob),
                           dim3(256), 0, s, d_probs, pk.p, tot1, hyp_r, it0, b, bmax, splits, thr2,
                           res_cnt.p, (const int32_t*)nullptr, (const int32_t*)nullptr);
    } else {
      // SURVEY 8d unit: 30 FLOP per (hypothesis, pair) -- the work of the exact formulation the prefilter
      // stands in for (the matrix pipe executes 64 FLOP per pair: bench.py reports both)
      prof_add_units("ransac_pre", 30.0 * eval_pairs);
      // the second stage runs when the few-survivor kernel will (decided from the previous round's counts, like the kernel
      // choice below); its rows are written by the survivors kernel itself
      static const int few_max = getenv("CS_RANSAC_FEW_MAX") ? std::max(atoi(getenv("CS_RANSAC_FEW_MAX")), 1) : 1024;
      const int surv_cap = std::min(max_surv_prev, b);
      const bool few = max_surv_prev <= few_max || b <= few_max;
      const bool run_s2 = stage2 && few && surv_cap <= PF_S2_CAP;
      Stage2Rows s2rows{};
      if (run_s2) {
        s2rows.A16 = A16.p + (size_t)cur.par * n_prob * bmax * PF_K;
        s2rows.c_h = c_h.p + (size_t)cur.par * n_prob * bmax;
        s2rows.stat = pf_stat.p;
        s2rows.tcap = pf_tcap;
        s2rows.A16s = A16s.p;
        s2rows.c_hs = c_hs.p;
        s2rows.cnt2 = cnt2.p;
        s2rows.n_surv2 = n_surv2.p;
      }
      hipLaunchKernelGGL(k_ransac_survivors, dim3((unsigned)((b + 255) / 256), (unsigned)n_prob),
                         dim3(256), 0, s, d_probs, cnt_up_r, it0, b, bmax, res_cnt.p, cand_err.p, hlist.p,
                         d_nsurv, s2rows);
      // exact counts of the survivors; few hypotheses, so the pair range is split finely
      int lsplits = 16;
      while (lsplits > 1 && m_max / lsplits < RC_CHUNK) --lsplits;
      {
        ProfScope prof("ransac_eval", s);
        // the previous round's survivor counts pick the kernel (both are exact for any count)
        // (round 3, K = 32 prefilter: 32 / 64 / 128 / 256 measured, 128 the fastest by ~1 %.  The K = 16 form of round 4
        // leaves 3 - 4x the survivors, ~16 per problem and round on the chair shape: the list kernel -- one workgroup of
        // 256 hypothesis lanes per problem -- then ran in every fourth round at 670 us)
        // (the first prefiltered round has no previous count, but a chunk of b <= few_max hypotheses cannot leave more)
        if (few) {
          // pair slices short enough for a thread to keep its pairs in registers across the survivors (8 per thread)
          int fslices = 8;
          while (fslices < 32 && m_max > fslices * 2048) fslices *= 2;
          // survivor slots per (slice, problem): a workgroup loads its pairs once and walks its share of the survivors, so
          // FEW slots amortise the load (chair, same box: 2 / 4 / 8 / 16 slots -> 1 431 / 1 454 / 1 424 / 1 370 queries/s)
          int fslots = surv_cap <= 256 ? 4 : 8;
          if (getenv("CS_RANSAC_FEW_SLOTS")) fslots = std::max(1, atoi(getenv("CS_RANSAC_FEW_SLOTS")));
          if (getenv("CS_RANSAC_FEW_SLICES")) fslices = std::max(1, atoi(getenv("CS_RANSAC_FEW_SLICES")));
          const int32_t* list_p = hlist.p;
          const int32_t* list_n = d_nsurv;
          const int32_t* list_cnt2 = nullptr;
          int list_stride = bmax;
          const int s2_pslots = cur.pslots;
          const int32_t* s2_xcd_prob = cur.xcd_prob;
          const XcdTab& s2_xtab = cur.xtab;
          // CS_RANSAC_KNOCKOUT (TIMING EXPERIMENT ONLY, results are wrong): bit 0 skips the exact count of the survivors, bit 1
          // the second-stage launch -- what a perfect bound in front of them could save at most (profiles/r5f_*)
          static const int knockout = getenv("CS_RANSAC_KNOCKOUT") ? atoi(getenv("CS_RANSAC_KNOCKOUT")) : 0;
          if (run_s2 && !(knockout & 2)) {
            // K = 32 bound of the survivors (rows compacted by k_ransac_survivors): one small prefilter launch over all
            // pairs, then the list filtered again
            // tiles for the WHOLE capacity: this round's survivor counts are not known on the host (surv_cap comes from the
            // previous round and only picks kernels that are exact for any count); workgroups past a problem's list leave at once
            const int s2tiles = std::max(1, (std::min(b, PF_S2_CAP) + PF_HYP - 1) / PF_HYP);
            int s2splits = 8;
            while (s2splits > 1 && m_max / s2splits < 8 * PF_ROWS) --s2splits;
            hipLaunchKernelGGL((k_ransac_prefilter<2, false>), dim3((unsigned)(8 * s2_pslots * s2tiles * s2splits)), dim3(256), 0,
                               s, d_probs, off16.p, B32.p, A16s.p, c_hs.p, 0, PF_S2_CAP, PF_S2_CAP, s2splits, s2_xcd_prob,
                               s2_xtab, s2_pslots, s2tiles, cnt2.p, (unsigned long long*)nullptr, d_nsurv);
            // the list filter of the second stage runs inside k_ransac_count_few (it skips the entries whose K = 32 bound is
            // below the best); the separate compaction kernel only for the statistics / CS_RANSAC_FUSE_S2LIST=0
            static const bool fuse_s2 = !(getenv("CS_RANSAC_FUSE_S2LIST") && getenv("CS_RANSAC_FUSE_S2LIST")[0] == '0') &&
                                        !getenv("CS_RANSAC_STAGE2_STATS");
            if (fuse_s2) {
              list_cnt2 = cnt2.p;
            } else {
              hipLaunchKernelGGL(k_ransac_stage2_survivors, dim3((unsigned)((b + 255) / 256), (unsigned)n_prob), dim3(256), 0, s,
                                 d_probs, hlist.p, d_nsurv, bmax, cnt2.p, hlist2.p, n_surv2.p, s2_total.p);
              list_p = hlist2.p;
              list_n = n_surv2.p;
            }
          }
          if (!(knockout & 1))
            hipLaunchKernelGGL(k_ransac_count_few, dim3((unsigned)fslices, (unsigned)n_prob, (unsigned)fslots), dim3(256), 0, s,
                               d_probs, pk.p, tot1, hyp_r, bmax, thr2, scale, res_cnt.p, cand_err.p, list_p, list_n, list_stride,
                               (knockout & 2) ? (const int32_t*)nullptr : list_cnt2);
          err_known = true;
        } else {
          const int ltiles = tiles < 4 ? tiles : 4;  // tile slots; the kernel strides over longer lists
          hipLaunchKernelGGL(k_ransac_count<true>, dim3((unsigned)(ltiles * lsplits), (unsigned)n_prob),
                             dim3(256), 0, s, d_probs, pk.p, tot1, hyp_r, it0, b, bmax, lsplits, thr2,
                             res_cnt.p, hlist.p, d_nsurv);
        }
      }
      if (check) {
        CS_HIP_CHECK(hipMemset2DAsync(exact_dbg.p, sizeof(int32_t) * bmax, 0, sizeof(int32_t) * b,
                                      n_prob, s));
        hipLaunchKernelGGL(k_ransac_count<false>, dim3((unsigned)(tiles * splits), (unsigned)n_prob),
                           dim3(256), 0, s, d_probs, pk.p, tot1, hyp_r, it0, b, bmax, splits, thr2,
                           exact_dbg.p, (const int32_t*)nullptr, (const int32_t*)nullptr);
        hipLaunchKernelGGL(k_ransac_check_bound, dim3((unsigned)((b + 255) / 256), (unsigned)n_prob),
                           dim3(256), 0, s, d_probs, exact_dbg.p, cnt_up_r, it0, b, bmax, chk_stats.p);
        if (run_s2)
          hipLaunchKernelGGL(k_ransac_check_bound2, dim3(PF_S2_CAP / 256, (unsigned)n_prob), dim3(256), 0, s, d_probs,
                             exact_dbg.p, hlist.p, d_nsurv, bmax, cnt2.p, chk_stats.p);
      }
    }
    static const hipError_t scan1_lds = hipFuncSetAttribute(
        reinterpret_cast<const void*>(k_ransac_scan1), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    CS_REQUIRE(scan1_lds == hipSuccess, CS_ERR_HIP, "cs_ransac_batch: cannot reserve LDS for k_ransac_scan1");
    // the best-candidate update (k_ransac_scan2) runs inside scan1 when the errors of all survivors are already known
    // (CS_RANSAC_FUSE_SCAN=0: three kernels as before)
    static const bool fuse_scan = !(getenv("CS_RANSAC_FUSE_SCAN") && getenv("CS_RANSAC_FUSE_SCAN")[0] == '0');
    const bool fused_scan = err_known && fuse_scan;
    hipLaunchKernelGGL(k_ransac_scan1, dim3((unsigned)n_prob), dim3(256), sizeof(int32_t) * (b + 64), s, d_probs, n_prob,
                       res_cnt.p, it0, b, bmax, ransac_n, max_iter, log_1mc, cand.p, d_nactive,
                       fused_scan ? cand_err.p : (const unsigned long long*)nullptr, hyp_r, nsurv_of(cur.par ^ 1),
                       nactive_of(cur.par ^ 1));
    if (!err_known)
      hipLaunchKernelGGL(k_ransac_err, dim3(8, (unsigned)n_prob), dim3(256), 0, s, d_probs, pk.p,
                         tot1, hyp_r, bmax, cand.p, thr2, scale, cand_err.p);
    if (!fused_scan)
      hipLaunchKernelGGL(k_ransac_scan2, dim3((unsigned)ceil_div(n_prob, 64)), dim3(64), 0, s,
                         d_probs, n_prob, hyp_r, cand.p, cand_err.p, err_known ? 1 : 0, it0, bmax, nsurv_of(cur.par ^ 1),
                         nactive_of(cur.par ^ 1));
    CS_LAUNCH_CHECK();
    // the per-problem state (est_k, done), the survivor counts and the activity counter come back in
    // one copy behind a synchronisation the chunk loop needs anyway
    std::vector<RansacProb> prev;
    if (pf) prev = hp;
    CS_HIP_CHECK(hipMemcpyAsync(h_state, state.p, st_bytes, hipMemcpyDeviceToHost, s));
    CS_HIP_CHECK(hipEventRecord(round_done.e, s));
    // front half of the next round: on the side stream once both rounds are prefiltered (before that
    // the rounds are short and the exact count of round i+1 would compete with round i), else behind
    // this round on the main stream
    Front nxt;
    const bool have_next = it0 + b < max_iter;
    if (have_next) {
      const bool nxt_pf = pf_alloc && it0 + b >= pf_from;
      hipStream_t st = (side && pf && nxt_pf) ? side : s;
      nxt = enqueue_front(it0 + b, cur.par ^ 1, st);
      if (nxt.on_side) {
        side_pending = true;
        side_par = nxt.par;
      }
      CS_LAUNCH_CHECK();
    }
    CS_HIP_CHECK(hipEventSynchronize(round_done.e));
    memcpy(hp.data(), h_state, st_probs);
    memcpy(h_surv.data(), h_state + st_probs + cur.par * st_cnt, sizeof(int32_t) * n_prob);
    int h_active = 0;
    memcpy(&h_active, h_state + st_probs + cur.par * st_cnt + st_surv, sizeof(int));
    if (pf) {
      max_surv_prev = 0;
      for (int p = 0; p < n_prob; ++p)
        if (!prev[p].done) {
          tot_surv += (unsigned long long)h_surv[p];
          if (h_surv[p] > max_surv_prev) max_surv_prev = h_surv[p];
        }
    }
    if (h_active == 0 || !have_next) break;
    cur = nxt;
  }
  // a speculative front half may still be running on the side stream (its workgroups see done = 1 and
  // leave): the scratch buffers go back to the main stream's pool only behind it
  if (side_pending) CS_HIP_CHECK(hipStreamWaitEvent(s, front_done[side_par].e, 0));
  if (trace_n) {
    std::vector<unsigned long long> ht(trace_n);
    CS_HIP_CHECK(hipMemcpy(ht.data(), trace.p, sizeof(unsigned long long) * trace_n, hipMemcpyDeviceToHost));
    FILE* f = fopen(getenv("CS_PF_TRACE_FILE") ? getenv("CS_PF_TRACE_FILE") : "/tmp/pf_trace.bin", "wb");
    if (f) {
      fwrite(ht.data(), sizeof(unsigned long long), trace_n, f);
      fclose(f);
    }
  }
  g_pf_stats[3] += tot_surv;
  g_pf_stats[4] += tot_eval;
  if (stage2 && getenv("CS_RANSAC_STAGE2_STATS")) {   // diagnostics: survivors of the first / second stage of this call
    unsigned long long h2 = 0;
    CS_HIP_CHECK(hipMemcpyAsync(&h2, s2_total.p, sizeof(h2), hipMemcpyDeviceToHost, s));
    CS_HIP_CHECK(hipStreamSynchronize(s));
    fprintf(stderr, "[cs_ransac_batch] prefilter survivors: stage 1 %llu, stage 2 %llu of %llu hypotheses\n", tot_surv, h2,
            tot_eval);
  }
  if (check) {
    unsigned long long h_stats[4] = {0, 0, 0, 0};
    CS_HIP_CHECK(hipMemcpyAsync(h_stats, chk_stats.p, sizeof(h_stats), hipMemcpyDeviceToHost, s));
    CS_HIP_CHECK(hipStreamSynchronize(s));
    g_pf_stats[0] += h_stats[0];
    g_pf_stats[1] += h_stats[1];
    g_pf_stats[2] += h_stats[2];
    CS_REQUIRE(h_stats[0] == 0, CS_ERR_INTERNAL,
               "cs_ransac_batch: prefilter bound violated for %llu hypotheses", h_stats[0]);
  }
  hipLaunchKernelGGL(k_ransac_finish, dim3((unsigned)ceil_div(n_prob, 64)), dim3(64), 0, s,
                     d_probs, n_prob, scale, d_T, d_inliers, d_rmse, d_iters);
  CS_LAUNCH_CHECK();
  CS_HIP_CHECK(hipStreamSynchronize(s));
  side_drain.clean = true;
  side_hyp_drain.clean = true;
  return CS_OK;
}

}  // extern "C"
