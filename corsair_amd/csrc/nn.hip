// Exact nearest-neighbour searches of the CORSAIR post-processing path, all in f64 so that the
// returned indices equal those of the reference's SciPy f64 routines:
//   cs_l2_topk      <- scipy cdist + argsort            (utils/retrieval.py:139-177)
//   cs_knn_feat     <- KDTree(feat1).query(feat0, k)     (utils/find_nn.py:43-49, utils/eval_pose.py:48-79,
//                                                         utils/symmetry.py:145-179)
//   cs_chamfer_1dir <- apply_transform + KDTree 1-NN     (utils/preprocess.py:39-48,67-70)
// The canonical distance is one f64 fma chain over the feature dimension in ascending order; ties go
// to the smaller index.  Every returned index / distance is that of the canonical chain.  The fast
// paths only decide WHICH rows get the canonical evaluation:
//   * 16-d k-NN: shortlist by |t|^2 - 2 q.t on the f16 matrix cores (k_knn_f16), canonical rescore,
//     verification of the shortlist, exhaustive recomputation of unverified queries (k_knn_feat);
//   * Chamfer / Hausdorff: arg-min of |t|^2 - 2 p.t on the f64 matrix pipe, canonical chain on the
//     candidates (k_chamfer_mfma);
//   * descriptor top-k at stress sizes: f64 matrix-pipe shortlist + canonical rescore (k_topk_mfma).
// The exhaustive VALU kernels (k_knn_feat, k_chamfer, k_dist_matrix + k_row_topk) remain selectable
// by environment variable and are what the fast paths are tested against.
#include <stdlib.h>

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <vector>

#include "common.h"

namespace cs {

// ------------------------------------------------------------------------------------------
// feature k-NN
// ------------------------------------------------------------------------------------------
struct KnnWork {
  int64_t q0;   // first query row of the tile (row of d_qf)
  int64_t t0;   // first target row of the problem (row of d_tf)
  int64_t o0;   // first output row of the tile (problem-major)
  int32_t qn;   // query rows in this tile (<= 256)
  int32_t tn;   // target rows
  int32_t prob;
  int32_t pad;
};

constexpr int KNN_MAXK = 8;
constexpr int KNN_TT = 128;  // target rows per LDS tile

template <int DIM>
__global__ __launch_bounds__(256) void k_knn_feat(const KnnWork* __restrict__ work,
                                                  const float* __restrict__ qf,
                                                  const float* __restrict__ tf, int k,
                                                  const int32_t* __restrict__ qlabel,
                                                  const int32_t* __restrict__ tlabel,
                                                  const int32_t* __restrict__ perm,
                                                  int32_t* __restrict__ out_idx,
                                                  double* __restrict__ out_dist,
                                                  const int32_t* __restrict__ tile_flag,
                                                  const int32_t* __restrict__ qflag) {
  // fallback mode (k_knn_rescore_f16 flagged some queries): only flagged tiles run, only flagged
  // queries are written
  if (tile_flag && !tile_flag[blockIdx.x]) return;
  // targets are converted to f64 once per tile (the inner loop is f64-VALU bound: one v_cvt less
  // per dimension and pair)
  __shared__ double t_lds[KNN_TT * DIM];
  __shared__ int32_t tl_lds[KNN_TT];
  const KnnWork wk = work[blockIdx.x];
  const int tid = threadIdx.x;
  const bool active = tid < wk.qn;
  const int64_t qrow = wk.q0 + (active ? tid : 0);

  double q[DIM];
#pragma unroll
  for (int c = 0; c < DIM; ++c) q[c] = (double)qf[qrow * DIM + c];
  int want = -1;
  const bool use_labels = qlabel != nullptr;
  if (use_labels) {
    int ql = qlabel[qrow];
    want = (ql >= 0 && ql < 8) ? perm[wk.prob * 8 + ql] : -2;
  }

  double bd[KNN_MAXK];
  int32_t bi[KNN_MAXK];
#pragma unroll
  for (int j = 0; j < KNN_MAXK; ++j) {
    bd[j] = INFINITY;
    bi[j] = -1;
  }

  for (int tbase = 0; tbase < wk.tn; tbase += KNN_TT) {
    const int tcount = min(KNN_TT, wk.tn - tbase);
    __syncthreads();
    for (int i = tid; i < tcount * DIM; i += 256) t_lds[i] = (double)tf[(wk.t0 + tbase) * DIM + i];
    if (use_labels)
      for (int i = tid; i < tcount; i += 256) tl_lds[i] = tlabel[wk.t0 + tbase + i];
    __syncthreads();
    if (!active) continue;
    for (int j = 0; j < tcount; ++j) {
      if (use_labels && tl_lds[j] != want) continue;
      double d = 0.0;
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        double diff = q[c] - t_lds[j * DIM + c];
        d = fma(diff, diff, d);
      }
      if (d < bd[KNN_MAXK - 1]) {
        // insert (d, tbase + j) into the running top-8 (ascending); strict < keeps the earlier
        // index on ties.  The first k entries are the answer.
        double cd = d;
        int32_t ci = tbase + j;
        bool carry = false;  // once placed, the displaced tail shifts down unconditionally
#pragma unroll
        for (int s = 0; s < KNN_MAXK; ++s) {
          if (carry || cd < bd[s]) {
            carry = true;
            double td = bd[s];
            int32_t ti = bi[s];
            bd[s] = cd;
            bi[s] = ci;
            cd = td;
            ci = ti;
          }
        }
      }
    }
  }
  if (active && (!qflag || qflag[wk.o0 + tid])) {
    for (int j = 0; j < k; ++j) {
      // (unrolled select keeps bd/bi in registers)
      double dj = INFINITY;
      int32_t ij = -1;
#pragma unroll
      for (int s = 0; s < KNN_MAXK; ++s)
        if (s == j) {
          dj = bd[s];
          ij = bi[s];
        }
      const int64_t orow = wk.o0 + tid;
      out_idx[orow * k + j] = ij;
      if (out_dist) out_dist[orow * k + j] = ij >= 0 ? sqrt(dj) : INFINITY;
    }
  }
}

// ------------------------------------------------------------------------------------------
// feature k-NN on the f64 matrix pipe (16-d features, the registration path's shape).
// dist(q, t) = |q|^2 + |t|^2 - 2 q.t with the dot products on v_mfma_f64_16x16x4_f64: the VALU is left
// with 3 f64 ops per pair instead of 32.  The expansion rounds differently from the canonical chain,
// so it only SHORTLISTS: every lane keeps the KNM_KK best rows of its quarter of the targets by the
// expanded distance, k_knn_rescore re-evaluates the 4 x KNM_KK candidates of a query with the
// canonical chain sum_c (q_c - t_c)^2 and ranks them by (distance, row).  The result equals the exact
// kernel's unless more than KNM_KK - k rows of one quarter tie with the k-th neighbour to within the
// rounding of the expansion (~1e-15 relative) -- the caveat every f64 distance-matrix method has.
// Labelled searches (part-to-part correspondences): the targets of a segment are visited in label
// order (stable), so a wave skips the 16-row tiles whose labels cannot match its 16 queries.
// ------------------------------------------------------------------------------------------
using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int KNM_NG = 1;               // 16-query groups per wave (each A fragment is used NG times)
constexpr int KNM_QT = 64 * KNM_NG;     // queries per workgroup
constexpr int KNM_TT = 256;             // target rows per LDS stage (one row per thread)
constexpr int KNM_PITCH = 18;           // floats per LDS row: conflict-free ds_read_b32 of the A fragments
constexpr int KNM_KK = 8;               // shortlist per lane (k <= KNM_KK - 2)
constexpr int KNM_PEND = 4;             // pending (not yet ranked) candidates per lane

__global__ void k_seg_keys(const int64_t* __restrict__ off, int n_seg, const int32_t* __restrict__ label,
                           uint32_t* __restrict__ keys, int32_t* __restrict__ rows) {
  const int sg = blockIdx.y;
  const int64_t b = off[sg], e = off[sg + 1];
  for (int64_t i = b + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = label[i];
    keys[i] = (uint32_t)sg * 16u + (uint32_t)((l >= 0 && l < 8) ? l : 8);
    rows[i] = (int32_t)(i - b);  // row local to the segment
  }
}

// lab_start[seg * 10 + l] = first row (label order, local to the segment) whose label key is >= l,
// l = 0..9 (keys: 0..7 parts, 8 = no part); one thread per entry, binary search in the sorted keys
__global__ void k_label_starts(const int64_t* __restrict__ off, int n_seg,
                               const uint32_t* __restrict__ keys_sorted, int32_t* __restrict__ lab_start) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_seg * 10) return;
  const int sg = i / 10, l = i - sg * 10;
  const int64_t b = off[sg], e = off[sg + 1];
  const uint32_t key = (uint32_t)sg * 16u + (uint32_t)l;
  int64_t lo = b, hi = e;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys_sorted[mid] < key) lo = mid + 1; else hi = mid;
  }
  lab_start[i] = (int32_t)(lo - b);
}

__global__ __launch_bounds__(256) void k_knn_mfma16(const KnnWork* __restrict__ work,
                                                    const float* __restrict__ qf,
                                                    const float* __restrict__ tf,
                                                    const double* __restrict__ qnorm,
                                                    const double* __restrict__ tnorm,
                                                    const int32_t* __restrict__ qlabel,
                                                    const int32_t* __restrict__ tlabel,
                                                    const int32_t* __restrict__ perm,
                                                    const int32_t* __restrict__ torder,
                                                    const int32_t* __restrict__ lab_start,
                                                    int32_t* __restrict__ cand_i) {
  // double-buffered stage: features (f32, converted when the fragment is read), |t|^2, label, row id
  __shared__ float t_lds[2][KNM_TT * KNM_PITCH];
  __shared__ double tn_lds[2][KNM_TT];
  __shared__ int32_t tl_lds[2][KNM_TT];
  __shared__ int32_t ti_lds[2][KNM_TT];
  __shared__ int32_t wrange[2][4];
  const KnnWork wk = work[blockIdx.x];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;  // query within a group (B side); target row within a 16-row tile (A side)
  const int kq = lane >> 4;   // k slot of the operands; row group of the results
  const bool use_labels = qlabel != nullptr;
  double qb[KNM_NG][4], my_qn[KNM_NG];
  int want[KNM_NG];
  bool qvalid[KNM_NG];
  int wmin = 0x7fffffff, wmax = -2;
#pragma unroll
  for (int g = 0; g < KNM_NG; ++g) {
    const int qloc = wave * 16 * KNM_NG + 16 * g + col;
    qvalid[g] = qloc < wk.qn;
    const int64_t qrow = wk.q0 + (qvalid[g] ? qloc : 0);
#pragma unroll
    // B[k = 4 s + kq][query], pre-scaled by -2 (exact): with the accumulator preloaded with |t|^2 the
    // chain delivers |t|^2 - 2 q.t, the ranking value of the query (|q|^2 is the same for all targets)
    for (int s4 = 0; s4 < 4; ++s4) qb[g][s4] = -2.0 * (double)qf[qrow * 16 + 4 * s4 + kq];
    my_qn[g] = 0.0;
    want[g] = -1;
    if (use_labels) {
      const int ql = qlabel[qrow];
      want[g] = (qvalid[g] && ql >= 0 && ql < 8) ? perm[wk.prob * 8 + ql] : -2;
      if (want[g] >= 0) wmin = min(wmin, want[g]);
      wmax = max(wmax, want[g]);
    }
  }
  // label window of the wave's queries (tile skipping)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    wmin = min(wmin, __shfl_xor(wmin, off));
    wmax = max(wmax, __shfl_xor(wmax, off));
  }
  // Ranked shortlist (ascending) plus a small unranked pending list per lane.  A candidate below the
  // lane's threshold is only appended (a register shift); the pending lists of the whole wave are
  // ranked together when one of them is full.  Ranking on every hit would run the 8-slot insertion
  // for nearly every element, because some lane of the 64 almost always has a hit.
  // Labelled search: targets are visited in label order, and the queries of a workgroup (sorted by
  // part) want only one or two labels: the scan covers just the target rows of those labels.
  int t_lo = 0, t_hi = wk.tn;
  if (use_labels && lab_start != nullptr) {
    if (lane == 0) {
      wrange[0][wave] = wmin;
      wrange[1][wave] = wmax;
    }
    __syncthreads();
    const int bmin = min(min(wrange[0][0], wrange[0][1]), min(wrange[0][2], wrange[0][3]));
    const int bmax = max(max(wrange[1][0], wrange[1][1]), max(wrange[1][2], wrange[1][3]));
    if (bmin > bmax) {
      t_hi = 0;  // no query of this workgroup has a part
    } else {
      t_lo = lab_start[wk.pad * 10 + bmin];
      t_hi = lab_start[wk.pad * 10 + bmax + 1];
    }
  }
  double bd[KNM_NG][KNM_KK], pd[KNM_NG][KNM_PEND];
  int32_t bi[KNM_NG][KNM_KK], pi[KNM_NG][KNM_PEND];
  int pn[KNM_NG];
#pragma unroll
  for (int g = 0; g < KNM_NG; ++g) {
    pn[g] = 0;
#pragma unroll
    for (int j = 0; j < KNM_KK; ++j) {
      bd[g][j] = INFINITY;
      bi[g][j] = 0x7fffffff;
    }
#pragma unroll
    for (int j = 0; j < KNM_PEND; ++j) {
      pd[g][j] = INFINITY;
      pi[g][j] = 0x7fffffff;
    }
  }
  auto rank_pending = [&](int g) {
#pragma unroll
    for (int e = 0; e < KNM_PEND; ++e) {
      double cd = pd[g][e];  // +inf in unused slots: never inserted
      int32_t ci = pi[g][e];
      pd[g][e] = INFINITY;
      if (cd < bd[g][KNM_KK - 1]) {
        bool carry = false;
#pragma unroll
        for (int s2 = 0; s2 < KNM_KK; ++s2) {
          if (carry || cd < bd[g][s2]) {
            carry = true;
            const double td = bd[g][s2];
            const int32_t ti = bi[g][s2];
            bd[g][s2] = cd;
            bi[g][s2] = ci;
            cd = td;
            ci = ti;
          }
        }
      }
    }
    pn[g] = 0;
  };
  // staging registers: thread tid owns row tid of the stage
  float4 sf[4];
  double sn;
  int32_t sl, si;
  auto stage_load = [&](int tbase) {
    const int j = tid;
    const bool ok = tbase + j < t_hi;
    const int src = ok ? (torder ? torder[wk.t0 + tbase + j] : tbase + j) : 0;
    const float4* rp = reinterpret_cast<const float4*>(tf + (wk.t0 + src) * 16);
#pragma unroll
    for (int c = 0; c < 4; ++c) sf[c] = rp[c];
    sn = ok ? tnorm[wk.t0 + src] : INFINITY;  // +inf distance for rows past the segment
    const int tl = (ok && use_labels) ? tlabel[wk.t0 + src] : (ok ? 0 : 8);
    sl = (tl >= 0 && tl < 8) ? tl : 8;         // 8 = no part: matches no query, sorts last
    si = ok ? src : 0x7fffffff;
  };
  auto stage_store = [&](int b) {
    float2* dst = reinterpret_cast<float2*>(&t_lds[b][tid * KNM_PITCH]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dst[2 * c] = make_float2(sf[c].x, sf[c].y);
      dst[2 * c + 1] = make_float2(sf[c].z, sf[c].w);
    }
    tn_lds[b][tid] = sn;
    tl_lds[b][tid] = sl;
    ti_lds[b][tid] = si;
  };
  double thr = INFINITY;  // pruning threshold of the lane's query (see the ranking step below)
  if (t_hi > t_lo) {
    stage_load(t_lo);
    stage_store(0);
  }
  __syncthreads();
  int buf = 0;
  for (int tbase = t_lo; tbase < t_hi; tbase += KNM_TT) {
    const int tcount = min(KNM_TT, t_hi - tbase);
    const bool more = tbase + KNM_TT < t_hi;
    if (more) stage_load(tbase + KNM_TT);  // global loads in flight during the tiles below
    // two 16-row tiles per iteration: their MFMA chains are independent, so the second chain issues
    // while the first drains, and the LDS reads of both are in flight together (rows past tcount carry
    // |t|^2 = +inf and never enter a shortlist)
    static_assert(KNM_NG == 1, "the two-tile loop below is written for one query group per wave");
    for (int t = 0; t < (tcount + 15) / 16; t += 2) {
      const float* ap = &t_lds[buf][(16 * t + col) * KNM_PITCH + kq];
      double a0[4], a1[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        a0[s4] = (double)ap[4 * s4];                        // A[target row][k = 4 s + kq]
        a1[s4] = (double)ap[16 * KNM_PITCH + 4 * s4];
      }
      f64x4 acc0, acc1;
      int tl0[4], tl1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0[r] = tn_lds[buf][16 * t + kq + 4 * r];
        acc1[r] = tn_lds[buf][16 * t + 16 + kq + 4 * r];
        tl0[r] = tl_lds[buf][16 * t + kq + 4 * r];
        tl1[r] = tl_lds[buf][16 * t + 16 + kq + 4 * r];
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s4], qb[0][s4], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s4], qb[0][s4], acc1, 0, 0, 0);
      }
      // acc[r] = |t|^2 - 2 q.t for target row 16 t (+16) + kq + 4 r and the lane's query
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double dist = u ? acc1[r] : acc0[r];
          if (use_labels && (u ? tl1[r] : tl0[r]) != want[0]) dist = INFINITY;
          if (dist < thr) {
#pragma unroll
            for (int e = KNM_PEND - 1; e > 0; --e) {
              pd[0][e] = pd[0][e - 1];
              pi[0][e] = pi[0][e - 1];
            }
            pd[0][0] = dist;
            pi[0][0] = ti_lds[buf][16 * t + 16 * u + kq + 4 * r];
            ++pn[0];
          }
          if (__any(pn[0] == KNM_PEND)) {
            rank_pending(0);
            // The four lanes of a query (kq = 0..3) scan disjoint quarters of the targets.  Whichever of
            // them already holds KNM_KK candidates below tau bounds the query's KNM_KK-th best by tau,
            // so all four may prune with the smallest of their KNM_KK-th values.
            thr = bd[0][KNM_KK - 1];
            thr = fmin(thr, __shfl_xor(thr, 16));
            thr = fmin(thr, __shfl_xor(thr, 32));
          }
        }
      }
    }
    if (more) stage_store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int g = 0; g < KNM_NG; ++g) {
    rank_pending(g);
    const int qloc = wave * 16 * KNM_NG + 16 * g + col;
    if (qvalid[g]) {
      const int64_t base = ((wk.o0 + qloc) * 4 + kq) * KNM_KK;
#pragma unroll
      for (int j = 0; j < KNM_KK; ++j) cand_i[base + j] = bi[g][j];
    }
  }
}

// One thread per query: canonical distances of its 4 * KNM_KK candidates, k best by (distance, row).
__global__ void k_knn_rescore16(const KnnWork* __restrict__ work, const float* __restrict__ qf,
                                const float* __restrict__ tf, const int32_t* __restrict__ cand_i,
                                int k, int32_t* __restrict__ out_idx, double* __restrict__ out_dist) {
  const KnnWork wk = work[blockIdx.x];
  const int qloc = threadIdx.x;
  if (qloc >= wk.qn) return;
  const int64_t qrow = wk.q0 + qloc;
  double q[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) q[c] = (double)qf[qrow * 16 + c];
  double bd[KNN_MAXK];
  int32_t bi[KNN_MAXK];
#pragma unroll
  for (int j = 0; j < KNN_MAXK; ++j) {
    bd[j] = INFINITY;
    bi[j] = 0x7fffffff;
  }
  const int32_t* ci = cand_i + (wk.o0 + qloc) * 4 * KNM_KK;
  for (int c = 0; c < 4 * KNM_KK; ++c) {
    const int32_t row = ci[c];
    if (row == 0x7fffffff) continue;
    const float* tp = tf + (wk.t0 + row) * 16;
    double d = 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const double diff = q[e] - (double)tp[e];
      d = fma(diff, diff, d);
    }
    // ordered insertion by (d, row)
    double cd = d;
    int32_t cr = row;
    bool carry = false;
#pragma unroll
    for (int s2 = 0; s2 < KNN_MAXK; ++s2) {
      if (carry || cd < bd[s2] || (cd == bd[s2] && cr < bi[s2])) {
        carry = true;
        const double td = bd[s2];
        const int32_t ti = bi[s2];
        bd[s2] = cd;
        bi[s2] = cr;
        cd = td;
        cr = ti;
      }
    }
  }
  const int64_t orow = wk.o0 + qloc;
  for (int j = 0; j < k; ++j) {
    double dj = INFINITY;
    int32_t ij = 0x7fffffff;
#pragma unroll
    for (int s2 = 0; s2 < KNN_MAXK; ++s2)
      if (s2 == j) {
        dj = bd[s2];
        ij = bi[s2];
      }
    const bool have = ij != 0x7fffffff;
    out_idx[orow * k + j] = have ? ij : -1;
    if (out_dist) out_dist[orow * k + j] = have ? sqrt(dj) : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------
// feature k-NN shortlist on the f16 matrix cores (16-d features, k <= 6): the default path.
// On gfx950 the f64 MFMA runs on the vector unit's double-precision ALUs (78.6 TF either way; measured:
// the f64-MFMA kernel above cannot overlap its MFMAs with its own VALU work), so it stays ~3x above its
// bound.  Here |t|^2 - 2 q.t is evaluated like the RANSAC prefilter: every feature is split into f16
// hi + lo, hi*hi + lo*hi + hi*lo (48 products) is three v_mfma_f32_32x32x16_f16 per 32 x 32 tile (true
// matrix cores, ~20x the f64 rate, co-issuing with the VALU), the accumulator is preloaded with |t|^2.
//   rows = targets (LDS, staged by LDS-DMA from a 112-B-pitch f16 image in visiting order),
//   cols = queries (registers): a lane owns one query and 16 of the tile's 32 target rows.
// The result is only a SHORTLIST (2 x KNF_KK candidates per query by the approximate value).
// k_knn_rescore_f16 re-evaluates them with the canonical f64 chain, ranks by (distance, row) and
// VERIFIES the shortlist: every target that was not kept has an approximate value >= tau (the final
// pruning threshold), hence an exact one >= tau - eps; if the exact k-th distance is not below that,
// the query is flagged and recomputed by the exhaustive kernel (k_knn_feat<16>, flagged tiles only).
// The answer therefore equals the exhaustive kernel's for every query.
// ------------------------------------------------------------------------------------------
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int KNF_PITCH = 56;   // halfs per image row (112 B): conflict-free ds_read_b128 fragments
constexpr int KNF_ROWS = 192;   // target rows per LDS stage (6 MFMA row tiles, 21 KiB)
constexpr int KNF_NG = 2;       // 32-query groups per wave
constexpr int KNF_QT = 4 * 32 * KNF_NG;  // queries per workgroup
static_assert(KNF_QT == 256, "tiles of the f16 path and of the exhaustive fallback must coincide");
constexpr int KNF_KK = 8;       // shortlist per lane (two lanes per query); 6 was measured: 23 of 189 517 queries fail the
                                // verification and their exhaustive recomputation costs more than the shorter lists save
constexpr int KNF_PEND = 4;

__device__ __forceinline__ void knf_split(float v, _Float16* hi, _Float16* lo) {
  const _Float16 h = (_Float16)v;
  *hi = h;
  *lo = (_Float16)(v - (float)h);
}

// target image: row j of the image = target (t0 + torder[t0 + j]) (or t0 + j): [th(16) | tl(16) | th(16) | 0(8)],
// tn32 = |t|^2 (f64 chain, rounded up to f32 is not needed: the verification budget covers its rounding),
// ti32 = row local to the segment.  seg_t2max[seg] = max |t|^2 (error budget of the verification).
__global__ void k_knf_pack_targets(const float* __restrict__ tf, const int64_t* __restrict__ toff, int n_seg,
                                   const int32_t* __restrict__ torder, _Float16* __restrict__ img,
                                   float* __restrict__ tn32, int32_t* __restrict__ ti32,
                                   unsigned* __restrict__ seg_t2max_bits) {
  const int sg = blockIdx.y;
  const int64_t b = toff[sg], e = toff[sg + 1];
  float mx = 0.f;
  for (int64_t j = b + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < e; j += (int64_t)gridDim.x * blockDim.x) {
    const int32_t loc = torder ? torder[j] : (int32_t)(j - b);
    const float* f = tf + (b + loc) * 16;
    union {
      _Float16 h[KNF_PITCH];
      uint4 v[7];
    } row;
    double n2 = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      _Float16 hi, lo;
      knf_split(f[c], &hi, &lo);
      row.h[c] = hi;
      row.h[16 + c] = lo;
      row.h[32 + c] = hi;
      n2 = fma((double)f[c], (double)f[c], n2);
    }
#pragma unroll
    for (int c = 48; c < KNF_PITCH; ++c) row.h[c] = (_Float16)0.0f;
    uint4* dst = reinterpret_cast<uint4*>(img + j * KNF_PITCH);
#pragma unroll
    for (int c = 0; c < 7; ++c) dst[c] = row.v[c];
    tn32[j] = (float)n2;
    ti32[j] = loc;
    mx = fmaxf(mx, (float)n2 * 1.0000002f);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(&seg_t2max_bits[sg], __float_as_uint(mx));
}

// query operand rows: [-2 qh(16) | -2 qh(16) | -2 ql(16)] (scaling by 2 is exact in f16 below the range limit)
__global__ void k_knf_pack_queries(const float* __restrict__ qf, int64_t n, _Float16* __restrict__ qrows,
                                   float* __restrict__ qn32) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  union {
    _Float16 h[48];
    uint4 v[6];
  } row;
  double n2 = 0.0;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    _Float16 hi, lo;
    knf_split(-2.0f * qf[i * 16 + c], &hi, &lo);
    row.h[c] = hi;
    row.h[16 + c] = hi;
    row.h[32 + c] = lo;
    n2 = fma((double)qf[i * 16 + c], (double)qf[i * 16 + c], n2);
  }
  qn32[i] = (float)n2 * 1.0000002f;   // |q|^2, rounded up: error budget of the threshold pass
  uint4* dst = reinterpret_cast<uint4*>(qrows + i * 48);
#pragma unroll
  for (int c = 0; c < 6; ++c) dst[c] = row.v[c];
}

// PASS = 1 (round 5): the THRESHOLD pass.  The shortlist kernel is bound by its hits, not by the matrix cores (MFMA busy
// 0.07): a lane starts with thr = +inf, its running 8th best falls like a record process (~55 hits per lane, and with 64
// lanes a hit in SOME lane on 70 % of the value steps), and every hit step pays the pending list and its ranking.  This pass
// bounds the query's k-th distance BEFORE the shortlist pass from tile minima alone -- no per-value work: one MFMA (hi.hi,
// 16 of the 48 products) per 32 x 32 tile, eight v_min3 for the tile's minimum of the lane's 16 rows, six v_med3 to keep the
// lane's six smallest tile minima.  The k-th smallest b_k is attained by k DIFFERENT rows, so the exact k-th distance is
// <= b_k + eps1 and every true neighbour has a full approximate value <= b_k + eps1 + eps3 =: thr0 -- the shortlist pass then
// starts from thr0 instead of +inf and sees ~8 hits per lane.  eps1 = 2^-9 (|q|^2 + |t|^2max) covers the hi.hi-only value
// (2 x 2^-11 relative on either side of q.t, times the factor 2, charged twice), eps3 = 2^-15 (...) the three-MFMA value as in
// k_knn_rescore_f16.  Nothing here decides a result: the shortlist is verified against its final threshold as before and a
// query whose list is short or unverifiable is recomputed exhaustively.
template <int PASS>
__global__ __launch_bounds__(256) void k_knn_f16(const KnnWork* __restrict__ work,
                                                 const _Float16* __restrict__ qrows,
                                                 const _Float16* __restrict__ img,
                                                 const float* __restrict__ tn32,
                                                 const int32_t* __restrict__ ti32,
                                                 const int32_t* __restrict__ qlabel,
                                                 const int32_t* __restrict__ perm,
                                                 const int32_t* __restrict__ lab_start,
                                                 int32_t* __restrict__ cand_i, float* __restrict__ cand_tau,
                                                 // PASS 1 writes thr0, PASS 0 starts from it (nullptr: from +inf)
                                                 float* __restrict__ thr0, const float* __restrict__ qn32,
                                                 const unsigned* __restrict__ seg_t2max_bits, int kq) {
  constexpr int STAGE_BYTES = KNF_ROWS * KNF_PITCH * 2;  // 21504
  constexpr int STAGE_KIB = STAGE_BYTES / 1024;
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_BYTES];
  __shared__ __attribute__((aligned(16))) float tn_s[2][KNF_ROWS];
  __shared__ int32_t wrange[2][4];
  const KnnWork wk = work[blockIdx.x];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int col = lane & 31;
  const bool use_labels = qlabel != nullptr;
  f16x8 bop[KNF_NG][3];
  int want[KNF_NG];
  bool qvalid[KNF_NG];
  int wmin = 0x7fffffff, wmax = -2;
#pragma unroll
  for (int g = 0; g < KNF_NG; ++g) {
    const int qloc = wave * 32 * KNF_NG + 32 * g + col;
    qvalid[g] = qloc < wk.qn;
    const int64_t qrow = wk.q0 + (qvalid[g] ? qloc : 0);
    const _Float16* row = qrows + qrow * 48 + 8 * half;
#pragma unroll
    for (int m = 0; m < 3; ++m) bop[g][m] = *reinterpret_cast<const f16x8*>(row + 16 * m);
    want[g] = -1;
    if (use_labels) {
      const int ql = qlabel[qrow];
      want[g] = (qvalid[g] && ql >= 0 && ql < 8) ? perm[wk.prob * 8 + ql] : -2;
      if (want[g] >= 0) wmin = min(wmin, want[g]);
      wmax = max(wmax, want[g]);
    }
  }
  int lab_lo = -1, lab_hi = -1;  // label passes (one pass, label -1, without labels)
  if (use_labels) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      wmin = min(wmin, __shfl_xor(wmin, off));
      wmax = max(wmax, __shfl_xor(wmax, off));
    }
    if (lane == 0) {
      wrange[0][wave] = wmin;
      wrange[1][wave] = wmax;
    }
    __syncthreads();
    lab_lo = min(min(wrange[0][0], wrange[0][1]), min(wrange[0][2], wrange[0][3]));
    lab_hi = max(max(wrange[1][0], wrange[1][1]), max(wrange[1][2], wrange[1][3]));
    if (lab_hi > 7) lab_hi = 7;
    if (lab_lo > lab_hi) lab_hi = lab_lo - 1;  // nothing to scan
  }
  // ranked shortlist + pending list per lane and group (see k_knn_mfma16 for the scheme)
  float bd[KNF_NG][KNF_KK], pd[KNF_NG][KNF_PEND], thr[KNF_NG];
  int32_t bi[KNF_NG][KNF_KK], pi[KNF_NG][KNF_PEND];
  int pn[KNF_NG];
  float tb[KNF_NG][6];   // PASS 1: the lane's six smallest tile minima, ascending
#pragma unroll
  for (int g = 0; g < KNF_NG; ++g) {
    pn[g] = 0;
    thr[g] = INFINITY;
    if (PASS == 0 && thr0) {
      const int qloc = wave * 32 * KNF_NG + 32 * g + col;
      thr[g] = thr0[wk.o0 + (qvalid[g] ? qloc : 0)];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) tb[g][j] = INFINITY;
#pragma unroll
    for (int j = 0; j < KNF_KK; ++j) {
      bd[g][j] = INFINITY;
      bi[g][j] = 0x7fffffff;
    }
#pragma unroll
    for (int j = 0; j < KNF_PEND; ++j) {
      pd[g][j] = INFINITY;
      pi[g][j] = 0x7fffffff;
    }
  }
  auto rank_pending = [&](int g) {
#pragma unroll
    for (int e = 0; e < KNF_PEND; ++e) {
      float cd = pd[g][e];
      int32_t ci = pi[g][e];
      pd[g][e] = INFINITY;
      if (cd < bd[g][KNF_KK - 1]) {
        bool carry = false;
#pragma unroll
        for (int s2 = 0; s2 < KNF_KK; ++s2) {
          if (carry || cd < bd[g][s2]) {
            carry = true;
            const float td = bd[g][s2];
            const int32_t ti = bi[g][s2];
            bd[g][s2] = cd;
            bi[g][s2] = ci;
            cd = td;
            ci = ti;
          }
        }
      }
    }
    pn[g] = 0;
    // the two lanes of a query scan disjoint halves of every tile: either one's KNF_KK-th value bounds
    // the query's KNF_KK-th best
    float t = bd[g][KNF_KK - 1];
    t = fminf(t, __shfl_xor(t, 32));
    thr[g] = fminf(thr[g], t);   // (never above the threshold pass's bound while the lists are still filling)
  };
  const char* gimg = reinterpret_cast<const char*>(img + (int64_t)wk.t0 * KNF_PITCH) + lane * 16;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  for (int lab = lab_lo; lab <= lab_hi; ++lab) {
    int t_lo = 0, t_hi = wk.tn;
    if (use_labels) {
      if (lab_start == nullptr) break;
      t_lo = lab_start[wk.pad * 10 + lab];
      t_hi = lab_start[wk.pad * 10 + lab + 1];
    }
    if (t_hi <= t_lo) continue;
    float thr_eff[KNF_NG];
#pragma unroll
    for (int g = 0; g < KNF_NG; ++g) thr_eff[g] = (!use_labels || want[g] == lab) ? thr[g] : -INFINITY;
    // (LDS-DMA as inline asm + |t|^2 / row ids loaded before it and stored after the stage's compute: see k_topk_f16)
    auto issue_dma = [&](int b, int base) {
      const char* gp = gimg + (int64_t)base * (KNF_PITCH * 2);
#pragma unroll
      for (int i = 0; i < (STAGE_KIB + 3) / 4; ++i) {
        const int piece = wave + 4 * i;
        if (piece < STAGE_KIB) lds_dma16(gp + piece * 1024, lds_base + b * STAGE_BYTES + piece * 1024);
      }
    };
    auto load_rows = [&](int base, float& tn) {   // unconditional (clamped) loads
      int r = base + (tid % KNF_ROWS);
      r = r > t_hi - 1 ? t_hi - 1 : r;
      r = r < t_lo ? t_lo : r;
      tn = tn32[wk.t0 + r];
    };
    auto store_rows = [&](int b, int base, float tn) {
      if (tid < KNF_ROWS) tn_s[b][tid] = base + tid < t_hi ? tn : INFINITY;  // rows past the label's range can never be hit
    };
    __syncthreads();  // the previous label pass may still read the buffers
    {
      float tn0;
      load_rows(t_lo, tn0);
      issue_dma(0, t_lo);
      store_rows(0, t_lo, tn0);
    }
    int buf = 0;
    for (int base = t_lo; base < t_hi; base += KNF_ROWS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const bool more = base + KNF_ROWS < t_hi;
      float tn_next;
      load_rows(base + KNF_ROWS, tn_next);
      if (more) issue_dma(buf ^ 1, base + KNF_ROWS);
#pragma unroll 1
      for (int t = 0; t < KNF_ROWS / 32; ++t) {
        if (base + 32 * t >= t_hi) break;  // whole tile past the range (block-uniform)
        const _Float16* arow =
            reinterpret_cast<const _Float16*>(lds + buf * STAGE_BYTES) + (t * 32 + col) * KNF_PITCH + 8 * half;
        const int32_t tile_pos = base + 32 * t + 4 * half;
        f16x8 a[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) a[m] = *reinterpret_cast<const f16x8*>(arow + 16 * m);
        // accumulator input: |t|^2 of the 16 rows this lane owns: (r & 3) + 8 (r >> 2) + 4 half
        f32x16 c16;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 v = *reinterpret_cast<const float4*>(&tn_s[buf][t * 32 + 8 * q4 + 4 * half]);
          c16[4 * q4 + 0] = v.x; c16[4 * q4 + 1] = v.y; c16[4 * q4 + 2] = v.z; c16[4 * q4 + 3] = v.w;
        }
#pragma unroll
        for (int g = 0; g < KNF_NG; ++g) {
          f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bop[g][0], c16, 0, 0, 0);
          if (PASS == 1) {
            float m = fminf(fminf(d[0], d[1]), d[2]);
            m = fminf(fminf(m, d[3]), d[4]);
            m = fminf(fminf(m, d[5]), d[6]);
            m = fminf(fminf(m, d[7]), d[8]);
            m = fminf(fminf(m, d[9]), d[10]);
            m = fminf(fminf(m, d[11]), d[12]);
            m = fminf(fminf(m, d[13]), d[14]);
            m = fminf(m, d[15]);
            if (use_labels && want[g] != lab) m = INFINITY;
            // sorted insertion without a branch: new j-th = median of (old j-1-th, old j-th, m)
            const float o0 = tb[g][0], o1 = tb[g][1], o2 = tb[g][2], o3 = tb[g][3], o4 = tb[g][4], o5 = tb[g][5];
            tb[g][0] = fminf(o0, m);
            tb[g][1] = __builtin_amdgcn_fmed3f(o0, o1, m);
            tb[g][2] = __builtin_amdgcn_fmed3f(o1, o2, m);
            tb[g][3] = __builtin_amdgcn_fmed3f(o2, o3, m);
            tb[g][4] = __builtin_amdgcn_fmed3f(o3, o4, m);
            tb[g][5] = __builtin_amdgcn_fmed3f(o4, o5, m);
            continue;
          }
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bop[g][1], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2], bop[g][2], d, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            // one compare + one scalar branch per value while no lane of the wave has a hit (three of four values
            // once a few hundred targets have been seen); the pending-list push only runs behind it
#ifdef KNF_NOHIT
            const bool hit = d[r] < thr_eff[g] - 1.0e30f;
#else
            const bool hit = d[r] < thr_eff[g];
#endif
            if (__any(hit)) {
              if (hit) {
#pragma unroll
                for (int e = KNF_PEND - 1; e > 0; --e) {
                  pd[g][e] = pd[g][e - 1];
                  pi[g][e] = pi[g][e - 1];
                }
                pd[g][0] = d[r];
                pi[g][0] = tile_pos + ((r & 3) + 8 * (r >> 2));   // position in the image; its row id is looked up at the end
                ++pn[g];
              }
              if (__any(pn[g] == KNF_PEND)) {
                rank_pending(g);
                thr_eff[g] = (!use_labels || want[g] == lab) ? thr[g] : -INFINITY;
              }
            }
          }
        }
      }
      if (more) store_rows(buf ^ 1, base + KNF_ROWS, tn_next);
      buf ^= 1;
    }
  }
  if (PASS == 1) {
#pragma unroll
    for (int g = 0; g < KNF_NG; ++g) {
      float bk = INFINITY;
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j == kq - 1) bk = tb[g][j];
      bk = fminf(bk, __shfl_xor(bk, 32));     // either lane's k-th smallest tile minimum bounds the query's k-th value
      const int qloc = wave * 32 * KNF_NG + 32 * g + col;
      if (qvalid[g] && half == 0) {
        const float budget = qn32[wk.q0 + qloc] + __uint_as_float(seg_t2max_bits[wk.pad]);
        // (+ eps1 + eps3, rounded up, and strictly above every value it has to admit)
        thr0[wk.o0 + qloc] = bk + budget * (0x1.0p-9f + 0x1.0p-14f) + fabsf(bk) * 0x1.0p-20f;
      }
    }
    return;
  }
#pragma unroll
  for (int g = 0; g < KNF_NG; ++g) {
    rank_pending(g);
    const int qloc = wave * 32 * KNF_NG + 32 * g + col;
    if (qvalid[g]) {
      const int64_t base = ((wk.o0 + qloc) * 2 + half) * KNF_KK;
      // shortlist entries are positions in the target image (no LDS read on the hit path): row ids only now
#pragma unroll
      for (int j = 0; j < KNF_KK; ++j) cand_i[base + j] = bi[g][j] != 0x7fffffff ? ti32[wk.t0 + bi[g][j]] : 0x7fffffff;
      if (half == 0) cand_tau[wk.o0 + qloc] = thr[g];
    }
  }
}

// One thread per query: canonical distances of its 2 * KNF_KK candidates, k best by (distance, row), and
// the verification of the shortlist (see the header of this section).  flag[tile] != 0 -> k_knn_feat
// recomputes that tile's flagged queries exhaustively.
__global__ void k_knn_rescore_f16(const KnnWork* __restrict__ work, const float* __restrict__ qf,
                                  const float* __restrict__ tf, const int32_t* __restrict__ cand_i,
                                  const float* __restrict__ cand_tau,
                                  const unsigned* __restrict__ seg_t2max_bits, int k,
                                  int32_t* __restrict__ out_idx, double* __restrict__ out_dist,
                                  int32_t* __restrict__ qflag, int32_t* __restrict__ tile_flag) {
  const KnnWork wk = work[blockIdx.x];
  const int qloc = threadIdx.x;
  if (qloc >= wk.qn) return;
  const int64_t qrow = wk.q0 + qloc;
  double q[16], qn2 = 0.0;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    q[c] = (double)qf[qrow * 16 + c];
    qn2 = fma(q[c], q[c], qn2);
  }
  double bd[KNN_MAXK];
  int32_t bi[KNN_MAXK];
#pragma unroll
  for (int j = 0; j < KNN_MAXK; ++j) {
    bd[j] = INFINITY;
    bi[j] = 0x7fffffff;
  }
  const int32_t* ci = cand_i + (wk.o0 + qloc) * 2 * KNF_KK;
  int n_cand = 0;
  for (int c = 0; c < 2 * KNF_KK; ++c) {
    const int32_t row = ci[c];
    if (row == 0x7fffffff) continue;
    ++n_cand;
    const float* tp = tf + (wk.t0 + row) * 16;
    double d = 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const double diff = q[e] - (double)tp[e];
      d = fma(diff, diff, d);
    }
    double cd = d;
    int32_t cr = row;
    bool carry = false;
#pragma unroll
    for (int s2 = 0; s2 < KNN_MAXK; ++s2) {
      if (carry || cd < bd[s2] || (cd == bd[s2] && cr < bi[s2])) {
        carry = true;
        const double td = bd[s2];
        const int32_t ti = bi[s2];
        bd[s2] = cd;
        bi[s2] = cr;
        cd = td;
        cr = ti;
      }
    }
  }
  const int64_t orow = wk.o0 + qloc;
  double dk = INFINITY;  // exact k-th distance
  for (int j = 0; j < k; ++j) {
    double dj = INFINITY;
    int32_t ij = 0x7fffffff;
#pragma unroll
    for (int s2 = 0; s2 < KNN_MAXK; ++s2)
      if (s2 == j) {
        dj = bd[s2];
        ij = bi[s2];
      }
    const bool have = ij != 0x7fffffff;
    out_idx[orow * k + j] = have ? ij : -1;
    if (out_dist) out_dist[orow * k + j] = have ? sqrt(dj) : INFINITY;
    dk = dj;
  }
  // Verification.  Targets that were dropped have approximate value >= tau, i.e. exact
  // |t|^2 - 2 q.t >= tau - eps with eps covering: 48 f32 accumulation steps and the dropped lo*lo
  // products relative to sum |terms| <= |q|^2 + |t|^2, the hi+lo split residuals, the f32 rounding of
  // |t|^2 -- together < 2^-17 (|q|^2 + |t|^2_max); charged 2^-15.  If the shortlists were never filled
  // (tau = +inf) every target of the query's part is a candidate and nothing was dropped.
  const float tau = cand_tau[orow];
  const double t2max = (double)__uint_as_float(seg_t2max_bits[wk.pad]);
  const double eps = 0x1.0p-15 * (qn2 + t2max);
  bool ok = true;
  if (tau < INFINITY) {
    // need: exact k-th (as |t|^2 - 2 q.t = d - |q|^2) strictly below every dropped target's exact value;
    // equality would need the (distance, row) tie rule, which the shortlist does not know
    ok = n_cand >= k && (dk - qn2) < (double)tau - eps;
  }
  if (!(qn2 < 1.0e8) || !(t2max < 1.0e8)) ok = false;  // outside the f16 range: not trusted at all
  qflag[orow] = ok ? 0 : 1;
  if (!ok) atomicOr(&tile_flag[blockIdx.x], 1);
}

// ------------------------------------------------------------------------------------------
// descriptor distance matrix + row top-k
// ------------------------------------------------------------------------------------------
constexpr int DM_QT = 16;   // queries per block
constexpr int DM_CT = 64;   // catalog rows per block
constexpr int DM_DC = 64;   // feature chunk staged in LDS

// D2[q, x] = sum_c (q_c - x_c)^2 in f64, c ascending.  thread = (catalog row j, query group g of 4).
__global__ __launch_bounds__(256) void k_dist_matrix(const float* __restrict__ Q, int64_t nq,
                                                     const float* __restrict__ X, int64_t nx,
                                                     int d, int64_t x_begin, int64_t x_count,
                                                     double* __restrict__ D2) {
  __shared__ float q_lds[DM_QT * DM_DC];
  __shared__ float x_lds[DM_DC * (DM_CT + 1)];
  const int tid = threadIdx.x;
  const int j = tid & 63;
  const int g = tid >> 6;
  const int64_t q0 = (int64_t)blockIdx.y * DM_QT;
  const int64_t xl0 = (int64_t)blockIdx.x * DM_CT;  // local to the slab
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int c0 = 0; c0 < d; c0 += DM_DC) {
    const int dc = min(DM_DC, d - c0);
    __syncthreads();
    for (int i = tid; i < DM_QT * DM_DC; i += 256) {
      int r = i / DM_DC, c = i - r * DM_DC;
      float v = 0.f;
      if (q0 + r < nq && c < dc) v = Q[(q0 + r) * d + c0 + c];
      q_lds[i] = v;
    }
    for (int i = tid; i < DM_CT * DM_DC; i += 256) {
      int r = i / DM_DC, c = i - r * DM_DC;
      float v = 0.f;
      if (xl0 + r < x_count && c < dc) v = X[(x_begin + xl0 + r) * d + c0 + c];
      x_lds[c * (DM_CT + 1) + r] = v;
    }
    __syncthreads();
    for (int c = 0; c < dc; ++c) {
      double xv = (double)x_lds[c * (DM_CT + 1) + j];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        double diff = (double)q_lds[(g * 4 + u) * DM_DC + c] - xv;
        acc[u] = fma(diff, diff, acc[u]);
      }
    }
  }
  if (xl0 + j < x_count) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int64_t qi = q0 + g * 4 + u;
      if (qi < nq) D2[qi * x_count + xl0 + j] = acc[u];
    }
  }
}

constexpr int TK_CAP = 2048;  // LDS sort capacity (entries)

__device__ __forceinline__ bool key_less(unsigned long long da, int ia, unsigned long long db,
                                         int ib) {
  return da < db || (da == db && ia < ib);
}

// One block per query: merge the carried top-k with a slab of squared distances by a bitonic sort
// of (distance bits, index) in LDS.  carry_* hold k entries per query (distance = +inf when unset).
__global__ __launch_bounds__(256) void k_row_topk(const double* __restrict__ D2, int64_t x_count,
                                                  int64_t x_begin, int k,
                                                  unsigned long long* carry_d, int* carry_i) {
  __shared__ unsigned long long sd[TK_CAP];
  __shared__ int si[TK_CAP];
  const int tid = threadIdx.x;
  const int64_t qi = blockIdx.x;
  const unsigned long long INF_BITS = 0x7ff0000000000000ULL;
  for (int64_t base = 0; base < x_count; base += TK_CAP - k) {
    const int cnt = (int)min((int64_t)(TK_CAP - k), x_count - base);
    __syncthreads();
    for (int i = tid; i < TK_CAP; i += 256) {
      unsigned long long dv = INF_BITS;
      int iv = 0x7fffffff;
      if (i < k) {
        dv = carry_d[qi * k + i];
        iv = carry_i[qi * k + i];
      } else if (i - k < cnt) {
        dv = (unsigned long long)__double_as_longlong(D2[qi * x_count + base + (i - k)]);
        iv = (int)(x_begin + base + (i - k));
      }
      sd[i] = dv;
      si[i] = iv;
    }
    __syncthreads();
    for (int size = 2; size <= TK_CAP; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < TK_CAP / 2; t += 256) {
          int lo = (t / stride) * stride * 2 + (t % stride);
          int hi = lo + stride;
          bool up = ((lo & size) == 0);
          unsigned long long dl = sd[lo], dh = sd[hi];
          int il = si[lo], ih = si[hi];
          bool swap = up ? key_less(dh, ih, dl, il) : key_less(dl, il, dh, ih);
          if (swap) {
            sd[lo] = dh;
            si[lo] = ih;
            sd[hi] = dl;
            si[hi] = il;
          }
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < k; i += 256) {
      carry_d[qi * k + i] = sd[i];
      carry_i[qi * k + i] = si[i];
    }
  }
}

__global__ void k_topk_init(unsigned long long* carry_d, int* carry_i, int64_t n) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) {
    carry_d[t] = 0x7ff0000000000000ULL;
    carry_i[t] = 0x7fffffff;
  }
}
// squared: write the squared distances the ranking was made on (cs_l2_topk_sq: shard merges compare exactly
// what the kernels compared; two different squares can share one rounded square root)
// (an explicit argument of every internal top-k function: no side channel between the entry points)
__global__ void k_topk_finish(const unsigned long long* carry_d, const int* carry_i, int64_t n,
                              int64_t* idx, double* dist, int squared) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) {
    int i = carry_i[t];
    idx[t] = i == 0x7fffffff ? -1 : (int64_t)i;
    const double d2 = __longlong_as_double((long long)carry_d[t]);
    if (dist) dist[t] = squared ? d2 : sqrt(d2);
  }
}

// ------------------------------------------------------------------------------------------
// Large-scale descriptor top-k (BASELINE config C5: 10^6 x 10^6 x 256-d, top-10) on the f64 matrix
// pipe.  d^2 = |q|^2 + |x|^2 - 2 q.x with the dot products on v_mfma_f64_16x16x4_f64 (absolute error
// ~1e-15 for unit descriptors) is used ONLY to shortlist: every lane keeps the TKM_KK best rows of
// the catalog rows it sees for its query; the shortlist (k + margin per lane) is then re-scored with
// the canonical chain sum_c (q_c - x_c)^2 and ranked by (distance, index), so the returned ids and
// distances are those of the exact kernel unless more than `margin` catalog rows tie with the k-th
// neighbour to within ~2e-15 -- the same caveat cdist itself has.
// ------------------------------------------------------------------------------------------
constexpr int TKM_QT = 64;     // queries per block (16 per wave)
constexpr int TKM_XT = 64;     // catalog rows per tile
constexpr int TKM_DC = 64;     // feature chunk in LDS
constexpr int TKM_PITCH = TKM_DC + 2;  // doubles; 2*(pitch) = 4 mod 64 banks: conflict-free b64 reads
constexpr int TKM_KK = 12;     // shortlist per lane (k <= TKM_KK - 2)

__global__ void k_row_norms(const float* __restrict__ X, int64_t n, int d, double* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int c = 0; c < d; ++c) {
    const double v = (double)X[i * d + c];
    s = fma(v, v, s);
  }
  out[i] = s;
}

// grid: x = query tile, y = catalog split.  cand_*: [nq][nsplit*4][TKM_KK]
__global__ __launch_bounds__(256) void k_topk_mfma(const float* __restrict__ Q, int64_t nq,
                                                   const float* __restrict__ X, int64_t nx, int d,
                                                   const double* __restrict__ qn,
                                                   const double* __restrict__ xn, int nsplit,
                                                   double* __restrict__ cand_d,
                                                   int* __restrict__ cand_i) {
  __shared__ double q_lds[TKM_QT * TKM_PITCH];
  __shared__ double x_lds[TKM_XT * TKM_PITCH];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 15;   // query within the wave's 16, and catalog row within an MFMA tile (A side)
  const int kq = lane >> 4;    // k slot of the operands, and row group of the results
  const int64_t q0 = (int64_t)blockIdx.x * TKM_QT;
  const int64_t per = ((nx + nsplit - 1) / nsplit + TKM_XT - 1) / TKM_XT * TKM_XT;
  const int64_t xb = (int64_t)blockIdx.y * per;
  const int64_t xe = min(nx, xb + per);
  const int64_t my_q = q0 + wave * 16 + col;
  const double my_qn = my_q < nq ? qn[my_q] : 0.0;

  double bd[TKM_KK];
  int bi[TKM_KK];
#pragma unroll
  for (int j = 0; j < TKM_KK; ++j) {
    bd[j] = INFINITY;
    bi[j] = 0x7fffffff;
  }

  for (int64_t x0 = xb; x0 < xe; x0 += TKM_XT) {
    f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < d; c0 += TKM_DC) {
      const int dc = min(TKM_DC, d - c0);
      __syncthreads();
      for (int i = tid; i < TKM_QT * TKM_DC; i += 256) {
        const int r = i / TKM_DC, c = i - r * TKM_DC;
        double v = 0.0;
        if (q0 + r < nq && c < dc) v = (double)Q[(q0 + r) * d + c0 + c];
        q_lds[r * TKM_PITCH + c] = v;
      }
      for (int i = tid; i < TKM_XT * TKM_DC; i += 256) {
        const int r = i / TKM_DC, c = i - r * TKM_DC;
        double v = 0.0;
        if (x0 + r < xe && c < dc) v = (double)X[(x0 + r) * d + c0 + c];
        x_lds[r * TKM_PITCH + c] = v;
      }
      __syncthreads();
      const double* qp = q_lds + (wave * 16 + col) * TKM_PITCH + kq;
      const double* xp = x_lds + col * TKM_PITCH + kq;
#pragma unroll 4
      for (int k4 = 0; k4 < TKM_DC; k4 += 4) {
        const double b = qp[k4];                       // B[k][query]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double a = xp[t * 16 * TKM_PITCH + k4];  // A[catalog row][k]
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    // results: acc[t][r] = dot(catalog row x0 + 16 t + kq + 4 r, query my_q)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = x0 + 16 * t + kq + 4 * r;
        double dist = INFINITY;
        if (row < xe) dist = fma(-2.0, acc[t][r], my_qn + xn[row]);
        if (dist < bd[TKM_KK - 1]) {
          double cd = dist;
          int ci = (int)row;
          bool carry = false;
#pragma unroll
          for (int s2 = 0; s2 < TKM_KK; ++s2) {
            if (carry || cd < bd[s2]) {
              carry = true;
              const double td = bd[s2];
              const int ti = bi[s2];
              bd[s2] = cd;
              bi[s2] = ci;
              cd = td;
              ci = ti;
            }
          }
        }
      }
    }
  }
  if (my_q < nq) {
    const int64_t base = (my_q * (nsplit * 4) + blockIdx.y * 4 + kq) * TKM_KK;
#pragma unroll
    for (int j = 0; j < TKM_KK; ++j) {
      cand_d[base + j] = bd[j];
      cand_i[base + j] = bi[j];
    }
  }
}

// Re-score the shortlist of one query with the canonical chain and keep the k best (dist, idx).
// One workgroup per query, one THREAD per candidate (the f64 chain of a candidate is sequential in the feature index:
// that order is the parity contract), 256 candidates at a time.  Round 5: the candidates' rows come through LDS in
// chunks of TKR_CH features -- 32 consecutive lanes copy one row's 128-byte piece, so a wave's load touches two cache
// lines, and every thread then walks ITS row in LDS (pitch TKR_CH + 1: conflict-free) -- and the ranking sorts the next
// power of two above the candidate count.  Before, every thread read its own row from global memory (64 cache lines per
// wave-load, 256 loads per thread) and the bitonic network always ran over TKM_MERGE_CAP = 1 024 slots: 2.97 ms per
// 10 240 x 144 candidates of 256-d, as much as a fifth of the shortlist kernel it follows.
constexpr int TKM_MERGE_CAP = 1024;
constexpr int TKR_CH = 32;
__global__ __launch_bounds__(256) void k_topk_rescore(const float* __restrict__ Q,
                                                      const float* __restrict__ X, int d,
                                                      const int* __restrict__ cand_i, int ncand,
                                                      int k, unsigned long long* carry_d,
                                                      int* carry_i) {
  __shared__ unsigned long long sd[TKM_MERGE_CAP];
  __shared__ int si[TKM_MERGE_CAP];
  __shared__ float rows[256][TKR_CH + 1];
  __shared__ float qs[TKR_CH];
  __shared__ int srow[256];
  const int tid = threadIdx.x;
  const int64_t qi = blockIdx.x;
  const unsigned long long INF_BITS = 0x7ff0000000000000ULL;
  int P = 2;
  while (P < ncand || P < k) P <<= 1;
  if (P > TKM_MERGE_CAP) P = TKM_MERGE_CAP;   // (the callers keep ncand <= TKM_MERGE_CAP)
  for (int gb = 0; gb < P; gb += 256) {
    const int i = gb + tid;
    const int row = i < ncand ? cand_i[qi * ncand + i] : 0x7fffffff;
    __syncthreads();   // the previous group has finished reading rows / srow
    srow[tid] = row;
    double acc = 0.0;
    const int n_rows = min(256, ncand - gb);   // candidates of this group (block-uniform, may be <= 0)
    for (int c0 = 0; c0 < d && n_rows > 0; c0 += TKR_CH) {
      const int cl = min(TKR_CH, d - c0);
      __syncthreads();   // srow visible; the previous chunk has been consumed
      {
        // thread (r0, c) copies column c of the rows r0, r0 + 8, ...: all of its loads are issued before the first one is
        // waited for (unconditional, clamped addresses -- a load under a condition is waited for at once)
        static_assert(TKR_CH == 32, "one 32-lane half-wave per row piece");
        const int c = tid & 31, r0 = tid >> 5;
        const int cc = c0 + (c < cl ? c : 0);
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int rr = srow[r0 + 8 * j];
          v[j] = X[(int64_t)(rr != 0x7fffffff ? rr : 0) * d + cc];
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) rows[r0 + 8 * j][c] = v[j];
      }
      if (tid < cl) qs[tid] = Q[qi * d + c0 + tid];
      __syncthreads();
      if (row != 0x7fffffff) {
        for (int c = 0; c < cl; ++c) {
          const double diff = (double)qs[c] - (double)rows[tid][c];
          acc = fma(diff, diff, acc);
        }
      }
    }
    if (i < TKM_MERGE_CAP) {
      sd[i] = row != 0x7fffffff ? (unsigned long long)__double_as_longlong(acc) : INF_BITS;
      si[i] = row;
    }
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < P / 2; t += 256) {
        const int lo = (t / stride) * stride * 2 + (t % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned long long dl = sd[lo], dh = sd[hi];
        const int il = si[lo], ih = si[hi];
        const bool swap = up ? key_less(dh, ih, dl, il) : key_less(dl, il, dh, ih);
        if (swap) {
          sd[lo] = dh;
          si[lo] = ih;
          sd[hi] = dl;
          si[hi] = il;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += 256) {
    carry_d[qi * k + i] = sd[i];
    carry_i[qi * k + i] = si[i];
  }
}


// ------------------------------------------------------------------------------------------
// The same shortlist on the f16 matrix cores (d = 64 / 128 / 256 / 512, k <= 10): 16x the f64 matrix rate.
// s~ = |x|^2 - 2 q.x with x = x_hi + x_lo and v = -2 q = v_hi + v_lo split into f16 (the lo * lo
// products are dropped): 3 v_mfma_f32_32x32x16_f16 per 16 features of a 32 x 32 (catalog x query) tile,
// accumulator preloaded with |x|^2.  Queries live in registers (32 per wave, 8 waves per workgroup:
// 256 queries reuse every staged catalog row), the catalog streams through LDS as a padded f16 image
// (row = [hi(16) | lo(16)] per 16 features + 16 B: odd 16-B slot pitch, conflict-free ds_read_b128),
// 64 rows per stage, copied by LDS-DMA one stage ahead.  A lane keeps the TKF_KK best rows of its half
// of the tile rows.  Exactness: the shortlist is re-scored with the canonical f64 chain (k_topk_rescore)
// and VERIFIED -- every row outside the shortlist has s~ >= tau (the smallest of the lanes' TKF_KK-th
// values), hence exact s >= tau - eps with eps = (3 d + 4) 2^-22 (|q|^2 + max |x|^2) bounding the f16
// pipeline error (split residuals 3 * 2^-21 |q||x|, 3 d + 1 f32 accumulations of partial sums
// <= |x|^2 + 2 |q||x|) plus an ABSOLUTE term 2^-24 sqrt(d) (2|q| + max|x|) for elements whose hi or lo part
// is an f16 subnormal (spacing 2^-24 whatever the magnitude: descriptors with norms << 1);
// queries whose exact k-th value is not below that go through the f64 path.
// d = 512 (TERMS = 2): the 32 queries of a wave would need 256 VGPRs for hi + lo operands, so only v_hi is
// kept in registers (128 VGPRs) and the x_hi v_lo term is dropped as well -- 2 MFMAs per 16 features, and
// eps grows by its bound 1.001 * 2^-10 |q| max|x| (about 1e-3 for unit descriptors, against a gap of ~0.1
// between the k-th neighbour and the shortlist threshold on the C5 data); 32-row stages (2 x 64.5 KiB of LDS).
// ------------------------------------------------------------------------------------------
constexpr int TKF_ROWS = 64;   // catalog rows per LDS stage (2 MFMA row tiles); the image is padded to whole 64s
constexpr int TKF_QT = 256;    // queries per workgroup (8 waves x 32)
constexpr int TKF_KK = 12;     // shortlist per lane (two lanes per query and catalog split)
constexpr int TKF_AHEAD = 2;   // MFMA steps by which the LDS reads of the A fragments run ahead

// f16 image of n rows (+ zero rows up to n_pad): per 16 features [hi(16) | lo(16)] of scale * X
__global__ void k_tkf_pack(const float* __restrict__ X, int64_t n, int64_t n_pad, int d, float scale,
                           _Float16* __restrict__ img, int pitch_h) {
  const int dch = d / 16;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n_pad * dch) return;
  const int64_t row = t / dch;
  const int c = (int)(t - row * dch);
  union {
    _Float16 h[32];
    uint4 v[4];
  } u;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double v = row < n ? (double)scale * (double)X[row * d + c * 16 + j] : 0.0;
    const _Float16 hi = (_Float16)v;
    u.h[j] = hi;
    u.h[16 + j] = (_Float16)(v - (double)hi);
  }
  uint4* dst = reinterpret_cast<uint4*>(img + row * pitch_h + c * 32);
#pragma unroll
  for (int j = 0; j < 4; ++j) dst[j] = u.v[j];
  if (c == 0 && pitch_h > dch * 32) {  // the 16-B pad of a catalog row (never read by the MFMA fragments; keep it defined)
    uint4* pad = reinterpret_cast<uint4*>(img + row * pitch_h + dch * 32);
    *pad = make_uint4(0u, 0u, 0u, 0u);
  }
}

__global__ void k_max_bits(const double* __restrict__ v, int64_t n, unsigned* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  float m = i < n ? __double2float_ru(v[i]) : 0.f;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));  // non-negative floats order like uints
}

// work item = (catalog split, query tile of TKF_QT); 512 threads; dynamic LDS: 2 stages + 2 x TKF_ROWS floats.
// cand_i: [nq][nsplit * 2][TKF_KK], tau: [nq][nsplit * 2]
template <int DCH, int TERMS, int ROWS>
__global__ __launch_bounds__(512) void k_topk_f16(const _Float16* __restrict__ qimg, int64_t nq,
                                                  const _Float16* __restrict__ ximg, int64_t nx,
                                                  const double* __restrict__ xn, int nsplit, int qtiles, int xcd_deal,
                                                  int* __restrict__ cand_i, float* __restrict__ tau) {
  // 1-D grid of 8 * ceil(qtiles * nsplit / 8) workgroups.  The hardware deals workgroup b to XCD b % 8: the work items
  // (catalog split, query tile) -- query tile fastest -- are cut into 8 contiguous runs, one per XCD, so that the
  // workgroups that share an XCD's 4-MiB L2 stream the SAME catalog split at about the same pace (each split's image
  // is then read from HBM / Infinity Cache about once per XCD that works on it instead of once per workgroup: at
  // 10 240 x 10^6 x 256-d, 40 query tiles x 6 splits, a workgroup needs 10.6 B per cycle at the full matrix rate, 6 TB/s
  // over the chip).  xcd_deal = 0 (CS_TOPK_XCD=0): item = workgroup id, the mapping of rounds 2-4.
  const int n_items = qtiles * nsplit;
  const int per_xcd = (n_items + 7) / 8;
  const int item = xcd_deal ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (item >= n_items) return;
  const int split = item / qtiles;
  const int qtile = item - split * qtiles;
  constexpr int PITCH_B = DCH * 64 + 16;              // bytes per image row
  constexpr int STAGE_BYTES = ROWS * PITCH_B;         // whole KiB for DCH = 4, 8, 16; 64.5 KiB for DCH = 32
  constexpr int STAGE_PIECES = (STAGE_BYTES + 1023) / 1024;   // 1-KiB LDS-DMA instructions (the last may be partial)
  constexpr int STAGE_PITCH = STAGE_PIECES * 1024;
  constexpr int NT = ROWS / 32;                       // MFMA row tiles per stage
  static_assert(STAGE_BYTES % 16 == 0 && ROWS % 32 == 0 && TKF_ROWS % ROWS == 0, "stage shape");
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  float* tn_s = reinterpret_cast<float*>(lds + 2 * STAGE_PITCH);  // [2][ROWS]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int col = lane & 31;
  const int64_t my_q = (int64_t)qtile * TKF_QT + wave * 32 + col;
  const int64_t per = ((nx + nsplit - 1) / nsplit + TKF_ROWS - 1) / TKF_ROWS * TKF_ROWS;
  const int64_t xb = (int64_t)split * per;
  const int64_t xe = min(nx, xb + per);
  // B operands of the wave's 32 queries: lane supplies k = 8 half .. +8 of every 16-feature chunk
  f16x8 qh[DCH], ql[TERMS == 3 ? DCH : 1];
  {
    const _Float16* qrow = qimg + (my_q < nq ? my_q : 0) * (int64_t)(DCH * 32) + 8 * half;
#pragma unroll
    for (int c = 0; c < DCH; ++c) {
      qh[c] = *reinterpret_cast<const f16x8*>(qrow + c * 32);
      if (TERMS == 3) ql[c] = *reinterpret_cast<const f16x8*>(qrow + c * 32 + 16);
    }
  }
  float bd[TKF_KK];
  int bi[TKF_KK];
#pragma unroll
  for (int j = 0; j < TKF_KK; ++j) {
    bd[j] = INFINITY;
    bi[j] = 0x7fffffff;
  }
  const char* gimg = reinterpret_cast<const char*>(ximg) + lane * 16;
  // A stage = the catalog image (LDS-DMA, written as inline asm: behind the builtin form hipcc waits vmcnt(0)
  // in front of the first ds_read that follows, i.e. for the stage it has just issued "one ahead" -- lds_dma16,
  // common.h) + the |x|^2 of its rows (a plain load, issued BEFORE the DMAs and stored to LDS AFTER the current
  // stage is computed: hipcc's wait for it is then the only wait in the loop body and the DMAs have landed by then).
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  auto issue_dma = [&](int b, int64_t base) {
    const char* gp = gimg + base * PITCH_B;
#pragma unroll
    for (int i = 0; i < (STAGE_PIECES + 7) / 8; ++i) {
      const int piece = wave + 8 * i;  // wave-uniform
      // (the last piece of a 64.5-KiB stage is half a KiB: the upper lanes sit it out)
      if (piece < STAGE_PIECES && piece * 1024 + lane * 16 < STAGE_BYTES)
        lds_dma16(gp + piece * 1024, lds_base + b * STAGE_PITCH + piece * 1024);
    }
  };
  auto load_norm = [&](int64_t base) {   // unconditional (clamped) load: a load under a condition is waited for at once
    int64_t r = base + (tid % ROWS);
    if (r > xe - 1) r = xe - 1;
    return (float)xn[r < xb ? xb : r];
  };
  auto store_norm = [&](int b, int64_t base, float v) {
    if (tid < ROWS) tn_s[b * ROWS + tid] = base + tid < xe ? v : INFINITY;
  };
  if (xb < xe) {
    const float v0 = load_norm(xb);
    issue_dma(0, xb);
    store_norm(0, xb, v0);
  }
  int buf = 0;
  for (int64_t base = xb; base < xe; base += ROWS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more = base + ROWS < xe;
    const float v_next = load_norm(base + ROWS);
    if (more) issue_dma(buf ^ 1, base + ROWS);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      // accumulator input: |x|^2 of the 16 rows this lane owns: (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 v = *reinterpret_cast<const float4*>(&tn_s[buf * ROWS + t * 32 + 8 * q4 + 4 * half]);
        acc[t][4 * q4 + 0] = v.x; acc[t][4 * q4 + 1] = v.y; acc[t][4 * q4 + 2] = v.z; acc[t][4 * q4 + 3] = v.w;
      }
    }
    const char* st = lds + buf * STAGE_PITCH + col * PITCH_B + half * 16;
    // The A fragments of step i + TKF_AHEAD are requested BEFORE the MFMAs of step i are issued (a ring of register
    // pairs): left to itself hipcc asks for a step's two ds_read_b128 right in front of its MFMAs and waits for them
    // at once -- the whole LDS latency once per 3 MFMAs, with two waves per SIMD the matrix pipe sat at 0.37.
    constexpr int NSTEP = DCH * NT;
    f16x8 rah[TKF_AHEAD + 1], ral[TKF_AHEAD + 1];
    auto fetch = [&](int i) {
      const int c = i / NT, t = i % NT;
      rah[i % (TKF_AHEAD + 1)] = *reinterpret_cast<const f16x8*>(st + t * 32 * PITCH_B + c * 64);
      ral[i % (TKF_AHEAD + 1)] = *reinterpret_cast<const f16x8*>(st + t * 32 * PITCH_B + c * 64 + 32);
    };
#pragma unroll
    for (int i = 0; i < TKF_AHEAD; ++i) fetch(i);
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
      if (i + TKF_AHEAD < NSTEP) fetch(i + TKF_AHEAD);
      __builtin_amdgcn_sched_barrier(0);   // keep the request in front of this step's MFMAs
      const int c = i / NT, t = i % NT;
      const f16x8 ah = rah[i % (TKF_AHEAD + 1)], al = ral[i % (TKF_AHEAD + 1)];
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[c], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[c], acc[t], 0, 0, 0);
      if (TERMS == 3) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[c], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // One test per 32 x 32 tile (v_min3 tree over the lane's 16 results against its KK-th best) instead of one per result:
    // late in the scan a wave finds a candidate in about one tile of four.  The insertion itself is branch-free
    // (one v_med3 per slot for the sorted values, two selects for the rows): written as the usual compare-and-carry loop
    // hipcc copied the whole list at every slot, ~1 000 cycles per event with 2.6 events per wave and stage -- together
    // with the 32 compare-and-branch sequences more than the stage's MFMAs.
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float m = acc[t][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = fminf(m, acc[t][r]);
      if (m < bd[TKF_KK - 1]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][r];
          if (v < bd[TKF_KK - 1]) {
            const int ci = (int)(base + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half);
            bool lt_cur = true;   // v < bd[KK - 1]
#pragma unroll
            for (int j = TKF_KK - 1; j >= 1; --j) {
              const bool lt_prev = v < bd[j - 1];
              bi[j] = lt_prev ? bi[j - 1] : (lt_cur ? ci : bi[j]);
              bd[j] = __builtin_amdgcn_fmed3f(bd[j - 1], v, bd[j]);
              lt_cur = lt_prev;
            }
            bi[0] = lt_cur ? ci : bi[0];
            bd[0] = fminf(bd[0], v);
          }
        }
      }
    }
    if (more) store_norm(buf ^ 1, base + ROWS, v_next);
    buf ^= 1;
  }
  if (my_q < nq) {
    const int64_t slot = my_q * (nsplit * 2) + split * 2 + half;
#pragma unroll
    for (int j = 0; j < TKF_KK; ++j) cand_i[slot * TKF_KK + j] = bi[j];
    tau[slot] = bd[TKF_KK - 1];
  }
}

template <int DCH, int TERMS, int ROWS>
static int launch_topk_f16(int qtiles, hipStream_t s, const _Float16* qimg, int64_t nq, const _Float16* ximg,
                           int64_t nx, const double* xn, int nsplit, int* cand_i, float* tau) {
  constexpr int LDS_BYTES = 2 * ((ROWS * (DCH * 64 + 16) + 1023) / 1024 * 1024) + 2 * ROWS * (int)sizeof(float);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_topk_f16<DCH, TERMS, ROWS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  CS_REQUIRE(attr == hipSuccess, CS_ERR_HIP, "cs_l2_topk: cannot reserve %d bytes of LDS", LDS_BYTES);
  const char* e = getenv("CS_TOPK_XCD");
  const int xcd_deal = !(e && e[0] == '0');
  const int n_items = qtiles * nsplit;
  const dim3 grid((unsigned)(xcd_deal ? 8 * ((n_items + 7) / 8) : n_items));
  hipLaunchKernelGGL((k_topk_f16<DCH, TERMS, ROWS>), grid, dim3(512), LDS_BYTES, s, qimg, nq, ximg, nx, xn, nsplit,
                     qtiles, xcd_deal, cand_i, tau);
  return CS_OK;
}

// One thread per query: is the re-scored k-th value provably below everything outside the shortlist?
// flagged queries (compact list + count) go through the f64 path.
__global__ void k_tkf_verify(const unsigned long long* __restrict__ carry_d, int k, int64_t nq,
                             const float* __restrict__ tau, int nlane, const double* __restrict__ qn,
                             const unsigned* __restrict__ xmax_bits, int d, int terms,
                             int* __restrict__ flagged, int* __restrict__ n_flagged) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= nq) return;
  float t = INFINITY;
  for (int j = 0; j < nlane; ++j) t = fminf(t, tau[q * nlane + j]);
  const double dk = __longlong_as_double((long long)carry_d[q * k + k - 1]);   // exact k-th squared distance
  const double xm = (double)__uint_as_float(*xmax_bits);   // max |x|^2 (rounded up)
  double eps = (double)(3 * d + 4) * 2.384185791015625e-07 * (qn[q] + xm)                    // relative: 2^-22
               + 5.9604644775390625e-08 * sqrt((double)d) * (2.0 * sqrt(qn[q]) + sqrt(xm));  // absolute: 2^-24
  if (terms < 3) eps += 1.001 * 9.765625e-04 * sqrt(qn[q]) * sqrt(xm);                      // dropped x_hi v_lo: 2^-10 |q||x|
  // f16 range: |x| and |-2 q| below 2e4 keep every hi part (and every product sum) finite; anything larger
  // (or not finite) cannot be trusted and takes the f64 path
  const bool in_range = __uint_as_float(*xmax_bits) < 4.0e8f && qn[q] < 1.0e8;
  const bool ok = in_range && dk - qn[q] < (double)t - eps;   // false for NaN / missing candidates
  if (!ok) flagged[atomicAdd(n_flagged, 1)] = (int)q;
}

__global__ void k_gather_rows(const float* __restrict__ X, int d, const int* __restrict__ rows, int64_t n,
                              float* __restrict__ out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * d) return;
  const int64_t r = t / d;
  out[t] = X[(int64_t)rows[r] * d + (t - r * d)];
}

__global__ void k_scatter_topk(const int64_t* __restrict__ idx, const double* __restrict__ dist, int k,
                               const int* __restrict__ rows, int64_t n, int64_t* __restrict__ out_idx,
                               double* __restrict__ out_dist) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * k) return;
  const int64_t r = t / k;
  const int64_t o = (int64_t)rows[r] * k + (t - r * k);
  out_idx[o] = idx[t];
  if (out_dist) out_dist[o] = dist[t];
}

// ------------------------------------------------------------------------------------------
// one-directional Chamfer
// ------------------------------------------------------------------------------------------
struct ChamferWork {
  int64_t s0;   // first source row of this tile (global)
  int64_t t0;   // first target row (global)
  int32_t sn;   // source rows in this tile (<= 256)
  int32_t tn;   // target rows
  int32_t prob;
  int32_t slot; // index into the partial-sum array
};

constexpr int CH_TT = 512;

__global__ __launch_bounds__(256) void k_chamfer(const ChamferWork* __restrict__ work,
                                                 const float* __restrict__ src,
                                                 const float* __restrict__ tgt,
                                                 const float* __restrict__ T, int reduce_max,
                                                 double* __restrict__ partial) {
  __shared__ float t_lds[CH_TT * 3];
  __shared__ double red[256];
  const ChamferWork wk = work[blockIdx.x];
  const int tid = threadIdx.x;
  const bool active = tid < wk.sn;
  const float* Tp = T + (int64_t)wk.prob * 16;
  double px = 0, py = 0, pz = 0;
  if (active) {
    const float* s = src + (wk.s0 + tid) * 3;
    double x = s[0], y = s[1], z = s[2];
    // row-major 4x4: p = R x + t, evaluated as fma(r0,x, fma(r1,y, fma(r2,z, t)))
    px = fma((double)Tp[0], x, fma((double)Tp[1], y, fma((double)Tp[2], z, (double)Tp[3])));
    py = fma((double)Tp[4], x, fma((double)Tp[5], y, fma((double)Tp[6], z, (double)Tp[7])));
    pz = fma((double)Tp[8], x, fma((double)Tp[9], y, fma((double)Tp[10], z, (double)Tp[11])));
  }
  double best = INFINITY;
  for (int tbase = 0; tbase < wk.tn; tbase += CH_TT) {
    const int tcount = min(CH_TT, wk.tn - tbase);
    __syncthreads();
    for (int i = tid; i < tcount * 3; i += 256) t_lds[i] = tgt[(wk.t0 + tbase) * 3 + i];
    __syncthreads();
    if (!active) continue;
    for (int j = 0; j < tcount; ++j) {
      double dx = px - (double)t_lds[3 * j + 0];
      double dy = py - (double)t_lds[3 * j + 1];
      double dz = pz - (double)t_lds[3 * j + 2];
      double d = fma(dz, dz, fma(dy, dy, dx * dx));
      best = d < best ? d : best;
    }
  }
  red[tid] = active ? sqrt(best) : 0.0;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] = reduce_max ? fmax(red[tid], red[tid + off]) : red[tid] + red[tid + off];
    __syncthreads();
  }
  if (tid == 0) partial[wk.slot] = red[0];
}

// The same search on the f64 matrix pipe: A = [tx, ty, tz, |t|^2] (targets, LDS), B = [-2px, -2py, -2pz, 1]
// (transformed sources, registers): one v_mfma_f64_16x16x4_f64 gives |t|^2 - 2 p.t for 16 x 16 pairs,
// the ranking value of each source.  Every lane keeps the arg-min of its quarter of the targets, the
// four candidates of a source are re-evaluated with the canonical chain and the smallest is the
// result -- equal to the exhaustive chain unless two targets are within the expansion's rounding
// (~5e-16 absolute in d^2) of the minimum, where the two values differ by less than that.
constexpr int CHM_NG = 4;               // 16-source groups per wave: every staged target row serves 256 sources
constexpr int CHM_ST = 64 * CHM_NG;     // sources per workgroup
constexpr int CHM_TT = 512;             // targets per LDS stage
// A-operand rows of every target, once per call: (x, y, z, |t|^2) in f64
__global__ void k_chamfer_pack(const float* __restrict__ tgt, int64_t n, double* __restrict__ t4g) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = tgt[3 * i], y = tgt[3 * i + 1], z = tgt[3 * i + 2];
  double* o = t4g + 4 * i;
  o[0] = x;
  o[1] = y;
  o[2] = z;
  o[3] = fma(z, z, fma(y, y, x * x));
}

__global__ __launch_bounds__(256) void k_chamfer_mfma(const ChamferWork* __restrict__ work,
                                                      const float* __restrict__ src,
                                                      const float* __restrict__ tgt,
                                                      const double* __restrict__ t4g,
                                                      const float* __restrict__ T, int reduce_max,
                                                      double* __restrict__ partial,
                                                      // != nullptr: only the tiles the f16 kernel could not vouch for
                                                      const int32_t* __restrict__ only_flagged) {
  __shared__ __attribute__((aligned(16))) double t4[CHM_TT * 4];
  __shared__ double red[CHM_ST];
  if (only_flagged && !only_flagged[blockIdx.x]) return;
  const ChamferWork wk = work[blockIdx.x];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 15;  // source within a group (B side); target row within a 16-row tile (A side)
  const int kq = lane >> 4;   // k slot of the operands; row group of the results
  const float* Tp = T + (int64_t)wk.prob * 16;
  double px[CHM_NG], py[CHM_NG], pz[CHM_NG], bq[CHM_NG], best[CHM_NG];
  int bidx[CHM_NG];
#pragma unroll
  for (int g = 0; g < CHM_NG; ++g) {
    const int sloc = (wave * CHM_NG + g) * 16 + col;
    const float* sp = src + (wk.s0 + (sloc < wk.sn ? sloc : 0)) * 3;
    const double x = sp[0], y = sp[1], z = sp[2];
    px[g] = fma((double)Tp[0], x, fma((double)Tp[1], y, fma((double)Tp[2], z, (double)Tp[3])));
    py[g] = fma((double)Tp[4], x, fma((double)Tp[5], y, fma((double)Tp[6], z, (double)Tp[7])));
    pz[g] = fma((double)Tp[8], x, fma((double)Tp[9], y, fma((double)Tp[10], z, (double)Tp[11])));
    bq[g] = kq == 0 ? -2.0 * px[g] : (kq == 1 ? -2.0 * py[g] : (kq == 2 ? -2.0 * pz[g] : 1.0));
    best[g] = INFINITY;
    bidx[g] = -1;
  }
  for (int tbase = 0; tbase < wk.tn; tbase += CHM_TT) {
    const int tcount = min(CHM_TT, wk.tn - tbase);
    __syncthreads();
    for (int j = tid; j < CHM_TT; j += 256) {
      double2 xy = make_double2(0.0, 0.0), zn = make_double2(0.0, INFINITY);  // rows past the segment never win
      if (j < tcount) {
        const double2* tp = reinterpret_cast<const double2*>(t4g + (wk.t0 + tbase + j) * 4);
        xy = tp[0];
        zn = tp[1];
      }
      *reinterpret_cast<double2*>(&t4[4 * j]) = xy;
      *reinterpret_cast<double2*>(&t4[4 * j + 2]) = zn;
    }
    __syncthreads();
    for (int t = 0; t < (tcount + 15) / 16; ++t) {
      const double a = t4[(16 * t + col) * 4 + kq];
#pragma unroll
      for (int g = 0; g < CHM_NG; ++g) {
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[g], acc, 0, 0, 0);
        // acc[r] = |t|^2 - 2 p.t of target row 16 t + kq + 4 r and source col of group g
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (acc[r] < best[g]) {
            best[g] = acc[r];
            bidx[g] = tbase + 16 * t + kq + 4 * r;
          }
        }
      }
    }
  }
  // canonical distance of this lane's candidate, then the smallest of the source's four lanes
#pragma unroll
  for (int g = 0; g < CHM_NG; ++g) {
    double d = INFINITY;
    if (bidx[g] >= 0) {
      const float* tp = tgt + (wk.t0 + bidx[g]) * 3;
      const double dx = px[g] - (double)tp[0], dy = py[g] - (double)tp[1], dz = pz[g] - (double)tp[2];
      d = fma(dz, dz, fma(dy, dy, dx * dx));
    }
    d = fmin(d, __shfl_xor(d, 16));
    d = fmin(d, __shfl_xor(d, 32));
    const int sloc = (wave * CHM_NG + g) * 16 + col;
    if (kq == 0) red[sloc] = sloc < wk.sn ? sqrt(d) : 0.0;
  }
  __syncthreads();
  for (int off = CHM_ST / 2; off > 0; off >>= 1) {
    if (tid < off) red[tid] = reduce_max ? fmax(red[tid], red[tid + off]) : red[tid] + red[tid + off];
    __syncthreads();
  }
  if (tid == 0) partial[wk.slot] = red[0];
}

// ------------------------------------------------------------------------------------------
// The same search on the f16 matrix cores (round 5).  The f64 matrix pipe of gfx950 is the vector unit's double-precision
// ALU (78.6 TF either way): k_chamfer_mfma sits at 0.23 of it with five VALU operations per result on top.  The ranking
// value |t|^2 - 2 p.t needs nine products when the coordinates are cut into f16 hi + lo (ph.th + ph.tl + pl.th): ONE
// v_mfma_f32_32x32x16_f16 per 32 targets x 32 sources with |t|^2 in the accumulator input, 16x the rate per pair.  It only
// RANKS: a lane (one source, 16 of a tile's 32 target rows) keeps the two best TILE minima and the third best value (eight
// v_min3 per tile, no index per value); at the end the rows of its two tiles are evaluated with the canonical f64 chain,
// and the result is vouched for when it lies below the third value minus the error budget of the approximation -- then no
// row outside the evaluated tiles can be nearer.  A workgroup with a source that fails the test is recomputed by
// k_chamfer_mfma (flag array, no host decision).  Coordinates are scaled by 2^9 before the cut so that the lo parts stay
// normal f16 numbers; |coordinate| >= 60 does not fit and takes the f64 kernel as well.  Results: the same canonical
// distances as the other two kernels (tests/test_gpu_post.py::test_chamfer_matches_oracle, all three paths).
constexpr int CHF_PITCH = 24;     // halfs per image row (48 B: conflict-free ds_read_b128 fragments)
constexpr int CHF_ROWS = 256;     // target rows per LDS stage (8 MFMA row tiles)
constexpr int CHF_NG = 2;         // 32-source groups per wave
constexpr float CHF_SCALE = 512.0f;
static_assert(4 * 32 * CHF_NG == CHM_ST, "the f16 kernel and its f64 fallback share the work items");

// target image: row j = [th(3) | tl(3) | th(3) | 0(7) | pad(8)] of the scaled coordinates, tn32[j] = |S t|^2 (f64 chain,
// rounded to f32); rows [n, n_pad) are zero with tn32 = +inf (they never win)
__global__ void k_chamfer_pack16(const float* __restrict__ tgt, int64_t n, int64_t n_pad, _Float16* __restrict__ img,
                                 float* __restrict__ tn32, float4* __restrict__ t4f) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  union {
    _Float16 h[CHF_PITCH];
    uint4 v[3];
  } row;
#pragma unroll
  for (int c = 0; c < CHF_PITCH; ++c) row.h[c] = (_Float16)0.0f;
  float t2 = INFINITY;
  if (i < n) {
    // (x, y, z, 0) in one 16-byte row: the final canonical evaluation reads a candidate row with ONE request instead of three
    t4f[i] = make_float4(tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2], 0.0f);
    double n2 = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = tgt[3 * i + c] * CHF_SCALE;     // exact (power of two) unless it overflows, which the kernel flags
      _Float16 hi, lo;
      knf_split(v, &hi, &lo);
      row.h[c] = hi;
      row.h[3 + c] = lo;
      row.h[6 + c] = hi;
      n2 = fma((double)v, (double)v, n2);
    }
    t2 = (float)n2;
  }
  uint4* dst = reinterpret_cast<uint4*>(img + i * CHF_PITCH);
#pragma unroll
  for (int c = 0; c < 3; ++c) dst[c] = row.v[c];
  tn32[i] = t2;
}

__global__ __launch_bounds__(256) void k_chamfer_f16(const ChamferWork* __restrict__ work, const float* __restrict__ src,
                                                     const float4* __restrict__ t4f, const _Float16* __restrict__ img,
                                                     const float* __restrict__ tn32, const float* __restrict__ T,
                                                     int reduce_max, double* __restrict__ partial,
                                                     int32_t* __restrict__ flag) {
  __shared__ __attribute__((aligned(16))) _Float16 a_s[2][CHF_ROWS * CHF_PITCH];
  __shared__ __attribute__((aligned(16))) float tn_s[2][CHF_ROWS];
  __shared__ double red[CHM_ST];
  __shared__ float wmax[4];
  __shared__ int wg_bad;
  const ChamferWork wk = work[blockIdx.x];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int col = lane & 31;
  const float* Tp = T + (int64_t)wk.prob * 16;
  if (tid == 0) wg_bad = 0;
  double px[CHF_NG], py[CHF_NG], pz[CHF_NG];
  f16x8 bop[CHF_NG];
  float b1[CHF_NG], b2[CHF_NG], b3[CHF_NG];
  int t1[CHF_NG], t2[CHF_NG];
  bool in_range = true;
#pragma unroll
  for (int g = 0; g < CHF_NG; ++g) {
    const int sloc = wave * 32 * CHF_NG + 32 * g + col;
    const float* sp = src + (wk.s0 + (sloc < wk.sn ? sloc : 0)) * 3;
    const double x = sp[0], y = sp[1], z = sp[2];
    px[g] = fma((double)Tp[0], x, fma((double)Tp[1], y, fma((double)Tp[2], z, (double)Tp[3])));
    py[g] = fma((double)Tp[4], x, fma((double)Tp[5], y, fma((double)Tp[6], z, (double)Tp[7])));
    pz[g] = fma((double)Tp[8], x, fma((double)Tp[9], y, fma((double)Tp[10], z, (double)Tp[11])));
    const float pf[3] = {(float)px[g] * CHF_SCALE, (float)py[g] * CHF_SCALE, (float)pz[g] * CHF_SCALE};
    _Float16 hi[3], lo[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      knf_split(-2.0f * pf[c], &hi[c], &lo[c]);
      in_range = in_range && fabsf(pf[c]) < 60.0f * CHF_SCALE;   // (NaN fails too)
    }
    // B rows k: 0..2 = -2 ph (x th), 3..5 = -2 ph (x tl), 6..8 = -2 pl (x th), 9..15 = 0; this lane holds k = 8 half .. + 7
    const _Float16 z0 = (_Float16)0.0f;
    if (half == 0)
      bop[g] = f16x8{hi[0], hi[1], hi[2], hi[0], hi[1], hi[2], lo[0], lo[1]};
    else
      bop[g] = f16x8{lo[2], z0, z0, z0, z0, z0, z0, z0};
    b1[g] = b2[g] = b3[g] = INFINITY;
    t1[g] = t2[g] = -1;
  }
  // register-staged double buffer: the next stage's 12 KiB image + |t|^2 are requested before this stage is computed
  const uint4* gimg = reinterpret_cast<const uint4*>(img + (int64_t)wk.t0 * CHF_PITCH);
  constexpr int U4_PER_STAGE = CHF_ROWS * CHF_PITCH * 2 / 16;   // 768 = 3 per thread
  uint4 st0, st1, st2;     // (three named registers: an array captured by the lambdas below went to scratch)
  float stn;
  float tmax2 = 0.0f;
  auto load_stage = [&](int base) {
    const uint4* g = gimg + (int64_t)base * (CHF_PITCH * 2 / 16) + tid;
    st0 = g[0];
    st1 = g[256];
    st2 = g[512];
    const int r = base + tid;
    stn = r < wk.tn ? tn32[wk.t0 + r] : INFINITY;   // rows past the segment (another problem's rows) never win
    if (r < wk.tn) tmax2 = fmaxf(tmax2, stn);
  };
  auto store_stage = [&](int b) {
    uint4* d = reinterpret_cast<uint4*>(a_s[b]) + tid;
    d[0] = st0;
    d[256] = st1;
    d[512] = st2;
    tn_s[b][tid] = stn;
  };
  static_assert(U4_PER_STAGE == 3 * 256, "three 16-byte pieces per thread");
  if (wk.tn > 0) {
    load_stage(0);
    store_stage(0);
  }
  int buf = 0;
  for (int base = 0; base < wk.tn; base += CHF_ROWS) {
    __syncthreads();
    const bool more = base + CHF_ROWS < wk.tn;
    if (more) load_stage(base + CHF_ROWS);
#pragma unroll 1
    for (int t = 0; t < CHF_ROWS / 32; ++t) {
      if (base + 32 * t >= wk.tn) break;   // whole tile past the range (block-uniform)
      const f16x8 a = *reinterpret_cast<const f16x8*>(a_s[buf] + (t * 32 + col) * CHF_PITCH + 8 * half);
      f32x16 c16;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 v = *reinterpret_cast<const float4*>(&tn_s[buf][t * 32 + 8 * q4 + 4 * half]);
        c16[4 * q4 + 0] = v.x; c16[4 * q4 + 1] = v.y; c16[4 * q4 + 2] = v.z; c16[4 * q4 + 3] = v.w;
      }
      const int tile = base / 32 + t;
#pragma unroll
      for (int g = 0; g < CHF_NG; ++g) {
        const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop[g], c16, 0, 0, 0);
        float m = fminf(fminf(d[0], d[1]), d[2]);
        m = fminf(fminf(m, d[3]), d[4]);
        m = fminf(fminf(m, d[5]), d[6]);
        m = fminf(fminf(m, d[7]), d[8]);
        m = fminf(fminf(m, d[9]), d[10]);
        m = fminf(fminf(m, d[11]), d[12]);
        m = fminf(fminf(m, d[13]), d[14]);
        m = fminf(m, d[15]);
        // sorted insertion of the tile minimum without a branch: new j-th value = median of (old j-1-th, old j-th, m);
        // the tile ids follow the two comparisons
        const float o1 = b1[g], o2 = b2[g];
        const bool lt1 = m < o1, lt2 = m < o2;
        b1[g] = fminf(o1, m);
        b2[g] = __builtin_amdgcn_fmed3f(o1, o2, m);
        b3[g] = __builtin_amdgcn_fmed3f(o2, b3[g], m);
        t2[g] = lt1 ? t1[g] : (lt2 ? tile : t2[g]);
        t1[g] = lt1 ? tile : t1[g];
      }
    }
    if (more) store_stage(buf ^ 1);
    buf ^= 1;
  }
  // largest |S t|^2 of the segment (error budget)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) tmax2 = fmaxf(tmax2, __shfl_xor(tmax2, off));
  if (lane == 0) wmax[wave] = tmax2;
  __syncthreads();
  const double tmax = sqrt((double)fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))) / (double)CHF_SCALE;
  bool bad = !in_range || !(tmax < 60.0);
  const float4* trow = t4f + wk.t0;
  auto eval_tile = [&](int g, int tile) {   // canonical distances of this lane's 16 rows of a tile: smallest
    double e = INFINITY;
    if (tile >= 0) {
      float4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = tile * 32 + 4 * half + (i & 3) + 8 * (i >> 2);
        v[i] = trow[row < wk.tn ? row : 0];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = tile * 32 + 4 * half + (i & 3) + 8 * (i >> 2);
        const double dx = px[g] - (double)v[i].x, dy = py[g] - (double)v[i].y, dz = pz[g] - (double)v[i].z;
        const double d = fma(dz, dz, fma(dy, dy, dx * dx));
        e = row < wk.tn ? fmin(e, d) : e;
      }
    }
    return e;
  };
#pragma unroll
  for (int g = 0; g < CHF_NG; ++g) {
    // Every unevaluated row has an approximate value >= the smallest tile minimum that was not evaluated (in scaled units
    // S^2 (|t|^2 - 2 p.t)), hence an exact one >= that - eps with eps covering: <= 16 f32 accumulation steps relative to
    // sum |terms| <= (|P| + |T|)^2, the dropped lo.lo products and the hi + lo cut residuals (2^-22 relative each), the f32
    // rounding of |T|^2 -- together < 2^-20 (|P| + |T|max)^2, charged 2^-19 -- and lo parts below the f16 normal range
    // should the matrix pipe flush them (<= 2^-14 per product side: 2^-10 (|P| + |T|max) covers all nine).
    const double pn2 = fma(pz[g], pz[g], fma(py[g], py[g], px[g] * px[g]));
    const double S = (double)CHF_SCALE, PT = S * (sqrt(pn2) + tmax);
    const double eps = 0x1.0p-19 * PT * PT + 0x1.0p-10 * PT;
    // best tile of either lane of the source first; the second tiles only where that does not settle it
    double e = eval_tile(g, t1[g]);
    e = fmin(e, __shfl_xor(e, 32));
    float rest = fminf(b2[g], __shfl_xor(b2[g], 32));     // smallest unevaluated tile minimum of the source
    bool ok = rest == INFINITY || (e - pn2) * S * S <= (double)rest - eps;
    if (__any(!ok)) {
      double e2 = ok ? INFINITY : eval_tile(g, t2[g]);
      e2 = fmin(e2, __shfl_xor(e2, 32));
      if (!ok) {
        e = fmin(e, e2);
        rest = fminf(b3[g], __shfl_xor(b3[g], 32));
        ok = rest == INFINITY || (e - pn2) * S * S <= (double)rest - eps;
      }
    }
    const int sloc = wave * 32 * CHF_NG + 32 * g + col;
    if (sloc < wk.sn && !ok) bad = true;
    if (half == 0) red[sloc] = sloc < wk.sn ? sqrt(e) : 0.0;
  }
  if (__any(bad) && lane == 0) atomicOr(&wg_bad, 1);
  __syncthreads();
  for (int off = CHM_ST / 2; off > 0; off >>= 1) {
    if (tid < off) red[tid] = reduce_max ? fmax(red[tid], red[tid + off]) : red[tid] + red[tid + off];
    __syncthreads();
  }
  if (tid == 0) {
    partial[wk.slot] = red[0];
    flag[blockIdx.x] = wg_bad;
  }
}

__global__ void k_chamfer_finish(const double* __restrict__ partial,
                                 const int32_t* __restrict__ slot_begin,
                                 const int64_t* __restrict__ src_count, int n_prob, int reduce_max,
                                 double* __restrict__ out) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_prob) return;
  double s = 0.0;
  for (int i = slot_begin[p]; i < slot_begin[p + 1]; ++i)
    s = reduce_max ? fmax(s, partial[i]) : s + partial[i];
  if (reduce_max)
    out[p] = src_count[p] > 0 ? s : NAN;
  else
    out[p] = src_count[p] > 0 ? s / (double)src_count[p] : NAN;
}

template <typename T>
static int upload(PoolBuf<T>& buf, const std::vector<T>& host, hipStream_t s) {
  if (!buf.alloc(host.size())) return CS_ERR_HIP;
  if (!host.empty())
    CS_HIP_CHECK(hipMemcpyAsync(buf.p, host.data(), host.size() * sizeof(T),
                                hipMemcpyHostToDevice, s));
  // pageable source: the runtime has staged the bytes when the call returns
  return CS_OK;
}

}  // namespace cs

using namespace cs;

// {queries answered by the f16 shortlist path, of those recomputed exhaustively}; counted only while
// CS_KNN_STATS=1 (the count costs a synchronisation)
static std::atomic<unsigned long long> g_knn_stats[2];
static std::atomic<unsigned long long> g_chamfer_stats[2];

__global__ void k_count_flags(const int32_t* __restrict__ flag, int64_t n, unsigned long long* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  unsigned long long v = (i < n && flag[i]) ? 1ULL : 0ULL;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

extern "C" {

void cs_knn_shortlist_stats(uint64_t out[2], int reset) {
  for (int i = 0; i < 2; ++i) {
    if (out) out[i] = g_knn_stats[i].load();
    if (reset) g_knn_stats[i].store(0);
  }
}

void cs_chamfer_f16_stats(uint64_t out[2], int reset) {
  for (int i = 0; i < 2; ++i) {
    if (out) out[i] = g_chamfer_stats[i].load();
    if (reset) g_chamfer_stats[i].store(0);
  }
}

int cs_knn_feat(const float* d_qf, const int64_t* h_qoff, const float* d_tf,
                const int64_t* h_toff, const int32_t* h_qseg, const int32_t* h_tseg, int n_prob,
                int dim, int k, const int32_t* d_qlabel, const int32_t* d_tlabel,
                const int32_t* d_perm, int32_t* d_idx, double* d_dist, void* stream) {
  CS_REQUIRE(d_qf && d_tf && h_qoff && h_toff && h_qseg && h_tseg && d_idx, CS_ERR_INVALID,
             "cs_knn_feat: NULL argument");
  CS_REQUIRE(k >= 1 && k <= KNN_MAXK, CS_ERR_UNSUPPORTED, "cs_knn_feat: k = %d not in [1, %d]", k,
             KNN_MAXK);
  CS_REQUIRE(dim == 16 || dim == 32 || dim == 3, CS_ERR_UNSUPPORTED,
             "cs_knn_feat: feature dimension %d not supported (3, 16, 32)", dim);
  CS_REQUIRE((d_qlabel == nullptr) == (d_tlabel == nullptr) &&
                 (d_qlabel == nullptr) == (d_perm == nullptr),
             CS_ERR_INVALID, "cs_knn_feat: labels and perm must be given together");
  if (n_prob <= 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  // 16-d features with k <= 6: f16 matrix-core shortlist + canonical rescore + verification (default);
  // CS_KNN_MFMA=64 selects the f64 matrix-pipe shortlist, CS_KNN_MFMA=0 the all-VALU exhaustive kernel
  const char* env = getenv("CS_KNN_MFMA");
  const bool small = dim == 16 && k <= KNM_KK - 2;
  const bool f16path = small && !(env && (env[0] == '0' || env[0] == '6'));
  const bool mfma = small && env && env[0] == '6';
  const int qtile = mfma ? KNM_QT : 256;
  std::vector<KnnWork> work;
  int64_t out_row = 0;
  int nqseg = 0, ntseg = 0;
  for (int p = 0; p < n_prob; ++p) {
    CS_REQUIRE(h_qseg[p] >= 0 && h_tseg[p] >= 0, CS_ERR_INVALID,
               "cs_knn_feat: negative segment id in problem %d", p);
    nqseg = h_qseg[p] + 1 > nqseg ? h_qseg[p] + 1 : nqseg;
    ntseg = h_tseg[p] + 1 > ntseg ? h_tseg[p] + 1 : ntseg;
    const int64_t q0 = h_qoff[h_qseg[p]], t0 = h_toff[h_tseg[p]];
    int64_t qn = h_qoff[h_qseg[p] + 1] - q0, tn = h_toff[h_tseg[p] + 1] - t0;
    CS_REQUIRE(qn >= 0 && tn >= 0 && tn < (1LL << 31), CS_ERR_INVALID,
               "cs_knn_feat: bad segment in problem %d", p);
    for (int64_t q = 0; q < qn; q += qtile) {
      KnnWork w;
      w.q0 = q0 + q;
      w.t0 = t0;
      w.o0 = out_row + q;
      w.qn = (int32_t)(qn - q < qtile ? qn - q : qtile);
      w.tn = (int32_t)tn;
      w.prob = p;
      w.pad = h_tseg[p];  // target segment (label-order tables of the MFMA path)
      work.push_back(w);
    }
    out_row += qn;
  }
  if (work.empty()) return CS_OK;
  PoolBuf<KnnWork> dwork;
  int rc = upload(dwork, work, s);
  if (rc) return rc;
  double knn_flop = 0.0;
  for (const KnnWork& w : work) knn_flop += 3.0 * (double)w.qn * (double)w.tn * (double)dim;
  if (f16path) {
    const int64_t nq_rows = h_qoff[nqseg], nt_rows = h_toff[ntseg];
    CS_REQUIRE(nt_rows < (1LL << 31) && ntseg < (1 << 27), CS_ERR_UNSUPPORTED,
               "cs_knn_feat: too many target rows / segments");
    PoolBuf<_Float16> qrows((size_t)(nq_rows ? nq_rows : 1) * 48);
    PoolBuf<_Float16> img((size_t)(nt_rows + KNF_ROWS) * KNF_PITCH);  // + one stage of slack for the last copy
    PoolBuf<float> tn32((size_t)(nt_rows ? nt_rows : 1)), tau((size_t)(out_row ? out_row : 1));
    PoolBuf<int32_t> ti32((size_t)(nt_rows ? nt_rows : 1)), cand((size_t)(out_row ? out_row : 1) * 2 * KNF_KK);
    PoolBuf<int32_t> qflag((size_t)(out_row ? out_row : 1)), tile_flag(work.size());
    PoolBuf<unsigned> t2max((size_t)ntseg);
    // threshold pass (k_knn_f16<1>): per query an upper bound of its k-th distance before the shortlist pass starts
    // (CS_KNN_TWOPASS=0: the shortlist pass alone, from +inf)
    const bool two_pass = k <= 6 && !(getenv("CS_KNN_TWOPASS") && getenv("CS_KNN_TWOPASS")[0] == '0');
    PoolBuf<float> thr0((size_t)(out_row ? out_row : 1)), qn32((size_t)(nq_rows ? nq_rows : 1));
    CS_REQUIRE(thr0.p && qn32.p, CS_ERR_HIP, "cs_knn_feat: scratch allocation failed");
    PoolBuf<int32_t> torder, rows_in, lab_start;
    PoolBuf<uint32_t> keys, keys_sorted;
    PoolBuf<int64_t> dtoff;
    PoolBuf<char> tmp;
    CS_REQUIRE(qrows.p && img.p && tn32.p && tau.p && ti32.p && cand.p && qflag.p && tile_flag.p && t2max.p,
               CS_ERR_HIP, "cs_knn_feat: scratch allocation failed");
    std::vector<int64_t> toff(h_toff, h_toff + ntseg + 1);
    rc = upload(dtoff, toff, s);
    if (rc) return rc;
    if (d_tlabel && nt_rows) {
      // label order of every target segment (stable: equal labels keep their row order)
      CS_REQUIRE(torder.alloc((size_t)nt_rows) && keys.alloc((size_t)nt_rows) &&
                     keys_sorted.alloc((size_t)nt_rows) && rows_in.alloc((size_t)nt_rows) &&
                     lab_start.alloc((size_t)ntseg * 10),
                 CS_ERR_HIP, "cs_knn_feat: scratch allocation failed");
      hipLaunchKernelGGL(k_seg_keys, dim3(16, (unsigned)ntseg), dim3(256), 0, s, dtoff.p, ntseg, d_tlabel,
                         keys.p, rows_in.p);
      int end_bit = 4;
      while ((1LL << end_bit) < (int64_t)ntseg * 16) ++end_bit;
      size_t tmp_bytes = 0;
      CS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.p, keys_sorted.p, rows_in.p,
                                                      torder.p, (int)nt_rows, 0, end_bit, s));
      CS_REQUIRE(tmp.alloc(tmp_bytes ? tmp_bytes : 1), CS_ERR_HIP, "cs_knn_feat: scratch allocation failed");
      CS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.p, keys_sorted.p, rows_in.p,
                                                      torder.p, (int)nt_rows, 0, end_bit, s));
      hipLaunchKernelGGL(k_label_starts, dim3((unsigned)ceil_div((int64_t)ntseg * 10, 256)), dim3(256), 0, s,
                         dtoff.p, ntseg, keys_sorted.p, lab_start.p);
    }
    CS_HIP_CHECK(hipMemsetAsync(t2max.p, 0, sizeof(unsigned) * ntseg, s));
    CS_HIP_CHECK(hipMemsetAsync(tile_flag.p, 0, sizeof(int32_t) * work.size(), s));
    if (nt_rows)
      hipLaunchKernelGGL(k_knf_pack_targets, dim3(16, (unsigned)ntseg), dim3(256), 0, s, d_tf, dtoff.p, ntseg,
                         (d_tlabel && nt_rows) ? torder.p : (const int32_t*)nullptr, img.p, tn32.p, ti32.p, t2max.p);
    if (nq_rows)
      hipLaunchKernelGGL(k_knf_pack_queries, dim3((unsigned)ceil_div(nq_rows, 256)), dim3(256), 0, s, d_qf,
                         nq_rows, qrows.p, qn32.p);
    {
      ProfScope prof("knn", s, knn_flop);
      const int32_t* labs = (d_tlabel && nt_rows) ? lab_start.p : (const int32_t*)nullptr;
      if (two_pass)
        hipLaunchKernelGGL(k_knn_f16<1>, dim3((unsigned)work.size()), dim3(256), 0, s, dwork.p, qrows.p, img.p, tn32.p, ti32.p,
                           d_qlabel, d_perm, labs, cand.p, tau.p, thr0.p, qn32.p, t2max.p, k);
      hipLaunchKernelGGL(k_knn_f16<0>, dim3((unsigned)work.size()), dim3(256), 0, s, dwork.p, qrows.p, img.p,
                         tn32.p, ti32.p, d_qlabel, d_perm, labs, cand.p, tau.p, two_pass ? thr0.p : (float*)nullptr, qn32.p,
                         t2max.p, k);
      hipLaunchKernelGGL(k_knn_rescore_f16, dim3((unsigned)work.size()), dim3(256), 0, s, dwork.p, d_qf, d_tf,
                         cand.p, tau.p, t2max.p, k, d_idx, d_dist, qflag.p, tile_flag.p);
      // exhaustive recomputation of the queries whose shortlist could not be verified (normally none)
      hipLaunchKernelGGL((k_knn_feat<16>), dim3((unsigned)work.size()), dim3(256), 0, s, dwork.p, d_qf, d_tf, k,
                         d_qlabel, d_tlabel, d_perm, d_idx, d_dist, tile_flag.p, qflag.p);
      CS_LAUNCH_CHECK();
    }
    const char* env_st = getenv("CS_KNN_STATS");
    if (env_st && env_st[0] == '1' && out_row > 0) {
      PoolBuf<unsigned long long> cnt(1);
      CS_REQUIRE(cnt.p, CS_ERR_HIP, "cs_knn_feat: scratch allocation failed");
      CS_HIP_CHECK(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), s));
      hipLaunchKernelGGL(k_count_flags, dim3((unsigned)ceil_div(out_row, 256)), dim3(256), 0, s, qflag.p, out_row,
                         cnt.p);
      unsigned long long h = 0;
      CS_HIP_CHECK(hipMemcpyAsync(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost, s));
      CS_HIP_CHECK(hipStreamSynchronize(s));
      g_knn_stats[0] += (unsigned long long)out_row;
      g_knn_stats[1] += h;
    }
    return CS_OK;  // scratch goes back to this thread's stream-ordered cache; outputs are valid in stream order
  }
  if (mfma) {
    const int64_t nq_rows = h_qoff[nqseg], nt_rows = h_toff[ntseg];
    CS_REQUIRE(nt_rows < (1LL << 31) && ntseg < (1 << 27), CS_ERR_UNSUPPORTED,
               "cs_knn_feat: too many target rows / segments");
    PoolBuf<double> qnorm((size_t)(nq_rows ? nq_rows : 1)), tnorm((size_t)(nt_rows ? nt_rows : 1));
    PoolBuf<int32_t> cand((size_t)(out_row ? out_row : 1) * 4 * KNM_KK);
    PoolBuf<int32_t> torder;
    PoolBuf<uint32_t> keys, keys_sorted;
    PoolBuf<int32_t> rows_in, lab_start;
    PoolBuf<int64_t> dtoff;
    PoolBuf<char> tmp;
    CS_REQUIRE(qnorm.p && tnorm.p && cand.p, CS_ERR_HIP, "cs_knn_feat: scratch allocation failed");
    if (nq_rows)
      hipLaunchKernelGGL(k_row_norms, dim3((unsigned)ceil_div(nq_rows, 256)), dim3(256), 0, s, d_qf,
                         nq_rows, 16, qnorm.p);
    if (nt_rows)
      hipLaunchKernelGGL(k_row_norms, dim3((unsigned)ceil_div(nt_rows, 256)), dim3(256), 0, s, d_tf,
                         nt_rows, 16, tnorm.p);
    if (d_tlabel && nt_rows) {
      // label order of every target segment (stable: equal labels keep their row order)
      std::vector<int64_t> toff(h_toff, h_toff + ntseg + 1);
      rc = upload(dtoff, toff, s);
      if (rc) return rc;
      CS_REQUIRE(torder.alloc((size_t)nt_rows) && keys.alloc((size_t)nt_rows) &&
                     keys_sorted.alloc((size_t)nt_rows) && rows_in.alloc((size_t)nt_rows),
                 CS_ERR_HIP, "cs_knn_feat: scratch allocation failed");
      hipLaunchKernelGGL(k_seg_keys, dim3(16, (unsigned)ntseg), dim3(256), 0, s, dtoff.p, ntseg, d_tlabel,
                         keys.p, rows_in.p);
      int end_bit = 4;
      while ((1LL << end_bit) < (int64_t)ntseg * 16) ++end_bit;
      size_t tmp_bytes = 0;
      CS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.p, keys_sorted.p, rows_in.p,
                                                      torder.p, (int)nt_rows, 0, end_bit, s));
      CS_REQUIRE(tmp.alloc(tmp_bytes ? tmp_bytes : 1), CS_ERR_HIP, "cs_knn_feat: scratch allocation failed");
      CS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.p, keys_sorted.p, rows_in.p,
                                                      torder.p, (int)nt_rows, 0, end_bit, s));
      CS_REQUIRE(lab_start.alloc((size_t)ntseg * 10), CS_ERR_HIP, "cs_knn_feat: scratch allocation failed");
      hipLaunchKernelGGL(k_label_starts, dim3((unsigned)ceil_div((int64_t)ntseg * 10, 256)), dim3(256), 0, s,
                         dtoff.p, ntseg, keys_sorted.p, lab_start.p);
    }
    {
      ProfScope prof("knn", s, knn_flop);
      hipLaunchKernelGGL(k_knn_mfma16, dim3((unsigned)work.size()), dim3(256), 0, s, dwork.p, d_qf, d_tf,
                         qnorm.p, tnorm.p, d_qlabel, d_tlabel, d_perm, d_tlabel ? torder.p : nullptr,
                         d_tlabel ? lab_start.p : nullptr, cand.p);
      hipLaunchKernelGGL(k_knn_rescore16, dim3((unsigned)work.size()), dim3(KNM_QT), 0, s, dwork.p, d_qf,
                         d_tf, cand.p, k, d_idx, d_dist);
      CS_LAUNCH_CHECK();
    }
    return CS_OK;  // scratch goes back to this thread's stream-ordered cache; outputs are valid in stream order
  }
  {
    ProfScope prof("knn", s, knn_flop);
    dim3 grid((unsigned)work.size());
    const int32_t* none = nullptr;
    if (dim == 16)
      hipLaunchKernelGGL((k_knn_feat<16>), grid, dim3(256), 0, s, dwork.p, d_qf, d_tf, k,
                         d_qlabel, d_tlabel, d_perm, d_idx, d_dist, none, none);
    else if (dim == 32)
      hipLaunchKernelGGL((k_knn_feat<32>), grid, dim3(256), 0, s, dwork.p, d_qf, d_tf, k,
                         d_qlabel, d_tlabel, d_perm, d_idx, d_dist, none, none);
    else
      hipLaunchKernelGGL((k_knn_feat<3>), grid, dim3(256), 0, s, dwork.p, d_qf, d_tf, k, d_qlabel,
                         d_tlabel, d_perm, d_idx, d_dist, none, none);
    CS_LAUNCH_CHECK();
  }
  return CS_OK;  // scratch goes back to this thread's stream-ordered cache; outputs are valid in stream order
}

// f64 matrix-pipe shortlist + canonical re-score (k <= TKM_KK - 2)
static int topk_f64_shortlist(const float* d_q, int64_t nq, const float* d_x, int64_t nx, int d, int k,
                              int64_t* d_idx, double* d_dist, hipStream_t s, int squared) {
  int nsplit = (int)(2048 / ceil_div(nq, TKM_QT));
  if (nsplit < 1) nsplit = 1;
  if (nsplit > 64) nsplit = 64;
  while (nsplit > 1 && (nx / nsplit < 4 * TKM_XT || nsplit * 4 * TKM_KK > TKM_MERGE_CAP)) --nsplit;
  const int ncand = nsplit * 4 * TKM_KK;
  PoolBuf<double> qn(nq), xn(nx), cand_d((size_t)nq * ncand);
  PoolBuf<int> cand_i((size_t)nq * ncand);
  PoolBuf<unsigned long long> cd((size_t)nq * k);
  PoolBuf<int> ci((size_t)nq * k);
  CS_REQUIRE(qn.p && xn.p && cand_d.p && cand_i.p && cd.p && ci.p, CS_ERR_HIP,
             "cs_l2_topk: scratch allocation failed");
  hipLaunchKernelGGL(k_row_norms, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, s, d_q, nq, d, qn.p);
  hipLaunchKernelGGL(k_row_norms, dim3((unsigned)ceil_div(nx, 256)), dim3(256), 0, s, d_x, nx, d, xn.p);
  hipLaunchKernelGGL(k_topk_mfma, dim3((unsigned)ceil_div(nq, TKM_QT), (unsigned)nsplit), dim3(256), 0,
                     s, d_q, nq, d_x, nx, d, qn.p, xn.p, nsplit, cand_d.p, cand_i.p);
  hipLaunchKernelGGL(k_topk_rescore, dim3((unsigned)nq), dim3(256), 0, s, d_q, d_x, d, cand_i.p, ncand,
                     k, cd.p, ci.p);
  hipLaunchKernelGGL(k_topk_finish, dim3((unsigned)ceil_div(nq * k, 256)), dim3(256), 0, s, cd.p,
                     ci.p, nq * k, d_idx, d_dist, squared);
  CS_LAUNCH_CHECK();
  return CS_OK;  // scratch goes back to this thread's stream-ordered cache; outputs are valid in stream order
}

// {queries through the f16 shortlist, of those recomputed by the f64 path}
static std::atomic<unsigned long long> g_topk_stats[2];

// f16 matrix-core shortlist (d = 64 / 128 / 256 / 512, k <= 10), canonical re-score, verification, f64 path for
// the queries that fail it.  Waits for the stream once (the number of such queries).
// what topk_f16_shortlist makes of the catalog before it looks at a query: the padded f16 hi / lo image, the f64 row
// norms and the bit pattern of their maximum.  A cs_topk_catalog keeps them across calls.
struct TkfCatalog {
  const _Float16* ximg = nullptr;
  const double* xn = nullptr;
  const unsigned* xmax = nullptr;
};
static int tkf_pitch(int d) { return (d / 16) * 32 + 8; }   // halfs per catalog image row (16 B of padding)
static int64_t tkf_rows_padded(int64_t nx) { return ceil_div(nx, TKF_ROWS) * TKF_ROWS; }
static int tkf_prepare_catalog(const float* d_x, int64_t nx, int d, _Float16* ximg, double* xn, unsigned* xmax,
                               hipStream_t s) {
  const int dch = d / 16;
  const int64_t n_pad = tkf_rows_padded(nx);
  CS_HIP_CHECK(hipMemsetAsync(xmax, 0, sizeof(unsigned), s));
  hipLaunchKernelGGL(k_row_norms, dim3((unsigned)ceil_div(nx, 256)), dim3(256), 0, s, d_x, nx, d, xn);
  hipLaunchKernelGGL(k_max_bits, dim3((unsigned)ceil_div(nx, 256)), dim3(256), 0, s, xn, nx, xmax);
  hipLaunchKernelGGL(k_tkf_pack, dim3((unsigned)ceil_div(n_pad * dch, 256)), dim3(256), 0, s, d_x, nx, n_pad, d, 1.0f,
                     ximg, tkf_pitch(d));
  CS_LAUNCH_CHECK();
  return CS_OK;
}

static int topk_f16_shortlist(const float* d_q, int64_t nq, const float* d_x, int64_t nx, int d, int k,
                              int64_t* d_idx, double* d_dist, hipStream_t s, int squared,
                              const TkfCatalog* prepared = nullptr) {
  const int dch = d / 16;
  const int pitch_h = tkf_pitch(d);
  const int64_t qtiles = ceil_div(nq, TKF_QT);
  // catalog splits: one full round of workgroups.  d >= 256 leaves room for ONE workgroup per CU (2 x 65 KiB
  // of LDS stages): 256 slots, filled without a partial second round (every split adds 24 candidates per
  // query to the exact re-score: 10 240 x 10^6 x 256-d went from 30 to 22.5 ms with 6 splits instead of 13);
  // the smaller images fit two per CU
  const int wg_slots = getenv("CS_TOPK_SLOTS") ? atoi(getenv("CS_TOPK_SLOTS")) : (d >= 256 ? 256 : 512);
  int nsplit = (int)(wg_slots / qtiles);   // (rounded DOWN: one workgroup more than slots costs a whole second round)
  if (nsplit < 2) nsplit = 2;   // (10^6 x 10^6: two half-catalog rounds measured 8 % faster than one full one)
  if (nsplit > 32) nsplit = 32;
  while (nsplit > 1 && nx / nsplit < 8 * TKF_ROWS) --nsplit;
  if (nsplit < 1) nsplit = 1;
  const int nlane = nsplit * 2;
  const int ncand = nlane * TKF_KK;  // <= 768 <= TKM_MERGE_CAP
  const int64_t n_pad = tkf_rows_padded(nx);
  PoolBuf<_Float16> qimg((size_t)nq * dch * 32), ximg_own;
  PoolBuf<double> qn(nq), xn_own;
  PoolBuf<unsigned> xmax_own;
  PoolBuf<int> cand_i((size_t)nq * ncand), ci((size_t)nq * k), flagged((size_t)nq + 1);
  PoolBuf<float> tau((size_t)nq * nlane);
  PoolBuf<unsigned long long> cd((size_t)nq * k);
  TkfCatalog cat;
  if (prepared) {
    cat = *prepared;
  } else {   // (no handle: the catalog is prepared for this call only)
    ximg_own.alloc((size_t)n_pad * pitch_h);
    xn_own.alloc((size_t)nx);
    xmax_own.alloc(1);
    CS_REQUIRE(ximg_own.p && xn_own.p && xmax_own.p, CS_ERR_HIP, "cs_l2_topk: scratch allocation failed");
    const int rc0 = tkf_prepare_catalog(d_x, nx, d, ximg_own.p, xn_own.p, xmax_own.p, s);
    if (rc0) return rc0;
    cat.ximg = ximg_own.p;
    cat.xn = xn_own.p;
    cat.xmax = xmax_own.p;
  }
  CS_REQUIRE(qimg.p && qn.p && cand_i.p && ci.p && flagged.p && tau.p && cd.p,
             CS_ERR_HIP, "cs_l2_topk: scratch allocation failed");
  int* const n_flagged = flagged.p + nq;
  CS_HIP_CHECK(hipMemsetAsync(n_flagged, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_row_norms, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, s, d_q, nq, d, qn.p);
  hipLaunchKernelGGL(k_tkf_pack, dim3((unsigned)ceil_div(nq * dch, 256)), dim3(256), 0, s, d_q, nq, nq, d,
                     -2.0f, qimg.p, dch * 32);
  const int terms = dch == 32 ? 2 : 3;
  int rc = dch == 32   ? launch_topk_f16<32, 2, 32>((int)qtiles, s, qimg.p, nq, cat.ximg, nx, cat.xn, nsplit, cand_i.p, tau.p)
           : dch == 16 ? launch_topk_f16<16, 3, 64>((int)qtiles, s, qimg.p, nq, cat.ximg, nx, cat.xn, nsplit, cand_i.p, tau.p)
           : dch == 8  ? launch_topk_f16<8, 3, 64>((int)qtiles, s, qimg.p, nq, cat.ximg, nx, cat.xn, nsplit, cand_i.p, tau.p)
                       : launch_topk_f16<4, 3, 64>((int)qtiles, s, qimg.p, nq, cat.ximg, nx, cat.xn, nsplit, cand_i.p, tau.p);
  if (rc) return rc;
  hipLaunchKernelGGL(k_topk_rescore, dim3((unsigned)nq), dim3(256), 0, s, d_q, d_x, d, cand_i.p, ncand, k,
                     cd.p, ci.p);
  hipLaunchKernelGGL(k_tkf_verify, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, s, cd.p, k, nq, tau.p,
                     nlane, qn.p, cat.xmax, d, terms, flagged.p, n_flagged);
  hipLaunchKernelGGL(k_topk_finish, dim3((unsigned)ceil_div(nq * k, 256)), dim3(256), 0, s, cd.p, ci.p,
                     nq * k, d_idx, d_dist, squared);
  CS_LAUNCH_CHECK();
  int h_flagged = 0;
  CS_HIP_CHECK(download_async(&h_flagged, n_flagged, sizeof(int), s));
  CS_HIP_CHECK(download_sync(s));
  g_topk_stats[0] += (unsigned long long)nq;
  g_topk_stats[1] += (unsigned long long)h_flagged;
  if (h_flagged > 0) {
    const int64_t nf = h_flagged;
    PoolBuf<float> qf((size_t)nf * d);
    PoolBuf<int64_t> fi((size_t)nf * k);
    PoolBuf<double> fd((size_t)nf * k);
    CS_REQUIRE(qf.p && fi.p && fd.p, CS_ERR_HIP, "cs_l2_topk: scratch allocation failed");
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)ceil_div(nf * d, 256)), dim3(256), 0, s, d_q, d, flagged.p,
                       nf, qf.p);
    rc = topk_f64_shortlist(qf.p, nf, d_x, nx, d, k, fi.p, fd.p, s, squared);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scatter_topk, dim3((unsigned)ceil_div(nf * k, 256)), dim3(256), 0, s, fi.p, fd.p, k,
                       flagged.p, nf, d_idx, d_dist);
    CS_LAUNCH_CHECK();
  }
  return CS_OK;
}

// ---- catalog handle -------------------------------------------------------------------------------------------
// The catalog of a retrieval run is fixed (evaluation.py:264-283 embeds the CAD library once and ranks every scan
// against it); what the f16 path makes of it -- image, norms, their maximum: 4.2 ms per call at 10^6 x 256 -- is made
// once here and reused by every cs_l2_topk_catalog call.  Shapes the f16 path does not take keep only the pointer.
struct cs_topk_catalog {
  const float* d_x = nullptr;
  int64_t nx = 0;
  int d = 0;
  _Float16* ximg = nullptr;
  double* xn = nullptr;
  unsigned* xmax = nullptr;
  hipEvent_t ready = nullptr;   // recorded behind the kernels that make ximg / xn / xmax
};

int cs_topk_catalog_create(const float* d_x, int64_t nx, int d, void* stream, cs_topk_catalog** out) {
  CS_REQUIRE(d_x && out, CS_ERR_INVALID, "cs_topk_catalog_create: NULL argument");
  *out = nullptr;
  CS_REQUIRE(d >= 1 && nx >= 1 && nx < (1LL << 31), CS_ERR_INVALID, "cs_topk_catalog_create: bad shape (%lld x %d)",
             (long long)nx, d);
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  cs_topk_catalog* c = new cs_topk_catalog();
  c->d_x = d_x;
  c->nx = nx;
  c->d = d;
  const bool f16_shape = (d == 64 || d == 128 || d == 256 || d == 512) && nx >= TKF_ROWS;
  if (f16_shape) {
    c->ximg = (_Float16*)pool_alloc((size_t)tkf_rows_padded(nx) * tkf_pitch(d) * sizeof(_Float16));
    c->xn = (double*)pool_alloc((size_t)nx * sizeof(double));
    c->xmax = (unsigned*)pool_alloc(sizeof(unsigned));
    int rc = (c->ximg && c->xn && c->xmax) ? tkf_prepare_catalog(d_x, nx, d, c->ximg, c->xn, c->xmax, s) : CS_ERR_HIP;
    if (rc == CS_OK && (hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
                        hipEventRecord(c->ready, s) != hipSuccess)) {
      set_error("cs_topk_catalog_create: could not record the ready event");
      rc = CS_ERR_HIP;
    }
    if (rc) {
      if (rc == CS_ERR_HIP && !(c->ximg && c->xn && c->xmax)) set_error("cs_topk_catalog_create: allocation failed");
      cs_topk_catalog_free(c);
      return rc;
    }
  }
  *out = c;
  return CS_OK;
}

void cs_topk_catalog_free(cs_topk_catalog* c) {
  if (!c) return;
  pool_free(c->ximg);
  pool_free(c->xn);
  pool_free(c->xmax);
  if (c->ready) (void)hipEventDestroy(c->ready);
  delete c;
}

static int l2_topk_impl(const float* d_q, int64_t nq, const float* d_x, int64_t nx, int d, int k, int64_t* d_idx,
                        double* d_dist, void* stream, const cs_topk_catalog* cat, int squared);

int cs_l2_topk_catalog(const float* d_q, int64_t nq, const cs_topk_catalog* cat, int k, int64_t* d_idx, double* d_dist,
                       int squared, void* stream) {
  CS_REQUIRE(cat, CS_ERR_INVALID, "cs_l2_topk_catalog: NULL catalog");
  // what the handle holds was made on the creating call's stream: every user waits for it in stream order
  if (cat->ready) CS_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, cat->ready, 0));
  return l2_topk_impl(d_q, nq, cat->d_x, cat->nx, cat->d, k, d_idx, d_dist, stream, cat, squared ? 1 : 0);
}

void cs_l2_topk_stats(uint64_t out[2], int reset) {
  for (int i = 0; i < 2; ++i) {
    if (out) out[i] = g_topk_stats[i].load();
    if (reset) g_topk_stats[i].store(0);
  }
}

int cs_l2_topk_sq(const float* d_q, int64_t nq, const float* d_x, int64_t nx, int d, int k, int64_t* d_idx,
                  double* d_dist2, void* stream) {
  return l2_topk_impl(d_q, nq, d_x, nx, d, k, d_idx, d_dist2, stream, nullptr, 1);
}

int cs_l2_topk(const float* d_q, int64_t nq, const float* d_x, int64_t nx, int d, int k,
               int64_t* d_idx, double* d_dist, void* stream) {
  return l2_topk_impl(d_q, nq, d_x, nx, d, k, d_idx, d_dist, stream, nullptr, 0);
}

static int l2_topk_impl(const float* d_q, int64_t nq, const float* d_x, int64_t nx, int d, int k, int64_t* d_idx,
                        double* d_dist, void* stream, const cs_topk_catalog* cat, int squared) {
  CS_REQUIRE(d_q && d_x && d_idx, CS_ERR_INVALID, "cs_l2_topk: NULL argument");
  CS_REQUIRE(d >= 1 && k >= 1 && k <= 1024 && k <= nx, CS_ERR_INVALID,
             "cs_l2_topk: need 1 <= k <= min(1024, nx) (k %d, nx %lld)", k, (long long)nx);
  CS_REQUIRE(nx < (1LL << 31), CS_ERR_UNSUPPORTED, "cs_l2_topk: catalog too large");
  if (nq == 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  ProfScope prof("topk", s, 2.0 * (double)nq * (double)nx * (double)d);
  // Large problems with a short list: shortlist on the matrix cores, exact re-score.  CS_TOPK_MFMA: "0"
  // exact slab path, "1" / "64" f64 matrix pipe, "16" f16 matrix cores (where the shape allows), unset:
  // f16 where the shape allows, else f64, for nq * nx >= 2^24.
  const char* force = getenv("CS_TOPK_MFMA");
  const bool big = (double)nq * (double)nx >= 16777216.0;
  const bool f16_shape = (d == 64 || d == 128 || d == 256 || d == 512) && k <= TKF_KK - 2 && nx >= TKF_ROWS;
  const bool want16 = force ? (force[0] == '1' && force[1] == '6') : big;
  const bool want64 = force ? (force[0] == '1' || force[0] == '6') : big;
  if (want16 && f16_shape) {
    TkfCatalog pre;
    const bool have = cat && cat->ximg;   // made for exactly (d_x, nx, d): the handle supplies all three
    if (have) {
      pre.ximg = cat->ximg;
      pre.xn = cat->xn;
      pre.xmax = cat->xmax;
    }
    return topk_f16_shortlist(d_q, nq, d_x, nx, d, k, d_idx, d_dist, s, squared, have ? &pre : nullptr);
  }
  if (want64 && k <= TKM_KK - 2) return topk_f64_shortlist(d_q, nq, d_x, nx, d, k, d_idx, d_dist, s, squared);
  // slab of catalog rows so that the f64 distance slab stays <= 1 GiB
  int64_t slab = (1LL << 27) / (nq > 0 ? nq : 1);
  if (slab < DM_CT) slab = DM_CT;
  if (slab > nx) slab = nx;
  PoolBuf<double> D2((size_t)nq * slab);
  PoolBuf<unsigned long long> cd((size_t)nq * k);
  PoolBuf<int> ci((size_t)nq * k);
  CS_REQUIRE(D2.p && cd.p && ci.p, CS_ERR_HIP, "cs_l2_topk: scratch allocation failed");
  hipLaunchKernelGGL(k_topk_init, dim3((unsigned)ceil_div(nq * k, 256)), dim3(256), 0, s, cd.p,
                     ci.p, nq * k);
  for (int64_t xb = 0; xb < nx; xb += slab) {
    int64_t xc = nx - xb < slab ? nx - xb : slab;
    dim3 grid((unsigned)ceil_div(xc, DM_CT), (unsigned)ceil_div(nq, DM_QT));
    hipLaunchKernelGGL(k_dist_matrix, grid, dim3(256), 0, s, d_q, nq, d_x, nx, d, xb, xc, D2.p);
    hipLaunchKernelGGL(k_row_topk, dim3((unsigned)nq), dim3(256), 0, s, D2.p, xc, xb, k, cd.p,
                       ci.p);
  }
  hipLaunchKernelGGL(k_topk_finish, dim3((unsigned)ceil_div(nq * k, 256)), dim3(256), 0, s, cd.p,
                     ci.p, nq * k, d_idx, d_dist, squared);
  CS_LAUNCH_CHECK();
  return CS_OK;  // scratch goes back to this thread's stream-ordered cache; outputs are valid in stream order
}

static int nn_dist_reduce(const float* d_src, const int64_t* h_soff, const float* d_tgt,
                          const int64_t* h_toff, const int32_t* h_src_seg, const int32_t* h_tgt_seg,
                          int n_prob, const float* d_T, int reduce_max, double* d_out, void* stream) {
  CS_REQUIRE(h_soff && h_toff && h_src_seg && h_tgt_seg && d_T && d_out, CS_ERR_INVALID,
             "cs_chamfer_1dir: NULL argument");
  if (n_prob <= 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  // CS_CHAMFER_MFMA=0 selects the exhaustive all-VALU chain kernel; CS_CHAMFER_F16=0 the f64 matrix-pipe kernel for every
  // tile (default: the f16 matrix-core ranking, the f64 kernel only for the tiles it flags)
  const char* env = getenv("CS_CHAMFER_MFMA");
  const bool mfma = !(env && env[0] == '0');
  const bool f16 = mfma && !(getenv("CS_CHAMFER_F16") && getenv("CS_CHAMFER_F16")[0] == '0');
  const int64_t stile = mfma ? CHM_ST : 256;
  std::vector<ChamferWork> work;
  std::vector<int32_t> slot_begin(n_prob + 1, 0);
  std::vector<int64_t> src_count(n_prob, 0);
  for (int p = 0; p < n_prob; ++p) {
    int ss = h_src_seg[p], ts = h_tgt_seg[p];
    CS_REQUIRE(ss >= 0 && ts >= 0, CS_ERR_INVALID, "cs_chamfer_1dir: negative segment id");
    int64_t sn = h_soff[ss + 1] - h_soff[ss], tn = h_toff[ts + 1] - h_toff[ts];
    CS_REQUIRE(sn >= 0 && tn >= 0 && tn < (1LL << 31), CS_ERR_INVALID,
               "cs_chamfer_1dir: bad segment in problem %d", p);
    slot_begin[p] = (int32_t)work.size();
    src_count[p] = sn;
    for (int64_t q = 0; q < sn; q += stile) {
      ChamferWork w;
      w.s0 = h_soff[ss] + q;
      w.t0 = h_toff[ts];
      w.sn = (int32_t)(sn - q < stile ? sn - q : stile);
      w.tn = (int32_t)tn;
      w.prob = p;
      w.slot = (int32_t)work.size();
      work.push_back(w);
    }
  }
  slot_begin[n_prob] = (int32_t)work.size();
  CS_REQUIRE(work.empty() || (d_src && d_tgt), CS_ERR_INVALID, "cs_chamfer_1dir: NULL point array");
  PoolBuf<ChamferWork> dwork;
  PoolBuf<int32_t> dslot;
  PoolBuf<int64_t> dcount;
  PoolBuf<double> partial(work.size() + 1), t4g;
  CS_REQUIRE(partial.p, CS_ERR_HIP, "cs_chamfer_1dir: scratch allocation failed");
  int rc = upload(dwork, work, s);
  if (!rc) rc = upload(dslot, slot_begin, s);
  if (!rc) rc = upload(dcount, src_count, s);
  if (rc) return rc;
  double ch_flop = 0.0;
  for (const ChamferWork& w : work) ch_flop += 8.0 * (double)w.sn * (double)w.tn;
  {
    ProfScope prof("chamfer", s, ch_flop);
    if (!work.empty()) {
      if (mfma) {
        int64_t nt_rows = 0;
        for (int p = 0; p < n_prob; ++p) nt_rows = std::max<int64_t>(nt_rows, h_toff[h_tgt_seg[p] + 1]);
        if (!t4g.alloc((size_t)(nt_rows ? nt_rows : 1) * 4)) {
          set_error("cs_chamfer_1dir: scratch allocation failed");
          return CS_ERR_HIP;
        }
        if (nt_rows)
          hipLaunchKernelGGL(k_chamfer_pack, dim3((unsigned)ceil_div(nt_rows, 256)), dim3(256), 0, s, d_tgt,
                             nt_rows, t4g.p);
        if (f16) {
          // image rows padded by one stage: the last stage of the last segment reads past its end
          const int64_t n_pad = nt_rows + CHF_ROWS;
          PoolBuf<_Float16> img16((size_t)n_pad * CHF_PITCH);
          PoolBuf<float> tn16((size_t)n_pad);
          PoolBuf<float4> t4f((size_t)(nt_rows ? nt_rows : 1));
          PoolBuf<int32_t> wflag(work.size());
          if (!img16.p || !tn16.p || !wflag.p || !t4f.p) {
            set_error("cs_chamfer_1dir: scratch allocation failed");
            return CS_ERR_HIP;
          }
          hipLaunchKernelGGL(k_chamfer_pack16, dim3((unsigned)ceil_div(n_pad, 256)), dim3(256), 0, s, d_tgt, nt_rows, n_pad,
                             img16.p, tn16.p, t4f.p);
          hipLaunchKernelGGL(k_chamfer_f16, dim3((unsigned)work.size()), dim3(256), 0, s, dwork.p, d_src, t4f.p, img16.p,
                             tn16.p, d_T, reduce_max, partial.p, wflag.p);
          hipLaunchKernelGGL(k_chamfer_mfma, dim3((unsigned)work.size()), dim3(256), 0, s, dwork.p, d_src, d_tgt, t4g.p, d_T,
                             reduce_max, partial.p, wflag.p);
          const bool stats = getenv("CS_CHAMFER_STATS") && getenv("CS_CHAMFER_STATS")[0] == '1';
          if (stats) {   // diagnostics / tests: tiles answered, of those recomputed by the f64 kernel (synchronises)
            PoolBuf<unsigned long long> cnt(1);
            unsigned long long h = 0;
            if (cnt.p && hipMemsetAsync(cnt.p, 0, 8, s) == hipSuccess) {
              hipLaunchKernelGGL(k_count_flags, dim3((unsigned)ceil_div((int64_t)work.size(), 256)), dim3(256), 0, s, wflag.p,
                                 (int64_t)work.size(), cnt.p);
              if (hipMemcpyAsync(&h, cnt.p, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
                g_chamfer_stats[0] += work.size();
                g_chamfer_stats[1] += h;
              }
            }
          }
        } else {
          hipLaunchKernelGGL(k_chamfer_mfma, dim3((unsigned)work.size()), dim3(256), 0, s, dwork.p, d_src,
                             d_tgt, t4g.p, d_T, reduce_max, partial.p, (const int32_t*)nullptr);
        }
      }
      else
        hipLaunchKernelGGL(k_chamfer, dim3((unsigned)work.size()), dim3(256), 0, s, dwork.p, d_src,
                           d_tgt, d_T, reduce_max, partial.p);
    }
    hipLaunchKernelGGL(k_chamfer_finish, dim3((unsigned)ceil_div(n_prob, 64)), dim3(64), 0, s,
                       partial.p, dslot.p, dcount.p, n_prob, reduce_max, d_out);
    CS_LAUNCH_CHECK();
  }
  return CS_OK;  // scratch goes back to this thread's stream-ordered cache; outputs are valid in stream order
}

int cs_chamfer_1dir(const float* d_src, const int64_t* h_soff, const float* d_tgt,
                    const int64_t* h_toff, const int32_t* h_src_seg, const int32_t* h_tgt_seg,
                    int n_prob, const float* d_T, double* d_out, void* stream) {
  return nn_dist_reduce(d_src, h_soff, d_tgt, h_toff, h_src_seg, h_tgt_seg, n_prob, d_T, 0, d_out,
                        stream);
}

int cs_hausdorff_1dir(const float* d_src, const int64_t* h_soff, const float* d_tgt,
                      const int64_t* h_toff, const int32_t* h_src_seg, const int32_t* h_tgt_seg,
                      int n_prob, const float* d_T, double* d_out, void* stream) {
  return nn_dist_reduce(d_src, h_soff, d_tgt, h_toff, h_src_seg, h_tgt_seg, n_prob, d_T, 1, d_out,
                        stream);
}

}  // extern "C"
