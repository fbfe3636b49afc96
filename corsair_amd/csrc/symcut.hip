// Symmetry part cut on gfx950: replaces symmetric_cut4 (utils/symmetry.py:182-259 of the reference).
//
// Per (cloud, anchor) -- k_symcut_select (steps 1-2, one workgroup), k_symcut_kmeans (step 3, one thread
// per restart), k_symcut_finish (step 4, one workgroup):
//   1. f64 squared feature distance of every voxel to the anchor (fma chain over the 16 channels)
//   2. exact selection of the n_nn (= 50) nearest voxels by an 8-pass MSB radix select on the
//      distance bit patterns (ties -> smaller row), emitted in ascending row order -- this is the
//      reference's  raw_pc[local_rank < 50]
//   3. the n_init (= 10) restarts of sklearn's KMeans(random_state=0, n_init=10) -- greedy k-means++ on
//      the tabulated draws of RandomState(0), Lloyd with sklearn's stopping rules -- one lane per
//      restart; first-best restart wins (see k_symcut_kmeans).
//   4. the acceptance-gate statistics: minimum centre distance, maximum mean member distance,
//      and the label histogram of the WHOLE cloud under the fitted centres.
// The gate itself (dist.min() > 0.15 > max(error), smallest std of label fractions) and the
// cyclic ordering of the 4 centres are a few flops per anchor and stay on the host.
#include <algorithm>
#include <vector>

#include "common.h"
#define KM_DRAWS_QUAL __device__
#include "kmeans_draws.h"

namespace cs {

constexpr int SYM_MAX_NN = 64;
constexpr int SYM_SEL_J = 32;    // keys per thread the selection keeps in registers (clouds of up to 8 192 rows)
constexpr int SYM_MAX_INIT = 10;  // restarts whose draws are tabulated (K = 4: 10 x 10 doubles)

__device__ __forceinline__ double dist2_3(double ax, double ay, double az, double bx, double by,
                                          double bz) {
  const double dx = ax - bx, dy = ay - by, dz = az - bz;
  return fma(dz, dz, fma(dy, dy, dx * dx));
}

struct KmState {
  double cx[4], cy[4], cz[4];
};

// nearest centre (ties -> lowest index) among the first K of st
__device__ __forceinline__ int nearest_center(const KmState& st, int K, double px, double py,
                                              double pz, double* dmin) {
  int best = 0;
  double bd = dist2_3(px, py, pz, st.cx[0], st.cy[0], st.cz[0]);
#pragma unroll
  for (int c = 1; c < 4; ++c) {
    if (c < K) {
      const double d = dist2_3(px, py, pz, st.cx[c], st.cy[c], st.cz[c]);
      if (d < bd) {
        bd = d;
        best = c;
      }
    }
  }
  *dmin = bd;
  return best;
}

// Exclusive prefix sum of one int per thread over a 256-thread workgroup (wave shuffles + 4 partials).
// All threads must call it; *total receives the sum.  wsum: 4 ints of LDS.
__device__ __forceinline__ int block_excl_scan256(int v, int* wsum, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  __syncthreads();  // wsum may still be read by a previous call
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) base += w < wave ? wsum[w] : 0;
  *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + incl - v;
}

// Step 1 for ALL anchors of a cloud at once: a thread owns one voxel, keeps its DIM features in registers
// and walks the anchors (their features in LDS), so a cloud's feature rows are read once instead of once
// per anchor (the per-anchor version fetched 1.5 GB per launch for 18 MB of features).  Same fma chain per
// (voxel, anchor) as before: the keys are bit-identical.  grid: x = 256-row chunk, y = cloud.
constexpr int SYM_KEYS_MAX_ANCHOR = 128;
template <int DIM>
__global__ __launch_bounds__(256) void k_symcut_keys(const float* __restrict__ feat,
                                                     const int64_t* __restrict__ off,
                                                     const int32_t* __restrict__ anchors, int n_anchor,
                                                     unsigned long long* __restrict__ key_scratch,
                                                     const int64_t* __restrict__ key_off) {
  __shared__ double af[SYM_KEYS_MAX_ANCHOR][DIM];
  const int cloud = blockIdx.y;
  const int64_t base = off[cloud];
  const int n = (int)(off[cloud + 1] - base);
  if ((int)blockIdx.x * 256 >= n) return;
  for (int a0 = 0; a0 < n_anchor; a0 += SYM_KEYS_MAX_ANCHOR) {
    const int na = min(SYM_KEYS_MAX_ANCHOR, n_anchor - a0);
    __syncthreads();
    for (int i = threadIdx.x; i < na * DIM; i += 256) {
      const int a = i / DIM, c = i - a * DIM;
      af[a][c] = (double)feat[(base + anchors[(int64_t)cloud * n_anchor + a0 + a]) * DIM + c];
    }
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row < n) {
      double f[DIM];
#pragma unroll
      for (int c = 0; c < DIM; ++c) f[c] = (double)feat[(base + row) * DIM + c];
      unsigned long long* keys = key_scratch + key_off[cloud] * n_anchor + (int64_t)a0 * n + row;
      for (int a = 0; a < na; ++a) {
        double d = 0.0;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          const double diff = af[a][c] - f[c];
          d = fma(diff, diff, d);
        }
        keys[(int64_t)a * n] = (unsigned long long)__double_as_longlong(d);  // d >= 0: bit order == value order
      }
    }
  }
}

// (keeping the distance keys in LDS instead of the global scratch was measured: no gain, the keys stay
// in L2)
template <int DIM>
__global__ __launch_bounds__(256) void k_symcut_select(
    const float* __restrict__ feat, const float* __restrict__ xyz,
    const int64_t* __restrict__ off, const int32_t* __restrict__ anchors, int n_anchor,
    const int32_t* __restrict__ Ks, int n_nn, int n_init, int max_iter, uint64_t seed,
    unsigned long long* __restrict__ key_scratch, const int64_t* __restrict__ key_off,
    double* __restrict__ out_centers, int32_t* __restrict__ out_counts,
    double* __restrict__ out_min_cdist, double* __restrict__ out_max_err,
    double* __restrict__ pts_g, int32_t* __restrict__ nsel_g) {
  __shared__ int hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_remaining;
  __shared__ int scan_a[256], scan_b[256];
  __shared__ int wsum[4];
  double (*pts)[3] = reinterpret_cast<double (*)[3]>(pts_g + (int64_t)blockIdx.x * SYM_MAX_NN * 3);

  const int blk = blockIdx.x;
  const int cloud = blk / n_anchor;
  const int tid = threadIdx.x;
  const int64_t base = off[cloud];
  const int n = (int)(off[cloud + 1] - base);
  const int K = Ks[cloud];
  const int n_sel = n < n_nn ? n : n_nn;
  double* oc = out_centers + (int64_t)blk * 12;
  if (n_sel < K || n == 0) {  // degenerate cloud: report a model the gate always rejects
    if (tid < 12) oc[tid] = 0.0;
    if (tid < 4) out_counts[(int64_t)blk * 4 + tid] = 0;
    if (tid == 0) {
      out_min_cdist[blk] = 0.0;
      out_max_err[blk] = INFINITY;
      nsel_g[blk] = 0;  // the later stages skip this (cloud, anchor)
    }
    return;
  }
  unsigned long long* keys = key_scratch + key_off[cloud] * n_anchor + (int64_t)(blk % n_anchor) * n;

  // ---- 1. distance keys: written by k_symcut_keys ---------------------------------------------
  // A cloud of up to 4 x 64 x SYM_SEL_J rows keeps its keys in REGISTERS for the radix passes and the compaction (one
  // read of the scratch instead of one per pass + two: round 3 measured 1.1 GB per call for 18 MB of features).  Thread
  // (wave w, lane l) holds rows w q + l + 64 j -- the layout the ordered compaction walks.  ~0 marks "no row" (a key is
  // the bit pattern of a finite non-negative double).
  const int sel_q = ((n + 3) / 4 + 63) / 64 * 64;
  const bool in_regs = sel_q <= 64 * SYM_SEL_J;
  unsigned long long kr[SYM_SEL_J];
  if (in_regs) {
    const int r0 = (tid >> 6) * sel_q + (tid & 63);
#pragma unroll
    for (int j = 0; j < SYM_SEL_J; ++j) {
      const int row = r0 + 64 * j;
      kr[j] = (64 * j < sel_q && row < n) ? keys[row] : ~0ULL;
    }
  }
  if (tid == 0) {
    s_prefix = 0;
    s_remaining = n_sel;
  }
  __syncthreads();

  // ---- 2. radix select of the n_sel-th smallest key ---------------------------------------
  // The passes stop as soon as the bin of the wanted rank holds exactly the keys still needed: every key
  // of that bin is then selected, whatever its lower digits are (typically after 4-5 of the 8 passes).
  __shared__ int s_done;
  int shift_final = 0;
  if (tid == 0) s_done = 0;
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    // squared distances of unit-norm features share their leading bytes: in the first passes nearly
    // every key lands in one or two bins, so equal consecutive digits are counted in a register and
    // flushed with one LDS atomic per run instead of one per key
    int last = -1, run = 0;
    auto tally = [&](unsigned long long key) {
      if (pass == 0 || (key >> (shift + 8)) == prefix) {
        const int digit = (int)((key >> shift) & 255ULL);
        if (digit == last) {
          ++run;
        } else {
          if (run) atomicAdd(&hist[last], run);
          last = digit;
          run = 1;
        }
      }
    };
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < SYM_SEL_J; ++j)
        if (kr[j] != ~0ULL) tally(kr[j]);
    } else {
      for (int i = tid; i < n; i += 256) tally(keys[i]);
    }
    if (run) atomicAdd(&hist[last], run);
    __syncthreads();
    {
      // digit = first bin whose inclusive count reaches the remaining rank (parallel scan of the 256 bins)
      const int rem = s_remaining;
      int total;
      const int hv = hist[tid];
      const int excl = block_excl_scan256(hv, wsum, &total);
      const bool here = excl < rem && excl + hv >= rem;  // exactly one bin (or none: all in bin 255)
      if (here) {
        s_remaining = rem - excl;
        s_prefix = (prefix << 8) | (unsigned long long)tid;
        if (excl + hv == rem) s_done = 1;  // the whole bin is selected
      }
      if (tid == 0 && total < rem) {  // cannot happen (rem <= selected keys); mirror the serial fallback
        s_remaining = rem - (total - hist[255]);
        s_prefix = (prefix << 8) | 255ULL;
      }
    }
    __syncthreads();
    shift_final = shift;
    if (s_done) break;
  }
  // selection = keys whose leading digits (key >> shift_final) are below the prefix, plus the first
  // need_eq (in row order) of those equal to it; after all 8 passes shift_final = 0: the exact k-th key
  const unsigned long long kth = s_prefix;
  const int need_eq = s_remaining;

  // ordered compaction (ascending row).  Each wave owns a contiguous quarter of the rows and walks it 64
  // rows at a time (coalesced key reads; ranks inside a step come from ballots): a counting pass, one
  // exchange of the four waves' totals, then the emitting pass.
  {
    const int lane = tid & 63, wave = tid >> 6;
    const int q = ((n + 3) / 4 + 63) / 64 * 64;
    const int w0 = wave * q, w1 = min(n, w0 + q);
    const unsigned long long below = (1ULL << lane) - 1ULL;
    int n_lt = 0, n_eq = 0;
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < SYM_SEL_J; ++j) {
        if (w0 + 64 * j >= w1) break;   // wave-uniform
        const bool in = kr[j] != ~0ULL;
        const unsigned long long key = kr[j] >> shift_final;
        n_lt += __popcll(__ballot(in && key < kth));
        n_eq += __popcll(__ballot(in && key == kth));
      }
    } else {
      for (int i = w0; i < w1; i += 64) {
        const bool in = i + lane < w1;
        const unsigned long long key = in ? keys[i + lane] >> shift_final : ~0ULL;
        n_lt += __popcll(__ballot(in && key < kth));
        n_eq += __popcll(__ballot(in && key == kth));
      }
    }
    if (lane == 0) {
      scan_a[wave] = n_lt;
      scan_b[wave] = n_eq;
    }
    __syncthreads();
    int eq_rank = 0, pos = 0;
    for (int w = 0; w < wave; ++w) {
      const int eq_taken = max(0, min(scan_b[w], need_eq - eq_rank));
      pos += scan_a[w] + eq_taken;
      eq_rank += scan_b[w];
    }
    auto emit = [&](int i, bool in, unsigned long long key) {
      const bool lt = in && key < kth, eq = in && key == kth;
      const unsigned long long eqm = __ballot(eq);
      const bool take = lt || (eq && eq_rank + __popcll(eqm & below) < need_eq);
      const unsigned long long tm = __ballot(take);
      const int my = pos + __popcll(tm & below);
      if (take && my < SYM_MAX_NN) {
        pts[my][0] = (double)xyz[(base + i + lane) * 3 + 0];
        pts[my][1] = (double)xyz[(base + i + lane) * 3 + 1];
        pts[my][2] = (double)xyz[(base + i + lane) * 3 + 2];
      }
      pos += __popcll(tm);
      eq_rank += __popcll(eqm);
    };
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < SYM_SEL_J; ++j) {
        if (w0 + 64 * j >= w1) break;   // wave-uniform
        emit(w0 + 64 * j, kr[j] != ~0ULL, kr[j] >> shift_final);
      }
    } else {
      for (int i = w0; i < w1; i += 64) {
        const bool in = i + lane < w1;
        emit(i, in, in ? keys[i + lane] >> shift_final : ~0ULL);
      }
    }
  }
  if (tid == 0) nsel_g[blk] = n_sel;
}

// ---- 3. k-means restarts: one THREAD per (cloud-anchor, restart).  Inside the fused kernel this phase
// ran on 10 lanes of one wave while the workgroup's other 246 threads waited; here every lane works.
//
// The fit is sklearn's KMeans(n_clusters=K, random_state=0, n_init=10).fit(nns) (utils/symmetry.py:216),
// restated from sklearn 1.7.2 (cluster/_kmeans.py: fit, _kmeans_plusplus, _kmeans_single_lloyd):
//   * KMeans.fit hands ONE RandomState(0) to the k-means++ seeding of every restart, and the stream does not
//     depend on the data: restart r consumes the constants KM_DRAWS[r*per, (r+1)*per), per = 1 + (K-1)*trials,
//     trials = 2 + int(log K)  (kmeans_draws.h, tools/gen_kmeans_draws.py) -- so restarts stay independent;
//   * first centre: RandomState.choice(n, p = 1/n) = searchsorted(cdf, u, "right"); every further centre:
//     `trials` candidates by searchsorted(cumsum(closest_d2), u * potential), the candidate with the smallest
//     new potential wins (first on ties);
//   * Lloyd: assign, means (an empty cluster takes the point farthest from its centre), stop when the labels
//     repeat ("strict") or the squared centre shift <= 1e-4 * mean(var(X, axis=0)); without strict
//     convergence the labels are re-assigned once more; inertia from the final centres and labels.
// Arithmetic: f64 on the un-centred points in the oracle's operation order (sklearn: f32, mean-centred) --
// decisions differ from sklearn's only on near-ties: 2 000 / 2 000 fits on real clouds give the same labels
// (tests/test_pins_cpu.py).  k_symcut_finish keeps the first restart, then any with smaller inertia AND a
// different clustering (sklearn's _is_same_clustering).
__global__ __launch_bounds__(256) void k_symcut_kmeans(const int32_t* __restrict__ Ks, int n_anchor,
                                                       int n_blk, int n_init, int max_iter, int per,
                                                       const double* __restrict__ pts_g,
                                                       const int32_t* __restrict__ nsel_g,
                                                       double* __restrict__ km_centers_g,
                                                       double* __restrict__ km_inertia_g,
                                                       unsigned long long* __restrict__ km_labels_g) {
  // dynamic LDS: [per][SYM_MAX_NN][3] points of this workgroup's cloud-anchors, then [n_nn][256] squared distances to the
  // nearest chosen centre.  per = cloud-anchors per workgroup, chosen by the host: as many as the 256 lanes have restarts
  // for and the LDS holds (25 at n_init = 10, n_nn = 50: 250 lanes busy, the 6 400 cloud-anchors of a chair step are ONE
  // round of 256 workgroups; round 3 capped this at 8 = 80 lanes, three rounds)
  extern __shared__ double km_lds[];
  double (*pts_s)[SYM_MAX_NN][3] = reinterpret_cast<double (*)[SYM_MAX_NN][3]>(km_lds);
  double* closest_s = km_lds + (size_t)per * SYM_MAX_NN * 3;
  const int tid = threadIdx.x;
  const int sub = tid / n_init;                          // which of them this thread works on
  const int blk = blockIdx.x * per + sub;
  for (int i = tid; i < per * SYM_MAX_NN * 3; i += 256) {
    const int sb = i / (SYM_MAX_NN * 3);
    const int64_t gb = (int64_t)blockIdx.x * per + sb;
    (&pts_s[0][0][0])[i] = gb < n_blk ? pts_g[gb * SYM_MAX_NN * 3 + (i - sb * SYM_MAX_NN * 3)] : 0.0;
  }
  __syncthreads();
  if (sub >= per || blk >= n_blk) return;
  const int n_sel = nsel_g[blk];
  if (n_sel == 0) return;
  const int K = Ks[blk / n_anchor];
  double (*pts)[3] = pts_s[sub];
  double* closest = closest_s + tid;                     // element i at closest[i * 256]
  const int init_id = tid - sub * n_init;
  const int trials = K == 2 ? 2 : 3;                     // 2 + int(log K), K in {2, 4}
  const double* u = KM_DRAWS + init_id * (1 + (K - 1) * trials);
  KmState st;
#pragma unroll
  for (int c = 0; c < 4; ++c) st.cx[c] = st.cy[c] = st.cz[c] = 0.0;
  // tol = mean(var(X, axis=0)) * 1e-4
  double tol;
  {
    double v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double m = 0.0, q = 0.0;
      for (int i = 0; i < n_sel; ++i) m += pts[i][a];
      m /= (double)n_sel;
      for (int i = 0; i < n_sel; ++i) q = fma(pts[i][a] - m, pts[i][a] - m, q);
      v[a] = q / (double)n_sel;
    }
    tol = ((v[0] + v[1]) + v[2]) / 3.0 * 1e-4;
  }
  // greedy k-means++ seeding
  {
    const double p_uniform = (double)(1.0f / (float)n_sel);
    double cdf_last = 0.0;
    for (int i = 0; i < n_sel; ++i) cdf_last += p_uniform;
    int c0 = n_sel - 1;
    double acc = 0.0;
    for (int i = 0; i < n_sel; ++i) {
      acc += p_uniform;
      if (acc / cdf_last > u[0]) {
        c0 = i;
        break;
      }
    }
    st.cx[0] = pts[c0][0];
    st.cy[0] = pts[c0][1];
    st.cz[0] = pts[c0][2];
  }
  double pot = 0.0;
  for (int i = 0; i < n_sel; ++i) {
    const double d = dist2_3(pts[i][0], pts[i][1], pts[i][2], st.cx[0], st.cy[0], st.cz[0]);
    closest[i * 256] = d;
    pot += d;
  }
#pragma unroll
  for (int c = 1; c < 4; ++c) {
    if (c < K) {
      int best_cand = -1;
      double best_pot = 0.0;
      for (int t = 0; t < trials; ++t) {
        const double rv = u[1 + (c - 1) * trials + t] * pot;
        int cand = n_sel - 1;
        double cum = 0.0;
        for (int i = 0; i < n_sel; ++i) {
          cum += closest[i * 256];
          if (cum >= rv) {
            cand = i;
            break;
          }
        }
        const double qx = pts[cand][0], qy = pts[cand][1], qz = pts[cand][2];
        double pc = 0.0;
        for (int i = 0; i < n_sel; ++i) {
          const double d = dist2_3(pts[i][0], pts[i][1], pts[i][2], qx, qy, qz);
          const double cl = closest[i * 256];
          pc += d < cl ? d : cl;
        }
        if (best_cand < 0 || pc < best_pot) {
          best_cand = cand;
          best_pot = pc;
        }
      }
      st.cx[c] = pts[best_cand][0];
      st.cy[c] = pts[best_cand][1];
      st.cz[c] = pts[best_cand][2];
      pot = best_pot;
      for (int i = 0; i < n_sel; ++i) {
        const double d = dist2_3(pts[i][0], pts[i][1], pts[i][2], st.cx[c], st.cy[c], st.cz[c]);
        if (d < closest[i * 256]) closest[i * 256] = d;
      }
    }
  }
  // Lloyd iterations
  unsigned long long lab_lo = ~0ULL, lab_hi = ~0ULL;  // 2 bits per point, 32 points per word
  bool strict = false;
  for (int it = 0; it < max_iter; ++it) {
    double sx[4] = {0, 0, 0, 0}, sy[4] = {0, 0, 0, 0}, sz[4] = {0, 0, 0, 0};
    int cn[4] = {0, 0, 0, 0};
    unsigned long long nlo = 0, nhi = 0;
    for (int i = 0; i < n_sel; ++i) {
      double dm;
      const double px = pts[i][0], py = pts[i][1], pz = pts[i][2];
      const int b = nearest_center(st, K, px, py, pz, &dm);
      if (i < 32)
        nlo |= (unsigned long long)b << (2 * i);
      else
        nhi |= (unsigned long long)b << (2 * (i - 32));
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (b == c) {
          sx[c] += px;
          sy[c] += py;
          sz[c] += pz;
          cn[c] += 1;
        }
      }
    }
    // empty clusters take the points farthest from their centres (rare: one each, farthest first)
    unsigned long long taken = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < K && cn[c] == 0) {
        int far = -1;
        double fd = -1.0;
        for (int i = 0; i < n_sel; ++i) {
          double dm;
          nearest_center(st, K, pts[i][0], pts[i][1], pts[i][2], &dm);
          if ((taken >> i) & 1ULL) dm = -1.0;
          if (far < 0 || dm > fd) {
            far = i;
            fd = dm;
          }
        }
        taken |= 1ULL << far;
        const int old = (int)((far < 32 ? nlo >> (2 * far) : nhi >> (2 * (far - 32))) & 3ULL);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          if (o == old) {
            sx[o] -= pts[far][0];
            sy[o] -= pts[far][1];
            sz[o] -= pts[far][2];
            cn[o] -= 1;
          }
        }
        sx[c] = pts[far][0];
        sy[c] = pts[far][1];
        sz[c] = pts[far][2];
        cn[c] = 1;
      }
    }
    double shift = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < K) {
        const double nx = cn[c] > 0 ? sx[c] / (double)cn[c] : st.cx[c];
        const double ny = cn[c] > 0 ? sy[c] / (double)cn[c] : st.cy[c];
        const double nz = cn[c] > 0 ? sz[c] / (double)cn[c] : st.cz[c];
        shift += dist2_3(nx, ny, nz, st.cx[c], st.cy[c], st.cz[c]);
        st.cx[c] = nx;
        st.cy[c] = ny;
        st.cz[c] = nz;
      }
    }
    const bool same = it > 0 && nlo == lab_lo && nhi == lab_hi;
    lab_lo = nlo;
    lab_hi = nhi;
    if (same) {
      strict = true;
      break;
    }
    if (shift <= tol) break;
  }
  double inertia = 0.0;
  if (!strict) {
    lab_lo = lab_hi = 0;
    for (int i = 0; i < n_sel; ++i) {
      double dm;
      const int b = nearest_center(st, K, pts[i][0], pts[i][1], pts[i][2], &dm);
      if (i < 32)
        lab_lo |= (unsigned long long)b << (2 * i);
      else
        lab_hi |= (unsigned long long)b << (2 * (i - 32));
      inertia += dm;
    }
  } else {
    for (int i = 0; i < n_sel; ++i) {
      const int b = (int)((i < 32 ? lab_lo >> (2 * i) : lab_hi >> (2 * (i - 32))) & 3ULL);
      double cx = st.cx[0], cy = st.cy[0], cz = st.cz[0];
#pragma unroll
      for (int c = 1; c < 4; ++c)
        if (b == c) {
          cx = st.cx[c];
          cy = st.cy[c];
          cz = st.cz[c];
        }
      inertia += dist2_3(pts[i][0], pts[i][1], pts[i][2], cx, cy, cz);
    }
  }
  const int64_t slot = (int64_t)blk * n_init + init_id;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    km_centers_g[slot * 12 + 3 * c + 0] = st.cx[c];
    km_centers_g[slot * 12 + 3 * c + 1] = st.cy[c];
    km_centers_g[slot * 12 + 3 * c + 2] = st.cz[c];
  }
  km_inertia_g[slot] = inertia;
  km_labels_g[slot * 2 + 0] = lab_lo;
  km_labels_g[slot * 2 + 1] = lab_hi;
}

template <int DIM>
__global__ __launch_bounds__(256) void k_symcut_finish(
    const float* __restrict__ xyz, const int64_t* __restrict__ off, int n_anchor,
    const int32_t* __restrict__ Ks, int n_init, const double* __restrict__ pts_g,
    const int32_t* __restrict__ nsel_g, const double* __restrict__ km_centers_g,
    const double* __restrict__ km_inertia_g, const unsigned long long* __restrict__ km_labels_g,
    double* __restrict__ out_centers, int32_t* __restrict__ out_counts, double* __restrict__ out_min_cdist,
    double* __restrict__ out_max_err) {
  __shared__ double sel_centers[12];
  __shared__ int counts[4];
  const int blk = blockIdx.x;
  const int cloud = blk / n_anchor;
  const int tid = threadIdx.x;
  const int n_sel = nsel_g[blk];
  if (n_sel == 0) return;  // degenerate: outputs written by k_symcut_select
  const int64_t base = off[cloud];
  const int n = (int)(off[cloud + 1] - base);
  const int K = Ks[cloud];
  double* oc = out_centers + (int64_t)blk * 12;
  const double (*pts)[3] = reinterpret_cast<const double (*)[3]>(pts_g + (int64_t)blk * SYM_MAX_NN * 3);
  const double (*km_centers)[12] = reinterpret_cast<const double (*)[12]>(km_centers_g + (int64_t)blk * n_init * 12);
  const double* km_inertia = km_inertia_g + (int64_t)blk * n_init;
  // ---- 4. best restart + gate statistics -----------------------------------------------------
  // Wave 0, one lane per restart / per selected point (n_init <= 10, n_sel <= 64).  Round 5: this section was ONE thread
  // walking the restarts' label words and the selected points with a global load (and its latency) per step while the
  // other 255 waited -- most of the kernel's 208 us per chair step.  The arithmetic is unchanged: what was a sequential
  // f64 sum (the per-cluster mean distance) is still summed by one lane in point order, from LDS.
  __shared__ double s_dist[SYM_MAX_NN];
  __shared__ int s_b[SYM_MAX_NN];
  KmState st;   // centres of the chosen restart (wave 0)
  if (tid < 64) {
    const unsigned long long* km_lab = km_labels_g + (int64_t)blk * n_init * 2;
    const bool has_r = tid < n_init;
    const double my_in = has_r ? km_inertia[tid] : 0.0;
    const unsigned long long my_lo = has_r ? km_lab[2 * tid] : 0ULL, my_hi = has_r ? km_lab[2 * tid + 1] : 0ULL;
    const bool act = tid < n_sel;
    const int sh = 2 * (tid & 31);
    // the first restart, then any with a smaller inertia AND a different clustering (sklearn _is_same_clustering: the
    // one-directional label mapping must be consistent, i.e. the points of one label of r carry ONE label of best)
    int best = 0;
    for (int r = 1; r < n_init; ++r) {
      if (!(__shfl(my_in, r) < __shfl(my_in, best))) continue;   // (wave-uniform)
      // (every shuffle with all 64 lanes active: a source lane that sits out of a divergent arm returns nothing)
      const unsigned long long r_lo = __shfl(my_lo, r), r_hi = __shfl(my_hi, r);
      const unsigned long long b_lo = __shfl(my_lo, best), b_hi = __shfl(my_hi, best);
      const unsigned long long rw = tid < 32 ? r_lo : r_hi;
      const unsigned long long bw = tid < 32 ? b_lo : b_hi;
      const int la = (int)((rw >> sh) & 3ULL), lb = (int)((bw >> sh) & 3ULL);
      bool same_clu = true;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned long long mk = __ballot(act && la == c);
        const int lb_first = __shfl(lb, mk ? __ffsll((long long)mk) - 1 : 0);
        if (__ballot(act && la == c && lb != lb_first)) same_clu = false;
      }
      if (!same_clu) best = r;
    }
    if (tid < 12) {
      const double v = tid < 3 * K ? km_centers[best][tid] : 0.0;
      sel_centers[tid] = v;
      oc[tid] = v;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      st.cx[c] = km_centers[best][3 * c + 0];
      st.cy[c] = km_centers[best][3 * c + 1];
      st.cz[c] = km_centers[best][3 * c + 2];
    }
    if (act) {
      double dm;
      s_b[tid] = nearest_center(st, K, pts[tid][0], pts[tid][1], pts[tid][2], &dm);
      s_dist[tid] = sqrt(dm);
    }
  }
  __syncthreads();
  {
    if (tid == 0) {
      double min_cd = INFINITY;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = c + 1; d < 4; ++d)
          if (d < K) {
            const double dd = sqrt(dist2_3(st.cx[c], st.cy[c], st.cz[c], st.cx[d], st.cy[d], st.cz[d]));
            if (dd < min_cd) min_cd = dd;
          }
      double esum[4] = {0, 0, 0, 0};
      int ecnt[4] = {0, 0, 0, 0};
      for (int i = 0; i < n_sel; ++i) {
        const int bsel = s_b[i];
        const double dist = s_dist[i];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (bsel == c) {
            esum[c] += dist;
            ecnt[c] += 1;
          }
      }
      double max_err = 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < K) {
          const double e = ecnt[c] > 0 ? esum[c] / (double)ecnt[c] : INFINITY;
          if (e > max_err) max_err = e;
        }
      }
      out_min_cdist[blk] = min_cd;
      out_max_err[blk] = max_err;
    }
  }
  if (tid < 4) counts[tid] = 0;
  __syncthreads();
  {
    KmState st;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      st.cx[c] = sel_centers[3 * c + 0];
      st.cy[c] = sel_centers[3 * c + 1];
      st.cz[c] = sel_centers[3 * c + 2];
    }
    int local[4] = {0, 0, 0, 0};
    // four points per thread and trip: their twelve loads are issued together (clamped addresses), not one point per
    // memory round trip
    for (int i0 = tid; i0 < n; i0 += 256 * 4) {
      float p[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        const int64_t g = base + (i < n ? i : i0);
#pragma unroll
        for (int a = 0; a < 3; ++a) p[u][a] = xyz[g * 3 + a];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        double dm;
        const int b = nearest_center(st, K, (double)p[u][0], (double)p[u][1], (double)p[u][2], &dm);
        const bool in = i0 + 256 * u < n;
#pragma unroll
        for (int c = 0; c < 4; ++c) local[c] += (in && b == c);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (local[c]) atomicAdd(&counts[c], local[c]);
  }
  __syncthreads();
  if (tid < 4) out_counts[(int64_t)blk * 4 + tid] = counts[tid];
}

__global__ void k_symcut_labels(const float* __restrict__ xyz, const int64_t* __restrict__ off,
                                const int32_t* __restrict__ Ks,
                                const double* __restrict__ centers, int32_t* __restrict__ labels) {
  const int cloud = blockIdx.y;
  const int64_t base = off[cloud];
  const int n = (int)(off[cloud + 1] - base);
  const int K = Ks[cloud];
  KmState st;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    st.cx[c] = centers[(int64_t)cloud * 12 + 3 * c + 0];
    st.cy[c] = centers[(int64_t)cloud * 12 + 3 * c + 1];
    st.cz[c] = centers[(int64_t)cloud * 12 + 3 * c + 2];
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double dm;
    labels[base + i] = nearest_center(st, K, (double)xyz[(base + i) * 3 + 0],
                                      (double)xyz[(base + i) * 3 + 1],
                                      (double)xyz[(base + i) * 3 + 2], &dm);
  }
}

}  // namespace cs

using namespace cs;

extern "C" {

int cs_symcut_fit(const float* d_feat, int dim, const float* d_xyz, const int64_t* h_off,
                  int n_cloud, const int32_t* d_anchor, int n_anchor, const int32_t* h_K,
                  int n_nn, int n_init, int max_iter, double* d_centers,
                  int32_t* d_counts, double* d_min_center_dist, double* d_max_error,
                  void* stream) {
  const uint64_t seed = 0;   // the draws are sklearn's random_state=0 stream (utils/symmetry.py:216), tabulated
  CS_REQUIRE(d_feat && d_xyz && h_off && d_anchor && h_K && d_centers && d_counts &&
                 d_min_center_dist && d_max_error,
             CS_ERR_INVALID, "cs_symcut_fit: NULL argument");
  CS_REQUIRE(dim == 16 || dim == 32, CS_ERR_UNSUPPORTED, "cs_symcut_fit: dim %d unsupported", dim);
  CS_REQUIRE(n_nn >= 4 && n_nn <= SYM_MAX_NN, CS_ERR_UNSUPPORTED,
             "cs_symcut_fit: n_nn %d not in [4, %d]", n_nn, SYM_MAX_NN);
  CS_REQUIRE(n_init >= 1 && n_init <= SYM_MAX_INIT, CS_ERR_UNSUPPORTED,
             "cs_symcut_fit: n_init %d not in [1, %d]", n_init, SYM_MAX_INIT);
  CS_REQUIRE(n_anchor >= 1 && max_iter >= 1, CS_ERR_INVALID, "cs_symcut_fit: bad counts");
  if (n_cloud <= 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  std::vector<int64_t> off(h_off, h_off + n_cloud + 1), key_off(n_cloud + 1, 0);
  std::vector<int32_t> Ks(h_K, h_K + n_cloud);
  for (int c = 0; c < n_cloud; ++c) {
    CS_REQUIRE(off[c + 1] >= off[c] && off[c + 1] - off[c] < (1LL << 31), CS_ERR_INVALID,
               "cs_symcut_fit: bad segment %d", c);
    CS_REQUIRE(Ks[c] == 2 || Ks[c] == 4, CS_ERR_UNSUPPORTED,
               "cs_symcut_fit: K must be 2 or 4 (utils/symmetry.py:246-257)");
    key_off[c + 1] = key_off[c] + (off[c + 1] - off[c]);
  }
  PoolBuf<int64_t> d_off(n_cloud + 1), d_koff(n_cloud + 1);
  PoolBuf<int32_t> d_K(n_cloud);
  PoolBuf<unsigned long long> keys((size_t)key_off[n_cloud] * n_anchor + 1);
  const size_t n_ca = (size_t)n_cloud * n_anchor;
  PoolBuf<double> pts_g(n_ca * SYM_MAX_NN * 3), kmc_g(n_ca * n_init * 12), kmi_g(n_ca * n_init);
  PoolBuf<int32_t> nsel_g(n_ca);
  PoolBuf<unsigned long long> kml_g(n_ca * n_init * 2);
  CS_REQUIRE(d_off.p && d_koff.p && d_K.p && keys.p && pts_g.p && kmc_g.p && kmi_g.p && nsel_g.p && kml_g.p, CS_ERR_HIP,
             "cs_symcut_fit: scratch allocation failed");
  CS_HIP_CHECK(hipMemcpyAsync(d_off.p, off.data(), sizeof(int64_t) * (n_cloud + 1),
                              hipMemcpyHostToDevice, s));
  CS_HIP_CHECK(hipMemcpyAsync(d_koff.p, key_off.data(), sizeof(int64_t) * (n_cloud + 1),
                              hipMemcpyHostToDevice, s));
  CS_HIP_CHECK(hipMemcpyAsync(d_K.p, Ks.data(), sizeof(int32_t) * n_cloud, hipMemcpyHostToDevice, s));
  {
    ProfScope prof("symcut", s);
    const int n_blk = n_cloud * n_anchor;
    dim3 grid((unsigned)n_blk);
    int64_t n_max = 0;
    for (int c = 0; c < n_cloud; ++c) n_max = std::max<int64_t>(n_max, off[c + 1] - off[c]);
    const dim3 kgrid((unsigned)std::max<int64_t>(ceil_div(n_max, 256), 1), (unsigned)n_cloud);
    if (dim == 16)
      hipLaunchKernelGGL((k_symcut_keys<16>), kgrid, dim3(256), 0, s, d_feat, d_off.p, d_anchor, n_anchor,
                         keys.p, d_koff.p);
    else
      hipLaunchKernelGGL((k_symcut_keys<32>), kgrid, dim3(256), 0, s, d_feat, d_off.p, d_anchor, n_anchor,
                         keys.p, d_koff.p);
    if (dim == 16)
      hipLaunchKernelGGL((k_symcut_select<16>), grid, dim3(256), 0, s, d_feat, d_xyz, d_off.p,
                         d_anchor, n_anchor, d_K.p, n_nn, n_init, max_iter, seed, keys.p,
                         d_koff.p, d_centers, d_counts, d_min_center_dist, d_max_error, pts_g.p, nsel_g.p);
    else
      hipLaunchKernelGGL((k_symcut_select<32>), grid, dim3(256), 0, s, d_feat, d_xyz, d_off.p,
                         d_anchor, n_anchor, d_K.p, n_nn, n_init, max_iter, seed, keys.p,
                         d_koff.p, d_centers, d_counts, d_min_center_dist, d_max_error, pts_g.p, nsel_g.p);
    // dynamic LDS: 100 KB for n_nn = 50 (one f64 per thread and selected voxel) + 1.5 KB per cloud-anchor of the workgroup
    const size_t closest_bytes = sizeof(double) * 256 * (size_t)n_nn;
    const size_t lds_budget = 156 * 1024;
    int per = 256 / n_init;
    const int per_fit = (int)((lds_budget - closest_bytes) / (sizeof(double) * SYM_MAX_NN * 3));
    if (per > per_fit) per = per_fit;
    if (per < 1) per = 1;
    const size_t km_lds_bytes = closest_bytes + sizeof(double) * SYM_MAX_NN * 3 * (size_t)per;
    CS_HIP_CHECK(hipFuncSetAttribute((const void*)k_symcut_kmeans, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget));
    hipLaunchKernelGGL(k_symcut_kmeans, dim3((unsigned)((n_blk + per - 1) / per)), dim3(256), km_lds_bytes, s, d_K.p, n_anchor,
                       n_blk, n_init, max_iter, per, pts_g.p, nsel_g.p, kmc_g.p, kmi_g.p, kml_g.p);
    hipLaunchKernelGGL((k_symcut_finish<16>), grid, dim3(256), 0, s, d_xyz, d_off.p, n_anchor, d_K.p,
                       n_init, pts_g.p, nsel_g.p, kmc_g.p, kmi_g.p, kml_g.p, d_centers, d_counts,
                       d_min_center_dist, d_max_error);
    CS_LAUNCH_CHECK();
  }
  return CS_OK;  // scratch goes back to this thread's stream-ordered cache; outputs are valid in stream order
}

int cs_symcut_labels(const float* d_xyz, const int64_t* h_off, int n_cloud, const int32_t* h_K,
                     const double* d_sel_centers, int32_t* d_labels, void* stream) {
  CS_REQUIRE(d_xyz && h_off && h_K && d_sel_centers && d_labels, CS_ERR_INVALID,
             "cs_symcut_labels: NULL argument");
  if (n_cloud <= 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  PoolBuf<int64_t> d_off(n_cloud + 1);
  PoolBuf<int32_t> d_K(n_cloud);
  CS_REQUIRE(d_off.p && d_K.p, CS_ERR_HIP, "cs_symcut_labels: scratch allocation failed");
  int64_t nmax = 0;
  for (int c = 0; c < n_cloud; ++c) {
    CS_REQUIRE(h_K[c] == 2 || h_K[c] == 4, CS_ERR_UNSUPPORTED, "cs_symcut_labels: K must be 2 or 4");
    if (h_off[c + 1] - h_off[c] > nmax) nmax = h_off[c + 1] - h_off[c];
  }
  CS_HIP_CHECK(hipMemcpyAsync(d_off.p, h_off, sizeof(int64_t) * (n_cloud + 1),
                              hipMemcpyHostToDevice, s));
  CS_HIP_CHECK(hipMemcpyAsync(d_K.p, h_K, sizeof(int32_t) * n_cloud, hipMemcpyHostToDevice, s));
  if (nmax > 0) {
    dim3 grid((unsigned)ceil_div(nmax, 256), (unsigned)n_cloud);
    hipLaunchKernelGGL(k_symcut_labels, grid, dim3(256), 0, s, d_xyz, d_off.p, d_K.p,
                       d_sel_centers, d_labels);
    CS_LAUNCH_CHECK();
  }
  return CS_OK;  // scratch goes back to this thread's stream-ordered cache; outputs are valid in stream order
}

}  // extern "C"
