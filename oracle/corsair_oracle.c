/*
 * corsair_oracle.c -- CPU restatement of the CORSAIR inference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under corsair_amd/ may import, link or call this file; it is
 * used by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py as the checker.
 *
 * Each function restates, in plain scalar C, the arithmetic of one reference stage (file:line
 * relative to the reference tree) in the canonical operation order that DESIGN.md fixes, so the
 * HIP kernels can be compared bit for bit (integers, indices, f32 features) or to 1e-12 (f64 sums
 * whose association differs).  Third-party arithmetic the reference delegates to (MinkowskiEngine
 * 0.5.5 sparse conv, Open3D 160209d0 RANSAC, SciPy cKDTree/cdist, scikit-learn KMeans) is
 * restated from its published algorithm; SciPy/sklearn are present in the container and are
 * used by the Python side of the oracle as cross-checks.  Parity status per stage: DESIGN.md.
 *
 * Build: gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp -shared -fPIC (oracle/native.py).
 */
#include <math.h>
#include <omp.h>
#include "kmeans_draws.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Threads used by the OpenMP loops (independent rows / problems only: results never depend on it). */
void oc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int oc_get_threads(void) { return omp_get_max_threads(); }

/* ------------------------------------------------------------------------------------------
 * Sparse convolution forward + fused epilogue.
 * Restates ME.MinkowskiConvolution / MinkowskiConvolutionTranspose forward as used at
 * model/resunet.py:49-193, model/residual_block.py:41-53, model/fc.py:63-71, followed by eval-mode
 * MinkowskiBatchNorm (model/common.py:22) folded to scale/shift, the residual add
 * (model/residual_block.py:70) and MEF.relu (model/resunet.py:212-255).
 *   out[o,co] = epi( chain_{k asc} chain_{ci asc} fmaf(in[nbr[o][k],ci], W[k,ci,co], acc) )
 * nbr == NULL: 1x1 conv (identity map).
 * ---------------------------------------------------------------------------------------- */
static inline float oc_epilogue(float v, int c, const float* scale, const float* shift,
                                const float* res_row, int relu) {
  if (scale)
    v = fmaf(v, scale[c], shift[c]);
  else if (shift)
    v = v + shift[c];
  if (res_row) v = v + res_row[c];
  if (relu) v = v > 0.0f ? v : 0.0f;
  return v;
}

void oc_conv_fwd(const int32_t* nbr, int kvol, int64_t n_out, const float* in, int ld_in, int cin,
                 const float* w, int cout, const float* scale, const float* shift,
                 const float* residual, int ld_res, int relu, float* out, int ld_out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t o = 0; o < n_out; ++o) {
    float* acc = (float*)malloc(sizeof(float) * (size_t)cout);
    for (int co = 0; co < cout; ++co) acc[co] = 0.0f;
    for (int k = 0; k < kvol; ++k) {
      int64_t src = nbr ? nbr[o * kvol + k] : o;
      if (src < 0) continue;
      const float* x = in + src * ld_in;
      const float* wk = w + (int64_t)k * cin * cout;
      for (int ci = 0; ci < cin; ++ci) {
        const float xv = x[ci];
        const float* wr = wk + (int64_t)ci * cout;
        for (int co = 0; co < cout; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
      }
    }
    const float* res_row = residual ? residual + o * ld_res : 0;
    for (int co = 0; co < cout; ++co)
      out[o * ld_out + co] = oc_epilogue(acc[co], co, scale, shift, res_row, relu);
    free(acc);
  }
}

void oc_affine_act(int64_t n, int c, const float* in, int ld_in, const float* scale,
                   const float* shift, const float* residual, int ld_res, int relu, float* out,
                   int ld_out) {
  for (int64_t r = 0; r < n; ++r) {
    const float* res_row = residual ? residual + r * ld_res : 0;
    for (int col = 0; col < c; ++col)
      out[r * ld_out + col] = oc_epilogue(in[r * ld_in + col], col, scale, shift, res_row, relu);
  }
}

/* Row L2 normalisation (model/resunet.py:260-262; evaluation.py:231).  The sum of squares is
 * taken the way the wave kernel takes it: lane l accumulates channels l, l+64, ... with fmaf, then a
 * 6-level xor butterfly (offsets 32..1) adds the 64 lane sums. */
void oc_row_l2_normalize(int64_t n, int c, const float* in, int ld_in, float eps, float* out,
                         int ld_out) {
  for (int64_t r = 0; r < n; ++r) {
    const float* x = in + r * ld_in;
    float lane[64];
    for (int l = 0; l < 64; ++l) {
      float s = 0.0f;
      for (int i = l; i < c; i += 64) s = fmaf(x[i], x[i], s);
      lane[l] = s;
    }
    for (int off = 32; off >= 1; off >>= 1) {
      float nxt[64];
      for (int l = 0; l < 64; ++l) nxt[l] = lane[l] + lane[l ^ off];
      memcpy(lane, nxt, sizeof(lane));
    }
    float nrm = sqrtf(lane[0]);
    if (nrm < eps) nrm = eps;
    for (int i = 0; i < c; ++i) out[r * ld_out + i] = x[i] / nrm;
  }
}

/* ------------------------------------------------------------------------------------------
 * f64 squared-distance matrix, fma chain over the feature dimension in ascending order.
 * Restates scipy cdist at utils/retrieval.py:175 (squared; the caller takes sqrt / argsort).
 * ---------------------------------------------------------------------------------------- */
void oc_dist2_matrix(const float* q, int64_t nq, const float* x, int64_t nx, int d, double* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nq; ++i)
    for (int64_t j = 0; j < nx; ++j) {
      double acc = 0.0;
      for (int c = 0; c < d; ++c) {
        double diff = (double)q[i * d + c] - (double)x[j * d + c];
        acc = fma(diff, diff, acc);
      }
      out[i * nx + j] = acc;
    }
}

/* ------------------------------------------------------------------------------------------
 * Brute-force k-NN in feature space (restates KDTree(feat1).query(feat0, k), utils/find_nn.py:43-49;
 * optional part labels restate split_corr, utils/symmetry.py:145-179).  Ties -> smaller index.
 * idx int32 [nq,k] (-1 when fewer than k candidates); dist f64 [nq,k] optional (Euclidean).
 * ---------------------------------------------------------------------------------------- */
void oc_knn(const float* qf, int64_t nq, const float* tf, int64_t nt, int dim, int k,
            const int32_t* qlabel, const int32_t* tlabel, const int32_t* perm, int32_t* idx,
            double* dist) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < nq; ++i) {
    double bd[64];
    int32_t bi[64];
    for (int j = 0; j < k; ++j) {
      bd[j] = INFINITY;
      bi[j] = -1;
    }
    int want = -1;
    if (qlabel) {
      int ql = qlabel[i];
      want = (ql >= 0 && ql < 8) ? perm[ql] : -2;
    }
    for (int64_t j = 0; j < nt; ++j) {
      if (qlabel && tlabel[j] != want) continue;
      double acc = 0.0;
      for (int c = 0; c < dim; ++c) {
        double diff = (double)qf[i * dim + c] - (double)tf[j * dim + c];
        acc = fma(diff, diff, acc);
      }
      if (acc < bd[k - 1]) {
        int s = k - 1;
        while (s > 0 && acc < bd[s - 1]) {
          bd[s] = bd[s - 1];
          bi[s] = bi[s - 1];
          --s;
        }
        bd[s] = acc;
        bi[s] = (int32_t)j;
      }
    }
    for (int j = 0; j < k; ++j) {
      idx[i * k + j] = bi[j];
      if (dist) dist[i * k + j] = bi[j] >= 0 ? sqrt(bd[j]) : INFINITY;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * One-directional Chamfer (restates apply_transform + chamfer_kdtree_1direction,
 * utils/preprocess.py:39-48,67-70).  T: f32 row-major 4x4.  Returns mean nearest distance.
 * ---------------------------------------------------------------------------------------- */
double oc_chamfer_1dir(const float* src, int64_t ns, const float* tgt, int64_t nt, const float* T) {
  if (ns == 0) return NAN;
  double* nd = (double*)malloc(sizeof(double) * (size_t)ns);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < ns; ++i) {
    double x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
    double px = fma((double)T[0], x, fma((double)T[1], y, fma((double)T[2], z, (double)T[3])));
    double py = fma((double)T[4], x, fma((double)T[5], y, fma((double)T[6], z, (double)T[7])));
    double pz = fma((double)T[8], x, fma((double)T[9], y, fma((double)T[10], z, (double)T[11])));
    double best = INFINITY;
    for (int64_t j = 0; j < nt; ++j) {
      double dx = px - (double)tgt[3 * j], dy = py - (double)tgt[3 * j + 1],
             dz = pz - (double)tgt[3 * j + 2];
      double d = fma(dz, dz, fma(dy, dy, dx * dx));
      if (d < best) best = d;
    }
    nd[i] = sqrt(best);
  }
  double s = 0.0;
  for (int64_t i = 0; i < ns; ++i) s += nd[i];
  free(nd);
  return s / (double)ns;
}

/* ------------------------------------------------------------------------------------------
 * Counter-based RNG shared by RANSAC sampling and k-means seeding (splitmix64 finaliser).
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t oc_rng_u64(uint64_t seed, uint64_t a, uint64_t b) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ULL * (a * 64ULL + b + 1ULL);
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  x = x ^ (x >> 31);
  return x;
}
static inline uint32_t oc_rng_index(uint64_t seed, uint64_t a, uint64_t b, uint32_t m) {
  return (uint32_t)(((oc_rng_u64(seed, a, b) >> 32) * (uint64_t)m) >> 32);
}

void oc_rng_indices(uint64_t seed, uint64_t itr, int n, uint32_t m, int32_t* out) {
  for (int j = 0; j < n; ++j) out[j] = (int32_t)oc_rng_index(seed, itr, (uint64_t)j, m);
}

/* ------------------------------------------------------------------------------------------
 * Rigid least-squares fit without scale (restates Open3D TransformationEstimationPointToPoint(false)
 * = Eigen::umeyama(src, tgt, false), called from registration_ransac_based_on_correspondence,
 * utils/eval_pose.py:95-97).  Closed form via Horn's unit quaternion: the rotation is the
 * eigenvector of the largest eigenvalue of the 4x4 matrix N built from the cross-covariance;
 * solved with 5 cyclic Jacobi sweeps (worst relative off-diagonal after 5: 2.5e-12, five orders below the
 * f32 rounding of the stored hypothesis; after 6: round-off).  Rotation angle from h = (aqq - app) / 2 and
 * g = apq: t = sgn(h) g / (|h| + sqrt(h^2 + g^2)), i.e. the textbook sgn(theta) / (|theta| + sqrt(theta^2
 * + 1)) with theta = h / g without that division (h = 0 gives t = +1).  Equal to the SVD/Umeyama optimum
 * whenever that is unique.
 * ---------------------------------------------------------------------------------------- */
static void oc_jacobi4(double a[4][4], double v[4][4]) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 5; ++sweep)
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        const double apq = a[p][q];
        if (apq == 0.0) continue;
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
        for (int r = 0; r < 4; ++r) {
          if (r == p || r == q) continue;
          const double arp = a[r][p], arq = a[r][q];
          const double nrp = c * arp - s * arq;
          const double nrq = s * arp + c * arq;
          a[r][p] = nrp;
          a[p][r] = nrp;
          a[r][q] = nrq;
          a[q][r] = nrq;
        }
        for (int r = 0; r < 4; ++r) {
          const double vrp = v[r][p], vrq = v[r][q];
          v[r][p] = c * vrp - s * vrq;
          v[r][q] = s * vrp + c * vrq;
        }
      }
}

/* Largest eigenpair of the Horn matrix without iterating on the matrix (round 4; the same operation
 * sequence as horn_qcp in corsair_amd/csrc/ransac.hip).  N is symmetric and traceless, so its characteristic
 * polynomial is l^4 + c2 l^2 + c1 l + c0 with c2 = -2 |S|_F^2, c1 = -8 det S, c0 = det N (Theobald's QCP
 * observation).  All four roots are real, so Halley's iteration started at the upper bound sqrt(3) |S|_F
 * (>= sigma1 + sigma2 + sigma3 >= l_max) decreases monotonically onto the largest root with cubic order;
 * it stops after the step whose size is below 1e-6 l (the next error is the cube of that).  The eigenvector
 * is the row of adj(N - l I) with the largest diagonal entry (adj = kappa v v^T: that row is kappa v_k v
 * with |v_k| >= 1/2).  The pair is ACCEPTED only when the iteration converged within 8 steps and
 * P'(l) = prod(l - l_j) >= 0.02 l^3, i.e. the largest eigenvalue is well separated; otherwise (near-collinear
 * samples, reflections with sigma2 = sigma3, S = 0, non-finite input: 3 in 10^5 ten-point samples of the bench
 * workload) the caller runs the Jacobi solver above.  Measured on 5.4 M samples against LAPACK: eigenvector
 * within 3.2e-12, eigenvalue within 4.3e-14 relative on the accepted ones.  Returns 1 when q is valid. */
static int oc_horn_qcp(const double S[3][3], const double N[4][4], double q[4]) {
  double f2 = 0.0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) f2 = fma(S[a][b], S[a][b], f2);
  const double c2 = -2.0 * f2;
  const double detS = S[0][0] * (S[1][1] * S[2][2] - S[1][2] * S[2][1]) -
                      S[0][1] * (S[1][0] * S[2][2] - S[1][2] * S[2][0]) +
                      S[0][2] * (S[1][0] * S[2][1] - S[1][1] * S[2][0]);
  const double c1 = -8.0 * detS;
  /* 2x2 minors of rows (0,1) and (2,3): only the diagonal changes between N and N - l I */
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
  int conv = 0;
  for (int it = 0; it < 8 && !conv; ++it) {
    const double l2 = lam * lam;
    const double P = fma(fma(l2 + c2, lam, c1), lam, c0);
    const double dP = fma(fma(4.0, l2, 2.0 * c2), lam, c1);
    const double ddP = fma(12.0, l2, 2.0 * c2);
    const double d = (2.0 * P * dP) / fma(2.0 * dP, dP, -(P * ddP));
    lam = lam - d;
    conv = fabs(d) <= 1e-6 * lam; /* false for NaN */
  }
  {
    const double l2 = lam * lam;
    const double dP = fma(fma(4.0, l2, 2.0 * c2), lam, c1);
    if (!(conv && dP >= 0.02 * (l2 * lam))) return 0;
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
  /* the four rows of the adjugate (symmetric: the off-diagonal entries are computed once) */
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
  int k = 0;
  double best = fabs(a00);
  if (fabs(a11) > best) { best = fabs(a11); k = 1; }
  if (fabs(a22) > best) { best = fabs(a22); k = 2; }
  if (fabs(a33) > best) { best = fabs(a33); k = 3; }
  if (!(best > 0.0)) return 0;
  if (k == 0) { q[0] = a00; q[1] = a01; q[2] = a02; q[3] = a03; }
  else if (k == 1) { q[0] = a01; q[1] = a11; q[2] = a12; q[3] = a13; }
  else if (k == 2) { q[0] = a02; q[1] = a12; q[2] = a22; q[3] = a23; }
  else { q[0] = a03; q[1] = a13; q[2] = a23; q[3] = a33; }
  return 1;
}

static int oc_force_jacobi = 0; /* tests: the fallback alone (ORACLE-side switch only) */
void oc_rigid_fit_force_jacobi(int on) { oc_force_jacobi = on; }

/* ps, pt: n x 3 doubles.  R row-major 3x3, t 3.  Returns the eigen-solver used: 0 = closed-form
 * characteristic polynomial (oc_horn_qcp), 1 = Jacobi fallback. */
int oc_rigid_fit2(const double* ps, const double* pt, int n, double* R, double* t) {
  double cs[3] = {0, 0, 0}, ct[3] = {0, 0, 0};
  for (int j = 0; j < n; ++j)
    for (int a = 0; a < 3; ++a) {
      cs[a] += ps[3 * j + a];
      ct[a] += pt[3 * j + a];
    }
  for (int a = 0; a < 3; ++a) {
    cs[a] = cs[a] / (double)n;
    ct[a] = ct[a] / (double)n;
  }
  double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int j = 0; j < n; ++j) {
    double ds[3], dt[3];
    for (int a = 0; a < 3; ++a) {
      ds[a] = ps[3 * j + a] - cs[a];
      dt[a] = pt[3 * j + a] - ct[a];
    }
    for (int a = 0; a < 3; ++a)
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
  double qv[4];
  int path = 0;
  if (oc_force_jacobi || !oc_horn_qcp(S, N, qv)) {
    path = 1;
    oc_jacobi4(N, V);
    int m = 0;
    for (int c = 1; c < 4; ++c)
      if (N[c][c] > N[m][m]) m = c;
    for (int r = 0; r < 4; ++r) qv[r] = V[r][m];
  }
  double qw = qv[0], qx = qv[1], qy = qv[2], qz = qv[3];
  const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  qw = qw / qn;
  qx = qx / qn;
  qy = qy / qn;
  qz = qz / qn;
  R[0] = 1.0 - 2.0 * (qy * qy + qz * qz);
  R[1] = 2.0 * (qx * qy - qw * qz);
  R[2] = 2.0 * (qx * qz + qw * qy);
  R[3] = 2.0 * (qx * qy + qw * qz);
  R[4] = 1.0 - 2.0 * (qx * qx + qz * qz);
  R[5] = 2.0 * (qy * qz - qw * qx);
  R[6] = 2.0 * (qx * qz - qw * qy);
  R[7] = 2.0 * (qy * qz + qw * qx);
  R[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
  for (int a = 0; a < 3; ++a)
    t[a] = ct[a] - (R[3 * a + 0] * cs[0] + R[3 * a + 1] * cs[1] + R[3 * a + 2] * cs[2]);
  return path;
}

void oc_rigid_fit(const double* ps, const double* pt, int n, double* R, double* t) {
  (void)oc_rigid_fit2(ps, pt, n, R, t);
}

/* ------------------------------------------------------------------------------------------
 * Correspondence RANSAC, one problem, sequential.  Restates Open3D
 * RegistrationRANSACBasedOnCorrespondence as the reference calls it (utils/eval_pose.py:82-100:
 * identity correspondences, ransac_n = 10, defaults max_iteration 100000 / confidence 0.999,
 * point-to-point without scaling, no checkers) run on ONE thread, with the counter-based RNG.
 * Evaluation arithmetic in f64 like Open3D's (the reference hands it float64 points,
 * utils/eval_pose.py:83-86; Matrix4d * Vector4d, squaredNorm and the comparison with
 * max_correspondence_distance^2 are double), inlier error in fixed point (exact, order-free sums).
 * The best transform is cast to f32 at the end, where the reference casts it (utils/symmetry.py:274).
 * ---------------------------------------------------------------------------------------- */
void oc_ransac(const float* src, const float* tgt, int64_t m, double max_corr, int ransac_n,
               int max_iter, double confidence, uint64_t seed, float* T16, int32_t* inliers,
               double* rmse, int32_t* iters) {
  for (int c = 0; c < 16; ++c) T16[c] = (c % 5 == 0) ? 1.0f : 0.0f;
  *inliers = 0;
  *rmse = 0.0;
  *iters = 0;
  if (m < ransac_n) return;
  const double thr2 = max_corr * max_corr;
  int ex = 0;
  (void)frexp(thr2, &ex);
  const double scale = ldexp(1.0, 38 - ex); /* terms < 2^38, fewer than 2^24 pairs: the u64 sum is exact */
  const double log_1mc = log(1.0 - confidence);
  int est_k = max_iter;
  int best_cnt = 0;
  uint64_t best_err = 0;
  double bestT[12];
  double ps[64 * 3], pt[64 * 3];
  int itr = 0;
  for (; itr < max_iter; ++itr) {
    if (itr >= est_k) break;
    for (int j = 0; j < ransac_n; ++j) {
      const int64_t i = oc_rng_index(seed, (uint64_t)itr, (uint64_t)j, (uint32_t)m);
      for (int a = 0; a < 3; ++a) {
        ps[3 * j + a] = (double)src[3 * i + a];
        pt[3 * j + a] = (double)tgt[3 * i + a];
      }
    }
    double Rd[9], td[3];
    oc_rigid_fit(ps, pt, ransac_n, Rd, td);
    const double* R = Rd;
    const double* t = td;
    int cnt = 0;
    uint64_t err = 0;
    /* integer sums: the thread split does not change the result */
#pragma omp parallel for reduction(+ : cnt, err) schedule(static) if (m >= 8192)
    for (int64_t i = 0; i < m; ++i) {
      const double sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
      /* canonical chain: p = R s + t accumulated from the translation, d = p - q, all f64 */
      const double dx = fma(R[2], sz, fma(R[1], sy, fma(R[0], sx, t[0]))) - (double)tgt[3 * i];
      const double dy = fma(R[5], sz, fma(R[4], sy, fma(R[3], sx, t[1]))) - (double)tgt[3 * i + 1];
      const double dz = fma(R[8], sz, fma(R[7], sy, fma(R[6], sx, t[2]))) - (double)tgt[3 * i + 2];
      const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
      if (d2 < thr2) {
        cnt += 1;
        err += (uint64_t)(d2 * scale);
      }
    }
    if (cnt > best_cnt || (cnt == best_cnt && cnt > 0 && err < best_err)) {
      best_cnt = cnt;
      best_err = err;
      for (int a = 0; a < 3; ++a) {
        bestT[4 * a + 0] = R[3 * a + 0];
        bestT[4 * a + 1] = R[3 * a + 1];
        bestT[4 * a + 2] = R[3 * a + 2];
        bestT[4 * a + 3] = t[a];
      }
      double ratio = (double)cnt / (double)m;
      if (ratio > 1.0) ratio = 1.0;
      double pw = 1.0;
      for (int j = 0; j < ransac_n; ++j) pw = pw * ratio;
      const double den = log(1.0 - pw);
      if (den < 0.0) {
        const double est = log_1mc / den;
        if (est < (double)est_k) est_k = (int)ceil(est);
      }
    }
  }
  *iters = itr;
  *inliers = best_cnt;
  if (best_cnt > 0) {
    for (int c = 0; c < 12; ++c) T16[c] = (float)bestT[c];
    *rmse = sqrt(((double)best_err / scale) / (double)best_cnt);
  }
}

/* Batched front end used for CPU-baseline timing: problems are independent, one per OpenMP task. */
void oc_ransac_batch(const float* src, const float* tgt, const int64_t* off, int n_prob,
                     double max_corr, int ransac_n, int max_iter, double confidence, uint64_t seed,
                     float* T, int32_t* inliers, double* rmse, int32_t* iters) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int p = 0; p < n_prob; ++p)
    oc_ransac(src + 3 * off[p], tgt + 3 * off[p], off[p + 1] - off[p], max_corr, ransac_n,
              max_iter, confidence, seed, T + 16 * p, inliers + p, rmse + p, iters + p);
}

/* ------------------------------------------------------------------------------------------
 * Symmetry part cut statistics for one (cloud, anchor).  Restates the body of the anchor loop of
 * symmetric_cut4 (utils/symmetry.py:198-236); sklearn.KMeans(n_clusters=K, random_state=0, n_init=10)
 * is restated on its own constant RandomState(0) draws (kmeans_draws.h) and pinned against sklearn
 * itself by tests/test_pins_cpu.py (2 000 / 2 000 identical label vectors on real clouds).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  double d;
  int32_t i;
} oc_key;
static int oc_key_cmp(const void* a, const void* b) {
  const oc_key* x = (const oc_key*)a;
  const oc_key* y = (const oc_key*)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return (x->i > y->i) - (x->i < y->i);
}
static int oc_int_cmp(const void* a, const void* b) {
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}
static inline double oc_d2(const double* a, const double* b) {
  const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return fma(dz, dz, fma(dy, dy, dx * dx));
}
static int oc_nearest(const double cen[4][3], int K, const double* p, double* dmin) {
  int best = 0;
  double bd = oc_d2(p, cen[0]);
  for (int c = 1; c < K; ++c) {
    const double d = oc_d2(p, cen[c]);
    if (d < bd) {
      bd = d;
      best = c;
    }
  }
  *dmin = bd;
  return best;
}

void oc_symcut_fit_one(const float* feat, int dim, const float* xyz, int n, int anchor, int K,
                       int n_nn, int n_init, int max_iter, uint64_t seed, double* centers /*12*/,
                       int32_t* counts /*4*/, double* min_cdist, double* max_err,
                       int32_t* nn_rows /* n_nn, optional */) {
  for (int c = 0; c < 12; ++c) centers[c] = 0.0;
  for (int c = 0; c < 4; ++c) counts[c] = 0;
  *min_cdist = 0.0;
  *max_err = INFINITY;
  const int n_sel = n < n_nn ? n : n_nn;
  if (n_sel < K || n == 0) return;
  oc_key* keys = (oc_key*)malloc(sizeof(oc_key) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    double d = 0.0;
    for (int c = 0; c < dim; ++c) {
      const double diff = (double)feat[(int64_t)anchor * dim + c] - (double)feat[(int64_t)i * dim + c];
      d = fma(diff, diff, d);
    }
    keys[i].d = d;
    keys[i].i = i;
  }
  qsort(keys, (size_t)n, sizeof(oc_key), oc_key_cmp);
  int32_t rows[64];
  for (int i = 0; i < n_sel; ++i) rows[i] = keys[i].i;
  free(keys);
  qsort(rows, (size_t)n_sel, sizeof(int32_t), oc_int_cmp); /* raw_pc[local_rank < 50]: row order */
  if (nn_rows)
    for (int i = 0; i < n_nn; ++i) nn_rows[i] = i < n_sel ? rows[i] : -1;
  double pts[64][3];
  for (int i = 0; i < n_sel; ++i)
    for (int a = 0; a < 3; ++a) pts[i][a] = (double)xyz[3 * (int64_t)rows[i] + a];

  /* ---- sklearn KMeans(n_clusters=K, random_state=0, n_init=10).fit(nns)  (utils/symmetry.py:216) ----
   * restated from sklearn 1.7.2 cluster/_kmeans.py (fit :1453-1535, _kmeans_plusplus :163-250,
   * _kmeans_single_lloyd :624-742, _relocate_empty_clusters_dense in _k_means_common.pyx); the uniform
   * draws of its RandomState(0) stream are constants (kmeans_draws.h).  Arithmetic here is f64 on the
   * un-centred points (sklearn: f32, mean-centred): decisions can only differ on near-ties. */
  const int trials = 2 + (int)log((double)K);            /* n_local_trials */
  const int per = 1 + (K - 1) * trials;                  /* draws per restart */
  if (n_init * per > KM_N_DRAWS) n_init = KM_N_DRAWS / per;
  (void)seed;                                            /* random_state=0 is the only tabulated stream */
  /* tol = mean(var(X, axis=0)) * 1e-4 */
  double tol;
  {
    double v[3];
    for (int a = 0; a < 3; ++a) {
      double m = 0.0, q = 0.0;
      for (int i = 0; i < n_sel; ++i) m += pts[i][a];
      m /= (double)n_sel;
      for (int i = 0; i < n_sel; ++i) q = fma(pts[i][a] - m, pts[i][a] - m, q);
      v[a] = q / (double)n_sel;
    }
    tol = ((v[0] + v[1]) + v[2]) / 3.0 * 1e-4;
  }
  /* RandomState.choice(n, p = 1/n): cdf = cumsum(p) / cumsum(p)[-1], searchsorted(side="right") */
  const double p_uniform = (double)(1.0f / (float)n_sel);
  double cdf_last = 0.0;
  for (int i = 0; i < n_sel; ++i) cdf_last += p_uniform;

  double best_cen[4][3];
  int best_lab[64];
  double best_inertia = INFINITY;
  int have_best = 0;
  for (int init = 0; init < n_init; ++init) {
    const double* u = KM_DRAWS + init * per;
    double cen[4][3] = {{0}};
    double closest[64];
    int c0 = n_sel - 1;
    {
      double acc = 0.0;
      for (int i = 0; i < n_sel; ++i) {
        acc += p_uniform;
        if (acc / cdf_last > u[0]) {
          c0 = i;
          break;
        }
      }
    }
    for (int a = 0; a < 3; ++a) cen[0][a] = pts[c0][a];
    double pot = 0.0;
    for (int i = 0; i < n_sel; ++i) {
      closest[i] = oc_d2(pts[i], pts[c0]);
      pot += closest[i];
    }
    for (int c = 1; c < K; ++c) {
      int best_cand = -1;
      double best_pot = 0.0;
      for (int t = 0; t < trials; ++t) {
        const double rv = u[1 + (c - 1) * trials + t] * pot;
        /* np.searchsorted(stable_cumsum(closest), rv) (side="left"), clipped to n - 1 */
        int cand = n_sel - 1;
        double cum = 0.0;
        for (int i = 0; i < n_sel; ++i) {
          cum += closest[i];
          if (cum >= rv) {
            cand = i;
            break;
          }
        }
        double pc = 0.0;
        for (int i = 0; i < n_sel; ++i) {
          const double d = oc_d2(pts[i], pts[cand]);
          pc += d < closest[i] ? d : closest[i];
        }
        if (best_cand < 0 || pc < best_pot) { /* np.argmin: first minimum */
          best_cand = cand;
          best_pot = pc;
        }
      }
      for (int a = 0; a < 3; ++a) cen[c][a] = pts[best_cand][a];
      pot = best_pot;
      for (int i = 0; i < n_sel; ++i) {
        const double d = oc_d2(pts[i], pts[best_cand]);
        if (d < closest[i]) closest[i] = d;
      }
    }
    int lab[64], prev[64];
    for (int i = 0; i < n_sel; ++i) prev[i] = -1;
    int strict = 0;
    for (int it = 0; it < max_iter; ++it) {
      double sum[4][3] = {{0}};
      int cn[4] = {0, 0, 0, 0};
      double dist[64];
      for (int i = 0; i < n_sel; ++i) {
        const int b = oc_nearest(cen, K, pts[i], &dist[i]);
        lab[i] = b;
        for (int a = 0; a < 3; ++a) sum[b][a] += pts[i][a];
        cn[b] += 1;
      }
      /* empty clusters take the points farthest from their centres (one each, farthest first) */
      for (int c = 0; c < K; ++c) {
        if (cn[c] != 0) continue;
        int far = 0;
        for (int i = 1; i < n_sel; ++i)
          if (dist[i] > dist[far]) far = i;
        dist[far] = -1.0;
        const int old = lab[far];
        for (int a = 0; a < 3; ++a) {
          sum[old][a] -= pts[far][a];
          sum[c][a] = pts[far][a];
        }
        cn[c] = 1;
        cn[old] -= 1;
      }
      double shift = 0.0;
      for (int c = 0; c < K; ++c) {
        double nc[3];
        for (int a = 0; a < 3; ++a) nc[a] = cn[c] > 0 ? sum[c][a] / (double)cn[c] : cen[c][a];
        shift += oc_d2(nc, cen[c]);
        for (int a = 0; a < 3; ++a) cen[c][a] = nc[a];
      }
      int same = 1;
      for (int i = 0; i < n_sel && same; ++i) same = lab[i] == prev[i];
      if (same) {
        strict = 1;
        break;
      }
      if (shift <= tol) break;
      memcpy(prev, lab, sizeof(int) * (size_t)n_sel);
    }
    double inertia = 0.0;
    for (int i = 0; i < n_sel; ++i) {
      double dm;
      if (!strict) lab[i] = oc_nearest(cen, K, pts[i], &dm);
      else dm = oc_d2(pts[i], cen[lab[i]]);
      inertia += dm;
    }
    /* keep the first restart, then any with a smaller inertia AND a different clustering
     * (_is_same_clustering: one-directional label mapping) */
    int take = !have_best;
    if (have_best && inertia < best_inertia) {
      int mapping[4] = {-1, -1, -1, -1}, same_clu = 1;
      for (int i = 0; i < n_sel && same_clu; ++i) {
        if (mapping[lab[i]] == -1) mapping[lab[i]] = best_lab[i];
        else if (mapping[lab[i]] != best_lab[i]) same_clu = 0;
      }
      take = !same_clu;
    }
    if (take) {
      have_best = 1;
      best_inertia = inertia;
      memcpy(best_cen, cen, sizeof(cen));
      memcpy(best_lab, lab, sizeof(int) * (size_t)n_sel);
    }
  }
  for (int c = 0; c < K; ++c)
    for (int a = 0; a < 3; ++a) centers[3 * c + a] = best_cen[c][a];
  double mcd = INFINITY;
  for (int c = 0; c < K; ++c)
    for (int d = c + 1; d < K; ++d) {
      const double dd = sqrt(oc_d2(best_cen[c], best_cen[d]));
      if (dd < mcd) mcd = dd;
    }
  double esum[4] = {0, 0, 0, 0};
  int ecnt[4] = {0, 0, 0, 0};
  for (int i = 0; i < n_sel; ++i) {
    double dm;
    const int b = oc_nearest(best_cen, K, pts[i], &dm);
    esum[b] += sqrt(dm);
    ecnt[b] += 1;
  }
  double me = 0.0;
  for (int c = 0; c < K; ++c) {
    const double e = ecnt[c] > 0 ? esum[c] / (double)ecnt[c] : INFINITY;
    if (e > me) me = e;
  }
  *min_cdist = mcd;
  *max_err = me;
  for (int i = 0; i < n; ++i) {
    double p[3] = {(double)xyz[3 * (int64_t)i], (double)xyz[3 * (int64_t)i + 1],
                   (double)xyz[3 * (int64_t)i + 2]};
    double dm;
    counts[oc_nearest(best_cen, K, p, &dm)] += 1;
  }
}

void oc_symcut_fit(const float* feat, int dim, const float* xyz, int n, const int32_t* anchors,
                   int n_anchor, int K, int n_nn, int n_init, int max_iter, uint64_t seed,
                   double* centers, int32_t* counts, double* min_cdist, double* max_err) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int a = 0; a < n_anchor; ++a)
    oc_symcut_fit_one(feat, dim, xyz, n, anchors[a], K, n_nn, n_init, max_iter, seed,
                      centers + 12 * a, counts + 4 * a, min_cdist + a, max_err + a, 0);
}

void oc_symcut_labels(const float* xyz, int n, int K, const double* centers /*12*/,
                      int32_t* labels) {
  double cen[4][3];
  for (int c = 0; c < 4; ++c)
    for (int a = 0; a < 3; ++a) cen[c][a] = centers[3 * c + a];
  for (int i = 0; i < n; ++i) {
    double p[3] = {(double)xyz[3 * (int64_t)i], (double)xyz[3 * (int64_t)i + 1],
                   (double)xyz[3 * (int64_t)i + 2]};
    double dm;
    labels[i] = oc_nearest(cen, K, p, &dm);
  }
}

/* Voxel index of the reference's  np.floor(xyz_f32 / voxel)  (utils/Info/CADLib.py:108-109). */
void oc_voxel_index(const float* xyz, int64_t n, float voxel, int32_t* out) {
  for (int64_t i = 0; i < 3 * n; ++i) out[i] = (int32_t)floorf(xyz[i] / voxel);
}
