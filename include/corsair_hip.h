/*
 * corsair_hip.h -- C ABI of libcorsair_hip.so, the MI355X (gfx950) native hot path of
 * CORSAIR inference: sparse-voxel ResUNet operators, descriptor top-k, feature k-NN,
 * one-directional Chamfer, batched correspondence RANSAC and the symmetry part-cut.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own:
 * its hot path sits behind the MinkowskiEngine Python operator API and three Python function
 * seams (utils/retrieval.py, utils/eval_pose.py, utils/preprocess.py).  Every entry point below
 * names the reference interface it replaces (file:line relative to the reference tree).
 *
 * Conventions
 *   - plain C, no torch types; every pointer prefixed d_ is a DEVICE pointer owned by the caller
 *     (e.g. tensor.data_ptr()); h_ is a host pointer.  The library owns only the opaque map
 *     handles and an internal scratch pool.
 *   - every call takes the hipStream_t to launch on as a void* (0 = default stream).  Work is
 *     enqueued in stream order: device outputs (d_*) are valid once the stream reaches that point
 *     (synchronise it before reading them on the host); host outputs (h_*, return values, sizes of
 *     maps) are valid on return.  Scratch memory is cached per host thread and handed out again in
 *     stream order, so a thread should keep to one stream (switching streams drains the old one);
 *     different threads may drive different streams concurrently.
 *   - return value 0 = success, negative = error; cs_last_error() returns the (thread-local)
 *     message.  The Python host raises RuntimeError(cs_last_error()), mirroring ME's
 *     RuntimeError on coordinate-key mismatch.
 *   - rows of feature matrices are addressed with an explicit leading dimension (ld, in
 *     elements) so a consumer can write straight into a slice of a wider buffer (this is how
 *     ME.cat -- model/resunet.py:239,246,253 -- is made free).
 */
#ifndef CORSAIR_HIP_H
#define CORSAIR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_OK 0
#define CS_ERR_INVALID (-1)
#define CS_ERR_HIP (-2)
#define CS_ERR_RANGE (-3)
#define CS_ERR_DUPLICATE (-4)
#define CS_ERR_UNSUPPORTED (-5)
#define CS_ERR_INTERNAL (-6) /* a self-check of the library failed (e.g. CS_RANSAC_CHECK) */

typedef struct cs_coordmap cs_coordmap;   /* coordinates of one tensor stride + hash index */
typedef struct cs_kernelmap cs_kernelmap; /* output-stationary neighbour table of one conv  */

const char* cs_last_error(void);
int cs_version(void);
/* number of visible HIP devices (does not initialise a context) */
int cs_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Coordinate maps.  Replaces ME.SparseTensor(feat, coords) coordinate-manager creation
 * (evaluation.py:215-218,246-249) and the strided coordinate generation inside
 * ME.MinkowskiConvolution(stride=2) (model/resunet.py:64-72,80-87,95-103).
 * d_coords: int32 [n,4] rows (batch, x, y, z), unique, |x|,|y|,|z| < 32768, 0 <= batch < 65536.
 * Row order of the created map == input order (evaluation.py:227-229 relies on it).
 * cs_coordmap_stride: output coords = unique rows of floor(c / (s*ts)) * (s*ts), ordered by
 * FIRST OCCURRENCE in input-row order; new tensor stride s*ts.
 * ---------------------------------------------------------------------------------------- */
int cs_coordmap_create(const int32_t* d_coords, int64_t n, int tensor_stride, void* stream,
                       cs_coordmap** out);
int cs_coordmap_stride(const cs_coordmap* in, int stride, void* stream, cs_coordmap** out);
/* All coordinate levels a network will ask for, at once (ME's coordinate manager makes the strided maps one by one as
 * model/resunet.py:64-103 reaches each stride-2 convolution): out[0] = cs_coordmap_create(d_coords, n, tensor_stride),
 * out[l] = cs_coordmap_stride(out[l-1], 2) for l < n_levels <= 4 -- identical coordinates, row order and tables, built from
 * the stride-1 rows in one pass with one host wait.  n_batch > 0 announces rows grouped by sample with batch indices
 * < n_batch (sparse_collate's order, utils/Info/CADLib.py:148-178): the per-sample segments the LDS kernel-map path uses are
 * then made in the same pass (n_batch <= 0: on first use, as for cs_coordmap_create). */
int cs_coordmap_pyramid(const int32_t* d_coords, int64_t n, int tensor_stride, int n_levels, int n_batch, void* stream,
                        cs_coordmap** out);
int64_t cs_coordmap_size(const cs_coordmap* m);
int cs_coordmap_tensor_stride(const cs_coordmap* m);
const int32_t* cs_coordmap_coords(const cs_coordmap* m); /* device int32 [n,4] */
void cs_coordmap_free(cs_coordmap* m);

/* ------------------------------------------------------------------------------------------
 * Kernel maps.  Replaces the kernel-map generation + caching of ME's coordinate manager for
 * kernel_size=3 (27 offsets, k = (dx+1) + 3(dy+1) + 9(dz+1)), dilation 1.
 *   transposed == 0: out row o gathers in rows at  o + delta_k * ts_in   (ts_out in {ts_in, 2 ts_in})
 *   transposed == 1: out row o (fine map) gathers in rows (coarse map) at  o - delta_k * ts_out,
 *                    i.e. the strided map with in/out swapped and the same k
 *                    (ME.MinkowskiConvolutionTranspose, model/resunet.py:110-118,131-139,152-160).
 * The table is int32 [n_out, 27], entry = in row or -1.
 * cs_kernelmap_export writes the canonical (k, in_row, out_row) triples sorted by (k, out_row)
 * for bit-exact parity tests; returns the number of pairs (or negative error).
 * ---------------------------------------------------------------------------------------- */
int cs_kernelmap_build(const cs_coordmap* in, const cs_coordmap* out, int kernel_size,
                       int transposed, void* stream, cs_kernelmap** km);
/* cs_kernelmap_build only enqueues work on `stream`; the pair count reaches the host behind it and
 * cs_kernelmap_num_pairs waits for that copy the first time it is asked (then it is cached). */
/* The n maps of one batch (MinkowskiEngine builds them lazily, one per convolution, through the coordinate manager:
 * model/resunet.py:163-237 is what asks for them): the same maps as n calls of cs_kernelmap_build, built as independent
 * chains on several streams that fork from and join `stream` inside the call (CS_KMAP_STREAMS=1..5, default 4).
 * On failure no map is returned (km[i] = NULL for all i). */
int cs_kernelmap_build_many(int n, const cs_coordmap* const* in, const cs_coordmap* const* out, const int* kernel_size,
                            const int* transposed, void* stream, cs_kernelmap** km);
int64_t cs_kernelmap_num_pairs(const cs_kernelmap* km);
int64_t cs_kernelmap_rows(const cs_kernelmap* km);
const int32_t* cs_kernelmap_table(const cs_kernelmap* km);
int64_t cs_kernelmap_export(const cs_kernelmap* km, int32_t* d_k, int32_t* d_in, int32_t* d_out,
                            int64_t capacity, void* stream);
void cs_kernelmap_free(cs_kernelmap* km);

/* ------------------------------------------------------------------------------------------
 * Sparse convolution forward with fused epilogue.  Replaces ME.MinkowskiConvolution /
 * ME.MinkowskiConvolutionTranspose forward (model/resunet.py:49-193, model/residual_block.py:41-53,
 * model/fc.py:63-71) and, fused, ME.MinkowskiBatchNorm in eval mode (model/common.py:22),
 * MEF.relu (model/resunet.py:212-255) and SparseTensor.__iadd__ (model/residual_block.py:70).
 *   out[o, :] = epilogue( sum_{k ascending} sum_{ci ascending} in[nbr[o][k], ci] * W[k, ci, :] )
 * accumulated as ONE f32 fma chain in exactly that order (f32-input MFMA is such a chain), then
 *   v = scale ? fma(v, scale[c], shift[c]) : (shift ? v + shift[c] : v);
 *   v = residual ? v + residual[o, c] : v;   v = relu ? max(v, 0) : v.
 * km == NULL means kernel_size 1 (identity map, n_out == n_in, W is [cin, cout]).
 * ---------------------------------------------------------------------------------------- */
int cs_conv_fwd(const cs_kernelmap* km, int64_t n_in, int64_t n_out, const float* d_in, int ld_in,
                int cin, const float* d_w, int cout, const float* d_scale, const float* d_shift,
                const float* d_residual, int ld_res, int relu, float* d_out, int ld_out,
                void* stream);

/* EXPERIMENT, off by default (reduced precision "only behind a parity-checked flag").  With the environment
 * variable CS_CONV_SPLIT=3 (or 2) cs_conv_fwd evaluates the layers the LDS-DMA kernel serves on the bf16 matrix cores:
 * every f32 operand cut into 3 (2) bf16 pieces, 6 (3) products per 16 channels accumulated in f32 -- NOT the fma chain
 * above, not bit-identical to the reference's ME kernels; tests/test_gpu_sparse.py bounds the drift, and
 * tools/conv_split_report.py reports speed and ranking differences.  CS_CONV_SPLIT_CACHE=1 keeps a layer's cut weights
 * per weight pointer (valid while the caller keeps the weights alive and unchanged); this call drops them. */
int cs_conv_split_reset(void);

/* Eval-mode batch-norm / bias / residual / ReLU as a stand-alone op (for the unfused, op-by-op
 * MinkowskiEngine-compatible path): same epilogue formula as cs_conv_fwd applied to d_in. */
int cs_affine_act(int64_t n, int c, const float* d_in, int ld_in, const float* d_scale,
                  const float* d_shift, const float* d_residual, int ld_res, int relu,
                  float* d_out, int ld_out, void* stream);

/* Row L2 normalisation out = in / max(||in||_2, eps) (eps = 0 reproduces model/resunet.py:260-262;
 * eps = 1e-12 reproduces nn.functional.normalize at evaluation.py:231,264). */
int cs_row_l2_normalize(int64_t n, int c, const float* d_in, int ld_in, float eps, float* d_out,
                        int ld_out, void* stream);

/* Per-sample column-wise max over rows (model/fc.py:23-29,124-125: split_batch + feat.max(0)).
 * d_batch: int32 batch index of every row, read with stride batch_ld (pass the coords pointer
 * and 4).  d_out: [n_batch, c]; rows of absent samples are -inf. */
int cs_segmented_max(int64_t n, int c, const float* d_in, int ld_in, const int32_t* d_batch,
                     int batch_ld, int n_batch, float* d_out, void* stream);

/* Instance normalisation over the rows of every sample (ME.MinkowskiInstanceNorm, used by the IN
 * variants of the network through model/common.py:23-24; MinkowskiEngine semantics: biased variance,
 * eps 1e-8 added to the variance, affine weight / bias [c]):
 *   out[r, ch] = (in[r, ch] - mean[b, ch]) / sqrt(var[b, ch] + eps) * weight[ch] + bias[ch].
 * d_seg: int32 [n_batch + 1] DEVICE row offsets of the samples (rows grouped by sample, ascending);
 * d_weight / d_bias may be NULL.  Summation order is fixed (see conv.hip) so that the CPU oracle
 * reproduces the result bit for bit.  Returns without waiting for the stream. */
int cs_instance_norm(int64_t n, int c, const float* d_in, int ld_in, const int32_t* d_seg, int n_batch,
                     const float* d_weight, const float* d_bias, float eps, float* d_out, int ld_out,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * Voxel quantisation.  Replaces ME.utils.sparse_quantize(floor(xyz/voxel), return_index=True,
 * return_maps_only=True) (utils/Info/CADLib.py:106-121, datasets/CategoryDataset.py:179-197):
 * for every cloud segment keep the first point of each voxel; kept indices ascending.
 * Grid index = floor(x / voxel) in the cloud's own type (NumPy: `f32 array / python float` divides in f32 —
 * the catalog side, CADLib.py:106-121).  d_xyz f32 [n,3]; h_offsets int64 [n_seg+1] (host); d_keep_idx int64 [n] (capacity n, indices
 * into the concatenated cloud); d_grid int32 [n,4] (batch, x, y, z) of kept rows;
 * h_out_offsets int64 [n_seg+1] (host) receives the kept segment boundaries.
 * ---------------------------------------------------------------------------------------- */
int cs_voxelize(const float* d_xyz, const int64_t* h_offsets, int n_seg, double voxel_size,
                int64_t* d_keep_idx, int32_t* d_grid, int64_t* h_out_offsets, void* stream);
/* The same for f64 clouds, floor(x / voxel) in f64: what the QUERY side of the reference quantises —
 * datasets/CategoryDataset.py:179-197 floors the f64 output of apply_transform (datasets/ScannetDataset.py:
 * 278-282, utils/preprocess.py:39-48), evaluation-shapenet.py:97-119 floors `pc @ R.T + t` (f64) and casts
 * the KEPT points to f32 afterwards (:103-105).  Narrowing such a cloud to f32 first moves a point across a
 * voxel boundary on about 2 % of posed clouds; this entry does not narrow.  d_xyz f64 [n,3]; the rest as
 * cs_voxelize.  The caller casts the kept origins to f32 after the selection, as the reference does. */
int cs_voxelize_f64(const double* d_xyz, const int64_t* h_offsets, int n_seg, double voxel_size,
                    int64_t* d_keep_idx, int32_t* d_grid, int64_t* h_out_offsets, void* stream);

/* ------------------------------------------------------------------------------------------
 * Descriptor retrieval.  Replaces scipy cdist + full argsort at utils/retrieval.py:139-177:
 * for every query the k catalog rows with the smallest Euclidean distance, ascending, ties
 * broken by smaller catalog index.  Distances are evaluated in f64 exactly as
 * sum_c (double(q_c) - double(x_c))^2 (c ascending, fma chain); d_dist receives sqrt of it.
 * d_q f32 [nq,d], d_x f32 [nx,d]; d_idx int64 [nq,k]; d_dist f64 [nq,k] (may be NULL).
 * ---------------------------------------------------------------------------------------- */
int cs_l2_topk(const float* d_q, int64_t nq, const float* d_x, int64_t nx, int d, int k,
               int64_t* d_idx, double* d_dist, void* stream);
/* Same, but d_dist2 receives the SQUARED distances the ranking was made on: what a multi-GPU run merges the
 * per-shard lists with (catalog sharded over ranks, corsair_amd/sharding.py sharded_topk; the reference is
 * single-device).  Merging by (squared distance, global index) reproduces the single-device list bit for bit. */
int cs_l2_topk_sq(const float* d_q, int64_t nq, const float* d_x, int64_t nx, int d, int k, int64_t* d_idx,
                  double* d_dist2, void* stream);
/* Diagnostics of the large-size path (nq * nx >= 2^24, d = 64 / 128 / 256, k <= 10: shortlist on the
 * f16 matrix cores, exact re-score, verification): {queries that took it, queries recomputed by the
 * f64 path because the verification failed (ties at the k-th neighbour)}. */
void cs_l2_topk_stats(uint64_t out[2], int reset);
/* A retrieval run ranks every scan against ONE fixed library (evaluation.py:264-283: `lib_desc` is embedded once,
 * utils/retrieval.py:139-177 ranks against it): the handle keeps what the matrix-core path derives from the catalog
 * (f16 image, f64 norms) across calls; results are those of cs_l2_topk / cs_l2_topk_sq (squared != 0) on the same
 * arrays.  The caller keeps d_x alive and unchanged for the life of the handle. */
typedef struct cs_topk_catalog cs_topk_catalog;
int cs_topk_catalog_create(const float* d_x, int64_t nx, int d, void* stream, cs_topk_catalog** out);
int cs_l2_topk_catalog(const float* d_q, int64_t nq, const cs_topk_catalog* catalog, int k, int64_t* d_idx,
                       double* d_dist, int squared, void* stream);
void cs_topk_catalog_free(cs_topk_catalog* catalog);

/* ------------------------------------------------------------------------------------------
 * Batched feature k-NN.  Replaces find_knn_cpu / KDTree(feat1).query(feat0, k)
 * (utils/find_nn.py:43-49) as used by find_kcorr (utils/eval_pose.py:48-79) and split_corr
 * (utils/symmetry.py:145-179).  The feature matrices are segmented by host offset tables
 * (h_qoff / h_toff); problem p searches, for every row of query segment h_qseg[p], the k nearest
 * rows of target segment h_tseg[p] (f64 squared distance, c ascending fma chain, ties -> smaller
 * index).  Optional part labels (one per feature row) restrict the search (split_corr): target
 * row j is a candidate of query row i iff d_tlabel[j] == d_perm[p*8 + d_qlabel[i]].
 * Output rows are problem-major (all rows of problem 0, then problem 1, ...):
 * d_idx int32 [sum_p nq_p, k] = target row index LOCAL to the target segment, -1 if fewer than k
 * candidates; d_dist f64 same shape, optional (Euclidean distance).
 * ---------------------------------------------------------------------------------------- */
int cs_knn_feat(const float* d_qf, const int64_t* h_qoff, const float* d_tf,
                const int64_t* h_toff, const int32_t* h_qseg, const int32_t* h_tseg, int n_prob,
                int dim, int k, const int32_t* d_qlabel, const int32_t* d_tlabel,
                const int32_t* d_perm, int32_t* d_idx, double* d_dist, void* stream);

/* Diagnostics of cs_knn_feat's default path for 16-d features (f16 matrix-core shortlist, canonical f64
 * re-evaluation, verification, exhaustive recomputation of what could not be verified): out =
 * {queries answered, of those recomputed exhaustively}, counted only while the environment variable
 * CS_KNN_STATS=1.  CS_KNN_MFMA=64 selects the f64 matrix-pipe shortlist, CS_KNN_MFMA=0 the exhaustive
 * all-VALU kernel; all three return the same indices and distances. */
void cs_knn_shortlist_stats(uint64_t out[2], int reset);

/* ------------------------------------------------------------------------------------------
 * One-directional Chamfer.  Replaces apply_transform + chamfer_kdtree_1direction
 * (utils/preprocess.py:39-48,67-70): problem p transforms the source segment with the f32
 * 4x4 row-major matrix d_T[p] (f64 arithmetic), finds for every source point the Euclidean
 * distance to the nearest target point, writes the mean to d_out[p] (f64).
 * Source / target segments are selected per problem by index into the offset tables.
 * ---------------------------------------------------------------------------------------- */
/* Diagnostics of the f16 matrix-core ranking inside cs_chamfer_1dir / cs_hausdorff_1dir (default path): {256-source tiles
 * answered, of those recomputed by the f64 matrix-pipe kernel because a source was within the approximation's error budget
 * of a tie}, counted only while CS_CHAMFER_STATS=1.  CS_CHAMFER_F16=0: the f64 kernel for everything; CS_CHAMFER_MFMA=0: the
 * exhaustive all-VALU chain.  All three return the same canonical distances. */
void cs_chamfer_f16_stats(uint64_t out[2], int reset);
int cs_chamfer_1dir(const float* d_src, const int64_t* h_soff, const float* d_tgt,
                    const int64_t* h_toff, const int32_t* h_src_seg, const int32_t* h_tgt_seg,
                    int n_prob, const float* d_T, double* d_out, void* stream);

/* Directed Hausdorff distance: same arguments, d_out[p] = MAX over source points of the distance to
 * the nearest target point.  Replaces one direction of chamfer_max in the geometric symmetry test
 * of evaluation-shapenet.py:122-155 (get_symmetry_label), SURVEY 8f rank 2. */
int cs_hausdorff_1dir(const float* d_src, const int64_t* h_soff, const float* d_tgt,
                      const int64_t* h_toff, const int32_t* h_src_seg, const int32_t* h_tgt_seg,
                      int n_prob, const float* d_T, double* d_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batched correspondence RANSAC.  Replaces registration_based_on_corr ->
 * o3d.pipelines.registration.registration_ransac_based_on_correspondence(src, tgt, corr,
 * max_corr, ransac_n=10) (utils/eval_pose.py:82-100).  Problem p owns correspondences
 * [h_off[p], h_off[p+1]) of d_src/d_tgt (f32 [M,3], pair i = (src_i, tgt_i)).
 * Semantics = Open3D's loop run on one thread, with a counter-based RNG:
 *   for itr in [0, max_iter): stop if itr >= est_k; sample ransac_n pairs (with replacement,
 *   idx = rng(seed, itr, j) mod-free multiply-shift); T = rigid least-squares fit (no scale);
 *   inliers = #{ |T src_i - tgt_i|^2 < max_corr^2 }, err = sum of inlier squared distances -- T, the
 *   transform of the points and the comparison in f64, as Open3D evaluates them (the reference
 *   converts the points to float64, utils/eval_pose.py:83-86; max_corr is the Python float);
 *   better = more inliers, or equal inliers and smaller err; on improvement
 *   est_k = min(est_k, ceil(log(1-confidence) / log(1 - (inliers/M)^ransac_n))).
 * d_T f32 [n_prob,16] row-major 4x4 = the best f64 transform cast once at the end, where the reference
 * casts it (utils/symmetry.py:274) (identity if no hypothesis had an inlier);
 * d_inliers int32 [n_prob]; d_rmse f64 [n_prob]; d_iters int32 [n_prob] = iterations consumed.
 * ---------------------------------------------------------------------------------------- */
int cs_ransac_batch(const float* d_src, const float* d_tgt, const int64_t* h_off, int n_prob,
                    double max_corr, int ransac_n, int max_iter, double confidence, uint64_t seed,
                    float* d_T, int32_t* d_inliers, double* d_rmse, int32_t* d_iters,
                    void* stream);
/* Diagnostics of the inlier-count prefilter inside cs_ransac_batch (an f16 matrix-core pass that
 * computes an UPPER bound of every hypothesis' inlier count; only hypotheses whose bound reaches
 * the current best are evaluated exactly, so results do not change).  out = {bound violations,
 * hypotheses checked, sum of (bound - exact)} accumulated by runs with the environment variable
 * CS_RANSAC_CHECK=1 (which recomputes every hypothesis exactly and fails with CS_ERR_INTERNAL on a
 * violation), then {survivors, hypotheses generated} of every run.  CS_RANSAC_PREFILTER=0 turns the
 * prefilter off. */
void cs_ransac_prefilter_stats(uint64_t out[5], int reset);

/* ------------------------------------------------------------------------------------------
 * Symmetry part cut.  Replaces symmetric_cut4 (utils/symmetry.py:182-259) for a batch of
 * clouds: for every (cloud, anchor) the 50 feature-nearest voxels, the fit of
 * sklearn's KMeans(n_clusters=K, random_state=0, n_init=n_init) on their xyz (utils/symmetry.py:216:
 * greedy k-means++ on the constant uniform draws of numpy's RandomState(0) -- the reference hard-codes
 * random_state=0, so the entry point takes no seed --, Lloyd with sklearn's stopping rules, first-best
 * restart; n_init <= 10: that many restarts' draws are tabulated, more is CS_ERR_UNSUPPORTED), the
 * statistics the acceptance gate needs, and finally the labels of every voxel under the accepted model.
 *   cs_symcut_fit: d_feat f32 [N,dim], d_xyz f32 [N,3], h_off int64 [n_cloud+1],
 *     d_anchor int32 [n_cloud, n_anchor] (row index local to the cloud), h_K int32 [n_cloud] in {2,4};
 *     outputs per (cloud, anchor): d_centers f64 [.,4,3], d_counts int32 [.,4] (labels of the WHOLE
 *     cloud), d_min_center_dist f64, d_max_error f64 (max over clusters of mean distance of the
 *     50-NN members to their centre).
 *   cs_symcut_labels: labels of every voxel for the chosen centres (d_sel_centers f64 [n_cloud,4,3]),
 *     argmin distance, ties -> smaller centre index.
 * ---------------------------------------------------------------------------------------- */
int cs_symcut_fit(const float* d_feat, int dim, const float* d_xyz, const int64_t* h_off,
                  int n_cloud, const int32_t* d_anchor, int n_anchor, const int32_t* h_K,
                  int n_nn, int n_init, int max_iter, double* d_centers,
                  int32_t* d_counts, double* d_min_center_dist, double* d_max_error,
                  void* stream);
int cs_symcut_labels(const float* d_xyz, const int64_t* h_off, int n_cloud, const int32_t* h_K,
                     const double* d_sel_centers, int32_t* d_labels, void* stream);

/* ------------------------------------------------------------------------------------------
 * Correspondence assembly of sym_pose.  Replaces the index plumbing between find_kcorr / split_corr and
 * registration_based_on_corr (utils/eval_pose.py:48-87, utils/symmetry.py:145-179, 303-356): the reference
 * builds, per part configuration, `xyzA_corrs` / `xyzB_corrs` by boolean-mask gathers and np.concatenate.
 *   cs_partition_by_label: d_label int32 [N] (part label per voxel, clamped to 0..7), d_off int64 [n_cloud+1]
 *     (DEVICE) -> d_order int64 [N]: rows of every cloud stably partitioned by label (parts in order, original
 *     order inside a part).
 *   cs_cfg_bad: d_nn int32 [.., k], d_first int64 [n_cfg+1] (DEVICE, row ranges) -> d_bad int32 [n_cfg]: 1 when a
 *     range holds a negative neighbour (a CAD part with fewer than k voxels; the reference dies there, this
 *     build drops the configuration, DESIGN "Deliberate differences").
 *   cs_corr_assemble: d_desc int64 [n_cfg,5] (DEVICE) = {q_first, n_first, t_first, len, out_first} per
 *     configuration; writes d_src / d_tgt f32 [sum len * k, 3]: query point of d_rows[q_first + i] (d_rows NULL:
 *     row q_first + i) repeated k times against CAD points t_first + d_nn[n_first + i][0..k).  max_len = largest len.
 * ---------------------------------------------------------------------------------------- */
int cs_partition_by_label(const int32_t* d_label, const int64_t* d_off, int n_cloud, int64_t* d_order, void* stream);
int cs_cfg_bad(const int32_t* d_nn, int k, const int64_t* d_first, int n_cfg, int32_t* d_bad, void* stream);
int cs_corr_assemble(const float* d_xyz0, const float* d_xyz1, const int64_t* d_rows, const int32_t* d_nn, int k,
                     const int64_t* d_desc, int n_cfg, int64_t max_len, float* d_src, float* d_tgt, void* stream);

/* ------------------------------------------------------------------------------------------
 * Profiling hooks for bench.py: when enabled the library brackets the launches of each named
 * kernel family with hipEvents on the launch stream and accumulates the elapsed time.
 * names: "conv", "ransac_eval", "ransac_pre", "ransac_hyp", "knn", "chamfer", "topk", "symcut",
 * "kmap".
 * ---------------------------------------------------------------------------------------- */
void cs_prof_enable(int on);
void cs_prof_reset(void);
int cs_prof_get(const char* name, double* total_ms, int64_t* launches);
/* algorithmic work (FLOP) the bracketed launches of the family performed since the last reset */
int cs_prof_get_units(const char* name, double* units);

/* Return the calling thread's cached scratch memory to the HIP runtime.  Scratch is cached per host
 * thread (one stream per thread), so the library may be driven from several threads at once. */
void cs_pool_trim(void);
/* out = {live scratch/handle blocks, blocks freed by a thread other than the one that allocated them,
 * bytes cached by the calling thread}.  A handle (cs_coordmap_free / cs_kernelmap_free) may be dropped on
 * any thread: a foreign block is released with hipFree (device-synchronising), never recycled into the
 * freeing thread's stream-ordered cache. */
void cs_pool_stats(uint64_t out[3]);

#ifdef __cplusplus
}
#endif
#endif /* CORSAIR_HIP_H */
