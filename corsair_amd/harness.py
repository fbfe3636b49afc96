"""The build's counterpart of the hot part of the reference's evaluation.py:207-383:
catalog / query feature extraction, descriptor retrieval, symmetry-aided registration and the metric
aggregation, on the MI355X-native library.  Inputs are synthetic or arbitrary f32 / f64 clouds;
dataset parsing, checkpoints-on-disk and the GUI of the reference are out of scope.

Differences in structure (not in results) from the reference loops:
  * voxelisation + collate run on the GPU (cs_voxelize) instead of in DataLoader workers;
  * per-sample feature splits are CSR offsets, not 32 boolean-mask gathers per batch
    (evaluation.py:226-229);
  * registrations are batched (registration.sym_pose_batch) instead of one Python iteration each.
"""
from __future__ import annotations

import os

from dataclasses import dataclass, field

import numpy as np
import torch

from . import backend as B
from . import engine as E
from . import registration as R
from .utils.eval_pose import eval_pose


@dataclass
class Config:
    """Hyper-parameters of evaluation.py:41-65 (dataclass defaults there, not CLI flags)."""
    voxel_size: float = 0.03
    k_nn: int = 5
    max_corr: float = 0.2
    random_seed: int = 31
    n_points: int = 10000
    batch_size: int = 32
    # clouds per forward of the network when the caller gives no batch size of its own.  The reference's DataLoader hands the
    # network 32 (= batch_size); the rows of an eval-mode forward do not depend on what else is in the batch, and a 32-cloud
    # batch leaves the stride-4 / 8 layers with fewer workgroups than the chip has CUs: 128 here
    embed_batch_size: int = 128
    ransac_max_iter: int = 100000
    ransac_confidence: float = 0.999


@dataclass
class EmbeddedSet:
    """Features of a set of clouds, packed: voxel features F [sumN,16], origins [sumN,3], offsets
    (host list, len n+1), global descriptors [n,256] (row-normalised)."""
    F: torch.Tensor
    origin: torch.Tensor
    offsets: list
    desc: torch.Tensor

    def __len__(self):
        return len(self.offsets) - 1

    def gather(self, ids):
        """Sub-set (with repetition) as a new packed set; device-side row gather."""
        dev = self.F.device
        off = np.asarray(self.offsets, dtype=np.int64)
        ids = np.asarray(ids, dtype=np.int64)
        lens = off[ids + 1] - off[ids]
        new_off = np.concatenate([[0], np.cumsum(lens)])
        starts = torch.from_numpy(off[ids] - new_off[:-1]).to(dev)
        rows = torch.arange(int(new_off[-1]), device=dev) + torch.repeat_interleave(
            starts, torch.from_numpy(lens).to(dev), output_size=int(new_off[-1]))
        return EmbeddedSet(self.F[rows], self.origin[rows], new_off.tolist(),
                           self.desc[torch.from_numpy(ids).to(dev)])


class Pipeline:
    def __init__(self, state_dict, embedding_state_dict, device="cuda", config=None):
        self.cfg = config or Config()
        self.device = torch.device(device)
        self.engine = E.ResUNetEngine(state_dict, embedding_state_dict, device=self.device)

    # ---- feature extraction (evaluation.py:213-269) ----------------------------------------------
    def embed_batch(self, xyz, offsets):
        """xyz f32 [n,3] device (concatenated raw clouds, already normalised), offsets host list.
        Returns an EmbeddedSet for the batch."""
        return self.embed_batch_raw(xyz, offsets, self.cfg.voxel_size)

    def embed_batch_raw(self, xyz, offsets, voxel_size):
        """xyz f32 or f64: quantised in its own type (backend.voxelize); the kept points of an f64 cloud
        become f32 origins AFTER the selection (evaluation-shapenet.py:103-105; the collate of
        datasets/ChairDataset.py:204-237 does the same to the f64 `rot_coords`)."""
        keep, grid, out_off = B.voxelize(xyz, offsets, voxel_size)
        origin = xyz[keep].to(torch.float32)
        feats = torch.ones((grid.shape[0], 1), dtype=torch.float32, device=xyz.device)
        out, feat8, maps = self.engine.forward(grid, feats, n_batch=len(offsets) - 1)
        return EmbeddedSet(out, origin, out_off, self._descriptors(feat8, maps, len(offsets) - 1))

    def _descriptors(self, feat8, maps, n):
        """Global descriptors [n,256]; a checkpoint without `embedding_state_dict` (evaluation-shapenet.py:283-289 loads
        the network only) gives an empty [n,0] block: registration needs none, retrieval would fail loudly."""
        if self.engine.emb is None:
            return torch.zeros((n, 0), dtype=torch.float32, device=feat8.device)
        return self.engine.embed(feat8, maps, n)

    def embed_groups(self, groups, voxel_size=None):
        """One forward over several groups of clouds that differ in type: groups = [(xyz, offsets), ...], each
        quantised by its own cs_voxelize / cs_voxelize_f64 call, then collated into one batch (sample index =
        position in the concatenation of the groups).  evaluation-shapenet.py:299-310 does this for a model
        (f32, load_pc) and its posed copy (f64, generate_test_pc_pair) in every forward."""
        vs = self.cfg.voxel_size if voxel_size is None else voxel_size
        grids, origins, out_off, base = [], [], [0], 0
        for xyz, offsets in groups:
            keep, grid, off = B.voxelize(xyz, offsets, vs)
            if base:
                grid = grid.clone()
                grid[:, 0] += base
            grids.append(grid)
            origins.append(xyz[keep].to(torch.float32))
            out_off += [out_off[-1] + int(o) for o in off[1:]]
            base += len(offsets) - 1
        grid = torch.cat(grids)
        feats = torch.ones((grid.shape[0], 1), dtype=torch.float32, device=grid.device)
        out, feat8, maps = self.engine.forward(grid, feats, n_batch=base)
        return EmbeddedSet(out, torch.cat(origins), out_off, self._descriptors(feat8, maps, base))

    def embed_clouds(self, clouds, batch_size=None):
        """clouds: list of f32 or f64 [n,3] NumPy arrays (host), one type per call: a cloud is quantised in
        the type it arrives in, so promoting or narrowing here would move points across voxel boundaries.
        Forwards of batch_size clouds (default cfg.embed_batch_size; the reference's DataLoader: 32 -- same rows either way)."""
        bs = batch_size or self.cfg.embed_batch_size
        kinds = {np.asarray(c).dtype for c in clouds}
        if len(kinds) > 1 or not kinds <= {np.dtype(np.float32), np.dtype(np.float64)}:
            raise TypeError("embed_clouds: clouds must all be float32 or all be float64, got %s" % sorted(map(str, kinds)))
        sets = []
        for i in range(0, len(clouds), bs):
            chunk = clouds[i:i + bs]
            xyz = torch.from_numpy(np.concatenate(chunk, 0)).to(self.device)
            off = np.concatenate([[0], np.cumsum([len(c) for c in chunk])]).tolist()
            sets.append(self.embed_batch(xyz, off))
        return concat_sets(sets)

    def voxel_counts(self, clouds, chunk=64):
        """Voxels each cloud quantises to (cs_voxelize / cs_voxelize_f64 only): the weight multi-GPU runs balance their
        shards by (N1 varies 1.6 k - 8 k per model)."""
        counts = []
        for i in range(0, len(clouds), chunk):
            part = [np.asarray(c) for c in clouds[i:i + chunk]]
            off = np.concatenate([[0], np.cumsum([len(c) for c in part])]).tolist()
            _, _, out_off = B.voxelize(torch.from_numpy(np.concatenate(part, 0)).to(self.device), off, self.cfg.voxel_size)
            counts += [int(v) for v in np.diff(out_off)]
        return counts

    # ---- retrieval (evaluation.py:272-283) ----------------------------------------------------------
    def retrieve(self, q_desc, lib_desc, k):
        return B.l2_topk(q_desc, lib_desc, k)

    # ---- registration (evaluation.py:297-331) -----------------------------------------------------
    def register(self, queries, cads, syms, anchor_ids=None, use_symmetry=True, force_gate=False,
                 query_anchors=None):
        """queries / cads: EmbeddedSets of equal length (pair p = query p vs cad p)."""
        c = self.cfg
        return R.sym_pose_batch(queries.F, queries.origin, queries.offsets, cads.F, cads.origin,
                                cads.offsets, syms, c.k_nn, c.max_corr, 0, anchor_ids, 100,
                                c.ransac_max_iter, c.ransac_confidence, use_symmetry, force_gate,
                                query_anchors)


def concat_sets(sets):
    off = [0]
    for s in sets:
        off += [o + off[-1] for o in s.offsets[1:]]
    return EmbeddedSet(torch.cat([s.F for s in sets]), torch.cat([s.origin for s in sets]), off,
                       torch.cat([s.desc for s in sets]))


# ---- metric aggregation (evaluation.py:334-383) -----------------------------------------------------
def aggregate(r_losses, t_losses, chamfer=None):
    """Mean RRE (deg) and RRE <= 5/15/45 deg, mean RTE and RTE <= .02/.05/.10/.15 (fractions),
    mean Chamfer -- the numbers of the README tables (README.md:175-178,215-218)."""
    r = np.asarray(r_losses, np.float64)
    t = np.asarray(t_losses, np.float64)
    out = {
        "rre_mean_deg": float(np.mean(r) / np.pi * 180),
        "rre_5": float(np.sum(r <= 5 / 180 * np.pi) / len(r)),
        "rre_15": float(np.sum(r <= 15 / 180 * np.pi) / len(r)),
        "rre_45": float(np.sum(r <= 45 / 180 * np.pi) / len(r)),
        "rte_mean": float(np.mean(t)),
        "rte_002": float(np.sum(t <= 0.02) / len(t)),
        "rte_005": float(np.sum(t <= 0.05) / len(t)),
        "rte_010": float(np.sum(t <= 0.10) / len(t)),
        "rte_015": float(np.sum(t <= 0.15) / len(t)),
    }
    if chamfer is not None:
        out["chamfer_mean"] = float(np.mean(np.asarray(chamfer, np.float64)))
    return out


def pose_losses(T_est, T0s, T1s, syms):
    """eval_pose over a batch (evaluation.py:319-322).  T_est f32 [P,4,4] (device or host)."""
    if isinstance(T_est, torch.Tensor):
        T_est = T_est.cpu().numpy()
    t_l, r_l = [], []
    for p in range(len(T_est)):
        t, r = eval_pose(T_est[p], T0s[p], T1s[p], int(syms[p]))
        t_l.append(t)
        r_l.append(r)
    return np.asarray(t_l), np.asarray(r_l)


# ---- the evaluation entry (evaluation.py:207-441) ---------------------------------------------------------
@dataclass
class EvalResult:
    """Everything App.__init__ of evaluation.py computes after the model is loaded."""
    stat: dict                      # precision, top1_error, top1_predict, gt (evaluation.py:272-283)
    per_query: dict                 # the nine arrays of evaluation.py:421-441 (cache.NAMES)
    ransac: dict                    # aggregate() of the vanilla RANSAC poses (evaluation.py:334-358)
    sym: dict                       # aggregate() of the symmetry-aided poses
    sym_success_rate: float
    from_cache: bool
    report: str                     # the log block of evaluation.py:359-383


def _report(ransac, sym, rate):
    def block(title, a, tail):
        return (f"\n==================================================================\n"
                f"{title}:\n"
                f"translation error: {a['rte_mean']},\n"
                f"rte 0.02: {a['rte_002']}, rte 0.05: {a['rte_005']}, rte 0.10: {a['rte_010']}, rte 0.15: {a['rte_015']}\n"
                f"------------------------------------------------------------------\n"
                f"rotation error: {a['rre_mean_rad']},\n"
                f"rre 5: {a['rre_5']}, rre 15: {a['rre_15']}, rre 45: {a['rre_45']},\n"
                f"chamfer distance: {a['chamfer_mean']}" + tail)

    return (block("vanilla ransac", ransac, "") +
            block("sym ransac", sym, "\n==================================================================\n") +
            f"\nsym success rate: {rate}")


def run_eval(pipe, catalog, queries, best_match, table, base_T, lib_T, syms, category="chair",
             register_top1=True, cache_dir=None, ignore_cache=False, force_gate=False, batch_size=None, in_flight=1):
    """The reference's evaluation, end to end (evaluation.py:207-441), on the MI355X path.

    catalog / queries: lists of [n,3] clouds (already normalised; f32 catalog clouds as CADLib loads them,
    f64 posed queries as apply_transform leaves them -- each is quantised in its own type) or EmbeddedSets; best_match int
    [Q] (annotated CAD of every query), table f64 [C,C] pairwise Chamfer of the catalog (diag 0),
    base_T [Q,4,4] / lib_T [C,4,4] ground-truth poses (`base_T`, `pos_T` of the datasets), syms int [C].
      1. feature extraction of both sets in batches (evaluation.py:213-269),
      2. scan2cad_retrieval_eval with Precision@M = 0.1 C (evaluation.py:272-283),
      3. registration of every query against its top-1 prediction (or the GT CAD) with sym_pose, and
         eval_pose of both estimates (evaluation.py:297-331) -- batched, not one Python iteration each;
         skipped when the nine cache files exist (evaluation.py:287, _load_data),
      4. aggregation + the log block (evaluation.py:334-383), 5. the result cache (evaluation.py:421-441).
    in_flight: registration batches in flight (host threads x HIP streams; identical results).  Returns an EvalResult."""
    from . import cache as C_

    cfg = pipe.cfg
    bs = batch_size or cfg.batch_size
    ebs = cfg.embed_batch_size   # (batch_size is the REGISTRATION batch; the network's forward has its own size)
    cat = catalog if isinstance(catalog, EmbeddedSet) else pipe.embed_clouds(catalog, ebs)
    qs = queries if isinstance(queries, EmbeddedSet) else pipe.embed_clouds(queries, ebs)
    Q = len(qs)
    best_match = np.asarray(best_match).astype(np.int64)
    syms = np.asarray(syms)
    stat = retrieval_stat(pipe, qs.desc, cat.desc, best_match, table)

    per_query = None if (ignore_cache or cache_dir is None) else C_.load_results(cache_dir, category, register_top1)
    from_cache = per_query is not None
    if per_query is None:
        pos_idx = np.asarray(stat["top1_predict" if register_top1 else "gt"], dtype=np.int64)
        per_query = register_queries(pipe, qs, np.arange(Q), cat, pos_idx, syms, base_T, lib_T, force_gate, bs, in_flight)
        if cache_dir is not None:
            C_.save_results(cache_dir, category, register_top1, per_query)
    return finish_eval(stat, per_query, from_cache)


def retrieval_stat(pipe, q_desc, lib_desc, best_match, table):
    """scan2cad_retrieval_eval with Precision@M, M = int(0.1 C) (evaluation.py:272-283): the ranking comes from
    pipe.retrieve (cs_l2_topk), the statistics from utils.retrieval."""
    from .utils import retrieval

    pos_n = int(0.1 * np.asarray(table).shape[1])
    rank = pipe.retrieve(q_desc, lib_desc, max(pos_n, 1))
    rank = rank.cpu().numpy() if torch.is_tensor(rank) else np.asarray(rank)
    return retrieval.scan2cad_retrieval_eval_rank(rank, table, best_match, pos_n)


# Worker threads (and one torch stream each per device) of the batches-in-flight mode, kept for the life of the process: the
# library gives every host thread its own side streams and scratch cache and HIP maps all streams of a process onto a handful
# of hardware queues, so a fresh set of threads per evaluation (a ThreadPoolExecutor per call, rounds 3-4) kept adding live
# streams -- repeated evaluations in one process (bench.py --scaling strong) then share queues between workers.
_WORKERS = {}
_WORKER_STREAMS = {}


def _worker(w):
    from concurrent.futures import ThreadPoolExecutor

    if w not in _WORKERS:
        _WORKERS[w] = ThreadPoolExecutor(max_workers=1, thread_name_prefix="corsair-register%d" % w)
    return _WORKERS[w]


def _worker_stream(dev, w):
    key = (dev.type, dev.index, w)
    if key not in _WORKER_STREAMS:
        _WORKER_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _WORKER_STREAMS[key]


def register_queries(pipe, qs, query_ids, cat, pos_idx, syms, base_T, lib_T, force_gate=False, batch_size=None,
                     in_flight=1):
    """The registration loop of evaluation.py:297-331 over the embedded queries `qs`, whose GLOBAL query numbers are
    `query_ids` (they seed the anchor draws, so a query gives the same result whichever rank or batch it lands in):
    sym_pose against CAD pos_idx[q] and eval_pose of both estimates, batched.  Returns the nine arrays of
    evaluation.py:421-441 (cache.NAMES) for these queries, in the order of `query_ids`.
    in_flight > 1: that many batches at a time, each on its own host thread and HIP stream (batches are independent;
    the library's scratch cache is per thread and stream-ordered) -- same results, the host work of one batch hides
    behind the kernels of the others (bench.py's `batches_in_flight`)."""
    from . import cache as C_

    bs = batch_size or pipe.cfg.batch_size
    query_ids = np.asarray(query_ids, dtype=np.int64)
    if not len(query_ids):
        return {k: np.zeros((0, 4, 4) if k.startswith("Ts_est") else 0, C_.DTYPES[k]) for k in C_.NAMES}
    batches = [np.arange(s, min(len(query_ids), s + bs)) for s in range(0, len(query_ids), bs)]

    def one(loc):
        ids = query_ids[loc]
        q = qs.gather(loc)
        cads = cat.gather(pos_idx[ids])
        cad_sym = syms[pos_idx[ids]]
        res = pipe.register(q, cads, cad_sym, anchor_ids=[(2 * int(i), 2 * int(i) + 1) for i in ids],
                            force_gate=force_gate)
        Tr, Tb, cdr, cdb = (t.cpu().numpy() for t in (res.T_ransac, res.T_best, res.cd_ransac, res.cd_best))
        T0 = [base_T[i] for i in ids]
        T1 = [lib_T[j] for j in pos_idx[ids]]
        t_r, r_r = pose_losses(Tr, T0, T1, cad_sym)
        t_s, r_s = pose_losses(Tb, T0, T1, cad_sym)
        return {"Ts_est_ransac": Tr, "Ts_est_best": Tb, "t_losses_ransac": t_r, "t_losses_sym": t_s,
                "r_losses_ransac": r_r, "r_losses_sym": r_s, "sym_ransac_success": np.asarray(res.ok),
                "chamfer_dist_ransac": cdr, "chamfer_dist_sym": cdb}

    depth = min(int(in_flight), len(batches))
    on_gpu = torch.device(getattr(pipe, "device", "cpu")).type == "cuda"
    if depth <= 1 or not on_gpu:
        done = [one(loc) for loc in batches]
    else:
        dev = torch.device(pipe.device)
        torch.cuda.synchronize(dev)          # the embedded sets were made on the caller's stream
        done = [None] * len(batches)

        def work(w):
            if dev.index is not None:
                torch.cuda.set_device(dev.index)
            st = _worker_stream(dev, w)
            with torch.cuda.stream(st):
                for i in range(w, len(batches), depth):
                    done[i] = one(batches[i])
                st.synchronize()

        if os.environ.get("CORSAIR_REGISTER_FRESH_THREADS") == "1":     # (A/B: a fresh set of threads per call, as in rounds 3-4)
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(max_workers=depth, thread_name_prefix="corsair-register") as ex:
                for f in [ex.submit(work, w) for w in range(depth)]:
                    f.result()
        else:
            for f in [_worker(w).submit(work, w) for w in range(depth)]:
                f.result()               # surfaces worker failures
    return {k: np.concatenate([np.asarray(d[k]) for d in done]) for k in C_.NAMES}


def finish_eval(stat, per_query, from_cache):
    """Aggregation + the log block (evaluation.py:334-383) over the per-query arrays of ALL queries."""
    ransac = aggregate(per_query["r_losses_ransac"], per_query["t_losses_ransac"], per_query["chamfer_dist_ransac"])
    sym = aggregate(per_query["r_losses_sym"], per_query["t_losses_sym"], per_query["chamfer_dist_sym"])
    for a, r in ((ransac, per_query["r_losses_ransac"]), (sym, per_query["r_losses_sym"])):
        a["rre_mean_rad"] = float(np.mean(np.asarray(r, np.float64)))
    rate = float(np.mean(per_query["sym_ransac_success"]))
    return EvalResult(stat, per_query, ransac, sym, rate, from_cache, _report(ransac, sym, rate))


# ---- synthetic Scan2CAD-shaped workload ---------------------------------------------------------------
@dataclass
class SyntheticScan2CAD:
    """Chair-sized synthetic evaluation set, the input of run_eval when the Scan2CAD annotations are
    not available: C catalog clouds, Q queries = posed copies of catalog clouds (known GT pose),
    symmetry labels with the chair label statistics; `table()` computes the pairwise
    Chamfer table the retrieval metric needs (utils/pc_dist.py, 2000 points per cloud)."""
    n_catalog: int = 652
    n_query: int = 993
    n_points: int = 10000
    catalog: list = field(default_factory=list)
    queries: list = field(default_factory=list)
    query_T: list = field(default_factory=list)
    query_cad: list = field(default_factory=list)
    sym: np.ndarray = None

    def build(self, catalog_ids=None, query_ids=None):
        from . import synth

        cids = list(range(self.n_catalog)) if catalog_ids is None else list(catalog_ids)
        self.catalog = [synth.make_cloud(c, 15000)[: self.n_points] for c in cids]
        self.sym = np.ones(len(cids), np.int32)
        self.sym[::326] = 4  # chair labels: 650 x 1, 2 x 4 (configs/03001627_scan2cad_rot_sym_label.txt)
        qids = list(range(self.n_query)) if query_ids is None else list(query_ids)
        self.queries, self.query_T, self.query_cad = [], [], []
        for q in qids:
            cad = q % len(cids)
            T = synth.random_pose(q, max_trans=0.0)
            # the reference's query is the scan in the fixed test rotation fix_trans[idx,0]
            # (datasets/ScannetDataset.py:274); here: the CAD cloud itself under a seeded rotation,
            # re-sampled (other 10k of the 15k points) so voxel occupancy differs
            pc = synth.make_cloud(cids[cad], 15000)[15000 - self.n_points:]
            self.queries.append(synth.apply_pose(pc, T, np.float64))   # apply_transform yields f64
            self.query_T.append(T)
            self.query_cad.append(cad)
        return self

    def table(self, n_points=2000):
        from .utils import pc_dist

        t = pc_dist.compute_dist([c[:n_points] for c in self.catalog])
        np.fill_diagonal(t, 0.0)                     # datasets/ScannetDataset.py:65-66
        return t

    def eval_inputs(self):
        """(catalog, queries, best_match, base_T, lib_T, syms) for run_eval; the CADs sit in their
        canonical frame (lib_T = identity, like the bench's T1)."""
        return (self.catalog, self.queries, np.asarray(self.query_cad), np.stack(self.query_T),
                np.stack([np.eye(4)] * len(self.catalog)), self.sym)


# ---- file-level entry: `python -m corsair_amd.harness` (evaluation.py:68-129,195-201) -----------------------------------
def load_cloud_dir(path, n_points, what):
    """Sorted *.npy files of a directory, first n_points rows each (load_raw_pc, utils/preprocess.py:27-29 through
    datasets/Reader.py:75-86), in the type they are stored in.  Returns (names, clouds)."""
    import os

    names = sorted(f for f in os.listdir(path) if f.endswith(".npy"))
    if not names:
        raise FileNotFoundError(f"{what}: no .npy cloud in {path}")
    clouds = []
    for f in names:
        a = np.load(os.path.join(path, f))
        if a.ndim != 2 or a.shape[1] < 3:
            raise ValueError(f"{what}: {f} is not an [n,3] cloud (shape {a.shape})")
        a = np.ascontiguousarray(a[:n_points, :3])
        clouds.append(a if a.dtype in (np.float32, np.float64) else a.astype(np.float32))
    return names, clouds


def read_sym_labels(path, names):
    """`<cad path> <label>` per line (configs/*_scan2cad_rot_sym_label.txt, read at evaluation.py:175-179): matched to
    the catalog files by basename when every name occurs, by line order otherwise."""
    import os

    rows = [ln.split() for ln in open(path) if ln.strip()]
    by_name = {os.path.basename(r[0]): int(r[-1]) for r in rows}
    if all(n in by_name for n in names):
        return np.asarray([by_name[n] for n in names], np.int32)
    if len(rows) != len(names):
        raise ValueError(f"{path}: {len(rows)} labels for {len(names)} catalog clouds")
    return np.asarray([int(r[-1]) for r in rows], np.int32)


def build_parser():
    import argparse

    ap = argparse.ArgumentParser(
        prog="python -m corsair_amd.harness",
        description="CORSAIR evaluation on the MI355X path from FILES: a reference checkpoint (utils/ckpts.py format) and "
                    "directories of .npy clouds; the counterpart of `python evaluation.py` (evaluation.py:68-129) with the "
                    "Scan2CAD annotation parsing (out of scope, SURVEY 2) replaced by plain arrays.")
    ap.add_argument("--checkpoint", "--ckpt", required=True, help="torch.save dict with state_dict / embedding_state_dict "
                                                                  "(evaluation.py:195-201)")
    ap.add_argument("--catalog-dir", required=True, help="CAD clouds, one [n,3] .npy each (sorted by name = catalog index)")
    ap.add_argument("--query-dir", required=True, help="scan clouds, one [n,3] .npy each (sorted by name = query index)")
    ap.add_argument("--category", default="table", choices=["table", "chair"])
    ap.add_argument("--query-poses", help=".npy [Q,4,4] or [Q,k,4,4] (configs/fix_trans.npy: element [:,0]) applied to the "
                                          "queries with apply_transform (f64) before quantisation; default: none")
    ap.add_argument("--lib-poses", help=".npy [C,4,4] ground-truth poses of the CADs (`pos_T`); default identity")
    ap.add_argument("--best-match", help=".npy int [Q]: annotated CAD index of every query; default: query i -> i mod C")
    ap.add_argument("--table", help=".npy f64 [C,C] pairwise Chamfer of the catalog (configs/<catid>_scan2cad.npy); "
                                    "default: computed from the first 2000 points of every CAD (utils/pc_dist.py)")
    ap.add_argument("--sym-labels", help="`<path> <label>` per line (configs/<catid>_scan2cad_rot_sym_label.txt); default 1")
    ap.add_argument("--cache-dir", default=None, help="load / save the nine result files (evaluation.py:390-441)")
    ap.add_argument("--register-gt", action="store_false", dest="register_top1", help="register the annotated CAD")
    ap.add_argument("--ignore-cache", action="store_true")
    ap.add_argument("--n-points", type=int, default=10000)
    ap.add_argument("--batch-size", type=int, default=32, help="queries per registration batch")
    ap.add_argument("--embed-batch-size", type=int, default=128,
                    help="clouds per forward of the network (the reference's DataLoader: 32; the rows do not depend on it)")
    ap.add_argument("--ransac-max-iter", type=int, default=100000)
    ap.add_argument("--in-flight", type=int, default=3, help="registration batches in flight (host threads x HIP streams)")
    ap.add_argument("--device", default="cuda", choices=["cuda"], help="there is no CPU path")
    return ap


def main(argv=None):
    """Returns the EvalResult (and prints the log block of evaluation.py:359-383)."""
    from . import synth
    from .utils import ckpts, pc_dist

    a = build_parser().parse_args(argv)
    # (eight hardware queues for the streams of the registration batches in flight: bench.py's note; only effective when
    # nothing has touched the GPU yet -- the command-line entry --, an explicit setting wins)
    import os

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    sd, esd = ckpts.load_state_dicts(a.checkpoint)
    if esd is None:
        raise SystemExit("checkpoint has no embedding_state_dict: retrieval needs the descriptor head (evaluation.py:199)")
    cfg = Config(n_points=a.n_points, batch_size=a.batch_size, embed_batch_size=a.embed_batch_size,
                 ransac_max_iter=a.ransac_max_iter)
    pipe = Pipeline(sd, esd, device=a.device, config=cfg)
    cad_names, catalog = load_cloud_dir(a.catalog_dir, a.n_points, "catalog")
    _, queries = load_cloud_dir(a.query_dir, a.n_points, "queries")
    C, Q = len(catalog), len(queries)
    base_T = np.stack([np.eye(4)] * Q)
    if a.query_poses:
        P = np.load(a.query_poses)
        base_T = np.asarray(P[:Q, 0] if P.ndim == 4 else P[:Q], np.float64)
        if base_T.shape != (Q, 4, 4):
            raise SystemExit(f"--query-poses: need {Q} 4x4 poses, got {P.shape}")
        queries = [synth.apply_pose(q, T, np.float64) for q, T in zip(queries, base_T)]     # apply_transform: f64
    lib_T = np.asarray(np.load(a.lib_poses), np.float64) if a.lib_poses else np.stack([np.eye(4)] * C)
    best_match = np.load(a.best_match).astype(np.int64) if a.best_match else np.arange(Q) % C
    syms = read_sym_labels(a.sym_labels, cad_names) if a.sym_labels else np.ones(C, np.int32)
    if a.table:
        table = np.array(np.load(a.table), np.float64)
    else:
        table = pc_dist.compute_dist([c[:2000] for c in catalog])
    np.fill_diagonal(table, 0.0)                                     # datasets/ScannetDataset.py:65-66
    if table.shape != (C, C) or len(best_match) != Q or len(lib_T) != C:
        raise SystemExit(f"shape mismatch: table {table.shape}, best_match {len(best_match)}, lib poses {len(lib_T)} "
                         f"for C = {C}, Q = {Q}")
    # the clouds and tables loaded above live as long as the process: out of the cycle collector's way (a full collection
    # otherwise walks them between registration batches: 10 - 50 ms pauses)
    import gc
    gc.collect()
    gc.freeze()
    res = run_eval(pipe, catalog, queries, best_match, table, base_T, lib_T, syms, a.category, a.register_top1,
                   a.cache_dir, a.ignore_cache, False, a.batch_size, a.in_flight)
    print(f"category: {a.category}")
    print(f"precision: {res.stat['precision']}\ntop1_error: {res.stat['top1_error']}")
    print(res.report)
    return res


if __name__ == "__main__":
    main()
