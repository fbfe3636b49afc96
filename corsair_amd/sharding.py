"""Multi-GPU sharding of the CORSAIR inference path (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

The reference is single-device.  The path shards by independent units:
  * catalog embedding: balanced by VOXEL COUNT, not item count (N1 varies 1.6 k - 8 k per
    model): every rank counts the voxels of an interleaved slice (cs_voxelize only), the counts are
    all-gathered (C int64, the only other collective) and `balanced_shards` deals the models out
    largest-first to the lightest rank.  `shard_ids` (plain interleave) is the count-free fallback;
  * queries: embedding, retrieval and registration of a query are independent of every other query.
The only exchange is ONE all-gather of the embedded catalog after the catalog pass: the global
descriptors (f32 [C/world, 256] per rank, needed by every query's top-k) and the per-voxel features
+ origins of the CAD models (needed to register against whichever CAD retrieval selects).  Shard
sizes differ, so payloads are padded to the largest shard and trimmed after the collective.
"""
from __future__ import annotations

import numpy as np
import torch

from .harness import EmbeddedSet, concat_sets


def shard_ids(n, rank, world):
    """Item ids owned by `rank` (interleaved)."""
    return list(range(rank, n, world))


def global_order(n, world):
    """Permutation that maps the concatenation of all shards (rank-major) back to item order."""
    return np.argsort(np.concatenate([np.arange(r, n, world) for r in range(world)]), kind="stable")


def balanced_shards(weights, world):
    """Longest-processing-time-first partition of items with the given weights (voxel counts) over
    `world` ranks: items in order of decreasing weight (ties: smaller id) go to the rank with the
    smallest load so far (ties: lowest rank).  Deterministic, identical on every rank.  Returns a list
    of `world` ascending id lists.  The maximum load is within one item of the mean."""
    w = np.asarray(weights, dtype=np.int64)
    order = np.lexsort((np.arange(len(w)), -w))
    load = np.zeros(world, dtype=np.int64)
    shards = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += int(w[i])
    return [sorted(s) for s in shards]


def shard_order(shards):
    """Permutation that maps the rank-major concatenation of `shards` back to item order."""
    flat = np.concatenate([np.asarray(s, dtype=np.int64) for s in shards]) if shards else np.zeros(0, np.int64)
    return np.argsort(flat, kind="stable")


def imbalance(weights, shards):
    """max / mean load of a partition (1.0 = perfect)."""
    w = np.asarray(weights, dtype=np.float64)
    loads = np.asarray([w[s].sum() if len(s) else 0.0 for s in shards])
    return float(loads.max() / max(loads.mean(), 1e-30))


def all_gather_counts(dist, local_ids, local_counts, n_items, world):
    """Voxel counts of all `n_items` items from the per-rank (ids, counts) slices: one all-gather of an
    int64 [n_items] vector in which a rank fills only its own entries (the rest are zero)."""
    v = torch.zeros(n_items, dtype=torch.int64)
    if len(local_ids):
        v[torch.as_tensor(list(local_ids), dtype=torch.int64)] = torch.as_tensor(list(local_counts), dtype=torch.int64)
    if dist is None:
        return v.numpy()
    if dist.get_backend() == "nccl":
        v = v.cuda()
    out = [torch.empty_like(v) for _ in range(world)]
    dist.all_gather(out, v)
    return torch.stack(out).sum(0).cpu().numpy()


def _stage_device(dist, t):
    """RCCL moves device tensors; gloo (CPU rehearsal / tests) is given host tensors."""
    if t.is_cuda and dist.get_backend() != "nccl":
        return torch.device("cpu")
    return t.device


def _all_gather_padded(dist, t, rows, world):
    """all_gather of a 2-D tensor whose row count differs per rank (padded to `rows`)."""
    stage = _stage_device(dist, t)
    pad = torch.zeros((rows,) + tuple(t.shape[1:]), device=stage, dtype=t.dtype)
    pad[: t.shape[0]] = t.to(stage)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [o.to(t.device) for o in out]


def all_gather_embedded(dist, eset, world):
    """All-gather an EmbeddedSet (descriptors + voxel features + origins + offsets).  Returns the
    list of per-rank sets."""
    dev = eset.F.device
    sizes_local = torch.tensor([eset.F.shape[0], len(eset)], device=_stage_device(dist, eset.F),
                               dtype=torch.int64)
    sizes = [torch.zeros_like(sizes_local) for _ in range(world)]
    dist.all_gather(sizes, sizes_local)
    sizes = torch.stack(sizes).cpu().numpy()
    max_rows, max_n = int(sizes[:, 0].max()), int(sizes[:, 1].max())
    F = _all_gather_padded(dist, eset.F, max_rows, world)
    O = _all_gather_padded(dist, eset.origin, max_rows, world)
    D = _all_gather_padded(dist, eset.desc, max_n, world)
    off = torch.tensor(eset.offsets, device=dev, dtype=torch.int64)[:, None]
    offs = _all_gather_padded(dist, off, max_n + 1, world)
    sets = []
    for r in range(world):
        rows, n = int(sizes[r, 0]), int(sizes[r, 1])
        sets.append(EmbeddedSet(F[r][:rows], O[r][:rows], offs[r][: n + 1, 0].cpu().tolist(), D[r][:n]))
    return sets


def gather_catalog(dist, local_set, n_items, world, shards=None):
    """Full catalog (item order) from the per-rank shards.  `shards` = the id lists the ranks embedded
    (balanced_shards); default: the interleaved assignment of shard_ids."""
    if world == 1 or dist is None:
        return local_set
    sets = all_gather_embedded(dist, local_set, world)
    order = global_order(n_items, world) if shards is None else shard_order(shards)
    return concat_sets(sets).gather(order)


# ---- catalog-sharded descriptor top-k (the 10^6 x 10^6 case, 1 GB of descriptors) ---------------
def merge_topk(d2_lists, gid_lists, k):
    """k best of the concatenated per-shard candidate lists, ranked by (squared distance, global id) -- the
    order cs_l2_topk itself uses (ties -> smaller catalog index).  d2_lists / gid_lists: per shard a [Q, k_s]
    float64 / int64 tensor (absent entries: id -1 with distance inf).  Two stable sorts give the lexicographic
    order: by id first, then by distance."""
    d2 = torch.cat(list(d2_lists), dim=1)
    gid = torch.cat(list(gid_lists), dim=1)
    big = torch.iinfo(torch.int64).max
    key = torch.where(gid < 0, torch.full_like(gid, big), gid)
    o1 = torch.sort(key, dim=1, stable=True).indices
    d2, gid = torch.gather(d2, 1, o1), torch.gather(gid, 1, o1)
    o2 = torch.sort(d2, dim=1, stable=True).indices[:, :k]
    return torch.gather(gid, 1, o2), torch.gather(d2, 1, o2)


def _all_gather_rows(dist, t, world):
    """all_gather of equally shaped tensors (one flat collective where the backend has it)."""
    stage = _stage_device(dist, t)
    src = t.to(stage).contiguous()
    out = torch.empty((world,) + tuple(src.shape), device=stage, dtype=src.dtype)
    if hasattr(dist, "all_gather_into_tensor") and dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, src)
    else:
        parts = [out[r] for r in range(world)]
        dist.all_gather(parts, src)
    return out.to(t.device)


def sharded_topk(dist, q_local, local_topk, shard_first, k, rank, world):
    """Top-k of this rank's queries against a catalog sharded over the ranks (rank r holds the contiguous rows
    [shard_first, shard_first + n_r)).  The QUERIES travel, not the catalog: (1) all-gather the
    queries (ranks may contribute DIFFERENT numbers, also none: the counts are exchanged first and the payload is
    padded to the largest), (2) `local_topk(q_all) -> (local row idx int64 [., k], squared distance f64 [., k])`
    against the own shard (cs_l2_topk_sq) for the real rows of every rank, (3) all-gather the candidate
    (distance, global id) lists, (4) merge the `world` lists of the own queries.  Returns (ids int64 [Q, k],
    d2 f64 [Q, k]): identical to a single-device top-k over the whole catalog, bit for bit.  A rank whose shard is
    empty (more ranks than catalog rows) contributes lists of (-1, inf)."""
    if dist is None or world == 1:
        idx, d2 = local_topk(q_local)
        return idx + (idx >= 0) * shard_first, d2
    Q = int(q_local.shape[0])
    counts = _all_gather_rows(dist, torch.tensor([Q], dtype=torch.int64, device=q_local.device), world)[:, 0].tolist()
    q_max = max(counts)
    if q_max == 0:
        return (torch.zeros((0, k), dtype=torch.int64, device=q_local.device),
                torch.zeros((0, k), dtype=torch.float64, device=q_local.device))
    if min(counts) == q_max:
        q_all = _all_gather_rows(dist, q_local, world).reshape(world * Q, -1)
    else:
        pad = torch.zeros((q_max,) + tuple(q_local.shape[1:]), dtype=q_local.dtype, device=q_local.device)
        pad[:Q] = q_local
        got = _all_gather_rows(dist, pad, world)
        q_all = torch.cat([got[r, : counts[r]] for r in range(world)])
    idx, d2 = local_topk(q_all)
    if idx.shape[1] < k:         # a shard with fewer than k rows (or none): absent entries are (-1, inf)
        fill = k - idx.shape[1]
        idx = torch.cat([idx, torch.full((idx.shape[0], fill), -1, dtype=idx.dtype, device=idx.device)], dim=1)
        d2 = torch.cat([d2, torch.full((d2.shape[0], fill), float("inf"), dtype=d2.dtype, device=d2.device)], dim=1)
    gid = torch.where(idx >= 0, idx + shard_first, idx)
    gids = _all_gather_rows(dist, gid, world)                 # [world (shard), sum of counts, k]
    d2s = _all_gather_rows(dist, d2, world)
    first = sum(counts[:rank])
    mine = slice(first, first + Q)
    return merge_topk([d2s[r, mine] for r in range(world)], [gids[r, mine] for r in range(world)], k)


def catalog_shard(n_items, rank, world):
    """Contiguous row range [first, last) of the descriptor catalog owned by `rank`."""
    per = (n_items + world - 1) // world
    return min(rank * per, n_items), min((rank + 1) * per, n_items)


# ---- the whole evaluation on N ranks, ending in ONE result set (evaluation.py:207-441) ------------------------------
def voxel_balanced_shards(dist, pipe, clouds, rank, world):
    """Shards of a list of clouds balanced by VOXEL count: every rank quantises an interleaved slice
    (`pipe.voxel_counts`: cs_voxelize only), the counts are all-gathered, balanced_shards deals the items out.
    Identical on every rank."""
    n = len(clouds)
    mine = shard_ids(n, rank, world)
    counts = pipe.voxel_counts([clouds[c] for c in mine]) if mine else []
    return balanced_shards(all_gather_counts(dist, mine, counts, n, world), world)


def embed_catalog_sharded(pipe, dist, rank, world, catalog, batch_size=None):
    """The catalog pass of a multi-GPU evaluation: clouds dealt to the ranks by voxel count, embedded, ONE all-gather
    (RCCL over xGMI) of descriptors + voxel features + origins; returns the full catalog in item order on every rank."""
    if dist is None or world == 1:
        return pipe.embed_clouds(list(catalog), batch_size)
    shards = voxel_balanced_shards(dist, pipe, catalog, rank, world)
    local = pipe.embed_clouds([catalog[c] for c in shards[rank]], batch_size) if shards[rank] else _empty_set(pipe.device)
    return gather_catalog(dist, local, len(catalog), world, shards)


def run_eval_sharded(pipe, dist, rank, world, catalog, queries, best_match, table, base_T, lib_T, syms, category="chair",
                     register_top1=True, cache_dir=None, force_gate=False, batch_size=None, in_flight=1):
    """harness.run_eval on `world` ranks (one process per GPU), strong scaling: ONE evaluation of all Q queries.

      1. catalog clouds dealt to the ranks by voxel count, embedded, ONE all-gather of the embedded catalog
         (descriptors + voxel features + origins: any rank can register against whichever CAD retrieval selects);
      2. query clouds dealt to the ranks by voxel count and embedded; their descriptors (Q x 256 f32, 1 MB for the
         chair set) are all-gathered so that every rank holds the retrieval statistics of evaluation.py:272-283;
      3. every rank registers ITS queries against their top-1 (or annotated) CAD -- no collective;
      4. the per-query outputs (the nine arrays of evaluation.py:421-441) are all-gathered and put back into query
         order; rank 0 aggregates, prints and writes the result cache once (evaluation.py:334-383,421-441).
    Anchor draws are seeded by the GLOBAL query number and every kernel on the path computes a sample independently of
    its batch neighbours, so the result equals the single-rank run_eval bit for bit (tests/test_sharding_gloo.py,
    tests/test_gpu_next_rows.py).  Returns the EvalResult on every rank (`report` is printed by the caller)."""
    from . import cache as C_
    from . import harness as H

    if dist is None or world == 1:
        return H.run_eval(pipe, catalog, queries, best_match, table, base_T, lib_T, syms, category, register_top1,
                          cache_dir, True, force_gate, batch_size, in_flight)
    cfg = pipe.cfg
    bs = batch_size or cfg.batch_size
    dev = pipe.device
    Q = len(queries)
    best_match = np.asarray(best_match).astype(np.int64)
    syms = np.asarray(syms)
    # 1. catalog (an EmbeddedSet = already embedded and gathered, e.g. once for several evaluations)
    ebs = cfg.embed_batch_size   # (batch_size is the registration batch)
    cat = catalog if isinstance(catalog, EmbeddedSet) else embed_catalog_sharded(pipe, dist, rank, world, catalog, ebs)
    # 2. queries
    qshards = voxel_balanced_shards(dist, pipe, queries, rank, world)
    mine = np.asarray(qshards[rank], dtype=np.int64)
    qs = pipe.embed_clouds([queries[q] for q in mine], ebs) if len(mine) else _empty_set(dev)
    q_max = max(len(s_) for s_ in qshards)
    order = shard_order(qshards)
    descs = _all_gather_padded(dist, qs.desc, q_max, world)
    desc_all = torch.cat([descs[r][: len(qshards[r])] for r in range(world)])[torch.from_numpy(order).to(qs.desc.device)]
    stat = H.retrieval_stat(pipe, desc_all, cat.desc, best_match, table)
    # 3. registration of the own queries
    pos_idx = np.asarray(stat["top1_predict" if register_top1 else "gt"], dtype=np.int64)
    local = H.register_queries(pipe, qs, mine, cat, pos_idx, syms, base_T, lib_T, force_gate, bs, in_flight)
    # 4. one result set, query order
    per_query = {}
    for name in C_.NAMES:
        a = np.asarray(local[name])
        # explicit width and a wire type fixed by NAME: a rank without queries (Q < world, or an empty balanced shard)
        # holds 0-size arrays -- reshape(0, -1) of those is an error, and a type taken from the data would differ from
        # its peers' (the rank would die before, or mis-pair in, the collective the others wait in)
        wire = np.float32 if C_.DTYPES[name] == np.float32 else np.float64
        a = a.reshape(len(a), 16 if name.startswith("Ts_est") else 1).astype(wire)
        t = torch.from_numpy(np.ascontiguousarray(a))
        parts = _all_gather_padded(dist, t.cuda() if dist.get_backend() == "nccl" else t, q_max, world)
        full = torch.cat([parts[r][: len(qshards[r])] for r in range(world)]).cpu().numpy()[order]
        if name.startswith("Ts_est"):
            per_query[name] = full.reshape(Q, 4, 4).astype(np.float32)
        elif name == "sym_ransac_success":
            per_query[name] = full[:, 0] != 0
        else:
            per_query[name] = full[:, 0].astype(C_.DTYPES[name])
    if rank == 0 and cache_dir is not None:
        C_.save_results(cache_dir, category, register_top1, per_query)
    return H.finish_eval(stat, per_query, False)


def _empty_set(dev):
    return EmbeddedSet(torch.zeros((0, 16), device=dev), torch.zeros((0, 3), device=dev), [0],
                       torch.zeros((0, 256), device=dev))
