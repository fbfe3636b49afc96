"""Multi-GPU sharding of the CORSAIR inference path (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

The reference is single-device.  The path shards by independent units:
  * catalog embedding: CAD model c is embedded by rank c % world (interleaved, so the voxel-count
    imbalance between models averages out);
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


def gather_catalog(dist, local_set, n_items, world):
    """Full catalog (item order) from the per-rank shards."""
    if world == 1 or dist is None:
        return local_set
    shards = all_gather_embedded(dist, local_set, world)
    return concat_sets(shards).gather(global_order(n_items, world))
