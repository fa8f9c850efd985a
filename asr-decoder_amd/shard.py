"""Utterance sharding across the GPUs of one node, and the one collective of the path.

Utterances share nothing but the read-only graph: the graph is replicated on every
GPU, utterance u goes to rank ``u % world`` ... here in contiguous blocks (rank r decodes utterances
[r*B, (r+1)*B)), every rank runs its own frame loop, and the only exchange is ONE gather of the
final results per batch: an all_gather of a fixed-shape float32 tensor [B][3 + Lmax] =
{n_words, tot_score, lm_score, word ids...} (RCCL over xGMI on GPUs -- ``nccl`` backend -- or
``gloo`` on CPU in the tests).  Word ids are < 2^24, so float32 carries them exactly.
"""
from __future__ import annotations

import numpy as np

HEADER = 3


def shard_range(rank, world, per_rank):
    """Global utterance indices decoded by ``rank``."""
    return range(rank * per_rank, (rank + 1) * per_rank)


def pack_results(results, lmax):
    """results: list of dicts with words / tot_score / lm_score (wfstdec.BatchDecoder.best_paths)."""
    out = np.zeros((len(results), HEADER + lmax), np.float32)
    for i, r in enumerate(results):
        w = np.asarray(r["words"])
        if w.size and int(w.max()) >= (1 << 24):
            raise ValueError("word id does not fit the float32 gather payload")
        k = min(int(w.shape[0]), lmax)
        out[i, 0] = int(w.shape[0])
        out[i, 1] = r["tot_score"]
        out[i, 2] = r["lm_score"]
        out[i, HEADER:HEADER + k] = w[:k]
    return out


def unpack_results(packed):
    res = []
    for row in np.asarray(packed):
        n = int(row[0])
        k = min(n, row.shape[0] - HEADER)
        res.append(dict(n_words=n, tot_score=float(row[1]), lm_score=float(row[2]),
                        words=row[HEADER:HEADER + k].astype(np.int32)))
    return res


def gather_results(packed, device=None):
    """all_gather the per-rank [B][3+Lmax] blocks; returns the [world*B][3+Lmax] array in global
    utterance order (every rank gets it; rank 0 is the consumer).  No-op without a process group."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(packed)
    t = torch.from_numpy(np.ascontiguousarray(packed, dtype=np.float32))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out, dim=0).cpu().numpy()
