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


# ---- lattices (lattice mode): length-prefixed byte blobs -----------------------
def lattice_to_bytes(lat):
    """One lattice (wfstdec.BatchDecoder.raw_lattice dict, or None) in the reference's on-disk
    format (Lattice::Write, reference newfst/lattice-fst.cc:38-64): u64 states, i32 start, per state
    {i32 final, u64 arcs, arcs x {i32 ilabel, i32 olabel, f32 graph, f32 acoustic, i32 next}}.
    None (no lattice) is the empty lattice: 0 states, start -1."""
    import struct

    if lat is None:
        return struct.pack("<Qi", 0, -1)
    S = int(lat["n_states"])
    src = np.asarray(lat["a_src"])
    counts = np.bincount(src, minlength=S).astype(np.int64)
    arc_t = np.dtype([("il", "<i4"), ("ol", "<i4"), ("g", "<f4"), ("ac", "<f4"), ("to", "<i4")])
    arcs = np.zeros(len(src), arc_t)
    arcs["il"], arcs["ol"], arcs["g"], arcs["ac"], arcs["to"] = (lat["a_ilabel"], lat["a_olabel"], lat["a_graph"],
                                                                 lat["a_acoustic"], lat["a_dst"])
    if len(src) and np.any(np.diff(src) < 0):
        arcs = arcs[np.argsort(src, kind="stable")]
    raw = arcs.tobytes()
    out = [struct.pack("<Qi", S, 0)]
    off = 0
    fin = np.asarray(lat["st_final"])
    for s in range(S):
        n = int(counts[s])
        out.append(struct.pack("<iQ", int(fin[s]), n))
        out.append(raw[off * 20:(off + n) * 20])
        off += n
    return b"".join(out)


def gather_lattices(blobs, device=None):
    """blobs: this rank's per-utterance byte strings.  Two collectives: an all_gather of the blob
    lengths [B], then an all_gather of the rank's blobs concatenated and padded to the longest
    rank.  Returns the list of all world*B blobs in global utterance order."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(blobs)
    world = dist.get_world_size()
    lens = torch.tensor([len(b) for b in blobs], dtype=torch.int64)
    if device is not None:
        lens = lens.to(device)
    all_lens = [torch.empty_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens)
    all_lens = [t.cpu().numpy() for t in all_lens]
    pad = max(int(l.sum()) for l in all_lens)
    mine = np.zeros(pad, np.uint8)
    cat = b"".join(blobs)
    mine[:len(cat)] = np.frombuffer(cat, np.uint8)
    t = torch.from_numpy(mine)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    res = []
    for r in range(world):
        buf = out[r].cpu().numpy().tobytes()
        o = 0
        for n in all_lens[r]:
            res.append(buf[o:o + int(n)])
            o += int(n)
    return res
