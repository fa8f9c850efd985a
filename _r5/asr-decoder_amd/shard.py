"""Utterance sharding across the GPUs of one node, and the one collective step of the path.

Utterances share nothing but the read-only graph: the graph is replicated on every
GPU, rank r decodes the contiguous block of utterances [r*B, (r+1)*B), every rank runs its own frame
loop, and the only exchange is the gather of the final results per batch (RCCL over xGMI on GPUs --
``nccl`` backend -- or ``gloo`` on CPU in the tests): an all_gather of the fixed-shape int32 header
[B][3] = {n_words, tot_score bits, lm_score bits}, then an all_gather of the ranks' word ids,
concatenated and padded to the longest rank.  Nothing is truncated and nothing is converted: word
ids and float bit patterns travel as int32.
"""
from __future__ import annotations

import numpy as np

HEADER = 3


def shard_range(rank, world, per_rank):
    """Global utterance indices decoded by ``rank``."""
    return range(rank * per_rank, (rank + 1) * per_rank)


def pack_results(results):
    """results: list of dicts with words / tot_score / lm_score (wfstdec.BatchDecoder.best_paths).
    Returns (header int32 [B][3], words int32 [sum of lengths])."""
    hdr = np.zeros((len(results), HEADER), np.int32)
    ws = []
    for i, r in enumerate(results):
        w = np.asarray(r["words"], dtype=np.int64)
        if w.size and (int(w.min()) < -(1 << 31) or int(w.max()) >= (1 << 31)):
            raise ValueError("word id does not fit int32")
        hdr[i, 0] = int(w.shape[0])
        hdr[i, 1:3] = np.asarray([r["tot_score"], r["lm_score"]], np.float32).view(np.int32)
        ws.append(w.astype(np.int32))
    words = np.concatenate(ws) if ws else np.zeros(0, np.int32)
    return hdr, words


def unpack_results(hdr, words):
    """Inverse of pack_results for one or several ranks' blocks laid end to end."""
    res, o = [], 0
    hdr = np.asarray(hdr, np.int32)
    sc = hdr[:, 1:3].copy().view(np.float32)
    for i in range(hdr.shape[0]):
        n = int(hdr[i, 0])
        res.append(dict(n_words=n, tot_score=float(sc[i, 0]), lm_score=float(sc[i, 1]),
                        words=np.asarray(words[o:o + n], np.int32).copy()))
        o += n
    return res


def gather_results(packed, device=None):
    """packed = pack_results(...) of this rank.  Two collectives (headers, then padded word ids);
    returns the list of all world*B result dicts in global utterance order (every rank gets it; rank 0
    is the consumer).  Without a process group: this rank's own results."""
    import torch
    import torch.distributed as dist

    hdr, words = packed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return unpack_results(hdr, words)
    world = dist.get_world_size()
    th = torch.from_numpy(np.ascontiguousarray(hdr, dtype=np.int32))
    if device is not None:
        th = th.to(device)
    all_h = [torch.empty_like(th) for _ in range(world)]
    dist.all_gather(all_h, th)
    all_h = [t.cpu().numpy() for t in all_h]
    pad = max(1, max(int(h[:, 0].sum()) for h in all_h))
    mine = np.zeros(pad, np.int32)
    mine[:words.shape[0]] = words
    tw = torch.from_numpy(mine)
    if device is not None:
        tw = tw.to(device)
    all_w = [torch.empty_like(tw) for _ in range(world)]
    dist.all_gather(all_w, tw)
    res = []
    for r in range(world):
        res.extend(unpack_results(all_h[r], all_w[r].cpu().numpy()))
    return res


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
