#!/usr/bin/env python3
"""Headline benchmark: frames/s decoded by the MI355X WFST token-passing decoder.

Workload = BASELINE.json configs[1] ("1xMI355X, batch=128 utterances, ~10M-arc HCLG, beam=13,
precomputed nnet3 log-likelihoods"): synthetic "hclg-like" graph of SURVEY.md section 8(d) (2.85M
states / ~10.1M arcs, seed 7), float32[300][3000] log-likelihoods per utterance (seed = global
utterance index).  HEADLINE = beam-only pruning (beam 13, max_active never binding, min_active 0:
the regime where the result is bit-identical to the reference CPU decoder) on log-likelihoods with
272 live hypotheses per utterance (synth.make_loglikes_multi; why not the single planted path:
DESIGN.md section 5).  The line also carries `service_point`: the section-8(d) single-planted-path
generator at the reference service's max_active 7000 / min_active 200, with the word-level
divergence of the GPU result from the reference decoder's own over the whole batch.

One "step" = one pass of the hot path over one batch: InitDecoding -> AdvanceDecoding(all frames)
-> FinalizeDecoding -> GetBestPath + LatticeToVector for every utterance (reference call sequence
kaldi-nnet3bin/kaldi-hclg-my-decoder.cc:97-129).  Log-likelihoods are resident in HBM before the
timed region starts.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: utterances shard embarrassingly (128 per GPU, graph replicated, weak scaling); the only
collective is the gather of the final word-id results per step (RCCL).
Rank 0 prints ONE JSON line, under 4 KB (summary_line(): metric, value, config, roofline, cpu_baseline and one dict of scalars
per leg); everything else -- curves, counts, notes, every leg's own config and roofline -- goes to bench_detail.json.

roofline.achieved: algorithmic bytes (SURVEY.md 8(d) per-unit figures x the units processed) of the slowest
kernel / its time, from hipEvent pairs around every launch on the stream it is launched on.  The decoder runs
its channels as two groups on two streams (wfst_options.channel_groups), whose launches overlap: the time is
then the kernel's BUSY time (union of its launches' intervals) and the per-launch figures sit in
roofline.per_launch; with --groups 1 both definitions coincide (DESIGN.md section 3 "Roofline accounting").

Other BASELINE configs through the same script (not the headline; `metric` says which):
    --biglm                          configs[3]: on-the-fly LM rescoring
    --lattice-links N [--determinize] [--prune-interval K]   configs[4]: lattice-generating decode
    --beam 15 --lattice-beam 8 --lattice-links 25165824 --arena-per-frame 60000 --max-tokens 262144
    --host-feed                      PCIe-inclusive (host matrices handed over inside the step)
"""
import argparse
import importlib
import json
import os
import sys
import threading
import time

# Eight hardware queues for the process's HIP streams (the runtime's default is four: a decoder with four channel groups has five
# streams, and two streams on one queue take turns) -- read by the HIP runtime when it starts, i.e. before anything below touches the
# GPU.  With them, biglm and lattice decoders of 128 channels are asked for FOUR channel groups below (wfst_options.channel_groups;
# the library's own default is three, it reads no environment).  An exported value wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--states", type=int, default=2850000)
    ap.add_argument("--pdfs", type=int, default=3000)
    ap.add_argument("--beam", type=float, default=13.0)
    ap.add_argument("--max-active", type=int, default=1000000, help="1000000 never binds (beam-only pruning: the bit-exact parity regime); 7000 = the reference service's operating point")
    ap.add_argument("--min-active", type=int, default=0)
    ap.add_argument("--workload", choices=["multi", "single"], default="multi",
                    help="multi: many live hypotheses (synth.make_loglikes_multi); single: one planted path (SURVEY 8(d))")
    ap.add_argument("--paths", type=int, default=272)
    ap.add_argument("--mu", type=float, default=None, help="noise mean (default -4.0 multi, -2.0 single)")
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--groups", type=int, default=0, help="channel groups, each with its own stream and hipGraph (0 = library default 1); "
                    "2 overlaps two half-batches: +7 %% frames/s, but per-launch accounting then covers half a batch")
    ap.add_argument("--host-feed", action="store_true", help="hand the log-likelihoods over as HOST matrices every step "
                    "(wfst_decoder_advance_host): the PCIe-inclusive rate; never the headline")
    ap.add_argument("--host-pageable", action="store_true", help="--host-feed from PAGEABLE host memory (staged copies, the call waits for every slice's "
                                                                 "upload) instead of page-locked matrices")
    ap.add_argument("--max-tokens", type=int, default=65536, help="wfst_limits.max_tokens_per_frame (the headline workload peaks at 40 k tokens "
                    "in one frame; the service-point legs, whose frames reach beyond that before max_active cuts them, run with 131072)")
    ap.add_argument("--default-limits", action="store_true", help="wfst_limits all zero: the LIBRARY's defaults for max_tokens_per_frame (262144 at this "
                    "max_active) and arena_tokens (4 M), instead of the values this script passes")
    ap.add_argument("--arena-per-frame", type=int, default=0, help="token arena per utterance = frames x this (raise it for wider beams); "
                    "300 x 13900 stays below 2^22 tokens, where a token's backpointer has room for its state's degree code "
                    "(wfst_device.h: the expansion then skips the row-header loads); the heaviest utterance of rank 0's workload needs 3.4 M. "
                    "0 = 13900 at every N (the token collection takes care of an utterance that outgrows it)")
    ap.add_argument("--lattice-links", type=int, default=0, help="> 0: lattice mode (BASELINE configs[4]): record forward links "
                    "(capacity per utterance), prune by lattice_beam at finalize; the step then also takes the n-best")
    ap.add_argument("--lattice-beam", type=float, default=7.0)
    ap.add_argument("--determinize", action="store_true", help="lattice mode: the step also builds every utterance's DETERMINIZED lattice "
                    "on the device (GetLattice, base-inl.h:850-866) and fetches it")
    ap.add_argument("--prune-interval", type=int, default=25, help="config prune_interval (reference default 25): lattice mode back-prunes and "
                    "compacts every this many frames; >= --frames: once, at FinalizeDecoding (offline batches that have the memory)")
    ap.add_argument("--nbest", type=int, default=5, help="n of the n-best taken per utterance in lattice mode")
    ap.add_argument("--raw-nbest", action="store_true", help="lattice mode with --determinize: the step's n-best is the short list taken from the RAW "
                                                              "lattice (round 5's step) instead of NShortestPath on the determinized one")
    ap.add_argument("--postprocess", action="store_true", help="lattice mode: after the timed steps, the service's post-processing of the whole "
                    "batch -- GetLattice with the second LM pass and the 10-best of it -- through the batched calls, next to one channel alone")
    ap.add_argument("--debug", type=int, default=0, help="wfst_options.debug (kernel phase timers 32 closure / 64 insert / 128 expand: "
                    "timing experiments only, printed on stderr when the decoder is freed)")
    ap.add_argument("--no-fuse", action="store_true", help="graph without fused epsilon closures (wfst_graph_options.fuse_closures = 0): "
                    "the separate closure pass runs every frame")
    ap.add_argument("--log2-parts", type=int, default=-1, help="wfst_options.log2_partitions (-1 = library default)")
    ap.add_argument("--log2-lds", type=int, default=0, help="wfst_options.log2_lds_slots (0 = library default)")
    ap.add_argument("--joint-max", type=int, default=0, help="wfst_options.joint_max (0 = library default)")
    ap.add_argument("--expand-wgs", type=int, default=0, help="wfst_options.expand_workgroups (0 = library default)")
    ap.add_argument("--insert-wgs", type=int, default=0, help="wfst_options.insert_workgroups (0 = library default)")
    ap.add_argument("--pipeline-determinizer", action="store_true", help="lattice mode with --determinize: utterance k's lattices are "
                    "determinized on a side stream WHILE the channels decode utterance k + 1 (wfst_decoder_prefetch_determinized_detached) and "
                    "fetched one step later; the last step's are waited for inside the timed region")
    ap.add_argument("--no-prefetch", action="store_true", help="lattice mode with --determinize: the determinizer starts when the first "
                    "lattice is asked for (behind the best paths and the n-best lists) instead of right after FinalizeDecoding")
    ap.add_argument("--tile-tokens", type=int, default=0, help="wfst_options.tile_tokens (0 = library default)")
    ap.add_argument("--row-align", type=int, default=0, help="wfst_graph_options.row_align_slots (0 = library default)")
    ap.add_argument("--no-hip-graph", action="store_true", help="enqueue the frame loop kernel by kernel (rocprofv3 --pmc passes)")
    ap.add_argument("--cpu-sample", type=int, default=128, help="utterances checked bit for bit against the CPU decoder -- by default the whole batch of rank 0, decoded by the reference on the host threads in a few seconds -- (0 = skip "
                    "the CPU legs: parity sample, cpu_baseline, service_point divergence)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="wall time of each timed CPU-baseline leg (1 thread, all cores)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores leg (0 = every CPU this process may run on)")
    ap.add_argument("--no-service-point", action="store_true", help="skip the second workload (single planted path, 7000/200)")
    ap.add_argument("--no-legs", action="store_true", help="skip the biglm (BASELINE configs[3]) and beam-15 lattice (configs[4]) legs the "
                    "default run appends to its line (each a child process running this script with --biglm / --lattice-links)")
    ap.add_argument("--only-legs", default="", help="comma-separated leg names: run only these of the headline run's legs (quick checks)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="keep the live parity sample but skip the timed CPU legs (the legs' runs)")
    ap.add_argument("--biglm", action="store_true", help="BASELINE configs[3]: on-the-fly LM rescoring (wfst_decoder_create_biglm) with a "
                    "synthetic bigram (old) / trigram (new) LM pair over the graph's 50k words; NOT the headline")
    ap.add_argument("--lm-old", default="20000,5,0,0", help="old LM: bigram contexts, successors, trigram contexts, successors")
    ap.add_argument("--lm-new", default="40000,6,100000,3", help="new LM: same four numbers")
    ap.add_argument("--lm-pairs", type=int, default=1 << 20, help="LM pair states per utterance (wfst_limits.lm_pairs)")
    ap.add_argument("--graph-cache", default="/tmp/wfst_bench_graph_%d.bin")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child runs of this script, "
                    "FETCH_SIZE and WRITE_SIZE, one step each with the kernels enqueued one by one): quote the stored passes of profiles/traffic_latest.json")
    ap.add_argument("--no-profile-step", action="store_true", help="skip the extra instrumented step behind the timed region (the rocprofv3 --pmc "
                                                                   "child passes: every launch the counters see then belongs to the --steps steps)")
    ap.add_argument("--detail-out", default="", help="where the FULL result (every leg, curve, count and note) is written as JSON; default "
                    "bench_detail.json beside this script (and gpurun_out/bench_detail.json when that directory exists).  The final "
                    "stdout line is the compact summary of it (summary_line(): < 4 KB, scalars only)")
    return ap.parse_args()


def make_utts(synth, g, m, first, count, T, P, a):
    out = np.empty((count, T, P), np.float32)
    for i in range(count):
        if a.workload == "multi":
            out[i] = synth.make_loglikes_multi(g, T, P, m, seed=first + i, n_paths=a.paths, mu=a.mu, sigma=a.sigma,
                                               jitter=0.5, ac_lo=0.5)[0]
        else:
            out[i] = synth.make_loglikes(g, T, P, m, seed=first + i, mu=a.mu, sigma=a.sigma)[0]
    return out


def affinity_cpus():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_decoder():
    """The CPU checker / baseline: the UNMODIFIED reference (oracle/_ref, kind "reference") when its
    prebuilt library is present, else our C restatement (kind "port")."""
    import pyoracle

    if os.path.exists(pyoracle.REF_SO):
        return "reference", pyoracle.RefDecoder()
    pyoracle.build_oracle()
    return "port", pyoracle.OracleDecoder()


class quiet_stderr:
    """silence the reference's LOG_COM chatter on stderr"""

    def __enter__(self):
        self.saved = os.dup(2)
        self.devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.devnull, 2)

    def __exit__(self, *exc):
        os.dup2(self.saved, 2)
        os.close(self.saved)
        os.close(self.devnull)


def cpu_decode_all(dec, graph_path, cd, mats, m, n_threads, big=None):
    """Decode every matrix once on n_threads host threads (results for parity / divergence).
    big = (old LM path, new LM path): the biglm decoder (the restatement runs in fixed mode)."""
    import pyoracle

    h = dec.load_graph(graph_path)
    # (never hand the reference a max_active beyond 10^6: it sizes its hash table by it, base-inl.h:27 -- 68 GB per decoder at INT_MAX)
    cfg = pyoracle.Config(**dict(cd, max_active=min(int(cd.get("max_active", 1000000)), 1000000)))
    lms = [pyoracle.Lm(dec, big[0], -1.0), pyoracle.Lm(dec, big[1], 1.0)] if big else None
    results = [None] * len(mats)
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= len(mats):
                return
            if lms:
                results[i] = pyoracle.biglm_decode(dec, h, cfg, lms[0], lms[1], mats[i], m, fixed=True)
            else:
                results[i] = dec.decode(h, cfg, mats[i], m)

    with quiet_stderr():
        th = [threading.Thread(target=work) for _ in range(max(1, min(n_threads, len(mats))))]
        [t.start() for t in th]
        [t.join() for t in th]
    for L in lms or []:
        L.free()
    dec.free_graph(h)
    return results


def cpu_timed(kind, dec, graph_path, cd, mats, m, n_threads, seconds, big=None, lattice=None):
    """frames/s of the CPU decoder: n_threads host threads, ONE decoder object per thread over one
    shared read-only graph -- the reference service's threading model (v2-asrbin/v2-asr-service.cc:
    95-105) -- each looping over the batch's utterances (thread t takes t, t + n_threads, ...) for
    `seconds` of wall time (ref_timed_loop / oracle_timed_loop).  Returns (frames/s, wall s, frames, extra).
    lattice = n of the n-best: the reference's LATTICE pipeline per utterance (ref_lattice_timed_loop: decode with forward links
    and back-pruning, best path, GetRawLattice, DeterminizeLatticeWrapper, NShortestPath(n)); extra then carries the stage times
    summed over the threads and the determinizer's ms per lattice (reference library only)."""
    import ctypes as C

    import pyoracle

    h = dec.load_graph(graph_path)
    cfg = pyoracle.Config(**dict(cd, max_active=min(int(cd.get("max_active", 1000000)), 1000000)))   # (see cpu_decode_all)
    lms = [pyoracle.Lm(dec, big[0], -1.0), pyoracle.Lm(dec, big[1], 1.0)] if big else None
    if lattice is not None and kind != "reference":
        raise RuntimeError("the lattice pipeline's CPU baseline is the reference's own (oracle/_ref absent)")
    f = getattr(dec.lib, ("ref" if kind == "reference" else "oracle") + ("_biglm" if big else "_lattice" if lattice is not None else "") + "_timed_loop")
    f.restype = C.c_longlong
    keep = [np.ascontiguousarray(x, np.float32) for x in mats]
    ptrs = (C.c_void_p * len(keep))(*[x.ctypes.data for x in keep])
    Ts = np.asarray([x.shape[0] for x in keep], np.int32)
    stride = int(keep[0].shape[1])
    mm = np.ascontiguousarray(m, np.int32)
    frames = [0] * n_threads
    stage = np.zeros((n_threads, 4), np.float64)
    cnt = np.zeros((n_threads, 4), np.int64)

    def work(t):
        el, nw = C.c_double(0), C.c_longlong(0)
        head = [C.c_void_p(h), C.byref(cfg)] + ([C.c_void_p(lms[0].h), C.c_void_p(lms[1].h)] if lms else [])
        if lattice is not None:
            frames[t] = f(*head, ptrs, Ts.ctypes.data_as(C.POINTER(C.c_int)), len(keep), stride,
                          mm.ctypes.data_as(C.POINTER(C.c_int)), int(mm.shape[0] - 1), t, n_threads, C.c_double(seconds), 1, int(lattice),
                          C.byref(el), stage[t].ctypes.data_as(C.POINTER(C.c_double)), cnt[t].ctypes.data_as(C.POINTER(C.c_longlong)))
            return
        frames[t] = f(*head, ptrs, Ts.ctypes.data_as(C.POINTER(C.c_int)), len(keep), stride,
                      mm.ctypes.data_as(C.POINTER(C.c_int)), int(mm.shape[0] - 1), t, n_threads, C.c_double(seconds),
                      C.byref(el), C.byref(nw))

    with quiet_stderr():
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
    for L in lms or []:
        L.free()
    dec.free_graph(h)
    extra = {}
    if lattice is not None:
        st, cn = stage.sum(0), cnt.sum(0)
        extra = {"stage_thread_seconds": {"decode_finalize_best_path": float(st[0]), "get_raw_lattice": float(st[1]),
                                          "determinize": float(st[2]), "nbest": float(st[3])},
                 "lattices": int(cn[0]), "mean_raw_states": float(cn[1]) / max(int(cn[0]), 1), "mean_determinized_states": float(cn[2]) / max(int(cn[0]), 1),
                 "mean_nbest_paths": float(cn[3]) / max(int(cn[0]), 1),
                 "determinizer_ms_per_lattice": 1e3 * float(st[2]) / max(int(cn[0]), 1)}
    return sum(frames) / dt, dt, sum(frames), extra


def edit_distance(a, b):
    """word-level Levenshtein distance (what nbest-compute-wer counts, kaldi-bin/bin)"""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def divergence(gpu_res, cpu_res):
    """GPU results vs the CPU decoder's own, utterance by utterance."""
    n = len(gpu_res)
    ident = errs = nref = same_words = with_path = 0
    gap = 0.0
    signed = []   # (tot_first - tot_second) / |tot_second| of the utterances both sides found a path for: > 0 = the first one's path costs more
    for r, o in zip(gpu_res, cpu_res):
        same = (np.array_equal(o.words, r["words"]) and np.array_equal(o.tids, r["tids"]) and
                np.float32(o.tot_score).tobytes() == np.float32(r["tot_score"]).tobytes())
        ident += int(same)
        same_words += int(np.array_equal(o.words, r["words"]))
        errs += edit_distance(o.words, r["words"])
        nref += len(o.words)
        with_path += int(bool(r["ok"]))
        if o.ok and r["ok"] and o.tot_score != 0:
            gap = max(gap, abs(r["tot_score"] - o.tot_score) / abs(o.tot_score))
            signed.append((float(r["tot_score"]) - float(o.tot_score)) / abs(float(o.tot_score)))
    sg = np.asarray(signed, np.float64) if signed else np.zeros(1)
    return {"utterances": n, "bit_identical": ident, "same_words": same_words, "word_errors": errs, "ref_words": nref,
            "wer": errs / float(max(nref, 1)), "max_rel_cost_gap": gap,
            # path cost is the only quality measure without transcripts: the SIGN of the difference says which side searched better
            "utterances_with_path": with_path,
            "signed_rel_cost_gap": {"mean": float(sg.mean()), "median": float(np.median(sg)),
                                    "first_cheaper": int((sg < 0).sum()), "second_cheaper": int((sg > 0).sum()), "equal": int((sg == 0).sum()),
                                    "note": "(tot_first - tot_second) / |tot_second| over utterances with a path on both sides; first = the "
                                            "decoder under test (GPU, or the reference at hash_ratio 3), second = the reference"}}


def measure_traffic(a, workload_args):
    """HBM bytes per launch of each kernel class, measured NOW: two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE, then
    WRITE_SIZE: separate passes, counters only, no trace domain), one step each, one channel group, kernels enqueued one by one
    (--no-hip-graph) so that every dispatch is attributed.  Corrected as MI355X_MICROARCH.md's HBM section prescribes for gfx950:
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes (FETCH_SIZE counts 128-byte fabric requests at 64 bytes; both are reported in KB).
    Returns ({class: bytes per launch}, {class: launches}, seconds) or None."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    t0 = time.time()
    tmp = tempfile.mkdtemp(prefix="wfst_pmc_", dir="/tmp")
    klass = lambda k: ("expand" if "expand_kernel" in k else "insert" if "insert_kernel" in k else
                       "closure" if ("closure_kernel" in k or "lattice_prune" in k) else None)
    kb = collections.defaultdict(float)
    launches = collections.defaultdict(int)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "1",
                   "--warmup", "0", "--cpu-sample", "0", "--no-service-point", "--no-legs", "--no-traffic", "--no-hip-graph", "--no-profile-step", "--groups", "1",
                   "--detail-out", os.path.join(tmp, ctr + ".json")] + workload_args
            pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            fs = glob.glob(os.path.join(out, "*", "*_counter_collection.csv"))
            if pr.returncode != 0 or not fs:
                log("traffic pass %s failed (rc %d): %s" % (ctr, pr.returncode, pr.stderr.decode()[-300:]))
                return None
            seen = collections.defaultdict(int)
            with open(fs[0]) as f:
                for r in csv.DictReader(f):
                    c = klass(r["Kernel_Name"])
                    if c is None or r["Counter_Name"] != ctr:
                        continue
                    kb[c] += (2.0 if ctr == "FETCH_SIZE" else 1.0) * float(r["Counter_Value"])
                    seen[c] += 1
            for c, n in seen.items():
                launches[c] = max(launches[c], n)
    except Exception as e:  # (a measurement that cannot be made is reported as such, it does not take the bench down)
        log("traffic measurement failed: %r" % (e,))
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return ({c: kb[c] * 1024.0 / max(launches[c], 1) for c in kb}, dict(launches), time.time() - t0)


def traffic_per_launch(child_bytes_per_launch, child_launches_per_step, launches_per_step):
    """HBM bytes per launch of THIS run from a counter pass made with another launch shape (one channel group: whole-batch launches):
    the class's bytes of one step, over this run's launches of the class per step.  (Round 5's line was 2x too large: the child ran
    two steps -- the timed one and the instrumented one -- and its launch count was taken for one step's.)"""
    return float(child_bytes_per_launch) * float(child_launches_per_step) / float(max(launches_per_step, 1))


def stored_launches_per_step(entry, klass, default):
    """profiles/traffic_latest.json: `launches_per_step` since round 6; the rounds before stored `launches` of passes that ran TWO steps"""
    if "launches_per_step" in entry:
        return entry["launches_per_step"].get(klass, default)
    return entry.get("launches", {}).get(klass, 2 * default) / 2.0


def dropin_leg(a, gpath, mats, m, cd, gpu_res, cpu_baseline, threads=64, chunk=25, repeat=17, warm=1):
    """wfst-decode --threads=64 [--pool=64] --chunk=25 --pull over the batch's utterances: frames/s of the C++ DecoderItf mirror in the
    reference service's shape, and the words of every utterance against the batch decoder's (which the run has checked against the
    reference).  The list is decoded `repeat` times over, the first `warm` passes before the clock (the headline's warm-up steps for
    this shape: the first utterances of a process pay its graph captures, code loading and buffer allocations -- 110-150 ms, which four
    passes only diluted: round 6's first figure, kept as pool_cold_4_passes_value)."""
    import re
    import struct
    import subprocess
    import tempfile

    host = os.path.join(ROOT, "asr-decoder_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    cli = os.path.join(host, "wfst-decode")
    # (the utterances' file -- 460 MB -- in memory where the box has a tmpfs: on a disk-backed /tmp its write-back ran beside the
    # leg's first CLI runs, which then came out at half the rate of the later ones)
    tmp = tempfile.mkdtemp(prefix="wfst_dropin_", dir="/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and __import__("shutil").disk_usage("/dev/shm").free > (2 << 30) else "/tmp")
    try:
        B, T, P = mats.shape
        np.asarray(m, "<i4").tofile(os.path.join(tmp, "tid2pdf.bin"))
        with open(os.path.join(tmp, "decoder.conf"), "w") as f:
            f.write("--beam=%g\n--max-active=%d\n--min-active=%d\n--lattice-beam=%g\n--prune-interval=%d\n--beam-delta=%g\n" % (
                cd["beam"], cd["max_active"], cd["min_active"], cd["lattice_beam"], cd["prune_interval"], cd["beam_delta"]))
        with open(os.path.join(tmp, "ll.bin"), "wb") as f:
            for i in range(B):
                key = ("utt%04d" % i).encode()
                f.write(struct.pack("<i", len(key)) + key + struct.pack("<ii", T, P))
                f.write(np.ascontiguousarray(mats[i], "<f4").tobytes())
        os.sync()
        common = [cli, "--tid2pdf=" + os.path.join(tmp, "tid2pdf.bin"), "--chunk=%d" % chunk,
                  "--max-frames=%d" % (T + 2), "--max-tokens=%d" % a.max_tokens, "--arena-tokens=%d" % int(T * a.arena_per_frame)]
        tail = [os.path.join(tmp, "decoder.conf"), gpath, os.path.join(tmp, "ll.bin")]
        want = {"utt%04d" % i: [int(w) for w in gpu_res[i]["words"]] for i in range(B) if gpu_res[i]["ok"]}
        o = {"unit": "frames/s", "threads": threads, "chunk_frames": chunk, "utterances": int(B) * (repeat - warm), "passes_over_the_batch": repeat - warm,
             "warm_up_passes": warm,
             "what": "wfst-decode --threads=%d --chunk=%d: %d host threads, one DecoderItf object each; pool / private: every score pulled through "
                     "LogLikelihood(frame, transition-id) of a DecodableMatrixScaledMapped-shaped decodable (6000 indices a frame, the graph reads "
                     "column ilabel); pool_matrix: MatrixDecodable rows of 3000 pdf columns taken in one piece (the graph reads tid2pdf[ilabel]); "
                     "host -> device inside the timed region" % (threads, chunk, threads)}
        # (the headline shape three times, spread over the leg -- first, in the middle, last --, the best of them: fresh processes on a
        # box whose other legs have just ended vary by more than the shapes differ; all three are kept in the detail)
        pool_tag = ("pool", ["--pool=%d" % threads, "--pull"])
        for tag, extra in (pool_tag, ("pool_matrix", ["--pool=%d" % threads]), ("private", ["--pull"]), pool_tag,
                           # the one-line drop-in: GpuLatticeDecoder::ShareDevice(64) once, the threads construct (graph, config) decoders as ever
                           ("shared", ["--share=%d" % threads, "--pull"]),
                           # twice the threads over one shared decoder of 128 channels: the device's frames are latency bound -- a call
                           # of 128 channels takes it 58 us a frame, one of 64 channels 45
                           ("shared_threads128", ["--threads=128", "--share=128", "--pull"]),
                           # (the process's first utterances inside the clock, four passes: round 6's first way of counting)
                           ("pool_cold_4_passes", ["--pool=%d" % threads, "--pull", "--repeat=4", "--warm=0"]),
                           # utterances cut to 50-100 % of their frames: boundaries that do not fall together (frames/s only: the words
                           # of cut utterances are not the batch decoder's; ragged utterances against the oracle: tests/test_gpu_host_cli.py)
                           ("pool_ragged", ["--pool=%d" % threads, "--pull", "--ragged=50"]), pool_tag):
            best = None
            if not any(x.startswith("--threads=") for x in extra):
                extra = ["--threads=%d" % threads] + extra
            if tag == "private":   # (33 k frames/s: two timed passes are 2.3 s)
                extra = extra + ["--repeat=%d" % (warm + 2), "--warm=%d" % warm]
            elif "--repeat=4" not in extra:
                extra = extra + ["--repeat=%d" % repeat, "--warm=%d" % warm]
            for rep in range(1):
                p = subprocess.run(common + extra + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
                err = p.stderr.decode(errors="replace")
                if p.returncode != 0:
                    raise RuntimeError("wfst-decode (%s) failed: %s" % (tag, err[-400:]))
                mw = re.search(r"LOG Timed passes: (\d+) frames in (\S+) s", err)
                if mw:
                    fps = float(mw.group(1)) / float(mw.group(2))
                else:
                    mt = re.search(r"LOG Time taken (\S+)s", err)
                    mf = re.search(r"Frames decoded in all passes: (\d+)", err) or re.search(r"per frame is \S+ over (\d+) frames", err)
                    fps = float(mf.group(1)) / float(mt.group(1))
                if best is None or fps > best[0]:
                    best = (fps, p.stdout.decode(), err)
            fps, stdout, err = best
            o.setdefault(tag + "_values_all_runs", []).append(fps)
            if fps < o.get(tag + "_value", 0.0):
                continue   # (an earlier run of the shape was faster: its record stays)
            got = {l.split()[0]: [int(w) for w in l.split()[1:]] for l in stdout.strip().splitlines()}
            same = sum(1 for k, w in want.items() if got.get(k) == w)
            o[tag + "_value"] = fps
            if tag != "pool_ragged":
                o[tag + "_same_words_as_batch_decoder"] = "%d/%d" % (same, len(want))
            mp = re.search(r"mean batch ([\d.]+)", err)
            if mp and tag == "pool":
                o["pool_mean_advance_batch"] = float(mp.group(1))
            o[tag + "_log"] = [l for l in err.splitlines() if l.startswith("LOG pool") or l.startswith("LOG Time")]
            o[tag + "_loadavg_before"] = os.getloadavg()[0]
        o["value"] = o["pool_value"]
        o["ms_per_step"] = 1e3 * B * T / o["pool_value"]
        if cpu_baseline and "threads_to_value" in cpu_baseline:
            tv = cpu_baseline["threads_to_value"]
            k = "64" if "64" in tv else max(tv, key=lambda q: int(q))
            o["reference_threads"] = int(k)
            o["reference_value"] = tv[k]
        return o
    finally:
        import shutil

        shutil.rmtree(tmp, ignore_errors=True)


LINE_LIMIT = 4000   # bytes of the final stdout line (the driver parses it; round 4's 29 KB line was not parsed)


def _short(x):
    """numbers as the line carries them: 6 significant digits, numpy scalars as Python numbers"""
    if isinstance(x, np.generic):
        x = x.item()
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None   # strict JSON: no NaN / Infinity
        return float("%.6g" % x)
    raise TypeError("summary_line carries scalars only, not %r" % type(x))


def _ratio(text):
    """'16/16 sampled utterances bit-exact ...' -> '16/16'"""
    import re

    m = re.match(r"\s*(\d+/\d+)", text or "")
    return m.group(1) if m else None


def leg_scalars(o):
    """One leg of the line: scalars only (its full dict stays in bench_detail.json)."""
    if "error" in o:
        return {"error": str(o["error"])[:120]}
    r, c, b = o.get("roofline", {}), o.get("config", {}), o.get("cpu_baseline", {})
    k = {"value": o.get("value"), "ms_per_step": o.get("ms_per_step"), "steps": o.get("steps"),
         "kernel": r.get("kernel"), "frac": r.get("frac"), "whole_path_frac": r.get("whole_path", {}).get("frac_over_step_time"),
         "whole_path_frac_8d": r.get("whole_path", {}).get("frac_8d_formula_over_step_time"),
         "parity": _ratio(c.get("parity")), "utterances_with_path": c.get("utterances_with_path"),
         "cpu_baseline_value": b.get("value"), "cpu_baseline_cores": b.get("cores"), "cpu_baseline_kind": b.get("kind")}
    if "lattice_parity" in c:
        k["lattice_parity"] = _ratio(c["lattice_parity"])
    if "determinizer_ms_per_lattice" in b:
        k["cpu_determinizer_ms_per_lattice"] = b["determinizer_ms_per_lattice"]
    dl = c.get("determinized_lattices", {})
    if "gpu_ms_per_lattice_mean" in dl:
        k["gpu_determinizer_ms_per_lattice_mean"] = dl["gpu_ms_per_lattice_mean"]
        k["gpu_determinizer_ms_per_lattice_max"] = dl["gpu_ms_per_lattice_max"]
    for dk in ("divergence_vs_reference", "divergence_vs_port"):   # service-point legs: word-level divergence over the whole batch
        if dk in o:
            k["bit_identical"] = "%d/%d" % (o[dk]["bit_identical"], o[dk]["utterances"])
            k["wer_vs_cpu"] = o[dk]["wer"]
    for dk in ("reference_self_divergence_hash_ratio_3_vs_2", "port_self_divergence_hash_ratio_3_vs_2"):
        if dk in o:
            k["cpu_self_bit_identical"] = "%d/%d" % (o[dk]["bit_identical"], o[dk]["utterances"])
            k["cpu_self_wer"] = o[dk]["wer"]
    if "spread" in o:
        k["cpu_self_wer_max"] = o["spread"]["reference_vs_reference_wer_range"][1]
        k["wer_vs_cpu_max"] = o["spread"]["gpu_vs_reference_wer_range"][1]
    if "degraded_frames" in o:
        k["degraded_frames"] = o["degraded_frames"]
    for dk in ("pool_value", "pool_matrix_value", "shared_value", "shared_threads128_value", "private_value", "reference_value", "reference_threads", "pool_mean_advance_batch",
               "threads", "chunk_frames", "warm_up_passes", "pool_cold_4_passes_value"):
        if dk in o:
            k[dk] = o[dk]
    if "pool_same_words_as_batch_decoder" in o:
        k["parity"] = o["pool_same_words_as_batch_decoder"]
    return {a: _short(v) for a, v in k.items() if v is not None}


def summary_line(out, detail_path=None):
    """The ONE stdout line: strict JSON under LINE_LIMIT bytes.  Top level = the bench contract's keys; config, roofline and
    cpu_baseline hold scalars; `legs` holds one dict of scalars per further configuration (BASELINE configs[3], configs[4], the
    service-point workloads).  Shape of reference: the one RTF line of kaldi-nnet3bin/kaldi-hclg-my-decoder.cc:189-192."""
    top = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data")}
    top["metric"] = str(top["metric"])[:100]
    top["data"] = str(top.get("data", ""))[:64]
    c = out.get("config", {})
    cfg = {k: c[k] for k in ("workload", "global_batch", "frames_per_utt", "parallelism", "rtfx", "channel_groups",
                             "mean_active_tokens_per_frame", "peak_tokens_in_a_frame", "max_tokens_per_frame_limit", "degraded_frames",
                             "utterances_with_path", "fused_epsilon_closures", "decoder_paths_same_on_every_rank", "regime") if k in c}
    if "regime" in cfg:
        cfg["regime"] = str(cfg["regime"])[:60]
    if "workload" in cfg:
        cfg["workload"] = str(cfg["workload"])[:150]
    for k in ("parity", "lattice_parity", "parity_per_rank_sample"):
        if k in c:
            cfg[k] = _ratio(c[k])
    if "parity" in c:
        cfg["parity_vs"] = "oracle (biglm, fixed mode)" if "fixed mode" in c["parity"] else "reference" if "reference" in c["parity"] else "oracle"
    if "gather_check" in c:
        cfg["gather_check"] = "%d/%d" % (c["gather_check"]["bit_exact_vs_oracle"], c["gather_check"]["utterances"])
    if "lattice_gather_check" in c:
        cfg["lattice_gather_check"] = c["lattice_gather_check"]
    top["config"] = {k: w for k, w in ((k, _short(v)) for k, v in cfg.items()) if w is not None}
    r = out.get("roofline")
    if r:
        rr = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                    "avg_launch_ms", "launches", "channel_groups", "profiled_step_ms", "traffic_over_algorithmic")}
        rr["traffic_measured_in_run"] = bool(r.get("traffic_measured_in_run", False))
        rr["whole_path_frac"] = r.get("whole_path", {}).get("frac_over_step_time")
        for k in ("expand", "insert"):
            rr[k + "_ms_per_step"] = r.get("kernel_ms_per_step", {}).get(k)
            if "kernel_busy_ms_per_step" in r:
                rr[k + "_busy_ms_per_step"] = r["kernel_busy_ms_per_step"].get(k)
        if "per_launch" in r:
            rr["per_launch_frac"] = r["per_launch"]["frac"]
        top["roofline"] = {k: _short(v) for k, v in rr.items() if v is not None or k == "traffic"}
    b = out.get("cpu_baseline")
    if b:
        bb = {k: b.get(k) for k in ("value", "unit", "cores", "kind", "single_thread_value", "all_cpus_value", "cpu_model", "affinity_cpus")}
        bb["sample"] = str(b.get("sample", ""))[:48]
        top["cpu_baseline"] = {k: _short(v) for k, v in bb.items() if v is not None}
    legs = {}
    for name, o in out.get("legs", {}).items():
        legs[name] = leg_scalars(o)
    if legs:
        top["legs"] = legs
    if detail_path:
        top["detail"] = detail_path

    def plain(o):
        if isinstance(o, np.generic):
            return o.item()
        raise TypeError("not JSON serialisable: %r" % type(o))

    def dump():
        return json.dumps(top, default=plain, allow_nan=False, separators=(",", ":"))

    # (a line over the limit loses its optional parts -- strings first, then the legs' scalars from the least telling one up, then
    # whole legs from the last one -- rather than its contract keys, and is printed in any case)
    # (what a leg keeps when the line has to shrink, most telling first)
    order = ("value", "ms_per_step", "parity", "error", "pool_value", "shared_value", "shared_threads128_value", "private_value", "reference_value", "pool_matrix_value", "bit_identical",
             "wer_vs_cpu", "cpu_self_wer", "cpu_baseline_value", "frac", "whole_path_frac", "lattice_parity", "gpu_determinizer_ms_per_lattice_mean",
             "gpu_determinizer_ms_per_lattice_max", "cpu_determinizer_ms_per_lattice", "whole_path_frac_8d", "utterances_with_path", "steps",
             "wer_vs_cpu_max", "cpu_self_wer_max", "pool_mean_advance_batch", "degraded_frames", "cpu_self_bit_identical", "kernel",
             "cpu_baseline_cores", "cpu_baseline_kind", "reference_threads", "threads", "chunk_frames", "warm_up_passes", "pool_cold_4_passes_value")
    line = dump()
    if len(line) > LINE_LIMIT:
        for part, key, n in (("cpu_baseline", "sample", 60), ("config", "workload", 120), ("config", "parallelism", 40), ("config", "regime", 50)):
            if part in top and key in top[part]:
                top[part][key] = str(top[part][key])[:n]
        top["metric"] = top["metric"][:100]
        line = dump()
    for keep in range(len(order) - 1, 5, -1):
        if len(line) <= LINE_LIMIT or "legs" not in top:
            break
        top["legs"] = {name: {k: v for k, v in leg.items() if k in order[:keep]} for name, leg in top["legs"].items()}
        line = dump()
    while len(line) > LINE_LIMIT and top.get("legs"):
        top["legs"].pop(next(reversed(top["legs"])))   # (last added first; bench_detail.json has them all)
        top["legs_dropped_from_line"] = top.get("legs_dropped_from_line", 0) + 1
        line = dump()
    return line


def write_detail(out, path):
    """The full result -- what the line summarises -- as indented JSON (NaN / Infinity as null)."""
    def clean(o):
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, np.generic):
            o = o.item()
        if isinstance(o, np.ndarray):
            return clean(o.tolist())
        if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
            return None
        return o

    txt = json.dumps(clean(out), indent=1, allow_nan=False)
    paths = [path] if path else [os.path.join(ROOT, "bench_detail.json")]
    if not path and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for q in paths:
        try:
            with open(q + ".tmp", "w") as f:
                f.write(txt)
            os.replace(q + ".tmp", q)
            written = written or q
        except OSError as e:   # (a read-only tree: the line still goes out)
            log("bench detail not written to %s: %r" % (q, e))
    return written


def oracle_counts(graph_path, cd, mats, m, order_free=False, want_paths=None):
    """N/E/Z work counts of the CPU restatement (algorithmic bytes come from the
    CPU restatement's counts, not from GPU-side expansion).  order_free: the oracle's order-free
    mode; want_paths: list that receives (transition-ids, tot_score) per utterance."""
    import ctypes as C

    import pyoracle

    pyoracle.build_oracle()
    orc = pyoracle.OracleDecoder()
    orc.set_order_free(order_free)
    if want_paths is not None:
        want_paths[:] = [None] * len(mats)
    f = orc.lib.oracle_decode_ex
    f.restype = C.c_int
    h = orc.load_graph(graph_path)
    tot = np.zeros(8, np.int64)
    cfg = pyoracle.Config(**dict(cd, max_active=min(int(cd.get("max_active", 1000000)), 1000000)))   # (see cpu_decode_all)
    lock = threading.Lock()
    idx = [0]

    def work():
        while True:
            with lock:
                i = idx[0]
                idx[0] += 1
            if i >= len(mats):
                return
            ll = np.ascontiguousarray(mats[i], np.float32)
            T, stride = ll.shape
            mp = 4 * T + 64
            ib = [np.zeros(mp, np.int32) for _ in range(4)]
            fb = [np.zeros(mp, np.float32) for _ in range(2)]
            n = [C.c_int(0) for _ in range(6)]
            sc = [C.c_float(0), C.c_float(0)]
            ex = np.zeros(8, np.int64)
            ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
            fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
            f(C.c_void_p(h), C.byref(cfg), fp(ll), T, stride, ip(m), int(m.shape[0] - 1), 0, 1, 1,
              ip(ib[0]), ip(ib[1]), fp(fb[0]), fp(fb[1]), mp, C.byref(n[0]), C.byref(sc[0]), C.byref(sc[1]),
              ip(ib[2]), mp, C.byref(n[1]), ip(ib[3]), mp, C.byref(n[2]), None, None, -1, None, None, 0, None,
              C.byref(n[3]), C.byref(n[4]), ex.ctypes.data_as(C.POINTER(C.c_int64)))
            with lock:
                tot[:] += ex
                if want_paths is not None:
                    want_paths[i] = (ib[3][: n[2].value].copy(), np.float32(sc[0].value))

    th = [threading.Thread(target=work) for _ in range(min(len(mats), os.cpu_count() or 1))]
    [t.start() for t in th]
    [t.join() for t in th]
    orc.set_order_free(False)
    orc.free_graph(h)
    return dict(N=int(tot[0]), E=int(tot[1]), Z=int(tot[2]), ties_on_best_path=int(tot[5]))


def launch_ranks(a):
    """`python bench.py --gpus N` with no rank environment: start the N ranks ourselves -- `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py <same arguments>` as a CHILD process (no exec, and nothing in this parent has touched torch or the
    GPU), relay its output (rank 0's JSON line is the last stdout line) and exit with its code.  The shape it stands for: N workers
    over one shared graph, v2-asrbin/v2-asr-service.cc:95-105 -- here one process and one graph replica per GPU."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:   # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    log("[launcher] %s" % " ".join(cmd))
    pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    last = None
    for raw in pr.stdout:   # (stderr goes straight through; stdout is relayed line by line so that the JSON line stays the last one)
        line = raw.decode(errors="replace").rstrip("\n")
        if line.startswith("{") and line.endswith("}"):
            last = line
        else:
            log(line)
    rc = pr.wait()
    if last is not None:
        print(last, flush=True)
    raise SystemExit(rc)


def main():
    a = parse()
    if a.mu is None:
        a.mu = -4.0 if a.workload == "multi" else -2.0
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        launch_ranks(a)   # (before torch is imported: the parent never initialises HIP)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (a.gpus, a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible; there is no CPU fallback)")
    # WFST_BENCH_SHARE_GPU=1 (testing only): all ranks use GPU 0 and gather over gloo, so the N>1
    # code path can be exercised on a 1-GPU box; the driver's runs use one GPU per rank and RCCL.
    share = os.environ.get("WFST_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    pkg = importlib.import_module("asr-decoder_amd")
    synth, wfstdec, shard = pkg.synth, pkg.wfstdec, pkg.shard
    # one build per node: rank 0 compiles if the library is stale, the others wait (N ranks writing one .so at once
    # would race); every rank then loads the same file
    if world == 1:
        pkg.build.build()
    else:
        if rank == 0:
            pkg.build.build()
        dist.barrier()

    B, T, P = a.batch, a.frames, a.pdfs
    if a.arena_per_frame <= 0:
        # the SAME decoder configuration at every N (an arena above 2^22 tokens would drop the degree codes: every rank of an N > 1
        # run would then read the row headers the N = 1 headline skips); an utterance that outgrows it collects its tokens
        a.arena_per_frame = 13900
    n_tid = 2 * P
    m = synth.default_tid2pdf(n_tid)
    cd = dict(beam=a.beam, max_active=a.max_active, min_active=a.min_active, lattice_beam=a.lattice_beam,
              prune_interval=a.prune_interval, beam_delta=0.5)

    # ---- inputs (untimed) -------------------------------------------------------------------
    t0 = time.time()
    gpath = (a.graph_cache % a.states) + (".r%d" % rank if world > 1 else "")
    g = None
    if os.path.exists(gpath):
        try:
            g = synth.Graph.read(gpath)
        except (IOError, ValueError):
            g = None  # a partial file from an interrupted run: rebuild
    if g is None:
        g = synth.make_hclg_like(a.states, seed=7, n_tid=n_tid)
        g.write(gpath + ".tmp%d" % os.getpid())
        os.replace(gpath + ".tmp%d" % os.getpid(), gpath)
    log("[rank %d] graph: %d states, %d arcs (%.1fs)" % (rank, g.n_states, g.n_arcs, time.time() - t0))
    t0 = time.time()
    mats = make_utts(synth, g, m, rank * B, B, T, P, a)
    if os.environ.get("WFST_BENCH_SAME_UTT"):  # experiment: every channel decodes the same utterance (no load imbalance)
        mats[:] = mats[int(os.environ["WFST_BENCH_SAME_UTT"])]
    ll_dev = torch.from_numpy(mats).to(dev)  # [B][T][P] resident in HBM
    log("[rank %d] log-likelihoods: %d x [%d x %d] (%.1fs)" % (rank, B, T, P, time.time() - t0))

    graph = wfstdec.Graph.from_arrays(g.start, g.final_state, g.state_info, g.arcs, device=local_rank,
                                      options=wfstdec.GraphOptions(fuse_closures=0 if a.no_fuse else 1, **({"row_align_slots": a.row_align} if a.row_align else {})))
    graph.set_tid2pdf(m)
    big, lm_dev, lm_info = None, [None, None], None
    post_lms = None
    if a.biglm or a.postprocess:
        lmsynth = importlib.import_module("asr-decoder_amd.lmsynth")
        V = int(g.arcs["olabel"].max())
        t0 = time.time()
        big = []
        for tag, spec, order, seed in (("old", a.lm_old, 2, 41), ("new", a.lm_new, 3, 42)):
            nb, s2, nt, s3 = (int(x) for x in spec.split(","))
            lp = "/tmp/wfst_bench_lm_%s_%d_%s.bin%s" % (tag, V, spec.replace(",", "_"), ".r%d" % rank if world > 1 else "")
            if not os.path.exists(lp):
                lm = lmsynth.make_lm(V, 3 if nt > 0 else 2, nb, s2, nt, s3, seed=seed)
                lm.to_fsa().write(lp + ".tmp%d" % os.getpid())
                os.replace(lp + ".tmp%d" % os.getpid(), lp)
            big.append(lp)
        lm_dev = [wfstdec.Lm.load(big[0], -1.0, device=local_rank), wfstdec.Lm.load(big[1], 1.0, device=local_rank)]
        lm_info = [x.info() for x in lm_dev]
        if not a.biglm:   # (--postprocess: the LMs of the service's second pass, not of the search)
            post_lms, lm_dev, big = lm_dev, [None, None], None
        log("[rank %d] LMs: old %d states / %d arcs, new %d states / %d arcs (%.1fs)" % (
            rank, lm_info[0]["n_states"], lm_info[0]["n_arcs"], lm_info[1]["n_states"], lm_info[1]["n_arcs"], time.time() - t0))
    stream = torch.cuda.current_stream(dev).cuda_stream
    if a.groups <= 0 and (a.biglm or a.lattice_links > 0) and B >= 128 and int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) >= 8:
        a.groups = 4   # (chains of small launches: a fourth group finds room -- with eight hardware queues only, see the top of the file)
    opt = wfstdec.Options(use_hip_graph=0 if a.no_hip_graph else 1, debug=a.debug, **({"channel_groups": a.groups} if a.groups > 0 else {}),
                          **({"expand_workgroups": a.expand_wgs} if a.expand_wgs > 0 else {}),
                          **({"log2_partitions": a.log2_parts} if a.log2_parts >= 0 else {}),
                          **({"log2_lds_slots": a.log2_lds} if a.log2_lds > 0 else {}),
                          **({"joint_max": a.joint_max} if a.joint_max > 0 else {}),
                          **({"insert_workgroups": a.insert_wgs} if a.insert_wgs > 0 else {}),
                          **({"tile_tokens": a.tile_tokens} if a.tile_tokens > 0 else {}))

    def new_decoder(cfg_dict, max_tokens=None):
        return wfstdec.BatchDecoder(graph, wfstdec.Config(**cfg_dict), B, max_frames=0 if a.default_limits else T + 2,
                                    max_tokens_per_frame=0 if a.default_limits else (max_tokens or a.max_tokens),
                                    arena_tokens=0 if a.default_limits else int(T * a.arena_per_frame), stream=stream, lattice_links=a.lattice_links,
                                    options=opt, old_lm=lm_dev[0], new_lm=lm_dev[1], lm_pairs=a.lm_pairs if a.biglm else 0)

    dec = new_decoder(cd)
    ready = [T] * B
    tb = {"init": 0.0, "advance_enqueue": 0.0, "finalize": 0.0, "sync": 0.0, "best_paths": 0.0, "n": 0}

    gathered = []  # world > 1: every utterance's result in global order, from the last step's gather
    gathered_lat = []  # world > 1, lattice mode with --determinize: every utterance's determinized lattice (on-disk format), likewise

    def make_step(dec, ll_dev, host_rows):
        ptrs = [ll_dev[i].data_ptr() for i in range(B)]
        pipe = {"primed": False}

        def step():
            t0 = time.perf_counter()
            dec.init()
            t1 = time.perf_counter()
            chunk = int(os.environ.get("WFST_BENCH_CHUNK", "0") or 0)   # (experiment: the utterances handed over in chunks of this many frames, call after call)
            for upto in (range(chunk, T + chunk, chunk) if chunk > 0 else (T,)):
                part = [min(r, upto) for r in ready]
                if a.host_feed:
                    dec.advance_host(host_rows, part)
                else:
                    dec.advance(ptrs, part, P)
            t2 = time.perf_counter()
            dec.finalize()
            # --determinize: the service's order -- GetLattice (GetRawLattice + DeterminizeLatticeWrapper), then GetNbest = NShortestPath
            # on THAT lattice (kaldi-online-nnet3-my-decoder.cc:97-105) -- both started here, on the determinizer's stream
            # (wfst_decoder_prefetch_nbest); the short list from the RAW lattice (wfst_decoder_get_nbest) is the step's n-best only
            # where no determinized lattice is asked for
            det_nbest = a.lattice_links > 0 and a.determinize and not a.no_prefetch and not a.raw_nbest
            if a.lattice_links > 0 and a.determinize and a.pipeline_determinizer:
                # (harvests the utterances of the step before, starts these: the subset construction runs on its own stream while
                # the channels decode the NEXT step's utterances -- wfst_decoder_prefetch_determinized_detached)
                dec.prefetch_determinized(detached=True, nbest=a.nbest if det_nbest else 0)
            elif a.lattice_links > 0 and a.determinize and not a.no_prefetch:
                dec.prefetch_determinized(nbest=a.nbest if det_nbest else 0)   # GetLattice's determinizer starts now, beside the best paths
            t3 = time.perf_counter()
            if os.environ.get("WFST_BENCH_BREAKDOWN"):
                dec.sync()
            t4 = time.perf_counter()
            res = dec.best_paths(cap=2 * T + 64)
            t4a = time.perf_counter()
            if a.lattice_links > 0:
                if not det_nbest:
                    nb = dec.nbest(a.nbest)
                    for r, paths in zip(res, nb):
                        r["nbest"] = paths
                tb["nbest"] = tb.get("nbest", 0.0) + time.perf_counter() - t4a
                tb["best_paths_only"] = tb.get("best_paths_only", 0.0) + t4a - t4
                if a.determinize and a.pipeline_determinizer:
                    if pipe["primed"]:   # the lattices (and n-best lists) of the step before (every step decodes the same utterances)
                        for r, L in zip(res, dec.prefetched_lattices()):
                            r["det"] = L
                        if det_nbest:
                            res.nbest_lazy = dec.prefetched_nbest(a.nbest)   # (fetched now; the per-path views are put together when looked at)
                    pipe["primed"] = True
                elif a.determinize:
                    for r, L in zip(res, dec.determinized_lattices()):
                        r["det"] = L
                    if det_nbest:
                        res.nbest_lazy = dec.nbest_paths_all(a.nbest)
            t5 = time.perf_counter()
            for k, v in zip(("init", "advance_enqueue", "finalize", "sync", "best_paths"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                tb[k] += v
            tb["n"] += 1
            if world > 1:  # the path's only collective: gather the final results (RCCL)
                gathered[:] = shard.gather_results(shard.pack_results(res), device=None if share else dev)
                if a.lattice_links > 0 and a.determinize and all("det" in r for r in res):
                    # ... and the final lattices (north_star: "RCCL over xGMI only to gather final lattices"): every utterance's
                    # determinized lattice as a length-prefixed blob in the reference's on-disk format (Lattice::Write)
                    gathered_lat[:] = shard.gather_lattices([shard.lattice_to_bytes(r["det"]) for r in res], device=None if share else dev)
            return res

        def drain(res):
            """pipelined determinizer: the lattices of the LAST step, inside the timed region (K steps = K determinizations)"""
            if a.lattice_links > 0 and a.determinize and a.pipeline_determinizer and res is not None:
                dec.harvest_prefetched()
                for r, L in zip(res, dec.prefetched_lattices()):
                    r["det"] = L
                if not a.no_prefetch and not a.raw_nbest:
                    res.nbest_lazy = dec.prefetched_nbest(a.nbest)

        step.drain = drain
        return step

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(step, warmup, steps):
        import gc

        res = None
        for _ in range(warmup):
            res = step()
        fence()
        # (the interpreter's cyclic collector is kept out of the timed region: with torch's millions of objects alive, the one full
        # collection an early step triggers is a ~35 ms pause of the HOST -- a lattice step builds 128 x 5 result dicts -- that has
        # nothing to do with the decoder; collected before, re-enabled after)
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        per_step = [] if os.environ.get("WFST_BENCH_PER_STEP") else None   # (experiment: every step's own wall time, to stderr)
        for _ in range(steps):
            ts = time.perf_counter()
            res = step()
            if per_step is not None:
                per_step.append(round(1e3 * (time.perf_counter() - ts), 2))
        if per_step:
            log("[rank %d] ms of each timed step: %s" % (rank, per_step))
        if hasattr(step, "drain"):
            step.drain(res)
        fence()
        dt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, res

    host_rows = [mats[i] for i in range(B)]
    if a.host_feed and not a.host_pageable:
        # page-locked host matrices, as a caller that feeds the device keeps them (wfst_host_alloc; here torch's pinned allocator):
        # wfst_decoder_advance_host then uploads by DMA slice by slice, a slice ahead of the search, and returns when enqueued
        # (ONE page-locked block for the batch, the utterances equally spaced in it: every slice of all 128 channels goes up as one 2-D copy)
        pinned = torch.from_numpy(np.ascontiguousarray(mats)).pin_memory()
        host_rows = [pinned[i].numpy() for i in range(B)]
    step = make_step(dec, ll_dev, host_rows)
    dt, res = timed(step, a.warmup, a.steps)
    frames_total = world * B * T * a.steps
    value = frames_total / dt
    if os.environ.get("WFST_BENCH_BREAKDOWN"):
        log("[rank %d] host-side ms per step: %s" % (rank, {k: round(1000.0 * v / max(tb["n"], 1), 2) for k, v in tb.items() if k != "n"}))

    if a.postprocess and a.lattice_links > 0 and rank == 0 and post_lms is not None:
        # the service's post-processing (kaldi-online-nnet3-my-decoder.cc:50-105) of the batch the last step left finalized: the
        # second LM pass of every determinized lattice, then the 10-best of the result -- one launch per stage, a workgroup per
        # lattice -- next to ONE channel's request served alone
        P1, P2 = post_lms
        ptrs_post = [ll_dev[i].data_ptr() for i in range(B)]
        dec.rescore_lattices(P1, P2)   # (first use: workspaces)
        dec.nbest_paths_batch(10, P1, P2)
        one = []
        for c in (0, B // 2, B - 1):   # (other requests than the kept ones: computed alone -- which also takes the workspace slots)
            t3 = time.perf_counter()
            dec.nbest_paths(c, 11, P1, P2)
            one.append(time.perf_counter() - t3)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        dec.rescore_lattices(P1, P2)       # determinize x 128, ComposeLattice x 2 x 128, fetch
        lat2 = [dec.rescored_lattice(c, P1, P2) for c in range(B)]
        t1 = time.perf_counter()
        dec.nbest_paths_batch(10, P1, P2)  # NShortestPath x 128 on the rescored lattices the slots still hold, fetch
        nb2 = [dec.nbest_paths(c, 10, P1, P2) for c in range(B)]
        t2 = time.perf_counter()
        # ... and in the order a service runs it on a freshly finalized batch: GetLattice's determinizer started right behind
        # FinalizeDecoding (wfst_decoder_prefetch_determinized), the best paths and the short n-best lists fetched beside it, then the
        # second pass -- which finds the determinized lattices in the workspace slots -- and the 10-best
        dec.init()
        dec.advance(ptrs_post, ready, P)
        dec.finalize()
        torch.cuda.synchronize(dev)
        s0 = time.perf_counter()
        dec.prefetch_determinized()
        dec.best_paths(cap=2 * T + 64)
        dec.nbest(a.nbest)
        s1 = time.perf_counter()
        dec.rescore_lattices(P1, P2)
        lat3 = [dec.rescored_lattice(c, P1, P2) for c in range(B)]
        s2 = time.perf_counter()
        dec.nbest_paths_batch(10, P1, P2)
        nb3 = [dec.nbest_paths(c, 10, P1, P2) for c in range(B)]
        s3 = time.perf_counter()
        same3 = all((x is None) == (y is None) and (x is None or all(np.array_equal(x[k], y[k]) for k in x)) for x, y in zip(lat3, lat2))
        post = {"utterances": B, "second_pass_lattices_ms": 1e3 * (t1 - t0), "nbest10_of_them_ms": 1e3 * (t2 - t1),
                "behind_a_prefetch": {"best_paths_and_short_nbest_ms": 1e3 * (s1 - s0), "second_pass_lattices_ms": 1e3 * (s2 - s1),
                                      "nbest10_of_them_ms": 1e3 * (s3 - s2), "same_lattices_as_without": bool(same3),
                                      "what": "FinalizeDecoding -> wfst_decoder_prefetch_determinized -> best paths + 5-best (the determinizer runs "
                                              "beside them) -> wfst_decoder_rescore_lattices (starts from the determinized lattices the slots hold) "
                                              "-> wfst_decoder_nbest_paths_batch(10)"},
                "one_channel_alone_ms": 1e3 * float(np.median(one)),
                "lattices": sum(x is not None for x in lat2), "mean_rescored_states": float(np.mean([x["n_states"] for x in lat2 if x is not None] or [0])),
                "mean_paths": float(np.mean([len(x) for x in nb2])),
                "what": "wfst_decoder_rescore_lattices + fetch of all; wfst_decoder_nbest_paths_batch(10) + fetch of all; "
                        "wfst_decoder_get_nbest_paths(11) of one channel alone (determinize + ComposeLattice x 2 + NShortestPath), median of three"}
    else:
        post = None
    # ---- roofline pass: one more step with HIP events around every kernel launch -----------
    gstats = [dec.stats(c) for c in range(B)]
    profiled_step_ms = 0.0
    if not a.no_profile_step:
        dec.set_profiling(True)
        torch.cuda.synchronize(dev)
        tp0 = time.perf_counter()
        res_p = step()
        if hasattr(step, "drain"):
            step.drain(res_p)
        torch.cuda.synchronize(dev)
        profiled_step_ms = 1e3 * (time.perf_counter() - tp0)   # the instrumented step itself (slower than a timed one: event pairs, no hipGraph)
    prof = dec.profile()
    if not a.no_profile_step:
        dec.set_profiling(False)

    regime = ("beam-only pruning (max_active never binds, min_active 0): bit-exact best-path parity with the reference CPU decoder"
              if a.max_active >= 1000000 and a.min_active == 0 else
              "max_active %d / min_active %d: where they bind the reference's cutoff depends on its hash-list visiting order "
              "(order-free parity, DESIGN.md section 4)" % (a.max_active, a.min_active))
    out = {
        # (`metric` stays under 100 characters -- the driver's record cuts it there; what the number is measured on and in which
        # pruning regime is config.workload / config.regime)
        "metric": ("frames/sec decoded with on-the-fly LM rescoring (biglm, BASELINE configs[3])" if a.biglm else
                   "frames/sec decoded, log-likelihoods handed over as %s host matrices every step (PCIe-inclusive)" % ("pageable" if a.host_pageable else "page-locked") if a.host_feed else
                   "frames/sec decoded (RTFx = value/100) at fixed beam" if a.lattice_links == 0 else
                   "frames/sec decoded with lattice generation%s (BASELINE configs[4])" % (" + determinization" if a.determinize else "")),
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic: seeded hclg-like graph, %s log-likes" % (
            "%d-hypothesis" % a.paths if a.workload == "multi" else "single-planted-path"),
        "config": {
            "workload": ("BASELINE configs[3] (biglm) on " if a.biglm else "") +
                        "BASELINE configs[1]: batch=%d utterances/GPU x %d frames, %d-arc HCLG, beam=%g, "
                        "max_active=%d, min_active=%d, %d pdfs" % (B, T, g.n_arcs, a.beam, a.max_active, a.min_active, P) +
                        (("; old LM %d states / %d arcs (scale -1), new LM %d states / %d arcs" % (
                            lm_info[0]["n_states"], lm_info[0]["n_arcs"], lm_info[1]["n_states"], lm_info[1]["n_arcs"])) if a.biglm else ""),
            "global_batch": world * B, "frames_per_utt": T, "parallelism": "utterance-sharded x%d (graph replicated)" % world,
            "rtfx": value / 100.0,
            "channel_groups": int(dec.n_groups),
            "regime": regime,
            "what": ("every word-labelled arc costs new LM - old LM, tokens keyed by (graph state, LM pair state); parity: bit-exact with the CPU "
                     "restatement of the reference's biglm decoder in its fixed DiffArpaLm mode (DESIGN.md section 4)" if a.biglm else
                     "forward links, lattice-beam pruning every prune_interval frames and at finalize, %d-best per utterance%s" % (
                         a.nbest, " + the determinized lattice of every utterance" if a.determinize else "") if a.lattice_links > 0 else
                     "InitDecoding -> AdvanceDecoding(all frames) -> FinalizeDecoding -> GetBestPath + LatticeToVector for every utterance"),
        },
    }
    pf = dec.path_flags()
    out["config"]["decoder_paths"] = pf
    if post is not None:
        out["postprocess"] = post
    if world > 1:
        # N > 1: every rank runs the configuration rank 0 reports (the kernel paths a decoder takes follow from its limits)
        tv = torch.tensor([pf[k] for k in sorted(pf)], dtype=torch.int64, device="cpu" if share else dev)
        tmin, tmax = tv.clone(), tv.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        out["config"]["decoder_paths_same_on_every_rank"] = bool(torch.equal(tmin, tmax))
        # N > 1: a live parity sample on EVERY rank -- its first two utterances against the CPU restatement (bit for bit: words,
        # transition-ids, tot_score) --, summed over the ranks
        import pyoracle

        if rank == 0:
            pyoracle.build_oracle()
        dist.barrier()
        orc = pyoracle.OracleDecoder()
        h = orc.load_graph(gpath)
        okr = 0
        for i in range(min(2, B)):
            o = orc.decode(h, pyoracle.Config(**cd), mats[i], m)
            okr += int(np.array_equal(o.words, res[i]["words"]) and np.array_equal(o.tids, res[i]["tids"]) and
                       np.float32(o.tot_score).tobytes() == np.float32(res[i]["tot_score"]).tobytes())
        orc.free_graph(h)
        tt = torch.tensor([okr, min(2, B)], dtype=torch.int64, device="cpu" if share else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        out["config"]["parity_per_rank_sample"] = "%d/%d utterances (two per rank) bit-exact vs the CPU restatement" % (int(tt[0].item()), int(tt[1].item()))
    if rank == 0 and world > 1 and os.environ.get("WFST_BENCH_CHECK_GATHER") == "1":
        # test hook (tests/test_gpu_multirank.py): the gathered results of ALL ranks against the oracle
        import pyoracle

        pyoracle.build_oracle()
        orc = pyoracle.OracleDecoder()
        h = orc.load_graph(gpath)
        sa = argparse.Namespace(**vars(a))
        ok = 0
        for u, r in enumerate(gathered):
            ll = make_utts(synth, g, m, u, 1, T, P, sa)[0]
            o = orc.decode(h, pyoracle.Config(**cd), ll, m)
            ok += int(np.array_equal(o.words, r["words"]) and np.float32(o.tot_score).tobytes() == np.float32(r["tot_score"]).tobytes()
                      and np.float32(o.lm_score).tobytes() == np.float32(r["lm_score"]).tobytes())
        orc.free_graph(h)
        out["config"]["gather_check"] = {"utterances": len(gathered), "bit_exact_vs_oracle": ok}
        if gathered_lat:
            # the gathered lattices of ALL ranks against what ONE rank makes of the same utterances: rank 0 decodes every rank's block
            # itself (its own decoder, after the timed region) and compares the blobs byte for byte
            same = 0
            for r in range(world):
                blk = make_utts(synth, g, m, r * B, B, T, P, sa)
                tb_ = torch.from_numpy(blk).to(dev)
                dec.init()
                dec.advance([tb_[i].data_ptr() for i in range(B)], ready, P)
                dec.finalize()
                dec.best_paths(cap=2 * T + 64)
                for i in range(B):
                    same += int(shard.lattice_to_bytes(dec.determinized_lattice(i)) == gathered_lat[r * B + i])
                del tb_
            out["config"]["lattice_gather_check"] = "%d/%d" % (same, len(gathered_lat))
    if rank == 0:
        N = sum(s["N"] for s in gstats)
        E = sum(s["E"] for s in gstats)
        Z = sum(s["Z"] for s in gstats)
        toks = sum(s["tokens"] for s in gstats)
        out["config"]["mean_active_tokens_per_frame"] = toks / float(B * (T + 1))
        out["config"]["mean_expanded_tokens_per_frame"] = N / float(B * T)
        out["config"]["peak_tokens_in_a_frame"] = max(s["peak_tokens"] for s in gstats)
        out["config"]["max_tokens_per_frame_limit"] = "library default (262144 at a finite max_active, arena 4 M tokens)" if a.default_limits else a.max_tokens
        out["config"]["max_tokens_per_frame_note"] = ("a best-path decoder does not fail at this limit, it goes on from the limit-th cheapest token "
                                                      "(degraded_frames counts the frames on which it did); the limit also sizes the arena's collection "
                                                      "reserve (a quarter of the arena at most): every gc_stride-th frame (reserve / limit - 1, at most 16) runs the classic three launches")
        if a.lattice_links == 0 and not a.biglm:
            out["config"]["degraded_frames"] = int(sum(dec.degraded_frames(c) for c in range(B)))
        out["config"]["utterances_with_path"] = int(sum(1 for r in res if r["ok"]))
        if getattr(res, "nbest_lazy", None) is not None:   # (outside the timed region: the n-best lists as per-utterance entries)
            for i, r in enumerate(res):
                if res.nbest_lazy[i] is not None:
                    r["nbest"] = res.nbest_lazy[i]
        if a.lattice_links > 0 and any("nbest" in r for r in res):
            # the 1-best of the step's n-best against the best path: the same word sequence (the determinized lattice keeps, for
            # every word sequence, its cheapest path)
            nb_ok = sum(1 for r in res if r["ok"] and r.get("nbest") and np.array_equal(np.asarray(r["nbest"][0]["words"]), np.asarray(r["words"])))
            out["config"]["nbest_1best_same_words_as_best_path"] = "%d/%d" % (nb_ok, sum(1 for r in res if r["ok"]))
            out["config"]["nbest_from"] = ("NShortestPath on the DETERMINIZED lattice, behind the determinizer on its stream (wfst_decoder_prefetch_nbest)"
                                           if (a.determinize and not a.no_prefetch and not a.raw_nbest) else "the k-best search over the raw lattice (wfst_decoder_get_nbest)")
        if a.lattice_links > 0 and a.determinize:
            dl = [r["det"] for r in res if r.get("det") is not None]
            out["config"]["determinized_lattices"] = {
                "utterances": len(dl), "mean_states": float(np.mean([d["n_states"] for d in dl])) if dl else 0.0,
                "mean_arcs": float(np.mean([len(d["a_src"]) for d in dl])) if dl else 0.0}
            # the device's own time per lattice (the launch lasts as long as its largest lattice; each lattice's workgroup clocks
            # itself: wfst_decoder_get_determinizer_ms), beside the reference's DeterminizeLatticeWrapper on a host core (cpu_baseline)
            dms = [x for x in (dec.determinizer_ms(c) for c in range(B)) if x is not None]
            if dms:
                out["config"]["determinized_lattices"]["gpu_ms_per_lattice_mean"] = float(np.mean(dms))
                out["config"]["determinized_lattices"]["gpu_ms_per_lattice_max"] = float(np.max(dms))
                out["config"]["determinized_lattices"]["gpu_ms_per_lattice_median"] = float(np.median(dms))
            if a.pipeline_determinizer:
                out["config"]["determinizer"] = ("pipelined (wfst_decoder_prefetch_determinized_detached): a step's lattices are determinized on a "
                                                 "side stream beside the NEXT step's decode and fetched one step later; the last step's are waited for "
                                                 "inside the timed region (K steps = K determinizations of 128 lattices)")
        # ---- CPU baseline + live parity on a bounded sample ---------------------------------
        scale = 1.0
        cpus = affinity_cpus()
        do_cpu = a.cpu_sample > 0 and world == 1  # the CPU legs are reported at N=1 only
        if do_cpu:
            ns = min(a.cpu_sample, B)
            nth = a.cpu_threads or cpus
            kind, cdec = cpu_decoder()
            sample = [mats[i] for i in range(ns)]
            if a.biglm:
                # parity target: the restatement in FIXED DiffArpaLm mode (the reference as written hands the
                # pair id to both LMs, newlm/diff-lm.h:80,86); the timed baseline below is the reference's own
                # biglm decoder where its library is present -- same loop, same amount of LM work
                import pyoracle

                pyoracle.build_oracle()
                cres = cpu_decode_all(pyoracle.OracleDecoder(), gpath, cd, sample, m, min(ns, cpus), big=big)
            else:
                cres = cpu_decode_all(cdec, gpath, cd, sample, m, min(ns, cpus))
            dv = divergence(res[:ns], cres)
            if not a.no_cpu_baseline:
                # timed legs: one thread, then every CPU this process may run on
                lat_n = a.nbest if (a.lattice_links > 0 and a.determinize and kind == "reference") else None
                fps1, cdt1, fr1, ex1 = cpu_timed(kind, cdec, gpath, cd, list(mats), m, 1, a.cpu_seconds, big=big, lattice=lat_n)
                fps, cdt, fr, exn = cpu_timed(kind, cdec, gpath, cd, list(mats), m, nth, a.cpu_seconds, big=big, lattice=lat_n)
                # a point in between (how the CPU decoder scales over one shared graph: it is bound by random access to it)
                curve = {}
                for nmid in (8, 32, 64):
                    if nmid < nth:
                        curve[str(nmid)] = cpu_timed(kind, cdec, gpath, cd, list(mats), m, nmid, max(2.0, a.cpu_seconds / 3), big=big, lattice=lat_n)[0]
                cpu_model = ""
                try:
                    with open("/proc/cpuinfo") as f:
                        cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
                except OSError:
                    pass
                curve = dict(curve, **{"1": fps1, str(nth): fps})
                best_n = max(curve, key=lambda k: curve[k])   # the CPU's best: more threads than that lose (random access to one shared graph)
                out["cpu_baseline"] = {"value": curve[best_n], "unit": "frames/s", "cores": int(best_n), "kind": kind,
                                       "single_thread_value": fps1, "all_cpus_value": fps, "cpu_model": cpu_model,
                                       "sample": "rank 0's %d utterances, looped by every thread; each host thread has its decoder object over one shared "
                                                 "graph; legs of 1, 8, 32 and %d threads (%.1fs wall for the first and the "
                                                 "last: %d and %d frames decoded); value = the best leg" % (B, nth, cdt, fr1, fr),
                                       "threads_to_value": curve,
                                       "affinity_cpus": cpus, "host_cpus": os.cpu_count()}
                if a.biglm:
                    out["cpu_baseline"]["what"] = ("the reference's biglm decoder (kaldi-hclg-my-decoder-biglm.cc:80-102: InitDecoding, AdvanceDecoding, "
                                                   "FinalizeDecoding, GetBestPath) with the same two LMs" if kind == "reference" else "the restatement's biglm decoder")
                elif lat_n is not None:
                    out["cpu_baseline"]["what"] = ("the reference's lattice pipeline per utterance: lattice-mode decode (forward links, PruneActiveTokens every "
                                                   "prune_interval frames), FinalizeDecoding, GetBestPath, GetRawLattice, DeterminizeLatticeWrapper, "
                                                   "NShortestPath(%d) (kaldi-online-nnet3-my-decoder.cc:50-105)" % lat_n)
                    out["cpu_baseline"]["determinizer_ms_per_lattice"] = ex1.get("determinizer_ms_per_lattice")   # one thread: a core to itself
                    out["cpu_baseline"]["one_thread_stages"] = ex1
                    out["cpu_baseline"]["all_threads_stages"] = exn
                elif a.lattice_links > 0:
                    out["cpu_baseline"]["what"] = ("the reference decoder's best-path call sequence (it always records forward links and back-prunes); "
                                                   "GetRawLattice / determinizer / n-best not included")
            if a.lattice_links > 0:
                # lattice mode: the raw lattice itself (GetRawLattice after FinalizeDecoding), state by state and arc by arc
                # against the CPU restatement in its order-free mode (deviation 6), on the first utterances
                import pyoracle

                pyoracle.build_oracle()
                orc = pyoracle.OracleDecoder()
                orc.set_order_free(True)
                hh = orc.load_graph(gpath)
                nl, okl = min(ns, 16), 0

                def same_raw(i):
                    O = pyoracle.oracle_raw_lattice(orc, hh, pyoracle.Config(**cd), mats[i], m)   # (one shared graph, a decode per thread: as cpu_decode_all)
                    dl = raw_dev[i]
                    if dl is None or O is None:
                        return 0
                    L = pyoracle.RawLattice(True, dl["n_states"], 0, dl["st_final"], dl["a_src"], dl["a_dst"], dl["a_ilabel"], dl["a_olabel"],
                                            dl["a_graph"], dl["a_acoustic"], dl["st_frame"], dl["st_state"], dl["st_cost"])
                    return int(L.n_states == O.n_states and np.array_equal(L.labelled_arcs(), O.labelled_arcs()))

                raw_dev = [dec.raw_lattice(i) for i in range(nl)]
                from concurrent.futures import ThreadPoolExecutor

                with ThreadPoolExecutor(max_workers=min(nl, max(1, min(16, cpus)))) as ex:
                    okl = sum(ex.map(same_raw, range(nl)))
                orc.set_order_free(False)
                orc.free_graph(hh)
                out["config"]["lattice_parity"] = "%d/%d sampled raw lattices arc for arc equal to the CPU restatement's (order-free mode)" % (okl, nl)
            out["config"]["parity"] = "%d/%d sampled utterances bit-exact (words, transition-ids, tot_score) vs the %s CPU decoder" % (
                dv["bit_identical"], ns, "oracle (biglm, fixed mode)" if a.biglm else "reference" if kind == "reference" else "oracle")
            if a.biglm:
                oc = {k: sum(r.extra[k] for r in cres) for k in ("N", "E", "Z", "L", "L_eps")}
                oc["ties_on_best_path"] = sum(r.extra["ties"] for r in cres)
                oc["lm_pairs_max"] = max(r.extra["lm_pairs"] for r in cres)
                n_lm = oc["L"]
            else:
                oc = oracle_counts(gpath, cd, sample, m)
            if dv["bit_identical"] < ns and not a.biglm:
                # where max_active / min_active bind, the reference's cutoff depends on tokens it
                # keeps in hash-list visiting order; the order-independent restatement of the same
                # algorithm (oracle, order-free mode) is what the GPU must equal bit for bit
                paths = []
                oracle_counts(gpath, cd, sample, m, order_free=True, want_paths=paths)
                ex2 = sum(1 for i in range(ns) if np.array_equal(paths[i][0], res[i]["tids"]) and
                          np.float32(paths[i][1]).tobytes() == np.float32(res[i]["tot_score"]).tobytes())
                out["config"]["parity"] += "; %d/%d bit-exact vs the oracle in order-free mode" % (ex2, ns)
                out["config"]["divergence_vs_%s_sample" % kind] = dv
            gs = {k: sum(gstats[i][k] for i in range(ns)) for k in ("N", "E", "Z")}
            out["config"]["work_counts_sample"] = {"oracle": oc, "gpu": gs}
            # algorithmic bytes from the CPU restatement's counts, scaled from the sample to the batch
            # by the GPU's own (identically defined) N+E counters
            if gs["N"] + gs["E"] > 0:
                scale = (oc["N"] + oc["E"]) / float(gs["N"] + gs["E"])
        # Algorithmic bytes: 28 B per traversed emitting arc (16 B arc + 4 B log-like
        # + 8 B hash min-update), 24 B per expanded token (8 B {state,cost} + 8 B arc range + 8 B
        # backpointer/arena write), 24 B per traversed epsilon arc.  Per kernel (DESIGN.md "Roofline
        # accounting"): expand = 20 E + 16 N, insert = 8 E + 8 N, closure = 24 Z.
        fused = not a.no_fuse and not a.biglm and (a.lattice_links == 0 or not (a.debug & 0x1000))
        if fused and do_cpu:
            # fused epsilon closures: the closure's arcs are priced by the expansion (16 B pseudo arc) and merged by the
            # insert launch (8 B hash min-update); Z = the CPU restatement's count (the GPU's own counts one closure
            # path per CANDIDATE at a state, the reference one per TOKEN), scaled from the sample like N and E
            Zo = oc["Z"] * (E / float(max(oc["E"], 1)))
            kb = {"expand": scale * (20.0 * E + 16.0 * N) + 16.0 * Zo, "insert": scale * (8.0 * E + 8.0 * N) + 8.0 * Zo, "closure": 0.0}
        else:
            kb = {"expand": scale * (20.0 * E + 16.0 * N), "insert": scale * (8.0 * E + 8.0 * N), "closure": 24.0 * Z}
        Z8d = (oc["Z"] * (E / float(max(oc["E"], 1)))) if (fused and do_cpu) else float(Z)   # traversed epsilon arcs of the batch (the CPU restatement's count where the closures are fused)
        out["config"]["fused_epsilon_closures"] = bool(fused)
        if a.lattice_links > 0:
            # lattice mode adds (DESIGN.md "Roofline accounting"): 16 B per forward link recorded (insert launch); per back-pruning
            # sweep 16 B link + 8 B {extra, cost} of its destination per link priced and 8 + 8 B per token priced; per compaction 12 B
            # per item scanned (8 B token pair / 16 B link) and 2 x 16 B per survivor moved -- counted on the device
            # (wfst_decoder_get_lattice_stats), one step's worth
            ls = [dec.lattice_stats(c) for c in range(B)]
            lk = {k: float(sum(x[k] for x in ls)) for k in ls[0]}
            kb["insert"] += 16.0 * lk["links_recorded"]
            kb["closure"] += 24.0 * lk["walk_links"] + 16.0 * lk["walk_tokens"] + 12.0 * lk["compaction_scanned"] + 32.0 * lk["compaction_moved"]
            out["config"]["lattice_work_per_step"] = lk
        if a.biglm and do_cpu and oc["E"] > 0:
            # biglm: + 96 B per word-labelled arc traversed (8 B pair key; per LM 16 B state record, 4 B x ~4 probes of
            # the word-sorted arcs, 8 B arc {weight, next}; 8 B pair-table slot), + 4 B LM pair id per token and record
            Lb = n_lm * (E / float(max(oc["E"], 1)))   # look-ups of the whole batch, scaled from the sample by the emitting arcs
            Leps = oc["L_eps"] * (E / float(max(oc["E"], 1)))   # ... of them on epsilon arcs: made by the closure pass, charged to it
            kb["expand"] += 96.0 * (Lb - Leps) + scale * (4.0 * E + 4.0 * N)
            kb["insert"] += scale * (4.0 * E + 4.0 * N)
            kb["closure"] += 96.0 * Leps + 4.0 * Z
            # ... and what the closure PASS of a biglm decoder moves besides its epsilon arcs (the counters read 16x
            # the 24 Z model): ProcessNonemitting's worklist looks at EVERY token of the frame for epsilon arcs -- the token and its
            # state's row header, 16 B each (base-inl.h:376-381: `NumInputEpsilons(state) != 0` per element) --, GetCutoff reads every
            # token's cost once more (16-byte token records, base-inl.h:139-234), and every token the closure creates or improves is a
            # 16 B record + 4 B pair id written and a 16 B slot of the frame's (state, pair) table probed.  Counts: the frame's tokens
            # from the device (wfst_decoder_get_stats), the closure's arrivals = Z.
            kb["closure"] += 32.0 * toks + 16.0 * toks + 36.0 * Z
            out["config"]["closure_pass_model"] = ("24 Z + 96 L_eps + 4 Z (epsilon arcs, LM look-ups on them, pair ids) + 48 B per frontier token "
                                                   "(worklist: token + row header; GetCutoff: the token again) + 36 B per epsilon arrival (token record, pair id, table slot)")
            out["config"]["lm_lookups_per_step"] = Lb
            out["config"]["lm_lookups_on_epsilon_arcs_per_step"] = Leps
        dom = max(("expand", "insert", "closure"), key=lambda k: prof[k + "_ms"])
        k_ms, k_n, k_bytes = prof[dom + "_ms"], prof[dom + "_launches"], kb[dom]
        per_launch_bytes = k_bytes / max(k_n, 1)
        avg_ms = k_ms / max(k_n, 1)
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        all_ms = prof["expand_ms"] + prof["insert_ms"] + prof["closure_ms"]
        # HBM traffic per launch: rocprofv3 --pmc passes cannot be collected inside this process; they are child runs of this script
        # (measure_traffic), made NOW.  A measurement that cannot be made leaves `traffic` null -- the stored passes of an earlier
        # run (profiles/traffic_latest.json) are quoted under `traffic_stored`, with their origin, never under `traffic`.
        traffic, traffic_src, traffic_stored = None, None, None
        tj = os.path.join(ROOT, "profiles", "traffic_latest.json")
        # the stored passes were made on three configurations: the headline workload, the biglm leg and the beam-15 lattice leg
        std = a.batch == 128 and a.frames == 300 and a.states == 2850000 and a.workload == "multi" and a.max_active == 1000000 and a.min_active == 0
        which = None
        if std and a.biglm and a.beam == 13.0 and a.lattice_links == 0:
            which = "biglm"
        elif std and not a.biglm and a.lattice_links > 0 and a.beam == 15.0 and a.lattice_beam == 8.0:
            which = "lattice_beam15"
        elif std and not a.biglm and a.lattice_links == 0 and a.beam == 13.0:
            which = "headline"
        if which is not None and os.path.exists(tj):
            try:
                tjd = json.load(open(tj))
                ts = tjd.get(which, {}).get(dom + "_bytes_per_launch")
                if ts is not None:   # (stored per whole-batch launch of a one-group run: brought to this run's launch count)
                    traffic_stored = {"bytes_per_launch": traffic_per_launch(ts, stored_launches_per_step(tjd[which], dom, k_n), k_n),
                                      "origin": "NOT measured in this run: profiles/traffic_latest.json [%s] (%s)" % (which, tjd.get("origin", "stored rocprofv3 --pmc passes"))}
            except Exception:
                traffic_stored = None
        traffic_src = "not measured in this run" + (" (--no-traffic)" if a.no_traffic else "")
        measured = None
        ng = int(dec.n_groups)
        if world == 1 and not a.no_traffic and not a.no_hip_graph:
            wl = ["--batch", str(B), "--frames", str(T), "--states", str(a.states), "--pdfs", str(P), "--beam", str(a.beam), "--max-active", str(a.max_active),
                  "--min-active", str(a.min_active), "--workload", a.workload, "--paths", str(a.paths), "--mu", str(a.mu), "--sigma", str(a.sigma),
                  "--max-tokens", str(a.max_tokens), "--arena-per-frame", str(a.arena_per_frame), "--lattice-beam", str(a.lattice_beam),
                  "--prune-interval", str(a.prune_interval)]
            if a.biglm:
                wl += ["--biglm", "--lm-old", a.lm_old, "--lm-new", a.lm_new, "--lm-pairs", str(a.lm_pairs)]
            if a.lattice_links > 0:
                wl += ["--lattice-links", str(a.lattice_links), "--nbest", str(a.nbest)] + (["--determinize"] if a.determinize else [])
            if a.default_limits:
                wl += ["--default-limits"]
            # (the children allocate a decoder of their own on this GPU: this process lets go of its own first -- everything below
            # that needs it has been read; the service-point and leg runs make theirs anew)
            dec.free()
            dec = None
            measured = measure_traffic(a, wl)
            if measured is not None:
                # (the passes run ONE step with one channel group: a launch there covers the whole batch; per launch of THIS run =
                # the child's bytes per launch x its launches per step / this run's launches per step)
                traffic = traffic_per_launch(measured[0].get(dom, 0.0), measured[1].get(dom, 0), k_n)
                traffic_src = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child runs of this script (ONE step each and no "
                               "instrumented extra step, one channel group, kernels enqueued one by one, %.0f s), (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch" % measured[2])
            else:
                traffic_src = "the rocprofv3 --pmc child passes of this run failed: traffic is null (stderr has the reason)"
        step_ms = 1000.0 * dt / a.steps
        whole_bytes = sum(kb.values())
        out["roofline"] = {"bound": "hbm", "kernel": dom + "_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                           "traffic_measured_in_run": measured is not None,
                           "traffic_per_class": ({"bytes_per_launch": measured[0], "launches_per_step": measured[1], "steps": 1} if measured is not None else None),
                           "traffic_stored": traffic_stored,
                           "traffic_over_algorithmic": (traffic / per_launch_bytes) if (traffic and per_launch_bytes > 0) else None,
                           "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": avg_ms, "launches": k_n,
                           "kernel_ms_per_step": {k: prof[k + "_ms"] for k in ("expand", "insert", "closure")},
                           "all_kernels_achieved_GBs": (whole_bytes / (all_ms * 1e-3) / 1e9) if all_ms > 0 else 0.0,
                           "profiled_step_ms": profiled_step_ms,
                           "whole_path": {"algorithmic_bytes_per_step": whole_bytes,
                                          "formula": ("28 E + 24 N + 24 Z (SURVEY.md 8(d)), counts of one step of this rank" +
                                                      ("; + lattice terms (builder-defined, DESIGN.md 'Roofline accounting'): 16 B per forward link recorded, "
                                                       "24 B per link and 16 B per token priced by a back-pruning sweep, 12 B per item scanned and 32 B per "
                                                       "survivor moved by a compaction" if a.lattice_links > 0 else "") +
                                                      ("; + biglm terms (builder-defined): 96 B per LM look-up (charged to the kernel that makes it: expansion for emitting arcs, closure pass for epsilon arcs), 4 B pair id per token and record; closure pass: 48 B per frontier token (ProcessNonemitting's worklist scan + GetCutoff) and 36 B per epsilon arrival" if a.biglm else "")),
                                          "frac_over_kernel_time": (whole_bytes / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if all_ms > 0 else 0.0,
                                          "frac_over_step_time": whole_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          # ... and on SURVEY.md 8(d)'s own three terms alone (no builder-defined additions)
                                          "bytes_8d_formula": scale * (28.0 * E + 24.0 * N) + 24.0 * (Z8d),
                                          "frac_8d_formula_over_step_time": (scale * (28.0 * E + 24.0 * N) + 24.0 * (Z8d)) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "measured": "hipEvent pairs around every launch on the stream it is launched on, one extra step after the timed region"}
        if ng > 1:
            # Channel groups: each group's launches run on its own stream and overlap the other group's (that is what the
            # groups are for), so "one launch" shares the chip with another launch of the same kernel for part of its
            # duration and bytes-of-one-launch / its-duration is not the rate the chip sustains.  `achieved` is therefore
            # taken over the kernel's BUSY time: bytes of all its launches in the step / the time during which at least one
            # of them was executing (union of the launches' event intervals).  With one group the two definitions coincide.
            # The per-launch figures stay in `per_launch` for the cross-check against rocprofv3's average duration.
            busy = prof[dom + "_busy_ms"]
            R = out["roofline"]
            R["per_launch"] = {"achieved": R["achieved"], "frac": R["frac"], "algorithmic_bytes_per_launch": per_launch_bytes,
                               "avg_launch_ms": avg_ms, "launches": k_n,
                               "note": "%d channel groups: a launch covers 1/%d of the batch and overlaps the other group's launches" % (ng, ng)}
            R["achieved"] = (k_bytes / (busy * 1e-3) / 1e9) if busy > 0 else 0.0
            R["frac"] = R["achieved"] / HBM_PEAK_GBS
            R["channel_groups"] = ng
            R["kernel_busy_ms_per_step"] = {k: prof[k + "_busy_ms"] for k in ("expand", "insert", "closure")}
            R["achieved_definition"] = ("algorithmic bytes of all %s launches of one step / time during which at least one of them was "
                                        "executing (the groups' launches overlap; per-launch figures under per_launch)" % (dom + "_kernel"))
    # ---- second workload: SURVEY 8(d) generator at the reference service's operating point ----
    if rank == 0 and world == 1 and not a.no_service_point and a.lattice_links == 0 and not a.host_feed and a.workload == "multi" and not a.biglm:
        if dec is not None:
            dec.free()
        dec = None
        t0 = time.time()
        cd2 = dict(cd, max_active=7000, min_active=200)
        n2 = max(2, a.steps // 4)

        def at_service_point(mats2, ll2, spread=False):
            """decode at 7000/200; with the CPU legs on: GPU result vs the reference decoder's own, and the
            reference against ITSELF with nothing but its hash table size changed (hash_ratio 3 instead of
            2: another visiting order of the same algorithm) -- the yardstick for the first number"""
            nonlocal dec
            dec = new_decoder(cd2)   # (the per-frame limit of a best-path decoder is no capacity: frames beyond it degrade, none did here)
            step2 = make_step(dec, ll2, [mats2[i] for i in range(B)])
            dt2, res2 = timed(step2, 1, n2)
            o = {"value": B * T * n2 / dt2, "unit": "frames/s", "ms_per_step": 1000.0 * dt2 / n2, "steps": n2,
                 "mean_active_tokens_per_frame": sum(dec.stats(c)["tokens"] for c in range(B)) / float(B * (T + 1)),
                 "degraded_frames": int(sum(dec.degraded_frames(c) for c in range(B)))}
            dec.free()
            dec = None
            if a.cpu_sample > 0:
                kind, cdec = cpu_decoder()
                cres2 = cpu_decode_all(cdec, gpath, cd2, list(mats2), m, affinity_cpus())
                o["divergence_vs_" + kind] = divergence(res2, cres2)
                cres3 = cpu_decode_all(cdec, gpath, dict(cd2, hash_ratio=3.0), list(mats2), m, affinity_cpus())
                as_res = lambda rs: [dict(ok=r.ok, words=r.words, tids=r.tids, tot_score=r.tot_score) for r in rs]
                o[kind + "_self_divergence_hash_ratio_3_vs_2"] = divergence(as_res(cres3), cres2)
                if spread:
                    # a third visiting order of the reference (hash_ratio 2.5) and every pair -- the reference's own
                    # spread, and the GPU (= the order-free restatement, bit for bit) against each of the three
                    cres25 = cpu_decode_all(cdec, gpath, dict(cd2, hash_ratio=2.5), list(mats2), m, affinity_cpus())
                    brief = lambda d: {k: d[k] for k in ("bit_identical", "same_words", "wer", "max_rel_cost_gap")}
                    sp_ = {"reference@2.5 vs reference@2": brief(divergence(as_res(cres25), cres2)),
                           "reference@3 vs reference@2": brief(o[kind + "_self_divergence_hash_ratio_3_vs_2"]),
                           "reference@3 vs reference@2.5": brief(divergence(as_res(cres3), cres25)),
                           "gpu vs reference@2": brief(o["divergence_vs_" + kind]),
                           "gpu vs reference@2.5": brief(divergence(res2, cres25)),
                           "gpu vs reference@3": brief(divergence(res2, cres3))}
                    rr = [v["wer"] for k, v in sp_.items() if not k.startswith("gpu")]
                    gg = [v["wer"] for k, v in sp_.items() if k.startswith("gpu")]
                    sp_["reference_vs_reference_wer_range"] = [min(rr), max(rr)]
                    sp_["gpu_vs_reference_wer_range"] = [min(gg), max(gg)]
                    sp_["note"] = ("the GPU computes ProcessEmitting's next_cutoff as the minimum over ALL arrivals before admitting any (the reference "
                                   "tightens it while it walks its hash list, base-inl.h:321-333, so every visiting order admits a superset): it is "
                                   "the limit point of the reference's orders, further from each of them than they are from each other, with no "
                                   "bias in path cost (signed_rel_cost_gap); profiles/r05_parity_spread.json, DESIGN.md section 4 deviation 2")
                    o["spread"] = sp_
            return o

        sa = argparse.Namespace(**vars(a))
        sa.workload, sa.mu = "single", -2.0
        mats2 = make_utts(synth, g, m, 0, B, T, P, sa)
        ll2 = torch.from_numpy(mats2).to(dev)
        sp = {"workload": "SURVEY.md 8(d) single-planted-path log-likelihoods (mu -2, sigma 1), same graph and batch, beam=%g, "
                          "max_active=7000, min_active=200 (v1-asrbin/conf/decoder.conf:4-8)" % a.beam}
        sp.update(at_service_point(mats2, ll2, spread=True))
        del ll2, mats2
        # third leg: the same generator calibrated as SURVEY 8(d) asks -- mu -2.6 gives ~5.5 k tokens per frame at
        # beam 13, ~4 k of them expanded -- so that max_active 7000 binds on a minority of the frames
        sa.mu = -2.6
        mats3 = make_utts(synth, g, m, 0, B, T, P, sa)
        ll3 = torch.from_numpy(mats3).to(dev)
        cp = {"workload": "SURVEY.md 8(d) single-planted-path log-likelihoods calibrated to ~5 k tokens per frame (mu -2.6, sigma 1), max_active=7000, min_active=200"}
        cp.update(at_service_point(mats3, ll3))   # (the three-order spread on the first workload only: a full pass of the reference over the batch less)
        del ll3, mats3
        sp["calibrated_workload_at_7000_200"] = cp
        hp = {"workload": "the headline log-likelihoods at max_active=7000, min_active=200"}
        hp.update(at_service_point(mats, ll_dev))
        sp["headline_workload_at_7000_200"] = hp
        # the reference's DEFAULT limits (lattice-faster-decoder-conf.h:35-44: max_active INT_MAX, min_active 200) on the headline
        # workload: min_active binds on the first frames of an utterance only -- the two-launch frames and the staged expansion stay
        def at_limits(cd3, label):
            nonlocal dec
            dec = new_decoder(cd3)
            step3 = make_step(dec, ll_dev, [mats[i] for i in range(B)])
            dt3, res3 = timed(step3, 1, max(4, n2))
            o = {"workload": label, "value": B * T * max(4, n2) / dt3, "unit": "frames/s", "ms_per_step": 1000.0 * dt3 / max(4, n2), "steps": max(4, n2)}
            dec.free()
            dec = None
            if a.cpu_sample > 0:
                kind, cdec = cpu_decoder()
                ns3 = min(a.cpu_sample, B)
                # (the reference sizes its hash table max_active x hash_ratio, base-inl.h:27: 68 GB per decoder object at INT_MAX --
                # on the host it runs with a max_active that never binds either, 10^6: the same search)
                cd3_cpu = dict(cd3, max_active=min(int(cd3["max_active"]), 1000000))
                o["divergence_vs_" + kind] = divergence(res3[:ns3], cpu_decode_all(cdec, gpath, cd3_cpu, [mats[i] for i in range(ns3)], m, min(ns3, affinity_cpus())))
            return o

        rdl = at_limits(dict(cd, max_active=2147483647, min_active=200),
                        "the headline log-likelihoods at the reference's default limits: max_active=INT_MAX, min_active=200")
        sp["divergence_note"] = ("where max_active/min_active bind, the reference's cutoff depends on its own hash-list visiting "
                                 "order (DESIGN.md section 4, deviation 2): it then differs from itself when only hash_ratio "
                                 "changes; the GPU computes the order-independent restatement")
        out["service_point"] = sp
        L = out.setdefault("legs", {})
        L["service_point_7000_200"] = {k: v for k, v in sp.items() if k not in ("calibrated_workload_at_7000_200", "headline_workload_at_7000_200")}
        L["calibrated_7000_200"] = cp
        L["headline_at_7000_200"] = hp
        L["reference_default_limits"] = rdl
        log("[rank 0] service point: %.1fs" % (time.time() - t0))
    headline_run = (not a.biglm and a.lattice_links == 0 and not a.host_feed and a.workload == "multi")
    if rank == 0 and world == 1 and headline_run and not a.no_legs:
        # ---- the other BASELINE configurations, driver-observed: configs[3] (biglm) and configs[4] (lattice-generating decode at
        # beam 15 + the determinized lattices), each a child process running this script on the same graph and utterances
        # for a few steps, with its own live parity sample, token counts, per-kernel times and roofline; the child's line
        # is embedded as it is (minus the bulky parts).  This process has released its decoder: the child has the GPU to itself.
        import subprocess

        if dec is not None:
            dec.free()
            dec = None
        del ll_dev
        torch.cuda.empty_cache()
        n2 = max(2, a.steps // 5)
        common = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--warmup", "1", "--no-service-point", "--no-legs", "--no-traffic",
                  "--batch", str(B), "--frames", str(T), "--states", str(a.states), "--pdfs", str(P)]
        # the reference's own CPU path timed beside configs[3] and configs[4] (1 / 32 / all host threads): its biglm decoder
        # (kaldi-hclg-my-decoder-biglm.cc:80-102) and its lattice pipeline (decode, GetRawLattice, DeterminizeLatticeWrapper,
        # NShortestPath: kaldi-online-nnet3-my-decoder.cc:50-105); the other legs vary the same two configurations and carry none
        cpu_on = ["--cpu-seconds", str(min(a.cpu_seconds, 4.0)), "--cpu-threads", str(min(64, a.cpu_threads or affinity_cpus()))] if (a.cpu_sample > 0 and not a.no_cpu_baseline) else ["--no-cpu-baseline"]
        cpu_off = ["--no-cpu-baseline"]
        legs = {# the headline with wfst_limits all zero: what a caller who sizes nothing gets
                "headline_library_default_limits": ["--default-limits", "--steps", str(max(6, n2)), "--cpu-sample", "4"] + cpu_off,
                # the headline with the log-likelihoods handed over as HOST matrices inside every step (PCIe-inclusive; never the reported value)
                "host_feed": ["--host-feed", "--steps", str(max(6, n2)), "--cpu-sample", "4"] + cpu_off,
                "biglm": ["--biglm", "--steps", str(max(6, n2)), "--cpu-sample", "32", "--max-tokens", "131072"] + cpu_on,
                # ... at lattice_beam 7 the reference's biglm final pruning (biglm.h:186-188) leaves 45 of the 128 utterances a path; at 14,
                # 121 of them: the same search (the beam is what it costs), a result for nearly every utterance
                "biglm_lattice_beam14": ["--biglm", "--lattice-beam", "14", "--steps", str(max(6, n2)), "--cpu-sample", "8", "--max-tokens", "131072"] + cpu_off,
                "lattice_beam13": ["--lattice-links", "25165824", "--steps", str(max(4, n2)), "--cpu-sample", "16", "--warmup", "2", "--postprocess"] + cpu_on,
                "lattice_beam15_no_determinizer": ["--beam", "15", "--lattice-beam", "8", "--lattice-links", "25165824", "--arena-per-frame", "60000",
                                                   "--max-tokens", "262144", "--steps", str(max(4, n2 // 2)), "--cpu-sample", "2", "--warmup", "2"] + cpu_off,
                "lattice_beam15": ["--beam", "15", "--lattice-beam", "8", "--lattice-links", "25165824", "--arena-per-frame", "60000",
                                   "--max-tokens", "262144", "--determinize", "--steps", str(max(4, n2 // 2)), "--cpu-sample", "16",
                                   "--warmup", "2"] + cpu_on,   # (the n-best / determinizer paths allocate their workspaces on first use)
                # ... and as a service that refills its channels at once would run it: utterance k's lattices determinized beside
                # utterance k + 1's decode (wfst_decoder_prefetch_determinized_detached), fetched one step later, the last step's
                # waited for inside the timed region
                "lattice_beam15_pipelined": ["--beam", "15", "--lattice-beam", "8", "--lattice-links", "25165824", "--arena-per-frame", "60000",
                                             "--max-tokens", "262144", "--determinize", "--pipeline-determinizer", "--steps", str(max(8, n2)),
                                             "--cpu-sample", "4", "--warmup", "2"] + cpu_off}
        L = out.setdefault("legs", {})
        only = [x for x in a.only_legs.split(",") if x]
        for name, extra in legs.items():
            if only and name not in only:
                continue
            t0 = time.time()
            dpath = "/tmp/wfst_bench_leg_%d_%s.json" % (os.getpid(), name)
            pr = None
            try:
                pr = subprocess.run(common + extra + ["--detail-out", dpath], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
                with open(dpath) as f:   # the child's FULL result (its own stdout line is the summary of it)
                    o = json.load(f)
                os.unlink(dpath)
                o["wall_s"] = time.time() - t0
                L[name] = o
            except Exception as e:  # a leg that fails is reported, not hidden
                L[name] = {"error": repr(e), "stderr_tail": pr.stderr.decode()[-600:] if pr is not None else ""}
            log("[rank 0] leg %s: %.1fs" % (name, time.time() - t0))
        # ---- the DROP-IN shape, measured (VERDICT r5 missing #2): the reference service's threading model -- 64 worker threads,
        # one DecoderItf object each (v2-asr/v2-asr-work-thread.h:66), every utterance fed in chunks of 25 frames through
        # LogLikelihood(frame, index) pulls -- through the C++ mirror's CLI: the objects over ONE 64-channel GpuChannelPool
        # (a batcher thread issues one C-ABI call per kind for whatever requests have arrived), beside the same threads over
        # private 1-channel device decoders (round 5's shape).  PCIe-inclusive by construction.
        try:
            t0 = time.time()
            if only and "dropin_threads64" not in only:
                raise KeyError("skipped (--only-legs)")
            L["dropin_threads64"] = dropin_leg(a, gpath, mats, m, cd, res, out.get("cpu_baseline"))
            log("[rank 0] leg dropin_threads64: %.1fs" % (time.time() - t0))
        except Exception as e:
            L["dropin_threads64"] = {"error": repr(e)}
        # legs that run the CPU's workload of another leg carry that leg's baseline (the reference decodes the same utterances at the same
        # beams: it has no pipelining to switch on, and its biglm search does not depend on lattice_beam)
        for name, sib in (("biglm_lattice_beam14", "biglm"), ("lattice_beam15_no_determinizer", "lattice_beam15"), ("lattice_beam15_pipelined", "lattice_beam15")):
            if name in L and sib in L and "cpu_baseline" not in L[name] and "cpu_baseline" in L[sib]:
                L[name]["cpu_baseline"] = dict(L[sib]["cpu_baseline"], measured_in_leg=sib)
    if rank == 0:
        # the full result to bench_detail.json, its summary -- strict JSON under 4 KB, scalars only -- as the ONE stdout line
        where = write_detail(out, a.detail_out)
        print(summary_line(out, os.path.basename(where) if where else None), flush=True)
    if dec is not None:
        dec.free()
    for L in list(lm_dev) + list(post_lms or []):
        if L is not None:
            L.free()
    graph.free()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
