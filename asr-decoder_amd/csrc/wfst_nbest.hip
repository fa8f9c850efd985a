// nbest_kernel: the n lowest-cost DISTINCT word sequences of a channel's pruned lattice, with the
// (total, graph) cost of the best path of each -- what the reference's service obtains with
// GetRawLattice -> DeterminizeLatticeWrapper -> NShortestPath -> ConvertNbestToVector ->
// LatticeToVector (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:50-105, newfst/lattice-determinize.h,
// newfst/lattice-to-nbest.cc): determinization in the lattice semiring keeps, for every word
// sequence, its lowest-cost path; n-shortest-paths over the determinized lattice then lists the n
// cheapest word sequences.  The same list falls out of a k-best dynamic program over the
// (acyclic, frame-layered) raw lattice that keeps per lattice state the K cheapest partial paths
// with DISTINCT word histories (identified by a 64-bit hash of the word sequence):
//   * a history that is not among the K cheapest distinct histories of an intermediate state cannot
//     be a prefix of one of the K cheapest distinct complete sequences (K cheaper distinct prefixes
//     extend through the same suffix), so nothing the answer needs is dropped for K >= n;
//   * equal histories are merged keeping the cheaper path = what determinization does.
// One 1024-thread workgroup per channel; one wavefront builds the list of one lattice state from
// the lists of the sources of its incoming arcs.  Frames in ascending order; epsilon arcs inside a
// frame are iterated to their fixpoint (the lattice has no epsilon cycles).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wfst_device.h"

namespace wfst {
namespace {

typedef unsigned long long u64;
constexpr int kNbThreads = 1024, kNbWaves = kNbThreads / 64;

__device__ __forceinline__ uint32_t nb_f2o(float f) {
  uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const uint32_t o = __shfl_xor(v, m, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ u64 wave_min_64(u64 v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const u64 o = __shfl_xor(v, m, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ u64 mix_word(u64 h, int32_t word) {  // history hash, splitmix64 finaliser
  u64 z = (h ^ (u64)(uint32_t)word) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct Cand {
  float tot, lm;
  u64 hash;
  int32_t prev, word;
  bool valid;
};

// Wave-wide: out of the 64 lanes' candidates pick up to K with distinct hashes in increasing
// (tot, hash, lm) order; result r lands in lane r.  Returns the number found (uniform).
__device__ int select_distinct(Cand c, int K, Cand *res) {
  const int lane = threadIdx.x & 63;
  Cand mine;
  mine.valid = false;
  mine.tot = mine.lm = 0.0f; mine.hash = 0; mine.prev = -1; mine.word = 0;
  int found = 0;
  for (; found < K; ++found) {
    const uint32_t t = wave_min_u32(c.valid ? nb_f2o(c.tot) : 0xFFFFFFFFu);
    if (!__ballot(c.valid)) break;
    const bool at = c.valid && nb_f2o(c.tot) == t;
    const u64 h = wave_min_64(at ? c.hash : ~0ull);
    const bool ah = at && c.hash == h;
    const uint32_t l = wave_min_u32(ah ? nb_f2o(c.lm) : 0xFFFFFFFFu);
    const u64 wm = __ballot(ah && nb_f2o(c.lm) == l);
    const int w = __ffsll((long long)wm) - 1;
    Cand b;
    b.tot = __shfl(c.tot, w, 64);
    b.lm = __shfl(c.lm, w, 64);
    b.hash = h;
    b.prev = __shfl(c.prev, w, 64);
    b.word = __shfl(c.word, w, 64);
    b.valid = true;
    if (lane == found) mine = b;
    if (c.valid && c.hash == h) c.valid = false;  // the same word history, more expensive
  }
  *res = mine;
  return found;
}

}  // namespace

__global__ __launch_bounds__(kNbThreads) void nbest_kernel(DecoderDev D, NbestDev N, const int32_t *chans) {
  const int slot = blockIdx.x;
  const int c = chans ? chans[slot] : slot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ChanCtl *ctl = D.ctl + c;
  const int nd = ctl->n_decoded, nt = ctl->lat_toks, na = ctl->lat_arcs;
  const int K = N.K;
  const int4 *toks = D.lat_toks + (size_t)c * D.lat_tok_cap;
  const LatArc *arcs = D.lat_arcs + (size_t)c * D.lat_arc_cap;
  int32_t *state_of = D.remap + (size_t)c * D.arena_cap;  // arena index -> lattice state (scratch of the pruning passes, free between them)
  NbEntry *list = N.list + (size_t)c * N.tok_cap * K;
  int32_t *S = N.scratch + (size_t)c * N.scratch_ints;
  int32_t *off = S;                        // [tok_cap + 1] start of a state's incoming arcs
  int32_t *cur = off + N.tok_cap + 1;      // [tok_cap]     fill cursor
  int32_t *cnt = cur + N.tok_cap;          // [tok_cap]     entries in a state's list
  int32_t *fbeg = cnt + N.tok_cap;         // [max_frames + 2] first state of a frame
  int32_t *fend = fbeg + D.max_frames + 2; // [max_frames + 2]
  int32_t *in_arcs = fend + D.max_frames + 2;  // [arc_cap]
  __shared__ int s_part[kNbThreads];
  __shared__ int s_changed;
  if (tid == 0) N.out_n[slot] = 0;
  __syncthreads();
  if (ctl->error || nt <= 0 || nd <= 0) return;
  if (nt > N.tok_cap || na > N.arc_cap) {
    if (tid == 0) N.out_n[slot] = -1;
    return;
  }
  // ---- index: arena index -> lattice state, frames, incoming-arc lists ------------------------
  for (int f = tid; f <= nd; f += kNbThreads) { fbeg[f] = 0; fend[f] = 0; }
  __syncthreads();
  for (int i = tid; i < nt; i += kNbThreads) {
    const int4 t = toks[i];
    state_of[t.x] = i;
    off[i] = 0; cnt[i] = 0;
    const int f = t.w & 0x3FFFFFFF;
    if (i == 0 || (toks[i - 1].w & 0x3FFFFFFF) != f) fbeg[f] = i;
    if (i == nt - 1 || (toks[i + 1].w & 0x3FFFFFFF) != f) fend[f] = i + 1;
  }
  if (tid == 0) off[nt] = 0;
  __syncthreads();
  for (int a = tid; a < na; a += kNbThreads) atomicAdd(&off[state_of[arcs[a].dst_tok]], 1);
  __syncthreads();
  {  // exclusive scan of off[0..nt) (one contiguous slice per thread)
    const int per = (nt + kNbThreads - 1) / kNbThreads, b = tid * per, e = min(nt, b + per);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += off[i];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int i = 0; i < kNbThreads; ++i) { const int v = s_part[i]; s_part[i] = run; run += v; }
    }
    __syncthreads();
    int run = s_part[tid];
    for (int i = b; i < e; ++i) { const int v = off[i]; off[i] = run; cur[i] = run; run += v; }
    if (tid == 0) off[nt] = na;
  }
  __syncthreads();
  for (int a = tid; a < na; a += kNbThreads) in_arcs[atomicAdd(&cur[state_of[arcs[a].dst_tok]], 1)] = a;
  __syncthreads();
  // ---- the start state --------------------------------------------------------------------
  const int root = state_of[0];  // the root token is arena entry 0; it survives every pruning
  if (tid == 0) {
    NbEntry e;
    e.tot = 0.0f; e.lm = 0.0f; e.hash = 0x243F6A8885A308D3ull; e.prev = -1; e.word = 0;
    list[(size_t)root * K] = e;
    cnt[root] = 1;
  }
  __syncthreads();
  // ---- frames in ascending order ------------------------------------------------------------
  for (int f = 0; f <= nd; ++f) {
    const int b = fbeg[f], e = fend[f];
    for (int round = 0; round < 4096; ++round) {
      if (tid == 0) s_changed = 0;
      __syncthreads();
      for (int t0 = b; t0 < e; t0 += kNbWaves) {
        const int t = t0 + wave;
        Cand res;
        res.valid = false;
        int found = 0;
        const bool work = t < e && t != root;
        if (work) {
          // candidates: every entry of every source list, extended by the arc; 48 new per pass
          const int a0 = off[t], a1 = off[t + 1];
          int ai = a0, ej = 0;  // next (arc, entry) to hand out -- uniform across the wave
          bool more = a0 < a1;
          Cand keep;
          keep.valid = false;
          while (more) {
            Cand cnd = keep;       // lanes 0..K-1 carry the best so far
            if (lane >= K) cnd.valid = false;
            // hand (arc, entry) pairs to lanes K..63 in order
            int want = lane - K, my_a = -1, my_e = 0;
            int wa = ai, we = ej, given = 0;
            // walk the arcs (uniform loop): arc wa contributes cnt[src] - we entries
            while (wa < a1 && given < 64 - K) {
              const LatArc A = arcs[in_arcs[wa]];
              const int src = state_of[A.src_tok];
              const int have = cnt[src] - we;
              const int take = min(have, 64 - K - given);
              if (want >= given && want < given + take) { my_a = wa; my_e = we + (want - given); }
              given += take;
              if (take == have) { ++wa; we = 0; } else { we += take; }
            }
            ai = wa; ej = we;
            more = wa < a1;
            if (my_a >= 0) {
              const LatArc A = arcs[in_arcs[my_a]];
              const int src = state_of[A.src_tok];
              const NbEntry E = list[(size_t)src * K + my_e];
              cnd.tot = E.tot + (A.graph + A.acoustic);  // LatticeToVector: tot += graph + acoustic
              cnd.lm = E.lm + A.graph;                   //                  lm  += graph
              cnd.hash = A.olabel ? mix_word(E.hash, A.olabel) : E.hash;
              cnd.prev = src * 16 + my_e;
              cnd.word = A.olabel;
              cnd.valid = true;
            }
            found = select_distinct(cnd, K, &keep);
          }
          res = keep;
        }
        // compare with what the state holds, then (all reads of this batch done) replace it
        bool diff = false;
        if (work) {
          if (found != cnt[t]) diff = true;
          if (lane < found) {
            const NbEntry O = list[(size_t)t * K + lane];
            // the backpointer too: an epsilon-source state of this frame may have re-ordered its list
            // since the last round without changing any cost here
            if (lane >= cnt[t] || __float_as_uint(O.tot) != __float_as_uint(res.tot) || O.hash != res.hash ||
                __float_as_uint(O.lm) != __float_as_uint(res.lm) || O.prev != res.prev || O.word != res.word)
              diff = true;
          }
          diff = __ballot(diff) != 0;
        }
        __syncthreads();
        if (work && diff) {
          if (lane < found) {
            NbEntry o;
            o.tot = res.tot; o.lm = res.lm; o.hash = res.hash; o.prev = res.prev; o.word = res.word;
            list[(size_t)t * K + lane] = o;
          }
          if (lane == 0) { cnt[t] = found; s_changed = 1; }
        }
        __syncthreads();
      }
      const int ch = s_changed;
      __syncthreads();
      if (!ch) break;
    }
  }
  // ---- the final states: merge their lists, trace the paths back ------------------------------
  if (wave == 0) {
    const int b = fbeg[nd], e = fend[nd], n = min(N.n, K);
    Cand keep;
    keep.valid = false;
    int found = 0;
    int t = b, ej = 0;
    bool more = true;
    while (more) {
      Cand cnd = keep;
      if (lane >= K) cnd.valid = false;
      int want = lane - K, my_t = -1, my_e = 0, given = 0;
      while (t < e && given < 64 - K) {
        const bool fin = (toks[t].w >> 30) & 1;
        const int have = fin ? cnt[t] - ej : 0;
        const int take = min(have, 64 - K - given);
        if (want >= given && want < given + take) { my_t = t; my_e = ej + (want - given); }
        given += take;
        if (take == have) { ++t; ej = 0; } else { ej += take; }
      }
      more = t < e;
      if (my_t >= 0) {
        const NbEntry E = list[(size_t)my_t * K + my_e];
        cnd.tot = E.tot; cnd.lm = E.lm; cnd.hash = E.hash; cnd.prev = my_t * 16 + my_e; cnd.word = 0;
        cnd.valid = true;
      }
      found = select_distinct(cnd, K, &keep);
    }
    found = min(found, n);
    if (lane == 0) N.out_n[slot] = found;
    if (lane < found) {
      const size_t o = (size_t)slot * N.n + lane;
      N.out_tot[o] = keep.tot;
      N.out_lm[o] = keep.lm;
      // pass 1: count the words; pass 2: write them back to front
      int nw = 0;
      for (int p = keep.prev; p >= 0;) {
        const NbEntry E = list[(size_t)(p >> 4) * K + (p & 15)];
        nw += E.word != 0;
        p = E.prev;
      }
      N.out_nwords[o] = nw;
      int32_t *w = N.out_words + o * N.max_words;
      int k = nw;
      for (int p = keep.prev; p >= 0;) {
        const NbEntry E = list[(size_t)(p >> 4) * K + (p & 15)];
        if (E.word != 0) { --k; if (k < N.max_words) w[k] = E.word; }
        p = E.prev;
      }
    }
  }
}

void launch_nbest(const DecoderDev &D, const NbestDev &N, const int32_t *chans, int cnt, hipStream_t s) {
  hipLaunchKernelGGL(nbest_kernel, dim3(cnt), dim3(kNbThreads), 0, s, D, N, chans);
}

}  // namespace wfst
