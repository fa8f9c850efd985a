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
  // [arc_cap] the incoming arcs of a state, side by side, each with what the list building needs of it: {source state, word,
  // graph, acoustic} -- one load where the arc index, the arc and the source token's state were three dependent ones (round 5)
  int4 *in_rec = reinterpret_cast<int4 *>(S);   // (first in the block, which is a multiple of 16 bytes: aligned)
  int32_t *off = S + 4 * (size_t)N.arc_cap; // [tok_cap + 1] start of a state's incoming arcs
  int32_t *cur = off + N.tok_cap + 1;      // [tok_cap]     fill cursor
  int32_t *cnt = cur + N.tok_cap;          // [tok_cap]     entries in a state's list
  int32_t *fbeg = cnt + N.tok_cap;         // [max_frames + 2] first state of a frame
  int32_t *fend = fbeg + D.max_frames + 2; // [max_frames + 2]
  int32_t *feps = fend + D.max_frames + 2; // [max_frames + 2] the frame has arcs between its own states (epsilon arcs)
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
  for (int f = tid; f <= nd; f += kNbThreads) { fbeg[f] = 0; fend[f] = 0; feps[f] = 0; }
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
  for (int a = tid; a < na; a += kNbThreads) {
    const LatArc A = arcs[a];
    atomicAdd(&off[state_of[A.dst_tok]], 1);
    if (A.is_eps) feps[A.src_frame] = 1;   // (an epsilon arc stays inside its frame)
  }
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
  for (int a = tid; a < na; a += kNbThreads) {
    const LatArc A = arcs[a];
    in_rec[atomicAdd(&cur[state_of[A.dst_tok]], 1)] = make_int4(state_of[A.src_tok], A.olabel, __float_as_int(A.graph), __float_as_int(A.acoustic));
  }
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
              const int src = in_rec[wa].x;
              const int have = cnt[src] - we;
              const int take = min(have, 64 - K - given);
              if (want >= given && want < given + take) { my_a = wa; my_e = we + (want - given); }
              given += take;
              if (take == have) { ++wa; we = 0; } else { we += take; }
            }
            ai = wa; ej = we;
            more = wa < a1;
            if (my_a >= 0) {
              const int4 R = in_rec[my_a];
              const int src = R.x;
              const float a_graph = __int_as_float(R.z), a_ac = __int_as_float(R.w);
              const NbEntry E = list[(size_t)src * K + my_e];
              cnd.tot = E.tot + (a_graph + a_ac);        // LatticeToVector: tot += graph + acoustic
              cnd.lm = E.lm + a_graph;                   //                  lm  += graph
              cnd.hash = R.y ? mix_word(E.hash, R.y) : E.hash;
              cnd.prev = src * 16 + my_e;
              cnd.word = R.y;
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
      // (a frame without arcs between its own states -- five in six -- is final after one round: its lists depend on earlier frames only)
      if (!ch || !feps[f]) break;
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
      // ONE walk down the path (a few hundred dependent hops): the words come out last to first, then the short list is turned round
      int32_t *w = N.out_words + o * N.max_words;
      int nw = 0;
      for (int p = keep.prev; p >= 0;) {
        const NbEntry E = list[(size_t)(p >> 4) * K + (p & 15)];
        if (E.word != 0) { if (nw < N.max_words) w[nw] = E.word; ++nw; }
        p = E.prev;
      }
      N.out_nwords[o] = nw;
      if (nw <= N.max_words) {
        for (int i = 0, j = nw - 1; i < j; ++i, --j) { const int32_t x = w[i]; w[i] = w[j]; w[j] = x; }
      } else {
        // (more words than the caller's buffer takes: it wants the FIRST max_words -- the walk met them last; once more, keeping those)
        int k = nw;
        for (int p = keep.prev; p >= 0;) {
          const NbEntry E = list[(size_t)(p >> 4) * K + (p & 15)];
          if (E.word != 0) { --k; if (k < N.max_words) w[k] = E.word; }
          p = E.prev;
        }
      }
    }
  }
}

// ---- nbest_paths_kernel: NShortestPath on a determinized (or rescored) lattice ---------------------------------------------
// The service's GetNbest (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:97-105) runs NShortestPath (newfst/lattice-to-nbest.cc:15-147)
// on the lattice GetLattice returns and hands every path out as a linear lattice with the lattice's own arcs on it
// (ConvertNbestToVector, :149-199).  Here: the same n cheapest paths -- cost of a path = its arc costs (graph + acoustic) added
// front to back in float, as NShortestPath's forward weights are -- by a k-best dynamic program over the (acyclic) lattice in
// topological order: the list of a state = the n cheapest of its predecessors' lists extended by the arcs between them.  Any n up
// to half the sort buffer (4096); paths come out as sequences of arc indices of the input lattice.  One workgroup per lattice:
// the candidate lists of a state are merged by a bitonic sort in LDS (chunked when they exceed the buffer: the n kept so far
// plus the next candidates), equal costs in the order (arc, rank).
constexpr int kNpThreads = 1024, kNpSlots = 8192;

__device__ __forceinline__ void np_sort(u64 *key, u64 *pay, int S2) {   // ascending bitonic sort of key[0..S2) (S2 a power of two), pay along
  const int tid = threadIdx.x;
  for (int k = 2; k <= S2; k <<= 1)
    for (int j = k >> 1; j >= 1; j >>= 1) {
      for (int i = tid; i < S2; i += kNpThreads) {
        const int l = i ^ j;
        if (l > i) {
          const u64 a = key[i], b = key[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            key[i] = b; key[l] = a;
            const u64 pa = pay[i]; pay[i] = pay[l]; pay[l] = pa;
          }
        }
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(kNpThreads) void nbest_paths_kernel(NbPathsDev P) {
  __shared__ u64 s_key[kNpSlots];
  __shared__ u64 s_pay[kNpSlots];
  __shared__ int s_flag, s_maxlevel, s_lnext, s_total;
  const int tid = threadIdx.x;
  {
    // one workgroup per lattice (a batch of GetNbest requests is one launch): workgroup b takes slot b of every buffer
    const size_t b = blockIdx.x;
    P.a += b * (size_t)P.in_stride; P.w += b * (size_t)P.in_stride; P.res += 4 * b;
    if (P.fin) P.fin += b * (size_t)P.fin_stride;
    P.ws += b * (size_t)P.ws_ints; P.lists += b * (size_t)P.list_cap;
    P.out += 4 * b; P.out_off += b * (size_t)(P.n + 1); P.out_tot += b * (size_t)P.n; P.out_arcs += b * (size_t)P.out_cap;
  }
  int32_t *out = P.out;
  if (tid == 0) { out[0] = 0; out[1] = 0; out[2] = 0; out[3] = 0; s_maxlevel = 0; s_lnext = 0; }
  __syncthreads();
  const int ns = P.res[0], na = P.res[1], n = P.n;
  if (P.res[2] != 0 || ns <= 0) {
    if (tid == 0) out[2] = P.res[2] != 0 ? 2 : 0;
    return;
  }
  const int nmax = ns > na ? ns : na;
  if (7ll * ns + 4ll * nmax + na + 16 > P.ws_ints || 2 * n > kNpSlots) {
    if (tid == 0) out[2] = 1;
    return;
  }
  int32_t *in_off = P.ws;            // [ns + 1]
  int32_t *in_fill = in_off + ns + 1;  // [ns]
  int32_t *level = in_fill + ns;     // [ns]
  int32_t *order = level + ns;       // [ns]
  int32_t *lvl_off = order + ns;     // [ns + 2]
  int32_t *cnt = lvl_off + ns + 2;   // [ns]
  int32_t *loff = cnt + ns;          // [ns]
  int32_t *in_arcs = loff + ns;      // [na]
  int32_t *pre = in_arcs + na;       // [nmax + 1]
  int32_t *csrc = pre + nmax + 1;    // [nmax]
  int32_t *carc = csrc + nmax;       // [nmax]
  const int4 *A = P.a;
  const float2 *Wt = P.w;
  // ---- incoming arcs by target state -------------------------------------------------------------------------------
  for (int s = tid; s <= ns; s += kNpThreads) { in_off[s] = 0; lvl_off[s] = 0; }
  if (tid == 0) lvl_off[ns + 1] = 0;
  for (int s = tid; s < ns; s += kNpThreads) { level[s] = -1; cnt[s] = 0; loff[s] = 0; }
  __syncthreads();
  for (int a = tid; a < na; a += kNpThreads) atomicAdd(&in_off[A[a].y + 1], 1);
  __syncthreads();
  if (tid == 0) {
    for (int s = 0; s < ns; ++s) in_off[s + 1] += in_off[s];
    level[0] = 0;
  }
  __syncthreads();
  for (int s = tid; s < ns; s += kNpThreads) in_fill[s] = in_off[s];
  __syncthreads();
  for (int a = tid; a < na; a += kNpThreads) in_arcs[atomicAdd(&in_fill[A[a].y], 1)] = a;
  __syncthreads();
  // (a state's incoming arcs in ascending arc order: the tie order must not depend on the atomics)
  for (int s = tid; s < ns; s += kNpThreads) {
    const int b = in_off[s], e = in_off[s + 1];
    for (int i = b + 1; i < e; ++i) {
      const int x = in_arcs[i];
      int j = i;
      while (j > b && in_arcs[j - 1] > x) { in_arcs[j] = in_arcs[j - 1]; --j; }
      in_arcs[j] = x;
    }
  }
  // ---- topological levels: the longest distance from the start state (state 0) -----------------------------------------
  for (int it = 0;; ++it) {
    if (tid == 0) s_flag = 0;
    __syncthreads();
    for (int a = tid; a < na; a += kNpThreads) {
      const int4 t = A[a];
      const int ls = level[t.x];
      if (ls >= 0 && level[t.y] < ls + 1) { atomicMax(&level[t.y], ls + 1); s_flag = 1; }
    }
    __syncthreads();
    const int f = s_flag;
    __syncthreads();
    if (!f) break;
    if (it > ns) {   // a cycle: not a lattice
      if (tid == 0) out[2] = 3;
      return;
    }
  }
  for (int s = tid; s < ns; s += kNpThreads)
    if (level[s] >= 0) { atomicAdd(&lvl_off[level[s] + 1], 1); atomicMax(&s_maxlevel, level[s]); }
  __syncthreads();
  if (tid == 0)
    for (int l = 0; l <= s_maxlevel; ++l) lvl_off[l + 1] += lvl_off[l];
  __syncthreads();
  const int n_reached = lvl_off[s_maxlevel + 1];
  for (int s = tid; s < ns; s += kNpThreads) in_fill[s] = 0;
  __syncthreads();
  for (int s = tid; s < ns; s += kNpThreads)
    if (level[s] >= 0) order[lvl_off[level[s]] + atomicAdd(&in_fill[level[s]], 1)] = s;
  __syncthreads();
  // ---- the lists, level by level ----------------------------------------------------------------------------------------
  if (tid == 0) {
    NbPathEntry e0;
    e0.cost = 0.0f; e0.arc = -1; e0.rank = 0; e0.pad = 0;
    if (P.list_cap > 0) P.lists[0] = e0;
    cnt[0] = 1; loff[0] = 0; s_lnext = 1;
  }
  __syncthreads();
  // idx < n_reached: state order[idx]; idx == n_reached: the super-final state (AddSuperFinalState, lattice-functions.cc:163-178)
  for (int idx = 0; idx <= n_reached; ++idx) {
    const bool super = idx == n_reached;
    const int t = super ? -1 : order[idx];
    if (t == 0) continue;   // the start state: the empty path only
    int m;
    if (!super) {
      m = in_off[t + 1] - in_off[t];
      for (int j = tid; j < m; j += kNpThreads) {
        const int arc = in_arcs[in_off[t] + j];
        csrc[j] = A[arc].x;
        carc[j] = arc;
      }
    } else {
      // the final states, in state order
      if (tid == 0) {
        int k = 0;
        for (int s = 0; s < ns; ++s) {
          const bool fin = P.fin ? P.fin[s] != 0 : s >= P.res[3];
          if (fin && level[s] >= 0 && cnt[s] > 0) { csrc[k] = s; carc[k] = -(s + 2); ++k; }
        }
        s_total = k;
      }
      __syncthreads();
      m = s_total;
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int j = 0; j < m; ++j) { pre[j] = run; run += cnt[csrc[j]]; }
      pre[m] = run;
      s_total = run;
    }
    __syncthreads();
    const int C = s_total;
    int kept = 0;
    for (int qbase = 0; qbase < C;) {
      const int take = min(C - qbase, kNpSlots - kept);
      for (int i = tid; i < take; i += kNpThreads) {
        const int q = qbase + i;
        int lo = 0, hi = m;   // pre[lo] <= q < pre[hi]
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (pre[mid] <= q) lo = mid; else hi = mid;
        }
        const int r = q - pre[lo], src = csrc[lo], arc = carc[lo];
        const NbPathEntry E = P.lists[(size_t)loff[src] + r];
        float add = 0.0f;
        if (arc >= 0) { const float2 w = Wt[arc]; add = w.x + w.y; }   // LatticeWeight::Value() (weigth.h:200)
        const float cost = E.cost + add;                               // NShortestPath: p.second + arc->_w.Value() (:121)
        s_key[kept + i] = ((u64)nb_f2o(cost) << 32) | (u64)(uint32_t)q;
        s_pay[kept + i] = ((u64)(uint32_t)arc << 32) | (u64)(uint32_t)r;
      }
      const int filled = kept + take;
      int S2 = 64;
      while (S2 < filled) S2 <<= 1;
      for (int i = filled + tid; i < S2; i += kNpThreads) { s_key[i] = ~0ull; s_pay[i] = 0; }
      __syncthreads();
      np_sort(s_key, s_pay, S2);
      kept = min(n, filled);
      qbase += take;
      // (the entries kept keep their keys: their candidate numbers are below every later one's)
    }
    __syncthreads();
    if (!super) {
      if (tid == 0) {
        loff[t] = s_lnext;
        cnt[t] = kept;
        s_lnext += kept;
      }
      __syncthreads();
      if ((int64_t)s_lnext > P.list_cap) {
        if (tid == 0) out[2] = 1;
        return;
      }
      for (int i = tid; i < kept; i += kNpThreads) {
        NbPathEntry e;
        e.cost = __uint_as_float(0);
        const uint32_t o = (uint32_t)(s_key[i] >> 32);
        const uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);   // inverse of nb_f2o
        e.cost = __uint_as_float(u);
        e.arc = (int32_t)(uint32_t)(s_pay[i] >> 32);
        e.rank = (int32_t)(uint32_t)(s_pay[i] & 0xFFFFFFFFu);
        e.pad = 0;
        P.lists[(size_t)loff[t] + i] = e;
      }
      __syncthreads();
    } else {
      // ---- the paths: backtrack each from its final state, write its arcs front to back -----------------------------------
      const int found = kept;
      for (int p = tid; p < found; p += kNpThreads) {
        int state = -((int32_t)(uint32_t)(s_pay[p] >> 32)) - 2, rank = (int32_t)(uint32_t)(s_pay[p] & 0xFFFFFFFFu), len = 0;
        for (;;) {
          const NbPathEntry E = P.lists[(size_t)loff[state] + rank];
          if (E.arc < 0) break;
          ++len;
          state = A[E.arc].x;
          rank = E.rank;
        }
        P.out_off[p + 1] = len;
        const uint32_t o = (uint32_t)(s_key[p] >> 32);
        P.out_tot[p] = __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu));
      }
      __syncthreads();
      if (tid == 0) {
        P.out_off[0] = 0;
        for (int p = 0; p < found; ++p) P.out_off[p + 1] += P.out_off[p];
        out[0] = found;
        out[1] = P.out_off[found];
        if (P.out_off[found] > P.out_cap) out[2] = 1;
      }
      __syncthreads();
      if (out[2] == 0)
        for (int p = tid; p < found; p += kNpThreads) {
          int state = -((int32_t)(uint32_t)(s_pay[p] >> 32)) - 2, rank = (int32_t)(uint32_t)(s_pay[p] & 0xFFFFFFFFu);
          int k = P.out_off[p + 1];
          for (;;) {
            const NbPathEntry E = P.lists[(size_t)loff[state] + rank];
            if (E.arc < 0) break;
            P.out_arcs[--k] = E.arc;
            state = A[E.arc].x;
            rank = E.rank;
          }
        }
    }
  }
}

void launch_nbest_paths(const NbPathsDev &P, int n_slots, hipStream_t s) { hipLaunchKernelGGL(nbest_paths_kernel, dim3(n_slots), dim3(kNpThreads), 0, s, P); }

void launch_nbest(const DecoderDev &D, const NbestDev &N, const int32_t *chans, int cnt, hipStream_t s) {
  hipLaunchKernelGGL(nbest_kernel, dim3(cnt), dim3(kNbThreads), 0, s, D, N, chans);
}

}  // namespace wfst
