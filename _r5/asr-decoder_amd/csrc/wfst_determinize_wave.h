// The subset construction of wfst_determinize.h run by ONE WAVE per lattice instead of one lane (device only).
//
// The algorithm, its order and its float arithmetic are those of wfst_determinize.h (the reference's LatticeDeterminizer,
// newfst/lattice-determinize.h:300-1468): output states leave a LIFO queue, an epsilon closure is a FIFO relaxation, subsets
// are matched within delta.  A lattice is ~30 dependent memory round trips per closure element on one lane (rocprofv3 counters on
// the bench's largest lattice: 8 M instructions in 124 M cycles -- the lane waits); this file cuts the round trips and runs the
// independent ones side by side:
//
//  * EpsilonClosure (:842-936), two thirds of them: the closure's element list, its FIFO and its state index live in LDS; the next
//    kDwWin queue entries are PRICED side by side, a lane per (entry, epsilon arc) -- the entry's row read in one go (the count of
//    leading epsilons sits beside the row's offset), the successor string found or made by ONE probe of the trie's table
//    (wfst_determinize.h: a node is its slot; made by a compare-and-swap on the key, so two lanes after the same (parent, label)
//    get the same node), the target's place in the state index looked up -- and then COMMITTED in queue order.  A queue entry's
//    offers depend on nothing but the entry itself (Element copied at push time, :864-865); whether they are made depends on the
//    entry still being its state's best when its turn comes (:874-875), whether one is taken on the target's best at that moment:
//    where no two offers of a window meet in one state and none reaches a state that has an entry in the window (checked through
//    marks in LDS: the rule, not the exception) the order of the commits does not matter and every lane commits its own offer,
//    queue positions by prefix sum; otherwise one lane commits them in the reference's order.  Bit for bit the sequential result;
//    what the lanes do ahead of the order are pure look-ups (an entry that turns out stale leaves at most trie nodes behind).
//  * everything else (ProcessFinal, the transition pairs, NormalizeSubset, the two subset tables) runs between the closures on
//    lane 0 through the functions of wfst_determinize.h.
//
// A closure that outgrows the LDS buffers is run again by det_closure() in the workspace's.
#ifndef WFST_DETERMINIZE_WAVE_H_
#define WFST_DETERMINIZE_WAVE_H_

#include <hip/hip_runtime.h>

#include "wfst_determinize.h"

namespace wfst {

// (the sizes can be shrunk at compile time -- tests/test_gpu_detwave_stress.py builds the harness with tiny ones so that every
// fall-back runs on ordinary lattices: closures that outgrow LDS, entries with more arcs than a window prices, strings longer than a
// lane's label buffer)
#ifndef DETW_CUR
#define DETW_CUR 1024
#endif
#ifndef DETW_ARCS
#define DETW_ARCS 4
#endif
#ifndef DETW_LABS
#define DETW_LABS 128
#endif
constexpr int kDwCur = DETW_CUR;     // closure elements held in LDS (a power of two, at most 1024: the sort keys carry the index in 10 bits)
constexpr int kDwQueue = DETW_CUR;   // FIFO ring
constexpr int kDwMap = 2048;     // state -> element index, open addressing
constexpr int kDwArcs = DETW_ARCS;   // epsilon arcs priced per entry in a window (an entry with more is priced alone, a lane per arc); 1, 2 or 4
constexpr int kDwWin = 64 / kDwArcs;   // queue entries priced side by side

struct DwShared {
  DetElem cur[kDwCur];
  DetElem queue[kDwQueue];
  uint16_t qidx[kDwQueue];       // the queue entry's state: its index in cur[]
  uint32_t map[kDwMap];          // 0 = empty, else (state << 11) | (index + 1)
  uint16_t cur_slot[kDwCur];     // where element i sits in map[] (cleared from here when the closure is done)
  uint16_t mark_idx[kDwCur];     // commit: the window's entry that sits on cur[i] (0xFFFF: none)
  uint16_t mark_best[kDwCur];    // ... the lane whose offer to cur[i] is the best so far; likewise for an empty slot of map[] (a new state)
  uint16_t mark_slot[kDwMap];
  uint32_t sortk[kDwCur];
  DetElem offer[64];
  uint16_t claim[256];           // detw_succ_wave: who takes an empty trie slot
  DetElem pair[64], psort[64];   // an output state's transition pairs (element, label) when there are at most 64: as made, then sorted by (label, state)
  int32_t plabel[64], plsort[64];
  DetElem stage[64];             // a closure's (minimal) result of up to 64 elements, beside its copy in the workspace: what lane 0 normalizes,
  DetElem sub[64];               // hashes and compares next -- and the initial subset it came from -- read at LDS latency
  int32_t bc[16];                // lane 0 -> wave
  long long tm[16];              // (development timers)
};

__host__ __device__ inline int64_t detw_extra_words(const DetCaps &, int32_t) { return 0; }

// compiler-level ordering of LDS / global traffic between the lanes of the one wave (no instruction: a wave's memory operations
// are issued in order)
#define DETW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// global-address-space accesses to the workspace (its pointers are generic: carved at run time)
#define DW_G(T) __attribute__((address_space(1))) T
typedef int dw_v4i __attribute__((ext_vector_type(4), aligned(4)));
__device__ inline int32_t dw_ld(const int32_t *p) { return *(const DW_G(int32_t) *)p; }
__device__ inline void dw_st(int32_t *p, int32_t v) { *(DW_G(int32_t) *)p = v; }
__device__ inline DetElem dw_ld_elem(const DetElem *p) {
  const dw_v4i v = *(const DW_G(dw_v4i) *)p;
  DetElem e;
  e.state = v.x; e.str = v.y; e.w1 = __int_as_float(v.z); e.w2 = __int_as_float(v.w);
  return e;
}
__device__ inline void dw_st_elem(DetElem *p, const DetElem &e) {
  dw_v4i v;
  v.x = e.state; v.y = e.str; v.z = __float_as_int(e.w1); v.w = __float_as_int(e.w2);
  *(DW_G(dw_v4i) *)p = v;
}
__device__ inline DetArc dw_ld_arc(const DetArc *p) {
  const dw_v4i v = *(const DW_G(dw_v4i) *)p;
  DetArc a;
  a.ilabel = v.x; a.olabel = v.y; a.w1 = __int_as_float(v.z); a.w2 = __int_as_float(v.w);
  a.to = *(const DW_G(int32_t) *)(&p->to);
  return a;
}

// Successor (:58-79) for the whole wave at once (every lane calls it; `need`: this lane has a string to extend): plain loads and
// stores like the one-lane det_succ() -- the trie's table is read through the L1 everywhere, an L2 atomic in between would leave
// stale lines there -- with the lanes that find the same empty slot settled through a claim word in LDS: one takes the slot and
// writes the key, the others look again (and find it, if they were after the same string).
__device__ inline int32_t detw_succ_wave(DetWs &W, uint16_t *claim /* [256] */, bool need, int32_t parent, int32_t label, int lane) {
  const uint32_t mask = (uint32_t)W.tr_hcap - 1u;
  const uint64_t key = det_key(parent, label);
  uint32_t s = det_hash2(parent, label) & mask;
  int32_t ans = 0;
  const int32_t pd = need ? dw_ld(W.tr_depth + parent) : 0;
  bool open = need && !W.err;
  int made = 0;
  for (int round = 0; round < (1 << 20); ++round) {
    if (!__ballot(open)) break;
    bool want = false;
    if (open) {
      const uint64_t k = *(const DW_G(uint64_t) *)(W.tr_key + s);
      if (k == key) { ans = (int32_t)s; open = false; }
      else if (k != kDetEmptyKey) s = (s + 1) & mask;
      else { want = true; claim[s & 255u] = (uint16_t)lane; }
    }
    DETW_SYNC();
    if (want && claim[s & 255u] == (uint16_t)lane) {
      *(DW_G(uint64_t) *)(W.tr_key + s) = key;
      dw_st(W.tr_depth + s, pd + 1);
      ans = (int32_t)s;
      open = false;
      ++made;
    }
    DETW_SYNC();
  }
  const unsigned long long mk = __ballot(made != 0);
  if (mk && lane == 0) {   // (the count is lane 0's: nobody else reads it)
    const int n = W.tr_n + __popcll(mk);
    W.tr_n = n;
    if (n >= W.cap.trie || 2 * (int64_t)n >= W.tr_hcap) W.err = 1;   // (the table stays at most half full; 1: the trie)
  }
  return ans;
}

__device__ inline uint32_t detw_mapslot(int32_t state) { return ((uint32_t)state * 2654435761u) >> (32 - 11); }   // kDwMap = 2^11

// index of `state` in cur[], or -1 with the empty slot its probe ended at
__device__ inline int detw_map_find(const DwShared &S, int32_t state, uint32_t *slot_out) {
  uint32_t h = detw_mapslot(state);
  for (;;) {
    const uint32_t v = S.map[h];
    if (v == 0) { *slot_out = h; return -1; }
    if ((int32_t)(v >> 11) == state) { *slot_out = h; return (int)(v & 2047u) - 1; }
    h = (h + 1) & (kDwMap - 1);
  }
}

// bitonic sort of S.sortk[0..n) (ascending), n <= kDwCur, by the wave
__device__ inline void detw_sort_keys(DwShared &S, int n, int lane) {
  int p = 1;
  while (p < n) p <<= 1;
  for (int i = n + lane; i < p; i += 64) S.sortk[i] = 0xFFFFFFFFu;
  DETW_SYNC();
  for (int k = 2; k <= p; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < p / 2; t += 64) {
        const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
        const bool up = ((lo & k) == 0);
        const uint32_t a = S.sortk[lo], b = S.sortk[hi];
        if ((a > b) == up) { S.sortk[lo] = b; S.sortk[hi] = a; }
      }
      DETW_SYNC();
    }
}

__device__ inline int detw_rank(unsigned long long mask, int lane) { return __popcll(mask & ((1ull << lane) - 1ull)); }

// One lane commits the window's offers in the reference's order (entries in queue order, an entry's arcs in row order).
// valid: the lanes that hold an offer; lane = entry * arcs_per_entry + arc.  Returns through S.bc {nc, queue length, overflow}.
__device__ inline void detw_commit_in_order(DetWs &W, DwShared &S, unsigned long long valid, int arcs_per_entry, int n_entries, int qh, int nc, int qn) {
  bool over = false;
  int tail = qn;
  for (int e = 0; e < n_entries && !over; ++e) {
    const unsigned long long m = (valid >> (e * arcs_per_entry)) & ((arcs_per_entry >= 64) ? ~0ull : ((1ull << arcs_per_entry) - 1ull));
    if (!m) continue;
    {  // the entry may have been overtaken by an offer committed since it was priced
      const DetElem el = S.queue[(qh + e) & (kDwQueue - 1)];
      const DetElem c = S.cur[S.qidx[(qh + e) & (kDwQueue - 1)]];
      if (!(c.str == el.str && c.w1 == el.w1 && c.w2 == el.w2)) continue;
    }
    for (unsigned long long mk = m; mk; mk &= mk - 1) {
      const DetElem nx = S.offer[e * arcs_per_entry + __ffsll((long long)mk) - 1];
      uint32_t slot;
      int idx = detw_map_find(S, nx.state, &slot);
      bool push = false;
      if (idx < 0) {
        if (nc >= kDwCur) { over = true; break; }
        idx = nc;
        S.map[slot] = ((uint32_t)nx.state << 11) | (uint32_t)(nc + 1);
        S.cur_slot[nc] = (uint16_t)slot;
        S.cur[nc++] = nx;
        push = true;
      } else {
        const DetElem c = S.cur[idx];
        if (det_cmp(W, nx.w1, nx.w2, nx.str, c.w1, c.w2, c.str) == 1) { S.cur[idx] = nx; push = true; }
      }
      if (push) {
        if (tail >= kDwQueue) { over = true; break; }
        S.queue[(qh + tail) & (kDwQueue - 1)] = nx;
        S.qidx[(qh + tail) & (kDwQueue - 1)] = (uint16_t)idx;
        ++tail;
      }
    }
  }
  S.bc[0] = nc; S.bc[1] = tail; S.bc[2] = over ? 1 : 0;
}

// EpsilonClosure of e[0..n) (global, one element per state) in place, by the wave; returns the new size (sorted by state),
// -1 when the LDS buffers were outgrown (nothing changed then but the trie: the caller runs det_closure()).
// minimal: ConvertToMinimal (:940-957) on the way out -- only the elements whose state has a labelled arc or is final are written.
__device__ inline int detw_closure(DetWs &W, DwShared &S, DetElem *e, int n, int lane, bool minimal) {
  if (n > kDwCur || W.n_states >= (1 << 21)) return -1;
  for (int i = lane; i < n; i += 64) {
    const DetElem x = dw_ld_elem(e + i);
    S.cur[i] = x;
    S.queue[i] = x;
    S.qidx[i] = (uint16_t)i;
    uint32_t h = detw_mapslot(x.state);
    for (;;) {
      if (atomicCAS(&S.map[h], 0u, ((uint32_t)x.state << 11) | (uint32_t)(i + 1)) == 0u) break;
      h = (h + 1) & (kDwMap - 1);
    }
    S.cur_slot[i] = (uint16_t)h;
  }
  DETW_SYNC();
  int nc = n, qh = 0, qn = n;
  bool over = false;
  while (qn > 0 && !over && !W.err) {
    const int win = qn < kDwWin ? qn : kDwWin;
    // ---- price the window: lane = entry * kDwArcs + arc ----------------------------------------------------------
    const int en = lane / kDwArcs, j = lane % kDwArcs;
    bool have = false, big = false;
    DetElem nx;
    nx.state = 0; nx.str = 0; nx.w1 = 0.0f; nx.w2 = 0.0f;
    int tgt = 0;
#ifdef DETW_TIMERS   // (development: tools/det_bench.hip -- the phases of a window, each drained before its clock is read)
#define DWT(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long now_ = clock64(); if (lane == 0) S.tm[k] += now_ - tq; tq = now_; } while (0)
    long long tq = clock64();
#else
#define DWT(k) do { } while (0)
#endif
    DetElem el; el.state = 0; el.str = 0; el.w1 = 0; el.w2 = 0;
    bool live_e = false;
    if (en < win) {
      const int qp = (qh + en) & (kDwQueue - 1);
      el = S.queue[qp];
      const DetElem c = S.cur[S.qidx[qp]];
      live_e = (c.str == el.str && c.w1 == el.w1 && c.w2 == el.w2);
    }
    DWT(0);
    int32_t a0 = 0, ne = 0;
    if (live_e) { a0 = dw_ld(W.off + el.state); ne = dw_ld(W.neps + el.state); }
    DWT(1);
    DetArc arc; arc.ilabel = 0; arc.olabel = 0; arc.w1 = 0; arc.w2 = 0; arc.to = 0;
    big = live_e && ne > kDwArcs;
    const bool mine = live_e && j < ne && !big;
    if (mine) arc = dw_ld_arc(W.arcs + a0 + j);
    DWT(2);
    if (mine && !det_is_zero(arc.w1, arc.w2)) {
      have = true;
      nx.state = arc.to;
      nx.w1 = el.w1 + arc.w1;
      nx.w2 = el.w2 + arc.w2;
      nx.str = el.str;
    }
    {
      const int32_t ns = detw_succ_wave(W, S.claim, have && arc.olabel != 0, el.str, arc.olabel, lane);
      if (have && arc.olabel != 0) nx.str = ns;
    }
    DWT(3);
    if (have) {
      uint32_t slot;
      const int idx = detw_map_find(S, nx.state, &slot);
      tgt = idx >= 0 ? idx : -1 - (int)slot;
      S.offer[lane] = nx;
    }
    DWT(4);
#ifdef DETW_TIMERS
    if (lane == 0) { S.tm[8] += 1; S.tm[9] += win; }
#endif
    const unsigned long long bigmask = __ballot(big);
    int use = win;
    if (bigmask) use = (__ffsll((long long)bigmask) - 1) / kDwArcs;
    if (use == 0) {
      // ---- the head entry has more epsilon arcs than a window prices: a lane per arc, 64 at a time, committed in order ----
      DETW_SYNC();
      const int qp = qh & (kDwQueue - 1);
      const DetElem el = S.queue[qp];
      const int32_t a0 = dw_ld(W.off + el.state), ne = dw_ld(W.neps + el.state);
      for (int32_t base = 0; base < ne && !over; base += 64) {
        bool live = false;
        DetArc arc2; arc2.ilabel = 0; arc2.olabel = 0; arc2.w1 = 0; arc2.w2 = 0; arc2.to = 0;
        if (base + lane < ne) {
          arc2 = dw_ld_arc(W.arcs + a0 + base + lane);
          live = !det_is_zero(arc2.w1, arc2.w2);
        }
        {
          const int32_t ns = detw_succ_wave(W, S.claim, live && arc2.olabel != 0, el.str, arc2.olabel, lane);
          if (live) {
            DetElem y;
            y.state = arc2.to;
            y.w1 = el.w1 + arc2.w1;
            y.w2 = el.w2 + arc2.w2;
            y.str = arc2.olabel != 0 ? ns : el.str;
            S.offer[lane] = y;
          }
        }
        const unsigned long long livemask = __ballot(live);
        DETW_SYNC();
        if (lane == 0) detw_commit_in_order(W, S, livemask, 64, 1, qh, nc, qn);
        DETW_SYNC();
        nc = S.bc[0]; qn = S.bc[1]; over = S.bc[2] != 0;
        DETW_SYNC();
      }
      qh = (qh + 1) & (kDwQueue - 1);
      --qn;
      continue;
    }
    if (en >= use) have = false;
    // ---- commit, every lane its own offer.  What the reference's order decides, and how it is kept:
    //  (1) an offer counts only if it beats the target's best; the best only improves, so an offer that loses against the best of
    //      the window's start loses at its turn too: dropped now;
    //  (2) an entry whose state is improved by an offer of an EARLIER entry of the window is stale at its turn (:874-875): the
    //      window ends in front of the first such entry (its turn comes next window, where the pricing finds it stale), so every
    //      entry committed here is alive at its turn whatever the others do;
    //  (3) offers that meet in one state: the reference takes them in turn and keeps the best; the ones it takes on the way sit in
    //      the queue as stale entries nobody reads -- only the best (the earliest of equals) is committed, at its own place in the
    //      queue's order.
    const int use0 = use;
    const bool have0 = have;   // (as priced)
    if (__ballot(have && tgt >= 0)) {   // (most windows only reach states new to the closure: nothing to look up then)
      if (lane < use0) S.mark_idx[S.qidx[(qh + lane) & (kDwQueue - 1)]] = (uint16_t)lane;   // which entry of the window sits on cur[i]
      DETW_SYNC();
      int kill = use0;
      if (have && tgt >= 0) {
        const DetElem c = S.cur[tgt];
        if (det_cmp(W, nx.w1, nx.w2, nx.str, c.w1, c.w2, c.str) != 1) have = false;   // (1)
        else {
          const int me = S.mark_idx[tgt];
          if (me < use0 && me > en) kill = me;                                            // (2)
        }
      }
      for (int d = 32; d > 0; d >>= 1) kill = min(kill, __shfl_xor(kill, d, 64));
      use = kill;
      if (en >= use) have = false;
      DETW_SYNC();
      if (lane < use0) S.mark_idx[S.qidx[(qh + lane) & (kDwQueue - 1)]] = 0xFFFFu;
    }
    // (3) the champion of every state that offers meet in
    bool mixed = false;
    if (__popcll(__ballot(have)) > 1) {   // (a single offer meets nobody)
      bool cand = have;
      uint16_t *mk = tgt >= 0 ? &S.mark_best[tgt] : &S.mark_slot[-1 - tgt];
      for (int round = 0; round < 64; ++round) {
        if (cand) *mk = (uint16_t)lane;
        DETW_SYNC();
        const int champ = cand ? (int)*mk : lane;
        bool beat = false;
        if (cand && champ != lane) {
          const DetElem o = S.offer[champ];
          if (o.state != nx.state) mixed = true;   // (two NEW states after one empty slot of the index: settled in order below)
          const int cmp = det_cmp(W, nx.w1, nx.w2, nx.str, o.w1, o.w2, o.str);
          beat = cmp == 1 || (cmp == 0 && lane < champ);
          if (!beat) cand = false;
        }
        DETW_SYNC();
        if (beat) *mk = (uint16_t)lane;   // (tells the champion it is beaten)
        DETW_SYNC();
        if (cand && champ == lane && (int)*mk != lane) cand = false;
        const bool settled = !cand || (champ == lane && (int)*mk == lane);
        if (!__ballot(!settled)) break;
      }
      have = cand;
    }
    const unsigned long long valid = __ballot(have);
    if (__ballot(mixed)) {
      // (never seen on a lattice of the bench: two states new to the closure whose probes of the state index end in one slot)
      const unsigned long long all = __ballot(have0 && en < use);   // the window's offers as priced: the in-order commit judges them itself
      DETW_SYNC();
      if (lane == 0) detw_commit_in_order(W, S, all, kDwArcs, use, qh, nc, qn);
      DETW_SYNC();
      nc = S.bc[0]; qn = S.bc[1] - use; over = S.bc[2] != 0;
      qh = (qh + use) & (kDwQueue - 1);
      DETW_SYNC();
      DWT(6);
      continue;
    }
    const bool is_new = have && tgt < 0;
    const unsigned long long newmask = __ballot(is_new);
    const int n_new = __popcll(newmask), n_push = __popcll(valid);
    if (nc + n_new > kDwCur || qn + n_push > kDwQueue) { over = true; break; }
    if (have) {
      int idx = tgt;
      if (is_new) {
        idx = nc + detw_rank(newmask, lane);
        const uint32_t slot = (uint32_t)(-1 - tgt);
        S.map[slot] = ((uint32_t)nx.state << 11) | (uint32_t)(idx + 1);
        S.cur_slot[idx] = (uint16_t)slot;
      }
      S.cur[idx] = nx;
      const int qp = (qh + qn + detw_rank(valid, lane)) & (kDwQueue - 1);
      S.queue[qp] = nx;
      S.qidx[qp] = (uint16_t)idx;
    }
    nc += n_new;
    qn += n_push - use;
    qh = (qh + use) & (kDwQueue - 1);
    DETW_SYNC();
    DWT(5);
  }
  // ---- out: clear the index, sort by state ------------------------------------------------------------------
  for (int i = lane; i < nc; i += 64) {
    S.map[S.cur_slot[i]] = 0u;
    S.sortk[i] = ((uint32_t)S.cur[i].state << 10) | (uint32_t)i;
  }
  DETW_SYNC();
  if (over || W.err) return -1;
  detw_sort_keys(S, nc, lane);
  int nout = 0;
  for (int i0 = 0; i0 < nc; i0 += 64) {
    const int i = i0 + lane;
    DetElem x;
    bool keep = false;
    if (i < nc) {
      x = S.cur[S.sortk[i] & 1023u];
      keep = !minimal || dw_ld(W.osf + x.state) != 0;
    }
    const unsigned long long km = __ballot(keep);
    if (keep) {
      const int o = nout + detw_rank(km, lane);
      dw_st_elem(e + o, x);
      if (o < 64) S.stage[o] = x;
    }
    nout += __popcll(km);
  }
  DETW_SYNC();
  return nout;
}

// the closure by the wave, or -- when it outgrows LDS -- by lane 0 in the workspace's buffers
__device__ inline int detw_closure_any(DetWs &W, DwShared &S, DetElem *e, int n, int lane, bool minimal) {
  int m = detw_closure(W, S, e, n, lane, minimal);
  if (m >= 0) return m;
  if (lane == 0) {
    int k = W.err ? 0 : det_closure(W, e, n);
    if (minimal && !W.err) k = det_minimal(W, e, k);
    for (int i = 0; i < k && i < 64; ++i) S.stage[i] = e[i];   // (as detw_closure leaves it)
    S.bc[3] = k;
  }
  DETW_SYNC();
  return S.bc[3];
}

// ProcessFinal + the transition pairs of output state `out` sorted by (label, state) into W.td / W.ta_label: the first half of
// det_process_state().  Returns the number of pairs.
__device__ inline int32_t detw_pairs(DetWs &W, int32_t out) {
  const int32_t n = W.os_len[out];
  {
    bool is_final = false;
    float f1 = __builtin_huge_valf(), f2 = __builtin_huge_valf();
    int32_t fs = 0;
    for (int32_t i = 0; i < n; ++i) {
      const DetElem el = W.pool[W.os_off[out] + i];
      if (!W.is_final[el.state]) continue;
      if (!is_final || det_cmp(W, el.w1, el.w2, el.str, f1, f2, fs) == 1) { is_final = true; f1 = el.w1; f2 = el.w2; fs = el.str; }
    }
    if (is_final) det_add_arc(W, out, 0, -1, f1, f2);
  }
  int32_t m = 0;
  for (int32_t i = 0; i < n && !W.err; ++i) {
    const DetElem el = W.pool[W.os_off[out] + i];
    for (int32_t a = W.off[el.state] + W.neps[el.state]; a < W.off[el.state + 1]; ++a) {
      const DetArc &arc = W.arcs[a];
      if (arc.ilabel == 0 || det_is_zero(arc.w1, arc.w2)) continue;
      if (m >= W.cap.tmp) { W.err = 6; break; }
      DetElem nx;
      nx.state = arc.to;
      nx.w1 = el.w1 + arc.w1;
      nx.w2 = el.w2 + arc.w2;
      nx.str = arc.olabel == 0 ? el.str : det_succ(W, el.str, arc.olabel);
      W.td[m] = nx;
      W.ta_label[m] = arc.ilabel;
      ++m;
    }
  }
  for (int32_t gap = m > 64 ? 40 : 1; gap >= 1; gap = gap > 1 ? (gap == 40 ? 13 : gap == 13 ? 4 : 1) : 0)
    for (int32_t i = gap; i < m; ++i) {
      const DetElem x = W.td[i];
      const int32_t xl = W.ta_label[i];
      int32_t j = i;
      while (j >= gap && (W.ta_label[j - gap] > xl || (W.ta_label[j - gap] == xl && W.td[j - gap].state > x.state))) {
        W.td[j] = W.td[j - gap]; W.ta_label[j] = W.ta_label[j - gap]; j -= gap;
      }
      W.td[j] = x; W.ta_label[j] = xl;
    }
  return m;
}

// detw_pairs() by the wave, for an output state of at most 64 elements with at most 64 labelled arcs between them (else lane 0
// runs detw_pairs()): a lane per element reads its row, the successor strings are made side by side, the pairs land in LDS and are
// ranked there.  Which of equal (label, state) pairs comes first does not matter: MakeSubsetUnique keeps the better one.
// Returns the number of pairs; *in_lds: they are in S.psort / S.plsort (else in W.td / W.ta_label).
__device__ inline int32_t detw_pairs_wave(DetWs &W, DwShared &S, int32_t out, int lane, bool *in_lds) {
  const int32_t n = dw_ld(W.os_len + out), o0 = dw_ld(W.os_off + out);
  DetElem el;
  el.state = 0; el.str = 0; el.w1 = 0.0f; el.w2 = 0.0f;
  int32_t a_lo = 0, cnt = 0;
  bool fin = false;
  if (n <= 64 && lane < n) {
    el = dw_ld_elem(W.pool + o0 + lane);
    fin = dw_ld(W.is_final + el.state) != 0;
    a_lo = dw_ld(W.off + el.state) + dw_ld(W.neps + el.state);
    cnt = dw_ld(W.off + el.state + 1) - a_lo;
  }
  int32_t tot = cnt, mx = cnt;
  for (int d = 32; d > 0; d >>= 1) { tot += __shfl_xor(tot, d, 64); mx = max(mx, __shfl_xor(mx, d, 64)); }
  if (n > 64 || tot > 64) {
    if (lane == 0) S.bc[5] = detw_pairs(W, out);
    DETW_SYNC();
    *in_lds = false;
    return S.bc[5];
  }
  // ProcessFinal (:1029-1060): the best final element, the first of equals
  unsigned long long fm = __ballot(fin);
  if (fm) {
    int best = __ffsll((long long)fm) - 1;
    float f1 = __shfl(el.w1, best, 64), f2 = __shfl(el.w2, best, 64);
    int32_t fs = __shfl(el.str, best, 64);
    for (unsigned long long mk = fm & (fm - 1); mk; mk &= mk - 1) {
      const int k = __ffsll((long long)mk) - 1;
      const float g1 = __shfl(el.w1, k, 64), g2 = __shfl(el.w2, k, 64);
      const int32_t gs = __shfl(el.str, k, 64);
      if (det_cmp(W, g1, g2, gs, f1, f2, fs) == 1) { f1 = g1; f2 = g2; fs = gs; }
    }
    if (lane == 0) det_add_arc(W, out, 0, -1, f1, f2);
  }
  // the labelled arcs of every element, arc j of each side by side
  int32_t m = 0;
  for (int32_t j = 0; j < mx; ++j) {
    DetArc arc;
    arc.ilabel = 0; arc.olabel = 0; arc.w1 = 0.0f; arc.w2 = 0.0f; arc.to = 0;
    bool live = false;
    if (j < cnt) {
      arc = dw_ld_arc(W.arcs + a_lo + j);
      live = arc.ilabel != 0 && !det_is_zero(arc.w1, arc.w2);
    }
    const int32_t ns = detw_succ_wave(W, S.claim, live && arc.olabel != 0, el.str, arc.olabel, lane);
    const unsigned long long lm = __ballot(live);
    if (live) {
      DetElem nx;
      nx.state = arc.to;
      nx.w1 = el.w1 + arc.w1;
      nx.w2 = el.w2 + arc.w2;
      nx.str = arc.olabel != 0 ? ns : el.str;
      const int pos = m + detw_rank(lm, lane);
      S.pair[pos] = nx;
      S.plabel[pos] = arc.ilabel;
    }
    m += __popcll(lm);
  }
  DETW_SYNC();
  // rank by (label, state), ties by position
  if (lane < m) {
    const int32_t ml = S.plabel[lane], ms = S.pair[lane].state;
    int r = 0;
    for (int k = 0; k < m; ++k) {
      const int32_t kl = S.plabel[k], ks = S.pair[k].state;
      r += (kl < ml || (kl == ml && (ks < ms || (ks == ms && k < lane)))) ? 1 : 0;
    }
    S.psort[r] = S.pair[lane];
    S.plsort[r] = ml;
  }
  DETW_SYNC();
  *in_lds = true;
  return m;
}

// NormalizeSubset (:1219-1252) of e[0..k), k <= 64, by the wave -- a lane per element: the best weight (the first of equals), the
// longest common prefix of the strings (every lane walks its own string up to the shallowest one's depth, then all walk together
// until they stand on one node), the weights divided, the prefix taken off every string (its remaining labels collected on the way
// up, the new string made label by label, all lanes side by side).  The same values as det_normalize(); what was (k - 1) walks one
// after the other is one walk.  Strings with more than kDwLabs labels left: lane 0 runs det_normalize().
constexpr int kDwLabs = DETW_LABS;
__device__ inline void detw_normalize_wave(DetWs &W, DwShared &S, DetElem *e, int k, int lane, float *t1, float *t2, int32_t *common) {
  const float inf = __builtin_huge_valf();
  if (k == 0) { *common = 0; *t1 = inf; *t2 = inf; return; }
  const bool on = lane < k;
  DetElem x;
  x.state = 0; x.str = 0; x.w1 = inf; x.w2 = inf;
  if (on) x = e[lane];
  // Plus (:303-308) over the elements in order: the first of the best
  float b1 = x.w1, b2 = x.w2;
  int bi = on ? lane : 64;
  for (int d = 32; d > 0; d >>= 1) {
    const float o1 = __shfl_xor(b1, d, 64), o2 = __shfl_xor(b2, d, 64);
    const int oi = __shfl_xor(bi, d, 64);
    const int c = (bi >= 64) ? -1 : (oi >= 64) ? 1 : det_wcmp(b1, b2, o1, o2);
    if (c == -1 || (c == 0 && oi < bi)) { b1 = o1; b2 = o2; bi = oi; }
  }
  // the strings' lowest common ancestor
  int32_t node = x.str;
  const int32_t dep0 = on ? dw_ld(W.tr_depth + node) : 0x7FFFFFFF;
  int32_t dmin = dep0;
  for (int d = 32; d > 0; d >>= 1) dmin = min(dmin, __shfl_xor(dmin, d, 64));
  int32_t dep = dep0;
  for (;;) {
    const bool up = on && dep > dmin;
    if (!__ballot(up)) break;
    if (up) { node = (int32_t)(uint32_t)(*(const DW_G(uint64_t) *)(W.tr_key + node)); --dep; }
  }
  int32_t plen = dmin;
  for (;;) {
    const int32_t first = __builtin_amdgcn_readfirstlane(node);   // (lane 0 is on: k >= 1)
    if (!__ballot(on && node != first)) break;
    if (on) node = (int32_t)(uint32_t)(*(const DW_G(uint64_t) *)(W.tr_key + node));
    --plen;
  }
  const int32_t pre = __builtin_amdgcn_readfirstlane(node);
  // the prefix off every string
  int32_t left = on ? dep0 - plen : 0, mx = left;
  for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
  if (plen > 0 && mx > kDwLabs) {   // (a string too long for a lane's label buffer: the sequential way, from the untouched elements)
    if (lane == 0) { det_normalize(W, e, k, t1, t2, common); S.bc[14] = __float_as_int(*t1); S.bc[15] = __float_as_int(*t2); S.bc[3] = *common; }
    DETW_SYNC();
    *t1 = __int_as_float(S.bc[14]); *t2 = __int_as_float(S.bc[15]); *common = S.bc[3];
    return;
  }
  if (on) det_divide(x.w1, x.w2, b1, b2);
  if (plen > 0) {
    int32_t *labs = W.labs + lane * kDwLabs;
    int32_t nd = x.str;
    for (int32_t t = 0; t < mx; ++t)
      if (t < left) {
        const uint64_t kk = *(const DW_G(uint64_t) *)(W.tr_key + nd);
        dw_st(labs + (left - 1 - t), (int32_t)(uint32_t)(kk >> 32));
        nd = (int32_t)(uint32_t)kk;
      }
    int32_t cur = 0;
    for (int32_t j = 0; j < mx; ++j) {
      const bool need = on && j < left;
      const int32_t lab = need ? dw_ld(labs + j) : 0;
      const int32_t r = detw_succ_wave(W, S.claim, need, cur, lab, lane);
      if (need) cur = r;
    }
    x.str = cur;
  }
  if (on) e[lane] = x;
  DETW_SYNC();
  *t1 = b1; *t2 = b2; *common = pre;
}

// The whole construction for one lattice, called by every thread of a workgroup (W carved, its tables cleared by det_init, a
// barrier behind both); wave 0 runs it, the other waves return.  timers (may be null): clock64 sums of lane 0.
__device__ inline int detw_run(DetWs &W, DwShared &S, long long *timers) {
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid >= 64) return 0;
  long long t_clo = 0, t_pairs = 0, t_sub = 0, t_fin = 0, t0 = clock64();
  for (int i = lane; i < kDwCur; i += 64) S.mark_idx[i] = 0xFFFFu;   // (every window puts its entries' marks back)
  if (lane == 0) {
    W.err = 0;
    W.tr_n = 1; W.tr_key[0] = kDetRootKey; W.tr_depth[0] = 0;
    W.pool_n = 0; W.os_n = 0; W.ih_n = 0; W.q_n = 0; W.oa_n = 0;
    W.ta[0].state = 0; W.ta[0].str = 0; W.ta[0].w1 = 0.0f; W.ta[0].w2 = 0.0f;
  }
  DETW_SYNC();
  if (W.n_states > 0) {
    const int m = detw_closure_any(W, S, W.ta, 1, lane, true);
    if (lane == 0) det_minimal_to_state(W, W.ta, m, false);
    DETW_SYNC();
    for (;;) {
      if (W.q_n <= 0 || W.err) break;
      long long c0 = clock64();
      if (lane == 0) S.bc[4] = W.queue[--W.q_n];
      DETW_SYNC();
      const int32_t o = S.bc[4];
      bool pairs_lds = false;
      const int32_t mp = detw_pairs_wave(W, S, o, lane, &pairs_lds);
      const DetElem *pel = pairs_lds ? S.psort : W.td;
      const int32_t *plab = pairs_lds ? S.plsort : W.ta_label;
      t_pairs += clock64() - c0;
      int32_t i = 0;
      while (i < mp && !W.err) {
        c0 = clock64();
        if (lane == 0) {
          const int32_t ilabel = plab[i];
          int32_t run = 0;
          while (i + run < mp && run <= 64 && plab[i + run] == ilabel) ++run;
          DetElem *sub = run <= 64 ? S.sub : W.te;   // (a handful of elements as a rule: kept in LDS)
          S.bc[13] = run <= 64 ? 1 : 0;
          int32_t k = 0;
          // MakeSubsetUnique (:1184-1216): the elements of one state merged, the better (weight, string) kept
          while (i < mp && plab[i] == ilabel) {
            DetElem cur = pel[i];
            ++i;
            while (i < mp && plab[i] == ilabel && pel[i].state == cur.state) {
              const DetElem x = pel[i];
              if (det_cmp(W, x.w1, x.w2, x.str, cur.w1, cur.w2, cur.str) == 1) { cur.w1 = x.w1; cur.w2 = x.w2; cur.str = x.str; }
              ++i;
            }
            sub[k++] = cur;
          }
          S.bc[6] = i; S.bc[8] = k; S.bc[9] = ilabel;
        }
        DETW_SYNC();
        {
          const int32_t k = S.bc[8];
          DetElem *sub = S.bc[13] ? S.sub : W.te;
          float t1, t2;
          int32_t common;
          if (k <= 64) detw_normalize_wave(W, S, sub, k, lane, &t1, &t2, &common);
          else if (lane == 0) det_normalize(W, sub, k, &t1, &t2, &common);
          if (lane == 0) {
            // InitialToStateId, first half: the look-up
            const uint32_t b = det_subset_hash(sub, k) & ((uint32_t)W.ih_hcap - 1u);
            int32_t found = -1;
            for (int32_t q = W.ih_head[b]; q >= 0; q = W.ih_next[q])
              if (det_subset_equal(sub, k, W.pool + W.ih_off[q], W.ih_len[q], W.delta)) { found = q; break; }
            if (found >= 0) {
              det_add_arc(W, o, S.bc[9], W.ih_state[found], t1 + W.ih_w1[found], t2 + W.ih_w2[found]);
            } else {
              if (k > W.cap.tmp) W.err = 6;
              for (int32_t q = 0; q < k && !W.err; ++q) W.ta[q] = sub[q];
            }
            S.bc[7] = found; S.bc[10] = (int32_t)b;
            S.bc[11] = __float_as_int(t1); S.bc[12] = __float_as_int(t2);
          }
        }
        DETW_SYNC();
        i = S.bc[6];
        const int32_t found = S.bc[7], k = S.bc[8];
        t_sub += clock64() - c0;
        if (found < 0 && !W.err) {
          c0 = clock64();
          const int m2 = detw_closure_any(W, S, W.ta, k, lane, true);   // (the closure, already minimal)
          t_clo += clock64() - c0;
          c0 = clock64();
          DetElem *s = m2 <= 64 ? S.stage : W.ta;   // (the closure left its result in both)
          float w1 = 0.0f, w2 = 0.0f;
          int32_t str = 0;
          if (!W.err) {
            if (m2 <= 64) detw_normalize_wave(W, S, s, m2, lane, &w1, &w2, &str);
            else if (lane == 0) det_normalize(W, s, m2, &w1, &w2, &str);
          }
          if (lane == 0 && !W.err) {
            // InitialToStateId, second half
            DetElem *sub = S.bc[13] ? S.sub : W.te;
            const int32_t ans = det_minimal_to_state(W, s, m2, true);
            if (W.ih_n >= W.cap.initials) W.err = 4;
            else {
              const int32_t q = W.ih_n++;
              const uint32_t b = (uint32_t)S.bc[10];
              W.ih_off[q] = det_store(W, sub, k);
              W.ih_len[q] = k;
              W.ih_state[q] = ans; W.ih_w1[q] = w1; W.ih_w2[q] = w2; W.ih_str[q] = str;
              W.ih_next[q] = W.ih_head[b];
              W.ih_head[b] = q;
              det_add_arc(W, o, S.bc[9], ans, __int_as_float(S.bc[11]) + w1, __int_as_float(S.bc[12]) + w2);
            }
          }
          DETW_SYNC();
          t_fin += clock64() - c0;
        }
      }
    }
  }
  DETW_SYNC();
  if (lane == 0 && timers) { timers[0] = clock64() - t0; timers[1] = t_clo; timers[2] = t_pairs; timers[3] = t_sub; timers[4] = t_fin; for (int q = 0; q < 11; ++q) timers[5 + q] = S.tm[q]; }
  return W.err;
}

// (the development harness, tools/det_bench.hip) carve + init + run for one lattice, by a 256-thread workgroup -- as determinize_kernel
// does it: the trie's table first at 16 slots per raw state, the whole table if that is outgrown
__device__ inline void detw_run_block(const int32_t *off, const DetArc *arcs, const int32_t *fin, int32_t n_states, int32_t n_arcs,
                                      int32_t *ws, const DetCaps &caps, DetOutArc *out, int32_t *res, long long *timers, int variant) {
  __shared__ DetWs W;
  __shared__ DwShared S;
  __shared__ int s_err;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) {
    W.n_states = n_states; W.n_arcs = n_arcs; W.off = off; W.arcs = arcs; W.is_final = fin; W.delta = 1.0f / 1024;
    det_carve(W, ws, caps, n_states);
  }
  for (int i = tid; i < kDwMap; i += blockDim.x) S.map[i] = 0u;
  if (tid < 16) S.tm[tid] = 0;
  __syncthreads();
  const int32_t hcap_full = W.tr_hcap;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tid == 0) {
      int32_t h = hcap_full;
      if (attempt == 0) { h = 4096; while (h < 16 * n_states && h < hcap_full) h <<= 1; }
      W.tr_hcap = h < hcap_full ? h : hcap_full;
    }
    __syncthreads();
    det_init(W, tid, blockDim.x);
    __syncthreads();
    {
      const int e = detw_run(W, S, timers);
      if (tid == 0) s_err = e;
    }
    __syncthreads();
    if (!(s_err == 1 && W.tr_hcap < hcap_full)) break;
  }
  if (tid >= 64) return;
  if (lane == 0) { res[0] = W.os_n; res[1] = W.oa_n; res[2] = W.err; res[3] = W.tr_n; }
  if (out) {
    const int32_t na = W.oa_n < caps.arcs ? W.oa_n : caps.arcs;
    for (int i = lane; i < na; i += 64) out[i] = W.oarcs[i];
  }
  (void)variant;
}

}  // namespace wfst
#endif
