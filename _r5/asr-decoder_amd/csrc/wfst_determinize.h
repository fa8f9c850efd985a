// Lattice determinization in the (graph, acoustic) lattice semiring -- the reference's
// LatticeDeterminizer (newfst/lattice-determinize.h:300-1468, Kaldi's DeterminizeLattice) as its wrapper
// uses it (newfst/lattice-determinize-api.cc:5-21): Invert (words become the input labels, transition-ids
// the output labels), ArcSort, Determinize, OutputNoolabel (the transition-id strings are dropped),
// Invert.  The result is the word-level deterministic lattice the service rescoring and its n-best
// read (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:58-89).
//
// One lattice is ONE sequential subset construction: which of two determinized states that agree within
// delta = 1/1024 becomes the representative depends on the order they are created in (MinimalToStateId /
// InitialToStateId match weights approximately, :1254-1322), so the reference's LIFO order is kept and
// the parallelism is across the lattices of a batch (one workgroup each; the CSR build is workgroup-wide,
// the construction runs on one lane).  Everything lives in flat int32 / float arrays of a per-lattice
// workspace: a trie for the strings (LatticeStringRepository, :35-298: a string is a node, the common
// prefix of two strings their lowest common ancestor), an element pool for the subsets, two chained hash
// tables (minimal and initial subsets), the LIFO queue, the output arcs.
//
// This header compiles for the device (hipcc) and for the host (g++): the host build exists ONLY for
// tests/test_determinize_host.py, which holds the code against the reference's own determinizer
// (oracle/_ref) lattice by lattice; the product runs the device build (determinize_kernel).
#ifndef WFST_DETERMINIZE_H_
#define WFST_DETERMINIZE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define WFST_HD __host__ __device__
#else
#define WFST_HD
#endif

namespace wfst {

struct DetArc { int32_t ilabel, olabel; float w1, w2; int32_t to; };   // inverted: ilabel = word, olabel = transition-id
struct DetElem { int32_t state, str; float w1, w2; };                    // Element (:602-619): str = trie node, 0 = empty
struct DetOutArc { int32_t src, ilabel, next; float w1, w2; };           // TempArc (:622-628) without its string; next -1 = final weight

struct DetCaps {
  int32_t trie, pool, states, initials, arcs, tmp;  // nodes, elements, output states, initial subsets, output arcs, scratch elements
};

struct DetWs {
  // input lattice (CSR, arcs of a state sorted by ilabel: epsilons first)
  int32_t n_states, n_arcs;
  const int32_t *off;
  const DetArc *arcs;
  const int32_t *is_final;
  int32_t *osf;                 // [n_states] IsIsymbolOrFinal (:999-1024)
  int32_t *neps;                // [n_states] leading arcs with input label 0 (the row is sorted: epsilons first)
  // string trie: one open-addressed table, a node IS its slot -- key {parent (low), label (high)}, depth beside it; slot 0 is the
  // empty string.  (One probe finds or makes a successor; round 4 kept node arrays beside a table of node ids: two dependent
  // loads to find one, five memory round trips to make one -- the construction is bound by exactly those.)
  uint64_t *tr_key;
  int32_t *tr_depth;
  int32_t tr_n, tr_hcap;        // nodes in use (the table is kept at most half full), slots in use
  // element pool; output states (minimal subsets) and initial subsets are slices of it
  DetElem *pool;
  int32_t pool_n;
  int32_t *os_off, *os_len, *os_next, *mh_head;   // output state -> slice, hash chain; bucket heads
  int32_t os_n, mh_cap;
  int32_t *ih_off, *ih_len, *ih_next, *ih_state, *ih_str, *ih_head;
  float *ih_w1, *ih_w2;
  int32_t ih_n, ih_hcap;
  int32_t *queue;
  int32_t q_n;
  DetOutArc *oarcs;
  int32_t oa_n;
  // scratch
  DetElem *ta, *tb, *tc, *td, *te;  // [tmp] each
  DetElem *tb_lo, *tc_lo;           // [tmp_lo] each, or null: a faster home (the device: LDS) for the closure's queue and element
  int32_t tmp_lo;                   // list -- what the one lane stores and reads back right away; a closure that outgrows it is run again in tb / tc
  int32_t *ta_label;            // [tmp] labels beside ta
  int32_t *cl_idx;              // [n_states] closure: state -> index in tc, -1
  int32_t *labs;                // [tmp] label sequences
  DetCaps cap;
  float delta;
  int32_t err;                  // a capacity was exceeded: 1 trie, 2 pool, 3 states, 4 initial subsets, 5 arcs, 6 scratch
};

// int32 words a workspace needs (the caller carves them with det_carve)
WFST_HD inline int64_t det_words(const DetCaps &c, int32_t n_states) {
  int64_t w = 0;
  w += 2 * (int64_t)n_states;           // osf, neps
  w += 6 * (int64_t)c.trie + 2;         // the trie's table: 2 x trie slots (rounded down to a power of two by det_carve) of key (8 bytes) + depth
  w += 4 * (int64_t)c.pool;             // pool
  w += 3 * (int64_t)c.states + 2 * (int64_t)c.states;  // os_off/len/next + mh_head
  w += 7 * (int64_t)c.initials + 2 * (int64_t)c.initials;
  w += c.states;                        // queue
  w += 5 * (int64_t)c.arcs;             // oarcs
  w += 5 * 4 * (int64_t)c.tmp + 2 * (int64_t)c.tmp;    // ta tb tc td te, ta_label, labs
  w += n_states;                        // cl_idx
  return w + 64;
}

WFST_HD inline int32_t det_pow2_le(int64_t x) {
  int32_t p = 1;
  while ((int64_t)p * 2 <= x) p *= 2;
  return p;
}

WFST_HD inline void det_carve(DetWs &W, int32_t *base, const DetCaps &c, int32_t n_states) {
  int32_t *p = base;
  W.cap = c;
  W.tb_lo = nullptr; W.tc_lo = nullptr; W.tmp_lo = 0;   // (the caller may set them after det_carve)
  W.osf = p; p += n_states;
  W.neps = p; p += n_states;
  W.tr_hcap = det_pow2_le(2 * (int64_t)c.trie);
  p += ((uintptr_t)p & 7) ? 1 : 0;      // (8-byte keys)
  W.tr_key = reinterpret_cast<uint64_t *>(p); p += 4 * (int64_t)c.trie;
  W.tr_depth = p; p += 2 * (int64_t)c.trie;
  W.pool = reinterpret_cast<DetElem *>(p); p += 4 * (int64_t)c.pool;
  W.os_off = p; p += c.states;
  W.os_len = p; p += c.states;
  W.os_next = p; p += c.states;
  W.mh_cap = det_pow2_le(2 * (int64_t)c.states);
  W.mh_head = p; p += 2 * (int64_t)c.states;
  W.ih_off = p; p += c.initials;
  W.ih_len = p; p += c.initials;
  W.ih_next = p; p += c.initials;
  W.ih_state = p; p += c.initials;
  W.ih_str = p; p += c.initials;
  W.ih_w1 = reinterpret_cast<float *>(p); p += c.initials;
  W.ih_w2 = reinterpret_cast<float *>(p); p += c.initials;
  W.ih_hcap = det_pow2_le(2 * (int64_t)c.initials);
  W.ih_head = p; p += 2 * (int64_t)c.initials;
  W.queue = p; p += c.states;
  W.oarcs = reinterpret_cast<DetOutArc *>(p); p += 5 * (int64_t)c.arcs;
  W.ta = reinterpret_cast<DetElem *>(p); p += 4 * (int64_t)c.tmp;
  W.tb = reinterpret_cast<DetElem *>(p); p += 4 * (int64_t)c.tmp;
  W.tc = reinterpret_cast<DetElem *>(p); p += 4 * (int64_t)c.tmp;
  W.td = reinterpret_cast<DetElem *>(p); p += 4 * (int64_t)c.tmp;
  W.te = reinterpret_cast<DetElem *>(p); p += 4 * (int64_t)c.tmp;
  W.ta_label = p; p += c.tmp;
  W.labs = p; p += c.tmp;
  W.cl_idx = p; p += n_states;
}

// ---- weights: newfst/weigth.h:192-357 -------------------------------------------------------------
WFST_HD inline bool det_is_zero(float a, float b) { const float inf = __builtin_huge_valf(); return a == inf && b == inf; }
// LatticeWeightCompare (:283-300): +1 = the first is better (cheaper)
WFST_HD inline int det_wcmp(float a1, float a2, float b1, float b2) {
  const float f1 = a1 + a2, f2 = b1 + b2;
  if (f1 < f2) return 1;
  if (f1 > f2) return -1;
  if (a1 < b1) return 1;
  if (a1 > b1) return -1;
  return 0;
}
// ApproxEqual (:345-354)
WFST_HD inline bool det_approx(float a1, float a2, float b1, float b2, float delta) {
  if (a1 == b1 && a2 == b2) return true;
  const float d = (a1 + a2) - (b1 + b2);
  return (d < 0 ? -d : d) <= delta;
}
// Divide (:318-337)
WFST_HD inline void det_divide(float &a1, float &a2, float b1, float b2) {
  const float inf = __builtin_huge_valf();
  const float a = a1 - b1, b = a2 - b2;
  if (a != a || b != b || a == -inf || b == -inf || a == inf || b == inf) { a1 = inf; a2 = inf; return; }
  a1 = a; a2 = b;
}

// ---- strings: LatticeStringRepository (lattice-determinize.h:35-298) as a trie ---------------------------
WFST_HD inline uint32_t det_hash2(int32_t a, int32_t b) {
  uint32_t h = (uint32_t)a * 2654435761u;
  h ^= ((uint32_t)b + 0x9E3779B9u) * 0x85EBCA6Bu;
  h ^= h >> 15;
  return h * 0x2C1B3C6Du;
}
constexpr uint64_t kDetEmptyKey = ~0ull, kDetRootKey = ~0ull - 1;
WFST_HD inline uint64_t det_key(int32_t parent, int32_t label) { return (uint64_t)(uint32_t)parent | ((uint64_t)(uint32_t)label << 32); }
WFST_HD inline int32_t det_parent(const DetWs &W, int32_t n) { return (int32_t)(uint32_t)W.tr_key[n]; }
WFST_HD inline int32_t det_label(const DetWs &W, int32_t n) { return (int32_t)(uint32_t)(W.tr_key[n] >> 32); }
// Successor (:58-79)
WFST_HD inline int32_t det_succ(DetWs &W, int32_t parent, int32_t label) {
  const uint32_t mask = (uint32_t)W.tr_hcap - 1u;
  const uint64_t key = det_key(parent, label);
  uint32_t s = det_hash2(parent, label) & mask;
  for (;;) {
    const uint64_t k = W.tr_key[s];
    if (k == key) return (int32_t)s;
    if (k == kDetEmptyKey) break;
    s = (s + 1) & mask;
  }
  if (W.tr_n >= W.cap.trie || 2 * (int64_t)W.tr_n >= W.tr_hcap) { W.err = 1; return 0; }  // 1: trie
  ++W.tr_n;
  W.tr_key[s] = key;
  W.tr_depth[s] = W.tr_depth[parent] + 1;
  return (int32_t)s;
}
// the longest common prefix of two strings = their lowest common ancestor (CommonPrefix / ReduceToCommonPrefix, :95-128)
WFST_HD inline int32_t det_lca(const DetWs &W, int32_t a, int32_t b) {
  while (W.tr_depth[a] > W.tr_depth[b]) a = det_parent(W, a);
  while (W.tr_depth[b] > W.tr_depth[a]) b = det_parent(W, b);
  while (a != b) { a = det_parent(W, a); b = det_parent(W, b); }
  return a;
}
// the labels of string `s` below depth `from`, in order, into W.labs; returns their number
WFST_HD inline int32_t det_labels(DetWs &W, int32_t s, int32_t from) {
  const int32_t n = W.tr_depth[s] - from;
  if (n > W.cap.tmp) { W.err = 6; return 0; }
  for (int32_t i = n - 1; i >= 0; --i) { const uint64_t k = W.tr_key[s]; W.labs[i] = (int32_t)(uint32_t)(k >> 32); s = (int32_t)(uint32_t)k; }
  return n;
}
// RemovePrefix (:131-142)
WFST_HD inline int32_t det_remove_prefix(DetWs &W, int32_t s, int32_t n) {
  if (n == 0) return s;
  const int32_t k = det_labels(W, s, n);
  int32_t ans = 0;
  for (int32_t i = 0; i < k; ++i) ans = det_succ(W, ans, W.labs[i]);
  return ans;
}
// Concatenate (:81-93)
WFST_HD inline int32_t det_concat(DetWs &W, int32_t a, int32_t b) {
  if (a == 0) return b;
  if (b == 0) return a;
  const int32_t k = det_labels(W, b, 0);
  int32_t ans = a;
  for (int32_t i = 0; i < k; ++i) ans = det_succ(W, ans, W.labs[i]);
  return ans;
}
// Compare (:966-997): weight first; then the LONGER string is the worse one; then lexicographic
WFST_HD inline int det_cmp(const DetWs &W, float a1, float a2, int32_t as, float b1, float b2, int32_t bs) {
  const int wc = det_wcmp(a1, a2, b1, b2);
  if (wc != 0) return wc;
  if (as == bs) return 0;
  const int32_t al = W.tr_depth[as], bl = W.tr_depth[bs];
  if (al > bl) return -1;
  if (al < bl) return 1;
  // equal lengths, different strings: the first position they differ at is just below their lowest common ancestor
  int32_t a = as, b = bs;
  while (det_parent(W, a) != det_parent(W, b)) { a = det_parent(W, a); b = det_parent(W, b); }
  return det_label(W, a) < det_label(W, b) ? -1 : 1;
}

WFST_HD inline void det_sort_by_state(DetElem *e, int32_t n) {  // subsets are small: insertion sort (Shell gaps for the odd large one)
  for (int32_t gap = n > 64 ? 40 : 1; gap >= 1; gap = gap > 1 ? (gap == 40 ? 13 : gap == 13 ? 4 : 1) : 0)
    for (int32_t i = gap; i < n; ++i) {
      const DetElem x = e[i];
      int32_t j = i;
      while (j >= gap && e[j - gap].state > x.state) { e[j] = e[j - gap]; j -= gap; }
      e[j] = x;
    }
}

// EpsilonClosure (:842-936) of subset e[0..n) (one element per state), in place; returns the new size, sorted by state
WFST_HD inline int32_t det_closure(DetWs &W, DetElem *e, int32_t n) {
  for (int pass = (W.tmp_lo > 0 && n <= W.tmp_lo) ? 0 : 1; pass < 2; ++pass) {
    DetElem *cur = pass == 0 ? W.tc_lo : W.tc;     // the current best element of every state reached
    DetElem *queue = pass == 0 ? W.tb_lo : W.tb;   // FIFO of elements to expand (a ring: an improved state is queued again)
    const int32_t cap = pass == 0 ? W.tmp_lo : W.cap.tmp;
    int32_t ncur = 0;
    int32_t qh = 0, qt = 0, qn = 0;   // ring: head, tail, elements queued (no 64-bit modulo: the lane is instruction-bound)
    bool over = false;
    for (int32_t i = 0; i < n; ++i) {
      if (ncur >= cap) { over = true; break; }
      W.cl_idx[e[i].state] = ncur;
      cur[ncur++] = e[i];
      queue[qt] = e[i];
      if (++qt == cap) qt = 0;
      ++qn;
    }
    bool replaced = false;
    while (qn > 0 && !W.err && !over) {
      const DetElem el = queue[qh];
      if (++qh == cap) qh = 0;
      --qn;
      if (replaced) {  // a better element for this state is further down the queue: skip the stale one
        const DetElem &c = cur[W.cl_idx[el.state]];
        if (c.str != el.str || c.w1 != el.w1 || c.w2 != el.w2) continue;
      }
      const int32_t a0 = W.off[el.state], a1 = a0 + W.neps[el.state];   // sorted: the epsilons lead the row
      for (int32_t a = a0; a < a1; ++a) {
        const DetArc &arc = W.arcs[a];
        if (det_is_zero(arc.w1, arc.w2)) continue;
        DetElem nx;
        nx.state = arc.to;
        nx.w1 = el.w1 + arc.w1;
        nx.w2 = el.w2 + arc.w2;
        nx.str = arc.olabel == 0 ? el.str : det_succ(W, el.str, arc.olabel);
        const int32_t idx = W.cl_idx[nx.state];
        bool push = false;
        if (idx < 0) {
          if (ncur >= cap) { over = true; break; }
          W.cl_idx[nx.state] = ncur;
          cur[ncur++] = nx;
          push = true;
        } else if (det_cmp(W, nx.w1, nx.w2, nx.str, cur[idx].w1, cur[idx].w2, cur[idx].str) == 1) {
          cur[idx].w1 = nx.w1; cur[idx].w2 = nx.w2; cur[idx].str = nx.str;
          push = true;
          replaced = true;
        }
        if (push) {
          if (qn >= cap) { over = true; break; }
          queue[qt] = nx;
          if (++qt == cap) qt = 0;
          ++qn;
        }
      }
    }
    if (over) {
      for (int32_t i = 0; i < ncur; ++i) W.cl_idx[cur[i].state] = -1;
      if (pass == 0) continue;   // (outgrew the fast buffers: once more in the workspace's; the trie nodes made so far are found again)
      W.err = 6;
      return 0;
    }
    for (int32_t i = 0; i < ncur; ++i) { W.cl_idx[cur[i].state] = -1; e[i] = cur[i]; }
    det_sort_by_state(e, ncur);
    return ncur;
  }
  return 0;
}

// ConvertToMinimal (:940-957)
WFST_HD inline int32_t det_minimal(const DetWs &W, DetElem *e, int32_t n) {
  int32_t k = 0;
  for (int32_t i = 0; i < n; ++i)
    if (W.osf[e[i].state]) e[k++] = e[i];
  return k;
}

// NormalizeSubset (:1219-1252)
WFST_HD inline void det_normalize(DetWs &W, DetElem *e, int32_t n, float *t1, float *t2, int32_t *common) {
  if (n == 0) { *common = 0; *t1 = __builtin_huge_valf(); *t2 = __builtin_huge_valf(); return; }
  float w1 = e[0].w1, w2 = e[0].w2;
  int32_t pre = e[0].str;
  for (int32_t i = 1; i < n; ++i) {
    if (!(det_wcmp(w1, w2, e[i].w1, e[i].w2) >= 0)) { w1 = e[i].w1; w2 = e[i].w2; }  // Plus (:303-308)
    pre = det_lca(W, pre, e[i].str);
  }
  const int32_t plen = W.tr_depth[pre];
  for (int32_t i = 0; i < n; ++i) {
    det_divide(e[i].w1, e[i].w2, w1, w2);
    e[i].str = det_remove_prefix(W, e[i].str, plen);
  }
  *common = pre;
  *t1 = w1; *t2 = w2;
}

WFST_HD inline uint32_t det_subset_hash(const DetElem *e, int32_t n) {  // SubsetKey (:643-658): states and strings only
  uint32_t h = 2166136261u;
  for (int32_t i = 0; i < n; ++i) h = (h ^ det_hash2(e[i].state, e[i].str)) * 16777619u;
  return h;
}
WFST_HD inline bool det_subset_equal(const DetElem *a, int32_t na, const DetElem *b, int32_t nb, float delta) {  // SubsetEqual (:662-685)
  if (na != nb) return false;
  for (int32_t i = 0; i < na; ++i)
    if (a[i].state != b[i].state || a[i].str != b[i].str || !det_approx(a[i].w1, a[i].w2, b[i].w1, b[i].w2, delta)) return false;
  return true;
}

WFST_HD inline int32_t det_store(DetWs &W, const DetElem *e, int32_t n) {  // a copy in the pool
  if ((int64_t)W.pool_n + n > W.cap.pool) { W.err = 2; return 0; }
  const int32_t o = W.pool_n;
  for (int32_t i = 0; i < n; ++i) W.pool[o + i] = e[i];
  W.pool_n += n;
  return o;
}

// MinimalToStateId (:1310-1324).  (A new key goes to the FRONT of its bucket, as in libstdc++'s unordered_map: where
// several stored subsets match within delta, the most recent is found.)
WFST_HD inline int32_t det_minimal_to_state(DetWs &W, const DetElem *e, int32_t n, bool look) {
  const uint32_t b = det_subset_hash(e, n) & ((uint32_t)W.mh_cap - 1u);
  if (look)
    for (int32_t s = W.mh_head[b]; s >= 0; s = W.os_next[s])
      if (det_subset_equal(e, n, W.pool + W.os_off[s], W.os_len[s], W.delta)) return s;
  if (W.os_n >= W.cap.states) { W.err = 3; return 0; }
  const int32_t s = W.os_n++;
  W.os_off[s] = det_store(W, e, n);
  W.os_len[s] = n;
  W.os_next[s] = W.mh_head[b];
  W.mh_head[b] = s;
  W.queue[W.q_n++] = s;
  return s;
}

// InitialToStateId (:1265-1306); the subset e[0..n) (normalized, before the epsilon closure) is left untouched
WFST_HD inline int32_t det_initial_to_state(DetWs &W, const DetElem *e, int32_t n, float *r1, float *r2, int32_t *rstr) {
  const uint32_t b = det_subset_hash(e, n) & ((uint32_t)W.ih_hcap - 1u);
  for (int32_t k = W.ih_head[b]; k >= 0; k = W.ih_next[k])
    if (det_subset_equal(e, n, W.pool + W.ih_off[k], W.ih_len[k], W.delta)) {
      *r1 = W.ih_w1[k]; *r2 = W.ih_w2[k]; *rstr = W.ih_str[k];
      return W.ih_state[k];
    }
  if (n > W.cap.tmp) { W.err = 6; return 0; }
  DetElem *s = W.ta;   // (the caller's subset lives elsewhere)
  for (int32_t i = 0; i < n; ++i) s[i] = e[i];
  int32_t m = det_closure(W, s, n);
  m = det_minimal(W, s, m);
  float w1, w2;
  int32_t str;
  det_normalize(W, s, m, &w1, &w2, &str);
  const int32_t ans = det_minimal_to_state(W, s, m, true);
  if (W.ih_n >= W.cap.initials) { W.err = 4; return ans; }
  const int32_t k = W.ih_n++;
  W.ih_off[k] = det_store(W, e, n);
  W.ih_len[k] = n;
  W.ih_state[k] = ans; W.ih_w1[k] = w1; W.ih_w2[k] = w2; W.ih_str[k] = str;
  W.ih_next[k] = W.ih_head[b];
  W.ih_head[b] = k;
  *r1 = w1; *r2 = w2; *rstr = str;
  return ans;
}

WFST_HD inline void det_add_arc(DetWs &W, int32_t src, int32_t ilabel, int32_t next, float w1, float w2) {
  if (W.oa_n >= W.cap.arcs) { W.err = 5; return; }
  DetOutArc a;
  a.src = src; a.ilabel = ilabel; a.next = next; a.w1 = w1; a.w2 = w2;
  W.oarcs[W.oa_n++] = a;
}

// ProcessState = ProcessFinal + ProcessTransitions (:1029-1181)
WFST_HD inline void det_process_state(DetWs &W, int32_t out) {
  const int32_t n = W.os_len[out];
  // (the pool may not move, but slices of it are appended while we work: index it afresh)
  {  // ProcessFinal
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
  // ProcessTransitions: every non-epsilon arc out of every element, sorted by (label, next state)
  int32_t m = 0;
  for (int32_t i = 0; i < n && !W.err; ++i) {
    const DetElem el = W.pool[W.os_off[out] + i];
    for (int32_t a = W.off[el.state]; a < W.off[el.state + 1]; ++a) {
      const DetArc &arc = W.arcs[a];
      if (arc.ilabel == 0 || det_is_zero(arc.w1, arc.w2)) continue;
      if (m >= W.cap.tmp) { W.err = 6; break; }
      DetElem nx;
      nx.state = arc.to;
      nx.w1 = el.w1 + arc.w1;
      nx.w2 = el.w2 + arc.w2;
      nx.str = arc.olabel == 0 ? el.str : det_succ(W, el.str, arc.olabel);
      W.tc[m] = nx;          // (tc is free: the closure is not running)
      W.ta_label[m] = arc.ilabel;
      ++m;
    }
  }
  // sort the pairs: first on the label, then on the state (PairComparator, :728-743)
  for (int32_t gap = m > 64 ? 40 : 1; gap >= 1; gap = gap > 1 ? (gap == 40 ? 13 : gap == 13 ? 4 : 1) : 0)
    for (int32_t i = gap; i < m; ++i) {
      const DetElem x = W.tc[i];
      const int32_t xl = W.ta_label[i];
      int32_t j = i;
      while (j >= gap && (W.ta_label[j - gap] > xl || (W.ta_label[j - gap] == xl && W.tc[j - gap].state > x.state))) {
        W.tc[j] = W.tc[j - gap]; W.ta_label[j] = W.ta_label[j - gap]; j -= gap;
      }
      W.tc[j] = x; W.ta_label[j] = xl;
    }
  // (tc is the closure's scratch and the closure runs inside ProcessTransition: the pairs move to td)
  for (int32_t i = 0; i < m; ++i) W.td[i] = W.tc[i];
  int32_t i = 0;
  while (i < m && !W.err) {
    const int32_t ilabel = W.ta_label[i];
    // ProcessTransition (:1153-1181) on the range with this label
    DetElem *sub = W.te;
    int32_t k = 0;
    // MakeSubsetUnique (:1184-1216): merge the elements of one state, keeping the better (weight, string)
    while (i < m && W.ta_label[i] == ilabel) {
      DetElem cur = W.td[i];
      ++i;
      while (i < m && W.ta_label[i] == ilabel && W.td[i].state == cur.state) {
        const DetElem &o = W.td[i];
        if (det_cmp(W, o.w1, o.w2, o.str, cur.w1, cur.w2, cur.str) == 1) { cur.w1 = o.w1; cur.w2 = o.w2; cur.str = o.str; }
        ++i;
      }
      sub[k++] = cur;   // (k <= m <= tmp)
    }
    float t1, t2, n1, n2;
    int32_t common, nstr;
    det_normalize(W, sub, k, &t1, &t2, &common);
    const int32_t next = det_initial_to_state(W, sub, k, &n1, &n2, &nstr);
    // (the arc's string Concatenate(common, nstr) is dropped by OutputNoolabel: not built)
    det_add_arc(W, out, ilabel, next, t1 + n1, t2 + n2);   // Times(tot_weight, next_tot_weight)
  }
}

// Table initialisation, shared out over `nthreads` callers (the device calls it workgroup-wide, tid = thread index;
// the host once with (0, 1)).  A barrier must separate it from det_run.
WFST_HD inline void det_init(DetWs &W, int32_t tid, int32_t nthreads) {
  for (int32_t i = tid; i < W.tr_hcap; i += nthreads) W.tr_key[i] = kDetEmptyKey;   // (the slots in use: the device starts with a part of the table)
  for (int32_t i = tid; i < 2 * W.cap.states; i += nthreads) W.mh_head[i] = -1;
  for (int32_t i = tid; i < 2 * W.cap.initials; i += nthreads) W.ih_head[i] = -1;
  for (int32_t s = tid; s < W.n_states; s += nthreads) {
    W.cl_idx[s] = -1;
    int32_t y = W.is_final[s] ? 1 : 0;   // IsIsymbolOrFinal (:999-1024)
    int32_t ne = 0;
    for (int32_t a = W.off[s]; a < W.off[s + 1]; ++a) {
      if (W.arcs[a].ilabel == 0) { ++ne; continue; }
      if (!det_is_zero(W.arcs[a].w1, W.arcs[a].w2)) { y = 1; break; }
    }
    W.osf[s] = y;
    W.neps[s] = ne;
  }
}

// InitializeDeterminization + the main loop (:551-600, 795-838), after det_init.  Returns 0, or which capacity was exceeded.
WFST_HD inline int det_run(DetWs &W) {
  W.err = 0;
  W.tr_n = 1; W.tr_key[0] = kDetRootKey; W.tr_depth[0] = 0;   // the empty string: slot 0
  W.pool_n = 0; W.os_n = 0; W.ih_n = 0; W.q_n = 0; W.oa_n = 0;
  if (W.n_states <= 0) return 0;
  {
    DetElem *s = W.ta;
    s[0].state = 0; s[0].str = 0; s[0].w1 = 0.0f; s[0].w2 = 0.0f;
    int32_t m = det_closure(W, s, 1);
    m = det_minimal(W, s, m);
    det_minimal_to_state(W, s, m, false);   // state 0: not normalized (:799-838)
  }
  while (W.q_n > 0 && !W.err) {
    const int32_t out = W.queue[--W.q_n];   // LIFO (:561-563)
    det_process_state(W, out);
  }
  return W.err;
}

}  // namespace wfst
#endif
