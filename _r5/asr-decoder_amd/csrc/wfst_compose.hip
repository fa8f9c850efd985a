// compose2_kernel: the service's second LM pass on a determinized lattice -- ComposeLattice with the old LM, then with the
// new one (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:53-78 under --use-second; newfst/compose-lat-inl.h:15-130), each
// followed by Connect (newfst/connect-fst.cc:10-22) -- on the determinized lattice that determinize_kernel left in its workspace
// slot, against the LM automata resident in HBM (wfst_lm: ComposeArpaLm::GetArc / Final / Start = lm_getarc / lm_final_cost /
// LmDev::start).
//
// ComposeLattice walks the pairs (lattice state, LM state) breadth first: a pair's arcs are the lattice state's arcs, an arc
// with a word label steps the LM (backing off until the word is found) and adds its cost, an arc into a FINAL lattice state
// also adds the LM's final cost of the LM state it arrives with (and makes the composed state final).  New pairs are numbered
// in the order they are met; the queue is first-in first-out, so pairs are processed in the order of their numbers and the
// queue is the state array itself.  A determinized lattice is a hundred or two states: one workgroup, the walk on one lane,
// the table initialisation and the trimming sweeps on all of them.
#include "wfst_device.h"

namespace wfst {

constexpr int kCmpThreads = 256;

struct CmpArc { int32_t src, dst, olabel; float g, ac; };   // ilabel is 0 throughout (OutputNoolabel, lattice-determinize.h:307-377)

// One ComposeLattice + Connect.  in: n_in states, fin_in[], arcs (any order); out: arcs, fin_out[], *n_out.  Returns false on a
// capacity overflow.  Workspace: off[n_in + 1], ord[a_in], keys[2 * pair_cap] (u64), pst[pair_cap] (lattice state), plm[pair_cap],
// keep[pair_cap], renum[pair_cap].
__device__ bool compose_once(const LmDev &L, float scale, int n_in, const int32_t *fin_in, int a_in, const CmpArc *in, int pair_cap,
                             int arc_cap, int32_t *ws, CmpArc *out, int32_t *fin_out, int *n_out, int *a_out) {
  const int tid = threadIdx.x;
  int32_t *off = ws;
  int32_t *ord = off + n_in + 1;
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(ord + a_in + ((a_in + n_in + 1) & 1));   // (8-byte aligned)
  int32_t *kid = reinterpret_cast<int32_t *>(keys + 2 * (size_t)pair_cap);
  int32_t *pst = kid + 2 * (size_t)pair_cap;
  int32_t *plm = pst + pair_cap;
  int32_t *keep = plm + pair_cap;
  int32_t *renum = keep + pair_cap;
  __shared__ int s_n, s_a, s_ok, s_changed;
  const unsigned hmask = 2u * (unsigned)pair_cap - 1u;
  for (int i = tid; i < 2 * pair_cap; i += kCmpThreads) keys[i] = ~0ull;
  for (int i = tid; i <= n_in; i += kCmpThreads) off[i] = 0;
  __syncthreads();
  if (tid == 0) {
    bool ok = n_in <= pair_cap;
    // CSR of the input by source state, a state's arcs in their given order (counting sort; the cursors sit in `keep`, which is
    // free until the trimming)
    for (int a = 0; a < a_in && ok; ++a) off[in[a].src + 1]++;
    for (int s = 0; s < n_in && ok; ++s) off[s + 1] += off[s];
    for (int s = 0; s < n_in && ok; ++s) keep[s] = off[s];
    for (int a = 0; a < a_in && ok; ++a) ord[keep[in[a].src]++] = a;
    int n = 0, na = 0;
    auto intern = [&](int st, int lm, bool *fresh) -> int {
      const unsigned long long key = (unsigned long long)(unsigned)st | ((unsigned long long)(unsigned)lm << 32);
      unsigned h = ((unsigned)st * 2654435761u) ^ (((unsigned)lm + 0x9E3779B9u) * 0x85EBCA6Bu);
      h ^= h >> 15;
      unsigned slot = (h * 0x2C1B3C6Du) & hmask;
      for (;;) {
        if (keys[slot] == key) { *fresh = false; return kid[slot]; }
        if (keys[slot] == ~0ull) break;
        slot = (slot + 1) & hmask;
      }
      if (n >= pair_cap) { ok = false; *fresh = false; return 0; }
      keys[slot] = key;
      kid[slot] = n;
      pst[n] = st;
      plm[n] = lm;
      fin_out[n] = 0;
      *fresh = true;
      return n++;
    };
    bool fresh;
    if (ok) intern(0, L.start, &fresh);   // (clat->Start(), fst->Start()): state 0 of a determinized lattice is its start
    for (int id = 0; id < n && ok; ++id) {
      const int s1 = pst[id], s2 = plm[id];
      for (int k = off[s1]; k < off[s1 + 1] && ok; ++k) {
        const CmpArc A = in[ord[k]];
        int next2 = s2;
        float lw = 0.0f;
        if (A.olabel != 0) lm_getarc(L, s2, A.olabel, &next2, &lw);   // ComposeArpaLm::GetArc: always matches (backs off to the unigram)
        const int nid = intern(A.dst, next2, &fresh);
        if (!ok) break;
        // "Because state isn't save final score, so add final score in arc" (compose-lat-inl.h:97-106)
        float final_score = 0.0f;
        if (fin_in[A.dst]) {
          final_score = lm_final_cost(L, next2);
          if (final_score == __builtin_huge_valf()) final_score = 0.0f;
          else fin_out[nid] = 1;
        }
        CmpArc O;
        O.src = id;
        O.dst = nid;
        if (A.olabel == 0) {
          O.olabel = 0;
          O.ac = A.ac;
          O.g = A.g + final_score * scale;                 // :113-114
        } else {
          O.olabel = A.olabel;
          O.ac = A.ac + 0.0f * scale;                      // lweight.Value2() is 0 (compose-arpalm.cc:66)
          O.g = A.g + (lw + final_score) * scale;          // :120-121
        }
        if (na >= arc_cap) { ok = false; break; }
        out[na++] = O;
      }
    }
    s_n = n;
    s_a = na;
    s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  if (!s_ok) return false;
  const int n = s_n, na = s_a;
  // Connect: every composed state is accessible by construction; keep those that reach a final state (sweeps to the fixpoint)
  for (int i = tid; i < n; i += kCmpThreads) keep[i] = fin_out[i];
  __syncthreads();
  for (;;) {
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int a = tid; a < na; a += kCmpThreads)
      if (keep[out[a].dst] && !keep[out[a].src]) { keep[out[a].src] = 1; s_changed = 1; }
    __syncthreads();
    const int ch = s_changed;
    __syncthreads();
    if (!ch) break;
  }
  if (tid == 0) {
    int m = 0;
    for (int i = 0; i < n; ++i) renum[i] = keep[i] ? m++ : -1;
    for (int i = 0; i < n; ++i)
      if (renum[i] >= 0) fin_out[renum[i]] = fin_out[i];
    int ma = 0;
    for (int a = 0; a < na; ++a) {
      CmpArc O = out[a];
      if (renum[O.src] < 0 || renum[O.dst] < 0) continue;
      O.src = renum[O.src];
      O.dst = renum[O.dst];
      out[ma++] = O;
    }
    *n_out = m;
    *a_out = ma;
  }
  __syncthreads();
  return true;
}

__global__ __launch_bounds__(kCmpThreads) void compose2_kernel(DetDev X, CmpDev Y, LmDev lm1, LmDev lm2) {
  const int tid = threadIdx.x;
  {
    // one workgroup per lattice: workgroup b composes the determinized lattice of the determinizer's workspace slot b into slot b
    // of the composition's buffers (a batch of the service's --use-second requests is ONE launch, not one per utterance)
    const size_t b = blockIdx.x;
    X.result += 4 * b; X.out_a += b * (size_t)X.out_cap; X.out_w += b * (size_t)X.out_cap;
    Y.ws += b * (size_t)Y.ws_ints; Y.result += 4 * b;
    Y.out_a += b * (size_t)Y.arc_cap; Y.out_w += b * (size_t)Y.arc_cap; Y.out_fin += b * (size_t)Y.pair_cap;
  }
  const int32_t *res = X.result;   // slot 0: {states, arcs, status, determinized states proper}
  int32_t *ores = Y.result;
  if (tid == 0) { ores[0] = 0; ores[1] = 0; ores[2] = 0; ores[3] = 0; }
  if (res[2] != 0 || res[0] <= 0) {
    if (tid == 0) ores[2] = res[2] != 0 ? 2 : 0;   // the determinizer's own failure / no lattice
    return;
  }
  const int ns = res[0], na = res[1], n_proper = res[3];
  // workspace: [stage A arcs | stage B arcs | fin A | fin B | per-pass tables]
  CmpArc *arcA = reinterpret_cast<CmpArc *>(Y.ws);
  CmpArc *arcB = arcA + Y.arc_cap;
  int32_t *finA = reinterpret_cast<int32_t *>(arcB + Y.arc_cap);
  int32_t *finB = finA + Y.pair_cap;
  int32_t *tab = finB + Y.pair_cap;
  __shared__ int s_n, s_a;
  if (ns > Y.pair_cap || na > Y.arc_cap) {
    if (tid == 0) ores[2] = 1;
    return;
  }
  // the determinized lattice: a final weight is an arc into a final state of its own (states >= n_proper)
  for (int a = tid; a < na; a += kCmpThreads) {
    const int4 t = X.out_a[a];
    const float2 w = X.out_w[a];
    CmpArc A;
    A.src = t.x; A.dst = t.y; A.olabel = t.z; A.g = w.x; A.ac = w.y;
    arcA[a] = A;
  }
  for (int s = tid; s < ns; s += kCmpThreads) finA[s] = s >= n_proper ? 1 : 0;
  __syncthreads();
  int n1 = 0, a1 = 0;
  if (!compose_once(lm1, 1.0f, ns, finA, na, arcA, Y.pair_cap, Y.arc_cap, tab, arcB, finB, &s_n, &s_a)) {
    if (tid == 0) ores[2] = 1;
    return;
  }
  __syncthreads();
  n1 = s_n; a1 = s_a;
  __syncthreads();
  if (!compose_once(lm2, 1.0f, n1, finB, a1, arcB, Y.pair_cap, Y.arc_cap, tab, arcA, finA, &s_n, &s_a)) {
    if (tid == 0) ores[2] = 1;
    return;
  }
  __syncthreads();
  const int n2 = s_n, a2 = s_a;
  for (int a = tid; a < a2; a += kCmpThreads) {
    const CmpArc A = arcA[a];
    Y.out_a[a] = make_int4(A.src, A.dst, A.olabel, finA[A.dst]);
    Y.out_w[a] = make_float2(A.g, A.ac);
  }
  for (int s = tid; s < n2; s += kCmpThreads) Y.out_fin[s] = finA[s];
  if (tid == 0) { ores[0] = n2; ores[1] = a2; }
}

void launch_compose2(const DetDev &X, const CmpDev &Y, const LmDev &lm1, const LmDev &lm2, int n_slots, hipStream_t s) {
  hipLaunchKernelGGL(compose2_kernel, dim3(n_slots), dim3(kCmpThreads), 0, s, X, Y, lm1, lm2);
}

}  // namespace wfst
