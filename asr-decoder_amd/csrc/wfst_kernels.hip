// HIP kernels of the batched WFST token-passing decoder, written for gfx950 (MI355X, wave64).
//
// One frame of the reference's hot loop (AdvanceDecoding, my-decoder/online-decoder-base-inl.h:
// 649-667) for ALL channels of a batch is two launches:
//
//   boundary_kernel  one 1024-thread workgroup per channel.
//       finalize part  = tail of ProcessEmitting + ProcessNonemitting (base-inl.h:353-431):
//                        collect the slots the expansion created, drop tokens that lost against
//                        the final next_cutoff, run the epsilon closure to its fixpoint inside
//                        the kernel, append the frame's tokens to the arena, clear the old table.
//       prep part      = GetCutoff (base-inl.h:138-234, exact k-th smallest by LDS radix select)
//                        + the best-token seeding of next_cutoff (base-inl.h:282-300).
//   expand_kernel    load-balanced ProcessEmitting inner loop (base-inl.h:311-347): a workgroup
//                    takes 256 frontier tokens, scans their emitting out-degrees in LDS and maps
//                    one lane to one arc, so low-degree HCLG states (2-3 arcs) fill wavefronts;
//                    survivors go into the channel's open-addressed next-state hash with a
//                    64-bit atomicMin of (orderable cost << 32 | arc).
//
// Float arithmetic follows the reference's operation order exactly (compiled with
// -ffp-contract=off): tot = (cur + (-loglike)) + graph; seed = (cur + graph) - loglike.
// There is no MFMA here: the path is irregular graph traversal, HBM/L2-latency bound.
#include "wfst_device.h"

namespace wfst {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t f2o(float f) {
  uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float o2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
  return __uint_as_float(u);
}
// L2-served loads for words that atomics of this launch may have changed (a plain load could be
// answered by a stale line of this CU's L1: MI355X_MICROARCH "inter-workgroup visibility").
template <class T>
__device__ __forceinline__ T ld_agent(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    u64 o = __shfl_xor(v, m, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ int lane_rank(u64 mask) {  // active lanes below this one
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}
__device__ __forceinline__ uint32_t hash_state(int32_t s, int log2cap) {
  return ((uint32_t)s * 2654435761u) >> (32 - log2cap);
}

// Open-addressed insert (linear probing).  Returns the slot of `state`, -1 if the table is full.
__device__ __forceinline__ int find_or_insert(int32_t *keys, int cap, int log2cap, int32_t state, bool *created) {
  uint32_t slot = hash_state(state, log2cap);
  const uint32_t mask = (uint32_t)cap - 1;
  *created = false;
  for (int p = 0; p < cap; ++p) {
    int32_t k = ld_agent(&keys[slot]);
    if (k == state) return (int)slot;
    if (k == kEmptyKey) {
      int32_t old = atomicCAS(&keys[slot], kEmptyKey, state);
      if (old == kEmptyKey) { *created = true; return (int)slot; }
      if (old == state) return (int)slot;
    }
    slot = (slot + 1) & mask;
  }
  return -1;
}
template <bool kAgent>
__device__ __forceinline__ int find_slot(const int32_t *keys, int cap, int log2cap, int32_t state) {
  uint32_t slot = hash_state(state, log2cap);
  const uint32_t mask = (uint32_t)cap - 1;
  for (int p = 0; p < cap; ++p) {
    int32_t k = kAgent ? ld_agent(&keys[slot]) : keys[slot];
    if (k == state) return (int)slot;
    if (k == kEmptyKey) return -1;
    slot = (slot + 1) & mask;
  }
  return -1;
}

// =========================================================================================
// expand_kernel: ProcessEmitting's inner loop (base-inl.h:311-347), load balanced.
//   grid (n_channels, tiles_per_channel), 256 threads.  blockIdx.x = channel, so with
//   n_channels % 8 == 0 all workgroups of a channel share an XCD (speed only: its hash table and
//   log-likelihood row then stay in one L2).
// =========================================================================================
constexpr int kExpandThreads = 256;

__global__ __launch_bounds__(kExpandThreads) void expand_kernel(DecoderDev D) {
  const int c = blockIdx.x;
  ChanCtl *ctl = D.ctl + c;
  if (!ctl->active) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = ctl->front_count;
  const int4 *tok = D.tok + (size_t)c * D.arena_cap + ctl->front_begin;
  const int tabT = ctl->cur_tab ^ 1;
  const size_t tab_off = ((size_t)c * 2 + tabT) * (size_t)D.cap;
  int32_t *keys = D.keys + tab_off;
  u64 *vals = D.vals + tab_off;
  int32_t *occ = D.occ + tab_off;
  const float cutoff = ctl->cur_cutoff, ab = ctl->adaptive_beam;
  const float *llrow = D.ll_base[c] + (size_t)ctl->n_decoded * D.stride;
  const float kInf = __builtin_huge_valf();

  __shared__ int s_base[kExpandThreads + 1];
  __shared__ int s_arcbeg[kExpandThreads];
  __shared__ float s_cost[kExpandThreads];
  __shared__ int s_wsum[kExpandThreads / 64];

  u64 nN = 0, nE = 0;
  for (int tile = blockIdx.y; tile * kExpandThreads < n; tile += gridDim.y) {
    const int i = tile * kExpandThreads + tid;
    int deg = 0, arcbeg = 0;
    float cost = 0.f;
    if (i < n) {
      int4 t = tok[i];
      cost = __int_as_float(t.y);
      if (cost <= cutoff) {  // base-inl.h:315
        uint2 si = D.g.state_info[t.x];
        deg = (int)(si.y >> kEpsBits);
        arcbeg = (int)(si.x + (si.y & kEpsMask));
        nN++;
        nE += deg;
      }
    }
    int incl = deg;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    s_cost[tid] = cost;
    s_arcbeg[tid] = arcbeg;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kExpandThreads / 64; ++w) {
      int v = s_wsum[w];
      if (w < wave) wbase += v;
      total += v;
    }
    s_base[tid] = wbase + incl - deg;
    if (tid == 0) s_base[kExpandThreads] = total;
    __syncthreads();

    float bound = o2f(ld_agent(&ctl->bound));
    for (int j0 = 0; j0 < total; j0 += kExpandThreads) {
      const int j = j0 + tid;
      const bool valid = j < total;
      float tot = kInf;
      int a = 0;
      int32_t nextstate = 0;
      if (valid) {
        int lo = 0, hi = kExpandThreads;  // s_base[lo] <= j < s_base[hi]
        while (hi - lo > 1) {
          int mid = (lo + hi) >> 1;
          if (s_base[mid] <= j) lo = mid; else hi = mid;
        }
        a = s_arcbeg[lo] + (j - s_base[lo]);
        const int4 arc = D.g.arcs[a];
        const float ac_cost = -llrow[arc.x];                      // base-inl.h:326
        tot = (s_cost[lo] + ac_cost) + __int_as_float(arc.z);     // base-inl.h:329
        nextstate = arc.w;
      }
      // base-inl.h:330-333: tighten next_cutoff by the best candidate seen (wave-aggregated)
      const float cand = wave_min_f(tot) + ab;
      if (cand < bound) {
        uint32_t old = 0;
        if (lane == 0) old = atomicMin(&ctl->bound, f2o(cand));
        old = __shfl(old, 0, 64);
        bound = fminf(o2f(old), cand);
      }
      if (valid && tot < bound) {
        // FindOrAddToken (base-inl.h:88-136) as insert-or-min on (cost, arc)
        const u64 packed = ((u64)f2o(tot) << 32) | (uint32_t)a;
        bool created;
        const int slot = find_or_insert(keys, D.cap, D.log2cap, nextstate, &created);
        if (slot < 0) {
          atomicOr(&ctl->error, kErrTableFull);
        } else if (created) {
          const int pos = atomicAdd(&ctl->n_occ[tabT], 1);
          occ[pos] = slot;
          atomicMin(&vals[slot], packed);
        } else if (packed < ld_agent(&vals[slot])) {
          atomicMin(&vals[slot], packed);
        }
      }
    }
    __syncthreads();
  }
  nN = wave_sum_u64(nN);
  nE = wave_sum_u64(nE);
  if (lane == 0 && (nN | nE)) {
    atomicAdd(&ctl->cnt_N, nN);
    atomicAdd(&ctl->cnt_E, nE);
  }
}

// =========================================================================================
// boundary_kernel and its pieces.  One 1024-thread workgroup per channel.
// =========================================================================================
constexpr int kBT = 1024;
constexpr int kBW = kBT / 64;

struct BoundaryShared {
  int nfront;
  int wl_n[2];
  int err;
  float redf[kBW];
  u64 red64[kBW];
  uint32_t hist[256];
  uint32_t sel_prefix, sel_k;
  int active;
};

// ProcessNonemitting to its fixpoint (base-inl.h:383-430) on table `tabB`, then commit the
// frontier: arena records with resolved backpointers.  On entry sh.nfront tokens are listed in
// front_slot (with toki assigned) and sh.wl_n[0] of them (those with epsilon arcs) in
// worklist[0]; sh.wl_n[1] == 0.  Returns the number of frontier tokens written.
__device__ int closure_and_commit(const DecoderDev &D, int c, ChanCtl *ctl, BoundaryShared &sh,
                                  int tabB, int tabA, int base, float cutoff, u64 *nZ_out) {
  const int tid = threadIdx.x;
  const size_t offB = ((size_t)c * 2 + tabB) * (size_t)D.cap;
  int32_t *keysB = D.keys + offB;
  u64 *valsB = D.vals + offB;
  int32_t *tokiB = D.toki + offB;
  int32_t *occB = D.occ + offB;
  int32_t *front_slot = D.front_slot + (size_t)c * D.max_tok;
  int32_t *wl = D.worklist + (size_t)c * 2 * D.wl_cap;
  u64 nZ = 0;

  int cur = 0;
  for (;;) {
    __syncthreads();
    const int nw = sh.wl_n[cur];
    if (nw == 0) break;
    int32_t *wl_cur = wl + (size_t)cur * D.wl_cap;
    int32_t *wl_nxt = wl + (size_t)(cur ^ 1) * D.wl_cap;
    for (int i = tid; i < nw; i += kBT) {
      const int S = wl_cur[i];
      const float cost = o2f((uint32_t)(ld_agent(&valsB[S]) >> 32));
      if (!(cost < cutoff)) continue;  // base-inl.h:391
      const int32_t state = ld_agent(&keysB[S]);
      const uint2 si = D.g.state_info[state];
      const int neps = (int)(si.y & kEpsMask);
      for (int e = 0; e < neps; ++e) {
        const int a = (int)si.x + e;
        const int4 arc = D.g.arcs[a];
        nZ++;
        const float tot = cost + __int_as_float(arc.z);  // base-inl.h:414
        if (!(tot < cutoff)) continue;                    // base-inl.h:415
        const uint32_t otot = f2o(tot);
        const u64 packed = ((u64)otot << 32) | (uint32_t)a;
        bool created;
        const int ds = find_or_insert(keysB, D.cap, D.log2cap, arc.w, &created);
        if (ds < 0) { atomicOr(&sh.err, kErrTableFull); continue; }
        if (created) {
          const int pos = atomicAdd(&ctl->n_occ[tabB], 1);
          occB[pos] = ds;
        }
        const u64 old = atomicMin(&valsB[ds], packed);
        if (packed < old) {
          const bool was_in = o2f((uint32_t)(old >> 32)) < cutoff;
          if (!was_in) {  // newly created token (or one that had lost against the cutoff)
            const int fpos = atomicAdd(&sh.nfront, 1);
            if (fpos < D.max_tok) { front_slot[fpos] = ds; tokiB[ds] = base + fpos; }
          }
          // base-inl.h:425: re-queue when the cost changed and the state has epsilon arcs
          if (otot < (uint32_t)(old >> 32) && (D.g.state_info[arc.w].y & kEpsMask)) {
            const int wp = atomicAdd(&sh.wl_n[cur ^ 1], 1);
            if (wp < D.wl_cap) wl_nxt[wp] = ds; else atomicOr(&sh.err, kErrWorklistFull);
          }
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      sh.wl_n[cur] = 0;
      if (sh.wl_n[cur ^ 1] > D.wl_cap) sh.wl_n[cur ^ 1] = D.wl_cap;
    }
    cur ^= 1;
  }
  __syncthreads();
  int nf = sh.nfront;
  if (nf > D.max_tok) { if (tid == 0) atomicOr(&sh.err, kErrFrontierFull); nf = D.max_tok; }
  if ((int64_t)base + nf > D.arena_cap) { if (tid == 0) atomicOr(&sh.err, kErrArenaFull); nf = 0; }

  // commit: one 16-byte record per token; backpointer = token of the winning arc's source state
  int4 *tok = D.tok + (size_t)c * D.arena_cap;
  const int32_t *keysA = tabA >= 0 ? D.keys + ((size_t)c * 2 + tabA) * (size_t)D.cap : nullptr;
  const int32_t *tokiA = tabA >= 0 ? D.toki + ((size_t)c * 2 + tabA) * (size_t)D.cap : nullptr;
  for (int pos = tid; pos < nf; pos += kBT) {
    const int slot = front_slot[pos];
    const u64 v = ld_agent(&valsB[slot]);
    const int32_t state = ld_agent(&keysB[slot]);
    const uint32_t arc = (uint32_t)v;
    int prev = -1;
    if (arc != kNoArc) {
      const int32_t srci = D.g.arc_src[arc];
      const int32_t src = srci & 0x7FFFFFFF;
      int ss;
      if (srci < 0) {  // epsilon arc: source token lives on this frame
        ss = find_slot<true>(keysB, D.cap, D.log2cap, src);
        prev = ss >= 0 ? ld_agent(&tokiB[ss]) : -2;
      } else {
        ss = keysA ? find_slot<false>(keysA, D.cap, D.log2cap, src) : -1;
        prev = ss >= 0 ? tokiA[ss] : -2;
      }
    }
    tok[base + pos] = make_int4(state, __float_as_int(o2f((uint32_t)(v >> 32))), prev, (int)arc);
  }
  *nZ_out = nZ;
  return nf;
}

__device__ void finalize_frame(const DecoderDev &D, int c, ChanCtl *ctl, BoundaryShared &sh) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int f = ctl->n_decoded;
  const int tabA = ctl->cur_tab, tabB = tabA ^ 1;
  const float cutoff = o2f(ctl->bound);
  const int base = ctl->front_begin + ctl->front_count;
  const int n_occB = ctl->n_occ[tabB];
  const size_t offB = ((size_t)c * 2 + tabB) * (size_t)D.cap;
  const int32_t *keysB = D.keys + offB;
  const u64 *valsB = D.vals + offB;
  int32_t *tokiB = D.toki + offB;
  const int32_t *occB = D.occ + offB;
  int32_t *front_slot = D.front_slot + (size_t)c * D.max_tok;
  int32_t *wl0 = D.worklist + (size_t)c * 2 * D.wl_cap;

  if (tid == 0) { sh.nfront = 0; sh.wl_n[0] = 0; sh.wl_n[1] = 0; sh.err = 0; }
  __syncthreads();
  // keep what the expansion created and that beats the FINAL next_cutoff (a subset of the
  // reference's tokens: its order-dependent extras, base-inl.h:330, are never expanded)
  for (int i0 = 0; i0 < n_occB; i0 += kBT) {
    const int i = i0 + tid;
    bool keep = false;
    int slot = 0;
    if (i < n_occB) {
      slot = occB[i];
      keep = o2f((uint32_t)(valsB[slot] >> 32)) < cutoff;
    }
    const u64 m = __ballot(keep);
    int wbase = 0;
    if (lane == 0 && m) wbase = atomicAdd(&sh.nfront, __popcll(m));
    wbase = __shfl(wbase, 0, 64);
    bool has_eps = false;
    if (keep) {
      const int pos = wbase + lane_rank(m);
      if (pos < D.max_tok) { front_slot[pos] = slot; tokiB[slot] = base + pos; }
      has_eps = (D.g.state_info[keysB[slot]].y & kEpsMask) != 0;  // base-inl.h:376-381
    }
    const u64 me = __ballot(has_eps);
    int ebase = 0;
    if (lane == 0 && me) ebase = atomicAdd(&sh.wl_n[0], __popcll(me));
    ebase = __shfl(ebase, 0, 64);
    if (has_eps) {
      const int wp = ebase + lane_rank(me);
      if (wp < D.wl_cap) wl0[wp] = slot; else atomicOr(&sh.err, kErrWorklistFull);
    }
  }
  __syncthreads();
  if (tid == 0 && sh.wl_n[0] > D.wl_cap) sh.wl_n[0] = D.wl_cap;

  u64 nZ = 0;
  const int nf = closure_and_commit(D, c, ctl, sh, tabB, tabA, base, cutoff, &nZ);

  // clear the table of the frame just expanded by walking its occupied-slot list
  __syncthreads();
  {
    const size_t offA = ((size_t)c * 2 + tabA) * (size_t)D.cap;
    int32_t *keysA = D.keys + offA;
    u64 *valsA = D.vals + offA;
    const int32_t *occA = D.occ + offA;
    const int nA = ctl->n_occ[tabA];
    for (int i = tid; i < nA; i += kBT) {
      const int s = occA[i];
      keysA[s] = kEmptyKey;
      valsA[s] = kEmptyVal;
    }
  }
  nZ = wave_sum_u64(nZ);
  if (lane == 0) sh.red64[tid >> 6] = nZ;
  __syncthreads();
  if (tid == 0) {
    u64 z = 0;
    for (int w = 0; w < kBW; ++w) z += sh.red64[w];
    int err = sh.err;
    if (f + 2 > D.max_frames + 1) err |= kErrFramesFull;
    else {
      D.frame_off[(size_t)c * (D.max_frames + 2) + f + 2] = base + nf;
      D.cutoff_hist[(size_t)c * (D.max_frames + 2) + f + 1] = cutoff;
    }
    ctl->cnt_Z += z;
    ctl->cnt_tok += (u64)nf;
    ctl->cnt_slots += (u64)ctl->n_occ[tabB];
    if (nf > ctl->peak_tokens) ctl->peak_tokens = nf;
    ctl->n_occ[tabA] = 0;
    ctl->front_begin = base;
    ctl->front_count = nf;
    ctl->cur_tab = tabB;
    ctl->n_decoded = f + 1;
    ctl->active = 0;
    if (err) ctl->error |= err;
  }
  __syncthreads();
}

// exact k-th smallest (0-based) cost of the frontier: what std::nth_element leaves at
// _tmp_array[k] (base-inl.h:190-193, 211-216).  MSB-first radix select, 8 bits per pass, LDS
// histogram.
__device__ float kth_smallest(const int4 *tok, int n, int k, BoundaryShared &sh) {
  const int tid = threadIdx.x;
  if (tid == 0) { sh.sel_prefix = 0; sh.sel_k = (uint32_t)k; }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int b = tid; b < 256; b += kBT) sh.hist[b] = 0;
    __syncthreads();
    const uint32_t prefix = sh.sel_prefix;
    for (int i = tid; i < n; i += kBT) {
      const uint32_t o = f2o(__int_as_float(tok[i].y));
      if ((o & hi_mask) == (prefix & hi_mask)) atomicAdd(&sh.hist[(o >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t kk = sh.sel_k, cum = 0;
      int b = 0;
      for (; b < 255; ++b) {
        if (kk < cum + sh.hist[b]) break;
        cum += sh.hist[b];
      }
      sh.sel_prefix = prefix | ((uint32_t)b << shift);
      sh.sel_k = kk - cum;
    }
    __syncthreads();
  }
  return o2f(sh.sel_prefix);
}

__device__ void prep_frame(const DecoderDev &D, int c, ChanCtl *ctl, const int32_t *target, BoundaryShared &sh) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float kInf = __builtin_huge_valf();
  if (tid == 0) {
    int act = (ctl->n_decoded < target[c]) && ctl->error == 0 && !ctl->finalized;
    if (act && ctl->n_decoded >= D.max_frames) { ctl->error |= kErrFramesFull; act = 0; }
    sh.active = act;
    if (!act) ctl->active = 0;
  }
  __syncthreads();
  if (!sh.active) return;
  const int n = ctl->front_count;
  const int4 *tok = D.tok + (size_t)c * D.arena_cap + ctl->front_begin;

  // best token (GetCutoff's running minimum; ties -> lowest index)
  u64 best = ~0ull;
  for (int i = tid; i < n; i += kBT) {
    const u64 v = ((u64)f2o(__int_as_float(tok[i].y)) << 32) | (uint32_t)i;
    best = v < best ? v : best;
  }
  best = wave_min_u64(best);
  if (lane == 0) sh.red64[wave] = best;
  __syncthreads();
  best = sh.red64[0];
  for (int w = 1; w < kBW; ++w) best = sh.red64[w] < best ? sh.red64[w] : best;
  __syncthreads();
  const float best_w = n > 0 ? o2f((uint32_t)(best >> 32)) : kInf;
  const int best_i = (int)(uint32_t)best;

  // GetCutoff, base-inl.h:138-234
  float cutoff, ab;
  if (D.max_active == 2147483647 && D.min_active == 0) {
    ab = D.beam;
    cutoff = best_w + D.beam;
  } else {
    const float beam_cutoff = best_w + D.beam;
    float min_active_cutoff = kInf, max_active_cutoff = kInf;
    if (n > D.max_active) max_active_cutoff = kth_smallest(tok, n, D.max_active, sh);
    if (max_active_cutoff < beam_cutoff) {
      ab = max_active_cutoff - best_w + D.beam_delta;
      cutoff = max_active_cutoff;
    } else {
      if (n > D.min_active) {
        if (D.min_active == 0) min_active_cutoff = best_w;
        else min_active_cutoff = kth_smallest(tok, n, D.min_active, sh);
      }
      if (min_active_cutoff > beam_cutoff) {
        ab = min_active_cutoff - best_w + D.beam_delta;
        cutoff = min_active_cutoff;
      } else {
        ab = D.beam;
        cutoff = beam_cutoff;
      }
    }
  }

  // seed next_cutoff from the best token's emitting arcs, base-inl.h:282-300
  float seed = kInf;
  if (n > 0) {
    const int4 bt = tok[best_i];
    const uint2 si = D.g.state_info[bt.x];
    const int deg = (int)(si.y >> kEpsBits), ab0 = (int)(si.x + (si.y & kEpsMask));
    const float *llrow = D.ll_base[c] + (size_t)ctl->n_decoded * D.stride;
    const float bc = __int_as_float(bt.y);
    for (int e = tid; e < deg; e += kBT) {
      const int4 arc = D.g.arcs[ab0 + e];
      const float tot_score = (bc + __int_as_float(arc.z)) - llrow[arc.x];  // base-inl.h:295
      seed = fminf(seed, tot_score);
    }
  }
  seed = wave_min_f(seed);
  if (lane == 0) sh.redf[wave] = seed;
  __syncthreads();
  if (tid == 0) {
    float s = sh.redf[0];
    for (int w = 1; w < kBW; ++w) s = fminf(s, sh.redf[w]);
    const float next_cutoff = s + ab;  // min(x)+ab == min(x+ab): float add is monotone
    ctl->cur_cutoff = cutoff;
    ctl->adaptive_beam = ab;
    ctl->bound = f2o(s < kInf ? next_cutoff : kInf);
    ctl->active = 1;
  }
}

__global__ __launch_bounds__(kBT) void boundary_kernel(DecoderDev D, const int32_t *target, int do_finalize, int do_prep) {
  __shared__ BoundaryShared sh;
  const int c = blockIdx.x;
  ChanCtl *ctl = D.ctl + c;
  if (do_finalize && ctl->active) finalize_frame(D, c, ctl, sh);
  __syncthreads();
  if (do_prep) prep_frame(D, c, ctl, target, sh);
}

// =========================================================================================
// init: InitDecoding (base-inl.h:40-67)
// =========================================================================================
__global__ __launch_bounds__(256) void clear_tables_kernel(DecoderDev D, const int32_t *chans) {
  const int c = chans ? chans[blockIdx.x] : blockIdx.x;
  const size_t off = (size_t)c * 2 * (size_t)D.cap;
  const size_t n = (size_t)2 * D.cap;
  for (size_t i = (size_t)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.y * blockDim.x) {
    D.keys[off + i] = kEmptyKey;
    D.vals[off + i] = kEmptyVal;
  }
}

__global__ __launch_bounds__(kBT) void init_kernel(DecoderDev D, const int32_t *chans) {
  __shared__ BoundaryShared sh;
  const int c = chans ? chans[blockIdx.x] : blockIdx.x;
  const int tid = threadIdx.x;
  ChanCtl *ctl = D.ctl + c;
  if (tid == 0) {
    ChanCtl z;
    memset(&z, 0, sizeof(z));
    *ctl = z;
    sh.err = 0; sh.wl_n[0] = 0; sh.wl_n[1] = 0; sh.nfront = 1;
    const size_t off0 = (size_t)c * 2 * (size_t)D.cap;
    bool created;
    const int slot = find_or_insert(D.keys + off0, D.cap, D.log2cap, D.g.start, &created);
    D.vals[off0 + slot] = ((u64)f2o(0.0f) << 32) | kNoArc;
    D.occ[off0] = slot;
    ctl->n_occ[0] = 1;
    D.front_slot[(size_t)c * D.max_tok] = slot;
    D.toki[off0 + slot] = 0;
    if (D.g.state_info[D.g.start].y & kEpsMask) { D.worklist[(size_t)c * 2 * D.wl_cap] = slot; sh.wl_n[0] = 1; }
  }
  __syncthreads();
  u64 nZ = 0;
  const int nf = closure_and_commit(D, c, ctl, sh, 0, -1, 0, D.beam, &nZ);  // ProcessNonemitting(_config._beam)
  __syncthreads();
  if (tid == 0) {
    D.frame_off[(size_t)c * (D.max_frames + 2) + 0] = 0;
    D.frame_off[(size_t)c * (D.max_frames + 2) + 1] = nf;
    D.cutoff_hist[(size_t)c * (D.max_frames + 2) + 0] = D.beam;
    ctl->front_begin = 0;
    ctl->front_count = nf;
    ctl->cur_tab = 0;
    ctl->cnt_tok = (u64)nf;
    ctl->peak_tokens = nf;
    if (sh.err) ctl->error |= sh.err;
  }
}

// =========================================================================================
// best path: BestPathEnd + TraceBackBestPath + GetBestPath (base-inl.h:1071-1200), one wave per
// channel.  Hops are written in start->final order; hop 0 is the root token's (0,0,One) arc.
// =========================================================================================
__global__ __launch_bounds__(64) void best_path_kernel(DecoderDev D, const int32_t *chans, int use_final, int cap,
                                                       int32_t *o_il, int32_t *o_ol, float *o_g, float *o_ac,
                                                       int32_t *n_hops) {
  const int bi = blockIdx.x;
  const int c = chans ? chans[bi] : bi;
  const int lane = threadIdx.x;
  const ChanCtl *ctl = D.ctl + c;
  const int n = ctl->front_count, nd = ctl->n_decoded;
  if (nd <= 0 || n == 0) {  // base-inl.h:1104-1108 / 1148-1154: no path
    if (lane == 0) n_hops[bi] = 0;
    return;
  }
  const int4 *tok = D.tok + (size_t)c * D.arena_cap;
  const int fb = ctl->front_begin;
  u64 best_all = ~0ull, best_fin = ~0ull;
  for (int i = lane; i < n; i += 64) {
    const int4 t = tok[fb + i];
    const u64 v = ((u64)f2o(__int_as_float(t.y)) << 32) | (uint32_t)(fb + i);
    best_all = v < best_all ? v : best_all;
    if (t.x == D.g.final_state) best_fin = v < best_fin ? v : best_fin;  // IsFinal, optimize-fst.h:189-192
  }
  best_all = wave_min_u64(best_all);
  best_fin = wave_min_u64(best_fin);
  if (lane != 0) return;
  const u64 best = (use_final && best_fin != ~0ull) ? best_fin : best_all;
  const int best_t = (int)(uint32_t)best;

  int len = 0;
  for (int t = best_t; t >= 0; t = tok[t].z) ++len;
  n_hops[bi] = len;
  if (len > cap) return;
  int32_t *il = o_il + (size_t)bi * cap, *ol = o_ol + (size_t)bi * cap;
  float *og = o_g + (size_t)bi * cap, *oa = o_ac + (size_t)bi * cap;
  const float *cut = D.cutoff_hist + (size_t)c * (D.max_frames + 2);
  const float *ll = D.ll_base[c];
  // forward links of frame f have met PruneForwardLinks iff a PruneActiveTokens pass started at
  // NumFramesDecoded() = m >= f+1 (base-inl.h:660-661, 445-476) or FinalizeDecoding ran
  const int m_last = ((nd - 1) / D.prune_interval) * D.prune_interval;
  int pos = len - 1, fr = nd;
  for (int t = best_t; t >= 0; --pos) {
    const int4 T = tok[t];
    const int prev = T.z;
    if (prev < 0) {  // base-inl.h:1193-1198
      il[pos] = 0; ol[pos] = 0; og[pos] = 0.f; oa[pos] = 0.f;
    } else {
      const int4 P = tok[prev];
      const float cb = __int_as_float(P.y), ct = __int_as_float(T.y);
      const int warc = T.w;
      const bool eps = D.g.arcs[warc].x < 0;
      const int fbp = eps ? fr : fr - 1;
      const uint2 si = D.g.state_info[P.x];
      const int ne = (int)(si.y & kEpsMask);
      const int hi = eps ? (int)si.x + ne : (int)si.x + ne + (int)(si.y >> kEpsBits);
      const bool pruned_once = ctl->finalized || m_last >= fbp + 1;
      const float *llrow = ll + (size_t)(eps ? 0 : fbp) * D.stride;
      int chosen = warc;
      // TraceBackBestPath takes the FIRST link bp->tok; links are prepended in arc order
      // (base-inl.h:340-341, 1169-1186), so a surviving parallel arc of higher index shadows the
      // winning one.
      for (int a = hi - 1; a > warc; --a) {
        const int4 B = D.g.arcs[a];
        if (B.w != T.x) continue;
        const float alt_ac = eps ? 0.f : -llrow[B.x];
        const float alt_tot = eps ? cb + __int_as_float(B.z) : (cb + alt_ac) + __int_as_float(B.z);
        if (!(alt_tot < cut[fr])) continue;  // link never created
        if (pruned_once && (0.0f + (alt_tot - ct)) > D.lattice_beam) continue;  // base-inl.h:524-532
        chosen = a;
        break;
      }
      const int4 C = D.g.arcs[chosen];
      il[pos] = D.g.arc_ilabel[chosen];
      ol[pos] = C.y;
      og[pos] = __int_as_float(C.z);
      oa[pos] = eps ? 0.f : -llrow[C.x];
      if (!eps) --fr;
    }
    t = prev;
  }
}

// =========================================================================================
// launch wrappers
// =========================================================================================
static __global__ void set_finalized_kernel(DecoderDev D, const int32_t *chans, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) D.ctl[chans ? chans[i] : i].finalized = 1;
}

void launch_init(const DecoderDev &D, const int32_t *chans, int n, hipStream_t s) {
  hipLaunchKernelGGL(clear_tables_kernel, dim3(n, 16), dim3(256), 0, s, D, chans);
  hipLaunchKernelGGL(init_kernel, dim3(n), dim3(kBT), 0, s, D, chans);
}
void launch_boundary(const DecoderDev &D, const int32_t *target, int do_finalize, int do_prep, hipStream_t s) {
  hipLaunchKernelGGL(boundary_kernel, dim3(D.n_channels), dim3(kBT), 0, s, D, target, do_finalize, do_prep);
}
void launch_expand(const DecoderDev &D, int tiles_per_channel, hipStream_t s) {
  hipLaunchKernelGGL(expand_kernel, dim3(D.n_channels, tiles_per_channel), dim3(kExpandThreads), 0, s, D);
}
void launch_set_finalized(const DecoderDev &D, const int32_t *chans, int n, hipStream_t s) {
  hipLaunchKernelGGL(set_finalized_kernel, dim3((n + 255) / 256), dim3(256), 0, s, D, chans, n);
}
void launch_best_path(const DecoderDev &D, const int32_t *chans, int n, int use_final, int cap, int32_t *ilabel,
                      int32_t *olabel, float *graph, float *ac, int32_t *n_hops, hipStream_t s) {
  hipLaunchKernelGGL(best_path_kernel, dim3(n), dim3(64), 0, s, D, chans, use_final, cap, ilabel, olabel, graph, ac,
                     n_hops);
}

}  // namespace wfst
