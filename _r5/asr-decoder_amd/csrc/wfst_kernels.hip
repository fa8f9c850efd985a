// HIP kernels of the batched WFST token-passing decoder, written for gfx950 (MI355X, wave64).
//
// One frame of the reference's hot loop (AdvanceDecoding, my-decoder/online-decoder-base-inl.h:
// 649-667) for ALL channels of a batch is three launches:
//
//   expand_kernel   ProcessEmitting's inner loop (base-inl.h:311-347), load balanced: a workgroup
//                   takes a tile of 1024 frontier tokens of one channel (tiles are listed per
//                   frame, so a channel with 8x the tokens owns 8x the tiles), scans their emitting
//                   out-degrees in LDS and maps one lane to one arc, so low-degree HCLG states
//                   (2-3 arcs) still fill wavefronts.  Survivors of the (evolving) next_cutoff are counting-sorted in LDS
//                   by hash partition of their next state and appended, coalesced, to that
//                   partition's bucket in HBM: one global atomic per partition per 1024 candidates
//                   instead of two per candidate (scattered atomics run at ~26 G/s on MI355X at
//                   any scope, half the rate of plain random gathers: tools/ubench_atomics.hip).
//                   The workgroup that finishes a channel's last tile lists that channel's insert
//                   work items (plan_channel).
//   insert_kernel   FindOrAddToken (base-inl.h:88-136) as insert-or-min in an LDS hash table, one
//                   work item = a group of hash partitions of one channel: ds_cmpst on the key, ds_min_u64 on
//                   (orderable cost << 32 | arc); the winner of each state writes the 16-byte
//                   token (state, cost, backpointer, arc) straight into the arena.
//   closure_kernel  ProcessNonemitting to its fixpoint inside the kernel (base-inl.h:353-431) on
//                   a direct-mapped per-channel table of the epsilon-target states (a state's whole
//                   closure in one round where the graph upload could flatten it), then
//                   GetCutoff (base-inl.h:138-234; exact k-th smallest by LDS radix select) and the
//                   best-token seeding of next_cutoff (base-inl.h:282-300) for the next frame.
//
// Float arithmetic follows the reference's operation order exactly (compiled with
// -ffp-contract=off): tot = (cur + (-loglike)) + graph; seed = (cur + graph) - loglike.
// There is no MFMA here: the path is irregular graph traversal, bound by HBM/L2 request rate.
#include <type_traits>

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
// Workgroup barrier for an LDS-only hand-off.  __syncthreads() carries a workgroup fence that hipcc lowers to
// s_waitcnt vmcnt(0) lgkmcnt(0) before s_barrier: every global load, store and atomic of the wave is drained at each
// barrier, so a phase's bucket stores or a prefetched arc load serialise with the next phase.  Where only LDS is handed
// over, wait for the LDS operations alone and leave the global ones in flight (cdna_hip_programming.md section 5,
// "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Wave-wide scans and reductions on the DPP crossbar (gfx9 row_shr / row_bcast modifiers of v_mov_b32): seven dependent VALU
// moves for all 64 lanes, no LDS traffic and -- unlike the __shfl family, which hipcc lowers to ds_bpermute_b32 with one
// precomputed address register per shuffle distance -- no index registers: the expansion kernels held a dozen VGPRs of shuffle
// addresses across their passes, at the 128-register limit (round 4: the compacting tiles spilled on them).
// dpp_ctrl: 0x111..0x11F row_shr:1..15, 0x142 row_bcast:15, 0x143 row_bcast:31.  A lane whose source lies outside its row, or
// that the row / bank mask excludes, keeps `old` = the operation's identity.
#define WFST_DPP(old, src, ctrl, rmask, bmask) __builtin_amdgcn_update_dpp((int)(old), (int)(src), ctrl, rmask, bmask, false)
// inclusive prefix sum over the wave's 64 lanes
__device__ __forceinline__ int wave_incl_scan(int v0) {
  int v = v0 + WFST_DPP(0, v0, 0x111, 0xf, 0xf);
  v += WFST_DPP(0, v0, 0x112, 0xf, 0xf);
  v += WFST_DPP(0, v0, 0x113, 0xf, 0xf);
  v += WFST_DPP(0, v, 0x114, 0xf, 0xe);
  v += WFST_DPP(0, v, 0x118, 0xf, 0xc);
  v += WFST_DPP(0, v, 0x142, 0xa, 0xf);
  v += WFST_DPP(0, v, 0x143, 0xc, 0xf);
  return v;
}
// the wave's total, in every lane (uniform)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane(wave_incl_scan((int)x), 63); }
__device__ __forceinline__ float wave_min_f(float x) {
  const int id = 0x7F800000;   // +inf
  int v0 = __float_as_int(x);
  auto mn = [](int a, int b) { return __float_as_int(fminf(__int_as_float(a), __int_as_float(b))); };
  int v = mn(v0, WFST_DPP(id, v0, 0x111, 0xf, 0xf));
  v = mn(v, WFST_DPP(id, v0, 0x112, 0xf, 0xf));
  v = mn(v, WFST_DPP(id, v0, 0x113, 0xf, 0xf));
  v = mn(v, WFST_DPP(id, v, 0x114, 0xf, 0xe));
  v = mn(v, WFST_DPP(id, v, 0x118, 0xf, 0xc));
  v = mn(v, WFST_DPP(id, v, 0x142, 0xa, 0xf));
  v = mn(v, WFST_DPP(id, v, 0x143, 0xc, 0xf));
  return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
__device__ __forceinline__ u64 wave_min_u64(u64 x) {
  uint32_t lo0 = (uint32_t)x, hi0 = (uint32_t)(x >> 32), lo = lo0, hi = hi0;
  auto step = [&](uint32_t slo, uint32_t shi, int ctrl_sel) {
    uint32_t tl, th;
    switch (ctrl_sel) {   // (dpp_ctrl and the masks are instruction immediates)
      case 0: tl = (uint32_t)WFST_DPP(-1, slo, 0x111, 0xf, 0xf); th = (uint32_t)WFST_DPP(-1, shi, 0x111, 0xf, 0xf); break;
      case 1: tl = (uint32_t)WFST_DPP(-1, slo, 0x112, 0xf, 0xf); th = (uint32_t)WFST_DPP(-1, shi, 0x112, 0xf, 0xf); break;
      case 2: tl = (uint32_t)WFST_DPP(-1, slo, 0x113, 0xf, 0xf); th = (uint32_t)WFST_DPP(-1, shi, 0x113, 0xf, 0xf); break;
      case 3: tl = (uint32_t)WFST_DPP(-1, slo, 0x114, 0xf, 0xe); th = (uint32_t)WFST_DPP(-1, shi, 0x114, 0xf, 0xe); break;
      case 4: tl = (uint32_t)WFST_DPP(-1, slo, 0x118, 0xf, 0xc); th = (uint32_t)WFST_DPP(-1, shi, 0x118, 0xf, 0xc); break;
      case 5: tl = (uint32_t)WFST_DPP(-1, slo, 0x142, 0xa, 0xf); th = (uint32_t)WFST_DPP(-1, shi, 0x142, 0xa, 0xf); break;
      default: tl = (uint32_t)WFST_DPP(-1, slo, 0x143, 0xc, 0xf); th = (uint32_t)WFST_DPP(-1, shi, 0x143, 0xc, 0xf); break;
    }
    const u64 t = ((u64)th << 32) | tl, a = ((u64)hi << 32) | lo, m = t < a ? t : a;
    lo = (uint32_t)m; hi = (uint32_t)(m >> 32);
  };
  step(lo0, hi0, 0); step(lo0, hi0, 1); step(lo0, hi0, 2);
  step(lo, hi, 3); step(lo, hi, 4); step(lo, hi, 5); step(lo, hi, 6);
  return ((u64)(uint32_t)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
}
__device__ __forceinline__ u64 wave_sum_u64(u64 v) {   // (the cold kernels' sums of wide counters)
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ int lane_rank(u64 mask) {  // active lanes below this one
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}
__device__ __forceinline__ uint32_t hash32(int32_t s) { return (uint32_t)s * 2654435761u; }
// partition = top log2part bits of the hash, LDS slot = the next log2lds bits
__device__ __forceinline__ int part_of(uint32_t h, int log2part) { return log2part ? (int)(h >> (32 - log2part)) : 0; }
__device__ __forceinline__ uint32_t lds_slot_of(uint32_t h, int log2part, int log2lds) {
  return (h >> (32 - log2part - log2lds)) & ((1u << log2lds) - 1u);
}

// =========================================================================================
// biglm (BASELINE configs[3]): on-the-fly LM rescoring.  A token's identity is (graph row, LM pair
// state) -- the reference's PairId (my-decoder/online-decoder-mempool-base-biglm.h:77-90) -- and every
// arc with an output label adds cost_newLM(word | history) - cost_oldLM(word | history), each LM
// walked from its OWN state (DiffArpaLm::GetArc with the pair's components, newlm/diff-lm.h:63-111;
// the reference text hands the pair id to both LMs, :80,86 -- oracle/wfst_oracle.c, "fixed" mode).
// =========================================================================================
__device__ __forceinline__ uint32_t hash_big(int32_t row, int32_t pair) {
  uint32_t h = (uint32_t)row * 2654435761u;
  h ^= ((uint32_t)pair + 0x9E3779B9u) * 0x85EBCA6Bu;
  h ^= h >> 15;
  return h * 0x2C1B3C6Du;
}
__device__ __forceinline__ u64 big_key(int32_t row, int32_t pair) { return (u64)(uint32_t)row | ((u64)(uint32_t)pair << 32); }

// (Fsa::GetArc / ComposeArpaLm::GetArc / ComposeArpaLm::Final on the device: fsa_getarc, lm_getarc, lm_final_cost in wfst_device.h)
__device__ __forceinline__ uint32_t hash_pair(int s1, int s2) {
  uint32_t h = (uint32_t)s1 * 7853u + (uint32_t)s2;  // PairHasher, util/stl-util.h:8-17 ...
  h *= 2654435761u;                                  // ... scrambled for an open-addressed table
  return h ^ (h >> 15);
}
// DiffArpaLm's _state_map.insert + _state_vec.push_back (newlm/diff-lm.h:92-103): the id of the pair
// (old-LM state, new-LM state), allocating it if new.  One open-addressed array per channel, id =
// slot; all accesses are agent-scope atomics (workgroups on other XCDs intern into the same table).
__device__ __forceinline__ int pair_intern(const DecoderDev &D, int c, ChanCtl *ctl, int s1, int s2) {
  u64 *keys = D.pair_keys + (size_t)c * D.pair_cap;
  const u64 key = (u64)(uint32_t)s1 | ((u64)(uint32_t)s2 << 32);
  const uint32_t mask = (uint32_t)D.pair_cap - 1u;
  uint32_t slot = hash_pair(s1, s2) & mask;
  for (int q = 0; q < D.pair_cap; ++q) {
    u64 k = ld_agent(&keys[slot]);
    if (k == kEmptyVal) {
      k = atomicCAS(&keys[slot], kEmptyVal, key);
      if (k == kEmptyVal) {
        // (the claimed slots are listed: the next InitDecoding empties exactly those -- clear_pairs_kernel)
        const int nth = atomicAdd(&ctl->pair_count, 1);
        if (nth >= (D.pair_cap >> 2) * 3) atomicOr(&ctl->error, kErrPairsFull);
        else D.pair_list[(size_t)c * D.pair_cap + nth] = (int32_t)slot;
        return (int)slot;
      }
    }
    if (k == key) return (int)slot;
    slot = (slot + 1) & mask;
  }
  atomicOr(&ctl->error, kErrPairsFull);
  return 0;
}
__device__ __forceinline__ int pair_find(const DecoderDev &D, int c, int s1, int s2) {  // -1: never interned
  const u64 *keys = D.pair_keys + (size_t)c * D.pair_cap;
  const u64 key = (u64)(uint32_t)s1 | ((u64)(uint32_t)s2 << 32);
  const uint32_t mask = (uint32_t)D.pair_cap - 1u;
  uint32_t slot = hash_pair(s1, s2) & mask;
  for (int q = 0; q < D.pair_cap; ++q) {
    const u64 k = ld_agent(&keys[slot]);
    if (k == key) return (int)slot;
    if (k == kEmptyVal) return -1;
    slot = (slot + 1) & mask;
  }
  return -1;
}
// NextLmState (biglm.h:54-70) for a non-epsilon output label: lm_score = Times(w_old, w_new).Value1();
// the LM states reached are returned for the caller to intern (or not: the next_cutoff seed and the
// traceback only need the score).
// The two LMs are walked in LOCKSTEP: a back-off level of either is one round trip -- the probe of its (state, word) table and,
// speculatively, the state's back-off record are asked for together, for both LMs at once -- instead of the old LM's whole
// chain, then the new one's (an LM step sits on the critical path of every round of the expansion and of the closure pass:
// the wavefront waits for its slowest lane's chain).  Same look-ups, same float sums (ComposeArpaLm::GetArc, lm_getarc).
__device__ __forceinline__ float lm_step_pk(const DecoderDev &D, u64 pk, int olabel, int *n1, int *n2) {   // pk: the pair's two LM states (pair_keys)
  int s0 = (int)(uint32_t)pk, s1 = (int)(uint32_t)(pk >> 32);
  float a0 = 0.0f, a1 = 0.0f;   // sums of the back-off weights so far
  float r0 = 0.0f, r1 = 0.0f;   // the results: -(back-offs + arc weight)
  int p0 = 0, p1 = 0;           // linear-probe offsets
  bool d0 = false, d1 = false;
  // kLmProbe consecutive slots of the (state, word) table per round trip (64 bytes, one or two lines): a probe sequence -- the
  // unsuccessful ones of a back-off above all -- ends within them nearly always, so a back-off level costs ONE trip whatever the
  // clustering (the wavefront waits for its slowest lane: the tail of the probe lengths was the tail of the launch)
  constexpr int kLmProbe = 4;
  while (!(d0 && d1)) {
    int4 e0[kLmProbe], e1[kLmProbe], b0 = make_int4(0, 0, 0, 0), b1 = b0;
    int2 x0 = make_int2(0, 0), x1 = x0;
#pragma unroll
    for (int q = 0; q < kLmProbe; ++q) e0[q] = e1[q] = make_int4(0, 0, 0, 0);
    if (!d0) {
      if (s0 == 0) x0 = D.lm_old.wt[olabel];
      else {
        const uint32_t h = lm_hash(s0, olabel) + (uint32_t)p0;
#pragma unroll
        for (int q = 0; q < kLmProbe; ++q) e0[q] = D.lm_old.hash[(h + (uint32_t)q) & D.lm_old.hmask];
        b0 = D.lm_old.st[s0];
      }
    }
    if (!d1) {
      if (s1 == 0) x1 = D.lm_new.wt[olabel];
      else {
        const uint32_t h = lm_hash(s1, olabel) + (uint32_t)p1;
#pragma unroll
        for (int q = 0; q < kLmProbe; ++q) e1[q] = D.lm_new.hash[(h + (uint32_t)q) & D.lm_new.hmask];
        b1 = D.lm_new.st[s1];
      }
    }
    if (!d0) {
      if (s0 == 0) { r0 = -1 * (a0 + __int_as_float(x0.x)); *n1 = x0.y; d0 = true; }
      else {
        bool decided = false;
#pragma unroll
        for (int q = 0; q < kLmProbe; ++q) {
          if (decided) continue;
          if (e0[q].x == s0 && e0[q].y == olabel) { r0 = -1 * (a0 + __int_as_float(e0[q].z)); *n1 = e0[q].w; d0 = true; decided = true; }
          else if (e0[q].x < 0) { a0 += __int_as_float(b0.z); s0 = b0.w; p0 = 0; decided = true; }   // no such arc: back off (compose-arpalm.cc:58-64)
        }
        if (!decided) p0 += kLmProbe;
      }
    }
    if (!d1) {
      if (s1 == 0) { r1 = -1 * (a1 + __int_as_float(x1.x)); *n2 = x1.y; d1 = true; }
      else {
        bool decided = false;
#pragma unroll
        for (int q = 0; q < kLmProbe; ++q) {
          if (decided) continue;
          if (e1[q].x == s1 && e1[q].y == olabel) { r1 = -1 * (a1 + __int_as_float(e1[q].z)); *n2 = e1[q].w; d1 = true; decided = true; }
          else if (e1[q].x < 0) { a1 += __int_as_float(b1.z); s1 = b1.w; p1 = 0; decided = true; }
        }
        if (!decided) p1 += kLmProbe;
      }
    }
  }
  return r0 + r1;
}
__device__ __forceinline__ float lm_step(const DecoderDev &D, int c, int pair, int olabel, int *n1, int *n2) {
  return lm_step_pk(D, ld_agent(&D.pair_keys[(size_t)c * D.pair_cap + pair]), olabel, n1, n2);
}

// debug phase timers (WFST_DBG & 32): slot k accumulates {sum, max, count} of 100 MHz ticks
__device__ __forceinline__ void dbg_phase(const DecoderDev &D, int k, unsigned long long &t_prev) {
  if (!(D.dbg & (k >= 22 ? 64 : k >= 11 ? 128 : k >= 6 ? 64 : 32))) return;   // (slots 22..: insert again)
  if (k >= 6 && (blockIdx.x & 15) != 0) return;  // sample 1/16 of the insert / expand workgroups
  const unsigned long long now = wall_clock64();
  const unsigned long long dt = now - t_prev;
  atomicAdd(&D.dbg_t[3 * k], dt);
  atomicMax(&D.dbg_t[3 * k + 1], dt);
  atomicAdd(&D.dbg_t[3 * k + 2], 1ull);
  t_prev = now;
}

constexpr int kEmitChunk = 16;   // entries of the channel's emitter list an insert item reserves up front (lattice mode on the fused rows)
constexpr int kHeavyItem = 900;  // records: insert items above this are handed out first

// Which buckets share a workgroup?  Partitions of a light channel hold a few dozen records each;
// the largest aligned group of 64 / 32 / ... / 2 partitions whose records fit ONE pass of a small LDS
// table becomes one work item, so a light channel costs 1-4 items while a heavy channel keeps all
// of its partitions separate.  (Halving steps: with 64/16/4 a typical channel sat just above a
// threshold and was cut four times finer than needed.  Greedy contiguous runs filled to joint_max
// were tried and are slower: ~700 full items per frame balance worse over 768 workgroups than
// ~1300 half-full ones.)  Returns the group of partition p: {first partition, size, records}.
__device__ __forceinline__ void partition_group(int P, int joint_max, int p, int ps /*inclusive prefix of counts, per lane*/,
                                                int cnt_p, int *g0, int *G, int *n) {
  *G = 1; *g0 = p; *n = cnt_p;
  for (int cand = P; cand >= 2; cand >>= 1) {
    const int s0 = p & ~(cand - 1);
    const int tot = __shfl(ps, s0 + cand - 1, 64) - (s0 ? __shfl(ps, s0 - 1, 64) : 0);
    if (tot <= joint_max) { *G = cand; *g0 = s0; *n = tot; return; }
  }
}

// An insert work item without records (group size 0) for channel c: listed where a channel of a two-launch frame has nothing
// to insert, so that an insert workgroup still closes its frame (frame_boundary_fused).
__device__ __forceinline__ void push_empty_item(const DecoderDev &D, int c, int group, int par) {
  FrameCtl *fc = D.fctl + group;
  const int idx = atomicAdd(&fc->n_small[par], 1);
  if (idx < D.item_cap / 2) D.items[(size_t)group * D.item_cap + D.item_cap - 1 - idx] = (int)((uint32_t)c << 16);
  else atomicOr(&D.ctl[c].error, kErrBucketFull);   // cannot happen: the list holds channels x partitions
}

// plan_channel: one wave lists the insert work items of channel c for this frame: item =
// channel << 16 | first partition << 8 | group size.  Run by the workgroup that finishes the
// channel's last expansion tile (all bucket counters of the channel are final then; they are only
// ever touched by device-scope atomics, so this wave reads them with atomic loads).
__device__ __forceinline__ void plan_channel(const DecoderDev &D, int c, int group, int par) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));   // (opaque: the addresses derived from it are computed here, not hoisted to the kernel's entry and spilled)
  FrameCtl *fc = D.fctl + group;
  const int P = D.n_part;
  const int32_t *cnts = D.bucket_cnt + (size_t)c * P;
  const int cnt = (lane < P) ? min(ld_agent(&cnts[lane]), D.bucket_cap) : 0;
  int ps = cnt;
  ps = wave_incl_scan(ps);
  int g0, G, n;
  partition_group(P, min(D.joint_max, (D.lds_slots * 3) >> 2), lane < P ? lane : 0, ps, cnt, &g0, &G, &n);
  const bool leader = lane < P && g0 == lane && n > 0;
  const u64 m = __ballot(leader);
  const int n_rec = __builtin_amdgcn_readlane(ps, 63);   // the channel's candidate records
  if (D.two_launch && lane == 0) {   // (read by the insert launch: frame_boundary_fused)
    // Can GetCutoff of the frame being built need a look at its tokens (frame_boundary_fused)?  Only with more tokens than
    // max_active / the per-frame limit -- and a frame has at most as many tokens as candidate records --, or with a min_active
    // behind a frame whose adaptive beam was not the plain beam.  Only then do the channel's insert items store their tokens
    // write-through, drain and count themselves out of stores_left (kRiskyBit rides above the item count).
    const int max_eff = D.soft_limit ? min(D.max_active, D.max_tok) : D.max_active;
    const bool risky = n_rec > max_eff || (D.min_active > 0 && !(D.ctl[c].adaptive_beam == D.beam));
    const int k = m ? __popcll(m) : 1;
    D.ctl[c].stores_left = k;
    D.ctl[c].items_left = k | (risky ? kRiskyBit : 0);
  }
  if (!m) {
    // no candidate survived: nothing to insert -- but with two launches per frame SOME insert workgroup has to close the
    // channel's frame: an empty item (group size 0)
    if (D.two_launch && lane == 0) push_empty_item(D, c, group, par);
    return;
  }
  // heavy items are listed from the front of items[], light ones from the back; the insert
  // workgroups walk heavy first
  const bool heavy = leader && n > kHeavyItem;
  const u64 mh = __ballot(heavy), ml = m & ~mh;
  int bh = 0, bl = 0;
  if (lane == 0) {
    if (mh) bh = atomicAdd(&fc->n_items[par], __popcll(mh));
    if (ml) bl = atomicAdd(&fc->n_small[par], __popcll(ml));
  }
  bh = __builtin_amdgcn_readfirstlane(bh);
  bl = __builtin_amdgcn_readfirstlane(bl);
  int slot = -1;   // the item's place in items[]
  if (leader) {
    const int v = (int)(((uint32_t)c << 16) | ((uint32_t)g0 << 8) | (uint32_t)G);  // c <= 32767 (wfst_decoder_create_ex)
    int32_t *items = D.items + (size_t)group * D.item_cap;
    if (heavy) {
      const int idx = bh + lane_rank(mh);
      if (idx < D.item_cap / 2) items[slot = idx] = v;
      else atomicOr(&D.ctl[c].error, kErrBucketFull);  // cannot happen: each half holds channels x partitions
    } else {
      const int idx = bl + lane_rank(ml);
      if (idx < D.item_cap / 2) items[slot = D.item_cap - 1 - idx] = v;
      else atomicOr(&D.ctl[c].error, kErrBucketFull);
    }
  }
  // the item's record prefix over its buckets, beside it (DecoderDev::item_pref): every lane is a bucket of exactly one group
  {
    const int gslot = __shfl(slot, g0, 64);
    const int below = __shfl(ps, max(g0 - 1, 0), 64);   // (every lane takes part in the shuffle)
    const int gbase = g0 ? below : 0;
    if (lane < P && gslot >= 0) D.item_pref[((size_t)group * D.item_cap + gslot) * 64 + (lane - g0)] = ps - gbase;
  }
}

// The end of an expansion tile (expand_body, expand_kernel_staged): the tile's work counters to the channel's control block,
// the countdown of the channel's tiles, and -- by the workgroup that finishes the channel's LAST tile -- the plan of its insert
// work items.  s_stat[4]: the workgroup's LDS accumulators {N, E, records, Z} (zero on entry; zero again on exit).
__device__ __forceinline__ void tile_tail(const DecoderDev &D, int c, ChanCtl *ctl, int group, int par, uint32_t nN, uint32_t nE,
                                          uint32_t nR, uint32_t nZf, uint32_t *s_stat) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));   // (opaque: see plan_channel)
  const int wave = threadIdx.x >> 6;
  // work counters: summed over the workgroup in LDS, then one set of atomics per tile from a wave that has nothing else to
  // wait for -- every atomic on the channel's control line queues behind the other tiles' (next_cutoff, the countdown)
  {
    const uint32_t wN = wave_sum_u32(nN), wE = wave_sum_u32(nE), wZ = wave_sum_u32(nZf);
    if (lane == 0) {
      if (wN) atomicAdd(&s_stat[0], wN);
      if (wE) atomicAdd(&s_stat[1], wE);
      if (nR) atomicAdd(&s_stat[2], nR);
      if (wZ) atomicAdd(&s_stat[3], wZ);
    }
  }
  // the channel's last tile plans its insert work items.  Every wave's bucket atomics have returned (their results were
  // used above), so an LDS-only barrier orders them before the countdown; only wave 0 waits for the countdown's answer
  lds_barrier();
  if (wave == 1 && lane < 4) {
    const uint32_t v = s_stat[lane];
    s_stat[lane] = 0;
    u64 *dst = lane == 0 ? &ctl->cnt_N : lane == 1 ? &ctl->cnt_E : lane == 2 ? &ctl->cnt_rec : &ctl->cnt_Z;
    if (v) atomicAdd(dst, (u64)v);   // (cnt_Z: closure paths priced, per candidate, not per token as the reference counts)
  }
  if (wave == 0) {
    int last = 0;
    if (lane == 0) last = atomicSub(&ctl->tiles_left, 1) == 1;
    last = __builtin_amdgcn_readfirstlane(last);
    if (last) plan_channel(D, c, group, par);
  }
}

// =========================================================================================
// expand_kernel: a fixed grid of 512-thread workgroups; workgroup w takes tile w of the frame's tile
// list, further tiles by ticket.
// =========================================================================================
constexpr int kExpandThreads = 256;
constexpr int kCandPerThread = 2;
constexpr int kChunk = kExpandThreads * kCandPerThread;  // candidates per counting-sort round
constexpr int kTokPerThread = 2;
constexpr int kTileTokens = kExpandThreads * kTokPerThread;  // frontier tokens per tile
constexpr int kLog2TileTokens = 9;
static_assert(kTileTokens == 1 << kLog2TileTokens, "the owner search of expand_body takes exactly log2(kTileTokens) steps");

// Work unit = one tile of kTileTokens frontier tokens of one channel.  prep_frame lists the tiles of
// all active channels (TileDesc); workgroup w takes tile w, then further tiles from a ticket
// counter, so a channel with 8x the tokens simply owns 8x the tiles (per-frame token counts are
// heavy-tailed across a batch; a fixed share of workgroups per channel made every frame wait for
// the heaviest one).
// kBig = biglm mode (the plain instantiation carries none of it).  (The fused rows -- pseudo arcs, degree codes, seed tiles --
// are the staged kernel's below; this one serves graphs whose closures cannot be folded, and biglm decoders.)
// kTimers: the phase timers of wfst_options.debug & 128 (their own instantiation: the production kernel carries neither
// the clock reads nor their registers -- it sits at the 80-VGPR limit, where every live value more is a spill).
template <bool kBig, bool kTimers = false>
__device__ __forceinline__ void expand_body(const DecoderDev &D, int group, int par) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  FrameCtl *fc = D.fctl + group;
  // (the workgroup's first tile descriptor is read together with the tile count, not after it: one round trip less at
  // the head of every launch; a slot beyond the count holds a stale descriptor that is not used)
  const TileDesc td_first = D.tiles[(size_t)group * D.tile_cap + min((int)blockIdx.x, D.tile_cap - 1)];
  const int total_tiles = fc->total_tiles[par];
  const float kInf = __builtin_huge_valf();
  const int P = D.n_part, log2part = D.log2part, bcap = D.bucket_cap;

  __shared__ int s_base[kTileTokens + 1];
  __shared__ int s_arcbeg[kTileTokens];
  __shared__ float s_cost[kTileTokens];
  __shared__ int s_wsum[kExpandThreads / 64];
  __shared__ int s_cnt[64], s_lbase[65], s_gbase[64];
  __shared__ int4 s_rec[kChunk];
  __shared__ int s_lm[kBig ? kTileTokens : 1];    // biglm: LM pair state of each token of the tile
  __shared__ u64 s_pk[kBig ? kTileTokens : 1];    //        and the pair's two LM states (asked for with the row headers: one round trip off every LM step)
  __shared__ int s_rec_lm[kBig ? kChunk : 1];     //        and of each sorted candidate
  __shared__ int s_ticket;

  __shared__ uint32_t s_stat[4];   // work counters of the tile, summed over the workgroup
  if (blockIdx.x == 0 && tid == 0) {  // lists of the previous step are consumed
    fc->total_tiles[par ^ 1] = 0;
    fc->ticket[par ^ 1] = 0;
    fc->n_items[par ^ 1] = 0;
    fc->n_small[par ^ 1] = 0;
    fc->item_ticket[par ^ 1] = 0;
  }
  const TileDesc *tiles = D.tiles + (size_t)group * D.tile_cap;
  if (tid < 4) s_stat[tid] = 0;   // (barriers follow before the first use)
  unsigned long long tq = kTimers ? wall_clock64() : 0ull;
  for (int t = blockIdx.x; t < total_tiles;) {
    const TileDesc td = t == (int)blockIdx.x ? td_first : tiles[t];
    const int c = td.chan;
    ChanCtl *ctl = D.ctl + c;
    // next_cutoff as it stands (it only ever tightens: a stale value prunes less, never wrongly); asked for here,
    // needed after the scan
    const uint32_t bound0 = ld_agent(&ctl->bound);
    const int n = td.tok_count;
    const int fbegin = td.tok_begin;
    const int4 *tok = D.tok + (size_t)c * D.arena_cap + fbegin;
    const float cutoff = td.cutoff, ab = td.adaptive_beam;
    const float *llrow = td.llrow;
    int4 *bucket = D.bucket + (size_t)c * P * bcap;
    int32_t *bucket_lm = kBig ? D.bucket_lm + (size_t)c * P * bcap : nullptr;
    int32_t *bucket_cnt = D.bucket_cnt + (size_t)c * P;
    uint32_t nN = 0, nE = 0, nR = 0;   // per thread and tile: 32 bits are plenty (and four registers less at the 80-VGPR limit)
    {
    // two adjacent frontier tokens per thread (a tile is 1024 tokens, so the tiles of a whole
    // batch fit the chip's resident workgroup slots in one wave)
    int deg[kTokPerThread], arcbeg[kTokPerThread];
    float cost[kTokPerThread];
    int4 tk[kTokPerThread];
#pragma unroll
    for (int j = 0; j < kTokPerThread; ++j) {
      const int i = tid * kTokPerThread + j;
      tk[j] = i < n ? tok[i] : make_int4(0, 0x7F800000, 0, 0);
      if constexpr (kBig) s_lm[i] = i < n ? D.tok_lm[(size_t)c * D.arena_cap + fbegin + i] : 0;
    }
    u64 pkv[kBig ? kTokPerThread : 1];
    if constexpr (kBig) {
#pragma unroll
      for (int j = 0; j < kTokPerThread; ++j) {
        const int i = tid * kTokPerThread + j;
        pkv[j] = i < n ? ld_agent(&D.pair_keys[(size_t)c * D.pair_cap + s_lm[i]]) : 0ull;   // (in flight with the header loads below)
      }
    }
#pragma unroll
    for (int j = 0; j < kTokPerThread; ++j) {
      const int i = tid * kTokPerThread + j;
      deg[j] = 0; arcbeg[j] = 0;
      cost[j] = __int_as_float(tk[j].y);
      int nem = 0;
      if (i < n && cost[j] <= cutoff) {  // base-inl.h:315
        const uint32_t dw = (uint32_t)D.g.arcs[tk[j].x].x;  // row header: (n_emit << 12) | n_eps
        nem = deg[j] = (int)(dw >> kEpsBits);
        arcbeg[j] = tk[j].x + 1 + (int)(dw & kEpsMask);
        nN++;
        nE += nem;
      }
      if constexpr (kBig) s_pk[i] = pkv[j];
    }
    int tsum = 0;
#pragma unroll
    for (int j = 0; j < kTokPerThread; ++j) tsum += deg[j];
    int incl = tsum;
    incl = wave_incl_scan(incl);
    if (lane == 63) s_wsum[wave] = incl;
#pragma unroll
    for (int j = 0; j < kTokPerThread; ++j) {
      s_cost[tid * kTokPerThread + j] = cost[j];
      s_arcbeg[tid * kTokPerThread + j] = arcbeg[j];
    }
    __syncthreads();
    if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 11, tq); }
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kExpandThreads / 64; ++w) {
      int v = s_wsum[w];
      if (w < wave) wbase += v;
      total += v;
    }
    {
      int run = wbase + incl - tsum;
#pragma unroll
      for (int j = 0; j < kTokPerThread; ++j) { s_base[tid * kTokPerThread + j] = run; run += deg[j]; }
    }
    if (tid == 0) s_base[kTileTokens] = total;
    if (tid < 64) s_cnt[tid] = 0;
    __syncthreads();

    float bound = o2f(bound0);
    const int tok0 = fbegin;  // arena index of the tile's first token
    for (int j0 = 0; j0 < total; j0 += kChunk) {
      int4 rec[kCandPerThread];
      int rec_lm[kCandPerThread];
      float tot[kCandPerThread];
      float tmin = kInf;
      // The candidates of a thread are taken through the stages TOGETHER -- owner search, arc load, second-slot
      // and log-likelihood load, pricing -- so that their memory round trips overlap.  (Written as one loop per
      // candidate the compiler keeps the candidates apart: search, arc, wait, log-likelihood, wait, then the
      // next candidate, i.e. twice as many serial HBM latencies per round.)  A lane without a candidate runs the
      // loads on slot 0 / column 0 and drops the result.
      int jv[kCandPerThread], lo[kCandPerThread], av[kCandPerThread];
      {
        int hi[kCandPerThread];
#pragma unroll
        for (int k = 0; k < kCandPerThread; ++k) {
          const int j = j0 + k * kExpandThreads + tid;
          jv[k] = j < total ? j : -1;
          lo[k] = 0; hi[k] = kTileTokens;  // s_base[lo] <= j < s_base[hi]
        }
#pragma unroll
        for (int step = 0; step < kLog2TileTokens; ++step) {
#pragma unroll
          for (int k = 0; k < kCandPerThread; ++k) {
            const int mid = (lo[k] + hi[k]) >> 1;
            if (s_base[mid] <= max(jv[k], 0)) lo[k] = mid; else hi[k] = mid;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < kCandPerThread; ++k) {
        const int off = max(jv[k], 0) - s_base[lo[k]];
        av[k] = s_arcbeg[lo[k]] + off;
        if (jv[k] < 0) av[k] = 0;
      }
      int4 arcv[kCandPerThread];
      int olv[kBig ? kCandPerThread : 1];
      float llv[kCandPerThread];
      // next_cutoff as it stands NOW, asked for with the arcs (it arrives with them): what the seed tile and the other tiles
      // of the channel have tightened since this tile's last look
      const uint32_t bfresh = ld_agent(&ctl->bound);
#pragma unroll
      for (int k = 0; k < kCandPerThread; ++k) arcv[k] = D.g.arcs[av[k]];
      if constexpr (kBig) {
#pragma unroll
        for (int k = 0; k < kCandPerThread; ++k) olv[k] = D.g.arc_olabel[av[k]];
      }
#pragma unroll
      for (int k = 0; k < kCandPerThread; ++k) llv[k] = llrow[jv[k] >= 0 ? (arcv[k].x & D.g.col_mask) : 0];
#pragma unroll
      for (int k = 0; k < kCandPerThread; ++k) {
        tot[k] = kInf;
        rec_lm[k] = 0;
        if (jv[k] >= 0) {
          const int4 arc = arcv[k];
          const int a = av[k];
          float graph_cost = __int_as_float(arc.z);
          if constexpr (kBig) {  // biglm.h:377-388: graph_cost = arc weight + lm_score, next LM state into the key
            const int ol = olv[k];
            float lm_score = 0.0f;
            rec_lm[k] = s_lm[lo[k]];
            if (ol != 0) {
              int n1, n2;
              lm_score = lm_step_pk(D, s_pk[lo[k]], ol, &n1, &n2);
              rec_lm[k] = pair_intern(D, c, ctl, n1, n2);
            }
            graph_cost = __int_as_float(arc.z) + lm_score;
          }
          const float ac_cost = -llv[k];                                  // base-inl.h:326
          tot[k] = (s_cost[lo[k]] + ac_cost) + graph_cost;                // base-inl.h:329
          rec[k] = make_int4(arc.w, __float_as_int(tot[k]), tok0 + lo[k], (int)((uint32_t)a | flags_of((uint32_t)arc.y)));
          tmin = fminf(tmin, tot[k]);
        }
      }
      if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 12, tq); }
      bound = fminf(bound, o2f(bfresh));
      // base-inl.h:330-333: tighten next_cutoff by the best candidate seen (wave-aggregated)
      const float cand = wave_min_f(tmin) + ab;
      if (cand < bound) {
        uint32_t old = 0;
        if (lane == 0) old = atomicMin(&ctl->bound, f2o(cand));
        old = __shfl(old, 0, 64);
        bound = fminf(o2f(old), cand);
      }
      // counting sort of the survivors by hash partition, in LDS
      int part[kCandPerThread], rank[kCandPerThread];
#pragma unroll
      for (int k = 0; k < kCandPerThread; ++k) {
        part[k] = -1;
        if (tot[k] < bound) {
          part[k] = part_of(kBig ? hash_big(rec[k].x, rec_lm[k]) : hash32(rec[k].x), log2part);
          rank[k] = atomicAdd(&s_cnt[part[k]], 1);
        }
      }
      lds_barrier();
      if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 13, tq); }
      if (tid < 64) {
        const int cnt = tid < P ? s_cnt[tid] : 0;
        int inc = cnt;
        inc = wave_incl_scan(inc);
        s_lbase[tid] = inc - cnt;
        if (tid == 63) s_lbase[64] = inc;
        int g = 0;
        if (cnt) {
          g = atomicAdd(&bucket_cnt[tid], cnt);
          if (g + cnt > bcap) atomicOr(&ctl->error, kErrBucketFull);
        }
        s_gbase[tid] = g;
        s_cnt[tid] = 0;
      }
      lds_barrier();
      if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 14, tq); }
#pragma unroll
      for (int k = 0; k < kCandPerThread; ++k)
        if (part[k] >= 0) {
          s_rec[s_lbase[part[k]] + rank[k]] = rec[k];
          if constexpr (kBig) s_rec_lm[s_lbase[part[k]] + rank[k]] = rec_lm[k];
        }
      lds_barrier();
      const int npass = s_lbase[64];
      nR += (tid == 0) ? (uint32_t)npass : 0u;
      for (int q = tid; q < npass; q += kExpandThreads) {
        const int4 r = s_rec[q];
        const int p = part_of(kBig ? hash_big(r.x, s_rec_lm[q]) : hash32(r.x), log2part);
        const int gi = s_gbase[p] + (q - s_lbase[p]);
        if (gi < bcap) {
          bucket[(size_t)p * bcap + gi] = r;
          if constexpr (kBig) bucket_lm[(size_t)p * bcap + gi] = s_rec_lm[q];
        }
      }
      lds_barrier();
      if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 15, tq); }
    }
    }
    tile_tail(D, c, ctl, group, par, nN, nE, nR, 0u, s_stat);
    // next tile: the first gridDim.x tiles are taken statically, the rest by ticket
    if (total_tiles <= (int)gridDim.x) break;
    if (tid == 0) s_ticket = (int)gridDim.x + atomicAdd(&fc->ticket[par], 1);
    __syncthreads();
    t = s_ticket;
    __syncthreads();
    if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 16, tq); }
  }
}

// The launch bounds' second number = waves per SIMD the kernel must fit.  The plain instantiation fits 80 VGPRs (6 waves)
// without spilling; the biglm one takes what it needs (124: four waves).
__global__ __launch_bounds__(kExpandThreads, 6) void expand_kernel_plain(DecoderDev D, int group, int par) { expand_body<false>(D, group, par); }
__global__ __launch_bounds__(kExpandThreads) void expand_kernel_biglm(DecoderDev D, int group, int par) { expand_body<true>(D, group, par); }
__global__ __launch_bounds__(kExpandThreads) void expand_kernel_plain_timed(DecoderDev D, int group, int par) { expand_body<false, true>(D, group, par); }
__global__ __launch_bounds__(kExpandThreads) void expand_kernel_biglm_timed(DecoderDev D, int group, int par) { expand_body<true, true>(D, group, par); }

// =========================================================================================
// expand_kernel_staged: the expansion of decoders on the fused rows (best-path and lattice, not biglm) with a tile's
// arcs STAGED IN LDS by the chip's gather DMA (global_load_lds_dwordx4 with per-lane source addresses, MI355X_MICROARCH
// "Indexed rows: gather into LDS").
//
// expand_body keeps a thread's candidates in registers, two at a time, so a 512-token tile is four to five ROUNDS of
// {arc load -> log-likelihood load -> bucket atomic -> write}: ~19 dependent memory round trips per tile, and the launch
// lasts as long as its slowest tile's chain.  Here every row slot of the tile (emitting arcs; both slots of the pseudo arcs)
// is one lane's 16-byte DMA into an LDS image of the tile -- 64 slots per wave instruction, all of a tile's instructions
// issued back to back, no register per slot -- then one 4-byte DMA per slot fetches its log-likelihood, and the whole
// tile is priced, pruned, ranked by hash partition (LDS atomics) and written to the buckets in ONE pass: tokens -> arcs ->
// log-likelihoods -> bucket atomics -> stores, five round trips whatever the tile holds, one global atomic per partition
// per TILE.  The records go to their bucket slots straight from the LDS image (rank within the tile's share of the
// bucket): neighbouring lanes write to different partitions, the lines fill in L2.
// A tile is kStTokens = 256 frontier tokens (one per thread); its slots beyond kStSlots are staged in further passes.
// LDS: 24 KB slots + 6 KB log-likelihoods + 4 KB per-token scan = 34.5 KB: four workgroups per CU.
// Arithmetic, pruning and record layout are those of the round-2 expansion of the fused rows (retired in round 4), to the bit.
// =========================================================================================
constexpr int kStThreads = 256;
constexpr int kStTokens = 256;   // (tiles of 128 / 192 / 320 / 384 tokens measured slower by 10-25 %, 512 the same)
constexpr int kLog2StTokens = kStTokens <= 256 ? 8 : kStTokens <= 512 ? 9 : 10;   // steps of the owner search
constexpr int kStSlotsGather = 1536;   // (1392 slots at five workgroups per CU -- every tile of a half-batch launch resident at once -- measured 11 % slower: 96 VGPRs spill)   // (1280 / 1024 with five / six workgroups per CU measured slower: more tiles need a second pass)
// kRow instantiation (DecoderDev::ll_row): the tile's whole log-likelihood ROW staged in LDS beside the arcs -- one coalesced
// 16-byte DMA per four columns, asked for with the tile's tokens -- instead of one 4-byte gather per arc slot (a request each at
// the memory side: 1.9 M per frame of 128 utterances against 0.9 M row lines; the gathers were a round trip of their own
// between the arcs and the pricing).  Rows of up to kStRowFloats columns, a multiple of four, 16-byte aligned (what
// wfst_decoder_advance checks); 22 KB slots + 12 KB row + 4 KB scan = 38.5 KB: still four workgroups per CU.
constexpr int kStSlotsRow = 1408;
constexpr int kStSuper = 4;   // a compacting tile holds up to kStSuper x kStTokens frontier tokens (frame_boundary_fused / prep_frame size it)
// Tokens per compacting tile of a frame of n tokens of which about n_live lie at or below the cutoff: as many as hold ~7/8 of a
// round's worth of live ones (a tile whose live tokens exceed a round pays a second round: the whole chain of round trips again)
__device__ __forceinline__ int super_tile_tokens(int n, int n_live) {
  const long long t = (long long)(kStTokens - kStTokens / 8) * n / max(n_live, 1);
  return (int)min((long long)kStSuper * kStTokens, max((long long)kStTokens, t & ~63ll));
}
constexpr int kStRowFloats = 3072;
typedef __attribute__((address_space(3))) void *lds_void_p;
typedef const __attribute__((address_space(1))) void *gbl_void_p;

template <bool kTimers, bool kRow>
__device__ __forceinline__ void expand_staged_body(const DecoderDev &D, int group, int par) {
  constexpr int kStSlots = kRow ? kStSlotsRow : kStSlotsGather;
  constexpr int kStIter = (kStSlots + kStThreads - 1) / kStThreads;   // slots per thread and pass
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long tq = kTimers ? wall_clock64() : 0ull;
  FrameCtl *fc = D.fctl + group;
  // XCD-aware static assignment: workgroups b and b + 8 share an XCD (round-robin dispatch), the tile list holds a channel's tiles
  // side by side -- within every aligned block of 128 workgroups, XCD x takes 16 CONSECUTIVE tiles (a bijection of the block that
  // needs no count: the first descriptor is asked for with everything else), so a channel's log-likelihood row and control line
  // are fetched into two or three L2s instead of all eight
#ifndef WFST_XCD_TILES
#define WFST_XCD_TILES 3   // log2 of the run of consecutive tiles an XCD takes (0: none)
#endif
  int b0 = (int)blockIdx.x;
  {
    constexpr int kRun = WFST_XCD_TILES, kBlk = 8 << kRun;
    if (kRun && (gridDim.x & (unsigned)(kBlk - 1)) == 0u) b0 = (b0 & ~(kBlk - 1)) | ((b0 & 7) << kRun) | ((b0 >> 3) & ((1 << kRun) - 1));
  }
  const TileDesc td_first = D.tiles[(size_t)group * D.tile_cap + min(b0, D.tile_cap - 1)];
  const int total_tiles = fc->total_tiles[par];
  const float kInf = __builtin_huge_valf();
  const int P = D.n_part, log2part = D.log2part, bcap = D.bucket_cap;

  __shared__ int4 s_arc[kStSlots];     // the tile's row slots as they stand in rows[]; a priced candidate's record replaces its slot
  __shared__ float s_ll[kRow ? 1 : kStSlots];     // the log-likelihood of each slot's column (gathered form)
  __shared__ __attribute__((aligned(16))) float s_row[kRow ? kStRowFloats : 4];   // kRow: the frame's log-likelihood row of the tile's channel
  __shared__ int s_base[kStTokens + 1], s_arcbeg[kStTokens], s_nemit[kStTokens];
  __shared__ float s_cost[kStTokens];
  __shared__ int s_wsum[kStThreads / 64], s_cnt[64], s_gbase[64];
  __shared__ int s_tidx[kStTokens];   // compacting tiles: the token (index within the tile) each thread owns this round
  __shared__ uint32_t s_stat[4], s_bound;
  __shared__ u64 s_best;   // best_exp: the tile's cheapest candidate, orderable cost << 32 | row of its state
  __shared__ int s_ticket;

  if (blockIdx.x == 0 && tid == 0) {  // lists of the previous step are consumed
    fc->total_tiles[par ^ 1] = 0;
    fc->ticket[par ^ 1] = 0;
    fc->n_items[par ^ 1] = 0;
    fc->n_small[par ^ 1] = 0;
    fc->item_ticket[par ^ 1] = 0;
  }
  const TileDesc *tiles = D.tiles + (size_t)group * D.tile_cap;
  if (tid < 4) s_stat[tid] = 0;
  if (tid < 64) s_cnt[tid] = 0;
  const float *row_have = nullptr;   // kRow: the row s_row holds
  TileDesc td = td_first;
  for (int t = b0; t < total_tiles;) {
    const int c = td.chan;
    ChanCtl *ctl = D.ctl + c;
    const int n = td.tok_count;
    const int tok0 = td.tok_begin;
    const float cutoff = td.cutoff, ab = td.adaptive_beam;
    const float *llrow = td.llrow;
    int4 *bucket = D.bucket + (size_t)c * P * bcap;
    int32_t *bucket_cnt = D.bucket_cnt + (size_t)c * P;
    // (the tile's work counts -- N, E, closure paths priced -- are read back from the scan arrays at the end of a round, the records
    // written go to the workgroup's LDS accumulator as they are counted: no register of them across the passes)
    bool counted = false;
    if (n == 0) {
      // SEED TILE (DecoderDev::seed_tiles): next_cutoff's seed from the best token's emitting arcs, base-inl.h:282-300 --
      // td.tok_begin = the token's row, td.cutoff = its cost.  (bc + w) - loglike as the reference writes it (:295), then
      // + adaptive_beam: min(x) + ab == min(x + ab), float addition being monotone
      const int brow = td.tok_begin;
      const float bc = td.cutoff;
      const uint32_t hx = (uint32_t)D.g.arcs[brow].x;
      const int deg = (int)(hx >> kEpsBits), ab0 = brow + 1 + (int)(hx & kEpsMask);
      float seed = kInf;
      for (int e = tid; e < deg; e += kStThreads) {
        const int4 arc = D.g.arcs[ab0 + e];
        seed = fminf(seed, (bc + __int_as_float(arc.z)) - llrow[arc.x & D.g.col_mask]);
      }
      seed = wave_min_f(seed);
      if (lane == 0 && seed < kInf) atomicMin(&ctl->bound, f2o(seed + ab));
    } else {
      // COMPACTING TILE (tok_count > kStTokens, up to kStSuper x as many: the frame boundary cuts such tiles where a binding
      // max_active -- or the per-frame limit -- leaves most of a frame's tokens above the cutoff): the costs alone are read first,
      // the tokens at or below the cutoff numbered, and the tile expanded in ROUNDS of kStTokens of them (one round as a rule:
      // the boundary sizes the tile to the expected share of live tokens) -- a tile of 256 consecutive tokens of which 60 are
      // expanded cost the same five round trips as a full one.
      const bool super = n > kStTokens;
      const int4 *tokc = D.tok + (size_t)c * D.arena_cap + tok0;
      int n_live = min(n, kStTokens);   // (compacting tiles: known after the first round's count)
      for (int rnd = 0; rnd * kStTokens < n_live; ++rnd) {
      // ---- the round's tokens: one per thread; its row slots = emitting arcs + two per pseudo arc -------------------
      {
        int my_i = tid;   // the token's index within the tile
        if (super) {   // (uniform)
          // the tile's live tokens, numbered in token order; this round takes numbers [rnd * kStTokens, + kStTokens).  (Counted afresh
          // every round -- a second round is rare -- rather than carried in three registers across the passes.)
          float cst[kStSuper];
#pragma unroll
          for (int r = 0; r < kStSuper; ++r) {
            const int i = r * kStTokens + tid;
            cst[r] = i < n ? __int_as_float(reinterpret_cast<const int *>(tokc + i)[1]) : kInf;
          }
          uint32_t live_mask = 0;
          int cnt = 0;
#pragma unroll
          for (int r = 0; r < kStSuper; ++r) {
            if (cst[r] <= cutoff) { live_mask |= 1u << r; ++cnt; }   // base-inl.h:315
          }
          int incl = cnt;
          incl = wave_incl_scan(incl);
          if (lane == 63) s_wsum[wave] = incl;
          __syncthreads();
          int wbase = 0, tot = 0;
#pragma unroll
          for (int w = 0; w < kStThreads / 64; ++w) {
            const int v = s_wsum[w];
            if (w < wave) wbase += v;
            tot += v;
          }
          const int live_base = wbase + incl - cnt - rnd * kStTokens;
          n_live = __builtin_amdgcn_readfirstlane(tot);
#pragma unroll
          for (int r = 0; r < kStSuper; ++r) {
            if (live_mask & (1u << r)) {
              const int p = live_base + __popc(live_mask & ((1u << r) - 1u));
              if (p >= 0 && p < kStTokens) s_tidx[p] = r * kStTokens + tid;
            }
          }
          __syncthreads();   // (s_tidx complete; s_wsum is the scan's again below)
          my_i = tid < n_live - rnd * kStTokens ? s_tidx[tid] : n;
        }
        int4 tk = make_int4(0, 0x7F800000, 0, 0);
        if (my_i < n) tk = tokc[my_i];
        const float cost = __int_as_float(tk.y);
        if constexpr (kTimers) { if (tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_phase(D, 20, tq); } }   // (descriptor + tokens landed)
        int nem = 0, slots = 0, arcbeg = 0;
        if (my_i < n && cost <= cutoff) {  // base-inl.h:315
          uint32_t code = kCodeUnknown;
          if (D.degcode) {   // the token's degree code (wfst_device.h): its arcs without a look at the row header
            const int zz = tk.z;
            const uint32_t rest = zz >= 0 ? (uint32_t)zz >> D.tok_idx_bits : zz <= kPrevUnresolved ? (uint32_t)(kPrevUnresolved - zz) : (kCodeUnknown >> 2);
            code = (rest << 2) | ((uint32_t)tk.w >> 30);
          }
          if (code != kCodeUnknown) {
            nem = (int)((code >> 2) & 15u);
            slots = nem + 2 * (int)(code >> 6);
            arcbeg = tk.x + 1 + (int)(code & 3u);
          } else {
            const int4 hdr = D.g.arcs[tk.x];  // row header: {(n_emit << 12) | n_eps, -, pseudo arcs, -}
            nem = (int)((uint32_t)hdr.x >> kEpsBits);
            slots = nem + 2 * hdr.z;
            arcbeg = tk.x + 1 + (int)((uint32_t)hdr.x & kEpsMask);
          }
        }
        s_nemit[tid] = nem;
        s_cost[tid] = cost;
        s_arcbeg[tid] = arcbeg;
        int incl = slots;
        incl = wave_incl_scan(incl);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kStThreads / 64; ++w) {
          const int v = s_wsum[w];
          if (w < wave) wbase += v;
          tot += v;
        }
        s_base[tid] = wbase + incl - slots;
        if (tid == 0) { s_base[kStTokens] = tot; s_bound = 0xFFFFFFFFu; s_best = ~0ull; }
        lds_barrier();   // (LDS only: the row's DMAs stay in flight)
      }
      const int total = s_base[kStTokens];
      counted = true;
      float bound = kInf;
      if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 11, tq); }
      for (int s0 = 0; s0 < total; s0 += kStSlots) {
        const int S = min(kStSlots, total - s0);
        // ---- (a) every slot of the pass: one lane's 16-byte DMA into the LDS image (64 consecutive slots per instruction) ----
        int lo[kStIter];
        {
          // the owner of each of this thread's slots: kStIter binary searches over the scanned offsets, run in LOCKSTEP (a search is
          // a chain of dependent LDS reads; one after the other they were ~2 us in front of the pass's last DMA)
#pragma unroll
          for (int i = 0; i < kStIter; ++i) lo[i] = 0;   // s_base[lo] <= s0 + j: the largest such index (s_base[0] = 0)
          static_assert(kStTokens == 1 << kLog2StTokens, "the owner search halves a power of two");
#pragma unroll
          for (int step = kStTokens >> 1; step >= 1; step >>= 1) {
            int bm[kStIter];
#pragma unroll
            for (int i = 0; i < kStIter; ++i) bm[i] = s_base[lo[i] + step];
#pragma unroll
            for (int i = 0; i < kStIter; ++i) {
              const int J = s0 + min((wave + i * (kStThreads / 64)) * 64 + lane, S - 1);
              if (bm[i] <= J) lo[i] += step;
            }
          }
          int ab_[kStIter], bs_[kStIter];
#pragma unroll
          for (int i = 0; i < kStIter; ++i) { ab_[i] = s_arcbeg[lo[i]]; bs_[i] = s_base[lo[i]]; }
#pragma unroll
          for (int i = 0; i < kStIter; ++i) {
            const int g = wave + i * (kStThreads / 64);   // this wave's i-th group of 64 slots
            const int j = g * 64 + lane;
            const int J = s0 + min(j, S - 1);
            if (g * 64 < S) {   // (wave-uniform)
              const int4 *src = D.g.arcs + (ab_[i] + (J - bs_[i]));
              if (j < S) __builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(s_arc + g * 64), 16, 0, 0);
            }
          }
        }
        if constexpr (kRow) {
          // the frame's log-likelihood row of this channel, BEHIND the arcs in issue order (a wave's vector-memory results come back
          // in issue order: the tokens and the arcs are what the tile waits for; the row is needed with the arcs, at the pricing):
          // 16 bytes per lane, 1 KB per wave instruction, straight into LDS
          if (llrow != row_have) {   // (a workgroup's consecutive tiles are often one channel's)
            const int n16 = D.stride >> 2;
#pragma unroll
            for (int i = 0; i < kStRowFloats / 4 / kStThreads; ++i) {
              const int g = wave + i * (kStThreads / 64);
              if (g * 64 < n16) {   // (wave-uniform)
                const int j = g * 64 + lane;
                if (j < n16) __builtin_amdgcn_global_load_lds((gbl_void_p)(llrow + 4 * j), (lds_void_p)(s_row + g * 256), 16, 0, 0);
              }
            }
            row_have = llrow;
          }
        }
        // next_cutoff as it stands now (the seed tile's and the other tiles' tightenings): asked for here, back with the arcs
        const uint32_t bfresh = ld_agent(&ctl->bound);
        const u64 best_seen = D.best_exp ? ld_agent(&ctl->best_next) : 0ull;   // (the frame's cheapest candidate so far)
        if constexpr (!kRow) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slots have landed (a lane reads back its own)
        if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 12, tq); }
        // ---- (b) the log-likelihood of every arc slot: one lane's 4-byte DMA (a pseudo arc's second slot has no column) ----
#pragma unroll
        for (int i = 0; i < kStIter; ++i) {
          const int g = wave + i * (kStThreads / 64);
          const int j = g * 64 + lane;
          if (g * 64 < S) {
            const int off = s0 + min(j, S - 1) - s_base[lo[i]], pi = off - s_nemit[lo[i]];
            const bool second = pi >= 0 && (pi & 1);
            const int col = (j < S && !second) ? (s_arc[j].x & D.g.col_mask) : 0;
            if (j < S) __builtin_amdgcn_global_load_lds((gbl_void_p)(llrow + col), (lds_void_p)(s_ll + g * 64), 4, 0, 0);
          }
        }
        }
        bound = fminf(bound, o2f(bfresh));
        if constexpr (kRow) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // (drains the DMAs -- kRow: the arcs and the channel's row; a pseudo arc's second slot may be another wave's)
        if constexpr (kRow && kTimers) { if (tid == 0) dbg_phase(D, 12, tq); }
        if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 13, tq); }
        // ---- (c) price every candidate; its record takes the place of its slot ------------------------------------------
        float tmin = kInf;
        u64 cbest = ~0ull;        // best_exp: this thread's cheapest candidate (emitting or epsilon arrival)
        uint32_t cand_mask = 0;   // bit i: slot i of this thread is a candidate (not a pseudo arc's second slot, not padding)
        // (kPrB slots at a time: their LDS reads are asked for together -- slot by slot, behind branches, a thread's pricing was a chain
        // of dependent LDS round trips per slot)
#ifndef WFST_PRB
#define WFST_PRB 1
#endif
        constexpr int kPrB = WFST_PRB;
#pragma unroll
        for (int i0 = 0; i0 < kStIter; i0 += kPrB) {
          int4 arc_[kPrB];
          int off_[kPrB], nem_[kPrB], abeg_[kPrB];
          float cost_[kPrB], ll_[kPrB];
#pragma unroll
          for (int q = 0; q < kPrB; ++q) {
            const int i = i0 + q;
            if (i >= kStIter) continue;   // (compile time)
            const int j = min((wave + i * (kStThreads / 64)) * 64 + lane, S - 1), l = lo[i];
            off_[q] = s0 + j - s_base[l];
            nem_[q] = s_nemit[l];
            abeg_[q] = s_arcbeg[l];
            cost_[q] = s_cost[l];
            arc_[q] = s_arc[j];
          }
#pragma unroll
          for (int q = 0; q < kPrB; ++q) {
            const int i = i0 + q;
            if (i >= kStIter) continue;
            ll_[q] = kRow ? s_row[arc_[q].x & D.g.col_mask] : s_ll[min((wave + i * (kStThreads / 64)) * 64 + lane, S - 1)];
          }
#pragma unroll
          for (int q = 0; q < kPrB; ++q) {
            const int i = i0 + q;
            if (i >= kStIter) continue;
            const int j = (wave + i * (kStThreads / 64)) * 64 + lane;
            if (j >= S) continue;
            const int l = lo[i];
            const int off = off_[q], pi = off - nem_[q];
            if (pi >= 0 && (pi & 1)) continue;   // second slot of a pseudo arc
            const bool pseudo = pi >= 0;
            const int a = abeg_[q] + off;     // the slot's index in rows[]
            const int4 arc = arc_[q];
            const float base_cost = (cost_[q] + (-ll_[q])) + __int_as_float(arc.z);   // base-inl.h:326-329
            int4 rec;
            if (pseudo) {
              // the emitting arc's arrival carried on over one path of the target's epsilon closure -- ((cur + ac) + w) + w_1
              // + ... + w_k in path order (base-inl.h:329, 414): an epsilon arrival at the path's end state; it does not
              // tighten next_cutoff (only emitting arcs do, base-inl.h:330-333 vs 415)
              const int4 leaf = (j + 1 < S) ? s_arc[j + 1] : D.g.arcs[a + 1];   // (a pair cut by the end of the pass)
              float tt = base_cost;
              if (leaf.z == 1) {
                tt = tt + __int_as_float(leaf.y);
              } else if (leaf.z == 2) {
                tt = (tt + __int_as_float(leaf.w)) + __int_as_float(leaf.y);
              } else {
                const float *pw = D.g.pseudo_w + (size_t)arc.y * kPseudoDepthMax;
                for (int u = 0; u < leaf.z; ++u) tt = tt + pw[u];
              }
              rec = make_int4(arc.w, __float_as_int(tt), kPrevUnresolved, (int)((uint32_t)leaf.x | kEpsRec));
            } else {
              rec = make_int4(arc.w, __float_as_int(base_cost), tok0 + (super ? s_tidx[l] : l), (int)((uint32_t)a | flags_of((uint32_t)arc.y)));
              tmin = fminf(tmin, base_cost);
            }
            if (D.degcode) {   // the degree code of the state arrived at rides in the record (expand_body)
              const uint32_t code = (uint32_t)arc.x >> kColBits;
              rec.w = (int)(((uint32_t)rec.w & 0x3FFFFFFFu) | (code << 30));
              rec.z = pseudo ? kPrevUnresolved - (int)(code >> 2) : (int)((uint32_t)rec.z | ((code >> 2) << D.tok_idx_bits));
            }
            s_arc[j] = rec;
            cand_mask |= 1u << i;
            if (D.best_exp) {
              const u64 cb = ((u64)f2o(__int_as_float(rec.y)) << 32) | (uint32_t)rec.x;
              cbest = cb < cbest ? cb : cbest;
            }
          }
        }
        // base-inl.h:330-333: next_cutoff tightened by the tile's best emitting candidate -- once for the whole tile
        {
          const float wmin = wave_min_f(tmin);
          if (lane == 0 && wmin < kInf) atomicMin(&s_bound, f2o(wmin + ab));
          if (D.best_exp) {
            // the cheapest candidate IS the cheapest token of the frame being built (FindOrAddToken keeps a state's minimum;
            // ties go to the lowest row): ChanCtl::best_next is complete when the expansion launch ends, and the insert launch
            // neither looks for the best token nor waits for its own atomics before it counts an item out
            const u64 cv = cbest < best_seen ? cbest : ~0ull;
            if (__ballot(cv != ~0ull)) {   // (wave-uniform; few waves of few tiles: the frame's best moves a handful of times)
              const u64 wb = wave_min_u64(cv);
              if (lane == 0) atomicMin(&s_best, wb);
            }
          }
          lds_barrier();
          const uint32_t tb = s_bound;
          if (o2f(tb) < bound) {
            if (tid == 0) atomicMin(&ctl->bound, tb);   // (the other tiles read it afresh: bfresh)
            bound = o2f(tb);
          }
          if (D.best_exp && tid == 0) {
            const u64 tbest = s_best;
            if (tbest < best_seen) atomicMin(&ctl->best_next, tbest);
          }
        }
        if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 14, tq); }
        // ---- (d) survivors: rank within the tile's share of their hash partition -------------------------------------
        int pr[kStIter];
#pragma unroll
        for (int i = 0; i < kStIter; ++i) {
          pr[i] = -1;
          if (!(cand_mask & (1u << i))) continue;
          const int j = (wave + i * (kStThreads / 64)) * 64 + lane;
          const int2 xy = *reinterpret_cast<const int2 *>(&s_arc[j]);
          if (__int_as_float(xy.y) < bound) {
            const int part = part_of(hash32(xy.x), log2part);
            pr[i] = (part << 16) | atomicAdd(&s_cnt[part], 1);
          }
        }
        lds_barrier();
        if (tid < 64) {
          const int cnt = tid < P ? s_cnt[tid] : 0;
          int g = 0;
          if (cnt) {
            g = atomicAdd(&bucket_cnt[tid], cnt);   // ONE global atomic per partition and tile
            if (g + cnt > bcap) atomicOr(&ctl->error, kErrBucketFull);
          }
          s_gbase[tid] = g;
          s_cnt[tid] = 0;
          const uint32_t wrote = wave_sum_u32((uint32_t)cnt);
          if (tid == 0 && wrote) atomicAdd(&s_stat[2], wrote);
        }
        lds_barrier();
        if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 15, tq); }
        // ---- (e) the records, straight from the LDS image to their bucket slots ------------------------------------------
#pragma unroll
        for (int i = 0; i < kStIter; ++i) {
          if (pr[i] < 0) continue;
          const int j = (wave + i * (kStThreads / 64)) * 64 + lane;
          const int p = pr[i] >> 16, gi = s_gbase[p] + (pr[i] & 0xFFFF);
          if (gi < bcap) bucket[(size_t)p * bcap + gi] = s_arc[j];
        }
        __syncthreads();   // the image is free for the next pass / tile
        if (tid == 0) { s_bound = 0xFFFFFFFFu; s_best = ~0ull; }
        if constexpr (kTimers) { if (tid == 0) dbg_phase(D, 16, tq); }
      }
      if ((rnd + 1) * kStTokens < n_live) {
        // (a further round follows: this round's counts now -- the last round's are read back by the tile's tail below)
        const bool exp = s_cost[tid] <= cutoff;
        const int nem = s_nemit[tid], zt = (s_base[tid + 1] - s_base[tid] - nem) >> 1;
        const uint32_t wN = (uint32_t)__popcll(__ballot(exp)), wE = wave_sum_u32(exp ? (uint32_t)nem : 0u),
                       wZ = wave_sum_u32(exp ? (uint32_t)zt : 0u);
        if (lane == 0) { if (wN) atomicAdd(&s_stat[0], wN); if (wE) atomicAdd(&s_stat[1], wE); if (wZ) atomicAdd(&s_stat[3], wZ); }
        __syncthreads();   // (s_tidx and the scan arrays are the next round's)
      }
      }   // (rounds)
    }
    {
      // (a token's closure paths priced = its pseudo arcs = (slots - emitting arcs) / 2; tokens beyond the round's count carry +inf)
      const bool exp = counted && s_cost[tid] <= cutoff;
      const int nem = exp ? s_nemit[tid] : 0, zt = exp ? (s_base[tid + 1] - s_base[tid] - nem) >> 1 : 0;
      tile_tail(D, c, ctl, group, par, exp ? 1u : 0u, (uint32_t)nem, 0u, (uint32_t)zt, s_stat);
    }
    // next tile: the first gridDim.x tiles are taken statically, the rest by ticket
    if (total_tiles <= (int)gridDim.x) break;
    if (tid == 0) s_ticket = (int)gridDim.x + atomicAdd(&fc->ticket[par], 1);
    __syncthreads();
    // (t is uniform: the next descriptor comes as ONE scalar load -- read through a vector register the compiler fetched it
    // field by field, three dependent round trips for every tile after a workgroup's first)
    t = __builtin_amdgcn_readfirstlane(s_ticket);
    __syncthreads();
    if (t < total_tiles) td = tiles[t];
    if constexpr (kTimers) tq = wall_clock64();
  }
}
__global__ __launch_bounds__(kStThreads, 4) void expand_kernel_staged(DecoderDev D, int group, int par) { expand_staged_body<false, false>(D, group, par); }
__global__ __launch_bounds__(kStThreads, 4) void expand_kernel_staged_timed(DecoderDev D, int group, int par) { expand_staged_body<true, false>(D, group, par); }
__global__ __launch_bounds__(kStThreads, 4) void expand_kernel_staged_row(DecoderDev D, int group, int par) { expand_staged_body<false, true>(D, group, par); }
__global__ __launch_bounds__(kStThreads, 4) void expand_kernel_staged_row_timed(DecoderDev D, int group, int par) { expand_staged_body<true, true>(D, group, par); }

// =========================================================================================
// insert_kernel.  A bucket whose records could overfill the LDS table is processed in 2^k
// sub-passes, each taking the states of one sub-hash class (records >= distinct states, so the
// test is safe).
// =========================================================================================
constexpr int kInsertThreads = 512;
#ifndef WFST_INSERT_UNROLL
#define WFST_INSERT_UNROLL 3
#endif
constexpr int kInsertUnroll = WFST_INSERT_UNROLL;   // (1536 records in one sweep: wfst_options.joint_max's default; a fourth record per thread cost five VGPRs at the 80-register limit)

// insert_kernel: a fixed grid of workgroups pulls the planned items (first gridDim.x statically,
// then by ticket); 512 threads, dynamic LDS = lds_slots * 12 bytes (16 in lattice mode).
// kLat = lattice mode (forward links recorded); the best-path instantiation carries none of it.
// kBig = biglm mode: 64-bit keys (graph row | LM pair state << 32), LDS = lds_slots * 16 bytes.
struct BoundaryLite {   // frame_boundary_fused's few words of LDS
  float redf[16];
  u64 best;
  int active, n, nd, front_begin, ntiles, tile_tokens, tile_start, pad_bits;
  // what the workgroup that counted the channel's last item out already holds (nothing is loaded behind the countdown):
  uint32_t hist[256];   // GetCutoff's radix select (the slow path: max_active / min_active / the per-frame limit bind)
  uint32_t sel_prefix, sel_k;
  float prev_ab;        // adaptive_beam of the frame being closed
  u64 h_best;         // ChanCtl::best_next as the expansion left it
  int h_risky;        // the frame's items stored their tokens write-through and count themselves out of stores_left
  int h_nf, h_err;    // the frame's token count (the countdown's own answer), error bits (as of the item's start | this frame's)
  uint32_t h_bound;   // the final next_cutoff
};
template <int kT>
__device__ __forceinline__ void frame_boundary_fused(const DecoderDev &D, int c, const int32_t *target, int chan_cnt, int group,
                                                     int par_next, bool do_prep, BoundaryLite &sh, uint32_t *sel_cache, int sel_cache_cap);
template <int kT, bool kSc1, class Sh, int kKeep>
__device__ __forceinline__ float kth_smallest_t(const int4 *tok, int n, int k, Sh &sh, uint32_t *cache, int cache_cap, uint32_t lo_o, uint32_t hi_o);
#ifndef WFST_NOINLINE_COLD
#define WFST_COLD __forceinline__
#else
#define WFST_COLD __attribute__((noinline))
#endif
#ifndef WFST_COLD_KEEP
#define WFST_COLD_KEEP 4   // costs a thread keeps in registers across the selection's four passes (the rest are re-read)
#endif
template <int kT>
__device__ WFST_COLD float kth_smallest_cold(const int4 *tok, int n, int k, BoundaryLite *sh, uint32_t *cache, int cache_cap, uint32_t lo_o, uint32_t hi_o) {
  return kth_smallest_t<kT, true, BoundaryLite, WFST_COLD_KEEP>(tok, n, k, *sh, cache, cache_cap, lo_o, hi_o);
}

template <bool kLat, bool kBig, bool kFused>
__device__ __forceinline__ void insert_body(const DecoderDev &D, int group, int par, const int32_t *target = nullptr, int boundary = 0,
                                            int chan_cnt = 0) {
  constexpr bool kTwo = kFused && !kLat && !kBig;   // the instantiation that can close a frame itself (two launches per frame)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int SLmax = D.lds_slots;
  typedef typename std::conditional<kBig, u64, int32_t>::type KeyT;
  const KeyT kNoKey = (KeyT)~(KeyT)0;  // == kEmptyKey for the 32-bit table
  u64 *vals = reinterpret_cast<u64 *>(smem);
  KeyT *keys = reinterpret_cast<KeyT *>(smem + (size_t)SLmax * 8);
  int32_t *tidx = reinterpret_cast<int32_t *>(smem + (size_t)SLmax * (8 + sizeof(KeyT)));  // lattice mode only
  // one struct, a multiple of 16 bytes, so the dynamic LDS region behind it stays 16-byte aligned
  // (64-bit LDS atomics on a misaligned table are replayed: cdna_hip_programming.md Guideline 17)
  struct __attribute__((aligned(16))) InsertShared {
    int nstates, gpos, wpos, ok, item, pad[3];
    int lw[12];   // lattice mode: links per wave of the item, [8] = the item's first link
    u64 best[kInsertThreads / 64];
    int pref[68];  // record offsets of the buckets of the current item
  };
  static_assert(sizeof(InsertShared) % 16 == 0, "keep the dynamic LDS base aligned");
  __shared__ InsertShared ish;
  int &s_nstates = ish.nstates, &s_gpos = ish.gpos, &s_wpos = ish.wpos, &s_ok = ish.ok, &s_item = ish.item, &s_last = ish.pad[0];
  int &s_ech = ish.pad[1], &s_efill = ish.pad[2];   // lattice mode on the fused rows: this item's chunk of the channel's emitter list, entries used
  constexpr bool kListEmit = kLat && kFused && !kBig;
  __shared__ BoundaryLite bsh;
  u64 *s_best = ish.best;
  int *s_pref = ish.pref;
  FrameCtl *fc = D.fctl + group;
  const int n_heavy = min(fc->n_items[par], D.item_cap / 2), n_items = n_heavy + min(fc->n_small[par], D.item_cap / 2);
  const int P = D.n_part;

  unsigned long long th = wall_clock64();   // (phase timers: the head of an item -- launch or ticket to its records' addresses)
  for (int it = blockIdx.x; it < n_items;) {
  // (signed decode: channels are < 32768, wfst_decoder_create_ex; the unsigned spelling costs 12 VGPRs and a wave of occupancy)
  // (the item's index is uniform: a scalar load; its record prefix -- written beside it by plan_channel -- comes in the same round trip)
  const int islot = __builtin_amdgcn_readfirstlane(it < n_heavy ? it : D.item_cap - 1 - (it - n_heavy));
  const int item = D.items[(size_t)group * D.item_cap + islot];
  const int my_pref = D.item_pref[((size_t)group * D.item_cap + islot) * 64 + lane];
  const int c = item >> 16, g0 = (item >> 8) & 0xFF, G = item & 0xFF;
  ChanCtl *ctl = D.ctl + c;
  // this item's chunk of the channel's emitter list: asked for now, its answer is looked at after pass 1 (the atomic's round
  // trip hides behind the record loads); the list's counter has a line of its own (the channel's control line is busy enough)
  int ech_reg = 0;
  if (kListEmit && tid == 0) {
    s_efill = 0;
    if (G) ech_reg = atomicAdd(&D.emit_cnt[c * 32], kEmitChunk);
  }
  int n = 0;
  {
    // records in the item's buckets: the inclusive prefix plan_channel left beside the item (every wave holds the same)
    n = G ? __shfl(my_pref, G - 1, 64) : 0;
    if (wave == 0 && lane < G) s_pref[lane + 1] = my_pref;
    if (tid == 0) s_pref[0] = 0;
  }
  int log2g = 0;
  while ((1 << log2g) < G) ++log2g;
  const int log2grp = D.log2part - log2g;  // hash bits that select this group of partitions
  const int4 *bucket0 = D.bucket + ((size_t)c * P + g0) * D.bucket_cap;
  const int32_t *bucket_lm0 = kBig ? D.bucket_lm + ((size_t)c * P + g0) * D.bucket_cap : nullptr;
  const uint32_t bound_o = ctl->bound;
  const float cutoff = o2f(bound_o);  // FINAL next_cutoff of this frame
  // two launches per frame: what the workgroup that counts the channel's last item out will need to close the frame, asked for
  // with everything else (complete since the expansion launch ended: the best candidate; the error bits so far)
  const u64 best_early = (kTwo && D.best_exp) ? ctl->best_next : ~0ull;
  const int err_early = kTwo ? ctl->error : 0;
  // (kRiskyBit of items_left, set by plan_channel: this frame's boundary may have to read the frame's tokens -- the bit does not
  // change during the launch, whatever the countdown does to the bits below it)
  const bool risky = kTwo && boundary && (ctl->items_left & kRiskyBit);
  // table sized to the load: the smallest power of two >= 4 n (records >= distinct states)
  int log2sl = 6;
  while ((1 << log2sl) < 4 * n && log2sl < D.log2lds) ++log2sl;
  const int SL = 1 << log2sl;
  const uint32_t mask = (uint32_t)SL - 1;
  const int base = ctl->front_begin + ctl->front_count;
  int4 *tok = D.tok + (size_t)c * D.arena_cap;
  u64 *evals = D.eps_vals + (size_t)c * D.ecap;
  int32_t *etoki = D.eps_toki + (size_t)c * D.ecap;
  int32_t *eocc = D.eps_occ_list + (size_t)c * D.wl_cap;
  int4 *wl = D.worklist + (size_t)c * 2 * D.wl_cap;
  u64 *occ_wl = reinterpret_cast<u64 *>(&ctl->eps_occ);  // {eps_occ, wl_n} bumped by one atomic

  // a single bucket too big for the table is processed in 2^k sub-passes by hash class
  int log2sub = 0;
  while (n > ((SL * 3) >> 2) << log2sub) ++log2sub;
  const int sub_shift = 32 - log2grp - log2sl - log2sub;
  if (sub_shift < 0 && tid == 0) atomicOr(&ctl->error, kErrTableFull);
  __syncthreads();
  // logical record i of the group -> (bucket, offset); *lm = its LM pair state (biglm)
  auto load_rec = [&](int i, int *lm) -> int4 {
    *lm = 0;
    if (i >= n) return make_int4(0, 0x7F800000, 0, 0);  // cost +inf: never below a cutoff
    int b = 0;
    if (G > 1) {
      int lo = 0, hi = G;  // s_pref[lo] <= i < s_pref[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_pref[mid] <= i) lo = mid; else hi = mid;
      }
      b = lo;
    }
    const size_t pos = (size_t)b * D.bucket_cap + (i - s_pref[b]);
    if constexpr (kBig) *lm = bucket_lm0[pos];
    return bucket0[pos];
  };
  auto key_of = [&](const int4 &r, int lm) -> KeyT {
    if constexpr (kBig) return big_key(r.x, lm);
    else return (KeyT)r.x;
  };
  auto hash_of = [&](const int4 &r, int lm) -> uint32_t { return kBig ? hash_big(r.x, lm) : hash32(r.x); };

  u64 best = ~0ull;
  if (tid == 0) dbg_phase(D, 22, th);
  unsigned long long tq = wall_clock64();
  for (int sub = 0; sub_shift >= 0 && sub < (1 << log2sub); ++sub) {
    __syncthreads();
    // (the first sweep's records are asked for BEFORE the table is cleared: their round trip runs behind the clearing)
    int4 r[kInsertUnroll];
    int rl[kInsertUnroll];
#pragma unroll
    for (int k = 0; k < kInsertUnroll; ++k) r[k] = load_rec(k * kInsertThreads + tid, &rl[k]);
    for (int i = tid; i < SL; i += kInsertThreads) { keys[i] = kNoKey; vals[i] = kEmptyVal; }
    if (tid == 0) { s_nstates = 0; s_wpos = 0; s_ok = 1; }
    __syncthreads();
    if (tid == 0) dbg_phase(D, 6, tq);

    // pass 1: insert-or-min.  Candidates that lost against the final cutoff are dropped here: the
    // reference keeps those order-dependent extras (base-inl.h:330) but never expands them.
    // (an item that fits one sweep -- every planned item does -- keeps its records in registers for pass 2)
    const bool one_sweep = n <= kInsertThreads * kInsertUnroll;
    for (int i0 = 0; i0 < n; i0 += kInsertThreads * kInsertUnroll) {
      if (i0 > 0) {
#pragma unroll
        for (int k = 0; k < kInsertUnroll; ++k) r[k] = load_rec(i0 + k * kInsertThreads + tid, &rl[k]);
      }
#pragma unroll
      for (int k = 0; k < kInsertUnroll; ++k) {
        if (!(__int_as_float(r[k].y) < cutoff)) continue;
        const uint32_t h = hash_of(r[k], rl[k]);
        if (log2sub && (int)((h >> sub_shift) & ((1u << log2sub) - 1u)) != sub) continue;
        uint32_t slot = lds_slot_of(h, log2grp, log2sl);
        const KeyT key = key_of(r[k], rl[k]);
        bool found = false;
        for (int q = 0; q < SL; ++q) {
          KeyT kk = keys[slot];
          if (kk == kNoKey) {
            kk = atomicCAS(&keys[slot], kNoKey, key);
            if (kk == kNoKey) { atomicAdd(&s_nstates, 1); found = true; break; }
          }
          if (kk == key) { found = true; break; }
          slot = (slot + 1) & mask;
        }
        if (found) atomicMin(&vals[slot], ((u64)f2o(__int_as_float(r[k].y)) << 32) | (uint32_t)r[k].w);
        else atomicOr(&ctl->error, kErrTableFull);
      }
    }
    __syncthreads();
    if (tid == 0) dbg_phase(D, 7, tq);
    const int ns = s_nstates;
    if (tid == 0) {
      int g = atomicAdd(&ctl->new_count, ns);
      s_gpos = g;
      if constexpr (kListEmit) s_ech = ech_reg;
      // (fused best-path decoders: the per-frame limit is a max_active, not a capacity -- DecoderDev::soft_limit -- and the frame
      // takes what the arena takes)
      if (!(kTwo && D.soft_limit) && g + ns > D.max_tok) { atomicOr(&ctl->error, kErrFrontierFull); s_ok = 0; }
      if ((int64_t)base + g + ns > D.arena_cap) { atomicOr(&ctl->error, kErrArenaFull); s_ok = 0; }
      // (the frame's boundary learns it from the countdown's answer: this item's tokens are counted but not written)
      if (kTwo && !s_ok) atomicOr(reinterpret_cast<u64 *>(&ctl->new_count), kFrameErrBit);
    }
    __syncthreads();
    if (!s_ok) break;
    if (tid == 0) dbg_phase(D, 8, tq);
    const int gpos = s_gpos;
    // (kTwo: the frame's tokens as a raw buffer -- base = the frame's first token, 2 GB of range; dword 3 as for any raw 32-bit buffer)
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t tok_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(tok + base), 0, 0x7FFFFFF0, 0x00020000);

    // pass 2: the record that won its state writes the token
    // (lattice mode: an item that fits one sweep -- every planned item does -- keeps each live
    // record's {source token, arc, cost} and LDS slot in registers for pass 3)
    int lk_src[kInsertUnroll], lk_arc[kInsertUnroll], lk_cost[kInsertUnroll], lk_slot[kInsertUnroll];
#pragma unroll
    for (int k = 0; k < kInsertUnroll; ++k) lk_slot[k] = -1;
    for (int i0 = 0; i0 < n; i0 += kInsertThreads * kInsertUnroll) {
      if (!one_sweep) {
#pragma unroll
        for (int k = 0; k < kInsertUnroll; ++k) r[k] = load_rec(i0 + k * kInsertThreads + tid, &rl[k]);
      }
#pragma unroll
      for (int k = 0; k < kInsertUnroll; ++k) {
        bool winner = false;
        u64 packed = 0;
        uint32_t wslot = 0;
        bool in_table = false;
        if (__int_as_float(r[k].y) < cutoff) {
          const uint32_t h = hash_of(r[k], rl[k]);
          if (!log2sub || (int)((h >> sub_shift) & ((1u << log2sub) - 1u)) == sub) {
            packed = ((u64)f2o(__int_as_float(r[k].y)) << 32) | (uint32_t)r[k].w;
            uint32_t slot = lds_slot_of(h, log2grp, log2sl);
            const KeyT key = key_of(r[k], rl[k]);
            for (int q = 0; q < SL; ++q) {
              const KeyT kk = keys[slot];
              if (kk == key) {
                // the record holding the state's minimum writes the token.  With fused closures two
                // candidates at one state can yield the SAME epsilon arrival (same last arc, costs equal
                // after rounding): the first to swap the slot's value away is the one
                // (the claimed value keeps the cost: lattice mode reads it back for the links, pass 3; its low word, arc bits and flags
                // all ones, is no record's -- row indices stay below kNoArc)
                winner = vals[slot] == packed && (!kFused || atomicCAS(&vals[slot], packed, packed | 0xFFFFFFFFull) == packed);
                in_table = true;
                break;
              }
              if (kk == kNoKey) break;
              slot = (slot + 1) & mask;
            }
            wslot = slot;
          }
        }
        if (kLat && in_table && !(kFused && ((uint32_t)r[k].w & kEpsRec))) {   // (a fused epsilon arrival is no link: epsilon_links)
          lk_src[k] = r[k].z; lk_arc[k] = (int)((uint32_t)r[k].w & kArcMask); lk_cost[k] = r[k].y; lk_slot[k] = (int)wslot;
        }
        const u64 wm = __ballot(winner);
        if (!wm) continue;
        int wb = 0;
        if (lane == 0) wb = atomicAdd(&s_wpos, __popcll(wm));
        wb = __builtin_amdgcn_readfirstlane(wb);
        int idx = 0;
        const uint32_t flags = (uint32_t)r[k].w & kFlagMask;
        if (winner) {
          idx = base + gpos + wb + lane_rank(wm);
          // {state, cost, source token, arc | flags}.  Two-launch decoders, on the frames whose GetCutoff may have to look at the
          // tokens (risky): WRITE-THROUGH (sc1) -- the workgroup that closes the frame reads the frame's costs in this same launch
          if (kTwo && risky) {   // (uniform)
            typedef int v4i_t __attribute__((ext_vector_type(4)));
            const v4i_t v = {r[k].x, r[k].y, r[k].z, r[k].w};
            __builtin_amdgcn_raw_buffer_store_b128(v, tok_rsrc, (gpos + wb + lane_rank(wm)) * 16, 0, 16);   // (aux 16 = sc1)
          } else {
            tok[idx] = r[k];
          }
          if constexpr (kBig) D.tok_lm[(size_t)c * D.arena_cap + idx] = rl[k];
          if (kLat) tidx[wslot] = idx;
          // (kTwo: the best token's graph ROW rides in the low word -- all the next frame's seed needs, DecoderDev::best_row)
          if (!(kTwo && D.best_exp)) {   // (best_exp: the expansion has found the best token already)
            const u64 b = (packed & 0xFFFFFFFF00000000ull) | (uint32_t)(kTwo ? r[k].x : idx);
            best = b < best ? b : best;
          }
        }
        // a token on an epsilon-TARGET state registers itself in the channel's direct-mapped
        // epsilon table (so an epsilon arc arriving later meets its cost); a token with epsilon
        // arcs OUT seeds the closure worklist
        // (fused closures: the epsilon arrivals are candidates like any other; nothing to register or seed --
        // except, in lattice mode, the token of an epsilon-target state for the link pass of the closure kernel)
        if constexpr (kFused) {
          if constexpr (kLat) {
            if (winner && (flags & kFlagEpsTarget)) {
              const int ord = (int)((uint32_t)D.g.arcs[(uint32_t)r[k].w & kArcMask].y & 0x7FFFFFFFu) - 1;
              if (ord >= 0) etoki[ord] = idx;
            }
          }
          if constexpr (kListEmit) {
            // a token with epsilon arcs out goes on the channel's emitter list (epsilon_links reads the list instead of sweeping
            // the frame): positions from the item's chunk, beyond it one by one
            const bool em = winner && (flags & kFlagOutEps);
            const u64 emm = __ballot(em);
            if (emm) {
              int eb = 0;
              if (lane == 0) eb = atomicAdd(&s_efill, __popcll(emm));
              eb = __shfl(eb, 0, 64);
              const int p = eb + lane_rank(emm);
              const u64 ovm = __ballot(em && p >= kEmitChunk);   // beyond the chunk: one more atomic for the wave's overflow
              int ob = 0;
              if (ovm) {
                if (lane == 0) ob = atomicAdd(&D.emit_cnt[c * 32], __popcll(ovm));
                ob = __shfl(ob, 0, 64);
              }
              if (em) {
                const int pos = p < kEmitChunk ? s_ech + p : ob + lane_rank(ovm);
                if (pos < 8 * D.wl_cap) reinterpret_cast<int32_t *>(D.worklist + (size_t)c * 2 * D.wl_cap)[pos] = idx;
              }
            }
          }
          continue;
        }
        const bool tgt = winner && (flags & kFlagEpsTarget);
        const bool seed = winner && (flags & kFlagOutEps);
        const u64 tm = __ballot(tgt), sm = __ballot(seed);
        if (!(tm | sm)) continue;
        u64 ob = 0;
        if (lane == 0) ob = atomicAdd(occ_wl, (u64)__popcll(tm) | ((u64)__popcll(sm) << 32));
        ob = __shfl(ob, 0, 64);
        if (tgt) {
          const uint32_t arc = (uint32_t)r[k].w & kArcMask;
          int ord;
          if constexpr (kBig) {
            // hashed epsilon table: claim the slot of (row, pair).  Keys of one frame are distinct
            // (one winner per key), so the claim cannot meet itself; other keys are probed past.
            u64 *ekeys = D.eps_keys + (size_t)c * D.ecap;
            const u64 key = big_key(r[k].x, rl[k]);
            const uint32_t emask = (uint32_t)D.ecap - 1u;
            uint32_t es = hash_big(r[k].x, rl[k]) & emask;
            ord = -1;
            for (int q = 0; q < D.ecap; ++q) {
              if (ld_agent(&ekeys[es]) == kEmptyVal && atomicCAS(&ekeys[es], kEmptyVal, key) == kEmptyVal) { ord = (int)es; break; }
              es = (es + 1) & emask;
            }
          } else {
            ord = (int)((uint32_t)D.g.arcs[arc].y & 0x7FFFFFFFu) - 1;
          }
          const int op = (int)(uint32_t)ob + lane_rank(tm);
          if (op < D.wl_cap && ord >= 0) {
            evals[ord] = (packed & 0xFFFFFFFF00000000ull) | arc;
            etoki[ord] = idx;
            eocc[op] = ord;
          } else atomicOr(&ctl->error, kErrWorklistFull);
        }
        if (seed) {
          const int wp = (int)(ob >> 32) + lane_rank(sm);
          if (wp < D.wl_cap) wl[wp] = make_int4(rl[k], r[k].x, r[k].y, 0);  // {LM pair (biglm), state, cost}
          else atomicOr(&ctl->error, kErrWorklistFull);
        }
      }
    }
    // pass 3 (lattice mode): every surviving candidate is a forward link of the lattice
    // (base-inl.h:340-341), source token -> the token that won its next state
    if (kLat) {
      __syncthreads();
      int4 *links = D.links + (size_t)c * D.link_cap;
      if (one_sweep) {
        // ONE atomic on the channel's link counter per item (a wave-by-wave append was ~800 returning atomics on one address per
        // frame of a heavy channel at beam 15, serialised in L2 -- and on the control line every other atomic of the frame uses):
        // links per thread -> prefix over the workgroup -> the item's block
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kInsertUnroll; ++k) cnt += lk_slot[k] >= 0;
        int incl = cnt;
        incl = wave_incl_scan(incl);
        if (lane == 63) ish.lw[wave] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kInsertThreads / 64; ++w) {
          const int v = ish.lw[w];
          if (w < wave) wbase += v;
          tot += v;
        }
        if (tid == 0) ish.lw[8] = tot ? atomicAdd(&ctl->link_count, tot) : 0;
        __syncthreads();
        int lp = ish.lw[8] + wbase + incl - cnt;
#pragma unroll
        for (int k = 0; k < kInsertUnroll; ++k) {
          if (lk_slot[k] < 0) continue;
          // (link_delta: link cost - cost of the token that won the state, the table's minimum)
          const int lw = D.link_delta ? __float_as_int(__int_as_float(lk_cost[k]) - o2f((uint32_t)(vals[lk_slot[k]] >> 32))) : lk_cost[k];
          if ((int64_t)lp < D.link_cap) links[lp] = make_int4(lk_src[k], tidx[lk_slot[k]], lk_arc[k], lw);
          else atomicOr(&ctl->error, kErrLinksFull);
          ++lp;
        }
      } else {
      for (int i0 = 0; i0 < n; i0 += kInsertThreads * kInsertUnroll) {
#pragma unroll
        for (int k = 0; k < kInsertUnroll; ++k) {
          int rlm;
          const int4 r = load_rec(i0 + k * kInsertThreads + tid, &rlm);
          bool live = false;
          int dst = 0;
          float dcost = 0.0f;
          if (__int_as_float(r.y) < cutoff && !(kFused && ((uint32_t)r.w & kEpsRec))) {
            const uint32_t h = hash_of(r, rlm);
            if (!log2sub || (int)((h >> sub_shift) & ((1u << log2sub) - 1u)) == sub) {
              uint32_t slot = lds_slot_of(h, log2grp, log2sl);
              const KeyT key = key_of(r, rlm);
              for (int q = 0; q < SL; ++q) {
                const KeyT kk = keys[slot];
                if (kk == key) { live = true; dst = tidx[slot]; dcost = o2f((uint32_t)(vals[slot] >> 32)); break; }
                if (kk == kNoKey) break;
                slot = (slot + 1) & mask;
              }
            }
          }
          const u64 lm = __ballot(live);
          if (!lm) continue;
          int lb = 0;
          if (lane == 0) lb = atomicAdd(&ctl->link_count, __popcll(lm));
          lb = __shfl(lb, 0, 64);
          if (live) {
            const int lp = lb + lane_rank(lm);
            if ((int64_t)lp < D.link_cap) links[lp] = make_int4(r.z, dst, (int)((uint32_t)r.w & kArcMask), D.link_delta ? __float_as_int(__int_as_float(r.y) - dcost) : r.y);
            else atomicOr(&ctl->error, kErrLinksFull);
          }
        }
      }
      }
    }
  }
  __syncthreads();
  if (tid == 0) dbg_phase(D, 9, tq);
  if constexpr (kListEmit) {   // the unused entries of the item's chunk of the emitter list
    if (G && tid < kEmitChunk && tid >= s_efill && s_ech + tid < 8 * D.wl_cap)
      reinterpret_cast<int32_t *>(D.worklist + (size_t)c * 2 * D.wl_cap)[s_ech + tid] = -1;
  }
  const bool own_best = !(kTwo && D.best_exp);   // (uniform)
  if (own_best) {
    best = wave_min_u64(best);
    if (lane == 0) s_best[wave] = best;
    __syncthreads();
  }
  if (tid == 0) {
    if (own_best) {
      u64 b = s_best[0];
      for (int w = 1; w < kInsertThreads / 64; ++w) b = s_best[w] < b ? s_best[w] : b;
      if (b != ~0ull) atomicMin(&ctl->best_next, b);
    }
    dbg_phase(D, 10, tq);
    int last = 0;
    if (kTwo && boundary) {
      // two launches per frame: the workgroup whose item is the channel's last closes the frame.  The countdown is a 64-bit
      // add on {new_count, items_left}: its answer carries the frame's token count (every item's allocation has RETURNED before
      // that item counts itself out) and the frame's error bit (set, where it is, by this same lane on this same word) -- so
      // with the best token known since the expansion (best_exp) nothing this workgroup sent has to be waited for, and nothing
      // is loaded behind the countdown.  Without best_exp the best token travels by atomicMin from here: drained first.
      if (own_best) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const u64 old = atomicAdd(reinterpret_cast<u64 *>(&ctl->new_count), 0xFFFFFFFF00000000ull);
      last = ((old >> 32) & 0xFFFFull) == 1;
      if (last) {
        bsh.h_nf = (int)(uint32_t)old;
        bsh.h_err = err_early | ((old & kFrameErrBit) ? kErrInternal : 0);   // (the sticky bit itself was set where the error arose)
        bsh.h_risky = risky ? 1 : 0;
        bsh.h_bound = bound_o;
        bsh.h_best = best_early;
      }
    }
    s_last = last;
    // (the ticket is taken when the item is DONE: asked for earlier, a busy workgroup would sit on an item that an idle one
    // could have had -- measured: +1.5 ms per step)
    s_item = n_items > (int)gridDim.x ? (int)gridDim.x + atomicAdd(&fc->item_ticket[par], 1) : n_items;   // (no ticket where every item had its workgroup from the start)
  }
  // two launches per frame: every wave's token stores have LANDED before the item counts itself out of stores_left (behind the
  // barrier below; the wait overlaps the countdown's round trip, which wave 0 is waiting for anyway)
  if (risky) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  it = s_item;
  const int last = s_last;
  __syncthreads();
  if (risky && tid == 0) atomicSub(&ctl->stores_left, 1);
  if (tid == 0) dbg_phase(D, 23, tq);   // (countdown + ticket answered)
  if constexpr (kTwo) {
    if (last) {
      // (the hash table's LDS is free now: the selection of a binding limit parks the frame's costs there)
      frame_boundary_fused<kInsertThreads>(D, c, target, chan_cnt, group, par ^ 1, boundary == 1, bsh, reinterpret_cast<uint32_t *>(smem), SLmax * 3);
      if (tid == 0) dbg_phase(D, 24, tq);   // (the frame boundary)
    }
  }
  th = wall_clock64();
  }
}

__global__ __launch_bounds__(kInsertThreads, 6) void insert_kernel_plain(DecoderDev D, int group, int par) { insert_body<false, false, false>(D, group, par); }
__global__ __launch_bounds__(kInsertThreads, 6) void insert_kernel_fused(DecoderDev D, int group, int par, const int32_t *target, int boundary, int chan_cnt) {
  insert_body<false, false, true>(D, group, par, target, boundary, chan_cnt);
}
__global__ __launch_bounds__(kInsertThreads) void insert_kernel_lattice(DecoderDev D, int group, int par) { insert_body<true, false, false>(D, group, par); }
__global__ __launch_bounds__(kInsertThreads) void insert_kernel_lattice_fused(DecoderDev D, int group, int par) { insert_body<true, false, true>(D, group, par); }
__global__ __launch_bounds__(kInsertThreads) void insert_kernel_biglm(DecoderDev D, int group, int par) { insert_body<false, true, false>(D, group, par); }
__global__ __launch_bounds__(kInsertThreads) void insert_kernel_lattice_biglm(DecoderDev D, int group, int par) { insert_body<true, true, false>(D, group, par); }

// =========================================================================================
// closure_kernel and its pieces.  One 1024-thread workgroup per channel (lattice decoders on the fused rows: DecoderDev::closure_slabs
// of them, which share the frame's epsilon links -- finalize_frame).
// =========================================================================================
constexpr int kBT = 1024;
constexpr int kBW = kBT / 64;
constexpr int kClosureUnroll = 4;

struct BoundaryShared {
  int nnew;       // tokens of the frame being built (continues ChanCtl::new_count)
  int wl_n[2];
  int err;
  int occ;        // epsilon-table entries touched this frame (continues ChanCtl::eps_occ)
  int nwon;       // tokens won by an epsilon arc this frame
  int nemit;      // lattice mode: tokens of the frame that emit epsilon links
  int nlinks;     // lattice mode: epsilon links this workgroup appended (epsilon_links)
  int last;       // closure launches with several workgroups per channel: this one is the last to finish its share
  int eps_total;  // ... and then: the epsilon links all of them appended
  float redf[kBW];
  u64 red64[kBW];
  u64 best;
  uint32_t hist[256];
  uint32_t sel_prefix, sel_k;
  int active;
  int tile_tokens;  // tokens per expansion tile of the coming frame (prep_frame)
};

// Forward links of the epsilon arcs of the frame being built (lattice mode; base-inl.h:421-422), once every token of the
// frame has its final cost -- after the closure's fixpoint, or, with fused closures, right after the insert launch.
// toki[] = the channel's direct-mapped epsilon table: the frame's token on each epsilon-target state.
template <bool kBig = false>
__device__ __forceinline__ void epsilon_links(const DecoderDev &D, int c, BoundaryShared &sh, int base, float cutoff, bool listed = false,
                                              int slab = 0, int n_slabs = 1) {
  // slab / n_slabs: the launch runs n_slabs workgroups per channel (closure_kernel on the fused rows of a lattice decoder: a heavy
  // channel's frame lists ten thousand emitters, a chain of a dozen sweeps on one workgroup while the other channels' are done
  // after one); this one takes share `slab` of the listed emitters and counts its links into sh.nlinks
  const int tid = threadIdx.x;
  // biglm: a token is (row, LM pair); an epsilon arc with a word label moves the LM (biglm.h:448-456), the link's cost carries
  // the LM difference, and the destination's token is found in the channel's HASHED epsilon table (keys beside toki[])
  const u64 *ekeys = kBig ? D.eps_keys + (size_t)c * D.ecap : nullptr;
  const int32_t *tlm = kBig ? D.tok_lm + (size_t)c * D.arena_cap : nullptr;
  int4 *tok = D.tok + (size_t)c * D.arena_cap;
  const int32_t *toki = D.eps_toki + (size_t)c * D.ecap;
  // Forward links of the epsilon arcs (base-inl.h:421-422): the reference regenerates a token's
  // links each time it is re-processed, so what remains are the links computed from its FINAL
  // cost.  Every token of the frame being built with epsilon arcs out and cost < cutoff emits them.
  int4 *links = D.links + (size_t)c * D.link_cap;
  const int n_frame = sh.nnew;
  // (1) the tokens that emit (a few percent of the frame), as arena indices (-1: none) in the channel's worklist space (free by
  // now): LISTED by the insert launch on the fused rows (insert_body, kListEmit -- every token it writes is below the final
  // cutoff), else compacted here by a sweep over the frame -- so that (2) runs its dependent gathers with full waves
  int32_t *emit = reinterpret_cast<int32_t *>(D.worklist + (size_t)c * 2 * D.wl_cap);
  int n_emit = 0;
  if (listed && D.emit_cnt[c * 32] <= 8 * D.wl_cap) {
    n_emit = D.emit_cnt[c * 32];
  } else {
    // (the list overflowed -- never seen --: the sweep below rebuilds it in the channel's worklist space, the work of ONE workgroup)
    if (slab != 0) return;
    n_slabs = 1;
    if (tid == 0) sh.nemit = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_frame; i0 += 4 * kBT) {  // 4 independent loads in flight per thread
      int4 T[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kBT + tid;
        T[u] = i < n_frame ? tok[base + i] : make_int4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kBT + tid;
        const bool on = i < n_frame && ((uint32_t)T[u].w & kFlagOutEps) && (__int_as_float(T[u].y) < cutoff);
        const u64 m = __ballot(on);
        int wb = 0;
        if ((tid & 63) == 0 && m) wb = atomicAdd(&sh.nemit, __popcll(m));
        wb = __builtin_amdgcn_readfirstlane(wb);
        if (on) emit[wb + lane_rank(m)] = base + i;
      }
    }
    __syncthreads();
    n_emit = sh.nemit;
  }
  // (equal shares of whole waves, however short the list: one path through the launch for light and heavy channels alike)
  const int share = ((n_emit + n_slabs - 1) / n_slabs + 63) & ~63, j_lo = min(n_emit, slab * share), j_hi = min(n_emit, j_lo + share);
  for (int j0 = j_lo; j0 < j_hi; j0 += kBT) {
    const int j = j0 + tid;
    int idx = -1, row = 0, neps = 0, npass = 0;
    float cost = 0.0f;
    int plm = 0;
    // cost of the arrival over epsilon arc a from this token, and (biglm) the LM pair it arrives with
    auto arrival = [&](int a, const int4 &arc, int *next_lm) -> float {
      if constexpr (kBig) {
        const int ol = D.g.arc_olabel[a];
        float lm_score = 0.0f;
        *next_lm = plm;
        if (ol != 0) {
          int n1, n2;
          lm_score = lm_step(D, c, plm, ol, &n1, &n2);
          *next_lm = pair_find(D, c, n1, n2);
        }
        return cost + (__int_as_float(arc.z) + lm_score);   // cur_cost + (arc weight + lm_score), biglm.h:448-452
      } else {
        *next_lm = 0;
        return cost + __int_as_float(arc.z);
      }
    };
    // the first epsilon arcs of the row are asked for WITH its header (their addresses need only the row; a state has one or
    // two epsilon arcs nearly always): token -> {header, arcs} -> destination tokens -> links, four round trips
    constexpr int kSpec = kBig ? 1 : 3;
    int4 A[kSpec];
    if (j < j_hi) idx = emit[j];
    if (idx >= 0) {
      const int4 T = tok[idx];
      row = T.x;
      cost = __int_as_float(T.y);
      if (!(cost < cutoff) || !((uint32_t)T.w & kFlagOutEps)) idx = -1;
    }
    if (idx >= 0) {
      if constexpr (kBig) plm = tlm[idx];
      const int4 hdr = D.g.arcs[row];
#pragma unroll
      for (int e = 0; e < kSpec; ++e) A[e] = D.g.arcs[row + 1 + e];   // (rows end in padding up to their line: reading past the last arc stays in the image)
      neps = (int)((uint32_t)hdr.x & kEpsMask);
      for (int e = 0; e < neps; ++e) { int nl; npass += arrival(row + 1 + e, e < kSpec ? A[e < kSpec ? e : 0] : D.g.arcs[row + 1 + e], &nl) < cutoff; }
    }
    // one atomicAdd per wave for all its links
    int ps = wave_incl_scan(npass);
    const int wtot = __builtin_amdgcn_readlane(ps, 63);
    int lb = 0;
    if ((tid & 63) == 0 && wtot) { lb = atomicAdd(&D.ctl[c].link_count, wtot); atomicAdd(&sh.nlinks, wtot); }
    lb = __shfl(lb, 0, 64);
    int lp = lb + ps - npass;
    for (int e = 0; e < neps && npass; ++e) {
      const int a = row + 1 + e;
      const int4 arc = e < kSpec ? A[e < kSpec ? e : 0] : D.g.arcs[a];
      int nlm;
      const float tot = arrival(a, arc, &nlm);
      if (!(tot < cutoff)) continue;
      int ord = (int)((uint32_t)arc.y & 0x7FFFFFFFu) - 1;
      if constexpr (kBig) {   // the slot of (destination row, LM pair) in the hashed table
        const u64 key = big_key(arc.w, nlm);
        const uint32_t emask = (uint32_t)D.ecap - 1u;
        uint32_t es = hash_big(arc.w, nlm) & emask;
        ord = -1;
        for (int q = 0; q < D.ecap && nlm >= 0; ++q) {
          const u64 kk = ld_agent(&ekeys[es]);
          if (kk == key) { ord = (int)es; break; }
          if (kk == kEmptyVal) break;
          es = (es + 1) & emask;
        }
      }
      // the destination's token of THIS frame (every epsilon arrival below the cutoff made one, through the insert launch or
      // the closure pass); an entry from an older frame would mean that invariant broke: reported, never linked
      const int dst = ord >= 0 ? ld_agent(&toki[ord]) : -1;
      if (dst < base) atomicOr(&sh.err, kErrInternal);
      else if ((int64_t)lp < D.link_cap) links[lp] = make_int4(idx, dst, a, __float_as_int(D.link_delta ? tot - __int_as_float(tok[dst].y) : tot));
      else atomicOr(&sh.err, kErrLinksFull);
      ++lp;
    }
  }
  __syncthreads();
}

// ProcessNonemitting to its fixpoint (base-inl.h:383-430) on the channel's direct-mapped epsilon
// table, then write the arena records of the tokens an epsilon arc created or improved.  On entry
// sh.wl_n[0] seeds are in worklist[0], sh.wl_n[1] == 0, sh.nnew / sh.occ continue the counters.
// kBig (biglm): the table is hashed by (row, LM pair) -- eps_keys beside eps_vals / eps_toki -- every
// epsilon arc with an output label moves the LM state (biglm.h:448-456), and the flattened closures
// (which know nothing of LM states) are not used.
template <bool kLat, bool kBig>
__device__ __forceinline__ void epsilon_closure(const DecoderDev &D, int c, BoundaryShared &sh, int base, float cutoff, u64 *nZ_out) {
  const int tid = threadIdx.x;
#ifndef WFST_BIG_CLOSURE_UNROLL
#define WFST_BIG_CLOSURE_UNROLL 1
#endif
  // worklist entries a thread takes through a sweep together.  biglm: ONE -- a frame seeds a few hundred entries for the 1024
  // threads, and an entry's LM state, keys and claims held four-fold put 21 registers of the closure launch into scratch
  constexpr int kCU = kBig ? WFST_BIG_CLOSURE_UNROLL : kClosureUnroll;
  ChanCtl *ctl = D.ctl + c;
  u64 *ekeys = kBig ? D.eps_keys + (size_t)c * D.ecap : nullptr;
  u64 *vals = D.eps_vals + (size_t)c * D.ecap;
  int32_t *toki = D.eps_toki + (size_t)c * D.ecap;
  int32_t *occ = D.eps_occ_list + (size_t)c * D.wl_cap;
  int32_t *won = D.eps_won_list + (size_t)c * D.wl_cap;
  int4 *wl = D.worklist + (size_t)c * 2 * D.wl_cap;
  int4 *tok = D.tok + (size_t)c * D.arena_cap;
  u64 nZ = 0;
  unsigned long long tq = wall_clock64();
  if (tid == 0 && (D.dbg & 32)) { atomicAdd(&D.dbg_t[57], 1ull); atomicMax(&D.dbg_t[58], (unsigned long long)sh.wl_n[0]); }
  if (tid == 0) sh.nwon = 0;

  int cur = 0;
  int dbg_rounds = 0;
  unsigned long long dbg_t0 = wall_clock64();
  for (;;) {
    __syncthreads();
    const int nw = sh.wl_n[cur];
    if (tid == 0 && (D.dbg & 32) && dbg_rounds == 1) {  // the first round is over
      const unsigned long long dt = wall_clock64() - dbg_t0;
      atomicAdd(&D.dbg_t[60], dt); atomicMax(&D.dbg_t[61], dt);
    }
    if (nw == 0) break;
    if (tid == 0 && (D.dbg & 32)) { atomicAdd(&D.dbg_t[56], 1ull); ++dbg_rounds; }
    const int4 *wl_cur = wl + (size_t)cur * D.wl_cap;
    int4 *wl_nxt = wl + (size_t)(cur ^ 1) * D.wl_cap;
    // four worklist entries per thread in flight (independent load chains issued together)
    for (int i0 = 0; i0 < nw; i0 += kBT * kCU) {
      int4 ent[kCU];
      uint2 si[kCU];
      int4 arc0[kCU];
      bool live[kCU];
#pragma unroll
      for (int k = 0; k < kCU; ++k) {
        const int i = i0 + k * kBT + tid;
        live[k] = i < nw;
        ent[k] = live[k] ? wl_cur[i] : make_int4(0, 0, 0x7F800000, 0);
        // entry = {-, state, cost when queued}; a later improvement of the same token queues
        // another entry, so a stale cost only repeats work the atomicMin below rejects
        live[k] = live[k] && (__int_as_float(ent[k].z) < cutoff);  // base-inl.h:391
      }
      uint32_t flat[kCU];
      // biglm: what an LM step will need is asked for as early as its address is known -- the pair's LM states with the row header,
      // the arc's output label with the arc -- two round trips off every round of the pass
      u64 pk[kBig ? kCU : 1];
      int ol0[kBig ? kCU : 1];
#pragma unroll
      for (int k = 0; k < kCU; ++k) {
        if constexpr (kBig) pk[k] = live[k] ? ld_agent(&D.pair_keys[(size_t)c * D.pair_cap + ent[k].x]) : 0ull;
        const int4 hdr = live[k] ? D.g.arcs[ent[k].y] : make_int4(0, 0, 0, 0);
        si[k] = make_uint2((uint32_t)ent[k].y + 1u, (uint32_t)hdr.x);
        flat[k] = kBig ? 0u : (uint32_t)hdr.w;  // (first eps_flat entry << 3) | entries; 0: iterate
      }
#pragma unroll
      for (int k = 0; k < kCU; ++k) {
        const bool has = live[k] && (si[k].y & kEpsMask);
        arc0[k] = !has ? make_int4(0, 0, 0, 0) : (flat[k] & 7u) ? D.g.eps_flat[flat[k] >> 3] : D.g.arcs[si[k].x];
        if constexpr (kBig) ol0[k] = has ? D.g.arc_olabel[si[k].x] : 0;
      }
      // FindOrAddToken (base-inl.h:88-136) for one epsilon arrival: one atomicMin on the state's own
      // slot.  kEpsWon in the low word makes an emitting arc win an exact cost tie, as the
      // reference's first-arrival rule does (emitting arcs are processed before the closure).
      // requeue: the state's own epsilon arcs still have to be followed from this cost
      // (base-inl.h:425); not for a flattened closure, whose deeper entries are those arcs.
      // The arrivals of a thread are priced level by level (level e = the e-th path of a flattened
      // closure, or the state's e-th epsilon arc) and the atomicMin of a whole level -- one per entry
      // in flight -- is issued before any of its results is looked at: a RETURNING global atomic is a
      // round trip of a few microseconds, and one after the other (up to 16 per thread on the heaviest
      // channel) they were what the slowest workgroup of the launch spent its time on.
      auto finish = [&](int ord, float tot, bool out_eps, int next_row, bool requeue, int next_lm, u64 packed, u64 old) {
        if (!(packed < old)) return;
        const uint32_t otot = (uint32_t)(packed >> 32);
        if (old == kEmptyVal) {  // a state no emitting arc reached: new token
          toki[ord] = base + atomicAdd(&sh.nnew, 1);
          const int op = atomicAdd(&sh.occ, 1);
          if (op < D.wl_cap) occ[op] = ord; else atomicOr(&sh.err, kErrWorklistFull);
        }
        if (old == kEmptyVal || !((uint32_t)old & kEpsWon)) {  // first epsilon win of this token
          const int wn = atomicAdd(&sh.nwon, 1);
          if (wn < D.wl_cap) won[wn] = ord; else atomicOr(&sh.err, kErrWorklistFull);
        }
        if (requeue && out_eps && otot < (uint32_t)(old >> 32)) {
          const int wp = atomicAdd(&sh.wl_n[cur ^ 1], 1);
          if (wp < D.wl_cap) wl_nxt[wp] = make_int4(next_lm, next_row, __float_as_int(tot), 0);
          else atomicOr(&sh.err, kErrWorklistFull);
        }
      };
      int nit[kCU];             // levels of entry k: paths of its flattened closure, or its epsilon arcs
      float pc[kCU][kFlatMax - 1];  // flattened closure: cost of path q (a later path's parent)
#pragma unroll
      for (int k = 0; k < kCU; ++k) {
        nit[k] = !live[k] ? 0 : (flat[k] & 7u) ? (int)(flat[k] & 7u) : (int)(si[k].y & kEpsMask);
#pragma unroll
        for (int q = 0; q < kFlatMax - 1; ++q) pc[k][q] = __builtin_huge_valf();
      }
      for (int e = 0;; ++e) {
        bool more = false;
#pragma unroll
        for (int k = 0; k < kCU; ++k) more |= e < nit[k];
        if (!__ballot(more)) break;
        int4 E[kCU];
        int OL[kBig ? kCU : 1];
#pragma unroll
        for (int k = 0; k < kCU; ++k) {
          E[k] = e >= nit[k] ? make_int4(0, 0, 0, 0)
                 : e == 0 ? arc0[k]
                 : (flat[k] & 7u) ? D.g.eps_flat[(flat[k] >> 3) + e] : D.g.arcs[si[k].x + e];
          if constexpr (kBig) OL[k] = e >= nit[k] ? 0 : e == 0 ? ol0[k] : D.g.arc_olabel[si[k].x + e];
        }
        int c_ord[kCU], c_row[kCU], c_lm[kCU];
        float c_tot[kCU];
        u64 c_packed[kCU], c_old[kCU];
        uint32_t c_flags[kCU];  // 1 live, 2 out_eps, 4 requeue
#pragma unroll
        for (int k = 0; k < kCU; ++k) {
          c_flags[k] = 0; c_ord[k] = 0; c_row[k] = 0; c_lm[k] = 0; c_tot[k] = 0.0f; c_packed[k] = 0;
          if (e >= nit[k]) continue;
          const float cost = __int_as_float(ent[k].z);
          int a;
          if (flat[k] & 7u) {
            // path e of the state's whole closure: cost = parent path's cost + weight, in path order; a
            // path whose parent or own cost is not below the cutoff is dead (base-inl.h:391,415)
            const int parent = (E[k].z & 7) - 1;
            float cp = cost;
#pragma unroll
            for (int q = 0; q < kFlatMax - 1; ++q) cp = (parent == q) ? pc[k][q] : cp;
            if (!(cp < cutoff)) continue;
            nZ++;
            const float tot = cp + __int_as_float(E[k].w);  // base-inl.h:414
            if (!(tot < cutoff)) continue;                   // base-inl.h:415
#pragma unroll
            for (int q = 0; q < kFlatMax - 1; ++q) pc[k][q] = (e == q) ? tot : pc[k][q];
            c_ord[k] = E[k].x; a = E[k].y; c_tot[k] = tot;
            c_flags[k] = 1u | ((E[k].z & 8) ? 2u : 0u);
          } else {
            a = (int)si[k].x + e;
            nZ++;
            float graph_cost = __int_as_float(E[k].z);
            if constexpr (kBig) {  // biglm.h:448-451
              const int ol = OL[k];
              float lm_score = 0.0f;
              c_lm[k] = ent[k].x;
              if (ol != 0) {
                int n1, n2;
                lm_score = lm_step_pk(D, pk[k], ol, &n1, &n2);
                c_lm[k] = pair_intern(D, c, ctl, n1, n2);
              }
              graph_cost = __int_as_float(E[k].z) + lm_score;
            }
            const float tot = cost + graph_cost;              // base-inl.h:414
            if (!(tot < cutoff)) continue;                    // base-inl.h:415
            c_ord[k] = (int)((uint32_t)E[k].y & 0x7FFFFFFFu) - 1;
            c_row[k] = E[k].w; c_tot[k] = tot;
            c_flags[k] = 1u | (((uint32_t)E[k].y & kFlagOutEps) ? 2u : 0u) | 4u;
            if constexpr (kBig) {  // find or claim the slot of (row, pair)
              const u64 key = big_key(c_row[k], c_lm[k]);
              const uint32_t emask = (uint32_t)D.ecap - 1u;
              uint32_t es = hash_big(c_row[k], c_lm[k]) & emask;
              c_ord[k] = -1;
              for (int q = 0; q < D.ecap; ++q) {
                u64 kk = ld_agent(&ekeys[es]);
                if (kk == kEmptyVal) kk = atomicCAS(&ekeys[es], kEmptyVal, key);
                if (kk == kEmptyVal || kk == key) { c_ord[k] = (int)es; break; }
                es = (es + 1) & emask;
              }
              if (c_ord[k] < 0) { atomicOr(&sh.err, kErrTableFull); c_flags[k] = 0; continue; }
            }
          }
          c_packed[k] = ((u64)f2o(c_tot[k]) << 32) | kEpsWon | ((c_flags[k] & 2u) ? kEpsOutBit : 0u) | (uint32_t)a;
        }
#pragma unroll
        for (int k = 0; k < kCU; ++k)
          c_old[k] = (c_flags[k] & 1u) ? atomicMin(&vals[c_ord[k]], c_packed[k]) : 0ull;
#pragma unroll
        for (int k = 0; k < kCU; ++k)
          if (c_flags[k] & 1u)
            finish(c_ord[k], c_tot[k], (c_flags[k] & 2u) != 0, c_row[k], (c_flags[k] & 4u) != 0, c_lm[k], c_packed[k], c_old[k]);
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
  if (tid == 0 && (D.dbg & 32)) atomicMax(&D.dbg_t[59], (unsigned long long)dbg_rounds);
  if (tid == 0) dbg_phase(D, 1, tq);
  const int nocc = min(sh.occ, D.wl_cap), nwon = min(sh.nwon, D.wl_cap);
  const bool fits = sh.nnew <= D.max_tok && (int64_t)base + sh.nnew <= D.arena_cap;
  if (!fits && tid == 0) atomicOr(&sh.err, sh.nnew > D.max_tok ? kErrFrontierFull : kErrArenaFull);
  // Tokens won by an epsilon arc get their record here.  Their backpointer (the token of the arc's
  // source state, on this same frame) is left as kPrevUnresolved: the traceback finds it by
  // scanning the frame for that state -- a few epsilon hops per utterance instead of a dependent
  // lookup chain per token on every frame's critical path.
  u64 best = ~0ull;
  for (int i0 = 0; fits && i0 < nwon; i0 += kBT * kCU) {
    int od[kCU];
#pragma unroll
    for (int k = 0; k < kCU; ++k) {
      const int i = i0 + k * kBT + tid;
      od[k] = i < nwon ? won[i] : -1;
    }
#pragma unroll
    for (int k = 0; k < kCU; ++k) {
      if (od[k] < 0) continue;
      const u64 v = ld_agent(&vals[od[k]]);
      const int idx = ld_agent(&toki[od[k]]);
      int32_t state;
      if constexpr (kBig) {
        const u64 key = ld_agent(&ekeys[od[k]]);
        state = (int32_t)(uint32_t)key;
        D.tok_lm[(size_t)c * D.arena_cap + idx] = (int32_t)(uint32_t)(key >> 32);
      } else {
        state = D.g.eps_target_state[od[k]];
      }
      tok[idx] = make_int4(state, __float_as_int(o2f((uint32_t)(v >> 32))), kPrevUnresolved,
                           (int)(((uint32_t)v & kArcMask) | kFlagEpsTarget | (((uint32_t)v & kEpsOutBit) ? kFlagOutEps : 0u)));
      const u64 b = (v & 0xFFFFFFFF00000000ull) | (uint32_t)(D.best_row ? state : idx);
      best = b < best ? b : best;
    }
  }
  best = wave_min_u64(best);
  if ((tid & 63) == 0) sh.red64[tid >> 6] = best;
  __syncthreads();  // every read of the table above is done before it is cleared
  if (tid == 0) dbg_phase(D, 2, tq);
  if (kLat && fits) epsilon_links<kBig>(D, c, sh, base, cutoff);
  for (int i0 = 0; i0 < nocc; i0 += kBT * kCU) {  // the list loads of a thread issued together
    int od[kCU];
#pragma unroll
    for (int k = 0; k < kCU; ++k) {
      const int i = i0 + k * kBT + tid;
      od[k] = i < nocc ? occ[i] : -1;
    }
#pragma unroll
    for (int k = 0; k < kCU; ++k) {
      if (od[k] < 0) continue;
      vals[od[k]] = kEmptyVal;
      if constexpr (kBig) ekeys[od[k]] = kEmptyVal;
    }
  }
  if (tid == 0) {
    u64 b = sh.red64[0];
    for (int w = 1; w < kBW; ++w) b = sh.red64[w] < b ? sh.red64[w] : b;
    sh.best = b;
  }
  *nZ_out = nZ;
  __syncthreads();
  if (tid == 0) dbg_phase(D, 3, tq);
}

// slab / n_slabs (lattice decoders on the fused rows, closure_kernel): n_slabs workgroups per channel share the frame's epsilon
// links; the one that finishes its share LAST (a 64-bit add on the channel's kClSlabWord: arrivals | links appended << 32) goes
// on to close the frame -- what the others changed it reads through L2 (the link counter, the error bits) -- and returns true,
// the others return false and are done.  No workgroup reads what another one stored with plain stores.
template <bool kLat, bool kBig>
__device__ __forceinline__ bool finalize_frame(const DecoderDev &D, int c, ChanCtl *ctl, BoundaryShared &sh, int slab = 0, int n_slabs = 1) {
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long tq = wall_clock64();
  const int f = ctl->n_decoded;
  const float cutoff = o2f(ctl->bound);
  const int base = ctl->front_begin + ctl->front_count;
  const bool shared_out = kLat && !kBig && n_slabs > 1;
  if (tid == 0) {
    sh.nnew = ctl->new_count;
    sh.occ = ctl->eps_occ < D.wl_cap ? ctl->eps_occ : D.wl_cap;
    sh.wl_n[0] = ctl->wl_n < D.wl_cap ? ctl->wl_n : D.wl_cap;
    sh.wl_n[1] = 0;
    sh.err = 0;
    sh.nlinks = 0;
    // every emitting link into the new frame is recorded (the insert launch is over): the epsilon
    // links of the frame start here  (shared_out: the other workgroups may have appended theirs already -- written below, from the totals)
    if (kLat && !shared_out && f + 1 <= D.max_frames) D.link_mid[(size_t)c * (D.max_frames + 3) + f + 1] = min(ctl->link_count, (int)D.link_cap);
  }
  // the insert workgroups read all bucket counters of the channel to form their groups, so the
  // counters stay untouched during that launch and are reset here
  if (slab == 0) for (int i = tid; i < D.n_part; i += kBT) D.bucket_cnt[(size_t)c * D.n_part + i] = 0;
  __syncthreads();
  if (tid == 0) dbg_phase(D, 0, tq);
  u64 nZ = 0;
  if (!kBig && D.fused) {
    if (tid == 0) sh.best = ~0ull;  // the frame is complete: its epsilon arrivals went through the insert launch
    __syncthreads();
    if constexpr (kLat) {
      // lattice mode: all that is left of ProcessNonemitting are the epsilon links, one flat pass over the frame
      const bool fits = sh.nnew <= D.max_tok && (int64_t)base + sh.nnew <= D.arena_cap && ctl->error == 0;
      if (fits) epsilon_links(D, c, sh, base, cutoff, true, slab, n_slabs);
      __syncthreads();
      if (tid == 0) dbg_phase(D, 1, tq);   // (the epsilon links of the frame)
      if (shared_out) {
        if (tid == 0) {
          if (sh.err) atomicOr(&ctl->error, sh.err);
          u64 *word = reinterpret_cast<u64 *>(D.prune_par + (size_t)c * kPruneParInts + kClSlabWord);
          const u64 old = atomicAdd(word, ((u64)(uint32_t)sh.nlinks << 32) | 1ull);
          sh.last = (int)(uint32_t)old == n_slabs - 1;
          sh.eps_total = (int)(old >> 32) + sh.nlinks;
          if (sh.last) atomicExch(word, 0ull);   // (for the next frame's launch)
        }
        __syncthreads();
        if (!sh.last) return false;
        if (tid == 0) {
          const int lend = ld_agent(&ctl->link_count);
          if (f + 1 <= D.max_frames) D.link_mid[(size_t)c * (D.max_frames + 3) + f + 1] = min(max(lend - sh.eps_total, 0), (int)D.link_cap);
          sh.err |= ld_agent(&ctl->error);
        }
      }
      if (tid == 0) D.emit_cnt[c * 32] = 0;   // (the next frame's insert launch lists afresh)
    }
  } else {
    epsilon_closure<kLat, kBig>(D, c, sh, base, cutoff, &nZ);
  }
  tq = wall_clock64();
  nZ = wave_sum_u64(nZ);
  if (lane == 0) sh.red64[tid >> 6] = nZ;
  __syncthreads();
  if (tid == 0) {
    u64 z = 0;
    for (int w = 0; w < kBW; ++w) z += sh.red64[w];
    int err = sh.err;
    int nf = sh.nnew;
    if ((!D.soft_limit && nf > D.max_tok) || (int64_t)base + nf > D.arena_cap) nf = 0;   // (soft_limit: the per-frame limit is a max_active)
    if (ctl->error | err) nf = 0;  // a channel that hit a limit stops producing tokens
    if (f + 2 > D.max_frames + 1) err |= kErrFramesFull;
    else {
      D.frame_off[(size_t)c * (D.max_frames + 2) + f + 2] = base + nf;
      D.cutoff_hist[(size_t)c * (D.max_frames + 2) + f + 1] = cutoff;
      if (kLat) {
        const int lend = min(shared_out ? ld_agent(&ctl->link_count) : ctl->link_count, (int)D.link_cap);
        D.lat_stats[(size_t)c * 4 + 0] += (u64)max(0, lend - D.link_off[(size_t)c * (D.max_frames + 3) + f + 1]);   // links recorded for this frame
        D.link_off[(size_t)c * (D.max_frames + 3) + f + 2] = lend;
      }
    }
    if (sh.best < ctl->best_next) ctl->best_next = sh.best;
    ctl->cnt_Z += z;
    ctl->cnt_tok += (u64)nf;
    if (nf > ctl->peak_tokens) ctl->peak_tokens = nf;
    ctl->front_begin = base;
    ctl->front_count = nf;
    ctl->n_decoded = f + 1;
    ctl->eps_occ = 0;
    ctl->wl_n = 0;
    ctl->active = 0;
    if (err) ctl->error |= err;
    dbg_phase(D, 4, tq);
  }
  __syncthreads();
  return true;
}

// exact k-th smallest (0-based) cost of the frontier: what std::nth_element leaves at
// _tmp_array[k] (base-inl.h:190-193, 211-216).  MSB-first radix select, 8 bits per pass, LDS
// histogram.  The first kSelKeep * 1024 costs stay in registers across the four passes (only a
// longer frontier is re-read from HBM), and the bin holding the k-th element is found by a
// wave-parallel prefix scan of the histogram (a serial scan by one thread cost ~7 us per pass).
constexpr int kSelKeep = 8;

// kT threads; kSc1: the costs are read with agent-scope (sc1) loads -- the tokens were written, write-through, by OTHER workgroups
// of this launch (frame_boundary_fused's slow path; cdna_hip_programming.md Guideline 16: every load of handed-off bytes).
template <int kT, bool kSc1, class Sh, int kKeep>
__device__ __forceinline__ float kth_smallest_t(const int4 *tok, int n, int k, Sh &sh, uint32_t *cache, int cache_cap, uint32_t lo_o, uint32_t hi_o) {
  // cache / cache_cap: LDS that is free during the selection (the insert workgroup's hash table): the orderable costs beyond the
  // ones kept in registers are parked there on the first pass instead of being read from memory again on the later ones.
  // lo_o / hi_o: orderable bounds of EVERY cost of the frame (the best token's cost; the frame's final next_cutoff, which every
  // token lies below) or 0 / ~0.  The bits the two share are the selection's prefix from the start: a frame's costs sit within
  // one beam of each other, so their top 12-16 bits agree -- a first pass over those bits put every token into ONE histogram bin,
  // n serialised LDS atomics on one address (33 us for the frame boundary of a 20 k-token frame, measured), and decided nothing.
  int tid = threadIdx.x;
  if (kSc1) asm volatile("" : "+v"(tid));   // (inside the insert kernel: see frame_boundary_fused)
  const int lane = tid & 63;
  auto cost_of = [&](int i) -> uint32_t {
    const int *p = reinterpret_cast<const int *>(tok + i) + 1;
    const int bits = kSc1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
    return f2o(__int_as_float(bits));
  };
  constexpr int kSelBatch = kSc1 ? 4 : 8;   // loads of costs in flight per thread beyond the kept ones (the insert kernel has 80 registers)
  uint32_t keep[kKeep];
#pragma unroll
  for (int j = 0; j < kKeep; ++j) {
    const int i = j * kT + tid;
    keep[j] = i < n ? cost_of(i) : 0xFFFFFFFFu;  // orderable +NaN: sorts last
  }
  const int n_cached = min(n, kKeep * kT + cache_cap);   // costs [kKeep * kT, n_cached) live in `cache` after the first pass
  if (hi_o <= lo_o) { lo_o = 0u; hi_o = 0xFFFFFFFFu; }
  const uint32_t diff = lo_o ^ hi_o;
  const int R = 32 - __clz((int)diff);   // low bits in which the costs can differ (diff != 0 here)
  const int n_pass = (R + 7) >> 3;
  if (tid == 0) { sh.sel_prefix = R >= 32 ? 0u : ((lo_o >> R) << R); sh.sel_k = (uint32_t)k; }
  for (int pass = 0; pass < n_pass; ++pass) {
    const int top = R - 8 * pass;            // this pass decides bits [shift, top)
    const int shift = max(0, top - 8);
    const uint32_t dmask = (1u << (top - shift)) - 1u;
    const uint32_t hi_mask = top >= 32 ? 0u : (0xFFFFFFFFu << top);
    for (int b = tid; b < 256; b += kT) sh.hist[b] = 0;
    __syncthreads();
    const uint32_t prefix = sh.sel_prefix;
#pragma unroll
    for (int j = 0; j < kKeep; ++j) {
      if (j * kT + tid < n && (keep[j] & hi_mask) == (prefix & hi_mask)) atomicAdd(&sh.hist[(keep[j] >> shift) & dmask], 1u);
    }
    // (the costs beyond the registers, kSelBatch loads in flight per thread: one after the other -- a load, its histogram atomic,
    // the next load -- a 20 k-token frame was forty dependent round trips per pass)
    for (int i0 = kKeep * kT + tid; i0 < n; i0 += kSelBatch * kT) {
      uint32_t o[kSelBatch];
#pragma unroll
      for (int u = 0; u < kSelBatch; ++u) {
        const int i = i0 + u * kT;
        o[u] = 0xFFFFFFFFu;
        if (i < n) o[u] = (i < n_cached && pass != 0) ? cache[i - kKeep * kT] : cost_of(i);
      }
#pragma unroll
      for (int u = 0; u < kSelBatch; ++u) {
        const int i = i0 + u * kT;
        if (i >= n) continue;
        if (pass == 0 && i < n_cached) cache[i - kKeep * kT] = o[u];   // (each thread re-reads its own entries: no barrier needed)
        if ((o[u] & hi_mask) == (prefix & hi_mask)) atomicAdd(&sh.hist[(o[u] >> shift) & dmask], 1u);
      }
    }
    __syncthreads();
    if (tid < 64) {  // one wave: lane l owns bins 4l..4l+3
      const uint32_t c0 = sh.hist[4 * lane], c1 = sh.hist[4 * lane + 1], c2 = sh.hist[4 * lane + 2], c3 = sh.hist[4 * lane + 3];
      uint32_t incl = (uint32_t)wave_incl_scan((int)(c0 + c1 + c2 + c3));
      const uint32_t kk = sh.sel_k;
      const u64 m = __ballot(incl > kk);  // first lane whose cumulative count passes k
      const int owner = m ? __ffsll((long long)m) - 1 : 63;
      if (lane == owner) {
        uint32_t cum = incl - (c0 + c1 + c2 + c3);
        int b = 4 * lane;
        if (kk >= cum + c0) { cum += c0; ++b; if (kk >= cum + c1) { cum += c1; ++b; if (kk >= cum + c2) { cum += c2; ++b; } } }
        sh.sel_prefix = prefix | ((uint32_t)b << shift);
        sh.sel_k = kk - cum;
      }
    }
    __syncthreads();
  }
  return o2f(sh.sel_prefix);
}
__device__ __forceinline__ float kth_smallest(const int4 *tok, int n, int k, BoundaryShared &sh, uint32_t lo_o = 0u, uint32_t hi_o = 0xFFFFFFFFu) {
  return kth_smallest_t<kBT, false, BoundaryShared, kSelKeep>(tok, n, k, sh, nullptr, 0, lo_o, hi_o);
}
// The selection of frame_boundary_fused's slow path, OUT OF LINE: a handful of frames take it, and inlined its registers (the costs
// kept across the four passes) and its code sit in the insert kernel's hot loops' allocation (insert +8 % per launch, measured).
// Nothing of the decoder block is passed: a function taking it by reference makes the compiler copy the kernel argument to scratch.

template <bool kBig>
__device__ __forceinline__ void prep_frame(const DecoderDev &D, int c, ChanCtl *ctl, const int32_t *target, BoundaryShared &sh,
                           int group, int par) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float kInf = __builtin_huge_valf();
  unsigned long long tq = wall_clock64();
  if (tid == 0) {
    int act = (ctl->n_decoded < target[c]) && ctl->error == 0 && !ctl->finalized;
    if (act && ctl->n_decoded >= D.max_frames) { ctl->error |= kErrFramesFull; act = 0; }
    sh.active = act;
    if (!act) ctl->active = 0;
  }
  __syncthreads();
  if (!sh.active) return;
  const int n = ctl->front_count;
  const int4 *tokc = D.tok + (size_t)c * D.arena_cap;
  const int4 *tok = tokc + ctl->front_begin;
  const u64 best = ctl->best_next;  // min (cost, arena index) over the frontier
  const float best_w = n > 0 ? o2f((uint32_t)(best >> 32)) : kInf;

  // GetCutoff, base-inl.h:138-234
  float cutoff, ab;
  int super_k = 1;
  if (D.max_active == 2147483647 && D.min_active == 0 && !D.soft_limit) {
    ab = D.beam;
    cutoff = best_w + D.beam;
  } else {
    const float beam_cutoff = best_w + D.beam;
    float min_active_cutoff = kInf, max_active_cutoff = kInf;
    // the selection's range (kth_smallest_t): fused decoders' frames lie between the best token's cost and the frame's final
    // next_cutoff (every token was admitted below it; the initial frame, built by InitDecoding, has none: bound 0)
    const uint32_t sel_lo = (D.fused && !kBig) ? f2o(best_w) : 0u, sel_hi = (D.fused && !kBig) ? ctl->bound : 0xFFFFFFFFu;
    // (soft_limit decoders: the per-frame token limit acts as a max_active -- the frame holds every token the arena took, the
    // expansion goes on from the limit-th cheapest: what the reference does at that max_active, base-inl.h:188-203)
    const int max_eff = D.soft_limit ? min(D.max_active, D.max_tok) : D.max_active;
    if (n > max_eff) {
      max_active_cutoff = kth_smallest(tok, n, max_eff, sh, sel_lo, sel_hi);
      if (tid == 0 && D.soft_limit && D.max_tok < D.max_active) D.degraded[c] += 1;
      if (max_active_cutoff < beam_cutoff) super_k = super_tile_tokens(n, max_eff);   // (compacting tiles: see frame_boundary_fused)
    }
    if (max_active_cutoff < beam_cutoff) {
      ab = max_active_cutoff - best_w + D.beam_delta;
      cutoff = max_active_cutoff;
    } else {
      if (n > D.min_active) {
        // (fused decoders whose frame was built under the plain beam: every token lies below best + beam -- frame_boundary_fused --,
        // the min_active-th cheapest too: any value at or below it decides the same, without the selection)
        const bool below_beam = D.fused && !kBig && ctl->n_decoded > 0 && ctl->adaptive_beam == D.beam;
        if (D.min_active == 0 || below_beam) min_active_cutoff = best_w;
        else min_active_cutoff = kth_smallest(tok, n, D.min_active, sh, sel_lo, sel_hi);
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

  // seed next_cutoff from the best token's emitting arcs, base-inl.h:282-300 (seed_tiles: left to the expansion's seed tile)
  float seed = kInf;
  if (n > 0 && !(!kBig && D.seed_tiles)) {
    int4 bt;
    if (D.best_row) bt = make_int4((int)(uint32_t)best, __float_as_int(best_w), 0, 0);   // the best token's row rides in best_next itself
    else bt = tokc[(uint32_t)best];
    const uint2 si = make_uint2((uint32_t)bt.x + 1u, (uint32_t)D.g.arcs[bt.x].x);
    const int deg = (int)(si.y >> kEpsBits), ab0 = (int)(si.x + (si.y & kEpsMask));
    const float *llrow = D.ll_base[c] + (size_t)ctl->n_decoded * D.stride;
    const float bc = __int_as_float(bt.y);
    const int blm = kBig ? D.tok_lm[(size_t)c * D.arena_cap + (uint32_t)best] : 0;
    for (int e = tid; e < deg; e += kBT) {
      const int4 arc = D.g.arcs[ab0 + e];
      float tot_score;
      if constexpr (kBig) {  // biglm.h:350-353: lm_score + tot_cost + weight - loglike (the pair is not interned here)
        const int ol = D.g.arc_olabel[ab0 + e];
        int n1, n2;
        const float lm_score = ol != 0 ? lm_step(D, c, blm, ol, &n1, &n2) : 0.0f;
        tot_score = ((lm_score + bc) + __int_as_float(arc.z)) - llrow[arc.x & D.g.col_mask];
      } else {
        tot_score = (bc + __int_as_float(arc.z)) - llrow[arc.x & D.g.col_mask];  // base-inl.h:295
      }
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
    ctl->new_count = 0;
    ctl->best_next = ~0ull;
    ctl->active = 1;
    // publish this channel's tiles for the expansion (any disjoint range will do).  Tile size: a tile's sort rounds are
    // serial (candidates / 512 per round), so smaller tiles shorten the launch -- as long as all tiles of the launch are
    // resident at once (~1500 workgroups; other groups' launches share them).  Judged per channel on its own token
    // count times the channels of the launch: 16 channels x 4.3 k tokens get 128-token tiles (16.7 -> 14.1 ms per step),
    // 64 channels and more the full 512.
    int tile_tokens = D.staged ? D.st_tile_tokens : kTileTokens;
    if ((int64_t)n * (int)gridDim.x <= 700ll * 256) tile_tokens = 256;
    if ((int64_t)n * (int)gridDim.x <= 700ll * 128) tile_tokens = 128;
    // (biglm at 64 tokens per tile -- one candidate, one LM walk per thread -- measured the same: 32.4 vs 32.1 ms per step)
    if (D.staged) tile_tokens = min(tile_tokens, D.st_tile_tokens);
    if (D.staged && !kBig && super_k > 1) tile_tokens = super_k;
    sh.tile_tokens = tile_tokens;
    const int ntiles = (n + tile_tokens - 1) / tile_tokens;
    sh.active = 0;
    const int nseed = (!kBig && D.seed_tiles && ntiles > 0) ? 1 : 0;   // the seed tile, listed first
    ctl->tiles_left = ntiles + nseed;
    if (ntiles == 0 && D.two_launch) {   // a channel without tokens: no tile will plan its insert items, yet its frame must be closed
      ctl->items_left = ctl->stores_left = 1;
      push_empty_item(D, c, group, par);
    }
    if (ntiles > 0) {
      const int start = atomicAdd(&D.fctl[group].total_tiles[par], ntiles + nseed);
      if (start + ntiles + nseed <= D.tile_cap) { sh.sel_k = (uint32_t)start; sh.active = ntiles + nseed; }
      else ctl->error |= kErrFrontierFull;
    }
  }
  __syncthreads();
  const int ntl = sh.active;
  const int nseed = (!kBig && D.seed_tiles && ntl > 0) ? 1 : 0;
  TileDesc *tiles = D.tiles + (size_t)group * D.tile_cap + sh.sel_k;
  for (int i = tid; i < ntl; i += kBT) {
    TileDesc td;
    td.chan = c;
    const int k = i - nseed;
    td.tok_begin = ctl->front_begin + k * sh.tile_tokens;
    td.tok_count = min(sh.tile_tokens, n - k * sh.tile_tokens);
    td.cutoff = cutoff;
    td.adaptive_beam = ab;
    td.pad = (int32_t)ctl->bound;   // next_cutoff's seed of the frame (the replay instantiations of the expansion start from it)
    td.llrow = D.ll_base[c] + (size_t)ctl->n_decoded * D.stride;
    if (k < 0) {   // the seed tile: the best token's row and cost
      td.tok_begin = (int32_t)(uint32_t)best;
      td.tok_count = 0;
      td.cutoff = best_w;
    }
    tiles[i] = td;
  }
  if (tid == 0) dbg_phase(D, 5, tq);
}

// =========================================================================================
// Two launches per frame (DecoderDev::two_launch): what closure_kernel<false,false> does for a fused best-path decoder -- close
// the frame the insert launch has just built (finalize_frame) and prepare the next one (prep_frame: GetCutoff, the seed of
// next_cutoff from the best token's arcs, the tile list) -- run by the insert workgroup that finishes the channel's last work
// item, kT threads.  Preconditions (wfst_decoder_create): max_active can never bind and min_active is 0, so GetCutoff is
// best + beam (base-inl.h:138-234 with both limits out of reach) and needs no look at the frame's tokens; the best token's
// cost AND graph row arrive in ChanCtl::best_next (atomicMin), so nothing another workgroup wrote with plain stores in this
// launch is read here: the fields other workgroups changed -- by atomics -- are read through L2 (ld_agent).  Float arithmetic
// as prep_frame's.
// =========================================================================================
template <int kT>
__device__ __forceinline__ void frame_boundary_fused(const DecoderDev &D, int c, const int32_t *target, int chan_cnt, int group,
                                                     int par_next, bool do_prep, BoundaryLite &sh, uint32_t *sel_cache, int sel_cache_cap) {
  // (an opaque copy of the thread index: what the boundary derives from it -- addresses of the tile list, of the counters -- is
  // then computed HERE; hoisted to the kernel's entry, as loop invariants of the item loop, those values lived across the insert
  // passes and were spilled there: 0.2 ms per step per spilled register, the spill stores sit in front of an item's first loads)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  ChanCtl *ctl = D.ctl + c;
  const float kInf = __builtin_huge_valf();
  // every item of the channel is done with the bucket counters: reset for the next expansion
  for (int i = tid; i < D.n_part; i += kT) D.bucket_cnt[(size_t)c * D.n_part + i] = 0;
  if (tid == 0) {
    const int f = ctl->n_decoded;                         // (unchanged during this launch)
    const int base = ctl->front_begin + ctl->front_count;
    const uint32_t bound_o = sh.h_bound;
    int nf = sh.h_nf;
    const int err = sh.h_err;
    const u64 best = D.best_exp ? sh.h_best : ld_agent(&ctl->best_next);
    int add_err = 0;
    // (the per-frame token limit is a max_active here, not a capacity: DecoderDev::soft_limit; the arena is one)
    if ((!D.soft_limit && nf > D.max_tok) || (int64_t)base + nf > D.arena_cap) nf = 0;   // (the insert workgroups have raised the error bit)
    if (err) nf = 0;   // a channel that hit a limit stops producing tokens
    if (f + 2 > D.max_frames + 1) add_err = kErrFramesFull;
    else {
      D.frame_off[(size_t)c * (D.max_frames + 2) + f + 2] = base + nf;
      D.cutoff_hist[(size_t)c * (D.max_frames + 2) + f + 1] = o2f(bound_o);
    }
    ctl->cnt_tok += (u64)nf;
    if (nf > ctl->peak_tokens) ctl->peak_tokens = nf;
    sh.prev_ab = ctl->adaptive_beam;   // (of the frame being closed)
    ctl->front_begin = base;
    ctl->front_count = nf;
    ctl->n_decoded = f + 1;
    ctl->active = 0;
    if (add_err) atomicOr(&ctl->error, add_err);
    // the next frame (prep_frame)
    int act = do_prep && (f + 1 < target[c]) && !(err | add_err) && !ctl->finalized;
    if (act && f + 1 >= D.max_frames) { atomicOr(&ctl->error, kErrFramesFull); act = 0; }
    sh.active = act;
    sh.n = nf;
    sh.nd = f + 1;
    sh.front_begin = base;
    sh.best = best;
  }
  __syncthreads();
  if (!sh.active) return;
  const int n = sh.n, nd = sh.nd;
  const u64 best = sh.best;
  const float best_w = n > 0 ? o2f((uint32_t)(best >> 32)) : kInf;
  // GetCutoff (base-inl.h:138-234), as prep_frame computes it.  The frame's tokens are needed only where a limit can bind:
  //  * more tokens than max_active (or than the per-frame limit, which acts as one): the exact max_active-th cheapest cost;
  //  * min_active: the frame was built below next_cutoff = best + adaptive_beam of the frame before (every candidate, emitting
  //    or epsilon arrival, costs at least the cheapest emitting one, whose cost + adaptive_beam IS the final next_cutoff:
  //    min(x) + b == min(x + b)); while that adaptive beam was the plain beam every token lies below best + beam, the
  //    min_active-th cheapest too, and the cutoff is best + beam without a look at the tokens; after a frame whose cutoff
  //    min_active (or max_active) set, the selection runs.
  // Those frames read the tokens the OTHER insert workgroups of this launch wrote: stored write-through (sc1), drained by every
  // storing wave, counted down in ChanCtl::stores_left behind the drain, read here with sc1 loads once that count is zero
  // (cdna_hip_programming.md Guideline 16; MI355X_MICROARCH.md, Valid forms, first row of the table).
  const float beam_cutoff = best_w + D.beam;
  const int max_eff = D.soft_limit ? min(D.max_active, D.max_tok) : D.max_active;
#ifdef WFST_LEAN_BOUNDARY   // (A/B builds: the boundary without its slow path)
  const bool need_max = false, need_min = false;
#else
  const bool need_max = n > max_eff;
  const bool need_min = D.min_active > 0 && n > D.min_active && !(sh.prev_ab == D.beam);
#endif
  float ab = D.beam, cutoff = beam_cutoff;
  if (need_max || need_min) {   // (uniform over the workgroup)
    if (tid == 0) {
      if (!sh.h_risky) atomicOr(&ctl->error, kErrInternal);   // (plan_channel's test covers every such frame: never expected)
      int spins = 0;
      while (sh.h_risky && ld_agent(&ctl->stores_left) != 0) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 22)) { atomicOr(&ctl->error, kErrInternal); break; }   // (never expected; reported, not hung on)
      }
      if (need_max && D.max_tok < D.max_active) D.degraded[c] += 1;   // (the per-frame limit, not the caller's max_active, binds)
    }
    __syncthreads();
    const int4 *tok = D.tok + (size_t)c * D.arena_cap + sh.front_begin;
    float min_active_cutoff = kInf, max_active_cutoff = kInf;
    if (n > D.min_active) min_active_cutoff = best_w;   // (min_active 0, or every token below best + beam: any value at or below it decides the same)
    // (ONE call site for the selection -- the max_active-th cheapest first, the min_active-th only where that does not bind:
    // inlined twice the selection was 12 KB of the insert kernel's 28, and the launch 4 % slower for code that rarely runs)
#pragma nounroll
    for (int stage = need_max ? 0 : 1; stage < 2; ++stage) {
      if (stage == 1 && !need_min) break;
      const float v = kth_smallest_cold<kT>(tok, n, stage == 0 ? max_eff : D.min_active, &sh, sel_cache, sel_cache_cap,
                                            f2o(best_w), sh.h_bound);   // (every token of the frame costs at least the best one's and less than its final next_cutoff)
      if (stage == 0) {
        max_active_cutoff = v;
        if (max_active_cutoff < beam_cutoff) break;
      } else {
        min_active_cutoff = v;
      }
    }
    if (max_active_cutoff < beam_cutoff) {
      ab = max_active_cutoff - best_w + D.beam_delta;
      cutoff = max_active_cutoff;
    } else if (min_active_cutoff > beam_cutoff) {
      ab = min_active_cutoff - best_w + D.beam_delta;
      cutoff = min_active_cutoff;
    }
  } else if (n <= D.min_active) {   // count <= min_active: min_active_cutoff stays +inf (base-inl.h:205-226)
    if (kInf > beam_cutoff) { ab = kInf - best_w + D.beam_delta; cutoff = kInf; }
  }
  const float *llrow = D.ll_base[c] + (size_t)nd * D.stride;
  // seed next_cutoff from the best token's emitting arcs, base-inl.h:282-300 (seed_tiles: left to the expansion's seed tile)
  float seed = kInf;
  if (n > 0 && !D.seed_tiles) {
    const int brow = (int)(uint32_t)best;
    const uint32_t hx = (uint32_t)D.g.arcs[brow].x;
    const int deg = (int)(hx >> kEpsBits), ab0 = brow + 1 + (int)(hx & kEpsMask);
    for (int e = tid; e < deg; e += kT) {
      const int4 arc = D.g.arcs[ab0 + e];
      seed = fminf(seed, (best_w + __int_as_float(arc.z)) - llrow[arc.x & D.g.col_mask]);  // base-inl.h:295
    }
  }
  if (!D.seed_tiles) {   // (uniform over the workgroup)
    seed = wave_min_f(seed);
    if (lane == 0) sh.redf[wave] = seed;
    __syncthreads();
  }
  if (tid == 0) {
    float s = kInf;
    if (!D.seed_tiles) for (int w = 0; w < kT / 64; ++w) s = fminf(s, sh.redf[w]);
    const float next_cutoff = s + ab;
    ctl->cur_cutoff = cutoff;
    ctl->adaptive_beam = ab;
    ctl->bound = f2o(s < kInf ? next_cutoff : kInf);
    ctl->new_count = 0;
    ctl->best_next = ~0ull;
    ctl->active = 1;
    int tile_tokens = D.staged ? D.st_tile_tokens : kTileTokens;   // (as prep_frame)
    if ((int64_t)n * chan_cnt <= 700ll * 256) tile_tokens = 256;
    if ((int64_t)n * chan_cnt <= 700ll * 128) tile_tokens = 128;
    if (D.staged) tile_tokens = min(tile_tokens, D.st_tile_tokens);
    // a binding max_active (or per-frame limit) leaves n - max_eff tokens above the cutoff: compacting tiles (expand_kernel_staged)
    // of as many tokens as hold one tile's worth of live ones
    if (D.staged && need_max && cutoff < beam_cutoff) tile_tokens = super_tile_tokens(n, max_eff);
    const int ntiles = (n + tile_tokens - 1) / tile_tokens;
    const int nseed = (D.seed_tiles && ntiles > 0) ? 1 : 0;   // the seed tile, listed first
    sh.tile_tokens = tile_tokens;
    sh.ntiles = 0;
    ctl->tiles_left = ntiles + nseed;
    if (ntiles == 0) {   // a channel without tokens: an empty insert item closes its next frame
      ctl->items_left = ctl->stores_left = 1;
      push_empty_item(D, c, group, par_next);
    } else {
      const int start = atomicAdd(&D.fctl[group].total_tiles[par_next], ntiles + nseed);
      if (start + ntiles + nseed <= D.tile_cap) { sh.tile_start = start; sh.ntiles = ntiles + nseed; }
      else atomicOr(&ctl->error, kErrFrontierFull);
    }
    sh.pad_bits = (int32_t)f2o(s < kInf ? next_cutoff : kInf);   // next_cutoff's seed as the tiles carry it (TileDesc::pad)
  }
  __syncthreads();
  const int ntl = sh.ntiles;
  TileDesc *tiles = D.tiles + (size_t)group * D.tile_cap + sh.tile_start;
  const int32_t pad_bits = sh.pad_bits;
  const int nseed = (D.seed_tiles && ntl > 0) ? 1 : 0;
  for (int i = tid; i < ntl; i += kT) {
    TileDesc td;
    td.chan = c;
    const int k = i - nseed;
    td.tok_begin = sh.front_begin + k * sh.tile_tokens;
    td.tok_count = min(sh.tile_tokens, n - k * sh.tile_tokens);
    td.cutoff = cutoff;
    td.adaptive_beam = ab;
    td.pad = pad_bits;
    td.llrow = llrow;
    if (k < 0) {   // the seed tile: the best token's row and cost
      td.tok_begin = (int32_t)(uint32_t)best;
      td.tok_count = 0;
      td.cutoff = best_w;
    }
    tiles[i] = td;
  }
}

// =========================================================================================
// Lattice-beam back-pruning (lattice mode): PruneActiveTokens every prune_interval frames
// (base-inl.h:438-480, called at :660-661), FinalizeDecoding (PruneForwardLinksFinal + PruneForwardLinks
// + PruneTokensForFrame, :482-607, 725-847), and the COMPACTION that keeps the token arena and the link
// store bounded; one 1024-thread workgroup per channel.
//
//   extra[t]  = min over t's links of (extra[next] + (link cost - cost_next)), links with more than
//               lattice_beam dropped.  The newest frame is the seed: extra 0 for every token
//               (PruneActiveTokens), or cost + final_cost - best (FinalizeDecoding).  link cost = (cost_t + ac)
//               + graph is the candidate cost the expansion computed, kept in the link record, so one 8-byte
//               gather {extra, cost} of the destination prices a link.
//   Frames are walked newest to oldest and the walk STOPS at the first frame none of whose extras moved by more
//   than delta = lattice_beam * prune_scale against the previous pass's -- the reference's own stopping rule
//   (extra_costs_changed, :458-461, :541-542), judged here on the frame's exact fixpoint where the reference judges
//   sweep by sweep over its token list (order dependent): a running pass may therefore stop at another frame than
//   the reference's, and a mid-utterance lattice may differ from the reference's at that moment by the links that
//   difference prices (it equals the order-free oracle's); FinalizeDecoding (delta 0, every frame walked, in the
//   reference too) ends at the same lattice either way.
//   The reference reaches the fixpoint inside a frame by sweeping token lists "while changed"; min is
//   order-independent, so atomicMin relaxation gives the same values.
//   Survivors are then moved down over the dead (tokens frame by frame, links segment by segment, indices
//   remapped, backpointers and the frontier included): the arena holds the surviving history plus the
//   raw frames since the last pass -- bounded, whatever the utterance length.
// =========================================================================================
constexpr int kPrLds = 16384;    // {extra, cost} pairs of the walk kept in LDS (128 KB): the frame being priced, and the frame after it where both fit
#ifndef WFST_PR_CHUNK
#define WFST_PR_CHUNK 8192
#endif
constexpr int kPrChunk = WFST_PR_CHUNK;   // items of one compaction sweep (kBT threads x 8)
#ifndef WFST_PR_SLABS
#define WFST_PR_SLABS 8
#endif
constexpr int kPrSlabs = WFST_PR_SLABS;      // workgroups per channel of a compaction's flag sweeps (prune_flags; round 5: 4 -> 8)
constexpr int kPrSlabBase = 40;  // their survivor counts in the channel's parameter block: [40, 40 + 2 x kPrSlabs)
constexpr int kPrParInts = kPruneParInts;   // a channel's parameter block (DecoderDev::prune_par): [0, 16) the compaction's {run, c_lo, tokens lo / hi, frame-0 bound, links lo / hi, nd, slab counts}; [16, 32) lattice_emit's counters; [60, 62) the closure launch's meeting word (kClSlabWord); then:
constexpr int kPrRawCount = 32;  // ... and of the raw frames' launches: workgroups of the channel that have finished their share of the frame,
constexpr int kPrRawChg = 34;    // [3]: "an extra moved" of an epsilon round, in rotation
#ifndef WFST_PR_RAW_J
#define WFST_PR_RAW_J 16
#endif
constexpr int kPrRawJ = WFST_PR_RAW_J;      // workgroups per channel and raw frame
struct ScanShared {
  u64 red[2][kBT / 64];
  int changed, any_changed, cnt, err;
  int wsum[kBT / 64];
  int flag[3];
};
struct PruneShared : ScanShared {
  // the walk: {orderable extra, cost bits} of the frame being priced and of the frame after it; the compaction (which runs after the walk) reuses the space for a sweep's exclusive prefix
  union {
    struct { uint2 e[kPrLds]; } w;
    int pre[kPrChunk + 1];
  };
};

// exclusive prefix sum of `v` over the workgroup's kBT threads; *total = sum
__device__ __forceinline__ int block_exscan(int v, ScanShared &ps, int *total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int incl = wave_incl_scan(v);
  __syncthreads();  // ps.wsum free again
  if (lane == 63) ps.wsum[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kBT / 64; ++w) {
    const int x = ps.wsum[w];
    if (w < wave) base += x;
    tot += x;
  }
  *total = tot;
  return base + incl - v;
}

// kFinal: FinalizeDecoding.  Returns with extras valid for every frame, dead tokens and links gone, and
// ctl->pruned_upto = n_decoded.
// raw_done (running passes only): the frames never priced before, [pruned_upto, n_decoded), have been priced by the launches of
// lattice_prune_raw_* (several workgroups per channel and frame, below): the walk starts at frame pruned_upto - 1.
template <bool kFinal>
__device__ __forceinline__ void prune_pass(const DecoderDev &D, int c, PruneShared &ps, bool raw_done = false) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ChanCtl *ctl = D.ctl + c;
  const int nd = ctl->n_decoded;
  int4 *tok = D.tok + (size_t)c * D.arena_cap;
  int4 *links = D.links + (size_t)c * D.link_cap;
  uint2 *extra = D.extra + (size_t)c * D.arena_cap;
  int32_t *remap = D.remap + (size_t)c * D.arena_cap;   // previous extras while walking, new indices while compacting
  int32_t *foff = D.frame_off + (size_t)c * (D.max_frames + 2);
  int32_t *loff = D.link_off + (size_t)c * (D.max_frames + 3);
  int32_t *lmid = D.link_mid + (size_t)c * (D.max_frames + 3);
  const float kInf = __builtin_huge_valf();
  const uint32_t kInfO = f2o(kInf);
  const float lb = D.lattice_beam;
  if (ctl->error) return;
  unsigned long long tq = wall_clock64();
  const int n_prev = ctl->pruned_upto;   // frames below hold the extras of an earlier pass
  const int fn = foff[nd], fn1 = foff[nd + 1];
  bool any_final = false;

  // link_extra = extra of the destination + (link cost - cost of the destination) (base-inl.h:524-526,
  // 782-784), from one 8-byte gather {extra (low), cost (high)}; +inf for a dead destination
  // A pass is a chain of short phases over a few thousand links or tokens each: latency, not bytes.  Every
  // thread therefore takes kPU items of a phase at once -- the link loads together, then the gathers of the
  // destinations' {extra, cost}, then the atomics -- instead of one dependent chain per item.
  constexpr int kPU = 6;   // (8: the same; 10: spills, +4 %)
  // f(i, L, le): link i = L is alive and its link_extra is le (maybe above lattice_beam)
  auto for_links = [&](int lo, int hi, auto &&f) {
    for (int i0 = lo; i0 < hi; i0 += kBT * kPU) {
      int4 L[kPU];
      u64 e[kPU];
#pragma unroll
      for (int u = 0; u < kPU; ++u) {
        const int i = i0 + u * kBT + tid;
        L[u] = i < hi ? links[i] : make_int4(-1, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < kPU; ++u) e[u] = L[u].x >= 0 ? ld_agent(reinterpret_cast<const u64 *>(&extra[L[u].y])) : 0ull;
#pragma unroll
      for (int u = 0; u < kPU; ++u) {
        if (L[u].x < 0) continue;
        const uint32_t eo = (uint32_t)e[u];
        const float le = eo >= kInfO ? kInf : o2f(eo) + (D.link_delta ? __int_as_float(L[u].w) : __int_as_float(L[u].w) - __int_as_float((int)(e[u] >> 32)));
        f(i0 + u * kBT + tid, L[u], le);
      }
    }
  };
  // the epsilon links of a frame, [e0, e1), to their fixpoint; then the dead ones are marked
  auto relax_eps = [&](int e0, int e1) {
    if (e0 >= e1) return;
    for (int round = 0; round < 4096; ++round) {
      if (tid == 0) ps.changed = 0;
      __syncthreads();
      for_links(e0, e1, [&](int, const int4 &L, float le) {
        if (!(le <= lb)) return;
        if (le < 0.0f) le = 0.0f;
        const uint32_t o = f2o(le);
        if (o < atomicMin(&extra[L.x].x, o)) ps.changed = 1;
      });
      __syncthreads();
      const int ch = ps.changed;
      __syncthreads();
      if (!ch) break;
    }
    for_links(e0, e1, [&](int i, const int4 &, float le) {
      if (!(le <= lb)) links[i].x = -1;
    });
    __syncthreads();
  };

  // ---- (1) the newest frame ---------------------------------------------------------------------
  if (kFinal) {
    // ComputeFinalCosts (base-inl.h:670-720) + PruneForwardLinksFinal (:725-824)
    // (biglm: ComputeFinalCosts of biglm.h:160-215 -- the LM's final cost enters best_cost_with_final for EVERY token, final
    // in the graph or not; only graph-final tokens are final, each with its LM final cost)
    const int32_t *tlm = D.big ? D.tok_lm + (size_t)c * D.arena_cap : nullptr;
    const u64 *pkeys = D.big ? D.pair_keys + (size_t)c * D.pair_cap : nullptr;
    auto lm_final = [&](int i) -> float {
      const u64 pk = pkeys[tlm[i]];
      return lm_final_cost(D.lm_old, (int)(uint32_t)pk) + lm_final_cost(D.lm_new, (int)(uint32_t)(pk >> 32));   // diff-lm.h:48-53
    };
    u64 b_all = ~0ull, b_fin = ~0ull;
    int any_fin = 0;
    for (int i = fn + tid; i < fn1; i += kBT) {
      const int4 t = tok[i];
      const u64 v = (u64)f2o(__int_as_float(t.y));
      b_all = v < b_all ? v : b_all;
      if (D.big) {
        const u64 w = (u64)f2o(__int_as_float(t.y) + lm_final(i));
        b_fin = w < b_fin ? w : b_fin;                      // best_cost_with_final: over all tokens
        any_fin |= t.x == D.g.final_state;
      } else if (t.x == D.g.final_state) b_fin = v < b_fin ? v : b_fin;
    }
    b_all = wave_min_u64(b_all);
    b_fin = wave_min_u64(b_fin);
    if (lane == 0) { ps.red[0][wave] = b_all; ps.red[1][wave] = b_fin; }
    __syncthreads();
    for (int w = 0; w < kBT / 64; ++w) { b_all = ps.red[0][w] < b_all ? ps.red[0][w] : b_all; b_fin = ps.red[1][w] < b_fin ? ps.red[1][w] : b_fin; }
    any_final = b_fin != ~0ull;
    if (D.big) {   // the final-cost set is non-empty iff some token is final in the graph
      if (tid == 0) ps.cnt = 0;
      __syncthreads();
      if (any_fin) ps.cnt = 1;
      __syncthreads();
      any_final = ps.cnt != 0;
    }
    const float final_best = o2f((uint32_t)(b_fin != ~0ull ? b_fin : b_all));
    for (int i = fn + tid; i < fn1; i += kBT) {
      const int4 t = tok[i];
      const float final_cost = !any_final ? 0.0f : t.x == D.g.final_state ? (D.big ? lm_final(i) : 0.0f) : kInf;
      float e = __int_as_float(t.y) + final_cost - final_best;  // base-inl.h:775
      if (e > lb) e = kInf;                                      // base-inl.h:815-816 (tokens without links)
      extra[i] = make_uint2(f2o(e), (uint32_t)t.y);
    }
    __syncthreads();
    relax_eps(lmid[nd], loff[nd + 1]);
  } else if (!raw_done) {
    // (link_delta: nothing reads a pair's cost half -- it is not fetched from the tokens either, here and below)
    for (int i = fn + tid; i < fn1; i += kBT) extra[i] = make_uint2(f2o(0.0f), D.link_delta ? 0u : (uint32_t)tok[i].y);
    __syncthreads();
  }

  // ---- (2) older frames, newest first ---------------------------------------------------------------
  // PruneActiveTokens walks on while a frame's extra costs moved by more than delta = lattice_beam * prune_scale
  // (base-inl.h:452-461; frames never priced before are always priced); FinalizeDecoding walks every frame
  // (delta 0, :838-843).  "Moved" is judged on the frame's exact fixpoint against the value before the pass
  // (the reference judges sweep by sweep over its token list: oracle/wfst_oracle.c prune_forward_links).
  //
  // A frame's pricing is a chain -- links -> {extra, cost} of their destinations -> atomicMin on their sources ->
  // (epsilon links: again, to the fixpoint) -- and a pass walks a hundred and more frames: with the pairs in HBM every
  // step of it was a dependent round trip.  A frame of up to kPrLds tokens (every frame an earlier pass has pruned; most
  // raw frames at beam 13) is therefore priced IN LDS: its pairs are built there from one coalesced read of the
  // tokens' costs, the frame after it is still there from the step before, the links come in by one coalesced read,
  // and the pairs go back to HBM with one coalesced store -- one round trip per frame instead of five.  Larger
  // frames take the HBM path (for_links / relax_eps above).
  const float delta = kFinal ? 0.0f : D.lattice_beam * D.prune_scale;
  int k_lo = nd;   // oldest frame re-priced by this pass
  bool moved = true;
  int have = -1, hb = 0;   // ps.w.e[hb] holds the final pairs of frame `have`
  u64 st_links = 0, st_toks = 0;   // links / tokens priced by this walk (wfst_decoder_get_lattice_stats)
  unsigned long long tw = wall_clock64();
  int k_first = nd - 1;
  if (!kFinal && raw_done && n_prev < nd) {
    // the raw frames are priced; did the oldest of them move (it was created with extra_cost 0, base-inl.h:103)?
    k_first = n_prev - 1;
    k_lo = n_prev;
    if (tid == 0) ps.any_changed = 0;
    __syncthreads();
    int ch = 0;
    for (int i = foff[n_prev] + tid; i < foff[n_prev + 1]; i += kBT) ch |= fabsf(o2f(extra[i].x) - 0.0f) > delta;
    if (ch) ps.any_changed = 1;
    __syncthreads();
    moved = ps.any_changed != 0;
    __syncthreads();
  }
  for (int k = k_first; k >= 0; --k) {
    if (tid == 0 && (D.dbg & 32)) { const unsigned long long now = wall_clock64(); atomicAdd(&D.dbg_t[(k + 1 < n_prev) ? 41 : 40], now - tw); tw = now; }
    const int fk = foff[k], fk1 = foff[k + 1], fk2 = foff[k + 2];
    const int nk = fk1 - fk, n1 = fk2 - fk1;
    const bool had_old = k < n_prev;
    if (!kFinal && had_old && !moved) break;
    k_lo = k;
    st_links += (u64)(lmid[k + 1] - loff[k + 1]) + 2ull * (u64)(loff[k + 1] - lmid[k]);   // (an epsilon link: priced, then confirmed)
    st_toks += (u64)nk;
    if (nk <= 2 * kPrLds) {
      // ---- the frame in LDS ----
      // (WIDE frames, kPrLds < nk <= 2 kPrLds -- the raw frames of a heavy channel at beam 15: only the 4-byte extras live in LDS,
      // 32 768 of them; the costs the epsilon links and the write-back need are read from the tokens)
      // (DecoderDev::link_delta, xmode: a link carries its cost relative to its destination's, so the 4-byte extras are ALL a frame's
      // pricing needs -- every frame takes the extras-only form, 32 768 of them fit, and the frame behind stays in LDS beside it)
      const bool xmode = D.link_delta != 0;
      const bool wide = xmode || nk > kPrLds;
      constexpr int kX = 2 * kPrLds;                        // extras the buffer holds
      uint32_t *Xb = reinterpret_cast<uint32_t *>(ps.w.e);
      // Placement: the pairs (xmode: extras) of frame k+1 sit at one end of the buffer (left there by the step before); frame k's go
      // to the other end.  Where both do not fit, frame k+1's are read from HBM instead (next_lds false).
      const bool next_had = have == k + 1;                 // frame k+1's pairs are in LDS (at the `hb` end)
      const bool next_fits = xmode ? n1 + nk <= kX : (!wide && n1 + nk <= kPrLds);
      uint2 *E1 = nullptr;
      uint32_t *E1x = nullptr;                              // (xmode)
      int cur_end;                                          // 0: frame k at the low end, 1: at the high end
      if (next_had && next_fits) {
        if (xmode) E1x = hb == 0 ? Xb : Xb + (kX - n1);
        else E1 = hb == 0 ? ps.w.e : ps.w.e + (kPrLds - n1);
        cur_end = hb ^ 1;
      } else if (next_fits) {                               // fetch them (through L2: the HBM path prices with atomics)
        if (xmode) {
          E1x = Xb;
          for (int i = tid; i < n1; i += kBT) Xb[i] = (uint32_t)ld_agent(reinterpret_cast<const u64 *>(&extra[fk1 + i]));
        } else {
          E1 = ps.w.e;
          for (int i = tid; i < n1; i += kBT) {
            const u64 v = ld_agent(reinterpret_cast<const u64 *>(&extra[fk1 + i]));
            ps.w.e[i] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
          }
        }
        cur_end = 1;
      } else {
        cur_end = 0;                                        // frame k alone; its successor's pairs come from HBM link by link
      }
      uint2 *E0 = (cur_end == 0 || wide) ? ps.w.e : ps.w.e + (kPrLds - nk);
      uint32_t *E0x = (xmode && cur_end == 1) ? Xb + (kX - nk) : Xb;   // (wide frames; xmode: every frame)
      // everything a SMALL frame needs from HBM is asked for at once, before the first barrier: the tokens' costs, their
      // extras of the previous pass, its emitting links and epsilon links (a pruned frame: a few hundred tokens, a
      // thousand links); a larger one streams them
      constexpr int kTU = 4;
      const bool small = nk <= kTU * kBT;
      const int m_lo = loff[k + 1], m_hi = lmid[k + 1];   // emitting links frame k -> k+1
      const int e0 = lmid[k], e1 = loff[k + 1];           // epsilon links inside frame k
      int cy[kTU];
      uint32_t ox[kTU];
      int4 ML[kPU], EL[kPU];
#pragma unroll
      for (int u = 0; u < kTU; ++u) {
        const int i = u * kBT + tid;
        cy[u] = (small && !xmode && i < nk) ? tok[fk + i].y : 0;
        ox[u] = (small && had_old && i < nk) ? extra[fk + i].x : 0u;
      }
#pragma unroll
      for (int u = 0; u < kPU; ++u) {
        const int i = m_lo + u * kBT + tid;
        ML[u] = i < m_hi ? links[i] : make_int4(-1, 0, 0, 0);
      }
      const bool eps_in_regs = e1 - e0 <= kBT * kPU;
#pragma unroll
      for (int u = 0; u < kPU; ++u) {
        const int i = e0 + u * kBT + tid;
        EL[u] = (eps_in_regs && i < e1) ? links[i] : make_int4(-1, 0, 0, 0);
      }
      if (small && !wide) {
#pragma unroll
        for (int u = 0; u < kTU; ++u) {
          const int i = u * kBT + tid;
          if (i < nk) E0[i] = make_uint2(kInfO, (uint32_t)cy[u]);
        }
      } else if (small) {   // (xmode)
#pragma unroll
        for (int u = 0; u < kTU; ++u) {
          const int i = u * kBT + tid;
          if (i < nk) E0x[i] = kInfO;
        }
      } else {
        if (wide) {
          for (int i = tid; i < nk; i += kBT) E0x[i] = kInfO;
        } else {
        for (int i0 = 0; i0 < nk; i0 += 4 * kBT) {   // four loads in flight per thread
          int c4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int i = i0 + u * kBT + tid; c4[u] = i < nk ? tok[fk + i].y : 0; }
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int i = i0 + u * kBT + tid; if (i < nk) E0[i] = make_uint2(kInfO, (uint32_t)c4[u]); }
        }
        }
      }
      // (flags: three in rotation, so that a round needs ONE barrier -- flag r % 3 is raised in round r and read after the round's
      // barrier; the next round's flag is cleared during this round, when nobody reads or raises it)
      if (tid < 3) ps.flag[tid] = 0;
      __syncthreads();
      auto price0 = [&](const int4 &X) -> float {   // a link into frame k itself (epsilon links)
        uint2 en;
        if (wide) { en.x = E0x[X.y - fk]; en.y = (xmode || en.x >= kInfO) ? 0u : (uint32_t)tok[X.y].y; }
        else en = E0[X.y - fk];
        return en.x >= kInfO ? kInf : o2f(en.x) + (D.link_delta ? __int_as_float(X.w) : __int_as_float(X.w) - __uint_as_float(en.y));
      };
      auto min0 = [&](int i, uint32_t o) -> uint32_t { return wide ? atomicMin(&E0x[i], o) : atomicMin(&E0[i].x, o); };
      // emitting links frame k -> k+1
      for (int i0 = m_lo; i0 < m_hi; i0 += kBT * kPU) {
        if (i0 != m_lo) {
#pragma unroll
          for (int u = 0; u < kPU; ++u) {
            const int i = i0 + u * kBT + tid;
            ML[u] = i < m_hi ? links[i] : make_int4(-1, 0, 0, 0);
          }
        }
        u64 en[kPU];
        if (E1) {
#pragma unroll
          for (int u = 0; u < kPU; ++u) {
            const uint2 v = ML[u].x >= 0 ? E1[ML[u].y - fk1] : make_uint2(0, 0);
            en[u] = (u64)v.x | ((u64)v.y << 32);
          }
        } else if (E1x) {
#pragma unroll
          for (int u = 0; u < kPU; ++u) en[u] = ML[u].x >= 0 ? (u64)E1x[ML[u].y - fk1] : 0ull;
        } else {
#pragma unroll
          for (int u = 0; u < kPU; ++u) en[u] = ML[u].x >= 0 ? ld_agent(reinterpret_cast<const u64 *>(&extra[ML[u].y])) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kPU; ++u) {
          if (ML[u].x < 0) continue;
          const uint32_t eo = (uint32_t)en[u];
          float le = eo >= kInfO ? kInf : o2f(eo) + (D.link_delta ? __int_as_float(ML[u].w) : __int_as_float(ML[u].w) - __int_as_float((int)(en[u] >> 32)));
          if (!(le <= lb)) { links[i0 + u * kBT + tid].x = -1; continue; }
          if (le < 0.0f) le = 0.0f;
          min0(ML[u].x - fk, f2o(le));
        }
      }
      __syncthreads();
      // the epsilon links inside frame k, to their fixpoint
      if (e0 < e1) {
        for (int round = 0; round < 4096; ++round) {
          int ch = 0;
          if (eps_in_regs) {
#pragma unroll
            for (int u = 0; u < kPU; ++u) {
              if (EL[u].x < 0) continue;
              float le = price0(EL[u]);
              if (!(le <= lb)) continue;
              if (le < 0.0f) le = 0.0f;
              const uint32_t o = f2o(le);
              if (o < min0(EL[u].x - fk, o)) ch = 1;
            }
          } else {
            for (int i = e0 + tid; i < e1; i += kBT) {
              const int4 X = links[i];
              if (X.x < 0) continue;
              float le = price0(X);
              if (!(le <= lb)) continue;
              if (le < 0.0f) le = 0.0f;
              const uint32_t o = f2o(le);
              if (o < min0(X.x - fk, o)) ch = 1;
            }
          }
          const int fl = round % 3;
          if (ch) ps.flag[fl] = 1;
          if (tid == 0) ps.flag[(fl + 1) % 3] = 0;   // (last read two barriers ago, next raised after this round's barrier)
          __syncthreads();
          if (!ps.flag[fl]) break;
        }
        // the dead ones are marked
        if (eps_in_regs) {
#pragma unroll
          for (int u = 0; u < kPU; ++u) {
            if (EL[u].x < 0) continue;
            if (!(price0(EL[u]) <= lb)) links[e0 + u * kBT + tid].x = -1;
          }
        } else {
          for (int i = e0 + tid; i < e1; i += kBT) {
            const int4 X = links[i];
            if (X.x >= 0 && !(price0(X) <= lb)) links[i].x = -1;
          }
        }
      }
      // the frame's pairs go to HBM; did they move?
      int ch = 0;
      if (small) {
#pragma unroll
        for (int u = 0; u < kTU; ++u) {
          const int i = u * kBT + tid;
          if (i >= nk) continue;
          const uint2 v = wide ? make_uint2(E0x[i], (uint32_t)cy[u]) : E0[i];
          extra[fk + i] = v;
          if (!kFinal) {
            const float now = o2f(v.x);
            const float was = had_old ? o2f(ox[u]) : 0.0f;   // a token is created with extra_cost 0 (base-inl.h:103)
            ch |= fabsf(now - was) > delta;                   // (inf - inf = NaN: not "moved", as in the reference)
          }
        }
      } else {
        for (int i0 = 0; i0 < nk; i0 += 4 * kBT) {
          uint32_t o4[4];
          int c4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kBT + tid;
            o4[u] = (!kFinal && had_old && i < nk) ? extra[fk + i].x : 0u;
            c4[u] = (wide && !xmode && i < nk) ? tok[fk + i].y : 0;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kBT + tid;
            if (i >= nk) continue;
            const uint2 v = wide ? make_uint2(E0x[i], (uint32_t)c4[u]) : E0[i];
            if (!kFinal) {
              const float now = o2f(v.x);
              const float was = had_old ? o2f(o4[u]) : 0.0f;
              ch |= fabsf(now - was) > delta;
            }
            extra[fk + i] = v;
          }
        }
      }
      if (!kFinal) {
        if (tid == 0) ps.any_changed = 0;
        __syncthreads();
        if (ch) ps.any_changed = 1;
        __syncthreads();
        moved = ps.any_changed != 0;
      }
      __syncthreads();   // (the next frame rewrites the other end of the buffer and the flags)
      have = (wide && !xmode) ? -1 : k;   // (a wide frame leaves extras only: its predecessor reads the pairs from HBM; xmode: extras are all it needs)
      hb = cur_end;
      if (tid == 0 && (D.dbg & 32)) atomicAdd(&D.dbg_t[42], 1ull);
      continue;
    }
    // ---- the frame in HBM (more tokens than the LDS buffers take) ----
    have = -1;
    if (tid == 0 && (D.dbg & 32)) atomicAdd(&D.dbg_t[43], 1ull);
    for (int i0 = fk; i0 < fk1; i0 += kBT * kPU) {
      int cy[kPU];
      uint32_t ox[kPU];
#pragma unroll
      for (int u = 0; u < kPU; ++u) {
        const int i = i0 + u * kBT + tid;
        cy[u] = (i < fk1 && !D.link_delta) ? tok[i].y : 0;
        ox[u] = (had_old && i < fk1) ? extra[i].x : 0u;
      }
#pragma unroll
      for (int u = 0; u < kPU; ++u) {
        const int i = i0 + u * kBT + tid;
        if (i >= fk1) continue;
        if (had_old) remap[i] = (int32_t)ox[u];
        extra[i] = make_uint2(kInfO, (uint32_t)cy[u]);
      }
    }
    __syncthreads();
    // emitting links frame k -> k+1
    for_links(loff[k + 1], lmid[k + 1], [&](int i, const int4 &L, float le) {
      if (!(le <= lb)) { links[i].x = -1; return; }
      if (le < 0.0f) le = 0.0f;
      atomicMin(&extra[L.x].x, f2o(le));
    });
    __syncthreads();
    relax_eps(lmid[k], loff[k + 1]);
    if (!kFinal) {
      if (tid == 0) ps.any_changed = 0;
      __syncthreads();
      int ch = 0;
      for (int i = fk + tid; i < fk1; i += kBT) {
        const float now = o2f(ld_agent(&extra[i].x));
        const float was = had_old ? o2f((uint32_t)remap[i]) : 0.0f;   // a token is created with extra_cost 0 (base-inl.h:103)
        ch |= fabsf(now - was) > delta;                                // (inf - inf = NaN: not "moved", as in the reference)
      }
      if (ch) ps.any_changed = 1;
      __syncthreads();
      moved = ps.any_changed != 0;
      __syncthreads();
    }
  }
  __syncthreads();

  if (tid == 0) { D.lat_stats[(size_t)c * 4 + 1] += st_links; D.lat_stats[(size_t)c * 4 + 2] += st_toks; }
  if (tid == 0 && (D.dbg & 32)) {
    const unsigned long long now = wall_clock64();
    atomicAdd(&D.dbg_t[51], now - tq); atomicAdd(&D.dbg_t[52], 1ull); atomicAdd(&D.dbg_t[53], (unsigned long long)(nd - k_lo));
    // (how long a channel's walk lasts, 0.4 ms bins: [80, 91) channels that priced their raw frames here, [117, 128) behind lattice_prune_raw_kernel)
    if (!kFinal) atomicAdd(&D.dbg_t[(raw_done ? 117 : 80) + min(10, (int)((now - tq) / 40000ull))], 1ull);
    tq = now;
  }
  // ---- (3) compaction of what this pass priced for the FIRST time: tokens of frames [c_lo, nd], links from the
  // epsilon links of frame c_lo on.  That is where nearly everything dies (a raw frame keeps a percent or two of
  // its tokens).  Frames priced before only lose the odd token now and then: there the dead stay where they are
  // as holes (a token is dead iff its extra is +inf, a link iff it was marked; lattice_emit_kernel skips both) --
  // moving the whole history down for them would cost more than the walk itself.
  if (kFinal) {
    if (tid == 0) ctl->pruned_upto = nd + 1;   // every frame is priced, the newest included; nothing moves any more
    __syncthreads();
    return;
  }
  {
    // ... until the holes add up: every eighth pass, or with the arena or the link store half full, the
    // whole re-priced range is compacted
    const bool full = (nd / D.prune_interval) % 8 == 0 || 2 * (int64_t)foff[nd + 1] > D.arena_cap ||
                      2 * (int64_t)ctl->link_count > D.link_cap;
    const int c_lo = full ? k_lo : max(k_lo, min(n_prev, nd));
    // The compaction runs as launches of its own (prune_flags: several workgroups per channel; prune_move), which read the range here
    if (tid == 0) {
      int32_t *pp = D.prune_par + (size_t)c * kPrParInts;
      pp[0] = 1;                                   // a pass ran: compact
      pp[1] = c_lo;
      pp[2] = foff[c_lo]; pp[3] = foff[nd + 1];    // tokens [range_lo, end)
      pp[4] = (c_lo == 0) ? foff[1] : 0;           // PruneActiveTokens never calls PruneTokensForFrame(0) (base-inl.h:471-476): frame 0
                                                   // keeps its dead tokens, link-less, until FinalizeDecoding
      pp[5] = loff[c_lo]; pp[6] = loff[nd + 1];    // links [l_lo, l_end): from the emitting links INTO frame c_lo (their destinations move)
      pp[7] = nd;
    }
    __syncthreads();
  }
  (void)any_final;
}

// ---- the compaction of a running pass -----------------------------------------------------------------------------------------
// What the pass priced for the FIRST time is compacted: tokens of frames [c_lo, nd], links from the epsilon links of frame c_lo on.
// That is where nearly everything dies (a raw frame keeps a percent or two of its tokens).  Frames priced before only lose the odd
// token now and then: there the dead stay where they are as holes (a token is dead iff its extra is +inf, a link iff it was marked;
// lattice_emit_kernel skips both) until the holes add up (every eighth pass, or arena / link store half full: `full` above).
//
// FLAT sweeps over the whole range, and the MOVES run over the survivors only.  The FLAG sweeps -- nine tenths of the items a
// compaction touches -- are shared out over kPrSlabs workgroups per channel (prune_flags, a launch of its own: the kernel boundary
// orders it after the walk and before the moves): slab g of the token range and of the link range gets slab-relative ranks,
//   remap[i] = rank of a surviving token in its slab, ~(survivors of the slab below i) of a dead one;  lpre[i] likewise for links,
// the survivors' old indices listed per slab (at the slab's own offset of surv[] / lsurv[]: a slab has no more survivors than
// items), and its survivor count in the channel's parameter block.  prune_move (one workgroup: the moves are in place and ordered)
// turns the slab counts into bases -- new index = range_lo + base[slab] + rank -- and moves the survivors down.
// Scratch: the channel's lat_toks / lat_arcs (they hold nothing between two lattice_emit launches; int32 views).
constexpr int kPrFU = 16;                     // items per thread of a FLAG sweep (4 bytes each in flight)
constexpr int kPrFlagChunk = kBT * kPrFU;

__device__ __forceinline__ int pr_slab_len(int len) { return ((len + kPrSlabs - 1) / kPrSlabs + 63) & ~63; }

__device__ __forceinline__ void prune_flags(const DecoderDev &D, int c, int g, ScanShared &ps) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int32_t *pp = D.prune_par + (size_t)c * kPrParInts;
  if (pp[0] != 1) return;
  const int range_lo = pp[2], end = pp[3], f0_hi = pp[4], l_lo = pp[5], l_end = pp[6];
  const uint2 *extra = D.extra + (size_t)c * D.arena_cap;
  const int4 *links = D.links + (size_t)c * D.link_cap;
  int32_t *remap = D.remap + (size_t)c * D.arena_cap;
  int32_t *surv = reinterpret_cast<int32_t *>(D.lat_toks + (size_t)c * D.lat_tok_cap);    // [slab offset + rank] -> old index (tokens)
  int32_t *lsurv = reinterpret_cast<int32_t *>(D.lat_arcs + (size_t)c * D.lat_arc_cap);   // [slab offset + rank] -> old index (links)
  int32_t *lpre = lsurv + D.link_cap;                                                      // [old index] -> survivors of its slab below it
  const uint32_t kInfO = f2o(__builtin_huge_valf());
  // (a chunk's items in wave-coalesced order -- wave w, step u, lane l: item (w * kPrFU + u) * 64 + l -- and the survivors'
  // ranks from ballots: a lane reads 4 or 16 bytes beside its neighbour's)
  auto chunk_ranks = [&](const bool (&alive)[kPrFU], int base, int (&rank)[kPrFU]) -> int {
    u64 m[kPrFU];
    int wc = 0;
#pragma unroll
    for (int u = 0; u < kPrFU; ++u) { m[u] = __ballot(alive[u]); wc += __popcll(m[u]); }
    __syncthreads();  // ps.wsum free again
    if (lane == 0) ps.wsum[wave] = wc;
    __syncthreads();
    int wb = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kBT / 64; ++w) {
      const int x = ps.wsum[w];
      if (w < wave) wb += x;
      tot += x;
    }
    int run = base + wb;
#pragma unroll
    for (int u = 0; u < kPrFU; ++u) { rank[u] = run + lane_rank(m[u]); run += __popcll(m[u]); }
    return tot;
  };
  unsigned long long tw = wall_clock64();
  {  // tokens of slab g
    const int sl = pr_slab_len(end - range_lo), lo = min(end, range_lo + g * sl), hi = min(end, lo + sl);
    int cnt = 0;
    for (int c0 = lo; c0 < hi; c0 += kPrFlagChunk) {
      bool alive[kPrFU];
      int rank[kPrFU];
#pragma unroll
      for (int u = 0; u < kPrFU; ++u) {
        const int i = c0 + (wave * kPrFU + u) * 64 + lane;
        alive[u] = i < hi && (i < f0_hi || extra[i].x < kInfO);
      }
      const int tot = chunk_ranks(alive, cnt, rank);
#pragma unroll
      for (int u = 0; u < kPrFU; ++u) {
        const int i = c0 + (wave * kPrFU + u) * 64 + lane;
        if (i < hi) {
          remap[i] = alive[u] ? rank[u] : ~rank[u];
          if (alive[u]) surv[(lo - range_lo) + rank[u]] = i;
        }
      }
      cnt += tot;
    }
    if (tid == 0) pp[kPrSlabBase + g] = cnt;
  }
  if (tid == 0 && g == 0 && (D.dbg & 32)) { const unsigned long long now = wall_clock64(); atomicAdd(&D.dbg_t[44], now - tw); atomicAdd(&D.dbg_t[54], now - tw); tw = now; }
  {  // links of slab g
    const int sl = pr_slab_len(l_end - l_lo), lo = min(l_end, l_lo + g * sl), hi = min(l_end, lo + sl);
    int cnt = 0;
    for (int c0 = lo; c0 < hi; c0 += kPrFlagChunk) {
      bool alive[kPrFU];
      int rank[kPrFU];
#pragma unroll
      for (int u = 0; u < kPrFU; ++u) {
        const int i = c0 + (wave * kPrFU + u) * 64 + lane;
        alive[u] = i < hi && links[i].x >= 0;
      }
      const int tot = chunk_ranks(alive, cnt, rank);
#pragma unroll
      for (int u = 0; u < kPrFU; ++u) {
        const int i = c0 + (wave * kPrFU + u) * 64 + lane;
        if (i < hi) {
          lpre[i] = rank[u];
          if (alive[u]) lsurv[(lo - l_lo) + rank[u]] = i;
        }
      }
      cnt += tot;
    }
    if (tid == 0) pp[kPrSlabBase + kPrSlabs + g] = cnt;
  }
  if (tid == 0 && g == 0 && (D.dbg & 32)) { const unsigned long long now = wall_clock64(); atomicAdd(&D.dbg_t[46], now - tw); atomicAdd(&D.dbg_t[54], now - tw); }
}

// role 0: the tokens (frame offsets, moves, the channel's frontier and best token); role 1: the links (segment bounds, moves, the link
// counter) -- two workgroups side by side: the links' new endpoints come from remap[] and the slab bases, not from the moved tokens
__device__ __forceinline__ void prune_move(const DecoderDev &D, int c, int role, ScanShared &ps) {
  const int tid = threadIdx.x;
  int32_t *pp = D.prune_par + (size_t)c * kPrParInts;
  if (pp[0] != 1) return;
  ChanCtl *ctl = D.ctl + c;
  const int k_lo = pp[1], range_lo = pp[2], end = pp[3], l_lo = pp[5], l_end = pp[6], nd = pp[7];
  int4 *tok = D.tok + (size_t)c * D.arena_cap;
  int4 *links = D.links + (size_t)c * D.link_cap;
  uint2 *extra = D.extra + (size_t)c * D.arena_cap;
  int32_t *remap = D.remap + (size_t)c * D.arena_cap;
  int32_t *foff = D.frame_off + (size_t)c * (D.max_frames + 2);
  int32_t *loff = D.link_off + (size_t)c * (D.max_frames + 3);
  int32_t *lmid = D.link_mid + (size_t)c * (D.max_frames + 3);
  const int32_t *surv = reinterpret_cast<const int32_t *>(D.lat_toks + (size_t)c * D.lat_tok_cap);
  const int32_t *lsurv = reinterpret_cast<const int32_t *>(D.lat_arcs + (size_t)c * D.lat_arc_cap);
  const int32_t *lpre = lsurv + D.link_cap;
  unsigned long long tw = wall_clock64();
  // the slabs' bases: survivors of the slabs below
  int tb[kPrSlabs + 1], lb[kPrSlabs + 1];
  tb[0] = 0; lb[0] = 0;
#pragma unroll
  for (int g = 0; g < kPrSlabs; ++g) { tb[g + 1] = tb[g] + pp[kPrSlabBase + g]; lb[g + 1] = lb[g] + pp[kPrSlabBase + kPrSlabs + g]; }
  const int tsl = pr_slab_len(end - range_lo), lsl = pr_slab_len(l_end - l_lo);
  const int new_end = range_lo + tb[kPrSlabs], lnew = l_lo + lb[kPrSlabs];
  // new index of token i of the range (negative: dead, ~(survivors below it))
  auto tok_new = [&](int i) -> int {
    const int g = (i - range_lo) / tsl, r = remap[i];
    int base = 0;
#pragma unroll
    for (int q = 0; q < kPrSlabs; ++q) base = (q == g) ? tb[q] : base;
    return r >= 0 ? range_lo + base + r : ~(range_lo + base + ~r);
  };
  auto surv_at = [&](const int32_t *list, const int (&base)[kPrSlabs + 1], int sl, int j) -> int {   // old index of survivor j (0-based within the range)
    int g = 0;
#pragma unroll
    for (int q = 1; q < kPrSlabs; ++q) g += j >= base[q] ? 1 : 0;
    int bg = 0;
#pragma unroll
    for (int q = 0; q < kPrSlabs; ++q) bg = (q == g) ? base[q] : bg;
    return list[g * sl + (j - bg)];
  };
  if (tid == 0) ps.err = 0;
  __syncthreads();
  constexpr int kCU = kPrChunk / kBT;   // items per thread and sweep
  if (role == 0) {
  // the frames' new offsets: survivors below each old offset
  for (int f = k_lo + tid; f <= nd; f += kBT) {
    const int p = foff[f + 1];
    int q = new_end;
    if (p < end) { q = tok_new(p); q = q < 0 ? ~q : q; }
    foff[f + 1] = q;
  }
  __syncthreads();
  // the survivors move down (a sweep is read whole before it is written; new index <= old index, so a sweep's writes land on
  // positions that this sweep or an earlier one has read)
  for (int j0 = range_lo; j0 < new_end; j0 += kPrChunk) {
    int oi[kCU], sl[kCU];
    int4 rec[kCU];
    uint2 ex[kCU];
    int32_t *side = D.big ? D.tok_lm + (size_t)c * D.arena_cap : nullptr;   // biglm: the tokens' LM pair states move along
#pragma unroll
    for (int u = 0; u < kCU; ++u) {
      const int j = j0 + u * kBT + tid;
      oi[u] = j < new_end ? surv_at(surv, tb, tsl, j - range_lo) : -1;
    }
#pragma unroll
    for (int u = 0; u < kCU; ++u) {
      rec[u] = make_int4(0, 0, 0, 0);
      ex[u] = make_uint2(0, 0);
      sl[u] = 0;
      if (oi[u] >= 0) { rec[u] = tok[oi[u]]; ex[u] = extra[oi[u]]; if (side) sl[u] = side[oi[u]]; }
    }
#pragma unroll
    for (int u = 0; u < kCU; ++u) {
      // backpointer: a survivor's predecessor survives (the link between them is the token's own best one);
      // predecessors below the compacted range have not moved
      if (oi[u] >= 0 && rec[u].z >= range_lo) {
        rec[u].z = tok_new(rec[u].z);
        if (rec[u].z < 0) ps.err = 1;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kCU; ++u) {
      const int j = j0 + u * kBT + tid;
      if (oi[u] >= 0) { tok[j] = rec[u]; extra[j] = ex[u]; if (side) side[j] = sl[u]; }
    }
    __syncthreads();
  }
  if (tid == 0 && (D.dbg & 32)) { const unsigned long long now = wall_clock64(); atomicAdd(&D.dbg_t[45], now - tw); atomicAdd(&D.dbg_t[54], now - tw); tw = now; }
  if (tid == 0) {
    ctl->front_begin = foff[nd];
    ctl->front_count = foff[nd + 1] - foff[nd];
    const u64 b = ctl->best_next;
    if (b != ~0ull) {
      const int nb = tok_new((int)(uint32_t)b);
      ctl->best_next = nb >= 0 ? ((b & 0xFFFFFFFF00000000ull) | (uint32_t)nb) : ~0ull;
    }
    ctl->pruned_upto = nd;
    if (ps.err) atomicOr(&ctl->error, kErrInternal);  // never expected: a surviving token whose predecessor died
  }
  __syncthreads();
  return;
  }
  // a frame's segment = [link_off[f], link_mid[f]) emitting into it, [link_mid[f], link_off[f+1]) epsilon inside it
  auto link_new = [&](int p) -> int {   // survivors below link p, as a new index
    const int g = (p - l_lo) / lsl;
    int base = 0;
#pragma unroll
    for (int q = 0; q < kPrSlabs; ++q) base = (q == g) ? lb[q] : base;
    return l_lo + base + lpre[p];
  };
  for (int q = tid; q < 2 * (nd - k_lo) + 2; q += kBT) {
    int *slot = (q & 1) ? &loff[k_lo + (q >> 1) + 1] : &lmid[k_lo + (q >> 1)];
    const int p = *slot;
    *slot = p < l_end ? link_new(p) : lnew;
  }
  for (int j0 = l_lo; j0 < lnew; j0 += kPrChunk) {
    int4 L[kCU];
#pragma unroll
    for (int u = 0; u < kCU; ++u) {
      const int j = j0 + u * kBT + tid;
      const int oi = j < lnew ? surv_at(lsurv, lb, lsl, j - l_lo) : -1;
      L[u] = oi >= 0 ? links[oi] : make_int4(-1, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < kCU; ++u)
      if (L[u].x >= 0) {   // (an endpoint below the compacted range has not moved)
        if (L[u].x >= range_lo) L[u].x = tok_new(L[u].x);
        if (L[u].y >= range_lo) L[u].y = tok_new(L[u].y);
      }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kCU; ++u) {
      const int j = j0 + u * kBT + tid;
      if (j < lnew) links[j] = L[u];
    }
    __syncthreads();
  }
  if (tid == 0 && (D.dbg & 32)) { const unsigned long long now = wall_clock64(); atomicAdd(&D.dbg_t[47], now - tw); atomicAdd(&D.dbg_t[54], now - tw); atomicAdd(&D.dbg_t[48], (unsigned long long)(end - range_lo)); atomicAdd(&D.dbg_t[49], (unsigned long long)(l_end - l_lo)); atomicAdd(&D.dbg_t[50], (unsigned long long)((new_end - range_lo) + (lnew - l_lo))); }
  if (tid == 0) D.lat_stats[(size_t)c * 4 + 3] += (u64)(end - range_lo) + (u64)(l_end - l_lo) + (((u64)(new_end - range_lo) + (u64)(lnew - l_lo)) << 32);
  __syncthreads();
  if (tid == 0) ctl->link_count = lnew;
  __syncthreads();
}


// =========================================================================================
// Best-path decoders: token garbage collection.  A best-path decoder keeps no forward links, so nothing prunes its
// arena the way PruneActiveTokens prunes the reference's token lists; what GetBestPath can ever need is the
// backpointer FOREST of the current frontier.  When the arena passes its collection mark (an eighth of it left; then halfway
// between what survived and the capacity), every token reachable from the frontier is marked -- frame by frame,
// newest first; a token won by an epsilon arc (kPrevUnresolved) has its predecessor, the frame's token on the arc's
// source state, found through an LDS hash of the wanted states and WRITTEN BACK as an ordinary backpointer -- and the
// survivors are moved down (frame_off[], backpointers, frontier, best token remapped).  GetBestPath is unchanged by it
// (same chain, same costs; tests/test_gpu_token_gc.py) and the arena then bounds the raw frames between two
// collections, not the utterance.  One 1024-thread workgroup per channel, inside closure_kernel; only when the mark is
// passed (never on the bench workload).  biglm: several tokens of a frame may sit on the wanted state (one per LM
// state); all of them are kept and the backpointer stays unresolved.
// =========================================================================================
// The arena is collected when less than DecoderDev::gc_reserve of it is left: an eighth (and at least two frames' worth of tokens at
// the per-frame limit: one frame can add max_tokens_per_frame tokens before the next check); two-launch decoders, which look at
// the mark every gc_stride-th frame only, up to a quarter where that buys a longer stride -- the check is a third launch on that
// frame (wfst_capi.cc).  A collection is not cheap (~5 ms per million tokens in the arena), so it should be rare.  Arenas too
// small for that reserve collect when half full.
__device__ __forceinline__ int gc_base_mark(const DecoderDev &D) { return (int)(D.arena_cap - D.gc_reserve); }
constexpr int kGcNeedSlots = 2048;  // LDS hash of the states wanted in one sweep (power of two)
struct GcShared {
  int32_t key[kGcNeedSlots];    // wanted state (row), -1 empty
  int32_t found[kGcNeedSlots];  // the frame's token on it
  int n_need, n_new, overflow;
};

template <bool kBig>
__device__ __forceinline__ void gc_pass(const DecoderDev &D, int c, ScanShared &ps, GcShared &gs) {
  const int tid = threadIdx.x;
  ChanCtl *ctl = D.ctl + c;
  const int nd = ctl->n_decoded;
  int4 *tok = D.tok + (size_t)c * D.arena_cap;
  int32_t *side = (kBig || D.tok_lm) ? D.tok_lm + (size_t)c * D.arena_cap : nullptr;
  int32_t *remap = D.remap + (size_t)c * D.arena_cap;   // 0 dead, 1 marked, 2 marked and followed; then new indices
  int32_t *foff = D.frame_off + (size_t)c * (D.max_frames + 2);
  const int end = foff[nd + 1];
  const uint32_t idx_mask = D.tok_idx_bits >= 31 ? 0x7FFFFFFFu : ((1u << D.tok_idx_bits) - 1u);
  // (1) mark: the frontier, then what it reaches
  for (int i = tid; i < end; i += kBT) remap[i] = i >= foff[nd] ? 1 : 0;
  __syncthreads();
  for (int f = nd; f >= 0; --f) {
    const int lo = foff[f], hi = foff[f + 1];
    for (;;) {   // until no token of the frame is newly marked (epsilon chains inside the frame)
      for (int i = tid; i < kGcNeedSlots; i += kBT) { gs.key[i] = -1; gs.found[i] = -1; }
      if (tid == 0) { gs.n_need = 0; gs.n_new = 0; gs.overflow = 0; }
      __syncthreads();
      for (int i = lo + tid; i < hi; i += kBT) {
        if (remap[i] != 1) continue;
        const int4 t = tok[i];
        bool done = true;
        const int zi = t.z >= 0 ? (int)((uint32_t)t.z & idx_mask) : t.z;   // the arena index under a degree code, if any
        if (t.z <= kPrevUnresolved) {
          const int need = D.g.arc_src[(uint32_t)t.w & kArcMask] & 0x7FFFFFFF;
          // claim a slot for the wanted state (shared by every token that wants it); a full table: next sweep
          uint32_t slot = hash32(need) & (kGcNeedSlots - 1);
          done = false;
          if (atomicAdd(&gs.n_need, 0) < (kGcNeedSlots * 3) / 4) {
            for (int q = 0; q < kGcNeedSlots; ++q) {
              const int k = atomicCAS(&gs.key[slot], -1, need);
              if (k == -1) { atomicAdd(&gs.n_need, 1); done = true; break; }
              if (k == need) { done = true; break; }
              slot = (slot + 1) & (kGcNeedSlots - 1);
            }
          }
          if (!done) gs.overflow = 1;
        } else if (zi >= lo) {
          // a backpointer on this same frame (an epsilon hop resolved by an earlier collection)
          if (remap[zi] == 0) { remap[zi] = 1; gs.n_new = 1; }
        } else if (zi >= 0) {
          remap[zi] = 1;   // on the previous frame
        }
        if (done) remap[i] = 2;
      }
      __syncthreads();
      const int n_need = gs.n_need;
      {
        const int ov = gs.overflow, nw = gs.n_new;
        __syncthreads();   // (everyone has read the flags before they are reset)
        if (n_need == 0 && !ov) {
          if (!nw) break;
          continue;
        }
      }
      // the frame's tokens on wanted states
      for (int i = lo + tid; i < hi; i += kBT) {
        const int st = tok[i].x;
        uint32_t slot = hash32(st) & (kGcNeedSlots - 1);
        for (int q = 0; q < kGcNeedSlots; ++q) {
          const int k = gs.key[slot];
          if (k == -1) break;
          if (k == st) {
            if constexpr (!kBig) gs.found[slot] = i;
            if (remap[i] == 0) { remap[i] = 1; gs.n_new = 1; }
            break;
          }
          slot = (slot + 1) & (kGcNeedSlots - 1);
        }
      }
      __syncthreads();
      if constexpr (!kBig) {
        // one token per state and frame: the predecessor is unique -- written back as an ordinary backpointer
        for (int i = lo + tid; i < hi; i += kBT) {
          if (remap[i] != 2) continue;
          const int4 t = tok[i];
          if (t.z > kPrevUnresolved) continue;
          const int need = D.g.arc_src[(uint32_t)t.w & kArcMask] & 0x7FFFFFFF;
          uint32_t slot = hash32(need) & (kGcNeedSlots - 1);
          for (int q = 0; q < kGcNeedSlots; ++q) {
            const int k = gs.key[slot];
            if (k == -1) break;
            if (k == need) {
              if (gs.found[slot] < 0) ps.err = 2;   // the frame holds no token on the arc's source state: never expected
              else tok[i].z = (int)((uint32_t)gs.found[slot] | ((uint32_t)(kPrevUnresolved - t.z) << D.tok_idx_bits));
              break;
            }
            slot = (slot + 1) & (kGcNeedSlots - 1);
          }
        }
      }
      __syncthreads();
      {
        const int ov = gs.overflow, nw = gs.n_new;
        __syncthreads();
        if (!nw && !ov) break;
      }
    }
    __syncthreads();
  }
  // (2) move the survivors down, frame by frame
  constexpr int kCU = 8;
  int new_end = 0, old_lo = 0;
  if (tid == 0) ps.err = 0;
  __syncthreads();
  for (int f = 0; f <= nd; ++f) {
    const int old_hi = foff[f + 1];
    int base = new_end;
    for (int i0 = old_lo; i0 < old_hi; i0 += kBT * kCU) {
      bool alive[kCU];
      int cnt = 0;
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        const int i = i0 + tid * kCU + u;
        alive[u] = i < old_hi && remap[i] != 0;
        cnt += alive[u] ? 1 : 0;
      }
      int tot;
      int r = base + block_exscan(cnt, ps, &tot);
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        const int i = i0 + tid * kCU + u;
        if (i < old_hi) remap[i] = alive[u] ? r++ : -1;
      }
      base += tot;
    }
    __syncthreads();
    for (int i0 = old_lo; i0 < old_hi; i0 += kBT * kCU) {
      int ni[kCU], sw[kCU];
      int4 rec[kCU];
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        const int i = i0 + tid * kCU + u;
        ni[u] = i < old_hi ? remap[i] : -1;
        rec[u] = make_int4(0, 0, 0, 0);
        sw[u] = 0;
        if (ni[u] >= 0) {
          rec[u] = tok[i];
          if (side) sw[u] = side[i];
          if (rec[u].z >= 0) {   // predecessor: a survivor of the previous frame (or of this one), already renumbered
            const int nz = remap[(uint32_t)rec[u].z & idx_mask];
            if (nz < 0) ps.err = 1;
            rec[u].z = (int)((uint32_t)nz | ((uint32_t)rec[u].z & ~idx_mask));   // (the degree code above the index stays)
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < kCU; ++u)
        if (ni[u] >= 0) { tok[ni[u]] = rec[u]; if (side) side[ni[u]] = sw[u]; }
      __syncthreads();
    }
    new_end = base;
    old_lo = old_hi;
    __syncthreads();
    if (tid == 0) foff[f + 1] = new_end;
    __syncthreads();
  }
  if (tid == 0) {
    ctl->front_begin = foff[nd];
    ctl->front_count = foff[nd + 1] - foff[nd];
    const u64 b = ctl->best_next;
    if (b != ~0ull && !D.best_row) {   // (best_row: the low word is the token's graph row, which does not move)
      const int nb = remap[(uint32_t)b];
      ctl->best_next = nb >= 0 ? ((b & 0xFFFFFFFF00000000ull) | (uint32_t)nb) : ~0ull;
    }
    // next collection: at the usual mark, or -- when much survived -- halfway between what survived and the capacity
    // (two-launch decoders check the mark every gc_stride frames only: the room behind it covers that many frames)
    {
      int mark = max(gc_base_mark(D), new_end + (int)((D.arena_cap - new_end) / 2));
      if (D.two_launch) mark = max(gc_base_mark(D), (int)min((int64_t)mark, D.arena_cap - (int64_t)(D.gc_stride + 1) * D.max_tok));
      ctl->lat_arcs = mark;   // (best-path decoders: the collection mark)
    }
    ctl->lat_toks += 1;                                             // (                    collections so far)
    if (ps.err) ctl->error |= kErrInternal;  // never expected: a survivor whose predecessor was not marked or not found
  }
  __syncthreads();
}

template <bool kLat, bool kBig>
__global__ __launch_bounds__(kBT) void closure_kernel(DecoderDev D, const int32_t *target, int do_prep, int chan_off,
                                                      int group, int par, int chan_cnt, int n_slabs) {
  __shared__ BoundaryShared sh;
  // n_slabs workgroups per channel (finalize_frame; > 1 only behind an insert launch: no channel idle at the launch's start is
  // made active by it, so every workgroup of a channel reads the same ChanCtl::active)
  const int c = (int)blockIdx.x % chan_cnt + chan_off, slab = (int)blockIdx.x / chan_cnt;
  ChanCtl *ctl = D.ctl + c;
  if (ctl->active) {
    if (!finalize_frame<kLat, kBig>(D, c, ctl, sh, slab, n_slabs)) return;
  } else if (slab != 0) {
    return;
  }
  __syncthreads();
  if constexpr (kLat) {
    // (PruneActiveTokens, every prune_interval-th frame: lattice_prune_kernel, a launch of its own between this one -- which then
    // only closes the frame, do_prep 0 -- and the next expansion: its 128 KB of LDS would otherwise sit on every closure launch)
  } else {
    // best-path decoders: collect the arena's garbage when it passes its mark (gc_pass)
    __shared__ ScanShared ps;
    __shared__ GcShared gs;
    const int nd = ctl->n_decoded;
    if (do_prep && D.remap && nd > 0 && nd < target[c] && ctl->error == 0 && !ctl->finalized && nd < D.max_frames) {
      const int32_t *foff = D.frame_off + (size_t)c * (D.max_frames + 2);
      const int mark = ctl->lat_arcs > 0 ? ctl->lat_arcs : gc_base_mark(D);
      if (foff[nd + 1] > mark) gc_pass<kBig>(D, c, ps, gs);
    }
    __syncthreads();
  }
  if (do_prep) prep_frame<kBig>(D, c, ctl, target, sh, group, par);  // par: parity of the step it prepares
}

// PruneActiveTokens at the top of every prune_interval-th frame's iteration (base-inl.h:660-661), i.e. only when the channel
// goes on to decode that frame; then the frame's preparation (prep_frame: the pass has moved the frontier).  Launched by
// wfst_decoder_advance after the closure launch (do_prep 0) of the steps at which some channel of the group reaches a multiple
// of prune_interval; one 1024-thread workgroup per channel.
__device__ __forceinline__ bool prune_due(const DecoderDev &D, int c, const int32_t *target) {
  const ChanCtl *ctl = D.ctl + c;
  const int nd = ctl->n_decoded;
  return nd > 0 && nd % D.prune_interval == 0 && ctl->pruned_upto != nd && nd < target[c] && ctl->error == 0 && !ctl->finalized &&
         nd < D.max_frames;
}
// ... and is its share of never-priced links large enough for the several-workgroup path (lattice_prune_raw_kernel) to pay?  Its
// workgroups meet five times a frame; below a few hundred thousand links the one-workgroup walk in LDS is done sooner
// (beam 13 of the bench: 12 k links a frame; beam 15: 45 k, 400 k in the heaviest channels): DecoderDev::prune_raw_min.
#ifndef WFST_PR_RAW_MAXJ
#define WFST_PR_RAW_MAXJ 24
#endif
constexpr int kPrRawMaxJ = WFST_PR_RAW_MAXJ;        // workgroups a channel takes at most (a meeting of a hundred workgroups costs more than their shares save; 8 / 12 / 16 / 24 / 32 / 64: 142.1 / 138.9 / 139.1 / 138.5 / 140.5 / 153.0 ms per beam-15 step)
__device__ __forceinline__ int prune_raw_links(const DecoderDev &D, int c) {
  const ChanCtl *cl = D.ctl + c;
  const int32_t *lo = D.link_off + (size_t)c * (D.max_frames + 3);
  const int a = cl->pruned_upto < 0 ? 0 : cl->pruned_upto;
  return max(0, lo[cl->n_decoded + 1] - lo[a < cl->n_decoded ? a + 1 : cl->n_decoded]);
}
__device__ __forceinline__ bool prune_due_raw(const DecoderDev &D, int c, const int32_t *target) {
  return prune_due(D, c, target) && prune_raw_links(D, c) >= D.prune_raw_min;
}
template <bool kBig>
__global__ __launch_bounds__(kBT) void lattice_prune_kernel(DecoderDev D, const int32_t *target, int chan_off, int group, int par, int raw) {
  __shared__ PruneShared ps;
  __shared__ BoundaryShared sh;
  const int c = blockIdx.x + chan_off;
  int32_t *pp = D.prune_par + (size_t)c * kPrParInts;
  // (did lattice_prune_raw_kernel, the launch before, price this channel's raw frames?  The same function of the channel's control
  // block that launch decided by -- no flag to go stale)
  const bool raw_done = raw && prune_due_raw(D, c, target);
  if (threadIdx.x == 0) { pp[0] = 0; pp[kPrRawCount] = 0; }   // (the raw launch's meeting counter: back to 0 for the next pass)
  __syncthreads();
  if (prune_due(D, c, target)) prune_pass<false>(D, c, ps, raw_done);
  (void)sh; (void)group; (void)par;
}

// ---- the RAW frames of a running pass on several workgroups per channel (round 5) --------------------------------------------
// A pass prices the prune_interval frames decoded since the pass before for the first time -- nine tenths of the links it touches,
// a frame at a time, each frame after the one behind it -- and one workgroup per channel streamed them at what ONE compute unit
// moves (~20 GB/s), the launch as long as its heaviest channel.  lattice_prune_raw_kernel gives a channel kPrRawJ workgroups:
// each takes its share of a frame's emitting links -- priced against the final {extra, cost} pairs of the frame after it, atomicMin
// on the sources' extras in HBM -- and of its epsilon links, relaxed round by round to their fixpoint (a "changed" word per round
// in the channel's parameter block), the dead links marked, as relax_eps does; between the phases the channel's workgroups meet
// at a counter in that block (the grid is a few hundred workgroups, all resident: nobody waits for a workgroup that cannot
// start; a workgroup that waits longer than a few seconds gives up and flags the channel).  Everything the workgroups share inside
// the launch is written by atomics and read by agent-scope loads.  min is order independent and every link_extra is computed as in
// prune_pass: the extras, and with them which tokens and links die, are those of the single-workgroup walk bit for bit.
// lattice_prune_kernel then goes on with the frames priced before (its LDS path, the delta stopping rule) from frame pruned_upto - 1.
#ifndef WFST_PR_RAW_T
#define WFST_PR_RAW_T 256
#endif
#ifndef WFST_PR_RAW_U
#define WFST_PR_RAW_U 4
#endif
constexpr int kPrRawT = WFST_PR_RAW_T;
constexpr int kPrRawU = WFST_PR_RAW_U;

// all the workgroups of the channel meet: *seq counts this workgroup's barriers
__device__ __forceinline__ bool raw_barrier(int32_t *cnt, int J, int *seq, bool release) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores and atomics have been acknowledged
  __syncthreads();
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    if (release) __threadfence();
    const int want = (*seq + 1) * J;
    atomicAdd(cnt, 1);
    int ok = 1;
    for (unsigned spin = 0; ld_agent(cnt) < want; ++spin) {
      __builtin_amdgcn_s_sleep(4);
      if (spin > (1u << 19)) { ok = 0; break; }   // (~a quarter of a second; never expected: a workgroup of the channel that does not run)
    }
    if (release) __threadfence();
    s_ok = ok;
  }
  ++*seq;
  __syncthreads();
  return s_ok != 0;
}

constexpr int kPrRawMaxChan = 256;   // channels of a launch whose workgroups are shared out by their work (more: kPrRawJ each)

__global__ __launch_bounds__(kPrRawT) void lattice_prune_raw_kernel(DecoderDev D, const int32_t *target, int chan_off, int chan_cnt) {
  const int tid = threadIdx.x;
  // The launch's workgroups are shared out over its channels in proportion to the links their passes have to price (a channel's
  // heaviest frames have ten times the links of the batch's mean: with the same number of workgroups everywhere the launch would
  // last as long as that channel).  Every workgroup computes the same table from the channels' control blocks.
  __shared__ int s_base[kPrRawMaxChan + 1];
  __shared__ unsigned long long s_work[kPrRawMaxChan];
  int c, j, J;
  if (chan_cnt <= kPrRawMaxChan) {
    for (int q = tid; q < chan_cnt; q += kPrRawT) {
      const int cc = q + chan_off;
      s_work[q] = prune_due_raw(D, cc, target) ? 1ull + (unsigned long long)prune_raw_links(D, cc) : 0ull;
    }
    __syncthreads();
    if (tid == 0) {
      unsigned long long tot = 0;
      int n_due = 0;
      for (int q = 0; q < chan_cnt; ++q) { tot += s_work[q]; n_due += s_work[q] != 0; }
      const long long spare = (long long)gridDim.x - n_due;
      int b = 0;
      for (int q = 0; q < chan_cnt; ++q) {
        s_base[q] = b;
        if (s_work[q]) b += min(kPrRawMaxJ, 1 + (spare > 0 ? (int)((unsigned long long)spare * s_work[q] / tot) : 0));
      }
      s_base[chan_cnt] = b;
    }
    __syncthreads();
    const int w = (int)blockIdx.x;
    if (w >= s_base[chan_cnt]) return;   // (a workgroup without a share)
    int lo = 0, hi = chan_cnt - 1;   // the channel whose range holds w
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_base[mid] <= w) lo = mid; else hi = mid - 1;
    }
    if (!s_work[lo]) return;   // (never: a channel that is not due has the empty range [base, base), and the search takes the LAST channel whose base is <= w)
    c = lo + chan_off; j = w - s_base[lo]; J = s_base[lo + 1] - s_base[lo];
  } else {
    c = (int)blockIdx.x / kPrRawJ + chan_off; j = (int)blockIdx.x % kPrRawJ; J = kPrRawJ;
  }
  int32_t *pp = D.prune_par + (size_t)c * kPrParInts;
  if (!prune_due_raw(D, c, target)) return;
  ChanCtl *ctl = D.ctl + c;
  const int nd = ctl->n_decoded, n_prev = ctl->pruned_upto;
  const int4 *tok = D.tok + (size_t)c * D.arena_cap;
  int4 *links = D.links + (size_t)c * D.link_cap;
  uint2 *extra = D.extra + (size_t)c * D.arena_cap;
  const int32_t *foff = D.frame_off + (size_t)c * (D.max_frames + 2);
  const int32_t *loff = D.link_off + (size_t)c * (D.max_frames + 3);
  const int32_t *lmid = D.link_mid + (size_t)c * (D.max_frames + 3);
  const float kInf = __builtin_huge_valf();
  const uint32_t kInfO = f2o(kInf), kZeroO = f2o(0.0f);
  const float lb = D.lattice_beam;
  int32_t *cnt = pp + kPrRawCount, *chg = pp + kPrRawChg;
  int seq = 0;
  bool ok = true;
  // ---- the newest frame: extra 0 (PruneActiveTokens); every raw frame: +inf until its links are priced ----
  {
    const int lo = foff[n_prev < nd ? n_prev : nd], fn = foff[nd], hi = foff[nd + 1];
    // (agent-scope stores: the line does not stay behind in this XCD's L2, where a later agent-scope load would find it stale)
    for (int i = lo + j * kPrRawT + tid; i < hi; i += J * kPrRawT)
      __hip_atomic_store(reinterpret_cast<u64 *>(&extra[i]), (u64)(i >= fn ? kZeroO : kInfO) | (D.link_delta ? 0ull : (u64)(uint32_t)tok[i].y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (link_delta: the cost half is read by nobody)
    if (j == 0 && tid < 3) __hip_atomic_store(&chg[tid], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  ok = raw_barrier(cnt, J, &seq, true);   // (plain stores: written back before anybody's atomics and agent loads meet them)
  // link_extra as prune_pass computes it (base-inl.h:524-526)
  auto link_extra = [&](const int4 &L, u64 e) -> float {
    const uint32_t eo = (uint32_t)e;
    return eo >= kInfO ? kInf : o2f(eo) + (D.link_delta ? __int_as_float(L.w) : __int_as_float(L.w) - __int_as_float((int)(e >> 32)));
  };
  // f(i, L, le) over this workgroup's share of links [lo, hi)
  auto for_links = [&](int lo, int hi, auto &&f) {
    for (int i0 = lo + j * (kPrRawT * kPrRawU); i0 < hi; i0 += J * (kPrRawT * kPrRawU)) {
      int4 L[kPrRawU];
      u64 e[kPrRawU];
#pragma unroll
      for (int u = 0; u < kPrRawU; ++u) {
        const int i = i0 + u * kPrRawT + tid;
        L[u] = i < hi ? links[i] : make_int4(-1, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < kPrRawU; ++u) e[u] = L[u].x >= 0 ? ld_agent(reinterpret_cast<const u64 *>(&extra[L[u].y])) : 0ull;
#pragma unroll
      for (int u = 0; u < kPrRawU; ++u)
        if (L[u].x >= 0) f(i0 + u * kPrRawT + tid, L[u], link_extra(L[u], e[u]));
    }
  };
  u64 st_links = 0, st_toks = 0;
  for (int k = nd - 1; k >= n_prev && k >= 0 && ok; --k) {
    // ---- the emitting links frame k -> k + 1 ----
    const int m_lo = loff[k + 1], m_hi = lmid[k + 1], e0 = lmid[k], e1 = loff[k + 1];
    for_links(m_lo, m_hi, [&](int i, const int4 &L, float le) {
      if (!(le <= lb)) { links[i].x = -1; return; }
      if (le < 0.0f) le = 0.0f;
      atomicMin(&extra[L.x].x, f2o(le));
    });
    ok = raw_barrier(cnt, J, &seq, false);
    // ---- the epsilon links inside frame k, to their fixpoint ----
    if (e0 < e1) {
      for (int round = 0; round < 4096 && ok; ++round) {
        int ch = 0;
        for_links(e0, e1, [&](int, const int4 &L, float le) {
          if (!(le <= lb)) return;
          if (le < 0.0f) le = 0.0f;
          const uint32_t o = f2o(le);
          if (o < atomicMin(&extra[L.x].x, o)) ch = 1;
        });
        const int fl = round % 3;
        if (__any(ch) && (tid & 63) == 0) atomicOr(&chg[fl], 1);
        if (j == 0 && tid == 0) __hip_atomic_store(&chg[(fl + 1) % 3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (last read two barriers ago)
        ok = raw_barrier(cnt, J, &seq, false);
        if (!ld_agent(&chg[fl])) break;
      }
      for_links(e0, e1, [&](int i, const int4 &, float le) {
        if (!(le <= lb)) links[i].x = -1;
      });
      // (the flags of the next frame's rounds: every workgroup has read this frame's last one -- it was 0 -- or will read 0)
      if (j == 0 && tid < 3) __hip_atomic_store(&chg[tid], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    st_links += (u64)(m_hi - m_lo) + 2ull * (u64)(e1 - e0);
    st_toks += (u64)(foff[k + 1] - foff[k]);
  }
  if (j == 0 && tid == 0) {
    D.lat_stats[(size_t)c * 4 + 1] += st_links;
    D.lat_stats[(size_t)c * 4 + 2] += st_toks;
    if (!ok) atomicOr(&ctl->error, kErrInternal);
  }
}
// the compaction's flag sweeps: kPrSlabs workgroups per channel
__global__ __launch_bounds__(kBT) void lattice_prune_flags_kernel(DecoderDev D, int chan_off) {
  __shared__ ScanShared ps;
  prune_flags(D, (int)(blockIdx.x / kPrSlabs) + chan_off, (int)(blockIdx.x % kPrSlabs), ps);
}
// the compaction's moves, then the frame's preparation (prep_frame: the pass has moved the frontier)
template <bool kBig>
__global__ __launch_bounds__(kBT) void lattice_prune_move_kernel(DecoderDev D, const int32_t *target, int chan_off, int group, int par) {
  __shared__ ScanShared ps;
  __shared__ BoundaryShared sh;
  const int c = blockIdx.x + chan_off, role = blockIdx.y;
  prune_move(D, c, role, ps);
  __syncthreads();
  if (role == 0) prep_frame<kBig>(D, c, D.ctl + c, target, sh, group, par);
}

// =========================================================================================
// init: InitDecoding (base-inl.h:40-67)
// =========================================================================================
__global__ __launch_bounds__(kBT) void init_kernel(DecoderDev D, const int32_t *chans) {
  __shared__ BoundaryShared sh;
  const int c = chans ? chans[blockIdx.x] : blockIdx.x;
  const int tid = threadIdx.x;
  ChanCtl *ctl = D.ctl + c;
  u64 *vals = D.eps_vals + (size_t)c * D.ecap;
  // every closure leaves the epsilon table empty; after an error it may not be
  if (ctl->error || ctl->eps_occ || ctl->active) {
    for (int i = tid; i < D.ecap; i += kBT) vals[i] = kEmptyVal;
    if (D.big) for (int i = tid; i < D.ecap; i += kBT) D.eps_keys[(size_t)c * D.ecap + i] = kEmptyVal;
  }
  // (biglm: the channel's LM pair table was emptied by clear_pairs_kernel, the launch before this one)
  for (int i = tid; i < D.n_part; i += kBT) D.bucket_cnt[(size_t)c * D.n_part + i] = 0;
  __syncthreads();
  if (tid == 0) {
    if (D.lat_stats) for (int q = 0; q < 4; ++q) D.lat_stats[(size_t)c * 4 + q] = 0;
    D.emit_cnt[c * 32] = 0;
    D.degraded[c] = 0;
    ChanCtl z;
    memset(&z, 0, sizeof(z));
    z.best_next = ~0ull;
    *ctl = z;
    sh.err = 0; sh.wl_n[0] = 0; sh.wl_n[1] = 0; sh.nnew = 1; sh.occ = 0; sh.best = ~0ull;
    const uint32_t ne = D.g.start_eps, fl = flags_of(ne);
    D.tok[(size_t)c * D.arena_cap] = make_int4(D.g.start, __float_as_int(0.0f), -1, (int)(kNoArc | fl));
    int start_lm = 0;
    if (D.big) {  // biglm.h:112: start pair = (graph start, _diff_lm.Start())
      start_lm = (int)(hash_pair(D.lm_old.start, D.lm_new.start) & ((uint32_t)D.pair_cap - 1u));
      D.pair_keys[(size_t)c * D.pair_cap + start_lm] = (u64)(uint32_t)D.lm_old.start | ((u64)(uint32_t)D.lm_new.start << 32);
      D.tok_lm[(size_t)c * D.arena_cap] = start_lm;
      D.pair_list[(size_t)c * D.pair_cap] = start_lm;
      ctl->pair_count = 1;
    }
    if (fl & kFlagEpsTarget) {
      int ord = (int)(ne & 0x7FFFFFFFu) - 1;
      if (D.big) {
        ord = (int)(hash_big(D.g.start, start_lm) & ((uint32_t)D.ecap - 1u));
        D.eps_keys[(size_t)c * D.ecap + ord] = big_key(D.g.start, start_lm);
      }
      vals[ord] = ((u64)f2o(0.0f) << 32) | kNoArc;
      D.eps_toki[(size_t)c * D.ecap + ord] = 0;
      D.eps_occ_list[(size_t)c * D.wl_cap] = ord;
      sh.occ = 1;
    }
    if (fl & kFlagOutEps) { D.worklist[(size_t)c * 2 * D.wl_cap] = make_int4(start_lm, D.g.start, __float_as_int(0.0f), 0); sh.wl_n[0] = 1; }
  }
  __syncthreads();
  u64 nZ = 0;
  if (D.big && D.lattice) epsilon_closure<true, true>(D, c, sh, 0, D.beam, &nZ);
  else if (D.big) epsilon_closure<false, true>(D, c, sh, 0, D.beam, &nZ);
  else if (D.lattice) epsilon_closure<true, false>(D, c, sh, 0, D.beam, &nZ);  // ProcessNonemitting(_config._beam)
  else epsilon_closure<false, false>(D, c, sh, 0, D.beam, &nZ);
  if (D.degcode) {
    // degree codes (wfst_device.h): the tokens of the start state's closure -- the only ones a closure pass ever creates
    // for such a decoder -- take theirs from the row headers (the root keeps -1: no code, header path)
    __syncthreads();
    const int nf0 = min(sh.nnew, (int)min((int64_t)D.max_tok, D.arena_cap));
    for (int i = tid; i < nf0; i += blockDim.x) {
      int4 t = D.tok[(size_t)c * D.arena_cap + i];
      const int4 hdr = D.g.arcs[t.x];
      const uint32_t code = pack_code((uint32_t)hdr.x & kEpsMask, (uint32_t)hdr.x >> kEpsBits, (uint32_t)hdr.z);
      if (t.z == kPrevUnresolved) {
        t.w = (int)(((uint32_t)t.w & 0x3FFFFFFFu) | (code << 30));
        t.z = kPrevUnresolved - (int)(code >> 2);
      } else {
        t.w = (int)((uint32_t)t.w | (3u << 30));   // the root (z = -1) reads as "no code": its row header is read
      }
      D.tok[(size_t)c * D.arena_cap + i] = t;
    }
    __syncthreads();
  }
  if (tid == 0) {
    int nf = sh.nnew;
    if (sh.err || nf > D.max_tok || nf > D.arena_cap) nf = 0;
    D.frame_off[(size_t)c * (D.max_frames + 2) + 0] = 0;
    D.frame_off[(size_t)c * (D.max_frames + 2) + 1] = nf;
    if (D.lattice) {
      D.link_off[(size_t)c * (D.max_frames + 3) + 0] = 0;
      D.link_mid[(size_t)c * (D.max_frames + 3) + 0] = 0;
      D.link_off[(size_t)c * (D.max_frames + 3) + 1] = min(ctl->link_count, (int)D.link_cap);
    }
    D.cutoff_hist[(size_t)c * (D.max_frames + 2) + 0] = D.beam;
    const u64 root = ((u64)f2o(0.0f) << 32) | (D.best_row ? (uint32_t)D.g.start : 0u);
    ctl->best_next = sh.best < root ? sh.best : root;
    ctl->front_begin = 0;
    ctl->front_count = nf;
    ctl->cnt_tok = (u64)nf;
    ctl->peak_tokens = nf;
    if (sh.err) ctl->error |= sh.err;
  }
}

// =========================================================================================
// best path: BestPathEnd + TraceBackBestPath + GetBestPath (base-inl.h:1071-1200), one 256-thread
// workgroup per channel.  Lane 0 walks the backpointer chain once (one dependent load per hop)
// and records the token indices; all lanes then resolve the hops in parallel.  Hops are written
// in start->final order; hop 0 is the root token's (0,0,One) arc.
// =========================================================================================
constexpr int kBpThreads = 256;
constexpr int kBpChainLds = 4096;   // hops of the backpointer walk kept in LDS (the walk's own list; longer paths go on in HBM)
constexpr int kBpFrames = 3072;   // utterances up to this many frames keep their frame bounds in LDS (longer ones read them from HBM)

// kBig (biglm): final costs carry the LM's (ComputeFinalCosts, biglm.h:160-215), hop graph costs are arc
// weight + lm_score, an epsilon-won token's predecessor is found by (state, LM pair, cost), and after
// FinalizeDecoding the reference's final pruning can leave NO token (its final_best_cost ranges over
// every token's cost + LM final cost, graph-final or not, :186-188) -- reproduced: no path.
template <bool kBig>
__global__ __launch_bounds__(kBpThreads) void best_path_kernel(DecoderDev D, const int32_t *chans, int use_final, int cap,
                                                               int32_t *o_il, int32_t *o_ol, float *o_g, float *o_ac,
                                                               int32_t *n_hops, int32_t *chain) {
  const int bi = blockIdx.x;
  const int c = chans ? chans[bi] : bi;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ChanCtl *ctl = D.ctl + c;
  const int n = ctl->front_count, nd = ctl->n_decoded;
  if (nd <= 0 || n == 0) {  // base-inl.h:1104-1108 / 1148-1154: no path
    if (tid == 0) n_hops[bi] = 0;
    return;
  }
  __shared__ u64 s_all[kBpThreads / 64], s_fin[kBpThreads / 64], s_wf[kBpThreads / 64];
  __shared__ int s_len;
  const int4 *tok = D.tok + (size_t)c * D.arena_cap;
  const int32_t *tok_lm = kBig ? D.tok_lm + (size_t)c * D.arena_cap : nullptr;
  const int fb = ctl->front_begin;
  u64 best_all = ~0ull, best_fin = ~0ull, best_wf = ~0ull;  // best_wf: min cost + LM final cost over ALL tokens (biglm)
  unsigned long long tb0 = (D.dbg & 32) ? wall_clock64() : 0ull, tb_walk = 0, tb_scan = 0;
  int n_unres = 0;
  for (int i = tid; i < n; i += kBpThreads) {
    const int4 t = tok[fb + i];
    const u64 v = ((u64)f2o(__int_as_float(t.y)) << 32) | (uint32_t)(fb + i);
    best_all = v < best_all ? v : best_all;
    if constexpr (kBig) {
      const u64 pk = D.pair_keys[(size_t)c * D.pair_cap + tok_lm[fb + i]];
      const float lm_final = lm_final_cost(D.lm_old, (int)(uint32_t)pk) + lm_final_cost(D.lm_new, (int)(uint32_t)(pk >> 32));  // diff-lm.h:48-53
      const u64 w = ((u64)f2o(__int_as_float(t.y) + lm_final) << 32) | (uint32_t)(fb + i);
      best_wf = w < best_wf ? w : best_wf;
      if (t.x == D.g.final_state) best_fin = w < best_fin ? w : best_fin;
    } else {
      if (t.x == D.g.final_state) best_fin = v < best_fin ? v : best_fin;  // IsFinal, optimize-fst.h:189-192
    }
  }
  best_all = wave_min_u64(best_all);
  best_fin = wave_min_u64(best_fin);
  best_wf = wave_min_u64(best_wf);
  if (lane == 0) { s_all[wave] = best_all; s_fin[wave] = best_fin; s_wf[wave] = best_wf; }
  __syncthreads();
  int32_t *ch = chain + (size_t)bi * cap;
  const int32_t *foff_g = D.frame_off + (size_t)c * (D.max_frames + 2);
  // the frame bounds of the utterance in LDS (both the walk and the hop pass search them; from HBM a search was nine dependent loads)
  __shared__ int32_t s_foff[kBpFrames + 2];
  const bool foff_lds = nd + 2 <= kBpFrames + 2;
  if (foff_lds) for (int i = tid; i < nd + 2; i += kBpThreads) s_foff[i] = foff_g[i];
  const int32_t *foff = foff_lds ? s_foff : foff_g;
  __shared__ int32_t s_chain[kBpChainLds];   // the walk's hops, last hop first
  __shared__ int s_t, s_need, s_lo, s_hi, s_found;
  __shared__ float s_extra0;   // extra cost of the best path's tokens after FinalizeDecoding (0 except in biglm, below)
  if (tid == 0) {
    s_extra0 = 0.0f;
    for (int w = 1; w < kBpThreads / 64; ++w) {
      best_all = s_all[w] < best_all ? s_all[w] : best_all;
      best_fin = s_fin[w] < best_fin ? s_fin[w] : best_fin;
      best_wf = s_wf[w] < best_wf ? s_wf[w] : best_wf;
    }
    const u64 best = (use_final && best_fin != ~0ull) ? best_fin : best_all;
    s_t = (int)(uint32_t)best;
    if (kBig && ctl->finalized) {
      // PruneForwardLinksFinal (biglm.h:468-568): tok_extra_cost = tot_cost + final_cost - final_best_cost
      // of the cheapest candidate; above lattice_beam it -- and with it every token -- is pruned away
      const float fbc = o2f((uint32_t)(best_wf >> 32));
      const float own = (best_fin != ~0ull) ? o2f((uint32_t)(best_fin >> 32)) : (o2f((uint32_t)(best_all >> 32)) + 0.0f);
      if ((own - fbc) > D.lattice_beam) s_t = -1;
      // final_best_cost ranges over non-final tokens too (biglm.h:186-188), so the best final token -- and with it every
      // token of its path, whose links to their successors cost nothing extra -- carries this extra cost, and a parallel
      // arc survives the final pruning only with it counted (the hop loop below)
      s_extra0 = own - fbc;
    }
    s_len = 0;
  }
  // Walk the backpointer chain (last hop first, packed against the end of ch[]).  One thread
  // follows resolved backpointers; a token won by an epsilon arc carries kPrevUnresolved and the
  // whole workgroup scans its frame for the token of the arc's source state.
  if (tid == 0 && (D.dbg & 32)) { const unsigned long long now = wall_clock64(); atomicAdd(&D.dbg_t[110], now - tb0); tb0 = now; }
  for (;;) {
    __syncthreads();
    if (s_t < 0) break;
    unsigned long long tw0 = (tid == 0 && (D.dbg & 32)) ? wall_clock64() : 0ull;
    if (tid == 0) {
      // resolved backpointers are followed in one go (a dependent load per hop, nothing else on the chain); the walk stops at a
      // token won by an epsilon arc, whose predecessor the whole workgroup looks for
      int t = s_t, len = s_len;
      const uint32_t idx_mask = D.tok_idx_bits >= 31 ? 0x7FFFFFFFu : ((1u << D.tok_idx_bits) - 1u);   // (a degree code may sit above the index)
      s_need = -1;
      while (t >= 0 && len < (1 << 24)) {   // (the bound: a damaged arena must not hang the device)
        const int4 T = tok[t];
        // (the hop list stays in LDS until the walk is over: on this target a load issued behind a global store waits for the store)
        if (len < kBpChainLds) s_chain[len] = t;
        else if (len < cap) ch[cap - 1 - len] = t;
        ++len;
        if (T.z <= kPrevUnresolved) {
          int lo = 0, hi = nd + 1;  // frame of t: frame_off[f] <= t < frame_off[f+1]
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (foff[mid] <= t) lo = mid; else hi = mid;
          }
          s_lo = foff[lo];
          s_hi = foff[lo + 1];
          s_need = D.g.arc_src[(uint32_t)T.w & kArcMask] & 0x7FFFFFFF;
          s_found = -1;
          break;
        }
        t = T.z >= 0 ? (int)((uint32_t)T.z & idx_mask) : T.z;
      }
      s_t = t;
      s_len = len;
      if (D.dbg & 32) { const unsigned long long now = wall_clock64(); tb_walk += now - tw0; tw0 = now; }
    }
    __syncthreads();
    if (s_need >= 0) {
      ++n_unres;
      const int need = s_need, t = s_t;
      constexpr int kScanU = 4;   // states of the frame in flight per thread
      for (int i0 = s_lo + tid; i0 < s_hi; i0 += kBpThreads * kScanU) {
        int sx[kScanU];
#pragma unroll
        for (int u = 0; u < kScanU; ++u) {
          const int i = i0 + u * kBpThreads;
          sx[u] = i < s_hi ? reinterpret_cast<const int *>(tok + i)[0] : -1;
        }
#pragma unroll
        for (int u = 0; u < kScanU; ++u) {
        const int i = i0 + u * kBpThreads;
        if (sx[u] != need) continue;
        [[maybe_unused]] const int4 S = tok[i];
        if constexpr (kBig) {
          // several tokens may sit on the arc's source state, one per LM state: the predecessor is the
          // one whose LM state and cost lead to this token over the winning arc
          const int4 T = tok[t];
          const int a = (int)((uint32_t)T.w & kArcMask);
          const int ol = D.g.arc_olabel[a];
          int nlm = tok_lm[i];
          float lm_score = 0.0f;
          if (ol != 0) {
            int n1, n2;
            lm_score = lm_step(D, c, tok_lm[i], ol, &n1, &n2);
            nlm = pair_find(D, c, n1, n2);
          }
          const float tot = __int_as_float(S.y) + (__int_as_float(D.g.arcs[a].z) + lm_score);
          if (nlm != tok_lm[t] || __float_as_int(tot) != T.y) continue;
        }
        s_found = i;
        }
      }
      __syncthreads();
      if (tid == 0) s_t = s_found;  // -1 (never expected) ends the walk
      if (tid == 0 && (D.dbg & 32)) tb_scan += wall_clock64() - tw0;
    }
  }
  if (tid == 0 && (D.dbg & 32)) {
    atomicAdd(&D.dbg_t[111], tb_walk); atomicAdd(&D.dbg_t[112], tb_scan); atomicAdd(&D.dbg_t[114], (unsigned long long)s_len);
    atomicAdd(&D.dbg_t[115], (unsigned long long)n_unres); atomicAdd(&D.dbg_t[116], 1ull);
    tb0 = wall_clock64();
  }
  if (tid == 0) n_hops[bi] = s_len;
  __syncthreads();
  const int len = s_len;
  if (len > cap) return;
  for (int p = tid; p < min(len, kBpChainLds); p += kBpThreads) ch[cap - 1 - p] = s_chain[p];
  __syncthreads();
  int32_t *il = o_il + (size_t)bi * cap, *ol = o_ol + (size_t)bi * cap;
  float *og = o_g + (size_t)bi * cap, *oa = o_ac + (size_t)bi * cap;
  const float *cut = D.cutoff_hist + (size_t)c * (D.max_frames + 2);
  const float *ll = D.ll_base[c];
  // forward links of frame f have met PruneForwardLinks iff a PruneActiveTokens pass started at
  // NumFramesDecoded() = m >= f+1 (base-inl.h:660-661, 445-476) or FinalizeDecoding ran
  const int m_last = ((nd - 1) / D.prune_interval) * D.prune_interval;
  const float extra0 = s_extra0;
  for (int pos = tid; pos < len; pos += kBpThreads) {
    auto hop_at = [&](int q) { const int p = len - 1 - q; return p < kBpChainLds ? s_chain[p] : ch[cap - 1 - p]; };
    const int t = hop_at(pos);
    const int4 T = tok[t];
    const int prev = pos > 0 ? hop_at(pos - 1) : -1;  // the chain itself holds the resolved backpointers
    if (prev < 0) {  // base-inl.h:1193-1198
      il[pos] = 0; ol[pos] = 0; og[pos] = 0.f; oa[pos] = 0.f;
      continue;
    }
    // frame of t: the f with frame_off[f] <= t < frame_off[f+1]
    int lo = 0, hi = nd + 1;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (foff[mid] <= t) lo = mid; else hi = mid;
    }
    const int fr = lo;
    const int4 Pt = tok[prev];
    const float cb = __int_as_float(Pt.y), ct = __int_as_float(T.y);
    const int warc = (int)((uint32_t)T.w & kArcMask);
    const bool eps = prev >= foff[fr];  // backpointer on the same frame <=> epsilon hop
    const int fbp = eps ? fr : fr - 1;
    const uint2 si = make_uint2((uint32_t)Pt.x + 1u, (uint32_t)D.g.arcs[Pt.x].x);
    const int ne = (int)(si.y & kEpsMask);
    const int ahi = eps ? (int)si.x + ne : (int)si.x + ne + (int)(si.y >> kEpsBits);
    const bool pruned_once = ctl->finalized || m_last >= fbp + 1;
    const float *llrow = ll + (size_t)(eps ? 0 : fbp) * D.stride;
    // biglm: LM score of an arc taken from the predecessor's LM state, and the pair state it leads to
    // (-1: a pair no token was ever created with)
    auto arc_lm = [&](int a, float *lm_score) -> int {
      *lm_score = 0.0f;
      if constexpr (!kBig) return 0;
      else {
        const int ol = D.g.arc_olabel[a];
        if (ol == 0) return tok_lm[prev];
        int n1, n2;
        *lm_score = lm_step(D, c, tok_lm[prev], ol, &n1, &n2);
        return pair_find(D, c, n1, n2);
      }
    };
    int chosen = warc;
    // TraceBackBestPath takes the FIRST link bp->tok; links are prepended in arc order
    // (base-inl.h:340-341, 1169-1186), so a surviving parallel arc of higher index shadows the
    // winning one.
    for (int a = ahi - 1; a > warc; --a) {
      const int4 B = D.g.arcs[a];
      if (B.w != T.x) continue;
      float alt_g = __int_as_float(B.z);
      if constexpr (kBig) {
        float ls;
        if (arc_lm(a, &ls) != tok_lm[t]) continue;  // leads to another (state, LM state) token
        alt_g = __int_as_float(B.z) + ls;
      }
      const float alt_ac = eps ? 0.f : -llrow[B.x & D.g.col_mask];
      const float alt_tot = eps ? cb + alt_g : (cb + alt_ac) + alt_g;
      if (!(alt_tot < cut[fr])) continue;  // link never created
      if (pruned_once && ((ctl->finalized ? extra0 : 0.0f) + (alt_tot - ct)) > D.lattice_beam) continue;  // base-inl.h:524-532
      chosen = a;
      break;
    }
    const int4 C = D.g.arcs[chosen];
    il[pos] = D.g.arc_ilabel[chosen];
    ol[pos] = D.g.arc_olabel[chosen];
    if constexpr (kBig) {
      float ls;
      arc_lm(chosen, &ls);
      og[pos] = __int_as_float(C.z) + ls;  // graph_cost = arc weight + lm_score (biglm.h:380,450)
    } else {
      og[pos] = __int_as_float(C.z);
    }
    oa[pos] = eps ? 0.f : -llrow[C.x & D.g.col_mask];
  }
  if (tid == 0 && (D.dbg & 32)) atomicAdd(&D.dbg_t[113], wall_clock64() - tb0);
}

// GetRawLattice's raw material: every token and link alive right now, resolved to labels and costs, in the
// channel's compact lat_toks[] / lat_arcs[].  use_final != 0: final states by ComputeFinalCosts (base-inl.h:
// 670-720, 924-940): the graph-final tokens of the newest frame if there are any, else all of it.
constexpr int kEmitFrames = 4096, kEmitU = 8;   // frames whose bounds the emit kernels keep in LDS (longer utterances read them from HBM); items per thread and sweep
constexpr int kEmitSlabs = 8;                   // workgroups per channel of the emit sweeps
// GetRawLattice's listing (base-inl.h:869-975) of what is alive in the arena and in the link store -- after FinalizeDecoding, or
// mid-utterance -- into lat_toks[] (arena order = frame order: the n-best search relies on a frame's states being contiguous) and
// lat_arcs[] (any order).  Both stores hold their dead as holes (a channel at beam 13: 130-330 k arena entries for 1-2.5 k living
// tokens), so the listing is two sweeps over everything; kEmitSlabs workgroups per channel share them:
//   lattice_emit_kernel    slab g of the arena: slab-relative rank of every living token (remap[], -1 dead) and the slab's count;
//                          slab g of the link store: every living link appended to lat_arcs[] (one counter per channel, one atomic per wave)
//   lattice_emit_tokens_kernel   the slabs' bases from their counts; the living tokens of slab g written to lat_toks[base + rank]
// (the channel's counters sit in its parameter block, DecoderDev::prune_par[16..]: slab counts, link counter, error; reset by
// lattice_emit_reset_kernel).
struct EmitBounds {
  int foff[kEmitFrames + 2], lseg[2 * (kEmitFrames + 2)];   // lseg[2 f] = link_off[f], [2 f + 1] = link_mid[f]
};
__device__ __forceinline__ int emit_slab_len(int len) { return ((len + kEmitSlabs - 1) / kEmitSlabs + 63) & ~63; }

__global__ void lattice_emit_reset_kernel(DecoderDev D, const int32_t *chans, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 16) return;
  const int c = chans ? chans[i >> 4] : (i >> 4);
  D.prune_par[(size_t)c * kPrParInts + 16 + (i & 15)] = 0;
}

__global__ __launch_bounds__(kBT) void lattice_emit_kernel(DecoderDev D, const int32_t *chans, int use_final) {
  const int c = chans ? chans[blockIdx.x] : blockIdx.x;
  const int g = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  ChanCtl *ctl = D.ctl + c;
  const int nd = ctl->n_decoded;
  const int4 *links = D.links + (size_t)c * D.link_cap;
  LatArc *out_arcs = D.lat_arcs + (size_t)c * D.lat_arc_cap;
  const int32_t *foff = D.frame_off + (size_t)c * (D.max_frames + 2);
  const int32_t *loff = D.link_off + (size_t)c * (D.max_frames + 3);
  const int32_t *lmid = D.link_mid + (size_t)c * (D.max_frames + 3);
  const float *ll = D.ll_base[c];
  int32_t *pp = D.prune_par + (size_t)c * kPrParInts + 16;   // [g] living tokens of slab g, [8] links listed, [9] error
  int32_t *remap = D.remap + (size_t)c * D.arena_cap;
  if (ctl->error) return;
  // tokens.  Frames the back-pruning has priced (below pruned_upto) hold their dead as holes: extra = +inf -- except frame 0 before
  // FinalizeDecoding, whose dead tokens the reference keeps (link-less: PruneActiveTokens never calls PruneTokensForFrame(0),
  // base-inl.h:471-476).
  const uint2 *extra = D.extra + (size_t)c * D.arena_cap;
  const uint32_t kInfO = f2o(__builtin_huge_valf());
  const int pruned_upto = ctl->pruned_upto;
  const bool finalized = ctl->finalized != 0;
  __shared__ EmitBounds sb;
  __shared__ ScanShared ps;
  const bool in_lds = nd + 2 <= kEmitFrames + 2;
  if (in_lds) {
    for (int f = tid; f <= nd + 1; f += kBT) { sb.lseg[2 * f] = loff[f]; sb.lseg[2 * f + 1] = f <= nd ? lmid[f] : loff[f]; }
  }
  __syncthreads();
  {  // the slab's living tokens: rank within the slab (kEmitU consecutive tokens per thread and sweep, one workgroup-wide prefix per sweep)
    const int n_all = foff[nd + 1];
    const int f_pruned = foff[min(max(pruned_upto, 0), nd + 1)], f_one = foff[1];   // first token of the first unpriced frame; of frame 1
    const int sl = emit_slab_len(n_all), lo = min(n_all, g * sl), hi = min(n_all, lo + sl);
    int base = 0;
    for (int i0 = lo; i0 < hi; i0 += kBT * kEmitU) {
      bool alive[kEmitU];
      uint32_t ex[kEmitU];
#pragma unroll
      for (int u = 0; u < kEmitU; ++u) {
        const int i = i0 + tid * kEmitU + u;
        ex[u] = i < hi ? extra[i].x : 0u;
      }
      int cnt = 0;
#pragma unroll
      for (int u = 0; u < kEmitU; ++u) {
        const int i = i0 + tid * kEmitU + u;
        // dead = priced by the back-pruning (its frame is below pruned_upto), extra +inf, and not one of frame 0's before FinalizeDecoding
        alive[u] = i < hi && !(ex[u] >= kInfO && i < f_pruned && !(i < f_one && !finalized));
        cnt += alive[u] ? 1 : 0;
      }
      int tot;
      int p = base + block_exscan(cnt, ps, &tot);
#pragma unroll
      for (int u = 0; u < kEmitU; ++u) {
        const int i = i0 + tid * kEmitU + u;
        if (i < hi) remap[i] = alive[u] ? p : -1;
        p += alive[u] ? 1 : 0;
      }
      base += tot;
    }
    if (tid == 0) pp[g] = base;
  }
  // links: a flat pass over the slab, kEmitU per thread in flight; a living link's segment (found in the bounds) tells the source
  // frame and whether the arc is an epsilon
  auto emit_links = [&](auto lseg_at) {
    const int l_lo = lseg_at(0), l_hi = lseg_at(2 * (nd + 1));
    const int sl = emit_slab_len(l_hi - l_lo), lo = min(l_hi, l_lo + g * sl), hi = min(l_hi, lo + sl);
    for (int i0 = lo; i0 < hi; i0 += kBT * kEmitU) {
      int4 L[kEmitU];
#pragma unroll
      for (int u = 0; u < kEmitU; ++u) {
        const int i = i0 + u * kBT + tid;
        L[u] = i < hi ? links[i] : make_int4(-1, 0, 0, 0);
      }
      int4 A[kEmitU];
      int il[kEmitU], ol[kEmitU];
#pragma unroll
      for (int u = 0; u < kEmitU; ++u) {
        const bool on = L[u].x >= 0;
        A[u] = on ? D.g.arcs[L[u].z] : make_int4(0, 0, 0, 0);
        il[u] = on ? D.g.arc_ilabel[L[u].z] : 0;
        ol[u] = on ? D.g.arc_olabel[L[u].z] : 0;
      }
#pragma unroll
      for (int u = 0; u < kEmitU; ++u) {
        const bool on = L[u].x >= 0;
        const u64 m = __ballot(on);
        if (!m) continue;
        int wb = 0;
        if (lane == 0) wb = atomicAdd(&pp[8], __popcll(m));
        wb = __builtin_amdgcn_readfirstlane(wb);
        if (!on) continue;
        const int i = i0 + u * kBT + tid;
        int lo2 = 0, hi2 = 2 * (nd + 1);   // segment q: lseg[q] <= i < lseg[q + 1]; q = 2 f: emitting links into frame f, 2 f + 1: epsilon links inside f
        while (hi2 - lo2 > 1) {
          const int mid = (lo2 + hi2) >> 1;
          if (lseg_at(mid) <= i) lo2 = mid; else hi2 = mid;
        }
        const int f = lo2 >> 1;
        const bool eps = lo2 & 1;
        const int src_frame = eps ? f : f - 1;
        const int p = wb + lane_rank(m);
        if (p >= D.lat_arc_cap) { pp[9] = 1; continue; }
        LatArc o;
        o.src_tok = L[u].x; o.dst_tok = L[u].y;
        o.ilabel = eps ? 0 : il[u];
        o.olabel = ol[u];
        o.graph = __int_as_float(A[u].z);
        if (D.big && o.olabel != 0) {   // biglm: graph_cost = arc weight + lm_score from the source token's LM state (biglm.h:377-388, 448-452)
          int n1, n2;
          o.graph = __int_as_float(A[u].z) + lm_step(D, c, D.tok_lm[(size_t)c * D.arena_cap + L[u].x], o.olabel, &n1, &n2);
        }
        o.acoustic = eps ? 0.0f : -ll[(size_t)src_frame * D.stride + (A[u].x & D.g.col_mask)];
        o.src_frame = src_frame; o.is_eps = eps ? 1 : 0;
        out_arcs[p] = o;
      }
    }
  };
  if (in_lds) emit_links([&](int q) { return sb.lseg[q]; });
  else emit_links([&](int q) { return (q & 1) ? ((q >> 1) <= nd ? lmid[q >> 1] : loff[q >> 1]) : loff[q >> 1]; });
}

__global__ __launch_bounds__(kBT) void lattice_emit_tokens_kernel(DecoderDev D, const int32_t *chans, int use_final) {
  const int c = chans ? chans[blockIdx.x] : blockIdx.x;
  const int g = blockIdx.y;
  const int tid = threadIdx.x;
  ChanCtl *ctl = D.ctl + c;
  const int nd = ctl->n_decoded;
  const int4 *tok = D.tok + (size_t)c * D.arena_cap;
  int4 *out_toks = D.lat_toks + (size_t)c * D.lat_tok_cap;
  const int32_t *foff = D.frame_off + (size_t)c * (D.max_frames + 2);
  const int32_t *pp = D.prune_par + (size_t)c * kPrParInts + 16;
  const int32_t *remap = D.remap + (size_t)c * D.arena_cap;
  __shared__ int s_any_final, s_err;
  __shared__ int s_foff[kEmitFrames + 2];
  if (tid == 0) { s_any_final = 0; s_err = 0; }
  __syncthreads();
  if (ctl->error) return;
  if (use_final) {
    int any = 0;
    for (int i = foff[nd] + tid; i < foff[nd + 1]; i += kBT) any |= tok[i].x == D.g.final_state;
    if (any) s_any_final = 1;
  }
  const bool in_lds = nd + 2 <= kEmitFrames + 2;
  if (in_lds)
    for (int f = tid; f <= nd + 1; f += kBT) s_foff[f] = foff[f];
  __syncthreads();
  const bool any_final = s_any_final != 0;
  int base = 0, total = 0;
#pragma unroll
  for (int q = 0; q < kEmitSlabs; ++q) { base += q < g ? pp[q] : 0; total += pp[q]; }
  const int n_all = foff[nd + 1];
  const int sl = emit_slab_len(n_all), lo = min(n_all, g * sl), hi = min(n_all, lo + sl);
  auto write_slab = [&](auto foff_at) {
    for (int i0 = lo; i0 < hi; i0 += kBT * kEmitU) {
      int r[kEmitU];
#pragma unroll
      for (int u = 0; u < kEmitU; ++u) {
        const int i = i0 + u * kBT + tid;
        r[u] = i < hi ? remap[i] : -1;
      }
#pragma unroll
      for (int u = 0; u < kEmitU; ++u) {
        if (r[u] < 0) continue;
        const int i = i0 + u * kBT + tid, p = base + r[u];
        if (p >= D.lat_tok_cap) { s_err = 1; continue; }
        int flo = 0, fhi = nd + 1;  // frame of token i: foff[f] <= i < foff[f+1]
        while (fhi - flo > 1) {
          const int mid = (flo + fhi) >> 1;
          if (foff_at(mid) <= i) flo = mid; else fhi = mid;
        }
        const int4 t = tok[i];
        const int fin = (flo == nd && (!use_final || !any_final || t.x == D.g.final_state)) ? 1 : 0;
        out_toks[p] = make_int4(i, D.g.arcs[t.x].y, t.y, flo | (fin << 30));  // .y: the graph's own state id (row header)
      }
    }
  };
  if (in_lds) write_slab([&](int f) { return s_foff[f]; });
  else write_slab([&](int f) { return foff[f]; });
  __syncthreads();
  if (tid == 0 && (s_err || (g == 0 && pp[9]))) atomicOr(&ctl->error, kErrLinksFull);
  if (tid == 0 && g == 0) {
    ctl->lat_arcs = min(pp[8], D.lat_arc_cap);
    ctl->lat_toks = min(total, D.lat_tok_cap);
  }
}

__global__ __launch_bounds__(kBT) void lattice_finalize_kernel(DecoderDev D, const int32_t *chans) {
  __shared__ PruneShared ps;
  const int c = chans ? chans[blockIdx.x] : blockIdx.x;
  prune_pass<true>(D, c, ps);
}

// =========================================================================================
// launch wrappers
// =========================================================================================
static __global__ void set_finalized_kernel(DecoderDev D, const int32_t *chans, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) D.ctl[chans ? chans[i] : i].finalized = 1;
}

// DiffArpaLm::Reset (newlm/diff-lm.h:37-44, biglm.h:110): forget the utterance's LM pair states.  Its own
// launch, many workgroups per channel: plain stores here, agent-scope atomics in the launches after.
// The slots the utterance before claimed are listed (pair_intern: pair_list[0 .. pair_count)): those are emptied -- a few
// thousand of a million (the whole table was 2.1 GB of stores per InitDecoding of 128 channels); a table that overflowed
// (kErrPairsFull: claims beyond the list) is emptied whole.  Runs in front of init_kernel, which resets the count.
__global__ __launch_bounds__(256) void clear_pairs_kernel(DecoderDev D, const int32_t *chans) {
  const int c = chans ? chans[blockIdx.x] : blockIdx.x;
  u64 *pk = D.pair_keys + (size_t)c * D.pair_cap;
  const int n = D.ctl[c].pair_count;
  if (n >= (D.pair_cap >> 2) * 3) {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < D.pair_cap; i += gridDim.y * blockDim.x) pk[i] = kEmptyVal;
  } else {
    const int32_t *list = D.pair_list + (size_t)c * D.pair_cap;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) pk[list[i]] = kEmptyVal;
  }
}
void launch_init(const DecoderDev &D, const int32_t *chans, int n, hipStream_t s) {
  if (D.big) hipLaunchKernelGGL(clear_pairs_kernel, dim3(n, 8), dim3(256), 0, s, D, chans);
  hipLaunchKernelGGL(init_kernel, dim3(n), dim3(kBT), 0, s, D, chans);
}
// chan_off / chan_cnt: the channel group a launch covers (groups run on their own streams so that
// one group's latency-bound closure overlaps another group's expand / insert)
void launch_expand(const DecoderDev &D, int group, int par, int n_workgroups, hipStream_t s) {
  if (D.dbg & 128) {   // phase timers: their own instantiations
    if (D.fused && D.staged && D.ll_row) hipLaunchKernelGGL(expand_kernel_staged_row_timed, dim3(n_workgroups), dim3(kStThreads), 0, s, D, group, par);
    else if (D.fused && D.staged) hipLaunchKernelGGL(expand_kernel_staged_timed, dim3(n_workgroups), dim3(kStThreads), 0, s, D, group, par);
    else if (D.big) hipLaunchKernelGGL(expand_kernel_biglm_timed, dim3(n_workgroups), dim3(kExpandThreads), 0, s, D, group, par);
    else hipLaunchKernelGGL(expand_kernel_plain_timed, dim3(n_workgroups), dim3(kExpandThreads), 0, s, D, group, par);
    return;
  }
  if (D.big) hipLaunchKernelGGL(expand_kernel_biglm, dim3(n_workgroups), dim3(kExpandThreads), 0, s, D, group, par);
  else if (D.fused && D.staged && D.ll_row) hipLaunchKernelGGL(expand_kernel_staged_row, dim3(n_workgroups), dim3(kStThreads), 0, s, D, group, par);
  else if (D.fused && D.staged) hipLaunchKernelGGL(expand_kernel_staged, dim3(n_workgroups), dim3(kStThreads), 0, s, D, group, par);
  else hipLaunchKernelGGL(expand_kernel_plain, dim3(n_workgroups), dim3(kExpandThreads), 0, s, D, group, par);
}
void launch_insert(const DecoderDev &D, int chan_off, int chan_cnt, const int32_t *target, int boundary, int group, int par,
                   int n_workgroups, hipStream_t s) {
  const size_t lds = (size_t)D.lds_slots * ((D.big && D.lattice) ? 20 : D.big ? 16 : D.lattice ? 16 : 12);
  if (D.big && D.lattice) hipLaunchKernelGGL(insert_kernel_lattice_biglm, dim3(n_workgroups), dim3(kInsertThreads), lds, s, D, group, par);
  else if (D.big) hipLaunchKernelGGL(insert_kernel_biglm, dim3(n_workgroups), dim3(kInsertThreads), lds, s, D, group, par);
  else if (D.lattice && D.fused) hipLaunchKernelGGL(insert_kernel_lattice_fused, dim3(n_workgroups), dim3(kInsertThreads), lds, s, D, group, par);
  else if (D.lattice) hipLaunchKernelGGL(insert_kernel_lattice, dim3(n_workgroups), dim3(kInsertThreads), lds, s, D, group, par);
  else if (D.fused) hipLaunchKernelGGL(insert_kernel_fused, dim3(n_workgroups), dim3(kInsertThreads), lds, s, D, group, par, target, boundary, chan_cnt);
  else hipLaunchKernelGGL(insert_kernel_plain, dim3(n_workgroups), dim3(kInsertThreads), lds, s, D, group, par);
}
void launch_closure(const DecoderDev &D, int chan_off, int chan_cnt, const int32_t *target, int do_prep, int group, int par,
                    hipStream_t s, int after_insert) {
  // lattice decoders on the fused rows, behind an insert launch: the frame's epsilon links shared out over kClSlabs workgroups per channel
  const int ns = (after_insert && D.lattice && !D.big && D.fused && D.closure_slabs > 1) ? D.closure_slabs : 1;
  if (D.big && D.lattice)
    hipLaunchKernelGGL((closure_kernel<true, true>), dim3(chan_cnt), dim3(kBT), 0, s, D, target, do_prep, chan_off, group, par, chan_cnt, 1);
  else if (D.big)
    hipLaunchKernelGGL((closure_kernel<false, true>), dim3(chan_cnt), dim3(kBT), 0, s, D, target, do_prep, chan_off, group, par, chan_cnt, 1);
  else if (D.lattice)
    hipLaunchKernelGGL((closure_kernel<true, false>), dim3(chan_cnt * ns), dim3(kBT), 0, s, D, target, do_prep, chan_off, group, par, chan_cnt, ns);
  else
    hipLaunchKernelGGL((closure_kernel<false, false>), dim3(chan_cnt), dim3(kBT), 0, s, D, target, do_prep, chan_off, group, par, chan_cnt, 1);
}
void launch_lattice_prune_step(const DecoderDev &D, int chan_off, int chan_cnt, const int32_t *target, int group, int par, hipStream_t s, int stage) {
  // the raw frames (kPrRawJ workgroups per channel), the walk over the frames priced before (one workgroup per
  // channel, 130 KB of LDS), the compaction's flag sweeps (kPrSlabs workgroups per channel), its moves + the next frame's preparation
  const int raw = D.prune_raw ? 1 : 0;
  auto on = [&](int k) { return stage < 0 || stage == k; };
  if (raw && on(0)) hipLaunchKernelGGL(lattice_prune_raw_kernel, dim3(kPrRawJ * chan_cnt), dim3(kPrRawT), 0, s, D, target, chan_off, chan_cnt);
  if (on(1)) {
    if (D.big) hipLaunchKernelGGL(lattice_prune_kernel<true>, dim3(chan_cnt), dim3(kBT), 0, s, D, target, chan_off, group, par, raw);
    else hipLaunchKernelGGL(lattice_prune_kernel<false>, dim3(chan_cnt), dim3(kBT), 0, s, D, target, chan_off, group, par, raw);
  }
  if (on(2)) hipLaunchKernelGGL(lattice_prune_flags_kernel, dim3(chan_cnt * kPrSlabs), dim3(kBT), 0, s, D, chan_off);
  if (on(3)) {
    if (D.big) hipLaunchKernelGGL(lattice_prune_move_kernel<true>, dim3(chan_cnt, 2), dim3(kBT), 0, s, D, target, chan_off, group, par);
    else hipLaunchKernelGGL(lattice_prune_move_kernel<false>, dim3(chan_cnt, 2), dim3(kBT), 0, s, D, target, chan_off, group, par);
  }
}
// A pause of `us` microseconds on a stream (one wave that sleeps): staggers the channel groups' frame chains against each other
// (wfst_capi.cc advance_device).
static __global__ void delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
void launch_delay(int us, hipStream_t s) {
  if (us > 0) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, (unsigned long long)us * 100ull);   // (the constant 100 MHz clock)
}
void launch_set_finalized(const DecoderDev &D, const int32_t *chans, int n, hipStream_t s) {
  hipLaunchKernelGGL(set_finalized_kernel, dim3((n + 255) / 256), dim3(256), 0, s, D, chans, n);
}
void launch_best_path(const DecoderDev &D, const int32_t *chans, int n, int use_final, int cap, int32_t *ilabel,
                      int32_t *olabel, float *graph, float *ac, int32_t *n_hops, int32_t *chain, hipStream_t s) {
  if (D.big)
    hipLaunchKernelGGL(best_path_kernel<true>, dim3(n), dim3(kBpThreads), 0, s, D, chans, use_final, cap, ilabel, olabel, graph,
                       ac, n_hops, chain);
  else
    hipLaunchKernelGGL(best_path_kernel<false>, dim3(n), dim3(kBpThreads), 0, s, D, chans, use_final, cap, ilabel, olabel, graph,
                       ac, n_hops, chain);
}
void launch_lattice_emit(const DecoderDev &D, const int32_t *chans, int n, int use_final, hipStream_t s);
void launch_lattice_prune(const DecoderDev &D, const int32_t *chans, int n, hipStream_t s) {
  hipLaunchKernelGGL(lattice_finalize_kernel, dim3(n), dim3(kBT), 0, s, D, chans);
  launch_lattice_emit(D, chans, n, 1, s);
}
void launch_lattice_emit(const DecoderDev &D, const int32_t *chans, int n, int use_final, hipStream_t s) {
  hipLaunchKernelGGL(lattice_emit_reset_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, s, D, chans, n);
  hipLaunchKernelGGL(lattice_emit_kernel, dim3(n, kEmitSlabs), dim3(kBT), 0, s, D, chans, use_final);
  hipLaunchKernelGGL(lattice_emit_tokens_kernel, dim3(n, kEmitSlabs), dim3(kBT), 0, s, D, chans, use_final);
}
int insert_kernel_set_lds(int bytes) {
  int e = (int)hipFuncSetAttribute((const void *)insert_kernel_lattice, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e) return e;
  e = (int)hipFuncSetAttribute((const void *)insert_kernel_biglm, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e) return e;
  e = (int)hipFuncSetAttribute((const void *)insert_kernel_lattice_biglm, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e) return e;
  e = (int)hipFuncSetAttribute((const void *)insert_kernel_lattice_fused, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e) return e;
  e = (int)hipFuncSetAttribute((const void *)insert_kernel_fused, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e) return e;
  return (int)hipFuncSetAttribute((const void *)insert_kernel_plain, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
}  // namespace wfst
