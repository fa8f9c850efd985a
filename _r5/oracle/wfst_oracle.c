/*
 * TEST INFRASTRUCTURE -- CPU oracle for the WFST token-passing hot path.
 *
 * A plain-C restatement of the reference algorithm (datemoon/ASR-decoder,
 * class OnlineLatticeDecoderMempool).  It is the checker for the HIP path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product never calls it and has no CPU fallback.
 *
 * Parity status: PINNED.  The reference has no golden vectors for this path
 * (SURVEY.md section 4), so the oracle is pinned against the reference decoder
 * itself, compiled unmodified into oracle/_ref/libref_decoder.so
 * (oracle/Makefile, oracle/ref_driver.cc): tests/test_oracle_vs_reference.py
 * requires bit-identical words, transition-ids, per-hop costs and scores, and
 * tests/golden/ holds reference-generated vectors (tests/golden/make_golden.py)
 * that this file must reproduce where the reference tree is absent.
 *
 * biglm (BASELINE configs[3], class OnlineLatticeDecoderMempoolBiglm,
 * my-decoder/online-decoder-mempool-base-biglm.h + newlm/): the same loop over 64-bit
 * (graph state | LM pair state << 32) keys with the on-the-fly LM difference, in two modes:
 *   as-written  DiffArpaLm::GetArc hands the PAIR id to both LMs (newlm/diff-lm.h:80,86) and interns
 *               pair ids in visiting order -- pinned bit for bit to the compiled reference
 *               (tests/test_oracle_biglm.py, oracle/_ref);
 *   fixed       the pair's own components (pr.first / pr.second) -- the two-line change the reference
 *               evidently means; equal to as-written wherever the LM scores do not depend on the
 *               history (proved on unigram LM pairs), and the mode the HIP path implements.
 *
 * Every function cites the reference lines it follows; paths are relative to
 * /root/reference/src.  Float arithmetic is single precision, left to right,
 * no contraction (reference: -O2 -msse2, configure.ac:12-13).
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <time.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FLOAT_INF (1.0f / 0.0f)

/* ---- graph: newfst/arc.h:17-26, newfst/optimize-fst.h:13-48,226-280 ---- */
typedef struct { int ilabel, olabel; float w; int to; } Arc;
typedef struct { unsigned num_arcs, niepsilons, noepsilons; } StateInfo;
typedef struct {
  int start, final_state, n_states, n_arcs;
  StateInfo *si;
  int64_t *off; /* arc offset of each state (prefix sum of num_arcs) */
  Arc *arcs;
} Graph;

/* ---- config: my-decoder/lattice-faster-decoder-conf.h:21-44 ---- */
typedef struct {
  float beam; int max_active; int min_active; float lattice_beam;
  int prune_interval; float beam_delta; float hash_ratio; float prune_scale;
} Config;

/* ---- tokens and links: my-decoder/online-decoder-base.h:28-84 ---- */
struct Token;
typedef struct Link {
  struct Token *next_tok; int ilabel, olabel; float graph_cost, acoustic_cost;
  struct Link *next;
} Link;
typedef struct Token {
  float tot_cost, extra_cost; Link *links; struct Token *next; struct Token *backpointer;
  float extra_before; /* order-free pruning: extra_cost when PruneForwardLinks was entered */
  int is_final; /* stands for membership in _final_costs */
  float final_cost; /* its value there: 0, or the LM final cost in biglm mode */
  int state;    /* graph state (the reference token does not know it; used to label lattice states) */
  int lm_state; /* biglm: LM pair state of the token's key */
  int lat_id;   /* lattice state id during GetRawLattice */
  int tie;      /* audit only: an equal-cost rival arrived after this cost was set */
} Token;
typedef struct { Token *toks; int must_prune_forward_links, must_prune_tokens; } TokenList;

/* ---- HashList<StateId, Token*>: util/hash-list.h:13-105, hash-list-inl.h:15-173 ---- */
typedef uint64_t Key; /* StateId, or biglm's PairId = state + (lm_state << 32) (biglm.h:77-90) */
typedef struct Elem { Key key; Token *val; struct Elem *tail; } Elem;
typedef struct { size_t prev_bucket; Elem *last_elem; } Bucket;
#define NOBUCKET ((size_t)-1)
typedef struct {
  Elem *list_head; size_t bucket_list_tail; size_t hash_size;
  Bucket *buckets; size_t n_buckets; Elem *freed_head;
  Elem **blocks; size_t n_blocks, cap_blocks;
} HashList;

/* simple block pools standing in for MemPool<T> (util/mem-pool.h:17-65); allocation
 * order has no effect on results */
typedef struct PoolBlock { struct PoolBlock *next; } PoolBlock;
typedef struct { void *free_head; PoolBlock *blocks; size_t elem_size; } Pool;

static int g_order_free = 0; /* see process_emitting and prune_forward_links */

/* ---- LM automaton: newlm/arpa2fsa.h:22-247; binary file ArpaLm::Read :355-397 + Fsa::Read arpa2fsa.cc:68-176 ---- */
typedef struct { int wordid; float weight; int tostateid; } FsaArc;
typedef struct { int arc_num; float backoff_prob; int backoff_id; } FsaStateInfo;
typedef struct {
  int bos, eos, unk, order;
  int n_states, n_arcs;
  FsaStateInfo *st; int64_t *off; FsaArc *arcs;
} Lm;
/* DiffArpaLm: newlm/diff-lm.h:13-122 */
typedef struct { int a, b, id; } PairSlot;
typedef struct {
  const Lm *lm1, *lm2;
  int (*vec)[2]; int n_vec, cap_vec;      /* _state_vec */
  PairSlot *map; int map_cap, map_n;       /* _state_map (open addressing; only membership matters) */
  int start_pair[2];
  int fixed;                               /* 0: as written (pair id handed to both LMs), 1: pr.first / pr.second */
  int oob;                                 /* an LM state / word id outside the automaton was asked for (undefined in the reference) */
} DiffLm;

typedef struct {
  const Graph *g; Config cfg; HashList toks; DiffLm *dlm;
  TokenList *active; int n_active, cap_active;
  const Elem **queue; size_t n_queue, cap_queue;
  float *tmp; size_t n_tmp, cap_tmp;
  Pool tok_pool, link_pool;
  int num_toks, num_links, warned, finalized, any_final;
  float final_relative_cost, final_best_cost;
  int num_frames_decoded;
  /* decodable (DecodableMatrixScaledMapped with the scale pre-applied) */
  const float *ll; int T, stride; const int *tid2pdf; int frames_ready;
  /* work counters for the roofline's algorithmic bytes */
  int64_t cnt_N, cnt_E, cnt_Z, cnt_tok_created, cnt_link_created;
  int64_t cnt_L; /* biglm: arcs with an output label traversed (LM look-ups) */
  int64_t cnt_Leps; /* ... of them on epsilon arcs (ProcessNonemitting) */
} Decoder;

static void *pool_new(Pool *p) {
  if (!p->free_head) {
    size_t n = 1024, i;
    char *blk = (char *)malloc(sizeof(PoolBlock) + n * p->elem_size);
    ((PoolBlock *)blk)->next = p->blocks; p->blocks = (PoolBlock *)blk;
    char *base = blk + sizeof(PoolBlock);
    for (i = 0; i < n; ++i) { *(void **)(base + i * p->elem_size) = p->free_head; p->free_head = base + i * p->elem_size; }
  }
  void *r = p->free_head; p->free_head = *(void **)r; return r;
}
static void pool_del(Pool *p, void *e) { *(void **)e = p->free_head; p->free_head = e; }
static void pool_destroy(Pool *p) { while (p->blocks) { PoolBlock *n = p->blocks->next; free(p->blocks); p->blocks = n; } p->free_head = NULL; }

/* ---- HashList ---- */
static void hl_set_size(HashList *h, size_t size) { /* hash-list-inl.h:15-23 */
  h->hash_size = size;
  if (size > h->n_buckets) {
    h->buckets = (Bucket *)realloc(h->buckets, size * sizeof(Bucket));
    for (size_t i = h->n_buckets; i < size; ++i) { h->buckets[i].prev_bucket = 0; h->buckets[i].last_elem = NULL; }
    h->n_buckets = size;
  }
}
static Elem *hl_clear(HashList *h) { /* hash-list-inl.h:25-38 */
  for (size_t b = h->bucket_list_tail; b != NOBUCKET; b = h->buckets[b].prev_bucket) h->buckets[b].last_elem = NULL;
  h->bucket_list_tail = NOBUCKET;
  Elem *ans = h->list_head; h->list_head = NULL; return ans;
}
static void hl_delete(HashList *h, Elem *e) { e->tail = h->freed_head; h->freed_head = e; } /* :46-51 */
static void hl_delete_elems(HashList *h) { /* :53-61 */
  for (Elem *e = hl_clear(h), *t; e; e = t) { t = e->tail; hl_delete(h, e); }
}
static Elem *hl_new(HashList *h) { /* :86-104 */
  if (!h->freed_head) {
    size_t n = 1024;
    Elem *tmp = (Elem *)malloc(n * sizeof(Elem));
    for (size_t i = 0; i + 1 < n; ++i) tmp[i].tail = tmp + i + 1;
    tmp[n - 1].tail = NULL; h->freed_head = tmp;
    if (h->n_blocks == h->cap_blocks) { h->cap_blocks = h->cap_blocks ? 2 * h->cap_blocks : 16; h->blocks = (Elem **)realloc(h->blocks, h->cap_blocks * sizeof(Elem *)); }
    h->blocks[h->n_blocks++] = tmp;
  }
  Elem *a = h->freed_head; h->freed_head = a->tail; return a;
}
static Elem *hl_insert(HashList *h, Key key, Token *val) { /* hash-list-inl.h:128-173 */
  size_t index = (size_t)key % h->hash_size;
  Bucket *b = &h->buckets[index];
  if (b->last_elem) {
    Elem *head = (b->prev_bucket == NOBUCKET) ? h->list_head : h->buckets[b->prev_bucket].last_elem->tail;
    Elem *tail = b->last_elem->tail;
    for (Elem *e = head; e != tail; e = e->tail) if (e->key == key) return e;
  }
  Elem *elem = hl_new(h); elem->key = key; elem->val = val;
  if (!b->last_elem) {
    if (h->bucket_list_tail == NOBUCKET) h->list_head = elem;
    else h->buckets[h->bucket_list_tail].last_elem->tail = elem;
    elem->tail = NULL; b->last_elem = elem; b->prev_bucket = h->bucket_list_tail; h->bucket_list_tail = index;
  } else {
    elem->tail = b->last_elem->tail; b->last_elem->tail = elem; b->last_elem = elem;
  }
  return elem;
}

/* ---- LM automaton ---- */
/* Fsa::GetArc, newlm/arpa2fsa.cc:244-262 (+ FsaState::SearchArc / SearchStartArc, arpa2fsa.h:194-214).
 * The reference indexes without bounds checks; *oob is raised instead of reading outside. */
static int fsa_getarc(const Lm *lm, int id, int wordid, float *weight, int *tostateid, int *oob) {
  if (id < 0 || id >= lm->n_states) { *oob = 1; *weight = 0.0f; *tostateid = 0; return 1; }
  if (wordid == 0) { *weight = lm->st[id].backoff_prob; *tostateid = lm->st[id].backoff_id; return 1; }
  const FsaArc *arc = lm->arcs + lm->off[id], *hit = NULL;
  const int arc_num = lm->st[id].arc_num;
  if (id == 0) {                 /* id == start: arc[wordid] */
    if (wordid < 0 || wordid >= arc_num) { *oob = 1; *weight = 0.0f; *tostateid = 0; return 1; }
    hit = &arc[wordid];
  } else {
    int start = 0, end = arc_num - 1, mid = (start + end) / 2;
    while (start <= end) {
      if (arc[mid].wordid > wordid) end = mid - 1;
      else if (arc[mid].wordid < wordid) start = mid + 1;
      else { hit = &arc[mid]; break; }
      mid = (start + end) / 2;
    }
  }
  if (!hit) return 0;
  *weight = hit->weight; *tostateid = hit->tostateid; return 1;
}
/* ComposeArpaLm::GetArc, newlm/compose-arpalm.cc:52-70: back off until the word is found */
static void compose_getarc(const Lm *lm, int s, int ilabel, int *nextstate, float *value1, int *oob) {
  float weight = 0.0f, w_arc = 0.0f; int to = 0, guard = 0;
  while (!fsa_getarc(lm, s, ilabel, &w_arc, &to, oob)) {
    fsa_getarc(lm, s, 0, &w_arc, &to, oob);
    s = to; weight += w_arc;
    if (++guard > 64) { *oob = 1; break; }
  }
  weight += w_arc;
  *value1 = -1 * weight;
  *nextstate = to;
}
static int compose_start(const Lm *lm, int *oob) { /* compose-arpalm.cc:5-13 */
  float w = 0.0f; int to = 0;
  fsa_getarc(lm, 0, lm->bos, &w, &to, oob);
  return to;
}
static float compose_final(const Lm *lm, int s, int *oob) { /* compose-arpalm.cc:15-29 */
  float weight = 0.0f, w_arc = 0.0f; int to = 0, guard = 0;
  while (!fsa_getarc(lm, s, lm->eos, &w_arc, &to, oob)) {
    fsa_getarc(lm, s, 0, &w_arc, &to, oob);
    s = to; weight += w_arc;
    if (++guard > 64) { *oob = 1; break; }
  }
  weight += w_arc;
  return (float)(-1.0 * weight);
}

void *oracle_lm_load(const char *path, float scale) { /* ArpaLm::Read + Fsa::Read + Rescale */
  FILE *fp = fopen(path, "rb");
  if (!fp) return NULL;
  Lm *lm = (Lm *)calloc(1, sizeof(Lm));
  uint64_t ngram = 0; int ok = 1;
  ok = ok && fread(&lm->bos, 4, 1, fp) == 1 && fread(&lm->eos, 4, 1, fp) == 1 && fread(&lm->unk, 4, 1, fp) == 1 && fread(&ngram, 8, 1, fp) == 1;
  if (ok && ngram > 64) ok = 0;
  if (ok) { int tmp[64]; lm->order = (int)ngram; ok = fread(tmp, 4, (size_t)ngram, fp) == (size_t)ngram; }
  ok = ok && fread(&lm->n_states, 4, 1, fp) == 1 && lm->n_states > 0;
  if (ok) {
    lm->st = (FsaStateInfo *)malloc(sizeof(FsaStateInfo) * (size_t)lm->n_states);
    lm->off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)lm->n_states + 1));
    ok = fread(lm->st, sizeof(FsaStateInfo), (size_t)lm->n_states, fp) == (size_t)lm->n_states;
  }
  if (ok) {
    int64_t o = 0;
    for (int i = 0; i < lm->n_states; ++i) { lm->off[i] = o; if (lm->st[i].arc_num < 0) ok = 0; o += lm->st[i].arc_num; }
    lm->off[lm->n_states] = o;
    ok = ok && fread(&lm->n_arcs, 4, 1, fp) == 1 && (int64_t)lm->n_arcs == o;
  }
  if (ok) {
    lm->arcs = (FsaArc *)malloc(sizeof(FsaArc) * ((size_t)lm->n_arcs + 1));
    ok = fread(lm->arcs, sizeof(FsaArc), (size_t)lm->n_arcs, fp) == (size_t)lm->n_arcs;
  }
  fclose(fp);
  if (!ok) { free(lm->st); free(lm->off); free(lm->arcs); free(lm); return NULL; }
  if (scale != 1) { /* Fsa::Rescale, arpa2fsa.cc:264-275 */
    for (int i = 0; i < lm->n_arcs; ++i) lm->arcs[i].weight *= scale;
    for (int i = 0; i < lm->n_states; ++i) lm->st[i].backoff_prob *= scale;
  }
  return lm;
}
void oracle_lm_free(void *p) { Lm *lm = (Lm *)p; if (!lm) return; free(lm->st); free(lm->off); free(lm->arcs); free(lm); }
int oracle_lm_start(void *lm) { int oob = 0; return compose_start((const Lm *)lm, &oob); }
float oracle_lm_final(void *lm, int s) { int oob = 0; return compose_final((const Lm *)lm, s, &oob); }
void oracle_lm_getarc_many(void *lm, int n, const int *states, const int *words, int *next, float *value1) {
  int oob = 0;
  for (int i = 0; i < n; ++i) compose_getarc((const Lm *)lm, states[i], words[i], &next[i], &value1[i], &oob);
}

/* ---- DiffArpaLm: newlm/diff-lm.h:13-122 ---- */
static int dlm_intern(DiffLm *d, int a, int b) { /* _state_map.insert + _state_vec.push_back, :92-103 */
  if (2 * (d->map_n + 1) > d->map_cap) {
    int ncap = d->map_cap ? 2 * d->map_cap : 1024;
    PairSlot *nm = (PairSlot *)malloc(sizeof(PairSlot) * (size_t)ncap);
    for (int i = 0; i < ncap; ++i) nm[i].id = -1;
    for (int i = 0; i < d->map_cap; ++i) if (d->map[i].id >= 0) {
      uint32_t h = ((uint32_t)d->map[i].a * 7853u + (uint32_t)d->map[i].b) * 2654435761u;
      int j = (int)(h & (uint32_t)(ncap - 1));
      while (nm[j].id >= 0) j = (j + 1) & (ncap - 1);
      nm[j] = d->map[i];
    }
    free(d->map); d->map = nm; d->map_cap = ncap;
  }
  uint32_t h = ((uint32_t)a * 7853u + (uint32_t)b) * 2654435761u;
  int j = (int)(h & (uint32_t)(d->map_cap - 1));
  while (d->map[j].id >= 0) {
    if (d->map[j].a == a && d->map[j].b == b) return d->map[j].id;
    j = (j + 1) & (d->map_cap - 1);
  }
  if (d->n_vec == d->cap_vec) { d->cap_vec = d->cap_vec ? 2 * d->cap_vec : 1024; d->vec = (int (*)[2])realloc(d->vec, sizeof(int[2]) * (size_t)d->cap_vec); }
  d->map[j].a = a; d->map[j].b = b; d->map[j].id = d->n_vec; d->map_n++;
  d->vec[d->n_vec][0] = a; d->vec[d->n_vec][1] = b;
  return d->n_vec++;
}
static void dlm_reset(DiffLm *d) { /* :37-44 */
  d->n_vec = 0; d->map_n = 0;
  for (int i = 0; i < d->map_cap; ++i) d->map[i].id = -1;
  dlm_intern(d, d->start_pair[0], d->start_pair[1]); /* id 0 == _start_state */
}
static void dlm_init(DiffLm *d, const Lm *lm1, const Lm *lm2, int fixed) { /* :19-35 */
  memset(d, 0, sizeof(*d));
  d->lm1 = lm1; d->lm2 = lm2; d->fixed = fixed;
  d->start_pair[0] = compose_start(lm1, &d->oob); d->start_pair[1] = compose_start(lm2, &d->oob);
  dlm_reset(d);
}
static void dlm_free(DiffLm *d) { free(d->vec); free(d->map); }
static float dlm_final(DiffLm *d, int s) { /* :48-53 */
  if (s < 0 || s >= d->n_vec) { d->oob = 1; return 0.0f; }
  return compose_final(d->lm1, d->vec[s][0], &d->oob) + compose_final(d->lm2, d->vec[s][1], &d->oob);
}
/* OnlineLatticeDecoderMempoolBaseBiglm::NextLmState (biglm.h:54-70) over DiffArpaLm::GetArc (diff-lm.h:63-111) */
static int next_lm_state(DiffLm *d, int lm_state, int olabel, float *lm_score) {
  if (olabel == 0) { *lm_score = 0; return lm_state; }
  if (lm_state < 0 || lm_state >= d->n_vec) { d->oob = 1; *lm_score = 0; return lm_state; }
  /* as written: `_lm1.GetArc(s, ...)`, `_lm2.GetArc(s, ...)` with s the PAIR id (diff-lm.h:80,86);
   * fixed: the pair's own components */
  const int s1 = d->fixed ? d->vec[lm_state][0] : lm_state, s2 = d->fixed ? d->vec[lm_state][1] : lm_state;
  int n1, n2; float w1, w2;
  compose_getarc(d->lm1, s1, olabel, &n1, &w1, &d->oob);
  compose_getarc(d->lm2, s2, olabel, &n2, &w2, &d->oob);
  const int id = dlm_intern(d, n1, n2);
  *lm_score = w1 + w2; /* Times(w1, w2).Value1(), newfst/weigth.h:320 */
  return id;
}
#define KEY_STATE(k) ((int)(uint32_t)(k))             /* PairToState,   biglm.h:82-85 */
#define KEY_LM(k) ((int)(uint32_t)((k) >> 32))        /* PairToLmState, biglm.h:87-90 */
static inline Key make_key(int state, int lm_state) { return (Key)(uint32_t)state + ((Key)(uint32_t)lm_state << 32); } /* ConstructPair :77-80 */

/* ---- token / link allocation: online-decoder-mempool-base.h:33-74 ---- */
static Token *new_token(Decoder *d, float tot, float extra, Link *links, Token *next, Token *bp) {
  Token *t = (Token *)pool_new(&d->tok_pool);
  t->tot_cost = tot; t->extra_cost = extra; t->links = links; t->next = next; t->backpointer = bp; t->is_final = 0; t->tie = 0;
  t->final_cost = 0.0f; t->state = -1; t->lm_state = 0; t->lat_id = -1;
  d->num_toks++; d->cnt_tok_created++; return t;
}
static Link *new_link(Decoder *d, Token *nt, int il, int ol, float gc, float ac, Link *next) {
  Link *l = (Link *)pool_new(&d->link_pool);
  l->next_tok = nt; l->ilabel = il; l->olabel = ol; l->graph_cost = gc; l->acoustic_cost = ac; l->next = next;
  d->num_links++; d->cnt_link_created++; return l;
}
static void delete_token(Decoder *d, Token *t) { pool_del(&d->tok_pool, t); d->num_toks--; }
static void delete_link(Decoder *d, Link *l) { pool_del(&d->link_pool, l); d->num_links--; }
static void delete_forward_links(Decoder *d, Token *tok) { /* base-inl.h:8-19 */
  Link *l = tok->links, *m;
  while (l) { m = l->next; delete_link(d, l); l = m; }
  tok->links = NULL;
}

static inline float loglike(const Decoder *d, int frame, int index) {
  int col = d->tid2pdf ? d->tid2pdf[index] : index;
  return d->ll[(size_t)frame * d->stride + col];
}

static void active_resize(Decoder *d, int n) {
  if (n > d->cap_active) { d->cap_active = n * 2 + 16; d->active = (TokenList *)realloc(d->active, d->cap_active * sizeof(TokenList)); }
  for (int i = d->n_active; i < n; ++i) { d->active[i].toks = NULL; d->active[i].must_prune_forward_links = 1; d->active[i].must_prune_tokens = 1; }
  d->n_active = n;
}

static void clear_active_tokens(Decoder *d) { /* base-inl.h:69-85 */
  for (int i = 0; i < d->n_active; ++i)
    for (Token *tok = d->active[i].toks; tok;) { delete_forward_links(d, tok); Token *n = tok->next; delete_token(d, tok); tok = n; }
  d->n_active = 0;
}

/* FindOrAddToken: base-inl.h:88-136 */
static Elem *find_or_add_token(Decoder *d, Key key, int frame_plus_one, float tot_cost, Token *bp, int *changed) {
  Token **toks = &d->active[frame_plus_one].toks;
  Elem *e = hl_insert(&d->toks, key, NULL);
  if (e->val == NULL) {
    Token *nt = new_token(d, tot_cost, 0.0f, NULL, *toks, bp);
    nt->state = KEY_STATE(key); nt->lm_state = KEY_LM(key);
    *toks = nt; e->val = nt;
    if (changed) *changed = 1;
  } else {
    Token *tok = e->val;
    if (tok->tot_cost > tot_cost) { tok->tot_cost = tot_cost; tok->backpointer = bp; tok->tie = 0; if (changed) *changed = 1; }
    else {
      /* audit only: an equal cost through ANOTHER predecessor is a real tie (first arrival wins here, lowest
       * arc index on the GPU); the same predecessor arriving again is the closure re-processing a token */
      if (tok->tot_cost == tot_cost && tok->backpointer != bp) tok->tie = 1;
      if (changed) *changed = 0;
    }
  }
  return e;
}

/* k-th smallest (0-based) of a[0,n): the only observable of std::nth_element */
static float kth_smallest(float *a, int64_t n, int64_t k) {
  int64_t lo = 0, hi = n - 1;
  while (lo < hi) {
    float p = a[lo + (hi - lo) / 2];
    int64_t i = lo, j = hi;
    while (i <= j) {
      while (a[i] < p) ++i;
      while (a[j] > p) --j;
      if (i <= j) { float t = a[i]; a[i] = a[j]; a[j] = t; ++i; --j; }
    }
    if (k <= j) hi = j;
    else if (k >= i) lo = i;
    else return a[k];
  }
  return a[k];
}

/* GetCutoff: base-inl.h:138-234 */
static float get_cutoff(Decoder *d, Elem *list_head, size_t *tok_count, float *adaptive_beam, Elem **best_elem) {
  float best_weight = FLOAT_INF;
  size_t count = 0;
  const Config *c = &d->cfg;
  if (c->max_active == 2147483647 && c->min_active == 0) {
    for (Elem *e = list_head; e; e = e->tail, ++count) {
      float w = e->val->tot_cost;
      if (w < best_weight) { best_weight = w; if (best_elem) *best_elem = e; }
    }
    if (tok_count) *tok_count = count;
    if (adaptive_beam) *adaptive_beam = c->beam;
    return best_weight + c->beam;
  }
  d->n_tmp = 0;
  for (Elem *e = list_head; e; e = e->tail, ++count) {
    float w = e->val->tot_cost;
    if (d->n_tmp == d->cap_tmp) { d->cap_tmp = d->cap_tmp ? 2 * d->cap_tmp : 4096; d->tmp = (float *)realloc(d->tmp, d->cap_tmp * sizeof(float)); }
    d->tmp[d->n_tmp++] = w;
    if (w < best_weight) { best_weight = w; if (best_elem) *best_elem = e; }
  }
  if (tok_count) *tok_count = count;
  float beam_cutoff = best_weight + c->beam;
  float min_active_cutoff = FLOAT_INF, max_active_cutoff = FLOAT_INF;
  if (d->n_tmp > (size_t)c->max_active) max_active_cutoff = kth_smallest(d->tmp, (int64_t)d->n_tmp, (int64_t)c->max_active);
  if (max_active_cutoff < beam_cutoff) {
    if (adaptive_beam) *adaptive_beam = max_active_cutoff - best_weight + c->beam_delta;
    return max_active_cutoff;
  }
  if (d->n_tmp > (size_t)c->min_active) {
    if (c->min_active == 0) min_active_cutoff = best_weight;
    else /* the k-th smallest of the whole array equals the reference's nth_element over its
            (already partitioned) prefix [0, max_active) */
      min_active_cutoff = kth_smallest(d->tmp, (int64_t)d->n_tmp, (int64_t)c->min_active);
  }
  if (min_active_cutoff > beam_cutoff) {
    if (adaptive_beam) *adaptive_beam = min_active_cutoff - best_weight + c->beam_delta;
    return min_active_cutoff;
  }
  if (adaptive_beam) *adaptive_beam = c->beam;
  return beam_cutoff;
}

static void possibly_resize_hash(Decoder *d, size_t num_toks) { /* base-inl.h:236-244 */
  size_t new_sz = (size_t)((float)num_toks * d->cfg.hash_ratio);
  if (new_sz > d->toks.hash_size) hl_set_size(&d->toks, new_sz);
}

static void queue_push(Decoder *d, const Elem *e) {
  if (d->n_queue == d->cap_queue) { d->cap_queue = d->cap_queue ? 2 * d->cap_queue : 1024; d->queue = (const Elem **)realloc(d->queue, d->cap_queue * sizeof(Elem *)); }
  d->queue[d->n_queue++] = e;
}

/* ProcessNonemitting: base-inl.h:353-431 */
static void process_nonemitting(Decoder *d, float cutoff) {
  const Graph *g = d->g;
  int frame = d->n_active - 1;
  if (d->toks.list_head == NULL && !d->warned) d->warned = 1;
  for (const Elem *e = d->toks.list_head; e; e = e->tail)
    if (g->si[KEY_STATE(e->key)].niepsilons != 0) queue_push(d, e);
  while (d->n_queue) {
    const Elem *elem = d->queue[--d->n_queue];
    int state = KEY_STATE(elem->key), lm_state = KEY_LM(elem->key); Token *tok = elem->val;
    float cur_cost = tok->tot_cost;
    if (cur_cost >= cutoff) continue;
    delete_forward_links(d, tok);
    const Arc *arcs = g->arcs + g->off[state];
    unsigned n = g->si[state].num_arcs;
    for (unsigned i = 0; i < n; ++i) {
      const Arc *arc = &arcs[i];
      if (arc->ilabel == 0) {
        d->cnt_Z++;
        float graph_cost = arc->w;
        int next_lm = 0;
        if (d->dlm) { /* biglm.h:448-451 */
          float lm_score;
          if (arc->olabel != 0) { d->cnt_L++; d->cnt_Leps++; }
          next_lm = next_lm_state(d->dlm, lm_state, arc->olabel, &lm_score);
          graph_cost = arc->w + lm_score;
        }
        float tot_cost = cur_cost + graph_cost;
        if (tot_cost < cutoff) {
          int changed = 0;
          Elem *nt = find_or_add_token(d, make_key(arc->to, next_lm), frame, tot_cost, tok, &changed);
          tok->links = new_link(d, nt->val, 0, arc->olabel, graph_cost, 0, tok->links);
          if (changed && g->si[arc->to].niepsilons != 0) queue_push(d, nt);
        }
      }
    }
  }
}

/* ProcessEmitting: base-inl.h:246-351 */
/* "Order-free" variant used ONLY to state what the GPU path computes (tests/test_oracle_lattice.py,
 * tests/test_gpu_lattice.py).  The reference admits an arc when its cost is below the next_cutoff AS
 * IT STANDS when the arc is reached (base-inl.h:326-333), so arcs that are above the frame's FINAL
 * next_cutoff get in or not depending on the hash-list order.  With the flag set the final
 * next_cutoff is computed first and applied to every arc: the result is the subset of the reference's
 * tokens/links that does not depend on the visiting order.  Default 0 = the reference's behaviour. */
void oracle_set_order_free(int on) { g_order_free = on; }

static float process_emitting(Decoder *d) {
  const Graph *g = d->g;
  int nnetframe = d->num_frames_decoded;
  int frame = d->n_active - 1;
  active_resize(d, d->n_active + 1);
  Elem *final_toks = hl_clear(&d->toks);
  Elem *best_elem = NULL; float adaptive_beam; size_t tok_cnt = 0;
  float cur_cutoff = get_cutoff(d, final_toks, &tok_cnt, &adaptive_beam, &best_elem);
  /* biglm: PossiblyResizeHash (biglm.h:335) is the BASE class's and grows the base class's own, unused
   * `_toks`; the 64-bit-keyed list that holds the tokens keeps its constructor size (biglm.h:28,73) */
  if (!d->dlm) possibly_resize_hash(d, tok_cnt);
  float next_cutoff = FLOAT_INF;
  if (best_elem) {
    int state = KEY_STATE(best_elem->key), lm_state = KEY_LM(best_elem->key); Token *tok = best_elem->val;
    const Arc *arcs = g->arcs + g->off[state]; unsigned n = g->si[state].num_arcs;
    for (unsigned i = 0; i < n; ++i) {
      const Arc *arc = &arcs[i];
      if (arc->ilabel != 0) {
        float tot_score;
        if (d->dlm) { /* biglm.h:350-353 */
          float lm_score;
          next_lm_state(d->dlm, lm_state, arc->olabel, &lm_score);
          tot_score = lm_score + tok->tot_cost + arc->w - loglike(d, nnetframe, arc->ilabel);
        } else tot_score = tok->tot_cost + arc->w - loglike(d, nnetframe, arc->ilabel);
        if (tot_score + adaptive_beam < next_cutoff) next_cutoff = tot_score + adaptive_beam;
      }
    }
  }
  if (g_order_free) {
    for (Elem *e = final_toks; e; e = e->tail) {
      int state = KEY_STATE(e->key), lm_state = KEY_LM(e->key); Token *tok = e->val;
      if (!(tok->tot_cost <= cur_cutoff)) continue;
      const Arc *arcs = g->arcs + g->off[state]; unsigned n = g->si[state].num_arcs;
      for (unsigned i = 0; i < n; ++i)
        if (arcs[i].ilabel != 0) {
          float graph_cost = arcs[i].w;
          if (d->dlm) { float lm_score; next_lm_state(d->dlm, lm_state, arcs[i].olabel, &lm_score); graph_cost = arcs[i].w + lm_score; }
          float tot_cost = tok->tot_cost + -loglike(d, nnetframe, arcs[i].ilabel) + graph_cost;
          if (tot_cost + adaptive_beam < next_cutoff) next_cutoff = tot_cost + adaptive_beam;
        }
    }
  }
  for (Elem *e = final_toks, *e_tail; e; e = e_tail) {
    int state = KEY_STATE(e->key), lm_state = KEY_LM(e->key); Token *tok = e->val;
    if (tok->tot_cost <= cur_cutoff) {
      d->cnt_N++;
      const Arc *arcs = g->arcs + g->off[state]; unsigned n = g->si[state].num_arcs;
      for (unsigned i = 0; i < n; ++i) {
        const Arc *arc = &arcs[i];
        if (arc->ilabel != 0) {
          d->cnt_E++;
          float graph_cost = arc->w;
          int next_lm = 0;
          if (d->dlm) { /* biglm.h:377-380: the LM is asked before the acoustic score */
            float lm_score;
            if (arc->olabel != 0) d->cnt_L++;
            next_lm = next_lm_state(d->dlm, lm_state, arc->olabel, &lm_score);
            graph_cost = arc->w + lm_score;
          }
          float ac_cost = -loglike(d, nnetframe, arc->ilabel);
          float cur_cost = tok->tot_cost;
          float tot_cost = cur_cost + ac_cost + graph_cost;
          if (tot_cost >= next_cutoff) continue;
          else if (tot_cost + adaptive_beam < next_cutoff) next_cutoff = tot_cost + adaptive_beam;
          Elem *nt = find_or_add_token(d, make_key(arc->to, next_lm), frame + 1, tot_cost, tok, NULL);
          tok->links = new_link(d, nt->val, arc->ilabel, arc->olabel, graph_cost, ac_cost, tok->links);
        }
      }
    }
    e_tail = e->tail;
    hl_delete(&d->toks, e);
  }
  d->num_frames_decoded++;
  return next_cutoff;
}

/* PruneForwardLinks: base-inl.h:482-572.
 * Order-free variant (g_order_free, what the GPU computes): the reference sweeps the frame's token list
 * until a sweep moves no extra_cost by more than delta, and reports "changed" if any sweep did -- with
 * delta > 0 (PruneActiveTokens: lattice_beam * prune_scale) both the values it stops at (epsilon links
 * inside the frame, read one sweep stale) and the report depend on the order of the list, i.e. on the
 * hash-list order the tokens were created in.  The variant sweeps to the exact fixpoint and reports
 * "changed" iff a token's extra_cost ended more than delta away from where it was BEFORE the call.  It
 * coincides with the reference whenever no surviving token of the frame has an epsilon link to another
 * (the usual case) and always with delta = 0 (FinalizeDecoding: the final lattice never differs); in
 * general it is neither finer nor coarser.  tests/test_oracle_lattice.py counts how often the
 * mid-utterance lattices of the two differ on the goldens. */
static void prune_forward_links(Decoder *d, int fpo, int *extra_costs_changed, int *links_pruned, float delta) {
  *extra_costs_changed = 0; *links_pruned = 0;
  if (d->active[fpo].toks == NULL && !d->warned) d->warned = 1;
  const float sweep_delta = g_order_free ? 0.0f : delta;
  if (g_order_free) for (Token *tok = d->active[fpo].toks; tok; tok = tok->next) tok->extra_before = tok->extra_cost;
  int changed = 1;
  while (changed) {
    changed = 0;
    for (Token *tok = d->active[fpo].toks; tok; tok = tok->next) {
      Link *link, *prev_link = NULL;
      float tok_extra_cost = FLOAT_INF;
      for (link = tok->links; link;) {
        Token *nt = link->next_tok;
        float link_extra_cost = nt->extra_cost + ((tok->tot_cost + link->acoustic_cost + link->graph_cost) - nt->tot_cost);
        if (link_extra_cost > d->cfg.lattice_beam) {
          Link *nl = link->next;
          if (prev_link) prev_link->next = nl; else tok->links = nl;
          delete_link(d, link); link = nl; *links_pruned = 1;
        } else {
          if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;
          if (link_extra_cost < tok_extra_cost) tok_extra_cost = link_extra_cost;
          prev_link = link; link = link->next;
        }
      }
      if (fabsf(tok_extra_cost - tok->extra_cost) > sweep_delta) changed = 1;
      tok->extra_cost = tok_extra_cost;
    }
    if (changed && !g_order_free) *extra_costs_changed = 1;
  }
  if (g_order_free)
    for (Token *tok = d->active[fpo].toks; tok; tok = tok->next)
      if (fabsf(tok->extra_cost - tok->extra_before) > delta) *extra_costs_changed = 1;
}

/* PruneTokensForFrame: base-inl.h:578-607 */
static void prune_tokens_for_frame(Decoder *d, int fpo) {
  Token **toks = &d->active[fpo].toks;
  Token *tok, *next_tok, *prev_tok = NULL;
  for (tok = *toks; tok; tok = next_tok) {
    next_tok = tok->next;
    if (tok->extra_cost == FLOAT_INF) {
      if (prev_tok) prev_tok->next = tok->next; else *toks = tok->next;
      delete_token(d, tok);
    } else prev_tok = tok;
  }
}

/* PruneActiveTokens: base-inl.h:438-480 */
static void prune_active_tokens(Decoder *d, float delta) {
  int cur = d->n_active - 1;
  for (int f = cur - 1; f >= 0; f--) {
    if (d->active[f].must_prune_forward_links) {
      int links_pruned = 0, extra_costs_changed = 0;
      prune_forward_links(d, f, &extra_costs_changed, &links_pruned, delta);
      if (extra_costs_changed && f > 0) d->active[f - 1].must_prune_forward_links = 1;
      if (links_pruned) d->active[f].must_prune_tokens = 1;
      d->active[f].must_prune_forward_links = 0;
    }
    if (f + 1 < cur && d->active[f + 1].must_prune_tokens) {
      prune_tokens_for_frame(d, f + 1);
      d->active[f + 1].must_prune_tokens = 0;
    }
  }
}

/* ComputeFinalCosts: base-inl.h:670-720.  mark != 0 fills the final-cost set. */
static void compute_final_costs(Decoder *d, int mark, int *any_final, float *final_relative_cost, float *final_best_cost) {
  float best_cost = FLOAT_INF, best_cost_with_final = FLOAT_INF;
  int any = 0;
  for (const Elem *e = d->toks.list_head; e; e = e->tail) {
    Token *tok = e->val;
    int fst_final = (KEY_STATE(e->key) == d->g->final_state);
    if (tok->tot_cost < best_cost) best_cost = tok->tot_cost;
    if (mark) { tok->is_final = 0; tok->final_cost = 0.0f; }
    if (d->dlm) {
      /* biglm.h:160-215: the LM's final cost enters best_cost_with_final for EVERY token, final in the
       * graph or not; only graph-final tokens are entered in final_costs */
      float lm_final = dlm_final(d->dlm, KEY_LM(e->key));
      float cost_with_final = tok->tot_cost + lm_final;
      if (cost_with_final < best_cost_with_final) best_cost_with_final = cost_with_final;
      if (mark && fst_final) { tok->is_final = 1; tok->final_cost = lm_final; any = 1; }
    } else if (mark && fst_final) { tok->is_final = 1; any = 1; if (tok->tot_cost < best_cost_with_final) best_cost_with_final = tok->tot_cost; }
  }
  if (any_final) *any_final = any;
  if (final_relative_cost) {
    if (best_cost == FLOAT_INF && best_cost_with_final == FLOAT_INF) *final_relative_cost = FLOAT_INF;
    else *final_relative_cost = best_cost_with_final - best_cost;
  }
  if (final_best_cost) *final_best_cost = (best_cost_with_final != FLOAT_INF) ? best_cost_with_final : best_cost;
}

/* PruneForwardLinksFinal: base-inl.h:725-824 */
static void prune_forward_links_final(Decoder *d) {
  int fpo = d->n_active - 1;
  compute_final_costs(d, 1, &d->any_final, &d->final_relative_cost, &d->final_best_cost);
  d->finalized = 1;
  hl_delete_elems(&d->toks);
  int changed = 1; float delta = 1.0e-5f;
  while (changed) {
    changed = 0;
    for (Token *tok = d->active[fpo].toks; tok; tok = tok->next) {
      Link *link, *prev_link = NULL;
      float final_cost = !d->any_final ? 0.0f : (tok->is_final ? tok->final_cost : FLOAT_INF);
      float tok_extra_cost = tok->tot_cost + final_cost - d->final_best_cost;
      for (link = tok->links; link;) {
        Token *nt = link->next_tok;
        float link_extra_cost = nt->extra_cost + ((tok->tot_cost + link->acoustic_cost + link->graph_cost) - nt->tot_cost);
        if (link_extra_cost > d->cfg.lattice_beam) {
          Link *nl = link->next;
          if (prev_link) prev_link->next = nl; else tok->links = nl;
          delete_link(d, link); link = nl;
        } else {
          if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;
          if (link_extra_cost < tok_extra_cost) tok_extra_cost = link_extra_cost;
          prev_link = link; link = link->next;
        }
      }
      if (tok_extra_cost > d->cfg.lattice_beam) tok_extra_cost = FLOAT_INF;
      if (fabsf(tok->extra_cost - tok_extra_cost) > delta) changed = 1;
      tok->extra_cost = tok_extra_cost;
    }
  }
}

/* FinalizeDecoding: base-inl.h:829-847 */
static void finalize_decoding(Decoder *d) {
  int final_fpo = d->n_active - 1;
  prune_forward_links_final(d);
  for (int f = final_fpo - 1; f >= 0; --f) {
    int b1, b2;
    prune_forward_links(d, f, &b1, &b2, 0.0f);
    prune_tokens_for_frame(d, f + 1);
  }
  prune_tokens_for_frame(d, 0);
}

/* InitDecoding: base-inl.h:40-67 */
static void init_decoding(Decoder *d) {
  clear_active_tokens(d);
  hl_delete_elems(&d->toks);
  d->n_queue = 0; d->n_tmp = 0; d->warned = 0; d->finalized = 0; d->any_final = 0;
  active_resize(d, 1);
  Token *start_tok = new_token(d, 0.0f, 0.0f, NULL, NULL, NULL);
  start_tok->state = d->g->start;
  d->active[0].toks = start_tok;
  if (d->dlm) dlm_reset(d->dlm); /* biglm.h:110-112: start pair = (graph start, _diff_lm.Start() == 0) */
  hl_insert(&d->toks, make_key(d->g->start, 0), start_tok);
  d->num_frames_decoded = 0; /* set before the closure: it is not read there */
  process_nonemitting(d, d->cfg.beam);
  d->num_frames_decoded = 0;
}

/* AdvanceDecoding: base-inl.h:630-668 */
static void advance_decoding(Decoder *d, int max_num_frames) {
  int target = d->frames_ready;
  if (max_num_frames >= 0 && d->num_frames_decoded + max_num_frames < target) target = d->num_frames_decoded + max_num_frames;
  while (d->num_frames_decoded < target) {
    if ((d->n_active - 1) % d->cfg.prune_interval == 0) prune_active_tokens(d, d->cfg.lattice_beam * d->cfg.prune_scale);
    float cutoff = process_emitting(d);
    process_nonemitting(d, cutoff);
  }
}

/* ---- graph loading ---- */
static void graph_index(Graph *g) {
  g->off = (int64_t *)malloc(((size_t)g->n_states + 1) * sizeof(int64_t));
  int64_t o = 0;
  for (int i = 0; i < g->n_states; ++i) { g->off[i] = o; o += g->si[i].num_arcs; }
  g->off[g->n_states] = o;
}

void *oracle_graph_load(const char *path) { /* Fst::ReadFst, newfst/optimize-fst.h:226-280 */
  FILE *fp = fopen(path, "rb");
  if (!fp) return NULL;
  int hdr[6];
  if (fread(hdr, sizeof(int), 6, fp) != 6) { fclose(fp); return NULL; }
  Graph *g = (Graph *)calloc(1, sizeof(Graph));
  g->start = hdr[0]; g->final_state = hdr[1]; g->n_states = hdr[2]; g->n_arcs = hdr[3];
  g->si = (StateInfo *)malloc((size_t)g->n_states * sizeof(StateInfo));
  g->arcs = (Arc *)malloc((size_t)g->n_arcs * sizeof(Arc) + 16);
  int ok = fread(g->si, sizeof(StateInfo), g->n_states, fp) == (size_t)g->n_states &&
           fread(g->arcs, sizeof(Arc), g->n_arcs, fp) == (size_t)g->n_arcs;
  fclose(fp);
  if (ok) { graph_index(g); if (g->off[g->n_states] != g->n_arcs) ok = 0; }
  if (!ok) { free(g->si); free(g->arcs); free(g->off); free(g); return NULL; }
  return g;
}

void oracle_graph_free(void *gp) {
  Graph *g = (Graph *)gp; if (!g) return;
  free(g->si); free(g->arcs); free(g->off); free(g);
}

/* Decode one utterance.  Same argument list and meaning as ref_decode() in
 * oracle/ref_driver.cc; `extra` (nullable, 8 x int64) receives
 * {N, E, Z, tokens created, links created, tie hops on best path, quirk hops, 0}. */
static int decode_impl(void *gp, const Config *rc, DiffLm *dlm, const float *loglikes, int T, int stride,
                     const int *tid2pdf, int n_tid, int chunk, int do_finalize, int use_final_probs,
                     int *path_ilabel, int *path_olabel, float *path_graph, float *path_ac,
                     int max_path, int *n_path, float *tot_score, float *lm_score, int *words,
                     int max_words, int *n_words, int *tids, int max_tids, int *n_tids,
                     int *frame_ntoks, float *frame_best, int dump_frame, int *dump_states,
                     float *dump_costs, int dump_cap, int *dump_n, int *num_toks_end,
                     int *num_links_end, int64_t *extra) {
  (void)n_tid;
  Decoder D; memset(&D, 0, sizeof(D));
  Decoder *d = &D;
  d->g = (const Graph *)gp; d->cfg = *rc; d->dlm = dlm;
  d->toks.bucket_list_tail = NOBUCKET;
  d->tok_pool.elem_size = sizeof(Token); d->link_pool.elem_size = sizeof(Link);
  d->ll = loglikes; d->T = T; d->stride = stride; d->tid2pdf = tid2pdf; d->frames_ready = T;
  /* ctor: base-inl.h:27 (the reference would ask for ~68 GB with max_active=INT_MAX; the
   * oracle caps the *initial* size there, which the reference cannot run at all) */
  {
    float fs = (float)rc->max_active * rc->hash_ratio;
    size_t sz = fs > 1.0e9f ? (size_t)1000 : (size_t)fs;
    hl_set_size(&d->toks, sz);
  }
#define FRONTIER_STATS(idx) do { if (chunk == 1 && frame_ntoks) { int n_ = 0; float b_ = FLOAT_INF; \
    for (const Elem *e_ = d->toks.list_head; e_; e_ = e_->tail) { ++n_; if (e_->val->tot_cost < b_) b_ = e_->val->tot_cost; } \
    frame_ntoks[idx] = n_; frame_best[idx] = b_; } \
    if (chunk == 1 && dump_frame == (idx) && dump_n) { int n_ = 0; \
    for (const Elem *e_ = d->toks.list_head; e_; e_ = e_->tail) { if (n_ < dump_cap) { dump_states[n_] = KEY_STATE(e_->key); dump_costs[n_] = e_->val->tot_cost; } ++n_; } \
    *dump_n = n_; } } while (0)

  init_decoding(d);
  FRONTIER_STATS(0);
  if (chunk <= 0) { advance_decoding(d, -1); }
  else {
    for (int r = 0; r < T;) {
      r = (r + chunk < T) ? r + chunk : T;
      d->frames_ready = r;
      advance_decoding(d, -1);
      FRONTIER_STATS(r);
    }
  }
  if (do_finalize) finalize_decoding(d);
  if (num_toks_end) *num_toks_end = d->num_toks;
  if (num_links_end) *num_links_end = d->num_links;

  *n_path = 0; *n_words = 0; *n_tids = 0; *tot_score = 0; *lm_score = 0;
  int ok = 0;
  int64_t tie_hops = 0, quirk_hops = 0;
  /* BestPathEnd: base-inl.h:1096-1158 */
  if (d->n_active - 1 > 0) {
    int any_final = d->any_final;
    if (!d->finalized && use_final_probs) compute_final_costs(d, 1, &any_final, NULL, NULL);
    float best_cost = FLOAT_INF; Token *best_tok = NULL;
    for (Token *tok = d->active[d->n_active - 1].toks; tok; tok = tok->next) {
      float cost = tok->tot_cost;
      if (use_final_probs && any_final) { if (!tok->is_final) cost = FLOAT_INF; else cost += tok->final_cost; }
      if (cost < best_cost) { best_cost = cost; best_tok = tok; }
    }
    if (best_tok) {
      /* GetBestPath + TraceBackBestPath: base-inl.h:1071-1094,1160-1200.  Hops come out last
       * to first; the lattice is walked start->final by LatticeToVector, i.e. reversed. */
      int cap = 1024, n = 0;
      int *hi = (int *)malloc(cap * sizeof(int)), *ho = (int *)malloc(cap * sizeof(int));
      float *hg = (float *)malloc(cap * sizeof(float)), *ha = (float *)malloc(cap * sizeof(float));
      for (Token *tok = best_tok; tok;) {
        int il = 0, ol = 0; float gc = 0.0f, ac = 0.0f;
        if (tok->tie) tie_hops++;
        if (tok->backpointer) {
          Link *link;
          for (link = tok->backpointer->links; link; link = link->next)
            if (link->next_tok == tok) { il = link->ilabel; ol = link->olabel; gc = link->graph_cost; ac = link->acoustic_cost; break; }
          if (link && (tok->backpointer->tot_cost + ac) + gc != tok->tot_cost) quirk_hops++;
          if (!link) quirk_hops += 1000000; /* "Error tracing best-path back" (base-inl.h:1187-1191): never expected */
        }
        if (n == cap) { cap *= 2; hi = (int *)realloc(hi, cap * sizeof(int)); ho = (int *)realloc(ho, cap * sizeof(int)); hg = (float *)realloc(hg, cap * sizeof(float)); ha = (float *)realloc(ha, cap * sizeof(float)); }
        hi[n] = il; ho[n] = ol; hg[n] = gc; ha[n] = ac; ++n;
        tok = tok->backpointer;
      }
      /* LatticeToVector: newfst/lattice-functions.cc:179-217 */
      float tot = 0, lm = 0; int nw = 0, nt = 0;
      for (int k = n - 1, j = 0; k >= 0; --k, ++j) {
        if (j < max_path) { path_ilabel[j] = hi[k]; path_olabel[j] = ho[k]; path_graph[j] = hg[k]; path_ac[j] = ha[k]; }
        if (hi[k] != 0) { if (nt < max_tids) tids[nt] = hi[k]; nt++; }
        if (ho[k] != 0) { if (nw < max_words) words[nw] = ho[k]; nw++; }
        lm += hg[k];
        tot += hg[k] + ha[k];
      }
      *n_path = n; *n_words = nw; *n_tids = nt; *tot_score = tot; *lm_score = lm;
      free(hi); free(ho); free(hg); free(ha);
      ok = 1;
    }
  }
  if (extra) { extra[0] = d->cnt_N; extra[1] = d->cnt_E; extra[2] = d->cnt_Z; extra[3] = d->cnt_tok_created; extra[4] = d->cnt_link_created; extra[5] = tie_hops; extra[6] = quirk_hops;
               extra[7] = dlm ? ((int64_t)dlm->n_vec | ((int64_t)dlm->oob << 40)) : 0;
               if (dlm) { extra[8] = d->cnt_L; extra[9] = d->cnt_Leps; } /* biglm callers pass 10 slots */ }

  /* teardown */
  clear_active_tokens(d);
  hl_delete_elems(&d->toks);
  for (size_t i = 0; i < d->toks.n_blocks; ++i) free(d->toks.blocks[i]);
  free(d->toks.blocks); free(d->toks.buckets);
  pool_destroy(&d->tok_pool); pool_destroy(&d->link_pool);
  free(d->active); free(d->queue); free(d->tmp);
  return ok;
}

int oracle_decode_ex(void *gp, const Config *rc, const float *loglikes, int T, int stride,
                     const int *tid2pdf, int n_tid, int chunk, int do_finalize, int use_final_probs,
                     int *path_ilabel, int *path_olabel, float *path_graph, float *path_ac,
                     int max_path, int *n_path, float *tot_score, float *lm_score, int *words,
                     int max_words, int *n_words, int *tids, int max_tids, int *n_tids,
                     int *frame_ntoks, float *frame_best, int dump_frame, int *dump_states,
                     float *dump_costs, int dump_cap, int *dump_n, int *num_toks_end,
                     int *num_links_end, int64_t *extra) {
  return decode_impl(gp, rc, NULL, loglikes, T, stride, tid2pdf, n_tid, chunk, do_finalize, use_final_probs, path_ilabel,
                     path_olabel, path_graph, path_ac, max_path, n_path, tot_score, lm_score, words, max_words, n_words,
                     tids, max_tids, n_tids, frame_ntoks, frame_best, dump_frame, dump_states, dump_costs, dump_cap,
                     dump_n, num_toks_end, num_links_end, extra);
}

/* biglm: same arguments as ref_biglm_decode() in oracle/ref_driver.cc plus `fixed` (0: DiffArpaLm as
 * written, 1: pair components) and `extra` (10 x int64: [0..6] as oracle_decode_ex; [8] = labelled arcs traversed = LM look-ups; [7] = LM pair states
 * interned | (out-of-range LM access seen) << 40).  lm1 = old LM (rescaled by -1 at load), lm2 = new. */
int oracle_biglm_decode(void *gp, const Config *rc, void *lm1, void *lm2, int fixed, const float *loglikes, int T,
                        int stride, const int *tid2pdf, int n_tid, int chunk, int do_finalize, int use_final_probs,
                        int *path_ilabel, int *path_olabel, float *path_graph, float *path_ac, int max_path,
                        int *n_path, float *tot_score, float *lm_score, int *words, int max_words, int *n_words,
                        int *tids, int max_tids, int *n_tids, int *frame_ntoks, float *frame_best,
                        int *num_toks_end, int *num_links_end, int64_t *extra) {
  DiffLm dlm;
  dlm_init(&dlm, (const Lm *)lm1, (const Lm *)lm2, fixed);
  int ok = decode_impl(gp, rc, &dlm, loglikes, T, stride, tid2pdf, n_tid, chunk, do_finalize, use_final_probs, path_ilabel,
                       path_olabel, path_graph, path_ac, max_path, n_path, tot_score, lm_score, words, max_words, n_words,
                       tids, max_tids, n_tids, frame_ntoks, frame_best, -1, NULL, NULL, 0, NULL, num_toks_end,
                       num_links_end, extra);
  dlm_free(&dlm);
  return ok;
}

/* GetRawLattice (base-inl.h:869-975) after one AdvanceDecoding over all frames (+ FinalizeDecoding).
 * Same outputs as ref_raw_lattice() in oracle/ref_driver.cc plus, per lattice state, the frame and
 * graph state of its token (st_frame/st_gstate) so that another implementation can be compared
 * state by state.  State numbering: frame by frame in token-list order (the reference numbers by
 * TopSortTokens over an unordered_map keyed by pointers, i.e. implementation defined; compare up to
 * isomorphism).  State 0 is the start token. */
static int raw_lattice_impl(void *gp, const Config *rc, DiffLm *dlm, const float *loglikes, int T, int stride,
                       const int *tid2pdf, int n_tid, int do_finalize, int use_final_probs,
                       int max_states, int *n_states, int *start, int *st_final, int *st_frame,
                       int *st_gstate, float *st_cost, int max_arcs, int *n_arcs, int *a_src, int *a_dst, int *a_il,
                       int *a_ol, float *a_graph, float *a_ac) {
  (void)n_tid;
  Decoder D; memset(&D, 0, sizeof(D));
  Decoder *d = &D;
  d->g = (const Graph *)gp; d->cfg = *rc; d->dlm = dlm;
  d->toks.bucket_list_tail = NOBUCKET;
  d->tok_pool.elem_size = sizeof(Token); d->link_pool.elem_size = sizeof(Link);
  d->ll = loglikes; d->T = T; d->stride = stride; d->tid2pdf = tid2pdf; d->frames_ready = T;
  {
    float fs = (float)rc->max_active * rc->hash_ratio;
    size_t sz = fs > 1.0e9f ? (size_t)1000 : (size_t)fs;
    hl_set_size(&d->toks, sz);
  }
  init_decoding(d);
  advance_decoding(d, -1);
  if (do_finalize) finalize_decoding(d);
  *n_states = 0; *n_arcs = 0; *start = -1;
  int ok = 0;
  const int num_frames = d->n_active - 1;
  if (!(d->finalized && !use_final_probs) && num_frames > 0) {
    int any_final = d->any_final;
    if (!d->finalized && use_final_probs) compute_final_costs(d, 1, &any_final, NULL, NULL);
    ok = 1;
    int ns = 0;
    /* TopSortTokens (base-inl.h:976-1068): positions num_toks-1..0 in list order, an epsilon link to
     * a token placed earlier moves that token to a fresh position at the end, repeated until stable;
     * states are numbered by final position.  The reference walks an unordered_map keyed by the
     * token POINTER, so its numbering inside a frame is not reproducible; this restatement walks the
     * list in order (any order gives a valid topological numbering -- what the tests check). */
    for (int f = 0; f <= num_frames && ok; ++f) {
      if (d->active[f].toks == NULL) { ok = 0; break; }
      int n = 0;
      for (Token *t = d->active[f].toks; t; t = t->next) ++n;
      Token **arr = (Token **)malloc(sizeof(Token *) * (size_t)n);
      int cur_pos = 0;
      for (Token *t = d->active[f].toks; t; t = t->next) { arr[cur_pos] = t; t->lat_id = n - (++cur_pos); }
      size_t qcap = 64, qn = 0, qi = 0;
      Token **q = (Token **)malloc(sizeof(Token *) * qcap);
      for (int i = 0; i < n || qi < qn; ++i) {   /* the list once, then the reprocess queue until empty */
        Token *t = i < n ? arr[i] : q[qi++];
        for (Link *l = t->links; l; l = l->next)
          if (l->ilabel == 0 && l->next_tok->lat_id < t->lat_id) {   /* epsilon links stay inside the frame */
            l->next_tok->lat_id = cur_pos++;
            if (qn == qcap) { qcap *= 2; q = (Token **)realloc(q, sizeof(Token *) * qcap); }
            q[qn++] = l->next_tok;
          }
      }
      /* compact the positions of this frame into consecutive state ids (the reference leaves NULL gaps
       * in its list and skips them) */
      int *order = (int *)malloc(sizeof(int) * (size_t)(cur_pos > 0 ? cur_pos : 1));
      for (int i = 0; i < cur_pos; ++i) order[i] = -1;
      for (int i = 0; i < n; ++i) order[arr[i]->lat_id] = i;
      for (int i = 0; i < cur_pos; ++i) if (order[i] >= 0) arr[order[i]]->lat_id = ns++;
      free(order); free(q); free(arr);
    }
    if (ok) {
      int na = 0;
      for (int f = 0; f <= num_frames; ++f)
        for (Token *t = d->active[f].toks; t; t = t->next) {
          const int s = t->lat_id;
          float final_cost = 0.0f;
          int is_fin = 0;
          if (f == num_frames) {
            if (use_final_probs && any_final) { if (t->is_final) { is_fin = 1; final_cost = t->final_cost; } }
            else is_fin = 1;
          }
          if (s < max_states) { st_final[s] = is_fin; st_frame[s] = f; st_gstate[s] = t->state; st_cost[s] = t->tot_cost; }
          for (Link *l = t->links; l; l = l->next) {
            if (na < max_arcs) {
              a_src[na] = s; a_dst[na] = l->next_tok->lat_id; a_il[na] = l->ilabel; a_ol[na] = l->olabel;
              a_graph[na] = l->graph_cost + final_cost; a_ac[na] = l->acoustic_cost;
            }
            ++na;
          }
        }
      *n_states = ns; *n_arcs = na; *start = 0;
    }
  }
  clear_active_tokens(d);
  hl_delete_elems(&d->toks);
  for (size_t i = 0; i < d->toks.n_blocks; ++i) free(d->toks.blocks[i]);
  free(d->toks.blocks); free(d->toks.buckets);
  pool_destroy(&d->tok_pool); pool_destroy(&d->link_pool);
  free(d->active); free(d->queue); free(d->tmp);
  return ok;
}

int oracle_raw_lattice(void *gp, const Config *rc, const float *loglikes, int T, int stride,
                       const int *tid2pdf, int n_tid, int do_finalize, int use_final_probs,
                       int max_states, int *n_states, int *start, int *st_final, int *st_frame,
                       int *st_gstate, float *st_cost, int max_arcs, int *n_arcs, int *a_src, int *a_dst, int *a_il,
                       int *a_ol, float *a_graph, float *a_ac) {
  return raw_lattice_impl(gp, rc, NULL, loglikes, T, stride, tid2pdf, n_tid, do_finalize, use_final_probs, max_states, n_states, start,
                          st_final, st_frame, st_gstate, st_cost, max_arcs, n_arcs, a_src, a_dst, a_il, a_ol, a_graph, a_ac);
}

/* The same from the biglm decoder (my-decoder/online-decoder-mempool-base-biglm.h: a lattice decoder -- the service asks it for
 * GetRawLattice / GetLattice / n-best, kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:58,81,97-105): links carry graph cost = arc
 * weight + lm_score (:377-392, :448-458), FinalizeDecoding prunes with the LM's final costs (:469-560, ComputeFinalCosts
 * :160-215).  As in the base class's GetRawLattice a final token is only FLAGGED final: its final cost is added to the graph
 * cost of its out-links (base-inl.h:930-966), of which a graph-final token has none.  `fixed` as oracle_biglm_decode. */
int oracle_biglm_raw_lattice(void *gp, const Config *rc, void *lm1, void *lm2, int fixed, const float *loglikes, int T, int stride,
                             const int *tid2pdf, int n_tid, int do_finalize, int use_final_probs,
                             int max_states, int *n_states, int *start, int *st_final, int *st_frame,
                             int *st_gstate, float *st_cost, int max_arcs, int *n_arcs, int *a_src, int *a_dst, int *a_il,
                             int *a_ol, float *a_graph, float *a_ac) {
  DiffLm dlm;
  dlm_init(&dlm, (const Lm *)lm1, (const Lm *)lm2, fixed);
  const int ok = raw_lattice_impl(gp, rc, &dlm, loglikes, T, stride, tid2pdf, n_tid, do_finalize, use_final_probs, max_states, n_states,
                                  start, st_final, st_frame, st_gstate, st_cost, max_arcs, n_arcs, a_src, a_dst, a_il, a_ol, a_graph, a_ac);
  dlm_free(&dlm);
  return ok;
}

int oracle_decode(void *gp, const Config *rc, const float *loglikes, int T, int stride,
                  const int *tid2pdf, int n_tid, int chunk, int do_finalize, int use_final_probs,
                  int *path_ilabel, int *path_olabel, float *path_graph, float *path_ac,
                  int max_path, int *n_path, float *tot_score, float *lm_score, int *words,
                  int max_words, int *n_words, int *tids, int max_tids, int *n_tids,
                  int *frame_ntoks, float *frame_best, int dump_frame, int *dump_states,
                  float *dump_costs, int dump_cap, int *dump_n, int *num_toks_end,
                  int *num_links_end) {
  return oracle_decode_ex(gp, rc, loglikes, T, stride, tid2pdf, n_tid, chunk, do_finalize,
                          use_final_probs, path_ilabel, path_olabel, path_graph, path_ac, max_path,
                          n_path, tot_score, lm_score, words, max_words, n_words, tids, max_tids,
                          n_tids, frame_ntoks, frame_best, dump_frame, dump_states, dump_costs,
                          dump_cap, dump_n, num_toks_end, num_links_end, NULL);
}

/* CPU baseline leg of bench.py where the prebuilt reference library is absent ("port"): same
 * contract as ref_timed_loop() in oracle/ref_driver.cc -- decode mats[first], mats[first+step], ...
 * (wrapping) until `seconds` of wall time have passed; returns the frames decoded. */
long long oracle_timed_loop(void *gp, const Config *rc, const float *const *mats, const int *T, int n_mats,
                            int stride, const int *tid2pdf, int n_tid, int first, int step, double seconds,
                            double *elapsed, long long *words_out) {
  long long frames = 0, nwords = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  double dt = 0.0;
  for (int i = first % n_mats;; i = (i + step) % n_mats) {
    const int mp = 4 * T[i] + 64;
    int *ib = (int *)malloc(sizeof(int) * 4 * (size_t)mp);
    float *fb = (float *)malloc(sizeof(float) * 2 * (size_t)mp);
    int np = 0, nw = 0, nt = 0, a = 0, b = 0; float tot = 0, lm = 0;
    oracle_decode_ex(gp, rc, mats[i], T[i], stride, tid2pdf, n_tid, 0, 1, 1, ib, ib + mp, fb, fb + mp, mp, &np, &tot, &lm,
                     ib + 2 * mp, mp, &nw, ib + 3 * mp, mp, &nt, NULL, NULL, -1, NULL, NULL, 0, NULL, &a, &b, NULL);
    free(ib); free(fb);
    nwords += nw; frames += T[i];
    clock_gettime(CLOCK_MONOTONIC, &t1);
    dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (dt >= seconds) break;
  }
  if (elapsed) *elapsed = dt;
  if (words_out) *words_out = nwords;
  return frames;
}

/* oracle_timed_loop() for the biglm decoder (fixed mode); bench.py --biglm where oracle/_ref is absent. */
long long oracle_biglm_timed_loop(void *gp, const Config *rc, void *lm1, void *lm2, const float *const *mats, const int *T,
                                  int n_mats, int stride, const int *tid2pdf, int n_tid, int first, int step,
                                  double seconds, double *elapsed, long long *words_out) {
  long long frames = 0, nwords = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  double dt = 0.0;
  for (int i = first % n_mats;; i = (i + step) % n_mats) {
    const int mp = 4 * T[i] + 64;
    int *ib = (int *)malloc(sizeof(int) * 4 * (size_t)mp);
    float *fb = (float *)malloc(sizeof(float) * 2 * (size_t)mp);
    int np = 0, nw = 0, nt = 0, a = 0, b = 0; float tot = 0, lm = 0;
    oracle_biglm_decode(gp, rc, lm1, lm2, 1, mats[i], T[i], stride, tid2pdf, n_tid, 0, 1, 1, ib, ib + mp, fb, fb + mp, mp, &np,
                        &tot, &lm, ib + 2 * mp, mp, &nw, ib + 3 * mp, mp, &nt, NULL, NULL, &a, &b, NULL);
    free(ib); free(fb);
    nwords += nw; frames += T[i];
    clock_gettime(CLOCK_MONOTONIC, &t1);
    dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (dt >= seconds) break;
  }
  if (elapsed) *elapsed = dt;
  if (words_out) *words_out = nwords;
  return frames;
}
