/* wavesim.c -- ANALYSIS ONLY (tools/wavesim.py, tools/blocksim.py): wave- and block-level models of the product's one-lane-per-ray
 * find traversal on its exported Node4 / TriRec arrays.  Moved out of the parity oracle in round 3: a performance model is not
 * checker code.  Self-contained (its own ray / box / triangle arithmetic: a model needs the visit counts, not the oracle's
 * bit-exact spec); nothing in the product, the oracle or the tests links it.  build: tools/Makefile */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } orc_vec3;
typedef struct { orc_vec3 v0, e1, e2, Ng, n; } orc_tri;
static orc_vec3 v3(float x, float y, float z) { orc_vec3 r = {x, y, z}; return r; }
static float safe_inv(float d) { float ad = fabsf(d); float s = (ad < 1e-30f) ? copysignf(1e-30f, d) : d; return 1.0f / s; }
/* Moeller-Trumbore in Embree's formulation (strict near side, t <= tfar) */
static int tri_intersect(const orc_tri* T, orc_vec3 O, orc_vec3 D, float tnear, float tfar, float* t_out)
{
  const float Cx = T->v0.x - O.x, Cy = T->v0.y - O.y, Cz = T->v0.z - O.z;
  const float Rx = Cy * D.z - Cz * D.y, Ry = Cz * D.x - Cx * D.z, Rz = Cx * D.y - Cy * D.x;
  const float den = T->Ng.x * D.x + T->Ng.y * D.y + T->Ng.z * D.z;
  float U = Rx * T->e2.x + Ry * T->e2.y + Rz * T->e2.z;
  float V = Rx * T->e1.x + Ry * T->e1.y + Rz * T->e1.z;
  float Tt = T->Ng.x * Cx + T->Ng.y * Cy + T->Ng.z * Cz;
  if (den < 0.0f) { U = -U; V = -V; Tt = -Tt; }
  const float aden = fabsf(den);
  if (den == 0.0f || U < 0.0f || V < 0.0f || U + V > aden) return 0;
  const float t = Tt / aden;
  if (!(Tt > aden * tnear) || !(t <= tfar)) return 0;
  *t_out = t;
  return 1;
}

/* ------------------------------------------------------------------------- */
/* wave-level model of the product's one-lane-per-ray "while-while" traversal.  The     */
/* single-scan find is bound by the slowest WAVE's chain of dependent steps, so what a builder / ordering change buys   */
/* is measured here as wave-level node iterations and leaf rounds (a wave iterates while ANY of its lanes still has   */
/* an inner node / a leaf), on the product's exported Node4 / TriRec arrays.                                          */
/* mode bit0: cull popped entries whose entry distance exceeds best_t; bit1: (always) sort deferred children;         */
/* bit2: entry-distance ties at 0 (origin inside several child boxes) are ordered by EXIT distance, farthest first.    */
/* out[0] wave node iterations, out[1] wave leaf rounds, out[2] wave triangle iterations (sum over rounds of the       */
/* longest leaf), out[3] sum of lane node visits, out[4] max lane node visits, out[5] sum lane leaf visits.            */
/* ------------------------------------------------------------------------- */
/* cost model of orc_wavesim_ww (cycles): per wave-level node iteration / triangle iteration, by the number of rays still
 * walking at the start of the round: > t2 one lane per ray, <= t2 two lanes per ray, <= t4 four lanes per ray (a
 * cooperative leaf step tests a whole leaf at once).  Set by orc_wavesim_costs; out[6] accumulates the estimate. */
static struct { double n1, l1, n2, l2, n4, l4; uint32_t t2, t4; } g_ws_cost = {880, 300, 570, 400, 450, 400, 0, 0};
void orc_wavesim_costs(double n1, double l1, double n2, double l2, double n4, double l4, uint32_t t2, uint32_t t4)
{
  g_ws_cost.n1 = n1; g_ws_cost.l1 = l1; g_ws_cost.n2 = n2; g_ws_cost.l2 = l2; g_ws_cost.n4 = n4; g_ws_cost.l4 = l4;
  g_ws_cost.t2 = t2; g_ws_cost.t4 = t4;
}

typedef struct {
  uint32_t cur; int sp; int done; float best_t; uint32_t best_f; int found;
  float o[3], inv[3]; orc_vec3 O, D;
  uint32_t stack[128]; float stack_t[128];
  uint64_t nvisit, lvisit;
  uint32_t pend;   /* postponed leaf (mode 0x20000), 0 = none */
} ws_lane;

static int ws_box(const float* nd, uint32_t c, const float* o, const float* inv, float best_t, float* tn_out, float* tf_out)
{
  float tn = 0.0f, tf = best_t;
  for (int k = 0; k < 3; ++k) {
    const float lo = nd[8 * k + c], hi = nd[8 * k + 4 + c];
    float t0 = (lo - o[k]) * inv[k], t1 = (hi - o[k]) * inv[k];
    if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
    if (t0 > tn) tn = t0;
    if (t1 < tf) tf = t1;
  }
  *tn_out = tn; *tf_out = tf;
  return nd[c] < 1e29f && tn <= tf;
}

static void ws_pop(ws_lane* L, int mode)
{
  while (L->sp > 0) {
    --L->sp;
    if ((mode & 1) && L->stack_t[L->sp] > L->best_t) continue;
    L->cur = L->stack[L->sp];
    return;
  }
  L->done = 1;
}

/* tracking mode (round 3): per-lane best_t seeds (the previous scan's hit distance); NULL = cold */
static const float* g_seed_t = 0;
void orc_wavesim_seed(const float* t) { g_seed_t = t; }


/* frontier start (round 3 idea): instead of descending from the root, a ray starts with the nodes of BFS depth `fd` (and the
 * leaves above that depth) its own box tests accept, nearest first -- the top `fd` levels of per-ray descent are replaced by a
 * wave-cooperative frustum culling of the (<= 4^fd) frontier entries plus one box test per surviving candidate.  Here: exact
 * per-ray filter (what the kernel's per-lane filter computes), visits of the skipped levels are not counted. */
static int g_frontier_depth = 0;
static uint64_t g_frontier_cands = 0;
uint64_t orc_wavesim_frontier_cands(void) { uint64_t v = g_frontier_cands; g_frontier_cands = 0; return v; }
void orc_wavesim_frontier(int depth) { g_frontier_depth = depth; }
static int g_frontier_order = 0;
void orc_wavesim_frontier_order(int o) { g_frontier_order = o; }
static void ws_frontier(ws_lane* l, const uint32_t* nodes, uint32_t node, int depth, uint32_t* cand, float* ckey, int* nc)
{
  const float* nd = (const float*)(nodes + 32u * node);
  const uint32_t* ch = nodes + 32u * node + 24u;
  for (uint32_t c = 0; c < 4; ++c) {
    float tn, tf;
    if (!ws_box(nd, c, l->o, l->inv, l->best_t, &tn, &tf)) continue;
    if (!(ch[c] & 0x80000000u) && depth + 1 < g_frontier_depth) ws_frontier(l, nodes, ch[c], depth + 1, cand, ckey, nc);
    else if (*nc < 256) { cand[*nc] = ch[c]; ckey[*nc] = tn; (*nc)++; }
  }
}

int orc_wavesim_ww(const uint32_t* nodes, const uint32_t* tris, const float* O, const float* D, uint32_t nlanes,
                   float tfar, int mode, uint64_t out[8], float* t_out, uint32_t* face_out)
{
  double est = 0.0;
  if (nlanes > 64) return -1;
  ws_lane* L = (ws_lane*)calloc(nlanes ? nlanes : 1, sizeof(ws_lane));
  for (uint32_t i = 0; i < nlanes; ++i) {
    L[i].O = v3(O[3 * i], O[3 * i + 1], O[3 * i + 2]); L[i].D = v3(D[3 * i], D[3 * i + 1], D[3 * i + 2]);
    L[i].o[0] = L[i].O.x; L[i].o[1] = L[i].O.y; L[i].o[2] = L[i].O.z;
    L[i].inv[0] = safe_inv(L[i].D.x); L[i].inv[1] = safe_inv(L[i].D.y); L[i].inv[2] = safe_inv(L[i].D.z);
    L[i].best_t = (g_seed_t && g_seed_t[i] > 0.0f) ? g_seed_t[i] : tfar; L[i].best_f = 0xFFFFFFFFu;
    L[i].done = !(L[i].D.x == L[i].D.x && L[i].D.y == L[i].D.y && L[i].D.z == L[i].D.z);
    if (g_frontier_depth > 0 && !L[i].done) {
      uint32_t cand[256]; float ckey[256]; int nc = 0;
      ws_frontier(&L[i], nodes, 0, 0, cand, ckey, &nc);
      if (g_frontier_order == 0) {   /* fully sorted: nearest first, then by entry distance */
        for (int a = 0; a < nc; ++a) for (int b = a + 1; b < nc; ++b)
          if (ckey[b] < ckey[a]) { float t = ckey[a]; ckey[a] = ckey[b]; ckey[b] = t; uint32_t r = cand[a]; cand[a] = cand[b]; cand[b] = r; }
      } else {                       /* what round 3's first kernel did: only the nearest is found, the others stay in table order */
        int best = 0;
        for (int a = 1; a < nc; ++a) if (ckey[a] < ckey[best]) best = a;
        if (nc > 0) { float t = ckey[0]; ckey[0] = ckey[best]; ckey[best] = t; uint32_t r = cand[0]; cand[0] = cand[best]; cand[best] = r; }
        if (g_frontier_order == 2)   /* ... and the two nearest */
          { int b2 = 1; for (int a = 2; a < nc; ++a) if (ckey[a] < ckey[b2]) b2 = a;
            if (nc > 1) { float t = ckey[1]; ckey[1] = ckey[b2]; ckey[b2] = t; uint32_t r = cand[1]; cand[1] = cand[b2]; cand[b2] = r; } }
      }
      if (nc == 0) L[i].done = 1;
      else { for (int a = nc - 1; a >= 1; --a) { L[i].stack[L[i].sp] = cand[a]; L[i].stack_t[L[i].sp] = ckey[a]; L[i].sp++; } L[i].cur = cand[0]; }
      g_frontier_cands += (uint64_t)nc;
    }
  }
  memset(out, 0, 8 * sizeof(uint64_t));
  for (;;) {
    int any = 0; uint32_t walking = 0;
    for (uint32_t i = 0; i < nlanes; ++i) { any |= !L[i].done; walking += L[i].done ? 0u : 1u; }
    if (!any) break;
    const int coop = (walking <= g_ws_cost.t4) ? 4 : ((walking <= g_ws_cost.t2) ? 2 : 1);
    const double cn = coop == 4 ? g_ws_cost.n4 : (coop == 2 ? g_ws_cost.n2 : g_ws_cost.n1);
    const double cl = coop == 4 ? g_ws_cost.l4 : (coop == 2 ? g_ws_cost.l2 : g_ws_cost.l1);
    /* phase 1: node iterations while any lane holds an inner node (policy, mode bits 8..15 = T > 0: the phase is also
     * left as soon as at least T lanes hold a leaf) */
    const uint32_t leaf_trigger = ((uint32_t)mode >> 8) & 0xFFu;
    uint32_t steps_this_round = 0;
    const uint32_t alive_at_round_start = walking;
    for (;;) {
      int inner = 0; uint32_t holding = 0;
      /* mode 0x20000, "speculative while-while" (Aila & Laine 2009): a lane that arrives at a leaf parks it (one slot)
       * and keeps walking with the next stack entry instead of idling until the leaf phase */
      if (mode & 0x20000)
        for (uint32_t i = 0; i < nlanes; ++i) {
          ws_lane* l = &L[i];
          if (!l->done && (l->cur & 0x80000000u) && l->pend == 0 && l->sp > 0) { l->pend = l->cur; ws_pop(l, mode); if (l->done) { l->done = 0; l->cur = l->pend; l->pend = 0; } }
        }
      for (uint32_t i = 0; i < nlanes; ++i) {
        inner |= (!L[i].done && !(L[i].cur & 0x80000000u));
        holding += (!L[i].done && (L[i].cur & 0x80000000u)) ? 1u : 0u;
      }
      if (!inner) break;
      if (leaf_trigger && holding >= leaf_trigger) break;
      /* mode 0x40000: the product's leaf trigger (kernels.hip trace_lane_bf_tail): after at least one step of the round,
       * leave when the descending lanes are <= 40 % of the rays alive at the start of the round */
      if ((mode & 0x40000) && steps_this_round > 0) {
        uint32_t n_in = 0;
        for (uint32_t i = 0; i < nlanes; ++i) n_in += (!L[i].done && !(L[i].cur & 0x80000000u)) ? 1u : 0u;
        if (5u * n_in <= 2u * alive_at_round_start) break;
      }
      steps_this_round++;
      out[0]++;
      est += cn;
      for (uint32_t i = 0; i < nlanes; ++i) {
        ws_lane* l = &L[i];
        if (l->done || (l->cur & 0x80000000u)) continue;
        l->nvisit++;
        const float* nd = (const float*)(nodes + 32u * l->cur);
        const uint32_t* ch = nodes + 32u * l->cur + 24u;
        float key[4], key2[4]; uint32_t ref[4]; int nh = 0;
        for (uint32_t c = 0; c < 4; ++c) {
          float tn, tf;
          if (ws_box(nd, c, l->o, l->inv, l->best_t, &tn, &tf)) { key[nh] = tn; key2[nh] = (mode & 4) ? -tf : 0.0f; ref[nh] = ch[c]; nh++; }
        }
        for (int a = 0; a < nh; ++a) for (int b = a + 1; b < nh; ++b)
          if (key[b] < key[a] || (key[b] == key[a] && key2[b] < key2[a])) {
            float t = key[a]; key[a] = key[b]; key[b] = t; t = key2[a]; key2[a] = key2[b]; key2[b] = t;
            uint32_t r = ref[a]; ref[a] = ref[b]; ref[b] = r;
          }
        for (int a = nh - 1; a >= 1; --a) { if (l->sp < 128) { l->stack[l->sp] = ref[a]; l->stack_t[l->sp] = key[a]; l->sp++; } }
        if (nh > 0) l->cur = ref[0]; else ws_pop(l, mode);
      }
    }
    /* phase 2: one leaf round */
    uint32_t maxcnt = 0; int anyleaf = 0;
    for (uint32_t i = 0; i < nlanes; ++i) {
      ws_lane* l = &L[i];
      if (l->done || (!(l->cur & 0x80000000u) && l->pend == 0)) continue;
      anyleaf = 1;
      l->lvisit++;
      const uint32_t leaf = l->pend ? l->pend : l->cur;
      const uint32_t first = leaf & 0x0FFFFFFFu, cnt = ((leaf >> 28) & 7u) + 1u;
      out[7] += cnt;   /* triangle records fetched by lanes (64 B each) */
      if (cnt > maxcnt) maxcnt = cnt;
      for (uint32_t k = 0; k < cnt; ++k) {
        const float* r = (const float*)(tris + 16u * (first + k));
        orc_tri T;
        T.v0 = v3(r[0], r[1], r[2]); T.e1 = v3(r[3], r[4], r[5]); T.e2 = v3(r[6], r[7], r[8]);
        T.Ng = v3(r[9], r[10], r[11]); T.n = v3(r[12], r[13], r[14]);
        const uint32_t f = tris[16u * (first + k) + 15u];
        float t;
        if (tri_intersect(&T, l->O, l->D, 0.0f, tfar, &t)) {
          if (!l->found || t < l->best_t || (t == l->best_t && f < l->best_f)) { l->best_t = t; l->best_f = f; l->found = 1; }
        }
      }
      if (l->pend) l->pend = 0; else ws_pop(l, mode);
    }
    if (anyleaf) { out[1]++; out[2] += maxcnt; est += (coop == 1) ? cl * maxcnt : cl; }
  }
  out[6] = (uint64_t)est;
  for (uint32_t i = 0; i < nlanes; ++i) {
    out[3] += L[i].nvisit; if (L[i].nvisit > out[4]) out[4] = L[i].nvisit; out[5] += L[i].lvisit;
    if (t_out) t_out[i] = L[i].found ? L[i].best_t : -1.0f;
    if (face_out) face_out[i] = (mode & 0x10000) ? (uint32_t)L[i].nvisit : L[i].best_f;   /* analysis: per-lane node visits */
  }
  free(L);
  return 0;
}


/* ------------------------------------------------------------------------- */
/* block-level model of a work-sharing find: `nw` waves (tiles of 64 rays) run side by side with their own clocks;   */
/* a wave that has finished takes half of the still-walking rays of the wave that holds the most (state moves with  */
/* the ray: cur, stack, best hit), at a cost for both.  Analysis only (tools/blocksim.py): what could intra-CU ray    */
/* stealing win over the static tile -> wave assignment?  costs[]: node iteration with > 16 / <= 16 rays, leaf round  */
/* with > 16 / <= 16 rays, thief cost, victim cost; min_victim: rays a victim must hold.  out[0] = makespan without,  */
/* out[1] = with stealing, out[2] = steals.                                                                           */
/* ------------------------------------------------------------------------- */
static void bs_init_lane(ws_lane* L, const float* O, const float* D, float tfar)
{
  memset(L, 0, sizeof(*L));
  L->O = v3(O[0], O[1], O[2]); L->D = v3(D[0], D[1], D[2]);
  L->o[0] = L->O.x; L->o[1] = L->O.y; L->o[2] = L->O.z;
  L->inv[0] = safe_inv(L->D.x); L->inv[1] = safe_inv(L->D.y); L->inv[2] = safe_inv(L->D.z);
  L->best_t = tfar; L->best_f = 0xFFFFFFFFu;
  L->done = !(L->D.x == L->D.x && L->D.y == L->D.y && L->D.z == L->D.z);
}
/* one unit of work of a wave: a node iteration if any lane holds an inner node, else a leaf round; returns 0 node, 1 leaf, -1 idle */
static int bs_unit(ws_lane* L, uint32_t n, const uint32_t* nodes, const uint32_t* tris, float tfar, uint32_t* active_out)
{
  uint32_t active = 0; int inner = 0;
  for (uint32_t i = 0; i < n; ++i) { if (!L[i].done) { active++; inner |= !(L[i].cur & 0x80000000u); } }
  *active_out = active;
  if (!active) return -1;
  if (inner) {
    for (uint32_t i = 0; i < n; ++i) {
      ws_lane* l = &L[i];
      if (l->done || (l->cur & 0x80000000u)) continue;
      const float* nd = (const float*)(nodes + 32u * l->cur);
      const uint32_t* ch = nodes + 32u * l->cur + 24u;
      float key[4]; uint32_t ref[4]; int nh = 0;
      for (uint32_t c = 0; c < 4; ++c) { float tn, tf; if (ws_box(nd, c, l->o, l->inv, l->best_t, &tn, &tf)) { key[nh] = tn; ref[nh] = ch[c]; nh++; } }
      for (int a = 0; a < nh; ++a) for (int b = a + 1; b < nh; ++b) if (key[b] < key[a]) { float t = key[a]; key[a] = key[b]; key[b] = t; uint32_t r = ref[a]; ref[a] = ref[b]; ref[b] = r; }
      for (int a = nh - 1; a >= 1; --a) { if (l->sp < 128) { l->stack[l->sp] = ref[a]; l->stack_t[l->sp] = key[a]; l->sp++; } }
      if (nh > 0) l->cur = ref[0]; else ws_pop(l, 0);
    }
    return 0;
  }
  for (uint32_t i = 0; i < n; ++i) {
    ws_lane* l = &L[i];
    if (l->done) continue;
    const uint32_t first = l->cur & 0x0FFFFFFFu, cnt = ((l->cur >> 28) & 7u) + 1u;
    for (uint32_t k = 0; k < cnt; ++k) {
      const float* r = (const float*)(tris + 16u * (first + k));
      orc_tri T;
      T.v0 = v3(r[0], r[1], r[2]); T.e1 = v3(r[3], r[4], r[5]); T.e2 = v3(r[6], r[7], r[8]);
      T.Ng = v3(r[9], r[10], r[11]); T.n = v3(r[12], r[13], r[14]);
      const uint32_t f = tris[16u * (first + k) + 15u];
      float t;
      if (tri_intersect(&T, l->O, l->D, 0.0f, tfar, &t)) {
        if (!l->found || t < l->best_t || (t == l->best_t && f < l->best_f)) { l->best_t = t; l->best_f = f; l->found = 1; }
      }
    }
    ws_pop(l, 0);
  }
  return 1;
}
int orc_blocksim(const uint32_t* nodes, const uint32_t* tris, const float* O, const float* D, uint32_t nw, float tfar,
                 const double* costs, uint32_t min_victim, double out[3])
{
  if (nw == 0 || nw > 16) return -1;
  out[0] = out[1] = out[2] = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    ws_lane* L = (ws_lane*)calloc((size_t)nw * 64u, sizeof(ws_lane));
    uint32_t cnt[16]; double clk[16]; int finished[16];
    for (uint32_t w = 0; w < nw; ++w) {
      cnt[w] = 64; clk[w] = 0.0; finished[w] = 0;
      for (uint32_t i = 0; i < 64; ++i) bs_init_lane(&L[w * 64u + i], O + 3u * (w * 64u + i), D + 3u * (w * 64u + i), tfar);
    }
    for (;;) {
      int w = -1;
      for (uint32_t k = 0; k < nw; ++k) if (!finished[k] && (w < 0 || clk[k] < clk[w])) w = (int)k;
      if (w < 0) break;
      uint32_t active;
      const int u = bs_unit(&L[(uint32_t)w * 64u], cnt[w], nodes, tris, tfar, &active);
      if (u >= 0) { clk[w] += (u == 0) ? (active > 16 ? costs[0] : costs[1]) : (active > 16 ? costs[2] : costs[3]); continue; }
      /* idle: steal (second pass only) */
      int v = -1; uint32_t vact = 0;
      if (pass == 1) {
        for (uint32_t k = 0; k < nw; ++k) {
          if ((int)k == w || finished[k]) continue;
          uint32_t a = 0;
          for (uint32_t i = 0; i < cnt[k]; ++i) a += L[k * 64u + i].done ? 0u : 1u;
          if (a >= min_victim && a > vact) { vact = a; v = (int)k; }
        }
      }
      if (v < 0) { finished[w] = 1; continue; }
      /* every other walking ray of the victim moves to the thief */
      uint32_t moved = 0, seen = 0;
      for (uint32_t i = 0; i < cnt[v]; ++i) {
        ws_lane* l = &L[(uint32_t)v * 64u + i];
        if (l->done) continue;
        if ((seen++ & 1u) == 0u) continue;
        L[(uint32_t)w * 64u + moved] = *l;
        l->done = 1;
        moved++;
      }
      cnt[w] = moved;
      const double t0 = clk[w] > clk[v] ? clk[w] : clk[v];
      clk[w] = t0 + costs[4];
      clk[v] += costs[5];
      out[2] += 1.0;
    }
    double mk = 0.0;
    for (uint32_t k = 0; k < nw; ++k) if (clk[k] > mk) mk = clk[k];
    out[pass] = mk;
    free(L);
  }
  return 0;
}

