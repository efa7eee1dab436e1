/* pfsim.c -- ANALYSIS ONLY (not the parity oracle, not the product): wave-level cost model of the particle-filter
 * beam-evaluation kernel (rmcl_amd/csrc/kernels.hip: k_pf_update_persist and the round-3 multi-slot kernel) on the
 * product's own exported BVH4 arrays.  It answers, without a GPU, what a scheduling policy does to the two quantities
 * the PMC profile shows the kernel is bound by: VALU instruction issues per block and the fraction of lanes active in
 * them.  One block = `nrays` rays shared by 4 waves of 64 lanes through one queue; the waves run on their own clocks
 * (the wave with the smallest clock advances), every wave-level step costs a fixed number of instruction issues
 * (pfsim_costs) whatever the number of active lanes -- that is what "issue bound" means.
 *
 * policies:
 *   slots = 1 : the round-2 kernel.  A lane holds one ray; refill when >= refill_thr lanes of the wave are idle (or
 *               nobody is busy); node phase left early when <= tail_lanes lanes still descend and a lane holds a leaf;
 *               leaf phase = loop over the longest leaf of the wave.
 *   slots = K : every lane owns K ray slots (state in LDS columns).  Each wave-level step is ONE of {node step, leaf
 *               step, refill}; the wave takes the kind most lanes can take part in (a lane takes part with any one of
 *               its slots that needs that kind); refill is taken when >= refill_thr lanes have an empty slot or
 *               nothing else can run.
 * build: see tools/Makefile (gcc -O2 -shared).  driver: tools/pfsim.py
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LEAF 0x80000000u
#define DONE 0x7FFFFFFFu
#define MAXSLOT 4

typedef struct {
  double node, tri, leaf_fixed, refill_eval, refill_setup, loop, slot_ld_node, slot_ld_leaf;
} pfsim_cost;

static pfsim_cost g_cost = {125, 75, 12, 110, 200, 14, 14, 14};

void pfsim_costs(const double* c) { memcpy(&g_cost, c, sizeof(g_cost)); }
static int g_lds_rows = 20;
void pfsim_lds_rows(int r) { g_lds_rows = r; }

typedef struct {
  uint32_t cur;          /* DONE: no ray / finished */
  int has_ray;
  float o[3], d[3], inv[3], best_t;
  uint32_t stack[96]; int sp;
  uint32_t nvisit, lvisit;
} slot_t;

static float safe_inv(float d) { float ad = fabsf(d); float s = (ad < 1e-30f) ? copysignf(1e-30f, d) : d; return 1.0f / s; }

static void slot_start(slot_t* s, const float* O, const float* D, float tfar)
{
  memset(s, 0, sizeof(*s));
  for (int k = 0; k < 3; ++k) { s->o[k] = O[k]; s->d[k] = D[k]; s->inv[k] = safe_inv(D[k]); }
  s->best_t = tfar; s->has_ray = 1; s->cur = 0; s->sp = 0;
}

static void slot_pop(slot_t* s) { if (s->sp > 0) s->cur = s->stack[--s->sp]; else s->cur = DONE; }

static int g_tie_mode = 0;   /* 0: order children by entry distance alone; 1: ties at 0 (origin inside the box) by FARTHEST exit first; 2: nearest exit first */
void pfsim_tie_mode(int m) { g_tie_mode = m; }
static float g_last_tf;
static int box(const float* nd, uint32_t c, const slot_t* s, float* tn_out)
{
  float tn = 0.0f, tf = s->best_t;
  for (int k = 0; k < 3; ++k) {
    const float lo = nd[8 * k + c], hi = nd[8 * k + 4 + c];
    float t0 = (lo - s->o[k]) * s->inv[k], t1 = (hi - s->o[k]) * s->inv[k];
    if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
    if (t0 > tn) tn = t0;
    if (t1 < tf) tf = t1;
  }
  *tn_out = tn;
  g_last_tf = tf;
  return nd[c] < 1e29f && tn <= tf;
}

static int g_sort_steps = 0;   /* 0: full order (insertion sort over the hits); 3 / 4 / 5: the kernel's compare-exchange network cut after that many steps */
void pfsim_sort_steps(int n) { g_sort_steps = n; }
static void node_step_network(slot_t* s, const uint32_t* nodes)
{
  const float* nd = (const float*)(nodes + 32u * s->cur);
  const uint32_t* ch = nodes + 32u * s->cur + 24u;
  float key[4]; uint32_t ref[4];
  s->nvisit++;
  for (uint32_t c = 0; c < 4; ++c) { float tn; key[c] = box(nd, c, s, &tn) ? tn : 3e38f; ref[c] = ch[c]; }
#define CSW(i, j) if (key[j] < key[i]) { float t = key[i]; key[i] = key[j]; key[j] = t; uint32_t r = ref[i]; ref[i] = ref[j]; ref[j] = r; }
  CSW(0, 1) CSW(2, 3) CSW(0, 2)
  if (g_sort_steps >= 4) CSW(1, 3)
  if (g_sort_steps >= 5) CSW(1, 2)
#undef CSW
  for (int a = 3; a >= 1; --a) if (key[a] < 3e38f && s->sp < 96) s->stack[s->sp++] = ref[a];
  if (key[0] < 3e38f) s->cur = ref[0]; else slot_pop(s);
}

static void node_step(slot_t* s, const uint32_t* nodes)
{
  if (g_sort_steps) { node_step_network(s, nodes); return; }
  const float* nd = (const float*)(nodes + 32u * s->cur);
  const uint32_t* ch = nodes + 32u * s->cur + 24u;
  float key[4]; uint32_t ref[4]; int nh = 0;
  s->nvisit++;
  for (uint32_t c = 0; c < 4; ++c) { float tn; if (box(nd, c, s, &tn)) { key[nh] = tn; if (g_tie_mode && tn == 0.0f) key[nh] = (g_tie_mode == 1) ? -g_last_tf * 1e-20f : -1e-20f / (g_last_tf + 1e-20f); ref[nh] = ch[c]; nh++; } }
  for (int a = 0; a < nh; ++a) for (int b = a + 1; b < nh; ++b)
    if (key[b] < key[a]) { float t = key[a]; key[a] = key[b]; key[b] = t; uint32_t r = ref[a]; ref[a] = ref[b]; ref[b] = r; }
  for (int a = nh - 1; a >= 1; --a) if (s->sp < 96) s->stack[s->sp++] = ref[a];
  if (nh > 0) s->cur = ref[0]; else slot_pop(s);
}

static void tri_test(slot_t* s, const uint32_t* tris, uint32_t rec)
{
  const float* r = (const float*)(tris + 16u * rec);
  const float* v0 = r; const float* e1 = r + 3; const float* e2 = r + 6; const float* Ng = r + 9;
  float C[3] = {v0[0] - s->o[0], v0[1] - s->o[1], v0[2] - s->o[2]};
  float R[3] = {C[1] * s->d[2] - C[2] * s->d[1], C[2] * s->d[0] - C[0] * s->d[2], C[0] * s->d[1] - C[1] * s->d[0]};
  float den = Ng[0] * s->d[0] + Ng[1] * s->d[1] + Ng[2] * s->d[2];
  float U = R[0] * e2[0] + R[1] * e2[1] + R[2] * e2[2];
  float V = R[0] * e1[0] + R[1] * e1[1] + R[2] * e1[2];
  float T = Ng[0] * C[0] + Ng[1] * C[1] + Ng[2] * C[2];
  if (den < 0) { U = -U; V = -V; T = -T; }
  const float aden = fabsf(den);
  if (den != 0.0f && U >= 0 && V >= 0 && U + V <= aden && T > 0) { const float t = T / aden; if (t < s->best_t) s->best_t = t; }
}

static uint32_t leaf_count(uint32_t cur) { return ((cur >> 28) & 7u) + 1u; }

static void leaf_step(slot_t* s, const uint32_t* tris)
{
  const uint32_t first = s->cur & 0x0FFFFFFFu, cnt = leaf_count(s->cur);
  s->lvisit++;
  for (uint32_t i = 0; i < cnt; ++i) tri_test(s, tris, first + i);
  slot_pop(s);
}

typedef struct {
  slot_t slot[64][MAXSLOT];
  int exhausted[64];
  double clk;
  int finished;
} wave_t;

typedef struct {
  double issue;         /* sum of wave-level instruction issues */
  double lane_issue;    /* sum of issues x active lanes / 64 */
  double node_issue, leaf_issue, refill_issue, loop_issue;
  double node_lane, leaf_lane, refill_lane;
  double makespan;      /* max wave clock of the block (issues) */
  double nvisit, lvisit, rays;
  double deep_steps, node_steps, max_sp;   /* wave-level node steps in which a lane's stack reaches beyond the LDS rows */
} pfsim_out;

/* O, D: nrays x 3 floats (queue order); returns stats of ONE block */
int pfsim_block(const uint32_t* nodes, const uint32_t* tris, const float* O, const float* D, uint32_t nrays, float tfar,
                int slots, int refill_thr, int tail_lanes, int nwaves, pfsim_out* out)
{
  if (slots < 1 || slots > MAXSLOT || nwaves < 1 || nwaves > 16) return -1;
  wave_t* W = (wave_t*)calloc((size_t)nwaves, sizeof(wave_t));
  uint32_t next = 0;
  for (int w = 0; w < nwaves; ++w)
    for (int l = 0; l < 64; ++l)
      for (int k = 0; k < MAXSLOT; ++k) W[w].slot[l][k].cur = DONE;
  memset(out, 0, sizeof(*out));
#define CHARGE(wv, what, cost, active) { (wv)->clk += (cost); out->issue += (cost); out->lane_issue += (cost) * (double)(active) / 64.0; \
                                         out->what##_issue += (cost); }
  for (;;) {
    int w = -1;
    for (int k = 0; k < nwaves; ++k) if (!W[k].finished && (w < 0 || W[k].clk < W[w].clk)) w = k;
    if (w < 0) break;
    wave_t* wv = &W[w];
    if (slots == 1) {
      /* ---- round-2 kernel: one outer iteration ---- */
      int want = 0, busy = 0;
      for (int l = 0; l < 64; ++l) {
        const slot_t* s = &wv->slot[l][0];
        if (s->cur == DONE && !wv->exhausted[l]) want++;
        if (s->cur != DONE) busy++;
      }
      if (want == 0 && busy == 0) { wv->finished = 1; continue; }
      wv->clk += g_cost.loop; out->issue += g_cost.loop; out->loop_issue += g_cost.loop; out->lane_issue += g_cost.loop * (busy + want) / 64.0;
      if (want != 0 && (busy == 0 || want >= refill_thr)) {
        int n_eval = 0, n_setup = 0;
        for (int l = 0; l < 64; ++l) {
          slot_t* s = &wv->slot[l][0];
          if (!(s->cur == DONE && !wv->exhausted[l])) continue;
          if (s->has_ray) { n_eval++; out->nvisit += s->nvisit; out->lvisit += s->lvisit; out->rays += 1; s->has_ray = 0; }
          if (next < nrays) { slot_start(s, O + 3u * next, D + 3u * next, tfar); next++; n_setup++; }
          else wv->exhausted[l] = 1;
        }
        CHARGE(wv, refill, g_cost.refill_eval, n_eval);
        out->refill_lane += g_cost.refill_eval * n_eval / 64.0;
        CHARGE(wv, refill, g_cost.refill_setup, n_setup);
        out->refill_lane += g_cost.refill_setup * n_setup / 64.0;
      }
      for (;;) {
        int inner = 0, holding = 0;
        for (int l = 0; l < 64; ++l) {
          const slot_t* s = &wv->slot[l][0];
          if (s->cur == DONE) continue;
          if (s->cur & LEAF) holding++; else inner++;
        }
        if (!inner) break;
        if (inner <= tail_lanes && holding) break;
        CHARGE(wv, node, g_cost.node, inner);
        out->node_lane += g_cost.node * inner / 64.0;
        {
          int deep = 0;
          for (int l = 0; l < 64; ++l) { const slot_t* s = &wv->slot[l][0]; if (s->cur != DONE && !(s->cur & LEAF)) { if (s->sp + 1 + 3 > g_lds_rows) deep = 1; if (s->sp > out->max_sp) out->max_sp = s->sp; } }
          out->deep_steps += deep; out->node_steps += 1;
        }
        for (int l = 0; l < 64; ++l) { slot_t* s = &wv->slot[l][0]; if (s->cur != DONE && !(s->cur & LEAF)) node_step(s, nodes); }
      }
      {
        uint32_t maxcnt = 0; int holding = 0; uint32_t hist[9] = {0};
        for (int l = 0; l < 64; ++l) {
          slot_t* s = &wv->slot[l][0];
          if (s->cur == DONE || !(s->cur & LEAF)) continue;
          const uint32_t c = leaf_count(s->cur);
          holding++; hist[c]++; if (c > maxcnt) maxcnt = c;
          leaf_step(s, tris);
        }
        if (holding) {
          CHARGE(wv, leaf, g_cost.leaf_fixed, holding);
          out->leaf_lane += g_cost.leaf_fixed * holding / 64.0;
          for (uint32_t i = 1; i <= maxcnt; ++i) {
            int act = 0; for (uint32_t c = i; c <= 8; ++c) act += hist[c];
            CHARGE(wv, leaf, g_cost.tri, act);
            out->leaf_lane += g_cost.tri * act / 64.0;
          }
        }
      }
    } else {
      /* ---- multi-slot kernel: one wave-level step of one kind ---- */
      int n_node = 0, n_leaf = 0, n_empty = 0, n_any = 0;
      for (int l = 0; l < 64; ++l) {
        int hn = 0, hl = 0, he = 0;
        for (int k = 0; k < slots; ++k) {
          const slot_t* s = &wv->slot[l][k];
          if (s->cur == DONE) { if (s->has_ray || !wv->exhausted[l]) he = 1; }
          else if (s->cur & LEAF) hl = 1; else hn = 1;
        }
        n_node += hn; n_leaf += hl; n_empty += he; n_any += (hn | hl);
      }
      if (n_node == 0 && n_leaf == 0 && n_empty == 0) { wv->finished = 1; continue; }
      wv->clk += g_cost.loop; out->issue += g_cost.loop; out->loop_issue += g_cost.loop; out->lane_issue += g_cost.loop;
      int kind;  /* 0 node, 1 leaf, 2 refill */
      if (n_empty > 0 && (n_empty >= refill_thr || (n_node == 0 && n_leaf == 0))) kind = 2;
      else if (n_node == 0 && n_leaf == 0) kind = 2;
      else {
        /* a leaf step is cheaper than a node step: take it when at least tail_lanes/64 of ... simple rule: the larger crowd,
         * leaves win ties; never run a kind with nobody */
        kind = (n_leaf * 100 >= n_node * tail_lanes) ? 1 : 0;
        if (kind == 1 && n_leaf == 0) kind = 0;
        if (kind == 0 && n_node == 0) kind = 1;
      }
      if (kind == 2) {
        int n_eval = 0, n_setup = 0;
        for (int l = 0; l < 64; ++l) {
          /* one slot per lane per refill step */
          for (int k = 0; k < slots; ++k) {
            slot_t* s = &wv->slot[l][k];
            if (s->cur != DONE) continue;
            if (!s->has_ray && wv->exhausted[l]) continue;
            if (s->has_ray) { n_eval++; out->nvisit += s->nvisit; out->lvisit += s->lvisit; out->rays += 1; s->has_ray = 0; }
            if (next < nrays) { slot_start(s, O + 3u * next, D + 3u * next, tfar); next++; n_setup++; }
            else wv->exhausted[l] = 1;
            break;
          }
        }
        CHARGE(wv, refill, g_cost.refill_eval, n_eval);
        out->refill_lane += g_cost.refill_eval * n_eval / 64.0;
        CHARGE(wv, refill, g_cost.refill_setup, n_setup);
        out->refill_lane += g_cost.refill_setup * n_setup / 64.0;
      } else if (kind == 0) {
        const double c = g_cost.node + g_cost.slot_ld_node;
        CHARGE(wv, node, c, n_node);
        out->node_lane += c * n_node / 64.0;
        for (int l = 0; l < 64; ++l)
          for (int k = 0; k < slots; ++k) { slot_t* s = &wv->slot[l][k]; if (s->cur != DONE && !(s->cur & LEAF)) { node_step(s, nodes); break; } }
      } else {
        uint32_t maxcnt = 0; uint32_t hist[9] = {0};
        for (int l = 0; l < 64; ++l)
          for (int k = 0; k < slots; ++k) {
            slot_t* s = &wv->slot[l][k];
            if (s->cur != DONE && (s->cur & LEAF)) { const uint32_t c = leaf_count(s->cur); hist[c]++; if (c > maxcnt) maxcnt = c; leaf_step(s, tris); break; }
          }
        const double c = g_cost.leaf_fixed + g_cost.slot_ld_leaf;
        CHARGE(wv, leaf, c, n_leaf);
        out->leaf_lane += c * n_leaf / 64.0;
        for (uint32_t i = 1; i <= maxcnt; ++i) {
          int act = 0; for (uint32_t cc = i; cc <= 8; ++cc) act += hist[cc];
          CHARGE(wv, leaf, g_cost.tri, act);
          out->leaf_lane += g_cost.tri * act / 64.0;
        }
      }
    }
  }
  for (int k = 0; k < nwaves; ++k) if (W[k].clk > out->makespan) out->makespan = W[k].clk;
  free(W);
  return 0;
#undef CHARGE
}
