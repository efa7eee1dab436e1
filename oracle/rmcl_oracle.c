/*
 * rmcl_oracle.c -- CPU restatement of the RMCL / MICP-L hot path.
 * TEST INFRASTRUCTURE ONLY (see rmcl_oracle.h).  PARITY UNPINNED (see header).
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -fPIC -shared -pthread (oracle/Makefile).
 * -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
 */
#define _GNU_SOURCE
#include "rmcl_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------------- */
/* vector helpers with a fixed operation order                                */
/* ------------------------------------------------------------------------- */

static inline orc_vec3 v3(float x, float y, float z) { orc_vec3 r = {x, y, z}; return r; }
static inline orc_vec3 v_add(orc_vec3 a, orc_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline orc_vec3 v_sub(orc_vec3 a, orc_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline orc_vec3 v_scale(orc_vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
/* rmagine Vector3::dot: plain left-to-right sum of products */
static inline float v_dot_plain(orc_vec3 a, orc_vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* Embree's dot/cross on FMA hardware: madd(a.x,b.x,madd(a.y,b.y,a.z*b.z)), msub(...)
 * (embree4 common/math/vec3.h, kernels/geometry/triangle_intersector_moeller.h) */
static inline float v_dot_fma(orc_vec3 a, orc_vec3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static inline orc_vec3 v_cross_fma(orc_vec3 a, orc_vec3 b)
{
  return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

/* ------------------------------------------------------------------------- */
/* rmagine Quaternion / Transform (external; recollected upstream semantics,  */
/* SURVEY.md Appendix A): Hamilton product, q*p = q (p,0) q^-1, T*v = R v + t */
/* ------------------------------------------------------------------------- */

orc_quat orc_quat_mult(orc_quat a, orc_quat b)
{
  orc_quat r;
  r.w = ((a.w * b.w - a.x * b.x) - a.y * b.y) - a.z * b.z;
  r.x = ((a.w * b.x + a.x * b.w) + a.y * b.z) - a.z * b.y;
  r.y = ((a.w * b.y - a.x * b.z) + a.y * b.w) + a.z * b.x;
  r.z = ((a.w * b.z + a.x * b.y) - a.y * b.x) + a.z * b.w;
  return r;
}

orc_quat orc_quat_inv(orc_quat q)
{
  orc_quat r = {-q.x, -q.y, -q.z, q.w};
  return r;
}

orc_vec3 orc_quat_rotate(orc_quat q, orc_vec3 p)
{
  const orc_quat P = {p.x, p.y, p.z, 0.0f};
  const orc_quat PT = orc_quat_mult(orc_quat_mult(q, P), orc_quat_inv(q));
  return v3(PT.x, PT.y, PT.z);
}

orc_vec3 orc_transform_apply(orc_transform T, orc_vec3 p)
{
  return v_add(orc_quat_rotate(T.R, p), T.t);
}

orc_transform orc_transform_mult(orc_transform a, orc_transform b)
{
  /* P' = R1 (R2 P + t2) + t1 */
  orc_transform r;
  r.t = v_add(orc_quat_rotate(a.R, b.t), a.t);
  r.R = orc_quat_mult(a.R, b.R);
  r.stamp = a.stamp;
  return r;
}

orc_transform orc_transform_inv(orc_transform a)
{
  orc_transform r;
  r.R = orc_quat_inv(a.R);
  const orc_vec3 rt = orc_quat_rotate(r.R, a.t);
  r.t = v3(-rt.x, -rt.y, -rt.z);
  r.stamp = a.stamp;
  return r;
}

orc_transform orc_transform_identity(void)
{
  orc_transform r;
  r.R.x = 0; r.R.y = 0; r.R.z = 0; r.R.w = 1;
  r.t = v3(0, 0, 0);
  r.stamp = 0;
  return r;
}

/* rmagine EulerAngles -> Quaternion (ZYX, used by rmcl_localization.cpp:315) */
orc_quat orc_euler_to_quat(float roll, float pitch, float yaw)
{
  const float cr = cosf(roll / 2.0f), sr = sinf(roll / 2.0f);
  const float cp = cosf(pitch / 2.0f), sp = sinf(pitch / 2.0f);
  const float cy = cosf(yaw / 2.0f), sy = sinf(yaw / 2.0f);
  orc_quat q;
  q.w = cr * cp * cy + sr * sp * sy;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  return q;
}

/* ------------------------------------------------------------------------- */
/* mesh: triangle records + the oracle's own BVH2                             */
/* ------------------------------------------------------------------------- */

typedef struct {
  orc_vec3 v0, e1, e2, Ng, n; /* e1 = v0-v1, e2 = v2-v0, Ng = cross(e2,e1) (Embree Triangle4) */
} orc_tri;

typedef struct {
  float bmin[3];
  float bmax[3];
  uint32_t left_first; /* inner: index of left child (right = left+1); leaf: first prim */
  uint32_t count;      /* 0 = inner */
} orc_node;            /* 32 B: the "reference tree" of SURVEY.md 8(d) */

struct orc_mesh {
  uint32_t nf;
  orc_tri* tris;       /* indexed by ORIGINAL face id */
  uint32_t* prim;      /* BVH leaf order -> face id */
  orc_node* nodes;
  uint32_t n_nodes;
  uint32_t max_leaf;
  float pad;
  struct orc_node4* nodes4;   /* 4-wide collapse of `nodes` for the SSE walk (use_bvh = 2: the CPU-baseline path), built on first use */
  uint32_t n_nodes4;
  int bvh4_ready;             /* set (release) once nodes4 is complete */
};

static void tri_setup(orc_tri* T, orc_vec3 a, orc_vec3 b, orc_vec3 c)
{
  T->v0 = a;
  T->e1 = v_sub(a, b);
  T->e2 = v_sub(c, a);
  T->Ng = v_cross_fma(T->e2, T->e1);
  /* rmagine normalizeInplace: d = sqrt(x*x+y*y+z*z); x/=d ... */
  const float d = sqrtf((T->Ng.x * T->Ng.x + T->Ng.y * T->Ng.y) + T->Ng.z * T->Ng.z);
  if (d > 0.0f) T->n = v3(T->Ng.x / d, T->Ng.y / d, T->Ng.z / d);
  else T->n = v3(0, 0, 0);
}

/* Moeller-Trumbore in Embree's formulation (triangle_intersector_moeller.h):
 * scaled barycentrics, sign-adjusted, rejection tests before the single division.
 * Two-sided (no back-face culling).  Returns 1 and t on a hit in (tnear,tfar]: Embree's depth test is strict on
 * the near side (absDen * tnear < T), so a ray that starts exactly on a triangle does not hit it. */
static inline int tri_intersect(const orc_tri* T, orc_vec3 O, orc_vec3 D, float tnear, float tfar, float* t_out)
{
  const orc_vec3 C = v_sub(T->v0, O);
  const orc_vec3 R = v_cross_fma(C, D);
  const float den = v_dot_fma(T->Ng, D);
  const float aden = fabsf(den);
  float U = v_dot_fma(R, T->e2);
  float V = v_dot_fma(R, T->e1);
  float Tt = v_dot_fma(T->Ng, C);
  if (den < 0.0f) { U = -U; V = -V; Tt = -Tt; }
  if (!(den != 0.0f)) return 0;
  if (!(U >= 0.0f && V >= 0.0f && (U + V) <= aden)) return 0;
  const float t = Tt / aden;
  if (!(aden * tnear < Tt && t <= tfar)) return 0;
  *t_out = t;
  return 1;
}

/* --- BVH2 builder: binned SAH, top-down ---------------------------------- */

typedef struct { float bmin[3], bmax[3], c[3]; } prim_info;

typedef struct {
  orc_mesh* m;
  prim_info* info; /* indexed by face id */
  uint32_t n_alloc;
} build_ctx;

static inline void box_init(float* bmin, float* bmax)
{
  for (int k = 0; k < 3; ++k) { bmin[k] = FLT_MAX; bmax[k] = -FLT_MAX; }
}
static inline void box_grow(float* bmin, float* bmax, const float* omin, const float* omax)
{
  for (int k = 0; k < 3; ++k) { if (omin[k] < bmin[k]) bmin[k] = omin[k]; if (omax[k] > bmax[k]) bmax[k] = omax[k]; }
}
static inline float box_area(const float* bmin, const float* bmax)
{
  const float dx = bmax[0] - bmin[0], dy = bmax[1] - bmin[1], dz = bmax[2] - bmin[2];
  if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}

#define ORC_BINS 16

static void build_rec(build_ctx* bc, uint32_t node_id, uint32_t first, uint32_t count)
{
  orc_mesh* m = bc->m;
  orc_node* node = &m->nodes[node_id];
  float cmin[3], cmax[3];
  box_init(node->bmin, node->bmax);
  box_init(cmin, cmax);
  for (uint32_t i = first; i < first + count; ++i) {
    const prim_info* p = &bc->info[m->prim[i]];
    box_grow(node->bmin, node->bmax, p->bmin, p->bmax);
    box_grow(cmin, cmax, p->c, p->c);
  }
  for (int k = 0; k < 3; ++k) { node->bmin[k] -= m->pad; node->bmax[k] += m->pad; }
  if (count <= m->max_leaf) {
    node->left_first = first; node->count = count;
    return;
  }
  /* choose split */
  int best_axis = -1; int best_bin = -1; float best_cost = FLT_MAX;
  for (int axis = 0; axis < 3; ++axis) {
    const float ext = cmax[axis] - cmin[axis];
    if (!(ext > 0.0f)) continue;
    float bbmin[ORC_BINS][3], bbmax[ORC_BINS][3]; uint32_t bcnt[ORC_BINS];
    for (int b = 0; b < ORC_BINS; ++b) { box_init(bbmin[b], bbmax[b]); bcnt[b] = 0; }
    const float scale = (float)ORC_BINS / ext;
    for (uint32_t i = first; i < first + count; ++i) {
      const prim_info* p = &bc->info[m->prim[i]];
      int b = (int)((p->c[axis] - cmin[axis]) * scale);
      if (b >= ORC_BINS) b = ORC_BINS - 1;
      if (b < 0) b = 0;
      bcnt[b]++; box_grow(bbmin[b], bbmax[b], p->bmin, p->bmax);
    }
    float la[ORC_BINS - 1], ra[ORC_BINS - 1]; uint32_t lc[ORC_BINS - 1], rc[ORC_BINS - 1];
    float lmin[3], lmax[3], rmin[3], rmax[3]; uint32_t c = 0;
    box_init(lmin, lmax);
    for (int b = 0; b < ORC_BINS - 1; ++b) { box_grow(lmin, lmax, bbmin[b], bbmax[b]); c += bcnt[b]; la[b] = box_area(lmin, lmax); lc[b] = c; }
    box_init(rmin, rmax); c = 0;
    for (int b = ORC_BINS - 1; b > 0; --b) { box_grow(rmin, rmax, bbmin[b], bbmax[b]); c += bcnt[b]; ra[b - 1] = box_area(rmin, rmax); rc[b - 1] = c; }
    for (int b = 0; b < ORC_BINS - 1; ++b) {
      if (lc[b] == 0 || rc[b] == 0) continue;
      const float cost = la[b] * (float)lc[b] + ra[b] * (float)rc[b];
      if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
    }
  }
  uint32_t mid;
  if (best_axis < 0) {
    /* all centroids coincide: split in the middle */
    mid = first + count / 2;
  } else {
    const float ext = cmax[best_axis] - cmin[best_axis];
    const float scale = (float)ORC_BINS / ext;
    uint32_t i = first, j = first + count;
    while (i < j) {
      const prim_info* p = &bc->info[m->prim[i]];
      int b = (int)((p->c[best_axis] - cmin[best_axis]) * scale);
      if (b >= ORC_BINS) b = ORC_BINS - 1;
      if (b < 0) b = 0;
      if (b <= best_bin) { ++i; } else { --j; uint32_t tmp = m->prim[i]; m->prim[i] = m->prim[j]; m->prim[j] = tmp; }
    }
    mid = i;
    if (mid == first || mid == first + count) mid = first + count / 2;
  }
  const uint32_t left = m->n_nodes; m->n_nodes += 2;
  node->left_first = left; node->count = 0;
  build_rec(bc, left, first, mid - first);
  build_rec(bc, left + 1, mid, first + count - mid);
}

/* max_leaf bit 31 (ORC_MESH_NO_BVH): records only, no tree -- for maps of 10^7 faces that are only ever traced by brute force (the
 * recursive single-threaded build below would take longer than the check it serves); use_bvh != 0 on such a mesh is refused */
orc_mesh* orc_mesh_create(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, uint32_t max_leaf)
{
  const int no_bvh = (max_leaf & 0x80000000u) != 0;
  max_leaf &= 0x7FFFFFFFu;
  if (nf == 0 || max_leaf == 0) return NULL;
  orc_mesh* m = (orc_mesh*)calloc(1, sizeof(orc_mesh));
  m->nf = nf; m->max_leaf = max_leaf;
  m->tris = (orc_tri*)malloc(sizeof(orc_tri) * nf);
  m->prim = (uint32_t*)malloc(sizeof(uint32_t) * nf);
  m->nodes = (orc_node*)malloc(sizeof(orc_node) * (no_bvh ? 1 : 2 * (size_t)nf));
  prim_info* info = (prim_info*)malloc(sizeof(prim_info) * nf);
  float smin[3], smax[3]; box_init(smin, smax);
  for (uint32_t f = 0; f < nf; ++f) {
    orc_vec3 p[3];
    for (int k = 0; k < 3; ++k) {
      const uint32_t vi = faces[3 * f + k];
      if (vi >= nv) { free(info); orc_mesh_destroy(m); return NULL; }
      p[k] = v3(verts[3 * vi], verts[3 * vi + 1], verts[3 * vi + 2]);
    }
    tri_setup(&m->tris[f], p[0], p[1], p[2]);
    prim_info* pi = &info[f];
    box_init(pi->bmin, pi->bmax);
    for (int k = 0; k < 3; ++k) {
      const float q[3] = {p[k].x, p[k].y, p[k].z};
      box_grow(pi->bmin, pi->bmax, q, q);
    }
    for (int k = 0; k < 3; ++k) pi->c[k] = 0.5f * (pi->bmin[k] + pi->bmax[k]);
    box_grow(smin, smax, pi->bmin, pi->bmax);
    m->prim[f] = f;
  }
  float diag = 0.0f;
  for (int k = 0; k < 3; ++k) { const float d = smax[k] - smin[k]; if (d > diag) diag = d; }
  float amax = 0.0f;
  for (int k = 0; k < 3; ++k) { if (fabsf(smin[k]) > amax) amax = fabsf(smin[k]); if (fabsf(smax[k]) > amax) amax = fabsf(smax[k]); }
  m->pad = 1e-4f * (diag > amax ? diag : amax) + 1e-6f;
  if (no_bvh) { m->n_nodes = 0; free(info); return m; }
  build_ctx bc = {m, info, 0};
  m->n_nodes = 1;
  build_rec(&bc, 0, 0, nf);
  free(info);
  return m;
}

void orc_mesh_destroy(orc_mesh* m)
{
  if (!m) return;
  free(m->tris); free(m->prim); free(m->nodes); free(m->nodes4); free(m);
}

uint32_t orc_mesh_num_nodes(const orc_mesh* m) { return m->n_nodes; }

void orc_mesh_face_normals(const orc_mesh* m, float* out)
{
  for (uint32_t f = 0; f < m->nf; ++f) { out[3 * f] = m->tris[f].n.x; out[3 * f + 1] = m->tris[f].n.y; out[3 * f + 2] = m->tris[f].n.z; }
}

/* closest hit with the documented deterministic tie-break: min t, then min face id.
 * This brute-force loop is the bit-exact face-id authority. */
int orc_intersect_brute(const orc_mesh* m, orc_vec3 O, orc_vec3 D, float tnear, float tfar, float* t_out, uint32_t* face_out)
{
  float best_t = 0.0f; uint32_t best_f = 0xFFFFFFFFu; int found = 0;
  for (uint32_t f = 0; f < m->nf; ++f) {
    float t;
    if (tri_intersect(&m->tris[f], O, D, tnear, tfar, &t)) {
      if (!found || t < best_t || (t == best_t && f < best_f)) { best_t = t; best_f = f; found = 1; }
    }
  }
  if (found) { *t_out = best_t; *face_out = best_f; }
  return found;
}

static inline float safe_inv(float d)
{
  if (fabsf(d) < 1e-30f) d = copysignf(1e-30f, d);
  return 1.0f / d;
}

static inline int box_hit(const orc_node* n, const float* o, const float* inv, float tnear, float tfar, float* tn_out)
{
  float tn = tnear, tf = tfar;
  for (int k = 0; k < 3; ++k) {
    float t0 = (n->bmin[k] - o[k]) * inv[k];
    float t1 = (n->bmax[k] - o[k]) * inv[k];
    if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 > tn) tn = t0;
    if (t1 < tf) tf = t1;
  }
  *tn_out = tn;
  return tn <= tf * 1.0000004f;
}

int orc_intersect_bvh(const orc_mesh* m, orc_vec3 O, orc_vec3 D, float tnear, float tfar, float* t_out, uint32_t* face_out, orc_counters* cnt)
{
  if (m->n_nodes == 0) return orc_intersect_brute(m, O, D, tnear, tfar, t_out, face_out);   /* ORC_MESH_NO_BVH: same result, no tree */
  const float o[3] = {O.x, O.y, O.z};
  const float inv[3] = {safe_inv(D.x), safe_inv(D.y), safe_inv(D.z)};
  float best_t = tfar; uint32_t best_f = 0xFFFFFFFFu; int found = 0;
  uint32_t stack[128]; int sp = 0;
  uint64_t nv = 0, nt = 0;
  float tn;
  if (!(D.x == D.x && D.y == D.y && D.z == D.z)) goto done; /* NaN direction: miss */
  nv++;
  if (!box_hit(&m->nodes[0], o, inv, tnear, best_t, &tn)) goto done;
  stack[sp++] = 0;
  while (sp > 0) {
    const orc_node* n = &m->nodes[stack[--sp]];
    if (n->count > 0) {
      for (uint32_t i = 0; i < n->count; ++i) {
        const uint32_t f = m->prim[n->left_first + i];
        float t; nt++;
        if (tri_intersect(&m->tris[f], O, D, tnear, tfar, &t)) {
          if (!found || t < best_t || (t == best_t && f < best_f)) { best_t = t; best_f = f; found = 1; }
        }
      }
      continue;
    }
    const uint32_t l = n->left_first, r = l + 1;
    float tl, tr;
    nv += 2;
    const int hl = box_hit(&m->nodes[l], o, inv, tnear, best_t, &tl);
    const int hr = box_hit(&m->nodes[r], o, inv, tnear, best_t, &tr);
    if (hl && hr) {
      if (sp + 2 > 128) return -1;
      if (tl <= tr) { stack[sp++] = r; stack[sp++] = l; } else { stack[sp++] = l; stack[sp++] = r; }
    } else if (hl) { stack[sp++] = l; }
    else if (hr) { stack[sp++] = r; }
  }
done:
  if (cnt) { cnt->nodes_visited += nv; cnt->tris_tested += nt; cnt->rays += 1; }
  if (found) { *t_out = best_t; *face_out = best_f; }
  return found;
}

/* ------------------------------------------------------------------------- */
/* use_bvh = 2: the same closest hit through a 4-wide collapse of the BVH2 with the four children's slab tests in one  */
/* SSE register -- what a CPU ray caster of Embree's class does per ray (bvh4 + single-ray traversal), so that the      */
/* `cpu_baseline` of bench.py is not a strawman.  Same triangle test, same (min t, min face id) rule: the result is    */
/* the BVH2 walk's and the brute-force loop's bit for bit (tests/test_oracle.py).  Box test per child = box_hit's       */
/* arithmetic, lane-wise.                                                                                               */
/* ------------------------------------------------------------------------- */
#include <immintrin.h>
typedef struct orc_node4 {
  float lo[3][4], hi[3][4];
  int32_t child[4];   /* >= 0: orc_node4 index; < 0: ~(index of a BVH2 LEAF node); empty slot: inverted box, child = INT32_MIN */
} orc_node4;

static float n2_area(const orc_node* n) { return box_area(n->bmin, n->bmax); }

static uint32_t build_node4(orc_mesh* m, uint32_t n2)
{
  /* collapse: start with the two children of BVH2 node n2, open the inner candidate of largest area until four */
  uint32_t cand[4]; int nc = 0;
  cand[nc++] = m->nodes[n2].left_first; cand[nc++] = m->nodes[n2].left_first + 1;
  while (nc < 4) {
    int best = -1; float ba = -1.0f;
    for (int i = 0; i < nc; ++i)
      if (m->nodes[cand[i]].count == 0) { const float a = n2_area(&m->nodes[cand[i]]); if (a > ba) { ba = a; best = i; } }
    if (best < 0) break;
    const uint32_t l = m->nodes[cand[best]].left_first;
    cand[best] = l; cand[nc++] = l + 1;
  }
  const uint32_t me = m->n_nodes4++;
  for (int i = 0; i < 4; ++i) {
    orc_node4* N = &m->nodes4[me];
    if (i < nc) {
      const orc_node* c = &m->nodes[cand[i]];
      for (int k = 0; k < 3; ++k) { N->lo[k][i] = c->bmin[k]; N->hi[k][i] = c->bmax[k]; }
    } else {
      for (int k = 0; k < 3; ++k) { N->lo[k][i] = 1.0f; N->hi[k][i] = -1.0f; }   /* never entered */
      N->child[i] = INT32_MIN;
    }
  }
  for (int i = 0; i < nc; ++i) {
    const int32_t ref = (m->nodes[cand[i]].count > 0) ? ~(int32_t)cand[i] : (int32_t)build_node4(m, cand[i]);
    m->nodes4[me].child[i] = ref;   /* (re-index: nodes4 may have moved? no -- allocated once, below) */
  }
  return me;
}

static pthread_mutex_t g_bvh4_mtx = PTHREAD_MUTEX_INITIALIZER;
static void ensure_bvh4(orc_mesh* m)
{
  pthread_mutex_lock(&g_bvh4_mtx);
  if (!__atomic_load_n(&m->bvh4_ready, __ATOMIC_ACQUIRE)) {
    m->nodes4 = (orc_node4*)malloc(sizeof(orc_node4) * (size_t)(m->n_nodes ? m->n_nodes : 1));   /* <= one per BVH2 inner node */
    m->n_nodes4 = 0;
    if (m->nodes[0].count == 0) build_node4(m, 0);
    __atomic_store_n(&m->bvh4_ready, 1, __ATOMIC_RELEASE);
  }
  pthread_mutex_unlock(&g_bvh4_mtx);
}

int orc_intersect_bvh4(const orc_mesh* mc, orc_vec3 O, orc_vec3 D, float tnear, float tfar, float* t_out, uint32_t* face_out)
{
  if (mc->n_nodes == 0) return orc_intersect_brute(mc, O, D, tnear, tfar, t_out, face_out);   /* ORC_MESH_NO_BVH */
  orc_mesh* m = (orc_mesh*)mc;
  if (!__atomic_load_n(&m->bvh4_ready, __ATOMIC_ACQUIRE)) ensure_bvh4(m);
  if (m->nodes[0].count > 0) return orc_intersect_bvh(mc, O, D, tnear, tfar, t_out, face_out, NULL);   /* a single leaf */
  if (!(D.x == D.x && D.y == D.y && D.z == D.z)) return 0;
  const float inv3[3] = {safe_inv(D.x), safe_inv(D.y), safe_inv(D.z)};
  const __m128 ox = _mm_set1_ps(O.x), oy = _mm_set1_ps(O.y), oz = _mm_set1_ps(O.z);
  const __m128 ix = _mm_set1_ps(inv3[0]), iy = _mm_set1_ps(inv3[1]), iz = _mm_set1_ps(inv3[2]);
  const __m128 vnear = _mm_set1_ps(tnear), slack = _mm_set1_ps(1.0000004f);
  float best_t = tfar; uint32_t best_f = 0xFFFFFFFFu; int found = 0;
  int32_t stack[256]; float stack_t[256]; int sp = 0;
  stack[sp] = 0; stack_t[sp] = tnear; ++sp;
  while (sp > 0) {
    --sp;
    const int32_t ref = stack[sp];
    if (stack_t[sp] > best_t * 1.0000004f) continue;
    if (ref < 0) {
      const orc_node* n = &m->nodes[~ref];
      for (uint32_t i = 0; i < n->count; ++i) {
        const uint32_t f = m->prim[n->left_first + i];
        float t;
        if (tri_intersect(&m->tris[f], O, D, tnear, tfar, &t)) {
          if (!found || t < best_t || (t == best_t && f < best_f)) { best_t = t; best_f = f; found = 1; }
        }
      }
      continue;
    }
    const orc_node4* N = &m->nodes4[ref];
    /* box_hit, four children at once: t0 = (bmin - o) * inv, t1 = (bmax - o) * inv, ordered, intersected with [tnear, best_t] */
    const __m128 ax = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(N->lo[0]), ox), ix), bx = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(N->hi[0]), ox), ix);
    const __m128 ay = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(N->lo[1]), oy), iy), by = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(N->hi[1]), oy), iy);
    const __m128 az = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(N->lo[2]), oz), iz), bz = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(N->hi[2]), oz), iz);
    const __m128 tn = _mm_max_ps(_mm_max_ps(_mm_min_ps(ax, bx), _mm_min_ps(ay, by)), _mm_max_ps(_mm_min_ps(az, bz), vnear));
    const __m128 tf = _mm_min_ps(_mm_min_ps(_mm_max_ps(ax, bx), _mm_max_ps(ay, by)), _mm_min_ps(_mm_max_ps(az, bz), _mm_set1_ps(best_t)));
    int mask = _mm_movemask_ps(_mm_cmple_ps(tn, _mm_mul_ps(tf, slack)));
    if (!mask) continue;
    float tns[4];
    _mm_storeu_ps(tns, tn);
    /* push the hit children far to near (insertion into a sorted run of <= 4) */
    int order[4], no = 0;
    while (mask) {
      const int c = __builtin_ctz((unsigned)mask);
      mask &= mask - 1;
      if (N->child[c] == INT32_MIN) continue;
      int pos = no++;
      while (pos > 0 && tns[order[pos - 1]] < tns[c]) { order[pos] = order[pos - 1]; --pos; }
      order[pos] = c;
    }
    if (sp + no > 256) return -1;
    for (int i = 0; i < no; ++i) { stack[sp] = N->child[order[i]]; stack_t[sp] = tns[order[i]]; ++sp; }   /* nearest on top */
  }
  if (found) { *t_out = best_t; *face_out = best_f; }
  return found;
}

/* rmagine PinholeModel::getDirection (external; fields f = {fx, fy}, c = {cx, cy} pinned by
 * rmcl_ros/src/util/conversions.cpp:36-60): optical ray ((hid - cx)/fx, (vid - cy)/fy, 1) normalised, then
 * optical (x right, y down, z forward) -> sensor frame (x forward, y left, z up). */
orc_vec3 orc_pinhole_direction(const float* f, const float* c, uint32_t vid, uint32_t hid)
{
  const float pX = ((float)hid - c[0]) / f[0];
  const float pY = ((float)vid - c[1]) / f[1];
  const float d = sqrtf((pX * pX + pY * pY) + 1.0f * 1.0f);
  const orc_vec3 o = v3(pX / d, pY / d, 1.0f / d);
  return v3(o.z, -o.x, -o.y);
}

/* ------------------------------------------------------------------------- */
/* simulate                                                                   */
/* ------------------------------------------------------------------------- */

typedef struct {
  const orc_mesh* m;
  int kind; /* 0 spherical, 1 o1dn, 2 pinhole, 3 ondn */
  float pin_f[2], pin_c[2];
  const float* origs;
  const orc_spherical_model* sph;
  uint32_t width, height;
  orc_interval range;
  orc_vec3 orig;
  const float* dirs;
  const orc_transform* Tsb;
  const orc_transform* Tbm;
  uint32_t nposes;
  int use_bvh;
  const float* trig;   /* spherical, use_bvh = 2: cos(phi_v)[H] | sin(phi_v)[H] | cos(theta_h)[W] | sin(theta_h)[W] */
  uint8_t* hits; float* ranges; float* points; float* normals; uint32_t* face_ids;
  /* work distribution */
  volatile uint64_t next;
  uint64_t total;
  uint64_t grain;
  pthread_mutex_t mtx;
  orc_counters cnt;
} sim_job;

static void sim_range(sim_job* J, uint64_t begin, uint64_t end, orc_counters* cnt)
{
  const uint64_t per_pose = (uint64_t)J->width * J->height;
  const float nanv = NAN;
  uint32_t cur_pid = 0xFFFFFFFFu;
  orc_transform Tsm = orc_transform_identity(), Tms = Tsm;
  for (uint64_t g = begin; g < end; ++g) {
    const uint32_t pid = (uint32_t)(g / per_pose);
    const uint32_t loc = (uint32_t)(g % per_pose);
    const uint32_t vid = loc / J->width, hid = loc % J->width;
    if (pid != cur_pid) {
      Tsm = orc_transform_mult(J->Tbm[pid], *J->Tsb);
      Tms = orc_transform_inv(Tsm);
      cur_pid = pid;
    }
    orc_vec3 dir_s, orig_s, org_m;
    if (J->kind == 0) {
      /* rmagine SphericalModel::getDirection = polar2cartesian(phi, theta)
       * (convention pinned by rmcl_ros/src/util/conversions.cpp:174-188) */
      float cp, sp, ct, st;
      if (J->trig) {   /* the same four libm values, computed once per row / column of the scan instead of once per ray */
        cp = J->trig[vid]; sp = J->trig[J->height + vid]; ct = J->trig[2 * J->height + hid]; st = J->trig[2 * J->height + J->width + hid];
      } else {
        const float phi = J->sph->phi.min + (float)vid * J->sph->phi.inc;
        const float theta = J->sph->theta.min + (float)hid * J->sph->theta.inc;
        cp = cosf(phi); sp = sinf(phi); ct = cosf(theta); st = sinf(theta);
      }
      dir_s = v3(cp * ct, cp * st, sp);
      orig_s = v3(0, 0, 0);
      org_m = Tsm.t;
    } else if (J->kind == 1) {
      dir_s = v3(J->dirs[3 * loc], J->dirs[3 * loc + 1], J->dirs[3 * loc + 2]);
      orig_s = J->orig;
      org_m = orc_transform_apply(Tsm, orig_s);
    } else if (J->kind == 2) {
      dir_s = orc_pinhole_direction(J->pin_f, J->pin_c, vid, hid);
      orig_s = v3(0, 0, 0);
      org_m = Tsm.t;
    } else {
      /* OnDn: per-ray origin and direction (RCCEmbreeOnDn, RCCEmbree.cpp:102-130) */
      dir_s = v3(J->dirs[3 * loc], J->dirs[3 * loc + 1], J->dirs[3 * loc + 2]);
      orig_s = v3(J->origs[3 * loc], J->origs[3 * loc + 1], J->origs[3 * loc + 2]);
      org_m = orc_transform_apply(Tsm, orig_s);
    }
    const orc_vec3 dir_m = orc_quat_rotate(Tsm.R, dir_s);
    float t = 0; uint32_t face = 0xFFFFFFFFu;
    int hit;
    if (J->use_bvh == 2) hit = orc_intersect_bvh4(J->m, org_m, dir_m, 0.0f, J->range.max, &t, &face);
    else if (J->use_bvh) hit = orc_intersect_bvh(J->m, org_m, dir_m, 0.0f, J->range.max, &t, &face, cnt);
    else hit = orc_intersect_brute(J->m, org_m, dir_m, 0.0f, J->range.max, &t, &face);
    if (hit > 0) {
      if (J->hits) J->hits[g] = 1;
      if (J->ranges) J->ranges[g] = t;
      if (J->points) {
        orc_vec3 p = v_scale(dir_s, t);
        if (J->kind == 1 || J->kind == 3) p = v_add(p, orig_s);
        J->points[3 * g] = p.x; J->points[3 * g + 1] = p.y; J->points[3 * g + 2] = p.z;
      }
      if (J->normals) {
        orc_vec3 n = orc_quat_rotate(Tms.R, J->m->tris[face].n);
        /* flip towards the sensor */
        if (v_dot_plain(dir_s, n) > 0.0f) n = v3(-n.x, -n.y, -n.z);
        J->normals[3 * g] = n.x; J->normals[3 * g + 1] = n.y; J->normals[3 * g + 2] = n.z;
      }
      if (J->face_ids) J->face_ids[g] = face;
    } else {
      if (J->hits) J->hits[g] = 0;
      if (J->ranges) J->ranges[g] = J->range.max + 1.0f;
      if (J->points) { J->points[3 * g] = nanv; J->points[3 * g + 1] = nanv; J->points[3 * g + 2] = nanv; }
      if (J->normals) { J->normals[3 * g] = nanv; J->normals[3 * g + 1] = nanv; J->normals[3 * g + 2] = nanv; }
      if (J->face_ids) J->face_ids[g] = 0xFFFFFFFFu;
    }
  }
}

#define ORC_GRAIN 128 /* the reference's TBB grain (PCDSensorUpdaterEmbree.cpp:331) */

/* --- persistent worker pool -------------------------------------------------------------------------------
 * The reference parallelises with TBB (a process-wide pool of work-stealing workers).  The oracle keeps ONE pool
 * of pthreads alive across calls (spawning 256 threads per 131 072-ray scan cost more than the scan) and hands
 * worker w the chunks w, w + N, w + 2N, ... of the job: no shared counter, no contended cache line, every worker
 * writes its own slice of the outputs.  Chunks of 512 rays = 4 TBB grains keep neighbouring rays (same subtrees) in
 * one core's cache while giving every one of 256 workers >= 1 chunk of a 128x1024 scan. */
typedef void (*orc_pool_fn)(void* job, int worker, int nworkers);
static struct {
  pthread_mutex_t mtx;
  pthread_cond_t go, done;
  pthread_t* th;
  int n_alive;        /* threads created so far (ids 1..n_alive; the caller is worker 0) */
  int n_active;       /* workers taking part in the current job (including the caller) */
  uint64_t epoch;     /* bumped per job */
  int pending;        /* pool threads still running the current job */
  orc_pool_fn fn;
  void* job;
  int inited;
} g_pool;
static pthread_mutex_t g_pool_call = PTHREAD_MUTEX_INITIALIZER; /* one parallel region at a time */

static void* pool_thread(void* arg)
{
  const int id = (int)(intptr_t)arg;
  uint64_t seen = 0;
  pthread_mutex_lock(&g_pool.mtx);
  for (;;) {
    while (g_pool.epoch == seen) pthread_cond_wait(&g_pool.go, &g_pool.mtx);
    seen = g_pool.epoch;
    if (id < g_pool.n_active) {
      orc_pool_fn fn = g_pool.fn; void* job = g_pool.job; const int n = g_pool.n_active;
      pthread_mutex_unlock(&g_pool.mtx);
      fn(job, id, n);
      pthread_mutex_lock(&g_pool.mtx);
      if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.done);
    }
  }
  return NULL;
}

static void pool_run(orc_pool_fn fn, void* job, int nthreads)
{
  if (nthreads <= 1) { fn(job, 0, 1); return; }
  pthread_mutex_lock(&g_pool_call);
  if (!g_pool.inited) {
    pthread_mutex_init(&g_pool.mtx, NULL);
    pthread_cond_init(&g_pool.go, NULL);
    pthread_cond_init(&g_pool.done, NULL);
    g_pool.inited = 1;
  }
  pthread_mutex_lock(&g_pool.mtx);
  if (g_pool.n_alive < nthreads - 1) {
    g_pool.th = (pthread_t*)realloc(g_pool.th, sizeof(pthread_t) * (size_t)(nthreads - 1));
    for (int i = g_pool.n_alive; i < nthreads - 1; ++i) {
      /* a thread born now has seen = 0 < epoch only if epoch > 0: give it the current epoch by creating it while
       * we hold the mutex and letting it wait for the NEXT bump */
      pthread_create(&g_pool.th[i], NULL, pool_thread, (void*)(intptr_t)(i + 1));
    }
    g_pool.n_alive = nthreads - 1;
  }
  g_pool.fn = fn; g_pool.job = job; g_pool.n_active = nthreads; g_pool.pending = nthreads - 1;
  g_pool.epoch++;
  pthread_cond_broadcast(&g_pool.go);
  pthread_mutex_unlock(&g_pool.mtx);
  fn(job, 0, nthreads);
  pthread_mutex_lock(&g_pool.mtx);
  while (g_pool.pending != 0) pthread_cond_wait(&g_pool.done, &g_pool.mtx);
  pthread_mutex_unlock(&g_pool.mtx);
  pthread_mutex_unlock(&g_pool_call);
}

static void sim_worker(void* arg, int worker, int nworkers)
{
  sim_job* J = (sim_job*)arg;
  sim_job Jl = *J; /* private copy of the read-only job description */
  orc_counters local = {0, 0, 0};
  for (uint64_t b = (uint64_t)worker * Jl.grain; b < Jl.total; b += (uint64_t)nworkers * Jl.grain) {
    uint64_t e = b + Jl.grain; if (e > Jl.total) e = Jl.total;
    sim_range(&Jl, b, e, &local);
  }
  pthread_mutex_lock(&J->mtx);
  J->cnt.nodes_visited += local.nodes_visited; J->cnt.tris_tested += local.tris_tested; J->cnt.rays += local.rays;
  pthread_mutex_unlock(&J->mtx);
}

static int run_sim(sim_job* J, int nthreads, orc_counters* cnt)
{
  J->next = 0; J->total = (uint64_t)J->width * J->height * J->nposes;
  if (nthreads < 1) nthreads = 1;
  J->grain = 4 * ORC_GRAIN;
  pthread_mutex_init(&J->mtx, NULL);
  memset(&J->cnt, 0, sizeof(J->cnt));
  pool_run(sim_worker, J, nthreads);
  pthread_mutex_destroy(&J->mtx);
  if (cnt) *cnt = J->cnt;
  return 0;
}

int orc_simulate_spherical(const orc_mesh* m, const orc_spherical_model* model, const orc_transform* Tsb,
                           const orc_transform* Tbm, uint32_t nposes, int use_bvh, int nthreads,
                           uint8_t* hits, float* ranges, float* points, float* normals, uint32_t* face_ids,
                           orc_counters* cnt)
{
  sim_job J; memset(&J, 0, sizeof(J));
  J.m = m; J.kind = 0; J.sph = model; J.width = model->theta.size; J.height = model->phi.size;
  J.range = model->range; J.Tsb = Tsb; J.Tbm = Tbm; J.nposes = nposes; J.use_bvh = use_bvh;
  J.hits = hits; J.ranges = ranges; J.points = points; J.normals = normals; J.face_ids = face_ids;
  float* trig = NULL;
  if (use_bvh == 2) {
    /* the baseline path: the per-ray libm values, hoisted per row / column (identical values, hence identical rays) */
    const uint32_t H = J.height, W = J.width;
    trig = (float*)malloc(sizeof(float) * (2u * (size_t)H + 2u * (size_t)W + 1u));
    for (uint32_t v = 0; v < H; ++v) { const float phi = model->phi.min + (float)v * model->phi.inc; trig[v] = cosf(phi); trig[H + v] = sinf(phi); }
    for (uint32_t h = 0; h < W; ++h) { const float th = model->theta.min + (float)h * model->theta.inc; trig[2 * H + h] = cosf(th); trig[2 * H + W + h] = sinf(th); }
    J.trig = trig;
    if (!__atomic_load_n(&((orc_mesh*)m)->bvh4_ready, __ATOMIC_ACQUIRE)) ensure_bvh4((orc_mesh*)m);
  }
  const int rc = run_sim(&J, nthreads, cnt);
  free(trig);
  return rc;
}

int orc_simulate_o1dn(const orc_mesh* m, uint32_t width, uint32_t height, orc_interval range, orc_vec3 orig,
                      const float* dirs, const orc_transform* Tsb, const orc_transform* Tbm, uint32_t nposes,
                      int use_bvh, int nthreads, uint8_t* hits, float* ranges, float* points, float* normals,
                      uint32_t* face_ids, orc_counters* cnt)
{
  sim_job J; memset(&J, 0, sizeof(J));
  J.m = m; J.kind = 1; J.width = width; J.height = height; J.range = range; J.orig = orig; J.dirs = dirs;
  J.Tsb = Tsb; J.Tbm = Tbm; J.nposes = nposes; J.use_bvh = use_bvh;
  J.hits = hits; J.ranges = ranges; J.points = points; J.normals = normals; J.face_ids = face_ids;
  return run_sim(&J, nthreads, cnt);
}

/* ------------------------------------------------------------------------- */
/* statistics_p2l + CrossStatistics algebra + umeyama                         */
/* ------------------------------------------------------------------------- */

float orc_adaptive_max_dist(float max_dist, float adaptive_max_dist_min, double p)
{
  /* CorrespondencesCPU.cpp:21-23: float operands promoted to double, stored to float */
  return (float)((double)max_dist * (1.0 - p) + (double)adaptive_max_dist_min * p);
}

orc_cross_statistics orc_cross_statistics_identity(void)
{
  orc_cross_statistics s; memset(&s, 0, sizeof(s)); return s;
}

/* rm::CrossStatistics::operator+= : count-weighted (Chan) merge; covariance is
 * normalised by n and oriented model x dataset^T. */
orc_cross_statistics orc_cross_statistics_merge(orc_cross_statistics a, orc_cross_statistics b)
{
  orc_cross_statistics r;
  r.n_meas = a.n_meas + b.n_meas;
  if (r.n_meas == 0) return orc_cross_statistics_identity();
  const float w1 = (float)a.n_meas / (float)r.n_meas;
  const float w2 = (float)b.n_meas / (float)r.n_meas;
  r.dataset_mean = v_add(v_scale(a.dataset_mean, w1), v_scale(b.dataset_mean, w2));
  r.model_mean = v_add(v_scale(a.model_mean, w1), v_scale(b.model_mean, w2));
  const orc_vec3 m1 = v_sub(a.model_mean, r.model_mean), d1 = v_sub(a.dataset_mean, r.dataset_mean);
  const orc_vec3 m2 = v_sub(b.model_mean, r.model_mean), d2 = v_sub(b.dataset_mean, r.dataset_mean);
  const float mm1[3] = {m1.x, m1.y, m1.z}, dd1[3] = {d1.x, d1.y, d1.z};
  const float mm2[3] = {m2.x, m2.y, m2.z}, dd2[3] = {d2.x, d2.y, d2.z};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const float P1 = a.covariance[3 * i + j] * w1 + b.covariance[3 * i + j] * w2;
      const float P2 = (mm1[i] * dd1[j]) * w1 + (mm2[i] * dd2[j]) * w2;
      r.covariance[3 * i + j] = P1 + P2;
    }
  return r;
}

static void quat_to_mat(orc_quat q, float* M)
{
  /* rmagine Matrix3x3 <- Quaternion */
  M[0] = 2.0f * (q.w * q.w + q.x * q.x) - 1.0f; M[1] = 2.0f * (q.x * q.y - q.w * q.z); M[2] = 2.0f * (q.x * q.z + q.w * q.y);
  M[3] = 2.0f * (q.x * q.y + q.w * q.z); M[4] = 2.0f * (q.w * q.w + q.y * q.y) - 1.0f; M[5] = 2.0f * (q.y * q.z - q.w * q.x);
  M[6] = 2.0f * (q.x * q.z - q.w * q.y); M[7] = 2.0f * (q.y * q.z + q.w * q.x); M[8] = 2.0f * (q.w * q.w + q.z * q.z) - 1.0f;
}

/* Transform * CrossStatistics: means transformed, covariance rotated R C R^T */
orc_cross_statistics orc_cross_statistics_transform(orc_transform T, orc_cross_statistics s)
{
  orc_cross_statistics r;
  r.dataset_mean = orc_transform_apply(T, s.dataset_mean);
  r.model_mean = orc_transform_apply(T, s.model_mean);
  float R[9], tmp[9];
  quat_to_mat(T.R, R);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    float acc = 0.0f; for (int k = 0; k < 3; ++k) acc += R[3 * i + k] * s.covariance[3 * k + j];
    tmp[3 * i + j] = acc;
  }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    float acc = 0.0f; for (int k = 0; k < 3; ++k) acc += tmp[3 * i + k] * R[3 * j + k];
    r.covariance[3 * i + j] = acc;
  }
  r.n_meas = s.n_meas;
  return r;
}

/* gate + projection exactly as restated in-repo at rmcl_ros/src/micpl/MICPSensorCPU.cpp:70-84,
 * with Tpre applied to the dataset point first (rm::statistics_p2l). Returns 1 if kept. */
static inline int p2l_element(const orc_transform* Tpre, const float* dp, const float* mp, const float* mn,
                              float max_dist, orc_vec3* Di_out, orc_vec3* Mi_out)
{
  const orc_vec3 Di = orc_transform_apply(*Tpre, v3(dp[0], dp[1], dp[2]));
  const orc_vec3 Ii = v3(mp[0], mp[1], mp[2]);
  const orc_vec3 Ni = v3(mn[0], mn[1], mn[2]);
  const float spd = v_dot_plain(v_sub(Ii, Di), Ni);
  if (fabsf(spd) < max_dist) {
    *Di_out = Di;
    *Mi_out = v_add(Di, v_scale(Ni, spd));
    return 1;
  }
  return 0;
}

void orc_statistics_p2l_f32(const orc_transform* Tpre, const float* dp, const uint8_t* dm, const float* mp,
                            const float* mn, const uint8_t* mm, uint32_t n, float max_dist,
                            orc_cross_statistics* out)
{
  orc_cross_statistics s = orc_cross_statistics_identity();
  for (uint32_t i = 0; i < n; ++i) {
    if ((dm == NULL || dm[i] > 0) && (mm == NULL || mm[i] > 0)) {
      orc_vec3 Di, Mi;
      if (p2l_element(Tpre, dp + 3 * i, mp + 3 * i, mn + 3 * i, max_dist, &Di, &Mi)) {
        orc_cross_statistics one = orc_cross_statistics_identity();
        one.dataset_mean = Di; one.model_mean = Mi; one.n_meas = 1;
        s = orc_cross_statistics_merge(s, one);
      }
    }
  }
  *out = s;
}

void orc_statistics_p2l_f64(const orc_transform* Tpre, const float* dp, const uint8_t* dm, const float* mp,
                            const float* mn, const uint8_t* mm, uint32_t n, float max_dist, double* out15,
                            uint32_t* n_out)
{
  double sd[3] = {0, 0, 0}, sm[3] = {0, 0, 0}; uint32_t cnt = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if ((dm == NULL || dm[i] > 0) && (mm == NULL || mm[i] > 0)) {
      orc_vec3 Di, Mi;
      if (p2l_element(Tpre, dp + 3 * i, mp + 3 * i, mn + 3 * i, max_dist, &Di, &Mi)) {
        sd[0] += Di.x; sd[1] += Di.y; sd[2] += Di.z; sm[0] += Mi.x; sm[1] += Mi.y; sm[2] += Mi.z; cnt++;
      }
    }
  }
  memset(out15, 0, sizeof(double) * 15);
  *n_out = cnt;
  if (cnt == 0) return;
  double md[3], mmn[3];
  for (int k = 0; k < 3; ++k) { md[k] = sd[k] / cnt; mmn[k] = sm[k] / cnt; }
  double C[9] = {0};
  for (uint32_t i = 0; i < n; ++i) {
    if ((dm == NULL || dm[i] > 0) && (mm == NULL || mm[i] > 0)) {
      orc_vec3 Di, Mi;
      if (p2l_element(Tpre, dp + 3 * i, mp + 3 * i, mn + 3 * i, max_dist, &Di, &Mi)) {
        const double d[3] = {Di.x - md[0], Di.y - md[1], Di.z - md[2]};
        const double m_[3] = {Mi.x - mmn[0], Mi.y - mmn[1], Mi.z - mmn[2]};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[3 * r + c] += m_[r] * d[c];
      }
    }
  }
  for (int k = 0; k < 3; ++k) { out15[k] = md[k]; out15[3 + k] = mmn[k]; }
  for (int k = 0; k < 9; ++k) out15[6 + k] = C[k] / cnt;
}

/* The same statistics in ONE pass of raw double sums over static chunks on the worker pool -- what a tuned CPU reduction does (the
 * reference calls rm::statistics_p2l, whose CPU form is an OpenMP reduction).  Only bench.py's cpu_baseline leg times this form; the
 * checker uses the two forms above.  tests/test_oracle.py holds it to the two-pass value. */
typedef struct {
  const orc_transform* Tpre; const float* dp; const uint8_t* dm; const float* mp; const float* mn; const uint8_t* mm;
  uint32_t n; float max_dist;
  double (*part)[16];   /* per worker: sum d[3], sum m[3], sum m d^T [9], count */
} p2l_fast_job;

static void p2l_fast_worker(void* arg, int worker, int nworkers)
{
  const p2l_fast_job* J = (const p2l_fast_job*)arg;
  const uint32_t lo = (uint32_t)((uint64_t)J->n * (uint64_t)worker / (uint64_t)nworkers);
  const uint32_t hi = (uint32_t)((uint64_t)J->n * (uint64_t)(worker + 1) / (uint64_t)nworkers);
  double a[16];
  for (int k = 0; k < 16; ++k) a[k] = 0.0;
  const orc_transform T = *J->Tpre;
  for (uint32_t i = lo; i < hi; ++i) {
    if ((J->dm == NULL || J->dm[i] > 0) && (J->mm == NULL || J->mm[i] > 0)) {
      orc_vec3 Di, Mi;
      if (p2l_element(&T, J->dp + 3 * i, J->mp + 3 * i, J->mn + 3 * i, J->max_dist, &Di, &Mi)) {
        const double d[3] = {Di.x, Di.y, Di.z}, m_[3] = {Mi.x, Mi.y, Mi.z};
        for (int k = 0; k < 3; ++k) { a[k] += d[k]; a[3 + k] += m_[k]; }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a[6 + 3 * r + c] += m_[r] * d[c];
        a[15] += 1.0;
      }
    }
  }
  for (int k = 0; k < 16; ++k) J->part[worker][k] = a[k];
}

void orc_statistics_p2l_fast(const orc_transform* Tpre, const float* dp, const uint8_t* dm, const float* mp,
                             const float* mn, const uint8_t* mm, uint32_t n, float max_dist, int nthreads, orc_cross_statistics* out)
{
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  double part[256][16];
  p2l_fast_job J = {Tpre, dp, dm, mp, mn, mm, n, max_dist, part};
  pool_run(p2l_fast_worker, &J, nthreads);
  double a[16];
  for (int k = 0; k < 16; ++k) { a[k] = 0.0; for (int w = 0; w < nthreads; ++w) a[k] += part[w][k]; }
  orc_cross_statistics s = orc_cross_statistics_identity();
  const double cnt = a[15];
  if (cnt > 0.0) {
    const double md[3] = {a[0] / cnt, a[1] / cnt, a[2] / cnt}, mm_[3] = {a[3] / cnt, a[4] / cnt, a[5] / cnt};
    s.dataset_mean = v3((float)md[0], (float)md[1], (float)md[2]);
    s.model_mean = v3((float)mm_[0], (float)mm_[1], (float)mm_[2]);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) s.covariance[3 * r + c] = (float)(a[6 + 3 * r + c] / cnt - mm_[r] * md[c]);
    s.n_meas = (uint32_t)cnt;
  }
  *out = s;
}

/* symmetric 3x3 Jacobi eigen-decomposition (double). A = V diag(e) V^T */
static void jacobi_eig3(const double* Ain, double* V, double* e)
{
  double A[9]; memcpy(A, Ain, sizeof(A));
  for (int i = 0; i < 9; ++i) V[i] = 0;
  V[0] = V[4] = V[8] = 1;
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double dg = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-30 * dg || off == 0.0) break;
    for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
      const double apq = A[3 * p + q];
      if (apq == 0.0) continue;
      const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; ++k) { /* A <- A J */
        const double akp = A[3 * k + p], akq = A[3 * k + q];
        A[3 * k + p] = c * akp - s * akq; A[3 * k + q] = s * akp + c * akq;
      }
      for (int k = 0; k < 3; ++k) { /* A <- J^T A */
        const double apk = A[3 * p + k], aqk = A[3 * q + k];
        A[3 * p + k] = c * apk - s * aqk; A[3 * q + k] = s * apk + c * aqk;
      }
      for (int k = 0; k < 3; ++k) {
        const double vkp = V[3 * k + p], vkq = V[3 * k + q];
        V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq;
      }
    }
  }
  e[0] = A[0]; e[1] = A[4]; e[2] = A[8];
}

void orc_svd3(const double* A, double* U, double* w, double* V)
{
  /* B = A^T A = V S^2 V^T ; u_i = A v_i / s_i */
  double B[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double acc = 0; for (int k = 0; k < 3; ++k) acc += A[3 * k + i] * A[3 * k + j];
    B[3 * i + j] = acc;
  }
  double Vr[9], e[3];
  jacobi_eig3(B, Vr, e);
  int idx[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) if (e[idx[j]] > e[idx[i]]) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) V[3 * r + c] = Vr[3 * r + idx[c]];
    w[c] = e[idx[c]] > 0 ? sqrt(e[idx[c]]) : 0.0;
  }
  const double tol = 1e-9 * (w[0] > 0 ? w[0] : 1.0);
  double u[3][3];
  int have[3] = {0, 0, 0};
  for (int c = 0; c < 3; ++c) {
    if (w[c] > tol) {
      for (int r = 0; r < 3; ++r) { double acc = 0; for (int k = 0; k < 3; ++k) acc += A[3 * r + k] * V[3 * k + c]; u[c][r] = acc / w[c]; }
      /* re-orthonormalise against previous */
      for (int p = 0; p < c; ++p) if (have[p]) { double d = 0; for (int r = 0; r < 3; ++r) d += u[c][r] * u[p][r]; for (int r = 0; r < 3; ++r) u[c][r] -= d * u[p][r]; }
      double nrm = 0; for (int r = 0; r < 3; ++r) nrm += u[c][r] * u[c][r]; nrm = sqrt(nrm);
      if (nrm > 0) { for (int r = 0; r < 3; ++r) u[c][r] /= nrm; have[c] = 1; }
    }
  }
  if (!have[0]) { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; have[0] = 1; }
  if (!have[1]) {
    /* any unit vector orthogonal to u0 */
    int k = 0; if (fabs(u[0][1]) < fabs(u[0][k])) k = 1; if (fabs(u[0][2]) < fabs(u[0][k])) k = 2;
    double a[3] = {0, 0, 0}; a[k] = 1;
    double d = u[0][k];
    double nrm = 0;
    for (int r = 0; r < 3; ++r) { u[1][r] = a[r] - d * u[0][r]; nrm += u[1][r] * u[1][r]; }
    nrm = sqrt(nrm); for (int r = 0; r < 3; ++r) u[1][r] /= nrm;
    have[1] = 1;
  }
  if (!have[2]) {
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  }
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) U[3 * r + c] = u[c][r];
}

static double det3(const double* M)
{
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

static orc_quat mat_to_quat(const double* R)
{
  /* Shepperd's method */
  double q[4]; /* x y z w */
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    const double s = sqrt(tr + 1.0) * 2.0;
    q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2.0;
    q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2.0;
    q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s;
  } else {
    const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2.0;
    q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s;
  }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  orc_quat r = {(float)(q[0] / n), (float)(q[1] / n), (float)(q[2] / n), (float)(q[3] / n)};
  return r;
}

/* rm::umeyama_transform: C = U S V^T, R = U diag(1,1,sign(det U det V)) V^T,
 * t = model_mean - R dataset_mean; identity when n_meas == 0. */
orc_transform orc_umeyama_transform(const orc_cross_statistics* s)
{
  orc_transform T = orc_transform_identity();
  if (s->n_meas == 0) return T;
  double C[9], U[9], w[3], V[9];
  for (int i = 0; i < 9; ++i) C[i] = s->covariance[i];
  orc_svd3(C, U, w, V);
  double S[3] = {1, 1, 1};
  if (det3(U) * det3(V) < 0) S[2] = -1;
  double R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double acc = 0; for (int k = 0; k < 3; ++k) acc += U[3 * i + k] * S[k] * V[3 * j + k];
    R[3 * i + j] = acc;
  }
  T.R = mat_to_quat(R);
  const orc_vec3 Rd = orc_quat_rotate(T.R, s->dataset_mean);
  T.t = v_sub(s->model_mean, Rd);
  return T;
}

/* ------------------------------------------------------------------------- */
/* particle filter                                                            */
/* ------------------------------------------------------------------------- */

/* rm::Gaussian1D::operator+= : 1-D count-weighted merge (parity unpinned) */
orc_gaussian1d orc_gaussian1d_add(orc_gaussian1d a, orc_gaussian1d b)
{
  orc_gaussian1d r;
  r.n_meas = a.n_meas + b.n_meas;
  const float w1 = (float)a.n_meas / (float)r.n_meas;
  const float w2 = (float)b.n_meas / (float)r.n_meas;
  r.mean = a.mean * w1 + b.mean * w2;
  const float P1 = a.sigma * w1 + b.sigma * w2;
  const float P2 = ((a.mean - r.mean) * (a.mean - r.mean)) * w1 + ((b.mean - r.mean) * (b.mean - r.mean)) * w2;
  r.sigma = P1 + P2;
  return r;
}

/* evaluate_rcc, PCDSensorUpdaterEmbree.cpp:18-86, with UNIT face normals
 * (== optix/BeamEvaluateProgram.cu:77-120; SURVEY.md Appendix B.1 recommendation). */
float orc_evaluate_rcc(const orc_mesh* m, const orc_range_measurement* meas, const orc_pf_params* p, int use_bvh)
{
  const int real_hit = (meas->range >= p->sensor_range.min && meas->range <= p->sensor_range.max);
  float t = 0; uint32_t face = 0;
  int hit;
  /* correspondence_type 3 = the OptiX program's rules (BeamEvaluateProgram.cu:44-49,77-120): tmax 1e4, any hit counts */
  const int optix = (p->correspondence_type == 3);
  const float tfar = optix ? 1.0e4f : INFINITY;
  /* use_bvh 2: the BVH4 walk with SSE slab tests (same intersector, same tie rule: identical (t, face) -- tests/test_oracle.py);
   * it makes the FULL-size checks of C4 / C5 (25.6 M and 32 M rays) a matter of seconds */
  if (use_bvh == 2) hit = orc_intersect_bvh4(m, meas->orig, meas->dir, 0.0f, tfar, &t, &face);
  else if (use_bvh) hit = orc_intersect_bvh(m, meas->orig, meas->dir, 0.0f, tfar, &t, &face, NULL);
  else hit = orc_intersect_brute(m, meas->orig, meas->dir, 0.0f, tfar, &t, &face);
  const int sim_hit = (hit > 0) && (optix || t > p->sensor_range.min);
  float error;
  if (sim_hit) {
    if (real_hit) {
      const orc_vec3 preal = v_add(meas->orig, v_scale(meas->dir, meas->range));
      const orc_vec3 pint = v_add(meas->orig, v_scale(meas->dir, t));
      /* correspondence_type 2: Embree's rayhit.hit.Ng, not normalised (PCDSensorUpdaterEmbree.cpp:56-66) */
      error = fabsf(v_dot_plain(v_sub(pint, preal), p->correspondence_type == 2 ? m->tris[face].Ng : m->tris[face].n));
    } else error = p->real_miss_sim_hit_error;
  } else {
    error = real_hit ? p->real_hit_sim_miss_error : p->real_miss_sim_miss_error;
  }
  return error;
}

typedef struct {
  const orc_mesh* m; const orc_transform* poses; orc_particle_attributes* attrs; uint32_t n;
  const orc_range_measurement* beams; uint32_t nbeams; const orc_transform* Tsb; const orc_pf_params* p;
  int use_bvh; float* errors;
  volatile uint64_t next;
  uint64_t grain;
} pf_job;

static void pf_particle(pf_job* J, uint32_t i)
{
  const orc_pf_params* p = J->p;
  const float sq = p->dist_sigma * p->dist_sigma;
  const orc_transform Tsm = orc_transform_mult(J->poses[i], *J->Tsb);
  orc_particle_attributes a = J->attrs[i];
  for (uint32_t b = 0; b < J->nbeams; ++b) {
    /* meas_m = Tsm * meas_s (RangeMeasurement.hpp:28-42) */
    orc_range_measurement mm = J->beams[b];
    mm.dir = orc_quat_rotate(Tsm.R, J->beams[b].dir);
    mm.orig = orc_transform_apply(Tsm, J->beams[b].orig);
    float error;
    if (p->correspondence_type == 1) {
      /* evaluate_cpc (PCDSensorUpdaterEmbree.cpp:88-95): distance of meas_m.mean() (RangeMeasurement.hpp:17-20) */
      const orc_vec3 mean = v_add(mm.orig, v_scale(mm.dir, mm.range));
      float d; orc_vec3 cp; uint32_t face;
      error = (orc_closest_point(J->m, mean, J->use_bvh, &d, &cp, &face) > 0) ? d : NAN;
    } else {
      error = orc_evaluate_rcc(J->m, &mm, p, J->use_bvh);
    }
    if (J->errors) J->errors[(size_t)i * J->nbeams + b] = error;
    /* PCDSensorUpdaterEmbree.cpp:224: float argument, double exp/sqrt, float result */
    const float arg = -(error * error) / sq / 2;
    const float eval = (float)(exp((double)arg) / sqrt((double)(2 * sq) * M_PI));
    orc_gaussian1d meas = {eval, 0.0f, 1};
    a.likelihood = orc_gaussian1d_add(a.likelihood, meas);
    if (a.likelihood.n_meas > p->max_n_meas) a.likelihood.n_meas = p->max_n_meas;
  }
  J->attrs[i] = a;
}

static void pf_worker(void* arg, int worker, int nworkers)
{
  pf_job* J = (pf_job*)arg;
  pf_job Jl = *J;  /* private copy of the read-only fields (see sim_worker) */
  for (uint64_t b = (uint64_t)worker * Jl.grain; b < Jl.n; b += (uint64_t)nworkers * Jl.grain) {
    uint64_t e = b + Jl.grain; if (e > Jl.n) e = Jl.n;
    for (uint64_t i = b; i < e; ++i) pf_particle(&Jl, (uint32_t)i);
  }
}

int orc_pf_update(const orc_mesh* m, const orc_transform* poses, orc_particle_attributes* attrs, uint32_t n,
                  const orc_range_measurement* beams, uint32_t nbeams, const orc_transform* Tsb,
                  const orc_pf_params* p, int use_bvh, int nthreads, float* errors_out)
{
  pf_job J = {m, poses, attrs, n, beams, nbeams, Tsb, p, use_bvh, errors_out, 0, ORC_GRAIN};
  if (nthreads < 1) nthreads = 1;
  J.grain = (uint64_t)n / ((uint64_t)nthreads * 16u);   /* particles per fetch (see run_sim) */
  if (J.grain < 1) J.grain = 1;
  if (J.grain > ORC_GRAIN) J.grain = ORC_GRAIN;
  pool_run(pf_worker, &J, nthreads);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* checker for the PRODUCT's BVH arrays (layout: rmcl_amd/csrc/layout.h):    */
/* walks the exported Node4 / TriRec dwords with the oracle's intersector so */
/* the CPU suite can prove the builder never loses a triangle.               */
/* ------------------------------------------------------------------------- */
int orc_trace_bvh4(const uint32_t* nodes, uint32_t n_nodes, const uint32_t* tris, uint32_t n_tris,
                   orc_vec3 O, orc_vec3 D, float tnear, float tfar, float* t_out, uint32_t* face_out)
{
  const float o[3] = {O.x, O.y, O.z};
  const float inv[3] = {safe_inv(D.x), safe_inv(D.y), safe_inv(D.z)};
  float best_t = tfar; uint32_t best_f = 0xFFFFFFFFu; int found = 0;
  uint32_t stack[256]; int sp = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    const uint32_t ref = stack[--sp];
    if (ref & 0x80000000u) {
      const uint32_t first = ref & 0x0FFFFFFFu, cnt = ((ref >> 28) & 7u) + 1u;
      for (uint32_t i = 0; i < cnt; ++i) {
        if (first + i >= n_tris) return -2;
        const float* r = (const float*)(tris + 16u * (first + i));
        orc_tri T;
        T.v0 = v3(r[0], r[1], r[2]); T.e1 = v3(r[3], r[4], r[5]); T.e2 = v3(r[6], r[7], r[8]);
        T.Ng = v3(r[9], r[10], r[11]); T.n = v3(r[12], r[13], r[14]);
        const uint32_t f = tris[16u * (first + i) + 15u];
        float t;
        if (tri_intersect(&T, O, D, tnear, tfar, &t)) {
          if (!found || t < best_t || (t == best_t && f < best_f)) { best_t = t; best_f = f; found = 1; }
        }
      }
      continue;
    }
    if (ref >= n_nodes) return -2;
    const float* nd = (const float*)(nodes + 32u * ref);
    const uint32_t* ch = nodes + 32u * ref + 24u;
    const uint32_t nk = nodes[32u * ref + 28u];   /* number of valid children; unused slots are skipped */
    if (nk < 1 || nk > 4) return -2;
    for (uint32_t c = 0; c < nk; ++c) {
      orc_node bx;
      bx.bmin[0] = nd[0 + c]; bx.bmax[0] = nd[4 + c];
      bx.bmin[1] = nd[8 + c]; bx.bmax[1] = nd[12 + c];
      bx.bmin[2] = nd[16 + c]; bx.bmax[2] = nd[20 + c];
      float tn;
      if (box_hit(&bx, o, inv, tnear, best_t, &tn)) { if (sp >= 256) return -1; stack[sp++] = ch[c]; }
    }
  }
  if (found) { *t_out = best_t; *face_out = best_f; }
  return found;
}

/* rmagine SphericalModel::getDirection(vid, hid) for the whole image, buffer order vid*W + hid
 * (convention pinned by rmcl_ros/src/util/conversions.cpp:174-188) */
void orc_spherical_directions(const orc_spherical_model* model, float* out)
{
  const uint32_t H = model->phi.size, W = model->theta.size;
  for (uint32_t vid = 0; vid < H; ++vid) {
    const float phi = model->phi.min + (float)vid * model->phi.inc;
    const float cp = cosf(phi), sp = sinf(phi);
    for (uint32_t hid = 0; hid < W; ++hid) {
      const float theta = model->theta.min + (float)hid * model->theta.inc;
      float* o = out + 3u * ((size_t)vid * W + hid);
      o[0] = cp * cosf(theta); o[1] = cp * sinf(theta); o[2] = sp;
    }
  }
}

/* triangle records exactly as the oracle derives them, face-id order, 15 floats each */
void orc_mesh_tri_records(const orc_mesh* m, float* out)
{
  for (uint32_t f = 0; f < m->nf; ++f) {
    const orc_tri* T = &m->tris[f];
    float* o = out + 15u * f;
    o[0] = T->v0.x; o[1] = T->v0.y; o[2] = T->v0.z; o[3] = T->e1.x; o[4] = T->e1.y; o[5] = T->e1.z;
    o[6] = T->e2.x; o[7] = T->e2.y; o[8] = T->e2.z; o[9] = T->Ng.x; o[10] = T->Ng.y; o[11] = T->Ng.z;
    o[12] = T->n.x; o[13] = T->n.y; o[14] = T->n.z;
  }
}

int orc_simulate_pinhole(const orc_mesh* m, uint32_t width, uint32_t height, orc_interval range, const float* f,
                         const float* c, const orc_transform* Tsb, const orc_transform* Tbm, uint32_t nposes,
                         int use_bvh, int nthreads, uint8_t* hits, float* ranges, float* points, float* normals,
                         uint32_t* face_ids, orc_counters* cnt)
{
  sim_job J; memset(&J, 0, sizeof(J));
  J.m = m; J.kind = 2; J.width = width; J.height = height; J.range = range;
  J.pin_f[0] = f[0]; J.pin_f[1] = f[1]; J.pin_c[0] = c[0]; J.pin_c[1] = c[1];
  J.Tsb = Tsb; J.Tbm = Tbm; J.nposes = nposes; J.use_bvh = use_bvh;
  J.hits = hits; J.ranges = ranges; J.points = points; J.normals = normals; J.face_ids = face_ids;
  return run_sim(&J, nthreads, cnt);
}

int orc_simulate_ondn(const orc_mesh* m, uint32_t width, uint32_t height, orc_interval range, const float* origs,
                      const float* dirs, const orc_transform* Tsb, const orc_transform* Tbm, uint32_t nposes,
                      int use_bvh, int nthreads, uint8_t* hits, float* ranges, float* points, float* normals,
                      uint32_t* face_ids, orc_counters* cnt)
{
  sim_job J; memset(&J, 0, sizeof(J));
  J.m = m; J.kind = 3; J.width = width; J.height = height; J.range = range; J.origs = origs; J.dirs = dirs;
  J.Tsb = Tsb; J.Tbm = Tbm; J.nposes = nposes; J.use_bvh = use_bvh;
  J.hits = hits; J.ranges = ranges; J.points = points; J.normals = normals; J.face_ids = face_ids;
  return run_sim(&J, nthreads, cnt);
}

void orc_pinhole_directions(uint32_t width, uint32_t height, const float* f, const float* c, float* out)
{
  for (uint32_t vid = 0; vid < height; ++vid)
    for (uint32_t hid = 0; hid < width; ++hid) {
      const orc_vec3 d = orc_pinhole_direction(f, c, vid, hid);
      float* o = out + 3u * ((size_t)vid * width + hid);
      o[0] = d.x; o[1] = d.y; o[2] = d.z;
    }
}

/* ------------------------------------------------------------------------- */
/* particle-filter motion update: TFMotionUpdaterCPU::update inner loop        */
/* (rmcl_ros/src/rmcl/TFMotionUpdaterCPU.cpp:184-224) == particle_motion.cu:11-34 */
/* plus the wall-collision test collision_in_between (:17-50)                   */
/* ------------------------------------------------------------------------- */
int orc_collision_in_between(const orc_mesh* m, orc_vec3 p1, orc_vec3 p2, int use_bvh)
{
  orc_vec3 vec = v_sub(p2, p1);
  const float length = sqrtf((vec.x * vec.x + vec.y * vec.y) + vec.z * vec.z);
  if (length < 0.00001) return 0;
  vec = v3(vec.x / length, vec.y / length, vec.z / length);
  float t; uint32_t face;
  const int hit = use_bvh ? orc_intersect_bvh(m, p1, vec, 0.0f, length, &t, &face, NULL)
                          : orc_intersect_brute(m, p1, vec, 0.0f, length, &t, &face);
  return hit > 0;
}

void orc_pf_motion_update(const orc_mesh* m /* NULL: no collision test */, orc_transform* poses,
                          orc_particle_attributes* attrs, uint32_t n, const orc_transform* T_bnew_bold,
                          double forget_rate, uint32_t max_n_meas, int use_bvh)
{
  for (uint32_t i = 0; i < n; ++i) {
    const orc_transform pose_old = poses[i];
    orc_particle_attributes attr = attrs[i];
    const orc_transform pose_new = orc_transform_mult(pose_old, *T_bnew_bold);
    /* `n_meas -= forget_rate * n_meas` on a uint32: double arithmetic, truncating store */
    attr.likelihood.n_meas = (uint32_t)((double)attr.likelihood.n_meas - forget_rate * (double)attr.likelihood.n_meas);
    if (m && orc_collision_in_between(m, pose_old.t, pose_new.t, use_bvh)) {
      attr.likelihood.mean = 0.0f; attr.likelihood.sigma = 0.0f; attr.likelihood.n_meas = max_n_meas;
    }
    poses[i] = pose_new;
    attrs[i] = attr;
  }
}

/* ------------------------------------------------------------------------- */
/* closest-point correspondences: CPCEmbree::find (rmcl/src/rmcl/registration/ */
/* CPCEmbree.cpp:18-44) -> rm::EmbreeMap::closestPoint (external; Embree point   */
/* query + the closest-point-on-triangle routine of Embree's closest_point      */
/* tutorial, i.e. Ericson, Real-Time Collision Detection 5.1.5).                */
/* a = v0, ab = -e1, ac = e2, b = a + ab, c = a + ac (record form).             */
/* Tie-break of equidistant triangles: min squared distance, then min face id.  */
/* ------------------------------------------------------------------------- */
static inline orc_vec3 closest_point_triangle(const orc_tri* T, orc_vec3 p)
{
  const orc_vec3 a = T->v0;
  const orc_vec3 ab = v3(-T->e1.x, -T->e1.y, -T->e1.z), ac = T->e2;
  const orc_vec3 b = v_add(a, ab), c = v_add(a, ac);
  const orc_vec3 ap = v_sub(p, a);
  const float d1 = v_dot_plain(ab, ap), d2 = v_dot_plain(ac, ap);
  if (d1 <= 0.f && d2 <= 0.f) return a;
  const orc_vec3 bp = v_sub(p, b);
  const float d3 = v_dot_plain(ab, bp), d4 = v_dot_plain(ac, bp);
  if (d3 >= 0.f && d4 <= d3) return b;
  const orc_vec3 cp = v_sub(p, c);
  const float d5 = v_dot_plain(ab, cp), d6 = v_dot_plain(ac, cp);
  if (d6 >= 0.f && d5 <= d6) return c;
  const float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { const float v = d1 / (d1 - d3); return v_add(a, v_scale(ab, v)); }
  const float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { const float v = d2 / (d2 - d6); return v_add(a, v_scale(ac, v)); }
  const float va = d3 * d6 - d5 * d4;
  if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
    const float v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    return v_add(b, v_scale(v_sub(c, b), v));
  }
  const float denom = 1.f / ((va + vb) + vc);
  const float v = vb * denom, w = vc * denom;
  return v_add(v_add(a, v_scale(ab, v)), v_scale(ac, w));
}

static inline float dist2(orc_vec3 a, orc_vec3 b)
{
  const orc_vec3 d = v_sub(a, b);
  return (d.x * d.x + d.y * d.y) + d.z * d.z;
}

static inline float box_dist2(const orc_node* n, const float* p)
{
  float acc = 0.f;
  for (int k = 0; k < 3; ++k) {
    float d = 0.f;
    if (p[k] < n->bmin[k]) d = n->bmin[k] - p[k]; else if (p[k] > n->bmax[k]) d = p[k] - n->bmax[k];
    acc += d * d;
  }
  return acc;
}

int orc_closest_point(const orc_mesh* m, orc_vec3 P, int use_bvh, float* d_out, orc_vec3* cp_out, uint32_t* face_out)
{
  float best = INFINITY; uint32_t best_f = 0xFFFFFFFFu; orc_vec3 best_p = v3(0, 0, 0);
  if (!(P.x == P.x && P.y == P.y && P.z == P.z)) return 0;
  if (!use_bvh || m->n_nodes == 0) {
    for (uint32_t f = 0; f < m->nf; ++f) {
      const orc_vec3 q = closest_point_triangle(&m->tris[f], P);
      const float d2 = dist2(P, q);
      if (d2 < best || (d2 == best && f < best_f)) { best = d2; best_f = f; best_p = q; }
    }
  } else {
    const float p[3] = {P.x, P.y, P.z};
    uint32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
      const orc_node* n = &m->nodes[stack[--sp]];
      if (box_dist2(n, p) * 0.999999f > best) continue;
      if (n->count > 0) {
        for (uint32_t i = 0; i < n->count; ++i) {
          const uint32_t f = m->prim[n->left_first + i];
          const orc_vec3 q = closest_point_triangle(&m->tris[f], P);
          const float d2 = dist2(P, q);
          if (d2 < best || (d2 == best && f < best_f)) { best = d2; best_f = f; best_p = q; }
        }
        continue;
      }
      const uint32_t l = n->left_first, r = l + 1;
      const float dl = box_dist2(&m->nodes[l], p), dr = box_dist2(&m->nodes[r], p);
      if (sp + 2 > 128) return -1;
      if (dl <= dr) { stack[sp++] = r; stack[sp++] = l; } else { stack[sp++] = l; stack[sp++] = r; }
    }
  }
  if (best_f == 0xFFFFFFFFu) return 0;
  *d_out = sqrtf(best); *cp_out = best_p; *face_out = best_f;
  return 1;
}

/* CPCEmbree::find: model buffers sized like the dataset; hits = (d <= max_dist); point / normal back in the
 * sensor frame.  Non-finite dataset points give hits = 0 and NaN outputs. */
void orc_cpc_find(const orc_mesh* m, const orc_transform* Tsb, const orc_transform* Tbm, const float* dataset_points,
                  uint32_t n, float max_dist, int use_bvh, uint8_t* hits, float* dists, float* points, float* normals,
                  uint32_t* face_ids)
{
  const orc_transform Tsm = orc_transform_mult(*Tbm, *Tsb);
  const orc_transform Tms = orc_transform_inv(Tsm);
  for (uint32_t i = 0; i < n; ++i) {
    const orc_vec3 Pm = orc_transform_apply(Tsm, v3(dataset_points[3 * i], dataset_points[3 * i + 1], dataset_points[3 * i + 2]));
    float d; orc_vec3 cp; uint32_t face;
    if (orc_closest_point(m, Pm, use_bvh, &d, &cp, &face) > 0) {
      const orc_vec3 ps = orc_transform_apply(Tms, cp);
      const orc_vec3 ns = orc_quat_rotate(Tms.R, m->tris[face].n);
      if (hits) hits[i] = (d <= max_dist) ? 1 : 0;
      if (dists) dists[i] = d;
      if (points) { points[3 * i] = ps.x; points[3 * i + 1] = ps.y; points[3 * i + 2] = ps.z; }
      if (normals) { normals[3 * i] = ns.x; normals[3 * i + 1] = ns.y; normals[3 * i + 2] = ns.z; }
      if (face_ids) face_ids[i] = face;
    } else {
      if (hits) hits[i] = 0;
      if (dists) dists[i] = NAN;
      if (points) { points[3 * i] = NAN; points[3 * i + 1] = NAN; points[3 * i + 2] = NAN; }
      if (normals) { normals[3 * i] = NAN; normals[3 * i + 1] = NAN; normals[3 * i + 2] = NAN; }
      if (face_ids) face_ids[i] = 0xFFFFFFFFu;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* gladiator resampling (resampling.cu:41-219, GladiatorResamplerCPU.cpp)    */
/* ------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* simple_stats_kernel (resampling.cu:41-81): {sum, max} of likelihood.mean, max seeded with 0 like the kernel's
 * shared-memory init (:52-53); the sum is accumulated in double (order independent to float precision). */
orc_likelihood_stats orc_likelihood_stats_compute(const orc_particle_attributes* attrs, uint32_t n)
{
  double sum = 0.0; float mx = 0.0f;
  for (uint32_t i = 0; i < n; ++i) {
    const float L = attrs[i].likelihood.mean;
    sum += (double)L;
    if (L > mx) mx = L;
  }
  orc_likelihood_stats r = {(float)sum, mx};
  return r;
}

/* rmagine EulerAngles::set(Quaternion) (EXTERNAL; the textbook ZYX extraction) */
void orc_quat_to_euler(orc_quat q, float* roll, float* pitch, float* yaw)
{
  const float sinr_cosp = 2.0f * (q.w * q.x + q.y * q.z);
  const float cosr_cosp = 1.0f - 2.0f * (q.x * q.x + q.y * q.y);
  const float sinp = 2.0f * (q.w * q.y - q.z * q.x);
  const float siny_cosp = 2.0f * (q.w * q.z + q.x * q.y);
  const float cosy_cosp = 1.0f - 2.0f * (q.y * q.y + q.z * q.z);
  *roll = (float)atan2((double)sinr_cosp, (double)cosr_cosp);
  if (fabsf(sinp) >= 1.0f) *pitch = copysignf((float)(M_PI / 2.0), sinp);
  else *pitch = (float)asin((double)sinp);
  *yaw = (float)atan2((double)siny_cosp, (double)cosy_cosp);
}

static orc_quat euler_to_quat_d(float roll, float pitch, float yaw)
{
  const float cr = (float)cos((double)(roll / 2.0f)), sr = (float)sin((double)(roll / 2.0f));
  const float cp = (float)cos((double)(pitch / 2.0f)), sp = (float)sin((double)(pitch / 2.0f));
  const float cy = (float)cos((double)(yaw / 2.0f)), sy = (float)sin((double)(yaw / 2.0f));
  orc_quat q;
  q.w = cr * cp * cy + sr * sp * sy;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  return q;
}

static void box_muller(uint32_t a, uint32_t b, float* z0, float* z1)
{
  const double u1 = ((double)a + 0.5) * (1.0 / 4294967296.0), u2 = ((double)b + 0.5) * (1.0 / 4294967296.0);
  const double r = sqrt(-2.0 * log(u1)), ang = 6.283185307179586476925 * u2;
  *z0 = (float)(r * cos(ang));
  *z1 = (float)(r * sin(ang));
}

void orc_gladiator_resample(const orc_transform* poses, const orc_particle_attributes* attrs, uint32_t n,
                            orc_transform* poses_new, orc_particle_attributes* attrs_new, uint32_t first,
                            uint32_t count, const orc_gladiator_config* cfg, uint64_t seed, uint32_t step)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t champion = first + k;
    uint32_t ra[4], rb[4];
    const uint32_t c0[4] = {champion, step, 0u, 0u}, c1[4] = {champion, step, 1u, 0u};
    orc_philox4x32_10(c0, key, ra);
    orc_philox4x32_10(c1, key, rb);
    const uint32_t enemy = ra[0] % n;
    float Nd_tx, Nd_ty, Nd_tz, Nd_rx, Nd_ry, Nd_rz;
    box_muller(ra[1], ra[2], &Nd_tx, &Nd_ty);
    box_muller(ra[3], rb[0], &Nd_tz, &Nd_rx);
    box_muller(rb[1], rb[2], &Nd_ry, &Nd_rz);
    const float Lc = attrs[champion].likelihood.mean, Le = attrs[enemy].likelihood.mean;
    if (Le > Lc) {
      const orc_transform pose = poses[enemy];
      orc_transform pose_new = pose;
      orc_particle_attributes attrs_n = attrs[enemy];
      pose_new.t.x = pose_new.t.x + Nd_tx * cfg->min_noise_tx;
      pose_new.t.y = pose_new.t.y + Nd_ty * cfg->min_noise_ty;
      pose_new.t.z = pose_new.t.z + Nd_tz * cfg->min_noise_tz;
      float roll, pitch, yaw;
      orc_quat_to_euler(pose_new.R, &roll, &pitch, &yaw);
      roll = roll + Nd_rx * cfg->min_noise_roll;
      pitch = pitch + Nd_ry * cfg->min_noise_pitch;
      yaw = yaw + Nd_rz * cfg->min_noise_yaw;
      pose_new.R = euler_to_quat_d(roll, pitch, yaw);
      const orc_transform diff = orc_transform_mult(orc_transform_inv(pose), pose_new);
      const float t2 = (diff.t.x * diff.t.x + diff.t.y * diff.t.y) + diff.t.z * diff.t.z;
      const float trans_dist = (cfg->trans_dist_metric == 1u) ? t2 : sqrtf(t2);
      /* rmagine Quaternion::l2norm (EXTERNAL): the 4-vector norm, i.e. ~1 for a unit quaternion */
      const float rot_dist = sqrtf(((diff.R.w * diff.R.w + diff.R.x * diff.R.x) + diff.R.y * diff.R.y) + diff.R.z * diff.R.z);
      const float frs = (float)(1.0 - pow(1.0 - (double)cfg->likelihood_forget_per_meter, (double)trans_dist));
      const float frr = (float)(1.0 - pow(1.0 - (double)cfg->likelihood_forget_per_radian, (double)rot_dist));
      const float forget_rate = (frs > frr) ? frs : frr;
      const float remember_rate = (float)(1.0 - (double)forget_rate);
      attrs_n.likelihood.n_meas = (uint32_t)((float)attrs_n.likelihood.n_meas * remember_rate);
      poses_new[k] = pose_new;
      attrs_new[k] = attrs_n;
    } else {
      poses_new[k] = poses[champion];
      attrs_new[k] = attrs[champion];
    }
  }
}

/* ------------------------------------------------------------------------- */
/* residual resampling (ResidualResamplerCPU.cpp:55-203)                     */
/* ------------------------------------------------------------------------- */
/* The reference's loop, statement by statement: statistics {sum, max} of likelihood.mean in double (:72-85); then, until the new
 * cloud is full (:104): draw a random particle (:106), insert n = size_t(L / sum * N_new) copies of it -- clamped to the room
 * that is left (:115-121) --, each perturbed by Gaussians of width min_noise / (L / max) (:144-159) and with its n_meas reduced
 * by forget_per_meter^|dt|^2 * forget_per_radian^l2norm(dR) (:163-168).
 * Pinned where the reference is implementation-defined (mt19937 + uniform_int_distribution + normal_distribution<float>, all
 * libstdc++ specific): draw k takes particle philox(k, step, 2, 0)[0] % n; the six Gaussians of OUTPUT SLOT j come from
 * philox(j, step, 3, 0) / philox(j, step, 4, 0) by Box-Muller in double; pow() is evaluated in double and rounded to float
 * (the reference's std::pow(float, float)); a non-positive or NaN share inserts nothing.  max_draws bounds the loop (the
 * reference's does not terminate when every share truncates to 0).  Returns the number of slots filled; *n_draws = draws used. */
uint32_t orc_residual_resample(const orc_transform* poses, const orc_particle_attributes* attrs, uint32_t n,
                               orc_transform* poses_new, orc_particle_attributes* attrs_new, uint32_t n_new,
                               const orc_gladiator_config* cfg, uint64_t seed, uint32_t step, uint64_t max_draws,
                               uint64_t* n_draws)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  double weight_sum = 0.0, weight_max = 0.0;
  for (uint32_t i = 0; i < n; ++i) {
    const double v = (double)attrs[i].likelihood.mean;
    weight_sum += v;
    if (v > weight_max) weight_max = v;
  }
  uint32_t insertion_idx = 0;
  uint64_t k = 0;
  while (insertion_idx < n_new && k < max_draws && n > 0) {
    uint32_t r[4];
    const uint32_t ck[4] = {(uint32_t)k, step, 2u, (uint32_t)(k >> 32)};
    orc_philox4x32_10(ck, key, r);
    ++k;
    const uint32_t random_index = r[0] % n;
    const orc_transform pose = poses[random_index];
    const orc_particle_attributes at = attrs[random_index];
    const double L = (double)at.likelihood.mean;
    const double L_sum_normed = L / weight_sum, L_max_normed = L / weight_max;
    const double share = L_sum_normed * (double)n_new;
    const uint32_t left = n_new - insertion_idx;
    uint32_t n_ins = 0;
    if (share > 0.0) n_ins = (share >= (double)left) ? left : (uint32_t)share;
    for (uint32_t inner = 0; inner < n_ins; ++inner) {
      const uint32_t j = insertion_idx + inner;
      uint32_t ra[4], rb[4];
      const uint32_t c3[4] = {j, step, 3u, 0u}, c4[4] = {j, step, 4u, 0u};
      orc_philox4x32_10(c3, key, ra);
      orc_philox4x32_10(c4, key, rb);
      float Nd_tx, Nd_ty, Nd_tz, Nd_rx, Nd_ry, Nd_rz;
      box_muller(ra[0], ra[1], &Nd_tx, &Nd_ty);
      box_muller(ra[2], ra[3], &Nd_tz, &Nd_rx);
      box_muller(rb[0], rb[1], &Nd_ry, &Nd_rz);
      const float noise_tx = (float)((double)cfg->min_noise_tx / L_max_normed), noise_ty = (float)((double)cfg->min_noise_ty / L_max_normed);
      const float noise_tz = (float)((double)cfg->min_noise_tz / L_max_normed);
      const float noise_roll = (float)((double)cfg->min_noise_roll / L_max_normed), noise_pitch = (float)((double)cfg->min_noise_pitch / L_max_normed);
      const float noise_yaw = (float)((double)cfg->min_noise_yaw / L_max_normed);
      orc_transform pose_new = pose;
      orc_particle_attributes attrs_n = at;
      pose_new.t.x = pose_new.t.x + Nd_tx * noise_tx;
      pose_new.t.y = pose_new.t.y + Nd_ty * noise_ty;
      pose_new.t.z = pose_new.t.z + Nd_tz * noise_tz;
      float roll, pitch, yaw;
      orc_quat_to_euler(pose_new.R, &roll, &pitch, &yaw);
      roll = roll + Nd_rx * noise_roll;
      pitch = pitch + Nd_ry * noise_pitch;
      yaw = yaw + Nd_rz * noise_yaw;
      pose_new.R = euler_to_quat_d(roll, pitch, yaw);
      const orc_transform diff = orc_transform_mult(orc_transform_inv(pose), pose_new);
      const float trans_dist = (diff.t.x * diff.t.x + diff.t.y * diff.t.y) + diff.t.z * diff.t.z;   /* l2normSquared (:164) */
      const float rot_dist = sqrtf(((diff.R.w * diff.R.w + diff.R.x * diff.R.x) + diff.R.y * diff.R.y) + diff.R.z * diff.R.z);
      const float reduction_factor = (float)pow((double)cfg->likelihood_forget_per_meter, (double)trans_dist) *
                                     (float)pow((double)cfg->likelihood_forget_per_radian, (double)rot_dist);
      attrs_n.likelihood.n_meas = (uint32_t)((float)attrs_n.likelihood.n_meas * reduction_factor);
      poses_new[j] = pose_new;
      attrs_new[j] = attrs_n;
    }
    insertion_idx += n_ins;
  }
  if (n_draws) *n_draws = k;
  return insertion_idx;
}

/* ------------------------------------------------------------------------- */
/* PointCloud2 -> O1Dn model + dataset (conversions.cpp:869-1002 etc.)       */
/* ------------------------------------------------------------------------- */
int orc_pointcloud2_unpack(const uint8_t* data, uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step,
                           uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t datatype, orc_filter1d fh,
                           orc_filter1d fw, float range_min, float range_max, uint32_t* out_w, uint32_t* out_h,
                           float* dirs, float* ranges, float* points, uint8_t* mask, uint32_t* n_valid)
{
  if (datatype != 7u && datatype != 8u) return -1;     /* "Field X has unknown DataType" */
  if (fh.increment == 0 || fw.increment == 0) return -2;
  if (fw.skip_begin + fw.skip_end > width || fh.skip_begin + fh.skip_end > height) return -2;
  const uint32_t ow = (width - fw.skip_begin - fw.skip_end) / fw.increment;
  const uint32_t oh = (height - fh.skip_begin - fh.skip_end) / fh.increment;
  *out_w = ow; *out_h = oh;
  uint32_t nv = 0;
  if (!dirs) { if (n_valid) *n_valid = 0; return 0; }  /* size query */
  for (uint32_t ti = 0; ti < oh; ++ti) {
    const size_t si = (size_t)ti * fh.increment + fh.skip_begin;
    for (uint32_t tj = 0; tj < ow; ++tj) {
      const size_t sj = (size_t)tj * fw.increment + fw.skip_begin;
      const uint8_t* ptr = data + si * row_step + sj * point_step;
      float x, y, z;
      if (datatype == 7u) {
        memcpy(&x, ptr + off_x, 4); memcpy(&y, ptr + off_y, 4); memcpy(&z, ptr + off_z, 4);
      } else {
        double dx, dy, dz;
        memcpy(&dx, ptr + off_x, 8); memcpy(&dy, ptr + off_y, 8); memcpy(&dz, ptr + off_z, 8);
        x = (float)dx; y = (float)dy; z = (float)dz;
      }
      const size_t id = (size_t)ti * ow + tj;
      float range; orc_vec3 d;
      if (isfinite(x) && isfinite(y) && isfinite(z)) {
        range = sqrtf((x * x + y * y) + z * z);
        d = v3(x / range, y / range, z / range);
      } else {
        range = 0.0f; d = v3(0, 0, 0);
      }
      dirs[3 * id] = d.x; dirs[3 * id + 1] = d.y; dirs[3 * id + 2] = d.z;
      ranges[id] = range;
      /* unpackMessage: real_point = dir * range + orig (orig = 0 for a cloud in its own frame) */
      points[3 * id] = d.x * range + 0.0f; points[3 * id + 1] = d.y * range + 0.0f; points[3 * id + 2] = d.z * range + 0.0f;
      const int out = (range < range_min) || (range > range_max);
      mask[id] = out ? 0 : 1;
      if (!out) ++nv;
    }
  }
  if (n_valid) *n_valid = nv;
  return 0;
}

int orc_trace_bvh4_ordered(const uint32_t* nodes, const uint32_t* tris, orc_vec3 O, orc_vec3 D, float tnear, float tfar,
                           int mode, uint64_t counters[5], float* t_out, uint32_t* face_out)
{
  const float o[3] = {O.x, O.y, O.z};
  const float inv[3] = {safe_inv(D.x), safe_inv(D.y), safe_inv(D.z)};
  float best_t = tfar; uint32_t best_f = 0xFFFFFFFFu; int found = 0;
  uint32_t stack[256]; float stack_t[256]; int sp = 0;
  uint32_t cur = 0; int have = 1;
  while (have) {
    if (!(cur & 0x80000000u)) {
      counters[0]++;
      const float* nd = (const float*)(nodes + 32u * cur);
      const uint32_t* ch = nodes + 32u * cur + 24u;
      float key[4]; uint32_t ref[4]; int nh = 0;
      for (uint32_t c = 0; c < 4; ++c) {
        orc_node bx;
        bx.bmin[0] = nd[0 + c]; bx.bmax[0] = nd[4 + c];
        bx.bmin[1] = nd[8 + c]; bx.bmax[1] = nd[12 + c];
        bx.bmin[2] = nd[16 + c]; bx.bmax[2] = nd[20 + c];
        float tn;
        if (bx.bmin[0] < 1e29f && box_hit(&bx, o, inv, tnear, best_t, &tn)) { key[nh] = tn; ref[nh] = ch[c]; nh++; }
      }
      if (nh == 0) counters[1]++;
      /* nearest first; deferred ones sorted (mode bit1) or in slot order */
      for (int i = 0; i < nh; ++i) for (int j = i + 1; j < nh; ++j)
        if (key[j] < key[i] && (i == 0 || (mode & 2))) { float tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tr = ref[i]; ref[i] = ref[j]; ref[j] = tr; }
      for (int i = nh - 1; i >= 1; --i) { stack[sp] = ref[i]; stack_t[sp] = key[i]; sp++; }
      if (sp > (int)counters[4]) counters[4] = (uint64_t)sp;
      if (nh > 0) { cur = ref[0]; continue; }
    } else {
      counters[2]++;
      const uint32_t first = cur & 0x0FFFFFFFu, cnt = ((cur >> 28) & 7u) + 1u;
      for (uint32_t i = 0; i < cnt; ++i) {
        counters[3]++;
        const float* r = (const float*)(tris + 16u * (first + i));
        orc_tri T;
        T.v0 = v3(r[0], r[1], r[2]); T.e1 = v3(r[3], r[4], r[5]); T.e2 = v3(r[6], r[7], r[8]);
        T.Ng = v3(r[9], r[10], r[11]); T.n = v3(r[12], r[13], r[14]);
        const uint32_t f = tris[16u * (first + i) + 15u];
        float t;
        if (tri_intersect(&T, O, D, tnear, tfar, &t)) {
          if (!found || t < best_t || (t == best_t && f < best_f)) { best_t = t; best_f = f; found = 1; }
        }
      }
    }
    /* pop */
    have = 0;
    while (sp > 0) {
      --sp;
      if ((mode & 1) && stack_t[sp] > best_t) continue;
      cur = stack[sp]; have = 1; break;
    }
  }
  if (found) { *t_out = best_t; *face_out = best_f; }
  return found;
}

/* ------------------------------------------------------------------------- */
/* beam sampling of PCDSensorUpdaterEmbree::update (PCDSensorUpdaterEmbree.cpp:276-327) with the pinned stream:   */
/* MT19937 (Matsumoto & Nishimura 1998, the algorithm std::mt19937 is specified to be), index = draw % n_points.   */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } orc_mt19937;
static void mt_seed(orc_mt19937* g, uint32_t seed)
{
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}
static uint32_t mt_next(orc_mt19937* g)
{
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7FFFFFFFu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
  return y;
}
uint32_t orc_mt19937_draw(uint32_t seed, uint32_t n_skip)
{
  orc_mt19937 g; mt_seed(&g, seed);
  uint32_t v = 0;
  for (uint32_t i = 0; i <= n_skip; ++i) v = mt_next(&g);
  return v;
}

int orc_sample_beams_pointcloud2(const uint8_t* data, uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step,
                                 uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t datatype, uint32_t samples,
                                 uint32_t seed, orc_range_measurement* out, uint32_t* n_out)
{
  *n_out = 0;
  if (datatype != 7u && datatype != 8u) return -1;
  const uint64_t n_points = (uint64_t)width * height;
  if (n_points == 0) return samples ? -1 : 0;
  orc_mt19937 g; mt_seed(&g, seed);
  for (uint32_t s = 0; s < samples; ++s) {
    int valid = 0; float x = 0, y = 0, z = 0;
    for (int t = 0; t < 100 && !valid; ++t) {
      const uint64_t id = (uint64_t)mt_next(&g) % n_points;
      const uint8_t* ptr = data + (id / width) * row_step + (id % width) * point_step;
      if (datatype == 8u) { double d; memcpy(&d, ptr + off_x, 8); x = (float)d; memcpy(&d, ptr + off_y, 8); y = (float)d; memcpy(&d, ptr + off_z, 8); z = (float)d; }
      else { memcpy(&x, ptr + off_x, 4); memcpy(&y, ptr + off_y, 4); memcpy(&z, ptr + off_z, 4); }
      valid = (x == x) && (y == y) && (z == z);
    }
    if (!valid) break;
    orc_range_measurement m; memset(&m, 0, sizeof(m));
    const float norm = sqrtf((x * x + y * y) + z * z);
    m.dir = v3(x / norm, y / norm, z / norm);
    m.range = norm;
    m.cov[0] = m.cov[4] = m.cov[8] = 0.1f;
    out[(*n_out)++] = m;
  }
  return 0;
}
