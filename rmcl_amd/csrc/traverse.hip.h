// traverse.hip.h -- device-side traversal code of librmclhip (gfx950): ray / box / triangle arithmetic, the per-lane, quad and
// packet BVH4 traversals, the closest-point query and k_cpc_find.  Included by the PRODUCTION translation unit (kernels.hip)
// and by the experiments' one (kernels_lab.hip -> librmclhip_lab.so); everything lives in an unnamed namespace, so each
// library carries its own copy of what it instantiates.
//
// Wave64 design notes (DESIGN.md has the long form):
//  * packet traversal: one wave = one 8x8 (configurable) tile of the scan image.  The current
//    BVH4 node is WAVE-UNIFORM, so its 128 B are fetched with scalar loads (s_load_dwordx*)
//    into SGPRs and the per-lane slab tests read them as SGPR operands; triangles likewise.
//    The traversal stack is wave-uniform too and lives in ONE VGPR, one entry per lane,
//    pushed / popped with v_writelane / v_readlane.  Descent decisions are v_cmp ballots.
//  * per-lane traversal (incoherent particle-filter rays): per-lane stack in LDS laid out
//    [depth][lane] (bank-conflict free), nodes via global_load_dwordx4.
//  * the ray/triangle arithmetic (tri_accept) is an exact-order fp32 spec shared with the
//    parity oracle; the slab test is free-form but conservative (boxes are padded at build).
#pragma once
#include "kernels.h"

namespace rmclhip {

namespace {

typedef const __attribute__((address_space(4))) uint32_t* cu32p;  // constant AS: uniform loads -> SMEM
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) u32x16* cu32x16p;

// full 5-comparator ordering of the 4 children; the 3-comparator "nearest only" variant measured neutral on the
// sphere and 3-5 % slower on the occluded room / particle filter (profiles/r01d_*)
#define RMCL_FULL_SORT 1

constexpr uint32_t kNone = 0xFFFFFFFFu;

// closest hit of a ray: rec = index of the hit triangle's record, kNone (0xFFFFFFFF) for a miss; the ORIGINAL face id
// lives in dword 15 of that record (the epilogues read it together with the unit normal)
struct RayHit {
  float t;
  uint32_t rec;
};

__device__ __forceinline__ float asf(uint32_t u) { return __uint_as_float(u); }

__device__ __forceinline__ float safe_inv(float d) {
  const float ad = fabsf(d);
  const float s = (ad < 1e-30f) ? copysignf(1e-30f, d) : d;
  return 1.0f / s;
}

// Moeller-Trumbore in Embree's formulation; must match oracle/rmcl_oracle.c:tri_intersect op for op.
// Returns the barycentric acceptance; T/aden is left to the caller (so a packet can skip the divide).
// Depth test of the callers: Embree's near side is STRICT (absDen * tnear < T, tnear = 0 => T > 0): a ray that starts
// exactly on a triangle does not hit it; far side t <= tfar.
__device__ __forceinline__ bool tri_accept(f3 v0, f3 e1, f3 e2, f3 Ng, f3 O, f3 D, float& Tt, float& aden) {
  const f3 C = sub3(v0, O);
  const f3 R = cross_fma(C, D);
  const float den = dot_fma(Ng, D);
  aden = fabsf(den);
  float U = dot_fma(R, e2);
  float V = dot_fma(R, e1);
  Tt = dot_fma(Ng, C);
  if (den < 0.0f) { U = -U; V = -V; Tt = -Tt; }
  return (den != 0.0f) && (U >= 0.0f) && (V >= 0.0f) && ((U + V) <= aden);
}

typedef float f2 __attribute__((ext_vector_type(2)));

// One child's slab test.  px/py/pz = (min, max) plane pair of the child per axis; both plane distances of an
// axis are ONE packed FMA (v_pk_fma_f32).  Free-form arithmetic: conservative because the boxes are padded.
__device__ __forceinline__ void slab(f2 px, f2 py, f2 pz, f3 inv, f3 noi, float best_t, float& tn, float& tf) {
  const f2 ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
  const f2 nx = {noi.x, noi.x}, ny = {noi.y, noi.y}, nz = {noi.z, noi.z};
  const f2 tx = __builtin_elementwise_fma(px, ix, nx);
  const f2 ty = __builtin_elementwise_fma(py, iy, ny);
  const f2 tz = __builtin_elementwise_fma(pz, iz, nz);
  tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), 0.0f));
  tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), best_t));
}

// Per-ray constants of the sign-selected node fetch (layout.h): byte offsets, inside a 128-B node, of the group of
// four planes the ray ENTERS through and of the group it leaves through, per axis.
struct RaySlab {
  f3 inv, noi;
  uint32_t onx, ofx, ony, ofy, onz, ofz;
};

__device__ __forceinline__ RaySlab make_ray_slab(f3 O, f3 D) {
  RaySlab r;
  r.inv = mk3(safe_inv(D.x), safe_inv(D.y), safe_inv(D.z));
  r.noi = mk3(-(O.x * r.inv.x), -(O.y * r.inv.y), -(O.z * r.inv.z));
  r.onx = (r.inv.x < 0.0f) ? 16u : 0u;  r.ofx = 16u - r.onx;
  r.ony = (r.inv.y < 0.0f) ? 48u : 32u; r.ofy = 80u - r.ony;
  r.onz = (r.inv.z < 0.0f) ? 80u : 64u; r.ofz = 144u - r.onz;
  return r;
}

// The four children of inner node `cur` for one lane: seven global_load_dwordx4 (near / far plane groups of the
// three axes + child references), twelve packed FMAs (two children per instruction), then per child one
// max3 / min3 pair.  key = entry distance bits (>= 0, so they order like the floats) or kNone for a miss; unused
// slots hold an unreachable box (layout.h).  Free-form arithmetic: conservative because the boxes are padded.
__device__ __forceinline__ void node_keys_at(const char* nb, const RaySlab& rs, float best_t, uint32_t (&key)[4], uint32_t (&ref)[4]) {
  const uint4 qnx = *reinterpret_cast<const uint4*>(nb + rs.onx), qfx = *reinterpret_cast<const uint4*>(nb + rs.ofx);
  const uint4 qny = *reinterpret_cast<const uint4*>(nb + rs.ony), qfy = *reinterpret_cast<const uint4*>(nb + rs.ofy);
  const uint4 qnz = *reinterpret_cast<const uint4*>(nb + rs.onz), qfz = *reinterpret_cast<const uint4*>(nb + rs.ofz);
  const uint4 qch = *reinterpret_cast<const uint4*>(nb + 96);
  const f2 ix = {rs.inv.x, rs.inv.x}, iy = {rs.inv.y, rs.inv.y}, iz = {rs.inv.z, rs.inv.z};
  const f2 nx = {rs.noi.x, rs.noi.x}, ny = {rs.noi.y, rs.noi.y}, nz = {rs.noi.z, rs.noi.z};
  const f2 nx01 = __builtin_elementwise_fma(f2{asf(qnx.x), asf(qnx.y)}, ix, nx), nx23 = __builtin_elementwise_fma(f2{asf(qnx.z), asf(qnx.w)}, ix, nx);
  const f2 fx01 = __builtin_elementwise_fma(f2{asf(qfx.x), asf(qfx.y)}, ix, nx), fx23 = __builtin_elementwise_fma(f2{asf(qfx.z), asf(qfx.w)}, ix, nx);
  const f2 ny01 = __builtin_elementwise_fma(f2{asf(qny.x), asf(qny.y)}, iy, ny), ny23 = __builtin_elementwise_fma(f2{asf(qny.z), asf(qny.w)}, iy, ny);
  const f2 fy01 = __builtin_elementwise_fma(f2{asf(qfy.x), asf(qfy.y)}, iy, ny), fy23 = __builtin_elementwise_fma(f2{asf(qfy.z), asf(qfy.w)}, iy, ny);
  const f2 nz01 = __builtin_elementwise_fma(f2{asf(qnz.x), asf(qnz.y)}, iz, nz), nz23 = __builtin_elementwise_fma(f2{asf(qnz.z), asf(qnz.w)}, iz, nz);
  const f2 fz01 = __builtin_elementwise_fma(f2{asf(qfz.x), asf(qfz.y)}, iz, nz), fz23 = __builtin_elementwise_fma(f2{asf(qfz.z), asf(qfz.w)}, iz, nz);
  const float tnx[4] = {nx01.x, nx01.y, nx23.x, nx23.y}, tfx[4] = {fx01.x, fx01.y, fx23.x, fx23.y};
  const float tny[4] = {ny01.x, ny01.y, ny23.x, ny23.y}, tfy[4] = {fy01.x, fy01.y, fy23.x, fy23.y};
  const float tnz[4] = {nz01.x, nz01.y, nz23.x, nz23.y}, tfz[4] = {fz01.x, fz01.y, fz23.x, fz23.y};
  ref[0] = qch.x; ref[1] = qch.y; ref[2] = qch.z; ref[3] = qch.w;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float tn = fmaxf(fmaxf(fmaxf(tnx[c], tny[c]), tnz[c]), 0.0f);
    const float tf = fminf(fminf(fminf(tfx[c], tfy[c]), tfz[c]), best_t);
    key[c] = (tn <= tf) ? __float_as_uint(tn) : kNone;
  }
}

__device__ __forceinline__ void node_keys(const uint32_t* __restrict__ nodes, uint32_t cur, const RaySlab& rs, float best_t,
                                          uint32_t (&key)[4], uint32_t (&ref)[4]) {
  node_keys_at(reinterpret_cast<const char*>(nodes) + (static_cast<size_t>(cur) << 7), rs, best_t, key, ref);
}

// LDS-resident top of the tree (north_star: "LDS-staged node tiles"): the nodes are stored breadth-first, so the first
// kTop of them ARE the top levels; every block copies that prefix into its LDS once, and a lane whose current node
// index is below kTop reads it from there.  The choice is per lane and per step, so the node address is a FLAT
// pointer -- LDS aperture or global -- and the seven plane-group loads become flat_load_dwordx4: lanes still in the
// top levels are served by the LDS (~64 cycles), the others by L1/L2 as before, in one instruction stream.
template <int kTop>
__device__ __forceinline__ const char* node_address(const uint32_t* __restrict__ nodes, const uint32_t* lds_top, uint32_t cur) {
  const char* g = reinterpret_cast<const char*>(nodes);
  if (kTop == 0) return g + (static_cast<size_t>(cur) << 7);
  const char* l = reinterpret_cast<const char*>(lds_top);
  return ((cur < static_cast<uint32_t>(kTop)) ? l : g) + (static_cast<size_t>(cur) << 7);
}

// One triangle test of the per-lane traversals from the first three dwordx4 of its record (same arithmetic and acceptance
// as tri_accept's other callers: Tt > 0, t <= tfar, closest = (min t, then min face id)).  The face id is NOT read here:
// it only decides exact ties in t, so the loop tracks the RECORD of the best hit and fetches the two face ids in the
// (rare) tie branch; the caller's epilogue reads the winner's face id together with its normal.  One load fewer per
// triangle, and no dependent load on the hit path.  best_rec == kNone <=> no hit yet.
__device__ __forceinline__ void tri_update(uint4 a, uint4 b, uint4 c, uint32_t rec, const uint32_t* __restrict__ tris, f3 O, f3 D,
                                           float ray_tfar, float& best_t, uint32_t& best_rec) {
  const f3 v0 = mk3(asf(a.x), asf(a.y), asf(a.z));
  const f3 e1 = mk3(asf(a.w), asf(b.x), asf(b.y));
  const f3 e2 = mk3(asf(b.z), asf(b.w), asf(c.x));
  const f3 Ng = mk3(asf(c.y), asf(c.z), asf(c.w));
  float Tt, aden;
  const bool ok = tri_accept(v0, e1, e2, Ng, O, D, Tt, aden);
  if (ok) {
    const float t = Tt / aden;
    const bool acc = (Tt > 0.0f) && (t <= ray_tfar);
    bool closer = acc && (t < best_t);
    if (acc && (t == best_t) && (rec != best_rec)) {
      // exact tie: the smaller ORIGINAL face id wins; a first hit at exactly t == tfar wins against "no hit"
      closer = (best_rec == kNone) || (tris[static_cast<size_t>(rec) * 16u + 15u] < tris[static_cast<size_t>(best_rec) * 16u + 15u]);
    }
    best_t = closer ? t : best_t;
    best_rec = closer ? rec : best_rec;
  }
}

// A whole leaf (<= 4 records) in ONE memory round trip: the loop form waits for triangle i before it requests
// triangle i+1 -- up to four dependent round trips per leaf visit, and the wave runs as many as its fullest leaf has
// triangles.  Here the records of all four slots are requested together, unconditionally (a load under a wave-uniform
// branch makes the compiler wait for it at the end of the branch); a lane whose leaf is shorter re-requests its last
// record (same cache line, and a repeated test cannot change (best_t, best_rec)); the tests run in record order and
// skip slots no lane of the wave fills: results identical to the loop.
// `after_requests` runs between the record requests and the first test (kPipe: the lane's NEXT node is requested there, behind
// the records, so the records' arrival is not held up by it and its latency is covered by the tests)
template <class F>
__device__ __forceinline__ void leaf_batch_then(const uint32_t* __restrict__ tris, uint32_t cur, f3 O, f3 D, float ray_tfar,
                                                float& best_t, uint32_t& best_rec, F&& after_requests) {
  const uint32_t first = cur & 0x0FFFFFFFu;
  const uint32_t last = first + ((cur >> 28) & 7u);
  const bool w2 = __any(last > first), w3 = __any(last > first + 1u), w4 = __any(last > first + 2u);
  const uint32_t i1 = min(first + 1u, last), i2 = min(first + 2u, last), i3 = min(first + 3u, last);
  const uint4* t0 = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(first) * 4u;
  const uint4* t1 = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(i1) * 4u;
  const uint4* t2 = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(i2) * 4u;
  const uint4* t3 = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(i3) * 4u;
  const uint4 a0 = t0[0], b0 = t0[1], c0 = t0[2];
  const uint4 a1 = t1[0], b1 = t1[1], c1 = t1[2];
  const uint4 a2 = t2[0], b2 = t2[1], c2 = t2[2];
  const uint4 a3 = t3[0], b3 = t3[1], c3 = t3[2];
  after_requests();
  tri_update(a0, b0, c0, first, tris, O, D, ray_tfar, best_t, best_rec);
  if (w2) tri_update(a1, b1, c1, i1, tris, O, D, ray_tfar, best_t, best_rec);
  if (w3) tri_update(a2, b2, c2, i2, tris, O, D, ray_tfar, best_t, best_rec);
  if (w4) tri_update(a3, b3, c3, i3, tris, O, D, ray_tfar, best_t, best_rec);
}
__device__ __forceinline__ void leaf_batch(const uint32_t* __restrict__ tris, uint32_t cur, f3 O, f3 D, float ray_tfar,
                                           float& best_t, uint32_t& best_rec) {
  leaf_batch_then(tris, cur, O, D, ray_tfar, best_t, best_rec, [] {});
}

// the loop form of a leaf visit (one record per iteration), same rules
__device__ __forceinline__ void leaf_loop(const uint32_t* __restrict__ tris, uint32_t cur, f3 O, f3 D, float ray_tfar,
                                          float& best_t, uint32_t& best_rec) {
  const uint32_t first = cur & 0x0FFFFFFFu;
  const uint32_t cnt = ((cur >> 28) & 7u) + 1u;
  for (uint32_t i = 0; i < cnt; ++i) {
    const uint4* tp = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(first + i) * 4u;
    const uint4 a = tp[0], b = tp[1], c = tp[2];
    tri_update(a, b, c, first + i, tris, O, D, ray_tfar, best_t, best_rec);
  }
}

// maximum over the wave of a per-lane integer < 64, by six ballots (no cross-lane data movement)
__device__ __forceinline__ uint32_t wave_max_6bit(uint32_t v) {
  v = min(v, 63u);
  uint32_t m = 0;
#pragma unroll
  for (int b = 5; b >= 0; --b) {
    const uint32_t cand = m | (1u << b);
    if (__any(v >= cand)) m = cand;
  }
  return m;
}

// face id of a record (kInvalidFace for "no hit"): one dword of the record's last 16 B
__device__ __forceinline__ uint32_t record_face(const uint32_t* __restrict__ tris, uint32_t rec) {
  return (rec != kNone) ? tris[static_cast<size_t>(rec) * 16u + 15u] : kInvalidFace;
}

// node_keys on the quantised twin of the node (layout.h: Node4Q): FOUR loads instead of seven.  The plane distance
// t = (origin + q*scale - O) * inv is evaluated as q * (scale*inv) + (origin*inv - O*inv): six per-node
// instructions, then one (packed) FMA per plane as before; the bytes are widened with v_cvt_f32_ubyteN.
__device__ __forceinline__ void node_keys_q4(uint4 qa, uint4 qb, uint4 qc, uint4 qch, const RaySlab& rs, float best_t,
                                             uint32_t (&key)[4], uint32_t (&ref)[4]);
__device__ __forceinline__ void node_keys_q(const uint32_t* __restrict__ qnodes, uint32_t cur, const RaySlab& rs, float best_t,
                                            uint32_t (&key)[4], uint32_t (&ref)[4]) {
  const uint4* nb = reinterpret_cast<const uint4*>(qnodes) + static_cast<size_t>(cur) * 4u;
  node_keys_q4(nb[0], nb[1], nb[2], nb[3], rs, best_t, key, ref);
}
__device__ __forceinline__ void node_keys_q4(uint4 qa, uint4 qb, uint4 qc, uint4 qch, const RaySlab& rs, float best_t,
                                             uint32_t (&key)[4], uint32_t (&ref)[4]) {
  const float sx = asf(qa.w) * rs.inv.x, sy = asf(qb.x) * rs.inv.y, sz = asf(qb.y) * rs.inv.z;
  const float bx = fmaf(asf(qa.x), rs.inv.x, rs.noi.x), by = fmaf(asf(qa.y), rs.inv.y, rs.noi.y), bz = fmaf(asf(qa.z), rs.inv.z, rs.noi.z);
  const bool ngx = rs.inv.x < 0.0f, ngy = rs.inv.y < 0.0f, ngz = rs.inv.z < 0.0f;
  const uint32_t qnx = ngx ? qb.w : qb.z, qfx = ngx ? qb.z : qb.w;
  const uint32_t qny = ngy ? qc.y : qc.x, qfy = ngy ? qc.x : qc.y;
  const uint32_t qnz = ngz ? qc.w : qc.z, qfz = ngz ? qc.z : qc.w;
  ref[0] = qch.x; ref[1] = qch.y; ref[2] = qch.z; ref[3] = qch.w;
  // bytes -> floats (v_cvt_f32_ubyteN), two children per packed FMA
#define RMCL_Q2(w, a, b) f2{static_cast<float>(((w) >> (8 * (a))) & 0xFFu), static_cast<float>(((w) >> (8 * (b))) & 0xFFu)}
  const f2 sx2 = {sx, sx}, sy2 = {sy, sy}, sz2 = {sz, sz}, bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
  const f2 nx01 = __builtin_elementwise_fma(RMCL_Q2(qnx, 0, 1), sx2, bx2), nx23 = __builtin_elementwise_fma(RMCL_Q2(qnx, 2, 3), sx2, bx2);
  const f2 fx01 = __builtin_elementwise_fma(RMCL_Q2(qfx, 0, 1), sx2, bx2), fx23 = __builtin_elementwise_fma(RMCL_Q2(qfx, 2, 3), sx2, bx2);
  const f2 ny01 = __builtin_elementwise_fma(RMCL_Q2(qny, 0, 1), sy2, by2), ny23 = __builtin_elementwise_fma(RMCL_Q2(qny, 2, 3), sy2, by2);
  const f2 fy01 = __builtin_elementwise_fma(RMCL_Q2(qfy, 0, 1), sy2, by2), fy23 = __builtin_elementwise_fma(RMCL_Q2(qfy, 2, 3), sy2, by2);
  const f2 nz01 = __builtin_elementwise_fma(RMCL_Q2(qnz, 0, 1), sz2, bz2), nz23 = __builtin_elementwise_fma(RMCL_Q2(qnz, 2, 3), sz2, bz2);
  const f2 fz01 = __builtin_elementwise_fma(RMCL_Q2(qfz, 0, 1), sz2, bz2), fz23 = __builtin_elementwise_fma(RMCL_Q2(qfz, 2, 3), sz2, bz2);
#undef RMCL_Q2
  const float tnx[4] = {nx01.x, nx01.y, nx23.x, nx23.y}, tfx[4] = {fx01.x, fx01.y, fx23.x, fx23.y};
  const float tny[4] = {ny01.x, ny01.y, ny23.x, ny23.y}, tfy[4] = {fy01.x, fy01.y, fy23.x, fy23.y};
  const float tnz[4] = {nz01.x, nz01.y, nz23.x, nz23.y}, tfz[4] = {fz01.x, fz01.y, fz23.x, fz23.y};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float tn = fmaxf(fmaxf(fmaxf(tnx[c], tny[c]), tnz[c]), 0.0f);
    const float tf = fminf(fminf(fminf(tfx[c], tfy[c]), tfz[c]), best_t);
    key[c] = (tn <= tf) ? __float_as_uint(tn) : kNone;
  }
}

// The quantised node in SLOT ORDER (round 5; bvh_build.cpp sorts a node's children along the axis their centres spread most and
// leaves that axis in the low two mantissa bits of the x scale): hit[c] / ref[c] come back in the order THIS ray should visit them --
// slots 0..3 when its direction is positive on the node's axis, 3..0 otherwise (`rev_mask` bit a = direction negative on axis a).
// Nearest-first is only a heuristic here (the closest hit follows from the t bound alone), and this order costs one bit test and four
// selects where the sorting network of five compare-exchanges costs twenty-five instructions.
__device__ __forceinline__ void node_hits_q4_so(uint4 qa, uint4 qb, uint4 qc, uint4 qch, const RaySlab& rs, float best_t, uint32_t rev_mask,
                                                bool (&hit)[4], uint32_t (&ref)[4]) {
  const float sx = asf(qa.w) * rs.inv.x, sy = asf(qb.x) * rs.inv.y, sz = asf(qb.y) * rs.inv.z;
  const float bx = fmaf(asf(qa.x), rs.inv.x, rs.noi.x), by = fmaf(asf(qa.y), rs.inv.y, rs.noi.y), bz = fmaf(asf(qa.z), rs.inv.z, rs.noi.z);
  const bool ngx = rs.inv.x < 0.0f, ngy = rs.inv.y < 0.0f, ngz = rs.inv.z < 0.0f;
  const uint32_t qnx = ngx ? qb.w : qb.z, qfx = ngx ? qb.z : qb.w;
  const uint32_t qny = ngy ? qc.y : qc.x, qfy = ngy ? qc.x : qc.y;
  const uint32_t qnz = ngz ? qc.w : qc.z, qfz = ngz ? qc.z : qc.w;
#define RMCL_Q2(w, a, b) f2{static_cast<float>(((w) >> (8 * (a))) & 0xFFu), static_cast<float>(((w) >> (8 * (b))) & 0xFFu)}
  const f2 sx2 = {sx, sx}, sy2 = {sy, sy}, sz2 = {sz, sz}, bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
  const f2 nx01 = __builtin_elementwise_fma(RMCL_Q2(qnx, 0, 1), sx2, bx2), nx23 = __builtin_elementwise_fma(RMCL_Q2(qnx, 2, 3), sx2, bx2);
  const f2 fx01 = __builtin_elementwise_fma(RMCL_Q2(qfx, 0, 1), sx2, bx2), fx23 = __builtin_elementwise_fma(RMCL_Q2(qfx, 2, 3), sx2, bx2);
  const f2 ny01 = __builtin_elementwise_fma(RMCL_Q2(qny, 0, 1), sy2, by2), ny23 = __builtin_elementwise_fma(RMCL_Q2(qny, 2, 3), sy2, by2);
  const f2 fy01 = __builtin_elementwise_fma(RMCL_Q2(qfy, 0, 1), sy2, by2), fy23 = __builtin_elementwise_fma(RMCL_Q2(qfy, 2, 3), sy2, by2);
  const f2 nz01 = __builtin_elementwise_fma(RMCL_Q2(qnz, 0, 1), sz2, bz2), nz23 = __builtin_elementwise_fma(RMCL_Q2(qnz, 2, 3), sz2, bz2);
  const f2 fz01 = __builtin_elementwise_fma(RMCL_Q2(qfz, 0, 1), sz2, bz2), fz23 = __builtin_elementwise_fma(RMCL_Q2(qfz, 2, 3), sz2, bz2);
#undef RMCL_Q2
  const float tnx[4] = {nx01.x, nx01.y, nx23.x, nx23.y}, tfx[4] = {fx01.x, fx01.y, fx23.x, fx23.y};
  const float tny[4] = {ny01.x, ny01.y, ny23.x, ny23.y}, tfy[4] = {fy01.x, fy01.y, fy23.x, fy23.y};
  const float tnz[4] = {nz01.x, nz01.y, nz23.x, nz23.y}, tfz[4] = {fz01.x, fz01.y, fz23.x, fz23.y};
  bool h[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float tn = fmaxf(fmaxf(fmaxf(tnx[c], tny[c]), tnz[c]), 0.0f);
    const float tf = fminf(fminf(fminf(tfx[c], tfy[c]), tfz[c]), best_t);
    h[c] = tn <= tf;
  }
  const bool rev = ((rev_mask >> (qa.w & 3u)) & 1u) != 0u;
  hit[0] = rev ? h[3] : h[0]; hit[1] = rev ? h[2] : h[1]; hit[2] = rev ? h[1] : h[2]; hit[3] = rev ? h[0] : h[3];
  ref[0] = rev ? qch.w : qch.x; ref[1] = rev ? qch.z : qch.y; ref[2] = rev ? qch.y : qch.z; ref[3] = rev ? qch.x : qch.w;
}
__device__ __forceinline__ void node_hits_q_so(const uint32_t* __restrict__ qnodes, uint32_t cur, const RaySlab& rs, float best_t, uint32_t rev_mask,
                                               bool (&hit)[4], uint32_t (&ref)[4]) {
  const uint4* nb = reinterpret_cast<const uint4*>(qnodes) + static_cast<size_t>(cur) * 4u;
  node_hits_q4_so(nb[0], nb[1], nb[2], nb[3], rs, best_t, rev_mask, hit, ref);
}

#define RMCL_CSWAP(i, j)                                   \
  {                                                        \
    const bool sw_ = key[j] < key[i];                      \
    const uint32_t ka_ = sw_ ? key[j] : key[i];            \
    const uint32_t kb_ = sw_ ? key[i] : key[j];            \
    const uint32_t ra_ = sw_ ? ref[j] : ref[i];            \
    const uint32_t rb_ = sw_ ? ref[i] : ref[j];            \
    key[i] = ka_; key[j] = kb_; ref[i] = ra_; ref[j] = rb_; \
  }

// ---------------------------------------------------------------------------------------------
// packet traversal: wave-uniform node, scalar loads, stack in a VGPR (needs stack_need <= 64)
// ray_tfar < 0 marks an inactive lane.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void trace_packet(cu32p nodes, cu32p tris, f3 O, f3 D, float ray_tfar, uint32_t lane, RayHit& h) {
  const f3 inv = mk3(safe_inv(D.x), safe_inv(D.y), safe_inv(D.z));
  const f3 noi = mk3(-(O.x * inv.x), -(O.y * inv.y), -(O.z * inv.z));
  float best_t = ray_tfar;
  uint32_t best_face = kInvalidFace, best_rec = 0;

  int stk = 0;       // 64 wave-uniform entries, entry i in lane i
  uint32_t sp = 0;   // uniform
  uint32_t cur = 0;  // uniform; root is always an inner node
  for (;;) {
    if (!(cur & kLeafBit)) {
      // whole node in two s_load_dwordx16: dwords 0..15 = x and y plane groups; 16..31 = z groups, child[4], count
      const cu32x16p np = reinterpret_cast<cu32x16p>(nodes + cur * kNodeDwords);
      const u32x16 lo = np[0], hi = np[1];
      uint32_t key[4], ref[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float tn, tf;
        const f2 px = {asf(lo[c]), asf(lo[4 + c])}, py = {asf(lo[8 + c]), asf(lo[12 + c])};
        const f2 pz = {asf(hi[c]), asf(hi[4 + c])};
        slab(px, py, pz, inv, noi, best_t, tn, tf);
        ref[c] = hi[8 + c];
        const uint64_t m = __ballot(tn <= tf);  // unused slots hold an unreachable box (layout.h)
        uint32_t k = kNone;
        if (m != 0) {
          const int first = __builtin_ctzll(m);
          k = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(tn)), first));
        }
        key[c] = k;
      }
      RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
      if (key[3] != kNone) { stk = (lane == sp) ? static_cast<int>(ref[3]) : stk; ++sp; }
      if (key[2] != kNone) { stk = (lane == sp) ? static_cast<int>(ref[2]) : stk; ++sp; }
      if (key[1] != kNone) { stk = (lane == sp) ? static_cast<int>(ref[1]) : stk; ++sp; }
      if (key[0] != kNone) { cur = ref[0]; continue; }
    } else {
      const uint32_t first = cur & 0x0FFFFFFFu;
      const uint32_t cnt = ((cur >> 28) & 7u) + 1u;
      // the whole leaf (<= 4 records, 256 contiguous bytes) is requested at once: four s_load_dwordx16 in
      // flight cost one scalar-cache round trip instead of four (the record array is padded by 3 records)
      const cu32x16p tp = reinterpret_cast<cu32x16p>(tris + first * kTriDwords);
      const u32x16 trs[4] = {tp[0], tp[1], tp[2], tp[3]};
#pragma unroll
      for (uint32_t i = 0; i < kMaxLeafTris; ++i) {
        if (i < cnt) {
          const u32x16 tr = trs[i];
          const f3 v0 = mk3(asf(tr[0]), asf(tr[1]), asf(tr[2]));
          const f3 e1 = mk3(asf(tr[3]), asf(tr[4]), asf(tr[5]));
          const f3 e2 = mk3(asf(tr[6]), asf(tr[7]), asf(tr[8]));
          const f3 Ng = mk3(asf(tr[9]), asf(tr[10]), asf(tr[11]));
          const uint32_t face = tr[15];
          float Tt, aden;
          const bool ok = tri_accept(v0, e1, e2, Ng, O, D, Tt, aden);
          if (__ballot(ok) != 0) {
            const float t = Tt / aden;
            const bool acc = ok && (Tt > 0.0f) && (t <= ray_tfar);
            const bool closer = acc && ((t < best_t) || ((t == best_t) && (face < best_face)));
            best_t = closer ? t : best_t;
            best_face = closer ? face : best_face;
            best_rec = closer ? (first + i) : best_rec;
          }
        }
      }
    }
    if (sp == 0) break;
    --sp;
    cur = static_cast<uint32_t>(__builtin_amdgcn_readlane(stk, sp));
  }
  h.t = best_t;
  h.rec = (best_face != kInvalidFace) ? best_rec : kNone;
}

// ---------------------------------------------------------------------------------------------
// per-lane traversal: every lane walks its own path; stack in LDS [depth][blockDim]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void trace_lane(const uint32_t* __restrict__ nodes, const uint32_t* __restrict__ tris,
                                           f3 O, f3 D, float ray_tfar, uint32_t* __restrict__ lds_stack,
                                           uint32_t lds_stride, RayHit& h) {
  const RaySlab rs = make_ray_slab(O, D);
  float best_t = ray_tfar;
  uint32_t best_face = kInvalidFace, best_rec = 0;
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  uint32_t sp = 0;
  uint32_t cur = (ray_tfar >= 0.0f) ? 0u : kDone;
  while (cur != kDone) {
    if (!(cur & kLeafBit)) {
      uint32_t key[4], ref[4];
      node_keys(nodes, cur, rs, best_t, key, ref);
      RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
      if (key[3] != kNone) { lds_stack[sp * lds_stride] = ref[3]; ++sp; }
      if (key[2] != kNone) { lds_stack[sp * lds_stride] = ref[2]; ++sp; }
      if (key[1] != kNone) { lds_stack[sp * lds_stride] = ref[1]; ++sp; }
      if (key[0] != kNone) { cur = ref[0]; continue; }
    } else {
      const uint32_t first = cur & 0x0FFFFFFFu;
      const uint32_t cnt = ((cur >> 28) & 7u) + 1u;
      for (uint32_t i = 0; i < cnt; ++i) {
        const uint4* tp = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(first + i) * 4u;
        const uint4 a = tp[0], b = tp[1], c = tp[2], d = tp[3];
        const f3 v0 = mk3(asf(a.x), asf(a.y), asf(a.z));
        const f3 e1 = mk3(asf(a.w), asf(b.x), asf(b.y));
        const f3 e2 = mk3(asf(b.z), asf(b.w), asf(c.x));
        const f3 Ng = mk3(asf(c.y), asf(c.z), asf(c.w));
        const uint32_t face = d.w;
        float Tt, aden;
        const bool ok = tri_accept(v0, e1, e2, Ng, O, D, Tt, aden);
        if (ok) {
          const float t = Tt / aden;
          const bool acc = (Tt > 0.0f) && (t <= ray_tfar);
          const bool closer = acc && ((t < best_t) || ((t == best_t) && (face < best_face)));
          best_t = closer ? t : best_t;
          best_face = closer ? face : best_face;
          best_rec = closer ? (first + i) : best_rec;
        }
      }
    }
    if (sp == 0) { cur = kDone; }
    else { --sp; cur = lds_stack[sp * lds_stride]; }
  }
  h.t = best_t;
  h.rec = (best_face != kInvalidFace) ? best_rec : kNone;
}

// ---------------------------------------------------------------------------------------------
// per-lane traversal, "while-while" form (Aila & Laine): every lane first descends through inner nodes
// until it holds a leaf; only then does the wave run the (expensive) triangle tests, with most lanes
// active.  The per-lane stack is split: the first kLdsEntries live in LDS ([entry][lane], conflict free), deeper
// entries spill to private (scratch) memory.  A full 64-deep LDS stack costs 64 KB per 256-thread block (2 blocks
// per CU); a pure scratch stack keeps occupancy but measured 272 MB of HBM-side write traffic per C4 update; 16
// LDS entries (16 KB per block) catch almost every push.
// ---------------------------------------------------------------------------------------------
// kQuant: `nodes` points to the quantised Node4Q twins (four loads per node visit instead of seven)
// Where a ray starts when the top levels of its descent have been replaced by the frontier start (frontier_start below): the
// node it enters first and the number of further entries already on its LDS stack.
struct TraceStart {
  uint32_t cur, sp;
};

template <int kLdsEntries, bool kQuant = false, bool kLeafBatch = false, bool kVote = false>  // stack entries kept in LDS ([entry][lane]); the rest (up to 64 total) in scratch
__device__ __forceinline__ void trace_lane_ww(const uint32_t* __restrict__ nodes, const uint32_t* __restrict__ tris,
                                              f3 O, f3 D, float ray_tfar, uint32_t* __restrict__ lds_stack,
                                              uint32_t lds_stride, RayHit& h, const TraceStart* start = nullptr) {
  const RaySlab rs = make_ray_slab(O, D);
  float best_t = ray_tfar;
  uint32_t best_rec = kNone;
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  uint32_t priv[(kLdsEntries < 64) ? (64 - kLdsEntries) : 1];
  uint32_t sp = start ? start->sp : 0u;
  uint32_t cur = start ? start->cur : ((ray_tfar >= 0.0f) ? 0u : kDone);
#define RMCL_PUSH(v) { if (kLdsEntries >= 64 || sp < kLdsEntries) lds_stack[sp * lds_stride] = (v); else priv[sp - kLdsEntries] = (v); ++sp; }
#define RMCL_POP() { if (sp == 0) cur = kDone; else { --sp; if (kLdsEntries >= 64 || sp < kLdsEntries) cur = lds_stack[sp * lds_stride]; else cur = priv[sp - kLdsEntries]; } }
  for (;;) {
    const uint64_t m_act = __ballot(cur != kDone);
    if (m_act == 0) break;
    const uint32_t na = static_cast<uint32_t>(__popcll(m_act));   // rays alive at the start of this round
    // phase 1: inner nodes (kVote: the leaf trigger, see trace_lane_bf_tail)
    while ((cur != kDone) && !(cur & kLeafBit)) {
      uint32_t key[4], ref[4];
      if (kQuant) node_keys_q(nodes, cur, rs, best_t, key, ref);
      else node_keys(nodes, cur, rs, best_t, key, ref);
#ifdef RMCL_FULL_SORT
      RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
#else
      RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2)  // nearest child to slot 0; the deferred ones stay unordered
#endif
      if (key[3] != kNone) RMCL_PUSH(ref[3])
      if (key[2] != kNone) RMCL_PUSH(ref[2])
      if (key[1] != kNone) RMCL_PUSH(ref[1])
      if (key[0] != kNone) cur = ref[0];
      else RMCL_POP()
      if (kVote) {
        if (5u * static_cast<uint32_t>(__popcll(__ballot((cur != kDone) && !(cur & kLeafBit)))) <= 2u * na) break;
      }
    }
    // phase 2: this lane's leaf (if any)
    if ((cur != kDone) && (cur & kLeafBit)) {
      if (kLeafBatch) leaf_batch(tris, cur, O, D, ray_tfar, best_t, best_rec);
      else leaf_loop(tris, cur, O, D, ray_tfar, best_t, best_rec);
      RMCL_POP()
    }
  }
#undef RMCL_PUSH
#undef RMCL_POP
  h.t = best_t;
  h.rec = best_rec;
}

// ---------------------------------------------------------------------------------------------
// per-lane while-while traversal, BRANCH-FREE node step (the form every one-lane-per-ray kernel now uses).
// In vivo a node step of trace_lane_ww costs ~1300 cycles on a chip full of C2 waves (tools/wave_timeline.py) and ~880 for
// a lone wave (tools/probe_find.py: ~430 waiting for the node + ~450 of issue) -- and the ISSUE half was mostly control:
// every conditional push is a v_cmp + s_and_saveexec + branch + (LDS-or-scratch test, another saveexec pair) + a 32-bit
// multiply for `sp * stride`; the "nothing hit: pop" arm is another nest.  Here
//   * the stack row stride is the compile-time block size (shift-add addressing, no v_mul_lo_u32),
//   * the three deferred children are stored UNCONDITIONALLY at rows sp, sp', sp'' with sp advancing only past real hits
//     (the children are sorted, misses last, so a miss is overwritten by the next store or lands above the top),
//   * row 0 holds the sentinel kDone and the current top of the stack is fetched speculatively with the node, so
//     "no child hit -> pop" is two selects; an empty stack ends the ray without a test,
//   * node data is addressed as SGPR base + 32-bit lane offset (one v_lshl_add_u32 per load instead of 64-bit adds).
// Rows >= kRows live in private scratch as before; a wave whose lanes might touch them in this step (wave-uniform
// test) takes the general path.  Same visits, same arithmetic, same results as trace_lane_ww.
// ---------------------------------------------------------------------------------------------
// slab tests + keys of the four children from the seven 16-B groups of a node (near / far plane groups per axis + refs)
__device__ __forceinline__ void node_keys_from(uint4 qnx, uint4 qfx, uint4 qny, uint4 qfy, uint4 qnz, uint4 qfz, uint4 qch,
                                               const RaySlab& rs, float best_t, uint32_t (&key)[4], uint32_t (&ref)[4]) {
  const f2 ix = {rs.inv.x, rs.inv.x}, iy = {rs.inv.y, rs.inv.y}, iz = {rs.inv.z, rs.inv.z};
  const f2 nx = {rs.noi.x, rs.noi.x}, ny = {rs.noi.y, rs.noi.y}, nz = {rs.noi.z, rs.noi.z};
  const f2 nx01 = __builtin_elementwise_fma(f2{asf(qnx.x), asf(qnx.y)}, ix, nx), nx23 = __builtin_elementwise_fma(f2{asf(qnx.z), asf(qnx.w)}, ix, nx);
  const f2 fx01 = __builtin_elementwise_fma(f2{asf(qfx.x), asf(qfx.y)}, ix, nx), fx23 = __builtin_elementwise_fma(f2{asf(qfx.z), asf(qfx.w)}, ix, nx);
  const f2 ny01 = __builtin_elementwise_fma(f2{asf(qny.x), asf(qny.y)}, iy, ny), ny23 = __builtin_elementwise_fma(f2{asf(qny.z), asf(qny.w)}, iy, ny);
  const f2 fy01 = __builtin_elementwise_fma(f2{asf(qfy.x), asf(qfy.y)}, iy, ny), fy23 = __builtin_elementwise_fma(f2{asf(qfy.z), asf(qfy.w)}, iy, ny);
  const f2 nz01 = __builtin_elementwise_fma(f2{asf(qnz.x), asf(qnz.y)}, iz, nz), nz23 = __builtin_elementwise_fma(f2{asf(qnz.z), asf(qnz.w)}, iz, nz);
  const f2 fz01 = __builtin_elementwise_fma(f2{asf(qfz.x), asf(qfz.y)}, iz, nz), fz23 = __builtin_elementwise_fma(f2{asf(qfz.z), asf(qfz.w)}, iz, nz);
  const float tnx[4] = {nx01.x, nx01.y, nx23.x, nx23.y}, tfx[4] = {fx01.x, fx01.y, fx23.x, fx23.y};
  const float tny[4] = {ny01.x, ny01.y, ny23.x, ny23.y}, tfy[4] = {fy01.x, fy01.y, fy23.x, fy23.y};
  const float tnz[4] = {nz01.x, nz01.y, nz23.x, nz23.y}, tfz[4] = {fz01.x, fz01.y, fz23.x, fz23.y};
  ref[0] = qch.x; ref[1] = qch.y; ref[2] = qch.z; ref[3] = qch.w;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float tn = fmaxf(fmaxf(fmaxf(tnx[c], tny[c]), tnz[c]), 0.0f);
    const float tf = fminf(fminf(fminf(tfx[c], tfy[c]), tfz[c]), best_t);
    key[c] = (tn <= tf) ? __float_as_uint(tn) : kNone;
  }
}

__device__ __forceinline__ void node_keys_off(const uint32_t* __restrict__ nodes, uint32_t byte_off, const RaySlab& rs, float best_t,
                                              uint32_t (&key)[4], uint32_t (&ref)[4]) {
  // uniform base + zero-extended 32-bit offsets (map_create bounds the node array below 4 GB)
  const char* nb = reinterpret_cast<const char*>(nodes);
  const uint4 qnx = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.onx)), qfx = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.ofx));
  const uint4 qny = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.ony)), qfy = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.ofy));
  const uint4 qnz = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.onz)), qfz = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.ofz));
  const uint4 qch = *reinterpret_cast<const uint4*>(nb + (byte_off + 96u));
  node_keys_from(qnx, qfx, qny, qfy, qnz, qfz, qch, rs, best_t, key, ref);
}

// WAVE-UNIFORM node (north_star: "wavefront ballot for packet traversal"): ~60 % of the node steps of a C2 scan are taken by
// a wave whose active lanes all stand on the SAME node (the top of the tree, tools/probe_find.py).  A single scan is
// bound by the vector memory pipeline -- 7 x 16 B x 64 lanes = 7 KB through the 64 B/clk texture path per wave and
// step, 8 waves per CU -- so such a step fetches the node ONCE with scalar loads (scalar cache, not the vector path) into
// SGPRs and the lanes read the planes as scalar operands.  When the wave's rays also share the sign octant of their
// direction (every tile that does not straddle a coordinate plane) the near / far plane groups are selected by scalar
// address arithmetic exactly like the per-lane offsets, so the arithmetic -- and therefore keys, order and results --
// is identical to the vector path.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4* cu32x4p;

__device__ __forceinline__ uint4 sload4(const uint32_t* __restrict__ base, uint32_t byte_off) {
  const u32x4 v = *reinterpret_cast<cu32x4p>(reinterpret_cast<const __attribute__((address_space(4))) char*>((cu32p)(base)) + byte_off);
  return uint4{v.x, v.y, v.z, v.w};
}

// octant offsets of the wave (uniform): same values as RaySlab's per-lane ones
struct WaveOctant {
  uint32_t onx, ofx, ony, ofy, onz, ofz;
};

__device__ __forceinline__ void node_keys_uniform(const uint32_t* __restrict__ nodes, uint32_t cur_uniform, const WaveOctant& wo,
                                                  const RaySlab& rs, float best_t, uint32_t (&key)[4], uint32_t (&ref)[4]) {
  const uint32_t b = cur_uniform << 7;
  const uint4 qnx = sload4(nodes, b + wo.onx), qfx = sload4(nodes, b + wo.ofx);
  const uint4 qny = sload4(nodes, b + wo.ony), qfy = sload4(nodes, b + wo.ofy);
  const uint4 qnz = sload4(nodes, b + wo.onz), qfz = sload4(nodes, b + wo.ofz);
  const uint4 qch = sload4(nodes, b + 96u);
  node_keys_from(qnx, qfx, qny, qfy, qnz, qfz, qch, rs, best_t, key, ref);
}

// A node REQUESTED ahead of its use (kPipe of trace_lane_bf_tail): the seven sign-selected 16-B groups of node_keys_off in registers
struct NodeRegs { uint4 qnx, qfx, qny, qfy, qnz, qfz, qch; };
__device__ __forceinline__ NodeRegs node_req(const uint32_t* __restrict__ nodes, uint32_t byte_off, const RaySlab& rs) {
  const char* nb = reinterpret_cast<const char*>(nodes);
  NodeRegs r;
  r.qnx = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.onx)); r.qfx = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.ofx));
  r.qny = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.ony)); r.qfy = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.ofy));
  r.qnz = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.onz)); r.qfz = *reinterpret_cast<const uint4*>(nb + (byte_off + rs.ofz));
  r.qch = *reinterpret_cast<const uint4*>(nb + (byte_off + 96u));
  return r;
}

constexpr uint32_t kBfStride = 256u;  // stack row stride in dwords = threads per block of every kernel that calls trace_lane_bf

// kRows: stack rows in LDS per lane INCLUDING the sentinel row 0 (row r of this lane at lds_col[r * 256]); deeper entries
// (up to 64 in total, the builder's bound) in scratch
template <int kRows, bool kQuant = false, bool kLeafBatch = false, bool kUniform = false>
__device__ __forceinline__ void trace_lane_bf(const uint32_t* __restrict__ nodes, const uint32_t* __restrict__ tris, f3 O, f3 D,
                                              float ray_tfar, uint32_t* __restrict__ lds_col, RayHit& h, uint32_t* visits = nullptr) {
  uint32_t nvis = 0;  // node visits of this ray
  const RaySlab rs = make_ray_slab(O, D);
  // do all rays of the wave share the sign octant of their direction?  (lanes without a ray do not vote)
  WaveOctant wo = {0u, 0u, 0u, 0u, 0u, 0u};
  bool uni_oct = false;
  if (kUniform && !kQuant) {
    const bool live = ray_tfar >= 0.0f;
    const uint32_t oct = (rs.inv.x < 0.0f ? 1u : 0u) | (rs.inv.y < 0.0f ? 2u : 0u) | (rs.inv.z < 0.0f ? 4u : 0u);
    const uint64_t m_live = __ballot(live);
    if (m_live != 0) {
      const uint32_t o0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(oct), __builtin_ctzll(m_live)));
      uni_oct = __ballot(live && oct != o0) == 0;
      wo.onx = (o0 & 1u) ? 16u : 0u;  wo.ofx = 16u - wo.onx;
      wo.ony = (o0 & 2u) ? 48u : 32u; wo.ofy = 80u - wo.ony;
      wo.onz = (o0 & 4u) ? 80u : 64u; wo.ofz = 144u - wo.onz;
    }
  }
  float best_t = ray_tfar;
  uint32_t best_rec = kNone;
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  uint32_t priv[(kRows < 65) ? (65 - kRows) : 1];
  lds_col[0] = kDone;  // sentinel
  uint32_t sp = 1;     // first free row
  uint32_t cur = (ray_tfar >= 0.0f) ? 0u : kDone;
  // general row access (rare path)
#define RMCL_ROW_ST(r, v) { if ((r) < static_cast<uint32_t>(kRows)) lds_col[(r) * kBfStride] = (v); else priv[(r) - kRows] = (v); }
#define RMCL_ROW_LD(r) (((r) < static_cast<uint32_t>(kRows)) ? lds_col[(r) * kBfStride] : priv[(r) - kRows])
  while (__any(cur != kDone)) {
    // phase 1: inner nodes (cur < kDone <=> inner node: leaf references have bit 31 set)
    while (cur < kDone) {
      uint32_t key[4], ref[4];
      ++nvis;
      if (!__any(sp + 3u > static_cast<uint32_t>(kRows))) {
        // ---- fast path: every row this step can touch is in LDS (wave-uniform) ----
        const uint32_t top = lds_col[(sp - 1u) * kBfStride];
        const uint32_t c0 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(cur)));
        if (kUniform && !kQuant && uni_oct && !__any(cur != c0)) node_keys_uniform(nodes, c0, wo, rs, best_t, key, ref);
        else if (kQuant) node_keys_q(nodes, cur, rs, best_t, key, ref);
        else node_keys_off(nodes, cur << 7, rs, best_t, key, ref);
        RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
        lds_col[sp * kBfStride] = ref[3]; sp += (key[3] != kNone) ? 1u : 0u;
        lds_col[sp * kBfStride] = ref[2]; sp += (key[2] != kNone) ? 1u : 0u;
        lds_col[sp * kBfStride] = ref[1]; sp += (key[1] != kNone) ? 1u : 0u;
        const bool any = key[0] != kNone;
        cur = any ? ref[0] : top;
        sp = any ? sp : (sp - 1u);
      } else {
        if (kQuant) node_keys_q(nodes, cur, rs, best_t, key, ref);
        else node_keys_off(nodes, cur << 7, rs, best_t, key, ref);
        RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
        if (key[3] != kNone) { RMCL_ROW_ST(sp, ref[3]) ++sp; }
        if (key[2] != kNone) { RMCL_ROW_ST(sp, ref[2]) ++sp; }
        if (key[1] != kNone) { RMCL_ROW_ST(sp, ref[1]) ++sp; }
        if (key[0] != kNone) cur = ref[0];
        else { --sp; cur = RMCL_ROW_LD(sp); }
      }
    }
    // phase 2: this lane's leaf (if any)
    if (cur != kDone) {
      if (kLeafBatch) leaf_batch(tris, cur, O, D, ray_tfar, best_t, best_rec);
      else leaf_loop(tris, cur, O, D, ray_tfar, best_t, best_rec);
      --sp;
      cur = RMCL_ROW_LD(sp);
    }
  }
#undef RMCL_ROW_ST
#undef RMCL_ROW_LD
  h.t = best_t;
  h.rec = best_rec;
  if (visits) *visits = nvis;
}

// ---------------------------------------------------------------------------------------------
// quad-cooperative traversal: FOUR lanes per ray, lane c of the quad owns child slot c of the current node and
// triangle c of the current leaf (leaves hold <= 4 triangles).  A single scan is bound by the slowest ray's chain
// of dependent node fetches (tools/latency_explore.py: one wave alone takes 2/3 of the full scan's time), so the
// work of one step is spread over four lanes: one slab test instead of four, a rank computation over DPP
// quad_perm instead of a sorting network, up to four triangle tests at once.  The quad's stack (64 entries, the
// builder's bound) lives in LDS.  Same acceptance rules and tie-break as trace_lane_ww: identical results.
// ---------------------------------------------------------------------------------------------
template <int kCtrl>
__device__ __forceinline__ uint32_t quad_dpp(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), kCtrl, 0xF, 0xF, true));
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kQuadXor3 = 0x1B;  // quad_perm [1,0,3,2] [2,3,0,1] [3,2,1,0]
constexpr uint32_t kQuadStackEntries = 1u + 64u + 4u;            // sentinel + the builder's bound + scratch rows

// A traversal in progress, handed from one lane to a quad (see trace_lane_ww_tail): current node, number of stack
// entries already stored in rows 1..n_stack of the quad's column, and the best hit so far.
struct QuadResume {
  uint32_t cur, n_stack;
  float best_t;
  uint32_t best_face, best_rec;
};

template <bool kResume = false>
__device__ __forceinline__ void trace_quad(const uint32_t* __restrict__ nodes, const uint32_t* __restrict__ tris, f3 O,
                                           f3 D, float ray_tfar, uint32_t c, uint32_t ray, uint32_t* __restrict__ lds,
                                           RayHit& h, const QuadResume* resume = nullptr, uint32_t* visits = nullptr) {
  uint32_t nvis = 0;  // node visits of this ray (the mixed launch's cost measure)
  const RaySlab rs = make_ray_slab(O, D);
  float best_t = kResume ? resume->best_t : ray_tfar;
  uint32_t best_face = kResume ? resume->best_face : kInvalidFace, best_rec = kResume ? resume->best_rec : 0u;
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  // The quad's stack: entry e of ray r at byte (e*64 + r)*4 of `lds` (kQuadStackEntries rows).  Row 0 holds the
  // sentinel kDone, so that popping an empty stack ends the ray without a test; rows above the top are scratch:
  // every lane stores its child reference every step (deferred children below the new top, the rest above it)
  // and the top of the stack is fetched speculatively together with the node -- no branch in a node step.
  const char* nbase = reinterpret_cast<const char*>(nodes);
  char* sbase = reinterpret_cast<char*>(lds) + ray * 4u;
  if (c == 0u) *reinterpret_cast<uint32_t*>(sbase) = kDone;
  uint32_t spb = kResume ? ((resume->n_stack + 1u) << 8) : 256u;  // byte offset of the first free row
  uint32_t cur = (ray_tfar >= 0.0f) ? (kResume ? resume->cur : 0u) : kDone;
  // this lane's child inside a child-major node (layout.h: Node4C): 32 B = two dwordx4
  const uint32_t coff = c * 32u;
  const bool ngx = rs.inv.x < 0.0f, ngy = rs.inv.y < 0.0f, ngz = rs.inv.z < 0.0f;
  while (__any(cur != kDone)) {
    while (cur < kDone) {  // inner node (leaf references have bit 31 set)
      const uint4* nd = reinterpret_cast<const uint4*>(nbase + (cur << 7) + coff);
      const uint4 q0 = nd[0], q1 = nd[1];  // lo.x lo.y lo.z hi.x | hi.y hi.z ref pad
      const uint32_t top = *reinterpret_cast<const uint32_t*>(sbase + (spb - 256u));
      ++nvis;
      const float pnx = ngx ? asf(q0.w) : asf(q0.x), pfx = ngx ? asf(q0.x) : asf(q0.w);
      const float pny = ngy ? asf(q1.x) : asf(q0.y), pfy = ngy ? asf(q0.y) : asf(q1.x);
      const float pnz = ngz ? asf(q1.y) : asf(q0.z), pfz = ngz ? asf(q0.z) : asf(q1.y);
      const uint32_t ref = q1.z;
      const float tn = fmaxf(fmaxf(fmaxf(fmaf(pnx, rs.inv.x, rs.noi.x), fmaf(pny, rs.inv.y, rs.noi.y)), fmaf(pnz, rs.inv.z, rs.noi.z)), 0.0f);
      const float tf = fminf(fminf(fminf(fmaf(pfx, rs.inv.x, rs.noi.x), fmaf(pfy, rs.inv.y, rs.noi.y)), fmaf(pfz, rs.inv.z, rs.noi.z)), best_t);
      // unique keys: entry distance with the slot number in the two low mantissa bits; misses (unused slots hold
      // an unreachable box, layout.h) sort last
      const uint32_t key = ((tn <= tf) ? (__float_as_uint(tn) & ~3u) : 0xFFFFFFFCu) | c;
      const uint32_t k1 = quad_dpp<kQuadXor1>(key), k2 = quad_dpp<kQuadXor2>(key), k3 = quad_dpp<kQuadXor3>(key);
      const uint32_t rank = (k1 < key ? 1u : 0u) + (k2 < key ? 1u : 0u) + (k3 < key ? 1u : 0u);
      const uint32_t kmin = min(min(key, k1), min(k2, k3));
      // number of hits = 4 - misses; all four keys are known to every lane
      const uint32_t nh = (key < 0xFFFFFFFCu ? 1u : 0u) + (k1 < 0xFFFFFFFCu ? 1u : 0u) + (k2 < 0xFFFFFFFCu ? 1u : 0u) +
                          (k3 < 0xFFFFFFFCu ? 1u : 0u);
      const uint32_t sel = (key == kmin) ? ref : 0u;
      const uint32_t s1 = sel | quad_dpp<kQuadXor1>(sel);
      const uint32_t nearest = s1 | quad_dpp<kQuadXor2>(s1);
      // rows: deferred hits (rank 1..nh-1) at spb + (nh-1-rank), second nearest on top; rank 0 and the misses land
      // in the scratch rows at or above the new top
      const uint32_t row = (rank < nh) ? (nh - 1u - rank) : rank;
      *reinterpret_cast<uint32_t*>(sbase + spb + (row << 8)) = ref;
      const bool any = nh != 0u;
      cur = any ? nearest : top;
      spb = any ? (spb + ((nh - 1u) << 8)) : (spb - 256u);
    }
    if (cur != kDone) {
      const uint32_t first = cur & 0x0FFFFFFFu;
      const uint32_t cnt = ((cur >> 28) & 7u) + 1u;  // <= kMaxLeafTris = 4
      const uint32_t idx = first + ((c < cnt) ? c : (cnt - 1u));
      const uint4* tp = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(idx) * 4u;
      const uint4 a = tp[0], b = tp[1], cc = tp[2], d = tp[3];
      const uint32_t top = *reinterpret_cast<const uint32_t*>(sbase + (spb - 256u));
      const f3 v0 = mk3(asf(a.x), asf(a.y), asf(a.z));
      const f3 e1 = mk3(asf(a.w), asf(b.x), asf(b.y));
      const f3 e2 = mk3(asf(b.z), asf(b.w), asf(cc.x));
      const f3 Ng = mk3(asf(cc.y), asf(cc.z), asf(cc.w));
      float Tt, aden;
      const bool ok = tri_accept(v0, e1, e2, Ng, O, D, Tt, aden);
      const float t = Tt / aden;
      const bool acc = ok && (c < cnt) && (Tt > 0.0f) && (t <= ray_tfar);
      // quad minimum of (t, face): candidates that fail carry (+inf, invalid face) and never win
      float ct = acc ? t : __builtin_inff();
      uint32_t cf = acc ? d.w : kInvalidFace, cr = idx;
      {
        const float ot = __uint_as_float(quad_dpp<kQuadXor1>(__float_as_uint(ct)));
        const uint32_t of = quad_dpp<kQuadXor1>(cf), orr = quad_dpp<kQuadXor1>(cr);
        const bool take = (ot < ct) || ((ot == ct) && (of < cf));
        ct = take ? ot : ct; cf = take ? of : cf; cr = take ? orr : cr;
      }
      {
        const float ot = __uint_as_float(quad_dpp<kQuadXor2>(__float_as_uint(ct)));
        const uint32_t of = quad_dpp<kQuadXor2>(cf), orr = quad_dpp<kQuadXor2>(cr);
        const bool take = (ot < ct) || ((ot == ct) && (of < cf));
        ct = take ? ot : ct; cf = take ? of : cf; cr = take ? orr : cr;
      }
      const bool closer = (cf != kInvalidFace) && ((ct < best_t) || ((ct == best_t) && (cf < best_face)));
      best_t = closer ? ct : best_t;
      best_face = closer ? cf : best_face;
      best_rec = closer ? cr : best_rec;
      cur = top;
      spb -= 256u;
    }
  }
  h.t = best_t;
  h.rec = (best_face != kInvalidFace) ? best_rec : kNone;
  if (visits) *visits = nvis;
}

// trace_lane_ww whose LAST rays are finished by quads.  A single scan ends when its slowest ray ends, and that ray sits
// in a wave whose other lanes have long been idle: once at most kTailRays rays of the wave are still walking, each of
// them is handed to four lanes (state through LDS, its stack copied into a quad-layout column) and finishes with
// trace_quad -- shorter node steps, four triangles per leaf step -- instead of crawling on alone.  Same visits per
// ray up to ordering, same results.  LDS: lane stacks | quad-tail stacks (64 columns x kQuadStackEntries rows) | hand-over
// slots (4 waves x kTailRays x 12 dwords).
constexpr uint32_t kTailRays = 16;
constexpr uint32_t kTailXferDwords = 12;

template <int kLdsEntries, int kTop = 0, bool kLeafBatch = false, bool kVote = false>
__device__ __forceinline__ void trace_lane_ww_tail(const uint32_t* __restrict__ nodes, const uint32_t* __restrict__ cnodes,
                                                   const uint32_t* __restrict__ tris, f3 O, f3 D, float ray_tfar,
                                                   uint32_t* __restrict__ lds_stack, uint32_t lds_stride,
                                                   uint32_t* __restrict__ qstack, uint32_t* __restrict__ xfer_wave,
                                                   RayHit& h, const uint32_t* lds_top = nullptr) {
  const RaySlab rs = make_ray_slab(O, D);
  float best_t = ray_tfar;
  uint32_t best_rec = kNone;
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  uint32_t priv[(kLdsEntries < 64) ? (64 - kLdsEntries) : 1];
  uint32_t sp = 0;
  uint32_t cur = (ray_tfar >= 0.0f) ? 0u : kDone;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#define RMCL_PUSH(v) { if (kLdsEntries >= 64 || sp < kLdsEntries) lds_stack[sp * lds_stride] = (v); else priv[sp - kLdsEntries] = (v); ++sp; }
#define RMCL_POP() { if (sp == 0) cur = kDone; else { --sp; if (kLdsEntries >= 64 || sp < kLdsEntries) cur = lds_stack[sp * lds_stride]; else cur = priv[sp - kLdsEntries]; } }
  for (;;) {
    const uint64_t m_act = __ballot(cur != kDone);
    if (m_act == 0) break;
    const uint32_t na = static_cast<uint32_t>(__popcll(m_act));
    if (na <= kTailRays) {
      // ---- hand the remaining rays to quads ----
      const bool mine = cur != kDone;
      const uint32_t j = static_cast<uint32_t>(__popcll(m_act & ((1ull << lane) - 1ull)));
      if (mine) {
        uint32_t* x = xfer_wave + j * kTailXferDwords;
        x[0] = __float_as_uint(O.x); x[1] = __float_as_uint(O.y); x[2] = __float_as_uint(O.z);
        x[3] = __float_as_uint(D.x); x[4] = __float_as_uint(D.y); x[5] = __float_as_uint(D.z);
        x[6] = __float_as_uint(ray_tfar); x[7] = __float_as_uint(best_t);
        x[8] = record_face(tris, best_rec);  // the quad traversal carries (t, face id) pairs
        x[9] = best_rec;
        x[10] = cur; x[11] = sp;
        // the stack, bottom to top, into rows 1..sp of column (wave*16 + j) of the quad-layout region
        uint32_t* col = qstack + (wave * kTailRays + j);
        for (uint32_t e = 0; e < sp; ++e) {
          const uint32_t v = (kLdsEntries >= 64 || e < kLdsEntries) ? lds_stack[e * lds_stride] : priv[e - kLdsEntries];
          col[(e + 1u) * 64u] = v;
        }
      }
      __builtin_amdgcn_wave_barrier();  // the hand-over slots and stack columns are read by OTHER lanes of this wave
      const uint32_t q = lane >> 2, c = lane & 3u;
      const bool have = q < na;
      const uint32_t* x = xfer_wave + (have ? q : 0u) * kTailXferDwords;
      const f3 Oq = mk3(asf(x[0]), asf(x[1]), asf(x[2])), Dq = mk3(asf(x[3]), asf(x[4]), asf(x[5]));
      QuadResume rsm;
      rsm.cur = x[10]; rsm.n_stack = x[11]; rsm.best_t = asf(x[7]); rsm.best_face = x[8]; rsm.best_rec = x[9];
      const float tfq = have ? asf(x[6]) : -1.0f;
      RayHit hq;
      trace_quad<true>(cnodes, tris, Oq, Dq, tfq, c, wave * kTailRays + q, qstack, hq, &rsm);
      if (have && c == 0u) {
        uint32_t* y = xfer_wave + q * kTailXferDwords;
        y[7] = __float_as_uint(hq.t); y[9] = hq.rec;
      }
      __builtin_amdgcn_wave_barrier();
      if (mine) {
        const uint32_t* y = xfer_wave + j * kTailXferDwords;
        best_t = asf(y[7]); best_rec = y[9];
      }
      break;
    }
    // phase 1: inner nodes (kVote: left early by the leaf trigger of trace_lane_bf_tail; `na` = rays alive in this round)
    while ((cur != kDone) && !(cur & kLeafBit)) {
      uint32_t key[4], ref[4];
      node_keys_at(node_address<kTop>(nodes, lds_top, cur), rs, best_t, key, ref);
      RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
      if (key[3] != kNone) RMCL_PUSH(ref[3])
      if (key[2] != kNone) RMCL_PUSH(ref[2])
      if (key[1] != kNone) RMCL_PUSH(ref[1])
      if (key[0] != kNone) cur = ref[0];
      else RMCL_POP()
      if (kVote) {
        if (5u * static_cast<uint32_t>(__popcll(__ballot((cur != kDone) && !(cur & kLeafBit)))) <= 2u * na) break;
      }
    }
    // phase 2: this lane's leaf (if any)
    if ((cur != kDone) && (cur & kLeafBit)) {
      if (kLeafBatch) leaf_batch(tris, cur, O, D, ray_tfar, best_t, best_rec);
      else leaf_loop(tris, cur, O, D, ray_tfar, best_t, best_rec);
      RMCL_POP()
    }
  }
#undef RMCL_PUSH
#undef RMCL_POP
  h.t = best_t;
  h.rec = best_rec;
}

// trace_lane_bf whose LAST rays are finished by quads (see trace_lane_ww_tail): branch-free node steps and one-round-trip
// leaves while more than kTailRays rays of the wave are walking, then each remaining ray gets four lanes.
// LDS: lane stacks (kRows x 256) | quad-tail stacks (64 columns x kQuadStackEntries rows) | hand-over slots.
// kPre (experiment iv): after every leaf visit the lane requests the last 16 B (unit normal + face id) of the record that is
// its best hit so far into *pre (and remembers which record in *pre_rec): the epilogue of k_find then finds them in registers
// kPipe (round 3, kind 28): the node step is software-pipelined.  A lone wave's step is a dependent chain -- seven loads, ~130
// VALU instructions, the next node's address --, so the lane requests its NEXT node as soon as the nearest child is known (after
// three of the five compare-exchanges), finishes the ordering and the pushes under that request, keeps the top of its stack in
// a register (a pop costs no LDS round trip) and, at a leaf, requests the node it will pop behind the leaf's records, before
// the triangle tests.  Same visits in the same order: results and visit counts are those of the plain step.
// kSortSteps (kinds 29 / 30): 5 = the four children fully ordered; 4 = nearest first, farthest last, the middle two as they come;
// 3 = only the nearest found.  The visit ORDER changes (never the set of hits: min t, then min face id), a step is 5 / 10 VALU
// instructions shorter.
template <int kRows, bool kLeafBatch, int kLeafTrigger = 0, bool kQuant = false, bool kPre = false, bool kPipe = false, int kSortSteps = 5, bool kTail = true>   // kQuant: `nodes` are the 64-B quantised twins; kTail = false: no hand-over of the last rays to quads (qstack / xfer_wave unused)
__device__ __forceinline__ void trace_lane_bf_tail(const uint32_t* __restrict__ nodes, const uint32_t* __restrict__ cnodes,
                                                   const uint32_t* __restrict__ tris, f3 O, f3 D, float ray_tfar,
                                                   uint32_t* __restrict__ lds_col, uint32_t* __restrict__ qstack,
                                                   uint32_t* __restrict__ xfer_wave, RayHit& h, uint32_t* visits = nullptr,
                                                   const TraceStart* start = nullptr, uint32_t* dbg = nullptr, uint4* pre = nullptr,
                                                   uint32_t* pre_rec = nullptr, const RayHit* seed = nullptr) {
  uint4 pre_v = uint4{0u, 0u, 0u, 0u};
  uint32_t pre_r = kNone;
  const RaySlab rs = make_ray_slab(O, D);
  float best_t = seed ? seed->t : ray_tfar;          // (kind 32: the closest hit among the leaves the wave tested together)
  uint32_t best_rec = seed ? seed->rec : kNone;
  uint32_t nvis = 0;  // node visits of this ray
  uint32_t dbg_slow = 0, dbg_na = 0, dbg_tail = 0, dbg_leaf = 0;
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  uint32_t priv[(kRows < 65) ? (65 - kRows) : 1];
  lds_col[0] = kDone;
  uint32_t sp = start ? start->sp : 1u;   // a preset start: rows 1 .. sp-1 already hold this ray's pending entries
  uint32_t cur = start ? start->cur : ((ray_tfar >= 0.0f) ? 0u : kDone);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#define RMCL_ROW_ST(r, v) { if ((r) < static_cast<uint32_t>(kRows)) lds_col[(r) * kBfStride] = (v); else priv[(r) - kRows] = (v); }
#define RMCL_ROW_LD(r) (((r) < static_cast<uint32_t>(kRows)) ? lds_col[(r) * kBfStride] : priv[(r) - kRows])
  static_assert(!(kPipe && kQuant), "the pipelined step reads the 128-B nodes");
  // the pipelined loops touch LDS rows only (a row select between LDS and scratch would turn the re-read of the cached top into
  // a flat load -- one more instruction through the texture path per step, measured +12 %): a wave whose stacks come within
  // three rows of kRows leaves the pipelined form for good (`piped`, wave-uniform) and continues with the plain step
#define RMCL_TOP_LD(dst, spv) { (dst) = lds_col[(max((spv), 1u) - 1u) * kBfStride]; }
  // kPipe invariants at the top of every round: `nr` holds (a request for) node `cur` whenever cur < kDone, `top` = row sp - 1
  NodeRegs nr = {};
  uint32_t top = kDone;
  bool piped = kPipe;
  if constexpr (kPipe) {
    piped = !__any(sp + 3u > static_cast<uint32_t>(kRows));
    if (piped) {
      if (cur < kDone) nr = node_req(nodes, cur << 7, rs);
      RMCL_TOP_LD(top, sp)
    }
  }
  for (;;) {
    if constexpr (kPipe) piped = __all(piped);   // lanes that had left the node loop when the wave gave up the pipelined form follow
    const uint64_t m_act = __ballot(cur != kDone);
    if (m_act == 0) break;
    const uint32_t na = static_cast<uint32_t>(__popcll(m_act));
    // kTail = false (kinds 31 and 32, whose rays mostly start with nothing but the leaves their tile sees): no hand-over -- rays that only hold
    // leaves finish in a round or two of the lane loop, and the kernel is better off without the quad traversal's code and LDS
    // (C2 sphere-100k 13.4 -> 12.8 us; where stragglers still descend, as on the room, that costs 9 %: kind 23 is the kind for those maps)
    const bool hand_over = kTail && na <= kTailRays;
    if (hand_over) {
      // ---- hand the remaining rays to quads ----
      const bool mine = cur != kDone;
      const uint32_t j = static_cast<uint32_t>(__popcll(m_act & ((1ull << lane) - 1ull)));
      if (mine) {
        uint32_t* x = xfer_wave + j * kTailXferDwords;
        x[0] = __float_as_uint(O.x); x[1] = __float_as_uint(O.y); x[2] = __float_as_uint(O.z);
        x[3] = __float_as_uint(D.x); x[4] = __float_as_uint(D.y); x[5] = __float_as_uint(D.z);
        x[6] = __float_as_uint(ray_tfar); x[7] = __float_as_uint(best_t);
        x[8] = record_face(tris, best_rec);  // the quad traversal carries (t, face id) pairs
        x[9] = best_rec;
        x[10] = cur; x[11] = sp - 1u;        // rows 1..sp-1 hold this ray's pending entries
        uint32_t* col = qstack + (wave * kTailRays + j);
        for (uint32_t e = 1; e < sp; ++e) col[e * 64u] = RMCL_ROW_LD(e);
      }
      __builtin_amdgcn_wave_barrier();  // the hand-over slots and stack columns are read by OTHER lanes of this wave
      const uint32_t q = lane >> 2, c = lane & 3u;
      const bool have = q < na;
      const uint32_t* x = xfer_wave + (have ? q : 0u) * kTailXferDwords;
      const f3 Oq = mk3(asf(x[0]), asf(x[1]), asf(x[2])), Dq = mk3(asf(x[3]), asf(x[4]), asf(x[5]));
      QuadResume rsm;
      rsm.cur = x[10]; rsm.n_stack = x[11]; rsm.best_t = asf(x[7]); rsm.best_face = x[8]; rsm.best_rec = x[9];
      const float tfq = have ? asf(x[6]) : -1.0f;
      RayHit hq;
      uint32_t qvis = 0;
      uint64_t tq0 = 0, tq1 = 0;
      if (dbg) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq0) : : "memory");
      trace_quad<true>(cnodes, tris, Oq, Dq, tfq, c, wave * kTailRays + q, qstack, hq, &rsm, &qvis);
      if (dbg) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq1) : : "memory"); dbg_tail = static_cast<uint32_t>(tq1 - tq0); dbg_na = na; }
      if (have && c == 0u) {
        uint32_t* y = xfer_wave + q * kTailXferDwords;
        y[7] = __float_as_uint(hq.t); y[9] = hq.rec; y[10] = qvis;
      }
      __builtin_amdgcn_wave_barrier();
      if (mine) {
        const uint32_t* y = xfer_wave + j * kTailXferDwords;
        best_t = asf(y[7]); best_rec = y[9]; nvis += y[10];
      }
      break;
    }
    // phase 1: inner nodes.  kLeafTrigger > 0: the phase is also left as soon as that many lanes hold a leaf -- they would
    // otherwise idle through the descents of the others (the wave model: 65 -> 45 node iterations for the slowest tile of
    // the room, 32 -> 26 on the sphere, for one or two more leaf rounds); the stragglers resume in the next round.
#define RMCL_SORT4                                                                                           \
        RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2)                                                   \
        if constexpr (kSortSteps >= 4) RMCL_CSWAP(1, 3)                                                      \
        if constexpr (kSortSteps >= 5) RMCL_CSWAP(1, 2)
#define RMCL_BF_STEP                                                                                         \
    {                                                                                                        \
      uint32_t key[4], ref[4];                                                                               \
      ++nvis;                                                                                                \
      if (!__any(sp + 3u > static_cast<uint32_t>(kRows))) {                                                  \
        const uint32_t top = lds_col[(sp - 1u) * kBfStride];                                                 \
        if constexpr (kQuant) node_keys_q(nodes, cur, rs, best_t, key, ref); else node_keys_off(nodes, cur << 7, rs, best_t, key, ref);                                                \
        RMCL_SORT4                                                                                           \
        lds_col[sp * kBfStride] = ref[3]; sp += (key[3] != kNone) ? 1u : 0u;                                 \
        lds_col[sp * kBfStride] = ref[2]; sp += (key[2] != kNone) ? 1u : 0u;                                 \
        lds_col[sp * kBfStride] = ref[1]; sp += (key[1] != kNone) ? 1u : 0u;                                 \
        const bool any = key[0] != kNone;                                                                    \
        cur = any ? ref[0] : top;                                                                            \
        sp = any ? sp : (sp - 1u);                                                                           \
      } else {                                                                                               \
        ++dbg_slow;                                                                                          \
        if constexpr (kQuant) node_keys_q(nodes, cur, rs, best_t, key, ref); else node_keys_off(nodes, cur << 7, rs, best_t, key, ref);                                                \
        RMCL_SORT4                                                                                           \
        if (key[3] != kNone) { RMCL_ROW_ST(sp, ref[3]) ++sp; }                                               \
        if (key[2] != kNone) { RMCL_ROW_ST(sp, ref[2]) ++sp; }                                               \
        if (key[1] != kNone) { RMCL_ROW_ST(sp, ref[1]) ++sp; }                                               \
        if (key[0] != kNone) cur = ref[0];                                                                   \
        else { --sp; cur = RMCL_ROW_LD(sp); }                                                                \
      }                                                                                                      \
    }
#define RMCL_BF_STEP_PIPE                                                                                    \
    {                                                                                                        \
      uint32_t key[4], ref[4];                                                                               \
      ++nvis;                                                                                                \
      node_keys_from(nr.qnx, nr.qfx, nr.qny, nr.qfy, nr.qnz, nr.qfz, nr.qch, rs, best_t, key, ref);          \
      RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2)                                                     \
      const bool any = key[0] != kNone;                                                                      \
      const uint32_t nxt = any ? ref[0] : top;                                                               \
      if (nxt < kDone) nr = node_req(nodes, nxt << 7, rs);                                                   \
      RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)                                                                      \
      lds_col[sp * kBfStride] = ref[3]; sp += (key[3] != kNone) ? 1u : 0u;                                   \
      lds_col[sp * kBfStride] = ref[2]; sp += (key[2] != kNone) ? 1u : 0u;                                   \
      lds_col[sp * kBfStride] = ref[1]; sp += (key[1] != kNone) ? 1u : 0u;                                   \
      sp = any ? sp : (sp - 1u);                                                                             \
      RMCL_TOP_LD(top, sp)                                                                                   \
      cur = nxt;                                                                                             \
    }
    if (kPipe && piped) {
      while (cur < kDone) {
        if (__any(sp + 3u > static_cast<uint32_t>(kRows))) { piped = false; break; }   // wave-uniform
        RMCL_BF_STEP_PIPE
        if constexpr (kLeafTrigger > 0)
          if (static_cast<uint32_t>(kLeafTrigger) * static_cast<uint32_t>(__popcll(__ballot(cur < kDone))) <= 4u * na) break;
      }
    } else if constexpr (kLeafTrigger > 0) {
      // The loop stays the divergent per-lane while loop; the vote only needs the number of lanes still in it (the ballot of
      // a divergent loop counts exactly those) against `na`, the rays alive when the round began: waiting >= 1.5 x descending
      // <=> 5 x descending <= 2 x alive.  Checked after the step, so every round makes progress (a lane that keeps popping
      // leaves cannot starve the descending ones).
      while (cur < kDone) {
        RMCL_BF_STEP
        if (static_cast<uint32_t>(kLeafTrigger) * static_cast<uint32_t>(__popcll(__ballot(cur < kDone))) <= 4u * na) break;
      }
    } else {
      while (cur < kDone) RMCL_BF_STEP
    }
#undef RMCL_BF_STEP
#undef RMCL_SORT4
#undef RMCL_BF_STEP_PIPE
    // phase 2: this lane's leaf (if any; with a leaf trigger other lanes may still hold an inner node)
    if (kPipe && piped) {
      if (cur > kDone) {
        static_assert(!kPipe || kLeafBatch, "the pipelined step comes with the one-round-trip leaf");
        const uint32_t leaf = cur;
        --sp;
        cur = top;   // == row sp
        leaf_batch_then(tris, leaf, O, D, ray_tfar, best_t, best_rec, [&] {
          if (cur < kDone) nr = node_req(nodes, cur << 7, rs);
          RMCL_TOP_LD(top, sp)
        });
      }
    } else
    if (cur > kDone) {
      ++dbg_leaf;
      if (kLeafBatch) leaf_batch(tris, cur, O, D, ray_tfar, best_t, best_rec);
      else leaf_loop(tris, cur, O, D, ray_tfar, best_t, best_rec);
      if constexpr (kPre) {
        pre_r = best_rec;
        pre_v = reinterpret_cast<const uint4*>(tris)[static_cast<size_t>((best_rec != kNone) ? best_rec : 0u) * 4u + 3u];
      }
      --sp;
      cur = RMCL_ROW_LD(sp);
    }
  }
#undef RMCL_ROW_ST
#undef RMCL_ROW_LD
#undef RMCL_TOP_LD
  h.t = best_t;
  h.rec = best_rec;
  if constexpr (kPre) { *pre = pre_v; *pre_rec = pre_r; }
  if (visits) *visits = nvis;
  if (dbg) { dbg[0] = dbg_slow; dbg[1] = dbg_na; dbg[2] = dbg_tail; dbg[3] = dbg_leaf; }   // dbg[3]: leaf visits of this lane before the quad tail
}

// ---------------------------------------------------------------------------------------------
// FRONTIER START (round 3).  The 64 rays of a wave leave ONE origin through one tile of the scan image, and every one of them
// begins its traversal with the same few levels of the tree: ~45 % of all node visits of a C2 scan are spent above BFS depth 4
// (tools/wavesim.py: 10.5 -> 4.6 visits per ray, 13.5 -> 7.3 node iterations per wave without them).  Here the wave replaces
// those levels by ONE cooperative pass over the map's frontier table (<= 256 entries {box, reference} = every reference at BFS
// depth 4 and every leaf above it, bvh_build.cpp):
//   1. the tile's bounding pyramid: four planes through the origin spanned by the rays of the tile's corner lanes, each pushed
//      outward until EVERY active ray of the wave lies inside (wave-min of n.d over the lanes -- exact for any ray set, spherical
//      rows are small circles, O1Dn directions are data), times the farthest distance a hit can have;
//   2. each lane tests <= 4 frontier boxes against the pyramid (positive-vertex test: conservative);
//   3. the surviving entries are broadcast one by one (v_readlane) and every lane tests the box with ITS ray (the traversal's own
//      slab test): hits go onto the lane's stack, the nearest becomes its first node.
// A child's stored box lies inside its parent's, so "my ray hits this entry's box" is exactly the condition under which the
// descent from the root would have reached that reference: the set of subtrees visited is the same, only their order can
// differ -- and the closest hit (min t, then min face id) does not depend on the order.  Results are bit-identical
// (tests/test_gpu_find.py digests).  If a lane would collect more entries than its LDS rows hold, the wave starts at the root.
// kRow0: index of the first stack row (1 for the branch-free traversals whose row 0 is the sentinel, 0 for trace_lane_ww).
// ---------------------------------------------------------------------------------------------
// minimum over the 64 lanes, the same value in every lane: four DPP steps give every lane the minimum of its row of 16
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror -- one cycle each, no LDS crossbar), four v_readlane + three v_min
// combine the rows
template <int kCtrl>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), kCtrl, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_min_f32(float v) {
  v = fminf(v, dpp_f32<0xB1>(v));
  v = fminf(v, dpp_f32<0x4E>(v));
  v = fminf(v, dpp_f32<0x141>(v));   // row_half_mirror
  v = fminf(v, dpp_f32<0x140>(v));   // row_mirror
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fminf(fminf(r0, r1), fminf(r2, r3));
}
__device__ __forceinline__ float lane_bcast(float v, uint32_t src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), static_cast<int>(src_lane)));
}
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, uint32_t src_lane) {
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(src_lane)));
}

// The pyramid of one tile (= one wave's rays), in the SENSOR frame: four planes through the origin spanned by the corner rays of
// the tile, normals inward and unit length, and per plane how far outside it the tile's worst ray points per unit length
// (m = min over the rays of n.D, <= 0).  Both are properties of the sensor model and the tiling alone -- n.D is invariant under
// the pose's rotation -- so they are computed once per model (k_tile_planes) and a find only rotates four normals.
// out: n0.xyz m0 | n1.xyz m1 | n2.xyz m2 | n3.xyz m3 (wave-uniform values)
__device__ __forceinline__ void tile_planes_wave(f3 D, bool active, uint32_t tile_w_log2, float (&out)[16]) {
  const uint32_t tw = 1u << tile_w_log2;
  const uint32_t c0 = 0u, c1 = tw - 1u, c2 = 64u - tw, c3 = 63u;
  const f3 d0 = mk3(lane_bcast(D.x, c0), lane_bcast(D.y, c0), lane_bcast(D.z, c0));
  const f3 d1 = mk3(lane_bcast(D.x, c1), lane_bcast(D.y, c1), lane_bcast(D.z, c1));
  const f3 d2 = mk3(lane_bcast(D.x, c2), lane_bcast(D.y, c2), lane_bcast(D.z, c2));
  const f3 d3 = mk3(lane_bcast(D.x, c3), lane_bcast(D.y, c3), lane_bcast(D.z, c3));
  const f3 dc = add3(add3(d0, d1), add3(d2, d3));
  f3 n[4] = {cross_fma(d0, d1), cross_fma(d1, d3), cross_fma(d3, d2), cross_fma(d2, d0)};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float len2 = (n[k].x * n[k].x + n[k].y * n[k].y) + n[k].z * n[k].z;
    const bool usable = len2 > 1e-12f;    // degenerate (1-D tiles, coincident corner rays) or NaN: the plane is dropped (n = 0 keeps every box)
    float sc = usable ? __builtin_amdgcn_rsqf(len2) : 0.0f;
    if (dot_plain(n[k], dc) < 0.0f) sc = -sc;   // inward
    n[k] = usable ? scale3(n[k], sc) : mk3(0.f, 0.f, 0.f);
    const float s_lane = active ? dot_plain(n[k], D) : 0.0f;
    const float m = fminf(wave_min_f32(s_lane), 0.0f);
    out[4 * k] = n[k].x; out[4 * k + 1] = n[k].y; out[4 * k + 2] = n[k].z; out[4 * k + 3] = m;
  }
}

// Frontier start: instead of walking the top levels of the tree, the wave culls the map's frontier table (the <= 256 child
// references of BFS depth kFrontierDepth with their boxes) against the pyramid of its tile, then every lane tests the few
// survivors against its own ray and starts with the accepted ones on its stack, the two nearest on top.
// planes: this tile's 16 floats of the model's plane table (uniform address); Rsm: the pose's rotation sensor -> map.
template <int kRows, int kRow0>
__device__ __forceinline__ TraceStart frontier_start(const uint32_t* __restrict__ frontier, uint32_t n_frontier, f3 scene_center,
                                                     float scene_half_diag, const float* __restrict__ planes, quat Rsm, float tfar,
                                                     f3 O, f3 D, float ray_tfar, uint32_t lane, uint32_t* __restrict__ lds_col,
                                                     uint32_t lds_stride, uint32_t max_preload) {
  // max_preload (FindParams::frontier_max_preload, uniform): entries the start may leave on a lane's stack = the stack's 64 entries
  // minus what the deepest descent of THIS map's tree can still push (the builder's stack_need bounds a descent from the root, hence
  // from any frontier entry): a deep tree starts more waves at the root instead of overflowing the stack
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  const bool active = ray_tfar >= 0.0f;
  TraceStart root;
  root.cur = active ? 0u : kDone;
  root.sp = static_cast<uint32_t>(kRow0);
  if (n_frontier == 0u) return root;
  // my <= 4 entries of the table (coalesced: lane j takes entries j, j + 64, ...)
  const uint4* F = reinterpret_cast<const uint4*>(frontier);
  uint4 ea[4], eb[4];
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) {
    const uint32_t idx = min(lane + 64u * k, n_frontier - 1u);
    ea[k] = F[2u * idx];
    eb[k] = F[2u * idx + 1u];
  }
  // 1. the tile's pyramid in the map frame
  const uint4 P0 = sload4(reinterpret_cast<const uint32_t*>(planes), 0u), P1 = sload4(reinterpret_cast<const uint32_t*>(planes), 16u);
  const uint4 P2 = sload4(reinterpret_cast<const uint32_t*>(planes), 32u), P3 = sload4(reinterpret_cast<const uint32_t*>(planes), 48u);
  f3 n[4] = {qrot(Rsm, mk3(asf(P0.x), asf(P0.y), asf(P0.z))), qrot(Rsm, mk3(asf(P1.x), asf(P1.y), asf(P1.z))),
             qrot(Rsm, mk3(asf(P2.x), asf(P2.y), asf(P2.z))), qrot(Rsm, mk3(asf(P3.x), asf(P3.y), asf(P3.z)))};
  const float mq[4] = {asf(P0.w), asf(P1.w), asf(P2.w), asf(P3.w)};
  // the farthest a hit can be from this origin: inside the map's bounding sphere, and within the sensor's range
  const f3 oc = sub3(O, scene_center);
  const float reach = fminf(tfar, sqrtf((oc.x * oc.x + oc.y * oc.y) + oc.z * oc.z) + scene_half_diag);
  float off[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)   // offset in metres at the farthest possible hit + slack for the rounding of the rotation and of this test
    off[k] = mq[k] * reach - 1e-4f * reach - 1e-6f;
  // 2. my entries against the pyramid: the box's vertex farthest along n ("positive vertex") is at n.c + |n|.h from the origin
  // (c = centre - O, h = half extent); in doubled quantities (2c = lo + hi - 2 O, 2h = hi - lo) that is six FMAs per plane
  f3 an[4];
  float off2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    an[q] = mk3(fabsf(n[q].x), fabsf(n[q].y), fabsf(n[q].z));
    off2[q] = 2.0f * off[q];
  }
  const f3 O2 = mk3(2.0f * O.x, 2.0f * O.y, 2.0f * O.z);
  bool acc[4];
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) {
    const f3 lo = mk3(asf(ea[k].x), asf(ea[k].y), asf(ea[k].z)), hi = mk3(asf(ea[k].w), asf(eb[k].x), asf(eb[k].y));
    const f3 c2 = mk3((lo.x + hi.x) - O2.x, (lo.y + hi.y) - O2.y, (lo.z + hi.z) - O2.z);
    const f3 h2 = mk3(hi.x - lo.x, hi.y - lo.y, hi.z - lo.z);
    bool ok = (lane + 64u * k) < n_frontier;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v = fmaf(an[q].z, h2.z, fmaf(an[q].y, h2.y, fmaf(an[q].x, h2.x, fmaf(n[q].z, c2.z, fmaf(n[q].y, c2.y, n[q].x * c2.x)))));
      ok = ok && !(v < off2[q]);   // NaN -> keep
    }
    acc[k] = ok;
  }
  // 3. every surviving entry against every ray of the wave
  const RaySlab rs = make_ray_slab(O, D);
  // the two nearest accepted entries stay in registers (entered first / on top of the stack); the others go below them in
  // table order.  (Only the nearest: the room's hardest ray walks 28 instead of 19 nodes, tools/wavesim.py.)
  uint32_t first_ref = kDone, first_key = 0xFFFFFFFFu, second_ref = kDone, second_key = 0xFFFFFFFFu, sp = static_cast<uint32_t>(kRow0);
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) {
    if (64u * k >= n_frontier) break;   // wave-uniform
    uint64_t mask = __ballot(acc[k]);
    while (mask != 0ull) {
      const uint32_t j = static_cast<uint32_t>(__builtin_ctzll(mask));
      mask &= mask - 1ull;
      const float lx = lane_bcast(asf(ea[k].x), j), ly = lane_bcast(asf(ea[k].y), j), lz = lane_bcast(asf(ea[k].z), j);
      const float hx = lane_bcast(asf(ea[k].w), j), hy = lane_bcast(asf(eb[k].x), j), hz = lane_bcast(asf(eb[k].y), j);
      const uint32_t ref = lane_bcast(eb[k].z, j);
      const float tx0 = fmaf(lx, rs.inv.x, rs.noi.x), tx1 = fmaf(hx, rs.inv.x, rs.noi.x);
      const float ty0 = fmaf(ly, rs.inv.y, rs.noi.y), ty1 = fmaf(hy, rs.inv.y, rs.noi.y);
      const float tz0 = fmaf(lz, rs.inv.z, rs.noi.z), tz1 = fmaf(hz, rs.inv.z, rs.noi.z);
      const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.0f));
      const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), ray_tfar));
      if (active && tn <= tf) {
        const uint32_t key = __float_as_uint(tn);
        const bool n1 = key < first_key, n2 = key < second_key;
        const uint32_t pushed = n2 ? second_ref : ref;          // what leaves (or never enters) the nearest two
        second_ref = n1 ? first_ref : (n2 ? ref : second_ref);
        second_key = n1 ? first_key : (n2 ? key : second_key);
        first_ref = n1 ? ref : first_ref;
        first_key = n1 ? key : first_key;
        if (pushed != kDone) {
          if (sp < static_cast<uint32_t>(kRows)) lds_col[sp * lds_stride] = pushed;
          ++sp;
        }
      }
    }
  }
  if (second_ref != kDone) {
    if (sp < static_cast<uint32_t>(kRows)) lds_col[sp * lds_stride] = second_ref;
    ++sp;
  }
  // a lane that would need more rows than it has in LDS: the whole wave starts at the root instead
  if (__any(sp > min(static_cast<uint32_t>(kRows - 4), static_cast<uint32_t>(kRow0) + max_preload))) return root;
  TraceStart st;
  st.cur = first_ref;   // kDone: the ray misses every entry, i.e. the whole map
  st.sp = sp;
  return st;
}

// ---------------------------------------------------------------------------------------------
// COOPERATIVE DESCENT below the frontier (round 6, kinds 31 and 32).  The frontier start replaces the top four levels of every ray's descent
// by one cooperative pass; what remains on a resident map is still a chain of dependent, divergent node fetches (~4.6 node visits per
// ray of a C2 scan, 10 - 16 wave steps of ~1.3 k cycles each at two waves per SIMD).  But the 64 rays of a tile keep walking the SAME
// few subtrees below the frontier too: a 16 x 4 tile of a C2 scan sees about a dozen leaves.  So the wave keeps descending TOGETHER:
// the frontier's survivors are a list in LDS; per level, lane l tests child (l & 3) of survivor (l >> 2) -- one coalesced 32-B fetch per
// lane from the child-major nodes, the pyramid test of the frontier pass -- and the children inside the tile's pyramid become the next
// list (16 nodes per pass, up to two passes per level).  Leaves met on the way collect in a second list.  When no inner node is left
// (or the lists would outgrow 64 entries: a tile that sees a lot of the map, or a deep scene -- the level is then not expanded) every
// lane tests the final entries against ITS ray, exactly as the frontier start does with the frontier's survivors, and starts the
// ordinary per-lane traversal with the accepted ones on its stack: on the benchmark maps these are leaves only, so the divergent
// phase is a few leaf rounds.  One round trip per LEVEL and wave instead of one per node visit and ray.
// Same argument as the frontier start for the results: a stored child box lies inside its parent's, an entry is dropped only when its
// box lies outside the pyramid that contains every ray of the wave (up to the farthest possible hit), hence every subtree a ray's own
// descent would enter is entered; the closest hit (min t, then min face id) does not depend on the order.  Bit-identical to kind 23.
// ws: this wave's scratch, kDescentWaveDwords dwords, 16-B aligned.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kDescentCap = 64u, kDescentInnerCap = 32u;       // entries ({box, ref} = 8 dwords each) of the final list / of an inner list
constexpr uint32_t kDescentWaveDwords = (kDescentCap + 2u * kDescentInnerCap) * 8u;   // final list | inner list A | inner list B: 4 KB per wave
#define RMCL_WAVE_LDS_SYNC()                                   \
  {                                                            \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
  }

// kCoop (kind 32): the per-ray part without the sorted hand-over -- see the comment at the final list below.  `seed` receives the closest hit
// among the final LEAVES; the per-lane traversal that follows starts from it and only sees what the descent left unexpanded.
template <int kRows, int kRow0, bool kCoop = false>
__device__ __forceinline__ TraceStart frontier_descent_start(const uint32_t* __restrict__ frontier, uint32_t n_frontier,
                                                             const uint32_t* __restrict__ cnodes, const uint32_t* __restrict__ cnodes16,
                                                             f3 scene_center, float scene_half_diag,
                                                             const float* __restrict__ planes, quat Rsm, float tfar, f3 O, f3 D,
                                                             float ray_tfar, uint32_t lane, uint32_t* __restrict__ lds_col,
                                                             uint32_t lds_stride, uint32_t max_preload, uint32_t* __restrict__ ws,
                                                             uint32_t final_cap, uint32_t max_levels, uint32_t* dbg_levels = nullptr,
                                                             uint32_t* stamps = nullptr, const uint32_t* __restrict__ tris = nullptr,
                                                             RayHit* seed = nullptr) {
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  const bool active = ray_tfar >= 0.0f;
  // diagnostics (clocked lab instantiation only; stamps == nullptr folds all of it away): shader clock at the phase boundaries, after
  // everything in flight has arrived -- 0 entry, 1 frontier + planes here, 2 culled, 3 lists written, 4 + 2 L nodes of level L here,
  // 5 + 2 L level L compacted (L < 3), 10 final list lane-resident, 11 every ray has seen every entry
#define RMCL_STAMP(i) { if (stamps != nullptr) { uint64_t t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); stamps[i] = static_cast<uint32_t>(t_); } }
  RMCL_STAMP(0)
  if constexpr (kCoop) { seed->t = ray_tfar; seed->rec = kNone; }
  const uint32_t mask_leaf_cap = (max_levels >> 8) & 0xFFu;   // kCoop: most final leaves a ray may enter (bits 8..15 of the level word)
  max_levels &= 0xFFu;
  TraceStart root;
  root.cur = active ? 0u : kDone;
  root.sp = static_cast<uint32_t>(kRow0);
  if (n_frontier == 0u) return root;
  const uint4* Ft = reinterpret_cast<const uint4*>(frontier);
  uint4 ea[4], eb[4];
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) {
    const uint32_t idx = min(lane + 64u * k, n_frontier - 1u);
    ea[k] = Ft[2u * idx];
    eb[k] = Ft[2u * idx + 1u];
  }
  // the tile's pyramid in the map frame (frontier_start, steps 1 and 2)
  const uint4 P0 = sload4(reinterpret_cast<const uint32_t*>(planes), 0u), P1 = sload4(reinterpret_cast<const uint32_t*>(planes), 16u);
  const uint4 P2 = sload4(reinterpret_cast<const uint32_t*>(planes), 32u), P3 = sload4(reinterpret_cast<const uint32_t*>(planes), 48u);
  f3 n[4] = {qrot(Rsm, mk3(asf(P0.x), asf(P0.y), asf(P0.z))), qrot(Rsm, mk3(asf(P1.x), asf(P1.y), asf(P1.z))),
             qrot(Rsm, mk3(asf(P2.x), asf(P2.y), asf(P2.z))), qrot(Rsm, mk3(asf(P3.x), asf(P3.y), asf(P3.z)))};
  const float mq[4] = {asf(P0.w), asf(P1.w), asf(P2.w), asf(P3.w)};
  RMCL_STAMP(1)
  const f3 oc = sub3(O, scene_center);
  const float reach = fminf(tfar, sqrtf((oc.x * oc.x + oc.y * oc.y) + oc.z * oc.z) + scene_half_diag);
  f3 an[4];
  float off2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    an[q] = mk3(fabsf(n[q].x), fabsf(n[q].y), fabsf(n[q].z));
    off2[q] = 2.0f * (mq[q] * reach - 1e-4f * reach - 1e-6f);
  }
  const f3 O2 = mk3(2.0f * O.x, 2.0f * O.y, 2.0f * O.z);
  // a box {a.xyz = lo, a.w b.x b.y = hi} against the pyramid: its vertex farthest along every plane normal (conservative; NaN keeps)
  auto in_pyramid = [&](const uint4& a, const uint4& b) -> bool {
    const f3 lo = mk3(asf(a.x), asf(a.y), asf(a.z)), hi = mk3(asf(a.w), asf(b.x), asf(b.y));
    const f3 c2 = mk3((lo.x + hi.x) - O2.x, (lo.y + hi.y) - O2.y, (lo.z + hi.z) - O2.z);
    const f3 h2 = mk3(hi.x - lo.x, hi.y - lo.y, hi.z - lo.z);
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v = fmaf(an[q].z, h2.z, fmaf(an[q].y, h2.y, fmaf(an[q].x, h2.x, fmaf(n[q].z, c2.z, fmaf(n[q].y, c2.y, n[q].x * c2.x)))));
      ok = ok && !(v < off2[q]);
    }
    return ok;
  };
  bool acc[4];
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) acc[k] = ((lane + 64u * k) < n_frontier) && in_pyramid(ea[k], eb[k]);

  const RaySlab rs = make_ray_slab(O, D);
  uint32_t first_ref = kDone, first_key = 0xFFFFFFFFu, second_ref = kDone, second_key = 0xFFFFFFFFu, sp = static_cast<uint32_t>(kRow0);
  // one entry {box, ref} (wave-uniform values) against this lane's ray: the traversal's own slab test; the two nearest accepted entries
  // stay in registers (entered first / on top of the stack), the others go below them in list order
  auto offer = [&](float lx, float ly, float lz, float hx, float hy, float hz, uint32_t ref) {
    const float tx0 = fmaf(lx, rs.inv.x, rs.noi.x), tx1 = fmaf(hx, rs.inv.x, rs.noi.x);
    const float ty0 = fmaf(ly, rs.inv.y, rs.noi.y), ty1 = fmaf(hy, rs.inv.y, rs.noi.y);
    const float tz0 = fmaf(lz, rs.inv.z, rs.noi.z), tz1 = fmaf(hz, rs.inv.z, rs.noi.z);
    const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.0f));
    const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), ray_tfar));
    if (active && tn <= tf) {
      const uint32_t key = __float_as_uint(tn);
      const bool n1 = key < first_key, n2 = key < second_key;
      const uint32_t pushed = n2 ? second_ref : ref;
      second_ref = n1 ? first_ref : (n2 ? ref : second_ref);
      second_key = n1 ? first_key : (n2 ? key : second_key);
      first_ref = n1 ? ref : first_ref;
      first_key = n1 ? key : first_key;
      if (pushed != kDone) {
        if (sp < static_cast<uint32_t>(kRows)) lds_col[sp * lds_stride] = pushed;
        ++sp;
      }
    }
  };

  // how many entries survive, how many of them are inner nodes (wave-uniform)
  uint64_t m_all[4], m_in[4];
  uint32_t n_surv = 0, n_inner = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) {
    m_all[k] = __ballot(acc[k]);
    m_in[k] = __ballot(acc[k] && !(eb[k].z & kLeafBit));
    n_surv += static_cast<uint32_t>(__popcll(m_all[k]));
    n_inner += static_cast<uint32_t>(__popcll(m_in[k]));
  }
  uint32_t levels = 0;
  RMCL_STAMP(2)
  if (dbg_levels) *dbg_levels = min(n_surv, 63u);   // diagnostics: survivors of the frontier | entries after level 1 << 6 | 2 << 12 | 3 << 18 | final entries << 24
  if (cnodes != nullptr && n_inner != 0u && n_inner <= kDescentInnerCap && n_surv <= kDescentCap) {
    // ---- the descent: lists in this wave's scratch ----
    uint4* Fl = reinterpret_cast<uint4*>(ws);
    uint4* Acur = Fl + 2u * kDescentCap;
    uint4* Anext = Acur + 2u * kDescentInnerCap;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t nF = 0, nA = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      const uint64_t m_lf = m_all[k] & ~m_in[k];
      if (acc[k]) {
        const bool inner = !(eb[k].z & kLeafBit);
        const uint32_t idx = inner ? nA + static_cast<uint32_t>(__popcll(m_in[k] & lt)) : nF + static_cast<uint32_t>(__popcll(m_lf & lt));
        uint4* dst = inner ? Acur : Fl;
        dst[2u * idx] = ea[k];
        dst[2u * idx + 1u] = eb[k];
      }
      nF += static_cast<uint32_t>(__popcll(m_lf));
      nA += static_cast<uint32_t>(__popcll(m_in[k]));
    }
    RMCL_WAVE_LDS_SYNC()
    RMCL_STAMP(3)
    // two levels per round trip where the map carries the 16-wide twins (layout.h Node16C: a node's GRANDCHILDREN, child-major): lane l
    // tests entry (l & 15) of survivor (l >> 4), four nodes per pass -- the first levels below the frontier barely branch (2.8 -> 3.0 ->
    // 3.6 survivors per tile of a C2 scan on sphere-100k), so a level of four-wide nodes buys little for its round trip
    const bool wide = cnodes16 != nullptr;
    const uint32_t lsh = wide ? 4u : 2u, per_pass = wide ? 4u : 16u;
    const uint32_t s_in = lane >> lsh, c_in = lane & ((1u << lsh) - 1u);
    // A level that turns out not to fit costs a round trip for nothing, so the wave predicts before it fetches: the list grew by
    // g = total / previous total at the last level (at least 5/4); the next level is fetched only if total * g <= final_cap
    uint32_t prev_total = max(n_surv, 1u);
    while (nA != 0u && levels < max_levels) {
      {
        const uint32_t total = nF + nA;
        const uint32_t g_num = max(4u * total, 5u * prev_total);   // g = g_num / (4 prev_total)
        if (total * g_num > final_cap * 4u * prev_total) break;
        prev_total = total;
      }
      // expand every node of the current list: 16 nodes (x 4 children) or 4 nodes (x 16 grandchildren) per pass
      uint32_t nFn = nF, nAn = 0;
      for (uint32_t base = 0; base < nA; base += per_pass) {
        const uint32_t s = base + s_in;
        const bool have = s < nA;
        const uint32_t nref = have ? reinterpret_cast<const uint32_t*>(Acur)[8u * s + 6u] : 0u;
        const uint4* cn = wide ? reinterpret_cast<const uint4*>(cnodes16) + (static_cast<size_t>(nref) * 32u + 2u * c_in)
                               : reinterpret_cast<const uint4*>(cnodes) + (static_cast<size_t>(nref) * 8u + 2u * c_in);
        const uint4 ca = cn[0], cb = cn[1];
        if (base == 0u) { if (levels == 0u) RMCL_STAMP(4) else if (levels == 1u) RMCL_STAMP(6) else if (levels == 2u) RMCL_STAMP(8) }
        // (an unused child slot holds the far point, layout.h: never inside a scene -- but possibly inside an unbounded pyramid)
        const bool ok = have && (asf(ca.x) < 1.0e29f) && in_pyramid(ca, cb);
        const bool inner = !(cb.z & kLeafBit);
        const uint64_t mi = __ballot(ok && inner), ml = __ballot(ok && !inner);
        if (ok) {
          const uint32_t idx = inner ? nAn + static_cast<uint32_t>(__popcll(mi & lt)) : nFn + static_cast<uint32_t>(__popcll(ml & lt));
          if (idx < (inner ? kDescentInnerCap : kDescentCap)) {
            uint4* dst = inner ? Anext : Fl;
            dst[2u * idx] = ca;
            dst[2u * idx + 1u] = cb;
          }
        }
        nAn += static_cast<uint32_t>(__popcll(mi));
        nFn += static_cast<uint32_t>(__popcll(ml));
      }
      RMCL_WAVE_LDS_SYNC()
      if (levels == 0u) RMCL_STAMP(5) else if (levels == 1u) RMCL_STAMP(7) else if (levels == 2u) RMCL_STAMP(9)
      if (nFn + nAn > final_cap || nAn > kDescentInnerCap) break;   // this level does not fit: it stays unexpanded (its leaves, written above nF, are forgotten)
      nF = nFn;
      nA = nAn;
      uint4* t = Acur; Acur = Anext; Anext = t;
      ++levels;
      if (dbg_levels && levels <= 3u) *dbg_levels |= min(nF + nA, 63u) << (6u * levels);
    }
    if (dbg_levels) *dbg_levels |= (nF + nA) << 24;
    // the inner nodes left unexpanded join the final list (nF + nA <= kDescentCap by construction)
    if (lane < nA) {
      Fl[2u * (nF + lane)] = Acur[2u * lane];
      Fl[2u * (nF + lane) + 1u] = Acur[2u * lane + 1u];
    }
    const uint32_t n_leaves = nF;   // final entries [0, n_leaves) are leaves, [n_leaves, nF) inner nodes
    nF += nA;
    RMCL_WAVE_LDS_SYNC()
    // final entries lane-resident, then each against every ray of the wave
    const uint32_t jl = min(lane, max(nF, 1u) - 1u);
    const uint4 fa = Fl[2u * jl], fb = Fl[2u * jl + 1u];
    RMCL_STAMP(10)
    if constexpr (!kCoop) {
      for (uint32_t j = 0; j < nF; ++j)
        offer(lane_bcast(asf(fa.x), j), lane_bcast(asf(fa.y), j), lane_bcast(asf(fa.z), j), lane_bcast(asf(fa.w), j), lane_bcast(asf(fb.x), j),
              lane_bcast(asf(fb.y), j), lane_bcast(fb.z, j));
    } else {
      // Every closest hit is (min t, then min face id) whatever the order of the tests, and a leaf that reaches a ray's stack is tested
      // whatever its distance -- so the per-ray filter needs neither the entry distances in order nor a stack: one bit per final entry
      // (the list holds at most 64), the boxes read back from the list as broadcasts, and the ray then tests the leaves of its set bits
      // in list order.  Only inner nodes the descent left unexpanded (rare) go to the ray's stack for the ordinary traversal.
      uint32_t m_lo = 0u, m_hi = 0u;
      float e_tn = 0.0f;
      auto box_hit = [&](uint32_t j) -> bool {
        const uint4 a = Fl[2u * j], b = Fl[2u * j + 1u];   // (wave-uniform address: a broadcast read)
        const float tx0 = fmaf(asf(a.x), rs.inv.x, rs.noi.x), tx1 = fmaf(asf(a.w), rs.inv.x, rs.noi.x);
        const float ty0 = fmaf(asf(a.y), rs.inv.y, rs.noi.y), ty1 = fmaf(asf(b.x), rs.inv.y, rs.noi.y);
        const float tz0 = fmaf(asf(a.z), rs.inv.z, rs.noi.z), tz1 = fmaf(asf(b.y), rs.inv.z, rs.noi.z);
        e_tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.0f));
        const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), ray_tfar));
        return e_tn <= tf;   // (an inactive ray has ray_tfar < 0: never)
      };
      const uint32_t n_lo = min(n_leaves, 32u);
      for (uint32_t j = 0; j < n_lo; ++j) m_lo |= box_hit(j) ? (1u << j) : 0u;   // (unrolling by four measured 5 % slower: registers)
      for (uint32_t j = 32u; j < n_leaves; ++j) m_hi |= box_hit(j) ? (1u << (j - 32u)) : 0u;
      // the unexpanded inner nodes (behind the leaves in the list; none on most tiles) are the ordinary traversal's business: nearest
      // first as in the sorted form -- a far subtree entered before the near one is walked without a bound
      for (uint32_t j = n_leaves; j < nF; ++j) {
        if (box_hit(j)) {
          const uint32_t key = __float_as_uint(e_tn), ref = lane_bcast(fb.z, j);
          const bool n1 = key < first_key, n2 = key < second_key;
          const uint32_t pushed = n2 ? second_ref : ref;
          second_ref = n1 ? first_ref : (n2 ? ref : second_ref);
          second_key = n1 ? first_key : (n2 ? key : second_key);
          first_ref = n1 ? ref : first_ref;
          first_key = n1 ? key : first_key;
          if (pushed != kDone) {
            if (sp < static_cast<uint32_t>(kRows)) lds_col[sp * lds_stride] = pushed;
            ++sp;
          }
        }
      }
      // A ray that enters many of the boxes (a grazing ray along a wall) would test every one of their leaves here, where its own
      // ordered traversal stops at the first few: such a wave starts at the root like a wave whose stacks the sorted form overfills
      if (__any(static_cast<uint32_t>(__popc(m_lo) + __popc(m_hi)) > mask_leaf_cap)) return root;
      float best_t = ray_tfar;
      uint32_t best_rec = kNone;
      const uint32_t* Fw = reinterpret_cast<const uint32_t*>(Fl);
      while (__any((m_lo | m_hi) != 0u)) {
        uint32_t ref = kDone;
        if ((m_lo | m_hi) != 0u) {
          const bool lo = m_lo != 0u;
          const uint32_t w = lo ? m_lo : m_hi;
          const uint32_t bit = static_cast<uint32_t>(__builtin_ctz(w));
          m_lo = lo ? (m_lo & (m_lo - 1u)) : m_lo;
          m_hi = lo ? m_hi : (m_hi & (m_hi - 1u));
          ref = Fw[8u * (bit + (lo ? 0u : 32u)) + 6u];
        }
        if (ref > kDone) leaf_batch(tris, ref, O, D, ray_tfar, best_t, best_rec);
      }
      seed->t = best_t;
      seed->rec = best_rec;
    }
    RMCL_STAMP(11)
    if (stamps != nullptr) stamps[12] = levels;
  } else {
    // no descent (nothing but leaves survived, too many survivors, no child-major nodes): the frontier start's own third step
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      if (64u * k >= n_frontier) break;   // wave-uniform
      uint64_t mask = m_all[k];
      while (mask != 0ull) {
        const uint32_t j = static_cast<uint32_t>(__builtin_ctzll(mask));
        mask &= mask - 1ull;
        offer(lane_bcast(asf(ea[k].x), j), lane_bcast(asf(ea[k].y), j), lane_bcast(asf(ea[k].z), j), lane_bcast(asf(ea[k].w), j),
              lane_bcast(asf(eb[k].x), j), lane_bcast(asf(eb[k].y), j), lane_bcast(eb[k].z, j));
      }
    }
  }
  if (second_ref != kDone) {
    if (sp < static_cast<uint32_t>(kRows)) lds_col[sp * lds_stride] = second_ref;
    ++sp;
  }
  if (__any(sp > min(static_cast<uint32_t>(kRows - 4), static_cast<uint32_t>(kRow0) + max_preload))) return root;
  TraceStart st;
  st.cur = first_ref;
  st.sp = sp;
  return st;
#undef RMCL_STAMP
}

// ---------------------------------------------------------------------------------------------
// closest-point query (CPCEmbree::find -> rm::EmbreeMap::closestPoint): per-lane while-while traversal ordered
// by box distance, closest point on triangle = Embree closest_point tutorial / Ericson RTCD 5.1.5, in the exact
// operation order of oracle/rmcl_oracle.c:closest_point_triangle (a = v0, ab = -e1, ac = e2, b = a+ab, c = a+ac).
// Equidistant triangles: min squared distance, then min face id.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f3 closest_point_triangle(f3 a, f3 e1, f3 e2, f3 p) {
  const f3 ab = neg3(e1), ac = e2;
  const f3 b = add3(a, ab), c = add3(a, ac);
  const f3 ap = sub3(p, a);
  const float d1 = dot_plain(ab, ap), d2 = dot_plain(ac, ap);
  if (d1 <= 0.f && d2 <= 0.f) return a;
  const f3 bp = sub3(p, b);
  const float d3 = dot_plain(ab, bp), d4 = dot_plain(ac, bp);
  if (d3 >= 0.f && d4 <= d3) return b;
  const f3 cp = sub3(p, c);
  const float d5 = dot_plain(ab, cp), d6 = dot_plain(ac, cp);
  if (d6 >= 0.f && d5 <= d6) return c;
  const float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { const float v = d1 / (d1 - d3); return add3(a, scale3(ab, v)); }
  const float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { const float v = d2 / (d2 - d6); return add3(a, scale3(ac, v)); }
  const float va = d3 * d6 - d5 * d4;
  if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
    const float v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    return add3(b, scale3(sub3(c, b), v));
  }
  const float denom = 1.f / ((va + vb) + vc);
  const float v = vb * denom, w = vc * denom;
  return add3(add3(a, scale3(ab, v)), scale3(ac, w));
}

struct NearHit {
  float d2;
  uint32_t face;
  uint32_t rec;
  f3 p;
};

template <int kLdsEntries>
__device__ __forceinline__ void nearest_lane_ww(const uint32_t* __restrict__ nodes, const uint32_t* __restrict__ tris,
                                                f3 P, bool active, uint32_t* __restrict__ lds_stack,
                                                uint32_t lds_stride, NearHit& h, const NearHit* seed = nullptr) {
  float best = 3.0e38f;  // finite: unused node slots (box at 1e30 -> d2 = inf) never pass `d2 <= best`
  uint32_t best_face = kInvalidFace, best_rec = 0;
  f3 best_p = mk3(0.f, 0.f, 0.f);
  if (seed != nullptr) { best = seed->d2; best_face = seed->face; best_rec = seed->rec; best_p = seed->p; }   // (a bound without a candidate has face == kInvalidFace: every face wins a tie against it)
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  uint32_t priv[(kLdsEntries < 64) ? (64 - kLdsEntries) : 1];
  uint32_t sp = 0;
  uint32_t cur = active ? 0u : kDone;
#define RMCL_PUSH(v) { if (kLdsEntries >= 64 || sp < kLdsEntries) lds_stack[sp * lds_stride] = (v); else priv[sp - kLdsEntries] = (v); ++sp; }
#define RMCL_POP() { if (sp == 0) cur = kDone; else { --sp; if (kLdsEntries >= 64 || sp < kLdsEntries) cur = lds_stack[sp * lds_stride]; else cur = priv[sp - kLdsEntries]; } }
  while (__any(cur != kDone)) {
    while ((cur != kDone) && !(cur & kLeafBit)) {
      const uint4* np = reinterpret_cast<const uint4*>(nodes) + static_cast<size_t>(cur) * 8u;
      const uint4 qx0 = np[0], qx1 = np[1], qy0 = np[2], qy1 = np[3], qz0 = np[4], qz1 = np[5], qch = np[6];
      // (lower, upper) plane of child c per axis: q*0 hold the four lower planes, q*1 the four upper planes
      const f2 bx[4] = {{asf(qx0.x), asf(qx1.x)}, {asf(qx0.y), asf(qx1.y)}, {asf(qx0.z), asf(qx1.z)}, {asf(qx0.w), asf(qx1.w)}};
      const f2 by[4] = {{asf(qy0.x), asf(qy1.x)}, {asf(qy0.y), asf(qy1.y)}, {asf(qy0.z), asf(qy1.z)}, {asf(qy0.w), asf(qy1.w)}};
      const f2 bz[4] = {{asf(qz0.x), asf(qz1.x)}, {asf(qz0.y), asf(qz1.y)}, {asf(qz0.z), asf(qz1.z)}, {asf(qz0.w), asf(qz1.w)}};
      uint32_t ref[4] = {qch.x, qch.y, qch.z, qch.w};
      uint32_t key[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // squared distance to the (padded) child box: a conservative lower bound of any triangle inside
        const float dx = fmaxf(fmaxf(bx[c].x - P.x, P.x - bx[c].y), 0.f);
        const float dy = fmaxf(fmaxf(by[c].x - P.y, P.y - by[c].y), 0.f);
        const float dz = fmaxf(fmaxf(bz[c].x - P.z, P.z - bz[c].y), 0.f);
        const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        key[c] = (d2 <= best) ? __float_as_uint(d2) : kNone;
      }
      RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
      if (key[3] != kNone) RMCL_PUSH(ref[3])
      if (key[2] != kNone) RMCL_PUSH(ref[2])
      if (key[1] != kNone) RMCL_PUSH(ref[1])
      if (key[0] != kNone) cur = ref[0];
      else RMCL_POP()
    }
    if (cur != kDone) {
      const uint32_t first = cur & 0x0FFFFFFFu;
      const uint32_t cnt = ((cur >> 28) & 7u) + 1u;
      for (uint32_t i = 0; i < cnt; ++i) {
        const uint4* tp = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(first + i) * 4u;
        const uint4 a = tp[0], b = tp[1], c = tp[2], d = tp[3];
        const f3 v0 = mk3(asf(a.x), asf(a.y), asf(a.z));
        const f3 e1 = mk3(asf(a.w), asf(b.x), asf(b.y));
        const f3 e2 = mk3(asf(b.z), asf(b.w), asf(c.x));
        const uint32_t face = d.w;
        const f3 q = closest_point_triangle(v0, e1, e2, P);
        const f3 df = sub3(P, q);
        const float d2 = (df.x * df.x + df.y * df.y) + df.z * df.z;
        const bool closer = (d2 < best) || ((d2 == best) && (face < best_face));
        if (closer) { best = d2; best_face = face; best_rec = first + i; best_p = q; }
      }
      RMCL_POP()
    }
  }
#undef RMCL_PUSH
#undef RMCL_POP
  h.d2 = best;
  h.face = best_face;
  h.rec = best_rec;
  h.p = best_p;
}

// closest-point query with FOUR lanes per point (see trace_quad): lane c measures the distance to child box c and
// runs Ericson's closest-point-on-triangle for triangle c of a leaf -- the expensive, branchy part of this query is
// done for up to four triangles at once; (d2, face id) ties resolve exactly like nearest_lane_ww.
__device__ __forceinline__ void nearest_quad(const uint32_t* __restrict__ nodes, const uint32_t* __restrict__ tris, f3 P,
                                             bool active, uint32_t c, uint32_t ray, uint32_t* __restrict__ lds, NearHit& h,
                                             const NearHit* seed = nullptr) {
  float best = 3.0e38f;
  uint32_t best_face = kInvalidFace, best_rec = 0;
  f3 best_p = mk3(0.f, 0.f, 0.f);
  if (seed != nullptr) { best = seed->d2; best_face = seed->face; best_rec = seed->rec; best_p = seed->p; }   // (a bound without a candidate has face == kInvalidFace: every face wins a tie against it)
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  const char* nbase = reinterpret_cast<const char*>(nodes);
  char* sbase = reinterpret_cast<char*>(lds) + ray * 4u;
  if (c == 0u) *reinterpret_cast<uint32_t*>(sbase) = kDone;  // sentinel row (see trace_quad)
  uint32_t spb = 256u;
  uint32_t cur = active ? 0u : kDone;
  const uint32_t coff = c * 32u;  // this lane's child inside a child-major node (layout.h: Node4C)
  while (__any(cur != kDone)) {
    while (cur < kDone) {
      const uint4* nd = reinterpret_cast<const uint4*>(nbase + (cur << 7) + coff);
      const uint4 q0 = nd[0], q1 = nd[1];  // lo.x lo.y lo.z hi.x | hi.y hi.z ref pad
      const float lx = asf(q0.x), ly = asf(q0.y), lz = asf(q0.z), hx = asf(q0.w), hy = asf(q1.x), hz = asf(q1.y);
      const uint32_t ref = q1.z;
      const uint32_t top = *reinterpret_cast<const uint32_t*>(sbase + (spb - 256u));
      const float dx = fmaxf(fmaxf(lx - P.x, P.x - hx), 0.f);
      const float dy = fmaxf(fmaxf(ly - P.y, P.y - hy), 0.f);
      const float dz = fmaxf(fmaxf(lz - P.z, P.z - hz), 0.f);
      const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
      const uint32_t key = ((d2 <= best) ? (__float_as_uint(d2) & ~3u) : 0xFFFFFFFCu) | c;
      const uint32_t k1 = quad_dpp<kQuadXor1>(key), k2 = quad_dpp<kQuadXor2>(key), k3 = quad_dpp<kQuadXor3>(key);
      const uint32_t rank = (k1 < key ? 1u : 0u) + (k2 < key ? 1u : 0u) + (k3 < key ? 1u : 0u);
      const uint32_t kmin = min(min(key, k1), min(k2, k3));
      const uint32_t nh = (key < 0xFFFFFFFCu ? 1u : 0u) + (k1 < 0xFFFFFFFCu ? 1u : 0u) + (k2 < 0xFFFFFFFCu ? 1u : 0u) +
                          (k3 < 0xFFFFFFFCu ? 1u : 0u);
      const uint32_t sel = (key == kmin) ? ref : 0u;
      const uint32_t s1 = sel | quad_dpp<kQuadXor1>(sel);
      const uint32_t nearest = s1 | quad_dpp<kQuadXor2>(s1);
      const uint32_t row = (rank < nh) ? (nh - 1u - rank) : rank;
      *reinterpret_cast<uint32_t*>(sbase + spb + (row << 8)) = ref;
      const bool any = nh != 0u;
      cur = any ? nearest : top;
      spb = any ? (spb + ((nh - 1u) << 8)) : (spb - 256u);
    }
    if (cur != kDone) {
      const uint32_t first = cur & 0x0FFFFFFFu;
      const uint32_t cnt = ((cur >> 28) & 7u) + 1u;
      const uint32_t idx = first + ((c < cnt) ? c : (cnt - 1u));
      const uint4* tp = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(idx) * 4u;
      const uint4 a = tp[0], b = tp[1], cc = tp[2], d = tp[3];
      const uint32_t top = *reinterpret_cast<const uint32_t*>(sbase + (spb - 256u));
      const f3 v0 = mk3(asf(a.x), asf(a.y), asf(a.z));
      const f3 e1 = mk3(asf(a.w), asf(b.x), asf(b.y));
      const f3 e2 = mk3(asf(b.z), asf(b.w), asf(cc.x));
      f3 cq = closest_point_triangle(v0, e1, e2, P);
      const f3 df = sub3(P, cq);
      const float d2 = (df.x * df.x + df.y * df.y) + df.z * df.z;
      const bool valid = c < cnt;
      float cd = valid ? d2 : __builtin_inff();
      uint32_t cf = valid ? d.w : kInvalidFace, cr = idx;
#define RMCL_QMIN(CTRL)                                                                               \
      {                                                                                               \
        const float od = __uint_as_float(quad_dpp<CTRL>(__float_as_uint(cd)));                        \
        const uint32_t of = quad_dpp<CTRL>(cf), orr = quad_dpp<CTRL>(cr);                             \
        const float ox = __uint_as_float(quad_dpp<CTRL>(__float_as_uint(cq.x)));                      \
        const float oy = __uint_as_float(quad_dpp<CTRL>(__float_as_uint(cq.y)));                      \
        const float oz = __uint_as_float(quad_dpp<CTRL>(__float_as_uint(cq.z)));                      \
        const bool take = (od < cd) || ((od == cd) && (of < cf));                                     \
        cd = take ? od : cd; cf = take ? of : cf; cr = take ? orr : cr;                               \
        cq.x = take ? ox : cq.x; cq.y = take ? oy : cq.y; cq.z = take ? oz : cq.z;                    \
      }
      RMCL_QMIN(kQuadXor1)
      RMCL_QMIN(kQuadXor2)
#undef RMCL_QMIN
      const bool closer = (cf != kInvalidFace) && ((cd < best) || ((cd == best) && (cf < best_face)));
      if (closer) { best = cd; best_face = cf; best_rec = cr; best_p = cq; }
      cur = top;
      spb -= 256u;
    }
  }
  h.d2 = best;
  h.face = best_face;
  h.rec = best_rec;
  h.p = best_p;
}

struct CpcParams {
  const uint32_t* nodes;
  const uint32_t* tris;
  const float* dataset_points;
  uint32_t n;
  float max_dist;
  xform Tsm, Tms;
  uint8_t* hits;
  float* dists;
  float* points;
  float* normals;
  uint32_t* face_ids;
  // tracking: the record each point was closest to in the PREVIOUS call (nullable / kNone: none) and where this call's go.
  // The previous triangle is an actual candidate, so its distance is an upper bound of the answer: the query starts with it
  // and only visits boxes at most that far away -- the same result (min d2, then min face id), a fraction of the leaf visits
  // when the pose changed little between the calls (the case of an ICP loop).
  const uint32_t* seed_rec;
  uint32_t* rec_out;
  uint32_t n_tris;
  // bounded search (opt-in, rmclhip_rcc_set_cpc_bounded): nothing farther than this squared distance is looked for; a point
  // with no surface inside it gets the "not found" outputs.  3e38: unbounded (the reference's semantics).
  float bound_d2;
  // Round 4 -- the map's NEAR GRID (nullable): one triangle record per cell of a uniform grid over the map's box, the record closest
  // to the cell's centre (built once per map, on the first closest-point query).  A point without a tracking seed starts from the
  // record of its cell: an actual candidate, hence an upper bound of the answer within about a cell diagonal of it -- a COLD query
  // (first scan, new pose, no previous call) prunes like a tracked one.  Same results by construction.
  const uint32_t* near_grid;
  uint32_t gn[3];
  float gorg[3], ginv[3];
  // (grid build only) the query points are the centres of cells [0, n) of THIS grid instead of dataset_points
  uint32_t from_cells;
  uint32_t cn[3];
  float corg[3], cinv[3];
  float skip_d2;   // (grid build) a cell whose seed is farther than this gets NO record: query points live near the surface, and cells
                   // deep inside free space -- the centre of a hollow sphere: everything equidistant -- are the searches that cannot prune
};

// seed of a closest-point query from a record index: the record's own closest point (the query's arithmetic: revisiting the record
// changes nothing); sr >= n_tris: no seed
__device__ __forceinline__ NearHit near_seed_from_record(const uint32_t* __restrict__ tris, uint32_t sr, uint32_t n_tris, f3 Pm, bool usable) {
  NearHit seed;
  seed.d2 = 3.0e38f; seed.face = kInvalidFace; seed.rec = 0; seed.p = mk3(0.f, 0.f, 0.f);
  if (usable && sr < n_tris) {
    const uint4* tp = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(sr) * 4u;
    const uint4 a = tp[0], b = tp[1], cc = tp[2], d = tp[3];
    const f3 cq = closest_point_triangle(mk3(asf(a.x), asf(a.y), asf(a.z)), mk3(asf(a.w), asf(b.x), asf(b.y)),
                                         mk3(asf(b.z), asf(b.w), asf(cc.x)), Pm);
    const f3 df = sub3(Pm, cq);
    seed.d2 = (df.x * df.x + df.y * df.y) + df.z * df.z;
    seed.face = d.w; seed.rec = sr; seed.p = cq;
  }
  return seed;
}
// the record of the near-grid cell a point falls into (points outside the grid take the nearest cell: any record is a candidate)
__device__ __forceinline__ uint32_t near_grid_record(const uint32_t* __restrict__ cells, const uint32_t* gn, const float* gorg, const float* ginv, f3 Pm) {
  const float fx = fminf(fmaxf((Pm.x - gorg[0]) * ginv[0], 0.0f), static_cast<float>(gn[0] - 1u));
  const float fy = fminf(fmaxf((Pm.y - gorg[1]) * ginv[1], 0.0f), static_cast<float>(gn[1] - 1u));
  const float fz = fminf(fmaxf((Pm.z - gorg[2]) * ginv[2], 0.0f), static_cast<float>(gn[2] - 1u));
  return cells[(static_cast<uint32_t>(fz) * gn[1] + static_cast<uint32_t>(fy)) * gn[0] + static_cast<uint32_t>(fx)];
}

// kQuad: four lanes per dataset point (64 points per block) instead of one
template <bool kQuad>
__global__ void __launch_bounds__(256) k_cpc_find(const CpcParams p) {
  extern __shared__ uint32_t lds_dyn[];
  const uint32_t sub = threadIdx.x & 3u;
  const uint32_t i = kQuad ? (blockIdx.x * 64u + (threadIdx.x >> 2)) : (blockIdx.x * blockDim.x + threadIdx.x);
  const bool live = i < p.n;
  const uint32_t ii = live ? i : 0u;
  f3 Pm;
  if (p.from_cells) {
    const uint32_t cx = ii % p.cn[0], cyz = ii / p.cn[0], cy = cyz % p.cn[1], cz = cyz / p.cn[1];
    Pm = mk3(p.corg[0] + (static_cast<float>(cx) + 0.5f) / p.cinv[0], p.corg[1] + (static_cast<float>(cy) + 0.5f) / p.cinv[1],
             p.corg[2] + (static_cast<float>(cz) + 0.5f) / p.cinv[2]);
  } else {
    const float* dp = p.dataset_points + 3 * static_cast<size_t>(ii);
    Pm = xapply(p.Tsm, mk3(dp[0], dp[1], dp[2]));
  }
  const bool finite = (Pm.x == Pm.x) && (Pm.y == Pm.y) && (Pm.z == Pm.z);
  NearHit h, seed;
  {
    uint32_t sr = (p.seed_rec != nullptr) ? p.seed_rec[ii] : kNone;
    // no tracking seed: the record of the point's cell
    if (sr >= p.n_tris && p.near_grid != nullptr && live && finite) sr = near_grid_record(p.near_grid, p.gn, p.gorg, p.ginv, Pm);
    seed = near_seed_from_record(p.tris, sr, p.n_tris, Pm, live && finite);
  }
  if (p.from_cells && seed.d2 > p.skip_d2) {
    if (live && p.rec_out != nullptr && (!kQuad || sub == 0u)) p.rec_out[i] = kNone;
    return;   // (no barrier follows in this kernel)
  }
  if (!(seed.d2 <= p.bound_d2)) {   // bounded search: start from the bound itself (no candidate: every face wins a tie against it)
    seed.d2 = p.bound_d2; seed.face = kInvalidFace; seed.rec = 0; seed.p = mk3(0.f, 0.f, 0.f);
  }
  if (kQuad) nearest_quad(p.nodes, p.tris, Pm, live && finite, sub, threadIdx.x >> 2, lds_dyn, h, &seed);
  else nearest_lane_ww<16>(p.nodes, p.tris, Pm, live && finite, lds_dyn + threadIdx.x, blockDim.x, h, &seed);
  if (!live) return;
  if (p.rec_out != nullptr && (!kQuad || sub == 0u)) p.rec_out[i] = (h.face != kInvalidFace) ? h.rec : kNone;
  // quad: the four lanes of a point hold the same result and share the stores
  const bool w0 = !kQuad || sub == 0u, w1 = !kQuad || sub == 1u, w2 = !kQuad || sub == 2u;
  if (h.face != kInvalidFace) {
    const float d = sqrtf(h.d2);
    if (p.hits && w0) p.hits[i] = (d <= p.max_dist) ? 1 : 0;
    if (p.dists && w0) p.dists[i] = d;
    if (p.points && w1) {
      const f3 ps = xapply(p.Tms, h.p);
      p.points[3 * i] = ps.x; p.points[3 * i + 1] = ps.y; p.points[3 * i + 2] = ps.z;
    }
    if (p.normals && w2) {
      const uint4 nrec = reinterpret_cast<const uint4*>(p.tris)[static_cast<size_t>(h.rec) * 4u + 3u];
      const f3 ns = qrot(p.Tms.R, mk3(asf(nrec.x), asf(nrec.y), asf(nrec.z)));
      p.normals[3 * i] = ns.x; p.normals[3 * i + 1] = ns.y; p.normals[3 * i + 2] = ns.z;
    }
    if (p.face_ids && w0) p.face_ids[i] = h.face;
  } else {
    const float qn = __uint_as_float(0x7FC00000u);
    if (p.hits && w0) p.hits[i] = 0;
    if (p.dists && w0) p.dists[i] = qn;
    if (p.points && w1) { p.points[3 * i] = qn; p.points[3 * i + 1] = qn; p.points[3 * i + 2] = qn; }
    if (p.normals && w2) { p.normals[3 * i] = qn; p.normals[3 * i + 1] = qn; p.normals[3 * i + 2] = qn; }
    if (p.face_ids && w0) p.face_ids[i] = kInvalidFace;
  }
}

// rmagine PinholeModel::getDirection: optical ray ((hid - cx)/fx, (vid - cy)/fy, 1) normalised (Vector::normalize
// = divide by sqrt(x*x + y*y + z*z)), then optical (x right, y down, z forward) -> sensor (x forward, y left, z up).
// Same operation order as oracle/rmcl_oracle.c:orc_pinhole_direction (IEEE division / sqrt on both sides).
__device__ __forceinline__ f3 pinhole_direction(float fx, float fy, float cx, float cy, uint32_t vid, uint32_t hid) {
  const float pX = (static_cast<float>(hid) - cx) / fx;
  const float pY = (static_cast<float>(vid) - cy) / fy;
  const float d = sqrtf((pX * pX + pY * pY) + 1.0f * 1.0f);
  return mk3(1.0f / d, -(pX / d), -(pY / d));
}

}  // namespace
}  // namespace rmclhip
