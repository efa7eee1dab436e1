// layout.h -- HBM data layout of a map (shared by the host builder and the kernels).
//
// BVH4 node, 128 B (32 dwords), 128-B aligned: one wave fetches a whole node with two
// s_load_dwordx16 (packet traversal) or seven global_load_dwordx4 per lane (per-lane
// traversal).  Per axis the four children's lower planes and upper planes are two separate
// 16-B groups:
//
//   dword  0.. 3  x lower planes of children 0..3     dword  4.. 7  x upper planes
//   dword  8..11  y lower planes                      dword 12..15  y upper planes
//   dword 16..19  z lower planes                      dword 20..23  z upper planes
//   dword 24..27  child[4]
//   dword 28      number of valid children (1..4)        dword 29..31 reserved
//
// A ray knows from the sign of its direction which plane of an axis it enters through, so a lane
// fetches "the four near planes" and "the four far planes" of an axis with two dwordx4 loads whose
// offsets (0 / 16 B inside the axis group) are per-ray constants: the slab test needs no min / max to
// order the two planes, the plane distances of two children are ONE packed FMA (v_pk_fma_f32), and the
// entry / exit distance of a child is one v_max3 / v_min3.
//
// child reference: bit31 = 0 -> index of another Node4
//                  bit31 = 1 -> leaf: bits 28..30 = count-1 (1..8 triangles),
//                                     bits  0..27 = index of the first TriRec
// Unused slots hold a degenerate box far outside any scene (min = max = kFarPoint: the slab test can never
// accept it for a finite search interval) and a reference to leaf {record 0, count 1}, so that no kernel
// needs an "is this slot empty" test; should a pathological ray ever pass the box test, it only re-tests
// triangle 0.
//
// Triangle record, 64 B (16 dwords), stored in leaf order:
//   v0.xyz | e1.xyz (= v0-v1) | e2.xyz (= v2-v0) | Ng.xyz (= cross(e2,e1)) | n.xyz (unit) | face_id
// (Embree Triangle4 layout restated; the unit normal and the ORIGINAL face id ride in the
//  last 16 B so the epilogue fetches both with one dwordx4 load.)
#pragma once
#include <cstdint>

namespace rmclhip {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNodeDwords = 32;
constexpr uint32_t kTriDwords = 16;
constexpr uint32_t kMaxLeafTris = 4;
constexpr uint32_t kFrontierDepth = 4;   // BFS depth of the map's frontier table (bvh_build.h): <= 256 entries
constexpr uint32_t kPfLeafTris = 2;   // leaf size of the particle filter's own tree (bvh_build.h)
constexpr uint32_t kInvalidFace = 0xFFFFFFFFu;
constexpr float kFarPoint[3] = {1.0e30f, 2.0e30f, 3.0e30f};

struct alignas(128) Node4 {
  float x[8], y[8], z[8];  // per axis: lower planes of children 0..3, then upper planes of children 0..3
  uint32_t child[4];
  uint32_t n_children;
  uint32_t reserved[3];
};
static_assert(sizeof(Node4) == 128, "Node4 must be 128 B");

// Quantised twin of a Node4, 64 B (16 dwords), same index and same child references: the children's (padded) boxes
// as 8-bit offsets from the node's own corner, rounded outwards.  A lane reads it with FOUR dwordx4 loads
// instead of seven -- the incoherent traversals (particle filter, pose batches) are bound by the number of L1
// cache-line accesses, i.e. by load instructions per node visit, not by bytes or arithmetic.
//   dword  0.. 2  origin xyz         dword 3..5  scale xyz (plane = origin + q * scale)
//   dword  6      x lower q of children 0..3 (one byte each)   dword  7  x upper q
//   dword  8 / 9  y lower / upper                              dword 10 / 11  z lower / upper
//   dword 12..15  child[4]
// Unused slots: lower q = 255, upper q = 0 (an inverted box is never entered) + the harmless leaf reference.
struct alignas(64) Node4Q {
  float origin[3];
  float scale[3];
  uint32_t qx_lo, qx_hi, qy_lo, qy_hi, qz_lo, qz_hi;
  uint32_t child[4];
};
static_assert(sizeof(Node4Q) == 64, "Node4Q must be 64 B");

// Child-major twin of a Node4, 128 B, same index: child c occupies 32 B = {lo.x lo.y lo.z hi.x | hi.y hi.z ref pad}.
// The quad-cooperative traversals give each of a ray's four lanes ONE child: two dwordx4 loads per lane and node
// visit instead of seven dword loads.
struct alignas(128) Node4C {
  struct Child { float lo[3]; float hix; float hiy, hiz; uint32_t ref; uint32_t pad; } c[4];
};
static_assert(sizeof(Node4C) == 128, "Node4C must be 128 B");

// 16-wide twin of a node, 512 B, same index (round 6, built on the device from the Node4C array): entry 4 c + g = grandchild g of child
// c with the box child c's node stores for it; a child that is a leaf sits in entry 4 c itself; everything else is the far point.
// The cooperative descent of find kind 32 tests one entry per lane: two levels of the tree per round trip.
struct alignas(128) Node16C {
  Node4C::Child e[16];
};
static_assert(sizeof(Node16C) == 512, "Node16C must be 512 B");

struct alignas(64) TriRec {
  float v0[3];
  float e1[3];
  float e2[3];
  float Ng[3];
  float n[3];
  uint32_t face_id;
};
static_assert(sizeof(TriRec) == 64, "TriRec must be 64 B");

inline uint32_t make_leaf_ref(uint32_t first, uint32_t count) {
  return kLeafBit | ((count - 1u) << 28) | (first & 0x0FFFFFFFu);
}

}  // namespace rmclhip
