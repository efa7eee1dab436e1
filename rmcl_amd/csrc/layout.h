// layout.h -- HBM data layout of a map (shared by the host builder and the kernels).
//
// BVH4 node, 128 B (32 dwords), 128-B aligned: one wave fetches a whole node with two
// s_load_dwordx16 (packet traversal) or seven global_load_dwordx4 per lane (per-lane
// traversal).  Child boxes are SoA so that child c's six bounds sit at dword c of six
// consecutive 16-B vectors.
//
//   dword  0.. 3  minx[4]      dword 12..15  maxx[4]     dword 24..27  child[4]
//   dword  4.. 7  miny[4]      dword 16..19  maxy[4]     dword 28..31  reserved
//   dword  8..11  minz[4]      dword 20..23  maxz[4]
//
// child reference: bit31 = 0 -> index of another Node4
//                  bit31 = 1 -> leaf: bits 28..30 = count-1 (1..8 triangles),
//                                     bits  0..27 = index of the first TriRec
//                  0xFFFFFFFF   -> empty slot (its box is inverted: never hit)
//
// Triangle record, 64 B (16 dwords), stored in leaf order:
//   v0.xyz | e1.xyz (= v0-v1) | e2.xyz (= v2-v0) | Ng.xyz (= cross(e2,e1)) | n.xyz (unit) | face_id
// (Embree Triangle4 layout restated; the unit normal and the ORIGINAL face id ride in the
//  last 16 B so the epilogue fetches both with one dwordx4 load.)
#pragma once
#include <cstdint>

namespace rmclhip {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kEmptyRef = 0xFFFFFFFFu;
constexpr uint32_t kNodeDwords = 32;
constexpr uint32_t kTriDwords = 16;
constexpr uint32_t kMaxLeafTris = 4;
constexpr uint32_t kInvalidFace = 0xFFFFFFFFu;

struct alignas(128) Node4 {
  float minx[4], miny[4], minz[4];
  float maxx[4], maxy[4], maxz[4];
  uint32_t child[4];
  uint32_t reserved[4];
};
static_assert(sizeof(Node4) == 128, "Node4 must be 128 B");

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
