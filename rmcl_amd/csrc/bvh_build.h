// bvh_build.h -- host-side BVH construction for librmclhip (binned SAH BVH2 -> BVH4 collapse,
// breadth-first node order so that the top of the tree is one contiguous, cache/LDS-friendly
// prefix).  Replaces the scene commit the reference delegates to Embree / OptiX
// (rm::import_embree_map, micp_localization.cpp:187-195).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "layout.h"

namespace rmclhip {

struct BvhInfo {
  uint32_t n_faces = 0, n_vertices = 0, n_nodes = 0, max_depth = 0, stack_need = 0;
  float bbox_min[3] = {0, 0, 0}, bbox_max[3] = {0, 0, 0};
  float pad = 0.f;
};

struct BvhHost {
  std::vector<Node4> nodes;
  std::vector<Node4Q> qnodes;  // quantised twins, same indices
  std::vector<Node4C> cnodes;  // child-major twins, same indices
  std::vector<TriRec> tris;  // leaf order
  BvhInfo info;
};

// returns empty string on success, else an error message
std::string build_bvh(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, BvhHost& out);

}  // namespace rmclhip
