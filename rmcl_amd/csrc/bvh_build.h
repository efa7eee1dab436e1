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
  uint32_t n_nodes_pf = 0, max_depth_pf = 0, stack_need_pf = 0;  // the particle filter's tree (leaves <= kPfLeafTris)
  float bbox_min[3] = {0, 0, 0}, bbox_max[3] = {0, 0, 0};
  float pad = 0.f;
  // round 5: how often the stack bound had to be enforced (0 on ordinary meshes): object-median splits the height budget of the BVH2
  // forced, and BVH4 nodes expanded tallest-child-first; stack_need <= 64 holds for every mesh by construction
  uint32_t height_fallbacks = 0, guarded_nodes = 0;
  // round 6: spatial splits the builder took, and the triangle RECORDS they left (>= n_faces: a face referenced by k leaves has k
  // identical records; leaf references and every record index of the kernels count records)
  uint32_t spatial_splits = 0, n_records = 0;
};

struct BvhHost {
  std::vector<Node4> nodes;
  std::vector<Node4Q> qnodes;  // quantised twins, same indices
  std::vector<Node4C> cnodes;  // child-major twins, same indices
  std::vector<Node4> nodes_pf;    // the particle filter's cut of the same BVH2 (leaves <= kPfLeafTris): host-side only
  std::vector<Node4Q> qnodes_pf;  // ... and its quantised form, what k_pf_update_* reads
  std::vector<Node4C::Child> frontier;  // <= 4^kFrontierDepth entries {box, ref}: where a scan's rays can start (layout.h)
  std::vector<Node4C::Child> frontier_pf;   // the same for the filter's tree (nodes_pf / qnodes_pf)
  std::vector<TriRec> tris;  // leaf order (shared by both trees)
  BvhInfo info;
};

// returns empty string on success, else an error message
// max_leaf: largest leaf of the map's tree (1..kMaxLeafTris)
std::string build_bvh(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, BvhHost& out,
                      uint32_t max_leaf = kMaxLeafTris);

}  // namespace rmclhip
