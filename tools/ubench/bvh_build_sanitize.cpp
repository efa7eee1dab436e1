// bvh_build_sanitize.cpp -- the threaded host builder (rmcl_amd/csrc/bvh_build.cpp) under ThreadSanitizer / AddressSanitizer + UBSan,
// without a GPU.  profiles/r05_asan_sweep.txt (host section).
//   g++ -std=c++17 -O1 -g -fsanitize=thread -ffp-contract=off -Irmcl_amd/csrc rmcl_amd/csrc/bvh_build.cpp tools/ubench/bvh_build_sanitize.cpp -o /tmp/bvh_tsan -lpthread
//   g++ ... -fsanitize=address,undefined -fno-sanitize-recover=undefined ... -o /tmp/bvh_asan
//   RMCLHIP_BUILD_THREADS=8 /tmp/bvh_tsan <mesh: 0 soup | 1 exponential chain | 2 sliver fan | 3 nest> <triangles>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "bvh_build.h"

int main(int argc, char** argv) {
  const int kind = argc > 1 ? std::atoi(argv[1]) : 0;
  const int n = argc > 2 ? std::atoi(argv[2]) : 200000;
  std::vector<float> v;
  std::vector<uint32_t> f;
  auto tri = [&](const float* a, const float* b, const float* c) {
    const uint32_t base = static_cast<uint32_t>(v.size() / 3);
    for (int k = 0; k < 3; ++k) v.push_back(a[k]);
    for (int k = 0; k < 3; ++k) v.push_back(b[k]);
    for (int k = 0; k < 3; ++k) v.push_back(c[k]);
    f.push_back(base); f.push_back(base + 1); f.push_back(base + 2);
  };
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-10.f, 10.f), S(-0.2f, 0.2f);
  for (int i = 0; i < n; ++i) {
    float a[3], b[3], c[3];
    if (kind == 0) {          // random small triangles in a flat box
      const float o[3] = {U(rng), U(rng), U(rng) * 0.2f};
      for (int k = 0; k < 3; ++k) { a[k] = o[k] + S(rng); b[k] = o[k] + S(rng); c[k] = o[k] + S(rng); }
    } else if (kind == 1) {   // sizes grow geometrically along x: the SAH peels one triangle per level (height budget)
      const float s = std::pow(1.05f, static_cast<float>(i % 1500)), x = 20.f * s;
      a[0] = x; a[1] = 0.f; a[2] = 0.f; b[0] = x; b[1] = s; b[2] = 0.f; c[0] = x; c[1] = 0.f; c[2] = s;
    } else if (kind == 2) {   // slivers sharing an apex
      const float t0 = 6.2831853f * static_cast<float>(i) / static_cast<float>(n), t1 = 6.2831853f * static_cast<float>(i + 1) / static_cast<float>(n);
      a[0] = 0.f; a[1] = 0.f; a[2] = 0.3f * std::sin(40.f * t0);
      b[0] = 10.f * std::cos(t0); b[1] = 10.f * std::sin(t0); b[2] = 0.f; c[0] = 10.f * std::cos(t1); c[1] = 10.f * std::sin(t1); c[2] = 0.f;
    } else {                  // concentric triangles, each slightly larger than the last
      const float s = std::pow(1.2f, static_cast<float>(i % 200)) * 1e-3f, z = 1e-3f * static_cast<float>(i);
      a[0] = -s; a[1] = -s; a[2] = z; b[0] = s; b[1] = -s; b[2] = z; c[0] = 0.f; c[1] = s; c[2] = z;
    }
    tri(a, b, c);
  }
  rmclhip::BvhHost out;
  const std::string err = rmclhip::build_bvh(v.data(), static_cast<uint32_t>(v.size() / 3), f.data(), static_cast<uint32_t>(f.size() / 3), out);
  std::printf("mesh %d, %d triangles: '%s' nodes %u (filter %u) stack %u / %u height fallbacks %u guarded %u\n", kind, n, err.c_str(), out.info.n_nodes,
              out.info.n_nodes_pf, out.info.stack_need, out.info.stack_need_pf, out.info.height_fallbacks, out.info.guarded_nodes);
  return err.empty() ? 0 : 1;
}
