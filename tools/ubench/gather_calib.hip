// gather_calib.hip -- what does rocprofv3's FETCH_SIZE report for the access patterns of a BVH walk?  (profiles/r05_fetch_size_calibration.txt)
// /opt/skills/guides/MI355X_MICROARCH.md calibrates the counter for wide coalesced streams only (it reads 1/2 of the bytes there) and
// calls every other width uncalibrated.  The traversals read RANDOM 128-B nodes (eight 16-B loads per lane), 64-B quantised nodes and
// 64-B triangle records (four 16-B loads per lane).  This kernel does exactly that on a buffer far larger than the 256 MB MALL, every
// lane its own pseudo-random record, so the bytes that must cross the fabric are known: lanes x record size (records are touched once).
//   build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_calib.hip -o tools/ubench/gather_calib
//   run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d DIR -o NAME -- tools/ubench/gather_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int kRecBytes, bool kStream>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ buf, uint32_t n_rec, float* out) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  // a bijection of [0, 2^k): odd multiplier mod 2^k -- every record index is hit at most once when gridsize <= n_rec
  const uint32_t rec = kStream ? gid : ((gid * 2654435761u + 12345u) & (n_rec - 1u));
  const uint4* p = buf + static_cast<size_t>(rec) * (kRecBytes / 16);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kRecBytes / 16; ++i) { const uint4 v = p[i]; s += __uint_as_float(v.x) + __uint_as_float(v.w); }
  if (s == 123.456f) out[gid] = s;
}

template <int kRecBytes, bool kStream>
static void run(const uint4* buf, size_t bytes, float* out, const char* what) {
  uint32_t n_rec = 1;
  while (static_cast<size_t>(n_rec) * 2 * kRecBytes <= bytes) n_rec *= 2;
  const uint32_t lanes = 1u << 22;   // 4 Mi lanes
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k_gather<kRecBytes, kStream>), dim3(lanes / 256), dim3(256), 0, 0, buf, n_rec, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep == 2)
      std::printf("%-44s records %10u x %3d B  expected bytes %12zu  %8.3f ms  %7.1f GB/s\n", what, n_rec, kRecBytes,
                  static_cast<size_t>(lanes) * kRecBytes, ms, lanes * static_cast<double>(kRecBytes) / ms / 1e6);
  }
}

int main() {
  const size_t bytes = size_t{4} << 30;   // 4 GB
  uint4* buf = nullptr;
  float* out = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&buf), bytes) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
  (void)hipMalloc(reinterpret_cast<void**>(&out), (1u << 22) * sizeof(float));
  (void)hipMemset(buf, 1, bytes);
  (void)hipDeviceSynchronize();
  run<128, false>(buf, bytes, out, "k_gather<128,false> random 128-B records");
  run<64, false>(buf, bytes, out, "k_gather<64,false> random 64-B records");
  run<16, false>(buf, bytes, out, "k_gather<16,false> random 16-B loads");
  run<128, true>(buf, bytes, out, "k_gather<128,true> consecutive 128-B records");
  run<16, true>(buf, bytes, out, "k_gather<16,true> coalesced 16 B per lane");
  (void)hipFree(buf); (void)hipFree(out);
  return 0;
}
