// valu_rate.hip -- how many cycles does ONE SIMD of gfx950 need per wave64 VALU instruction?  (profiles/r04_valu_issue_rate.txt)
// DESIGN's round-2/3 analysis of k_pf_update_v3 priced a VALU instruction at 4 cycles per SIMD ("issue-bound": 1.03 G instructions x
// 4 / 1024 SIMDs = the kernel's duration); /opt/skills/guides/MI355X_MICROARCH.md lists v_fma_f32 at 2 cycles (SIMD-32).  This
// measures it for the instruction classes the traversal is made of, with 1 / 2 / 4 / 8 waves per SIMD and independent / dependent
// chains:  build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int kOp, bool kDep>
__global__ void __launch_bounds__(256) k_rate(float* out, uint32_t iters, unsigned long long* clk) {
  float a[8];
  uint32_t u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 7u + i; }
  const float b = 1.0001f, c = 0.5f;
  unsigned long long mask = 0x5555555555555555ull;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = kDep ? 0 : i;     // dependent: everything through accumulator 0
        if (kOp == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
        else if (kOp == 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
        else if (kOp == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
        else if (kOp == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
        else if (kOp == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[j]), "v"(b) : "vcc");
        else if (kOp == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&a[(j & 3) * 2])) : "v"(*reinterpret_cast<const double*>(&a[0])));
        else if (kOp == 6) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
        else if (kOp == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
        else if (kOp == 8) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[j]) : "v"(u[j]));
        else if (kOp == 9) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
        else if (kOp == 10) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "s"(mask));
        else if (kOp == 11) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
        else if (kOp == 12) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
        else if (kOp == 13) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(u[j]) : "v"(u[(j + 1) & 7]));
        else if (kOp == 14) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[j]) : "v"(u[j]));
        else if (kOp == 15) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
        else if (kOp == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(u[j]) : "v"(u[(j + 1) & 7]));
        else if (kOp == 17) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
        else if (kOp == 18) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(mask) : "v"(u[j]), "v"(u[(j + 1) & 7]));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + static_cast<float>(u[i]);
  s += static_cast<float>(mask & 1ull);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63u) == 0u) clk[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int kOp, bool kDep>
static void run(const char* name, int waves_per_simd) {
  // one block of 256 threads = 4 waves = one per SIMD of a CU; `waves_per_simd` blocks per CU, 256 CUs
  const int blocks = 256 * waves_per_simd;
  const uint32_t iters = 4096;
  float* out; unsigned long long* clk;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&clk, sizeof(unsigned long long) * blocks * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<kOp, kDep>), dim3(blocks), dim3(256), 0, 0, out, 16u, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<kOp, kDep>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += double(v); mean /= h.size();
  const double n_inst = double(iters) * 32.0;           // VALU instructions per wave
  std::printf("%-20s %-11s %d waves/SIMD: %6.2f cycles per instruction per wave (in-kernel clock) -> %5.2f cycles per instruction per SIMD; "
              "%.3f ms wall = %.2f cycles/instr/SIMD at 2.4 GHz\n", name, kDep ? "dependent" : "independent", waves_per_simd,
              mean / n_inst, mean / n_inst / waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (n_inst * waves_per_simd));
  hipFree(out); hipFree(clk);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0, false>("v_fma_f32", w);     run<0, true>("v_fma_f32", w);
    run<1, false>("v_min_f32", w);     run<2, false>("v_cndmask(vcc)", w);
    run<3, false>("v_add_u32", w);     run<4, false>("v_cmp_lt_f32", w);
    run<5, false>("v_pk_fma_f32", w);  run<6, false>("v_max3_f32", w);   run<7, false>("v_mul_f32", w);
    run<8, false>("v_cvt_f32_ubyte1", w);  run<9, false>("v_min_u32", w);  run<10, false>("v_cndmask(sgpr)", w);
    run<11, false>("v_perm_b32", w);   run<12, false>("v_lshl_add_u32", w);  run<13, false>("v_bfe_u32", w);
    run<14, false>("v_cvt_f32_u32", w);  run<15, false>("v_and_or_b32", w);  run<16, false>("v_mov_b32", w);
    run<17, false>("v_max3_u32", w);   run<18, false>("v_cmp_lt_u32(sgpr)", w);
  }
  return 0;
}
