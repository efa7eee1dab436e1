// tag_handoff.hip -- WHY did the host see a completion tag before the results it announces?  (profiles/r05_tag_handoff.txt)
// Round 3 measured (tools/determinism2.py, ~1 in 10^4 calls): a kernel writes a 64-B result block to pinned host memory, executes
// __threadfence_system(), then stores a tag; the polling host sees the NEW tag and, right after it, the PREVIOUS call's block.  The product
// has carried a {sequence number, xor of the result words} tag since then.  This program isolates the hand-off and counts, per variant and
// over N launches, how often the host reads a block that is not the one the tag announces:
//   A  block and tag in DIFFERENT host allocations (the product's layout), block written by one lane, fence, tag          [coherent]
//   B  block (56 B) and tag (8 B) in ONE 64-byte line, written by ONE wave-wide store instruction (16 lanes x 4 B)
//   C  block and tag in the same allocation, different 64-byte lines (128 B apart), one lane, fence, tag
//   D  as A, allocations made with hipHostMallocDefault (no explicit coherent / mapped flags)
//   E  as A, the tag stored with __ATOMIC_RELEASE at system scope instead of fence + relaxed store
//   F  as A, block written by 16 lanes of one wave (one store instruction), fence, tag by lane 0
//   build: hipcc --offload-arch=gfx950 -O3 tools/ubench/tag_handoff.hip -o tools/ubench/tag_handoff      run: tools/ubench/tag_handoff [launches]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

// a result word that depends on the sequence number and its position: a stale or torn block is recognisable
__host__ __device__ inline uint32_t word_of(uint32_t seq, uint32_t k) { return seq * 2654435761u + k * 40503u + 17u; }

template <int kVariant>
__global__ void __launch_bounds__(64) k_publish(uint32_t* block, unsigned long long* tag, uint32_t seq, const float* spin_src, float* sink) {
  // a little work in front, so that the stores are not the first thing the wave does
  float a = spin_src[threadIdx.x & 15];
  for (int i = 0; i < 64; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.678f) sink[0] = a;
  if (kVariant == 1) {
    // one 64-byte line: words 0..13 = block, words 14 / 15 = tag {seq, seq ^ 0xA5A5A5A5}
    if (threadIdx.x < 16) {
      const uint32_t k = threadIdx.x;
      const uint32_t v = k < 14 ? word_of(seq, k) : (k == 14 ? seq : (seq ^ 0xA5A5A5A5u));
      __hip_atomic_store(block + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  if (kVariant == 5) {
    if (threadIdx.x < 16) block[threadIdx.x] = word_of(seq, threadIdx.x);
  } else if (threadIdx.x == 0) {
    for (uint32_t k = 0; k < 16; ++k) block[k] = word_of(seq, k);
  }
  if (threadIdx.x == 0) {
    if (kVariant == 4) {
      __hip_atomic_store(tag, (static_cast<unsigned long long>(seq ^ 0xA5A5A5A5u) << 32) | seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      __threadfence_system();
      __hip_atomic_store(tag, (static_cast<unsigned long long>(seq ^ 0xA5A5A5A5u) << 32) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <int kVariant>
static void run(const char* what, uint32_t n, unsigned flags, bool same_alloc) {
  uint32_t* block = nullptr;
  unsigned long long* tag = nullptr;
  char* base = nullptr;
  if (kVariant == 1 || same_alloc) {
    if (hipHostMalloc(reinterpret_cast<void**>(&base), 4096, flags) != hipSuccess) { std::printf("%s: hipHostMalloc failed\n", what); return; }
    block = reinterpret_cast<uint32_t*>(base);
    tag = reinterpret_cast<unsigned long long*>(base + (kVariant == 1 ? 56 : 128));
  } else {
    if (hipHostMalloc(reinterpret_cast<void**>(&block), 64, flags) != hipSuccess || hipHostMalloc(reinterpret_cast<void**>(&tag), 16, flags) != hipSuccess) {
      std::printf("%s: hipHostMalloc failed\n", what);
      return;
    }
  }
  for (int k = 0; k < 16; ++k) block[k] = 0;
  *tag = 0;
  float *spin = nullptr, *sink = nullptr;
  (void)hipMalloc(reinterpret_cast<void**>(&spin), 64);
  (void)hipMalloc(reinterpret_cast<void**>(&sink), 64);
  (void)hipMemset(spin, 0, 64);
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  uint32_t stale = 0, torn = 0, timeouts = 0;
  const uint32_t nwords = kVariant == 1 ? 14u : 16u;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t seq = 1; seq <= n; ++seq) {
    hipLaunchKernelGGL((k_publish<kVariant>), dim3(1), dim3(64), 0, s, block, tag, seq, spin, sink);
    volatile const unsigned long long* vt = tag;
    uint32_t spins = 0;
    for (;;) {
      const unsigned long long t = *vt;
      if (static_cast<uint32_t>(t) == seq && static_cast<uint32_t>(t >> 32) == (seq ^ 0xA5A5A5A5u)) break;
#if defined(__x86_64__)
      _mm_pause();
#endif
      if (++spins > 200000000u) { ++timeouts; break; }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    volatile const uint32_t* vb = block;
    uint32_t n_new = 0, n_old = 0;
    for (uint32_t k = 0; k < nwords; ++k) {
      const uint32_t w = vb[k];
      n_new += w == word_of(seq, k);
      n_old += w == word_of(seq - 1, k) || (seq == 1 && w == 0);
    }
    if (n_new != nwords) { if (n_old == nwords) ++stale; else ++torn; }
  }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
  std::printf("%-96s %9u launches  stale %6u  torn %6u  timeouts %u  %.2f us / hand-off\n", what, n, stale, torn, timeouts, us);
  std::fflush(stdout);
  (void)hipStreamSynchronize(s);
  (void)hipStreamDestroy(s);
  (void)hipFree(spin); (void)hipFree(sink);
  if (base) (void)hipHostFree(base); else { (void)hipHostFree(block); (void)hipHostFree(tag); }
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? static_cast<uint32_t>(std::atoi(argv[1])) : 1000000u;
  const unsigned coh = hipHostMallocMapped | hipHostMallocCoherent;
  run<0>("A  block / tag in different allocations, one lane, fence, tag            (coherent: product layout)", n, coh, false);
  run<1>("B  block + tag in ONE 64-B line, one wave-wide store                     (coherent)", n, coh, true);
  run<2>("C  same allocation, lines 128 B apart, one lane, fence, tag              (coherent)", n, coh, true);
  run<3>("D  as A with hipHostMallocDefault", n, hipHostMallocDefault, false);
  run<4>("E  as A, tag stored with release semantics at system scope               (coherent)", n, coh, false);
  run<5>("F  as A, block written by one wave-wide store, fence, tag                (coherent)", n, coh, false);
  return 0;
}
