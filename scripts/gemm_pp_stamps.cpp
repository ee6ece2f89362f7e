// Cycle stamps of the ping-pong GEMM main loop (csrc/gemm5_tile.hpp compiled with GOAT_G5_TIMING): per-wave cycle sums of
// {LDS-DMA issue, fragment reads, barrier after MEM, MFMAs, vmcnt wait, barrier after MFMA} for the DMA-placement variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGOAT_G5_TIMING=1 -o scripts/gemm_pp_stamps.bin scripts/gemm_pp_stamps.cpp -Iinclude -Ivln-goat_amd/csrc
#include "gemm5_tile.hpp"
#include <stdio.h>
#include <string.h>
#include <vector>
using namespace goat_g5;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

static uint64_t rs = 88172645463325252ull;
static inline uint16_t rbf() {
  rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
  float f = (float)((rs >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
  uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16);
}
static void* devrand(size_t n) {
  std::vector<uint16_t> h(n);
  for (auto& x : h) x = rbf();
  void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice)); return d;
}

template <class CF, bool TA, bool TB, typename OutT>
static void run(const char* name, int M, int N, int K, bool zero = false) {
  void* A = devrand((size_t)M * K); void* B = devrand((size_t)N * K);
  if (zero) { CK(hipMemset(A, 0, (size_t)M * K * 2)); CK(hipMemset(B, 0, (size_t)N * K * 2)); }
  void* C; CK(hipMalloc(&C, (size_t)M * N * 4));
  G2Args a;
  a.A = A; a.B = B; a.C = C; a.bias = nullptr;
  a.lda = TA ? M : K; a.ldb = TB ? N : K; a.ldc = N; a.ldaux = 0;
  a.M = M; a.N = N; a.Kc = K;
  a.tiles_m = (M + CF::BM - 1) / CF::BM; a.tiles_n = (N + CF::BN - 1) / CF::BN;
  a.k_tiles_per_split = (K + 63) / 64;
  a.a_bytes = (uint32_t)((size_t)M * K * 2); a.b_bytes = (uint32_t)((size_t)N * K * 2);
  a.colsum = nullptr; a.accum = 0; a.group_m = a.tiles_m < 4 ? a.tiles_m : 4;
  const int nwg = a.tiles_m * a.tiles_n;
  uint32_t* st; CK(hipMalloc(&st, (size_t)nwg * 8 * 8 * 4)); CK(hipMemset(st, 0, (size_t)nwg * 8 * 8 * 4));
  a.aux = st;
  auto kern = pp_kernel<CF, TA, TB, OutT, GOAT_EPI_NONE, false>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 12; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), CF::SMEM, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  std::vector<uint32_t> h((size_t)nwg * 64);
  CK(hipMemcpy(h.data(), st, h.size() * 4, hipMemcpyDeviceToHost));
  const int nkt = (K + 63) / 64;
  double g[2][7] = {{0}};

  for (int b = 0; b < nwg; ++b) for (int w = 0; w < 8; ++w) for (int i = 0; i < 7; ++i) g[w >> 2][i] += h[((size_t)b * 8 + w) * 8 + i];
  printf("%-24s t%d%d %dx%dx%d%s: %d tiles, %d K-tiles, %.1f us (%.0f TF)  clock >= %.2f GHz (rounds x cycles per tile / time)\n", name, TA, TB, M, N, K, zero ? " ZERO-FILLED" : "", nwg, nkt,
         best * 1e3, 2.0 * M * N * K / best * 1e-9, ((nwg + 255) / 256) * (g[0][6] / (nwg * 4.0)) / (best * 1e6));
  const double dn = nwg * 4.0 * nkt;
  for (int q = 0; q < 2; ++q) {
    printf("   group %d per K-tile: dma-issue %5.0f  reads %5.0f  barrier %5.0f | mfma %5.0f  vmwait %5.0f  barrier %5.0f | period %5.0f cycles (ideal %d)   kernel-total %7.0f\n", q,
           g[q][0] / dn, g[q][1] / dn, g[q][2] / dn, g[q][3] / dn, g[q][4] / dn, g[q][5] / dn,
           (g[q][0] + g[q][1] + g[q][2] + g[q][3] + g[q][4] + g[q][5]) / dn, 2 * CF::MI * CF::NI * 4 * 32, g[q][6] / (nwg * 4.0));
  }
  printf("   block 0 waves: dma [");
  for (int w = 0; w < 8; ++w) printf("%u ", h[w * 8 + 0] / nkt);
  printf("] reads [");
  for (int w = 0; w < 8; ++w) printf("%u ", h[w * 8 + 1] / nkt);
  printf("] mfma [");
  for (int w = 0; w < 8; ++w) printf("%u ", h[w * 8 + 3] / nkt);
  printf("]\n");
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(st));
}

// pure MFMA stream (no memory traffic): what clock does the chip sustain on the matrix pipe alone, for random / zero operands?
template <int NT> __global__ __launch_bounds__(NT) void mfma_only(uint32_t* out, int iters, uint32_t seed) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      uint32_t h = (threadIdx.x * 2654435761u + i * 97u + e * 13u + blockIdx.x) * seed;
      h ^= h >> 15; h *= 0x7FEB352Du; h ^= h >> 13;
      const float fa = seed ? ((float)(h & 0xFFFF) / 32768.0f - 1.0f) : 0.f, fb = seed ? ((float)(h >> 16) / 32768.0f - 1.0f) * 0.05f : 0.f;
      a[i][e] = (bf16_t)fa; b[i][e] = (bf16_t)fb;
    }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const uint32_t t0 = g5_now();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + k) & 3], b[k], acc[i], 0, 0, 0);
  }
  const uint32_t t1 = g5_now();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t1 - t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = (uint32_t)s; }
}
static void mfma_clock(uint32_t seed, int threads = 512) {
  uint32_t* d; CK(hipMalloc(&d, 256 * 16 * 2 * 4)); CK(hipMemset(d, 0, 256 * 16 * 2 * 4));
  const int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0, 0));
    if (threads == 256) hipLaunchKernelGGL(mfma_only<256>, dim3(256), dim3(256), 0, 0, d, iters, seed);
    else if (threads == 512) hipLaunchKernelGGL(mfma_only<512>, dim3(256), dim3(512), 0, 0, d, iters, seed);
    else hipLaunchKernelGGL(mfma_only<768>, dim3(256), dim3(768), 0, 0, d, iters, seed);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  std::vector<uint32_t> h(256 * 32); CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  const int nw = threads / 64;
  double cyc = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) cyc += h[(b * 16 + w) * 2]; cyc /= 256.0 * nw;
  const double flop = 256.0 * nw * iters * 32 * (2.0 * 32 * 32 * 16);
  printf("MFMA only (%s operands, %d waves/SIMD): %.1f ms, %.0f TF, %.1f s_memtime ticks per MFMA per SIMD, %.2f ns per MFMA per SIMD, tick rate %.2f GHz\n", seed ? "random" : "zero", nw / 4, best,
         flop / best * 1e-9, cyc / (iters * 32.0) / (nw / 4.0), best * 1e6 / (iters * 32.0 * (nw / 4.0)), cyc / (best * 1e6));
  CK(hipFree(d));
}

int main() {
  CK(hipSetDevice(0));
  mfma_clock(12345u, 256); mfma_clock(12345u, 512); mfma_clock(12345u, 768);
  mfma_clock(0u, 256); mfma_clock(0u, 512); mfma_clock(0u, 768);
  if (getenv("MFMA_ONLY")) return 0;
  run<PCfg<4, 2, 2, 0>, false, false, bf16_t>("pp256 dma-first", 8192, 8192, 8192);
  run<PCfg<4, 2, 2, 1>, false, false, bf16_t>("pp256 dma-interleaved", 8192, 8192, 8192);
  run<PCfg<4, 2, 2, 1>, false, false, bf16_t>("pp256 dma-interleaved", 8192, 8192, 8192, true);
  run<PCfg<4, 2, 2, 2>, false, false, bf16_t>("pp256 dma-last", 8192, 8192, 8192);
  run<PCfg<4, 2, 3, 1>, false, false, bf16_t>("pp256b3 dma-interleaved", 8192, 8192, 8192);
  run<PCfg<4, 2, 2, 0>, false, false, bf16_t>("pp256 dma-first", 3840, 3072, 768);
  run<PCfg<4, 2, 2, 1>, false, false, bf16_t>("pp256 dma-interleaved", 3840, 3072, 768);
  run<PCfg<4, 2, 2, 2>, false, false, bf16_t>("pp256 dma-last", 3840, 3072, 768);
  run<PCfg<4, 2, 2, 0>, true, true, float>("pp256 dma-first", 3072, 3072, 3840);
  run<PCfg<4, 2, 2, 1>, true, true, float>("pp256 dma-interleaved", 3072, 3072, 3840);
  run<PCfg<4, 2, 2, 2>, true, true, float>("pp256 dma-last", 3072, 3072, 3840);
  run<PCfg<2, 2, 2, 0>, false, false, bf16_t>("pp128x256 dma-first", 8640, 768, 3072);
  run<PCfg<2, 2, 2, 1>, false, false, bf16_t>("pp128x256 dma-interl", 8640, 768, 3072);
  run<PCfg<2, 2, 2, 2>, false, false, bf16_t>("pp128x256 dma-last", 8640, 768, 3072);
  return 0;
}
