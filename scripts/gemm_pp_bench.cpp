// Stand-alone A/B + parity harness for the GEMM tile kernels of libgoat_hip.so (no torch: starts in a second on a fresh box).
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/gemm_pp_bench scripts/gemm_pp_bench.cpp -Iinclude -Lvln-goat_amd/csrc -lgoat_hip
//   LD_LIBRARY_PATH=vln-goat_amd/csrc gpurun_out/gemm_pp_bench [filter]
// Every case runs a BASELINE configuration (a gemm2_tile.hpp tile, the parity-tested kernel) and one or more CANDIDATE
// configurations through the same C entry point (goat_gemm_bf16), compares the results element by element (same contraction order ->
// bit-identical for unsplit launches) and times both on rotating operand buffers with HIP events.  Also: an LDS read-rate probe
// (ds_read_b128 vs ds_read_b64_tr_b16) used to price the transposed-operand layouts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include "goat_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint64_t rng_state = 0x1234567887654321ull;
static inline float frand() {   // uniform [-1, 1)
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}
static void* dev_random_bf16(size_t n, float scale) {
  std::vector<uint16_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = f2bf(frand() * scale);
  void* d;
  CK(hipMalloc(&d, n * 2));
  CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}

struct Conf { int tile, ns; const char* name; };
static inline int T(int bm, int bn) { return bm | (bn << 16); }
#ifndef GOAT_GEMM_PP
#define GOAT_GEMM_PP 0x200
#endif

struct Case {
  int ta, tb, M, N, K, epi, f32, split;
  Conf base;
  std::vector<Conf> cand;
};

static int run_gemm(hipStream_t st, const Case& c, const Conf& cf, const void* A, const void* B, void* C, const float* bias, void* aux, float* colsum) {
  const int64_t lda = c.ta ? c.M : c.K, ldb = c.tb ? c.N : c.K;
  return goat_gemm_bf16(st, c.ta, c.tb, c.f32 ? GOAT_F32 : GOAT_BF16, A, lda, B, ldb, C, c.N, c.M, c.N, c.K, bias, c.epi,
                        (c.epi == GOAT_EPI_MUL_DGELU || c.epi == GOAT_EPI_MUL_DRELU || c.epi == GOAT_EPI_GELU) ? aux : nullptr, c.N, c.split, cf.tile, cf.ns, colsum);
}

static double time_conf(hipStream_t st, const Case& c, const Conf& cf, void** A, void** B, void** C, const float* bias, void* aux, int ROT) {
  const double flop = 2.0 * c.M * c.N * c.K;
  const int n = flop > 1e11 ? 6 : 40;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int k = 0; k < ROT; ++k)
    if (run_gemm(st, c, cf, A[k], B[k], C[k], bias, aux, nullptr)) return -1;
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) run_gemm(st, c, cf, A[i % ROT], B[i % ROT], C[i % ROT], bias, aux, nullptr);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms * 1e3 / n < best) best = ms * 1e3 / n;
  }
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return best;
}

// ---- LDS read-rate probe ---------------------------------------------------------------------------------------------------
// Every wave issues NREAD reads per iteration from a 64 KiB window with the fragment address patterns of the GEMM tiles, then one
// s_waitcnt; cycles per wave-instruction per CU = elapsed cycles * (waves that share the LDS) ... reported per CU.
typedef uint32_t pu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pu32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void lds_probe(uint32_t* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i;
  __syncthreads();
  uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  uint32_t addr;
  if (MODE == 0) {          // ds_read_b128, K-contiguous image [rows][128 B], swizzled
    const int l31 = lane & 31, hi = lane >> 5;
    addr = base + (wave & 3) * 8192 + l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4);
  } else {                  // ds_read_b64_tr_b16, transposed image [64 k][W], W = 256 (MODE 1) / 512 (MODE 2), swizzled
    const int W = MODE == 1 ? 256 : 512;
    const int t15 = lane & 15, g = lane >> 4;
    addr = base + (8 * (g >> 1) + (t15 >> 2)) * W + ((((wave & 3)) ^ (t15 >> 2)) << 6) + (g & 1) * 32 + (t15 & 3) * 8;
  }
  uint32_t sink = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      pu32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(0) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(v[k]));
      sink += v[0].x;
    } else {
      pu32x2 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(0) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
      sink += v[0].x;
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    out[(blockIdx.x * 8 + wave) * 2] = (uint32_t)(t1 - t0);
    out[(blockIdx.x * 8 + wave) * 2 + 1] = sink;
  }
}

template <int MODE>
static void probe(const char* name, int threads) {
  uint32_t* d;
  CK(hipMalloc(&d, 256 * 8 * 2 * 4));
  CK(hipMemset(d, 0, 256 * 8 * 2 * 4));
  const int iters = 2000;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(lds_probe<MODE>, dim3(256), dim3(threads), 65536, 0, d, iters);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> h(256 * 8 * 2);
  CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  double cyc = 0;
  const int nw = threads / 64;
  for (int w = 0; w < nw; ++w) cyc += h[w * 2];
  cyc /= nw;
  const int per_iter = MODE == 0 ? 8 : 16;
  const double bytes = MODE == 0 ? 1024.0 : 512.0;
  const double inst_per_cu = (double)iters * per_iter * nw;
  printf("LDS probe %-28s %d waves/CU: %.2f cycles per wave-instruction (CU aggregate), %.1f B/clk/CU\n", name, nw, cyc / inst_per_cu,
         inst_per_cu * bytes / cyc);
  CK(hipFree(d));
}

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : "";
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  if (strstr(filter, "noprobe") == nullptr) {
    probe<0>("ds_read_b128 [rows][128B]", 256);
    probe<0>("ds_read_b128 [rows][128B]", 512);
    probe<1>("ds_read_b64_tr_b16 W=256", 256);
    probe<1>("ds_read_b64_tr_b16 W=256", 512);
    probe<2>("ds_read_b64_tr_b16 W=512", 256);
    probe<2>("ds_read_b64_tr_b16 W=512", 512);
  }
  const Conf PP256 = {T(256, 256), 2 | GOAT_GEMM_PP, "pp256x256"}, PP192 = {T(192, 256), 2 | GOAT_GEMM_PP, "pp192x256"};
  const Conf PP128x256 = {T(128, 256), 2 | GOAT_GEMM_PP, "pp128x256"}, PP256x128 = {T(256, 128), 2 | GOAT_GEMM_PP, "pp256x128"};
  const Conf PP128 = {T(128, 128), 2 | GOAT_GEMM_PP, "pp128x128"};
  const Conf O256 = {T(256, 256), 2, "g2 256x256 s2"}, O128x8 = {128, 0x102, "g2 128x128 8w s2"}, O128x256 = {T(128, 256), 3, "g2 128x256 s3"};
  const Conf O192x256 = {T(192, 256), 2, "g2 192x256 s2"}, O96 = {96, 3, "g2 96x128 s3"}, O256x128 = {256, 3, "g2 256x128 s3"};
  std::vector<Case> cases = {
      // parity-first small cases (ragged edges, every layout / epilogue)
      {0, 0, 512, 512, 256, 0, 0, 1, O128x8, {PP256, PP128x256, PP256x128, PP128}},
      {0, 0, 300, 520, 192, 0, 0, 1, O128x8, {PP256, PP192, PP128x256, PP256x128, PP128}},
      {0, 1, 300, 520, 192, 0, 0, 1, O128x8, {PP256, PP192, PP128x256, PP256x128, PP128}},
      {1, 1, 520, 264, 200, 0, 1, 1, O128x8, {PP256, PP128x256, PP256x128, PP128}},
      {1, 1, 520, 264, 200, 0, 0, 1, O128x8, {PP256, PP128x256}},
      {0, 0, 300, 520, 64, 0, 1, 1, O128x8, {PP256, PP128}},
      {0, 0, 300, 520, 192, 1, 0, 1, O128x8, {PP256, PP192, PP128x256}},
      {0, 1, 300, 520, 192, 3, 0, 1, O128x8, {PP256, PP192, PP128x256}},
      {0, 0, 300, 520, 192, 2, 0, 1, O128x8, {PP256}},
      {0, 1, 300, 520, 192, 4, 0, 1, O128x8, {PP256}},
      {1, 1, 520, 264, 1000, 0, 1, 3, O128x8, {PP256, PP128x256}},
      // the shapes of the step and of the vendor comparison
      {0, 0, 8192, 8192, 8192, 0, 0, 1, O256, {PP256}},
      {0, 0, 8640, 3072, 768, 0, 0, 1, O256, {PP256, PP128x256, PP256x128}},
      {0, 0, 8640, 3072, 768, 1, 0, 1, O256, {PP256, PP128x256}},
      {0, 0, 3840, 3072, 768, 0, 0, 1, O192x256, {PP256, PP192, PP128x256, PP256x128, PP128}},
      {0, 0, 3840, 3072, 768, 1, 0, 1, O192x256, {PP256, PP192, PP128x256}},
      {0, 1, 3840, 3072, 768, 3, 0, 1, O192x256, {PP256, PP192, PP128x256}},
      {0, 0, 3840, 2304, 768, 0, 0, 1, O128x8, {PP256, PP192, PP128x256, PP256x128, PP128}},
      {0, 0, 3840, 768, 3072, 0, 0, 1, O96, {PP128x256, PP256x128, PP128}},
      {0, 1, 3840, 768, 3072, 0, 0, 1, O96, {PP128x256, PP256x128, PP128}},
      {0, 0, 3840, 768, 768, 0, 0, 1, O96, {PP128x256, PP256x128, PP128}},
      {0, 0, 8640, 768, 3072, 0, 0, 1, O128x256, {PP256, PP128x256, PP256x128, PP128}},
      {0, 1, 8640, 3072, 768, 0, 0, 1, O256, {PP256, PP128x256}},
      {1, 1, 3072, 768, 3840, 0, 1, 1, O256x128, {PP256, PP128x256, PP256x128}},
      {1, 1, 3072, 3072, 3840, 0, 1, 1, O256x128, {PP256, PP128x256, PP256x128}},
      {1, 1, 768, 3072, 8640, 0, 1, 1, O256x128, {PP256, PP128x256, PP256x128}},
      {1, 1, 6144, 3072, 3840, 0, 1, 1, O256x128, {PP256, PP128x256, PP256x128}},
      {0, 0, 20480, 3072, 768, 0, 0, 1, O256, {PP256}},
      {0, 0, 20480, 768, 3072, 0, 0, 1, O128x256, {PP256, PP128x256, PP256x128}},
      {0, 0, 4096, 4096, 4096, 0, 0, 1, O256, {PP256}},
  };
  const int ROT = 4;
  int nbad = 0;
  for (size_t ci = 0; ci < cases.size(); ++ci) {
    const Case& c = cases[ci];
    char tag[128];
    snprintf(tag, sizeof tag, "t%d%d %dx%dx%d epi%d %s split%d", c.ta, c.tb, c.M, c.N, c.K, c.epi, c.f32 ? "f32" : "bf16", c.split);
    if (filter[0] && strstr(filter, "noprobe") == nullptr && strstr(tag, filter) == nullptr) continue;
    const size_t na = (size_t)c.M * c.K, nb = (size_t)c.N * c.K, nc = (size_t)c.M * c.N;
    const int rot = (na + nb + nc) * 2 * ROT > (3ull << 30) ? 2 : ROT;
    void *A[ROT], *B[ROT], *C[ROT];
    for (int k = 0; k < rot; ++k) {
      A[k] = dev_random_bf16(na, 1.0f);
      B[k] = dev_random_bf16(nb, 0.05f);
      CK(hipMalloc(&C[k], nc * 4));
    }
    for (int k = rot; k < ROT; ++k) { A[k] = A[k % rot]; B[k] = B[k % rot]; C[k] = C[k % rot]; }
    void* aux = dev_random_bf16(nc, 1.5f);
    void* aux2;
    CK(hipMalloc(&aux2, nc * 2));
    std::vector<float> hb(c.N);
    for (auto& x : hb) x = frand() * 0.2f;
    float* bias;
    CK(hipMalloc(&bias, c.N * 4));
    CK(hipMemcpy(bias, hb.data(), c.N * 4, hipMemcpyHostToDevice));
    const float* use_bias = (c.ta || c.split > 1 || c.epi == GOAT_EPI_MUL_DGELU || c.epi == GOAT_EPI_MUL_DRELU) ? nullptr : bias;
    float *cs0, *cs1;
    CK(hipMalloc(&cs0, c.M * 4));
    CK(hipMalloc(&cs1, c.M * 4));
    const size_t cbytes = nc * (c.f32 ? 4 : 2);
    std::vector<uint8_t> ref(cbytes), got(cbytes), refaux(nc * 2), gotaux(nc * 2);
    // reference
    CK(hipMemsetAsync(C[0], 0, cbytes, st));
    CK(hipMemsetAsync(cs0, 0, c.M * 4, st));
    void* auxw = c.epi == GOAT_EPI_GELU ? aux2 : aux;
    int rc = run_gemm(st, c, c.base, A[0], B[0], C[0], use_bias, auxw, c.ta ? cs0 : nullptr);
    CK(hipStreamSynchronize(st));
    if (rc) { printf("%-44s baseline %s rc %d\n", tag, c.base.name, rc); continue; }
    CK(hipMemcpy(ref.data(), C[0], cbytes, hipMemcpyDeviceToHost));
    if (c.epi == GOAT_EPI_GELU) CK(hipMemcpy(refaux.data(), aux2, nc * 2, hipMemcpyDeviceToHost));
    std::vector<float> hcs0(c.M), hcs1(c.M);
    CK(hipMemcpy(hcs0.data(), cs0, c.M * 4, hipMemcpyDeviceToHost));
    const double flop = 2.0 * c.M * c.N * c.K;
    const double tb = time_conf(st, c, c.base, A, B, C, use_bias, auxw, ROT);
    printf("%-44s %-18s %9.2f us %7.1f TF\n", tag, c.base.name, tb, flop / tb * 1e-6);
    for (const Conf& cf : c.cand) {
      CK(hipMemsetAsync(C[1], c.split > 1 ? 0 : 0xFF, cbytes, st));
      CK(hipMemsetAsync(cs1, 0, c.M * 4, st));
      if (c.epi == GOAT_EPI_GELU) CK(hipMemsetAsync(aux2, 0xFF, nc * 2, st));
      rc = run_gemm(st, c, cf, A[0], B[0], C[1], use_bias, auxw, c.ta ? cs1 : nullptr);
      hipError_t se = hipStreamSynchronize(st);
      if (rc || se != hipSuccess) { printf("%-44s %-18s rc %d sync %d\n", "", cf.name, rc, (int)se); if (se != hipSuccess) return 3; continue; }
      CK(hipMemcpy(got.data(), C[1], cbytes, hipMemcpyDeviceToHost));
      size_t nd = 0;
      double maxd = 0;
      for (size_t i = 0; i < nc; ++i) {
        float a, b;
        if (c.f32) { a = reinterpret_cast<float*>(ref.data())[i]; b = reinterpret_cast<float*>(got.data())[i]; }
        else { a = bf2f(reinterpret_cast<uint16_t*>(ref.data())[i]); b = bf2f(reinterpret_cast<uint16_t*>(got.data())[i]); }
        const double d = fabs((double)a - (double)b);
        if (!(d <= (c.split > 1 ? 1e-3 * (1 + fabs(a)) : 0.0))) { ++nd; }
        if (d > maxd || d != d) maxd = d;
      }
      size_t nda = 0;
      if (c.epi == GOAT_EPI_GELU) {
        CK(hipMemcpy(gotaux.data(), aux2, nc * 2, hipMemcpyDeviceToHost));
        nda = memcmp(gotaux.data(), refaux.data(), nc * 2) ? 1 : 0;
      }
      double csd = 0;
      if (c.ta) {
        CK(hipMemcpy(hcs1.data(), cs1, c.M * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < c.M; ++i) csd = fmax(csd, fabs(hcs0[i] - hcs1[i]) / (1 + fabs(hcs0[i])));
      }
      const bool ok = nd == 0 && nda == 0 && csd < 1e-3;
      if (!ok) ++nbad;
      const double tc = time_conf(st, c, cf, A, B, C, use_bias, auxw, ROT);
      printf("%-44s %-18s %9.2f us %7.1f TF  x%.3f  %s (mismatch %zu of %zu, max %.3g, aux %zu, colsum %.2g)\n", "", cf.name, tc, flop / tc * 1e-6, tb / tc,
             ok ? "PARITY-OK" : "PARITY-FAIL", nd, nc, maxd, nda, csd);
      fflush(stdout);
    }
    for (int k = 0; k < rot; ++k) { CK(hipFree(A[k])); CK(hipFree(B[k])); CK(hipFree(C[k])); }
    CK(hipFree(aux)); CK(hipFree(aux2)); CK(hipFree(bias)); CK(hipFree(cs0)); CK(hipFree(cs1));
  }
  printf("parity failures: %d\n", nbad);
  return nbad ? 1 : 0;
}
