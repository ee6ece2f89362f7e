// What is the fixed cost of one short-K GEMM launch made of?  (round 4: the K-sweep of round 1 put ~9.5 us of a 13-30 us launch outside the
// main loop.)  Stand-alone probe, no torch:
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/launch_floor scripts/launch_floor.cpp -Iinclude -Lvln-goat_amd/csrc -lgoat_hip
//   LD_LIBRARY_PATH=vln-goat_amd/csrc gpurun_out/launch_floor
// Every line is the time per launch of NL dependent launches on one stream, captured in a hipGraph (no host launch cost) and replayed,
// operands rotating through ROT buffer sets (cold in L2, and beyond the Infinity Cache for ROT * set > 256 MB):
//   empty      : grid x 512 threads with the GEMM's LDS allocation, no work                    -> dispatch + kernel boundary
//   store      : every workgroup writes its BM x BN bf16 tile (16-byte stores)                 -> + output write and drain
//   load       : every workgroup reads its A rows and B rows once (16-byte loads), K = 768       -> + cold operand fetch
//   load+store : both
//   gemm K=... : goat_gemm_bf16 on the same shape with the contraction cut to K
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

__global__ __launch_bounds__(512) void k_empty(int* p) {
  if (p == (int*)1) *p = 0;
}

// tile (tm, tn) of a [M, N] bf16 matrix, BM x BN, written as 16-byte pieces
__global__ __launch_bounds__(512) void k_store(uint16_t* C, int M, int N, int BM, int BN, int tiles_n) {
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int cpr = BN / 8;
  for (int i = threadIdx.x; i < BM * cpr; i += 512) {
    const int r = tm * BM + i / cpr, c = tn * BN + (i % cpr) * 8;
    if (r < M && c < N) *reinterpret_cast<uint4*>(C + (size_t)r * N + c) = make_uint4(i, i, i, i);
  }
}

__global__ __launch_bounds__(512) void k_load(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K, int BM, int BN, int tiles_n,
                                              int do_store) {
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int kp = K / 8;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int i = threadIdx.x; i < BM * kp; i += 512) {
    const int r = tm * BM + i / kp;
    if (r < M) {
      const uint4 v = *reinterpret_cast<const uint4*>(A + (size_t)r * K + (i % kp) * 8);
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  }
  for (int i = threadIdx.x; i < BN * kp; i += 512) {
    const int r = tn * BN + i / kp;
    if (r < N) {
      const uint4 v = *reinterpret_cast<const uint4*>(B + (size_t)r * K + (i % kp) * 8);
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  }
  if (do_store) {
    const int cpr = BN / 8;
    for (int i = threadIdx.x; i < BM * cpr; i += 512) {
      const int r = tm * BM + i / cpr, c = tn * BN + (i % cpr) * 8;
      if (r < M && c < N) *reinterpret_cast<uint4*>(C + (size_t)r * N + c) = acc;
    }
  } else if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) {
    C[0] = 1;
  }
}

template <class F>
static double graph_time(hipStream_t st, int NL, F&& launch) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < NL; ++i) launch(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms * 1e3 / NL < best) best = ms * 1e3 / NL;
  }
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return best;
}

// GOAT_FLOOR_RANDOM=1: operands are N(0,1)-like random bf16 instead of the constant 0x1111 (DVFS: constant operands draw less
// power and clock higher -- MI355X_MICROARCH.md "DVFS give-back"; the random numbers are what a training step sees)
static void fill(uint16_t* dst, size_t n, bool rnd) {
  if (!rnd) { CK(hipMemset(dst, 0x11, n * 2)); return; }
  static std::vector<uint16_t> host;
  if (host.size() < n) {
    size_t old = host.size();
    host.resize(n);
    uint32_t s = 12345u + (uint32_t)old;
    for (size_t i = old; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      const float f = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 22);      // uniform in [-2, 2)
      uint32_t b; memcpy(&b, &f, 4);
      host[i] = (uint16_t)(b >> 16);
    }
  }
  CK(hipMemcpy(dst, host.data(), n * 2, hipMemcpyHostToDevice));
}

int main(int argc, char** argv) {
  const bool rnd = getenv("GOAT_FLOOR_RANDOM") != nullptr;
  printf("operands: %s\n", rnd ? "random bf16 in [-2, 2)" : "constant 0x1111");
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  struct Shape { int M, N, K, bm, bn, ns; const char* what; };
  const Shape shapes[] = {
      {3840, 768, 768, 96, 128, 4, "out-proj 3840x768x768, tile 96x128 s4"},
      {3840, 3072, 768, 192, 256, 2 | GOAT_GEMM_PP, "FFN-up 3840x3072x768, tile 192x256 pp"},
      {3840, 2304, 768, 192, 256, 2 | GOAT_GEMM_PP, "QKV 3840x2304x768, tile 192x256 pp"},
      {3840, 768, 3072, 96, 128, 4, "FFN-down 3840x768x3072, tile 96x128 s4"},
      {1056, 768, 768, 64, 128, 4, "gmap 1056x768x768, tile 64x128 s4"},
  };
  const int NL = 96;
  for (const Shape& s : shapes) {
    const size_t set = ((size_t)s.M * s.K + (size_t)s.N * s.K + (size_t)s.M * s.N) * 2;
    int ROT = (int)(400e6 / set) + 1;
    if (ROT > NL) ROT = NL;
    std::vector<uint16_t*> A(ROT), B(ROT), C(ROT);
    for (int i = 0; i < ROT; ++i) {
      CK(hipMalloc(&A[i], (size_t)s.M * s.K * 2));
      CK(hipMalloc(&B[i], (size_t)s.N * s.K * 2));
      CK(hipMalloc(&C[i], (size_t)s.M * s.N * 2));
      fill(A[i], (size_t)s.M * s.K, rnd);
      fill(B[i], (size_t)s.N * s.K, rnd);
    }
    const int tiles_m = (s.M + s.bm - 1) / s.bm, tiles_n = (s.N + s.bn - 1) / s.bn, grid = tiles_m * tiles_n;
    const bool pp = (s.ns & GOAT_GEMM_PP) != 0;
    const int threads = (pp || s.bm >= 256) ? 512 : 256;
    const int lds = pp ? (s.bm * 64 * 2 * 2 + s.bn * 64 * 2 * 2) : (s.bm + s.bn) * 128 * (s.ns & 7);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("%s: %d workgroups x %d threads, %d KiB LDS, rotation %d sets (%.0f MB)\n", s.what, grid, threads, lds / 1024, ROT, ROT * set / 1e6);
    double t;
    t = graph_time(st, NL, [&](int i) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(threads), lds, st, (int*)nullptr); });
    printf("  %-22s %7.2f us\n", "empty", t);
    t = graph_time(st, NL, [&](int i) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(threads), 0, st, (int*)nullptr); });
    printf("  %-22s %7.2f us\n", "empty, no LDS", t);
    t = graph_time(st, NL, [&](int i) { hipLaunchKernelGGL(k_store, dim3(grid), dim3(512), 0, st, C[i % ROT], s.M, s.N, s.bm, s.bn, tiles_n); });
    printf("  %-22s %7.2f us   (%.1f MB)\n", "store", t, s.M * (double)s.N * 2 / 1e6);
    t = graph_time(st, NL, [&](int i) {
      hipLaunchKernelGGL(k_load, dim3(grid), dim3(512), 0, st, A[i % ROT], B[i % ROT], C[i % ROT], s.M, s.N, s.K, s.bm, s.bn, tiles_n, 0);
    });
    printf("  %-22s %7.2f us   (%.1f MB unique, %.1f MB requested)\n", "load", t, ((double)s.M + s.N) * s.K * 2 / 1e6,
           ((double)tiles_n * s.M + (double)tiles_m * s.N) * s.K * 2 / 1e6);
    t = graph_time(st, NL, [&](int i) {
      hipLaunchKernelGGL(k_load, dim3(grid), dim3(512), 0, st, A[i % ROT], B[i % ROT], C[i % ROT], s.M, s.N, s.K, s.bm, s.bn, tiles_n, 1);
    });
    printf("  %-22s %7.2f us\n", "load + store", t);
    for (int K : {64, 128, 256, 512, 768, 1536, 3072}) {
      if (K > s.K) break;
      int rc = 0;
      t = graph_time(st, NL, [&](int i) {
        rc |= goat_gemm_bf16(st, 0, 0, GOAT_BF16, A[i % ROT], s.K, B[i % ROT], s.K, C[i % ROT], s.N, s.M, s.N, K, nullptr, GOAT_EPI_NONE, nullptr, 0,
                             1, s.bm | (s.bn << 16), s.ns, nullptr);
      });
      printf("  gemm K=%-15d %7.2f us   (%.0f TFLOP/s)%s\n", K, t, 2.0 * s.M * s.N * K / t / 1e6, rc ? "  [launch error]" : "");
    }
    // epilogues of the FFN pair (cold operands): bias, bias + GELU with the pre-activation saved (aux store), x GELU'(aux) (aux load)
    if (s.N == 3072 || s.N == 2304) {
      float* bias;
      CK(hipMalloc(&bias, s.N * 4));
      CK(hipMemset(bias, 0, s.N * 4));
      std::vector<uint16_t*> X(ROT);
      for (int i = 0; i < ROT; ++i) { CK(hipMalloc(&X[i], (size_t)s.M * s.N * 2)); fill(X[i], (size_t)s.M * s.N, rnd); }
      struct E { int epi; bool bias, aux; const char* name; };
      const E es[] = {{GOAT_EPI_NONE, true, false, "bias"}, {GOAT_EPI_GELU, true, true, "bias+GELU, aux store"}, {GOAT_EPI_GELU, true, false, "bias+GELU, no aux"},
                      {GOAT_EPI_MUL_DGELU, false, true, "x GELU'(aux)"}};
      for (const E& e : es) {
        int rc = 0;
        t = graph_time(st, NL, [&](int i) {
          rc |= goat_gemm_bf16(st, 0, 0, GOAT_BF16, A[i % ROT], s.K, B[i % ROT], s.K, C[i % ROT], s.N, s.M, s.N, s.K, e.bias ? bias : nullptr, e.epi,
                               e.aux ? X[i % ROT] : nullptr, s.N, 1, s.bm | (s.bn << 16), s.ns, nullptr);
        });
        printf("  gemm K=%d %-22s %7.2f us   (%.0f TFLOP/s)%s\n", s.K, e.name, t, 2.0 * s.M * s.N * s.K / t / 1e6, rc ? "  [launch error]" : "");
      }
      for (int i = 0; i < ROT; ++i) CK(hipFree(X[i]));
      CK(hipFree(bias));
    }
    // the same GEMM with warm operands (one buffer set)
    {
      int rc = 0;
      t = graph_time(st, NL, [&](int i) {
        rc |= goat_gemm_bf16(st, 0, 0, GOAT_BF16, A[0], s.K, B[0], s.K, C[0], s.N, s.M, s.N, s.K, nullptr, GOAT_EPI_NONE, nullptr, 0, 1,
                             s.bm | (s.bn << 16), s.ns, nullptr);
      });
      printf("  gemm K=%d warm        %7.2f us   (%.0f TFLOP/s)%s\n", s.K, t, 2.0 * s.M * s.N * s.K / t / 1e6, rc ? "  [launch error]" : "");
    }
    for (int i = 0; i < ROT; ++i) {
      CK(hipFree(A[i]));
      CK(hipFree(B[i]));
      CK(hipFree(C[i]));
    }
  }
  return 0;
}
