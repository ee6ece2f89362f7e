// Micro-benchmark: how fast can one CU pull L2-resident bytes (a) into LDS with buffer_load ... lds (LDS-DMA) and
// (b) into VGPRs with buffer_load_dwordx4?  The GEMM tile shape is chosen from this number (bytes per MFMA cycle).
//   build:  hipcc --offload-arch=gfx950 -O3 -o scripts/l2_lds_bw scripts/l2_lds_bw.hip
//   run:    scripts/l2_lds_bw            (prints a table; every WG reads the same window, so all XCD L2s serve hits)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// mode 0: LDS-DMA.  Each wave issues PER 1-KiB pieces per iteration, keeps up to 2*PER in flight.
// rows = 1: 1 KiB contiguous per wave-instruction;  rows = 8: 8 rows x 128 B with a row stride of `ld` bytes (GEMM K-tile shape)
template <int PER, int MODE>
__global__ __launch_bounds__(512) void bw_kernel(const char* src, uint32_t window, uint32_t ld, int rows8, int iters, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)window, 0x00020000);
  // per-lane offset inside one piece
  uint32_t lo = rows8 ? (uint32_t)((lane >> 3) * ld + (lane & 7) * 16) : (uint32_t)(lane * 16);
  const uint32_t piece_span = rows8 ? 8 * ld : 1024;       // bytes of address space one piece covers
  uint32_t base = ((blockIdx.x * 7 + wave) * PER) * piece_span;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      uint32_t off = (base + j * piece_span) % (window - piece_span - 1024);
      off &= ~15u;
      if (MODE == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(smem + ((wave * 2 * PER + (it & 1) * PER + j) * 1024)), 16, lo + off, 0, 0, 0);
      } else {
        u32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(lo + off), "s"(r) : "memory");
        asm volatile("" ::"v"(v));
        acc ^= 1;
      }
    }
    base += nw * PER * piece_span * 3;
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <int PER, int MODE>
static double run(const char* d, uint32_t window, uint32_t ld, int rows8, int nthreads, int wgs, int iters, uint32_t* sink) {
  size_t smem = MODE == 0 ? (size_t)(nthreads / 64) * 2 * PER * 1024 : 0;
  auto k = bw_kernel<PER, MODE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(wgs), dim3(nthreads), smem, 0, d, window, ld, rows8, iters, sink);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(wgs), dim3(nthreads), smem, 0, d, window, ld, rows8, iters, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  double bytes = (double)wgs * (nthreads / 64) * iters * PER * 1024.0 * reps;
  return bytes / (ms * 1e-3);   // B/s
}

int main() {
  const uint32_t maxwin = 64u << 20;
  char* d; uint32_t* sink;
  CK(hipMalloc(&d, maxwin)); CK(hipMemset(d, 1, maxwin)); CK(hipMalloc(&sink, 64));
  const int iters = 400;
  printf("%-6s %-5s %-6s %-7s %-5s %-4s | %10s %12s %10s\n", "mode", "rows", "window", "threads", "wg/cu", "per", "TB/s chip", "GB/s per CU", "B/clk/CU");
  const uint32_t windows[] = {1u << 20, 2u << 20, 24u << 20};
  for (int mode = 0; mode < 2; ++mode)
    for (int rows8 = 0; rows8 < 2; ++rows8)
      for (uint32_t win : windows)
        for (int nthreads : {256, 512})
          for (int wgpc : {1, 2}) {
            for (int per : {2, 4, 8}) {
              if (mode == 0 && (size_t)(nthreads / 64) * 2 * per * 1024 * wgpc > 160 * 1024) continue;
              double bps;
#define RUN(P) bps = mode == 0 ? run<P, 0>(d, win, 1536, rows8, nthreads, 256 * wgpc, iters, sink) : run<P, 1>(d, win, 1536, rows8, nthreads, 256 * wgpc, iters, sink)
              if (per == 2) { RUN(2); } else if (per == 4) { RUN(4); } else { RUN(8); }
              printf("%-6s %-5s %4uMB %-7d %-5d %-4d | %10.2f %12.1f %10.1f\n", mode == 0 ? "ldsdma" : "vgpr", rows8 ? "8x128" : "1KiB", win >> 20,
                     nthreads, wgpc, per, bps / 1e12, bps / 256 / 1e9, bps / 256 / 2.4e9);
            }
          }
  // ---- round 5: the weight stream of a FULL-ROW fused kernel (BertSelfOutput = dense 768 -> 768 + dropout + residual +
  // LayerNorm in one kernel needs whole output rows per workgroup, i.e. every workgroup streams the WHOLE 768 x 768 bf16 weight,
  // 1.18 MB, through its LDS).  M = 3840 rows as 16- / 32- / 64-row tiles = 240 / 120 / 60 workgroups of 8 waves; each moves 1.18 MB
  // (19 iterations x 8 waves x 8 KiB) out of a 1.18 MB window that stays L2-resident.  Time per launch = the floor of such a kernel
  // before its first MFMA, its activation loads, its epilogue and the LayerNorm: compare with goat_gemm_bf16 (9.8 us) +
  // goat_ln_fwd (6.9 us) + one kernel boundary (1.7 us) = 18.4 us for the pair it would replace.
  printf("\nfull-row weight stream: 1.18 MB per workgroup from an L2-resident 1.18 MB window, 512 threads, 8 pieces per wave in flight\n");
  for (int wgs : {60, 120, 240}) {
    const uint32_t win = 768u * 1536u + 2048u;
    const int it = 19;
    double bps = run<8, 0>(d, win, 1536, 1, 512, wgs, it, sink);
    const double bytes = (double)wgs * it * 8 * 8 * 1024;
    printf("  %3d workgroups (%2d-row tiles): %6.2f us per launch   (%.1f TB/s over the chip, %.1f B/clk per active CU)\n", wgs, 3840 / wgs,
           bytes / bps * 1e6, bps / 1e12, bps / wgs / 2.4e9);
  }
  return 0;
}
