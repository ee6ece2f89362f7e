// Small glue kernels of the captured steps (round 4): what used to be torch elementwise launches inside the step graph.
//   goat_add_n       : out = sum of up to 8 tensors, float32 accumulation, one rounding — the gradient fan-in of a tensor with several
//                      consumers (hipops.fanout): one launch instead of the autograd engine's k - 1 pairwise `add` kernels
//   goat_zero_ranges : up to 16 byte ranges cleared by one launch (the gradient arena's per-step fills: dp.GradArena.zero)
#include "common.hpp"

namespace {

struct AddNArgs { const void* src[8]; int n; };

template <typename T>
__global__ __launch_bounds__(256) void add_n_kernel(AddNArgs a, T* __restrict__ out, int64_t nchunks, int64_t numel) {
  constexpr int EPC = DT<T>::EPC;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * 256) {
    const int64_t base = c * EPC;
    if (base + EPC <= numel) {
      Chunk<T> acc, t;
      acc.load(reinterpret_cast<const T*>(a.src[0]) + base);
#pragma unroll
      for (int i = 1; i < 8; ++i) {
        if (i < a.n) {
          t.load(reinterpret_cast<const T*>(a.src[i]) + base);
#pragma unroll
          for (int e = 0; e < EPC; ++e) acc.v[e] += t.v[e];
        }
      }
      acc.store_stream(out + base);
    } else {
      for (int64_t j = base; j < numel; ++j) {
        float s = 0.f;
        for (int i = 0; i < a.n; ++i) s += to_f(reinterpret_cast<const T*>(a.src[i])[j]);
        out[j] = from_f<T>(s);
      }
    }
  }
}

struct ZeroArgs { void* ptr[16]; int64_t end16[16]; int n; };      // end16[i] = running total of 16-byte chunks up to and including range i

__global__ __launch_bounds__(256) void zero_ranges_kernel(ZeroArgs a) {
  const int64_t total = a.end16[a.n - 1];
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (int64_t)gridDim.x * 256) {
    int r = 0;
#pragma unroll
    for (int i = 0; i < 15; ++i) r += (i + 1 < a.n && c >= a.end16[i]) ? 1 : 0;
    const int64_t off = c - (r ? a.end16[r - 1] : 0);
    reinterpret_cast<f32x4*>(a.ptr[r])[off] = z;
  }
}

}  // namespace

extern "C" int goat_add_n(void* stream, int dtype, const void* const* srcs, int n, void* out, int64_t numel) {
  if (!srcs || !out || n < 1 || n > 8) return GOAT_E_ARG;
  if (numel <= 0) return GOAT_E_SHAPE;
  AddNArgs a;
  a.n = n;
  for (int i = 0; i < 8; ++i) {
    a.src[i] = i < n ? srcs[i] : nullptr;
    if (i < n && (!srcs[i] || (reinterpret_cast<uintptr_t>(srcs[i]) & 15))) return GOAT_E_ARG;
  }
  if (reinterpret_cast<uintptr_t>(out) & 15) return GOAT_E_ARG;
  const int epc = dtype == GOAT_BF16 ? 8 : 4;
  const int64_t nchunks = (numel + epc - 1) / epc;
  int64_t blocks = (nchunks + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(add_n_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, st, a, (bf16_t*)out, nchunks, numel);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(add_n_kernel<float>, dim3((int)blocks), dim3(256), 0, st, a, (float*)out, nchunks, numel);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_zero_ranges(void* stream, void* const* ptrs, const int64_t* nbytes, int n) {
  if (!ptrs || !nbytes || n < 1 || n > 16) return GOAT_E_ARG;
  ZeroArgs a;
  a.n = n;
  int64_t tot = 0;
  for (int i = 0; i < 16; ++i) {
    if (i < n) {
      if (!ptrs[i] || (reinterpret_cast<uintptr_t>(ptrs[i]) & 15) || nbytes[i] <= 0 || (nbytes[i] & 15)) return GOAT_E_ARG;
      tot += nbytes[i] / 16;
    }
    a.ptr[i] = i < n ? ptrs[i] : nullptr;
    a.end16[i] = tot;
  }
  int64_t blocks = (tot + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_ranges_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  GOAT_LAUNCH_CHECK();
  return 0;
}
