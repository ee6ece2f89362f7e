// Shared device helpers for the gfx950 kernels of libgoat_hip.so (wave64, MFMA 32x32 fragments).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/goat_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

#define GOAT_LAUNCH_CHECK()                          \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int id = GOAT_F32;
  static constexpr int EPC = 4;  // elements per 16-byte chunk
};
template <> struct DT<bf16_t> {
  static constexpr int id = GOAT_BF16;
  static constexpr int EPC = 8;
};

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte result store of the HBM-bound row kernels.  GOAT_ROW_NT (default 1): non-temporal — a kernel that leaves its output
// dirty in the XCD L2s pays the write-back as a burst at its end (measured on the GEMM epilogue: 25.9 -> 21.5 us for 23.6 MB);
// streamed stores leave the L2 while the kernel still runs.  The consumer is another kernel on other CUs / XCDs anyway.
#ifndef GOAT_ROW_NT
#define GOAT_ROW_NT 1
#endif
__device__ __forceinline__ void goat_store_stream(f32x4* p, const f32x4& v) {
#if GOAT_ROW_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// ---- 16-byte chunk load/store as floats -------------------------------------------------------
template <typename T> struct Chunk;  // EPC values as float
template <> struct Chunk<float> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ __forceinline__ void store(float* p) const {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
  __device__ __forceinline__ void store_stream(float* p) const {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    goat_store_stream(reinterpret_cast<f32x4*>(p), t);
  }
};
template <> struct Chunk<bf16_t> {
  float v[8];
  __device__ __forceinline__ void load(const bf16_t* p) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  __device__ __forceinline__ void store(bf16_t* p) const {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
  __device__ __forceinline__ void store_stream(bf16_t* p) const {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    goat_store_stream(reinterpret_cast<f32x4*>(p), *reinterpret_cast<f32x4*>(&t));
  }
};

// ---- counter-based dropout RNG -----------------------------------------------------------------
// keep(i) for flat element index i under (seed, offset).  Stateless, so the backward pass regenerates the mask.
// The 64-bit seed is hashed ONCE per thread into two 32-bit keys; per element the work is a keyed 32-bit
// multiply-xorshift bijection of the PAIR index i>>1 (two quarter-rate v_mul_lo_u32 + full-rate ops) whose two 16-bit
// halves decide elements 2q and 2q+1 (drop if half < p*65536).  The first version hashed every element with two 64-bit
// multiplies (~8 quarter-rate ops): LayerNorm / dropout / attention kernels were VALU-bound on the mask, not on HBM.
struct GoatRng {
  uint32_t k0, k1;
  __device__ __forceinline__ explicit GoatRng(uint64_t seed) {
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    k0 = (uint32_t)x; k1 = (uint32_t)(x >> 32);
  }
  // 32 random bits for the counter pair {2q, 2q+1}
  __device__ __forceinline__ uint32_t pair_bits(uint64_t q) const {
    uint32_t x = ((uint32_t)q ^ k0) + __umul24((uint32_t)(q >> 32), 0x9E3779u);
    x ^= x >> 16; x *= 0x7FEB352Du;
    x += k1;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
  }
  __device__ __forceinline__ bool keep(uint64_t ctr, uint32_t thr16) const {
    const uint32_t h = pair_bits(ctr >> 1);
    return ((ctr & 1) ? (h >> 16) : (h & 0xFFFFu)) >= thr16;
  }
  // bit e of the result = keep(ctr0 + e), e < N (N even): N/2 hashes when ctr0 is even (offsets are handed out in
  // multiples of 8 and rows are multiples of the chunk width, so that is the path taken)
  template <int N>
  __device__ __forceinline__ uint32_t keep_bits(uint64_t ctr0, uint32_t thr16) const {
    uint32_t m = 0;
    if ((ctr0 & 1) == 0) {
      const uint64_t q0 = ctr0 >> 1;
#pragma unroll
      for (int j = 0; j < N / 2; ++j) {
        const uint32_t h = pair_bits(q0 + j);
        m |= ((h & 0xFFFFu) >= thr16 ? 1u : 0u) << (2 * j);
        m |= ((h >> 16) >= thr16 ? 1u : 0u) << (2 * j + 1);
      }
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) m |= (keep(ctr0 + e, thr16) ? 1u : 0u) << e;
    }
    return m;
  }
};
// Dropout bits of one (sample, head): a 32-bit key per block (from the 64-bit seed / offset / device counter of common.hpp's
// GoatRng, mixed once), then ONE 32-bit multiply-xorshift hash per PAIR of consecutive probability indices
// idx = q * Lk + key; its 16-bit halves decide the two elements.  The first version hashed every element through
// GoatRng::keep with 64-bit counters: 6 700 of 24 700 cycles of a forward block.  Forward and backward regenerate the same bits.
struct HeadRng {
  uint32_t k;
  __device__ __forceinline__ HeadRng(uint64_t seed, uint64_t offset, uint32_t bh) {
    uint64_t x = (seed + 0x9E3779B97F4A7C15ull * (offset + 1)) ^ ((uint64_t)bh * 0xD6E8FEB86659FD93ull);
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    k = (uint32_t)x;
  }
  __device__ __forceinline__ uint32_t pair(uint32_t pair_idx) const {
    uint32_t x = pair_idx ^ k;
    x *= 0x7FEB352Du; x ^= x >> 15;
    x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
  }
  __device__ __forceinline__ bool keep(uint32_t idx, uint32_t thr16) const {
    const uint32_t h = pair(idx >> 1);
    return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= thr16;
  }
  // bit e = keep(idx0 + e), e < 4
  __device__ __forceinline__ uint32_t keep4(uint32_t idx0, uint32_t thr16) const {
    uint32_t m = 0;
    if ((idx0 & 1u) == 0) {
      const uint32_t h0 = pair(idx0 >> 1), h1 = pair((idx0 >> 1) + 1);
      m |= ((h0 & 0xFFFFu) >= thr16 ? 1u : 0u) | ((h0 >> 16) >= thr16 ? 2u : 0u);
      m |= ((h1 & 0xFFFFu) >= thr16 ? 4u : 0u) | ((h1 >> 16) >= thr16 ? 8u : 0u);
    } else {
      const uint32_t h0 = pair(idx0 >> 1), h1 = pair((idx0 >> 1) + 1), h2 = pair((idx0 >> 1) + 2);
      m |= ((h0 >> 16) >= thr16 ? 1u : 0u) | ((h1 & 0xFFFFu) >= thr16 ? 2u : 0u);
      m |= ((h1 >> 16) >= thr16 ? 4u : 0u) | ((h2 & 0xFFFFu) >= thr16 ? 8u : 0u);
    }
    return m;
  }
};

__host__ __device__ __forceinline__ uint32_t goat_thr16(float p) {
  return (uint32_t)(p * 65536.0f + 0.5f);
}

// ---- wave64 reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// reduce across the 32 lanes of each half-wave (lanes sharing lane>>5)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- MFMA 32x32 tile step on a pair of 16-byte operand chunks --------------------------------------
// Lane l holds, for A: row (l&31) of the 32-row tile, the 16-byte k-chunk number (l>>5) of a 32-byte
// k-step; same for B.  bf16: one v_mfma_f32_32x32x16_bf16.  f32: the chunk is 4 floats and the k-order
// inside the step is permuted identically for A and B (a dot product is order-free), so four
// v_mfma_f32_32x32x2_f32 consume it exactly.
// C/D layout (both): col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16).
__device__ __forceinline__ void mma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const f32x4& a, const f32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
}
template <typename T> struct FragT;
template <> struct FragT<float> { typedef f32x4 type; };
template <> struct FragT<bf16_t> { typedef bf16x8 type; };

__device__ __forceinline__ int c_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): one rcp + one exp instead of libm's erff, which
// cost ~25 us per FFN GEMM epilogue.  erf-GELU as in P/model/Bert_backbone.py:41-47.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float y = 1.0f - poly * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  return 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// bf16-grade erf-GELU for the GEMM epilogues whose result is rounded to bf16 anyway (GOAT_G2_FASTGELU, default on).
// gelu(x) = x * (0.5 + s(xc)),  gelu'(x) = 0.5 + g(xc),  xc = clamp(x, -4, 4),  s, g odd polynomials of degree 15 (minimax fits
// of Phi(x) - 0.5 and Phi(x) - 0.5 + x phi(x) on [0, 4]; |error| <= 4.4e-4 resp. 2.7e-4 absolute, i.e. 1/10 of a bf16
// ulp at 1): 11 full-rate VALU operations instead of ~26 issue slots for the rcp + exp form above — the epilogue of a
// 192x256 tile evaluates 49 152 of them with nothing else to hide behind.  The float32 parity path keeps gelu_f / dgelu_f.
// minimax coefficients (tests/test_gelu_fit.py re-derives the error bounds)
#define GOAT_GELU_C0 3.986733939e-01f
#define GOAT_GELU_C1 -6.588784139e-02f
#define GOAT_GELU_C2 9.505397558e-03f
#define GOAT_GELU_C3 -1.006490218e-03f
#define GOAT_GELU_C4 7.485502142e-05f
#define GOAT_GELU_C5 -3.657131012e-06f
#define GOAT_GELU_C6 1.041959709e-07f
#define GOAT_GELU_C7 -1.301293275e-09f
#define GOAT_DGELU_C0 7.967216258e-01f
#define GOAT_DGELU_C1 -2.620298365e-01f
#define GOAT_DGELU_C2 5.591482115e-02f
#define GOAT_DGELU_C3 -7.687439989e-03f
#define GOAT_DGELU_C4 6.876451109e-04f
#define GOAT_DGELU_C5 -3.845953930e-05f
#define GOAT_DGELU_C6 1.213804624e-06f
#define GOAT_DGELU_C7 -1.641975819e-08f
#ifndef GOAT_G2_FASTGELU
#define GOAT_G2_FASTGELU 1
#endif
__device__ __forceinline__ float gelu_fast(float x) {
#if GOAT_G2_FASTGELU
  const float xc = fminf(fmaxf(x, -4.0f), 4.0f), t = xc * xc;
  float p = GOAT_GELU_C7;
  p = fmaf(p, t, GOAT_GELU_C6); p = fmaf(p, t, GOAT_GELU_C5); p = fmaf(p, t, GOAT_GELU_C4); p = fmaf(p, t, GOAT_GELU_C3);
  p = fmaf(p, t, GOAT_GELU_C2); p = fmaf(p, t, GOAT_GELU_C1); p = fmaf(p, t, GOAT_GELU_C0);
  return x * fmaf(xc, p, 0.5f);
#else
  return gelu_f(x);
#endif
}
__device__ __forceinline__ float dgelu_fast(float x) {
#if GOAT_G2_FASTGELU
  const float xc = fminf(fmaxf(x, -4.0f), 4.0f), t = xc * xc;
  float p = GOAT_DGELU_C7;
  p = fmaf(p, t, GOAT_DGELU_C6); p = fmaf(p, t, GOAT_DGELU_C5); p = fmaf(p, t, GOAT_DGELU_C4); p = fmaf(p, t, GOAT_DGELU_C3);
  p = fmaf(p, t, GOAT_DGELU_C2); p = fmaf(p, t, GOAT_DGELU_C1); p = fmaf(p, t, GOAT_DGELU_C0);
  return fmaf(xc, p, 0.5f);
#else
  return dgelu_f(x);
#endif
}
