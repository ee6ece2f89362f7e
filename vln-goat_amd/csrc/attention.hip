// goat_attn_fwd / goat_attn_bwd: masked multi-head attention (head_dim 64) on MFMA 32x32 tiles.
//
// GOAT's sequences are tiny (36 views, <=80..200 tokens, <=~60 map nodes), so instead of flash-style
// streaming one workgroup owns a whole (sample, head): forward = one wave per 32-query tile with the full
// score row-block held in accumulators (S^T = K·Q^T, so a softmax row is lane-local: keys live in the
// accumulator registers, queries across lanes); V (forward) / K, Q_i, dO_i (backward) are staged row-major
// in LDS and the "transposed" MFMA operands are gathered from there, exploiting that the k-order inside an
// MFMA step is free as long as A and B agree.  Dropout on the probabilities uses the stateless counter hash
// of common.hpp and is regenerated in the backward pass.
#include <cstdlib>
#include "attn_args.hpp"

namespace {

constexpr int HD = 64;  // head dim

template <typename T> struct AT {
  typedef typename FragT<T>::type Frag;
  static constexpr int NE = DT<T>::EPC;                 // elements per 16-B chunk
  static constexpr int ROWB = HD * (int)sizeof(T);      // bytes per head row
  static constexpr int KSTEPS = ROWB / 32;              // 32-byte k-steps over d (bf16 4, f32 8)
  static constexpr int LSTR = HD + NE;                  // LDS row stride (elements) of [row][64] tiles
  static constexpr int TSTEPS = 32 / (2 * NE);          // k-steps over a 32-wide tile (bf16 2, f32 4)
  static constexpr int PSTR = 32 + NE;                  // LDS row stride of [32][32] tiles
};

template <typename T>
__device__ __forceinline__ typename AT<T>::Frag zero_frag() {
  typename AT<T>::Frag f;
#pragma unroll
  for (int e = 0; e < AT<T>::NE; ++e) f[e] = (T)0.f;
  return f;
}

// 16-B chunk (ks, hi) of a 64-wide row in global memory
template <typename T>
__device__ __forceinline__ typename AT<T>::Frag gfrag(const T* row_ptr, bool valid, int ks, int hi) {
  typedef typename AT<T>::Frag Frag;
  if (!valid) return zero_frag<T>();
  return *reinterpret_cast<const Frag*>(row_ptr + (ks * 2 + hi) * AT<T>::NE);
}

// gather an MFMA B fragment "fixed column, accumulator-pattern rows" from a row-major LDS tile
template <typename T>
__device__ __forceinline__ typename AT<T>::Frag gather_crow(const T* lds, int stride, int row_base, int step, int col,
                                                             int lane) {
  typename AT<T>::Frag f;
#pragma unroll
  for (int e = 0; e < AT<T>::NE; ++e) f[e] = lds[(row_base + c_row(step * AT<T>::NE + e, lane)) * stride + col];
  return f;
}
// gather "fixed column, rows (2*step+hi)*NE + e"
template <typename T>
__device__ __forceinline__ typename AT<T>::Frag gather_lin(const T* lds, int stride, int step, int hi, int col) {
  typename AT<T>::Frag f;
#pragma unroll
  for (int e = 0; e < AT<T>::NE; ++e) f[e] = lds[((2 * step + hi) * AT<T>::NE + e) * stride + col];
  return f;
}
template <typename T>
__device__ __forceinline__ typename AT<T>::Frag acc_frag(const f32x16& a, int step) {
  typename AT<T>::Frag f;
#pragma unroll
  for (int e = 0; e < AT<T>::NE; ++e) f[e] = from_f<T>(a[step * AT<T>::NE + e]);
  return f;
}

// B fragment "fixed column, accumulator-pattern rows" of a row-major [row][64] LDS tile.
//   f32 : four ds_read_b32 (rows 8*step + 4*hi + e)
//   bf16: two ds_read_b64_tr_b16 (hardware 4x16 transpose): rows 16*step + 4*hi + {0..3} and +8
template <typename T>
__device__ __forceinline__ typename AT<T>::Frag bfrag_crow(const T* lds, int stride, int row_base, int step, int dt, int lane);
template <>
__device__ __forceinline__ f32x4 bfrag_crow<float>(const float* lds, int stride, int row_base, int step, int dt, int lane) {
  return gather_crow<float>(lds, stride, row_base, step, dt * 32 + (lane & 31), lane);
}
template <>
__device__ __forceinline__ bf16x8 bfrag_crow<bf16_t>(const bf16_t* lds, int stride, int row_base, int step, int dt, int lane) {
  typedef __attribute__((address_space(3))) bf16x4 lds_b4;
  const int g = lane >> 4, t15 = lane & 15;
  const int col = dt * 32 + (g & 1) * 16 + (t15 & 3) * 4;
  const int r0 = row_base + 16 * step + 4 * (g >> 1) + (t15 >> 2);
  bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(lds + r0 * stride + col));
  bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(lds + (r0 + 8) * stride + col));
  bf16x8 f;
  f[0] = v0[0]; f[1] = v0[1]; f[2] = v0[2]; f[3] = v0[3];
  f[4] = v1[0]; f[5] = v1[1]; f[6] = v1[2]; f[7] = v1[3];
  return f;
}

// cooperative copy of `nrows` 64-wide rows (global, strided) into LDS [rows_pad][LSTR]; rows >= nrows zero
template <typename T>
__device__ __forceinline__ void stage_rows(T* lds, const T* g, int64_t rs, int row0, int nrows_valid, int rows_pad,
                                           int tid, int nthreads) {
  constexpr int CPR = HD / AT<T>::NE;  // chunks per row
  for (int c = tid; c < rows_pad * CPR; c += nthreads) {
    int r = c / CPR, cc = c % CPR;
    uint4 v = {0u, 0u, 0u, 0u};
    if (row0 + r < nrows_valid) v = *reinterpret_cast<const uint4*>(g + (int64_t)(row0 + r) * rs + cc * AT<T>::NE);
    *reinterpret_cast<uint4*>(lds + r * AT<T>::LSTR + cc * AT<T>::NE) = v;
  }
}

// Dropout decision for probability (b, h, q, key).  bf16: the per-head 32-bit hash of the LDS-staged kernels (attention2.hip's
// HeadRng) — goat_attn_fwd and goat_attn_bwd may be served by different kernel families for one call (the staged backward needs
// more LDS than the staged forward), so both families must draw the same bits.  f32 (always these kernels): the 64-bit counter
// stream of GoatRng, as documented in the header.
template <typename T>
struct AttnMask {
  GoatRng g;
  HeadRng hr;
  uint64_t base;
  __device__ __forceinline__ AttnMask(const AttnArgs& p, int b, int h)
      : g(p.seed + (p.rng_dev ? *p.rng_dev : 0ull)),
        hr(p.seed + (p.rng_dev ? *p.rng_dev : 0ull), p.offset, (uint32_t)(b * p.nh + h)),
        base(p.offset + ((uint64_t)b * p.nh + h) * (uint64_t)p.Lq * (uint64_t)p.Lk) {}
  __device__ __forceinline__ bool keep(const AttnArgs& p, int q, int key, uint32_t thr) const {
    if (sizeof(T) == 2) return hr.keep((uint32_t)q * (uint32_t)p.Lk + (uint32_t)key, thr);
    return g.keep(base + (uint64_t)q * (uint64_t)p.Lk + key, thr);
  }
};

// ======================================================================================== forward
template <typename T, int NKT>
__global__ __launch_bounds__(512) void attn_fwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename AT<T>::Frag Frag;
  constexpr int KSTEPS = AT<T>::KSTEPS, LSTR = AT<T>::LSTR, TSTEPS = AT<T>::TSTEPS;
  T* vl = reinterpret_cast<T*>(smem);  // [NKT*32][LSTR]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
  const int q0 = (blockIdx.y * (blockDim.x >> 6) + wave) * 32;

  const T* Qb = reinterpret_cast<const T*>(p.Q) + b * p.q_bs + h * HD;
  const T* Kb = reinterpret_cast<const T*>(p.K) + b * p.k_bs + h * HD;
  const T* Vb = reinterpret_cast<const T*>(p.V) + b * p.v_bs + h * HD;
  T* Ob = reinterpret_cast<T*>(p.Ow) + b * p.o_bs + h * HD;

  stage_rows<T>(vl, Vb, p.v_rs, 0, p.Lk, NKT * 32, tid, blockDim.x);

  const int q = q0 + l31;
  const bool qv = q < p.Lq;
  Frag qf[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = gfrag<T>(Qb + (int64_t)q * p.q_rs, qv, ks, hi);

  // S^T tiles: rows = keys (accumulator regs), cols = queries (lanes)
  f32x16 s[NKT];
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[jt][r] = 0.f;
    const int key = jt * 32 + l31;
    const bool kv = key < p.Lk;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      Frag kf = gfrag<T>(Kb + (int64_t)key * p.k_rs, kv, ks, hi);
      mma32(s[jt], kf, qf[ks]);
    }
  }

  // scale + masks, row max
  float m = -INFINITY;
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = jt * 32 + c_row(r, lane);
      float v = -INFINITY;
      if (key < p.Lk) {
        v = s[jt][r] * p.scale;
        if (p.kmask) v += p.kmask[(int64_t)b * p.Lk + key];
        if (p.bias && qv) v += p.bias[((int64_t)b * p.Lq + q) * p.Lk + key];
      }
      s[jt][r] = v;
      m = fmaxf(m, v);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
  const float msafe = (m == -INFINITY) ? 0.f : m;
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float e = __expf(s[jt][r] - msafe);
      s[jt][r] = e;
      l += e;
    }
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (qv && hi == 0) p.lse[((int64_t)b * p.nh + h) * p.Lq + q] = (l > 0.f) ? (msafe + __logf(l)) : -INFINITY;

  const bool drop = p.p > 0.f;
  const uint32_t thr = goat_thr16(p.p);
  const float keep_scale = drop ? 1.f / (1.f - p.p) : 1.f;
  const AttnMask<T> rng(p, b, h);
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pv = s[jt][r] * inv;
      if (drop) {
        const int key = jt * 32 + c_row(r, lane);
        pv = rng.keep(p, q, key, thr) ? pv * keep_scale : 0.f;
      }
      s[jt][r] = pv;
    }

  __syncthreads();  // V staged

  // O (q x d) = P (q x keys) · V (keys x d)
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
    for (int st = 0; st < TSTEPS; ++st) {
      Frag pa = acc_frag<T>(s[jt], st);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        Frag vb = bfrag_crow<T>(vl, LSTR, jt * 32, st, dt, lane);
        mma32(o[dt], pa, vb);
      }
    }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + c_row(r, lane);
      if (qq < p.Lq) Ob[(int64_t)qq * p.o_rs + dt * 32 + l31] = from_f<T>(o[dt][r]);
    }
}

// ======================================================================================== backward
// Two barrier-free single-wave roles per (b, h); S and dP are recomputed in each (cheaper than the cross-wave
// reductions, LDS transposes and barriers of a one-workgroup-per-head kernel: 17 % -> 8 % of the step):
//   dQ role  (b, h, q-tile):  lane = query;  S^T = K·Q^T, dP^T = V·dO^T per key tile; dQ += dS·K
//   dKV role (b, h, key-tile): lane = key;   S = Q·K^T, dP = dO·V^T per query tile; dV += Pd^T·dO, dK += dS^T·Q
// "Transposed" B operands (K, dO, Q with the contraction index as the LDS row) come from bfrag_crow.
template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename AT<T>::Frag Frag;
  constexpr int KSTEPS = AT<T>::KSTEPS, LSTR = AT<T>::LSTR, TSTEPS = AT<T>::TSTEPS, NE = AT<T>::NE;
  T* kl = reinterpret_cast<T*>(smem);  // [nkt*32][LSTR]
  const int lane = threadIdx.x, hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
  const int q0 = blockIdx.y * 32, q = q0 + l31;
  const bool qv = q < p.Lq;
  const int nkt = (p.Lk + 31) / 32;
  const T* Qb = reinterpret_cast<const T*>(p.Q) + b * p.q_bs + h * HD;
  const T* Kb = reinterpret_cast<const T*>(p.K) + b * p.k_bs + h * HD;
  const T* Vb = reinterpret_cast<const T*>(p.V) + b * p.v_bs + h * HD;
  const T* Ob = reinterpret_cast<const T*>(p.O) + b * p.o_bs + h * HD;
  const T* dOb = reinterpret_cast<const T*>(p.dO) + b * p.do_bs + h * HD;
  T* dQb = reinterpret_cast<T*>(p.dQ) + b * p.dq_bs + h * HD;

  stage_rows<T>(kl, Kb, p.k_rs, 0, p.Lk, nkt * 32, lane, 64);
  Frag qf[KSTEPS], dof[KSTEPS];
  float dsum = 0.f;
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    qf[ks] = gfrag<T>(Qb + (int64_t)q * p.q_rs, qv, ks, hi);
    dof[ks] = gfrag<T>(dOb + (int64_t)q * p.do_rs, qv, ks, hi);
    Frag of = gfrag<T>(Ob + (int64_t)q * p.o_rs, qv, ks, hi);
#pragma unroll
    for (int e = 0; e < NE; ++e) dsum += to_f(dof[ks][e]) * to_f(of[e]);
  }
  dsum += __shfl_xor(dsum, 32, 64);
  const float lse_q = qv ? p.lse[((int64_t)b * p.nh + h) * p.Lq + q] : 0.f;
  const bool lse_ok = (lse_q != -INFINITY);
  const bool drop = p.p > 0.f;
  const uint32_t thr = goat_thr16(p.p);
  const float keep_scale = drop ? 1.f / (1.f - p.p) : 1.f;
  const AttnMask<T> rng(p, b, h);
  __syncthreads();

  f32x16 dq[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  for (int jt = 0; jt < nkt; ++jt) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    const int key_l = jt * 32 + l31;
    const bool kvl = key_l < p.Lk;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      Frag kf = *reinterpret_cast<const Frag*>(kl + key_l * LSTR + (ks * 2 + hi) * NE);
      Frag vf = gfrag<T>(Vb + (int64_t)key_l * p.v_rs, kvl, ks, hi);
      mma32(s, kf, qf[ks]);
      mma32(dp, vf, dof[ks]);
    }
    f32x16 ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = jt * 32 + c_row(r, lane);
      float pr = 0.f;
      if (qv && key < p.Lk && lse_ok) {
        float v = s[r] * p.scale;
        if (p.kmask) v += p.kmask[(int64_t)b * p.Lk + key];
        if (p.bias) v += p.bias[((int64_t)b * p.Lq + q) * p.Lk + key];
        pr = __expf(v - lse_q);
      }
      float keep = 1.f;
      if (drop) keep = rng.keep(p, q, key, thr) ? keep_scale : 0.f;
      const float d = pr * (dp[r] * keep - dsum);
      if (p.dbias && qv && key < p.Lk) atomicAdd(p.dbias + ((int64_t)b * p.Lq + q) * p.Lk + key, d);
      ds[r] = d * p.scale;
    }
#pragma unroll
    for (int st = 0; st < TSTEPS; ++st) {
      Frag a = acc_frag<T>(ds, st);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        Frag kb = bfrag_crow<T>(kl, LSTR, jt * 32, st, dt, lane);
        mma32(dq[dt], a, kb);
      }
    }
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + c_row(r, lane);
      if (qq < p.Lq) dQb[(int64_t)qq * p.dq_rs + dt * 32 + l31] = from_f<T>(dq[dt][r]);
    }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_dkv_kernel(AttnArgs p) {
  typedef typename AT<T>::Frag Frag;
  constexpr int KSTEPS = AT<T>::KSTEPS, LSTR = AT<T>::LSTR, TSTEPS = AT<T>::TSTEPS, NE = AT<T>::NE;
  __shared__ __attribute__((aligned(16))) T ql[32 * AT<T>::LSTR];
  __shared__ __attribute__((aligned(16))) T dol[32 * AT<T>::LSTR];
  __shared__ float rowd[32], rowl[32];
  const int lane = threadIdx.x, hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
  const int k0 = blockIdx.y * 32, key = k0 + l31;
  const bool kv = key < p.Lk;
  const T* Qb = reinterpret_cast<const T*>(p.Q) + b * p.q_bs + h * HD;
  const T* Kb = reinterpret_cast<const T*>(p.K) + b * p.k_bs + h * HD;
  const T* Vb = reinterpret_cast<const T*>(p.V) + b * p.v_bs + h * HD;
  const T* Ob = reinterpret_cast<const T*>(p.O) + b * p.o_bs + h * HD;
  const T* dOb = reinterpret_cast<const T*>(p.dO) + b * p.do_bs + h * HD;
  T* dKb = reinterpret_cast<T*>(p.dK) + b * p.dk_bs + h * HD;
  T* dVb = reinterpret_cast<T*>(p.dV) + b * p.dv_bs + h * HD;

  Frag kf[KSTEPS], vf[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    kf[ks] = gfrag<T>(Kb + (int64_t)key * p.k_rs, kv, ks, hi);
    vf[ks] = gfrag<T>(Vb + (int64_t)key * p.v_rs, kv, ks, hi);
  }
  const float kmv = kv ? (p.kmask ? p.kmask[(int64_t)b * p.Lk + key] : 0.f) : -INFINITY;
  const bool drop = p.p > 0.f;
  const uint32_t thr = goat_thr16(p.p);
  const float keep_scale = drop ? 1.f / (1.f - p.p) : 1.f;
  const AttnMask<T> rng(p, b, h);

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

  const int nqt = (p.Lq + 31) / 32;
  for (int it = 0; it < nqt; ++it) {
    const int q0 = it * 32;
    __syncthreads();
    stage_rows<T>(ql, Qb, p.q_rs, q0, p.Lq, 32, lane, 64);
    stage_rows<T>(dol, dOb, p.do_rs, q0, p.Lq, 32, lane, 64);
    {  // D_q and lse_q for row l31 of this query tile
      const int qq = q0 + l31;
      float dsum = 0.f;
      if (qq < p.Lq) {
        const T* orow = Ob + (int64_t)qq * p.o_rs + hi * 32;
        const T* drow = dOb + (int64_t)qq * p.do_rs + hi * 32;
#pragma unroll
        for (int c = 0; c < 32 / NE; ++c) {
          Chunk<T> a, d2;
          a.load(orow + c * NE);
          d2.load(drow + c * NE);
#pragma unroll
          for (int e = 0; e < NE; ++e) dsum += a.v[e] * d2.v[e];
        }
      }
      dsum += __shfl_xor(dsum, 32, 64);
      if (hi == 0) {
        rowd[l31] = dsum;
        rowl[l31] = (qq < p.Lq) ? p.lse[((int64_t)b * p.nh + h) * p.Lq + qq] : -INFINITY;
      }
    }
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      Frag qa = *reinterpret_cast<const Frag*>(ql + l31 * LSTR + (ks * 2 + hi) * NE);
      Frag da = *reinterpret_cast<const Frag*>(dol + l31 * LSTR + (ks * 2 + hi) * NE);
      mma32(s, qa, kf[ks]);
      mma32(dp, da, vf[ks]);
    }
    f32x16 pd, ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = c_row(r, lane);
      const int q = q0 + qr;
      const float lse_q = rowl[qr];
      float pr = 0.f;
      if (q < p.Lq && kv && lse_q != -INFINITY) {
        float v = s[r] * p.scale + kmv;
        if (p.bias) v += p.bias[((int64_t)b * p.Lq + q) * p.Lk + key];
        pr = __expf(v - lse_q);
      }
      float keep = 1.f;
      if (drop) keep = rng.keep(p, q, key, thr) ? keep_scale : 0.f;
      pd[r] = pr * keep;
      ds[r] = pr * (dp[r] * keep - rowd[qr]) * p.scale;
    }
#pragma unroll
    for (int st = 0; st < TSTEPS; ++st) {
      Frag ap = acc_frag<T>(pd, st);
      Frag as = acc_frag<T>(ds, st);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        Frag bo = bfrag_crow<T>(dol, LSTR, 0, st, dt, lane);
        mma32(dv[dt], ap, bo);
        Frag bq = bfrag_crow<T>(ql, LSTR, 0, st, dt, lane);
        mma32(dk[dt], as, bq);
      }
    }
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = k0 + c_row(r, lane);
      if (kk < p.Lk) {
        dKb[(int64_t)kk * p.dk_rs + dt * 32 + l31] = from_f<T>(dk[dt][r]);
        dVb[(int64_t)kk * p.dv_rs + dt * 32 + l31] = from_f<T>(dv[dt][r]);
      }
    }
}

template <typename T>
int launch_bwd2(hipStream_t st, const AttnArgs& a) {
  const int nkt = (a.Lk + 31) / 32, nqt = (a.Lq + 31) / 32;
  const size_t sm = (size_t)nkt * 32 * AT<T>::LSTR * sizeof(T);
  static size_t attr = 0;
  if (sm > 64 * 1024 && sm > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    if (e != hipSuccess) return (int)e;
    attr = sm;
  }
  hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, dim3(a.B * a.nh, nqt), dim3(64), sm, st, a);
  GOAT_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<T>, dim3(a.B * a.nh, nkt), dim3(64), 0, st, a);
  GOAT_LAUNCH_CHECK();
  return 0;
}

template <typename T>
size_t fwd_smem(int nkt) { return (size_t)nkt * 32 * AT<T>::LSTR * sizeof(T); }

template <typename K>
int set_smem(K kern, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

template <typename T, int NKT>
int launch_fwd(hipStream_t st, const AttnArgs& a) {
  const int nqt = (a.Lq + 31) / 32;
  const int wpb = nqt < 8 ? nqt : 8;  // waves per block
  dim3 grid(a.B * a.nh, (nqt + wpb - 1) / wpb);
  size_t sm = fwd_smem<T>(NKT);
  int e = set_smem(attn_fwd_kernel<T, NKT>, sm);
  if (e) return e;
  hipLaunchKernelGGL((attn_fwd_kernel<T, NKT>), grid, dim3(64 * wpb), sm, st, a);
  GOAT_LAUNCH_CHECK();
  return 0;
}
template <typename T>
int dispatch_fwd(hipStream_t st, const AttnArgs& a) {
  const int nkt = (a.Lk + 31) / 32;
#define GOAT_ATTN_CASE(N) \
  case N: return launch_fwd<T, N>(st, a);
  switch (nkt) {
    GOAT_ATTN_CASE(1) GOAT_ATTN_CASE(2) GOAT_ATTN_CASE(3) GOAT_ATTN_CASE(4)
    GOAT_ATTN_CASE(5) GOAT_ATTN_CASE(6) GOAT_ATTN_CASE(7) GOAT_ATTN_CASE(8)
  }
#undef GOAT_ATTN_CASE
  return GOAT_E_SHAPE;
}

bool use_v2() {       // GOAT_ATTN_V1=1: the round-1 kernels for every problem (A/B experiments)
  static const bool v = !(getenv("GOAT_ATTN_V1") && getenv("GOAT_ATTN_V1")[0] == '1');
  return v;
}

bool strides_ok(int dtype, int64_t rs, int64_t bs, const void* ptr) {
  const int epc = dtype == GOAT_BF16 ? 8 : 4;
  return (rs % epc) == 0 && (bs % epc) == 0 && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0;
}

}  // namespace

extern "C" int goat_attn_fwd(void* stream, int dtype, const void* Q, int64_t q_rs, int64_t q_bs, const void* K,
                             int64_t k_rs, int64_t k_bs, const void* V, int64_t v_rs, int64_t v_bs, void* O,
                             int64_t o_rs, int64_t o_bs, const float* kmask, const float* bias, float* lse, int B,
                             int nh, int Lq, int Lk, float scale, float p, uint64_t seed, uint64_t offset,
                             const uint64_t* rng_dev) {
  if (!Q || !K || !V || !O || !lse) return GOAT_E_ARG;
  if (B <= 0 || nh <= 0 || Lq <= 0 || Lk <= 0 || Lk > 256) return GOAT_E_SHAPE;
  if (dtype != GOAT_F32 && dtype != GOAT_BF16) return GOAT_E_ARG;
  if (!strides_ok(dtype, q_rs, q_bs, Q) || !strides_ok(dtype, k_rs, k_bs, K) || !strides_ok(dtype, v_rs, v_bs, V))
    return GOAT_E_SHAPE;
  AttnArgs a = {};
  a.Q = Q; a.K = K; a.V = V; a.Ow = O;
  a.q_rs = q_rs; a.q_bs = q_bs; a.k_rs = k_rs; a.k_bs = k_bs; a.v_rs = v_rs; a.v_bs = v_bs; a.o_rs = o_rs; a.o_bs = o_bs;
  a.kmask = kmask; a.bias = bias; a.lse = lse;
  a.B = B; a.nh = nh; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.p = p; a.seed = seed; a.offset = offset; a.rng_dev = rng_dev;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == GOAT_BF16 && use_v2()) {          // LDS-staged kernels (attention2.hip) for Lq, Lk <= 128
    const int rc = goat_attn2_fwd(st, a);
    if (rc != GOAT_E_SHAPE) return rc;
  }
  return dtype == GOAT_BF16 ? dispatch_fwd<bf16_t>(st, a) : dispatch_fwd<float>(st, a);
}

extern "C" int goat_attn_bwd(void* stream, int dtype, const void* Q, int64_t q_rs, int64_t q_bs, const void* K,
                             int64_t k_rs, int64_t k_bs, const void* V, int64_t v_rs, int64_t v_bs, const void* O,
                             int64_t o_rs, int64_t o_bs, const void* dO, int64_t do_rs, int64_t do_bs, void* dQ,
                             int64_t dq_rs, int64_t dq_bs, void* dK, int64_t dk_rs, int64_t dk_bs, void* dV,
                             int64_t dv_rs, int64_t dv_bs, const float* kmask, const float* bias, const float* lse,
                             float* dbias, int B, int nh, int Lq, int Lk, float scale, float p, uint64_t seed,
                             uint64_t offset, const uint64_t* rng_dev) {
  if (!Q || !K || !V || !O || !dO || !dQ || !dK || !dV || !lse) return GOAT_E_ARG;
  if (B <= 0 || nh <= 0 || Lq <= 0 || Lk <= 0 || Lk > 256) return GOAT_E_SHAPE;
  if (dtype != GOAT_F32 && dtype != GOAT_BF16) return GOAT_E_ARG;
  if (!strides_ok(dtype, q_rs, q_bs, Q) || !strides_ok(dtype, k_rs, k_bs, K) || !strides_ok(dtype, v_rs, v_bs, V) ||
      !strides_ok(dtype, o_rs, o_bs, O) || !strides_ok(dtype, do_rs, do_bs, dO) || !strides_ok(dtype, dq_rs, dq_bs, dQ))
    return GOAT_E_SHAPE;
  AttnArgs a = {};
  a.Q = Q; a.K = K; a.V = V; a.O = O; a.dO = dO; a.dQ = dQ; a.dK = dK; a.dV = dV;
  a.q_rs = q_rs; a.q_bs = q_bs; a.k_rs = k_rs; a.k_bs = k_bs; a.v_rs = v_rs; a.v_bs = v_bs; a.o_rs = o_rs; a.o_bs = o_bs;
  a.do_rs = do_rs; a.do_bs = do_bs; a.dq_rs = dq_rs; a.dq_bs = dq_bs; a.dk_rs = dk_rs; a.dk_bs = dk_bs;
  a.dv_rs = dv_rs; a.dv_bs = dv_bs;
  a.kmask = kmask; a.bias = bias; a.lse = const_cast<float*>(lse); a.dbias = dbias;
  a.B = B; a.nh = nh; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.p = p; a.seed = seed; a.offset = offset; a.rng_dev = rng_dev;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == GOAT_BF16 && use_v2()) {
    const int rc = goat_attn2_bwd(st, a);
    if (rc != GOAT_E_SHAPE) return rc;
  }
  return dtype == GOAT_BF16 ? launch_bwd2<bf16_t>(st, a) : launch_bwd2<float>(st, a);
}
