// HBM-bound row kernels: residual+dropout+LayerNorm (fwd/bwd), dropout, transpose(+column sums),
// adaptive panorama fusion, gather / segment-mean.  One wave64 per 768-wide row, 16-byte vector loads,
// wavefront shuffles for the reductions, statistics in f32.
#include <cstdlib>
#include "common.hpp"
#include <algorithm>
#include <vector>

namespace {

#ifndef GOAT_LN_ABL
#define GOAT_LN_ABL 0      // (ablation builds of scripts/ln_bench.py: 1 no column reduction, 2 no dropout hash, 4 no stores)
#endif
constexpr int MAXC_MAX = 8;  // 16-B chunks per lane kept in registers: H <= 64*MAXC*EPC (MAXC is a template parameter)

// ------------------------------------------------------------------------------------ LayerNorm fwd
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float p, uint64_t seed, uint64_t offset,
                                                     const uint64_t* __restrict__ rng_dev, T* __restrict__ y, T* __restrict__ zout, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int H, float p_out, uint64_t offset_out,
                                                     const T* __restrict__ post_add) {
  constexpr int EPC = DT<T>::EPC;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  if (rng_dev) seed += *rng_dev;
  const GoatRng rng(seed);
  const int nchunk = H / EPC;
  const bool drop = p > 0.f;
  const uint32_t thr = goat_thr16(p);
  const float ks = drop ? 1.f / (1.f - p) : 1.f;
  Chunk<T> v[MAXC];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const int64_t base = (int64_t)row * H + c * EPC;
      v[i].load(x + base);
      if (drop) {
        const uint32_t km = rng.keep_bits<EPC>(offset + base, thr);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[i].v[e] = ((km >> e) & 1u) ? v[i].v[e] * ks : 0.f;
      }
      if (res) {
        Chunk<T> r;
        r.load(res + base);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[i].v[e] += r.v[e];
      }
      // round z to the storage type so forward statistics and backward recomputation agree
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[i].v[e] = to_f(from_f<T>(v[i].v[e]));
      if (zout) v[i].store_stream(zout + base);
#pragma unroll
      for (int e = 0; e < EPC; ++e) sum += v[i].v[e];
    }
  }
  const float mu = wave_sum(sum) / H;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { float d = v[i].v[e] - mu; var += d * d; }
    }
  }
  const float rs = rsqrtf(wave_sum(var) / H + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      Chunk<T> o;
#pragma unroll
      for (int e = 0; e < EPC; ++e) o.v[e] = (v[i].v[e] - mu) * rs * gamma[c * EPC + e] + beta[c * EPC + e];
      if (post_add) {          // a second summand behind the norm: y = dropout(LayerNorm(z) + post_add) — the image embedding block adds
        Chunk<T> t;            // the normalised location features to the normalised view features (P/model/vilmodel_goat.py:340-344)
        t.load(post_add + (int64_t)row * H + c * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.v[e] += t.v[e];
      }
      if (p_out > 0.f) {       // dropout on the OUTPUT (embeddings: dropout(LayerNorm(e)), P/model/Bert_backbone.py:108-110)
        const uint32_t km = rng.keep_bits<EPC>(offset_out + (int64_t)row * H + c * EPC, goat_thr16(p_out));
        const float ko = 1.f / (1.f - p_out);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.v[e] = ((km >> e) & 1u) ? o.v[e] * ko : 0.f;
      }
      o.store_stream(y + (int64_t)row * H + c * EPC);
    }
  }
}

// ------------------------------------------------------------------------------------ LayerNorm bwd
// grid = NPART blocks of 4 waves; wave w of block b walks rows (b*4+w), +4*NPART, ...
// partial dgamma/dbeta per block -> ws[block][2][H]; ln_bwd_reduce sums them.
template <typename T, int MAXC, int RIF, int NWV>
__global__ __launch_bounds__(64 * NWV) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ z,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float p, uint64_t seed,
                                                     uint64_t offset, const uint64_t* __restrict__ rng_dev,
                                                     T* __restrict__ dx, T* __restrict__ dres,
                                                     float* __restrict__ ws, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     const T* __restrict__ dx_add, int pre_add, int M, int H, float p_out,
                                                     uint64_t offset_out, T* __restrict__ d_post) {
  constexpr int EPC = DT<T>::EPC;
  extern __shared__ float lsum[];  // [NWV][2][H]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (rng_dev) seed += *rng_dev;
  const GoatRng rng(seed);
  const int nchunk = H / EPC;
  const bool drop = p > 0.f;
  const uint32_t thr = goat_thr16(p);
  const float ks = drop ? 1.f / (1.f - p) : 1.f;
  float dg[MAXC][EPC], db[MAXC][EPC], g[MAXC][EPC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      dg[i][e] = 0.f; db[i][e] = 0.f;
      const int c = lane + 64 * i;
      g[i][e] = (c < nchunk) ? gamma[c * EPC + e] : 0.f;
    }
  // RIF rows per wave in flight: the loads of both rows are issued before the first reduction (this kernel is
  // latency-bound on its 16-B loads, not on the shuffles)
  const int rstride = gridDim.x * NWV;
  for (int row0 = blockIdx.x * NWV + wave; row0 < M; row0 += RIF * rstride) {
    Chunk<T> vdy[RIF][MAXC], vz[RIF][MAXC];
    float mu[RIF], rs[RIF], c1[RIF], c2[RIF];
#pragma unroll
    for (int u = 0; u < RIF; ++u) { c1[u] = 0.f; c2[u] = 0.f; }
    bool live[RIF];
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      const int row = row0 + u * rstride;
      live[u] = row < M;
      mu[u] = live[u] ? mean[row] : 0.f;
      rs[u] = live[u] ? rstd[row] : 0.f;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (live[u] && c < nchunk) {
          const int64_t base = (int64_t)row * H + c * EPC;
          vdy[u][i].load(dy + base);
          vz[u][i].load(z + base);
          if (dy2) {   // second upstream gradient of the same tensor (its other consumer): summed here instead of by an add kernel
            Chunk<T> t;
            t.load(dy2 + base);
#pragma unroll
            for (int e = 0; e < EPC; ++e) vdy[u][i].v[e] += t.v[e];
          }
          if (p_out > 0.f) {     // forward dropped its OUTPUT with this mask: the gradient of the normalised value is masked the same way
            const uint32_t km = rng.keep_bits<EPC>(offset_out + base, goat_thr16(p_out));
            const float ko = 1.f / (1.f - p_out);
#pragma unroll
            for (int e = 0; e < EPC; ++e) vdy[u][i].v[e] = ((km >> e) & 1u) ? vdy[u][i].v[e] * ko : 0.f;
          }
          if (d_post) vdy[u][i].store_stream(d_post + base);      // gradient of the summand added behind the norm (post_add)
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (live[u] && c < nchunk) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float xh = (vz[u][i].v[e] - mu[u]) * rs[u];
            const float d = vdy[u][i].v[e];
            dg[i][e] += d * xh;
            db[i][e] += d;
            const float dxh = d * g[i][e];
            c1[u] += dxh;
            c2[u] += dxh * xh;
            vz[u][i].v[e] = xh;
            vdy[u][i].v[e] = dxh;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      c1[u] = wave_sum(c1[u]) / H;
      c2[u] = wave_sum(c2[u]) / H;
    }
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      const int row = row0 + u * rstride;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (live[u] && c < nchunk) {
          const int64_t base = (int64_t)row * H + c * EPC;
          Chunk<T> o;
#pragma unroll
          for (int e = 0; e < EPC; ++e) o.v[e] = (vdy[u][i].v[e] - c1[u] - vz[u][i].v[e] * c2[u]) * rs[u];
#if GOAT_LN_ABL & 4
          asm volatile("" ::"v"(o.v[0]), "v"(o.v[EPC - 1]));
          continue;
#endif
          if (dx_add && pre_add) {     // gradient arriving at the PRE-NORM SUM z = residual + dropout(x) through its other consumer (the
            Chunk<T> t;                // skip connection that continues behind this LayerNorm): joins before the residual / dropout split
            t.load(dx_add + base);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.v[e] += t.v[e];
          }
          if (dres) o.store_stream(dres + base);
#if GOAT_LN_ABL & 2
          if (false) {
#else
          if (drop) {
#endif
            const uint32_t km = rng.keep_bits<EPC>(offset + base, thr);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.v[e] = ((km >> e) & 1u) ? o.v[e] * ks : 0.f;
          }
          if (dx_add && !pre_add) {     // gradient of x arriving through its OTHER consumer (the skip connection of a pre-LN block): summed on store
            Chunk<T> t;
            t.load(dx_add + base);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.v[e] += t.v[e];
          }
          if (dx) o.store_stream(dx + base);
        }
      }
    }
  }
  // block reduction of the column partials
#if GOAT_LN_ABL & 1
#pragma unroll
  for (int i = 0; i < MAXC; ++i) asm volatile("" ::"v"(dg[i][0]), "v"(db[i][0]), "v"(dg[i][EPC - 1]), "v"(db[i][EPC - 1]));
  return;
#endif
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        lsum[(wave * 2 + 0) * H + c * EPC + e] = dg[i][e];
        lsum[(wave * 2 + 1) * H + c * EPC + e] = db[i][e];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * H; i += 64 * NWV) {
    const int which = i / H, col = i % H;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) s += lsum[(w * 2 + which) * H + col];
    if (ws != nullptr) ws[(int64_t)blockIdx.x * 2 * H + i] = s;          // deterministic mode: per-block partials + ln_bwd_reduce_kernel
    else atomicAdd((which == 0 ? dgamma : dbeta) + col, s);                // default: one float atomic per block and column into the
  }                                                                        // (pre-zeroed) gradient; no second launch
}

// ---- bf16, H = 768 (every LayerNorm of the model at its hidden size): the backward on a register diet.
// The generic kernel above keeps two rows in flight as float (16-byte chunks: the second chunk of a 768-wide row occupies half the
// lanes) and needs 192 VGPRs — two waves per SIMD, and ONE slot beside a 240-VGPR GEMM workgroup of the other graph branch (measured
// in the step: 18.9 us against 10.7 alone).  Here a lane owns 12 elements of a row as THREE 8-BYTE chunks (all 64 lanes busy), the
// row being reduced lives in float, the NEXT row of the wave is in flight in its packed bf16 form (12 registers for dy | z, converted
// on use): <= 96 VGPRs -> five waves per SIMD alone, two beside a GEMM wave.  Same arithmetic, order and results as ln_bwd_kernel.
__device__ __forceinline__ void goat_store_stream8(bf16_t* p, const bf16x4& v) {
#if GOAT_ROW_NT
  __builtin_nontemporal_store(v, reinterpret_cast<bf16x4*>(p));
#else
  *reinterpret_cast<bf16x4*>(p) = v;
#endif
}

template <int NWV, int RIF, bool DROP, bool DY2, bool EXTRA>
__device__ __forceinline__ void ln_bwd768_body(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dy2,
                                                             const bf16_t* __restrict__ z, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, float p,
                                                             uint64_t seed, uint64_t offset, const uint64_t* __restrict__ rng_dev,
                                                             bf16_t* __restrict__ dx, bf16_t* __restrict__ dres, float* __restrict__ ws,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             const bf16_t* __restrict__ dx_add, int pre_add, int M, float p_out,
                                                             uint64_t offset_out, bf16_t* __restrict__ d_post) {
  constexpr int H = 768, NC = 3, E = 4;
  extern __shared__ float lsum[];  // [NWV][2][H] column partials, then gamma [H] (read per row: 12 registers less than holding it)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* __restrict__ lgam = lsum + NWV * 2 * H;
  float dg[NC][E], db[NC][E];
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < NC; ++i)
      *reinterpret_cast<f32x4*>(lgam + (lane + 64 * i) * E) = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * i) * E);
  }
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int e = 0; e < E; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; }
  __syncthreads();
  if (rng_dev) seed += *rng_dev;
  const int rstride = gridDim.x * NWV;
  for (int row0 = blockIdx.x * NWV + wave; row0 < M; row0 += RIF * rstride) {
    // RIF rows of the wave in flight in their packed form (the loads of all of them are issued before the first reduction)
    bf16x4 pdy[RIF][NC], pz[RIF][NC], pdy2[RIF][NC];
    float mus[RIF], rss[RIF];
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      const int row = row0 + u * rstride;
      if (row < M) {
        const uint32_t rb = (uint32_t)row * H + lane * E;      // (the launcher sends rows beyond 2^31 bytes to the generic kernel)
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          pdy[u][i] = *reinterpret_cast<const bf16x4*>(dy + rb + 64 * E * i);
          pz[u][i] = *reinterpret_cast<const bf16x4*>(z + rb + 64 * E * i);
          if (DY2) pdy2[u][i] = *reinterpret_cast<const bf16x4*>(dy2 + rb + 64 * E * i);
        }
        mus[u] = mean[row];
        rss[u] = rstd[row];
      }
    }
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      __builtin_amdgcn_sched_barrier(0);      // one row in float at a time: the conversions of row u + 1 stay behind row u's stores
      const int row = row0 + u * rstride;
      if (row >= M) break;
      float d[NC][E], xh[NC][E];
      const float mu = mus[u], rs = rss[u];
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          d[i][e] = (float)pdy[u][i][e];
          if (DY2) d[i][e] += (float)pdy2[u][i][e];      // the second upstream gradient of the same tensor (its other consumer)
          xh[i][e] = ((float)pz[u][i][e] - mu) * rs;
        }
      const uint32_t rbase = (uint32_t)row * H + lane * E;
      if (EXTRA) {
        if (p_out > 0.f) {       // forward dropped its OUTPUT with this mask
          const GoatRng rng(seed);
          const float ko = 1.f / (1.f - p_out);
#pragma unroll
          for (int i = 0; i < NC; ++i) {
            const uint32_t km = rng.keep_bits<E>(offset_out + rbase + 64 * E * i, goat_thr16(p_out));
#pragma unroll
            for (int e = 0; e < E; ++e) d[i][e] = ((km >> e) & 1u) ? d[i][e] * ko : 0.f;
          }
        }
        if (d_post) {
#pragma unroll
          for (int i = 0; i < NC; ++i) {
            bf16x4 t;
#pragma unroll
            for (int e = 0; e < E; ++e) t[e] = (bf16_t)d[i][e];
            goat_store_stream8(d_post + rbase + 64 * E * i, t);
          }
        }
      }
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(lgam + (lane + 64 * i) * E);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float dd = d[i][e];
          dg[i][e] += dd * xh[i][e];
          db[i][e] += dd;
          const float dxh = dd * g[e];
          c1 += dxh;
          c2 += dxh * xh[i][e];
          d[i][e] = dxh;
        }
      }
      c1 = wave_sum(c1) / H;
      c2 = wave_sum(c2) / H;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const uint32_t base = rbase + 64 * E * i;
        float o[E];
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = (d[i][e] - c1 - xh[i][e] * c2) * rs;
        if (EXTRA && dx_add && pre_add) {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>(dx_add + base);
#pragma unroll
          for (int e = 0; e < E; ++e) o[e] += (float)t[e];
        }
        if (dres) {
          bf16x4 t;
#pragma unroll
          for (int e = 0; e < E; ++e) t[e] = (bf16_t)o[e];
          goat_store_stream8(dres + base, t);
        }
        if (DROP) {
          const GoatRng rng(seed);
          const float ks = 1.f / (1.f - p);
          const uint32_t km = rng.keep_bits<E>(offset + base, goat_thr16(p));
#pragma unroll
          for (int e = 0; e < E; ++e) o[e] = ((km >> e) & 1u) ? o[e] * ks : 0.f;
        }
        if (EXTRA && dx_add && !pre_add) {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>(dx_add + base);
#pragma unroll
          for (int e = 0; e < E; ++e) o[e] += (float)t[e];
        }
        if (dx) {
          bf16x4 t;
#pragma unroll
          for (int e = 0; e < E; ++e) t[e] = (bf16_t)o[e];
          goat_store_stream8(dx + base, t);
        }
      }
    }
  }
  // block reduction of the column partials (as ln_bwd_kernel)
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int col = (lane + 64 * i) * E;
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < E; ++e) { a[e] = dg[i][e]; b[e] = db[i][e]; }
    *reinterpret_cast<f32x4*>(&lsum[(wave * 2 + 0) * H + col]) = a;
    *reinterpret_cast<f32x4*>(&lsum[(wave * 2 + 1) * H + col]) = b;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * H; i += 64 * NWV) {
    const int which = i / H, col = i % H;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) s += lsum[(w * 2 + which) * H + col];
    if (ws != nullptr) ws[(int64_t)blockIdx.x * 2 * H + i] = s;
    else atomicAdd((which == 0 ? dgamma : dbeta) + col, s);
  }
}

#define GOAT_LN768_PARAMS                                                                                                         \
  const bf16_t *__restrict__ dy, const bf16_t *__restrict__ dy2, const bf16_t *__restrict__ z, const float *__restrict__ gamma,    \
      const float *__restrict__ mean, const float *__restrict__ rstd, float p, uint64_t seed, uint64_t offset,                      \
      const uint64_t *__restrict__ rng_dev, bf16_t *__restrict__ dx, bf16_t *__restrict__ dres, float *__restrict__ ws,              \
      float *__restrict__ dgamma, float *__restrict__ dbeta, const bf16_t *__restrict__ dx_add, int pre_add, int M, float p_out,     \
      uint64_t offset_out, bf16_t *__restrict__ d_post
#define GOAT_LN768_ARGS dy, dy2, z, gamma, mean, rstd, p, seed, offset, rng_dev, dx, dres, ws, dgamma, dbeta, dx_add, pre_add, M, p_out, offset_out, d_post
// the common forms (post-LN blocks: dropout, forked output) are held to 128 VGPRs = four waves per SIMD; the forms with a joining
// skip gradient / output dropout / post-add gradient (pre-LN panorama blocks, embeddings) would spill there: three waves per SIMD
template <int NWV, int RIF, bool DROP, bool DY2>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(4, 8))) void ln_bwd768_kernel(GOAT_LN768_PARAMS) {
  ln_bwd768_body<NWV, RIF, DROP, DY2, false>(GOAT_LN768_ARGS);
}
template <int NWV, int RIF, bool DROP, bool DY2>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(3, 8))) void ln_bwd768x_kernel(GOAT_LN768_PARAMS) {
  ln_bwd768_body<NWV, RIF, DROP, DY2, true>(GOAT_LN768_ARGS);
}

// 256 threads = 16 columns x 16 part-groups; every thread sums nparts/16 partials with 4 independent chains
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int nparts, int H, int accumulate) {
  __shared__ float red[16][17];
  const int cx = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < 2 * H) {
    int b = g;
    for (; b + 48 < nparts; b += 64) {
      s0 += ws[(int64_t)b * 2 * H + i];
      s1 += ws[(int64_t)(b + 16) * 2 * H + i];
      s2 += ws[(int64_t)(b + 32) * 2 * H + i];
      s3 += ws[(int64_t)(b + 48) * 2 * H + i];
    }
    for (; b < nparts; b += 16) s0 += ws[(int64_t)b * 2 * H + i];
  }
  red[g][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && i < 2 * H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    float* dst = (i < H) ? dgamma + i : dbeta + (i - H);
    *dst = accumulate ? *dst + t : t;
  }
}

// Deferred reduction (goat_ln_bwd with accumulate == 2 leaves per-block partials behind): ONE launch at the end of the backward pass
// sums the partials of up to 64 LayerNorm calls into their (pre-zeroed / accumulating) gradient vectors.  blockIdx.y = entry.
struct LnPartial { const float* ws; float* dgamma; float* dbeta; int nparts; int pad; };
struct LnReduceArgs { LnPartial e[64]; int n; int H; };
__global__ __launch_bounds__(256) void ln_reduce_batched_kernel(LnReduceArgs a) {
  // entries are sorted by destination: the first entry of a run of equal destinations (a LayerNorm applied several times in one
  // backward pass: shared modules, BPTT) sums the whole run, the others leave — one writer per gradient word, fixed order
  __shared__ float red[16][17];
  const int e0 = blockIdx.y;
  if (e0 > 0 && a.e[e0 - 1].dgamma == a.e[e0].dgamma) return;
  const int H = a.H;
  const int cx = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int e = e0; e < a.n && a.e[e].dgamma == a.e[e0].dgamma; ++e) {
    const float* __restrict__ ws = a.e[e].ws;
    const int nparts = a.e[e].nparts;
    if (i < 2 * H) {
      int b = g;
      for (; b + 48 < nparts; b += 64) {
        s0 += ws[(int64_t)b * 2 * H + i];
        s1 += ws[(int64_t)(b + 16) * 2 * H + i];
        s2 += ws[(int64_t)(b + 32) * 2 * H + i];
        s3 += ws[(int64_t)(b + 48) * 2 * H + i];
      }
      for (; b < nparts; b += 16) s0 += ws[(int64_t)b * 2 * H + i];
    }
  }
  red[g][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && i < 2 * H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    float* dst = (i < H) ? a.e[e0].dgamma + i : a.e[e0].dbeta + (i - H);
    *dst += t;
  }
}

// ------------------------------------------------------------------------------------ dropout (+add)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                      int64_t n, float p, uint64_t seed, uint64_t offset,
                                                      const uint64_t* __restrict__ rng_dev) {
  constexpr int EPC = DT<T>::EPC;
  if (rng_dev) seed += *rng_dev;
  const GoatRng rng(seed);
  const uint32_t thr = goat_thr16(p);
  const bool drop = p > 0.f;
  const float ks = drop ? 1.f / (1.f - p) : 1.f;
  const int64_t nchunk = n / EPC;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * blockDim.x) {
    Chunk<T> v;
    v.load(x + c * EPC);
    if (drop) {
      const uint32_t km = rng.keep_bits<EPC>(offset + c * EPC, thr);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] = ((km >> e) & 1u) ? v.v[e] * ks : 0.f;
    }
    if (!BWD && res) {
      Chunk<T> r;
      r.load(res + c * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] += r.v[e];
    }
    v.store_stream(y + c * EPC);
  }
  // tail
  if (blockIdx.x == 0 && threadIdx.x < (n - nchunk * EPC)) {
    const int64_t i = nchunk * EPC + threadIdx.x;
    float v = to_f(x[i]);
    if (drop) v = rng.keep(offset + i, thr) ? v * ks : 0.f;
    if (!BWD && res) v += to_f(res[i]);
    y[i] = from_f<T>(v);
  }
}


// ------------------------------------------------------------------------------------ activation backward
// dx = dropmask(dy) * act'(u)   (act: 1 gelu, 2 relu): backward of  h = dropout_p(act(u)).
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ u, T* __restrict__ dx,
                                                      int64_t n, int act, float p, uint64_t seed, uint64_t offset,
                                                      const uint64_t* __restrict__ rng_dev) {
  constexpr int EPC = DT<T>::EPC;
  if (rng_dev) seed += *rng_dev;
  const GoatRng rng(seed);
  const uint32_t thr = goat_thr16(p);
  const bool drop = p > 0.f;
  const float ks = drop ? 1.f / (1.f - p) : 1.f;
  const int64_t nchunk = n / EPC;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * blockDim.x) {
    Chunk<T> d, uu;
    d.load(dy + c * EPC);
    uu.load(u + c * EPC);
    const uint32_t km = drop ? rng.keep_bits<EPC>(offset + c * EPC, thr) : 0xFFFFFFFFu;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float g = d.v[e];
      if (drop) g = ((km >> e) & 1u) ? g * ks : 0.f;
      g *= (act == 1) ? dgelu_f(uu.v[e]) : (uu.v[e] > 0.f ? 1.f : 0.f);
      d.v[e] = g;
    }
    d.store_stream(dx + c * EPC);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - nchunk * EPC)) {
    const int64_t i = nchunk * EPC + threadIdx.x;
    float g = to_f(dy[i]);
    const float uv = to_f(u[i]);
    if (drop) g = rng.keep(offset + i, thr) ? g * ks : 0.f;
    g *= (act == 1) ? dgelu_f(uv) : (uv > 0.f ? 1.f : 0.f);
    dx[i] = from_f<T>(g);
  }
}


// ------------------------------------------------------------------------------------ column sums
// block = 256 threads = 64 columns x 4 row lanes; each block walks a strip of rows, one atomic per column.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ld, int R, int C,
                                                     float* __restrict__ colsum, int rows_per_block) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(R, r0 + rows_per_block);
  float acc = 0.f;
  if (c < C)
    for (int r = r0 + ty; r < r1; r += 4) acc += to_f(x[(int64_t)r * ld + c]);
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < C) atomicAdd(colsum + c, part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]);
}


// ------------------------------------------------------------------------------------ softmax cross-entropy
// One 256-thread block per row of f32 logits [M, ld] with N valid columns (ld may be padded).
//   fwd: lse[m] = logsumexp(logits[m,:N]) ; loss[m] = lse - logits[m, target]       (F.cross_entropy, reduction none;
//        P/model/pretrain_goat.py:213-215 on the 576 x 50265 MLM scores)
//   bwd: dlogits[m,n] = (exp(l - lse) - [n == target]) * dloss[m] for n < N, 0 for the padding columns n in [N, ld_out)
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int64_t ld, int N,
                                                     const int64_t* __restrict__ targets, float* __restrict__ loss,
                                                     float* __restrict__ lse) {
  __shared__ float red[4];
  const int m = blockIdx.x;
  const float* row = logits + (int64_t)m * ld;
  float mx = -INFINITY;
  const int n4 = N >> 2;
  for (int c = threadIdx.x; c < n4; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + c * 4);
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  for (int c = n4 * 4 + threadIdx.x; c < N; c += 256) mx = fmaxf(mx, row[c]);
  mx = block_reduce(mx, red, true);
  float sum = 0.f;
  for (int c = threadIdx.x; c < n4; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + c * 4);
    sum += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx);
  }
  for (int c = n4 * 4 + threadIdx.x; c < N; c += 256) sum += __expf(row[c] - mx);
  sum = block_reduce(sum, red, false);
  if (threadIdx.x == 0) {
    const float l = mx + __logf(sum);
    lse[m] = l;
    const int64_t t = targets[m];
    // negative target: ignored row (F.cross_entropy's ignore_index), loss 0.  A target past the row (torch raises there; a kernel
    // cannot) poisons the loss with NaN instead of reading out of bounds
    loss[m] = t < 0 ? 0.f : (t < (int64_t)N ? l - row[t] : __builtin_nanf(""));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int64_t ld, int N,
                                                     const int64_t* __restrict__ targets, const float* __restrict__ lse,
                                                     const float* __restrict__ dloss, T* __restrict__ dlogits, int64_t ld_out) {
  const int m = blockIdx.x;
  const float* row = logits + (int64_t)m * ld;
  T* out = dlogits + (int64_t)m * ld_out;
  const int t = (int)targets[m];
  const float l = lse[m], g = t < 0 ? 0.f : (t < N ? dloss[m] : __builtin_nanf(""));      // ignored row: zero gradient; out-of-range target: NaN (see ce_fwd_kernel)
  for (int c = threadIdx.x * 4; c < (int)ld_out; c += 256 * 4) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = c + e;
      v[e] = (n < N) ? (__expf(row[n] - l) - (n == t ? 1.f : 0.f)) * g : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < (int)ld_out) out[c + e] = from_f<T>(v[e]);
  }
}

// ------------------------------------------------------------------------------------ transpose
// 64x64 tiles through LDS; each block walks RT consecutive row tiles of one column tile so the column
// sums (bias gradient) cost one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, int64_t ld_in, T* __restrict__ out,
                                                        int64_t ld_out, int R, int C, float* __restrict__ colsum, int RT) {
  __shared__ T tile[64][66];
  __shared__ float csum[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // ty 0..3
  const int c0 = blockIdx.y * 64;
  float acc = 0.f;
  for (int t = 0; t < RT; ++t) {
    const int r0 = (blockIdx.x * RT + t) * 64;
    if (r0 >= R && r0 >= (int)ld_out) break;
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
      const int r = r0 + i, c = c0 + tx;
      T v = (T)0.f;
      if (r < R && c < C) v = in[(int64_t)r * ld_in + c];
      tile[i][tx] = v;
      acc += to_f(v);
    }
    __syncthreads();
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
      const int c = c0 + i, r = r0 + tx;
      if (c < C && r < (int)ld_out) out[(int64_t)c * ld_out + r] = tile[tx][i];
    }
    __syncthreads();
  }
  if (colsum) {
    csum[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c0 + tx < C) atomicAdd(colsum + c0 + tx, csum[0][tx] + csum[1][tx] + csum[2][tx] + csum[3][tx]);
  }
}

// ------------------------------------------------------------------------------------ weight gradient of a short-input Linear
// dW[n, k] += sum_r dy[r, n] x[r, k] ; dbias[n] += sum_r dy[r, n]   for K <= 16 input features (the 7- / 14-wide position Linears,
// P/model/vilmodel_goat.py:300-303,406,475): block = 128 output columns x a chunk of rows, lane = 2 columns, wave = row phase.
// Replaces a padded-K GEMM into a temporary + slice copy + autograd's `grad += dW` (4 launches per Linear and step).
constexpr int SK_MAXK = 16, SK_BATCH = 8;
template <typename T, int KP, int SK_ROWS>      // SK_ROWS: rows per block (128; 32 for short inputs so that the grid still covers the chip)
__global__ __launch_bounds__(256) void wgrad_smallk_kernel(const T* __restrict__ dy, int64_t ld_dy, const T* __restrict__ x,
                                                           int64_t ld_x, int rows, int N, int K, float* __restrict__ dw,
                                                           int64_t ld_dw, float* __restrict__ dbias) {
  __shared__ float sm[4][128][SK_MAXK + 1];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (row indices stay scalar: x rows
  const int n0 = blockIdx.x * 128 + lane * 2;                                                    //  come through the scalar cache)
  const int r0 = blockIdx.y * SK_ROWS, r1 = min(rows, r0 + SK_ROWS);
  float acc[2][SK_MAXK], bsum[2] = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < SK_MAXK; ++k) { acc[0][k] = 0.f; acc[1][k] = 0.f; }
  const bool in0 = n0 < N, in1 = n0 + 1 < N;
  // SK_BATCH rows per trip: all their loads are issued before the first multiply (the kernel is latency-bound otherwise)
  for (int rb = r0 + wave * SK_BATCH; rb < r1; rb += 4 * SK_BATCH) {
    constexpr int EPC = DT<T>::EPC, NCH = KP / EPC;
    float d0[SK_BATCH], d1[SK_BATCH], xv[SK_BATCH][KP];
#pragma unroll
    for (int u = 0; u < SK_BATCH; ++u) {
      const int r = rb + u;
      const bool live = r < r1;
      d0[u] = (live && in0) ? to_f(dy[(int64_t)r * ld_dy + n0]) : 0.f;
      d1[u] = (live && in1) ? to_f(dy[(int64_t)r * ld_dy + n0 + 1]) : 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {          // one 16-byte load per chunk of the row (same address in every lane)
        Chunk<T> ch;
        if (live) ch.load(x + (int64_t)r * ld_x + c * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) xv[u][c * EPC + e] = live ? ch.v[e] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < SK_BATCH; ++u) {
      bsum[0] += d0[u]; bsum[1] += d1[u];
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        acc[0][k] += d0[u] * xv[u][k];
        acc[1][k] += d1[u] * xv[u][k];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int k = 0; k < SK_MAXK; ++k) sm[wave][lane * 2 + c][k] = acc[c][k];
    sm[wave][lane * 2 + c][SK_MAXK] = bsum[c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 128 * (K + 1); i += 256) {
    const int c = i / (K + 1), k = i % (K + 1), n = blockIdx.x * 128 + c;
    if (n >= N) continue;
    const int kk = k < K ? k : SK_MAXK;
    const float v = sm[0][c][kk] + sm[1][c][kk] + sm[2][c][kk] + sm[3][c][kk];
    if (k < K) atomicAdd(dw + (int64_t)n * ld_dw + k, v);
    else if (dbias) atomicAdd(dbias + n, v);
  }
}

// ------------------------------------------------------------------------------------ pano fusion
// one block (4 waves) per panorama; V <= 64 slots.  score_v = tanh(x_v·a + a0); w = softmax_v(score)
template <typename T>
__global__ __launch_bounds__(256) void pano_fusion_fwd_generic(const T* __restrict__ x, const float* __restrict__ a,
                                                              const float* __restrict__ a0, T* __restrict__ fused,
                                                              float* __restrict__ wsave, int V, int H) {
  __shared__ float sc[64];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* xb = x + (int64_t)n * V * H;
  for (int v = wave; v < V; v += 4) {
    float s = 0.f;
    for (int i = lane; i < H; i += 64) s += to_f(xb[(int64_t)v * H + i]) * a[i];
    s = wave_sum(s);
    if (lane == 0) sc[v] = tanhf(s + a0[0]);
  }
  __syncthreads();
  if (wave == 0) {
    float v = lane < V ? sc[lane] : -INFINITY;
    float m = wave_max(v);
    float e = lane < V ? __expf(v - m) : 0.f;
    float l = wave_sum(e);
    if (lane < V) { sc[lane] = e / l; wsave[(int64_t)n * V + lane] = e / l; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += 256) {
    float s = 0.f;
    for (int v = 0; v < V; ++v) s += sc[v] * to_f(xb[(int64_t)v * H + i]);
    fused[(int64_t)n * H + i] = from_f<T>(s);
  }
}

// backward: df[H] given.  g_v = x_v·df ; d(softmax in)_v = w_v (g_v - sum_u w_u g_u) ;
// dscore_v = d(softmax in)_v * (1 - tanh^2(x_v·a + a0)) ; dx_v = w_v*df + dscore_v * a
template <typename T>
__global__ __launch_bounds__(256) void pano_fusion_bwd_generic(const T* __restrict__ x, const float* __restrict__ a,
                                                              const float* __restrict__ a0,
                                                              const float* __restrict__ wsave, const T* __restrict__ dfused,
                                                              T* __restrict__ dx, float* __restrict__ da,
                                                              float* __restrict__ da0, int V, int H) {
  __shared__ float gl[64], tl[64], dsc[64];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* xb = x + (int64_t)n * V * H;
  const T* df = dfused + (int64_t)n * H;
  for (int v = wave; v < V; v += 4) {
    float g = 0.f, s = 0.f;
    for (int i = lane; i < H; i += 64) {
      const float xv = to_f(xb[(int64_t)v * H + i]);
      g += xv * to_f(df[i]);
      s += xv * a[i];
    }
    g = wave_sum(g);
    s = wave_sum(s);
    if (lane == 0) { gl[v] = g; tl[v] = tanhf(s + a0[0]); }
  }
  __syncthreads();
  if (wave == 0) {
    const float w = lane < V ? wsave[(int64_t)n * V + lane] : 0.f;
    const float g = lane < V ? gl[lane] : 0.f;
    const float wg = wave_sum(w * g);
    if (lane < V) dsc[lane] = w * (g - wg) * (1.f - tl[lane] * tl[lane]);
  }
  __syncthreads();
  for (int v = wave; v < V; v += 4) {
    const float w = wsave[(int64_t)n * V + v];
    const float d = dsc[v];
    for (int i = lane; i < H; i += 64)
      dx[((int64_t)n * V + v) * H + i] = from_f<T>(w * to_f(df[i]) + d * a[i]);
  }
  for (int i = threadIdx.x; i < H; i += 256) {
    float s = 0.f;
    for (int v = 0; v < V; ++v) s += dsc[v] * to_f(xb[(int64_t)v * H + i]);
    atomicAdd(da + i, s);
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int v = 0; v < V; ++v) s += dsc[v];
    atomicAdd(da0, s);
  }
}

// Single-pass versions (H % EPC == 0, H <= 64*EPC*MAXC): one block of 8 waves per panorama, wave w keeps rows w, w+8, ... (<= 8 of
// them) in registers as 16-byte chunks, so x is read once (the generic kernels above read it twice with 2-byte loads: 70 / 82 us
// for 240 panoramas of 36 x 768 against ~8 us here).  Cross-wave sums (the fused vector, da) go through LDS.
constexpr int PF_NW = 8, PF_RMAX = 8;
template <typename T, int MAXC>
__global__ __launch_bounds__(64 * PF_NW) void pano_fusion_fwd_kernel(const T* __restrict__ x, const float* __restrict__ a,
                                                                     const float* __restrict__ a0, T* __restrict__ fused,
                                                                     float* __restrict__ wsave, int V, int H) {
  constexpr int EPC = DT<T>::EPC;
  extern __shared__ float psm[];   // [PF_NW][H]
  __shared__ float sc[64];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = H / EPC;
  const T* xb = x + (int64_t)n * V * H;
  Chunk<T> xv[PF_RMAX][MAXC];
#pragma unroll
  for (int r = 0; r < PF_RMAX; ++r) {
    const int v = wave + PF_NW * r;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (v < V && c < nchunk) xv[r][i].load(xb + (int64_t)v * H + c * EPC);
      else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) xv[r][i].v[e] = 0.f;
      }
    }
  }
  float av[MAXC][EPC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < EPC; ++e) av[i][e] = c < nchunk ? a[c * EPC + e] : 0.f;
  }
  const float bias = a0[0];
#pragma unroll
  for (int r = 0; r < PF_RMAX; ++r) {
    const int v = wave + PF_NW * r;
    if (v < V) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int e = 0; e < EPC; ++e) s += xv[r][i].v[e] * av[i][e];
      s = wave_sum(s);
      if (lane == 0) sc[v] = tanhf(s + bias);
    }
  }
  __syncthreads();
  if (wave == 0) {
    const float v = lane < V ? sc[lane] : -INFINITY;
    const float m = wave_max(v);
    const float e = lane < V ? __expf(v - m) : 0.f;
    const float l = wave_sum(e);
    if (lane < V) { sc[lane] = e / l; wsave[(int64_t)n * V + lane] = e / l; }
  }
  __syncthreads();
  float acc[MAXC][EPC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[i][e] = 0.f;
#pragma unroll
  for (int r = 0; r < PF_RMAX; ++r) {
    const int v = wave + PF_NW * r;
    if (v < V) {
      const float w = sc[v];
#pragma unroll
      for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[i][e] += w * xv[r][i].v[e];
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) psm[wave * H + c * EPC + e] = acc[i][e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += 64 * PF_NW) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < PF_NW; ++w) s += psm[w * H + i];
    fused[(int64_t)n * H + i] = from_f<T>(s);
  }
}

template <typename T, int MAXC>
__global__ __launch_bounds__(64 * PF_NW) void pano_fusion_bwd_kernel(const T* __restrict__ x, const float* __restrict__ a,
                                                                     const float* __restrict__ a0,
                                                                     const float* __restrict__ wsave, const T* __restrict__ dfused,
                                                                     T* __restrict__ dx, float* __restrict__ da,
                                                                     float* __restrict__ da0, int V, int H) {
  constexpr int EPC = DT<T>::EPC;
  extern __shared__ float psm[];   // [PF_NW][H]
  __shared__ float gl[64], tl[64], dsc[64], wl[64];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = H / EPC;
  const T* xb = x + (int64_t)n * V * H;
  Chunk<T> xv[PF_RMAX][MAXC];
#pragma unroll
  for (int r = 0; r < PF_RMAX; ++r) {
    const int v = wave + PF_NW * r;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (v < V && c < nchunk) xv[r][i].load(xb + (int64_t)v * H + c * EPC);
      else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) xv[r][i].v[e] = 0.f;
      }
    }
  }
  float av[MAXC][EPC];
  Chunk<T> dfv[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) dfv[i].load(dfused + (int64_t)n * H + c * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      av[i][e] = c < nchunk ? a[c * EPC + e] : 0.f;
      if (c >= nchunk) dfv[i].v[e] = 0.f;
    }
  }
  const float bias = a0[0];
#pragma unroll
  for (int r = 0; r < PF_RMAX; ++r) {
    const int v = wave + PF_NW * r;
    if (v < V) {
      float g = 0.f, s = 0.f;
#pragma unroll
      for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int e = 0; e < EPC; ++e) { g += xv[r][i].v[e] * dfv[i].v[e]; s += xv[r][i].v[e] * av[i][e]; }
      g = wave_sum(g);
      s = wave_sum(s);
      if (lane == 0) { gl[v] = g; tl[v] = tanhf(s + bias); }
    }
  }
  __syncthreads();
  if (wave == 0) {
    const float w = lane < V ? wsave[(int64_t)n * V + lane] : 0.f;
    const float g = lane < V ? gl[lane] : 0.f;
    const float wg = wave_sum(w * g);
    const float d = w * (g - wg) * (1.f - tl[lane < V ? lane : 0] * tl[lane < V ? lane : 0]);
    if (lane < V) { dsc[lane] = d; wl[lane] = w; }
    const float tot = wave_sum(lane < V ? d : 0.f);
    if (lane == 0) atomicAdd(da0, tot);
  }
  __syncthreads();
  float acc[MAXC][EPC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[i][e] = 0.f;
#pragma unroll
  for (int r = 0; r < PF_RMAX; ++r) {
    const int v = wave + PF_NW * r;
    if (v < V) {
      const float w = wl[v], d = dsc[v];
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          o.v[e] = w * dfv[i].v[e] + d * av[i][e];
          acc[i][e] += d * xv[r][i].v[e];
        }
        if (c < nchunk) o.store_stream(dx + ((int64_t)n * V + v) * H + c * EPC);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) psm[wave * H + c * EPC + e] = acc[i][e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += 64 * PF_NW) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < PF_NW; ++w) s += psm[w * H + i];
    atomicAdd(da + i, s);
  }
}

// ------------------------------------------------------------------------------------ gather / segment mean
template <typename T>
__global__ __launch_bounds__(256) void gather_fwd_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx,
                                                         const int32_t* __restrict__ start, const float* __restrict__ scale,
                                                         const float* __restrict__ tok_w, T* __restrict__ out, int n_out, int H) {
  constexpr int EPC = DT<T>::EPC;
  const int nchunk = H / EPC;
  const int64_t total = (int64_t)n_out * nchunk;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / nchunk), c = (int)(t % nchunk);
    Chunk<T> acc;
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc.v[e] = 0.f;
    for (int j = start[i]; j < start[i + 1]; ++j) {
      const int s = idx[j];
      if (s >= 0) {
        Chunk<T> v;
        v.load(src + (int64_t)s * H + c * EPC);
        const float w = tok_w ? tok_w[j] : 1.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc.v[e] += v.v[e] * w;
      }
    }
    const float sc = scale ? scale[i] : 1.f;
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc.v[e] *= sc;
    acc.store(out + (int64_t)i * H + c * EPC);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_bwd_kernel(const T* __restrict__ dout, const int32_t* __restrict__ idx,
                                                         const int32_t* __restrict__ start, const float* __restrict__ scale,
                                                         float* __restrict__ dsrc, int n_out, int H) {
  constexpr int EPC = DT<T>::EPC;
  const int nchunk = H / EPC;
  const int64_t total = (int64_t)n_out * nchunk;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / nchunk), c = (int)(t % nchunk);
    if (start[i] == start[i + 1]) continue;
    Chunk<T> g;
    g.load(dout + (int64_t)i * H + c * EPC);
    const float sc = scale ? scale[i] : 1.f;
    for (int j = start[i]; j < start[i + 1]; ++j) {
      const int s = idx[j];
      if (s >= 0) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) atomicAdd(dsrc + (int64_t)s * H + c * EPC + e, g.v[e] * sc);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ embedding tables
// out[r] = word[ids[r]] (+ type[tids ? tids[r] : 0]) (+ pos[r % L]) — the three lookups and two adds of
// BertEmbeddings (P/model/Bert_backbone.py:98-113; position ids are arange(L)) in one pass; tables are the f32 masters.
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ word, const int64_t* __restrict__ ids,
                                                        const float* __restrict__ type_tab, const int64_t* __restrict__ tids,
                                                        const float* __restrict__ pos_tab, int L, T* __restrict__ out,
                                                        int rows, int H, int vocab, int* __restrict__ err) {
  constexpr int EPC = DT<T>::EPC;
  const int nchunk = H / EPC;
  const int64_t total = (int64_t)rows * nchunk;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(t / nchunk), c = (int)(t % nchunk);
    int64_t id = ids[r];
    if (id < 0 || id >= vocab) {  // torch raises on an out-of-range index; flag it and read row 0
      if (err) atomicOr(err, 1);
      id = 0;
    }
    float acc[EPC];
    {
      const float* w = word + id * H + c * EPC;
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(w + e);
        acc[e] = v[0]; acc[e + 1] = v[1]; acc[e + 2] = v[2]; acc[e + 3] = v[3];
      }
    }
    if (type_tab) {
      const float* w = type_tab + (tids ? tids[r] : 0) * H + c * EPC;
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(w + e);
        acc[e] += v[0]; acc[e + 1] += v[1]; acc[e + 2] += v[2]; acc[e + 3] += v[3];
      }
    }
    if (pos_tab) {
      const float* w = pos_tab + (int64_t)(r % L) * H + c * EPC;
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(w + e);
        acc[e] += v[0]; acc[e + 1] += v[1]; acc[e + 2] += v[2]; acc[e + 3] += v[3];
      }
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = acc[e];
    o.store(out + (int64_t)r * H + c * EPC);
  }
}

// scatter-add of d(out) into the (pre-zeroed) f32 table gradients.  Rows whose id equals the table's
// padding index contribute nothing (nn.Embedding(padding_idx=…) semantics, Bert_backbone.py:85-87 — this
// also zeroes the gradient of position row `pos_pad`, as in the reference).
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ ids,
                                                        const int64_t* __restrict__ tids, int L, float* __restrict__ dword,
                                                        float* __restrict__ dtype_tab, float* __restrict__ dpos, int rows,
                                                        int H, int vocab, int word_pad, int pos_pad) {
  // one wave per row; lane l owns columns l, l+64, ...: every atomic instruction of a wave covers 256 contiguous
  // bytes of one table row (the memory-side atomic units coalesce those; 32-B-strided lanes do not).
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwave = gridDim.x * (blockDim.x >> 6);
  for (int r = wave; r < rows; r += nwave) {
    const int64_t id = ids[r];
    float* dw = (dword && id >= 0 && id < vocab && id != word_pad) ? dword + id * H : nullptr;
    float* dt = (dtype_tab && tids) ? dtype_tab + tids[r] * H : nullptr;
    const int l = r % L;
    float* dp = (dpos && l != pos_pad) ? dpos + (int64_t)l * H : nullptr;
    if (!dw && !dt && !dp) continue;
    const T* g = dout + (int64_t)r * H;
    for (int c = lane; c < H; c += 64) {
      const float v = to_f(g[c]);
      if (dw) atomicAdd(dw + c, v);
      if (dt) atomicAdd(dt + c, v);
      if (dp) atomicAdd(dp + c, v);
    }
  }
}

// Small tables (nav-type, step-id, object-name embeddings: 3..64 rows): every token row lands on a handful of table rows, so the
// per-element atomics of the kernel above serialise on the same words (109 us for 8960 rows into a 3-row table).  Here a block
// owns 64 columns x a chunk of token rows, accumulates in LDS ([row phase][table row][column], one owner per word) and issues one
// atomic per table word and block.
constexpr int ES_MAXV = 64, ES_ROWS = 256;
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_small_kernel(const T* __restrict__ dout, const int64_t* __restrict__ ids,
                                                              float* __restrict__ dtab, int rows, int H, int vocab, int pad_id) {
  extern __shared__ float es[];      // [4][vocab][64]
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  for (int i = threadIdx.x; i < 4 * vocab * 64; i += 256) es[i] = 0.f;
  __syncthreads();
  const int r0 = blockIdx.y * ES_ROWS, r1 = min(rows, r0 + ES_ROWS);
  if (c < H)
    for (int r = r0 + ph; r < r1; r += 4) {
      const int64_t id = ids[r];
      if (id < 0 || id >= vocab || id == pad_id) continue;
      es[(ph * vocab + (int)id) * 64 + lane] += to_f(dout[(int64_t)r * H + c]);
    }
  __syncthreads();
  for (int i = threadIdx.x; i < vocab * 64; i += 256) {
    const int v = i >> 6, l = i & 63;
    const float t = es[(0 * vocab + v) * 64 + l] + es[(1 * vocab + v) * 64 + l] + es[(2 * vocab + v) * 64 + l] + es[(3 * vocab + v) * 64 + l];
    if (t != 0.f && blockIdx.x * 64 + l < H) atomicAdd(dtab + (int64_t)v * H + blockIdx.x * 64 + l, t);
  }
}

// ------------------------------------------------------------------------------------ tr16 probe
__global__ void probe_tr16_kernel(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[64 * 4];
  const int l = threadIdx.x;
  for (int j = 0; j < 4; ++j) sm[l * 4 + j] = (uint16_t)(l * 4 + j);
  __syncthreads();
  typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
  s16x4 t;
  const uint32_t addr = (uint32_t)(uintptr_t)(sm + l * 4);
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)t[j];
}

template <typename T>
int ln_check(int H) {
  const int epc = DT<T>::EPC;
  if (H % epc) return GOAT_E_SHAPE;
  if (H > 64 * MAXC_MAX * epc) return GOAT_E_SHAPE;
  return 0;
}
template <typename T>
int ln_maxc(int H) {  // smallest instantiated chunk count covering H
  const int need = (H / DT<T>::EPC + 63) / 64;
  return need <= 1 ? 1 : need <= 2 ? 2 : need <= 3 ? 3 : need <= 4 ? 4 : 8;
}
#define GOAT_LN_DISPATCH(T_, H_, ...)                        \
  switch (ln_maxc<T_>(H_)) {                                 \
    case 1: { constexpr int MC = 1; __VA_ARGS__; } break;    \
    case 2: { constexpr int MC = 2; __VA_ARGS__; } break;    \
    case 3: { constexpr int MC = 3; __VA_ARGS__; } break;    \
    case 4: { constexpr int MC = 4; __VA_ARGS__; } break;    \
    default: { constexpr int MC = 8; __VA_ARGS__; } break;   \
  }

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int goat_version(void) { return 101; }

extern "C" int goat_ln_fwd_do(void* stream, int dtype, const void* x, const void* residual, const float* gamma,
                              const float* beta, float eps, float p, uint64_t seed, uint64_t offset,
                              const uint64_t* rng_dev, void* y, void* z_out, float* mean, float* rstd, int M, int H, float p_out,
                              uint64_t offset_out, const void* post_add) {
  if (!x || !gamma || !beta || !y || !mean || !rstd) return GOAT_E_ARG;
  if (M <= 0 || !(p_out >= 0.f && p_out < 1.f)) return GOAT_E_SHAPE;
  if ((residual || p > 0.f) && !z_out) return GOAT_E_ARG;
  dim3 grid((M + 3) / 4);
  if (dtype == GOAT_BF16) {
    if (int e = ln_check<bf16_t>(H)) return e;
    GOAT_LN_DISPATCH(bf16_t, H, hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, MC>), grid, dim3(256), 0, ST(stream),
                                                    (const bf16_t*)x, (const bf16_t*)residual, gamma, beta, eps, p, seed,
                                                    offset, rng_dev, (bf16_t*)y, (bf16_t*)z_out, mean, rstd, M, H, p_out, offset_out,
                                                    (const bf16_t*)post_add));
  } else if (dtype == GOAT_F32) {
    if (int e = ln_check<float>(H)) return e;
    GOAT_LN_DISPATCH(float, H, hipLaunchKernelGGL((ln_fwd_kernel<float, MC>), grid, dim3(256), 0, ST(stream),
                                                   (const float*)x, (const float*)residual, gamma, beta, eps, p, seed,
                                                   offset, rng_dev, (float*)y, (float*)z_out, mean, rstd, M, H, p_out, offset_out,
                                                   (const float*)post_add));
  } else {
    return GOAT_E_ARG;
  }
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_ln_fwd(void* stream, int dtype, const void* x, const void* residual, const float* gamma,
                           const float* beta, float eps, float p, uint64_t seed, uint64_t offset,
                           const uint64_t* rng_dev, void* y, void* z_out, float* mean, float* rstd, int M, int H) {
  return goat_ln_fwd_do(stream, dtype, x, residual, gamma, beta, eps, p, seed, offset, rng_dev, y, z_out, mean, rstd, M, H, 0.f, 0,
                        nullptr);
}

#define GOAT_LN_BWD_PARTS 512
#ifndef GOAT_LN_BWD_WAVES
// waves per block of the LayerNorm backward.  Isolated (scripts/ln_bench.py, atomic mode, 3840 rows): 4 -> 20.4 us, 8 -> 15.8, 16 -> 42.  Inside
// the step the kernel runs with deferred column partials (no atomics) and usually NEXT TO a GEMM of the other graph branch whose
// workgroups hold 96-112 KiB of a CU's 160 KiB LDS: an 8-wave block needs 48 KiB for its column reduction and fits beside such a
// workgroup only just or not at all, a 4-wave block (24 KiB) always does.  Same-box A/B of the whole step, three alternations (round 4):
// 8 waves 5.647 / 5.672 / 5.662 ms, 4 waves 5.579 / 5.598 / 5.614 ms; 2 waves (12 KiB, twice the partial rows) is slower again (+0.06 ms).
#define GOAT_LN_BWD_WAVES 4
#endif
#ifndef GOAT_LN_BWD768
#define GOAT_LN_BWD768 1     // bf16 rows of 768: ln_bwd768_kernel (0: the generic kernel; also GOAT_LN_BWD_GENERIC=1 in the environment)
#endif
#ifndef GOAT_LN_RIF
#define GOAT_LN_RIF 2      // rows in flight per wave (bf16); measured: 4 rows / fewer partial blocks are slower (8.52-8.80 vs 8.45 ms/step)
#endif

extern "C" int goat_ln_bwd_ws_floats(int H) { return GOAT_LN_BWD_PARTS * 2 * H; }

extern "C" int goat_ln_bwd_nparts(int M) {       // partial rows written by goat_ln_bwd(..., accumulate = 2)
  int nparts = (M + GOAT_LN_BWD_WAVES * GOAT_LN_RIF - 1) / (GOAT_LN_BWD_WAVES * GOAT_LN_RIF);
  return nparts > GOAT_LN_BWD_PARTS ? GOAT_LN_BWD_PARTS : nparts;
}

extern "C" int goat_ln_reduce_batched(void* stream, const goat_ln_partial* entries, int n, int H) {
  if (!entries || n < 0 || H <= 0) return GOAT_E_ARG;
  // same destination -> adjacent (stable: call order kept inside a run); consecutive launches on one stream are ordered, so a run
  // cut by the 64-entry limit is still summed by one writer at a time
  std::vector<goat_ln_partial> sorted(entries, entries + n);
  std::stable_sort(sorted.begin(), sorted.end(), [](const goat_ln_partial& x, const goat_ln_partial& y) {
    return reinterpret_cast<uintptr_t>(x.dgamma) < reinterpret_cast<uintptr_t>(y.dgamma);
  });
  for (int k = 1; k < n; ++k)
    if (sorted[k].dgamma == sorted[k - 1].dgamma && sorted[k].dbeta != sorted[k - 1].dbeta) return GOAT_E_ARG;
  for (int first = 0; first < n; first += 64) {
    LnReduceArgs a;
    a.n = n - first < 64 ? n - first : 64;
    a.H = H;
    for (int k = 0; k < a.n; ++k) {
      const goat_ln_partial& e = sorted[first + k];
      if (!e.ws || !e.dgamma || !e.dbeta || e.nparts <= 0) return GOAT_E_ARG;
      a.e[k].ws = e.ws; a.e[k].dgamma = e.dgamma; a.e[k].dbeta = e.dbeta; a.e[k].nparts = e.nparts; a.e[k].pad = 0;
    }
    hipLaunchKernelGGL(ln_reduce_batched_kernel, dim3((2 * H + 15) / 16, a.n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    GOAT_LAUNCH_CHECK();
  }
  return 0;
}

namespace {
__global__ __launch_bounds__(256) void zero2_kernel(float* __restrict__ a, float* __restrict__ b, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { a[i] = 0.f; b[i] = 0.f; }
}
}  // namespace

extern "C" int goat_ln_bwd_do(void* stream, int dtype, const void* dy, const void* dy2, const void* z, const float* gamma,
                              const float* mean, const float* rstd, float p, uint64_t seed, uint64_t offset,
                              const uint64_t* rng_dev, void* dx, void* d_res, float* dgamma, float* dbeta, float* ws,
                              int M, int H, int accumulate, const void* dx_add, float p_out, uint64_t offset_out, void* d_post) {
  if (!dy || !z || !gamma || !mean || !rstd || !dgamma || !dbeta) return GOAT_E_ARG;
  if (M <= 0 || !(p_out >= 0.f && p_out < 1.f)) return GOAT_E_SHAPE;
  // deterministic mode (ws): 4-wave blocks, up to 512 per-block partial rows reduced by a second kernel (round 1).
  // atomic mode: GOAT_LN_BWD_WAVES-wave blocks (default 8) so that fewer blocks contend for the 2*H gradient words —
  // 512 four-wave blocks made the kernel 20 us instead of 12 + 5 (profiles/round2_ln_bench.txt)
  // accumulate == 2: the per-block partials stay in ws (goat_ln_bwd_nparts(M) rows of 2*H floats) and the caller reduces them
  // later with goat_ln_reduce_batched — the column reduction was 6.8 of the kernel's 15.8 us at 3840 rows (profiles/round2_ln_bench.txt)
  const int pre_add = (accumulate & 4) ? 1 : 0;      // GOAT_LN_ADD_BEFORE: dx_add joins the gradient of the pre-norm sum (see goat_hip.h)
  accumulate &= 3;
  const bool defer = accumulate == 2;
  if (defer && ws == nullptr) return GOAT_E_ARG;
  const bool det = ws != nullptr && !defer;
  const int nwv = det ? 4 : GOAT_LN_BWD_WAVES;
  int nparts = (M + nwv * GOAT_LN_RIF - 1) / (nwv * GOAT_LN_RIF);   // nwv waves x RIF rows in flight per block
  if (nparts > GOAT_LN_BWD_PARTS) nparts = GOAT_LN_BWD_PARTS;
  const size_t sm = (size_t)nwv * 2 * H * sizeof(float);
  if (sm > 160 * 1024) return GOAT_E_SHAPE;
  if (!det && !defer && !accumulate) {      // atomic mode writes by accumulation: an overwrite clears the two vectors first.
    // A KERNEL, not hipMemsetAsync: captured into a hipGraph (ROCm 7.2) the two memset nodes left every fourth float of the
    // vectors uncleared in a graph with parallel branches (found by tests/test_rollout_gpu.py: garbage in a LayerNorm weight
    // gradient of a replayed episode; GOAT_LN_DETERMINISTIC=1, which does not clear, was clean).
    hipLaunchKernelGGL(zero2_kernel, dim3((H + 255) / 256), dim3(256), 0, ST(stream), dgamma, dbeta, H);
  }
#define GOAT_LN_BWD_LAUNCH(T_, RIF_, NWV_)                                                                                    \
  do {                                                                                                                        \
    auto kern_ = ln_bwd_kernel<T_, MC, (MC > 3 ? 1 : RIF_), NWV_>;   /* rows wider than 1536 (bf16) / 768 (f32) elements: ONE row in   \
                                                                       flight (two spilled 54 .. 760 registers) */                    \
    if (sm > 64 * 1024) {                                                                                                     \
      static bool attr_ = false;                                                                                              \
      if (!attr_) {                                                                                                           \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) \
          return GOAT_E_SHAPE;                                                                                                \
        attr_ = true;                                                                                                         \
      }                                                                                                                       \
    }                                                                                                                         \
    hipLaunchKernelGGL(kern_, dim3(nparts), dim3(64 * NWV_), sm, ST(stream), (const T_*)dy, (const T_*)dy2, (const T_*)z, gamma, mean, \
                       rstd, p, seed, offset, rng_dev, (T_*)dx, (T_*)d_res, ws, dgamma, dbeta, (const T_*)dx_add, pre_add, M, H, \
                       p_out, offset_out, (T_*)d_post);                                                                       \
  } while (0)
  if (dtype == GOAT_BF16 && H == 768 && GOAT_LN_BWD768 && (int64_t)M * H * 2 < (1ll << 31) && !getenv("GOAT_LN_BWD_GENERIC")) {
    // the 96-VGPR form (ln_bwd768_kernel): same grid, same partial rows, same results
    const bool drop = p > 0.f, two = dy2 != nullptr, extra = dx_add != nullptr || p_out > 0.f || d_post != nullptr;
#define GOAT_LN768_LAUNCH(NWV_, DROP_, DY2_, EXTRA_)                                                                          \
    hipLaunchKernelGGL((EXTRA_ ? ln_bwd768x_kernel<NWV_, GOAT_LN_RIF, DROP_, DY2_> : ln_bwd768_kernel<NWV_, GOAT_LN_RIF, DROP_, DY2_>), dim3(nparts), dim3(64 * NWV_), sm + 768 * sizeof(float), ST(stream),          \
                       (const bf16_t*)dy, (const bf16_t*)dy2, (const bf16_t*)z, gamma, mean, rstd, p, seed, offset, rng_dev,  \
                       (bf16_t*)dx, (bf16_t*)d_res, ws, dgamma, dbeta, (const bf16_t*)dx_add, pre_add, M, p_out, offset_out,  \
                       (bf16_t*)d_post)
#define GOAT_LN768_FLAGS(NWV_)                                                       \
    do {                                                                             \
      if (extra) {                                                                   \
        if (drop) { if (two) GOAT_LN768_LAUNCH(NWV_, true, true, true); else GOAT_LN768_LAUNCH(NWV_, true, false, true); }      \
        else { if (two) GOAT_LN768_LAUNCH(NWV_, false, true, true); else GOAT_LN768_LAUNCH(NWV_, false, false, true); }         \
      } else {                                                                       \
        if (drop) { if (two) GOAT_LN768_LAUNCH(NWV_, true, true, false); else GOAT_LN768_LAUNCH(NWV_, true, false, false); }    \
        else { if (two) GOAT_LN768_LAUNCH(NWV_, false, true, false); else GOAT_LN768_LAUNCH(NWV_, false, false, false); }       \
      }                                                                              \
    } while (0)
    if (det) GOAT_LN768_FLAGS(4);
    else GOAT_LN768_FLAGS(GOAT_LN_BWD_WAVES);
#undef GOAT_LN768_FLAGS
#undef GOAT_LN768_LAUNCH
  } else if (dtype == GOAT_BF16) {
    if (int e = ln_check<bf16_t>(H)) return e;
    if (det) { GOAT_LN_DISPATCH(bf16_t, H, GOAT_LN_BWD_LAUNCH(bf16_t, GOAT_LN_RIF, 4)); }
    else { GOAT_LN_DISPATCH(bf16_t, H, GOAT_LN_BWD_LAUNCH(bf16_t, GOAT_LN_RIF, GOAT_LN_BWD_WAVES)); }
  } else if (dtype == GOAT_F32) {
    if (int e = ln_check<float>(H)) return e;
    if (det) { GOAT_LN_DISPATCH(float, H, GOAT_LN_BWD_LAUNCH(float, 2, 4)); }
    else { GOAT_LN_DISPATCH(float, H, GOAT_LN_BWD_LAUNCH(float, 2, GOAT_LN_BWD_WAVES)); }
  } else {
    return GOAT_E_ARG;
  }
#undef GOAT_LN_BWD_LAUNCH
  GOAT_LAUNCH_CHECK();
  if (det) {
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * H + 15) / 16), dim3(256), 0, ST(stream), ws, dgamma, dbeta, nparts,
                       H, accumulate);
    GOAT_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int goat_ln_bwd(void* stream, int dtype, const void* dy, const void* dy2, const void* z, const float* gamma,
                           const float* mean, const float* rstd, float p, uint64_t seed, uint64_t offset,
                           const uint64_t* rng_dev, void* dx, void* d_res, float* dgamma, float* dbeta, float* ws,
                           int M, int H, int accumulate, const void* dx_add) {
  return goat_ln_bwd_do(stream, dtype, dy, dy2, z, gamma, mean, rstd, p, seed, offset, rng_dev, dx, d_res, dgamma, dbeta, ws, M, H,
                        accumulate, dx_add, 0.f, 0, nullptr);
}

extern "C" int goat_dropout_add_fwd(void* stream, int dtype, const void* x, const void* residual, void* y, int64_t n,
                                    float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev) {
  if (!x || !y) return GOAT_E_ARG;
  if (n <= 0) return GOAT_E_SHAPE;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL((dropout_kernel<bf16_t, false>), dim3((int)blocks), dim3(256), 0, ST(stream), (const bf16_t*)x,
                       (const bf16_t*)residual, (bf16_t*)y, n, p, seed, offset, rng_dev);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL((dropout_kernel<float, false>), dim3((int)blocks), dim3(256), 0, ST(stream), (const float*)x,
                       (const float*)residual, (float*)y, n, p, seed, offset, rng_dev);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_dropout_bwd(void* stream, int dtype, const void* dy, void* dx, int64_t n, float p, uint64_t seed,
                                uint64_t offset, const uint64_t* rng_dev) {
  if (!dy || !dx) return GOAT_E_ARG;
  if (n <= 0) return GOAT_E_SHAPE;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL((dropout_kernel<bf16_t, true>), dim3((int)blocks), dim3(256), 0, ST(stream), (const bf16_t*)dy,
                       (const bf16_t*)nullptr, (bf16_t*)dx, n, p, seed, offset, rng_dev);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL((dropout_kernel<float, true>), dim3((int)blocks), dim3(256), 0, ST(stream), (const float*)dy,
                       (const float*)nullptr, (float*)dx, n, p, seed, offset, rng_dev);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_act_bwd(void* stream, int dtype, const void* dy, const void* u, void* dx, int64_t n, int act,
                            float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev) {
  if (!dy || !u || !dx) return GOAT_E_ARG;
  if (n <= 0) return GOAT_E_SHAPE;
  if (act != GOAT_EPI_GELU && act != GOAT_EPI_RELU) return GOAT_E_ARG;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, ST(stream), (const bf16_t*)dy,
                       (const bf16_t*)u, (bf16_t*)dx, n, act, p, seed, offset, rng_dev);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(act_bwd_kernel<float>, dim3((int)blocks), dim3(256), 0, ST(stream), (const float*)dy,
                       (const float*)u, (float*)dx, n, act, p, seed, offset, rng_dev);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_colsum(void* stream, int dtype, const void* x, int64_t ld, int R, int C, float* colsum) {
  if (!x || !colsum) return GOAT_E_ARG;
  if (R <= 0 || C <= 0) return GOAT_E_SHAPE;
  const int cb = (C + 63) / 64;
  int rb = (1024 + cb - 1) / cb;
  if (rb > (R + 31) / 32) rb = (R + 31) / 32;
  if (rb < 1) rb = 1;
  const int rows_per_block = (R + rb - 1) / rb;
  dim3 grid(cb, (R + rows_per_block - 1) / rows_per_block);
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, ST(stream), (const bf16_t*)x, ld, R, C, colsum,
                       rows_per_block);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, ST(stream), (const float*)x, ld, R, C, colsum,
                       rows_per_block);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_ce_fwd(void* stream, const float* logits, int64_t ld, int M, int N, const int64_t* targets,
                           float* loss, float* lse) {
  if (!logits || !targets || !loss || !lse) return GOAT_E_ARG;
  if (M <= 0 || N <= 0 || ld < N || (ld & 3) || (reinterpret_cast<uintptr_t>(logits) & 15)) return GOAT_E_SHAPE;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(M), dim3(256), 0, ST(stream), logits, ld, N, targets, loss, lse);
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_ce_bwd(void* stream, int dtype_out, const float* logits, int64_t ld, int M, int N,
                           const int64_t* targets, const float* lse, const float* dloss, void* dlogits, int64_t ld_out) {
  if (!logits || !targets || !lse || !dloss || !dlogits) return GOAT_E_ARG;
  if (M <= 0 || N <= 0 || ld < N || ld_out < N) return GOAT_E_SHAPE;
  if (dtype_out == GOAT_BF16)
    hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(M), dim3(256), 0, ST(stream), logits, ld, N, targets, lse, dloss,
                       (bf16_t*)dlogits, ld_out);
  else if (dtype_out == GOAT_F32)
    hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(M), dim3(256), 0, ST(stream), logits, ld, N, targets, lse, dloss,
                       (float*)dlogits, ld_out);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_transpose(void* stream, int dtype, const void* in, int64_t ld_in, void* out, int64_t ld_out, int R,
                              int C, float* colsum) {
  if (!in || !out) return GOAT_E_ARG;
  if (R <= 0 || C <= 0 || ld_out < R || ld_in < C) return GOAT_E_SHAPE;
  const int rtiles = (int)((ld_out + 63) / 64), ctiles = (C + 63) / 64;
  int RT = 1;
  while (RT < 16 && (int64_t)((rtiles + 2 * RT - 1) / (2 * RT)) * ctiles >= 1024) RT *= 2;
  dim3 grid((rtiles + RT - 1) / RT, ctiles);
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, ST(stream), (const bf16_t*)in, ld_in, (bf16_t*)out,
                       ld_out, R, C, colsum, RT);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, ST(stream), (const float*)in, ld_in, (float*)out,
                       ld_out, R, C, colsum, RT);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_wgrad_smallk(void* stream, int dtype, const void* dy, int64_t ld_dy, const void* x, int64_t ld_x, int rows,
                                 int N, int K, float* dw, int64_t ld_dw, float* dbias) {
  if (!dy || !x || !dw) return GOAT_E_ARG;
  if (rows <= 0 || N <= 0 || K <= 0 || K > SK_MAXK) return GOAT_E_SHAPE;
  const int rc = rows >= 2048 ? 128 : 32;          // rows per block
  dim3 grid((N + 127) / 128, (rows + rc - 1) / rc);
  // x rows are read as whole 16-byte chunks (columns >= K are multiplied but never stored): rows padded and aligned to a chunk
  const int epc = dtype == GOAT_BF16 ? 8 : 4, kp = (K + epc - 1) / epc * epc;
  if (ld_x < kp || (ld_x % epc) || (reinterpret_cast<uintptr_t>(x) & 15)) return GOAT_E_SHAPE;
#define GOAT_SK_LAUNCH(T_, KP_)                                                                                              \
  do {                                                                                                                       \
    if (rc == 128)                                                                                                           \
      hipLaunchKernelGGL((wgrad_smallk_kernel<T_, KP_, 128>), grid, dim3(256), 0, ST(stream), (const T_*)dy, ld_dy, (const T_*)x, ld_x, \
                         rows, N, K, dw, ld_dw, dbias);                                                                      \
    else                                                                                                                     \
      hipLaunchKernelGGL((wgrad_smallk_kernel<T_, KP_, 32>), grid, dim3(256), 0, ST(stream), (const T_*)dy, ld_dy, (const T_*)x, ld_x, \
                         rows, N, K, dw, ld_dw, dbias);                                                                      \
  } while (0)
  if (dtype == GOAT_BF16) {
    if (kp == 8) GOAT_SK_LAUNCH(bf16_t, 8); else GOAT_SK_LAUNCH(bf16_t, 16);
  } else if (dtype == GOAT_F32) {
    if (kp == 4) GOAT_SK_LAUNCH(float, 4); else if (kp == 8) GOAT_SK_LAUNCH(float, 8);
    else if (kp == 12) GOAT_SK_LAUNCH(float, 12); else GOAT_SK_LAUNCH(float, 16);
  } else {
    return GOAT_E_ARG;
  }
#undef GOAT_SK_LAUNCH
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_pano_fusion_fwd(void* stream, int dtype, const void* x, const float* a, const float* a0, void* fused,
                                    float* wsave, int N, int V, int H) {
  if (!x || !a || !a0 || !fused || !wsave) return GOAT_E_ARG;
  if (N <= 0 || V <= 0 || V > 64 || H <= 0) return GOAT_E_SHAPE;
  size_t sm = (size_t)PF_NW * H * sizeof(float);
#ifdef GOAT_PANO_GENERIC
  sm = 1 << 30;
#endif
  if (dtype == GOAT_BF16) {
    if (H % 8 == 0 && H <= 64 * 8 * 2 && sm <= 64 * 1024)
      hipLaunchKernelGGL((pano_fusion_fwd_kernel<bf16_t, 2>), dim3(N), dim3(64 * PF_NW), sm, ST(stream), (const bf16_t*)x, a, a0,
                         (bf16_t*)fused, wsave, V, H);
    else
      hipLaunchKernelGGL(pano_fusion_fwd_generic<bf16_t>, dim3(N), dim3(256), 0, ST(stream), (const bf16_t*)x, a, a0,
                         (bf16_t*)fused, wsave, V, H);
  } else if (dtype == GOAT_F32) {
    if (H % 4 == 0 && H <= 64 * 4 * 3 && sm <= 64 * 1024)
      hipLaunchKernelGGL((pano_fusion_fwd_kernel<float, 3>), dim3(N), dim3(64 * PF_NW), sm, ST(stream), (const float*)x, a, a0,
                         (float*)fused, wsave, V, H);
    else
      hipLaunchKernelGGL(pano_fusion_fwd_generic<float>, dim3(N), dim3(256), 0, ST(stream), (const float*)x, a, a0,
                         (float*)fused, wsave, V, H);
  } else {
    return GOAT_E_ARG;
  }
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_pano_fusion_bwd(void* stream, int dtype, const void* x, const float* a, const float* a0,
                                    const float* wsave, const void* dfused, void* dx, float* da, float* da0, int N,
                                    int V, int H) {
  if (!x || !a || !a0 || !wsave || !dfused || !dx || !da || !da0) return GOAT_E_ARG;
  if (N <= 0 || V <= 0 || V > 64 || H <= 0) return GOAT_E_SHAPE;
  size_t sm = (size_t)PF_NW * H * sizeof(float);
#ifdef GOAT_PANO_GENERIC
  sm = 1 << 30;
#endif
  if (dtype == GOAT_BF16) {
    if (H % 8 == 0 && H <= 64 * 8 * 2 && sm <= 64 * 1024)
      hipLaunchKernelGGL((pano_fusion_bwd_kernel<bf16_t, 2>), dim3(N), dim3(64 * PF_NW), sm, ST(stream), (const bf16_t*)x, a, a0,
                         wsave, (const bf16_t*)dfused, (bf16_t*)dx, da, da0, V, H);
    else
      hipLaunchKernelGGL(pano_fusion_bwd_generic<bf16_t>, dim3(N), dim3(256), 0, ST(stream), (const bf16_t*)x, a, a0,
                         wsave, (const bf16_t*)dfused, (bf16_t*)dx, da, da0, V, H);
  } else if (dtype == GOAT_F32) {
    if (H % 4 == 0 && H <= 64 * 4 * 3 && sm <= 64 * 1024)
      hipLaunchKernelGGL((pano_fusion_bwd_kernel<float, 3>), dim3(N), dim3(64 * PF_NW), sm, ST(stream), (const float*)x, a, a0, wsave,
                         (const float*)dfused, (float*)dx, da, da0, V, H);
    else
      hipLaunchKernelGGL(pano_fusion_bwd_generic<float>, dim3(N), dim3(256), 0, ST(stream), (const float*)x, a, a0, wsave,
                         (const float*)dfused, (float*)dx, da, da0, V, H);
  } else {
    return GOAT_E_ARG;
  }
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_gather_segmean_fwd(void* stream, int dtype, const void* src, int64_t src_rows, const int32_t* idx,
                                       const int32_t* start, const float* scale, void* out, int n_out, int H, const float* tok_w) {
  if (!src || !idx || !start || !out) return GOAT_E_ARG;
  if (n_out <= 0 || H <= 0 || src_rows <= 0) return GOAT_E_SHAPE;
  const int epc = dtype == GOAT_BF16 ? 8 : 4;
  if (H % epc) return GOAT_E_SHAPE;
  int64_t total = (int64_t)n_out * (H / epc);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(gather_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST(stream), (const bf16_t*)src, idx, start,
                       scale, tok_w, (bf16_t*)out, n_out, H);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(gather_fwd_kernel<float>, dim3(blocks), dim3(256), 0, ST(stream), (const float*)src, idx, start,
                       scale, tok_w, (float*)out, n_out, H);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_gather_segmean_bwd(void* stream, int dtype, const void* dout, const int32_t* idx,
                                       const int32_t* start, const float* scale, float* dsrc32, int n_out, int H) {
  if (!dout || !idx || !start || !dsrc32) return GOAT_E_ARG;
  if (n_out <= 0 || H <= 0) return GOAT_E_SHAPE;
  const int epc = dtype == GOAT_BF16 ? 8 : 4;
  if (H % epc) return GOAT_E_SHAPE;
  int64_t total = (int64_t)n_out * (H / epc);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(gather_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST(stream), (const bf16_t*)dout, idx, start,
                       scale, dsrc32, n_out, H);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(gather_bwd_kernel<float>, dim3(blocks), dim3(256), 0, ST(stream), (const float*)dout, idx, start,
                       scale, dsrc32, n_out, H);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_embed_fwd(void* stream, int dtype, const float* word, const int64_t* ids, const float* type_tab,
                             const int64_t* type_ids, const float* pos_tab, int L, void* out, int rows, int H, int vocab,
                             int* err_flag) {
  if (!word || !ids || !out) return GOAT_E_ARG;
  if (rows <= 0 || H <= 0 || vocab <= 0 || (pos_tab && L <= 0)) return GOAT_E_SHAPE;
  const int epc = dtype == GOAT_BF16 ? 8 : 4;
  if (H % epc) return GOAT_E_SHAPE;
  int64_t total = (int64_t)rows * (H / epc);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST(stream), word, ids, type_tab, type_ids,
                       pos_tab, L, (bf16_t*)out, rows, H, vocab, err_flag);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(embed_fwd_kernel<float>, dim3(blocks), dim3(256), 0, ST(stream), word, ids, type_tab, type_ids,
                       pos_tab, L, (float*)out, rows, H, vocab, err_flag);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_embed_bwd(void* stream, int dtype, const void* dout, const int64_t* ids, const int64_t* type_ids,
                             int L, float* dword, float* dtype_tab, float* dpos, int rows, int H, int vocab, int word_pad,
                             int pos_pad) {
  if (!dout || !ids) return GOAT_E_ARG;
  if (rows <= 0 || H <= 0 || vocab <= 0 || (dpos && L <= 0)) return GOAT_E_SHAPE;
  const int epc = dtype == GOAT_BF16 ? 8 : 4;
  if (H % epc) return GOAT_E_SHAPE;
  static const bool es_off = getenv("GOAT_NO_EMBED_SMALL") != nullptr;      // (diagnostics: A/B)
  if (!es_off && dword && !dtype_tab && !dpos && vocab <= ES_MAXV && rows >= 512) {      // single small table: LDS accumulation per block
    dim3 grid((H + 63) / 64, (rows + ES_ROWS - 1) / ES_ROWS);
    const size_t sm = (size_t)4 * vocab * 64 * sizeof(float);
    if (dtype == GOAT_BF16)
      hipLaunchKernelGGL(embed_bwd_small_kernel<bf16_t>, grid, dim3(256), sm, ST(stream), (const bf16_t*)dout, ids, dword, rows, H, vocab,
                         word_pad);
    else if (dtype == GOAT_F32)
      hipLaunchKernelGGL(embed_bwd_small_kernel<float>, grid, dim3(256), sm, ST(stream), (const float*)dout, ids, dword, rows, H, vocab,
                         word_pad);
    else
      return GOAT_E_ARG;
    GOAT_LAUNCH_CHECK();
    return 0;
  }
  int blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST(stream), (const bf16_t*)dout, ids, type_ids,
                       L, dword, dtype_tab, dpos, rows, H, vocab, word_pad, pos_pad);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(embed_bwd_kernel<float>, dim3(blocks), dim3(256), 0, ST(stream), (const float*)dout, ids, type_ids,
                       L, dword, dtype_tab, dpos, rows, H, vocab, word_pad, pos_pad);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_probe_tr16(void* stream, uint16_t* out) {
  if (!out) return GOAT_E_ARG;
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, ST(stream), out);
  GOAT_LAUNCH_CHECK();
  return 0;
}
