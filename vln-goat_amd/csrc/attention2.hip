// goat_attn2_fwd / goat_attn2_bwd: masked multi-head attention (head_dim 64, bf16) for GOAT's short sequences (<= 256 rows on
// either side), one workgroup per (sample, head).
//
// Why a second set of kernels: the round-1 kernels (attention.hip) run one wave per 32-row tile and fetch their MFMA fragments
// straight from global memory inside the tile loops (16 B per lane from 32 different rows per instruction, K re-read by every
// query tile, Q/dO re-staged per iteration behind a barrier, 2-byte result stores).  On the GOAT shapes they move 1 TB/s:
// 21.7 us forward / 49.4 us backward for the 48 x 12 heads of 80 tokens (profiles/round2_attention_kernels.txt) — latency,
// not bandwidth.  Here every operand of a (sample, head) is brought into LDS ONCE with coalesced 16-byte loads (8 lanes per
// 128-byte row), the tile loops touch only LDS and registers, and results leave through LDS as 16-byte row segments.
//   forward : waves = query tiles.  S^T = K·Q^T (softmax rows lane-local, as in round 1), O^T = V^T·P^T with swapped MFMA roles
//             so that a lane holds 4 consecutive head columns of its query row (8-byte LDS writes).
//   backward: waves = query tiles (dQ role: lane = query) + key tiles (dK/dV role: lane = key); S and dP are recomputed per
//             role (cheaper than a cross-wave reduction); D = rowsum(dO * O) is computed while dO is staged.
// Semantics (additive key mask, optional [B,Lq,Lk] bias with gradient, dropout on the probabilities from the stateless
// counter hash, LSE saved) are exactly those of attention.hip; P/model/Bert_backbone.py:246-290, transformer.py:172-176.
#include "attn_args.hpp"

namespace {

constexpr int HD = 64, NE = 8, LSTR = HD + NE;     // LDS row stride in elements (144 B: conflict-free 16-byte fragment reads)
constexpr int KSTEPS = 4, TSTEPS = 2;
constexpr int TILE = 32 * LSTR;                      // elements per 32-row tile
#ifndef GOAT_ATTN_TIMING     // experiments only: forward kernel writes 8 s_memtime stamps of wave 0 per block behind the LSE array (the caller allocates B*nh*8 extra words)
#define GOAT_ATTN_TIMING 0
#endif
#if GOAT_ATTN_TIMING
#define GOAT_STAMP(i_) do { if (tid == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); reinterpret_cast<uint32_t*>(p.lse + (int64_t)p.B * p.nh * p.Lq)[blockIdx.x * 8 + (i_)] = (uint32_t)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define GOAT_STAMP(i_) do { } while (0)
#endif

__device__ __forceinline__ bf16x8 lds_frag(const bf16_t* tile, int row, int ks, int hi) {
  return *reinterpret_cast<const bf16x8*>(tile + row * LSTR + (ks * 2 + hi) * NE);
}
// B fragment "fixed column, accumulator-pattern rows" of a row-major [row][64] LDS tile: two ds_read_b64_tr_b16
__device__ __forceinline__ bf16x8 bfrag_crow(const bf16_t* lds, int row_base, int step, int dt, int lane) {
  typedef __attribute__((address_space(3))) bf16x4 lds_b4;
  const int g = lane >> 4, t15 = lane & 15;
  const int col = dt * 32 + (g & 1) * 16 + (t15 & 3) * 4;
  const int r0 = row_base + 16 * step + 4 * (g >> 1) + (t15 >> 2);
  bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(lds + r0 * LSTR + col));
  bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(lds + (r0 + 8) * LSTR + col));
  bf16x8 f;
  f[0] = v0[0]; f[1] = v0[1]; f[2] = v0[2]; f[3] = v0[3];
  f[4] = v1[0]; f[5] = v1[1]; f[6] = v1[2]; f[7] = v1[3];
  return f;
}
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int step) {
  bf16x8 f;
#pragma unroll
  for (int e = 0; e < NE; ++e) f[e] = (bf16_t)a[step * NE + e];
  return f;
}
// One operand of a (sample, head): [nrows, 64] head slice in global memory (row stride rs) -> LDS [rows_pad][LSTR], zero beyond nrows
struct StageOp {
  const bf16_t* g;
  bf16_t* lds;
  int64_t rs;
  int nrows, rows_pad;
};
// Cooperative copy of NOP operands.  The 16-byte chunks of all operands form one index space; every lane keeps DEPTH loads in
// flight before the first LDS write (the per-chunk load -> wait -> write loop of the first version cost 12 dependent global
// round trips, 6 100 of the 24 700 cycles of a forward block).
template <int NOP, int DEPTH>
__device__ __forceinline__ void stage_ops(const StageOp (&op)[NOP], int tid, int nthreads) {
  int total = 0;
#pragma unroll
  for (int i = 0; i < NOP; ++i) total += op[i].rows_pad * 8;
  for (int c0 = tid; c0 < total; c0 += nthreads * DEPTH) {
    uint4 v[DEPTH];
    bf16_t* dst[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      // operand / row / chunk of this lane's d-th chunk by selects (ONE load site per chunk: branches around per-operand loads
      // made the compiler reuse the destination registers and wait for every load in turn)
      const int c = c0 + d * nthreads;
      int cl = c;
      const bf16_t* src = op[0].g;
      bool ld = false;
      dst[d] = nullptr;
#pragma unroll
      for (int i = 0; i < NOP; ++i) {
        const int n = op[i].rows_pad * 8;
        const bool in = (c < total) && cl >= 0 && cl < n;
        const int r = cl >> 3, cc = cl & 7;
        src = in ? op[i].g + (int64_t)r * op[i].rs + cc * NE : src;
        dst[d] = in ? op[i].lds + r * LSTR + cc * NE : dst[d];
        ld = in ? (r < op[i].nrows) : ld;
        cl -= n;
      }
      v[d] = uint4{0u, 0u, 0u, 0u};
      if (ld) v[d] = *reinterpret_cast<const uint4*>(src);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (dst[d] != nullptr) *reinterpret_cast<uint4*>(dst[d]) = v[d];
  }
}
// (HeadRng — the dropout bits of one (sample, head) — lives in common.hpp: the streaming bf16 kernels of attention.hip draw the
// same bits, so a forward and a backward pass may be served by different kernel families)
// the 16 mask values of this lane's keys in key tile jt (keys jt*32 + 4*hi + 8*g + {0..3}): four 16-byte LDS reads
__device__ __forceinline__ void load_kmask(const float* kml, int jt, int hi, float (&km)[16]) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(kml + jt * 32 + 4 * hi + 8 * g4);
#pragma unroll
    for (int e = 0; e < 4; ++e) km[4 * g4 + e] = t[e];
  }
}
// a wave's [32 x 64] result, lane = row, acc[dt][r] = column dt*32 + c_row(r, lane): 8-byte packed writes into the wave's own
// LDS tile, then 16-byte row segments to global rows row0 .. row0+31 (< nrows)
__device__ __forceinline__ void store_tile(bf16_t* tile, const f32x16 (&acc)[2], bf16_t* g, int64_t rs, int row0, int nrows, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16_t)acc[dt][4 * q4 + e];
      *reinterpret_cast<bf16x4*>(tile + l31 * LSTR + dt * 32 + 4 * hi + 8 * q4) = o;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same-wave LDS hand-over between lanes
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int idx = c * 64 + lane, r = idx >> 3, cc = idx & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(tile + r * LSTR + cc * NE);
    if (row0 + r < nrows) goat_store_stream(reinterpret_cast<f32x4*>(g + (int64_t)(row0 + r) * rs + cc * NE), *reinterpret_cast<const f32x4*>(&v));
  }
}

// ======================================================================================== forward
template <int NKT>
__global__ __launch_bounds__(256) void attn2_fwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nth = blockDim.x;
  const int hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
  const int nqt = (p.Lq + 31) / 32;
  bf16_t* kl = reinterpret_cast<bf16_t*>(smem);
  bf16_t* vl = kl + NKT * TILE;
  bf16_t* ql = vl + NKT * TILE;
  float* kml = reinterpret_cast<float*>(ql + nqt * TILE);      // additive key mask, -inf beyond Lk

  const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * HD;
  const bf16_t* Kb = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * HD;
  const bf16_t* Vb = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * HD;
  bf16_t* Ob = reinterpret_cast<bf16_t*>(p.Ow) + b * p.o_bs + h * HD;
  GOAT_STAMP(0);
  // (key mask and dropout counter requested before the operands: behind them they were a second / third dependent round trip)
  const float km_pre = tid < p.Lk ? (p.kmask ? p.kmask[(int64_t)b * p.Lk + tid] : 0.f) : -INFINITY;
  const uint64_t rng_bump = p.rng_dev ? *p.rng_dev : 0ull;
  {
    // lane (row group tid / 8, chunk tid % 8) moves its chunk of rows r0, r0 + nth / 8, ...: one 32-bit offset add per load on a
    // wave-uniform base (see attn2_bwd_shared_kernel: the flat chunk -> (operand, row, chunk) mapping of stage_ops cost ~80 instructions
    // per 16-byte load); 4 rows per lane and operand in flight (K, V, Q: 12 loads) before the first LDS write
    constexpr int U = 4;
    const int r0 = tid >> 3, cc8 = (tid & 7) * NE, rstep = nth >> 3;
    const int maxrows = (NKT > nqt ? NKT : nqt) * 32;
    for (int rb = 0; rb < maxrows; rb += U * rstep) {
      uint4 vk[U], vv[U], vq[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = rb + r0 + u * rstep;
        vk[u] = uint4{0u, 0u, 0u, 0u}; vv[u] = uint4{0u, 0u, 0u, 0u}; vq[u] = uint4{0u, 0u, 0u, 0u};
        if (r < p.Lk) {
          vk[u] = *reinterpret_cast<const uint4*>(Kb + (uint32_t)(r * (int)p.k_rs + cc8));
          vv[u] = *reinterpret_cast<const uint4*>(Vb + (uint32_t)(r * (int)p.v_rs + cc8));
        }
        if (r < p.Lq) vq[u] = *reinterpret_cast<const uint4*>(Qb + (uint32_t)(r * (int)p.q_rs + cc8));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = rb + r0 + u * rstep;
        if (r < NKT * 32) {
          *reinterpret_cast<uint4*>(kl + r * LSTR + cc8) = vk[u];
          *reinterpret_cast<uint4*>(vl + r * LSTR + cc8) = vv[u];
        }
        if (r < nqt * 32) *reinterpret_cast<uint4*>(ql + r * LSTR + cc8) = vq[u];
      }
    }
  }
  // (the mask is kept times log2(e): the softmax below runs on exp2 with the scale folded into one fma per element)
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  if (tid < NKT * 32) kml[tid] = km_pre * LOG2E;
  for (int i = tid + nth; i < NKT * 32; i += nth) kml[i] = (i < p.Lk ? (p.kmask ? p.kmask[(int64_t)b * p.Lk + i] : 0.f) : -INFINITY) * LOG2E;
  GOAT_STAMP(1);
  __syncthreads();
  GOAT_STAMP(2);

  const bool drop = p.p > 0.f;
  const uint32_t thr = goat_thr16(p.p);
  const float keep_scale = drop ? 1.f / (1.f - p.p) : 1.f;
  const HeadRng rng(p.seed + rng_bump, p.offset, (uint32_t)blockIdx.x);
  // a wave takes query tiles wave, wave + #waves, ... (one each up to 128 queries; two for the 129..256-row sequences)
  for (int qti = wave; qti < nqt; qti += (nth >> 6)) {
  const int q0 = qti * 32, q = q0 + l31;
  const bool qv = q < p.Lq;
  bf16_t* qt = ql + q0 * LSTR;
  bf16x8 qf[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = lds_frag(qt, l31, ks, hi);

  // S^T tiles: rows = keys (accumulator registers), cols = queries (lanes)
  f32x16 s[NKT];
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[jt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) mma32(s[jt], lds_frag(kl + jt * TILE, l31, ks, hi), qf[ks]);
  }
  GOAT_STAMP(3);
  // scale + additive key mask (registers), optional bias (one uniform branch around independent loads), row max — in the log2 domain
  float m = -INFINITY;
  const float sl2 = p.scale * LOG2E;
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt) {
    float km[16];
    load_kmask(kml, jt, hi, km);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[jt][r] = fmaf(s[jt][r], sl2, km[r]);
  }
  if (p.bias != nullptr) {
    const float* brow = p.bias + ((int64_t)b * p.Lq + (qv ? q : 0)) * p.Lk;
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = jt * 32 + c_row(r, lane);
        s[jt][r] += ((qv && key < p.Lk) ? brow[key] : 0.f) * LOG2E;
      }
  }
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, s[jt][r]);
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
  const float msafe = (m == -INFINITY) ? 0.f : m;
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(s[jt][r] - msafe);
      s[jt][r] = e;
      l += e;
    }
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (qv && hi == 0) p.lse[((int64_t)b * p.nh + h) * p.Lq + q] = (l > 0.f) ? (msafe * LN2 + __logf(l)) : -INFINITY;

  GOAT_STAMP(4);
  // dropout: the keep bits of 4 consecutive keys from 2 pair hashes (3 when q * Lk is odd)
  const uint32_t idx0 = (uint32_t)q * (uint32_t)p.Lk;
  if (drop && (p.Lk & 1) == 0) {       // even Lk: every group of 4 keys starts on a pair boundary — two hashes, no parity select
    const float ik = inv * keep_scale;
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const uint32_t pi = (idx0 + jt * 32 + 4 * hi + 8 * g4) >> 1;
        const uint32_t h0 = rng.pair(pi), h1 = rng.pair(pi + 1);
        s[jt][4 * g4 + 0] = (h0 & 0xFFFFu) >= thr ? s[jt][4 * g4 + 0] * ik : 0.f;
        s[jt][4 * g4 + 1] = (h0 >> 16) >= thr ? s[jt][4 * g4 + 1] * ik : 0.f;
        s[jt][4 * g4 + 2] = (h1 & 0xFFFFu) >= thr ? s[jt][4 * g4 + 2] * ik : 0.f;
        s[jt][4 * g4 + 3] = (h1 >> 16) >= thr ? s[jt][4 * g4 + 3] * ik : 0.f;
      }
  } else if (drop) {
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const uint32_t kb = rng.keep4(idx0 + jt * 32 + 4 * hi + 8 * g4, thr);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[jt][4 * g4 + e] = ((kb >> e) & 1u) ? s[jt][4 * g4 + e] * (inv * keep_scale) : 0.f;
      }
  } else {
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[jt][r] *= inv;
  }

  GOAT_STAMP(5);
  // O^T (d x q) = V^T (d x keys) · P^T (keys x q): lane = query, registers = head columns
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
    for (int st = 0; st < TSTEPS; ++st) {
      const bf16x8 pa = acc_frag(s[jt], st);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) mma32(o[dt], bfrag_crow(vl, jt * 32, st, dt, lane), pa);
    }
  GOAT_STAMP(6);
  store_tile(qt, o, Ob, p.o_rs, q0, p.Lq, lane);          // this query tile's Q rows are dead: nobody else reads them
  GOAT_STAMP(7);
  }
}

// ======================================================================================== backward
template <bool MULTI>      // MULTI: more roles than waves (129..256-row sequences)
__global__ __launch_bounds__(512) void attn2_bwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nth = blockDim.x;
  const int hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
  const int nqt = (p.Lq + 31) / 32, nkt = (p.Lk + 31) / 32;
  bf16_t* ql = reinterpret_cast<bf16_t*>(smem);
  bf16_t* dol = ql + nqt * TILE;
  bf16_t* kl = dol + nqt * TILE;
  bf16_t* vl = kl + nkt * TILE;
  float* Dl = reinterpret_cast<float*>(vl + nkt * TILE);      // D_q = sum_d dO[q,d] O[q,d]
  float* lsel = Dl + nqt * 32;
  float* kml = lsel + nqt * 32;
  // more roles than waves (sequences of 129..256 rows): a wave runs roles wave, wave + #waves, ... and hands its results out
  // through a staging tile of its own (the operand tiles are still being read by the other waves)
  const int nroles = nqt + nkt, nw = nth >> 6;
  constexpr bool multi = MULTI;
  bf16_t* stg = reinterpret_cast<bf16_t*>(kml + nkt * 32) + wave * TILE;

  const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * HD;
  const bf16_t* Kb = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * HD;
  const bf16_t* Vb = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * HD;
  const bf16_t* Ob = reinterpret_cast<const bf16_t*>(p.O) + b * p.o_bs + h * HD;
  const bf16_t* dOb = reinterpret_cast<const bf16_t*>(p.dO) + b * p.do_bs + h * HD;
  {
    const StageOp ops[3] = {{Qb, ql, p.q_rs, p.Lq, nqt * 32}, {Kb, kl, p.k_rs, p.Lk, nkt * 32}, {Vb, vl, p.v_rs, p.Lk, nkt * 32}};
    stage_ops<3, 8>(ops, tid, nth);
  }
  // dO, with D_q on the way: the 8 lanes that move a row's eight 16-byte chunks reduce their partial dot products
  // (4 chunks of dO and of O in flight per lane; every lane of a wave runs the same number of iterations)
  for (int c0 = tid; c0 < nqt * 32 * 8; c0 += nth * 4) {
    uint4 dv[4], ov[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int c = c0 + d * nth, r = c >> 3, cc = c & 7;
      dv[d] = uint4{0u, 0u, 0u, 0u};
      ov[d] = uint4{0u, 0u, 0u, 0u};
      if (c < nqt * 32 * 8 && r < p.Lq) {
        dv[d] = *reinterpret_cast<const uint4*>(dOb + (int64_t)r * p.do_rs + cc * NE);
        ov[d] = *reinterpret_cast<const uint4*>(Ob + (int64_t)r * p.o_rs + cc * NE);
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int c = c0 + d * nth, r = c >> 3, cc = c & 7;
      const bf16x8 d8 = *reinterpret_cast<const bf16x8*>(&dv[d]), o8 = *reinterpret_cast<const bf16x8*>(&ov[d]);
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < NE; ++e) part += (float)d8[e] * (float)o8[e];
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      part += __shfl_xor(part, 4, 64);
      if (c < nqt * 32 * 8) {
        *reinterpret_cast<uint4*>(dol + r * LSTR + cc * NE) = dv[d];
        if (cc == 0) Dl[r] = part;
      }
    }
  }
  for (int i = tid; i < nqt * 32; i += nth) lsel[i] = i < p.Lq ? p.lse[((int64_t)b * p.nh + h) * p.Lq + i] : -INFINITY;
  for (int i = tid; i < nkt * 32; i += nth) kml[i] = i < p.Lk ? (p.kmask ? p.kmask[(int64_t)b * p.Lk + i] : 0.f) : -INFINITY;
  __syncthreads();

  const bool drop = p.p > 0.f;
  const uint32_t thr = goat_thr16(p.p);
  const float keep_scale = drop ? 1.f / (1.f - p.p) : 1.f;
  const HeadRng rng(p.seed + (p.rng_dev ? *p.rng_dev : 0ull), p.offset, (uint32_t)blockIdx.x);
#define GOAT_DQB (reinterpret_cast<bf16_t*>(p.dQ) + b * p.dq_bs + h * HD)      // (formed at the stores: not live across the role bodies)
#define GOAT_DKB (reinterpret_cast<bf16_t*>(p.dK) + b * p.dk_bs + h * HD)
#define GOAT_DVB (reinterpret_cast<bf16_t*>(p.dV) + b * p.dv_bs + h * HD)
  f32x16 ra[2], rb[2];           // results: dQ role uses ra (dQ); dK/dV role ra = dK, rb = dV
  int role = wave;
  for (; role < (MULTI ? nroles : wave + 1); role += (MULTI ? nw : 1)) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ra[dt][r] = 0.f; rb[dt][r] = 0.f; }

  if (role < nqt) {
    // ---- dQ role: lane = query.  S^T = K·Q^T, dP^T = V·dO^T per key tile; dQ^T += K^T·dS^T
    const int q0 = role * 32, q = q0 + l31;
    const bool qv = q < p.Lq;
    bf16x8 qf[KSTEPS], dof[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      qf[ks] = lds_frag(ql + q0 * LSTR, l31, ks, hi);
      dof[ks] = lds_frag(dol + q0 * LSTR, l31, ks, hi);
    }
    const float dsum = Dl[q], lse_q = lsel[q];
    const bool ok = qv && (lse_q != -INFINITY);
    for (int jt = 0; jt < nkt; ++jt) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        mma32(s, lds_frag(kl + jt * TILE, l31, ks, hi), qf[ks]);
        mma32(dp, lds_frag(vl + jt * TILE, l31, ks, hi), dof[ks]);
      }
      // probabilities: exp(S scale + mask (+ bias) - lse); masked / padded keys carry -inf in kml, invalid queries lse = +inf
      float km[16];
      load_kmask(kml, jt, hi, km);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = s[r] * p.scale + km[r];
      if (p.bias != nullptr) {
        const float* brow = p.bias + ((int64_t)b * p.Lq + (qv ? q : 0)) * p.Lk;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = jt * 32 + c_row(r, lane);
          s[r] += (qv && key < p.Lk) ? brow[key] : 0.f;
        }
      }
      f32x16 ds;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const uint32_t kb = drop ? rng.keep4((uint32_t)q * (uint32_t)p.Lk + jt * 32 + 4 * hi + 8 * g4, thr) : 0xFu;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g4 + e;
          const float pr = ok ? __expf(s[r] - lse_q) : 0.f;
          const float keep = ((kb >> e) & 1u) ? keep_scale : 0.f;
          const float d = pr * (dp[r] * keep - dsum);
          ds[r] = d * p.scale;
          s[r] = d;
        }
      }
      if (p.dbias != nullptr && qv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = jt * 32 + c_row(r, lane);
          if (key < p.Lk) atomicAdd(p.dbias + ((int64_t)b * p.Lq + q) * p.Lk + key, s[r]);
        }
      }
#pragma unroll
      for (int st = 0; st < TSTEPS; ++st) {
        const bf16x8 a = acc_frag(ds, st);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma32(ra[dt], bfrag_crow(kl, jt * 32, st, dt, lane), a);
      }
    }
  } else {
    // ---- dK / dV role: lane = key.  S = Q·K^T, dP = dO·V^T per query tile; dV^T += dO^T·Pd, dK^T += Q^T·dS
    const int k0 = (role - nqt) * 32, key = k0 + l31;
    const bool kv = key < p.Lk;
    bf16x8 kf[KSTEPS], vf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      kf[ks] = lds_frag(kl + k0 * LSTR, l31, ks, hi);
      vf[ks] = lds_frag(vl + k0 * LSTR, l31, ks, hi);
    }
    const float kmv = kml[key];
    for (int it = 0; it < nqt; ++it) {
      const int q0 = it * 32;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        mma32(s, lds_frag(ql + q0 * LSTR, l31, ks, hi), kf[ks]);
        mma32(dp, lds_frag(dol + q0 * LSTR, l31, ks, hi), vf[ks]);
      }
      float lq[16], dq[16];                 // lse and D of this lane's 16 queries (q0 + 4*hi + 8*g + {0..3})
      load_kmask(lsel, it, hi, lq);
      load_kmask(Dl, it, hi, dq);
      if (p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = q0 + c_row(r, lane);
          s[r] = s[r] * p.scale + ((q < p.Lq && kv) ? p.bias[((int64_t)b * p.Lq + q) * p.Lk + key] : 0.f);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= p.scale;
      }
      f32x16 pd, ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = q0 + c_row(r, lane);
        // (padded queries: lse = -inf -> exp(+inf) would be inf: guarded; masked keys: kmv = -inf -> 0)
        const float pr = (lq[r] != -INFINITY && kv) ? __expf(s[r] + kmv - lq[r]) : 0.f;
        float keep = 1.f;
        if (drop) keep = rng.keep((uint32_t)q * (uint32_t)p.Lk + key, thr) ? keep_scale : 0.f;
        pd[r] = pr * keep;
        ds[r] = pr * (dp[r] * keep - dq[r]) * p.scale;
      }
#pragma unroll
      for (int st = 0; st < TSTEPS; ++st) {
        const bf16x8 ap = acc_frag(pd, st), as = acc_frag(ds, st);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          mma32(rb[dt], bfrag_crow(dol, q0, st, dt, lane), ap);
          mma32(ra[dt], bfrag_crow(ql, q0, st, dt, lane), as);
        }
      }
    }
  }
  if (!multi) break;      // one role per wave: results leave after the block barrier below, through the operand tiles
  if (role < nqt) {
    store_tile(stg, ra, GOAT_DQB, p.dq_rs, role * 32, p.Lq, lane);
  } else {
    const int kt = role - nqt;
    store_tile(stg, ra, GOAT_DKB, p.dk_rs, kt * 32, p.Lk, lane);
    store_tile(stg, rb, GOAT_DVB, p.dv_rs, kt * 32, p.Lk, lane);
  }
  }
  if (multi) return;
  __syncthreads();      // every wave is done reading the staged operands: their tiles become the result staging areas
  if (wave < nqt) {
    store_tile(ql + wave * TILE, ra, GOAT_DQB, p.dq_rs, wave * 32, p.Lq, lane);
  } else {
    const int kt = wave - nqt;
    store_tile(kl + kt * TILE, ra, GOAT_DKB, p.dk_rs, kt * 32, p.Lk, lane);
    store_tile(vl + kt * TILE, rb, GOAT_DVB, p.dv_rs, kt * 32, p.Lk, lane);
  }
#undef GOAT_DQB
#undef GOAT_DKB
#undef GOAT_DVB
}


// ======================================================================================== backward, shared dS (round 5)
// The kernel above computes S, dP, the exponentials and the dropout bits of every (query tile, key tile) pair TWICE: once in the
// query-major role that accumulates dQ and once in the key-major role that accumulates dK / dV (profiles/round4_step_breakdown.txt:
// 42 us x 17 per step, VALU-bound).  Here every pair is visited once, in the key-major orientation (lane = key, registers = queries):
//   phase 1: wave kt < nkt owns key tile kt, loops over the query tiles, accumulates dK^T and dV^T in registers as before and leaves
//            dS (already scaled, bf16) in an LDS image ds[query][key]; the bias gradient (un-scaled dS) is added from here (lane = key:
//            coalesced atomics);
//   barrier;
//   phase 2: the waves hand dK / dV out through the (dead) V tiles, then take the query tiles in turn: dQ^T[d, q] = sum_k K^T[d, k]
//            dS^T[k, q] — K^T by transposed LDS reads of the staged K rows, dS rows straight from the image (lane = query, 8
//            consecutive keys = one 16-byte read).  4 MFMAs per (query tile, key tile).
// Workgroup = max(nkt, 2) waves (the old kernel: nqt + nkt waves of 245 registers — ONE workgroup per CU, 576 text heads in three
// rounds over the 256 CUs; 3-wave workgroups fit twice).  LDS: the four operands as before + the dS image ((32 nqt) x (32 nkt + 8)
// bf16): 76 KiB for 80 x 80, 46 KiB for 36 x 36, 146 KiB for 160 x 160 (the old kernel needed its multi-role form there).
constexpr int DSP = 8;       // padding of a dS row in elements (16 bytes: row stride 16 mod 128 bytes -> conflict-free 16-byte row reads)
// A fragment "fixed column, 8 consecutive rows": rows row_base + 16*step + 8*hi + {0..7} of column dt*32 + l31
__device__ __forceinline__ bf16x8 bfrag_nrow(const bf16_t* lds, int row_base, int step, int dt, int lane) {
  typedef __attribute__((address_space(3))) bf16x4 lds_b4;
  const int g = lane >> 4, t15 = lane & 15;
  const int col = dt * 32 + (g & 1) * 16 + (t15 & 3) * 4;
  const int r0 = row_base + 16 * step + 8 * (g >> 1) + (t15 >> 2);
  bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(lds + r0 * LSTR + col));
  bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(lds + (r0 + 4) * LSTR + col));
  bf16x8 f;
  f[0] = v0[0]; f[1] = v0[1]; f[2] = v0[2]; f[3] = v0[3];
  f[4] = v1[0]; f[5] = v1[1]; f[6] = v1[2]; f[7] = v1[3];
  return f;
}

#if GOAT_ATTN_TIMING      // (experiments: the backward kernel's stamps go to `dbias` as uint32[B * nh * 8], wave 0 of every block; bias must be NULL)
#define GOAT_BSTAMP(i_) do { if (tid == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); reinterpret_cast<uint32_t*>(p.dbias)[blockIdx.x * 8 + (i_)] = (uint32_t)__builtin_amdgcn_s_memtime(); } } while (0)
#define GOAT_BSTAMP_NW(i_) do { if (tid == 0) reinterpret_cast<uint32_t*>(p.dbias)[blockIdx.x * 8 + (i_)] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)      /* no wait: issue time only */
#else
#define GOAT_BSTAMP(i_) do { } while (0)
#define GOAT_BSTAMP_NW(i_) do { } while (0)
#endif
// DSS_: dS row stride in elements (compile time: the 16 image stores of a pair take immediate offsets); DROP / LKE: dropout on / even Lk
// (uniform branches inside the unrolled element loops cost an s_cbranch each: this kernel is bound by the instructions ONE wave issues)
template <int DSS_, bool DROP, bool LKE>
__global__ __launch_bounds__(512) void attn2_bwd_shared_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nth = blockDim.x;
  const int hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
  const int nqt = (p.Lq + 31) / 32, nkt = (p.Lk + 31) / 32;
  constexpr int DSS = DSS_;                                     // dS row stride in elements (>= 32 nkt + DSP)
  bf16_t* ql = reinterpret_cast<bf16_t*>(smem);
  bf16_t* dol = ql + nqt * TILE;
  bf16_t* kl = dol + nqt * TILE;
  bf16_t* vl = kl + nkt * TILE;
  bf16_t* dsl = vl + nkt * TILE;
  float* Dl = reinterpret_cast<float*>(dsl + nqt * 32 * DSS);   // D_q = sum_d dO[q,d] O[q,d]
  float* lsel = Dl + nqt * 32;
  float* kml = lsel + nqt * 32;

  const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * HD;
  const bf16_t* Kb = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * HD;
  const bf16_t* Vb = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * HD;
  const bf16_t* Ob = reinterpret_cast<const bf16_t*>(p.O) + b * p.o_bs + h * HD;
  const bf16_t* dOb = reinterpret_cast<const bf16_t*>(p.dO) + b * p.do_bs + h * HD;
  GOAT_BSTAMP(0);
  // The small per-row vectors (log-sum-exp, key mask) and the device-side dropout counter are requested FIRST, into registers: behind the
  // operand staging they were a second and third dependent HBM round trip (cycle stamps, profiles/round5_attention_bwd_phases.txt:
  // staging 14.0k of a block's 36.5k cycles).  One value per lane covers up to 64 x #waves rows; longer sequences loop below.
  const int lrow = tid;
  const float lse_pre = (lrow < nqt * 32 && lrow < p.Lq) ? p.lse[((int64_t)b * p.nh + h) * p.Lq + lrow] : -INFINITY;      // (stored below as lse log2(e); +inf for padded / fully masked rows)
  const float km_pre = lrow < p.Lk ? (p.kmask ? p.kmask[(int64_t)b * p.Lk + lrow] : 0.f) : -INFINITY;
  const uint64_t rng_bump = p.rng_dev ? *p.rng_dev : 0ull;
  // Staging.  Lane (row group r0 = tid / 8, 16-byte chunk cc = tid % 8) moves chunk cc of rows r0, r0 + nth / 8, ... of every operand: the
  // chunk is fixed per lane and the row advances by a constant, so a load costs one 32-bit offset add on a wave-uniform base (the
  // first version mapped a flat chunk index over Q | K | V to (operand, row, chunk) per load: ~80 instructions of selects and 64-bit
  // multiplies per 16-byte load — cycle stamps, profiles/round5_attention_bwd_phases.txt: 7.4k cycles before the LAST load was even
  // issued, on an idle chip).  U = 4 rows per lane and operand cover 32 * #waves rows, i.e. K and V always (waves >= key tiles): the
  // 4 + 4 (dO, O) + 12 (Q, K, V) loads of a lane are all in flight before the first LDS write.  Rows past the sequence read as zero.
  constexpr int U = 4;
  const int r0 = tid >> 3, cc8 = (tid & 7) * NE, rstep = nth >> 3;
#define GOAT_ROWS_LOAD(v_, g_, rs_, n_, rbase_)                                                    \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                  \
    const int r = (rbase_) + r0 + u * rstep;                                                       \
    v_[u] = uint4{0u, 0u, 0u, 0u};                                                                 \
    if (r < (n_)) v_[u] = *reinterpret_cast<const uint4*>((g_) + (uint32_t)(r * (int)(rs_) + cc8)); \
  }
#define GOAT_ROWS_PUT(v_, l_, pad_, rbase_)                                                        \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                  \
    const int r = (rbase_) + r0 + u * rstep;                                                       \
    if (r < (pad_)) *reinterpret_cast<uint4*>((l_) + r * LSTR + cc8) = v_[u];                      \
  }
  // dO -> LDS, with D_q = sum_d dO[q, d] O[q, d] on the way: the 8 lanes of a row reduce their partial dot products (DPP, no LDS)
#define GOAT_DO_PUT(rbase_)                                                                        \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                  \
    const int r = (rbase_) + r0 + u * rstep;                                                       \
    const bf16x8 d8 = *reinterpret_cast<const bf16x8*>(&dv[u]), o8 = *reinterpret_cast<const bf16x8*>(&ov[u]); \
    float part = 0.f;                                                                              \
    _Pragma("unroll") for (int e = 0; e < NE; ++e) part += (float)d8[e] * (float)o8[e];            \
    part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0xB1, 0xF, 0xF, false));   /* lane ^ 1 */ \
    part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x4E, 0xF, 0xF, false));   /* lane ^ 2 */ \
    part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x141, 0xF, 0xF, false));  /* row_half_mirror: the other quad of the 8 */ \
    if (r < nqt * 32) {                                                                            \
      *reinterpret_cast<uint4*>(dol + r * LSTR + cc8) = dv[u];                                     \
      if ((tid & 7) == 0) Dl[r] = part;                                                            \
    }                                                                                              \
  }
  {
    uint4 dv[U], ov[U], vq[U], vk[U], vv[U];
    GOAT_ROWS_LOAD(dv, dOb, p.do_rs, p.Lq, 0);
    GOAT_ROWS_LOAD(ov, Ob, p.o_rs, p.Lq, 0);
    GOAT_ROWS_LOAD(vq, Qb, p.q_rs, p.Lq, 0);
    GOAT_ROWS_LOAD(vk, Kb, p.k_rs, p.Lk, 0);
    GOAT_ROWS_LOAD(vv, Vb, p.v_rs, p.Lk, 0);
    GOAT_BSTAMP_NW(7);
    GOAT_DO_PUT(0);
    GOAT_ROWS_PUT(vq, ql, nqt * 32, 0);
    GOAT_ROWS_PUT(vk, kl, nkt * 32, 0);
    GOAT_ROWS_PUT(vv, vl, nkt * 32, 0);
  }
  for (int rb = U * rstep; rb < nqt * 32; rb += U * rstep) {      // (more query tiles than waves: the remaining Q / dO / O rows)
    uint4 dv[U], ov[U], vq[U];
    GOAT_ROWS_LOAD(dv, dOb, p.do_rs, p.Lq, rb);
    GOAT_ROWS_LOAD(ov, Ob, p.o_rs, p.Lq, rb);
    GOAT_ROWS_LOAD(vq, Qb, p.q_rs, p.Lq, rb);
    GOAT_DO_PUT(rb);
    GOAT_ROWS_PUT(vq, ql, nqt * 32, rb);
  }
#undef GOAT_ROWS_LOAD
#undef GOAT_ROWS_PUT
#undef GOAT_DO_PUT
  // lsel = lse log2(e), +inf for padded and fully masked queries; kml = mask log2(e), -inf for masked / padded keys:
  // P = exp2(S scale log2(e) + kml - lsel) is then 0 wherever either says so, without a select
  constexpr float LOG2E = 1.4426950408889634f;
  if (lrow < nqt * 32) lsel[lrow] = lse_pre != -INFINITY ? lse_pre * LOG2E : INFINITY;
  if (lrow < nkt * 32) kml[lrow] = km_pre * LOG2E;
  for (int i = tid + nth; i < nqt * 32; i += nth) {
    const float l = i < p.Lq ? p.lse[((int64_t)b * p.nh + h) * p.Lq + i] : -INFINITY;
    lsel[i] = l != -INFINITY ? l * LOG2E : INFINITY;
  }
  for (int i = tid + nth; i < nkt * 32; i += nth) kml[i] = (i < p.Lk ? (p.kmask ? p.kmask[(int64_t)b * p.Lk + i] : 0.f) : -INFINITY) * LOG2E;
  GOAT_BSTAMP(1);
  __syncthreads();
  GOAT_BSTAMP(2);

  constexpr bool drop = DROP;
  const uint32_t thr = goat_thr16(p.p);
  const float keep_scale = drop ? 1.f / (1.f - p.p) : 1.f;
  const HeadRng rng(p.seed + rng_bump, p.offset, (uint32_t)blockIdx.x);
  f32x16 ra[2], rb[2];           // phase 1: ra = dK^T, rb = dV^T; phase 2: ra = dQ^T
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ra[dt][r] = 0.f; rb[dt][r] = 0.f; }

  if (wave < nkt) {
    // ---- phase 1, key tile `wave`: lane = key.  S = Q·K^T, dP = dO·V^T per query tile; dV^T += dO^T·Pd, dK^T += Q^T·dS; dS -> LDS
    const int k0 = wave * 32, key = k0 + l31;
    const bool kv = key < p.Lk;
    bf16x8 kf[KSTEPS], vf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      kf[ks] = lds_frag(kl + k0 * LSTR, l31, ks, hi);
      vf[ks] = lds_frag(vl + k0 * LSTR, l31, ks, hi);
    }
    const float kmv = kml[key];
    // (bias / its gradient: one 64-bit base per sample + 32-bit element offsets — with 64-bit addresses per register the compiler
    // hoisted sixteen of them out of the query-tile loop and spilled them)
    const float* bias_b = p.bias ? p.bias + (int64_t)b * p.Lq * p.Lk : nullptr;
    float* dbias_b = (p.dbias && !GOAT_ATTN_TIMING) ? p.dbias + (int64_t)b * p.Lq * p.Lk : nullptr;
    // dropout bits: the pair hash of (q * Lk + key) >> 1 serves this lane and its neighbour (key ^ 1) when q * Lk is even: with an even
    // Lk every lane hashes half of its 16 queries and takes the other half from lane ^ 1 (one DPP move instead of a second hash)
    constexpr bool lk_even = LKE;
    const float sl2 = p.scale * LOG2E;
    const uint32_t Lku = (uint32_t)p.Lk;
    const uint32_t hsh = lk_even ? (uint32_t)(key & 1) * 16u : 0u;      // even Lk: the element's half of the pair hash is the key's parity
    for (int it = 0; it < nqt; ++it) {
      const int q0 = it * 32;
      // S and dP of the pair: the first k-step starts from a zero accumulator operand (no 32 v_mov)
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(ql + q0 * LSTR, l31, 0, hi), kf[0], zero16, 0, 0, 0);
      f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(dol + q0 * LSTR, l31, 0, hi), vf[0], zero16, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < KSTEPS; ++ks) {
        mma32(s, lds_frag(ql + q0 * LSTR, l31, ks, hi), kf[ks]);
        mma32(dp, lds_frag(dol + q0 * LSTR, l31, ks, hi), vf[ks]);
      }
      float lq[16], dq[16];                 // lse log2(e) and D of this lane's 16 queries (q0 + 4*hi + 8*g + {0..3})
      load_kmask(lsel, it, hi, lq);
      load_kmask(Dl, it, hi, dq);
      // exponent's additive part per query: c = mask log2(e) - lse log2(e) (+ bias log2(e)); -inf kills the element
#pragma unroll
      for (int r = 0; r < 16; ++r) lq[r] = kmv - lq[r];
      if (p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = q0 + c_row(r, lane);
          lq[r] += ((q < p.Lq && kv) ? bias_b[(uint32_t)(q * p.Lk + key)] : 0.f) * LOG2E;
        }
      }
      // element index of (first query of this lane, key); register r adds ((r / 4) * 8 + r % 4) * Lk — wave-uniform
      const uint32_t ibase = (uint32_t)(q0 + 4 * hi) * Lku + (uint32_t)key;
      bf16x8 apf[TSTEPS], asf[TSTEPS];      // P (dropped) and dS as the MFMA fragments of k-step 0 / 1 (registers r8 / r8 + 8)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        constexpr int dummy = 0;
        (void)dummy;
        const uint32_t o0 = (uint32_t)((r8 >> 2) * 8 + (r8 & 3)) * Lku, o1 = o0 + 16u * Lku;      // (register r8 + 8: queries 16 further)
        // pair hashes of elements (query of register r8, key) and (query of register r8 + 8, key)
        uint32_t h0 = 0, h1 = 0;
        if (drop) {
          if (lk_even) {
            // even lanes hash register r8, odd lanes register r8 + 8 (lane and lane ^ 1 hold the same queries and share the pair index)
            const int odd = lane & 1;
            const uint32_t mine = rng.pair((ibase + (odd ? o1 : o0)) >> 1);
            const uint32_t theirs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, false);      // lane ^ 1 (quad_perm 1,0,3,2)
            h0 = (odd ? theirs : mine) >> hsh;
            h1 = (odd ? mine : theirs) >> hsh;
          } else {
            const uint32_t i0 = ibase + o0, i1 = ibase + o1;
            h0 = rng.pair(i0 >> 1) >> ((i0 & 1u) * 16u);
            h1 = rng.pair(i1 >> 1) >> ((i1 & 1u) * 16u);
          }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int r = r8 + 8 * half;
          const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, lq[r]));
          float keep = 1.f;
          if (drop) keep = ((half ? h1 : h0) & 0xFFFFu) >= thr ? keep_scale : 0.f;
          const float d = pr * (dp[r] * keep - dq[r]);
          s[r] = d;
          apf[half][r8] = (bf16_t)(pr * keep);
          asf[half][r8] = (bf16_t)(d * p.scale);
        }
      }
      if (dbias_b != nullptr && kv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = q0 + c_row(r, lane);
          if (q < p.Lq) atomicAdd(dbias_b + (uint32_t)(q * p.Lk + key), s[r]);
        }
      }
      // dS image: element (query, key) as bf16; the 32 lanes of a half write 64 contiguous bytes of one row
      {
        bf16_t* dl = dsl + (q0 + 4 * hi) * DSS + key;          // (compile-time stride: immediate offsets)
#pragma unroll
        for (int r = 0; r < 16; ++r) dl[((r >> 2) * 8 + (r & 3)) * DSS] = asf[r >> 3][r & 7];
      }
#pragma unroll
      for (int st = 0; st < TSTEPS; ++st) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          mma32(rb[dt], bfrag_crow(dol, q0, st, dt, lane), apf[st]);
          mma32(ra[dt], bfrag_crow(ql, q0, st, dt, lane), asf[st]);
        }
      }
    }
  }
  GOAT_BSTAMP(3);
  __syncthreads();      // the dS image is complete; Q, dO and V are dead from here on
  GOAT_BSTAMP(4);
#define GOAT_DQB (reinterpret_cast<bf16_t*>(p.dQ) + b * p.dq_bs + h * HD)
#define GOAT_DKB (reinterpret_cast<bf16_t*>(p.dK) + b * p.dk_bs + h * HD)
#define GOAT_DVB (reinterpret_cast<bf16_t*>(p.dV) + b * p.dv_bs + h * HD)
  if (wave < nkt) {
    store_tile(vl + wave * TILE, ra, GOAT_DKB, p.dk_rs, wave * 32, p.Lk, lane);
    store_tile(vl + wave * TILE, rb, GOAT_DVB, p.dv_rs, wave * 32, p.Lk, lane);
  }
  GOAT_BSTAMP(5);
  // ---- phase 2, query tiles wave, wave + #waves, ...: lane = query.  dQ^T += K^T · dS^T over all key tiles
  for (int qt = wave; qt < nqt; qt += (nth >> 6)) {
    const int q0 = qt * 32;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) ra[dt][r] = 0.f;
    for (int jt = 0; jt < nkt; ++jt) {
#pragma unroll
      for (int st = 0; st < TSTEPS; ++st) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(dsl + (q0 + l31) * DSS + jt * 32 + st * 16 + hi * 8);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma32(ra[dt], bfrag_nrow(kl, jt * 32, st, dt, lane), a);
      }
    }
    store_tile(ql + qt * TILE, ra, GOAT_DQB, p.dq_rs, q0, p.Lq, lane);
  }
  GOAT_BSTAMP(6);
#undef GOAT_DQB
#undef GOAT_DKB
#undef GOAT_DVB
}

template <typename K>
int set_smem(K kern, size_t bytes, size_t& cur) {
  if (bytes > 64 * 1024 && bytes > cur) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    cur = bytes;
  }
  return 0;
}

template <int NKT>
int launch_fwd(hipStream_t st, const AttnArgs& a) {
  const int nqt = (a.Lq + 31) / 32;
  const size_t sm = (size_t)(2 * NKT + nqt) * TILE * 2 + NKT * 32 * 4;
  static size_t cur = 0;
  if (int e = set_smem(attn2_fwd_kernel<NKT>, sm, cur)) return e;
  if (sm > 160 * 1024) return GOAT_E_SHAPE;
  hipLaunchKernelGGL(attn2_fwd_kernel<NKT>, dim3(a.B * a.nh), dim3(64 * (nqt < 2 ? 2 : (nqt < 4 ? nqt : 4))), sm, st, a);   // (one query tile: a second wave helps staging K / V)
  GOAT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int goat_attn2_fwd(hipStream_t st, const AttnArgs& a) {
  const int nkt = (a.Lk + 31) / 32, nqt = (a.Lq + 31) / 32;
  if (nkt > 8 || nqt > 8) return GOAT_E_SHAPE;
  // 16-byte row segments on every operand and on O
  if ((a.q_rs | a.k_rs | a.v_rs | a.o_rs | a.q_bs | a.k_bs | a.v_bs | a.o_bs) & 7) return GOAT_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(a.Ow) & 15)) return GOAT_E_SHAPE;
  switch (nkt) {
    case 1: return launch_fwd<1>(st, a);
    case 2: return launch_fwd<2>(st, a);
    case 3: return launch_fwd<3>(st, a);
    case 4: return launch_fwd<4>(st, a);
    case 5: return launch_fwd<5>(st, a);
    case 6: return launch_fwd<6>(st, a);
    case 7: return launch_fwd<7>(st, a);
    case 8: return launch_fwd<8>(st, a);
  }
  return GOAT_E_SHAPE;
}

int goat_attn2_bwd(hipStream_t st, const AttnArgs& a) {
  const int nkt = (a.Lk + 31) / 32, nqt = (a.Lq + 31) / 32;
  if (nkt > 8 || nqt > 8) return GOAT_E_SHAPE;
  if ((a.q_rs | a.k_rs | a.v_rs | a.o_rs | a.do_rs | a.dq_rs | a.dk_rs | a.dv_rs | a.q_bs | a.k_bs | a.v_bs | a.o_bs | a.do_bs | a.dq_bs |
       a.dk_bs | a.dv_bs) & 7)
    return GOAT_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(a.O) & 15) || (reinterpret_cast<uintptr_t>(a.dO) & 15) || (reinterpret_cast<uintptr_t>(a.dQ) & 15) ||
      (reinterpret_cast<uintptr_t>(a.dK) & 15) || (reinterpret_cast<uintptr_t>(a.dV) & 15))
    return GOAT_E_SHAPE;
  // GOAT_ATTN_BWD_DUP=1: the round-2 kernel (both roles recompute S / dP) for every problem (A/B experiments)
  static const bool shared_ds = !(getenv("GOAT_ATTN_BWD_DUP") && getenv("GOAT_ATTN_BWD_DUP")[0] == '1');
  if (shared_ds) {
    const int dss = nkt == 1 ? 40 : (nkt == 2 ? 72 : (nkt == 3 ? 104 : (nkt <= 5 ? 168 : 264)));      // dS row stride (elements): 32 nkt + 8 (of the class's largest nkt)
    const size_t sms = (size_t)(2 * nqt + 2 * nkt) * TILE * 2 + (size_t)nqt * 32 * dss * 2 + (size_t)(2 * nqt + nkt) * 32 * 4;
    if (sms <= 160 * 1024) {
      const bool drop = a.p > 0.f, lke = (a.Lk & 1) == 0;
      const dim3 grid(a.B * a.nh), block(64 * (nkt > 2 ? nkt : 2));
#define GOAT_BWD_LAUNCH(DSS_, D_, E_)                                                                   \
  do {                                                                                                  \
    static size_t cur_s = 0;                                                                            \
    if (int e = set_smem(attn2_bwd_shared_kernel<DSS_, D_, E_>, sms, cur_s)) return e;                  \
    hipLaunchKernelGGL((attn2_bwd_shared_kernel<DSS_, D_, E_>), grid, block, sms, st, a);               \
  } while (0)
#define GOAT_BWD_DE(DSS_)                                                                               \
  do {                                                                                                  \
    if (drop) { if (lke) GOAT_BWD_LAUNCH(DSS_, true, true); else GOAT_BWD_LAUNCH(DSS_, true, false); }  \
    else GOAT_BWD_LAUNCH(DSS_, false, true);              /* (no dropout: the parity flag is unused) */  \
  } while (0)
      if (dss == 40) GOAT_BWD_DE(40);
      else if (dss == 72) GOAT_BWD_DE(72);
      else if (dss == 104) GOAT_BWD_DE(104);
      else if (dss == 168) GOAT_BWD_DE(168);
      else GOAT_BWD_DE(264);
#undef GOAT_BWD_DE
#undef GOAT_BWD_LAUNCH
      GOAT_LAUNCH_CHECK();
      return 0;
    }
  }
  int nwv = nqt + nkt <= 8 ? nqt + nkt : 8;
  const size_t sm0 = (size_t)(2 * nqt + 2 * nkt) * TILE * 2 + (size_t)(2 * nqt + nkt) * 32 * 4;
  size_t sm = sm0 + (nqt + nkt > 8 ? (size_t)nwv * TILE * 2 : 0);
  // more roles than waves: every wave owns a staging tile, so fewer waves need less LDS.  200 x 200 (the 200-token instructions of the
  // fine-tuning episodes: 7 x 7 tiles) is 164.6 KiB on eight waves and 155.6 KiB on six — three rounds of roles instead of two, against
  // the streaming kernels of attention.hip it used to fall back to (48 + 39 us per call; profiles/round5_attention_L200_bwd.txt)
  static const int min_waves = getenv("GOAT_ATTN_BWD_MIN_WAVES") ? atoi(getenv("GOAT_ATTN_BWD_MIN_WAVES")) : 6;      // (8: the behaviour before, for A/B)
  while (sm > 160 * 1024 && nqt + nkt > 8 && nwv > min_waves) {
    --nwv;
    sm = sm0 + (size_t)nwv * TILE * 2;
  }
  if (sm > 160 * 1024) return GOAT_E_SHAPE;          // (e.g. 256 x 256: the caller falls back to the streaming kernels of attention.hip)
  static size_t cur = 0, cur_m = 0;
  if (nqt + nkt > 8) {
    if (int e = set_smem(attn2_bwd_kernel<true>, sm, cur_m)) return e;
    hipLaunchKernelGGL(attn2_bwd_kernel<true>, dim3(a.B * a.nh), dim3(64 * nwv), sm, st, a);
  } else {
    if (int e = set_smem(attn2_bwd_kernel<false>, sm, cur)) return e;
    hipLaunchKernelGGL(attn2_bwd_kernel<false>, dim3(a.B * a.nh), dim3(64 * nwv), sm, st, a);
  }
  GOAT_LAUNCH_CHECK();
  return 0;
}
