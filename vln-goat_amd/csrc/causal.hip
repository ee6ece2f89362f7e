// Small row kernels of GOAT's causal-learning heads (HBM/latency-bound, a few MB per call):
//   * tanh-attention pooling of the CFP heads                (P/model/pretrain_goat.py:502-515, M/models/vilmodel_GOAT.py:909-922)
//   * the "door" gate of BACL type_2 / FACL                  (P/model/vilmodel_goat.py:137-143, M/models/vilmodel_GOAT.py:147-153,548-552)
//   * probability-weighted dictionary sums of BACL type_1    (P/model/vilmodel_goat.py:115-118, M/models/vilmodel_GOAT.py:246-249)
// One wave64 per row / one block per sample; f32 arithmetic; parameter gradients by block partials + one atomic per column.
#include "common.hpp"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------------------------ tanh-attention pooling
// a = softmax_l(tanh(x_l)·w) over ALL L slots (no padding mask, as the reference); out = tanh(sum_l a_l x_l)
// Two launches per direction so that B*L rows / B*H/256 column slabs (not just B samples) are in flight:
//   fwd  (1) score[b,l] = tanh(x_l)·w            one wave per row
//        (2) per (b, 256-column slab): softmax over the L scores, out[cols] = tanh(sum_l a_l x[l,cols])
//   bwd  (1) da[b,l] = x_l·du,  du = dout (1 - out^2)   one wave per row
//        (2) per (b, slab): ds = a (da - a·da); dx[l,cols] = a_l du + ds_l w (1 - tanh^2 x); dw[cols] += sum_l ds_l tanh x
constexpr int POOL_MAXL = 256;
template <typename T>
__global__ __launch_bounds__(256) void pool_score_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ out, const float* __restrict__ dout,
                                                         float* __restrict__ score, int rows, int L, int H,
                                                         const float* __restrict__ smask = nullptr) {
  // forward (out == nullptr): score = tanh(x)·w ; backward: score = x·(dout (1 - out^2))
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * H;
  const int b = row / L;
  float s = 0.f;
  if (out == nullptr) {
    for (int i = lane; i < H; i += 64) s += tanhf(to_f(xr[i])) * w[i];
  } else {
    for (int i = lane; i < H; i += 64) {
      const float o = out[(int64_t)b * H + i];
      s += to_f(xr[i]) * dout[(int64_t)b * H + i] * (1.f - o * o);
    }
  }
  s = wave_sum(s);
  if (lane == 0) score[row] = smask != nullptr ? s + smask[row] : s;      // (-inf: a slot outside the batch's own padded width)
}

// block = 256 threads = 64 column quads x 4 row lanes; grid = (B, ceil(H/256))
template <typename T>
__global__ __launch_bounds__(256) void pool_out_kernel(const T* __restrict__ x, const float* __restrict__ score,
                                                       float* __restrict__ out, float* __restrict__ attn, int L, int H) {
  __shared__ float a[POOL_MAXL];
  __shared__ float red[4];
  __shared__ float part[4][4][64];
  typedef T quad __attribute__((ext_vector_type(4)));
  const int b = blockIdx.x, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float v = threadIdx.x < L ? score[(int64_t)b * L + threadIdx.x] : -INFINITY;
  const float m = block_max(v, red);
  const float e = threadIdx.x < L ? __expf(v - m) : 0.f;
  const float tot = block_sum(e, red);
  if (threadIdx.x < L) {
    a[threadIdx.x] = e / tot;
    if (blockIdx.y == 0) attn[(int64_t)b * L + threadIdx.x] = e / tot;
  }
  __syncthreads();
  const int c = blockIdx.y * 256 + tx * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < H)
    for (int l = ty; l < L; l += 4) {
      const quad q = *reinterpret_cast<const quad*>(x + ((int64_t)b * L + l) * H + c);
      const float al = a[l];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += al * to_f(q[k]);
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) part[ty][k][tx] = acc[k];
  __syncthreads();
  // thread (ty, tx) finishes column c + ty of quad tx
  const float sum = part[0][ty][tx] + part[1][ty][tx] + part[2][ty][tx] + part[3][ty][tx];
  if (c < H) out[(int64_t)b * H + c + ty] = tanhf(sum);
}

template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ attn, const float* __restrict__ out,
                                                       const float* __restrict__ dout, const float* __restrict__ da,
                                                       T* __restrict__ dx, float* __restrict__ dw, int L, int H) {
  __shared__ float al[POOL_MAXL], ds[POOL_MAXL];
  __shared__ float red[4];
  __shared__ float part[4][4][64];
  typedef T quad __attribute__((ext_vector_type(4)));
  const int b = blockIdx.x, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float a = threadIdx.x < L ? attn[(int64_t)b * L + threadIdx.x] : 0.f;
  const float d = threadIdx.x < L ? da[(int64_t)b * L + threadIdx.x] : 0.f;
  const float dot = block_sum(a * d, red);
  if (threadIdx.x < L) { al[threadIdx.x] = a; ds[threadIdx.x] = a * (d - dot); }
  __syncthreads();
  const int c = blockIdx.y * 256 + tx * 4;
  float dwp[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < H) {
    float du[4], wv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float o = out[(int64_t)b * H + c + k];
      du[k] = dout[(int64_t)b * H + c + k] * (1.f - o * o);
      wv[k] = w[c + k];
    }
    for (int l = ty; l < L; l += 4) {
      const int64_t base = ((int64_t)b * L + l) * H + c;
      const quad q = *reinterpret_cast<const quad*>(x + base);
      quad o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float t = tanhf(to_f(q[k]));
        o[k] = from_f<T>(al[l] * du[k] + ds[l] * wv[k] * (1.f - t * t));
        dwp[k] += ds[l] * t;
      }
      *reinterpret_cast<quad*>(dx + base) = o;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) part[ty][k][tx] = dwp[k];
  __syncthreads();
  const float sum = part[0][ty][tx] + part[1][ty][tx] + part[2][ty][tx] + part[3][ty][tx];
  if (c < H) atomicAdd(dw + c + ty, sum);
}

// ------------------------------------------------------------------------------------ door gate
// s = sigmoid(aug·wa + ba + ori·wo + bo) ; out = s*aug + (1-s)*ori       (one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void door_fwd_kernel(const T* __restrict__ aug, const T* __restrict__ ori,
                                                       const float* __restrict__ wa, const float* __restrict__ wo,
                                                       const float* __restrict__ ba, const float* __restrict__ bo,
                                                       T* __restrict__ out, float* __restrict__ gate, int rows, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* a = aug + (int64_t)row * H;
  const T* o = ori + (int64_t)row * H;
  float g = 0.f;
  for (int i = lane; i < H; i += 64) g += to_f(a[i]) * wa[i] + to_f(o[i]) * wo[i];
  g = wave_sum(g) + ba[0] + bo[0];
  const float s = 1.f / (1.f + __expf(-g));
  if (lane == 0) gate[row] = s;
  for (int i = lane; i < H; i += 64) out[(int64_t)row * H + i] = from_f<T>(s * to_f(a[i]) + (1.f - s) * to_f(o[i]));
}

// dgate = dout·(aug-ori) ; dpre = dgate s (1-s) ; daug = s dout + dpre wa ; dori = (1-s) dout + dpre wo ;
// dwa += sum_rows dpre aug ; dwo += sum_rows dpre ori ; dbias += sum_rows dpre   (the two biases share it)
constexpr int DOOR_MAXC = 16;   // columns per lane kept in registers: H <= 64*16
template <typename T>
__global__ __launch_bounds__(256) void door_bwd_kernel(const T* __restrict__ aug, const T* __restrict__ ori,
                                                       const float* __restrict__ wa, const float* __restrict__ wo,
                                                       const float* __restrict__ gate, const T* __restrict__ dout,
                                                       T* __restrict__ daug, T* __restrict__ dori, float* __restrict__ dwa,
                                                       float* __restrict__ dwo, float* __restrict__ dbias, float* __restrict__ dbias2,
                                                       int rows, int H) {
  extern __shared__ float sm[];   // [4 waves][2][H]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pa[DOOR_MAXC], po[DOOR_MAXC], pb = 0.f;
#pragma unroll
  for (int j = 0; j < DOOR_MAXC; ++j) { pa[j] = 0.f; po[j] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const T* a = aug + (int64_t)row * H;
    const T* o = ori + (int64_t)row * H;
    const T* d = dout + (int64_t)row * H;
    const float s = gate[row];
    float dg = 0.f;
    for (int i = lane; i < H; i += 64) dg += to_f(d[i]) * (to_f(a[i]) - to_f(o[i]));
    dg = wave_sum(dg);
    const float dpre = dg * s * (1.f - s);
    pb += dpre;
#pragma unroll
    for (int j = 0; j < DOOR_MAXC; ++j) {
      const int i = lane + 64 * j;
      if (i < H) {
        const float dv = to_f(d[i]), av = to_f(a[i]), ov = to_f(o[i]);
        daug[(int64_t)row * H + i] = from_f<T>(s * dv + dpre * wa[i]);
        dori[(int64_t)row * H + i] = from_f<T>((1.f - s) * dv + dpre * wo[i]);
        pa[j] += dpre * av;
        po[j] += dpre * ov;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < DOOR_MAXC; ++j) {
    const int i = lane + 64 * j;
    if (i < H) { sm[(wave * 2 + 0) * H + i] = pa[j]; sm[(wave * 2 + 1) * H + i] = po[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * H; i += 256) {
    const int which = i / H, col = i % H;
    const float v = sm[(0 * 2 + which) * H + col] + sm[(1 * 2 + which) * H + col] + sm[(2 * 2 + which) * H + col] +
                    sm[(3 * 2 + which) * H + col];
    atomicAdd((which ? dwo : dwa) + col, v);
  }
  if (lane == 0) {                       // (each wave's rows; lanes hold the same value after wave_sum)
    atomicAdd(dbias, pb);
    if (dbias2) atomicAdd(dbias2, pb);   // the second gate bias receives the same gradient (its own arena slice)
  }
}

// ------------------------------------------------------------------------------------ weighted dictionary sum
// out[b,:] = sum_k p[b,k] z[b,k,:]
template <typename T>
__global__ __launch_bounds__(256) void dict_wsum_fwd_kernel(const float* __restrict__ z, const float* __restrict__ p,
                                                            T* __restrict__ out, int K, int H) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < H; i += 256) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += p[(int64_t)b * K + k] * z[((int64_t)b * K + k) * H + i];
    out[(int64_t)b * H + i] = from_f<T>(s);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void dict_wsum_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ z,
                                                            const float* __restrict__ p, float* __restrict__ dz,
                                                            float* __restrict__ dp, int K, int H) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* d = dout + (int64_t)b * H;
  if (dz)
    for (int k = 0; k < K; ++k) {
      const float pk = p[(int64_t)b * K + k];
      for (int i = threadIdx.x; i < H; i += 256) dz[((int64_t)b * K + k) * H + i] = pk * to_f(d[i]);
    }
  if (dp)
    for (int k = wave; k < K; k += 4) {
      float s = 0.f;
      for (int i = lane; i < H; i += 64) s += z[((int64_t)b * K + k) * H + i] * to_f(d[i]);
      s = wave_sum(s);
      if (lane == 0) dp[(int64_t)b * K + k] = s;
    }
}


// ------------------------------------------------------------------------------------ CFP: 3 x symmetric InfoNCE
// loss_i = sum over x in {gmap, vp, fused} of 1/2 [ CE(x_loc[i] . txt_all^T / tau, t_i) + CE(txt_loc[i] . x_all^T / tau, t_i) ],
// t_i = target0 + i  (P/model/pretrain_goat.py:519-534; `all` = the rows of every data-parallel rank, = `loc` on one rank).
// Six (pair, direction) similarity problems of [Bl x Ba x H]: the reference issues ~70 ATen kernels for them and their
// gradients (18 mm, 19 div, 6 log_softmax ...: 0.7 ms of a CFP step); here one forward and one backward launch.
//   forward : block i: for each of the six problems S[j] = <A_pd[i], B_pd[j]> / tau for all j (one wave per j, lanes over H),
//             softmax over j saved to prob[pd][i][:]; loss[i] += sum_pd 0.5 * (lse - S[t_i])  — one writer per sample.
//   backward: every output tensor element has ONE writer that sums its contributions in a fixed order (no atomics: in a bf16
//             chain a 1e-7 summation-order difference is amplified to ~1e-3 by the roundings downstream, and the captured /
//             phased steps must reproduce the eager step).  G_pd = (prob_pd - onehot) * dloss_i / (2 tau);
//             dA_pd[i] = sum_j G[i,j] B_pd[j],  dB_pd[j] = sum_i G[i,j] A_pd[i].  An output "role" is (tensor, its terms): pointers
//             that alias (loc == all on one rank) are merged into one role on the host.
struct NceArgs {
  const float* a[6];      // [Bl, H] operand whose rows are the samples of this rank
  const float* b[6];      // [Ba, H] operand holding the candidates
  float* da[6];
  float* db[6];
  float* prob;            // [6, Bl, Ba]
  float* loss;            // [Bl]
  const float* dloss;     // [Bl]
  int Bl, Ba, H, target0;
  float inv_tau;
};

struct NceRole { float* out; int rows; int n; int pd[6]; int is_b[6]; };
struct NceBwdArgs { NceArgs p; NceRole role[8]; int nroles; int row_groups; };

__global__ __launch_bounds__(256) void infonce_fwd_kernel(NceArgs p) {
  extern __shared__ float sh[];           // [Ba] similarities
  __shared__ float lsum;
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) lsum = 0.f;
  for (int pd = 0; pd < 6; ++pd) {
    const float* arow = p.a[pd] + (int64_t)i * p.H;
    __syncthreads();
    for (int j = wave; j < p.Ba; j += 4) {
      const float* brow = p.b[pd] + (int64_t)j * p.H;
      float s = 0.f;
      for (int k = lane * 4; k < p.H; k += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(arow + k), y = *reinterpret_cast<const f32x4*>(brow + k);
        s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
      }
      s = wave_sum(s);
      if (lane == 0) sh[j] = s * p.inv_tau;
    }
    __syncthreads();
    if (wave == 0) {
      float m = -INFINITY;
      for (int j = lane; j < p.Ba; j += 64) m = fmaxf(m, sh[j]);
      m = wave_max(m);
      float l = 0.f;
      for (int j = lane; j < p.Ba; j += 64) l += __expf(sh[j] - m);
      l = wave_sum(l);
      const float lse = m + __logf(l);
      float* pr = p.prob + ((int64_t)pd * p.Bl + i) * p.Ba;
      for (int j = lane; j < p.Ba; j += 64) pr[j] = __expf(sh[j] - lse);
      if (lane == 0) lsum += 0.5f * (lse - sh[p.target0 + i]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) p.loss[i] += lsum;
}

// grid (role * row_groups + row group of 4 rows, 64-column tile); thread = (row in group, column)
__global__ __launch_bounds__(256) void infonce_bwd_kernel(NceBwdArgs q) {
  const NceArgs& p = q.p;
  const int ro = blockIdx.x / q.row_groups, rg = blockIdx.x % q.row_groups;
  const NceRole& role = q.role[ro];
  const int r = rg * 4 + (threadIdx.x >> 6), c = blockIdx.y * 64 + (threadIdx.x & 63);
  if (r >= role.rows || c >= p.H) return;
  const float half_tau = 0.5f * p.inv_tau;
  float tot = 0.f;
  for (int t = 0; t < role.n; ++t) {
    const int pd = role.pd[t];
    const float* pr = p.prob + (int64_t)pd * p.Bl * p.Ba;
    float acc = 0.f;
    if (!role.is_b[t]) {            // dA[r, c] = g_r * sum_j (P[r, j] - [j == t_r]) B[j, c]
      if (r < p.Bl) {
        const float* B = p.b[pd];
        const int tr = p.target0 + r;
        for (int j = 0; j < p.Ba; ++j) acc += (pr[(int64_t)r * p.Ba + j] - (j == tr ? 1.f : 0.f)) * B[(int64_t)j * p.H + c];
        acc *= p.dloss[r] * half_tau;
      }
    } else if (r < p.Ba) {          // dB[r, c] = sum_i g_i (P[i, r] - [r == t_i]) A[i, c]
      const float* A = p.a[pd];
      for (int i = 0; i < p.Bl; ++i)
        acc += (pr[(int64_t)i * p.Ba + r] - (r == p.target0 + i ? 1.f : 0.f)) * (p.dloss[i] * half_tau) * A[(int64_t)i * p.H + c];
    }
    tot += acc;
  }
  role.out[(int64_t)r * p.H + c] += tot;
}


// ------------------------------------------------------------------------------------ SAP logits / loss
// The tail of the single-action-prediction head (P/model/pretrain_goat.py:375-413; M/models/vilmodel_GOAT.py:803-839), one wave per
// sample.  With gs / ls the raw scores of the global / local heads and fw the fusion weight (sigmoid of fwl, or 0.5):
//   gl[g]    = masked_g ? -inf : gs[g] * fw               masked_g = visited | beyond the map (gvalid == 0 / g >= glens)
//   ll[w]    = masked_w ? -inf : ls[w] * (1 - fw)
//   fused[g] = gl[g] + sum_w M[g,w] * (masked_w ? 0 : ls[w] * (1 - fw))  (+ ll[0] on g == 0 when add_stop: the fine-tuning model)
//   loss     = CE(gl, ga) + CE(ll, la) + CE(fused, ga)    (labels given; a negative label contributes 0)
// The reference spells this as ~25 elementwise / masked_fill / bmm / log_softmax launches on [B, 22..64] tensors and twice as many
// in the backward pass; each masked_fill(...) is a clone (a memcpy node of the step graph) plus a kernel.
struct SapArgs {
  const void* gs; const void* ls; const void* fwl;           // scores [B,G] / [B,W], fusion logit [B] (nullptr: fw = 0.5)
  const uint8_t* gvis; const uint8_t* gvalid; const int64_t* glens; const uint8_t* lmask;
  const float* M; const int64_t* ga; const int64_t* la;
  float* gl; float* ll; float* fused; float* loss; float* lse;   // lse: [B,3]
  const float* dloss; const float* dgl; const float* dll; const float* dfused;
  void* dgs; void* dls; void* dfwl;
  int B, G, W, lmask_is_valid, add_stop, fw_sigmoid;
};

__device__ __forceinline__ bool sap_gmasked(const SapArgs& p, int b, int g) {
  bool m = false;
  if (p.gvis) m = m || p.gvis[(int64_t)b * p.G + g] != 0;
  if (p.gvalid) m = m || p.gvalid[(int64_t)b * p.G + g] == 0;
  if (p.glens) m = m || g >= (int)p.glens[b];
  return m;
}
__device__ __forceinline__ bool sap_lmasked(const SapArgs& p, int b, int w) {
  if (!p.lmask) return false;
  const bool v = p.lmask[(int64_t)b * p.W + w] != 0;
  return p.lmask_is_valid ? !v : v;
}
template <typename T>
__device__ __forceinline__ float sap_fw(const SapArgs& p, int b) {
  if (!p.fwl) return 0.5f;
  const float x = (float)reinterpret_cast<const T*>(p.fwl)[b];
  return p.fw_sigmoid ? 1.f / (1.f + __expf(-x)) : x;
}
// log-sum-exp of n values held one-per-(lane, iteration) in LDS row v (−inf entries allowed; all −inf -> −inf)
__device__ __forceinline__ float sap_lse(const float* v, int n, int lane) {
  float m = -INFINITY;
  for (int i = lane; i < n; i += 64) m = fmaxf(m, v[i]);
  m = wave_max(m);
  if (m == -INFINITY) return -INFINITY;
  float s = 0.f;
  for (int i = lane; i < n; i += 64) s += __expf(v[i] - m);
  return m + __logf(wave_sum(s));
}

template <typename T>
__global__ __launch_bounds__(64) void sap_fwd_kernel(SapArgs p) {
  extern __shared__ float sm[];              // [G] gl | [W] ll | [W] ll zero-filled | [G] fused
  float* s_gl = sm; float* s_ll = sm + p.G; float* s_lz = s_ll + p.W; float* s_fu = s_lz + p.W;
  const int b = blockIdx.x, lane = threadIdx.x;
  const float fw = sap_fw<T>(p, b);
  const T* gs = reinterpret_cast<const T*>(p.gs) + (int64_t)b * p.G;
  const T* ls = reinterpret_cast<const T*>(p.ls) + (int64_t)b * p.W;
  for (int g = lane; g < p.G; g += 64) {
    const float v = sap_gmasked(p, b, g) ? -INFINITY : (float)gs[g] * fw;
    s_gl[g] = v;
    p.gl[(int64_t)b * p.G + g] = v;
  }
  for (int w = lane; w < p.W; w += 64) {
    const bool m = sap_lmasked(p, b, w);
    const float r = (float)ls[w] * (1.f - fw);
    s_ll[w] = m ? -INFINITY : r;
    s_lz[w] = m ? 0.f : r;
    p.ll[(int64_t)b * p.W + w] = s_ll[w];
  }
  __syncthreads();
  for (int g = lane; g < p.G; g += 64) {
    float acc = 0.f;
    if (p.M) {
      const float* mr = p.M + ((int64_t)b * p.G + g) * p.W;
      for (int w = 0; w < p.W; ++w) acc += mr[w] * s_lz[w];
    }
    float v = s_gl[g] + acc;
    if (p.add_stop && g == 0) v += s_ll[0];
    s_fu[g] = v;
    p.fused[(int64_t)b * p.G + g] = v;
  }
  __syncthreads();
  if (p.lse) {
    const float lg = sap_lse(s_gl, p.G, lane), lw = sap_lse(s_ll, p.W, lane), lf = sap_lse(s_fu, p.G, lane);
    if (lane == 0) {
      p.lse[b * 3] = lg; p.lse[b * 3 + 1] = lw; p.lse[b * 3 + 2] = lf;
      if (p.loss) {
        const int ga = p.ga ? (int)p.ga[b] : -1, la = p.la ? (int)p.la[b] : -1;
        float l = 0.f;
        if (ga >= 0 && ga < p.G) l += (lg - s_gl[ga]) + (lf - s_fu[ga]);
        if (la >= 0 && la < p.W) l += lw - s_ll[la];
        p.loss[b] = l;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(64) void sap_bwd_kernel(SapArgs p) {
  extern __shared__ float sm[];              // [G] d fused (total)
  float* s_df = sm;
  const int b = blockIdx.x, lane = threadIdx.x;
  const float fw = sap_fw<T>(p, b);
  const T* gs = reinterpret_cast<const T*>(p.gs) + (int64_t)b * p.G;
  const T* ls = reinterpret_cast<const T*>(p.ls) + (int64_t)b * p.W;
  T* dgs = reinterpret_cast<T*>(p.dgs) + (int64_t)b * p.G;
  T* dls = reinterpret_cast<T*>(p.dls) + (int64_t)b * p.W;
  const float dl = p.dloss ? p.dloss[b] : 0.f;
  const int ga = (p.dloss && p.ga) ? (int)p.ga[b] : -1, la = (p.dloss && p.la) ? (int)p.la[b] : -1;
  const bool use_g = ga >= 0 && ga < p.G, use_l = la >= 0 && la < p.W;
  const float lg = p.lse ? p.lse[b * 3] : 0.f, lw = p.lse ? p.lse[b * 3 + 1] : 0.f, lf = p.lse ? p.lse[b * 3 + 2] : 0.f;
  float acc_fw = 0.f;
  for (int g = lane; g < p.G; g += 64) {
    const int64_t o = (int64_t)b * p.G + g;
    float dg = p.dgl ? p.dgl[o] : 0.f, df = p.dfused ? p.dfused[o] : 0.f;
    if (use_g) {
      const float glv = p.gl[o], fuv = p.fused[o];
      dg += dl * ((glv == -INFINITY ? 0.f : __expf(glv - lg)) - (g == ga ? 1.f : 0.f));
      df += dl * ((fuv == -INFINITY ? 0.f : __expf(fuv - lf)) - (g == ga ? 1.f : 0.f));
    }
    s_df[g] = df;
    const float dr = sap_gmasked(p, b, g) ? 0.f : dg + df;
    dgs[g] = (T)(dr * fw);
    acc_fw += dr * (float)gs[g];
  }
  __syncthreads();
  for (int w = lane; w < p.W; w += 64) {
    const int64_t o = (int64_t)b * p.W + w;
    float dw = p.dll ? p.dll[o] : 0.f;
    if (use_l) {
      const float llv = p.ll[o];
      dw += dl * ((llv == -INFINITY ? 0.f : __expf(llv - lw)) - (w == la ? 1.f : 0.f));
    }
    if (p.M) {
      const float* mc = p.M + (int64_t)b * p.G * p.W + w;
      for (int g = 0; g < p.G; ++g) dw += mc[(int64_t)g * p.W] * s_df[g];
    }
    if (p.add_stop && w == 0) dw += s_df[0];
    const float dr = sap_lmasked(p, b, w) ? 0.f : dw;
    dls[w] = (T)(dr * (1.f - fw));
    acc_fw -= dr * (float)ls[w];
  }
  acc_fw = wave_sum(acc_fw);
  if (lane == 0 && p.dfwl) reinterpret_cast<T*>(p.dfwl)[b] = (T)(p.fw_sigmoid ? acc_fw * fw * (1.f - fw) : acc_fw);
}

// ---- CFP fused vector: fo = go * w + vo * (1 - w), w = sigmoid(fwl)   (P/model/pretrain_goat.py:486-499: the glocal fusion weight on the
// pooled map / local vectors).  One block per sample; float32 vectors, the fusion logit in the compute dtype.
template <typename T>
__global__ __launch_bounds__(256) void cfp_mix_fwd_kernel(const float* __restrict__ go, const float* __restrict__ vo, const T* __restrict__ fwl,
                                                          float* __restrict__ fo, float* __restrict__ fw, int H) {
  const int b = blockIdx.x;
  const float w = 1.f / (1.f + __expf(-to_f(fwl[b])));
  if (threadIdx.x == 0) fw[b] = w;
  for (int h = threadIdx.x; h < H; h += 256) {
    const int64_t i = (int64_t)b * H + h;
    fo[i] = go[i] * w + vo[i] * (1.f - w);
  }
}
// dgo += dfo * w ; dvo += dfo * (1 - w) ; dfwl = w (1 - w) sum_h dfo (go - vo)
template <typename T>
__global__ __launch_bounds__(256) void cfp_mix_bwd_kernel(const float* __restrict__ go, const float* __restrict__ vo, const float* __restrict__ fw,
                                                          const float* __restrict__ dfo, float* __restrict__ dgo, float* __restrict__ dvo,
                                                          T* __restrict__ dfwl, int H, int accumulate) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float w = fw[b];
  float s = 0.f;
  for (int h = threadIdx.x; h < H; h += 256) {
    const int64_t i = (int64_t)b * H + h;
    const float d = dfo[i];
    s += d * (go[i] - vo[i]);
    dgo[i] = accumulate ? dgo[i] + d * w : d * w;
    dvo[i] = accumulate ? dvo[i] + d * (1.f - w) : d * (1.f - w);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dfwl[b] = from_f<T>((red[0] + red[1] + red[2] + red[3]) * w * (1.f - w));
}

// ---- Linear(H, 1): the last layer of ClsPrediction (P/model/pretrain_goat.py:27-38: global / local action scores, fusion logit, object scores).
// As a GEMM it is one output column: a 64 x 128 tile computing 1/128 of itself, a split-K weight gradient, and torch glue for the odd shapes
// around it.  Here: y[m] = x[m,:] . w + b with one wave per row; backward dx[m,:] = dy[m] w and per-block partials of dw / db added
// atomically (float32; the caller's arena slice or a zeroed temporary).  The weight is rounded to the activation dtype first, as the GEMM
// path's bf16 shadow was: same products, float32 accumulation.
template <typename T>
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                         T* __restrict__ y, int M, int H) {
  constexpr int EPC = DT<T>::EPC;
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f;
  for (int c = lane; c < H / EPC; c += 64) {
    Chunk<T> v;
    v.load(x + (int64_t)row * H + c * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) s += v.v[e] * to_f(from_f<T>(w[c * EPC + e]));
  }
  s = wave_sum(s);
  if (lane == 0) y[row] = from_f<T>(s + (b ? b[0] : 0.f));
}

template <typename T, int MAXC>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const T* __restrict__ dy,
                                                         T* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int M, int H) {
  constexpr int EPC = DT<T>::EPC;
  extern __shared__ float part[];          // [4][H]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[MAXC][EPC], wv[MAXC][EPC];
  float dbs = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      acc[i][e] = 0.f;
      const int c = lane + 64 * i;
      wv[i][e] = (c < H / EPC) ? to_f(from_f<T>(w[c * EPC + e])) : 0.f;
    }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float d = to_f(dy[row]);
    if (lane == 0) dbs += d;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < H / EPC) {
        const int64_t base = (int64_t)row * H + c * EPC;
        if (dw) {
          Chunk<T> v;
          v.load(x + base);
#pragma unroll
          for (int e = 0; e < EPC; ++e) acc[i][e] += d * v.v[e];
        }
        if (dx) {
          Chunk<T> o;
#pragma unroll
          for (int e = 0; e < EPC; ++e) o.v[e] = d * wv[i][e];
          o.store(dx + base);
        }
      }
    }
  }
  if (!dw) return;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < H / EPC)
#pragma unroll
      for (int e = 0; e < EPC; ++e) part[wave * H + c * EPC + e] = acc[i][e];
  }
  __syncthreads();
  for (int h = threadIdx.x; h < H; h += 256) atomicAdd(dw + h, part[h] + part[H + h] + part[2 * H + h] + part[3 * H + h]);
  if (db && lane == 0) atomicAdd(db, dbs);
}
}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int goat_attn_pool_fwd(void* stream, int dtype, const void* x, const float* w, float* out, float* attn,
                                  float* ws, int B, int L, int H, const float* slot_mask) {
  if (!x || !w || !out || !attn || !ws) return GOAT_E_ARG;
  if (B <= 0 || L <= 0 || L > POOL_MAXL || H <= 0 || (H % 4)) return GOAT_E_SHAPE;
  const int rows = B * L;
  dim3 g1((rows + 3) / 4), g2(B, (H + 255) / 256);
  if (dtype == GOAT_BF16) {
    hipLaunchKernelGGL(pool_score_kernel<bf16_t>, g1, dim3(256), 0, ST(stream), (const bf16_t*)x, w, nullptr, nullptr, ws, rows, L, H, slot_mask);
    hipLaunchKernelGGL(pool_out_kernel<bf16_t>, g2, dim3(256), 0, ST(stream), (const bf16_t*)x, ws, out, attn, L, H);
  } else if (dtype == GOAT_F32) {
    hipLaunchKernelGGL(pool_score_kernel<float>, g1, dim3(256), 0, ST(stream), (const float*)x, w, nullptr, nullptr, ws, rows, L, H, slot_mask);
    hipLaunchKernelGGL(pool_out_kernel<float>, g2, dim3(256), 0, ST(stream), (const float*)x, ws, out, attn, L, H);
  } else {
    return GOAT_E_ARG;
  }
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_attn_pool_bwd(void* stream, int dtype, const void* x, const float* w, const float* attn,
                                  const float* out, const float* dout, void* dx, float* dw, float* ws, int B, int L, int H) {
  if (!x || !w || !attn || !out || !dout || !dx || !dw || !ws) return GOAT_E_ARG;
  if (B <= 0 || L <= 0 || L > POOL_MAXL || H <= 0 || (H % 4)) return GOAT_E_SHAPE;
  const int rows = B * L;
  dim3 g1((rows + 3) / 4), g2(B, (H + 255) / 256);
  if (dtype == GOAT_BF16) {
    hipLaunchKernelGGL(pool_score_kernel<bf16_t>, g1, dim3(256), 0, ST(stream), (const bf16_t*)x, w, out, dout, ws, rows, L, H);
    hipLaunchKernelGGL(pool_bwd_kernel<bf16_t>, g2, dim3(256), 0, ST(stream), (const bf16_t*)x, w, attn, out, dout, ws, (bf16_t*)dx, dw, L, H);
  } else if (dtype == GOAT_F32) {
    hipLaunchKernelGGL(pool_score_kernel<float>, g1, dim3(256), 0, ST(stream), (const float*)x, w, out, dout, ws, rows, L, H);
    hipLaunchKernelGGL(pool_bwd_kernel<float>, g2, dim3(256), 0, ST(stream), (const float*)x, w, attn, out, dout, ws, (float*)dx, dw, L, H);
  } else {
    return GOAT_E_ARG;
  }
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_door_gate_fwd(void* stream, int dtype, const void* aug, const void* ori, const float* wa,
                                  const float* wo, const float* ba, const float* bo, void* out, float* gate, int rows, int H) {
  if (!aug || !ori || !wa || !wo || !ba || !bo || !out || !gate) return GOAT_E_ARG;
  if (rows <= 0 || H <= 0) return GOAT_E_SHAPE;
  const int blocks = (rows + 3) / 4;
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(door_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST(stream), (const bf16_t*)aug, (const bf16_t*)ori,
                       wa, wo, ba, bo, (bf16_t*)out, gate, rows, H);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(door_fwd_kernel<float>, dim3(blocks), dim3(256), 0, ST(stream), (const float*)aug, (const float*)ori,
                       wa, wo, ba, bo, (float*)out, gate, rows, H);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_door_gate_bwd(void* stream, int dtype, const void* aug, const void* ori, const float* wa,
                                  const float* wo, const float* gate, const void* dout, void* daug, void* dori, float* dwa,
                                  float* dwo, float* dbias, int rows, int H, float* dbias2) {
  if (!aug || !ori || !wa || !wo || !gate || !dout || !daug || !dori || !dwa || !dwo || !dbias) return GOAT_E_ARG;
  if (rows <= 0 || H <= 0 || H > 64 * DOOR_MAXC) return GOAT_E_SHAPE;
  int blocks = (rows + 3) / 4;
  if (blocks > 256) blocks = 256;
  const size_t sm = (size_t)8 * H * sizeof(float);
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(door_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), sm, ST(stream), (const bf16_t*)aug, (const bf16_t*)ori,
                       wa, wo, gate, (const bf16_t*)dout, (bf16_t*)daug, (bf16_t*)dori, dwa, dwo, dbias, dbias2, rows, H);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(door_bwd_kernel<float>, dim3(blocks), dim3(256), sm, ST(stream), (const float*)aug, (const float*)ori,
                       wa, wo, gate, (const float*)dout, (float*)daug, (float*)dori, dwa, dwo, dbias, dbias2, rows, H);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_dict_wsum_fwd(void* stream, int dtype_out, const float* z, const float* p, void* out, int B, int K,
                                  int H) {
  if (!z || !p || !out) return GOAT_E_ARG;
  if (B <= 0 || K <= 0 || H <= 0) return GOAT_E_SHAPE;
  if (dtype_out == GOAT_BF16)
    hipLaunchKernelGGL(dict_wsum_fwd_kernel<bf16_t>, dim3(B), dim3(256), 0, ST(stream), z, p, (bf16_t*)out, K, H);
  else if (dtype_out == GOAT_F32)
    hipLaunchKernelGGL(dict_wsum_fwd_kernel<float>, dim3(B), dim3(256), 0, ST(stream), z, p, (float*)out, K, H);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_dict_wsum_bwd(void* stream, int dtype_dout, const void* dout, const float* z, const float* p,
                                  float* dz, float* dp, int B, int K, int H) {
  if (!dout || !z || !p) return GOAT_E_ARG;
  if (B <= 0 || K <= 0 || H <= 0) return GOAT_E_SHAPE;
  if (!dz && !dp) return 0;
  if (dtype_dout == GOAT_BF16)
    hipLaunchKernelGGL(dict_wsum_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, ST(stream), (const bf16_t*)dout, z, p, dz, dp, K, H);
  else if (dtype_dout == GOAT_F32)
    hipLaunchKernelGGL(dict_wsum_bwd_kernel<float>, dim3(B), dim3(256), 0, ST(stream), (const float*)dout, z, p, dz, dp, K, H);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

// x_loc / txt_loc: [Bl, H]; x_all / txt_all: [Ba, H] (order gmap, vp, fused); gradients are ADDED to d* (pre-zeroed by the
// caller; pointers may alias when loc == all); prob: [6, Bl, Ba] scratch written by the forward and read by the backward.
static int nce_fill(NceArgs& a, const float* const* x_loc, const float* const* x_all, const float* txt_loc, const float* txt_all,
                    float* const* dx_loc, float* const* dx_all, float* dtxt_loc, float* dtxt_all, float* prob, int Bl, int Ba, int H,
                    int target0, float temperature) {
  if (!x_loc || !x_all || !txt_loc || !txt_all || !prob) return GOAT_E_ARG;
  if (Bl <= 0 || Ba < Bl || H <= 0 || (H % 4) || target0 < 0 || target0 + Bl > Ba || !(temperature > 0.f)) return GOAT_E_SHAPE;
  for (int k = 0; k < 3; ++k) {
    if (!x_loc[k] || !x_all[k]) return GOAT_E_ARG;
    a.a[2 * k] = x_loc[k]; a.b[2 * k] = txt_all;            // image -> all texts
    a.a[2 * k + 1] = txt_loc; a.b[2 * k + 1] = x_all[k];     // text -> all images
    a.da[2 * k] = dx_loc ? dx_loc[k] : nullptr; a.db[2 * k] = dtxt_all;
    a.da[2 * k + 1] = dtxt_loc; a.db[2 * k + 1] = dx_all ? dx_all[k] : nullptr;
  }
  a.prob = prob; a.Bl = Bl; a.Ba = Ba; a.H = H; a.target0 = target0; a.inv_tau = 1.f / temperature;
  return 0;
}

extern "C" int goat_infonce_fwd(void* stream, const float* const* x_loc, const float* const* x_all, const float* txt_loc,
                                const float* txt_all, float* loss, float* prob, int Bl, int Ba, int H, int target0, float temperature) {
  NceArgs a = {};
  if (!loss) return GOAT_E_ARG;
  if (int e = nce_fill(a, x_loc, x_all, txt_loc, txt_all, nullptr, nullptr, nullptr, nullptr, prob, Bl, Ba, H, target0, temperature)) return e;
  if ((size_t)Ba * 4 > 64 * 1024) return GOAT_E_SHAPE;
  a.loss = loss;
  hipLaunchKernelGGL(infonce_fwd_kernel, dim3(Bl), dim3(256), (size_t)Ba * 4, reinterpret_cast<hipStream_t>(stream), a);
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_infonce_bwd(void* stream, const float* const* x_loc, const float* const* x_all, const float* txt_loc,
                                const float* txt_all, const float* dloss, const float* prob, float* const* dx_loc, float* const* dx_all,
                                float* dtxt_loc, float* dtxt_all, int Bl, int Ba, int H, int target0, float temperature) {
  NceArgs a = {};
  if (!dloss) return GOAT_E_ARG;
  if (int e = nce_fill(a, x_loc, x_all, txt_loc, txt_all, dx_loc, dx_all, dtxt_loc, dtxt_all, const_cast<float*>(prob), Bl, Ba, H, target0,
                       temperature))
    return e;
  a.dloss = dloss;
  NceBwdArgs q = {};
  q.p = a;
  // roles: one per distinct output pointer; its terms in the fixed order pd = 0..5, dA before dB
  for (int pd = 0; pd < 6; ++pd)
    for (int is_b = 0; is_b < 2; ++is_b) {
      float* out = is_b ? a.db[pd] : a.da[pd];
      if (!out) continue;
      int ro = 0;
      while (ro < q.nroles && q.role[ro].out != out) ++ro;
      if (ro == q.nroles) { q.role[ro].out = out; q.role[ro].rows = 0; q.role[ro].n = 0; ++q.nroles; }
      NceRole& role = q.role[ro];
      role.pd[role.n] = pd; role.is_b[role.n] = is_b; ++role.n;
      const int rows = is_b ? Ba : Bl;
      if (rows > role.rows) role.rows = rows;
    }
  if (q.nroles == 0) return 0;
  q.row_groups = (Ba + 3) / 4;
  hipLaunchKernelGGL(infonce_bwd_kernel, dim3(q.nroles * q.row_groups, (H + 63) / 64), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), q);
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_rowdot_fwd(void* stream, int dtype, const void* x, const float* w, const float* b, void* y, int M, int H) {
  if (!x || !w || !y) return GOAT_E_ARG;
  if (M <= 0 || H <= 0 || (H % 8)) return GOAT_E_SHAPE;
  const int blocks = (M + 3) / 4;
  if (dtype == GOAT_BF16)
    hipLaunchKernelGGL(rowdot_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST(stream), (const bf16_t*)x, w, b, (bf16_t*)y, M, H);
  else if (dtype == GOAT_F32)
    hipLaunchKernelGGL(rowdot_fwd_kernel<float>, dim3(blocks), dim3(256), 0, ST(stream), (const float*)x, w, b, (float*)y, M, H);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_rowdot_bwd(void* stream, int dtype, const void* x, const float* w, const void* dy, void* dx, float* dw, float* db, int M,
                               int H) {
  if (!x || !w || !dy) return GOAT_E_ARG;
  if (M <= 0 || H <= 0 || (H % 8)) return GOAT_E_SHAPE;
  if (!dx && !dw) return 0;
  int blocks = (M + 3) / 4;
  if (blocks > 128) blocks = 128;          // (each block adds H + 1 partials atomically)
  const size_t sm = (size_t)4 * H * sizeof(float);
  if (dtype == GOAT_BF16) {
    if (H > 64 * 8 * 2) return GOAT_E_SHAPE;
    hipLaunchKernelGGL((rowdot_bwd_kernel<bf16_t, 2>), dim3(blocks), dim3(256), sm, ST(stream), (const bf16_t*)x, w, (const bf16_t*)dy, (bf16_t*)dx,
                       dw, db, M, H);
  } else if (dtype == GOAT_F32) {
    if (H > 64 * 4 * 4) return GOAT_E_SHAPE;
    hipLaunchKernelGGL((rowdot_bwd_kernel<float, 4>), dim3(blocks), dim3(256), sm, ST(stream), (const float*)x, w, (const float*)dy, (float*)dx, dw,
                       db, M, H);
  } else {
    return GOAT_E_ARG;
  }
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_cfp_mix_fwd(void* stream, int dtype_fwl, const float* go, const float* vo, const void* fwl, float* fo, float* fw, int B, int H) {
  if (!go || !vo || !fwl || !fo || !fw) return GOAT_E_ARG;
  if (B <= 0 || H <= 0) return GOAT_E_SHAPE;
  if (dtype_fwl == GOAT_BF16)
    hipLaunchKernelGGL(cfp_mix_fwd_kernel<bf16_t>, dim3(B), dim3(256), 0, ST(stream), go, vo, (const bf16_t*)fwl, fo, fw, H);
  else if (dtype_fwl == GOAT_F32)
    hipLaunchKernelGGL(cfp_mix_fwd_kernel<float>, dim3(B), dim3(256), 0, ST(stream), go, vo, (const float*)fwl, fo, fw, H);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_cfp_mix_bwd(void* stream, int dtype_fwl, const float* go, const float* vo, const float* fw, const float* dfo, float* dgo,
                                float* dvo, void* dfwl, int B, int H, int accumulate) {
  if (!go || !vo || !fw || !dfo || !dgo || !dvo || !dfwl) return GOAT_E_ARG;
  if (B <= 0 || H <= 0) return GOAT_E_SHAPE;
  if (dtype_fwl == GOAT_BF16)
    hipLaunchKernelGGL(cfp_mix_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, ST(stream), go, vo, fw, dfo, dgo, dvo, (bf16_t*)dfwl, H, accumulate);
  else if (dtype_fwl == GOAT_F32)
    hipLaunchKernelGGL(cfp_mix_bwd_kernel<float>, dim3(B), dim3(256), 0, ST(stream), go, vo, fw, dfo, dgo, dvo, (float*)dfwl, H, accumulate);
  else
    return GOAT_E_ARG;
  GOAT_LAUNCH_CHECK();
  return 0;
}

static int sap_check(const SapArgs& a, int dtype) {
  if (!a.gs || !a.ls || !a.gl || !a.ll || !a.fused) return GOAT_E_ARG;
  if (dtype != GOAT_BF16 && dtype != GOAT_F32) return GOAT_E_ARG;
  if (a.B <= 0 || a.G <= 0 || a.W <= 0 || (size_t)(2 * a.G + 2 * a.W) * 4 > 64 * 1024) return GOAT_E_SHAPE;
  return 0;
}

extern "C" int goat_sap_fuse_fwd(void* stream, int dtype, const void* gs, const void* ls, const void* fwl, int fw_sigmoid,
                                 const uint8_t* gvis, const uint8_t* gvalid, const int64_t* glens, const uint8_t* lmask,
                                 int lmask_is_valid, const float* M, int add_stop, const int64_t* ga, const int64_t* la, float* gl,
                                 float* ll, float* fused, float* loss, float* lse, int B, int G, int W) {
  SapArgs a = {};
  a.gs = gs; a.ls = ls; a.fwl = fwl; a.fw_sigmoid = fw_sigmoid; a.gvis = gvis; a.gvalid = gvalid; a.glens = glens; a.lmask = lmask;
  a.lmask_is_valid = lmask_is_valid; a.M = M; a.add_stop = add_stop; a.ga = ga; a.la = la; a.gl = gl; a.ll = ll; a.fused = fused;
  a.loss = loss; a.lse = lse; a.B = B; a.G = G; a.W = W;
  if (int e = sap_check(a, dtype)) return e;
  if (loss && !lse) return GOAT_E_ARG;
  const size_t smem = (size_t)(2 * G + 2 * W) * 4;
  if (dtype == GOAT_BF16) hipLaunchKernelGGL(sap_fwd_kernel<bf16_t>, dim3(B), dim3(64), smem, reinterpret_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(sap_fwd_kernel<float>, dim3(B), dim3(64), smem, reinterpret_cast<hipStream_t>(stream), a);
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_sap_fuse_bwd(void* stream, int dtype, const void* gs, const void* ls, const void* fwl, int fw_sigmoid,
                                 const uint8_t* gvis, const uint8_t* gvalid, const int64_t* glens, const uint8_t* lmask,
                                 int lmask_is_valid, const float* M, int add_stop, const int64_t* ga, const int64_t* la,
                                 const float* gl, const float* ll, const float* fused, const float* lse, const float* dloss,
                                 const float* dgl, const float* dll, const float* dfused, void* dgs, void* dls, void* dfwl, int B, int G,
                                 int W) {
  SapArgs a = {};
  a.gs = gs; a.ls = ls; a.fwl = fwl; a.fw_sigmoid = fw_sigmoid; a.gvis = gvis; a.gvalid = gvalid; a.glens = glens; a.lmask = lmask;
  a.lmask_is_valid = lmask_is_valid; a.M = M; a.add_stop = add_stop; a.ga = ga; a.la = la;
  a.gl = const_cast<float*>(gl); a.ll = const_cast<float*>(ll); a.fused = const_cast<float*>(fused); a.lse = const_cast<float*>(lse);
  a.dloss = dloss; a.dgl = dgl; a.dll = dll; a.dfused = dfused; a.dgs = dgs; a.dls = dls; a.dfwl = dfwl; a.B = B; a.G = G; a.W = W;
  if (int e = sap_check(a, dtype)) return e;
  if (!dgs || !dls || (dloss && !lse) || (fwl && !dfwl)) return GOAT_E_ARG;
  const size_t smem = (size_t)G * 4;
  if (dtype == GOAT_BF16) hipLaunchKernelGGL(sap_bwd_kernel<bf16_t>, dim3(B), dim3(64), smem, reinterpret_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(sap_bwd_kernel<float>, dim3(B), dim3(64), smem, reinterpret_cast<hipStream_t>(stream), a);
  GOAT_LAUNCH_CHECK();
  return 0;
}
