// Fused optimizer step on the gradient arena: goat_grad_sqnorm + goat_adamw_step (include/goat_hip.h).
// HBM-bound row kernels: per element 4 B gradient + 8 B moments (read + write) + 4 B parameter (read + write) + 2-4 B shadows,
// 16-byte accesses, one chunk of <= 65536 elements per workgroup so that 208 M parameters are ~3 300 workgroups.
#include "common.hpp"

namespace {

constexpr int CHUNK = 65536;

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ arena, const int64_t* __restrict__ ranges, int n_ranges,
                                                     float* out_sq) {
  // grid-stride over (range, CHUNK-sized piece) pairs
  float acc = 0.f;
  for (int r = 0; r < n_ranges; ++r) {
    const int64_t b = ranges[2 * r], e = ranges[2 * r + 1];
    for (int64_t base = b + (int64_t)blockIdx.x * 1024; base < e; base += (int64_t)gridDim.x * 1024) {
      const int64_t i = base + threadIdx.x * 4;
      if (i + 4 <= e && ((i & 3) == 0)) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(arena + i);
        acc += g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3];
      } else {
        for (int k = 0; k < 4; ++k)
          if (i + k < e) acc += arena[i + k] * arena[i + k];
      }
    }
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out_sq, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(const float* __restrict__ grad, float* __restrict__ m_, float* __restrict__ v_,
                                                    const goat_adamw_tensor* __restrict__ tensors, const int32_t* __restrict__ chunks,
                                                    float beta1, float beta2, float eps, float max_norm, const float* __restrict__ sq_norm) {
  const int ti = chunks[2 * blockIdx.x], first = chunks[2 * blockIdx.x + 1];
  const goat_adamw_tensor t = tensors[ti];
  float clip = 1.f;
  if (max_norm > 0.f && sq_norm != nullptr) {
    const float c = max_norm / (sqrtf(*sq_norm) + 1e-6f);        // torch.nn.utils.clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max = 1)
    clip = c < 1.f ? c : 1.f;
  }
  const int64_t n = t.numel;
  const int64_t end = (int64_t)first + CHUNK < n ? (int64_t)first + CHUNK : n;
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  bf16_t* s0 = reinterpret_cast<bf16_t*>(t.shadow0);
  bf16_t* s1 = reinterpret_cast<bf16_t*>(t.shadow1);
  float* sf = t.shadow_f32;
  const bool padded = t.cols > 0;            // shadow0 = row-padded image (leading dimension ld0): scalar stores, tiny tensors
  const bool vec = !padded && (sf == nullptr || (reinterpret_cast<uintptr_t>(sf) & 15) == 0) && ((t.arena_off & 3) == 0) && ((reinterpret_cast<uintptr_t>(t.param) & 15) == 0) &&
                   (s0 == nullptr || (reinterpret_cast<uintptr_t>(s0) & 7) == 0) && (s1 == nullptr || (reinterpret_cast<uintptr_t>(s1) & 7) == 0);
  for (int64_t i = first + threadIdx.x * 4; i < end; i += 256 * 4) {
    float g[4], m[4], v[4], p[4];
    const int cnt = (int)((end - i) < 4 ? (end - i) : 4);
    const bool full = vec && cnt == 4 && ((i & 3) == 0);
    if (full) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(grad + t.arena_off + i), m4 = *reinterpret_cast<const f32x4*>(m_ + t.arena_off + i),
                  v4 = *reinterpret_cast<const f32x4*>(v_ + t.arena_off + i), p4 = *reinterpret_cast<const f32x4*>(t.param + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) { g[k] = g4[k]; m[k] = m4[k]; v[k] = v4[k]; p[k] = p4[k]; }
    } else {
      for (int k = 0; k < cnt; ++k) { g[k] = grad[t.arena_off + i + k]; m[k] = m_[t.arena_off + i + k]; v[k] = v_[t.arena_off + i + k]; p[k] = t.param[i + k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < cnt) {
        const float gc = g[k] * clip;
        m[k] = m[k] * beta1 + omb1 * gc;                         // exp_avg.mul_(beta1).add_(grad, alpha = 1 - beta1)
        v[k] = v[k] * beta2 + omb2 * (gc * gc);                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(v[k]) + eps;
        p[k] = p[k] - t.step_size * (m[k] / denom);             // p.addcdiv_(exp_avg, denom, value = -step_size)
        if (t.decay > 0.f) p[k] = p[k] - t.decay * p[k];         // p.add_(p, alpha = -lr * weight_decay)   (after the update)
      }
    }
    if (full) {
      f32x4 m4, v4, p4;
#pragma unroll
      for (int k = 0; k < 4; ++k) { m4[k] = m[k]; v4[k] = v[k]; p4[k] = p[k]; }
      *reinterpret_cast<f32x4*>(m_ + t.arena_off + i) = m4;
      *reinterpret_cast<f32x4*>(v_ + t.arena_off + i) = v4;
      *reinterpret_cast<f32x4*>(t.param + i) = p4;
      bf16x4 b;
#pragma unroll
      for (int k = 0; k < 4; ++k) b[k] = (bf16_t)p[k];
      if (s0) *reinterpret_cast<bf16x4*>(s0 + i) = b;
      if (s1) *reinterpret_cast<bf16x4*>(s1 + i) = b;
      if (sf) *reinterpret_cast<f32x4*>(sf + i) = p4;
    } else {
      for (int k = 0; k < cnt; ++k) {
        m_[t.arena_off + i + k] = m[k]; v_[t.arena_off + i + k] = v[k]; t.param[i + k] = p[k];
        if (s0) s0[padded ? ((i + k) / t.cols) * (int64_t)t.ld0 + (i + k) % t.cols : i + k] = (bf16_t)p[k];
        if (s1) s1[i + k] = (bf16_t)p[k];
        if (sf) sf[i + k] = p[k];
      }
    }
  }
}

}  // namespace

extern "C" int goat_grad_sqnorm(void* stream, const float* arena, const int64_t* ranges, int n_ranges, float* out_sq) {
  if (!arena || !ranges || !out_sq || n_ranges < 0) return GOAT_E_ARG;
  if (n_ranges == 0) return 0;
  hipLaunchKernelGGL(sqnorm_kernel, dim3(1024), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), arena, ranges, n_ranges, out_sq);
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_adamw_step(void* stream, const float* grad_arena, float* exp_avg, float* exp_avg_sq, const goat_adamw_tensor* tensors,
                               const int32_t* chunks, int nchunks, float beta1, float beta2, float eps, float max_norm,
                               const float* sq_norm) {
  if (!grad_arena || !exp_avg || !exp_avg_sq || !tensors || !chunks || nchunks < 0) return GOAT_E_ARG;
  if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f)) return GOAT_E_ARG;      // as P/optim/adamw.py:44-51
  if (nchunks == 0) return 0;
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), grad_arena, exp_avg, exp_avg_sq,
                     tensors, chunks, beta1, beta2, eps, max_norm, sq_norm);
  GOAT_LAUNCH_CHECK();
  return 0;
}
