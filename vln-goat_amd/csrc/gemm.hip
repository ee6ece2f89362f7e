// goat_gemm_nt: C = epilogue(A[M,K] · B[N,K]^T + bias) on MFMA (gfx950).
//
// Structure (round-1 kernel): 128x128 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles;
// K-tile of 128 bytes per row (64 bf16 / 32 f32), register-staged global->LDS with two LDS stages and one
// barrier per K-tile; LDS rows padded to 144 B so ds_read_b128 fragment reads are bank-conflict free;
// epilogue staged through LDS so global stores (and aux loads/stores) are 16-B coalesced rows;
// XCD-aware 1-D tile order (tiles sharing an A row-panel land on the same XCD's L2).
// bf16 path: v_mfma_f32_32x32x16_bf16; f32 path: v_mfma_f32_32x32x2_f32 (exact f32).
#include "common.hpp"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int BKB = 128;           // bytes of K per row per K-tile
constexpr int LDS_STRIDE = 144;    // padded row stride in bytes
constexpr int STAGE_BYTES = (BM + BN) * LDS_STRIDE;  // 36864
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;          // 73728
constexpr int NTHREADS = 256;

struct GemmArgs {
  const void* A; const void* B; void* C; const float* bias; void* aux;
  int64_t lda, ldb, ldc, ldaux;
  int M, N, K;
  int tiles_m, tiles_n;
  int k_tiles_per_split;
};

template <typename T>
__device__ __forceinline__ void g2r(const T* __restrict__ base, int64_t ld, int row0, int nrows, int k0, int K,
                                    int tid, uint4 (&regs)[4]) {
  constexpr int EPC = DT<T>::EPC;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + NTHREADS * i;
    int r = c >> 3, cc = c & 7;
    int k = k0 + cc * EPC;
    uint4 v = {0u, 0u, 0u, 0u};
    if (row0 + r < nrows && k < K) v = *reinterpret_cast<const uint4*>(base + (int64_t)(row0 + r) * ld + k);
    regs[i] = v;
  }
}

__device__ __forceinline__ void r2s(char* lds, int tid, const uint4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + NTHREADS * i;
    int r = c >> 3, cc = c & 7;
    *reinterpret_cast<uint4*>(lds + r * LDS_STRIDE + cc * 16) = regs[i];
  }
}

template <typename T, typename OutT, int EPI, bool SPLITK>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename FragT<T>::type Frag;
  constexpr int BK = BKB / (int)sizeof(T);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  // XCD-aware bijective remap of the 1-D tile id (block b runs on XCD b%8).
  int nwg = gridDim.x, bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);

  int kt_begin = 0, kt_end = (p.K + BK - 1) / BK;
  if (SPLITK) {
    kt_begin = blockIdx.y * p.k_tiles_per_split;
    kt_end = min(kt_end, kt_begin + p.k_tiles_per_split);
    if (kt_begin >= kt_end) return;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[4];
  g2r<T>(A, p.lda, m0, p.M, kt_begin * BK, p.K, tid, ra);
  g2r<T>(B, p.ldb, n0, p.N, kt_begin * BK, p.K, tid, rb);
  r2s(smem, tid, ra);
  r2s(smem + BM * LDS_STRIDE, tid, rb);
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const bool has_next = (kt + 1 < kt_end);
    if (has_next) {
      g2r<T>(A, p.lda, m0, p.M, (kt + 1) * BK, p.K, tid, ra);
      g2r<T>(B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, tid, rb);
    }
    const char* sa = smem + cur * STAGE_BYTES;
    const char* sb = sa + BM * LDS_STRIDE;
#pragma unroll
    for (int ks = 0; ks < BKB / 32; ++ks) {
      Frag fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = *reinterpret_cast<const Frag*>(sa + (wm * 64 + i * 32 + l31) * LDS_STRIDE + ks * 32 + hi * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[j] = *reinterpret_cast<const Frag*>(sb + (wn * 64 + j * 32 + l31) * LDS_STRIDE + ks * 32 + hi * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma32(acc[i][j], fa[i], fb[j]);
    }
    if (has_next) {
      char* nx = smem + (cur ^ 1) * STAGE_BYTES;
      r2s(nx, tid, ra);
      r2s(nx + BM * LDS_STRIDE, tid, rb);
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  if (SPLITK) {
    float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = m0 + wm * 64 + i * 32 + c_row(r, lane);
          if (row < p.M && col < p.N) atomicAdd(C + (int64_t)row * p.ldc + col, acc[i][j][r]);
        }
      }
    return;
  }

  // staging tiles in LDS (main-loop buffers are free after the final barrier)
  constexpr int CT_STRIDE_T = BN + 16 / (int)sizeof(T);       // elements
  constexpr int CT_STRIDE_O = BN + 16 / (int)sizeof(OutT);
  T* ct_t = reinterpret_cast<T*>(smem);
  OutT* ct_o = reinterpret_cast<OutT*>(smem);
  T* aux = reinterpret_cast<T*>(p.aux);
  constexpr int EPC_T = DT<T>::EPC;
  constexpr int EPC_O = 16 / (int)sizeof(OutT);

  float auxv[2][2][16];
  if (EPI == GOAT_EPI_MUL_DGELU || EPI == GOAT_EPI_MUL_DRELU) {
    // coalesced load of the aux tile -> LDS -> C-layout registers
    for (int c = tid; c < BM * (BN / EPC_T); c += NTHREADS) {
      int r = c / (BN / EPC_T), cc = c % (BN / EPC_T);
      int row = m0 + r, col = n0 + cc * EPC_T;
      if (row < p.M) {
        if (col + EPC_T <= p.N && (p.ldaux % EPC_T) == 0) {
          *reinterpret_cast<uint4*>(ct_t + r * CT_STRIDE_T + cc * EPC_T) =
              *reinterpret_cast<const uint4*>(aux + (int64_t)row * p.ldaux + col);
        } else {
          for (int e = 0; e < EPC_T; ++e)
            if (col + e < p.N) ct_t[r * CT_STRIDE_T + cc * EPC_T + e] = aux[(int64_t)row * p.ldaux + col + e];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          auxv[i][j][r] = to_f(ct_t[(wm * 64 + i * 32 + c_row(r, lane)) * CT_STRIDE_T + wn * 64 + j * 32 + l31]);
    __syncthreads();
  }

  // bias + activation in registers
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int col = n0 + wn * 64 + j * 32 + l31;
    float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float u = acc[i][j][r] + bv;
        if (EPI == GOAT_EPI_MUL_DGELU) u = u * dgelu_f(auxv[i][j][r]);
        if (EPI == GOAT_EPI_MUL_DRELU) u = auxv[i][j][r] > 0.f ? u : 0.f;
        acc[i][j][r] = u;
      }
  }

  if ((EPI == GOAT_EPI_GELU || EPI == GOAT_EPI_RELU)) {
    if (aux != nullptr) {  // store pre-activation u
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            ct_t[(wm * 64 + i * 32 + c_row(r, lane)) * CT_STRIDE_T + wn * 64 + j * 32 + l31] = from_f<T>(acc[i][j][r]);
      __syncthreads();
      for (int c = tid; c < BM * (BN / EPC_T); c += NTHREADS) {
        int r = c / (BN / EPC_T), cc = c % (BN / EPC_T);
        int row = m0 + r, col = n0 + cc * EPC_T;
        if (row < p.M) {
          if (col + EPC_T <= p.N && (p.ldaux % EPC_T) == 0) {
            *reinterpret_cast<uint4*>(aux + (int64_t)row * p.ldaux + col) =
                *reinterpret_cast<const uint4*>(ct_t + r * CT_STRIDE_T + cc * EPC_T);
          } else {
            for (int e = 0; e < EPC_T; ++e)
              if (col + e < p.N) aux[(int64_t)row * p.ldaux + col + e] = ct_t[r * CT_STRIDE_T + cc * EPC_T + e];
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float u = acc[i][j][r];
          acc[i][j][r] = (EPI == GOAT_EPI_GELU) ? gelu_f(u) : fmaxf(u, 0.f);
        }
  }

  // output tile -> LDS -> coalesced rows
  OutT* C = reinterpret_cast<OutT*>(p.C);
  if (sizeof(OutT) * BM * CT_STRIDE_O <= (size_t)SMEM_BYTES) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ct_o[(wm * 64 + i * 32 + c_row(r, lane)) * CT_STRIDE_O + wn * 64 + j * 32 + l31] = from_f<OutT>(acc[i][j][r]);
    __syncthreads();
    const bool vec_ok = (p.ldc % EPC_O) == 0 && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    for (int c = tid; c < BM * (BN / EPC_O); c += NTHREADS) {
      int r = c / (BN / EPC_O), cc = c % (BN / EPC_O);
      int row = m0 + r, col = n0 + cc * EPC_O;
      if (row < p.M) {
        if (col + EPC_O <= p.N && vec_ok) {
          *reinterpret_cast<uint4*>(C + (int64_t)row * p.ldc + col) =
              *reinterpret_cast<const uint4*>(ct_o + r * CT_STRIDE_O + cc * EPC_O);
        } else {
          for (int e = 0; e < EPC_O; ++e)
            if (col + e < p.N) C[(int64_t)row * p.ldc + col + e] = ct_o[r * CT_STRIDE_O + cc * EPC_O + e];
        }
      }
    }
  }
}

template <typename T, typename OutT, int EPI, bool SPLITK>
int launch(hipStream_t st, const GemmArgs& a, int split) {
  auto kern = gemm_nt_kernel<T, OutT, EPI, SPLITK>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(a.tiles_m * a.tiles_n, SPLITK ? split : 1);
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), SMEM_BYTES, st, a);
  GOAT_LAUNCH_CHECK();
  return 0;
}

template <typename T, typename OutT>
int dispatch_epi(hipStream_t st, const GemmArgs& a, int epi, int split) {
  if (split > 1) return launch<T, float, GOAT_EPI_NONE, true>(st, a, split);
  switch (epi) {
    case GOAT_EPI_NONE: return launch<T, OutT, GOAT_EPI_NONE, false>(st, a, 1);
    case GOAT_EPI_GELU: return launch<T, OutT, GOAT_EPI_GELU, false>(st, a, 1);
    case GOAT_EPI_RELU: return launch<T, OutT, GOAT_EPI_RELU, false>(st, a, 1);
    case GOAT_EPI_MUL_DGELU: return launch<T, OutT, GOAT_EPI_MUL_DGELU, false>(st, a, 1);
    case GOAT_EPI_MUL_DRELU: return launch<T, OutT, GOAT_EPI_MUL_DRELU, false>(st, a, 1);
  }
  return GOAT_E_ARG;
}

}  // namespace

extern "C" int goat_gemm_nt(void* stream, int dtype_in, int dtype_out,
                            const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                            int M, int N, int K, const float* bias, int epilogue,
                            void* aux, int64_t ldaux, int split_k) {
  if (!A || !B || !C) return GOAT_E_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return GOAT_E_SHAPE;
  const int epc = (dtype_in == GOAT_BF16) ? 8 : 4;
  if (dtype_in != GOAT_BF16 && dtype_in != GOAT_F32) return GOAT_E_ARG;
  if ((K % epc) || (lda % epc) || (ldb % epc)) return GOAT_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return GOAT_E_SHAPE;
  if ((epilogue == GOAT_EPI_MUL_DGELU || epilogue == GOAT_EPI_MUL_DRELU) && !aux) return GOAT_E_ARG;
  if (split_k > 1 && (dtype_out != GOAT_F32 || epilogue != GOAT_EPI_NONE || bias)) return GOAT_E_ARG;
  if (dtype_in == GOAT_F32 && dtype_out != GOAT_F32) return GOAT_E_ARG;

  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
  a.M = M; a.N = N; a.K = K;
  a.tiles_m = (M + BM - 1) / BM;
  a.tiles_n = (N + BN - 1) / BN;
  const int bk = BKB / (dtype_in == GOAT_BF16 ? 2 : 4);
  const int kt = (K + bk - 1) / bk;
  if (split_k < 1) split_k = 1;
  if (split_k > kt) split_k = kt;
  a.k_tiles_per_split = (kt + split_k - 1) / split_k;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype_in == GOAT_BF16) {
    if (dtype_out == GOAT_BF16) return dispatch_epi<bf16_t, bf16_t>(st, a, epilogue, split_k);
    return dispatch_epi<bf16_t, float>(st, a, epilogue, split_k);
  }
  return dispatch_epi<float, float>(st, a, epilogue, split_k);
}
