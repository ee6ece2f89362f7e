// goat_gemm_bf16: pipelined bf16 MFMA GEMM for gfx950 with direct-to-LDS loads, all three operand layouts.
//
//   C[M,N] = epilogue( op(A) · op(B)^T )        contraction length Kc
//     TA=0: A is [M, Kc] (Kc contiguous)        TA=1: A is [Kc, M] (M contiguous)   -> "transposed" operand
//     TB=0: B is [N, Kc]                        TB=1: B is [Kc, N]
//   (TA,TB) = (0,0) forward  y = x W^T ; (0,1) dgrad  dx = dy W ; (1,1) wgrad  dW = dy^T x.
//
// Why a second GEMM kernel: GOAT's GEMMs are small (M 1-9 k rows, K 768-3072), so a workgroup sees only
// 12-48 K-tiles and the round-1 register-staged kernel (gemm.hip) was latency-bound (one tile in flight).
// Here every operand tile goes HBM/L2 -> LDS by `buffer_load ... lds` (LDS-DMA, no VGPR round trip) into a
// ring of NSTAGE stages with NSTAGE-1 tiles in flight across a single raw s_barrier per K-tile and counted
// vmcnt waits; out-of-range rows / contraction tails are zero-filled by the buffer descriptor's bounds check.
// LDS images are lane-linear (DMA constraint), so bank conflicts are removed by XOR-swizzling the *source*
// address and applying the same involution on the fragment reads.  Transposed operands are read with
// ds_read_b64_tr_b16 (hardware 4x16 transpose), so wgrad/dgrad need no transposed copies.  LDS reads are
// inline asm (hipcc would otherwise drain the DMA queue with vmcnt(0) before every ds_read).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "common.hpp"

namespace {

constexpr int BN = 128, BK = 64;
constexpr int NT = 256;   // threads of the 64- and 128-row tiles (4 waves); the 256-row tile runs 8 waves (nthreads<BM>())
// tile ids (the BM template argument): 64, 128, 256 = rows; TILE_128X8 = the 128-row tile run by eight waves (32x64 wave patches)
constexpr int TILE_128X8 = 129;
template <int BM> constexpr int tile_rows() { return BM == TILE_128X8 ? 128 : BM; }
template <int BM> constexpr int nthreads() { return (BM == 256 || BM == TILE_128X8) ? 512 : 256; }
#ifndef GOAT_GEMM_INTERLEAVE
#define GOAT_GEMM_INTERLEAVE 1
#endif
#ifndef GOAT_GEMM_FRAG_DEPTH
#define GOAT_GEMM_FRAG_DEPTH 2
#endif

struct G2Args {
  const void* A; const void* B; void* C; const float* bias; void* aux;
  int64_t lda, ldb, ldc, ldaux;
  int M, N, Kc;
  int tiles_m, tiles_n;
  int k_tiles_per_split;
  uint32_t a_bytes, b_bytes;  // buffer sizes for the bounds check
  float* colsum;              // TA only: colsum[m] += sum_k A[k,m]  (bias gradient fused into wgrad)
  int accum;                  // f32 output, no split: C += A·B (read-modify-write) instead of C = A·B
  int group_m;                // tile order: column-major inside groups of group_m tile rows (L2-sized 2-D blocks per XCD)
};

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ uint4 lds_read_b128(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint2 lds_read_tr16(uint32_t addr) {
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void wait_lgkm0() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N_> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N_> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// Per-operand tile geometry.  ROWS x RB bytes, lane-linear LDS image, swizzled source.
template <bool T, int BMN>
struct Tile {
  static constexpr int RB = T ? BMN * 2 : BK * 2;         // bytes per LDS row
  static constexpr int ROWS = T ? BK : BMN;
  static constexpr int BYTES = ROWS * RB;                 // 16 KiB (BMN=128) / 8 KiB (BMN=64)
  static constexpr int NINST = BYTES / 1024;              // DMA wave-instructions per tile
  static constexpr int RPB = 256 / RB > 0 ? 256 / RB : 1; // LDS rows per 256-B bank row
  static constexpr int C64 = RB / 64;                     // 64-B chunks per row

  // byte offset inside the *matrix* (global) for LDS linear offset o of the tile whose origin is (mn0, k0=0)
  __device__ static __forceinline__ uint32_t src_off(int o, int mn0, int64_t ld) {
    const int row = o / RB, slot = (o % RB) >> 4;
    if (!T) {
      const int c = slot ^ ((row >> 1) & 7);
      return (uint32_t)(((int64_t)(mn0 + row) * ld + c * 8) * 2);
    } else {
      const int c64 = (slot >> 2) ^ ((row / RPB) % C64);
      const int col = ((c64 << 2) | (slot & 3)) * 8;
      return (uint32_t)(((int64_t)row * ld + mn0 + col) * 2);
    }
  }
  // per-K-tile advance of the source offset in bytes
  __device__ static __forceinline__ uint32_t k_step(int64_t ld) { return T ? (uint32_t)(BK * ld * 2) : (uint32_t)(BK * 2); }
};

// fragment read addresses --------------------------------------------------------------------------
// non-transposed: lane (l31,hi) reads 16-B chunk (ks*2+hi) of row `row` -> slot = chunk ^ ((row>>1)&7)
__device__ __forceinline__ uint32_t frag_addr_n(uint32_t tile_base, int row, int ks, int hi) {
  return tile_base + row * (BK * 2) + ((((ks << 1) | hi) ^ ((row >> 1) & 7)) << 4);
}
// transposed: two ds_read_b64_tr_b16; `col0` = first column of this lane's 16-column block,
// t = lane&15 supplies the address of k-row (kbase + (t>>2)), columns col0 + 4*(t&3) .. +3
template <int RB, int RPB, int C64>
__device__ __forceinline__ uint32_t frag_addr_t(uint32_t tile_base, int kr, int col) {
  const int byte = col * 2;
  const int slot = byte >> 4;
  const int c64 = (slot >> 2) ^ ((kr / RPB) % C64);
  return tile_base + kr * RB + ((((c64 << 2) | (slot & 3))) << 4) + (byte & 15);
}

// blockIdx.x -> position in an order that gives every XCD (8 private L2s, workgroups dealt round-robin) ONE contiguous chunk
__device__ __forceinline__ int xcd_chunk_position(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// One output tile of one GEMM problem.  `bid` = position of the tile in the problem's tile order, `split` = K-split index.
template <bool TA, bool TB, typename OutT, int EPI, bool SPLITK, int BMID, int NSTAGE>
__device__ __forceinline__ void gemm2_tile(const G2Args& p, int bid, int split) {
#if defined(__HIP_DEVICE_COMPILE__)  // (host pass: the gfx950-only builtins below would silently drop the kernel stubs)
  constexpr int BM = tile_rows<BMID>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Tile<TA, BM> TLA;
  typedef Tile<TB, BN> TLB;
  constexpr int STAGE = TLA::BYTES + TLB::BYTES;
  // waves: NW/2 wave rows x 2 wave columns, each wave a (32*MI) x 64 patch.  BM 256 = eight waves of the 128-row tile's
  // patch: the B tile is shared by twice the rows, so a K-tile moves 48 KB through the L1 path for 2x the MFMA work (the
  // 128-row tile is balanced 1:1 against that path)
  // Eight waves on the 128-row tile (TILE_128X8): a lone workgroup per CU is limited by how fast its waves can ISSUE the
  // LDS-DMA instructions (each stalls the issuing wave for ~100 cycles); twice the waves halve that per-wave share.
  constexpr int NTH = nthreads<BMID>(), NW = NTH / 64;
  constexpr int WROWS = BM / (NW / 2);  // rows of a wave patch
  constexpr int MI = WROWS / 32;        // 32-row MFMA tiles per wave in M
  static_assert(MI == 1 || MI == 2, "wave patch is 32 or 64 rows");
  constexpr int IPWA = TLA::NINST / NW, IPWB = TLB::NINST / NW;   // DMA wave-instructions per wave and K-tile
  static_assert(IPWA >= 1 && IPWB >= 1, "every wave issues at least one DMA instruction per operand");
  constexpr int LOADS = IPWA + IPWB;
  // k-steps a K-tile's DMA instructions are spread over: a 2-stage ring waits for them at the very next barrier, so they
  // go behind the first two k-steps only; deeper rings have a whole K-tile of slack
  constexpr int SPREAD = BK / 16;
  constexpr bool INTERLEAVE = GOAT_GEMM_INTERLEAVE && NSTAGE >= 3;
  // fragment prefetch distance in k-steps (each k-step's fragments have their own registers); bounded by the 4-bit lgkmcnt
  constexpr int KSTEPS = BK / 16;
  constexpr int RD = MI * (TA ? 2 : 1) + 2 * (TB ? 2 : 1);            // ds_read instructions per k-step
  constexpr int FD = (GOAT_GEMM_FRAG_DEPTH * RD <= 15) ? GOAT_GEMM_FRAG_DEPTH : (15 / RD >= 1 ? 15 / RD : 1);

  // `wave` through readfirstlane: the compiler then keeps every wave-uniform quantity (the LDS addresses of this wave's DMA
  // pieces, hence M0) in SGPRs instead of a v_add + v_readfirstlane + s_mov chain in front of every buffer_load ... lds
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  // `bid` walks a contiguous chunk per XCD (each XCD has its own L2).  Inside the chunk tiles are visited
  // column-major within groups of group_m tile rows, so the ~64 workgroups an XCD runs at a time cover a compact
  // group_m x (64/group_m) block: its A and B panels are fetched into that L2 once, and the 8 XCD chunks form a 2-D
  // partition of C instead of 8 full-width stripes (measured: L2-miss traffic 8.3x -> see profiles/ of the algorithmic bytes).
  const int gsz = p.group_m * p.tiles_n;
  const int grp = bid / gsz, gi = bid - grp * gsz;
  const int gm = min(p.tiles_m - grp * p.group_m, p.group_m);
  const int tn = gi / gm, tm = grp * p.group_m + (gi - tn * gm);
  const int m0 = tm * BM, n0 = tn * BN;

  int kt_begin = 0, kt_end = (p.Kc + BK - 1) / BK;
  if (SPLITK) {
    kt_begin = split * p.k_tiles_per_split;
    kt_end = min(kt_end, kt_begin + p.k_tiles_per_split);
    if (kt_begin >= kt_end) return;
  }
  const int nkt = kt_end - kt_begin;

  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)p.b_bytes, 0x00020000);

  // per-lane source offsets of this wave's DMA instructions at k-tile 0 of this split
  uint32_t offa[IPWA], offb[IPWB];
  const uint32_t ka = TLA::k_step(p.lda), kb = TLB::k_step(p.ldb);
#pragma unroll
  for (int j = 0; j < IPWA; ++j)
    offa[j] = TLA::src_off((wave * IPWA + j) * 1024 + lane * 16, m0, p.lda) + (uint32_t)kt_begin * ka;
#pragma unroll
  for (int j = 0; j < IPWB; ++j)
    offb[j] = TLB::src_off((wave * IPWB + j) * 1024 + lane * 16, n0, p.ldb) + (uint32_t)kt_begin * kb;

#define GOAT_ISSUE(t_)                                                                                              \
  do {                                                                                                              \
    char* st_ = smem + ((t_) % NSTAGE) * STAGE;                                                                     \
    const uint32_t sa_ = (uint32_t)(t_) * ka, sb_ = (uint32_t)(t_) * kb;                                            \
    _Pragma("unroll") for (int j = 0; j < IPWA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(                  \
        ra, (lds_void*)(st_ + (wave * IPWA + j) * 1024), 16, offa[j], sa_, 0, 0);                               \
    _Pragma("unroll") for (int j = 0; j < IPWB; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(                  \
        rb, (lds_void*)(st_ + TLA::BYTES + (wave * IPWB + j) * 1024), 16, offb[j], sb_, 0, 0);                  \
  } while (0)
  // one DMA wave-instruction (number j_ of this wave's LOADS) of K-tile t_: the steady-state loop spreads a tile's
  // instructions over the four k-steps, behind their MFMAs, instead of issuing all of them between the barrier and the
  // first MFMA (each costs the wave ~60-100 issue cycles)
#define GOAT_ISSUE_ONE(t_, j_)                                                                                      \
  do {                                                                                                              \
    char* st_ = smem + ((t_) % NSTAGE) * STAGE;                                                                     \
    if ((j_) < IPWA)                                                                                            \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(st_ + (wave * IPWA + (j_)) * 1024), 16,          \
                                               offa[(j_) < IPWA ? (j_) : 0], (uint32_t)(t_) * ka, 0, 0);        \
    else                                                                                                            \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(st_ + TLA::BYTES + (wave * IPWB + (j_) - IPWA) * 1024), 16, \
                                               offb[(j_) >= IPWA ? (j_) - IPWA : 0], (uint32_t)(t_) * kb, 0, 0); \
  } while (0)

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t) GOAT_ISSUE(t);

  const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_void*)smem;  // LDS byte offset of the dynamic region
  const int t15 = lane & 15, g = lane >> 4;
  const bool do_colsum = TA && p.colsum != nullptr && tn == 0 && wn == 0;
  float bsum[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) bsum[i] = 0.f;

  for (int t = 0; t < nkt; ++t) {
    wait_vm<(NSTAGE - 2) * LOADS>();
    __builtin_amdgcn_s_barrier();
    // tile t+NSTAGE-1 (beyond the end: harmless, bounds-checked, lands in a stage nobody reads).  A 2-stage ring waits for it
    // at the very next barrier, so it is issued here, as early as possible; deeper rings have a whole K-tile of slack and
    // spread the instructions behind the MFMAs of the four k-steps (measured: -13...-18 % on the 3-stage shapes, +4...+17 %
    // on the 2-stage ones if done there too)
    if (!INTERLEAVE) GOAT_ISSUE(t + NSTAGE - 1);
    const uint32_t sa = smem_base + (t % NSTAGE) * STAGE;
    const uint32_t sb = sa + TLA::BYTES;
    // fragment reads are software-pipelined one k-step ahead of the MFMAs (a wave is alone on its SIMD,
    // so nothing else hides the LDS latency)
    bf16x8 fa[KSTEPS][MI], fb[KSTEPS][2];
#define GOAT_LOAD_FRAGS(ks_, buf_)                                                                                   \
  do {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                                \
      if (!TA) {                                                                                                    \
        uint4 v = lds_read_b128(frag_addr_n(sa, wm * WROWS + i * 32 + l31, (ks_), hi));                           \
        fa[buf_][i] = *reinterpret_cast<bf16x8*>(&v);                                                               \
      } else {                                                                                                      \
        const int col = wm * WROWS + i * 32 + (g & 1) * 16 + (t15 & 3) * 4;                                      \
        const int kr = (ks_) * 16 + 8 * (g >> 1) + (t15 >> 2);                                                      \
        uint2 v0 = lds_read_tr16(frag_addr_t<TLA::RB, TLA::RPB, TLA::C64>(sa, kr, col));                            \
        uint2 v1 = lds_read_tr16(frag_addr_t<TLA::RB, TLA::RPB, TLA::C64>(sa, kr + 4, col));                        \
        uint4 v = {v0.x, v0.y, v1.x, v1.y};                                                                         \
        fa[buf_][i] = *reinterpret_cast<bf16x8*>(&v);                                                               \
      }                                                                                                             \
    }                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
      if (!TB) {                                                                                                    \
        uint4 v = lds_read_b128(frag_addr_n(sb, wn * 64 + j * 32 + l31, (ks_), hi));                                \
        fb[buf_][j] = *reinterpret_cast<bf16x8*>(&v);                                                               \
      } else {                                                                                                      \
        const int col = wn * 64 + j * 32 + (g & 1) * 16 + (t15 & 3) * 4;                                            \
        const int kr = (ks_) * 16 + 8 * (g >> 1) + (t15 >> 2);                                                      \
        uint2 v0 = lds_read_tr16(frag_addr_t<TLB::RB, TLB::RPB, TLB::C64>(sb, kr, col));                            \
        uint2 v1 = lds_read_tr16(frag_addr_t<TLB::RB, TLB::RPB, TLB::C64>(sb, kr + 4, col));                        \
        uint4 v = {v0.x, v0.y, v1.x, v1.y};                                                                         \
        fb[buf_][j] = *reinterpret_cast<bf16x8*>(&v);                                                               \
      }                                                                                                             \
    }                                                                                                               \
  } while (0)
#define GOAT_KSTEP(ks_)                                                                                            \
  do {                                                                                                              \
    constexpr int left_ = (KSTEPS - (ks_) < FD ? KSTEPS - (ks_) : FD) - 1; /* later k-steps whose reads may stay in flight */ \
    wait_lgkm<left_ * RD>();                                                                                        \
    if ((ks_) + FD < KSTEPS) GOAT_LOAD_FRAGS((ks_) + FD < KSTEPS ? (ks_) + FD : 0, (ks_) + FD < KSTEPS ? (ks_) + FD : 0); \
    if (TA && do_colsum) {                                                                                          \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) bsum[i] += (float)fa[ks_][i][e];                              \
    }                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                  \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) mma32(acc[i][j], fa[ks_][i], fb[ks_][j]);                       \
    if (INTERLEAVE) {                                                                                               \
      /* the target stage was last read in iteration t-1, which every wave left before this iteration's barrier */  \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      _Pragma("unroll") for (int j = 0; j < LOADS; ++j)                                                             \
        if (j * SPREAD / LOADS == (ks_)) GOAT_ISSUE_ONE(t + NSTAGE - 1, j);                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
    }                                                                                                               \
  } while (0)
#pragma unroll
    for (int d = 0; d < FD; ++d) {
      if (d == 0) GOAT_LOAD_FRAGS(0, 0);
      if (d == 1) GOAT_LOAD_FRAGS(1, 1);
      if (d == 2) GOAT_LOAD_FRAGS(2, 2);
    }
    static_assert(KSTEPS == 4 && FD >= 1 && FD <= 3, "k-step unrolling below is written for BK = 64");
    GOAT_KSTEP(0);
    GOAT_KSTEP(1);
    GOAT_KSTEP(2);
    GOAT_KSTEP(3);
  }
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  if (TA && do_colsum) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      const int row = m0 + wm * WROWS + i * 32 + l31;
      if (hi == 0 && row < p.M) atomicAdd(p.colsum + row, v);
    }
  }

  // ------------------------------------------------------------------ epilogue (as gemm.hip)
  const int wrow0 = wm * WROWS, wcol0 = wn * 64;
  if (SPLITK) {
    float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wcol0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wrow0 + i * 32 + c_row(r, lane);
          if (row < p.M && col < p.N) atomicAdd(C + (int64_t)row * p.ldc + col, acc[i][j][r]);
        }
      }
    return;
  }
  typedef bf16_t T;
  constexpr int EPC_T = 8;
  constexpr int EPC_O = 16 / (int)sizeof(OutT);
  constexpr int CT_STRIDE_T = BN + EPC_T;
  constexpr int CT_STRIDE_O = BN + EPC_O;
  static_assert(sizeof(OutT) == 4 || sizeof(OutT) * BM * CT_STRIDE_O <= (size_t)(NSTAGE * STAGE), "epilogue staging must fit");
  T* ct_t = reinterpret_cast<T*>(smem);
  OutT* ct_o = reinterpret_cast<OutT*>(smem);
  T* aux = reinterpret_cast<T*>(p.aux);

  float auxv[MI][2][16];
  if (EPI == GOAT_EPI_MUL_DGELU || EPI == GOAT_EPI_MUL_DRELU) {
    for (int c = tid; c < BM * (BN / EPC_T); c += NTH) {
      const int r = c / (BN / EPC_T), cc = c % (BN / EPC_T);
      const int row = m0 + r, col = n0 + cc * EPC_T;
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
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          auxv[i][j][r] = to_f(ct_t[(wrow0 + i * 32 + c_row(r, lane)) * CT_STRIDE_T + wcol0 + j * 32 + l31]);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wcol0 + j * 32 + l31;
    const float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float u = acc[i][j][r] + bv;
        if (EPI == GOAT_EPI_MUL_DGELU) u = u * dgelu_f(auxv[i][j][r]);
        if (EPI == GOAT_EPI_MUL_DRELU) u = auxv[i][j][r] > 0.f ? u : 0.f;
        acc[i][j][r] = u;
      }
  }
  if (EPI == GOAT_EPI_GELU || EPI == GOAT_EPI_RELU) {
    if (aux != nullptr) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            ct_t[(wrow0 + i * 32 + c_row(r, lane)) * CT_STRIDE_T + wcol0 + j * 32 + l31] = from_f<T>(acc[i][j][r]);
      __syncthreads();
      for (int c = tid; c < BM * (BN / EPC_T); c += NTH) {
        const int r = c / (BN / EPC_T), cc = c % (BN / EPC_T);
        const int row = m0 + r, col = n0 + cc * EPC_T;
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
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float u = acc[i][j][r];
          acc[i][j][r] = (EPI == GOAT_EPI_GELU) ? gelu_f(u) : fmaxf(u, 0.f);
        }
  }
  OutT* C = reinterpret_cast<OutT*>(p.C);
  if (sizeof(OutT) == 4) {  // f32 output: 32 lanes = 128 contiguous bytes per row, store straight from the accumulators
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wcol0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wrow0 + i * 32 + c_row(r, lane);
          if (row < p.M && col < p.N) {
            OutT* dst = C + (int64_t)row * p.ldc + col;
            *dst = from_f<OutT>(p.accum ? acc[i][j][r] + to_f(*dst) : acc[i][j][r]);
          }
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ct_o[(wrow0 + i * 32 + c_row(r, lane)) * CT_STRIDE_O + wcol0 + j * 32 + l31] = from_f<OutT>(acc[i][j][r]);
  __syncthreads();
  const bool vec_ok = (p.ldc % EPC_O) == 0 && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  for (int c = tid; c < BM * (BN / EPC_O); c += NTH) {
    const int r = c / (BN / EPC_O), cc = c % (BN / EPC_O);
    const int row = m0 + r, col = n0 + cc * EPC_O;
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
#endif  // __HIP_DEVICE_COMPILE__
}

template <bool TA, bool TB, typename OutT, int EPI, bool SPLITK, int BM, int NSTAGE>
__global__ __launch_bounds__(nthreads<BM>()) void gemm2_kernel(G2Args p) {
  gemm2_tile<TA, TB, OutT, EPI, SPLITK, BM, NSTAGE>(p, xcd_chunk_position(blockIdx.x, gridDim.x), blockIdx.y);
}

// Grouped weight-gradient launch: up to GROUP_MAX (24) independent TN problems (dW_i = dY_i^T · X_i, float32 out, unsplit)
// share one grid, so the many small weight gradients of a layer fill the chip together instead of each being split
// along the contraction (atomics + a zero fill) to do so.  Problem i owns tiles [tile_start[i], tile_start[i+1]).
constexpr int GROUP_MAX = 24;   // (the argument block stays under the 4 KiB kernel-argument limit)
static_assert(sizeof(G2Args) * GROUP_MAX + 4 * (GROUP_MAX + 2) <= 4000, "GroupArgs must fit the kernel-argument segment");
struct GroupArgs {
  G2Args prob[GROUP_MAX];
  int tile_start[GROUP_MAX + 1];
  int n;
};
template <int BM, int NSTAGE>
__global__ __launch_bounds__(nthreads<BM>()) void gemm2_group_kernel(GroupArgs g) {
  const int pos = xcd_chunk_position(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i)
    if (i < g.n && pos >= g.tile_start[i]) pi = i;
  const G2Args p = g.prob[pi];
  gemm2_tile<true, true, float, GOAT_EPI_NONE, false, BM, NSTAGE>(p, pos - g.tile_start[pi], 0);
}

template <bool TA, bool TB, typename OutT, int EPI, bool SPLITK, int BM, int NSTAGE>
int launch2s(hipStream_t st, const G2Args& a, int split) {
  constexpr int SMEM = NSTAGE * (Tile<TA, tile_rows<BM>()>::BYTES + Tile<TB, BN>::BYTES);
  auto kern = gemm2_kernel<TA, TB, OutT, EPI, SPLITK, BM, NSTAGE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(a.tiles_m * a.tiles_n, SPLITK ? split : 1);
  hipLaunchKernelGGL(kern, grid, dim3(nthreads<BM>()), SMEM, st, a);
  GOAT_LAUNCH_CHECK();
  return 0;
}

// LDS ring depth: 2 stages = 64 KiB (bm 128) / 48 KiB (bm 64) -> 2-3 workgroups per CU (best on short, hot
// contractions); 3-4 stages = deeper prefetch, 1-2 workgroups per CU (best on long / cold contractions)
thread_local int g_nstage = 2;

template <bool TA, bool TB, typename OutT, int EPI, bool SPLITK, int BM>
int launch2(hipStream_t st, const G2Args& a, int split) {
  if (g_nstage == 2) return launch2s<TA, TB, OutT, EPI, SPLITK, BM, 2>(st, a, split);
  if (g_nstage == 3) return launch2s<TA, TB, OutT, EPI, SPLITK, BM, 3>(st, a, split);
  return launch2s<TA, TB, OutT, EPI, SPLITK, BM, 4>(st, a, split);
}

template <bool TA, bool TB, int BM>
int dispatch2(hipStream_t st, const G2Args& a, int dtype_out, int epi, int split) {
  if (split > 1) return launch2<TA, TB, float, GOAT_EPI_NONE, true, BM>(st, a, split);
  if (dtype_out == GOAT_F32) {
    if (epi != GOAT_EPI_NONE && epi != GOAT_EPI_ACCUM) return GOAT_E_ARG;
    return launch2<TA, TB, float, GOAT_EPI_NONE, false, BM>(st, a, 1);
  }
  switch (epi) {
    case GOAT_EPI_NONE: return launch2<TA, TB, bf16_t, GOAT_EPI_NONE, false, BM>(st, a, 1);
    case GOAT_EPI_GELU: return launch2<TA, TB, bf16_t, GOAT_EPI_GELU, false, BM>(st, a, 1);
    case GOAT_EPI_RELU: return launch2<TA, TB, bf16_t, GOAT_EPI_RELU, false, BM>(st, a, 1);
    case GOAT_EPI_MUL_DGELU: return launch2<TA, TB, bf16_t, GOAT_EPI_MUL_DGELU, false, BM>(st, a, 1);
    case GOAT_EPI_MUL_DRELU: return launch2<TA, TB, bf16_t, GOAT_EPI_MUL_DRELU, false, BM>(st, a, 1);
  }
  return GOAT_E_ARG;
}

}  // namespace

// Tile-order parameter: the group height that minimises the operand bytes the eight per-XCD L2s have to fetch,
// sum over XCDs of (distinct tile rows * BM + distinct tile columns * BN) under the kernel's own blockIdx -> tile map
// (XCD x owns one contiguous chunk of the grouped order).  Brute force once per (tiles_m, tiles_n, bm), then cached.
static int pick_group_m(int tiles_m, int tiles_n, int bm) {
  if (const char* e = getenv("GOAT_GEMM_GROUP_M")) {
    int g = atoi(e);
    return g < 1 ? 1 : (g > tiles_m ? tiles_m : g);
  }
  static std::mutex mu;
  static std::unordered_map<uint64_t, int> cache;
  const uint64_t key = ((uint64_t)tiles_m << 40) | ((uint64_t)tiles_n << 16) | (uint64_t)bm;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
  }
  const int nwg = tiles_m * tiles_n, q = nwg >> 3, r = nwg & 7;
  int best = 1;
  double best_cost = 1e300;
  std::vector<char> seen_m(tiles_m), seen_n(tiles_n);
  const int gmax = tiles_m < 64 ? tiles_m : 64;
  for (int gm = 1; gm <= gmax; ++gm) {
    double cost = 0;
    int pos = 0;
    for (int x = 0; x < 8; ++x) {
      const int cnt = x < r ? q + 1 : q;
      std::fill(seen_m.begin(), seen_m.end(), 0);
      std::fill(seen_n.begin(), seen_n.end(), 0);
      int dm = 0, dn = 0;
      for (int i = 0; i < cnt; ++i, ++pos) {
        const int gsz = gm * tiles_n, grp = pos / gsz, gi = pos - grp * gsz;
        const int h = std::min(tiles_m - grp * gm, gm);
        const int tn = gi / h, tm = grp * gm + (gi - tn * h);
        if (!seen_m[tm]) { seen_m[tm] = 1; ++dm; }
        if (!seen_n[tn]) { seen_n[tn] = 1; ++dn; }
      }
      cost += (double)dm * bm + (double)dn * BN;
    }
    if (cost < best_cost - 1e-9) { best_cost = cost; best = gm; }
  }
  std::lock_guard<std::mutex> lk(mu);
  cache[key] = best;
  return best;
}


extern "C" int goat_gemm_bf16(void* stream, int trans_a, int trans_b, int dtype_out,
                              const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                              int M, int N, int Kc, const float* bias, int epilogue,
                              void* aux, int64_t ldaux, int split_k, int bm, int nstage, float* colsum) {
  if (!A || !B || !C) return GOAT_E_ARG;
  const bool eight = (nstage & GOAT_GEMM_8WAVES) != 0;
  nstage &= ~GOAT_GEMM_8WAVES;
  if (nstage < 2 || nstage > 4) return GOAT_E_ARG;
  g_nstage = nstage;
  if (colsum && !trans_a) return GOAT_E_ARG;
  if (M <= 0 || N <= 0 || Kc <= 0) return GOAT_E_SHAPE;
  if ((lda % 8) || (ldb % 8)) return GOAT_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return GOAT_E_SHAPE;
  // contraction tails: transposed operands are zero-filled by the bounds check; K-contiguous operands need
  // whole 64-wide tiles (callers route other shapes to goat_gemm_nt)
  if ((!trans_a || !trans_b) && (Kc % BK)) return GOAT_E_SHAPE;
  if (trans_a && !trans_b) return GOAT_E_ARG;
  if ((epilogue == GOAT_EPI_MUL_DGELU || epilogue == GOAT_EPI_MUL_DRELU) && !aux) return GOAT_E_ARG;
  if (epilogue == GOAT_EPI_ACCUM && (dtype_out != GOAT_F32 || bias)) return GOAT_E_ARG;
  if (split_k > 1 && (dtype_out != GOAT_F32 || (epilogue != GOAT_EPI_NONE && epilogue != GOAT_EPI_ACCUM) || bias)) return GOAT_E_ARG;
  if (bm != 64 && bm != 128 && bm != 256) return GOAT_E_ARG;
  if (eight && bm != 128) return GOAT_E_ARG;
  if (bm == 256 && nstage > 3) return GOAT_E_ARG;      // 48 KiB stages: 3 is the deepest ring in 160 KiB of LDS
  const int64_t a_rows = trans_a ? Kc : M, b_rows = trans_b ? Kc : N;
  const int64_t a_bytes = a_rows * lda * 2, b_bytes = b_rows * ldb * 2;
  if (a_bytes >= (1ll << 31) || b_bytes >= (1ll << 31)) return GOAT_E_SHAPE;

  G2Args a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
  a.M = M; a.N = N; a.Kc = Kc;
  a.tiles_m = (M + bm - 1) / bm;
  a.tiles_n = (N + BN - 1) / BN;
  a.a_bytes = (uint32_t)a_bytes; a.b_bytes = (uint32_t)b_bytes;
  a.colsum = colsum;
  a.accum = epilogue == GOAT_EPI_ACCUM;
  a.group_m = pick_group_m(a.tiles_m, a.tiles_n, bm);
  const int kt = (Kc + BK - 1) / BK;
  if (split_k < 1) split_k = 1;
  if (split_k > kt) split_k = kt;
  a.k_tiles_per_split = (kt + split_k - 1) / split_k;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define GOAT_G2(TA_, TB_) \
  (bm == 128 ? (eight ? dispatch2<TA_, TB_, TILE_128X8>(st, a, dtype_out, epilogue, split_k)     \
                      : dispatch2<TA_, TB_, 128>(st, a, dtype_out, epilogue, split_k))           \
   : bm == 256 ? dispatch2<TA_, TB_, 256>(st, a, dtype_out, epilogue, split_k)   \
               : dispatch2<TA_, TB_, 64>(st, a, dtype_out, epilogue, split_k))
  if (!trans_a && !trans_b) return GOAT_G2(false, false);
  if (!trans_a && trans_b) return GOAT_G2(false, true);
  return GOAT_G2(true, true);
#undef GOAT_G2
}

template <int BM, int NSTAGE>
static int launch_group(hipStream_t st, const GroupArgs& g) {
  constexpr int SMEM = NSTAGE * (Tile<true, tile_rows<BM>()>::BYTES + Tile<true, BN>::BYTES);
  auto kern = gemm2_group_kernel<BM, NSTAGE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.tile_start[g.n]), dim3(nthreads<BM>()), SMEM, st, g);
  GOAT_LAUNCH_CHECK();
  return 0;
}

extern "C" int goat_wgrad_grouped(void* stream, const goat_wgrad_problem* probs, int n, int bm, int nstage) {
  if (!probs || n < 1 || n > GROUP_MAX) return GOAT_E_ARG;
  const bool eight = (nstage & GOAT_GEMM_8WAVES) != 0;       // as in goat_gemm_bf16: the 128-row tile on eight waves
  nstage &= ~GOAT_GEMM_8WAVES;
  if ((bm != 64 && bm != 128) || nstage < 2 || nstage > 4 || (eight && bm != 128)) return GOAT_E_ARG;
  GroupArgs g;
  g.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const goat_wgrad_problem& q = probs[i];
    if (!q.dy || !q.x || !q.dw) return GOAT_E_ARG;
    if (q.rows <= 0 || q.n_out <= 0 || q.n_in <= 0) return GOAT_E_SHAPE;
    if ((q.ld_dy % 8) || (q.ld_x % 8)) return GOAT_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(q.dy) & 15) || (reinterpret_cast<uintptr_t>(q.x) & 15)) return GOAT_E_SHAPE;
    const int64_t a_bytes = (int64_t)q.rows * q.ld_dy * 2, b_bytes = (int64_t)q.rows * q.ld_x * 2;
    if (a_bytes >= (1ll << 31) || b_bytes >= (1ll << 31)) return GOAT_E_SHAPE;
    G2Args& a = g.prob[i];
    a.A = q.dy; a.B = q.x; a.C = q.dw; a.bias = nullptr; a.aux = nullptr;
    a.lda = q.ld_dy; a.ldb = q.ld_x; a.ldc = q.ld_dw; a.ldaux = 0;
    a.M = q.n_out; a.N = q.n_in; a.Kc = q.rows;
    a.tiles_m = (q.n_out + bm - 1) / bm;
    a.tiles_n = (q.n_in + BN - 1) / BN;
    a.k_tiles_per_split = (q.rows + BK - 1) / BK;
    a.a_bytes = (uint32_t)a_bytes; a.b_bytes = (uint32_t)b_bytes;
    a.colsum = q.dbias;
    a.accum = q.accumulate ? 1 : 0;
    a.group_m = pick_group_m(a.tiles_m, a.tiles_n, bm);
    g.tile_start[i] = tiles;
    tiles += a.tiles_m * a.tiles_n;
  }
  for (int i = n; i <= GROUP_MAX; ++i) g.tile_start[i] = tiles;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bm == 128 && eight) {
    if (nstage == 2) return launch_group<TILE_128X8, 2>(st, g);
    if (nstage == 3) return launch_group<TILE_128X8, 3>(st, g);
    return launch_group<TILE_128X8, 4>(st, g);
  }
  if (bm == 128) {
    if (nstage == 2) return launch_group<128, 2>(st, g);
    if (nstage == 3) return launch_group<128, 3>(st, g);
    return launch_group<128, 4>(st, g);
  }
  if (nstage == 2) return launch_group<64, 2>(st, g);
  if (nstage == 3) return launch_group<64, 3>(st, g);
  return launch_group<64, 4>(st, g);
}
