// Device code of goat_gemm_bf16 / goat_wgrad_grouped: one pipelined bf16 MFMA GEMM tile for gfx950 with direct-to-LDS loads,
// all three operand layouts, for a family of workgroup tiles.  Included by gemm2.hip (4-wave tiles, 128-column tiles, the C
// entry points) and gemm3.hip (the 8-wave 192/256-wide tiles), which are compiled in parallel.
//
//   C[M,N] = epilogue( op(A) · op(B)^T )        contraction length Kc
//     TA=0: A is [M, Kc] (Kc contiguous)        TA=1: A is [Kc, M] (M contiguous)   -> "transposed" operand
//     TB=0: B is [N, Kc]                        TB=1: B is [Kc, N]
//   (TA,TB) = (0,0) forward  y = x W^T ; (0,1) dgrad  dx = dy W ; (1,1) wgrad  dW = dy^T x.
//
// Every operand K-tile goes L2 -> LDS by `buffer_load ... lds` (LDS-DMA, no VGPR round trip) into a ring of NSTAGE stages
// with NSTAGE-1 tiles in flight across a single raw s_barrier per K-tile and counted vmcnt waits; out-of-range rows /
// contraction tails are zero-filled by the buffer descriptor's bounds check.  LDS images are lane-linear (DMA constraint), so
// bank conflicts are removed by XOR-swizzling the *source* address and applying the same involution on the fragment reads.
// Transposed operands are read with ds_read_b64_tr_b16 (hardware 4x16 transpose), so wgrad/dgrad need no transposed copies.
// LDS reads are inline asm (hipcc would otherwise drain the DMA queue with vmcnt(0) before every ds_read).
//
// Tile family (Cfg<WM, WN, MI, NI>): WM x WN waves, each wave a (32*MI) x (32*NI) patch of v_mfma_f32_32x32x16_bf16 blocks,
// BM = 32*MI*WM rows, BN = 32*NI*WN columns, BK = 64.  Why the big tiles: scripts/l2_lds_bw.hip measures what a CU can pull
// from its XCD's L2 into LDS at 50-58 B/clk when the data is L2-resident and 18-40 B/clk when it comes from the Infinity
// Cache; at the MFMA peak a 128x128 tile consumes 64 B/clk, 256x128 48, 192x256 37, 256x256 32 (profiles/round2_l2_lds_bw.txt).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "common.hpp"

namespace goat_g2 {

constexpr int BK = 64;
#ifndef GOAT_GEMM_FRAG_DEPTH
#define GOAT_GEMM_FRAG_DEPTH 2
#endif
#ifndef GOAT_G2_ROTATED       // 1: barrier at the top of a K-tile's last k-step (see the main loop); 0: barrier between K-tiles.
#define GOAT_G2_ROTATED 0     // Measured (profiles/round2_gemm_epilogue_ab.txt): equal on the 192/256-wide tiles at K = 768, 3-12 % slower on
#endif                        // the 128-wide and 3-4-stage configurations, 5 % faster only at 8192^3: the plain order is the default
#ifndef GOAT_G2_SPREAD        // rotated loop: k-steps over which a tile's LDS-DMA instructions are issued (1..3)
#define GOAT_G2_SPREAD 1
#endif
#ifndef GOAT_G2_SETPRIO       // s_setprio 1 around every MFMA cluster
#define GOAT_G2_SETPRIO 0
#endif
#ifndef GOAT_G2_NOEPI         // experiments only: skip the epilogue (stores nothing)
#define GOAT_G2_NOEPI 0
#endif
#ifndef GOAT_G2_TIMING        // experiments only: per-wave cycle sums {wait for DMA + last fragments, barrier, compute phase, total}
#define GOAT_G2_TIMING 0      // written to `aux` as uint32[(block * waves + wave) * 4 ..] (call with epilogue NONE and an aux buffer)
#endif

template <int WM_, int WN_, int MI_, int NI_>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, MI = MI_, NI = NI_;
  static constexpr int NW = WM * WN, NTH = NW * 64;
  static constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
};
// tile ids of the C ABI (bm | bn << 16; bn = 0 means 128; GOAT_GEMM_8WAVES in nstage selects the 8-wave 128x128 tile)
typedef Cfg<2, 2, 1, 2> T64;        //  64 x 128, 4 waves
typedef Cfg<2, 2, 2, 2> T128;       // 128 x 128, 4 waves
typedef Cfg<4, 2, 1, 2> T128X8;     // 128 x 128, 8 waves (32x64 wave patches: twice the waves issue the tile's LDS-DMA)
typedef Cfg<4, 2, 2, 2> T256;       // 256 x 128, 8 waves
typedef Cfg<2, 4, 2, 2> T128x256;   // 128 x 256, 8 waves
typedef Cfg<2, 4, 3, 2> T192x256;   // 192 x 256, 8 waves  (M = 3840 = 20 x 192: 240 tiles at N = 3072)
typedef Cfg<4, 2, 2, 3> T256x192;   // 256 x 192, 8 waves  (N = 2304 = 12 x 192)
typedef Cfg<2, 4, 4, 2> T256x256;   // 256 x 256, 8 waves
typedef Cfg<2, 3, 3, 2> T192x192;   // 192 x 192, SIX waves (3840 x 2304: 20 x 12 = 240 tiles — one round on 256 CUs; 256 x 192 gives 180)
typedef Cfg<1, 4, 3, 1> T96;        //  96 x 128, 4 waves  (M = 3840 = 40 x 96: 240 tiles at N = 768 — one round on 256 CUs, against 180 tiles of 128 x 128)

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
template <int N_> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N_> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
// 16-byte result store (inline asm ends in `s_nop 1`: the compiler does not know the statement is a >64-bit VMEM store and would
// otherwise overwrite the data registers inside the store-data hazard window — seen as isolated wrong elements).
// GOAT_G2_STORE: 0 plain, 1 nt (non-temporal), 2 sc1 (write-through: the line leaves the XCD's L2 right
// away instead of in the write-back burst at the end of the kernel, MI355X_MICROARCH.md "publish-large")
#ifndef GOAT_G2_STORE
#define GOAT_G2_STORE 1       // measured: 3840x3072x768 25.9 (plain) -> 21.5 us (nt), 8640x3072x768 64.1 -> 47.7 us
#endif
typedef uint32_t g2_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(void* dst, const uint4& v) {
#if GOAT_G2_STORE == 1
  const g2_u32x4 q = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(q) : "memory");
#elif GOAT_G2_STORE == 2
  const g2_u32x4 q = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(q) : "memory");
#else
  *reinterpret_cast<uint4*>(dst) = v;
#endif
}

// Per-operand tile geometry.  ROWS x RB bytes, lane-linear LDS image, swizzled source.
template <bool T, int BMN>
struct Tile {
  static constexpr int RB = T ? BMN * 2 : BK * 2;         // bytes per LDS row
  static constexpr int ROWS = T ? BK : BMN;
  static constexpr int BYTES = ROWS * RB;                 // 16 KiB (BMN=128) / 8 KiB (BMN=64) / 32 KiB (BMN=256)
  static constexpr int NINST = BYTES / 1024;              // DMA wave-instructions per tile
  static constexpr int RPB = 256 / RB > 0 ? 256 / RB : 1; // LDS rows per 256-B bank row
  static constexpr int C64 = RB / 64;                     // 64-B chunks per row
  static_assert(!T || (C64 & (C64 - 1)) == 0, "transposed operand tiles need a power-of-two width (XOR swizzle of 64-B chunks)");

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

// Saved-activation block rows fetched ahead of the epilogue (two of them in flight); empty in instantiations without the prefetch.
template <bool ON, int CHUNKS> struct AuxRows {
  uint4 v[2][CHUNKS];
  __device__ __forceinline__ uint4* row(int i) { return v[i]; }
};
template <int CHUNKS> struct AuxRows<false, CHUNKS> {
  __device__ __forceinline__ uint4* row(int) { return nullptr; }
};

// One 32-row block row of a wave's patch of the saved pre-activation (bf16, G2Args::aux): CHUNKS 16-byte pieces per lane, rows
// and columns past the problem read as zero.
template <int CHUNKS, int CPR>
__device__ __forceinline__ void load_aux_rows(const G2Args& p, int row_w, int col_w, int lane, uint4* dst) {
  constexpr int EPC = 8;
  const bf16_t* auxp = reinterpret_cast<const bf16_t*>(p.aux);
  const bool vec = auxp != nullptr && (p.ldaux % EPC) == 0 && ((reinterpret_cast<uintptr_t>(auxp) & 15) == 0);
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int idx = c * 64 + lane, r = idx / CPR, cc = idx % CPR;
    const int row = row_w + r, col = col_w + cc * EPC;
    uint4 raw = {0u, 0u, 0u, 0u};
    if (row < p.M) {
      if (col + EPC <= p.N && vec) {
        raw = *reinterpret_cast<const uint4*>(auxp + (int64_t)row * p.ldaux + col);
      } else {
        bf16_t* rv = reinterpret_cast<bf16_t*>(&raw);
        for (int e = 0; e < EPC; ++e)
          if (col + e < p.N) rv[e] = auxp[(int64_t)row * p.ldaux + col + e];
      }
    }
    dst[c] = raw;
  }
}

template <class CF, bool TA, bool TB, int NSTAGE>
constexpr int smem_bytes() { return NSTAGE * (Tile<TA, CF::BM>::BYTES + Tile<TB, CF::BN>::BYTES); }

// One output tile of one GEMM problem.  `bid` = position of the tile in the problem's tile order, `split` = K-split index.
template <class CF, bool TA, bool TB, typename OutT, int EPI, bool SPLITK, int NSTAGE>
__device__ __forceinline__ void gemm2_tile(const G2Args& p, int bid, int split) {
#if defined(__HIP_DEVICE_COMPILE__)  // (host pass: the gfx950-only builtins below would silently drop the kernel stubs)
  constexpr int BM = CF::BM, BN = CF::BN, MI = CF::MI, NI = CF::NI, NW = CF::NW, NTH = CF::NTH, WN = CF::WN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Tile<TA, BM> TLA;
  typedef Tile<TB, BN> TLB;
  constexpr int STAGE = TLA::BYTES + TLB::BYTES;
  constexpr int WROWS = 32 * MI, WCOLS = 32 * NI;       // the wave patch
  constexpr int IPWA = TLA::NINST / NW, IPWB = TLB::NINST / NW;   // DMA wave-instructions per wave and K-tile
  static_assert(IPWA >= 1 && IPWB >= 1 && IPWA * NW == TLA::NINST && IPWB * NW == TLB::NINST,
                "every wave issues a whole number (>= 1) of DMA instructions per operand");
  constexpr int LOADS = IPWA + IPWB;
  constexpr int KSTEPS = BK / 16;
  // Where a K-tile's DMA instructions are issued: deep rings (>= 3 stages) have a whole K-tile of slack, so the instructions
  // go behind the MFMAs of all four k-steps (each costs the issuing wave ~60-100 cycles); a 2-stage ring waits for the tile
  // at the very next barrier.  There the small tiles issue everything right after the barrier (measured +4...+17 % if
  // spread), the big ones (>= 2048 MFMA cycles per K-tile) spread over the first two k-steps.
  constexpr bool BIG = (MI * NI >= 6);
  constexpr bool INTERLEAVE = NSTAGE >= 3 || BIG;
  constexpr int SPREAD = NSTAGE >= 3 ? KSTEPS : 2;
  // fragment prefetch distance in k-steps (each k-step's fragments have their own registers); bounded by the 4-bit lgkmcnt
  constexpr bool SWAP = !SPLITK && sizeof(OutT) == 2;                  // bf16 results: swapped MFMA operand roles (see the epilogue)
  constexpr int RD = MI * (TA ? 2 : 1) + NI * (TB ? 2 : 1);            // ds_read instructions per k-step
  // (the counted wait in front of k-step ks allows (FD-1)*RD reads to stay in flight: THAT must fit the 4-bit lgkmcnt; with more
  // than 15 reads issued the wave simply stalls at issue.  Round 1 required FD*RD <= 15 and ran every transposed-operand tile
  // — all weight gradients — one k-step ahead only.)
  // RD = 16 (the 4 x 4 patch with both operands transposed): one k-step ahead, the wait count clamped to 15 — LDS reads return
  // in order, so waiting for "at most 15 outstanding" completes every read of the current k-step and the first of the next.
  constexpr int FD = RD == 16 ? 2 : ((GOAT_GEMM_FRAG_DEPTH - 1) * RD <= 15) ? GOAT_GEMM_FRAG_DEPTH : (15 / RD + 1 >= 1 ? (15 / RD + 1 < 3 ? 15 / RD + 1 : 3) : 1);
  static_assert(RD <= 16 && ((FD - 1) * RD <= 15 || RD == 16), "the reads a counted wait leaves in flight must fit the lgkmcnt counter");

  // `wave` through readfirstlane: the compiler then keeps every wave-uniform quantity (the LDS addresses of this wave's DMA
  // pieces, hence M0) in SGPRs instead of a v_add + v_readfirstlane + s_mov chain in front of every buffer_load ... lds
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int hi = lane >> 5, l31 = lane & 31;

  // `bid` walks a contiguous chunk per XCD (each XCD has its own L2).  Inside the chunk tiles are visited
  // column-major within groups of group_m tile rows, so the workgroups an XCD runs at a time cover a compact
  // group_m x (n/group_m) block: its A and B panels are fetched into that L2 once, and the 8 XCD chunks form a 2-D
  // partition of C instead of 8 full-width stripes.
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
  // one DMA wave-instruction (number j_ of this wave's LOADS) of K-tile t_
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

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_void*)smem;  // LDS byte offset of the dynamic region
  const int t15 = lane & 15, g = lane >> 4;
  const bool do_colsum = TA && p.colsum != nullptr && tn == 0 && wn == 0;
  float bsum[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) bsum[i] = 0.f;
  bf16x8 fa[KSTEPS][MI], fb[KSTEPS][NI];

  // fragments of k-step ks_ of the stage at (sa_, sb_) -> fa[buf_], fb[buf_]
#define GOAT_LOAD_FRAGS(sa_, sb_, ks_, buf_)                                                                         \
  do {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                                \
      if (!TA) {                                                                                                    \
        uint4 v = lds_read_b128(frag_addr_n((sa_), wm * WROWS + i * 32 + l31, (ks_), hi));                        \
        fa[buf_][i] = *reinterpret_cast<bf16x8*>(&v);                                                               \
      } else {                                                                                                      \
        const int col = wm * WROWS + i * 32 + (g & 1) * 16 + (t15 & 3) * 4;                                      \
        const int kr = (ks_) * 16 + 8 * (g >> 1) + (t15 >> 2);                                                      \
        uint2 v0 = lds_read_tr16(frag_addr_t<TLA::RB, TLA::RPB, TLA::C64>((sa_), kr, col));                         \
        uint2 v1 = lds_read_tr16(frag_addr_t<TLA::RB, TLA::RPB, TLA::C64>((sa_), kr + 4, col));                     \
        uint4 v = {v0.x, v0.y, v1.x, v1.y};                                                                         \
        fa[buf_][i] = *reinterpret_cast<bf16x8*>(&v);                                                               \
      }                                                                                                             \
    }                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                                                \
      if (!TB) {                                                                                                    \
        uint4 v = lds_read_b128(frag_addr_n((sb_), wn * WCOLS + j * 32 + l31, (ks_), hi));                          \
        fb[buf_][j] = *reinterpret_cast<bf16x8*>(&v);                                                               \
      } else {                                                                                                      \
        const int col = wn * WCOLS + j * 32 + (g & 1) * 16 + (t15 & 3) * 4;                                         \
        const int kr = (ks_) * 16 + 8 * (g >> 1) + (t15 >> 2);                                                      \
        uint2 v0 = lds_read_tr16(frag_addr_t<TLB::RB, TLB::RPB, TLB::C64>((sb_), kr, col));                         \
        uint2 v1 = lds_read_tr16(frag_addr_t<TLB::RB, TLB::RPB, TLB::C64>((sb_), kr + 4, col));                     \
        uint4 v = {v0.x, v0.y, v1.x, v1.y};                                                                         \
        fb[buf_][j] = *reinterpret_cast<bf16x8*>(&v);                                                               \
      }                                                                                                             \
    }                                                                                                               \
  } while (0)
#define GOAT_MMA(ks_)                                                                                              \
  do {                                                                                                              \
    if (TA && do_colsum) {       /* bias gradient: v_dot2c_f32_bf16 (pair . (1, 1) + acc), four per fragment instead of eight conversions + eight adds (round 6) */ \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                \
        _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                                          \
          const bf16x2 pr_ = {fa[ks_][i][2 * e2], fa[ks_][i][2 * e2 + 1]};                                          \
          bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pr_, bf16x2{(bf16_t)1.0f, (bf16_t)1.0f}, bsum[i], false);       \
        }                                                                                                           \
    }                                                                                                               \
    if (GOAT_G2_SETPRIO) __builtin_amdgcn_s_setprio(1);                                                             \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                  \
      _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                                              \
        if (SWAP) mma32(acc[i][j], fb[ks_][j], fa[ks_][i]);   /* lane = row of C, registers = columns (bf16 epilogue) */ \
        else mma32(acc[i][j], fa[ks_][i], fb[ks_][j]);        /* lane = column, registers = rows (f32 stores / atomics) */ \
      }                                                                                                             \
    if (GOAT_G2_SETPRIO) __builtin_amdgcn_s_setprio(0);                                                             \
  } while (0)

#if GOAT_G2_ROTATED
  // ---- rotated software pipeline -------------------------------------------------------------------------------
  // The single barrier of a K-tile sits at the top of its LAST k-step: by then every wave holds the tile's last fragments in
  // registers, so the stage is free one k-step before its MFMAs end.  Behind that barrier a wave (1) reads the first FD
  // k-steps of fragments of the NEXT tile, (2) issues the MFMAs of the last k-step of THIS tile, which cover the latency
  // of (1), and (3) issues the LDS-DMA of tile t+NSTAGE into the stage just freed, so that a DMA has NSTAGE-1 whole K-tiles
  // to land even in a 2-stage ring.  (The round-1 loop had its barrier between K-tiles: every wave met it with nothing
  // queued on the MFMA pipe, then waited for the first fragment reads — 400-600 of ~2900 cycles per K-tile.)
  constexpr int DSPREAD = GOAT_G2_SPREAD;      // k-steps the DMA instructions of one tile are spread over (1: all at once)
  static_assert(DSPREAD >= 1 && DSPREAD <= 3, "the DMA of a tile is issued within the three k-steps that follow the barrier");
#pragma unroll
  for (int t = 0; t < NSTAGE; ++t) GOAT_ISSUE(t);                   // every stage is free at the start
  wait_vm<(NSTAGE - 1) * LOADS>();
  __builtin_amdgcn_s_barrier();                                      // tile 0 is visible to every wave
#pragma unroll
  for (int d = 0; d < FD; ++d) {
    if (d == 0) GOAT_LOAD_FRAGS(smem_base, smem_base + TLA::BYTES, 0, 0);
    if (d == 1) GOAT_LOAD_FRAGS(smem_base, smem_base + TLA::BYTES, 1, 1);
    if (d == 2) GOAT_LOAD_FRAGS(smem_base, smem_base + TLA::BYTES, 2, 2);
  }
  static_assert(KSTEPS == 4 && FD >= 1 && FD <= 3, "k-step unrolling below is written for BK = 64");
  // DMA instruction j_ of the tile whose issue started in iteration (t_) - slot: slot 0 = right behind the barrier
#define GOAT_DMA_SLOT(slot_)                                                                                        \
  do {                                                                                                              \
    if ((slot_) < DSPREAD && ((slot_) == 0 || t > 0)) {                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      _Pragma("unroll") for (int j = 0; j < LOADS; ++j)                                                             \
        if (j * DSPREAD / LOADS == (slot_)) GOAT_ISSUE_ONE(t + NSTAGE - ((slot_) == 0 ? 0 : 1), j);               \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
    }                                                                                                               \
  } while (0)
#define GOAT_KSTEP(ks_)                                                                                            \
  do {                                                                                                              \
    constexpr int left_ = (KSTEPS - (ks_) < FD ? KSTEPS - (ks_) : FD) - 1; /* later k-steps whose reads may stay in flight */ \
    wait_lgkm<(left_ * RD > 15 ? 15 : left_ * RD)>();                                                                                        \
    if ((ks_) + FD < KSTEPS) GOAT_LOAD_FRAGS(sa, sb, (ks_) + FD < KSTEPS ? (ks_) + FD : 0, (ks_) + FD < KSTEPS ? (ks_) + FD : 0); \
    GOAT_MMA(ks_);                                                                                                  \
    GOAT_DMA_SLOT((ks_) + 1);                                                                                       \
  } while (0)
#if GOAT_G2_TIMING
  uint32_t tm_wait = 0, tm_bar = 0, tm_comp = 0;
  const uint32_t tm_start = (uint32_t)__builtin_amdgcn_s_memtime();
  uint32_t tm_c = tm_start;
#endif
  for (int t = 0; t < nkt; ++t) {
    const uint32_t sa = smem_base + (t % NSTAGE) * STAGE, sb = sa + TLA::BYTES;
    const uint32_t na = smem_base + ((t + 1) % NSTAGE) * STAGE, nb = na + TLA::BYTES;
    GOAT_KSTEP(0);
    GOAT_KSTEP(1);
    GOAT_KSTEP(2);
#if GOAT_G2_TIMING
    const uint32_t tm_a = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    wait_lgkm<0>();                                     // the last fragments of tile t are in registers: this wave is done with the stage
    wait_vm<(NSTAGE - 2) * LOADS>();                    // this wave's pieces of tile t+1 have landed
#if GOAT_G2_TIMING
    const uint32_t tm_b = (uint32_t)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
#if GOAT_G2_TIMING
    const uint32_t tm_c2 = (uint32_t)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tm_comp += tm_a - tm_c; tm_wait += tm_b - tm_a; tm_bar += tm_c2 - tm_b; tm_c = tm_c2;
#endif
#pragma unroll
    for (int d = 0; d < FD; ++d) {
      if (d == 0) GOAT_LOAD_FRAGS(na, nb, 0, 0);
      if (d == 1) GOAT_LOAD_FRAGS(na, nb, 1, 1);
      if (d == 2) GOAT_LOAD_FRAGS(na, nb, 2, 2);
    }
    GOAT_MMA(3);
    GOAT_DMA_SLOT(0);      // tile t+NSTAGE -> the stage just freed (beyond the end: harmless, bounds-checked or never read)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef GOAT_DMA_SLOT
#else
  // ---- round-1 loop: barrier between K-tiles ---------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t) GOAT_ISSUE(t);
#if GOAT_G2_TIMING
  uint32_t tm_wait = 0, tm_bar = 0, tm_comp = 0;
  const uint32_t tm_start = (uint32_t)__builtin_amdgcn_s_memtime();
  uint32_t tm_c = tm_start;
#endif
  for (int t = 0; t < nkt; ++t) {
#if GOAT_G2_TIMING
    const uint32_t tm_a = (uint32_t)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    wait_vm<(NSTAGE - 2) * LOADS>();
#if GOAT_G2_TIMING
    const uint32_t tm_b = (uint32_t)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
#if GOAT_G2_TIMING
    const uint32_t tm_c2 = (uint32_t)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tm_comp += tm_a - tm_c; tm_wait += tm_b - tm_a; tm_bar += tm_c2 - tm_b; tm_c = tm_c2;
#endif
    if (!INTERLEAVE) GOAT_ISSUE(t + NSTAGE - 1);
    const uint32_t sa = smem_base + (t % NSTAGE) * STAGE;
    const uint32_t sb = sa + TLA::BYTES;
#define GOAT_KSTEP(ks_)                                                                                            \
  do {                                                                                                              \
    constexpr int left_ = (KSTEPS - (ks_) < FD ? KSTEPS - (ks_) : FD) - 1;                                          \
    wait_lgkm<(left_ * RD > 15 ? 15 : left_ * RD)>();                                                                                        \
    if ((ks_) + FD < KSTEPS) GOAT_LOAD_FRAGS(sa, sb, (ks_) + FD < KSTEPS ? (ks_) + FD : 0, (ks_) + FD < KSTEPS ? (ks_) + FD : 0); \
    GOAT_MMA(ks_);                                                                                                  \
    if (INTERLEAVE) {                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      _Pragma("unroll") for (int j = 0; j < LOADS; ++j)                                                             \
        if (j * SPREAD / LOADS == (ks_)) GOAT_ISSUE_ONE(t + NSTAGE - 1, j);                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
    }                                                                                                               \
  } while (0)
#pragma unroll
    for (int d = 0; d < FD; ++d) {
      if (d == 0) GOAT_LOAD_FRAGS(sa, sb, 0, 0);
      if (d == 1) GOAT_LOAD_FRAGS(sa, sb, 1, 1);
      if (d == 2) GOAT_LOAD_FRAGS(sa, sb, 2, 2);
    }
    static_assert(KSTEPS == 4 && FD >= 1 && FD <= 3, "k-step unrolling below is written for BK = 64");
    GOAT_KSTEP(0);
    GOAT_KSTEP(1);
    GOAT_KSTEP(2);
    GOAT_KSTEP(3);
  }
#endif
#undef GOAT_MMA
  // activation-derivative epilogues (FFN dgrad): the saved pre-activation of this wave's patch is fetched one 32-row block row
  // AHEAD of its use — block row 0 here, behind the last MFMAs and across the pipeline drain, block row i + 1 while block row i is
  // being combined and stored — so that only the first of the MI dependent HBM round trips is (partly) exposed.  Round 2 loaded each
  // block row right where it was needed: 399 TFLOP/s on 3840 x 3072 x 768 against 548 for the same shape without the multiply.
  constexpr bool DACT_ = !SPLITK && sizeof(OutT) == 2 && (EPI == GOAT_EPI_MUL_DGELU || EPI == GOAT_EPI_MUL_DRELU);
  constexpr int EPC_ = 8, CPR_ = WCOLS / EPC_, CHUNKS_ = 32 * CPR_ / 64;
  // (only in instantiations whose accumulators leave room for two block rows of it, and through a free function: the first version
  //  of this — a [&] lambda defined here in every instantiation — cost the 256 x 256 transposed-operand kernels 5 to 162 spilled
  //  registers although they never call it, and the grouped weight-gradient launch of that tile went from 297 to 446 us;
  //  profiles/round3_gemm_spill_check.txt)
  constexpr bool PF_ = DACT_ && (MI * NI * 16 + 2 * CHUNKS_ * 4 <= 128);
  AuxRows<PF_, CHUNKS_> auxr;
  if constexpr (PF_) load_aux_rows<CHUNKS_, CPR_>(p, m0 + wm * WROWS, n0 + wn * WCOLS, lane, auxr.row(0));
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
#undef GOAT_ISSUE
#undef GOAT_ISSUE_ONE
#undef GOAT_LOAD_FRAGS
#undef GOAT_KSTEP

  if (TA && do_colsum) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      const int row = m0 + wm * WROWS + i * 32 + l31;
      if (hi == 0 && row < p.M) atomicAdd(p.colsum + row, v);
    }
  }

#if GOAT_G2_TIMING
  if (p.aux != nullptr && lane == 0 && EPI == GOAT_EPI_NONE) {
    uint32_t* o = reinterpret_cast<uint32_t*>(p.aux) + ((size_t)blockIdx.x * NW + wave) * 4;
    o[0] = tm_wait; o[1] = tm_bar; o[2] = tm_comp; o[3] = (uint32_t)__builtin_amdgcn_s_memtime() - tm_start;
  }
#endif
  // ------------------------------------------------------------------ epilogue
  if (GOAT_G2_NOEPI) {      // keep every accumulator live (an unused one would take its MFMAs with it)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  const int wrow0 = wm * WROWS, wcol0 = wn * WCOLS;
  if (SPLITK) {
    float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = n0 + wcol0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wrow0 + i * 32 + c_row(r, lane);
          if (row < p.M && col < p.N) atomicAdd(C + (int64_t)row * p.ldc + col, acc[i][j][r]);
        }
      }
    return;
  }
  if (sizeof(OutT) == 4) {  // f32 output: 32 lanes = 128 contiguous bytes per row, store straight from the accumulators
    float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = n0 + wcol0 + j * 32 + l31;
        const float bcol = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wrow0 + i * 32 + c_row(r, lane);
          if (row < p.M && col < p.N) {
            float* dst = C + (int64_t)row * p.ldc + col;
            const float u = acc[i][j][r] + bcol;
            if (p.accum) *dst = u + *dst;
            else if (GOAT_G2_STORE) __builtin_nontemporal_store(u, dst);      // streamed out of L2 (see store16)
            else *dst = u;
          }
        }
      }
    return;
  }
  // bf16 output.  The MFMA operand roles are swapped for these kernels (SWAP: D = B·A^T, i.e. lane = row of C, the 16
  // registers of a block = 4 groups of 4 consecutive columns), so a lane packs 4 results into 8 bytes.  Each wave stages
  // one 32-row block row of its patch at a time through its OWN slice of the (now free) LDS ring — no workgroup barrier:
  // a wave starts storing the moment its last MFMA retires — and writes it out as whole rows of the patch (128-B / 192-B
  // segments, 16 B per lane).  Round 1 staged the whole tile with 2-byte LDS writes between two __syncthreads():
  // 9.6 of the 26.8 us of the 3840x3072x768 launch (profiles/round2_gemm_epilogue_ab.txt).
  typedef bf16_t T;
  constexpr int EPC = 8;
  constexpr int RBY = WCOLS * 2 + 16;          // staging row stride in bytes (16 B pad: the 8-byte writes of 16 rows hit 16 bank pairs)
  constexpr int WSLICE = 32 * RBY;             // per-wave staging slice: 4.5 KiB (64 columns) / 6.5 KiB (96)
  static_assert(NW * WSLICE <= NSTAGE * STAGE, "per-wave epilogue slices must fit the LDS ring");
  constexpr int CPR = WCOLS / EPC;             // 16-byte chunks per patch row
  constexpr int CHUNKS = 32 * CPR / 64;        // chunks per lane and block row
  static_assert(32 * CPR % 64 == 0, "a block row is a whole number of 16-byte chunks per lane");
  constexpr bool DACT = (EPI == GOAT_EPI_MUL_DGELU || EPI == GOAT_EPI_MUL_DRELU);
  constexpr bool ACT = (EPI == GOAT_EPI_GELU || EPI == GOAT_EPI_RELU);
  T* aux = reinterpret_cast<T*>(p.aux);
  T* C = reinterpret_cast<T*>(p.C);
  const bool c_vec = (p.ldc % EPC) == 0 && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  const bool aux_vec = aux != nullptr && (p.ldaux % EPC) == 0 && ((reinterpret_cast<uintptr_t>(aux) & 15) == 0);
  const uint32_t ws = smem_base + wave * WSLICE;                 // this wave's slice (LDS byte address)
  char* wsp = smem + wave * WSLICE;
  const int col_w = n0 + wn * WCOLS;                             // first column of the wave patch
  // bias of this lane's columns: block j, group q -> columns j*32 + 4*hi + 8*q + {0..3}
  f32x4 bv[NI][4];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = col_w + j * 32 + 4 * hi + 8 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[j][q][e] = (p.bias != nullptr && col + e < p.N) ? p.bias[col + e] : 0.f;
    }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row_w = m0 + wm * WROWS + i * 32;                  // first row of this block row
    if (DACT) {
      // the saved pre-activation block row (fetched one block row ahead, see above) -> slice -> each lane picks up its own
      // 4-column groups; the next block row's loads are issued first and stay in flight behind this one's arithmetic and stores
      if constexpr (PF_) {
        if (i + 1 < MI) load_aux_rows<CHUNKS, CPR>(p, row_w + 32, col_w, lane, auxr.row((i + 1) & 1));
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
          const int idx = c * 64 + lane, r = idx / CPR, cc = idx % CPR;
          *reinterpret_cast<uint4*>(wsp + r * RBY + cc * 16) = auxr.row(i & 1)[c];
        }
      } else {           // large accumulator tiles: the block row is fetched where it is used (round-2 behaviour)
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
          const int idx = c * 64 + lane, r = idx / CPR, cc = idx % CPR;
          uint4 now;
          load_aux_rows<1, CPR>(p, row_w, col_w, idx, &now);     // (chunk c of this lane = chunk 0 of "lane" idx)
          *reinterpret_cast<uint4*>(wsp + r * RBY + cc * 16) = now;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS hand-over between lanes
    }
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        char* slot = wsp + l31 * RBY + (j * 32 + 4 * hi + 8 * q) * 2;
        float u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = acc[i][j][4 * q + e] + bv[j][q][e];
        if (DACT) {
          const bf16x4 a4 = *reinterpret_cast<const bf16x4*>(slot);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float av = (float)a4[e];
            u[e] = (EPI == GOAT_EPI_MUL_DGELU) ? u[e] * dgelu_fast(av) : (av > 0.f ? u[e] : 0.f);
          }
        }
        bf16x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (bf16_t)u[e];
        *reinterpret_cast<bf16x4*>(slot) = o4;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS hand-over between lanes
    // write-out: whole patch rows, 16 bytes per lane.  Activation epilogues store the staged pre-activation to `aux` (if
    // given) and the activation of the same bf16 values to C from this one pass.
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      const int idx = c * 64 + lane, r = idx / CPR, cc = idx % CPR;
      const int row = row_w + r, col = col_w + cc * EPC;
      uint4 raw = *reinterpret_cast<const uint4*>(wsp + r * RBY + cc * 16);
      if (row >= p.M || col >= p.N) continue;
      if (ACT) {
        if (aux != nullptr) {
          if (col + EPC <= p.N && aux_vec) {
            store16(aux + (int64_t)row * p.ldaux + col, raw);
          } else {
            const T* rv = reinterpret_cast<const T*>(&raw);
            for (int e = 0; e < EPC; ++e)
              if (col + e < p.N) aux[(int64_t)row * p.ldaux + col + e] = rv[e];
          }
        }
        bf16x8 v = *reinterpret_cast<bf16x8*>(&raw);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float u = (float)v[e];
          const float h = (EPI == GOAT_EPI_GELU) ? gelu_fast(u) : fmaxf(u, 0.f);
          v[e] = (bf16_t)h;
        }
        raw = *reinterpret_cast<uint4*>(&v);
      }
      if (col + EPC <= p.N && c_vec) {
        store16(C + (int64_t)row * p.ldc + col, raw);
      } else {
        const T* rv = reinterpret_cast<const T*>(&raw);
        for (int e = 0; e < EPC; ++e)
          if (col + e < p.N) C[(int64_t)row * p.ldc + col + e] = rv[e];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS hand-over between lanes      // the slice is rewritten by the next block row
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <class CF, bool TA, bool TB, typename OutT, int EPI, bool SPLITK, int NSTAGE>
__global__ __launch_bounds__(CF::NTH) void gemm2_kernel(G2Args p) {
  gemm2_tile<CF, TA, TB, OutT, EPI, SPLITK, NSTAGE>(p, xcd_chunk_position(blockIdx.x, gridDim.x), blockIdx.y);
}

// Grouped weight-gradient launch: up to GROUP_MAX (24) independent TN problems (dW_i = dY_i^T · X_i, float32 out, unsplit)
// share one grid, so the many small weight gradients of a layer fill the chip together instead of each being split
// along the contraction (atomics + a zero fill) to do so.  Problem i owns tiles [tile_start[i], tile_start[i+1]).
constexpr int GROUP_MAX = 48;   // (the argument block stays under the 4 KiB kernel-argument limit)
// One problem of a grouped launch, 64 bytes: what cannot be recomputed on the device.  (Round 1-5 passed a whole G2Args — 120 bytes —
// per problem: 24 problems per launch; the text encoder's 47 weight gradients then took two launches, each with its own ramp and
// tail: 16 -> 24 problems per launch was worth 1 % of the step, profiles/round5_wgrad_group_max.txt.)
struct GroupProb {
  const void* A; const void* B; void* C; float* colsum;
  int lda, ldb, ldc;
  int M, N, Kc;
  short accum, group_m;
  int pad_;
  template <int BM, int BN>
  __host__ __device__ __forceinline__ G2Args expand() const {
    G2Args a;
    a.A = A; a.B = B; a.C = C; a.bias = nullptr; a.aux = nullptr;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = 0;
    a.M = M; a.N = N; a.Kc = Kc;
    a.tiles_m = (M + BM - 1) / BM;
    a.tiles_n = (N + BN - 1) / BN;
    a.k_tiles_per_split = (Kc + BK - 1) / BK;
    a.a_bytes = (uint32_t)Kc * (uint32_t)lda * 2u;      // transposed operands: Kc rows of lda / ldb elements (checked < 2 GiB on the host)
    a.b_bytes = (uint32_t)Kc * (uint32_t)ldb * 2u;
    a.colsum = colsum;
    a.accum = accum;
    a.group_m = group_m;
    return a;
  }
};
static_assert(sizeof(GroupProb) == 64, "GroupProb is 64 bytes");
static_assert(sizeof(GroupProb) * GROUP_MAX + 4 * (2 * GROUP_MAX + 8) + 32 <= 4000, "GroupArgs (and the balanced launch's iteration table) must fit the kernel-argument segment");
struct GroupArgs {
  GroupProb prob[GROUP_MAX];
  int tile_start[GROUP_MAX + 1];
  int n;
};
template <class CF, int NSTAGE>
__global__ __launch_bounds__(CF::NTH) void gemm2_group_kernel(GroupArgs g) {
  const int pos = xcd_chunk_position(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i)
    if (i < g.n && pos >= g.tile_start[i]) pi = i;
  const G2Args p = g.prob[pi].template expand<CF::BM, CF::BN>();
  gemm2_tile<CF, true, true, float, GOAT_EPI_NONE, false, NSTAGE>(p, pos - g.tile_start[pi], 0);
}

template <class CF, bool TA, bool TB, typename OutT, int EPI, bool SPLITK, int NSTAGE>
int launch2s(hipStream_t st, const G2Args& a, int split) {
  constexpr int SMEM = smem_bytes<CF, TA, TB, NSTAGE>();
  static_assert(SMEM <= 160 * 1024, "LDS ring exceeds the CU's 160 KiB");
  auto kern = gemm2_kernel<CF, TA, TB, OutT, EPI, SPLITK, NSTAGE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(a.tiles_m * a.tiles_n, SPLITK ? split : 1);
  hipLaunchKernelGGL(kern, grid, dim3(CF::NTH), SMEM, st, a);
  GOAT_LAUNCH_CHECK();
  return 0;
}

// ring depths a tile supports: whatever fits 160 KiB of LDS, 2..4
template <class CF, bool TA, bool TB, typename OutT, int EPI, bool SPLITK>
int launch2(hipStream_t st, const G2Args& a, int split, int nstage) {
  constexpr int STAGE = smem_bytes<CF, TA, TB, 1>();
  if (nstage == 2) return launch2s<CF, TA, TB, OutT, EPI, SPLITK, 2>(st, a, split);
  if constexpr (3 * STAGE <= 160 * 1024) {
    if (nstage == 3) return launch2s<CF, TA, TB, OutT, EPI, SPLITK, 3>(st, a, split);
  }
  if constexpr (4 * STAGE <= 160 * 1024 && CF::BM * CF::BN <= 128 * 128) {
    if (nstage == 4) return launch2s<CF, TA, TB, OutT, EPI, SPLITK, 4>(st, a, split);
  }
  return GOAT_E_ARG;
}

template <class CF, bool TA, bool TB>
int dispatch2(hipStream_t st, const G2Args& a, int dtype_out, int epi, int split, int nstage) {
  if (split > 1) return launch2<CF, TA, TB, float, GOAT_EPI_NONE, true>(st, a, split, nstage);
  if (dtype_out == GOAT_F32) {
    if (epi != GOAT_EPI_NONE && epi != GOAT_EPI_ACCUM) return GOAT_E_ARG;
    return launch2<CF, TA, TB, float, GOAT_EPI_NONE, false>(st, a, 1, nstage);
  }
  if constexpr (TA) {      // weight-gradient layout: bf16 results only without an epilogue (the activation epilogues belong to
    if (epi != GOAT_EPI_NONE) return GOAT_E_ARG;      // forward / dgrad; not instantiating them saves a quarter of the build)
    return launch2<CF, TA, TB, bf16_t, GOAT_EPI_NONE, false>(st, a, 1, nstage);
  }
  switch (epi) {
    case GOAT_EPI_NONE: return launch2<CF, TA, TB, bf16_t, GOAT_EPI_NONE, false>(st, a, 1, nstage);
    case GOAT_EPI_GELU: return launch2<CF, TA, TB, bf16_t, GOAT_EPI_GELU, false>(st, a, 1, nstage);
    case GOAT_EPI_RELU: return launch2<CF, TA, TB, bf16_t, GOAT_EPI_RELU, false>(st, a, 1, nstage);
    case GOAT_EPI_MUL_DGELU: return launch2<CF, TA, TB, bf16_t, GOAT_EPI_MUL_DGELU, false>(st, a, 1, nstage);
    case GOAT_EPI_MUL_DRELU: return launch2<CF, TA, TB, bf16_t, GOAT_EPI_MUL_DRELU, false>(st, a, 1, nstage);
  }
  return GOAT_E_ARG;
}

// all operand layouts a tile supports (transposed operands need a power-of-two tile width on that side)
template <class CF>
int dispatch_layout(hipStream_t st, const G2Args& a, int trans_a, int trans_b, int dtype_out, int epi, int split, int nstage) {
  constexpr bool POW2_M = (CF::BM & (CF::BM - 1)) == 0, POW2_N = (CF::BN & (CF::BN - 1)) == 0;
  if (!trans_a && !trans_b) return dispatch2<CF, false, false>(st, a, dtype_out, epi, split, nstage);
  if constexpr (POW2_N) {
    if (!trans_a && trans_b) return dispatch2<CF, false, true>(st, a, dtype_out, epi, split, nstage);
    if constexpr (POW2_M) {
      if (trans_a && trans_b) return dispatch2<CF, true, true>(st, a, dtype_out, epi, split, nstage);
    }
  }
  return GOAT_E_ARG;
}

template <class CF, int NSTAGE>
int launch_group(hipStream_t st, const GroupArgs& g) {
  constexpr int SMEM = smem_bytes<CF, true, true, NSTAGE>();
  auto kern = gemm2_group_kernel<CF, NSTAGE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.tile_start[g.n]), dim3(CF::NTH), SMEM, st, g);
  GOAT_LAUNCH_CHECK();
  return 0;
}

}  // namespace goat_g2

// gemm3.hip: the 8-wave 192/256-wide tiles (tile = bm | bn << 16)
int goat_g3_dispatch(hipStream_t st, const goat_g2::G2Args& a, int bm, int bn, int trans_a, int trans_b, int dtype_out, int epi,
                     int split, int nstage);
int goat_g3_group(hipStream_t st, const goat_g2::GroupArgs& g, int bm, int bn, int nstage);
