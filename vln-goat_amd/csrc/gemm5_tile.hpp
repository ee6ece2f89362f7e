// Device code of the "ping-pong" bf16 MFMA GEMM tile for gfx950 (goat_gemm_bf16 / goat_wgrad_grouped with nstage | GOAT_GEMM_PP).
//
//   C[M,N] = epilogue( op(A) · op(B)^T ),  same operand layouts, argument block (G2Args), tile order and epilogues as
//   gemm2_tile.hpp; what differs is the main loop.
//
// Why a second main loop.  gemm2_tile.hpp runs its eight waves in lockstep: one s_barrier per K-tile, after which every wave
// first issues its LDS-DMA and fragment reads and then its MFMAs.  The two waves that share a SIMD therefore want the memory
// pipes at the same time and the matrix pipe at the same time; cycle stamps (profiles/round2_gemm_mainloop_cycle_stamps.txt)
// show the older wave of a SIMD computing for ~1700 cycles of a 2048-cycle K-tile and then waiting ~900 cycles at the barrier
// for the younger one: 75 % MFMA efficiency between barriers at 8192^3 on the 256 x 256 tile.
//
// Here the workgroup (512 threads) is two GROUPS of four waves — wave w and wave w + 4 share SIMD w % 4 — that run half a
// K-tile period out of phase, separated by one s_barrier per SLOT:
//
//        slot      2t                2t+1               2t+2               2t+3
//   group 0   MEM  (tile t)     MFMA (tile t)      MEM  (tile t+1)    MFMA (tile t+1)
//   group 1   MFMA (tile t-1)   MEM  (tile t)      MFMA (tile t)      MEM  (tile t+1)
//
//   MEM  = issue this group's LDS-DMA pieces of a future K-tile, read ALL fragments of tile t (4 k-steps) into registers,
//          s_waitcnt lgkmcnt(0);
//   MFMA = nothing but the tile's MI*NI*4 v_mfma_f32_32x32x16_bf16 under s_setprio 1, then s_waitcnt vmcnt(0) for the pieces
//          issued one slot earlier (they had a whole MFMA phase to land).
// On every SIMD one wave feeds the matrix pipe while its partner uses the LDS / texture path, in every slot.
//
// The 64*MI x 128*NI tile is cut so that the groups share only the B block: group g owns tile rows [g*32*MI, (g+1)*32*MI)
// (its own A half-block, A_g) and all columns (wave patch 32*MI x 32*NI).  LDS: two buffers per A half-block, NB (2 or 3) buffers
// for B.  Who loads what, and when a buffer is free:
//   B[t]   is read by group 0 in slot 2t and by group 1 in slot 2t+1  -> group 0 issues B[t+1] (NB = 2) at the top of slot 2t,
//          waits for it at the end of slot 2t+1;
//   A_0[t] is read by group 0 in slot 2t                              -> group 1 issues A_0[t+2] in slot 2t+1 (its MEM phase);
//   A_1[t] is read by group 1 in slot 2t+1                            -> group 1 issues A_1[t+1] in slot 2t+1 (other buffer).
// Every wave issues the same number of pieces per period (group 0: the B block, group 1: both A half-blocks) and every piece has
// at least one full MFMA phase (>= 1024 cycles on 256 x 256) in flight before anyone waits for it.
#pragma once
#include "gemm2_tile.hpp"

namespace goat_g5 {
using namespace goat_g2;

#ifndef GOAT_G5_SETPRIO
#define GOAT_G5_SETPRIO 1
#endif

#ifndef GOAT_G5_TIMING       // experiments only: per-wave cycle sums of the phases, written to `aux` as uint32[(block * 8 + wave) * 8 + i],
#define GOAT_G5_TIMING 0     // i = {DMA issue, fragment reads, barrier after MEM, MFMAs, vmcnt wait, barrier after MFMA, total} (epilogue NONE)
#endif
__device__ __forceinline__ uint32_t g5_now() {
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return (uint32_t)t;
}

// VAR (experiments; the product instantiates the measured best): bits 0-1 = where the MEM phase issues its LDS-DMA pieces
//   0: all pieces first, then the fragment reads      1: one piece after every three fragment reads
//   2: all fragment reads first, then the pieces (before the lgkmcnt wait)
template <int MI_, int NI_, int NB_, int VAR_ = 0>
struct PCfg {
  static constexpr int MI = MI_, NI = NI_, NB = NB_, VAR = VAR_;
  static constexpr int BM = 64 * MI, BN = 128 * NI, NTH = 512;
  static constexpr int AH = MI * 4096;     // bytes of one A half-block (32*MI rows x 64 k)
  static constexpr int BSZ = NI * 16384;   // bytes of one B block (128*NI columns x 64 k)
  static constexpr int SMEM = 4 * AH + NB * BSZ;
};
// Product tiles (DMA placement 1: measured equal to or ahead of the other two with the cycle-stamp harness scripts/gemm_pp_stamps.cpp; a third B
// buffer — PCfg<4, 2, 3>, 160 KiB — buys nothing: the pieces already have a whole MFMA phase to land).
typedef PCfg<4, 2, 2, 1> P256x256;   // 128 KiB
typedef PCfg<3, 2, 2, 1> P192x256;   // 112 KiB  (M = 3840 = 20 x 192: 240 tiles at N = 3072; K-contiguous A only)
typedef PCfg<2, 2, 2, 1> P128x256;   //  96 KiB
typedef PCfg<4, 1, 2, 1> P256x128;   //  96 KiB
typedef PCfg<2, 1, 2, 1> P128x128;   //  64 KiB (two workgroups per CU)

template <int OFF> __device__ __forceinline__ uint4 lds_read_b128_o(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int OFF> __device__ __forceinline__ uint2 lds_read_tr16_o(uint32_t addr) {
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

// One operand block of a K-tile: R "major" elements (rows of a K-contiguous operand / columns of a transposed one) x 64 k,
// R*128 bytes, moved as 1-KiB LDS-DMA pieces.  LDS images are lane-linear (DMA constraint); bank conflicts of the fragment reads
// are removed by XOR-swizzling the SOURCE address of every 16-byte (K-contiguous) / 64-byte (transposed) chunk and applying the
// same involution in the fragment read address (gemm2_tile.hpp uses the same images).
//   K-contiguous: image [R rows][128 B]; 16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 7).
//   transposed  : image [64 k-rows][W = 2R bytes]; 64-byte chunk c of k-row kr is stored at chunk c ^ x(kr),
//                 x = kr & 3 (W >= 256: four k-rows of a tr read land in four different 64-byte bank groups), (kr >> 1) & 1 (W = 128).
template <bool T, int R>
struct Blk {
  static constexpr int BYTES = R * 128;
  static constexpr int NP = BYTES / 1024;                                         // DMA pieces per block
  static constexpr int W = T ? R * 2 : 128;                                       // bytes per LDS row
  static constexpr int RPP = 1024 / W > 0 ? 1024 / W : 1;                         // LDS rows per piece
  static constexpr int NPAR = T ? (RPP == 2 ? 2 : (RPP == 1 ? 4 : 1)) : 2;        // distinct per-lane source offsets (piece number mod NPAR)
  static_assert(!T || (W >= 128 && W <= 1024 && (W & (W - 1)) == 0), "transposed blocks: 64..512 columns, a power of two");
  // per-lane source byte offset relative to the piece's origin, for pieces with (piece % NPAR) == par
  __device__ static __forceinline__ uint32_t lane_off(int lane, int par, int64_t ld) {
    if (!T) {
      const int r = lane >> 3, s = lane & 7;
      return (uint32_t)(r * ld * 2) + (uint32_t)((s ^ (r >> 1) ^ (par << 2)) << 4);
    } else {
      const int o = lane * 16, krl = o / W, bir = o % W, c64 = bir >> 6, sub = (bir >> 4) & 3;
      const int x = W == 128 ? ((krl >> 1) & 1) : ((par * RPP + krl) & 3);
      return (uint32_t)(krl * ld * 2) + (uint32_t)(((c64 ^ x) << 6) + (sub << 4));
    }
  }
  // wave-uniform source byte offset of piece pc of the block whose first major element is mn0, at contraction offset k0
  __device__ static __forceinline__ uint32_t piece_org(int pc, int mn0, int k0, int64_t ld) {
    if (!T) return (uint32_t)((((int64_t)(mn0 + 8 * pc)) * ld + k0) * 2);
    return (uint32_t)((((int64_t)(k0 + pc * RPP)) * ld + mn0) * 2);
  }
  // lane part of the fragment read address.  K-contiguous: index = k-step; transposed: index = 32-column group q of the block.
  __device__ static __forceinline__ uint32_t frag_lane(int lane, int idx) {
    if (!T) {
      const int l31 = lane & 31, hi = lane >> 5;
      return (uint32_t)(l31 * 128 + ((((idx << 1) | hi) ^ ((l31 >> 1) & 7)) << 4));
    } else {
      const int t15 = lane & 15, g = lane >> 4;
      const int x = W == 128 ? ((t15 >> 3) & 1) : (t15 >> 2);
      return (uint32_t)((8 * (g >> 1) + (t15 >> 2)) * W + ((idx ^ x) << 6) + (g & 1) * 32 + (t15 & 3) * 8);
    }
  }
};

// fragment of k-step KS, 32-element group Q (compile-time part; the run-time part of the group is folded into v[])
template <bool T, int W, int KS, int Q>
__device__ __forceinline__ bf16x8 pp_frag(const uint32_t (&v)[4], uint32_t off) {
  if constexpr (!T) {
    uint4 r = lds_read_b128_o<Q * 4096>(v[KS] + off);
    return *reinterpret_cast<bf16x8*>(&r);
  } else {
    const uint32_t a = v[Q] + off;
    uint2 r0 = lds_read_tr16_o<(KS * 16) * W>(a);
    uint2 r1 = lds_read_tr16_o<(KS * 16 + 4) * W>(a);
    uint4 r = {r0.x, r0.y, r1.x, r1.y};
    return *reinterpret_cast<bf16x8*>(&r);
  }
}

// ---- contraction-balanced ("stream-K") grouped launches: one SEGMENT = K-tiles [kt_begin, kt_end) of one output tile.
// A workgroup that does not own the END of the tile's contraction leaves its accumulators (and bias-gradient partial sums) in its
// workspace slot and raises the slot's per-wave flags; the workgroup that owns the end (the FINISHER) waits for the n_in slots
// below its own (they are written first thing by lower-numbered workgroups, see pp_group_sk_kernel), adds them in a fixed order
// (deterministic) and runs the normal epilogue.  Slot layout (floats): [wave 8][(i * NI + j) * 4 + q][lane 64][4], then
// [wave 8][i][lane 64] bias partial sums.  Data crosses XCDs, whose L2s are not coherent with each other inside a kernel: it is
// written and read with system-scope (sc0 sc1) 16-byte accesses, 1 KiB contiguous per wave instruction, the flags with
// agent-scope atomics.
struct SkSeg {
  int kt_begin, kt_end;
  float* part_out;          // non-finisher: this workgroup's slot (else nullptr)
  uint32_t* flag_out;       // 8 flags (one per wave) of that slot
  const float* part_in;     // finisher: slot of the nearest contributor (workgroup w - 1); contributor c is at part_in - c * slot_floats
  uint32_t* flag_in;        //           its flags; contributor c at flag_in - c * 8
  int n_in;
  int slot_floats;
};
// Sixteen 16-byte pieces of a slot (four accumulator tiles' worth per lane) in ONE asm statement each way, so that the register cost
// of the exchange is fixed (64 temporaries on the load side, none on the store side) — left to the scheduler, the 128 loads of a
// 256 x 256 tile were hoisted over each other and the main loop spilled.  `base` is wave-uniform, voff = lane * 16.
__device__ __forceinline__ void sk_store_batch(const float* base, uint32_t voff, const f32x4 (&v)[16]) {
  const float *b0 = base, *b1 = base + 1024, *b2 = base + 2048, *b3 = base + 3072;
  asm volatile(
      "global_store_dwordx4 %16, %0, %17 sc0 sc1\n\tglobal_store_dwordx4 %16, %1, %17 offset:1024 sc0 sc1\n\t"
      "global_store_dwordx4 %16, %2, %17 offset:2048 sc0 sc1\n\tglobal_store_dwordx4 %16, %3, %17 offset:3072 sc0 sc1\n\t"
      "global_store_dwordx4 %16, %4, %18 sc0 sc1\n\tglobal_store_dwordx4 %16, %5, %18 offset:1024 sc0 sc1\n\t"
      "global_store_dwordx4 %16, %6, %18 offset:2048 sc0 sc1\n\tglobal_store_dwordx4 %16, %7, %18 offset:3072 sc0 sc1\n\t"
      "global_store_dwordx4 %16, %8, %19 sc0 sc1\n\tglobal_store_dwordx4 %16, %9, %19 offset:1024 sc0 sc1\n\t"
      "global_store_dwordx4 %16, %10, %19 offset:2048 sc0 sc1\n\tglobal_store_dwordx4 %16, %11, %19 offset:3072 sc0 sc1\n\t"
      "global_store_dwordx4 %16, %12, %20 sc0 sc1\n\tglobal_store_dwordx4 %16, %13, %20 offset:1024 sc0 sc1\n\t"
      "global_store_dwordx4 %16, %14, %20 offset:2048 sc0 sc1\n\tglobal_store_dwordx4 %16, %15, %20 offset:3072 sc0 sc1\n\t"
      "s_nop 1"
      :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]),
         "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(voff), "s"(b0), "s"(b1), "s"(b2), "s"(b3)
      : "memory");
}
__device__ __forceinline__ void sk_load_batch(const float* base, uint32_t voff, f32x4 (&v)[16]) {
  const float *b0 = base, *b1 = base + 1024, *b2 = base + 2048, *b3 = base + 3072;
  asm volatile(
      "global_load_dwordx4 %0, %16, %17 sc0 sc1\n\tglobal_load_dwordx4 %1, %16, %17 offset:1024 sc0 sc1\n\t"
      "global_load_dwordx4 %2, %16, %17 offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %3, %16, %17 offset:3072 sc0 sc1\n\t"
      "global_load_dwordx4 %4, %16, %18 sc0 sc1\n\tglobal_load_dwordx4 %5, %16, %18 offset:1024 sc0 sc1\n\t"
      "global_load_dwordx4 %6, %16, %18 offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %7, %16, %18 offset:3072 sc0 sc1\n\t"
      "global_load_dwordx4 %8, %16, %19 sc0 sc1\n\tglobal_load_dwordx4 %9, %16, %19 offset:1024 sc0 sc1\n\t"
      "global_load_dwordx4 %10, %16, %19 offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %11, %16, %19 offset:3072 sc0 sc1\n\t"
      "global_load_dwordx4 %12, %16, %20 sc0 sc1\n\tglobal_load_dwordx4 %13, %16, %20 offset:1024 sc0 sc1\n\t"
      "global_load_dwordx4 %14, %16, %20 offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %15, %16, %20 offset:3072 sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]),
        "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
      : "v"(voff), "s"(b0), "s"(b1), "s"(b2), "s"(b3)
      : "memory");
}

// SK: 0 = a whole tile (every other launch form); segments of a contraction-balanced group: 1 = publish, 2 = finish, 3 = whole tile
template <class CF, bool TA, bool TB, typename OutT, int EPI, bool SPLITK, int SK = 0>
__device__ __forceinline__ void pp_tile(const G2Args& p, int bid, int split, const SkSeg* sk = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MI = CF::MI, NI = CF::NI, NB = CF::NB, BM = CF::BM, BN = CF::BN, AH = CF::AH, BSZ = CF::BSZ;
  static_assert(SK == 0 || (!SPLITK && sizeof(OutT) == 4 && TA), "segments: float32 weight-gradient tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Blk<TA, 32 * MI> BA;      // one A half-block
  typedef Blk<TB, 128 * NI> BB;     // the B block
  static_assert(BA::BYTES == AH && BB::BYTES == BSZ, "block sizes");
  constexpr int PPW_A = BA::NP / 4, PPW_B = BB::NP / 4;     // pieces per wave: A half-block (group 1 issues two of them), B block (group 0)
  static_assert(PPW_A * 4 == BA::NP && PPW_B * 4 == BB::NP, "pieces divide over the four waves of a group");
  constexpr int WROWS = 32 * MI, WCOLS = 32 * NI;
  constexpr bool SWAP = !SPLITK && sizeof(OutT) == 2;
  constexpr uint32_t B_BASE = 4 * AH;

  int tid_ = threadIdx.x;
  if constexpr (SK != 0) asm volatile("" : "+v"(tid_));      // (called in a loop: keeps the lane-only address parts from being hoisted out of it and held in registers across the tiles)
  const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int hi = lane >> 5, l31 = lane & 31;

  const int gsz = p.group_m * p.tiles_n;
  const int grpi = bid / gsz, gi = bid - grpi * gsz;
  const int gm = min(p.tiles_m - grpi * p.group_m, p.group_m);
  const int tn = gi / gm, tm = grpi * p.group_m + (gi - tn * gm);
  const int m0 = tm * BM, n0 = tn * BN;

  int kt_begin = 0, kt_end = (p.Kc + BK - 1) / BK;
  if (SPLITK) {
    kt_begin = split * p.k_tiles_per_split;
    kt_end = min(kt_end, kt_begin + p.k_tiles_per_split);
    if (kt_begin >= kt_end) return;
  }
  if constexpr (SK != 0) { kt_begin = sk->kt_begin; kt_end = sk->kt_end; }
  const int nkt = kt_end - kt_begin;

  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)p.b_bytes, 0x00020000);

  // fragment read addresses (lane parts): A of this group's half-block, B of this wave's column patch
  uint32_t va[4], vb[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    va[x] = BA::frag_lane(lane, x) + (uint32_t)(grp * 2 * AH);
    if (!TB) vb[x] = BB::frag_lane(lane, x) + (uint32_t)(wn * WCOLS * 128) + B_BASE;
    else vb[x] = BB::frag_lane(lane, wn * NI + (x < NI ? x : 0)) + B_BASE;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 fa[4][MI], fb[4][NI];

  // bias gradient (column tile 0 only).  Whole tiles (SK == 0): the four waves of a group hold the same A fragments, so wave wn sums fragment row block
  // i == wn — a quarter of the work each, every output row still has ONE writer (deterministic).  Segments of a contraction-balanced launch keep the
  // one-wave form (their partial sums travel through the workspace in that layout).
  constexpr bool BS_SPLIT = (SK == 0);
  const bool do_colsum = TA && p.colsum != nullptr && tn == 0 && (BS_SPLIT ? wn < MI : wn == 0);
  float bsum[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) bsum[i] = 0.f;

  const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_void*)smem;
  const uint32_t ka = TA ? (uint32_t)(BK * p.lda * 2) : (uint32_t)(BK * 2);     // source advance per K-tile
  const uint32_t kb = TB ? (uint32_t)(BK * p.ldb * 2) : (uint32_t)(BK * 2);
  const int k0 = kt_begin * BK;

  // all fragments of the K-tile in A buffer ab_ (byte offset inside this group's pair of buffers) and B buffer bb_
#define PP_FRAGS_KS(KS, ao_, bo_)                                                                  \
  do {                                                                                             \
    fa[KS][0] = pp_frag<TA, BA::W, KS, 0>(va, ao_);                                                \
    if constexpr (MI > 1) fa[KS][1 < MI ? 1 : 0] = pp_frag<TA, BA::W, KS, 1>(va, ao_);             \
    if constexpr (MI > 2) fa[KS][2 < MI ? 2 : 0] = pp_frag<TA, BA::W, KS, 2>(va, ao_);             \
    if constexpr (MI > 3) fa[KS][3 < MI ? 3 : 0] = pp_frag<TA, BA::W, KS, 3>(va, ao_);             \
    fb[KS][0] = pp_frag<TB, BB::W, KS, 0>(vb, bo_);                                                \
    if constexpr (NI > 1) fb[KS][1 < NI ? 1 : 0] = pp_frag<TB, BB::W, KS, 1>(vb, bo_);             \
  } while (0)
  // Bias gradient of a weight-gradient tile (column tile 0 only): the sum of the A fragments over the contraction.  Rounds 1-5: ONE wave (wave column 0)
  // converted and added every fragment in FRONT of its MFMA phase — 128 cvt + add per K-tile, ~1000 cycles against the phase's 1024 — and the tiles that
  // carry it (one column tile in three for a 768-wide input) held their whole round back: grouped launches ran 882 TFLOP/s with it and 1039 without at
  // 3840 rows, 950 / 1236 at 20 480 (profiles/round6_wgrad_bias_ab.txt).  Round 6: four v_dot2c_f32_bf16 (pair . (1, 1) + acc) per fragment, issued
  // BEHIND the tile's MFMAs (while the matrix pipe drains), and the four waves of a group — they hold the same A fragments — take one 32-row block each
  // (BS_SPLIT): 16 instructions per K-tile and wave, one writer per output row as before.  With it 1032 against 1033 without at 3840 rows.
  // (Interleaving the sums with the MFMAs: a second copy of the loop made the 256 x 256 kernel spill 836 registers; one loop with the sums behind a
  //  wave-uniform branch per (k-step, row block) ran 694 TFLOP/s WITH OR WITHOUT a bias — the branches break the MFMA stream: profiles/round6_wgrad_bias_ab.txt.)
#ifndef GOAT_BSUM_DOT2
#define GOAT_BSUM_DOT2 1        // (0: the conversions + adds of rounds 1-5, for same-box A/B builds)
#endif
#if GOAT_BSUM_DOT2
#define PP_BSUM(frag_, acc_)                                                                       \
  do {                                                                                             \
    _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                             \
      const bf16x2 pr_ = {frag_[2 * e2], frag_[2 * e2 + 1]};                                       \
      acc_ = __builtin_amdgcn_fdot2_f32_bf16(pr_, bf16x2{(bf16_t)1.0f, (bf16_t)1.0f}, acc_, false); \
    }                                                                                              \
  } while (0)
#else
#define PP_BSUM(frag_, acc_)                                                                       \
  do { _Pragma("unroll") for (int e = 0; e < 8; ++e) acc_ += (float)frag_[e]; } while (0)
#endif
#define PP_MMA_ALL()                                                                               \
  do {                                                                                             \
    if (GOAT_G5_SETPRIO) __builtin_amdgcn_s_setprio(1);                                            \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                               \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                               \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                           \
          if (SWAP) mma32(acc[i][j], fb[ks][j], fa[ks][i]);                                        \
          else mma32(acc[i][j], fa[ks][i], fb[ks][j]);                                             \
        }                                                                                          \
    if (GOAT_G5_SETPRIO) __builtin_amdgcn_s_setprio(0);                                            \
    if (TA && do_colsum) {                                                                         \
      if constexpr (BS_SPLIT) {                                                                    \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                             \
          if (wn == i) { _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) PP_BSUM(fa[ks][i], bsum[i]); } \
      } else {                                                                                     \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                           \
          _Pragma("unroll") for (int i = 0; i < MI; ++i) PP_BSUM(fa[ks][i], bsum[i]);              \
      }                                                                                            \
    }                                                                                              \
  } while (0)

  static_assert(NI <= 2 && MI <= 4, "fragment macros are written for MI <= 4, NI <= 2");

  constexpr int DPL = CF::VAR & 3;            // LDS-DMA placement inside the MEM phase (see PCfg)
#if GOAT_G5_TIMING
  uint32_t tms[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) tms[i] = 0;
  const uint32_t tm_start = g5_now();
  uint32_t tm_c = tm_start;
#define PP_STAMP(i_) do { const uint32_t n_ = g5_now(); tms[i_] += n_ - tm_c; tm_c = n_; } while (0)
#else
#define PP_STAMP(i_) do { } while (0)
#endif
  // One slot pair of a group.  PP_ISSUE_RANGE(lo, hi) issues this wave's LDS-DMA pieces [lo, hi) of the period; LOADS of them.
#define PP_PERIOD(LOADS_, ao_, bo_, WAITVM_, LASTBAR_)                                             \
  do {                                                                                             \
    if (DPL == 0) PP_ISSUE_RANGE(0, LOADS_);                                                       \
    PP_STAMP(0);                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    PP_FRAGS_KS(0, ao_, bo_);                                                                      \
    if (DPL == 1) { __builtin_amdgcn_sched_barrier(0); PP_ISSUE_RANGE(0, LOADS_ / 4); __builtin_amdgcn_sched_barrier(0); } \
    PP_FRAGS_KS(1, ao_, bo_);                                                                      \
    if (DPL == 1) { __builtin_amdgcn_sched_barrier(0); PP_ISSUE_RANGE(LOADS_ / 4, LOADS_ / 2); __builtin_amdgcn_sched_barrier(0); } \
    PP_FRAGS_KS(2, ao_, bo_);                                                                      \
    if (DPL == 1) { __builtin_amdgcn_sched_barrier(0); PP_ISSUE_RANGE(LOADS_ / 2, 3 * LOADS_ / 4); __builtin_amdgcn_sched_barrier(0); } \
    PP_FRAGS_KS(3, ao_, bo_);                                                                      \
    if (DPL == 1) { __builtin_amdgcn_sched_barrier(0); PP_ISSUE_RANGE(3 * LOADS_ / 4, LOADS_); __builtin_amdgcn_sched_barrier(0); } \
    if (DPL == 2) { __builtin_amdgcn_sched_barrier(0); PP_ISSUE_RANGE(0, LOADS_); __builtin_amdgcn_sched_barrier(0); } \
    wait_lgkm<0>();                                                                                \
    PP_STAMP(1);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    PP_STAMP(2);                                                                                   \
    PP_MMA_ALL();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    PP_STAMP(3);                                                                                   \
    WAITVM_;                                                                                       \
    PP_STAMP(4);                                                                                   \
    if (LASTBAR_) __builtin_amdgcn_s_barrier();                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    PP_STAMP(5);                                                                                   \
  } while (0)

  if (grp == 0) {
    // ---------------------------------------------------------------- group 0: loads B, computes rows [0, 32*MI)
    uint32_t vo[BB::NPAR];
    const int par0 = (wn * PPW_B) % BB::NPAR;
#pragma unroll
    for (int q = 0; q < BB::NPAR; ++q) vo[q] = BB::lane_off(lane, (q + par0) % BB::NPAR, p.ldb);
    const uint32_t org0 = BB::piece_org(wn * PPW_B, n0, k0, p.ldb);
    const uint32_t pstep = BB::piece_org(1, 0, 0, p.ldb);      // source distance between consecutive pieces
    // pieces [lo_, hi_) of this wave's share of B[t_] -> B buffer buf_
#define PP_ISSUE_B(t_, buf_, lo_, hi_)                                                                         \
  do {                                                                                                         \
    char* dst_ = smem + B_BASE + (buf_) * BSZ + wn * PPW_B * 1024;                                             \
    const uint32_t so_ = org0 + (uint32_t)(t_) * kb;                                                           \
    _Pragma("unroll") for (int j = (lo_); j < (hi_); ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(        \
        rb, (lds_void*)(dst_ + j * 1024), 16, vo[j % BB::NPAR], so_ + (uint32_t)j * pstep, 0, 0);              \
  } while (0)
    PP_ISSUE_B(0, 0, 0, PPW_B);
    if (NB == 3 && nkt > 1) PP_ISSUE_B(1, 1, 0, PPW_B);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
#if GOAT_G5_TIMING
    tm_c = g5_now();
#endif
    int bbuf = 0;                                   // B buffer of tile t
    for (int t = 0; t < nkt; ++t) {
      const int nbuf = bbuf + 1 == NB ? 0 : bbuf + 1;
      const int n2 = nbuf + 1 == NB ? 0 : nbuf + 1;
      const uint32_t ao = smem_base + (uint32_t)((t & 1) * AH), bo = smem_base + (uint32_t)(bbuf * BSZ);
      // MEM phase in slot 2t (issues B[t+1], with three buffers B[t+2]), MFMA phase in slot 2t+1
#define PP_ISSUE_RANGE(lo_, hi_)                                                                   \
  do {                                                                                             \
    if (NB == 2) { if (t + 1 < nkt) PP_ISSUE_B(t + 1, nbuf, lo_, hi_); }                           \
    else { if (t + 2 < nkt) PP_ISSUE_B(t + 2, n2, lo_, hi_); }                                     \
  } while (0)
#define PP_WAITVM_G0                                                                               \
  do {                                                                                             \
    if (NB == 2 || t + 2 >= nkt) wait_vm<0>();                                                     \
    else wait_vm<PPW_B>();       /* B[t+1] has landed, B[t+2] may stay in flight */                \
  } while (0)
      PP_PERIOD(PPW_B, ao, bo, PP_WAITVM_G0, true);
#undef PP_ISSUE_RANGE
#undef PP_WAITVM_G0
      bbuf = nbuf;
    }
#undef PP_ISSUE_B
  } else {
    // ---------------------------------------------------------------- group 1: loads both A half-blocks, computes rows [32*MI, 64*MI)
    uint32_t vo[BA::NPAR];
    const int par0 = (wn * PPW_A) % BA::NPAR;
#pragma unroll
    for (int q = 0; q < BA::NPAR; ++q) vo[q] = BA::lane_off(lane, (q + par0) % BA::NPAR, p.lda);
    const uint32_t org0 = BA::piece_org(wn * PPW_A, m0, k0, p.lda);
    const uint32_t org1 = BA::piece_org(wn * PPW_A, m0 + WROWS, k0, p.lda);
    const uint32_t pstep = BA::piece_org(1, 0, 0, p.lda);
    // pieces [lo_, hi_) of this wave's share of half h_ (0/1) of K-tile t_ -> buffer buf_ of that half
#define PP_ISSUE_A(h_, t_, buf_, lo_, hi_)                                                                     \
  do {                                                                                                         \
    char* dst_ = smem + ((h_) * 2 + (buf_)) * AH + wn * PPW_A * 1024;                                          \
    const uint32_t so_ = ((h_) ? org1 : org0) + (uint32_t)(t_) * ka;                                           \
    _Pragma("unroll") for (int j = (lo_); j < (hi_); ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(        \
        ra, (lds_void*)(dst_ + j * 1024), 16, vo[j % BA::NPAR], so_ + (uint32_t)j * pstep, 0, 0);              \
  } while (0)
    PP_ISSUE_A(0, 0, 0, 0, PPW_A);
    PP_ISSUE_A(1, 0, 0, 0, PPW_A);
    if (nkt > 1) PP_ISSUE_A(0, 1, 1, 0, PPW_A);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();                   // slot 0: group 0 reads tile 0, nothing to compute here yet
#if GOAT_G5_TIMING
    tm_c = g5_now();
#endif
    int bbuf = 0;
    for (int t = 0; t < nkt; ++t) {
      const uint32_t ao = smem_base + (uint32_t)((t & 1) * AH), bo = smem_base + (uint32_t)(bbuf * BSZ);
      // MEM phase in slot 2t+1: pieces [0, PPW_A) = A_1[t+1], [PPW_A, 2 PPW_A) = A_0[t+2]; MFMA phase in slot 2t+2
#define PP_ISSUE_RANGE(lo_, hi_)                                                                   \
  do {                                                                                             \
    if ((lo_) < PPW_A && t + 1 < nkt) PP_ISSUE_A(1, t + 1, (t + 1) & 1, (lo_), ((hi_) < PPW_A ? (hi_) : PPW_A)); \
    if ((hi_) > PPW_A && t + 2 < nkt) PP_ISSUE_A(0, t + 2, t & 1, ((lo_) > PPW_A ? (lo_) - PPW_A : 0), (hi_) - PPW_A); \
  } while (0)
      PP_PERIOD(2 * PPW_A, ao, bo, wait_vm<0>(), (t + 1 < nkt));   // (group 0 has left its loop after its last MFMA phase)
#undef PP_ISSUE_RANGE
      bbuf = bbuf + 1 == NB ? 0 : bbuf + 1;
    }
#undef PP_ISSUE_A
  }
#if GOAT_G5_TIMING
  if (p.aux != nullptr && lane == 0 && EPI == GOAT_EPI_NONE) {
    uint32_t* o = reinterpret_cast<uint32_t*>(p.aux) + ((size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = tms[i];
    o[6] = g5_now() - tm_start;
    o[7] = (uint32_t)nkt;
  }
#endif
#undef PP_PERIOD
#undef PP_STAMP
#undef PP_FRAGS_KS
#undef PP_MMA_ALL
#undef PP_BSUM

  // From here on the LDS ring is free: the last fragment reads of both groups completed before the barrier that ended the last
  // MEM phase, and no LDS-DMA is in flight.  Group 0 starts its epilogue while group 1 is in its last MFMA phase.
  const int wrow0 = grp * WROWS, wcol0 = wn * WCOLS;
  if constexpr (SK == 1 || SK == 2) {
    constexpr int WAVE_FLOATS = MI * NI * 16 * 64, BS0 = 8 * WAVE_FLOATS;
    static_assert((MI * NI) % 4 == 0, "the exchange moves four accumulator tiles per batch");
    const uint32_t voff = (uint32_t)lane * 16u;
    if constexpr (SK == 1) {
      const float* o = sk->part_out + wave * WAVE_FLOATS;
#pragma unroll
      for (int b = 0; b < MI * NI / 4; ++b) {
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int idx = b * 4 + k;
            const f32x16& t = acc[idx / NI][idx % NI];
            v[k * 4 + q] = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
          }
        sk_store_batch(o + b * 4096, voff, v);
      }
      if (do_colsum) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          __hip_atomic_store(sk->part_out + BS0 + (wave * MI + i) * 64 + lane, bsum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(sk->flag_out + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    } else {
    for (int c = 0; c < sk->n_in; ++c) {
      uint32_t* fl = sk->flag_in - c * 8 + wave;
      while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(4);
      asm volatile("" ::: "memory");
      const float* o = sk->part_in - (int64_t)c * sk->slot_floats + wave * WAVE_FLOATS;
      // (plain loads: the publisher wrote through to memory before raising the flag, and this XCD's L2 cannot hold an older copy of the
      //  slot — it is only ever read here, after the flag, and L2s are invalidated between kernels)
#pragma unroll
      for (int idx = 0; idx < MI * NI; ++idx)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(o + (idx * 4 + q) * 256 + lane * 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[idx / NI][idx % NI][4 * q + e] += v[e];
        }
      if (do_colsum) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          bsum[i] += __hip_atomic_load(sk->part_in - (int64_t)c * sk->slot_floats + BS0 + (wave * MI + i) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(fl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
    }
  }
  if (TA && do_colsum) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (BS_SPLIT && i != wn) continue;          // (this wave summed row block wn only)
      float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      const int row = m0 + wrow0 + i * 32 + l31;
      if (hi == 0 && row < p.M) atomicAdd(p.colsum + row, v);
    }
  }
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
  if (sizeof(OutT) == 4) {
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
            else __builtin_nontemporal_store(u, dst);
          }
        }
      }
    return;
  }
  // bf16 output: as gemm2_tile.hpp — swapped MFMA operand roles (lane = row of C, 4 consecutive columns per register group),
  // staged one 32-row block row at a time through the wave's own slice of the free LDS, written out as 16-byte row pieces.
  typedef bf16_t T;
  constexpr int EPC = 8;
  constexpr int RBY = WCOLS * 2 + 16;
  constexpr int WSLICE = 32 * RBY;
  static_assert(8 * WSLICE <= CF::SMEM, "per-wave epilogue slices must fit the LDS ring");
  constexpr int CPR = WCOLS / EPC;
  constexpr int CHUNKS = 32 * CPR / 64;
  static_assert(32 * CPR % 64 == 0, "a block row is a whole number of 16-byte chunks per lane");
  constexpr bool DACT = (EPI == GOAT_EPI_MUL_DGELU || EPI == GOAT_EPI_MUL_DRELU);
  constexpr bool ACT = (EPI == GOAT_EPI_GELU || EPI == GOAT_EPI_RELU);
  T* aux = reinterpret_cast<T*>(p.aux);
  T* C = reinterpret_cast<T*>(p.C);
  const bool c_vec = (p.ldc % EPC) == 0 && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  const bool aux_vec = aux != nullptr && (p.ldaux % EPC) == 0 && ((reinterpret_cast<uintptr_t>(aux) & 15) == 0);
  char* wsp = smem + wave * WSLICE;
  const int col_w = n0 + wcol0;
  f32x4 bv[DACT ? 1 : NI][4];          // (the activation-derivative epilogues take no bias: C = (A·B) * act'(aux))
  if (!DACT) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = col_w + j * 32 + 4 * hi + 8 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][q][e] = (p.bias != nullptr && col + e < p.N) ? p.bias[col + e] : 0.f;
      }
  }
  // activation-derivative epilogues (FFN dgrad): the saved pre-activation of the wave patch is fetched up to three block rows AHEAD, into
  // the registers the fragments occupied until the last MFMA phase (CHUNKS 16-byte pieces per lane and block row): one exposed HBM
  // round trip per tile instead of one per block row (the first version of this epilogue: 20480 x 3072 x 768 at 548 TFLOP/s against
  // 771 with the GELU epilogue).  Four rows at once would not fit beside the 128 accumulator registers of the 256 x 256 tile.
  constexpr int AD = MI < 3 ? MI : 3;
  uint4 auxv[DACT ? AD : 1][CHUNKS];
  if (DACT) {
#pragma unroll
    for (int i = 0; i < AD; ++i) load_aux_rows<CHUNKS, CPR>(p, m0 + wrow0 + i * 32, col_w, lane, auxv[i]);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row_w = m0 + wrow0 + i * 32;
    if (DACT) {
#pragma unroll
      for (int c = 0; c < CHUNKS; ++c) {
        const int idx = c * 64 + lane, r = idx / CPR, cc = idx % CPR;
        *reinterpret_cast<uint4*>(wsp + r * RBY + cc * 16) = auxv[DACT ? i % AD : 0][c];
      }
      if (i + AD < MI) load_aux_rows<CHUNKS, CPR>(p, row_w + AD * 32, col_w, lane, auxv[DACT ? i % AD : 0]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        char* slot = wsp + l31 * RBY + (j * 32 + 4 * hi + 8 * q) * 2;
        float u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = DACT ? acc[i][j][4 * q + e] : acc[i][j][4 * q + e] + bv[DACT ? 0 : j][q][e];
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <class CF, bool TA, bool TB, typename OutT, int EPI, bool SPLITK>
__global__ __launch_bounds__(512) void pp_kernel(G2Args p) {
  pp_tile<CF, TA, TB, OutT, EPI, SPLITK>(p, xcd_chunk_position(blockIdx.x, gridDim.x), blockIdx.y);
}

template <class CF>
__global__ __launch_bounds__(512) void pp_group_kernel(GroupArgs g) {
  const int pos = xcd_chunk_position(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i)
    if (i < g.n && pos >= g.tile_start[i]) pi = i;
  const G2Args p = g.prob[pi].template expand<CF::BM, CF::BN>();
  pp_tile<CF, true, true, float, GOAT_EPI_NONE, false>(p, pos - g.tile_start[pi], 0);
}

// Contraction-balanced grouped weight gradients.  pp_group_kernel gives every output tile a workgroup: 432 tiles of 256 x 256 on 256
// CUs are two rounds for 1.69 rounds of work, and a group that mixes contraction lengths (8640-row panorama problems beside 3840-row
// text problems) ends with most CUs idle.  Here the grid is ONE workgroup per CU and the group's K-tile iterations — tiles in the
// group's tile order, K-tiles in order inside a tile — are cut into gridDim.x equal contiguous ranges.  A range is at most: the tail
// of a tile another workgroup started, whole tiles, the head of a tile the next workgroup finishes.  The HEAD segment (the only one
// whose result has to travel) runs FIRST and goes to the workgroup's slot; the segment that begins mid-tile runs LAST and waits for
// the slots of the workgroups below it: a workgroup only ever waits for lower positions, which published before doing anything
// else, so the wait is short and cannot deadlock (positions of one XCD chunk are dispatched in order; across a chunk boundary the
// producer is one of the at most seven workgroups everyone else does not depend on).  Sums are formed in a fixed order: results are
// deterministic (they differ from pp_group_kernel's by float32 summation order only).
struct SkGroupArgs {
  GroupArgs g;
  int iter_start[GROUP_MAX + 1];      // first K-tile iteration of problem i in the group's iteration order
  float* ws;                          // gridDim.x slots of slot_floats floats
  uint32_t* flags;                    // gridDim.x x 8, zero between launches (finishers clear what they consume)
  int slot_floats;
};
static_assert(sizeof(SkGroupArgs) <= 4000, "SkGroupArgs must fit the kernel-argument segment");

template <class CF>
__global__ __launch_bounds__(512) void pp_group_sk_kernel(SkGroupArgs s) {
  const int G = gridDim.x;
  const int w = xcd_chunk_position(blockIdx.x, G);
  const int W = s.iter_start[s.g.n];
  const int b0 = (int)((int64_t)w * W / G), b1 = (int)((int64_t)(w + 1) * W / G);
  if (b0 >= b1) return;
  // tile bounds [T0, T0 + kt) of the tile that holds iteration x, and its problem (unrolled search: only run before the loop, so that
  // the 25 iteration offsets are not kept in scalar registers across the tiles)
  auto locate = [&](int x, int& pi, int& T0, int& kt) {
    pi = 0;
#pragma unroll
    for (int i = 1; i < GROUP_MAX; ++i)
      if (i < s.g.n && x >= s.iter_start[i]) pi = i;
    kt = (s.g.prob[pi].Kc + BK - 1) / BK;
    T0 = s.iter_start[pi] + (x - s.iter_start[pi]) / kt * kt;
  };
  int piF, T0F, ktF, piL, T0L, ktL;
  locate(b0, piF, T0F, ktF);
  locate(b1 - 1, piL, T0L, ktL);
  const int first_end = min(b1, T0F + ktF);                  // first segment [b0, first_end)
  const bool head = (T0L + ktL > b1) && T0L > b0;            // a separate last segment [T0L, b1) that does not finish its tile
  const bool lone = first_end < T0F + ktF;                   // the whole range lies inside one tile and does not finish it
  const int mid_end = head ? T0L : b1;
  SkSeg sg;
  sg.slot_floats = s.slot_floats;
  sg.part_out = s.ws + (int64_t)w * s.slot_floats;
  sg.flag_out = s.flags + w * 8;
  sg.part_in = s.ws + (int64_t)(w - 1) * s.slot_floats;
  sg.flag_in = s.flags + (w - 1) * 8;
  sg.n_in = 0;
  // 1. the segment whose result travels (three inlined copies of the tile — publish / whole / finish — instead of one with both
  //    exchange paths: with both, the 256 x 256 tile spilled inside its main loop)
  if (head || lone) {
    const int pi = __builtin_amdgcn_readfirstlane(head ? piL : piF), T0 = head ? T0L : T0F, kt = head ? ktL : ktF;
    sg.kt_begin = (head ? T0L : b0) - T0;
    sg.kt_end = b1 - T0;
    const G2Args p = s.g.prob[pi].template expand<CF::BM, CF::BN>();
    pp_tile<CF, true, true, float, GOAT_EPI_NONE, false, 1>(p, (T0 - s.iter_start[pi]) / kt, 0, &sg);
    if (lone) return;
  }
  // 2. whole tiles [first_end, mid_end)
  int xm = first_end, pim = piF;
  if (xm >= s.iter_start[pim + 1]) ++pim;
  while (xm < mid_end) {
    pim = __builtin_amdgcn_readfirstlane(pim);
    const G2Args p = s.g.prob[pim].template expand<CF::BM, CF::BN>();
    const int kt = p.k_tiles_per_split;
    sg.kt_begin = 0;
    sg.kt_end = kt;
    pp_tile<CF, true, true, float, GOAT_EPI_NONE, false, 3>(p, (xm - s.iter_start[pim]) / kt, 0, &sg);
    xm += kt;
    if (xm >= s.iter_start[pim + 1]) ++pim;
  }
  // 3. the first segment [b0, first_end): finishes its tile; contributors are the workgroups below whose range reaches into the tile
  {
    int n = 0;
    if (b0 > T0F)
      while (w - 1 - n >= 0 && (int)((int64_t)(w - n) * W / G) > T0F) ++n;
    sg.n_in = n;
    sg.kt_begin = b0 - T0F;
    sg.kt_end = first_end - T0F;
    const int pi = __builtin_amdgcn_readfirstlane(piF);
    const G2Args p = s.g.prob[pi].template expand<CF::BM, CF::BN>();
    pp_tile<CF, true, true, float, GOAT_EPI_NONE, false, 2>(p, (T0F - s.iter_start[pi]) / ktF, 0, &sg);
  }
}

template <class CF, bool TA, bool TB, typename OutT, int EPI, bool SPLITK>
int pp_launch(hipStream_t st, const G2Args& a, int split) {
  static_assert(CF::SMEM <= 160 * 1024, "LDS exceeds the CU's 160 KiB");
  auto kern = pp_kernel<CF, TA, TB, OutT, EPI, SPLITK>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(a.tiles_m * a.tiles_n, SPLITK ? split : 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), CF::SMEM, st, a);
  GOAT_LAUNCH_CHECK();
  return 0;
}

template <class CF, bool TA, bool TB>
int pp_dispatch2(hipStream_t st, const G2Args& a, int dtype_out, int epi, int split) {
  if (split > 1) return pp_launch<CF, TA, TB, float, GOAT_EPI_NONE, true>(st, a, split);
  if (dtype_out == GOAT_F32) {
    if (epi != GOAT_EPI_NONE && epi != GOAT_EPI_ACCUM) return GOAT_E_ARG;
    return pp_launch<CF, TA, TB, float, GOAT_EPI_NONE, false>(st, a, 1);
  }
  if constexpr (TA) {      // weight-gradient layout: bf16 results only without an epilogue
    if (epi != GOAT_EPI_NONE) return GOAT_E_ARG;
    return pp_launch<CF, TA, TB, bf16_t, GOAT_EPI_NONE, false>(st, a, 1);
  } else {
    switch (epi) {
      case GOAT_EPI_NONE: return pp_launch<CF, TA, TB, bf16_t, GOAT_EPI_NONE, false>(st, a, 1);
      case GOAT_EPI_GELU: return pp_launch<CF, TA, TB, bf16_t, GOAT_EPI_GELU, false>(st, a, 1);
      case GOAT_EPI_RELU: return pp_launch<CF, TA, TB, bf16_t, GOAT_EPI_RELU, false>(st, a, 1);
      case GOAT_EPI_MUL_DGELU: return pp_launch<CF, TA, TB, bf16_t, GOAT_EPI_MUL_DGELU, false>(st, a, 1);
      case GOAT_EPI_MUL_DRELU: return pp_launch<CF, TA, TB, bf16_t, GOAT_EPI_MUL_DRELU, false>(st, a, 1);
    }
    return GOAT_E_ARG;
  }
}

template <class CF>
int pp_dispatch_layout(hipStream_t st, const G2Args& a, int trans_a, int trans_b, int dtype_out, int epi, int split) {
  if (!trans_a && !trans_b) return pp_dispatch2<CF, false, false>(st, a, dtype_out, epi, split);
  if (!trans_a && trans_b) return pp_dispatch2<CF, false, true>(st, a, dtype_out, epi, split);
  if constexpr ((CF::MI & (CF::MI - 1)) == 0) {      // a transposed A half-block needs a power-of-two width
    if (trans_a && trans_b) return pp_dispatch2<CF, true, true>(st, a, dtype_out, epi, split);
  }
  return GOAT_E_ARG;
}

template <class CF>
int pp_launch_group(hipStream_t st, const GroupArgs& g) {
  auto kern = pp_group_kernel<CF>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.tile_start[g.n]), dim3(512), CF::SMEM, st, g);
  GOAT_LAUNCH_CHECK();
  return 0;
}


// bytes of workspace a contraction-balanced group launch needs for tile CF on this device (slots + flags)
template <class CF>
constexpr int pp_sk_slot_floats() { return 8 * CF::MI * CF::NI * 16 * 64 + 8 * CF::MI * 64; }
inline int pp_cu_count();
template <class CF>
int pp_launch_group_sk(hipStream_t st, const GroupArgs& g, void* ws, int64_t ws_bytes) {
  auto kern = pp_group_sk_kernel<CF>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  static_assert(CF::SMEM > 80 * 1024, "one workgroup per CU (the wait relies on every workgroup being resident or ahead in dispatch order)");
  const int G = pp_cu_count();
  SkGroupArgs s;
  s.g = g;
  int it = 0;
  for (int i = 0; i < g.n; ++i) {
    s.iter_start[i] = it;
    it += (g.tile_start[i + 1] - g.tile_start[i]) * ((g.prob[i].Kc + BK - 1) / BK);
  }
  for (int i = g.n; i <= GROUP_MAX; ++i) s.iter_start[i] = it;
  if (it < G) return GOAT_E_SHAPE;                                     // fewer iterations than workgroups: nothing to balance
  s.slot_floats = pp_sk_slot_floats<CF>();
  const int64_t flag_off = ((int64_t)G * s.slot_floats * 4 + 255) / 256 * 256;
  if (ws == nullptr || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < flag_off + (int64_t)G * 32) return GOAT_E_ARG;
  s.ws = reinterpret_cast<float*>(ws);
  s.flags = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ws) + flag_off);
  hipLaunchKernelGGL(kern, dim3(G), dim3(512), CF::SMEM, st, s);
  GOAT_LAUNCH_CHECK();
  return 0;
}
template <class CF>
int64_t pp_sk_ws_bytes() {
  const int G = pp_cu_count();
  return ((int64_t)G * pp_sk_slot_floats<CF>() * 4 + 255) / 256 * 256 + (int64_t)G * 32;
}


// ======================================================================================== persistent form (round 5)
// pp_kernel gives every tile a workgroup of its own: a tile's first K-tile is requested when its workgroup starts (cold: ~1-2 us before
// the first MFMA) and its epilogue (bias / activation, LDS transposition, 16-byte stores: a third of a K = 768 launch together with
// the prologue) runs with the matrix pipe idle.  When a problem has more tiles than the chip has CUs (M = 20 480 at
// per-rank batch 256: 1 284 tiles of 192 x 256) the grid here is ONE workgroup per CU and every workgroup walks its share of its XCD's
// chunk of the tile order; the first K-tile of the NEXT tile (B[0] by group 0; A_0[0], A_1[0], A_0[1] by group 1 — what the prologue
// issues) is requested BEFORE the epilogue of the current one, so the requests cross the fabric while the results are converted and
// stored.  The epilogue's per-wave staging slices move out of the way of those DMA targets: into B buffer 1 (the first BSZ / WSLICE
// waves) and A_1 buffer 1 (the rest), which receive nothing before the barrier that opens the next tile's main loop (every wave
// passes it after its own epilogue).  bf16 results, unsplit, K-contiguous A (TA = false); same tile order inside an XCD chunk, same
// arithmetic per tile: results are bit-identical to pp_kernel's.
template <class CF, bool TB, int EPI>
__device__ __forceinline__ void pp_tiles_persist(const G2Args& p, int pos0, int pos_step, int pos_end) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool TA = false;
  constexpr int MI = CF::MI, NI = CF::NI, NB = CF::NB, BM = CF::BM, BN = CF::BN, AH = CF::AH, BSZ = CF::BSZ;
  static_assert(NB == 2, "the persistent form is written for two B buffers");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Blk<TA, 32 * MI> BA;
  typedef Blk<TB, 128 * NI> BB;
  constexpr int PPW_A = BA::NP / 4, PPW_B = BB::NP / 4;
  constexpr int WROWS = 32 * MI, WCOLS = 32 * NI;
  constexpr bool SWAP = true;
  constexpr uint32_t B_BASE = 4 * AH;
  if (pos0 >= pos_end) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int hi = lane >> 5, l31 = lane & 31;
  const int nkt = (p.Kc + BK - 1) / BK;

  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)p.b_bytes, 0x00020000);

  uint32_t va[4], vb[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    va[x] = BA::frag_lane(lane, x) + (uint32_t)(grp * 2 * AH);
    if (!TB) vb[x] = BB::frag_lane(lane, x) + (uint32_t)(wn * WCOLS * 128) + B_BASE;
    else vb[x] = BB::frag_lane(lane, wn * NI + (x < NI ? x : 0)) + B_BASE;
  }
  f32x16 acc[MI][NI];
  bf16x8 fa[4][MI], fb[4][NI];
  constexpr bool do_colsum = false;
  float bsum[MI];
  (void)bsum;

  const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_void*)smem;
  const uint32_t ka = (uint32_t)(BK * 2);
  const uint32_t kb = TB ? (uint32_t)(BK * p.ldb * 2) : (uint32_t)(BK * 2);

#define PP_FRAGS_KS(KS, ao_, bo_)                                                                  \
  do {                                                                                             \
    fa[KS][0] = pp_frag<TA, BA::W, KS, 0>(va, ao_);                                                \
    if constexpr (MI > 1) fa[KS][1 < MI ? 1 : 0] = pp_frag<TA, BA::W, KS, 1>(va, ao_);             \
    if constexpr (MI > 2) fa[KS][2 < MI ? 2 : 0] = pp_frag<TA, BA::W, KS, 2>(va, ao_);             \
    if constexpr (MI > 3) fa[KS][3 < MI ? 3 : 0] = pp_frag<TA, BA::W, KS, 3>(va, ao_);             \
    fb[KS][0] = pp_frag<TB, BB::W, KS, 0>(vb, bo_);                                                \
    if constexpr (NI > 1) fb[KS][1 < NI ? 1 : 0] = pp_frag<TB, BB::W, KS, 1>(vb, bo_);             \
  } while (0)
#define PP_MMA_ALL()                                                                               \
  do {                                                                                             \
    if (GOAT_G5_SETPRIO) __builtin_amdgcn_s_setprio(1);                                            \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                               \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                               \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) mma32(acc[i][j], fb[ks][j], fa[ks][i]);     \
    if (GOAT_G5_SETPRIO) __builtin_amdgcn_s_setprio(0);                                            \
  } while (0)
  static_assert(NI <= 2 && MI <= 4, "fragment macros are written for MI <= 4, NI <= 2");
  static_assert((CF::VAR & 3) == 1, "the persistent form uses DMA placement 1");
#define PP_PERIOD(LOADS_, ao_, bo_, WAITVM_, LASTBAR_)                                             \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    PP_FRAGS_KS(0, ao_, bo_);                                                                      \
    { __builtin_amdgcn_sched_barrier(0); PP_ISSUE_RANGE(0, LOADS_ / 4); __builtin_amdgcn_sched_barrier(0); } \
    PP_FRAGS_KS(1, ao_, bo_);                                                                      \
    { __builtin_amdgcn_sched_barrier(0); PP_ISSUE_RANGE(LOADS_ / 4, LOADS_ / 2); __builtin_amdgcn_sched_barrier(0); } \
    PP_FRAGS_KS(2, ao_, bo_);                                                                      \
    { __builtin_amdgcn_sched_barrier(0); PP_ISSUE_RANGE(LOADS_ / 2, 3 * LOADS_ / 4); __builtin_amdgcn_sched_barrier(0); } \
    PP_FRAGS_KS(3, ao_, bo_);                                                                      \
    { __builtin_amdgcn_sched_barrier(0); PP_ISSUE_RANGE(3 * LOADS_ / 4, LOADS_); __builtin_amdgcn_sched_barrier(0); } \
    wait_lgkm<0>();                                                                                \
    __builtin_amdgcn_s_barrier();                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    PP_MMA_ALL();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    WAITVM_;                                                                                       \
    if (LASTBAR_) __builtin_amdgcn_s_barrier();                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)

  // tile position -> (first row, first column), pp_tile's order
  const int gsz = p.group_m * p.tiles_n;
#define PP_COORDS(pos_, m0_, n0_)                                                                  \
  do {                                                                                             \
    const int grpi_ = (pos_) / gsz, gi_ = (pos_) - grpi_ * gsz;                                    \
    const int gm_ = min(p.tiles_m - grpi_ * p.group_m, p.group_m);                                 \
    const int tn_ = gi_ / gm_, tm_ = grpi_ * p.group_m + (gi_ - tn_ * gm_);                        \
    m0_ = tm_ * BM; n0_ = tn_ * BN;                                                                \
  } while (0)

  // lane parts of the DMA source addresses (tile-independent)
  uint32_t vo[(BB::NPAR > BA::NPAR ? BB::NPAR : BA::NPAR)];
  uint32_t pstep;
  if (grp == 0) {
    const int par0 = (wn * PPW_B) % BB::NPAR;
#pragma unroll
    for (int q = 0; q < BB::NPAR; ++q) vo[q] = BB::lane_off(lane, (q + par0) % BB::NPAR, p.ldb);
    pstep = BB::piece_org(1, 0, 0, p.ldb);
  } else {
    const int par0 = (wn * PPW_A) % BA::NPAR;
#pragma unroll
    for (int q = 0; q < BA::NPAR; ++q) vo[q] = BA::lane_off(lane, (q + par0) % BA::NPAR, p.lda);
    pstep = BA::piece_org(1, 0, 0, p.lda);
  }
  // pieces [lo_, hi_) of this wave's share of B[t_] of the tile whose B origin is org_ -> B buffer buf_
#define PP_ISSUE_B(org_, t_, buf_, lo_, hi_)                                                                   \
  do {                                                                                                         \
    char* dst_ = smem + B_BASE + (buf_) * BSZ + wn * PPW_B * 1024;                                             \
    const uint32_t so_ = (org_) + (uint32_t)(t_) * kb;                                                         \
    _Pragma("unroll") for (int j = (lo_); j < (hi_); ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(        \
        rb, (lds_void*)(dst_ + j * 1024), 16, vo[j % BB::NPAR], so_ + (uint32_t)j * pstep, 0, 0);              \
  } while (0)
  // pieces [lo_, hi_) of this wave's share of the A half-block with origin org_ of K-tile t_ -> buffer buf_ of half h_
#define PP_ISSUE_A(org_, h_, t_, buf_, lo_, hi_)                                                               \
  do {                                                                                                         \
    char* dst_ = smem + ((h_) * 2 + (buf_)) * AH + wn * PPW_A * 1024;                                          \
    const uint32_t so_ = (org_) + (uint32_t)(t_) * ka;                                                         \
    _Pragma("unroll") for (int j = (lo_); j < (hi_); ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(        \
        ra, (lds_void*)(dst_ + j * 1024), 16, vo[j % BA::NPAR], so_ + (uint32_t)j * pstep, 0, 0);              \
  } while (0)

  int m0, n0;
  PP_COORDS(pos0, m0, n0);
  // the first tile's first K-tile (pp_tile's prologue)
  if (grp == 0) {
    const uint32_t org0 = BB::piece_org(wn * PPW_B, n0, 0, p.ldb);
    PP_ISSUE_B(org0, 0, 0, 0, PPW_B);
  } else {
    const uint32_t org0 = BA::piece_org(wn * PPW_A, m0, 0, p.lda), org1 = BA::piece_org(wn * PPW_A, m0 + WROWS, 0, p.lda);
    PP_ISSUE_A(org0, 0, 0, 0, 0, PPW_A);
    PP_ISSUE_A(org1, 1, 0, 0, 0, PPW_A);
    if (nkt > 1) PP_ISSUE_A(org0, 0, 1, 1, 0, PPW_A);
  }

  // epilogue staging (see the header comment): wave-private slices outside the next tile's first DMA targets
  typedef bf16_t T;
  constexpr int EPC = 8;
  constexpr int RBY = WCOLS * 2 + 16;
  constexpr int WSLICE = 32 * RBY;
  constexpr int NFIT = BSZ / WSLICE < 8 ? BSZ / WSLICE : 8;
  static_assert((8 - NFIT) * WSLICE <= AH, "the staging slices that do not fit B buffer 1 must fit A_1 buffer 1");
  char* wsp = wave < NFIT ? smem + B_BASE + BSZ + wave * WSLICE : smem + 3 * AH + (wave - NFIT) * WSLICE;
  constexpr int CPR = WCOLS / EPC;
  constexpr int CHUNKS = 32 * CPR / 64;
  constexpr bool DACT = (EPI == GOAT_EPI_MUL_DGELU || EPI == GOAT_EPI_MUL_DRELU);
  constexpr bool ACT = (EPI == GOAT_EPI_GELU || EPI == GOAT_EPI_RELU);
  T* aux = reinterpret_cast<T*>(p.aux);
  T* C = reinterpret_cast<T*>(p.C);
  const bool c_vec = (p.ldc % EPC) == 0 && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  const bool aux_vec = aux != nullptr && (p.ldaux % EPC) == 0 && ((reinterpret_cast<uintptr_t>(aux) & 15) == 0);
  const int wrow0 = grp * WROWS, wcol0 = wn * WCOLS;

  for (int pos = pos0; pos < pos_end; pos += pos_step) {
    const int npos = pos + pos_step;
    const bool has_next = npos < pos_end;
    int nm0 = 0, nn0 = 0;
    if (has_next) PP_COORDS(npos, nm0, nn0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (grp == 0) {
      const uint32_t org0 = BB::piece_org(wn * PPW_B, n0, 0, p.ldb);
      wait_vm<0>();                                 // this tile's B[0] (and the previous tile's result stores)
      __builtin_amdgcn_s_barrier();
      int bbuf = 0;
      for (int t = 0; t < nkt; ++t) {
        const int nbuf = bbuf ^ 1;
        const uint32_t ao = smem_base + (uint32_t)((t & 1) * AH), bo = smem_base + (uint32_t)(bbuf * BSZ);
#define PP_ISSUE_RANGE(lo_, hi_) do { if (t + 1 < nkt) PP_ISSUE_B(org0, t + 1, nbuf, lo_, hi_); } while (0)
        PP_PERIOD(PPW_B, ao, bo, wait_vm<0>(), true);
#undef PP_ISSUE_RANGE
        bbuf = nbuf;
      }
      if (has_next) {
        const uint32_t orgn = BB::piece_org(wn * PPW_B, nn0, 0, p.ldb);
        PP_ISSUE_B(orgn, 0, 0, 0, PPW_B);
      }
    } else {
      const uint32_t org0 = BA::piece_org(wn * PPW_A, m0, 0, p.lda), org1 = BA::piece_org(wn * PPW_A, m0 + WROWS, 0, p.lda);
      wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();                 // slot 0: group 0 reads tile 0, nothing to compute here yet
      int bbuf = 0;
      for (int t = 0; t < nkt; ++t) {
        const uint32_t ao = smem_base + (uint32_t)((t & 1) * AH), bo = smem_base + (uint32_t)(bbuf * BSZ);
#define PP_ISSUE_RANGE(lo_, hi_)                                                                   \
  do {                                                                                             \
    if ((lo_) < PPW_A && t + 1 < nkt) PP_ISSUE_A(org1, 1, t + 1, (t + 1) & 1, (lo_), ((hi_) < PPW_A ? (hi_) : PPW_A)); \
    if ((hi_) > PPW_A && t + 2 < nkt) PP_ISSUE_A(org0, 0, t + 2, t & 1, ((lo_) > PPW_A ? (lo_) - PPW_A : 0), (hi_) - PPW_A); \
  } while (0)
        PP_PERIOD(2 * PPW_A, ao, bo, wait_vm<0>(), (t + 1 < nkt));
#undef PP_ISSUE_RANGE
        bbuf ^= 1;
      }
      if (has_next) {
        const uint32_t orgn0 = BA::piece_org(wn * PPW_A, nm0, 0, p.lda), orgn1 = BA::piece_org(wn * PPW_A, nm0 + WROWS, 0, p.lda);
        PP_ISSUE_A(orgn0, 0, 0, 0, 0, PPW_A);
        PP_ISSUE_A(orgn1, 1, 0, 0, 0, PPW_A);
        if (nkt > 1) PP_ISSUE_A(orgn0, 0, 1, 1, 0, PPW_A);
      }
    }

    // ---- epilogue of tile (m0, n0): pp_tile's bf16 epilogue on the relocated staging slices
    {
      const int col_w = n0 + wcol0;
      f32x4 bv[DACT ? 1 : NI][4];
      if (!DACT) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = col_w + j * 32 + 4 * hi + 8 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[j][q][e] = (p.bias != nullptr && col + e < p.N) ? p.bias[col + e] : 0.f;
          }
      }
      constexpr int AD = MI < 3 ? MI : (MI == 4 ? 2 : 3);      // (256 x 256: three rows ahead spill beside the persistent loop's extra state)
      uint4 auxv[DACT ? AD : 1][CHUNKS];
      if (DACT) {
#pragma unroll
        for (int i = 0; i < AD; ++i) load_aux_rows<CHUNKS, CPR>(p, m0 + wrow0 + i * 32, col_w, lane, auxv[i]);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row_w = m0 + wrow0 + i * 32;
        if (DACT) {
#pragma unroll
          for (int c = 0; c < CHUNKS; ++c) {
            const int idx = c * 64 + lane, r = idx / CPR, cc = idx % CPR;
            *reinterpret_cast<uint4*>(wsp + r * RBY + cc * 16) = auxv[DACT ? i % AD : 0][c];
          }
          if (i + AD < MI) load_aux_rows<CHUNKS, CPR>(p, row_w + AD * 32, col_w, lane, auxv[DACT ? i % AD : 0]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            char* slot = wsp + l31 * RBY + (j * 32 + 4 * hi + 8 * q) * 2;
            float u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = DACT ? acc[i][j][4 * q + e] : acc[i][j][4 * q + e] + bv[DACT ? 0 : j][q][e];
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    m0 = nm0;
    n0 = nn0;
  }
#undef PP_ISSUE_A
#undef PP_ISSUE_B
#undef PP_COORDS
#undef PP_PERIOD
#undef PP_FRAGS_KS
#undef PP_MMA_ALL
#endif  // __HIP_DEVICE_COMPILE__
}

template <class CF, bool TB, int EPI>
__global__ __launch_bounds__(512) void pp_persist_kernel(G2Args p) {
  // workgroup w runs on XCD w % 8 (observed placement; only locality depends on it) and walks that XCD's chunk of the tile order
  const int nt = p.tiles_m * p.tiles_n, nwg = gridDim.x, w = blockIdx.x, x = w & 7, j = w >> 3;
  const int q = nt >> 3, r = nt & 7;
  const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, cnt = q + (x < r ? 1 : 0);
  const int J = (nwg >> 3) + ((nwg & 7) > x ? 1 : 0);
  pp_tiles_persist<CF, TB, EPI>(p, start + j, J, start + cnt);
}

inline int pp_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

template <class CF, bool TB, int EPI>
int pp_launch_persist(hipStream_t st, const G2Args& a) {
  static_assert(CF::SMEM <= 160 * 1024, "LDS exceeds the CU's 160 KiB");
  auto kern = pp_persist_kernel<CF, TB, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int nt = a.tiles_m * a.tiles_n, slots = pp_cu_count() * (CF::SMEM <= 80 * 1024 ? 2 : 1);
  hipLaunchKernelGGL(kern, dim3(nt < slots ? nt : slots), dim3(512), CF::SMEM, st, a);
  GOAT_LAUNCH_CHECK();
  return 0;
}

// persistent launches: bf16 results, unsplit, K-contiguous A; anything else -> GOAT_E_ARG
template <class CF>
int pp_dispatch_persist(hipStream_t st, const G2Args& a, int trans_a, int trans_b, int dtype_out, int epi, int split) {
  if (trans_a || split > 1 || dtype_out != GOAT_BF16) return GOAT_E_ARG;
  if (!trans_b) {
    switch (epi) {
      case GOAT_EPI_NONE: return pp_launch_persist<CF, false, GOAT_EPI_NONE>(st, a);
      case GOAT_EPI_GELU: return pp_launch_persist<CF, false, GOAT_EPI_GELU>(st, a);
      case GOAT_EPI_RELU: return pp_launch_persist<CF, false, GOAT_EPI_RELU>(st, a);
      case GOAT_EPI_MUL_DGELU: return pp_launch_persist<CF, false, GOAT_EPI_MUL_DGELU>(st, a);
      case GOAT_EPI_MUL_DRELU: return pp_launch_persist<CF, false, GOAT_EPI_MUL_DRELU>(st, a);
    }
  } else {
    switch (epi) {
      case GOAT_EPI_NONE: return pp_launch_persist<CF, true, GOAT_EPI_NONE>(st, a);
      case GOAT_EPI_GELU: return pp_launch_persist<CF, true, GOAT_EPI_GELU>(st, a);
      case GOAT_EPI_RELU: return pp_launch_persist<CF, true, GOAT_EPI_RELU>(st, a);
      case GOAT_EPI_MUL_DGELU: return pp_launch_persist<CF, true, GOAT_EPI_MUL_DGELU>(st, a);
      case GOAT_EPI_MUL_DRELU: return pp_launch_persist<CF, true, GOAT_EPI_MUL_DRELU>(st, a);
    }
  }
  return GOAT_E_ARG;
}

}  // namespace goat_g5

// gemm5.hip
int goat_g5_dispatch(hipStream_t st, const goat_g2::G2Args& a, int bm, int bn, int trans_a, int trans_b, int dtype_out, int epi,
                     int split, int nstage, bool persist);
int goat_g5_group(hipStream_t st, const goat_g2::GroupArgs& g, int bm, int bn, int nstage);
int goat_g5_group_sk(hipStream_t st, const goat_g2::GroupArgs& g, int bm, int bn, int nstage, void* ws, int64_t ws_bytes);
int64_t goat_g5_group_sk_ws_bytes(int bm, int bn);
