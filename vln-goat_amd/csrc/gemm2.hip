// goat_gemm_bf16 / goat_wgrad_grouped: C entry points, tile-order choice, and the instantiations of the 4-wave and
// 128-column tiles.  The tile code itself is gemm2_tile.hpp; the 8-wave 192/256-wide tiles are instantiated in gemm3.hip.
#include "gemm5_tile.hpp"

using namespace goat_g2;

// Tile-order parameter: the group height that minimises the operand bytes the eight per-XCD L2s have to fetch,
// sum over XCDs of (distinct tile rows * BM + distinct tile columns * BN) under the kernel's own blockIdx -> tile map
// (XCD x owns one contiguous chunk of the grouped order).  Brute force once per (tiles_m, tiles_n, bm, bn), then cached.
static int pick_group_m(int tiles_m, int tiles_n, int bm, int bn) {
  if (const char* e = getenv("GOAT_GEMM_GROUP_M")) {
    int g = atoi(e);
    return g < 1 ? 1 : (g > tiles_m ? tiles_m : g);
  }
  static std::mutex mu;
  static std::unordered_map<uint64_t, int> cache;
  const uint64_t key = ((uint64_t)tiles_m << 44) | ((uint64_t)tiles_n << 24) | ((uint64_t)bm << 12) | (uint64_t)bn;
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
      cost += (double)dm * bm + (double)dn * bn;
    }
    if (cost < best_cost - 1e-9) { best_cost = cost; best = gm; }
  }
  std::lock_guard<std::mutex> lk(mu);
  cache[key] = best;
  return best;
}


// tile = bm | bn << 16 (bn = 0: 128 columns).  4-wave / 128-column tiles live here, the rest in gemm3.hip.
static bool tile_ok(int bm, int bn, bool eight) {
  if (bn == 128 && bm == 96) return !eight;
  if (bn == 128) return bm == 64 || bm == 128 || bm == 256 ? (!eight || bm == 128) : false;
  if (eight) return false;
  return (bm == 128 && bn == 256) || (bm == 192 && bn == 256) || (bm == 256 && bn == 192) || (bm == 256 && bn == 256) || (bm == 192 && bn == 192);
}

// tiles of the ping-pong main loop: 64*MI x 128*NI, nstage = number of B buffers
static bool pp_tile_ok(int bm, int bn, int nstage) {
  return nstage == 2 && ((bm == 256 && bn == 256) || (bm == 192 && bn == 256) || (bm == 128 && bn == 256) || (bm == 256 && bn == 128) || (bm == 128 && bn == 128));
}

extern "C" int goat_gemm_bf16(void* stream, int trans_a, int trans_b, int dtype_out,
                              const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                              int M, int N, int Kc, const float* bias, int epilogue,
                              void* aux, int64_t ldaux, int split_k, int bm, int nstage, float* colsum) {
  if (!A || !B || !C) return GOAT_E_ARG;
  const bool eight = (nstage & GOAT_GEMM_8WAVES) != 0;
  const bool pp = (nstage & GOAT_GEMM_PP) != 0;               // the ping-pong main loop (gemm5_tile.hpp)
  const bool persist = (nstage & GOAT_GEMM_PERSIST) != 0;     // ... as one workgroup per CU walking the tiles
  nstage &= ~(GOAT_GEMM_8WAVES | GOAT_GEMM_PP | GOAT_GEMM_PERSIST);
  if (nstage < 2 || nstage > 4) return GOAT_E_ARG;
  if (persist && !pp) return GOAT_E_ARG;
  int bn = (bm >> 16) & 0xFFFF;
  bm &= 0xFFFF;
  if (bn == 0) bn = 128;
  if (colsum && !trans_a) return GOAT_E_ARG;
  if (M <= 0 || N <= 0 || Kc <= 0) return GOAT_E_SHAPE;
  if ((lda % 8) || (ldb % 8)) return GOAT_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return GOAT_E_SHAPE;
  // contraction tails: transposed operands are zero-filled by the bounds check; K-contiguous operands need
  // whole 64-wide tiles (callers route other shapes to goat_gemm_nt)
  if ((!trans_a || !trans_b) && (Kc % BK)) return GOAT_E_SHAPE;
  if (trans_a && !trans_b) return GOAT_E_ARG;
  if ((epilogue == GOAT_EPI_MUL_DGELU || epilogue == GOAT_EPI_MUL_DRELU) && (!aux || bias)) return GOAT_E_ARG;   // C = (A·B) * act'(aux): no bias
  if (epilogue == GOAT_EPI_ACCUM && (dtype_out != GOAT_F32 || bias)) return GOAT_E_ARG;
  if (split_k > 1 && (dtype_out != GOAT_F32 || (epilogue != GOAT_EPI_NONE && epilogue != GOAT_EPI_ACCUM) || bias)) return GOAT_E_ARG;
  if (pp ? (eight || !pp_tile_ok(bm, bn, nstage)) : !tile_ok(bm, bn, eight)) return GOAT_E_ARG;
  const int64_t a_rows = trans_a ? Kc : M, b_rows = trans_b ? Kc : N;
  const int64_t a_bytes = a_rows * lda * 2, b_bytes = b_rows * ldb * 2;
  if (a_bytes >= (1ll << 31) || b_bytes >= (1ll << 31)) return GOAT_E_SHAPE;

  G2Args a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
  a.M = M; a.N = N; a.Kc = Kc;
  a.tiles_m = (M + bm - 1) / bm;
  a.tiles_n = (N + bn - 1) / bn;
  a.a_bytes = (uint32_t)a_bytes; a.b_bytes = (uint32_t)b_bytes;
  a.colsum = colsum;
  a.accum = epilogue == GOAT_EPI_ACCUM;
  a.group_m = pick_group_m(a.tiles_m, a.tiles_n, bm, bn);
  const int kt = (Kc + BK - 1) / BK;
  if (split_k < 1) split_k = 1;
  if (split_k > kt) split_k = kt;
  a.k_tiles_per_split = (kt + split_k - 1) / split_k;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (pp) return goat_g5_dispatch(st, a, bm, bn, trans_a, trans_b, dtype_out, epilogue, split_k, nstage, persist);
  if (bn != 128 || bm == 96) return goat_g3_dispatch(st, a, bm, bn, trans_a, trans_b, dtype_out, epilogue, split_k, nstage);
  if (bm == 64) return dispatch_layout<T64>(st, a, trans_a, trans_b, dtype_out, epilogue, split_k, nstage);
  if (bm == 256) return dispatch_layout<T256>(st, a, trans_a, trans_b, dtype_out, epilogue, split_k, nstage);
  if (eight) return dispatch_layout<T128X8>(st, a, trans_a, trans_b, dtype_out, epilogue, split_k, nstage);
  return dispatch_layout<T128>(st, a, trans_a, trans_b, dtype_out, epilogue, split_k, nstage);
}

template <class CF>
static int group_stages(hipStream_t st, const GroupArgs& g, int nstage) {
  constexpr int STAGE = smem_bytes<CF, true, true, 1>();
  if (nstage == 2) return launch_group<CF, 2>(st, g);
  if constexpr (3 * STAGE <= 160 * 1024) {
    if (nstage == 3) return launch_group<CF, 3>(st, g);
  }
  if constexpr (4 * STAGE <= 160 * 1024 && CF::BM * CF::BN <= 128 * 128) {
    if (nstage == 4) return launch_group<CF, 4>(st, g);
  }
  return GOAT_E_ARG;
}

// argument block of a grouped launch from the caller's problem list (validation shared by the two entry points)
static int build_group(const goat_wgrad_problem* probs, int n, int bm, int bn, GroupArgs& g) {
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
    if (q.ld_dy >= (1ll << 31) || q.ld_x >= (1ll << 31) || q.ld_dw >= (1ll << 31)) return GOAT_E_SHAPE;
    GroupProb& a = g.prob[i];
    a.A = q.dy; a.B = q.x; a.C = q.dw; a.colsum = q.dbias;
    a.lda = (int)q.ld_dy; a.ldb = (int)q.ld_x; a.ldc = (int)q.ld_dw;
    a.M = q.n_out; a.N = q.n_in; a.Kc = q.rows;
    a.accum = q.accumulate ? 1 : 0;
    const int tiles_m = (q.n_out + bm - 1) / bm, tiles_n = (q.n_in + bn - 1) / bn;
    a.group_m = (short)pick_group_m(tiles_m, tiles_n, bm, bn);
    a.pad_ = 0;
    g.tile_start[i] = tiles;
    tiles += tiles_m * tiles_n;
  }
  for (int i = n; i <= GROUP_MAX; ++i) g.tile_start[i] = tiles;
  return 0;
}

extern "C" int goat_wgrad_grouped(void* stream, const goat_wgrad_problem* probs, int n, int bm, int nstage) {
  if (!probs || n < 1 || n > GROUP_MAX) return GOAT_E_ARG;
  const bool eight = (nstage & GOAT_GEMM_8WAVES) != 0;       // as in goat_gemm_bf16: the 128-row tile on eight waves
  const bool pp = (nstage & GOAT_GEMM_PP) != 0;
  nstage &= ~(GOAT_GEMM_8WAVES | GOAT_GEMM_PP);
  int bn = (bm >> 16) & 0xFFFF;
  bm &= 0xFFFF;
  if (bn == 0) bn = 128;
  if (nstage < 2 || nstage > 4 || (bm & (bm - 1)) || (bn & (bn - 1))) return GOAT_E_ARG;
  if (pp ? (eight || !pp_tile_ok(bm, bn, nstage) || (bm == 128 && bn == 128)) : !tile_ok(bm, bn, eight)) return GOAT_E_ARG;
  GroupArgs g;
  if (int e = build_group(probs, n, bm, bn, g)) return e;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (pp) return goat_g5_group(st, g, bm, bn, nstage);
  if (bn != 128) return goat_g3_group(st, g, bm, bn, nstage);
  if (bm == 64) return group_stages<T64>(st, g, nstage);
  if (bm == 256) return group_stages<T256>(st, g, nstage);
  if (eight) return group_stages<T128X8>(st, g, nstage);
  return group_stages<T128>(st, g, nstage);
}

// Contraction-balanced form of the grouped launch (gemm5_tile.hpp: pp_group_sk_kernel): ping-pong tiles 256x256 / 128x256 / 256x128.
extern "C" int goat_wgrad_grouped_balanced(void* stream, const goat_wgrad_problem* probs, int n, int bm, void* workspace, int64_t workspace_bytes) {
  if (!probs || n < 1 || n > GROUP_MAX) return GOAT_E_ARG;
  int bn = (bm >> 16) & 0xFFFF;
  bm &= 0xFFFF;
  if (bn == 0) bn = 128;
  if (goat_g5_group_sk_ws_bytes(bm, bn) < 0) return GOAT_E_ARG;
  GroupArgs g;
  if (int e = build_group(probs, n, bm, bn, g)) return e;
  return goat_g5_group_sk(reinterpret_cast<hipStream_t>(stream), g, bm, bn, 2, workspace, workspace_bytes);
}

extern "C" int goat_wgrad_balanced_ws_bytes(int bm) {
  int bn = (bm >> 16) & 0xFFFF;
  bm &= 0xFFFF;
  if (bn == 0) bn = 128;
  return (int)goat_g5_group_sk_ws_bytes(bm, bn);       // 256 slots of <= 264 KiB: fits an int
}
