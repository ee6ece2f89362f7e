// The ping-pong main loop (gemm5_tile.hpp) instantiated for its tile family; reached through goat_gemm_bf16 /
// goat_wgrad_grouped with nstage | GOAT_GEMM_PP (nstage 2 = number of B buffers in LDS).
#include "gemm5_tile.hpp"

using namespace goat_g5;

int goat_g5_dispatch(hipStream_t st, const goat_g2::G2Args& a, int bm, int bn, int trans_a, int trans_b, int dtype_out, int epi,
                     int split, int nstage, bool persist) {
  if (persist) {
    if (bm == 256 && bn == 256 && nstage == 2) return pp_dispatch_persist<P256x256>(st, a, trans_a, trans_b, dtype_out, epi, split);
    if (bm == 192 && bn == 256 && nstage == 2) return pp_dispatch_persist<P192x256>(st, a, trans_a, trans_b, dtype_out, epi, split);
    if (bm == 128 && bn == 256 && nstage == 2) return pp_dispatch_persist<P128x256>(st, a, trans_a, trans_b, dtype_out, epi, split);
    if (bm == 256 && bn == 128 && nstage == 2) return pp_dispatch_persist<P256x128>(st, a, trans_a, trans_b, dtype_out, epi, split);
    if (bm == 128 && bn == 128 && nstage == 2) return pp_dispatch_persist<P128x128>(st, a, trans_a, trans_b, dtype_out, epi, split);
    return GOAT_E_ARG;
  }
  if (bm == 256 && bn == 256 && nstage == 2) return pp_dispatch_layout<P256x256>(st, a, trans_a, trans_b, dtype_out, epi, split);
  if (bm == 192 && bn == 256 && nstage == 2) return pp_dispatch_layout<P192x256>(st, a, trans_a, trans_b, dtype_out, epi, split);
  if (bm == 128 && bn == 256 && nstage == 2) return pp_dispatch_layout<P128x256>(st, a, trans_a, trans_b, dtype_out, epi, split);
  if (bm == 256 && bn == 128 && nstage == 2) return pp_dispatch_layout<P256x128>(st, a, trans_a, trans_b, dtype_out, epi, split);
  if (bm == 128 && bn == 128 && nstage == 2) return pp_dispatch_layout<P128x128>(st, a, trans_a, trans_b, dtype_out, epi, split);
  return GOAT_E_ARG;
}

int goat_g5_group(hipStream_t st, const goat_g2::GroupArgs& g, int bm, int bn, int nstage) {
  if (bm == 256 && bn == 256 && nstage == 2) return pp_launch_group<P256x256>(st, g);
  if (bm == 128 && bn == 256 && nstage == 2) return pp_launch_group<P128x256>(st, g);
  if (bm == 256 && bn == 128 && nstage == 2) return pp_launch_group<P256x128>(st, g);
  return GOAT_E_ARG;
}

int goat_g5_group_sk(hipStream_t st, const goat_g2::GroupArgs& g, int bm, int bn, int nstage, void* ws, int64_t ws_bytes) {
  if (bm == 256 && bn == 256 && nstage == 2) return pp_launch_group_sk<P256x256>(st, g, ws, ws_bytes);
  if (bm == 128 && bn == 256 && nstage == 2) return pp_launch_group_sk<P128x256>(st, g, ws, ws_bytes);
  if (bm == 256 && bn == 128 && nstage == 2) return pp_launch_group_sk<P256x128>(st, g, ws, ws_bytes);
  return GOAT_E_ARG;
}

int64_t goat_g5_group_sk_ws_bytes(int bm, int bn) {
  if (bm == 256 && bn == 256) return pp_sk_ws_bytes<P256x256>();
  if (bm == 128 && bn == 256) return pp_sk_ws_bytes<P128x256>();
  if (bm == 256 && bn == 128) return pp_sk_ws_bytes<P256x128>();
  return -1;
}
