// The 8-wave 192/256-wide tiles of goat_gemm_bf16 / goat_wgrad_grouped (gemm2_tile.hpp), instantiated in their own
// translation unit so that they compile in parallel with gemm2.hip.
#include "gemm2_tile.hpp"

using namespace goat_g2;

int goat_g3_dispatch(hipStream_t st, const G2Args& a, int bm, int bn, int trans_a, int trans_b, int dtype_out, int epi,
                     int split, int nstage) {
  if (bm == 256 && bn == 256) return dispatch_layout<T256x256>(st, a, trans_a, trans_b, dtype_out, epi, split, nstage);
  if (bm == 192 && bn == 256) return dispatch_layout<T192x256>(st, a, trans_a, trans_b, dtype_out, epi, split, nstage);
  if (bm == 256 && bn == 192) return dispatch_layout<T256x192>(st, a, trans_a, trans_b, dtype_out, epi, split, nstage);
  if (bm == 128 && bn == 256) return dispatch_layout<T128x256>(st, a, trans_a, trans_b, dtype_out, epi, split, nstage);
  if (bm == 192 && bn == 192) return dispatch_layout<T192x192>(st, a, trans_a, trans_b, dtype_out, epi, split, nstage);
  if (bm == 96 && bn == 128) return dispatch_layout<T96>(st, a, trans_a, trans_b, dtype_out, epi, split, nstage);
  return GOAT_E_ARG;
}

template <class CF>
static int group_stages3(hipStream_t st, const GroupArgs& g, int nstage) {
  constexpr int STAGE = smem_bytes<CF, true, true, 1>();
  if (nstage == 2) return launch_group<CF, 2>(st, g);
  if constexpr (3 * STAGE <= 160 * 1024) {
    if (nstage == 3) return launch_group<CF, 3>(st, g);
  }
  return GOAT_E_ARG;
}

int goat_g3_group(hipStream_t st, const GroupArgs& g, int bm, int bn, int nstage) {
  if (bm == 256 && bn == 256) return group_stages3<T256x256>(st, g, nstage);
  if (bm == 128 && bn == 256) return group_stages3<T128x256>(st, g, nstage);
  return GOAT_E_ARG;
}
