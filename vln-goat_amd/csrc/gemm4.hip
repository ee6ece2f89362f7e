// The 256 x 256 tile on four waves (128 x 128 wave patches, GOAT_GEMM_WIDE_PATCH) for the weight-gradient layout
// (dW = dY^T X: both operands transposed, float32 output) of goat_gemm_bf16 / goat_wgrad_grouped — its own translation unit
// (256 accumulator registers per lane: the slowest kernels of the build to compile).
#include "gemm2_tile.hpp"

using namespace goat_g2;

int goat_g4_dispatch(hipStream_t st, const G2Args& a, int split, int nstage) {
  if (nstage != 2) return GOAT_E_ARG;
  if (split > 1) return launch2s<T256x256W4, true, true, float, GOAT_EPI_NONE, true, 2>(st, a, split);
  return launch2s<T256x256W4, true, true, float, GOAT_EPI_NONE, false, 2>(st, a, 1);
}

int goat_g4_group(hipStream_t st, const GroupArgs& g, int nstage) {
  if (nstage != 2) return GOAT_E_ARG;
  return launch_group<T256x256W4, 2>(st, g);
}
