// Argument block shared by the attention kernels (attention.hip: one wave per tile, any dtype / length; attention2.hip: the
// LDS-staged bf16 kernels for sequences of up to 128 rows).
#pragma once
#include "common.hpp"

struct AttnArgs {
  const void *Q, *K, *V, *O, *dO;
  void *Ow, *dQ, *dK, *dV;
  int64_t q_rs, q_bs, k_rs, k_bs, v_rs, v_bs, o_rs, o_bs, do_rs, do_bs;
  int64_t dq_rs, dq_bs, dk_rs, dk_bs, dv_rs, dv_bs;
  const float *kmask, *bias;
  float* lse;
  float* dbias;
  int B, nh, Lq, Lk;
  float scale, p;
  uint64_t seed, offset;
  const uint64_t* rng_dev;
};

// attention2.hip; return GOAT_E_SHAPE when the problem is outside their range (the caller then uses the general kernels)
int goat_attn2_fwd(hipStream_t st, const AttnArgs& a);
int goat_attn2_bwd(hipStream_t st, const AttnArgs& a);
