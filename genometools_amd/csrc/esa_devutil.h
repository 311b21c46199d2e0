// esa_devutil.h -- wave / block level scan helpers shared by the kernels
// (wave64, 256-thread blocks).
#pragma once
#include "esa_prims.h"

constexpr int SC_THREADS = 256;

template <int OP> __device__ __forceinline__ u32 sc_op(u32 a, u32 b) {
  return OP == SCAN_SUM ? a + b : (a > b ? a : b);
}

// inclusive scan across the 64 lanes of a wave: four shifts inside the rows of
// 16 lanes and two row broadcasts, all as DPP operands of the VALU (a
// __shfl_up goes through the LDS crossbar: six dependent round trips of ~100
// cycles each, which was most of the latency of a block scan).  0 is the
// identity of both operations; a lane without a source reads it.
template <int OP> __device__ __forceinline__ u32 wave_scan_incl(u32 v) {
  v = sc_op<OP>(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false));  // row_shr:1
  v = sc_op<OP>(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false));  // row_shr:2
  v = sc_op<OP>(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false));  // row_shr:4
  v = sc_op<OP>(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false));  // row_shr:8
  v = sc_op<OP>(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false));  // row_bcast:15
  v = sc_op<OP>(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false));  // row_bcast:31
  return v;
}

// block-wide exclusive scan of one value per thread (256 threads);
// returns the exclusive prefix, *total gets the block total
template <int OP, int THREADS = SC_THREADS>
__device__ __forceinline__ u32 block_scan_excl(u32 v, u32 *total, u32 *lds4) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u32 inc = wave_scan_incl<OP>(v);
  if (lane == 63) lds4[w] = inc;
  __syncthreads();
  u32 carry = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < THREADS / 64; i++) {
    u32 s = lds4[i];
    if (i < w) carry = sc_op<OP>(carry, s);
    tot = sc_op<OP>(tot, s);
  }
  __syncthreads();
  *total = tot;
  // the inclusive value of the lane below (wave_shr:1; lane 0 reads 0)
  const u32 prev = (u32) __builtin_amdgcn_update_dpp(0, (int) inc, 0x138, 0xf, 0xf, false);
  return sc_op<OP>(carry, prev);
}


__device__ __forceinline__ u32 block_scan_excl_sum(u32 v, u32 *total, u32 *lds4) {
  return block_scan_excl<SCAN_SUM>(v, total, lds4);
}
__device__ __forceinline__ u32 block_scan_excl_max(u32 v, u32 *total, u32 *lds4) {
  return block_scan_excl<SCAN_MAX>(v, total, lds4);
}
