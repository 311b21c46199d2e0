// esa_devutil.h -- wave / block level scan helpers shared by the kernels
// (wave64, 256-thread blocks).
#pragma once
#include "esa_prims.h"

constexpr int SC_THREADS = 256;

template <int OP> __device__ __forceinline__ u32 sc_op(u32 a, u32 b) {
  return OP == SCAN_SUM ? a + b : (a > b ? a : b);
}

// inclusive scan across the 64 lanes of a wave
template <int OP> __device__ __forceinline__ u32 wave_scan_incl(u32 v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 o = __shfl_up(v, d, 64);
    if (lane >= d) v = sc_op<OP>(v, o);
  }
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
  u32 prev = __shfl_up(inc, 1, 64);
  if (lane == 0) prev = 0;
  return sc_op<OP>(carry, prev);
}


__device__ __forceinline__ u32 block_scan_excl_sum(u32 v, u32 *total, u32 *lds4) {
  return block_scan_excl<SCAN_SUM>(v, total, lds4);
}
__device__ __forceinline__ u32 block_scan_excl_max(u32 v, u32 *total, u32 *lds4) {
  return block_scan_excl<SCAN_MAX>(v, total, lds4);
}
