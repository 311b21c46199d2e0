// esa_prims.h -- device-wide primitives (scan, radix sort) of the ESA engine.
#pragma once
#include "esa_common.h"

enum { SCAN_SUM = 0, SCAN_MAX = 1 };

// workspace (in u32 words) a scan of n elements needs
u64 scan_workspace_words(u64 n);
// out[i] = op over in[0..i) (exclusive) or in[0..i] (inclusive); in == out ok
int scan_u32(int op, const u32 *in, u32 *out, u64 n, bool inclusive, u32 *ws,
             hipStream_t st);

// workspace (u32 words) for sorting n pairs
u64 radix_workspace_words(u64 n);
// Stable LSD radix sort of (key, value) pairs (K = u64 or u32 keys) on the digits
// (key >> shifts[p]) & ((1 << widths[p]) - 1), p = 0..npasses-1 (least
// significant digit first, widths <= 8).  Ping-pongs between (a) and (b); the
// result is in (a) if npasses is even, else in (b).  If ev_pairs != NULL, a
// start/stop event pair is recorded around every scatter launch
// (ev_pairs[2*i], ev_pairs[2*i+1]) and *n_ev is advanced.  dig_a/dig_b: optional
// n-byte scratch arrays; when given, every scatter pass also writes the next
// pass's digit of each pair as one byte and the next histogram reads those
// bytes instead of the keys.
template <typename K, typename V>
int radix_sort_pairs(K *keys_a, V *vals_a, K *keys_b, V *vals_b, u64 n,
                     const int *shifts, const int *widths, int npasses,
                     u32 *ws, hipStream_t st, hipEvent_t *ev_pairs,
                     int *n_ev, u8 *dig_a = nullptr, u8 *dig_b = nullptr);

// For a caller that makes the first pass of a sort itself (the DNA keygen does:
// it writes its keys partitioned on the lowest digit straight away).  ws is the
// sort's workspace; its first ceil(n / 4096) * 256 words are the tile-major digit
// histogram the caller has filled (row t = counts of tile t, 4096 pairs per
// tile); on return every entry is the global output index where that tile's
// pairs of that digit start, exactly what the sort's own passes use.
int radix_scan_tile_hist(u32 *ws, u64 n, hipStream_t st);
// the same scan over `nrows` rows of a tile-major histogram that lies anywhere
// (rows of 256 counters; scanws: radix_rows_workspace_words(nrows) words)
u64 radix_rows_workspace_words(u64 nrows);
int radix_scan_tile_rows(u32 *hist, u32 nrows, u32 *scanws, hipStream_t st);

// One pass of the same sort whose VALUES are not read but made on the fly:
// value of input pair i = offset + (head of the tie group of entry i), from the
// tie bitmap ("entry i has the key of entry i-1", one bit per entry) and the
// per-word carries (head of the group that reaches into word w).  The rank-table
// build uses it for its first partition pass: no array of heads is written or
// read.  Result in (keys_b, vals_b).
struct GroupHeadValues {
  const u64 *tiebits;
  const u32 *carry;
  u64 nwords;     // words of the bitmap (and entries of carry)
  u32 offset;
};
int radix_pass_group_heads(const u32 *keys_a, GroupHeadValues gen, u32 *keys_b,
                           u32 *vals_b, u64 n, int shift, int width, u32 *ws,
                           hipStream_t st);

// One UNSTABLE partition pass of (u32, u32) pairs on (key >> shift) & ((1 <<
// width) - 1): pairs of one digit end up together, in any order (LDS atomics
// instead of the ballot ranking: bound by bandwidth, not by instructions).  For
// the FIRST partition pass of the rank-table build, whose keys are positions
// in no order that matters (a second pass on higher bits has to be a stable
// one).  radix_pass_group_heads partitions this way too.
int radix_partition_u32(const u32 *keys_a, const u32 *vals_a, u32 *keys_b, u32 *vals_b, u64 n,
                        int shift, int width, u32 *ws, hipStream_t st);
