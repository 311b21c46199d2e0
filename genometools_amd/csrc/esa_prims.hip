// esa_prims.hip -- device-wide primitives of the ESA engine, hand-written for
// gfx950 (wave64): u32 scans (sum / max), and a stable LSD radix sort of
// (u64 key, u32 value) pairs built from LDS-staged digit histograms and
// wavefront ballot ranking.  No rocPRIM/hipCUB.
#include "esa_prims.h"
#include "esa_devutil.h"
#include <stdlib.h>
#include <vector>

// ===========================================================================
// scans
// ===========================================================================
namespace {

constexpr int SC_ITEMS = 16;
constexpr int SC_TILE = SC_THREADS * SC_ITEMS;  // 4096

template <int OP>
__global__ __launch_bounds__(SC_THREADS) void k_scan_reduce(
    const u32 *__restrict__ in, u64 n, u32 *__restrict__ blocksum) {
  __shared__ u32 lds4[4];
  const u64 base = (u64) blockIdx.x * SC_TILE + (u64) threadIdx.x * SC_ITEMS;
  u32 acc = 0;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; i++) {
    u64 idx = base + i;
    if (idx < n) acc = sc_op<OP>(acc, in[idx]);
  }
  u32 tot;
  (void) block_scan_excl<OP>(acc, &tot, lds4);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

// scans one tile per block; carry-in from blockprefix (exclusive prefix of the
// block sums) when not NULL
template <int OP, bool INCLUSIVE>
__global__ __launch_bounds__(SC_THREADS) void k_scan_tile(
    const u32 *__restrict__ in, u32 *__restrict__ out, u64 n,
    const u32 *__restrict__ blockprefix) {
  __shared__ u32 lds4[4];
  const u64 base = (u64) blockIdx.x * SC_TILE + (u64) threadIdx.x * SC_ITEMS;
  u32 v[SC_ITEMS];
  u32 acc = 0;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; i++) {
    u64 idx = base + i;
    v[i] = idx < n ? in[idx] : 0;
    acc = sc_op<OP>(acc, v[i]);
  }
  u32 tot;
  u32 pre = block_scan_excl<OP>(acc, &tot, lds4);
  if (blockprefix != nullptr) pre = sc_op<OP>(pre, blockprefix[blockIdx.x]);
#pragma unroll
  for (int i = 0; i < SC_ITEMS; i++) {
    u64 idx = base + i;
    u32 incl = sc_op<OP>(pre, v[i]);
    if (idx < n) out[idx] = INCLUSIVE ? incl : pre;
    pre = incl;
  }
}

}  // namespace

u64 scan_workspace_words(u64 n) {
  u64 words = 0;
  while (n > SC_TILE) {
    n = div_up(n, SC_TILE);
    words += n + 16;
  }
  return words + 16;
}

template <int OP>
static int scan_rec(const u32 *in, u32 *out, u64 n, bool inclusive, u32 *ws,
                    hipStream_t st) {
  if (n == 0) return 0;
  const u64 nblocks = div_up(n, SC_TILE);
  if (nblocks == 1) {
    if (inclusive)
      k_scan_tile<OP, true><<<1, SC_THREADS, 0, st>>>(in, out, n, nullptr);
    else
      k_scan_tile<OP, false><<<1, SC_THREADS, 0, st>>>(in, out, n, nullptr);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  u32 *sums = ws;
  k_scan_reduce<OP><<<(u32) nblocks, SC_THREADS, 0, st>>>(in, n, sums);
  HIP_TRY(hipGetLastError());
  TRY(scan_rec<OP>(sums, sums, nblocks, false, ws + nblocks + 16, st));
  if (inclusive)
    k_scan_tile<OP, true><<<(u32) nblocks, SC_THREADS, 0, st>>>(in, out, n, sums);
  else
    k_scan_tile<OP, false><<<(u32) nblocks, SC_THREADS, 0, st>>>(in, out, n, sums);
  HIP_TRY(hipGetLastError());
  return 0;
}

int scan_u32(int op, const u32 *in, u32 *out, u64 n, bool inclusive, u32 *ws,
             hipStream_t st) {
  return op == SCAN_SUM ? scan_rec<SCAN_SUM>(in, out, n, inclusive, ws, st)
                        : scan_rec<SCAN_MAX>(in, out, n, inclusive, ws, st);
}

// ===========================================================================
// radix sort
// ===========================================================================
namespace {

constexpr int RS_THREADS = 512;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;       // 4096 pairs per block
constexpr int RS_WAVE_CHUNK = RS_ITEMS * 64;         // 1024 consecutive pairs
constexpr int RADIX = 256;
constexpr int OS_MAXPASS = 8;
constexpr u64 OS_AUX_WORDS = 2 * OS_MAXPASS * RADIX + 64;  // totals, bases, flag

// Per-tile digit histogram, tile-major: hist[tile * 256 + d] (one coalesced
// 1 KB row per tile, for this kernel's store and the scatter kernel's load).
// The column scan below turns it into every tile's global write base per
// digit.
template <typename K>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(
    const K *__restrict__ keys, u64 n, int shift, u32 mask,
    u32 *__restrict__ hist, u32 ntiles) {
  __shared__ u32 h[RS_WAVES][RADIX];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < RS_WAVES * RADIX; i += RS_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const u64 base = (u64) blockIdx.x * RS_TILE;
  constexpr int PER16 = 16 / (int) sizeof(K);   // keys per 16-byte load
  if (base + RS_TILE <= n && (reinterpret_cast<uintptr_t>(keys) & 15) == 0) {
    // full tile: 16 bytes per lane and load (the order inside the tile does not
    // matter to a histogram)
    K k[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS / PER16; j++) {
      const uint4 q = *reinterpret_cast<const uint4 *>(
          keys + base + ((u64) j * RS_THREADS + tid) * PER16);
      if (sizeof(K) == 8) {
        k[j * PER16] = (K) (((u64) q.y << 32) | q.x);
        k[j * PER16 + 1] = (K) (((u64) q.w << 32) | q.z);
      } else {
        k[j * PER16] = (K) q.x;
        k[j * PER16 + 1] = (K) q.y;
        if (PER16 > 2) {
          k[j * PER16 + (PER16 > 2 ? 2 : 0)] = (K) q.z;
          k[j * PER16 + (PER16 > 2 ? 3 : 0)] = (K) q.w;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) atomicAdd(&h[w][(u32) (k[j] >> shift) & mask], 1u);
  } else {
#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) {
      u64 idx = base + (u64) j * RS_THREADS + tid;
      if (idx < n) {
        u32 d = (u32) (keys[idx] >> shift) & mask;
        atomicAdd(&h[w][d], 1u);
      }
    }
  }
  __syncthreads();
  if (tid < RADIX) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES; i++) c += h[i][tid];
    hist[(u64) blockIdx.x * RADIX + tid] = c;
  }
}

// Exclusive scan of the tile-major histogram in digit-major order:
//   out[t][d] = sum_{d' < d} total[d'] + sum_{t' < t} hist[t'][d]
// (1) column sums of chunks of CS_ROWS tiles, (2) one block turns them into
// chunk bases, (3) every chunk rewrites its rows with running column prefixes.
// Thread d owns column d; every row access is one coalesced 1 KB line.
constexpr int CS_ROWS = 512;

__global__ __launch_bounds__(RADIX) void k_cs_chunksum(
    const u32 *__restrict__ hist, u32 ntiles, u32 *__restrict__ chunksum) {
  const u32 t0 = blockIdx.x * CS_ROWS;
  const u32 t1 = t0 + CS_ROWS < ntiles ? t0 + CS_ROWS : ntiles;
  // eight independent loads in flight per thread (one dependent add per row
  // made this walk latency-bound: 55 us for 512 rows)
  u32 a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u32 t = t0;
  for (; t + 8 <= t1; t += 8) {
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] += hist[(u64) (t + k) * RADIX + threadIdx.x];
  }
  for (; t < t1; t++) a[0] += hist[(u64) t * RADIX + threadIdx.x];
  chunksum[(u64) blockIdx.x * RADIX + threadIdx.x] =
      ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// one workgroup of 4 x 256 threads: thread (q, d) walks quarter q of column d
constexpr int CB_PARTS = 4;
__global__ __launch_bounds__(RADIX * CB_PARTS) void k_cs_chunkbase(
    u32 *__restrict__ chunksum, u32 nchunks) {
  __shared__ u32 s_scan[RADIX * CB_PARTS / 64];
  __shared__ u32 s_part[CB_PARTS][RADIX];
  __shared__ u32 s_colbase[RADIX];
  const u32 d = threadIdx.x & (RADIX - 1), q = threadIdx.x / RADIX;
  const u32 per = (nchunks + CB_PARTS - 1) / CB_PARTS;
  const u32 c0 = q * per < nchunks ? q * per : nchunks;
  const u32 c1 = c0 + per < nchunks ? c0 + per : nchunks;
  // column totals of the quarters, then the exclusive scan of the whole columns
  // over the digits
  u32 tot = 0;
  for (u32 c = c0; c < c1; c++) tot += chunksum[(u64) c * RADIX + d];
  s_part[q][d] = tot;
  __syncthreads();
  u32 col = 0, before = 0;
  if (q == 0) {
#pragma unroll
    for (int i = 0; i < CB_PARTS; i++) col += s_part[i][d];
  }
  // (the first 256 threads carry the column totals, the others add nothing and
  // come behind them)
  u32 all;
  const u32 colbase =
      block_scan_excl<SCAN_SUM, RADIX * CB_PARTS>(q == 0 ? col : 0u, &all, s_scan);
  if (q == 0) s_colbase[d] = colbase;
  __syncthreads();
  for (u32 i = 0; i < q; i++) before += s_part[i][d];
  u32 run = s_colbase[d] + before;
  for (u32 c = c0; c < c1; c++) {
    const u32 v = chunksum[(u64) c * RADIX + d];
    chunksum[(u64) c * RADIX + d] = run;
    run += v;
  }
}

__global__ __launch_bounds__(RADIX) void k_cs_rows(u32 *__restrict__ hist, u32 ntiles,
                                                   const u32 *__restrict__ chunkbase) {
  const u32 t0 = blockIdx.x * CS_ROWS;
  const u32 t1 = t0 + CS_ROWS < ntiles ? t0 + CS_ROWS : ntiles;
  u32 run = chunkbase[(u64) blockIdx.x * RADIX + threadIdx.x];
  u32 t = t0;
  for (; t + 8 <= t1; t += 8) {   // loads of eight rows in flight, then the running sums
    u32 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = hist[(u64) (t + k) * RADIX + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      hist[(u64) (t + k) * RADIX + threadIdx.x] = run;
      run += v[k];
    }
  }
  for (; t < t1; t++) {
    const u32 v = hist[(u64) t * RADIX + threadIdx.x];
    hist[(u64) t * RADIX + threadIdx.x] = run;
    run += v;
  }
}

// the same histogram from the digit bytes the previous scatter pass left
// behind (1 B per pair instead of the 8-byte key)
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist_bytes(
    const u8 *__restrict__ dig, u64 n, u32 *__restrict__ hist, u32 ntiles) {
  __shared__ u32 h[RS_WAVES][RADIX];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < RS_WAVES * RADIX; i += RS_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const u64 base = (u64) blockIdx.x * RS_TILE + (u64) tid * RS_ITEMS;
  if (base + RS_ITEMS <= n) {
    static_assert(RS_ITEMS == 8, "one 8-byte load per thread");
    const u64 v = *reinterpret_cast<const u64 *>(dig + base);
#pragma unroll
    for (int j = 0; j < 8; j++) atomicAdd(&h[w][(u32) (v >> (8 * j)) & 255u], 1u);
  } else {
    for (int j = 0; j < RS_ITEMS; j++)
      if (base + j < n) atomicAdd(&h[w][dig[base + j]], 1u);
  }
  __syncthreads();
  if (tid < RADIX) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES; i++) c += h[i][tid];
    hist[(u64) blockIdx.x * RADIX + tid] = c;
  }
}

// lanes of this wave that hold the same 8-bit digit (all 64 lanes active)
__device__ __forceinline__ u64 match_digit(u32 d) {
  u64 m = ~0ull;
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const bool bit = (d >> b) & 1u;
    const u64 bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}

// Stable scatter of one tile.  Wave w owns the 1024 consecutive pairs
// [w*1024, w*1024+1024) of the tile; item j of lane l is pair w*1024+j*64+l,
// so every load is a fully coalesced 512-byte (keys) / 256-byte (values)
// wave access and the (wave, item, lane) order is the input order.  Ranking:
// ballot match inside the wave + a per-wave running digit counter in LDS.
// The tile is then staged in LDS in digit order and written out so that
// neighbouring lanes write neighbouring addresses of the same digit run.
// (Two other ranking schemes were measured in-process against this one at
// 3 Gbp -- register ballots + pipelined LDS counter adds: 122 ms per 6 passes;
// lane masks through LDS atomic-or: 110 ms; this one: 106.5 ms -- so ranking
// is not what limits the kernel.)
constexpr u32 OS_CHUNK = 8;
constexpr u32 OS_AGG = 1u << 30, OS_INCL = 2u << 30, OS_VAL = (1u << 30) - 1u;
constexpr u32 OS_SPIN_LIMIT = 1u << 21;
constexpr int OS_WIN = 8;

__device__ __forceinline__ u32 os_tile(u32 b) {
  const u32 x = b & 7u, q = b >> 3;
  return (q / OS_CHUNK) * (8u * OS_CHUNK) + x * OS_CHUNK + (q % OS_CHUNK);
}

// XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its L2),
// so they take CONSECUTIVE tiles; the output runs of consecutive tiles are
// adjacent in every digit region, and the cache line two runs share is then
// completed inside one L2 instead of leaving two XCDs as two partial writes.
__device__ __forceinline__ u32 xcd_tile(u32 b, u32 ntiles) {
  const u32 per = (ntiles + 7u) >> 3;
  return (b & 7u) * per + (b >> 3);
}

// The kernel is VALU-bound (a wave64 instruction occupies its 16-lane SIMD for
// four cycles; ISA count x 4 cycles accounts for the measured time), so the
// ranking is written for instruction count:
//  * the lanes with the same digit are  m &= ~(ballot(bit b) ^ (own bit b spread
//    over the word))  -- one sign-extending bit-field extract, one compare and
//    one v_bitop3 per half and bit, against the nine instructions the compiler
//    makes of the select form  m &= bit ? bal : ~bal;
//  * v_mbcnt counts the matching lanes below this one without a lane mask;
//  * every lane reads the wave's running counter itself (same-address LDS
//    reads broadcast), then the first lane of the group adds the group size:
//    no leader search and no cross-lane shuffle.  LDS operations of one wave
//    execute in order, so the next item's read sees this item's update.
//  * full tiles (all but the last) run without per-item range checks.
// (Measured and dropped: 16-byte loads, turned into the (item, lane) order
// through the wave's own slice of the staging area -- 81.4 against 78.2 ms for
// five 3 G-pair passes; the histogram kernel, which needs no order, does gain
// from 16-byte loads.)
typedef __attribute__((address_space(3))) volatile u16 lds_vu16;

// no generator: the values are read
struct ReadValues {
  static constexpr bool active = false;
  __device__ __forceinline__ void prefetch(u64, int, u64 &, u32 &) const {}
  __device__ __forceinline__ u32 make(u64, u64, u32) const { return 0; }
};
// head of the tie group of entry i (GroupHeadValues of esa_prims.h).  The 64
// lanes of a wave ask, per item, for 64 consecutive entries = ONE bitmap word
// (item j of lane l is entry base + j*64 + l, base a multiple of 64): lane j
// fetches word and carry of item j once, the items take them by lane read.
struct MakeGroupHeads {
  static constexpr bool active = true;
  GroupHeadValues g;
  __device__ __forceinline__ void prefetch(u64 wave_first, int lane, u64 &t, u32 &c) const {
    if (lane < RS_ITEMS) {
      u64 w = (wave_first >> 6) + (u64) lane;
      if (w >= g.nwords) w = g.nwords - 1;      // behind the table: never used
      t = g.tiebits[w];
      c = g.carry[w];
    }
  }
  __device__ __forceinline__ u32 make(u64 i, u64 t, u32 c) const {
    const int b = (int) (i & 63);
    const u64 below = b == 63 ? ~0ull : ((2ull << b) - 1ull);
    const u64 z = ~t & below;
    return g.offset + (z ? ((u32) i & ~63u) + (63u - (u32) __clzll((long long) z)) : c);
  }
};

template <bool FULL, bool DIG, typename K, typename V, typename VG = ReadValues>
__device__ __forceinline__ void rs_scatter_tile(
    const K *__restrict__ keys_in, const V *__restrict__ vals_in,
    K *__restrict__ keys_out, V *__restrict__ vals_out, const u32 valid, int shift,
    u32 mask, u32 gbase, u8 *__restrict__ dig_out, int next_shift, u32 next_mask,
    K *s_key, V *s_val, u16 *s_cnt_generic /* [RS_WAVES][RADIX] */, u32 *s_obase,
    const VG vg = VG(), u64 first = 0 /* index of the tile's first pair */) {
  // volatile: lanes read counters that other lanes of the wave have updated
  lds_vu16 *s_cnt = (lds_vu16 *) s_cnt_generic;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  u32 *s_scan = reinterpret_cast<u32 *>(s_key);
  K key[RS_ITEMS];
  V val[RS_ITEMS];
  u32 rk[RS_ITEMS];   // rank inside the wave's stream << 8 | digit
  u64 vg_t = 0;
  u32 vg_c = 0;
  if (VG::active) vg.prefetch(first + (u64) w * RS_WAVE_CHUNK, lane, vg_t, vg_c);
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 e = (u32) w * RS_WAVE_CHUNK + (u32) j * 64 + lane;
    // (the lane reads OUTSIDE the branch: in a part-filled tile lane j, which holds the
    // word of item j, may itself have no entry of that item -- read from a lane that
    // sits out the branch, the word came back as 0 and the entries of a group that
    // reaches into the table's last word got themselves as their heads)
    const u64 gen_t = VG::active ? __shfl(vg_t, j, 64) : 0ull;
    const u32 gen_c = VG::active ? __shfl(vg_c, j, 64) : 0u;
    if (FULL || e < valid) {
      key[j] = keys_in[e];
      val[j] = VG::active ? (V) vg.make(first + e, gen_t, gen_c) : vals_in[e];
    } else {
      key[j] = (K) ~(K) 0;
      val[j] = 0;
    }
  }
  __syncthreads();   // counters are zero
  lds_vu16 *cnt_w = s_cnt + w * RADIX;
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 e = (u32) w * RS_WAVE_CHUNK + (u32) j * 64 + lane;
    // out-of-range pairs go to the last digit; being the last pairs of the
    // tile they end up behind all valid ones and are never written
    const u32 d = (FULL || e < valid) ? ((u32) (key[j] >> shift) & mask) : (RADIX - 1);
    u32 mlo = ~0u, mhi = ~0u;   // lanes holding the same digit
#pragma unroll
    for (int b = 0; b < 8; b++) {
      u32 sx = (u32) ((int) (d << (31 - b)) >> 31);   // bit b of d, spread
      asm volatile("" : "+v"(sx));   // compare THIS register, not a shifted copy of d
      const u64 bal = __ballot(sx != 0);
      // m & ~(bal ^ sx) as ONE three-input bit operation (truth table 0x90)
      mlo = __builtin_amdgcn_bitop3_b32(mlo, (u32) bal, sx, 0x90);
      mhi = __builtin_amdgcn_bitop3_b32(mhi, (u32) (bal >> 32), sx, 0x90);
    }
    const u32 intra = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
    const u32 old = cnt_w[d];
    if (intra == 0) cnt_w[d] = (u16) (old + (u32) __popc(mlo) + (u32) __popc(mhi));
    rk[j] = ((old + intra) << 8) | d;
  }
  __syncthreads();
  // digit totals over the waves; every (wave, digit) counter becomes the
  // tile-local position where that wave's pairs of that digit start
  // (threads 0..255 own one digit each; the scan needs all threads)
  {
    u32 c[RS_WAVES];
    u32 tot = 0;
    if (tid < RADIX) {
#pragma unroll
      for (int i = 0; i < RS_WAVES; i++) {
        c[i] = s_cnt[i * RADIX + tid];
        tot += c[i];
      }
    }
    u32 blocktot;
    u32 dbase = block_scan_excl<SCAN_SUM, RS_THREADS>(tot, &blocktot, s_scan);
    if (tid < RADIX) {
      s_obase[tid] = gbase - dbase;
#pragma unroll
      for (int i = 0; i < RS_WAVES; i++) {
        s_cnt[i * RADIX + tid] = (u16) dbase;
        dbase += c[i];
      }
    }
  }
  __syncthreads();   // also: the scan scratch inside s_key is free again
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 d = rk[j] & 255u;
    const u32 pos = (u32) cnt_w[d] + (rk[j] >> 8);
    s_key[pos] = key[j];
    s_val[pos] = val[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 e = (u32) j * RS_THREADS + tid;
    if (FULL || e < valid) {
      const K k = s_key[e];
      const u32 d = (u32) (k >> shift) & mask;
      const u32 g = s_obase[d] + e;
      keys_out[g] = k;
      vals_out[g] = s_val[e];
      if (DIG) dig_out[g] = (u8) ((u32) (k >> next_shift) & next_mask);
    }
  }
}

template <typename K, typename V, int XCD>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(
    const K *__restrict__ keys_in, const V *__restrict__ vals_in,
    K *__restrict__ keys_out, V *__restrict__ vals_out, u32 last_valid, int shift,
    u32 mask, const u32 *__restrict__ hist_scanned, u32 ntiles,
    u8 *__restrict__ dig_out, int next_shift, u32 next_mask) {
  // last_valid: pairs of the last tile, from the host (not min(n - tile_base,
  // 4096) in here: see the note at k_rs_scatter_gen)
  // 53 KB of LDS in all, so that three workgroups (24 waves) share a CU:
  // 16-bit counters (a tile has 4096 pairs) and the scan scratch laid over
  // the key staging area, which is not written before the scan is done
  __shared__ K s_key[RS_TILE];
  __shared__ V s_val[RS_TILE];
  __shared__ u16 s_cnt[RS_WAVES * RADIX];  // running counters, then staging base
  __shared__ u32 s_obase[RADIX];           // global base minus local start
  static_assert(sizeof(K) * RS_TILE + sizeof(V) * RS_TILE + sizeof(u16) * RS_WAVES * RADIX
                + sizeof(u32) * RADIX <= 53 * 1024 || sizeof(K) + sizeof(V) > 12,
                "three workgroups per CU");

  const int tid = threadIdx.x;
  const u32 tile = XCD == 2 ? os_tile(blockIdx.x)
                            : (XCD ? xcd_tile(blockIdx.x, ntiles) : blockIdx.x);
  if (tile >= ntiles) return;   // whole block leaves together
  const u64 tile_base = (u64) tile * RS_TILE;
  const u32 valid = tile + 1u == ntiles ? last_valid : (u32) RS_TILE;
  for (int i = tid; i < RS_WAVES * RADIX / 2; i += RS_THREADS)
    reinterpret_cast<u32 *>(s_cnt)[i] = 0;
  // this tile's global write base per digit: strided, latency-bound load,
  // issued first so that it is back long before it is needed
  u32 gbase = 0;
  if (tid < RADIX) gbase = hist_scanned[(u64) tile * RADIX + tid];
  // (the digit-byte side output is an experiment switch, off by default: it
  // takes the checked variant so that the two hot variants stay branch-free)
  if (dig_out != nullptr)
    rs_scatter_tile<false, true, K, V>(keys_in + tile_base, vals_in + tile_base, keys_out,
                                       vals_out, valid, shift, mask, gbase, dig_out,
                                       next_shift, next_mask, s_key, s_val, s_cnt, s_obase);
  else if (valid == (u32) RS_TILE)
    rs_scatter_tile<true, false, K, V>(keys_in + tile_base, vals_in + tile_base, keys_out,
                                       vals_out, valid, shift, mask, gbase, dig_out,
                                       next_shift, next_mask, s_key, s_val, s_cnt, s_obase);
  else
    rs_scatter_tile<false, false, K, V>(keys_in + tile_base, vals_in + tile_base, keys_out,
                                        vals_out, valid, shift, mask, gbase, dig_out,
                                        next_shift, next_mask, s_key, s_val, s_cnt, s_obase);
}

// ---------------------------------------------------------------------------
// chained-scan scatter ("onesweep"): no per-tile histogram pass and no scan.
// Every tile publishes its digit counts in a status word per digit
// (flag | value in ONE 32-bit word, so no fence is needed), looks back over the
// status words of the preceding tiles until it meets an inclusive prefix, and
// publishes its own inclusive prefix.  Tiles are dealt to workgroups in a
// chunk-permuted order (workgroups b, b+8, ... share an XCD and take
// OS_CHUNK consecutive tiles at a time) so that neighbouring output runs are
// still completed inside one L2.  Every spin is bounded; a pass that times out
// (*errflag) is redone by the caller with the histogram/scan/scatter path.
// ---------------------------------------------------------------------------
// digit totals of all passes in one read of the keys
template <typename K>
__global__ __launch_bounds__(256) void k_os_totals(
    const K *__restrict__ keys, u64 n, const int *__restrict__ shifts,
    const int *__restrict__ widths, int npasses, u32 *__restrict__ totals) {
  __shared__ u32 h[OS_MAXPASS][RADIX];
  for (int i = threadIdx.x; i < OS_MAXPASS * RADIX; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  int sh[OS_MAXPASS];
  u32 mk[OS_MAXPASS];
#pragma unroll
  for (int p = 0; p < OS_MAXPASS; p++) {
    sh[p] = p < npasses ? shifts[p] : 0;
    mk[p] = p < npasses ? (1u << widths[p]) - 1u : 0u;
  }
  for (u64 i = (u64) blockIdx.x * 256 + threadIdx.x; i < n; i += (u64) gridDim.x * 256) {
    const K k = keys[i];
#pragma unroll
    for (int p = 0; p < OS_MAXPASS; p++)
      if (p < npasses) atomicAdd(&h[p][(u32) (k >> sh[p]) & mk[p]], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < npasses * RADIX; i += 256)
    if ((&h[0][0])[i]) atomicAdd(&totals[i], (&h[0][0])[i]);
}

// per pass: exclusive scan of the 256 digit totals (one block of 256 threads)
__global__ __launch_bounds__(256) void k_os_bases(const u32 *__restrict__ totals,
                                                  u32 *__restrict__ bases) {
  __shared__ u32 s_scan[4];
  const u32 v = totals[blockIdx.x * RADIX + threadIdx.x];
  u32 tot;
  bases[blockIdx.x * RADIX + threadIdx.x] = block_scan_excl<SCAN_SUM>(v, &tot, s_scan);
}

template <typename K, typename V>
__global__ __launch_bounds__(RS_THREADS) void k_os_scatter(
    const K *__restrict__ keys_in, const V *__restrict__ vals_in,
    K *__restrict__ keys_out, V *__restrict__ vals_out, u64 n, int shift,
    u32 mask, const u32 *__restrict__ digit_base, u32 *__restrict__ status,
    u32 ntiles, u32 *__restrict__ errflag) {
  __shared__ K s_key[RS_TILE];
  __shared__ V s_val[RS_TILE];
  __shared__ u32 s_cnt[RS_WAVES][RADIX];
  __shared__ u32 s_dbase[RADIX];
  __shared__ u32 s_obase[RADIX];
  __shared__ u32 s_tot[RADIX];
  __shared__ u32 s_scan[RS_WAVES];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile = os_tile(blockIdx.x);
  if (tile >= ntiles) return;   // whole block leaves together
  const u64 tile_base = (u64) tile * RS_TILE;
  const u32 valid = (u32) ((n - tile_base) < (u64) RS_TILE ? (n - tile_base)
                                                            : (u64) RS_TILE);
  for (int i = tid; i < RS_WAVES * RADIX; i += RS_THREADS)
    (&s_cnt[0][0])[i] = 0;
  u32 gbase = 0;
  if (tid < RADIX) gbase = digit_base[tid];

  K key[RS_ITEMS];
  V val[RS_ITEMS];
  u32 rk[RS_ITEMS];
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 e = (u32) w * RS_WAVE_CHUNK + (u32) j * 64 + lane;
    if (e < valid) {
      key[j] = keys_in[tile_base + e];
      val[j] = vals_in[tile_base + e];
    } else {
      key[j] = (K) ~(K) 0;
      val[j] = 0;
    }
  }
  __syncthreads();
  const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 e = (u32) w * RS_WAVE_CHUNK + (u32) j * 64 + lane;
    const u32 d = e < valid ? ((u32) (key[j] >> shift) & mask) : (RADIX - 1);
    const u64 m = match_digit(d);
    const u32 intra = (u32) __popcll(m & lt);
    const int leader = __ffsll((unsigned long long) m) - 1;
    u32 old = 0;
    if (lane == leader) {
      old = s_cnt[w][d];
      s_cnt[w][d] = old + (u32) __popcll(m);
    }
    old = __shfl(old, leader, 64);
    rk[j] = ((old + intra) << 8) | d;
  }
  __syncthreads();
  {
    u32 tot = 0;
    if (tid < RADIX) {
#pragma unroll
      for (int i = 0; i < RS_WAVES; i++) {
        const u32 c = s_cnt[i][tid];
        s_cnt[i][tid] = tot;
        tot += c;
      }
      // the padding of a short last tile was counted under the last digit
      if (tid == RADIX - 1) tot -= (u32) RS_TILE - valid;
    }
    u32 blocktot;
    u32 padded = tot + ((tid == RADIX - 1) ? (u32) RS_TILE - valid : 0u);
    u32 dbase = block_scan_excl<SCAN_SUM, RS_THREADS>(tid < RADIX ? padded : 0u,
                                                      &blocktot, s_scan);
    if (tid < RADIX) {
      s_tot[tid] = tot;
      s_dbase[tid] = dbase;
      s_obase[tid] = gbase;
    }
  }
  __syncthreads();
  // publish this tile's digit counts as early as possible: two digits per
  // 8-byte store (each 32-bit half is a complete flag|value word)
  u64 *status64 = reinterpret_cast<u64 *>(status);
  if (tid < RADIX / 2) {
    const u32 fl = tile == 0 ? OS_INCL : OS_AGG;
    const u64 pack = (u64) (fl | s_tot[2 * tid]) | ((u64) (fl | s_tot[2 * tid + 1]) << 32);
    __hip_atomic_store(status64 + (u64) tile * (RADIX / 2) + tid, pack,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // stage the tile in LDS in digit order: needs tile-local offsets only, and
  // gives the preceding tiles time to publish theirs
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 d = rk[j] & 255u;
    const u32 pos = s_dbase[d] + s_cnt[w][d] + (rk[j] >> 8);
    s_key[pos] = key[j];
    s_val[pos] = val[j];
  }
  if (tid < RADIX / 2) {
    // ---- chained scan over the tiles, one thread per digit pair
    u32 excl0 = 0, excl1 = 0;
    if (tile > 0) {
      // walk back OS_WIN tiles at a time: the loads of one window are
      // independent and in flight together, so a hop costs a fraction of a
      // memory round trip
      u32 t = tile, spins = 0, hops = 0;
      bool done = false;
      while (!done) {
        u64 v[OS_WIN];
#pragma unroll
        for (int k = 0; k < OS_WIN; k++) {
          const u32 tt = t > (u32) k ? t - 1 - k : 0u;
          v[k] = __hip_atomic_load(status64 + (u64) tt * (RADIX / 2) + tid,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = 0; k < OS_WIN; k++) {
          if (done) break;
          const u32 lo = (u32) v[k], hi = (u32) (v[k] >> 32);
          const u32 f = lo >> 30;   // both halves always carry the same flag
          if (f == 0) {             // not published yet: re-read from here
            if (++spins > OS_SPIN_LIMIT) { *errflag = 1; done = true; }
            __builtin_amdgcn_s_sleep(1);
            break;
          }
          excl0 += lo & OS_VAL;
          excl1 += hi & OS_VAL;
          t--;
          hops++;
          if (f == 2) done = true;  // tile 0 always publishes an inclusive prefix
        }
      }
      if (tid == 0 && errflag[1] != 0) atomicAdd(&errflag[2], hops);
      const u64 pack = (u64) (OS_INCL | (excl0 + s_tot[2 * tid])) |
                       ((u64) (OS_INCL | (excl1 + s_tot[2 * tid + 1])) << 32);
      __hip_atomic_store(status64 + (u64) tile * (RADIX / 2) + tid, pack,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_obase[2 * tid] += excl0 - s_dbase[2 * tid];
    s_obase[2 * tid + 1] += excl1 - s_dbase[2 * tid + 1];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 e = (u32) j * RS_THREADS + tid;
    if (e < valid) {
      const K k = s_key[e];
      const u32 d = (u32) (k >> shift) & mask;
      const u32 g = s_obase[d] + e;
      keys_out[g] = k;
      vals_out[g] = s_val[e];
    }
  }
}

// ---------------------------------------------------------------------------
// Partition of one tile WITHOUT keeping the input order inside a digit: the
// place of a pair is an LDS atomic on its digit's counter -- a dozen
// instructions per pair instead of the ~125 of the ballot ranking above, which
// the stable passes of a sort are bound by.  For the partition passes of the
// rank-table build: their keys are positions (a permutation: uniform digits,
// hardly two lanes on one counter) and the scatter into the window behind them
// does not care about the order.  Histogram and column scan are the sort's.
// ---------------------------------------------------------------------------
template <bool FULL, typename K, typename V, typename VG>
__device__ __forceinline__ void rs_partition_tile(
    const K *__restrict__ keys_in, const V *__restrict__ vals_in,
    K *__restrict__ keys_out, V *__restrict__ vals_out, const u32 valid, int shift,
    u32 mask, u32 gbase, K *s_key, V *s_val, u32 *s_cnt /* [RADIX], zeroed */,
    u32 *s_obase, u32 *s_scan, const VG vg, u64 first) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  K key[RS_ITEMS];
  V val[RS_ITEMS];
  u32 rk[RS_ITEMS];   // place inside the digit's run << 8 | digit
  u64 vg_t = 0;
  u32 vg_c = 0;
  if (VG::active) vg.prefetch(first + (u64) w * RS_WAVE_CHUNK, lane, vg_t, vg_c);
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 e = (u32) w * RS_WAVE_CHUNK + (u32) j * 64 + lane;
    // (the lane reads OUTSIDE the branch: in a part-filled tile lane j, which holds the
    // word of item j, may itself have no entry of that item -- read from a lane that
    // sits out the branch, the word came back as 0 and the entries of a group that
    // reaches into the table's last word got themselves as their heads)
    const u64 gen_t = VG::active ? __shfl(vg_t, j, 64) : 0ull;
    const u32 gen_c = VG::active ? __shfl(vg_c, j, 64) : 0u;
    if (FULL || e < valid) {
      key[j] = keys_in[e];
      val[j] = VG::active ? (V) vg.make(first + e, gen_t, gen_c) : vals_in[e];
    } else {
      key[j] = 0;
      val[j] = 0;
    }
  }
  __syncthreads();   // counters are zero
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 e = (u32) w * RS_WAVE_CHUNK + (u32) j * 64 + lane;
    if (FULL || e < valid) {
      const u32 d = (u32) (key[j] >> shift) & mask;
      rk[j] = (atomicAdd(&s_cnt[d], 1u) << 8) | d;
    }
  }
  __syncthreads();
  {
    const u32 tot = tid < RADIX ? s_cnt[tid] : 0u;
    u32 blocktot;
    const u32 dbase = block_scan_excl<SCAN_SUM, RS_THREADS>(tot, &blocktot, s_scan);
    if (tid < RADIX) {
      s_obase[tid] = gbase - dbase;
      s_cnt[tid] = dbase;       // where the digit's run starts in the staging area
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 e = (u32) w * RS_WAVE_CHUNK + (u32) j * 64 + lane;
    if (FULL || e < valid) {
      const u32 pos = s_cnt[rk[j] & 255u] + (rk[j] >> 8);
      s_key[pos] = key[j];
      s_val[pos] = val[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const u32 e = (u32) j * RS_THREADS + tid;
    if (FULL || e < valid) {
      const K k = s_key[e];
      const u32 g = s_obase[(u32) (k >> shift) & mask] + e;
      keys_out[g] = k;
      vals_out[g] = s_val[e];
    }
  }
}

// one unstable partition pass (XCD-aware tile order), values read (VG =
// ReadValues) or generated
template <typename K, typename V, typename VG>
__global__ __launch_bounds__(RS_THREADS) void k_rs_partition(
    const K *__restrict__ keys_in, const V *__restrict__ vals_in, const VG vg,
    K *__restrict__ keys_out, V *__restrict__ vals_out, u32 last_valid, int shift, u32 mask,
    const u32 *__restrict__ hist_scanned, u32 ntiles) {
  __shared__ K s_key[RS_TILE];
  __shared__ V s_val[RS_TILE];
  __shared__ u32 s_cnt[RADIX];
  __shared__ u32 s_obase[RADIX];
  __shared__ u32 s_scan[RS_WAVES];
  const int tid = threadIdx.x;
  const u32 tile = xcd_tile(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  const u64 tile_base = (u64) tile * RS_TILE;
  const bool last = tile + 1u == ntiles;
  if (tid < RADIX) s_cnt[tid] = 0;
  u32 gbase = 0;
  if (tid < RADIX) gbase = hist_scanned[(u64) tile * RADIX + tid];
  const V *vin = VG::active ? nullptr : vals_in + tile_base;
  if (!last || last_valid == (u32) RS_TILE)
    rs_partition_tile<true, K, V, VG>(keys_in + tile_base, vin, keys_out, vals_out, (u32) RS_TILE,
                                      shift, mask, gbase, s_key, s_val, s_cnt, s_obase, s_scan,
                                      vg, tile_base);
  else
    rs_partition_tile<false, K, V, VG>(keys_in + tile_base, vin, keys_out, vals_out, last_valid,
                                       shift, mask, gbase, s_key, s_val, s_cnt, s_obase, s_scan,
                                       vg, tile_base);
}

}  // namespace

// chunk sums of the column scan (one 256-word row per CS_ROWS tiles)
static u64 cs_workspace_words(u64 ntiles) {
  return (ntiles / CS_ROWS + 2) * RADIX;
}

u64 radix_workspace_words(u64 n) {
  u64 ntiles = div_up(n, RS_TILE);
  u64 hist = ntiles * RADIX;
  return hist + cs_workspace_words(ntiles) + 64 + OS_AUX_WORDS;
}

// tuning switch (A/B measurements): GTAMD_XCD_REMAP=0 disables the remap
static bool g_xcd_remap = true;
static int g_xcd_mode = 1;
// digit side arrays: opt-in (GTAMD_DIGBYTES=1).  Measured at 3 Gbp: the
// histogram passes get 17 ms cheaper, but the byte stores cost the scatter
// kernel 2 ms per pass, net -4.6 ms (1 %): not worth 2 B/pair of memory.
static bool g_no_digbytes = true;
// chained-scan scatter: opt-in (GTAMD_ONESWEEP=1).  Measured at 3 Gbp: the
// look-back walks 39 tiles on average (status hop latency across XCDs x tile
// rate), which eats most of what the saved histogram pass gives back: sort
// 151.7 ms vs 158.7 ms.  Correct, but not worth a spin loop by default.
static bool g_onesweep = false;

template <typename K, typename V>
int radix_sort_pairs(K *keys_a, V *vals_a, K *keys_b, V *vals_b, u64 n,
                     const int *shifts, const int *widths, int npasses,
                     u32 *ws, hipStream_t st, hipEvent_t *ev_pairs,
                     int *n_ev, u8 *dig_a, u8 *dig_b) {
  if (n == 0) return 0;
  {
    const char *e = getenv("GTAMD_XCD_REMAP");
    g_xcd_remap = !(e != nullptr && e[0] == '0');
    g_xcd_mode = (e != nullptr && e[0] == '2') ? 2 : 1;
    const char *db = getenv("GTAMD_DIGBYTES");
    g_no_digbytes = !(db != nullptr && db[0] == '1');
    const char *o = getenv("GTAMD_ONESWEEP");
    g_onesweep = o != nullptr && o[0] == '1';
    if (g_onesweep) g_no_digbytes = true;   // the chained kernel writes no digit bytes
  }
  if (n >= (1ull << 32)) {
    gtamd_set_error("radix_sort_pairs: %llu pairs exceed the 32-bit index "
                    "range of one sort", (unsigned long long) n);
    return -1;
  }
  const u32 ntiles = (u32) div_up(n, RS_TILE);
  const u32 last_valid = (u32) (n - (u64) (ntiles - 1) * RS_TILE);
  u32 *hist = ws;
  u32 *scanws = ws + (u64) ntiles * RADIX;
  K *kin = keys_a, *kout = keys_b;
  V *vin = vals_a, *vout = vals_b;
  // ---- chained-scan path for big sorts
  bool chained[OS_MAXPASS] = {false};
  u32 *aux = scanws + cs_workspace_words(ntiles) + 32;
  u32 *d_totals = aux, *d_bases = aux + OS_MAXPASS * RADIX,
      *d_flag = aux + 2 * OS_MAXPASS * RADIX;
  int *d_shifts = reinterpret_cast<int *>(d_flag + 8), *d_widths = d_shifts + OS_MAXPASS;
  if (g_onesweep && n >= (1u << 20) && npasses <= OS_MAXPASS) {
    HIP_TRY(hipMemsetAsync(d_totals, 0, OS_MAXPASS * RADIX * 4, st));
    HIP_TRY(hipMemcpyAsync(d_shifts, shifts, npasses * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_widths, widths, npasses * sizeof(int), hipMemcpyHostToDevice, st));
    k_os_totals<K><<<2048, 256, 0, st>>>(kin, n, d_shifts, d_widths, npasses, d_totals);
    HIP_TRY(hipGetLastError());
    k_os_bases<<<npasses, 256, 0, st>>>(d_totals, d_bases);
    HIP_TRY(hipGetLastError());
    std::vector<u32> h_tot((size_t) npasses * RADIX);
    HIP_TRY(hipMemcpyAsync(h_tot.data(), d_totals, h_tot.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int p = 0; p < npasses; p++) {
      u32 mx = 0;
      for (int d = 0; d < RADIX; d++) mx = h_tot[(size_t) p * RADIX + d] > mx ? h_tot[(size_t) p * RADIX + d] : mx;
      chained[p] = mx < (1u << 30);   // status words carry 30-bit prefixes
    }
  }
  for (int p = 0; p < npasses; p++) {
    const u32 mask = (1u << widths[p]) - 1u;
    if (chained[p]) {
      HIP_TRY(hipMemsetAsync(hist, 0, (u64) ntiles * RADIX * 4, st));
      HIP_TRY(hipMemsetAsync(d_flag, 0, 12, st));
      if (getenv("GTAMD_OS_STATS") != nullptr) {
        const u32 one = 1;
        HIP_TRY(hipMemcpyAsync(d_flag + 1, &one, 4, hipMemcpyHostToDevice, st));
      }
      if (ev_pairs != nullptr) HIP_TRY(hipEventRecord(ev_pairs[2 * *n_ev], st));
      const u32 groups = (ntiles + 8u * OS_CHUNK - 1u) / (8u * OS_CHUNK);
      k_os_scatter<K, V><<<groups * 8u * OS_CHUNK, RS_THREADS, 0, st>>>(
          kin, vin, kout, vout, n, shifts[p], mask, d_bases + p * RADIX, hist,
          ntiles, d_flag);
      HIP_TRY(hipGetLastError());
      if (ev_pairs != nullptr) HIP_TRY(hipEventRecord(ev_pairs[2 * *n_ev + 1], st));
      u32 h_flags[3] = {0, 0, 0};
      HIP_TRY(hipMemcpyAsync(h_flags, d_flag, 12, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      const u32 h_flag = h_flags[0];
      if (h_flags[1])
        fprintf(stderr, "gtamd: chained scan pass %d: %.2f look-back hops per tile\n", p,
                (double) h_flags[2] / ntiles);
      if (h_flag == 0) {
        if (ev_pairs != nullptr) (*n_ev)++;
        K *tk = kin; kin = kout; kout = tk;
        V *tv = vin; vin = vout; vout = tv;
        continue;
      }
      // a look-back timed out (the dispatch order assumption did not hold):
      // the input of this pass is untouched, redo it the classic way
      fprintf(stderr, "gtamd: chained scan timed out in pass %d, falling back\n", p);
    }
    // digit bytes: written by the previous pass into din, this pass leaves the
    // next pass's digits in dout
    const bool have_dig = dig_a != nullptr && p > 0 && !g_no_digbytes;
    u8 *din = (p & 1) ? dig_a : dig_b, *dout = (p & 1) ? dig_b : dig_a;
    if (dig_a == nullptr || p + 1 >= npasses || g_no_digbytes) dout = nullptr;
    const int nsh = p + 1 < npasses ? shifts[p + 1] : 0;
    const u32 nmk = p + 1 < npasses ? (1u << widths[p + 1]) - 1u : 0u;
    if (have_dig)
      k_rs_hist_bytes<<<ntiles, RS_THREADS, 0, st>>>(din, n, hist, ntiles);
    else
      k_rs_hist<K><<<ntiles, RS_THREADS, 0, st>>>(kin, n, shifts[p], mask, hist,
                                               ntiles);
    HIP_TRY(hipGetLastError());
    {
      const u32 nchunks = (ntiles + CS_ROWS - 1) / CS_ROWS;
      k_cs_chunksum<<<nchunks, RADIX, 0, st>>>(hist, ntiles, scanws);
      HIP_TRY(hipGetLastError());
      k_cs_chunkbase<<<1, RADIX * CB_PARTS, 0, st>>>(scanws, nchunks);
      HIP_TRY(hipGetLastError());
      k_cs_rows<<<nchunks, RADIX, 0, st>>>(hist, ntiles, scanws);
      HIP_TRY(hipGetLastError());
    }
    if (ev_pairs != nullptr) HIP_TRY(hipEventRecord(ev_pairs[2 * *n_ev], st));
    if (g_xcd_mode == 2)
      k_rs_scatter<K, V, 2><<<((ntiles + 8u * OS_CHUNK - 1u) / (8u * OS_CHUNK)) * 8u * OS_CHUNK, RS_THREADS, 0, st>>>(
          kin, vin, kout, vout, last_valid, shifts[p], mask, hist, ntiles, dout, nsh, nmk);
    else if (!g_xcd_remap)
      k_rs_scatter<K, V, 0><<<ntiles, RS_THREADS, 0, st>>>(
          kin, vin, kout, vout, last_valid, shifts[p], mask, hist, ntiles, dout, nsh, nmk);
    else
      k_rs_scatter<K, V, 1><<<((ntiles + 7u) >> 3) * 8u, RS_THREADS, 0, st>>>(
          kin, vin, kout, vout, last_valid, shifts[p], mask, hist, ntiles, dout, nsh, nmk);
    HIP_TRY(hipGetLastError());
    if (ev_pairs != nullptr) {
      HIP_TRY(hipEventRecord(ev_pairs[2 * *n_ev + 1], st));
      (*n_ev)++;
    }
    K *tk = kin; kin = kout; kout = tk;
    V *tv = vin; vin = vout; vout = tv;
  }
  return 0;
}

// scatter pass with generated values (XCD-aware tile order).  last_valid: pairs
// of the last tile, from the host -- the in-kernel form  min(n - tile_base, 4096)
// was miscompiled in this kernel (hipcc 7.2: the s_cselect that follows the
// 64-bit compare read a stale SCC, the partial tile ran as a full one).
template <typename K, typename VG>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter_gen(
    const K *__restrict__ keys_in, const VG vg, K *__restrict__ keys_out,
    u32 *__restrict__ vals_out, u32 last_valid, int shift, u32 mask,
    const u32 *__restrict__ hist_scanned, u32 ntiles) {
  __shared__ K s_key[RS_TILE];
  __shared__ u32 s_val[RS_TILE];
  __shared__ u16 s_cnt[RS_WAVES * RADIX];
  __shared__ u32 s_obase[RADIX];
  const int tid = threadIdx.x;
  const u32 tile = xcd_tile(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  const u64 tile_base = (u64) tile * RS_TILE;
  const bool last = tile + 1u == ntiles;
  for (int i = tid; i < RS_WAVES * RADIX / 2; i += RS_THREADS)
    reinterpret_cast<u32 *>(s_cnt)[i] = 0;
  u32 gbase = 0;
  if (tid < RADIX) gbase = hist_scanned[(u64) tile * RADIX + tid];
  if (!last || last_valid == (u32) RS_TILE)
    rs_scatter_tile<true, false, K, u32, VG>(keys_in + tile_base, nullptr, keys_out, vals_out,
                                             (u32) RS_TILE, shift, mask, gbase, nullptr, 0, 0u,
                                             s_key, s_val, s_cnt, s_obase, vg, tile_base);
  else
    rs_scatter_tile<false, false, K, u32, VG>(keys_in + tile_base, nullptr, keys_out, vals_out,
                                              last_valid, shift, mask, gbase, nullptr, 0, 0u,
                                              s_key, s_val, s_cnt, s_obase, vg, tile_base);
}

int radix_pass_group_heads(const u32 *keys_a, GroupHeadValues gen, u32 *keys_b,
                           u32 *vals_b, u64 n, int shift, int width, u32 *ws,
                           hipStream_t st) {
  if (n == 0) return 0;
  if (n >= (1ull << 32)) {
    gtamd_set_error("radix_pass_group_heads: %llu pairs exceed the 32-bit index range",
                    (unsigned long long) n);
    return -1;
  }
  const u32 ntiles = (u32) div_up(n, RS_TILE);
  const u32 mask = (1u << width) - 1u;
  const u32 last_valid = (u32) (n - (u64) (ntiles - 1) * RS_TILE);
  k_rs_hist<u32><<<ntiles, RS_THREADS, 0, st>>>(keys_a, n, shift, mask, ws, ntiles);
  HIP_TRY(hipGetLastError());
  if (radix_scan_tile_hist(ws, n, st) != 0) return -1;
  MakeGroupHeads mg;
  mg.g = gen;
  const char *e = getenv("GTAMD_STABLE_PARTITION");   // A/B switch: the sort's own kernel
  if (e != nullptr && e[0] == '1')
    k_rs_scatter_gen<u32, MakeGroupHeads><<<((ntiles + 7u) >> 3) * 8u, RS_THREADS, 0, st>>>(
        keys_a, mg, keys_b, vals_b, last_valid, shift, mask, ws, ntiles);
  else
    k_rs_partition<u32, u32, MakeGroupHeads><<<((ntiles + 7u) >> 3) * 8u, RS_THREADS, 0, st>>>(
        keys_a, nullptr, mg, keys_b, vals_b, last_valid, shift, mask, ws, ntiles);
  HIP_TRY(hipGetLastError());
  return 0;
}

int radix_partition_u32(const u32 *keys_a, const u32 *vals_a, u32 *keys_b, u32 *vals_b, u64 n,
                        int shift, int width, u32 *ws, hipStream_t st) {
  if (n == 0) return 0;
  if (n >= (1ull << 32)) {
    gtamd_set_error("radix_partition_u32: %llu pairs exceed the 32-bit index range",
                    (unsigned long long) n);
    return -1;
  }
  const char *e = getenv("GTAMD_STABLE_PARTITION");
  if (e != nullptr && e[0] == '1')
    return radix_sort_pairs<u32, u32>(const_cast<u32 *>(keys_a), const_cast<u32 *>(vals_a), keys_b,
                                      vals_b, n, &shift, &width, 1, ws, st, nullptr, nullptr);
  const u32 ntiles = (u32) div_up(n, RS_TILE);
  const u32 mask = (1u << width) - 1u;
  const u32 last_valid = (u32) (n - (u64) (ntiles - 1) * RS_TILE);
  k_rs_hist<u32><<<ntiles, RS_THREADS, 0, st>>>(keys_a, n, shift, mask, ws, ntiles);
  HIP_TRY(hipGetLastError());
  if (radix_scan_tile_hist(ws, n, st) != 0) return -1;
  k_rs_partition<u32, u32, ReadValues><<<((ntiles + 7u) >> 3) * 8u, RS_THREADS, 0, st>>>(
      keys_a, vals_a, ReadValues(), keys_b, vals_b, last_valid, shift, mask, ws, ntiles);
  HIP_TRY(hipGetLastError());
  return 0;
}

int radix_scan_tile_hist(u32 *ws, u64 n, hipStream_t st) {
  const u32 ntiles = (u32) div_up(n, RS_TILE);
  return radix_scan_tile_rows(ws, ntiles, ws + (u64) ntiles * RADIX, st);
}

u64 radix_rows_workspace_words(u64 nrows) { return cs_workspace_words(nrows) + 64; }

int radix_scan_tile_rows(u32 *hist, u32 ntiles, u32 *scanws, hipStream_t st) {
  if (ntiles == 0) return 0;
  const u32 nchunks = (ntiles + CS_ROWS - 1) / CS_ROWS;
  k_cs_chunksum<<<nchunks, RADIX, 0, st>>>(hist, ntiles, scanws);
  HIP_TRY(hipGetLastError());
  k_cs_chunkbase<<<1, RADIX * CB_PARTS, 0, st>>>(scanws, nchunks);
  HIP_TRY(hipGetLastError());
  k_cs_rows<<<nchunks, RADIX, 0, st>>>(hist, ntiles, scanws);
  HIP_TRY(hipGetLastError());
  return 0;
}

template int radix_sort_pairs<u64, u32>(u64 *, u32 *, u64 *, u32 *, u64,
                                        const int *, const int *, int, u32 *,
                                        hipStream_t, hipEvent_t *, int *, u8 *,
                                        u8 *);
template int radix_sort_pairs<u32, u32>(u32 *, u32 *, u32 *, u32 *, u64,
                                        const int *, const int *, int, u32 *,
                                        hipStream_t, hipEvent_t *, int *, u8 *,
                                        u8 *);
template int radix_sort_pairs<u32, u64>(u32 *, u64 *, u32 *, u64 *, u64,
                                        const int *, const int *, int, u32 *,
                                        hipStream_t, hipEvent_t *, int *, u8 *,
                                        u8 *);
template int radix_sort_pairs<u64, u64>(u64 *, u64 *, u64 *, u64 *, u64,
                                        const int *, const int *, int, u32 *,
                                        hipStream_t, hipEvent_t *, int *, u8 *,
                                        u8 *);
