// Dev microbenchmark (not part of the product): level A of the MSD first sort
// (k_msd_scatter_a of csrc/esa_msd.h, DNA) on a random 2-bit text, its phases cut
// apart, and variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/levela tools/microbench/levela.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64; typedef uint8_t u8; typedef uint16_t u16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ inline u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct Text { const u64 *tb; const u64 *sp; u64 n, nw_tb, nw_sp; };
__device__ __forceinline__ u64 tb_word(const Text &t, u64 w) { return w < t.nw_tb ? t.tb[w] : 0ull; }
__device__ __forceinline__ u64 sp_word(const Text &t, u64 w) { return w < t.nw_sp ? t.sp[w] : 0ull; }
constexpr int KP_PER = 8, SYMS = 20, LOW_BITS = 24, DSHIFT = 19;
constexpr u32 P_WILD = 5, P_SEP = 6, P_UNDEF = 7;
constexpr int MS_TILE = 4096, MS_THREADS = 512, MS_WAVES = 8, MS_ITEMS = 8, MS_WCHUNK = 512, MS_PAD = 520;

__device__ __forceinline__ void dna_keys8(const Text &t, u64 p0, u64 (&key)[KP_PER]) {
  const u64 w = p0 >> 5;
  const int o = (int) (p0 & 31) * 2;
  const u64 hi = tb_word(t, w), lo = tb_word(t, w + 1);
  const u64 a_hi = o ? (hi << o) | (lo >> (64 - o)) : hi;
  const u64 a_lo = lo << o;
  const u64 sw = p0 >> 6;
  const int so = (int) (p0 & 63);
  const u64 s0 = sp_word(t, sw), s1 = sp_word(t, sw + 1);
  const u64 S = so ? (s0 >> so) | (s1 << (64 - so)) : s0;
  u32 pay;
  if (p0 == 0) pay = P_UNDEF;
  else {
    const u32 c = o ? (u32) (hi >> (64 - o)) & 3u : (u32) tb_word(t, w - 1) & 3u;
    const bool sp = c < 2u && (so ? (s0 >> (so - 1)) & 1ull : sp_word(t, sw - 1) >> 63);
    pay = sp ? ((c & 1u) ? P_SEP : P_WILD) : c;
  }
  if ((S & ((1ull << (KP_PER + SYMS - 1)) - 1ull)) == 0) {
#pragma unroll
    for (int g = 0; g < KP_PER; g++) {
      const u64 win = g ? (a_hi << (2 * g)) | (a_lo >> (64 - 2 * g)) : a_hi;
      key[g] = (win & (~0ull << LOW_BITS)) | pay;
      pay = (u32) (win >> 62);
    }
    return;
  }
#pragma unroll
  for (int g = 0; g < KP_PER; g++) {
    const u64 win = g ? (a_hi << (2 * g)) | (a_lo >> (64 - 2 * g)) : a_hi;
    const u64 s = (S >> g) & ((1ull << SYMS) - 1ull);
    const int d = s ? __ffsll((unsigned long long) s) - 1 : SYMS;
    if (d == 0) key[g] = (~0ull << DSHIFT) | pay;
    else {
      u64 pre = win >> LOW_BITS;
      u32 dc = 0;
      if (d < SYMS) { pre |= (1ull << (2 * (SYMS - d))) - 1ull; dc = (u32) (SYMS - d); }
      key[g] = (pre << LOW_BITS) | ((u64) dc << DSHIFT) | pay;
    }
    const u32 c = (u32) (win >> 62);
    pay = (c < 2u && (s & 1ull)) ? ((c & 1u) ? P_SEP : P_WILD) : c;
  }
}
// one suffix (variant without transposition)
__device__ __forceinline__ u64 dna_key1(const Text &t, u64 p) {
  const u64 w = p >> 5;
  const int o = (int) (p & 31) * 2;
  const u64 hi = tb_word(t, w), lo = tb_word(t, w + 1);
  const u64 win = o ? (hi << o) | (lo >> (64 - o)) : hi;
  const u64 sw = p >> 6;
  const int so = (int) (p & 63);
  const u64 s0 = sp_word(t, sw), s1 = sp_word(t, sw + 1);
  const u64 S = so ? (s0 >> so) | (s1 << (64 - so)) : s0;
  u32 pay;
  if (p == 0) pay = P_UNDEF;
  else {
    const u32 c = o ? (u32) (hi >> (64 - o)) & 3u : (u32) tb_word(t, w - 1) & 3u;
    const bool sp = c < 2u && (so ? (s0 >> (so - 1)) & 1ull : sp_word(t, sw - 1) >> 63);
    pay = sp ? ((c & 1u) ? P_SEP : P_WILD) : c;
  }
  const u64 s = S & ((1ull << SYMS) - 1ull);
  if (s == 0) return (win & (~0ull << LOW_BITS)) | pay;
  const int d = __ffsll((unsigned long long) s) - 1;
  if (d == 0) return (~0ull << DSHIFT) | pay;
  u64 pre = (win >> LOW_BITS) | ((1ull << (2 * (SYMS - d))) - 1ull);
  return (pre << LOW_BITS) | ((u64) (SYMS - d) << DSHIFT) | pay;
}

template <int NB>
__device__ __forceinline__ void ms_match(u32 d, u32 &intra, u32 &group) {
  u32 mlo = ~0u, mhi = ~0u;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    u32 sx = (u32) ((int) (d << (31 - b)) >> 31);
    asm volatile("" : "+v"(sx));
    const u64 bal = __ballot(sx != 0);
    mlo = __builtin_amdgcn_bitop3_b32(mlo, (u32) bal, sx, 0x90);
    mhi = __builtin_amdgcn_bitop3_b32(mhi, (u32) (bal >> 32), sx, 0x90);
  }
  intra = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
  group = (u32) __popc(mlo) + (u32) __popc(mhi);
}
typedef __attribute__((address_space(3))) volatile u16 ms_vu16;
__device__ __forceinline__ u32 wave_scan_incl(u32 v) {
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);
  return v;
}
template <int THREADS>
__device__ __forceinline__ u32 block_scan_excl(u32 v, u32 *total, u32 *lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u32 inc = wave_scan_incl(v);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  u32 carry = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < THREADS / 64; i++) { u32 s = lds[i]; if (i < w) carry += s; tot += s; }
  __syncthreads();
  *total = tot;
  return carry + (u32) __builtin_amdgcn_update_dpp(0, (int) inc, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 ms_xcd_tile(u32 b, u32 ntiles) {
  const u32 per = (ntiles + 7u) >> 3;
  return (b & 7u) * per + (b >> 3);
}

__global__ __launch_bounds__(MS_THREADS) void k_hist_a(Text t, u64 N, u32 *__restrict__ hist) {
  __shared__ u32 h[MS_WAVES][256];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < MS_WAVES * 256; i += MS_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const u64 p0 = (u64) blockIdx.x * MS_TILE + (u64) tid * KP_PER;
  if (p0 < N) {
    u64 key[KP_PER];
    dna_keys8(t, p0, key);
#pragma unroll
    for (int g = 0; g < KP_PER; g++) atomicAdd(&h[w][(u32) (key[g] >> 56)], 1u);
  }
  __syncthreads();
  if (tid < 256) {
    u32 c = 0;
    for (int i = 0; i < MS_WAVES; i++) c += h[i][tid];
    hist[(u64) blockIdx.x * 256 + tid] = c;
  }
}
// harness only: scanned[tile][d] = start of bin d + entries of digit d in earlier tiles
__global__ void k_colsum(const u32 *hist, u32 ntiles, u32 *tot) {
  const u32 d = threadIdx.x; u32 s = 0;
  for (u32 t = blockIdx.x; t < ntiles; t += gridDim.x) s += hist[(u64) t * 256 + d];
  atomicAdd(&tot[d], s);
}
__global__ void k_colscan(const u32 *hist, u32 ntiles, const u32 *binstart, u32 *scanned) {
  // one block per digit
  const u32 d = blockIdx.x;
  __shared__ u32 s_scan[16];
  __shared__ u32 s_run;
  if (threadIdx.x == 0) s_run = binstart[d];
  __syncthreads();
  for (u32 base = 0; base < ntiles; base += 1024) {
    const u32 t = base + threadIdx.x;
    const u32 v = t < ntiles ? hist[(u64) t * 256 + d] : 0u;
    u32 tot;
    const u32 e = block_scan_excl<1024>(v, &tot, s_scan);
    if (t < ntiles) scanned[(u64) t * 256 + d] = s_run + e;
    __syncthreads();
    if (threadIdx.x == 0) s_run += tot;
    __syncthreads();
  }
}

// PHASES: 1 keygen only, 2 + transposition, 3 + ranking, 4 + offsets, 5 + staging, 6 all
template <int PHASES, bool DIRECT>
__global__ __launch_bounds__(MS_THREADS) void k_scatter_a(
    Text t, u64 N, u32 last_valid, const u32 *__restrict__ scanned, u32 ntiles,
    u32 *__restrict__ k1out, u8 *__restrict__ xout, u32 *__restrict__ pout, u32 *sink) {
  __shared__ u64 s_t[5120];
  __shared__ u16 s_cnt_mem[MS_WAVES * 256];
  __shared__ u32 s_obase[256];
  __shared__ u32 s_scan[MS_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile = ms_xcd_tile(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  const u64 tile_base = (u64) tile * MS_TILE;
  const u32 valid = tile + 1u == ntiles ? last_valid : (u32) MS_TILE;
  for (int i = tid; i < MS_WAVES * 256 / 2; i += MS_THREADS) reinterpret_cast<u32 *>(s_cnt_mem)[i] = 0;
  u32 gbase = 0;
  if (tid < 256) gbase = scanned[(u64) tile * 256 + tid];
  u64 key[MS_ITEMS];
  u32 rk[MS_ITEMS];
  if (DIRECT) {
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) {
      const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
      key[j] = tile_base + e < N ? dna_key1(t, tile_base + e) : ~0ull;
    }
  } else {
    u64 key8[KP_PER];
    const u64 p0 = tile_base + (u64) tid * KP_PER;
    if (p0 < N) dna_keys8(t, p0, key8);
    else for (int g = 0; g < KP_PER; g++) key8[g] = ~0ull;
    if (PHASES == 1) { u64 x = 0; for (int g = 0; g < KP_PER; g++) x ^= key8[g]; if (x == 0x1234567ull) sink[0] = 1; return; }
#pragma unroll
    for (int g = 0; g < KP_PER; g++) s_t[g * MS_PAD + tid] = key8[g];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) {
      const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
      key[j] = s_t[(e & 7u) * MS_PAD + (e >> 3)];
    }
  }
  if (PHASES <= 2) { u64 x = 0; for (int j = 0; j < MS_ITEMS; j++) x ^= key[j]; if (x == 0x1234567ull) sink[0] = 1; return; }
  __syncthreads();
  ms_vu16 *cnt_w = (ms_vu16 *) s_cnt_mem + w * 256;
#pragma unroll
  for (int j = 0; j < MS_ITEMS; j++) {
    const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
    const u32 d = e < valid ? (u32) (key[j] >> 56) : 255u;
    u32 intra, group;
    ms_match<8>(d, intra, group);
    const u32 old = cnt_w[d];
    if (intra == 0) cnt_w[d] = (u16) (old + group);
    rk[j] = ((old + intra) << 8) | d;
  }
  if (PHASES == 3) { u32 x = 0; for (int j = 0; j < MS_ITEMS; j++) x ^= rk[j]; if (x == 0x12345678u) sink[0] = 1; return; }
  __syncthreads();
  {
    ms_vu16 *s_cnt = (ms_vu16 *) s_cnt_mem;
    u32 c[MS_WAVES];
    u32 tot = 0;
    if (tid < 256) {
#pragma unroll
      for (int i = 0; i < MS_WAVES; i++) { c[i] = s_cnt[i * 256 + tid]; tot += c[i]; }
    }
    u32 all;
    u32 dbase = block_scan_excl<MS_THREADS>(tot, &all, s_scan);
    if (tid < 256) {
      s_obase[tid] = gbase - dbase;
#pragma unroll
      for (int i = 0; i < MS_WAVES; i++) { s_cnt[i * 256 + tid] = (u16) dbase; dbase += c[i]; }
    }
  }
  __syncthreads();
  u32 *s_k1 = reinterpret_cast<u32 *>(s_t);
  u32 *s_p = s_k1 + MS_TILE;
  u8 *s_x = reinterpret_cast<u8 *>(s_p + MS_TILE);
  u8 *s_d = s_x + MS_TILE;
#pragma unroll
  for (int j = 0; j < MS_ITEMS; j++) {
    const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
    const u32 d = rk[j] & 255u;
    const u32 pos = (u32) cnt_w[d] + (rk[j] >> 8);
    s_k1[pos] = (u32) (key[j] >> 24);
    s_p[pos] = (u32) tile_base + e;
    s_x[pos] = (u8) ((((u32) (key[j] >> 19) & 31u) << 3) | ((u32) key[j] & 7u));
    s_d[pos] = (u8) d;
  }
  __syncthreads();
  if (PHASES == 5) { if (s_k1[tid] == 0x12345678u && s_p[tid] == 77u) sink[0] = 1; return; }
#pragma unroll
  for (int j = 0; j < MS_ITEMS; j++) {
    const u32 e = (u32) j * MS_THREADS + tid;
    if (e < valid) {
      const u32 g = s_obase[s_d[e]] + e;
      k1out[g] = s_k1[e];
      pout[g] = s_p[e];
      xout[g] = s_x[e];
    }
  }
}

__global__ void k_init_text(u64 *tb, u64 nw) {
  const u64 w = (u64) blockIdx.x * 256 + threadIdx.x;
  if (w < nw) tb[w] = mix64(w + 99);
}
__global__ void k_checksum(const u32 *k1, const u32 *p, const u8 *x, u64 N, unsigned long long *out) {
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const u64 v = mix64(((u64) k1[i] << 32) ^ p[i] ^ ((u64) x[i] << 40) ^ (i * 0x9E3779B97F4A7C15ull));
  atomicAdd(out, (unsigned long long) (v & 0xFFFFFFFFull));
}

// ---- the output pattern alone: tiles of TILE entries, every tile a run of TILE/256 entries in
// each of 256 bins (uniform digits), written as MODE 0: u32 + u32 + u8 arrays, 1: u64 + u8,
// 2: u32 + u32 only, 3: u64 only
template <int TILE, int MODE>
__global__ __launch_bounds__(512) void k_pattern(u32 ntiles, u32 *__restrict__ a, u32 *__restrict__ b,
                                                 u8 *__restrict__ x, u64 binsize) {
  const u32 tile = ms_xcd_tile(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  constexpr int RUN = TILE / 256;
#pragma unroll
  for (int j = 0; j < TILE / 512; j++) {
    const u32 e = (u32) j * 512 + threadIdx.x;
    const u64 g = (u64) (e / RUN) * binsize + (u64) tile * RUN + (e % RUN) + (MODE >= 4 ? ((e / RUN) * 7u) % 16u : 0u);
    if (MODE == 0 || MODE == 2 || MODE == 4) { a[g] = e; b[g] = tile; }
    if (MODE == 4) x[g] = (u8) e;
    if (MODE == 1 || MODE == 3) reinterpret_cast<u64 *>(a)[g] = ((u64) e << 32) | tile;
    if (MODE == 0 || MODE == 1) x[g] = (u8) e;
  }
}

// ---- variant P: persistent workgroups, the next tile's text words and histogram row are
// loaded before this tile's stores are issued (the vector memory pipeline of a CU is in
// order: a load issued behind a tile's stores waits until they have drained)
struct KeyIn { u64 hi, lo, s0, s1, pw, ps; };
__device__ __forceinline__ KeyIn dna_load8(const Text &t, u64 p0) {
  KeyIn k;
  const u64 w = p0 >> 5, sw = p0 >> 6;
  k.hi = tb_word(t, w); k.lo = tb_word(t, w + 1);
  k.s0 = sp_word(t, sw); k.s1 = sp_word(t, sw + 1);
  k.pw = (p0 & 31) == 0 && p0 > 0 ? tb_word(t, w - 1) : 0ull;
  k.ps = (p0 & 63) == 0 && p0 > 0 ? sp_word(t, sw - 1) : 0ull;
  return k;
}
__device__ __forceinline__ void dna_keys8_from(const KeyIn &in, u64 p0, u64 (&key)[KP_PER]) {
  const int o = (int) (p0 & 31) * 2;
  const u64 hi = in.hi, lo = in.lo;
  const u64 a_hi = o ? (hi << o) | (lo >> (64 - o)) : hi;
  const u64 a_lo = lo << o;
  const int so = (int) (p0 & 63);
  const u64 s0 = in.s0, s1 = in.s1;
  const u64 S = so ? (s0 >> so) | (s1 << (64 - so)) : s0;
  u32 pay;
  if (p0 == 0) pay = P_UNDEF;
  else {
    const u32 c = o ? (u32) (hi >> (64 - o)) & 3u : (u32) in.pw & 3u;
    const bool sp = c < 2u && (so ? (s0 >> (so - 1)) & 1ull : in.ps >> 63);
    pay = sp ? ((c & 1u) ? P_SEP : P_WILD) : c;
  }
  if ((S & ((1ull << (KP_PER + SYMS - 1)) - 1ull)) == 0) {
#pragma unroll
    for (int g = 0; g < KP_PER; g++) {
      const u64 win = g ? (a_hi << (2 * g)) | (a_lo >> (64 - 2 * g)) : a_hi;
      key[g] = (win & (~0ull << LOW_BITS)) | pay;
      pay = (u32) (win >> 62);
    }
    return;
  }
#pragma unroll
  for (int g = 0; g < KP_PER; g++) {
    const u64 win = g ? (a_hi << (2 * g)) | (a_lo >> (64 - 2 * g)) : a_hi;
    const u64 s = (S >> g) & ((1ull << SYMS) - 1ull);
    const int d = s ? __ffsll((unsigned long long) s) - 1 : SYMS;
    if (d == 0) key[g] = (~0ull << DSHIFT) | pay;
    else {
      u64 pre = win >> LOW_BITS;
      u32 dc = 0;
      if (d < SYMS) { pre |= (1ull << (2 * (SYMS - d))) - 1ull; dc = (u32) (SYMS - d); }
      key[g] = (pre << LOW_BITS) | ((u64) dc << DSHIFT) | pay;
    }
    const u32 c = (u32) (win >> 62);
    pay = (c < 2u && (s & 1ull)) ? ((c & 1u) ? P_SEP : P_WILD) : c;
  }
}
template <bool PREFETCH>
__global__ __launch_bounds__(MS_THREADS) void k_scatter_a_p(
    Text t, u64 N, u32 last_valid, const u32 *__restrict__ scanned, u32 ntiles, u32 per_wg,
    u32 *__restrict__ k1out, u8 *__restrict__ xout, u32 *__restrict__ pout) {
  __shared__ u64 s_t[5120];
  __shared__ u16 s_cnt_mem[MS_WAVES * 256];
  __shared__ u32 s_obase[256];
  __shared__ u32 s_scan[MS_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // workgroup b works on tiles [b * per_wg, (b + 1) * per_wg): neighbours in the grid (one XCD
  // gets every eighth workgroup) -- bins' frontiers move in step
  // (per_wg == 0: interleaved -- workgroup b takes the tiles b, b + grid, ...: the workgroups
  // that run together work on neighbouring tiles, as a grid of one tile per workgroup does)
  // (one XCD gets every eighth workgroup: it works on an eighth of the tiles, so that the
  // lines two neighbouring tiles share meet in one L2)
  const u32 xper = (ntiles + 7u) >> 3;
  const u32 step = per_wg ? 1u : gridDim.x >> 3;
  const u32 t_first = per_wg ? blockIdx.x * per_wg : (blockIdx.x & 7u) * xper + (blockIdx.x >> 3);
  const u32 x_end = ((blockIdx.x & 7u) + 1u) * xper < ntiles ? ((blockIdx.x & 7u) + 1u) * xper : ntiles;
  const u32 t_end = per_wg ? (t_first + per_wg < ntiles ? t_first + per_wg : ntiles) : x_end;
  if (t_first >= ntiles) return;
  KeyIn cur = dna_load8(t, (u64) t_first * MS_TILE + (u64) tid * KP_PER);
  u32 gcur = tid < 256 ? scanned[(u64) t_first * 256 + tid] : 0u;
  for (u32 tile = t_first; tile < t_end; tile += step) {
    const u64 tile_base = (u64) tile * MS_TILE;
    const u32 valid = tile + 1u == ntiles ? last_valid : (u32) MS_TILE;
    KeyIn nxt = cur;
    u32 gnxt = 0;
    if (PREFETCH && tile + step < t_end) {
      nxt = dna_load8(t, tile_base + (u64) step * MS_TILE + (u64) tid * KP_PER);
      if (tid < 256) gnxt = scanned[(u64) (tile + step) * 256 + tid];
    }
    for (int i = tid; i < MS_WAVES * 256 / 2; i += MS_THREADS) reinterpret_cast<u32 *>(s_cnt_mem)[i] = 0;
    const u32 gbase = gcur;
    u64 key[MS_ITEMS];
    u32 rk[MS_ITEMS];
    {
      u64 key8[KP_PER];
      const u64 p0 = tile_base + (u64) tid * KP_PER;
      if (p0 < N) dna_keys8_from(cur, p0, key8);
      else for (int g = 0; g < KP_PER; g++) key8[g] = ~0ull;
#pragma unroll
      for (int g = 0; g < KP_PER; g++) s_t[g * MS_PAD + tid] = key8[g];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) {
      const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
      key[j] = s_t[(e & 7u) * MS_PAD + (e >> 3)];
    }
    __syncthreads();
    ms_vu16 *cnt_w = (ms_vu16 *) s_cnt_mem + w * 256;
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) {
      const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
      const u32 d = e < valid ? (u32) (key[j] >> 56) : 255u;
      u32 intra, group;
      ms_match<8>(d, intra, group);
      const u32 old = cnt_w[d];
      if (intra == 0) cnt_w[d] = (u16) (old + group);
      rk[j] = ((old + intra) << 8) | d;
    }
    __syncthreads();
    {
      ms_vu16 *s_cnt = (ms_vu16 *) s_cnt_mem;
      u32 c[MS_WAVES];
      u32 tot = 0;
      if (tid < 256) {
#pragma unroll
        for (int i = 0; i < MS_WAVES; i++) { c[i] = s_cnt[i * 256 + tid]; tot += c[i]; }
      }
      u32 all;
      u32 dbase = block_scan_excl<MS_THREADS>(tot, &all, s_scan);
      if (tid < 256) {
        s_obase[tid] = gbase - dbase;
#pragma unroll
        for (int i = 0; i < MS_WAVES; i++) { s_cnt[i * 256 + tid] = (u16) dbase; dbase += c[i]; }
      }
    }
    __syncthreads();
    u32 *s_k1 = reinterpret_cast<u32 *>(s_t);
    u32 *s_p = s_k1 + MS_TILE;
    u8 *s_x = reinterpret_cast<u8 *>(s_p + MS_TILE);
    u8 *s_d = s_x + MS_TILE;
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) {
      const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
      const u32 d = rk[j] & 255u;
      const u32 pos = (u32) cnt_w[d] + (rk[j] >> 8);
      s_k1[pos] = (u32) (key[j] >> 24);
      s_p[pos] = (u32) tile_base + e;
      s_x[pos] = (u8) ((((u32) (key[j] >> 19) & 31u) << 3) | ((u32) key[j] & 7u));
      s_d[pos] = (u8) d;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) {
      const u32 e = (u32) j * MS_THREADS + tid;
      if (e < valid) {
        const u32 g = s_obase[s_d[e]] + e;
        k1out[g] = s_k1[e];
        pout[g] = s_p[e];
        xout[g] = s_x[e];
      }
    }
    if (PREFETCH) { cur = nxt; gcur = gnxt; }
    else if (tile + step < t_end) {
      cur = dna_load8(t, tile_base + (u64) step * MS_TILE + (u64) tid * KP_PER);
      gcur = tid < 256 ? scanned[(u64) (tile + step) * 256 + tid] : 0u;
    }
    __syncthreads();      // (the staging area is read by the stores above)
  }
}

// ---- the pattern a write-combining scatter would make: a workgroup takes K consecutive
// tiles; runs start OFF entries into a line (not aligned), but what is written per tile and
// bin is the whole line that became full (the rest waits in LDS); the head and the tail of
// the K tiles' range are partial lines, once per workgroup and bin
template <int K>
__global__ __launch_bounds__(512) void k_pattern_wc(u32 ngroups, u32 *__restrict__ a, u32 *__restrict__ b,
                                                    u8 *__restrict__ x, u64 binsize) {
  const u32 grp = ms_xcd_tile(blockIdx.x, ngroups);
  if (grp >= ngroups) return;
  for (int k = 0; k <= K; k++) {
    // step k < K: the line that tile k completes (all but the first: the head); step K: the tail
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u32 e = (u32) j * 512 + threadIdx.x;
      const u32 bin = e / 16, off = (bin * 7u) % 16u;
      // entries [lo, hi) of the bin's range of this group, in units of entries from the group's start
      const u32 lo = k == 0 ? 0u : (u32) k * 16u - off, hi = k == K ? (u32) K * 16u : (u32) (k + 1) * 16u - off;
      const u32 i = lo + (e % 16);
      if (i < hi && (k < K || off != 0)) {
        const u64 g = (u64) bin * binsize + (u64) grp * K * 16 + off + i;
        a[g] = e; b[g] = grp; x[g] = (u8) e;
      }
    }
  }
}

// ---- variant T8: tiles of 8192 entries, 1024 threads, 2 workgroups per CU (75 KB of LDS):
// runs of 32 entries per digit and tile
constexpr int T8_TILE = 8192, T8_THREADS = 1024, T8_WAVES = 16, T8_PAD = 1032;
__global__ __launch_bounds__(T8_THREADS) void k_hist_a8(Text t, u64 N, u32 *__restrict__ hist) {
  __shared__ u32 h[T8_WAVES][256];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < T8_WAVES * 256; i += T8_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const u64 p0 = (u64) blockIdx.x * T8_TILE + (u64) tid * KP_PER;
  if (p0 < N) {
    u64 key[KP_PER];
    dna_keys8(t, p0, key);
#pragma unroll
    for (int g = 0; g < KP_PER; g++) atomicAdd(&h[w][(u32) (key[g] >> 56)], 1u);
  }
  __syncthreads();
  if (tid < 256) {
    u32 c = 0;
    for (int i = 0; i < T8_WAVES; i++) c += h[i][tid];
    hist[(u64) blockIdx.x * 256 + tid] = c;
  }
}
__global__ __launch_bounds__(T8_THREADS) void k_scatter_a8(
    Text t, u64 N, const u32 *__restrict__ scanned, u32 ntiles,
    u32 *__restrict__ k1out, u8 *__restrict__ xout, u32 *__restrict__ pout) {
  __shared__ u64 s_t[8 * T8_PAD];                 // 66 KB: transposition, then staging
  __shared__ u16 s_cnt_mem[T8_WAVES * 256];
  __shared__ u32 s_obase[256];
  __shared__ u32 s_scan[T8_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile = ms_xcd_tile(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  const u64 tile_base = (u64) tile * T8_TILE;
  for (int i = tid; i < T8_WAVES * 256 / 2; i += T8_THREADS) reinterpret_cast<u32 *>(s_cnt_mem)[i] = 0;
  u32 gbase = 0;
  if (tid < 256) gbase = scanned[(u64) tile * 256 + tid];
  u64 key[8];
  u32 rk[8];
  {
    u64 key8[KP_PER];
    const u64 p0 = tile_base + (u64) tid * KP_PER;
    if (p0 < N) dna_keys8(t, p0, key8);
    else for (int g = 0; g < KP_PER; g++) key8[g] = ~0ull;
#pragma unroll
    for (int g = 0; g < KP_PER; g++) s_t[g * T8_PAD + tid] = key8[g];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const u32 e = (u32) w * 512 + (u32) j * 64 + lane;
    key[j] = s_t[(e & 7u) * T8_PAD + (e >> 3)];
  }
  __syncthreads();
  ms_vu16 *cnt_w = (ms_vu16 *) s_cnt_mem + w * 256;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const u32 d = (u32) (key[j] >> 56);
    u32 intra, group;
    ms_match<8>(d, intra, group);
    const u32 old = cnt_w[d];
    if (intra == 0) cnt_w[d] = (u16) (old + group);
    rk[j] = ((old + intra) << 8) | d;
  }
  __syncthreads();
  {
    ms_vu16 *s_cnt = (ms_vu16 *) s_cnt_mem;
    u32 c[T8_WAVES];
    u32 tot = 0;
    if (tid < 256) {
#pragma unroll
      for (int i = 0; i < T8_WAVES; i++) { c[i] = s_cnt[i * 256 + tid]; tot += c[i]; }
    }
    u32 all;
    u32 dbase = block_scan_excl<T8_THREADS>(tot, &all, s_scan);
    if (tid < 256) {
      s_obase[tid] = gbase - dbase;
#pragma unroll
      for (int i = 0; i < T8_WAVES; i++) { s_cnt[i * 256 + tid] = (u16) dbase; dbase += c[i]; }
    }
  }
  __syncthreads();
  u32 *s_k1 = reinterpret_cast<u32 *>(s_t);
  u16 *s_p = reinterpret_cast<u16 *>(s_k1 + T8_TILE);
  u8 *s_x = reinterpret_cast<u8 *>(s_p + T8_TILE);
  u8 *s_d = s_x + T8_TILE;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const u32 e = (u32) w * 512 + (u32) j * 64 + lane;
    const u32 d = rk[j] & 255u;
    const u32 pos = (u32) cnt_w[d] + (rk[j] >> 8);
    s_k1[pos] = (u32) (key[j] >> 24);
    s_p[pos] = (u16) e;
    s_x[pos] = (u8) ((((u32) (key[j] >> 19) & 31u) << 3) | ((u32) key[j] & 7u));
    s_d[pos] = (u8) d;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const u32 e = (u32) j * T8_THREADS + tid;
    const u32 g = s_obase[s_d[e]] + e;
    k1out[g] = s_k1[e];
    pout[g] = (u32) tile_base + s_p[e];
    xout[g] = s_x[e];
  }
}

// ---- do two partial writes of a line merge in the L2?  Aligned runs of 16 entries, every run
// written in two halves: HOW 0 by the same workgroup, one half after the other (a barrier
// between); HOW 1 the second half by the workgroup of the NEXT tile (as the real kernel's
// unaligned runs are: the neighbour completes the line)
template <int HOW>
__global__ __launch_bounds__(512) void k_pattern_halves(u32 ntiles, u32 *__restrict__ a, u32 *__restrict__ b,
                                                        u8 *__restrict__ x, u64 binsize) {
  const u32 tile = ms_xcd_tile(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  for (int half = 0; half < 2; half++) {
    const u32 t2 = HOW == 1 && half == 1 ? (tile + 1 < ntiles ? tile + 1 : 0u) : tile;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u32 e = (u32) j * 512 + threadIdx.x;
      if (((e % 16) < 8) == (half == 0)) {
        const u64 g = (u64) (e / 16) * binsize + (u64) t2 * 16 + (e % 16);
        a[g] = e; b[g] = tile; x[g] = (u8) e;
      }
    }
    if (HOW == 0) __syncthreads();
  }
}

int main() {
  const u64 N = 1ull << 31;                     // symbols = entries
  const u32 ntiles = (u32) (N / MS_TILE);
  u64 *tb, *sp; u32 *hist, *scanned, *binstart, *k1, *p, *sink; u8 *x; unsigned long long *sum;
  const u64 nw_tb = N / 32 + 2, nw_sp = N / 64 + 2;
  CK(hipMalloc(&tb, nw_tb * 8)); CK(hipMalloc(&sp, nw_sp * 8));
  CK(hipMalloc(&hist, (u64) ntiles * 256 * 4)); CK(hipMalloc(&scanned, (u64) ntiles * 256 * 4));
  CK(hipMalloc(&binstart, 257 * 4)); CK(hipMalloc(&k1, N * 4)); CK(hipMalloc(&p, N * 4)); CK(hipMalloc(&x, N + 4096));
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&sum, 8));
  k_init_text<<<(u32) (nw_tb / 256 + 1), 256>>>(tb, nw_tb);
  CK(hipMemset(sp, 0, nw_sp * 8));
  Text t = {tb, sp, N, nw_tb, nw_sp};
  k_hist_a<<<ntiles, MS_THREADS>>>(t, N, hist);
  CK(hipMemset(binstart, 0, 257 * 4));
  k_colsum<<<1024, 256>>>(hist, ntiles, binstart + 1);
  CK(hipDeviceSynchronize());
  std::vector<u32> hb(257);
  CK(hipMemcpy(hb.data(), binstart, 257 * 4, hipMemcpyDeviceToHost));
  for (int d = 1; d <= 256; d++) hb[d] += hb[d - 1];
  CK(hipMemcpy(binstart, hb.data(), 257 * 4, hipMemcpyHostToDevice));
  k_colscan<<<256, 1024>>>(hist, ntiles, binstart, scanned);
  CK(hipDeviceSynchronize());
  printf("N = %llu, %u tiles, bins end at %u\n", (unsigned long long) N, ntiles, hb[256]);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, bool check, auto &&fn) {
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
      CK(hipEventRecord(e0));
      fn();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    unsigned long long h = 0;
    if (check) {
      CK(hipMemset(sum, 0, 8));
      k_checksum<<<(u32) (N / 256), 256>>>(k1, p, x, N, sum);
      CK(hipMemcpy(&h, sum, 8, hipMemcpyDeviceToHost));
    }
    printf("%-46s %7.3f ms  (x 3e9/N = %6.2f ms)  %016llx\n", name, best, best * 3e9 / (double) N, h);
  };
  timeit("hist_a", false, [&] { k_hist_a<<<ntiles, MS_THREADS>>>(t, N, hist); });
  timeit("1 keygen only", false, [&] { k_scatter_a<1, false><<<ntiles, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, k1, x, p, sink); });
  timeit("2 + transposition", false, [&] { k_scatter_a<2, false><<<ntiles, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, k1, x, p, sink); });
  timeit("3 + ranking", false, [&] { k_scatter_a<3, false><<<ntiles, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, k1, x, p, sink); });
  timeit("5 + offsets + staging", false, [&] { k_scatter_a<5, false><<<ntiles, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, k1, x, p, sink); });
  timeit("6 all (the engine's kernel)", true, [&] { k_scatter_a<6, false><<<ntiles, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, k1, x, p, sink); });
  timeit("direct keys: 2 keygen", false, [&] { k_scatter_a<2, true><<<ntiles, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, k1, x, p, sink); });
  timeit("direct keys: 3 + ranking", false, [&] { k_scatter_a<3, true><<<ntiles, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, k1, x, p, sink); });
  timeit("direct keys: 6 all", true, [&] { k_scatter_a<6, true><<<ntiles, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, k1, x, p, sink); });
  {
    // tiles of 8192: their own histogram and column scan
    const u32 nt8 = (u32) (N / T8_TILE);
    k_hist_a8<<<nt8, T8_THREADS>>>(t, N, hist);
    k_colscan<<<256, 1024>>>(hist, nt8, binstart, scanned);
    CK(hipDeviceSynchronize());
    timeit("T8 tiles of 8192, 1024 threads", true, [&] { k_scatter_a8<<<nt8, T8_THREADS>>>(t, N, scanned, nt8, k1, x, p); });
    // (back to tiles of 4096 for what follows)
    k_hist_a<<<ntiles, MS_THREADS>>>(t, N, hist);
    k_colscan<<<256, 1024>>>(hist, ntiles, binstart, scanned);
    CK(hipDeviceSynchronize());
  }
  for (u32 g : {256u * 3, 256u * 6, 256u * 12, 256u * 48}) {
    char nm[96];
    snprintf(nm, sizeof nm, "P  interleaved, grid %u, prefetch", g);
    timeit(nm, true, [&] { k_scatter_a_p<true><<<g, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, 0, k1, x, p); });
    snprintf(nm, sizeof nm, "P  interleaved, grid %u, no prefetch", g);
    timeit(nm, true, [&] { k_scatter_a_p<false><<<g, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, 0, k1, x, p); });
  }
  for (u32 per : {64u}) {
    char nm[96];
    const u32 g = (ntiles + per - 1) / per;
    snprintf(nm, sizeof nm, "P  persistent, %u tiles per workgroup, prefetch", per);
    timeit(nm, true, [&] { k_scatter_a_p<true><<<g, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, per, k1, x, p); });
    snprintf(nm, sizeof nm, "P  persistent, %u tiles per workgroup, no prefetch", per);
    timeit(nm, true, [&] { k_scatter_a_p<false><<<g, MS_THREADS>>>(t, N, MS_TILE, scanned, ntiles, per, k1, x, p); });
  }
  {
    // (a and b as one allocation for the u64 modes)
    u32 *ab; CK(hipMalloc(&ab, N * 8 + 4096));
    auto pat = [&](const char *name, auto &&fn) { timeit(name, false, fn); };
    pat("pattern 4096: u32+u32+u8 (runs of 16)", [&] { k_pattern<4096, 0><<<(u32) (N / 4096), 512>>>((u32) (N / 4096), ab, ab + N, x, N / 256); });
    pat("pattern 4096: u32+u32+u8, runs not aligned", [&] { k_pattern<4096, 4><<<(u32) (N / 4096), 512>>>((u32) (N / 4096), ab, ab + N + 64, x, N / 256); });
    pat("pattern 8192: u32+u32+u8, runs not aligned", [&] { k_pattern<8192, 4><<<(u32) (N / 8192), 512>>>((u32) (N / 8192), ab, ab + N + 64, x, N / 256); });
    pat("write-combining pattern, 4 tiles a workgroup", [&] { k_pattern_wc<4><<<(u32) (N / 4096 / 4), 512>>>((u32) (N / 4096 / 4), ab, ab + N + 64, x, N / 256); });
    pat("write-combining pattern, 8 tiles a workgroup", [&] { k_pattern_wc<8><<<(u32) (N / 4096 / 8), 512>>>((u32) (N / 4096 / 8), ab, ab + N + 64, x, N / 256); });
    pat("write-combining pattern, 16 tiles a workgroup", [&] { k_pattern_wc<16><<<(u32) (N / 4096 / 16), 512>>>((u32) (N / 4096 / 16), ab, ab + N + 64, x, N / 256); });
    pat("write-combining pattern, 64 tiles a workgroup", [&] { k_pattern_wc<64><<<(u32) (N / 4096 / 64), 512>>>((u32) (N / 4096 / 64), ab, ab + N + 64, x, N / 256); });
    pat("aligned runs in two halves, same workgroup", [&] { k_pattern_halves<0><<<(u32) (N / 4096), 512>>>((u32) (N / 4096), ab, ab + N + 64, x, N / 256); });
    pat("aligned runs in two halves, second by the neighbour", [&] { k_pattern_halves<1><<<(u32) (N / 4096), 512>>>((u32) (N / 4096), ab, ab + N + 64, x, N / 256); });
    pat("pattern 4096: u64+u8", [&] { k_pattern<4096, 1><<<(u32) (N / 4096), 512>>>((u32) (N / 4096), ab, ab + N, x, N / 256); });
    pat("pattern 4096: u32+u32", [&] { k_pattern<4096, 2><<<(u32) (N / 4096), 512>>>((u32) (N / 4096), ab, ab + N, x, N / 256); });
    pat("pattern 4096: u64", [&] { k_pattern<4096, 3><<<(u32) (N / 4096), 512>>>((u32) (N / 4096), ab, ab + N, x, N / 256); });
    pat("pattern 8192: u32+u32+u8 (runs of 32)", [&] { k_pattern<8192, 0><<<(u32) (N / 8192), 512>>>((u32) (N / 8192), ab, ab + N, x, N / 256); });
    pat("pattern 8192: u64+u8", [&] { k_pattern<8192, 1><<<(u32) (N / 8192), 512>>>((u32) (N / 8192), ab, ab + N, x, N / 256); });
    pat("pattern 16384: u32+u32+u8 (runs of 64)", [&] { k_pattern<16384, 0><<<(u32) (N / 16384), 512>>>((u32) (N / 16384), ab, ab + N, x, N / 256); });
    pat("pattern 16384: u64+u8", [&] { k_pattern<16384, 1><<<(u32) (N / 16384), 512>>>((u32) (N / 16384), ab, ab + N, x, N / 256); });
    pat("pattern 65536: u32+u32+u8 (runs of 256)", [&] { k_pattern<65536, 0><<<(u32) (N / 65536), 512>>>((u32) (N / 65536), ab, ab + N, x, N / 256); });
  }
  return 0;
}
