// esa_pck.hip -- the packed index of `gt packedindex` (INDEX.bdx) assembled on
// the device from the resident BWT and suffix array (C ABI: include/gtamd_pck.h).
//
// File layout (src/match/eis-blockcomp.c:1888-2094, restated in DESIGN.md 9a):
//   header | ext header (locate) | zeros up to 8192 | cw records of all buckets,
//   one bit string | var parts of all buckets, one bit string | region list
// A bucket covers L = blockSize x bucketBlocks positions of the BWT.  Its
// constant-width record: number of occurrences of every letter before the
// bucket, bit offset of its var part, bits of its permutation indices, the
// composition index of each block, [one locate bit per position]; its var part:
// the permutation index of each block (as many bits as its composition needs),
// then the locate marks: [count, (position in bucket,] text position [)]...
//
// Two passes over the tables (tiles of <= 256 buckets, one thread per bucket):
// k_pck_tile<false> counts per tile (letters, var bits, region starts/ends),
// one small scan, k_pck_tile<true> recomputes the tile and writes every field
// with atomicOr into the zeroed image (bit strings are most significant bit
// first, so a field is OR-ed into byte-swapped 64-bit words).  Block -> index
// pair through a table of sigma^blockSize entries built once per geometry.
// The few bits the reference's staging buffers leave stale in the file (in the
// last bucket only) are reproduced by the host from the image's own tail.
#include <algorithm>
#include <vector>
#include "esa_common.h"
#include "esa_devutil.h"
#include "esa_pck_replay.h"
#include "../../include/gtamd_pck.h"

// from esa_engine.hip
extern "C" int gtamd_esa_internal_info(const gtamd_esa_ctx *c, int *device, u32 *sigma,
                                       u32 *numparts);

namespace {

constexpr int PCK_THREADS = 256;
constexpr u32 PCK_TILE_POS = 16384;     // positions a tile stages in LDS at most
constexpr u32 PCK_MAX_SIGMA = 30;
constexpr u32 PCK_EXTRA_COLS = 3;       // var bits, region starts, region ends
constexpr u64 PCK_TAIL_RECORDS = 65536; // var offsets of the last buckets kept for the replay
constexpr u32 LDS_SPECIAL = 30;         // LDS code of the wildcard; separator 31
constexpr u32 LDS_MARK = 0x40;

// gt_requiredUInt64Bits, src/core/bitpackstringop.c:60-79
__host__ __device__ inline u32 reqbits(u64 v) {
  u32 r = 1;
  while (v >>= 1) r++;
  return r;
}

struct PckGeom {
  u64 N;                 // entries of the tables
  u64 nb;                // buckets
  u64 first_special_row; // rows from here on hold suffixes that start with a special
  u64 cw_base_bit, var_base_bit;  // bit positions in the image
  u64 lut_entries;       // sigma^B, 0: indices are computed, not looked up
  u64 inv_L;             // ceil(2^32 / L)
  u32 sigma, B, K, L, LP;
  u32 T, ntiles;
  u32 locint, locmask, loc_pow2, loc_bitmap, loc_count;
  u32 comp_idx_bits, var_off_bits, cb_off_bits, bits_orig_pos;
  u32 cw_bits, pre_var_idx, pre_cb_off, pre_comp_idx, pre_cw_ext;
  u32 sym_bits[PCK_MAX_SIGMA + 2], sym_off[PCK_MAX_SIGMA + 2];
  u32 lds_cw_off, lds_cw_words, lds_var_off, lds_var_words;   // EMIT: LDS copies of the tile's bit strings (0 words: none)
  u32 reversible, bits_orig_rank;   // -sprank: specials sorted reversibly, bits per rank
  u64 total_specials;
};

// ---- block -> (composition index, permutation index, bits) -------------------
__host__ __device__ inline u64 binom(u32 n, u32 k) {
  if (k > n) return 0;
  if (k > n - k) k = n - k;
  u64 r = 1;
  for (u32 i = 1; i <= k; i++) r = r * (n - k + i) / i;
  return r;
}
__host__ __device__ inline u64 factorial(u32 n) {
  u64 r = 1;
  while (n > 1) r *= n--;
  return r;
}
__host__ __device__ inline u64 arrangements(const u32 *cnt, u32 sigma, u32 total) {
  u64 r = factorial(total);
  for (u32 s = 0; s < sigma; s++) r /= factorial(cnt[s]);
  return r;
}
// index of the composition in ascending order of (count of letter 0, count of
// letter 1, ...), src/match/eis-seqblocktranslate.c:156-246
__host__ __device__ inline u64 composition_index(const u32 *cnt, u32 sigma, u32 B) {
  u64 r = 0;
  u32 left = B;
  for (u32 i = 0; i + 1 < sigma; i++) {
    const u32 k = sigma - i - 1;
    for (u32 v = 0; v < cnt[i]; v++) r += binom(left - v + k - 1, k - 1);
    left -= cnt[i];
  }
  return r;
}
// packed result: permutation index (bits 0..39) | composition index (40..57) |
// bits of the permutation index (58..63)
__host__ __device__ inline u64 pack_indices(u64 perm, u64 comp, u32 pbits) {
  return perm | (comp << 40) | ((u64) pbits << 58);
}
// symbols of a block (letters only) -> packed indices; cnt is scratch for sigma counters
__device__ inline u64 block_indices(const u8 *sym, u32 sigma, u32 B, u32 *cnt) {
  for (u32 s = 0; s < sigma; s++) cnt[s] = 0;
  for (u32 i = 0; i < B; i++) cnt[sym[i]]++;
  const u64 comp = composition_index(cnt, sigma, B);
  u64 m = arrangements(cnt, sigma, B);
  const u32 pbits = m > 1 ? reqbits(m - 1) : 0;
  // rank of the block among the arrangements of its multiset: of the m
  // arrangements of what is left, m * cnt[s] / left start with letter s
  u64 perm = 0;
  u32 left = B;
  for (u32 i = 0; i < B && m > 1; i++) {
    u32 below = 0;
    for (u32 s = 0; s < sym[i]; s++) below += cnt[s];
    perm += m * below / left;
    m = m * cnt[sym[i]] / left;
    cnt[sym[i]]--;
    left--;
  }
  return pack_indices(perm, comp, pbits);
}

__global__ void k_pck_lut(u64 entries, u32 sigma, u32 B, u64 *lut) {
  const u64 idx = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= entries) return;
  u8 sym[16];
  u32 cnt[PCK_MAX_SIGMA + 2];
  u64 v = idx;
  for (int i = (int) B - 1; i >= 0; i--) { sym[i] = (u8) (v % sigma); v /= sigma; }
  lut[idx] = block_indices(sym, sigma, B, cnt);
}

// ---- bit strings, most significant bit first ---------------------------------
__device__ __forceinline__ void put_bits(u64 *img, u64 bit, u32 nbits, u64 v) {
  if (nbits == 0) return;
  // the reference's gt_bsStore* keep the low nbits bits of a value that does not
  // fit (its size bound for the var part can be too small)
  if (nbits < 64) v &= (1ull << nbits) - 1;
  const u64 w = bit >> 6;
  const u32 o = (u32) (bit & 63);
  if (o + nbits <= 64) {
    const u64 x = v << (64 - o - nbits);
    if (x) atomicOr((unsigned long long *) &img[w], (unsigned long long) __builtin_bswap64(x));
  } else {
    const u32 r = o + nbits - 64;
    const u64 hi = v >> r, lo = v << (64 - r);
    if (hi) atomicOr((unsigned long long *) &img[w], (unsigned long long) __builtin_bswap64(hi));
    if (lo) atomicOr((unsigned long long *) &img[w + 1], (unsigned long long) __builtin_bswap64(lo));
  }
}

// where the fields of a tile go: into an LDS copy of the tile's part of the bit
// string (words relative to w0, in bit-string order: byte-swapped when they
// leave), or straight into the image when the tile's part does not fit
struct BitSink {
  u64 *lds;   // nullptr: global
  u64 w0;     // first 64-bit word of the tile's part
  u64 *img;
};
__device__ __forceinline__ void sink_put(const BitSink &k, u64 bit, u32 nbits, u64 v) {
  if (k.lds == nullptr) { put_bits(k.img, bit, nbits, v); return; }
  if (nbits == 0) return;
  if (nbits < 64) v &= (1ull << nbits) - 1;
  const u64 w = (bit >> 6) - k.w0;
  const u32 o = (u32) (bit & 63);
  if (o + nbits <= 64) {
    const u64 x = v << (64 - o - nbits);
    if (x) atomicOr((unsigned long long *) &k.lds[w], (unsigned long long) x);
  } else {
    const u32 r = o + nbits - 64;
    const u64 hi = v >> r, lo = v << (64 - r);
    if (hi) atomicOr((unsigned long long *) &k.lds[w], (unsigned long long) hi);
    if (lo) atomicOr((unsigned long long *) &k.lds[w + 1], (unsigned long long) lo);
  }
}
// the tile's words leave in whole lines; the first and the last word may be
// shared with the neighbouring tiles (or the other bit string)
__device__ __forceinline__ void sink_flush(const BitSink &k, u32 nwords) {
  if (k.lds == nullptr) return;
  for (u32 i = threadIdx.x; i < nwords; i += blockDim.x) {
    const u64 x = __builtin_bswap64(k.lds[i]);
    if (i == 0 || i + 1 == nwords) { if (x) atomicOr((unsigned long long *) &k.img[k.w0 + i], (unsigned long long) x); }
    else k.img[k.w0 + i] = x;
  }
}

// A thread writes the fields of its record in increasing bit order: they are put
// together in a 64-bit accumulator and leave word by word -- the first and the
// last word of a record by atomic OR (the neighbouring records share them), the
// words between by a plain store (nobody else has bits there, the memory is
// zeroed).  Five LDS operations per record instead of one or two per field.
struct BitWriter {
  BitSink k;
  u64 acc, word;
  u32 used;
  bool first;
  __device__ __forceinline__ void begin(const BitSink &sink, u64 bit) {
    k = sink; word = bit >> 6; used = (u32) (bit & 63); acc = 0; first = true;
  }
  __device__ __forceinline__ void flush_word() {
    if (acc) {
      if (k.lds != nullptr) {
        u64 *d = &k.lds[word - k.w0];
        if (first) atomicOr((unsigned long long *) d, (unsigned long long) acc); else *d = acc;
      } else {
        const u64 x = __builtin_bswap64(acc);
        if (first) atomicOr((unsigned long long *) &k.img[word], (unsigned long long) x); else k.img[word] = x;
      }
    }
    word++; acc = 0; used = 0; first = false;
  }
  // zeros up to bit position `bit` (not before the current position)
  __device__ __forceinline__ void skip_to(u64 bit) {
    const u64 tw = bit >> 6;
    while (word < tw) flush_word();
    used = (u32) (bit & 63);
  }
  __device__ __forceinline__ void put(u32 nbits, u64 v) {
    if (nbits == 0) return;
    // (the reference's gt_bsStore* keep the low nbits bits of a value that does not fit)
    if (nbits < 64) v &= (1ull << nbits) - 1;
    if (used + nbits <= 64) {
      acc |= v << (64 - used - nbits);
      used += nbits;
      if (used == 64) flush_word();
    } else {
      const u32 r = used + nbits - 64;
      acc |= v >> r;
      flush_word();
      acc = v << (64 - r);
      used = r;
    }
  }
  __device__ __forceinline__ void put_at(u64 bit, u32 nbits, u64 v) { skip_to(bit); put(nbits, v); }
  __device__ __forceinline__ void finish() {
    if (acc) {
      if (k.lds != nullptr) atomicOr((unsigned long long *) &k.lds[word - k.w0], (unsigned long long) acc);
      else atomicOr((unsigned long long *) &k.img[word], (unsigned long long) __builtin_bswap64(acc));
    }
    acc = 0;
  }
};

__global__ void k_pck_count_specials(const u8 *bwt, u64 N, unsigned long long *out) {
  __shared__ u32 s4[4];
  u32 c = 0;
  const u64 nvec = (((uintptr_t) bwt) & 15) == 0 ? N / 16 : 0;   // 16 bytes per load when aligned
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (u64) gridDim.x * blockDim.x) {
    const uint4 v = reinterpret_cast<const uint4 *>(bwt)[i];
    const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // bytes >= 254: all of the top seven bits set
      const u32 t = w[k] & (w[k] >> 1) & (w[k] >> 2) & (w[k] >> 3) & (w[k] >> 4) & (w[k] >> 5) & (w[k] >> 6);
      c += (u32) __popc(t & 0x02020202u);
    }
  }
  for (u64 i = nvec * 16 + (u64) blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64) gridDim.x * blockDim.x)
    c += bwt[i] >= 254;
  u32 tot;
  (void) block_scan_excl_sum(c, &tot, s4);
  if (threadIdx.x == 0 && tot) atomicAdd(out, (unsigned long long) tot);
}

// ---- -sprank: ranks of the specials of the text ---------------------------------
// The text is not at hand here, the tables are: position q of the text holds a
// special iff some row r has a special BWT symbol and suf[r] == q + 1.
__global__ void k_pck_special_bitmap(const u8 *__restrict__ bwt, const u64 *__restrict__ suf, u64 N,
                                     unsigned long long *bits) {
  for (u64 r = (u64) blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (u64) gridDim.x * blockDim.x)
    if (bwt[r] >= 254) {
      const u64 v = suf[r];
      if (v) atomicOr(&bits[(v - 1) >> 6], 1ull << ((v - 1) & 63));
    }
}
__global__ void k_pck_popc_words(const u64 *__restrict__ bits, u64 nwords, u32 *__restrict__ cnt) {
  const u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nwords) cnt[i] = (u32) __popcll(bits[i]);
}
// specials in text[0, q): specialsRank, src/match/eis-specialsrank.c:160-190
__device__ __forceinline__ u64 special_rank(const u64 *__restrict__ bits, const u32 *__restrict__ pre, u64 q) {
  const u64 w = q >> 6;
  const u32 o = (u32) (q & 63);
  return (u64) pre[w] + (o ? (u64) __popcll(bits[w] & ((1ull << o) - 1)) : 0);
}

// the table entries of the blocks b8 .. b8 + 7 of a bucket (its symbols at
// `mine` in LDS): the eight look-ups are issued together
__device__ __forceinline__ void block_entries8(const PckGeom &g, const u8 *mine,
                                               const u64 *__restrict__ lut, u32 b8, u32 nblk,
                                               u64 e[8]) {
  u32 code[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    u32 c = 0;
    if (b8 + k < nblk) {
      const u8 *bp = mine + (size_t) (b8 + k) * g.B;
      for (u32 i = 0; i < g.B; i++) {
        const u32 x = bp[i] & 63u;
        c = c * g.sigma + (x >= LDS_SPECIAL ? 0u : x);    // region symbols fall back to letter 0
      }
    }
    code[k] = c;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) e[k] = b8 + k < nblk ? lut[code[k]] : 0;
}

// ---- the tile kernel -----------------------------------------------------------
// tile_tot: (sigma + 3) columns of ntiles u64: COUNT writes the tile's totals,
// the scan turns each column into exclusive prefixes, EMIT reads them.
template <bool EMIT>
__global__ __launch_bounds__(PCK_THREADS) void k_pck_tile(
    PckGeom g, const u8 *__restrict__ bwt, const u64 *__restrict__ suf,
    const u64 *__restrict__ lut, u64 *tile_tot, u64 *img, u64 *rstart, u64 *rend,
    u64 *tail_off, u64 tail_first, const u64 *__restrict__ spbits, const u32 *__restrict__ sppre) {
  extern __shared__ u8 smem[];
  __shared__ u32 s4[4];
  const u32 tid = threadIdx.x;
  const u64 tile = blockIdx.x;
  const u64 b0 = tile * g.T;                               // first bucket of the tile
  const u32 nbk = (u32) min((u64) g.T, g.nb - b0);         // buckets in this tile
  const u64 p0 = b0 * g.L;                                 // first position
  const u32 npos = (u32) min((u64) nbk * g.L, g.N > p0 ? g.N - p0 : 0);
  u8 *s_sym = smem;                                        // [T][LP]
  u16 *s_cnt = (u16 *) (smem + (((size_t) g.T * g.LP + 15) & ~(size_t) 15));  // [sigma][T]

  // stage the symbols (and the locate marks) of the tile, bucket-major with an
  // odd word stride; positions behind the end read as letter 0 (the fill of the
  // last block, eis-blockcomp.c:587-589)
  // (eight positions per thread and step: all loads of a step are issued before
  // the first is used -- one position per step left the tile waiting for memory
  // 64 times in a row)
  constexpr int SU = 8;
  for (u32 base = 0; base < nbk * g.L; base += PCK_THREADS * SU) {
    u32 cc[SU];
    u64 vv[SU];
#pragma unroll
    for (int k = 0; k < SU; k++) {
      const u32 i = base + (u32) k * PCK_THREADS + tid;
      cc[k] = i < npos ? (u32) bwt[p0 + i] : 0u;
    }
    if (g.locint) {
#pragma unroll
      for (int k = 0; k < SU; k++) {
        const u32 i = base + (u32) k * PCK_THREADS + tid;
        vv[k] = i < npos ? suf[p0 + i] : 1u;
      }
    }
#pragma unroll
    for (int k = 0; k < SU; k++) {
      const u32 i = base + (u32) k * PCK_THREADS + tid;
      if (i >= nbk * g.L) break;
      u32 code = 0;
      if (i < npos) {
        const u64 p = p0 + i;
        const u32 c = cc[k];
        code = c >= 254 ? LDS_SPECIAL + (c - 254) : c;
        if (g.locint) {
          // addLocateInfo, eis-bwtseq-extinfo.c:420-441: every locint-th text
          // position, and the positions where letters and specials meet
          // (isSortModeTransition :343-382): the symbol before the suffix is the
          // BWT symbol, the suffix starts with a special iff its row lies in the
          // tail of the table
          const u64 v = vv[k];
          const bool hit = g.loc_pow2 ? (v & g.locmask) == 0 : (v % g.locint) == 0;
          // (-sprank: the specials are sorted reversibly, no marks where they meet letters)
          const bool tr = !g.reversible && (c >= 254) != (p >= g.first_special_row);
          if (hit || tr) code |= LDS_MARK;
        }
      }
      // i / L by multiplication (exact for i, L <= 16384: i * (L - 1) < 2^32)
      const u32 q = (u32) (((u64) i * g.inv_L) >> 32);
      s_sym[q * g.LP + (i - q * g.L)] = (u8) code;
    }
  }
  for (u32 i = tid; i < g.sigma * g.T; i += PCK_THREADS) s_cnt[i] = 0;
  __syncthreads();

  // one thread per bucket
  const bool live = tid < nbk;
  const u64 bucket = b0 + tid;
  const u64 bpos = bucket * g.L;
  const u32 len = live ? (u32) min((u64) g.L, g.N > bpos ? g.N - bpos : 0) : 0;
  const u32 nblk = (len + g.B - 1) / g.B;
  const u8 *mine = s_sym + (size_t) tid * g.LP;
  u32 pbits_sum = 0, nmarks = 0, nstart = 0, nend = 0, nranks = 0;
  if (live) {
    // symbol before / behind the bucket (region borders)
    u32 prev = 0xff, next = 0xff;
    if (bpos > 0) prev = tid > 0 ? (mine[-(int) g.LP + (int) g.L - 1] & 63u)
                                 : (bwt[bpos - 1] >= 254 ? LDS_SPECIAL + (bwt[bpos - 1] - 254) : 0u);
    if (len && bpos + len < g.N)
      next = (tid + 1 < nbk) ? (mine[g.LP] & 63u)
                             : (bwt[bpos + len] >= 254 ? LDS_SPECIAL + (bwt[bpos + len] - 254) : 0u);
    for (u32 off = 0; off < len; off++) {
      const u32 raw = mine[off];
      const u32 c = raw & 63u;
      if (c < LDS_SPECIAL) s_cnt[c * g.T + tid]++;
      else {
        const u32 before = off ? (mine[off - 1] & 63u) : prev;
        const u32 after = off + 1 < len ? (mine[off + 1] & 63u) : next;
        nstart += before != c;
        nend += after != c;
        nranks++;
      }
      nmarks += (raw & LDS_MARK) != 0;
    }
    if (g.lut_entries) {
      for (u32 b8 = 0; b8 < nblk; b8 += 8) {
        u64 e[8];
        block_entries8(g, mine, lut, b8, nblk, e);
#pragma unroll
        for (int k = 0; k < 8; k++) pbits_sum += (u32) (e[k] >> 58);
      }
    } else {
      u32 cnt_scratch[PCK_MAX_SIGMA + 2];
      for (u32 b = 0; b < nblk; b++) {
        u8 bs[16];
        for (u32 i = 0; i < g.B; i++) { const u32 c = mine[b * g.B + i] & 63u; bs[i] = (u8) (c >= LDS_SPECIAL ? 0u : c); }
        pbits_sum += (u32) (block_indices(bs, g.sigma, g.B, cnt_scratch) >> 58);
      }
    }
  }
  // bits of the bucket's var part
  u32 varbits = pbits_sum;
  if (live && g.locint) {
    if (g.loc_count) varbits += reqbits(len) + nmarks * (reqbits((u64) len - 1) + g.bits_orig_pos);
    else varbits += nmarks * g.bits_orig_pos;
    varbits += nranks * g.bits_orig_rank;
  }
  __syncthreads();

  // prefix sums over the buckets of the tile
  u32 tot;
  const u32 var_ex = block_scan_excl_sum(live ? varbits : 0, &tot, s4);
  const u32 tile_var_bits = tot;
  if (!EMIT) { if (tid == 0) tile_tot[(u64) g.sigma * g.ntiles + tile] = tot; }
  const u32 st_ex = block_scan_excl_sum(nstart, &tot, s4);
  if (!EMIT) { if (tid == 0) tile_tot[(u64) (g.sigma + 1) * g.ntiles + tile] = tot; }
  const u32 en_ex = block_scan_excl_sum(nend, &tot, s4);
  if (!EMIT) { if (tid == 0) tile_tot[(u64) (g.sigma + 2) * g.ntiles + tile] = tot; }

  const u64 cwbit = g.cw_base_bit + bucket * g.cw_bits;
  // EMIT: the tile's parts of the two bit strings are put together in LDS and
  // leave in whole lines (a tile whose part does not fit writes straight into
  // the image)
  BitSink cw_sink = { nullptr, 0, img }, var_sink = { nullptr, 0, img };
  u32 cw_nw = 0, var_nw = 0;
  if (EMIT) {
    const u64 cw_first = g.cw_base_bit + b0 * g.cw_bits, cw_end = cw_first + (u64) nbk * g.cw_bits;
    const u64 var_first = g.var_base_bit + tile_tot[(u64) g.sigma * g.ntiles + tile],
              var_end = var_first + tile_var_bits;
    const u64 cwn = ((cw_end + 63) >> 6) - (cw_first >> 6), vn = ((var_end + 63) >> 6) - (var_first >> 6);
    if (g.lds_cw_words && cwn <= g.lds_cw_words) {
      cw_sink.lds = (u64 *) (smem + g.lds_cw_off); cw_sink.w0 = cw_first >> 6; cw_nw = (u32) cwn;
    }
    if (g.lds_var_words && tile_var_bits && vn <= g.lds_var_words) {
      var_sink.lds = (u64 *) (smem + g.lds_var_off); var_sink.w0 = var_first >> 6; var_nw = (u32) vn;
    }
    for (u32 i = tid; i < cw_nw; i += PCK_THREADS) cw_sink.lds[i] = 0;
    for (u32 i = tid; i < var_nw; i += PCK_THREADS) var_sink.lds[i] = 0;
    __syncthreads();
  }
  BitWriter cw_w, var_w;
  cw_w.begin(cw_sink, cwbit);
  var_w.begin(var_sink, 0);
  for (u32 s = 0; s < g.sigma; s++) {
    const u32 ex = block_scan_excl_sum(live ? (u32) s_cnt[s * g.T + tid] : 0, &tot, s4);
    if (!EMIT) { if (tid == 0) tile_tot[(u64) s * g.ntiles + tile] = tot; }
    else if (live)   // occurrences before the bucket, updateIdxOutput eis-blockcomp.c:1847-1855
      cw_w.put_at(cwbit + g.sym_off[s], g.sym_bits[s], tile_tot[(u64) s * g.ntiles + tile] + ex);
  }
  if (!EMIT) return;
  if (live) {
  const u64 var_off = tile_tot[(u64) g.sigma * g.ntiles + tile] + var_ex;
  cw_w.put_at(cwbit + g.pre_var_idx, g.var_off_bits, var_off);
  if (bucket >= tail_first) tail_off[bucket - tail_first] = var_off;   // for the replay of the last records
  if (g.locint) cw_w.put_at(cwbit + g.pre_cb_off, g.cb_off_bits, pbits_sum);
  u64 vbit = g.var_base_bit + var_off;
  var_w.begin(var_sink, vbit);
  u64 ridx_s = tile_tot[(u64) (g.sigma + 1) * g.ntiles + tile] + st_ex;
  u64 ridx_e = tile_tot[(u64) (g.sigma + 2) * g.ntiles + tile] + en_ex;
  {
    u32 prev = 0xff, next = 0xff;
    if (bpos > 0) prev = tid > 0 ? (mine[-(int) g.LP + (int) g.L - 1] & 63u)
                                 : (bwt[bpos - 1] >= 254 ? LDS_SPECIAL + (bwt[bpos - 1] - 254) : 0u);
    if (len && bpos + len < g.N)
      next = (tid + 1 < nbk) ? (mine[g.LP] & 63u)
                             : (bwt[bpos + len] >= 254 ? LDS_SPECIAL + (bwt[bpos + len] - 254) : 0u);
    {
      if (nranks)
        // region list of the specials, gt_SRLAddPosition eis-seqranges.c:183-213
        for (u32 off = 0; off < len; off++) {
          const u32 c = mine[off] & 63u;
          if (c >= LDS_SPECIAL) {
            const u32 before = off ? (mine[off - 1] & 63u) : prev;
            const u32 after = off + 1 < len ? (mine[off + 1] & 63u) : next;
            if (before != c) rstart[ridx_s++] = (bpos + off) | ((u64) (c - LDS_SPECIAL) << 63);
            if (after != c) rend[ridx_e++] = bpos + off + 1;
          }
        }
      // append2IdxOutput, eis-blockcomp.c:1762-1775
      if (g.lut_entries) {
        for (u32 b8 = 0; b8 < nblk; b8 += 8) {
          u64 e[8];
          block_entries8(g, mine, lut, b8, nblk, e);
#pragma unroll
          for (int k = 0; k < 8; k++)
            if (b8 + k < nblk) {
              cw_w.put_at(cwbit + g.pre_comp_idx + (b8 + k) * g.comp_idx_bits, g.comp_idx_bits,
                          (e[k] >> 40) & 0x3ffffu);
              const u32 pb = (u32) (e[k] >> 58);
              var_w.put(pb, e[k] & 0xffffffffffull);
              vbit += pb;
            }
        }
      } else {
        u32 cnt_scratch[PCK_MAX_SIGMA + 2];
        for (u32 b = 0; b < nblk; b++) {
          u8 bs[16];
          for (u32 i = 0; i < g.B; i++) { const u32 c = mine[b * g.B + i] & 63u; bs[i] = (u8) (c >= LDS_SPECIAL ? 0u : c); }
          const u64 e = block_indices(bs, g.sigma, g.B, cnt_scratch);
          cw_w.put_at(cwbit + g.pre_comp_idx + b * g.comp_idx_bits, g.comp_idx_bits, (e >> 40) & 0x3ffffu);
          const u32 pb = (u32) (e >> 58);
          var_w.put(pb, e & 0xffffffffffull);
          vbit += pb;
        }
      }
    }
  }
  if (g.locint) {
    // addLocateInfo, eis-bwtseq-extinfo.c:384-541
    const u32 bits_bwt_pos = reqbits((u64) len - 1);
    if (g.loc_count) { const u32 bc = reqbits(len); var_w.put(bc, nmarks); vbit += bc; }
    for (u32 o0 = 0; o0 < len; o0 += 64) {
      const u32 m = min(64u, len - o0);
      u64 marks = 0;
      for (u32 i = 0; i < m; i++) marks |= (u64) ((mine[o0 + i] & LDS_MARK) ? 1u : 0u) << i;
      if (g.loc_bitmap) cw_w.put_at(cwbit + g.pre_cw_ext + o0, m, __brevll(marks) >> (64 - m));
      // the text positions of the marked rows, eight loads at a time
      while (marks) {
        u32 off[8];
        u64 v[8];
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          off[k] = 0;
          if (marks) { off[k] = o0 + (u32) __builtin_ctzll(marks); marks &= marks - 1; cnt = k + 1; }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = k < cnt ? suf[bpos + off[k]] : 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (k < cnt) {
            if (g.loc_count) { var_w.put(bits_bwt_pos, off[k]); vbit += bits_bwt_pos; }
            u64 x = v[k];
            if (g.reversible) x = g.loc_pow2 ? x >> __popc(g.locmask) : x / g.locint;
            var_w.put(g.bits_orig_pos, x);
            vbit += g.bits_orig_pos;
          }
      }
    }
    if (g.bits_orig_rank && nranks)
      // the symbols sorted by rank (specials, the undefined symbol before suffix 0):
      // their rank among the specials of the text, eis-bwtseq-extinfo.c:452-471, 528-541
      for (u32 o = 0; o < len; o++)
        if ((mine[o] & 63u) >= LDS_SPECIAL) {
          const u64 v = suf[bpos + o];
          var_w.put(g.bits_orig_rank, v ? special_rank(spbits, sppre, v - 1) : g.total_specials);
          vbit += g.bits_orig_rank;
        }
  }
  cw_w.finish();
  var_w.finish();
  }  // live
  __syncthreads();
  sink_flush(cw_sink, cw_nw);
  sink_flush(var_sink, var_nw);
}

// exclusive prefix sums of every column of tile_tot; totals[c] = column sum
__global__ __launch_bounds__(PCK_THREADS) void k_pck_scan_cols(u64 *tile_tot, u32 ntiles, u64 *totals) {
  __shared__ u64 s_w[4];
  u64 *col = tile_tot + (u64) blockIdx.x * ntiles;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u64 carry = 0;
  for (u32 base = 0; base < ntiles; base += PCK_THREADS) {
    const u32 i = base + threadIdx.x;
    const u64 v = i < ntiles ? col[i] : 0;
    u64 inc = v;
    for (int d = 1; d < 64; d <<= 1) {
      const u64 o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    u64 before = 0, all = 0;
    for (int k = 0; k < 4; k++) { if (k < w) before += s_w[k]; all += s_w[k]; }
    if (i < ntiles) col[i] = carry + before + inc - v;
    carry += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// region records, struct seqRange eis-seqranges-priv.h:25-63: start (uint64),
// then 1 symbol bit and 63 length bits, most significant first
__global__ void k_pck_regions(const u64 *rstart, const u64 *rend, u64 nregions, u64 N, u32 B,
                              u8 *dst) {
  const u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nregions) return;
  u64 start, len, sym;
  if (i < nregions) {
    start = rstart[i] & ~(1ull << 63);
    sym = rstart[i] >> 63;
    len = rend[i] - start;
  } else {          // terminator just beyond the sequence, eis-blockcomp.c:2461-2464
    start = N + B; len = 1; sym = 0;
  }
  u8 *r = dst + 16 * i;
  const u64 be = (sym << 63) | len;
  for (int k = 0; k < 8; k++) {
    r[k] = (u8) (start >> (8 * k));
    r[8 + k] = (u8) (be >> (8 * (7 - k)));
  }
}

// context map, gt_BWTSCRFMapAdvance src/match/eis-bwtseq-context.c:158-177: the row
// of the suffix behind every 2^ilog-th text position, uniform entries behind a
// 32-bit header
__global__ void k_pck_ctxmap(const u64 *__restrict__ suf, u64 N, u32 ilog, u32 bits, u64 *img) {
  const u64 mask = (1ull << ilog) - 1;
  for (u64 r = (u64) blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (u64) gridDim.x * blockDim.x) {
    const u64 v = suf[r];
    const u64 op = v ? v - 1 : N - 1;
    if ((op & mask) == 0) put_bits(img, 32 + (op >> ilog) * bits, bits, r);
  }
}

}  // namespace

struct gtamd_pck {
  int device;
  hipStream_t st;
  hipEvent_t ev0, ev1;
  u64 *lut; u64 lut_entries; u32 lut_sigma, lut_B;
  u8 *img; u64 img_cap;
  u64 *tile_tot; u64 tile_tot_cap;
  u64 *rlist; u64 rlist_cap;
  u64 *d_totals;
  u64 *d_tail;
  u64 tail_cap;          // buckets d_tail has room for
  u64 *spbits; u64 spbits_cap;     // -sprank: bitmap of the text's specials, word prefix counts,
  u32 *sppre; u64 sppre_cap;       // scan workspace
  u32 *spws; u64 spws_cap;
  u8 *cxm; u64 cxm_cap, cxm_bytes;   // image of INDEX.<ilog>cxm
  gtamd_pck_info info;
  bool built;
};

extern "C" int gtamd_pck_default_toggles(uint32_t block_size, uint32_t bucket_blocks,
                                         uint32_t locate_interval, int locbitmap) {
  GTAMD_ABI_BEGIN
  // gt_computePackedIndexDefaults / estimateBestLocateTypeFeature,
  // src/match/eis-bwtseq-param.c:69-103
  if (locbitmap >= 0) return locbitmap ? GTAMD_PCK_LOCATE_BITMAP : GTAMD_PCK_LOCATE_COUNT;
  if (!locate_interval) return 0;
  const u32 seg = block_size * bucket_blocks;
  if (seg > (seg + 1) * reqbits(seg) / locate_interval) return GTAMD_PCK_LOCATE_COUNT;
  return GTAMD_PCK_LOCATE_BITMAP;
  GTAMD_ABI_END(-1)
}

extern "C" gtamd_pck *gtamd_pck_create(int device) {
  GTAMD_ABI_BEGIN
  if (gtamd_device_count() <= device || device < 0) {
    gtamd_set_error("no HIP device %d available (this library has no CPU fallback)", device);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) { gtamd_set_error("hipSetDevice(%d) failed", device); return nullptr; }
  gtamd_pck *p = new gtamd_pck();
  memset(p, 0, sizeof *p);
  p->device = device;
  if (hipStreamCreate(&p->st) != hipSuccess || hipEventCreate(&p->ev0) != hipSuccess ||
      hipEventCreate(&p->ev1) != hipSuccess ||
      hipMalloc(&p->d_totals, (PCK_MAX_SIGMA + 8) * sizeof(u64)) != hipSuccess ||
      hipMalloc(&p->d_tail, PCK_TAIL_RECORDS * sizeof(u64)) != hipSuccess) {
    gtamd_set_error("cannot create the packed-index builder on device %d", device);
    delete p;
    return nullptr;
  }
  p->tail_cap = PCK_TAIL_RECORDS;
  return p;
  GTAMD_ABI_END(nullptr)
}

extern "C" void gtamd_pck_destroy(gtamd_pck *p) {
  if (p == nullptr) return;
  (void) hipSetDevice(p->device);
  (void) hipStreamSynchronize(p->st);
  if (p->lut) (void) hipFree(p->lut);
  if (p->img) (void) hipFree(p->img);
  if (p->tile_tot) (void) hipFree(p->tile_tot);
  if (p->rlist) (void) hipFree(p->rlist);
  if (p->d_totals) (void) hipFree(p->d_totals);
  if (p->d_tail) (void) hipFree(p->d_tail);
  if (p->spbits) (void) hipFree(p->spbits);
  if (p->sppre) (void) hipFree(p->sppre);
  if (p->spws) (void) hipFree(p->spws);
  if (p->cxm) (void) hipFree(p->cxm);
  (void) hipEventDestroy(p->ev0);
  (void) hipEventDestroy(p->ev1);
  (void) hipStreamDestroy(p->st);
  delete p;
}

template <typename T> static int grow(T **buf, u64 *cap, u64 need_bytes) {
  if (*cap >= need_bytes && *buf != nullptr) return 0;
  if (*buf) (void) hipFree(*buf);
  *buf = nullptr; *cap = 0;
  if (hipMalloc((void **) buf, need_bytes) != hipSuccess) {
    gtamd_set_error("cannot allocate %llu bytes of device memory for the packed index",
                    (unsigned long long) need_bytes);
    return -1;
  }
  *cap = need_bytes;
  return 0;
}

// ---- the bits the reference leaves stale in the last bucket: esa_pck_replay.h --
static int img_read(void *user, uint64_t offset, uint64_t count, uint8_t *dst) {
  gtamd_pck *p = (gtamd_pck *) user;
  return hipMemcpy(dst, p->img + offset, count, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
static int img_write(void *user, uint64_t offset, uint64_t count, const uint8_t *src) {
  gtamd_pck *p = (gtamd_pck *) user;
  return hipMemcpy(p->img + offset, src, count, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
static int fix_stale_bits(gtamd_pck *p, const PckGeom &g, u64 var_bits_total,
                          const std::vector<u64> &tail_off) {
  PckTailGeom t;
  t.N = g.N; t.nb = g.nb; t.L = g.L; t.B = g.B; t.locint = g.locint; t.loc_bitmap = g.loc_bitmap;
  t.cw_bits = g.cw_bits; t.pre_comp_idx = g.pre_comp_idx; t.pre_cw_ext = g.pre_cw_ext;
  t.comp_idx_bits = g.comp_idx_bits;
  t.cw_data_pos = g.cw_base_bit / 8; t.var_data_pos = g.var_base_bit / 8;
  const int rc = pck_fix_stale_bits(t, var_bits_total, tail_off, img_read, img_write, p);
  if (rc == -1) gtamd_set_error("packed index: cannot copy the tail of the image between device and host");
  if (rc == -2) gtamd_set_error("packed index: the replay of the staging buffers needs more than the last %llu buckets",
                                (unsigned long long) tail_off.size());
  if (rc == -3) gtamd_set_error("packed index: the var offsets of the last %llu buckets read back from the device "
                                "are not increasing or exceed the %llu bits of the var part",
                                (unsigned long long) tail_off.size(), (unsigned long long) var_bits_total);
  return rc;
}

extern "C" int gtamd_pck_build(gtamd_pck *p, const uint8_t *bwt, const uint64_t *suf,
                               uint64_t total_len, uint32_t sigma, uint64_t longest,
                               const gtamd_pck_params *pp) {
  GTAMD_ABI_BEGIN
  if (p == nullptr || bwt == nullptr || pp == nullptr) { gtamd_set_error("invalid argument to gtamd_pck_build"); return -1; }
  const u32 B = pp->block_size, K = pp->bucket_blocks, locint = pp->locate_interval;
  const bool loc_bitmap = locint && (pp->feature_toggles & GTAMD_PCK_LOCATE_BITMAP);
  const bool loc_count = locint && !loc_bitmap && (pp->feature_toggles & GTAMD_PCK_LOCATE_COUNT);
  const bool reversible = locint && (pp->feature_toggles & GTAMD_PCK_REVERSIBLY_SORTED);
  if (B < 1 || B > 16 || K < 1 || (u64) B * K > PCK_TILE_POS || sigma < 2 || sigma > PCK_MAX_SIGMA) {
    gtamd_set_error("packed index: block size %u x %u blocks per bucket over %u letters is outside "
                    "what the device builder supports (block size <= 16, bucket <= %u positions)",
                    B, K, sigma, PCK_TILE_POS);
    return -1;
  }
  if (locint && !loc_bitmap && !loc_count) { gtamd_set_error("packed index: locate information wanted but neither bitmap nor count mode chosen"); return -1; }
  if (locint && suf == nullptr) { gtamd_set_error("packed index: locate information needs the suffix array"); return -1; }
  if (total_len < 2) { gtamd_set_error("packed index: empty sequence"); return -1; }
  if (total_len >= (1ull << 40)) { gtamd_set_error("packed index: more than 2^40 positions"); return -1; }
  if (pp->feature_toggles & ~(GTAMD_PCK_LOCATE_BITMAP | GTAMD_PCK_LOCATE_COUNT | GTAMD_PCK_REVERSIBLY_SORTED)) {
    gtamd_set_error("packed index: feature toggles %d not supported", (int) pp->feature_toggles);
    return -1;
  }
  if (reversible && suf == nullptr) { gtamd_set_error("packed index: -sprank needs the suffix array"); return -1; }
  HIP_TRY(hipSetDevice(p->device));
  p->built = false;

  PckGeom g;
  memset(&g, 0, sizeof g);
  g.N = total_len; g.sigma = sigma; g.B = B; g.K = K; g.L = B * K;
  g.LP = 4 * (2 * ((g.L + 7) / 8) + 1);
  g.inv_L = ((1ull << 32) + g.L - 1) / g.L;
  g.nb = (total_len + 1) / g.L + (((total_len + 1) % g.L) ? 1 : 0);     // numBuckets, eis-blockcomp.c:1633-1638
  g.T = std::max<u32>(1, std::min<u32>(PCK_THREADS, PCK_TILE_POS / g.L));
  g.ntiles = (u32) div_up(g.nb, g.T);
  g.locint = locint; g.loc_bitmap = loc_bitmap; g.loc_count = loc_count;
  g.loc_pow2 = locint && (locint & (locint - 1)) == 0; g.locmask = locint ? locint - 1 : 0;
  // widths the counting pass needs: gt_newGenBlockEncIdxSeq eis-blockcomp.c:336-339,
  // 477-501; initAddLocateInfoState eis-bwtseq-extinfo.c:253-337
  const u32 bits_per_ulong = reqbits(total_len - 1);
  g.comp_idx_bits = reqbits(binom(B + sigma - 1, sigma - 1) - 1);
  u32 even[PCK_MAX_SIGMA + 2];
  for (u32 s = 0; s < sigma; s++) even[s] = B / sigma + (s < B % sigma ? 1u : 0u);
  const u64 max_perms = arrangements(even, sigma, B);
  const u32 max_perm_idx_bits = reqbits(max_perms - 1);
  if (g.comp_idx_bits > 18 || max_perm_idx_bits > 40) {
    gtamd_set_error("packed index: block size %u over %u letters needs wider indices than the device builder packs", B, sigma);
    return -1;
  }
  const u64 cw_ext_bits = loc_bitmap ? g.L : 0;
  g.cb_off_bits = locint ? reqbits((u64) max_perm_idx_bits * K) : 0;
  // initAddLocateInfoState, eis-bwtseq-extinfo.c:279-285
  if (locint) g.bits_orig_pos = reversible ? reqbits((total_len - 1) / locint) : reqbits(total_len - 1);
  g.reversible = reversible;

  // block -> index pair table
  u64 entries = 1;
  for (u32 i = 0; i < B && entries <= (1ull << 24); i++) entries *= sigma;
  if (entries > (1ull << 24)) entries = 0;
  if (entries && (p->lut == nullptr || p->lut_sigma != sigma || p->lut_B != B)) {
    if (p->lut) { (void) hipFree(p->lut); p->lut = nullptr; }
    if (hipMalloc(&p->lut, entries * sizeof(u64)) != hipSuccess) { gtamd_set_error("packed index: cannot allocate the block table"); return -1; }
    k_pck_lut<<<(u32) div_up(entries, 256), 256, 0, p->st>>>(entries, sigma, B, p->lut);
    HIP_TRY(hipGetLastError());
    p->lut_entries = entries; p->lut_sigma = sigma; p->lut_B = B;
  }
  g.lut_entries = entries;

  const u64 ncols = sigma + PCK_EXTRA_COLS;
  u64 tt_cap_bytes = p->tile_tot_cap;
  TRY(grow(&p->tile_tot, &tt_cap_bytes, ncols * g.ntiles * sizeof(u64)));
  p->tile_tot_cap = tt_cap_bytes;

  HIP_TRY(hipEventRecord(p->ev0, p->st));
  // rows from first_special_row on hold the suffixes that start with a special:
  // as many as the BWT holds specials (every special position p is the symbol
  // before suffix p + 1; the undefined symbol before suffix 0 stands for the
  // virtual end)
  HIP_TRY(hipMemsetAsync(p->d_totals, 0, (PCK_MAX_SIGMA + 8) * sizeof(u64), p->st));
  k_pck_count_specials<<<1024, 256, 0, p->st>>>(bwt, total_len, (unsigned long long *) p->d_totals);
  HIP_TRY(hipGetLastError());
  u64 nspecial = 0;
  HIP_TRY(hipStreamSynchronize(p->st));
  HIP_TRY(hipMemcpy(&nspecial, p->d_totals, 8, hipMemcpyDeviceToHost));
  // (numbers read back from the device are checked before they size or index
  // anything on the host: a bogus one ends in -1 + message, not in a container
  // of 2^64 entries)
  if (nspecial == 0 || nspecial > total_len) {
    gtamd_set_error("packed index: %llu special symbols counted in a BWT of %llu entries",
                    (unsigned long long) nspecial, (unsigned long long) total_len);
    return -1;
  }
  g.first_special_row = total_len - nspecial;
  if (reversible) {
    // buildSpRTable / gt_createBWTSeqGeneric, eis-bwtseq-construct.c:206-229,
    // eis-bwtseq-extinfo.c:585-600: one rank per special of the text (the BWT holds
    // them all, plus the undefined symbol before suffix 0)
    g.total_specials = nspecial - 1;
    if (g.total_specials >= (1ull << 32)) { gtamd_set_error("packed index: -sprank with more than 2^32 specials"); return -1; }
    g.bits_orig_rank = reqbits(g.total_specials);
    const u64 nwords = total_len / 64 + 1;
    u64 cap = p->spbits_cap; TRY(grow(&p->spbits, &cap, nwords * 8)); p->spbits_cap = cap;
    cap = p->sppre_cap; TRY(grow(&p->sppre, &cap, nwords * 4)); p->sppre_cap = cap;
    cap = p->spws_cap; TRY(grow(&p->spws, &cap, scan_workspace_words(nwords) * 4 + 64)); p->spws_cap = cap;
    HIP_TRY(hipMemsetAsync(p->spbits, 0, nwords * 8, p->st));
    k_pck_special_bitmap<<<2048, 256, 0, p->st>>>(bwt, suf, total_len, (unsigned long long *) p->spbits);
    HIP_TRY(hipGetLastError());
    k_pck_popc_words<<<(u32) div_up(nwords, 256), 256, 0, p->st>>>(p->spbits, nwords, p->sppre);
    HIP_TRY(hipGetLastError());
    TRY(scan_u32(SCAN_SUM, p->sppre, p->sppre, nwords, false, p->spws, p->st));
  }

  const size_t lds = (((size_t) g.T * g.LP + 15) & ~(size_t) 15) + (size_t) sigma * g.T * sizeof(u16);
  k_pck_tile<false><<<g.ntiles, PCK_THREADS, lds, p->st>>>(g, bwt, suf, p->lut, p->tile_tot, nullptr, nullptr, nullptr,
                                                          nullptr, 0, nullptr, nullptr);
  HIP_TRY(hipGetLastError());
  k_pck_scan_cols<<<(u32) ncols, PCK_THREADS, 0, p->st>>>(p->tile_tot, g.ntiles, p->d_totals);
  HIP_TRY(hipGetLastError());
  u64 totals[PCK_MAX_SIGMA + 8];
  HIP_TRY(hipStreamSynchronize(p->st));
  HIP_TRY(hipMemcpy(totals, p->d_totals, ncols * sizeof(u64), hipMemcpyDeviceToHost));
  const u64 var_bits_total = totals[sigma], nregions = totals[sigma + 1];
  if (totals[sigma + 2] != nregions) { gtamd_set_error("packed index: region starts and ends disagree"); return -1; }
  if (nregions > total_len + 1 || var_bits_total > (total_len + g.L) * PCK_REPLAY_MAX_BITS_PER_POSITION) {
    gtamd_set_error("packed index: counting pass returned %llu regions and %llu var bits for %llu positions",
                    (unsigned long long) nregions, (unsigned long long) var_bits_total,
                    (unsigned long long) total_len);
    return -1;
  }

  // widths of the occurrence counters: symSumBitsDefaultSetup eis-blockcomp.c:757-774
  // without sequence statistics (trsuftab); with them (mkindex) as many bits as the
  // occurrences of each letter need, eis-blockcomp.c:385-437 -- the counting pass
  // has just counted them
  u64 regular = 0;
  u32 off = 0;
  for (u32 s = 0; s < sigma; s++) {
    regular += totals[s];
    g.sym_bits[s] = pp->with_statistics ? reqbits(totals[s]) : bits_per_ulong;
    g.sym_off[s] = off;
    off += g.sym_bits[s];
  }
  const u32 sym_sum_bits = off;
  // vwBits eis-blockcomp.c:1659-1689, locBitsUpperBounds eis-bwtseq-extinfo.c:195-251
  u64 max_var_bits_total = g.nb * ((u64) max_perm_idx_bits * K), max_var_ext_bits_per_bucket = 0;
  if (locint) {
    const u64 last_pos = total_len - 1;
    u64 extra = 0;
    if (locint > 1 && !reversible) {
      extra = std::min(total_len / 2, total_len - total_len / locint);
      if (pp->with_statistics) {
        // symbols outside the value-sorted range as newSeqStatsFromCharDist counts
        // them (eis-suffixerator-interface.c:176-206, eis-bwtseq-extinfo.c:302-314)
        const u64 nonval = total_len - regular + 1;
        extra = std::min(extra, std::min(nonval, total_len - nonval));
      }
    }
    const u64 dlen[2] = { g.L, total_len % g.L };
    const u64 drep[2] = { (total_len + 1) / g.L, ((total_len + 1) % g.L) ? 1ull : 0ull };
    u64 max_seg = 0, tot = 0;
    for (int i = 0; i < 2; i++) { max_seg = std::max(max_seg, dlen[i]); if (loc_count) tot += reqbits(dlen[i]) * drep[i]; }
    tot += (total_len / locint + extra) * ((loc_count ? reqbits(max_seg) : 0) + g.bits_orig_pos);
    if (g.bits_orig_rank) {
      // specialsRank(seqLen): the specials and the terminator -- which the reference's
      // sample table counts twice when seqLen falls on a sample position
      // (eis-specialsrank.c:108-128, 160-190; interval 2^bits(bits(seqLen)),
      // eis-bwtseq-construct.c:217-222)
      u64 bound = g.total_specials + 1;
      if (g.total_specials && total_len % (1ull << reqbits(reqbits(total_len))) == 0) bound++;
      tot += bound * g.bits_orig_rank;
    }
    max_var_ext_bits_per_bucket = max_seg * ((loc_count ? reqbits(last_pos) : 0) + g.bits_orig_pos + g.bits_orig_rank)
                                  + (loc_count ? reqbits(max_seg) : 0);
    max_var_bits_total += tot;
  }
  g.var_off_bits = reqbits(max_var_bits_total);
  g.pre_var_idx = sym_sum_bits;
  g.pre_cb_off = g.pre_var_idx + g.var_off_bits;
  g.pre_comp_idx = g.pre_cb_off + g.cb_off_bits;
  g.pre_cw_ext = g.pre_comp_idx + g.comp_idx_bits * K;
  g.cw_bits = g.pre_cw_ext + (u32) cw_ext_bits;
  // header, blockEncIdxSeqHeaderLength eis-blockcomp.c:1919-1946
  const u32 num_modes = 2;
  u64 header_len = 4 + 4 + 8 + 8 + 12 + 12 + 8 + 8 + 8 + 4 * sigma + 8 + 8 + 8 + 12 + 4 * num_modes;
  if (g.cb_off_bits) header_len += 8 + 12 + 12;
  const u64 cw_data_pos = div_up(header_len + (locint ? 8 + 16 : 0) + (g.bits_orig_rank ? 8 + 8 : 0), 8192) * 8192;
  const u64 cw_len = ((u64) g.cw_bits * g.nb + 7) / 8;
  const u64 var_data_pos = cw_data_pos + cw_len;
  g.cw_base_bit = cw_data_pos * 8;
  g.var_base_bit = var_data_pos * 8;
  const u64 range_enc_pos = var_data_pos + var_bits_total / 8 + ((var_bits_total % 8) ? 1 : 0);
  const u64 file_bytes = range_enc_pos + 8 + 16 * (nregions + 1);

  u64 img_cap = p->img_cap;
  TRY(grow(&p->img, &img_cap, ((file_bytes + 7) & ~7ull) + 64));
  p->img_cap = img_cap;
  u64 rl_cap = p->rlist_cap;
  TRY(grow(&p->rlist, &rl_cap, std::max<u64>(1, nregions) * 2 * sizeof(u64)));
  p->rlist_cap = rl_cap;
  HIP_TRY(hipMemsetAsync(p->img, 0, ((file_bytes + 7) & ~7ull) + 64, p->st));
  u64 tail_n = std::min<u64>(g.nb, std::max<u64>(PCK_TAIL_RECORDS, p->tail_cap));
  // LDS copies of a tile's parts of the bit strings: the whole cw part, and as
  // much of the 64 KB as is left (at most 24 KB) for the var part -- a tile with
  // more var bits writes them straight into the image, and so does every tile
  // when the cw part alone does not fit
  size_t lds_emit = (lds + 15) & ~(size_t) 15;
  {
    const u64 cw_words = ((u64) g.T * g.cw_bits + 63) / 64 + 2;
    if (lds_emit + cw_words * 8 + 1024 * 8 <= 65536 && getenv("GTAMD_PCK_DIRECT") == nullptr) {
      const u64 var_words = std::min<u64>(3072, (65536 - lds_emit - cw_words * 8) / 8);
      g.lds_cw_off = (u32) lds_emit; g.lds_cw_words = (u32) cw_words; lds_emit += cw_words * 8;
      g.lds_var_off = (u32) lds_emit; g.lds_var_words = (u32) var_words; lds_emit += var_words * 8;
    }
  }
  k_pck_tile<true><<<g.ntiles, PCK_THREADS, lds_emit, p->st>>>(g, bwt, suf, p->lut, p->tile_tot, (u64 *) p->img,
                                                         p->rlist, p->rlist + std::max<u64>(1, nregions),
                                                         p->d_tail, g.nb - tail_n, p->spbits, p->sppre);
  HIP_TRY(hipGetLastError());
  k_pck_regions<<<(u32) div_up(nregions + 1, 256), 256, 0, p->st>>>(
      p->rlist, p->rlist + std::max<u64>(1, nregions), nregions, total_len, B, p->img + range_enc_pos + 8);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(p->ev1, p->st));
  HIP_TRY(hipStreamSynchronize(p->st));

  // header + extension header, writeIdxHeader eis-blockcomp.c:1984-2094,
  // writeLocateInfoHeader eis-bwtseq-extinfo.c:59-76
  {
    std::vector<u8> h((size_t) header_len + 128, 0);
    u64 o = 8;
    auto put32 = [&](u64 v) { const u32 x = (u32) v; memcpy(&h[(size_t) o], &x, 4); o += 4; };
    auto put64 = [&](u64 v) { memcpy(&h[(size_t) o], &v, 8); o += 8; };
    memcpy(&h[0], "BDX", 4);
    { const u32 x = (u32) (div_up(header_len, 8192) * 8192); memcpy(&h[4], &x, 4); }
    put32(0x424b535a); put32(B);
    put32(0x42424c4b); put32(K);
    put32(0x564f4646); put64(var_data_pos);
    put32(0x524f4646); put64(range_enc_pos);
    put32(0x53454c45); put64(total_len);
    put32(0x53504254); put32(bits_per_ulong);
    put32(0x56444f42); put32(g.var_off_bits);
    put32(0x53534254); put32(sigma);
    for (u32 s = 0; s < sigma; s++) put32(g.sym_bits[s]);
    put32(0x42454642); put32(0);
    put32(0x52454642); put32(0);
    put32(0x4e4d524e); put32(num_modes);
    put32(1); put32(2);        // BLOCK_COMPOSITION_INCLUDE, REGIONS_LIST
    if (g.cb_off_bits) {
      put32(0x43424d42); put32(g.cb_off_bits);
      put32(0x43455842); put64(cw_ext_bits);
      put32(0x4d455842); put64(max_var_ext_bits_per_bucket);
    }
    if (o != header_len) { gtamd_set_error("packed index: header length mismatch"); return -1; }
    if (locint) {
      put32(0x45480000u | 1111u); put32(16);
      put64(longest); put32(locint); put32((u32) pp->feature_toggles);
      if (g.bits_orig_rank) {
        // writeRankSortHeader, eis-bwtseq-extinfo.c:106-122: SORTMODE_VALUE 0, SORTMODE_RANK 2
        put32(0x45480000u | 1112u); put32(8);
        put32(g.bits_orig_rank);
        const uint16_t m0 = 0, m2 = 2;
        memcpy(&h[(size_t) o], &m0, 2); o += 2;
        memcpy(&h[(size_t) o], &m2, 2); o += 2;
      }
    }
    HIP_TRY(hipMemcpy(p->img, h.data(), (size_t) o, hipMemcpyHostToDevice));
    const u64 nr = nregions + 1;
    HIP_TRY(hipMemcpy(p->img + range_enc_pos, &nr, 8, hipMemcpyHostToDevice));
  }
  for (;;) {
    std::vector<u64> tail_off((size_t) tail_n);
    HIP_TRY(hipMemcpy(tail_off.data(), p->d_tail, tail_n * sizeof(u64), hipMemcpyDeviceToHost));
    const int frc = fix_stale_bits(p, g, var_bits_total, tail_off);
    if (frc == 0) break;
    if (frc != -2 || tail_n >= g.nb) return -1;
    // the replay wants var offsets of more buckets than were kept: keep 16 times
    // as many and emit again (the emission ORs the same bits into the same places
    // and the replay has not patched anything yet, so the image is unchanged)
    tail_n = std::min<u64>(g.nb, tail_n * 16);
    if (tail_n > p->tail_cap) {
      (void) hipFree(p->d_tail);
      p->d_tail = nullptr; p->tail_cap = 0;
      if (hipMalloc(&p->d_tail, tail_n * sizeof(u64)) != hipSuccess) {
        gtamd_set_error("packed index: cannot allocate the var offsets of %llu buckets", (unsigned long long) tail_n);
        return -1;
      }
      p->tail_cap = tail_n;
    }
    k_pck_tile<true><<<g.ntiles, PCK_THREADS, lds_emit, p->st>>>(g, bwt, suf, p->lut, p->tile_tot, (u64 *) p->img,
                                                           p->rlist, p->rlist + std::max<u64>(1, nregions),
                                                           p->d_tail, g.nb - tail_n, p->spbits, p->sppre);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(p->st));
  }

  memset(&p->info, 0, sizeof p->info);
  p->info.file_bytes = file_bytes; p->info.cw_data_pos = cw_data_pos; p->info.var_data_pos = var_data_pos;
  p->info.range_enc_pos = range_enc_pos; p->info.num_buckets = g.nb; p->info.num_regions = nregions + 1;
  p->info.var_bits = var_bits_total; p->info.cw_bits = g.cw_bits;
  HIP_TRY(hipEventElapsedTime(&p->info.build_ms, p->ev0, p->ev1));
  p->built = true;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_pck_build_from_esa(gtamd_pck *p, const gtamd_esa_ctx *esa,
                                        const gtamd_pck_params *pp) {
  GTAMD_ABI_BEGIN
  if (p == nullptr || esa == nullptr || pp == nullptr) { gtamd_set_error("invalid argument to gtamd_pck_build_from_esa"); return -1; }
  int device = 0; u32 sigma = 0, numparts = 0;
  TRY(gtamd_esa_internal_info(esa, &device, &sigma, &numparts));
  if (numparts != 1) { gtamd_set_error("packed index: needs the tables of a whole-table build"); return -1; }
  if (device != p->device) { gtamd_set_error("packed index: the tables live on device %d, the builder on %d", device, p->device); return -1; }
  const u8 *bwt = (const u8 *) gtamd_esa_table_device(esa, GTAMD_TAB_BWT);
  const u64 *suf = (const u64 *) gtamd_esa_table_device(esa, GTAMD_TAB_SUF);
  gtamd_esa_stats st;
  TRY(gtamd_esa_get_stats(esa, &st));
  if (bwt == nullptr || (suf == nullptr && pp->locate_interval)) {
    gtamd_set_error("packed index: the last run did not produce the .bwt%s table", suf == nullptr ? " / .suf" : "");
    return -1;
  }
  return gtamd_pck_build(p, bwt, suf, st.numberofallsortedsuffixes, sigma, st.longest, pp);
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_pck_build_host(gtamd_pck *p, const uint8_t *bwt, const uint64_t *suf,
                                    uint64_t total_len, uint32_t sigma, uint64_t longest,
                                    const gtamd_pck_params *pp) {
  GTAMD_ABI_BEGIN
  if (p == nullptr || bwt == nullptr || pp == nullptr) { gtamd_set_error("invalid argument to gtamd_pck_build_host"); return -1; }
  HIP_TRY(hipSetDevice(p->device));
  u8 *d_bwt = nullptr;
  u64 *d_suf = nullptr;
  int rc = -1;
  if (hipMalloc(&d_bwt, total_len) != hipSuccess ||
      (suf != nullptr && hipMalloc(&d_suf, total_len * sizeof(u64)) != hipSuccess)) {
    gtamd_set_error("packed index: cannot allocate device memory for the tables of %llu entries",
                    (unsigned long long) total_len);
  } else if (hipMemcpy(d_bwt, bwt, total_len, hipMemcpyHostToDevice) != hipSuccess ||
             (suf != nullptr &&
              hipMemcpy(d_suf, suf, total_len * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess)) {
    gtamd_set_error("packed index: cannot copy the tables to the device");
  } else
    rc = gtamd_pck_build(p, d_bwt, d_suf, total_len, sigma, longest, pp);
  if (d_bwt) (void) hipFree(d_bwt);
  if (d_suf) (void) hipFree(d_suf);
  return rc;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_pck_get_info(const gtamd_pck *p, gtamd_pck_info *info) {
  GTAMD_ABI_BEGIN
  if (p == nullptr || !p->built || info == nullptr) { gtamd_set_error("no packed index built"); return -1; }
  *info = p->info;
  return 0;
  GTAMD_ABI_END(-1)
}
extern "C" const void *gtamd_pck_image_device(const gtamd_pck *p) {
  GTAMD_ABI_BEGIN
  return (p != nullptr && p->built) ? p->img : nullptr;
  GTAMD_ABI_END(nullptr)
}
extern "C" int gtamd_pck_image_copy(gtamd_pck *p, void *dst, uint64_t offset, uint64_t count) {
  GTAMD_ABI_BEGIN
  if (p == nullptr || !p->built) { gtamd_set_error("no packed index built"); return -1; }
  if (count == 0) return 0;
  if (dst == nullptr || offset > p->info.file_bytes || count > p->info.file_bytes - offset) {
    gtamd_set_error("packed index: range [%llu,+%llu) outside the %llu bytes of the image",
                    (unsigned long long) offset, (unsigned long long) count,
                    (unsigned long long) p->info.file_bytes);
    return -1;
  }
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipMemcpy(dst, p->img + offset, count, hipMemcpyDeviceToHost));
  return 0;
  GTAMD_ABI_END(-1)
}

// ---- the context map (-ctxilog, `gt packedindex mkctxmap`) ------------------------
extern "C" int gtamd_pck_ctxmap_build(gtamd_pck *p, const uint64_t *suf, uint64_t total_len,
                                      int ilog, int *ilog_used) {
  GTAMD_ABI_BEGIN
  if (p == nullptr || suf == nullptr || total_len < 2) { gtamd_set_error("invalid argument to gtamd_pck_ctxmap_build"); return -1; }
  // CTX_MAP_ILOG_AUTOSIZE, initBWTSeqContextRetrieverFactory eis-bwtseq-context.c:65-68
  if (ilog < 0) ilog = (int) reqbits(reqbits(total_len));
  // ctxMapILogIsValid, eis-bwtseq-context-param.h:36-45
  if ((u32) ilog >= reqbits(total_len) || ilog > 62) {
    gtamd_set_error("context map: interval 2^%d is not smaller than the sequence", ilog);
    return -1;
  }
  HIP_TRY(hipSetDevice(p->device));
  const u32 bits = reqbits(total_len - 1);
  const u64 nentries = (total_len + (1ull << ilog) - 1) >> ilog;
  const u64 size = 4 + (bits * nentries + 7) / 8;
  u64 cap = p->cxm_cap;
  TRY(grow(&p->cxm, &cap, ((size + 7) & ~7ull) + 16));
  p->cxm_cap = cap;
  p->cxm_bytes = 0;
  HIP_TRY(hipMemsetAsync(p->cxm, 0, ((size + 7) & ~7ull) + 16, p->st));
  k_pck_ctxmap<<<4096, 256, 0, p->st>>>(suf, total_len, (u32) ilog, bits, (u64 *) p->cxm);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(p->st));
  // header: interval log and entry width, 16 bits each, most significant bit
  // first (BWTSeqCRMapOpen :291-296); the file is created by writing the first
  // character of its own suffix ".<ilog>cxm" to its last byte (:297-300): the
  // unused bits of the last byte are those of '.'
  const u8 hdr[4] = { (u8) (ilog >> 8), (u8) ilog, (u8) (bits >> 8), (u8) bits };
  HIP_TRY(hipMemcpy(p->cxm, hdr, 4, hipMemcpyHostToDevice));
  const u32 used = (u32) ((bits * nentries) % 8);
  if (used) {
    u8 last = 0;
    HIP_TRY(hipMemcpy(&last, p->cxm + size - 1, 1, hipMemcpyDeviceToHost));
    last |= (u8) ('.' & ((1u << (8 - used)) - 1));
    HIP_TRY(hipMemcpy(p->cxm + size - 1, &last, 1, hipMemcpyHostToDevice));
  }
  p->cxm_bytes = size;
  if (ilog_used != nullptr) *ilog_used = ilog;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_pck_ctxmap_build_from_esa(gtamd_pck *p, const gtamd_esa_ctx *esa, int ilog,
                                               int *ilog_used) {
  GTAMD_ABI_BEGIN
  if (p == nullptr || esa == nullptr) { gtamd_set_error("invalid argument to gtamd_pck_ctxmap_build_from_esa"); return -1; }
  int device = 0; u32 sigma = 0, numparts = 0;
  TRY(gtamd_esa_internal_info(esa, &device, &sigma, &numparts));
  if (numparts != 1 || device != p->device) { gtamd_set_error("context map: needs the suffix array of a whole-table build on the builder's device"); return -1; }
  const u64 *suf = (const u64 *) gtamd_esa_table_device(esa, GTAMD_TAB_SUF);
  if (suf == nullptr) { gtamd_set_error("context map: the last run did not produce the .suf table"); return -1; }
  return gtamd_pck_ctxmap_build(p, suf, gtamd_esa_table_entries(esa, GTAMD_TAB_SUF), ilog, ilog_used);
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_pck_ctxmap_build_host(gtamd_pck *p, const uint64_t *suf, uint64_t total_len,
                                           int ilog, int *ilog_used) {
  GTAMD_ABI_BEGIN
  if (p == nullptr || suf == nullptr) { gtamd_set_error("invalid argument to gtamd_pck_ctxmap_build_host"); return -1; }
  HIP_TRY(hipSetDevice(p->device));
  u64 *d_suf = nullptr;
  int rc = -1;
  if (hipMalloc(&d_suf, total_len * sizeof(u64)) != hipSuccess)
    gtamd_set_error("context map: cannot allocate device memory for %llu suffix-array entries", (unsigned long long) total_len);
  else if (hipMemcpy(d_suf, suf, total_len * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess)
    gtamd_set_error("context map: cannot copy the suffix array to the device");
  else
    rc = gtamd_pck_ctxmap_build(p, d_suf, total_len, ilog, ilog_used);
  if (d_suf) (void) hipFree(d_suf);
  return rc;
  GTAMD_ABI_END(-1)
}

extern "C" uint64_t gtamd_pck_ctxmap_bytes(const gtamd_pck *p) { return p != nullptr ? p->cxm_bytes : 0; }

extern "C" int gtamd_pck_ctxmap_copy(gtamd_pck *p, void *dst, uint64_t offset, uint64_t count) {
  GTAMD_ABI_BEGIN
  if (p == nullptr || p->cxm_bytes == 0) { gtamd_set_error("no context map built"); return -1; }
  if (count == 0) return 0;
  if (dst == nullptr || offset > p->cxm_bytes || count > p->cxm_bytes - offset) { gtamd_set_error("context map: range outside the image"); return -1; }
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipMemcpy(dst, p->cxm + offset, count, hipMemcpyDeviceToHost));
  return 0;
  GTAMD_ABI_END(-1)
}
