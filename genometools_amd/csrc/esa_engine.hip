// esa_engine.hip -- the hot path of `gt suffixerator` re-designed for MI355X:
// suffix array + LCP + BWT of an encoded sequence, everything resident in HBM.
//
// Pipeline (one stream, no host round trips except two scalar read-backs):
//   pack      bytes -> 2-bit (or 5-bit) words + special-position bitmap
//   keygen    one 64-bit key per suffix (KEY_SYMS symbols + special code +
//             preceding symbol), position as the 32-bit value
//   sort      stable LSD radix sort of (key, position)        [esa_prims.hip]
//   finalize  widen positions to .suf, LCP from clz(key xor key'), BWT from
//             the key payload, mark suffixes tied on the whole key
//   pairs     tie groups of two (most tied suffixes of a genome: low-copy
//             repeats): one comparison on the packed text each, amortised
//             along the text; groups of three and four through the same list
//   refine    what is left: prefix doubling on a rank table (rank of suffix
//             p+h decides among suffixes equal on h symbols), built only for
//             the windows of positions the rounds can reach
//   fixtied   LCP (direct word compare on the packed text) and BWT of the
//             tied suffixes, .llv pairs
// A part build (one of R lexicographic ranges, one per GPU) makes the keys of
// its text tile, exchanges the pairs with the range owners and keeps the rank
// table cut by text position (see "part builds" below).
//
// What the reference does instead (for the record, not followed): bucket by a
// k-mer prefix (src/match/sfx-suffixer.c:1703,2012), then sort each bucket with
// a multikey quicksort family on 32-base words (src/match/sfx-bentsedg.c:1095),
// LCP as a side output (src/match/sfx-lcpvalues.c:622).  The tables are
// algorithm independent; only the order relation
// (src/core/encseq.c:6449-6530, src/match/sfx-bentsedg.c:75-80) is shared.
#include <stdarg.h>
#include <stdlib.h>
#include <chrono>
#include <functional>
#include <type_traits>
#include <vector>
#include "../../include/gtamd_esa.h"
#include "esa_prims.h"
#include "esa_devutil.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void gtamd_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
extern "C" const char *gtamd_esa_last_error(void) { return g_err; }

extern "C" int gtamd_device_count(void) {
  GTAMD_ABI_BEGIN
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
  GTAMD_ABI_END(0)
}

// host only: a container sized by the caller inside the barrier (tests/test_abi.py)
extern "C" int gtamd_abi_selftest(uint64_t host_bytes) {
  GTAMD_ABI_BEGIN
  std::vector<u8> probe((size_t) host_bytes);
  if (!probe.empty()) probe[probe.size() - 1] = 1;
  return probe.empty() ? 0 : (int) probe[probe.size() - 1] - 1;
  GTAMD_ABI_END(-1)
}

// ---------------------------------------------------------------------------
// packed text access
// ---------------------------------------------------------------------------
struct Text {
  const u64 *tb;   // packed symbols
  const u64 *sp;   // special bitmap, bit (p & 63) of word p >> 6; bit n set
  u64 n;           // totallength
  u64 nw_tb, nw_sp;
};

__device__ __forceinline__ u64 tb_word(const Text &t, u64 w) {
  return w < t.nw_tb ? t.tb[w] : 0ull;
}
__device__ __forceinline__ u64 sp_word(const Text &t, u64 w) {
  return w < t.nw_sp ? t.sp[w] : 0ull;
}
// bit j = "position p + j is special", 64 positions
__device__ __forceinline__ u64 sp_window(const Text &t, u64 p) {
  const u64 w = p >> 6;
  const int o = (int) (p & 63);
  u64 r = sp_word(t, w) >> o;
  if (o) r |= sp_word(t, w + 1) << (64 - o);
  return r;
}
__device__ __forceinline__ bool is_special(const Text &t, u64 p) {
  return (sp_word(t, p >> 6) >> (p & 63)) & 1ull;
}

template <int BITS> struct Sym;
template <> struct Sym<2> {
  static constexpr int WIN = 32;  // symbols delivered by window()
  // 32 symbols starting at p, first symbol in the top bits
  static __device__ __forceinline__ u64 window(const Text &t, u64 p) {
    const u64 w = p >> 5;
    const int o = (int) (p & 31) * 2;
    const u64 hi = tb_word(t, w);
    if (o == 0) return hi;
    return (hi << o) | (tb_word(t, w + 1) >> (64 - o));
  }
  static __device__ __forceinline__ u32 at(const Text &t, u64 p) {
    return (u32) (tb_word(t, p >> 5) >> (62 - 2 * (int) (p & 31))) & 3u;
  }
};
template <> struct Sym<5> {
  static constexpr int WIN = 11;  // at least 11 valid symbols
  static __device__ __forceinline__ u64 window(const Text &t, u64 p) {
    const u64 w = p / 12;
    const int r = (int) (p - w * 12);
    const u64 hi = tb_word(t, w) << (5 * r);
    if (r <= 1) return hi;
    return hi | (tb_word(t, w + 1) >> (5 * (12 - r)));
  }
  static __device__ __forceinline__ u32 at(const Text &t, u64 p) {
    const u64 w = p / 12;
    const int r = (int) (p - w * 12);
    return (u32) (tb_word(t, w) >> (59 - 5 * r)) & 31u;
  }
};

template <int BITS> struct Pay {
  static constexpr u32 PB = KeyLayout<BITS>::PAYLOAD_BITS;
  static constexpr u32 WILD = (1u << PB) - 3u, SEP = (1u << PB) - 2u,
                       UNDEF = (1u << PB) - 1u;
  // code of the symbol in front of suffix p
  static __device__ __forceinline__ u32 before(const Text &t, u64 p) {
    if (p == 0) return UNDEF;
    const u32 c = Sym<BITS>::at(t, p - 1);
    // a special is stored as code 0 or 1, so any other code needs no look at
    // the bitmap (one random line less per look-up, most of the time)
    if (c >= 2u) return c;
    if (is_special(t, p - 1)) return (c & 1u) ? SEP : WILD;
    return c;
  }
  // .bwt byte, src/match/sfx-run.c:173-210
  static __device__ __forceinline__ u8 to_bwt(u32 code) {
    if (code == SEP) return (u8) GTAMD_SEPARATOR;
    if (code >= WILD) return (u8) GTAMD_WILDCARD;  // wildcard, or UNDEFBWTCHAR
    return (u8) code;
  }
};

template <int BITS> struct Key {
  using L = KeyLayout<BITS>;
  static constexpr int SYMS = L::KEY_SYMS;
  // symbols two suffixes share FOR SURE when the first sort leaves them tied: all
  // of the key's; eight for the 5-bit alphabets, whose MSD sort (esa_msd.h, FMT 1)
  // looks at eight symbols and the class of the ninth.  Everything behind the sort
  // starts from this bound (a smaller one is never wrong: comparisons start
  // earlier, the doubling rounds at a smaller offset).
  static constexpr int KNOWN = BITS == 2 ? SYMS : 8;
  static constexpr int PFX_BITS = SYMS * BITS;
  static constexpr int LOW_BITS = 64 - PFX_BITS;          // below the prefix
  static constexpr int DSHIFT = LOW_BITS - L::DCODE_BITS; // dcode sits right
                                                          // below the prefix
  static constexpr u32 DMAX = (1u << L::DCODE_BITS) - 1u;  // suffix starts special
  static constexpr u64 PAY_MASK = (1ull << L::PAYLOAD_BITS) - 1ull;
  static_assert(DSHIFT >= L::PAYLOAD_BITS, "key layout does not fit 64 bits");
  // the unsorted bits between dcode and payload carry the position bits above
  // 2^32 (part builds of sequences with n >= 2^32: the 32-bit value of the
  // sort holds the low half), so the pairs stay 12 bytes
  static constexpr int SPARE_SHIFT = L::PAYLOAD_BITS;
  static constexpr int SPARE_BITS = DSHIFT - L::PAYLOAD_BITS;   // 16 (DNA), 5
  static constexpr u64 SPARE_MASK = (1ull << SPARE_BITS) - 1ull;
  static __device__ __forceinline__ u64 poshi(u64 key) {
    return (key >> SPARE_SHIFT) & SPARE_MASK;
  }
  static __device__ __forceinline__ u64 with_poshi(u64 key, u64 p) {
    return key | ((p >> 32) << SPARE_SHIFT);
  }
  static __device__ __forceinline__ u32 dcode(u64 key) {
    return (u32) (key >> DSHIFT) & DMAX;
  }
  // number of letters in front of the first special, capped at SYMS
  static __device__ __forceinline__ u32 letters(u64 key) {
    const u32 dc = dcode(key);
    return dc == 0 ? (u32) SYMS : (dc == DMAX ? 0u : (u32) SYMS - dc);
  }
};

// common prefix of two suffixes p, q known to share l symbols: compare the
// packed text word by word, stop at the first special on either side
// (specials never match: src/core/encseq.c:6449-6530, sfx-linlcp.c:93,162)
template <int BITS>
__device__ u64 lcp_extend(const Text &t, u64 p, u64 q, u64 l,
                          u64 cap = ~0ull) {
  constexpr int S = Sym<BITS>::WIN;
  for (;;) {
    if (l >= cap) return l;
    u64 x = Sym<BITS>::window(t, p + l) ^ Sym<BITS>::window(t, q + l);
    if (BITS * S < 64) x &= ~0ull << (64 - BITS * S);
    int m = x ? __clzll((long long) x) / BITS : S;
    u64 s = sp_window(t, p + l) | sp_window(t, q + l);
    if (S < 64) s &= (1ull << S) - 1ull;
    const int ds = s ? __ffsll((unsigned long long) s) - 1 : S;
    const int step = m < ds ? m : ds;
    l += (u64) step;
    if (step < S) return l;
  }
}

// The same for matches that can be long (the pair comparison, the LCP walk: copies
// of thousands of symbols): FOUR windows per step on the 2-bit text.  A step of
// lcp_extend is one round trip to memory for 32 symbols, and nothing of the next
// step can be issued before this one has decided -- a wave whose one lane walks a
// copy of 4 K symbols sits through 128 of them.  Here the ten text words and the six
// bitmap words of 128 symbols are in flight together (reading behind the first
// difference is harmless: the accessors return 0 behind the text).
// p_first (may be nullptr): whether suffix p comes before suffix q -- the first
// difference decides, a special is larger than every letter, two specials compare
// by position.  The symbols and special bits at the first difference are in the words
// the step has loaded: asking the text again would be one more round trip to memory
// per comparison.
template <int BITS, int W = 4>
__device__ u64 lcp_extend_long(const Text &t, u64 p, u64 q, u64 l, bool *p_first = nullptr) {
  static_assert(W == 4 || W == 8, "windows per step");
  if (BITS != 2) {
    l = lcp_extend<BITS>(t, p, q, l);
    if (p_first != nullptr) {
      const bool spa = is_special(t, p + l), spb = is_special(t, q + l);
      *p_first = (spa || spb) ? ((spa && spb) ? p < q : spb)
                              : Sym<BITS>::at(t, p + l) < Sym<BITS>::at(t, q + l);
    }
    return l;
  }
  for (;;) {
    const u64 P = p + l, Q = q + l;
    const u64 wp = P >> 5, wq = Q >> 5;
    const int op = (int) (P & 31) * 2, oq = (int) (Q & 31) * 2;
    u64 tp[W + 1], tq[W + 1], sp[W / 2 + 1], sq[W / 2 + 1];
#pragma unroll
    for (int i = 0; i <= W; i++) { tp[i] = tb_word(t, wp + i); tq[i] = tb_word(t, wq + i); }
    const u64 swp = P >> 6, swq = Q >> 6;
    const int sop = (int) (P & 63), soq = (int) (Q & 63);
#pragma unroll
    for (int i = 0; i <= W / 2; i++) { sp[i] = sp_word(t, swp + i); sq[i] = sp_word(t, swq + i); }
#pragma unroll
    for (int k = 0; k < W; k++) {
      const u64 a = op ? (tp[k] << op) | (tp[k + 1] >> (64 - op)) : tp[k];
      const u64 b = oq ? (tq[k] << oq) | (tq[k + 1] >> (64 - oq)) : tq[k];
      const u64 x = a ^ b;
      const int m = x ? __clzll((long long) x) / 2 : 32;
      // special bits of the 64 positions from P + 64 (k / 2) / from Q + ...
      const u64 Sp = sop ? (sp[k / 2] >> sop) | (sp[k / 2 + 1] << (64 - sop)) : sp[k / 2];
      const u64 Sq = soq ? (sq[k / 2] >> soq) | (sq[k / 2 + 1] << (64 - soq)) : sq[k / 2];
      const u32 sb = (u32) ((Sp | Sq) >> (32 * (k & 1)));
      const int ds = sb ? __ffs((int) sb) - 1 : 32;
      const int step = m < ds ? m : ds;
      l += (u64) step;
      if (step < 32) {
        if (p_first != nullptr) {
          const int at = 32 * (k & 1) + step;           // bit of the deciding position in Sp, Sq
          const bool spa = (Sp >> at) & 1ull, spb = (Sq >> at) & 1ull;
          *p_first = (spa || spb) ? ((spa && spb) ? p < q : spb)
                                  : ((a >> (62 - 2 * step)) & 3ull) < ((b >> (62 - 2 * step)) & 3ull);
        }
        return l;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// pack: bytes -> packed words + special bitmap
// ---------------------------------------------------------------------------
// readmode of the reference (src/core/readmode.h): the sequence is read in
// reverse (rev: symbol p comes from position n - 1 - p) and / or complemented
// (cpl: letter c becomes 3 - c, DNA only; specials stay what they are)
template <int BITS>
__global__ __launch_bounds__(256) void k_pack_symbols(
    const u8 *__restrict__ enc, u64 n, u64 *__restrict__ tb, u64 nwords, int rev, int cpl) {
  constexpr int SPW = KeyLayout<BITS>::SYMS_PER_WORD;
  const u64 w = (u64) blockIdx.x * 256 + threadIdx.x;
  if (w >= nwords) return;
  const u64 base = w * SPW;
  u64 word = 0;
#pragma unroll
  for (int i = 0; i < SPW; i++) {
    const u64 p = base + i;
    u64 c = 0;
    if (p < n) {
      const u32 b = enc[rev ? n - 1 - p : p];
      // a special keeps its kind in the low bit: 0 wildcard, 1 separator
      c = b >= GTAMD_WILDCARD ? (b == GTAMD_SEPARATOR ? 1u : 0u) : (cpl ? 3u - b : b);
    }
    word |= c << (64 - BITS * (i + 1));
  }
  tb[w] = word;
}

__global__ __launch_bounds__(256) void k_pack_specials(
    const u8 *__restrict__ enc, u64 n, u64 *__restrict__ sp, u64 nwords, int rev) {
  const u64 w = (u64) blockIdx.x * 256 + threadIdx.x;
  if (w >= nwords) return;
  const u64 base = w * 64;
  u64 word = 0;
  if (!rev && base + 64 <= n && (((uintptr_t) (enc + base)) & 15) == 0) {
    const uint4 *v = reinterpret_cast<const uint4 *>(enc + base);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint4 q = v[k];
      const u32 parts[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
          if (((parts[a] >> (8 * b)) & 255u) >= GTAMD_WILDCARD)
            word |= 1ull << (k * 16 + a * 4 + b);
    }
  } else {
    for (int i = 0; i < 64; i++) {
      const u64 p = base + i;
      if (p < n ? enc[rev ? n - 1 - p : p] >= GTAMD_WILDCARD : p == n) word |= 1ull << i;
    }
  }
  if (base <= n && n < base + 64) word |= 1ull << (n - base);  // virtual end
  sp[w] = word;
}

// ---------------------------------------------------------------------------
// keygen
// ---------------------------------------------------------------------------
template <int BITS>
__device__ __forceinline__ u64 make_key(const Text &t, u64 p) {
  using K = Key<BITS>;
  constexpr int SYMS = K::SYMS;
  const u32 pay = Pay<BITS>::before(t, p);
  u64 s = sp_window(t, p) & ((1ull << SYMS) - 1ull);
  const int d = s ? __ffsll((unsigned long long) s) - 1 : SYMS;
  if (d == 0)  // suffix starts with a special (or is the virtual end)
    return (~0ull << K::DSHIFT) | pay;
  u64 pre = Sym<BITS>::window(t, p) >> K::LOW_BITS;  // SYMS symbols
  u32 dc = 0;
  if (d < SYMS) {
    pre |= (1ull << (BITS * (SYMS - d))) - 1ull;  // pad behind the d letters
    dc = (u32) (SYMS - d);
  }
  return (pre << K::LOW_BITS) | ((u64) dc << K::DSHIFT) | pay;
}

// keys of the suffixes [first, end) (a whole table, or one text tile of a part
// build), written to local indices 0 .. end - first; the value is the low half
// of the position, the high half rides in the key's spare bits
template <int BITS>
__global__ __launch_bounds__(256) void k_keygen(Text t, u64 first, u64 end,
                                                u64 *__restrict__ keys,
                                                u32 *__restrict__ vals) {
  const u64 base = first + (u64) blockIdx.x * 1024;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u64 p = base + (u64) j * 256 + threadIdx.x;
    if (p < end) {
      keys[p - first] = Key<BITS>::with_poshi(make_key<BITS>(t, p), p);
      vals[p - first] = (u32) p;
    }
  }
}

// DNA: one thread makes the keys of FOUR consecutive suffixes from one pair of
// text words and one pair of bitmap words (the general kernel spends ~140
// instructions per suffix, mostly on re-deriving the same windows, and is
// bound by them, not by its 36 GB of stores); stores are 16 bytes per lane.
// Same keys as make_key<2>, bit for bit.  `first` is a multiple of 4.
__global__ __launch_bounds__(256) void k_keygen_dna(Text t, u64 first, u64 end,
                                                    u64 *__restrict__ keys,
                                                    u32 *__restrict__ vals) {
  using K = Key<2>;
  using P = Pay<2>;
  constexpr int SYMS = K::SYMS;
  static_assert(28 + 3 + SYMS <= 64, "four windows must fit one word pair");
  const u64 p0 = first + ((u64) blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= end) return;
  const u64 w = p0 >> 5;
  const int o = (int) (p0 & 31) * 2;           // 0, 8, ..., 56
  const u64 hi = tb_word(t, w), lo = tb_word(t, w + 1);
  const u64 a_hi = o ? (hi << o) | (lo >> (64 - o)) : hi;   // symbols p0 .. p0+31
  const u64 a_lo = lo << o;                                 // the ones behind
  const u64 sw = p0 >> 6;
  const int so = (int) (p0 & 63);              // multiple of 4
  const u64 s0 = sp_word(t, sw), s1 = sp_word(t, sw + 1);
  const u64 S = so ? (s0 >> so) | (s1 << (64 - so)) : s0;   // specials p0 .. p0+63
  // the symbol in front of p0
  u32 pay;
  if (p0 == 0) {
    pay = P::UNDEF;
  } else {
    const u32 c = o ? (u32) (hi >> (64 - o)) & 3u : (u32) tb_word(t, w - 1) & 3u;
    const bool sp = c < 2u && (so ? (s0 >> (so - 1)) & 1ull : sp_word(t, sw - 1) >> 63);
    pay = sp ? ((c & 1u) ? P::SEP : P::WILD) : c;
  }
  const u64 poshi = (p0 >> 32) << K::SPARE_SHIFT;   // (p0 + 3 has the same high half)
  u64 key[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const u64 win = g ? (a_hi << (2 * g)) | (a_lo >> (64 - 2 * g)) : a_hi;
    const u64 s = (S >> g) & ((1ull << SYMS) - 1ull);
    const int d = s ? __ffsll((unsigned long long) s) - 1 : SYMS;
    if (d == 0) {
      key[g] = (~0ull << K::DSHIFT) | pay | poshi;
    } else {
      u64 pre = win >> K::LOW_BITS;
      u32 dc = 0;
      if (d < SYMS) {
        pre |= (1ull << (2 * (SYMS - d))) - 1ull;
        dc = (u32) (SYMS - d);
      }
      key[g] = (pre << K::LOW_BITS) | ((u64) dc << K::DSHIFT) | pay | poshi;
    }
    // in front of the next suffix: this one's first symbol
    const u32 c = (u32) (win >> 62);
    pay = (c < 2u && (s & 1ull)) ? ((c & 1u) ? P::SEP : P::WILD) : c;
  }
  const u64 l0 = p0 - first;
  if (p0 + 4 <= end) {
    *reinterpret_cast<ulonglong2 *>(keys + l0) = make_ulonglong2(key[0], key[1]);
    *reinterpret_cast<ulonglong2 *>(keys + l0 + 2) = make_ulonglong2(key[2], key[3]);
    *reinterpret_cast<uint4 *>(vals + l0) =
        make_uint4((u32) p0, (u32) p0 + 1u, (u32) p0 + 2u, (u32) p0 + 3u);
  } else {
    for (int g = 0; g < 4 && p0 + g < end; g++) {
      keys[l0 + g] = key[g];
      vals[l0 + g] = (u32) (p0 + g);
    }
  }
}

// DNA, whole-table builds: keygen and the first radix pass in one.  The lowest
// sorted digit is dcode (5 bits), which is 0 for every suffix with no special
// among its first 20 symbols -- 99.5 % of a genome -- so that pass is a stable
// partition that moves almost nothing: all dcode-0 pairs stay in text order,
// the few others go behind them by class.  Written as a pass of the general
// sort it costs a keygen store, a histogram read and a full scatter (29 ms at
// 3 Gbp).  Here: k_dc_hist_dna counts the classes per 4096-position tile from
// the special bitmap alone; after the sort's own column scan k_keygen_pass0_dna
// makes the keys (16 consecutive suffixes per thread from one word pair), packs
// the tile's dcode-0 pairs in LDS and writes them as one coalesced run at their
// final place of this pass; the rare others are ranked by one wave, in
// position order, with the ballot match of the scatter kernel.
constexpr int KP_TILE = 4096;          // = the sort's tile
constexpr int KP_PER = 8;              // consecutive suffixes per thread
constexpr int KP_THREADS = KP_TILE / KP_PER;   // 512: 8 waves per workgroup, 3 workgroups per CU
constexpr int KP_ROW = 256;            // = words per tile in the sort's histogram

__device__ __forceinline__ u32 dna_dcode_of(u64 s /* special bits of 20 positions */) {
  constexpr int SYMS = Key<2>::SYMS;
  if (s == 0) return 0;
  const int d = __ffsll((unsigned long long) s) - 1;
  return d == 0 ? Key<2>::DMAX : (u32) (SYMS - d);
}

__global__ __launch_bounds__(KP_THREADS) void k_dc_hist_dna(Text t, u64 N,
                                                            u32 *__restrict__ hist) {
  constexpr int SYMS = Key<2>::SYMS;
  __shared__ u32 h[32];
  if (threadIdx.x < 32) h[threadIdx.x] = 0;
  __syncthreads();
  const u64 p0 = (u64) blockIdx.x * KP_TILE + (u64) threadIdx.x * KP_PER;
  u32 n0 = 0;
  if (p0 < N) {
    const int npos = N - p0 < KP_PER ? (int) (N - p0) : KP_PER;
    const u64 S = sp_window(t, p0);
    if ((S & ((1ull << (npos + SYMS - 1)) - 1ull)) == 0) {
      n0 = (u32) npos;
    } else {
      for (int g = 0; g < npos; g++) {
        const u32 dc = dna_dcode_of((S >> g) & ((1ull << SYMS) - 1ull));
        if (dc == 0) n0++; else atomicAdd(&h[dc], 1u);
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) n0 += __shfl_xor(n0, d, 64);
  if ((threadIdx.x & 63) == 0 && n0) atomicAdd(&h[0], n0);
  __syncthreads();
  if (threadIdx.x < KP_ROW)
    hist[(u64) blockIdx.x * KP_ROW + threadIdx.x] = threadIdx.x < 32 ? h[threadIdx.x] : 0u;
}

typedef __attribute__((address_space(3))) volatile u32 lds_vu32;

__global__ __launch_bounds__(KP_THREADS) void k_keygen_pass0_dna(
    Text t, u64 N, const u32 *__restrict__ hist_scanned, u64 *__restrict__ keys,
    u32 *__restrict__ vals) {
  using K = Key<2>;
  using P = Pay<2>;
  constexpr int SYMS = K::SYMS;
  static_assert(32 - KP_PER + (KP_PER - 1) + SYMS <= 64 && 32 % KP_PER == 0,
                "the windows of a thread must fit one word pair");
  __shared__ u64 s_key[KP_TILE];
  __shared__ u32 s_val[KP_TILE];
  __shared__ u32 s_base[32], s_run_mem[32];
  __shared__ u32 s_scan[KP_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const u64 tile_base = (u64) blockIdx.x * KP_TILE;
  const u32 tile_valid = (u32) (N - tile_base < (u64) KP_TILE ? N - tile_base : (u64) KP_TILE);
  if (tid < 32) {
    s_base[tid] = hist_scanned[(u64) blockIdx.x * KP_ROW + tid];
    s_run_mem[tid] = 0;
  }
  const u64 p0 = tile_base + (u64) tid * KP_PER;
  const int npos = p0 >= N ? 0 : (N - p0 < KP_PER ? (int) (N - p0) : KP_PER);
  u64 key[KP_PER];
  u32 zero = 0;                       // bit g: suffix p0 + g has dcode 0
  if (npos > 0) {
    const u64 w = p0 >> 5;
    const int o = (int) (p0 & 31) * 2;           // a multiple of 2 * KP_PER
    const u64 hi = tb_word(t, w), lo = tb_word(t, w + 1);
    const u64 a_hi = o ? (hi << o) | (lo >> (64 - o)) : hi;   // symbols p0 .. p0+31
    const u64 a_lo = lo << o;
    const u64 sw = p0 >> 6;
    const int so = (int) (p0 & 63);              // a multiple of KP_PER
    const u64 s0 = sp_word(t, sw), s1 = sp_word(t, sw + 1);
    const u64 S = so ? (s0 >> so) | (s1 << (64 - so)) : s0;   // specials p0 .. p0+63
    u32 pay;
    if (p0 == 0) {
      pay = P::UNDEF;
    } else {
      const u32 c = o ? (u32) (hi >> (64 - o)) & 3u : (u32) tb_word(t, w - 1) & 3u;
      const bool sp = c < 2u && (so ? (s0 >> (so - 1)) & 1ull : sp_word(t, sw - 1) >> 63);
      pay = sp ? ((c & 1u) ? P::SEP : P::WILD) : c;
    }
#pragma unroll
    for (int g = 0; g < KP_PER; g++) {
      const u64 win = g ? (a_hi << (2 * g)) | (a_lo >> (64 - 2 * g)) : a_hi;
      const u64 s = (S >> g) & ((1ull << SYMS) - 1ull);
      const int d = s ? __ffsll((unsigned long long) s) - 1 : SYMS;
      if (d == 0) {
        key[g] = (~0ull << K::DSHIFT) | pay;
      } else {
        u64 pre = win >> K::LOW_BITS;
        u32 dc = 0;
        if (d < SYMS) {
          pre |= (1ull << (2 * (SYMS - d))) - 1ull;
          dc = (u32) (SYMS - d);
        }
        key[g] = (pre << K::LOW_BITS) | ((u64) dc << K::DSHIFT) | pay;
      }
      if (s == 0 && g < npos) zero |= 1u << g;
      const u32 c = (u32) (win >> 62);
      pay = (c < 2u && (s & 1ull)) ? ((c & 1u) ? P::SEP : P::WILD) : c;
    }
  }
  // places inside the tile: dcode-0 pairs from the front in position order,
  // the others behind them, also in position order
  u32 cnt0;
  u32 off0 = block_scan_excl<SCAN_SUM, KP_THREADS>((u32) __popc(zero), &cnt0, s_scan);
  const u32 before = (u32) tid * KP_PER < tile_valid ? (u32) tid * KP_PER : tile_valid;
  u32 offr = cnt0 + (before - off0);
#pragma unroll
  for (int g = 0; g < KP_PER; g++) {
    if (g < npos) {
      const u32 at = ((zero >> g) & 1u) ? off0++ : offr++;
      s_key[at] = key[g];
      s_val[at] = (u32) (p0 + g);
    }
  }
  __syncthreads();
  // the dcode-0 run, 16 bytes per lane where the output index allows it
  const u32 ob0 = s_base[0];
  {
    const u32 head = (ob0 & 1u) < cnt0 ? (ob0 & 1u) : cnt0;     // to an even index
    if ((u32) tid < head) keys[ob0 + tid] = s_key[tid];
    const u32 pairs = (cnt0 - head) >> 1;
    for (u32 q = (u32) tid; q < pairs; q += KP_THREADS) {
      const u32 e = head + 2u * q;
      *reinterpret_cast<ulonglong2 *>(keys + ob0 + e) = make_ulonglong2(s_key[e], s_key[e + 1]);
    }
    if (tid == 0 && head + 2u * pairs < cnt0) keys[ob0 + cnt0 - 1] = s_key[cnt0 - 1];
  }
  {
    const u32 mis = (4u - (ob0 & 3u)) & 3u;                      // to a multiple of 4
    const u32 head = mis < cnt0 ? mis : cnt0;
    if ((u32) tid < head) vals[ob0 + tid] = s_val[tid];
    const u32 quads = (cnt0 - head) >> 2;
    for (u32 q = (u32) tid; q < quads; q += KP_THREADS) {
      const u32 e = head + 4u * q;
      *reinterpret_cast<uint4 *>(vals + ob0 + e) =
          make_uint4(s_val[e], s_val[e + 1], s_val[e + 2], s_val[e + 3]);
    }
    const u32 done = head + 4u * quads;
    if ((u32) tid < cnt0 - done) vals[ob0 + done + tid] = s_val[done + tid];
  }
  const u32 nrare = tile_valid - cnt0;
  if (nrare == 0 || tid >= 64) return;
  // one wave ranks the rest by class, 64 at a time
  lds_vu32 *s_run = (lds_vu32 *) s_run_mem;
  for (u32 base = 0; base < nrare; base += 64) {
    const u32 idx = base + (u32) lane;
    const bool valid = idx < nrare;
    const u64 k = valid ? s_key[cnt0 + idx] : 0ull;
    const u32 v = valid ? s_val[cnt0 + idx] : 0u;
    const u32 c = valid ? K::dcode(k) : 32u;      // 32: no pair in this lane
    u32 mlo = ~0u, mhi = ~0u;
#pragma unroll
    for (int b = 0; b < 6; b++) {
      u32 sx = (u32) ((int) (c << (31 - b)) >> 31);
      asm volatile("" : "+v"(sx));
      const u64 bal = __ballot(sx != 0);
      mlo = __builtin_amdgcn_bitop3_b32(mlo, (u32) bal, sx, 0x90);
      mhi = __builtin_amdgcn_bitop3_b32(mhi, (u32) (bal >> 32), sx, 0x90);
    }
    const u32 intra = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
    const u32 old = s_run[c & 31u];
    if (valid && intra == 0) s_run[c] = old + (u32) __popc(mlo) + (u32) __popc(mhi);
    if (valid) {
      const u32 g = s_base[c] + old + intra;
      keys[g] = k;
      vals[g] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// part builds.  The suffix array is cut into R lexicographic ranges of (almost)
// equal size, the reference's -parts idea (src/match/sfx-partssuf.c:172-347,
// filter src/match/sfx-suffixer.c:375-398); part r builds slice r of every
// table.  What a part does per suffix falls with R:
//   * the TEXT is cut into R tiles of T positions; part r makes the keys of its
//     tile only and sends every (key, position) pair to the part whose key
//     range it falls into (alltoallv); received in source order the pairs of a
//     part are in text order again, which the stable sort needs;
//   * the rank table (inverse suffix array) of the prefix doubling is cut by
//     TEXT POSITION: part t holds the ranks of the suffixes that start in tile
//     t.  The owner of position q is q / T -- no owner map, no text look-up --
//     so a round is: queries (offset in the tile) to the tile owners, answers
//     back, and the new ranks of refined suffixes to the tile owners.
// ---------------------------------------------------------------------------
constexpr int PART_BITS = 14;
constexpr int PART_BINS = 1 << PART_BITS;
constexpr u32 DEST_LOCAL = 0xFEu;   // handled in place, not sent
constexpr u32 DEST_NONE = 0xFFu;    // nothing to do for this item
constexpr int DEST_MAXPARTS = 128;  // parts of one build (destination fits a byte)

// leading PART_BITS bits of the key of suffix p (the bits the range partition
// looks at): the padded symbol prefix only, without dcode and payload
template <int BITS>
__device__ __forceinline__ u32 key_bin(const Text &t, u64 p) {
  using K = Key<BITS>;
  constexpr int SYMS = K::SYMS;
  u64 s = sp_window(t, p) & ((1ull << SYMS) - 1ull);
  const int d = s ? __ffsll((unsigned long long) s) - 1 : SYMS;
  if (d == 0) return (u32) PART_BINS - 1u;
  u64 pre = Sym<BITS>::window(t, p) >> K::LOW_BITS;
  if (d < SYMS) pre |= (1ull << (BITS * (SYMS - d))) - 1ull;
  return (u32) (pre >> (K::PFX_BITS - PART_BITS));
}

// histogram of the key bins over every `stride`-th suffix of [first, end):
// enough to place the range cuts; exact slice sizes come from the exchange
template <int BITS>
__global__ __launch_bounds__(256) void k_key_hist(Text t, u64 first, u64 end, u64 stride,
                                                  u32 *__restrict__ hist) {
  __shared__ u32 h[PART_BINS];
  for (int i = threadIdx.x; i < PART_BINS; i += 256) h[i] = 0;
  __syncthreads();
  // sample positions: the multiples of stride inside the tile
  const u64 s0 = (first + stride - 1) / stride;
  const u64 s1 = (end + stride - 1) / stride;
  for (u64 i = s0 + (u64) blockIdx.x * 256 + threadIdx.x; i < s1;
       i += (u64) gridDim.x * 256)
    atomicAdd(&h[key_bin<BITS>(t, i * stride)], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < PART_BINS; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

// ---- bucketing by destination part, without a sort ---------------------------
// k_dest_count: destination of every item (kept as a byte) and, per block of 256
// items, how many go to each part -- laid out part-major, so that ONE exclusive
// scan over numparts * blocks counters gives every block its write offset inside
// every part's segment of the send buffer.  k_dest_place then puts the items
// there (stable); items for this part itself are handled on the spot.
// F: dest(j) -> part, DEST_LOCAL or DEST_NONE; local(j); emit(j, at).
template <typename F>
__global__ __launch_bounds__(256) void k_dest_count(F f, u64 m, u32 numparts, u64 nblocks,
                                                    u8 *__restrict__ dest,
                                                    u32 *__restrict__ bcount) {
  __shared__ u32 s_cnt[4][DEST_MAXPARTS];
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u32 d = DEST_NONE;
  if (j < m) {
    d = f.dest(j);
    dest[j] = (u8) d;
  }
  for (u32 r = 0; r < numparts; r++) {
    const u64 mr = __ballot(d == r);
    if (lane == 0) s_cnt[w][r] = (u32) __popcll(mr);
  }
  __syncthreads();
  if (threadIdx.x < numparts)
    bcount[(u64) threadIdx.x * nblocks + blockIdx.x] =
        s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] +
        s_cnt[3][threadIdx.x];
}

template <typename F>
__global__ __launch_bounds__(256) void k_dest_place(F f, u64 m, u32 numparts, u64 nblocks,
                                                    const u8 *__restrict__ dest,
                                                    const u32 *__restrict__ boff) {
  __shared__ u32 s_cnt[4][DEST_MAXPARTS];
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u32 d = j < m ? dest[j] : DEST_NONE;
  u32 before = 0;                 // items of this wave's lower lanes to the same part
  for (u32 r = 0; r < numparts; r++) {
    const u64 mr = __ballot(d == r);
    if (lane == 0) s_cnt[w][r] = (u32) __popcll(mr);
    if (d == r) before = (u32) __popcll(mr & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (j >= m || d == DEST_NONE) return;
  if (d == DEST_LOCAL) { f.local(j); return; }
  u32 at = boff[(u64) d * nblocks + blockIdx.x] + before;
  for (int x = 0; x < w; x++) at += s_cnt[x][d];
  f.emit(j, at);
}

// send counts per part from the scanned block counters: segment boundaries
__global__ void k_dest_totals(const u32 *__restrict__ boff, const u32 *__restrict__ bcount,
                              u32 numparts, u64 nblocks, u32 *__restrict__ counts) {
  const u32 r = threadIdx.x;
  if (r >= numparts) return;
  const u64 first = (u64) r * nblocks, last = first + nblocks - 1;
  counts[r] = boff[last] + bcount[last] - boff[first];
}

// tile geometry of a part build: position q lives in tile q / T
struct Tiles {
  u64 T;        // positions per tile
  u32 self;     // this part
  __device__ __forceinline__ u32 owner(u64 q, u64 *off) const {
    const u64 d = q / T;
    *off = q - d * T;
    return (u32) d;
  }
};

// (key, position) pairs of the own text tile to the owners of their key range
struct OwnerOfKey {
  const u64 *keys;
  const u32 *vals;
  const u8 *owner;     // bin -> part
  u64 *keys_out;
  u32 *vals_out;
  __device__ __forceinline__ u32 dest(u64 j) const {
    return owner[keys[j] >> (64 - PART_BITS)];
  }
  __device__ __forceinline__ void local(u64) const {}
  __device__ __forceinline__ void emit(u64 j, u32 at) const {
    keys_out[at] = keys[j];
    vals_out[at] = vals[j];
  }
};

// first ranks of the suffixes [c0, c0 + m) of this part's slice to the owners
// of their text positions: rank = slice offset + head of the entry's tie group
template <typename P> struct InitialRanks {
  const P *sa;
  const u64 *tiebits;
  const u32 *carry;
  u64 c0, index_offset;
  Tiles tl;
  P *isa;              // this part's tile of the rank table
  u32 *srec;           // records: offset in the owner's tile, rank (one or two words)
  const u32 *sel;      // windows of 2^wb positions whose ranks travel (nullptr: all)
  int wb;
  __device__ __forceinline__ u32 dest(u64 j) const {
    u64 off;
    const u64 p = (u64) sa[c0 + j];
    if (sel != nullptr) {
      const u64 w = p >> wb;
      if (!((sel[w >> 5] >> (w & 31)) & 1u)) return DEST_NONE;
    }
    const u32 d = tl.owner(p, &off);
    return d == tl.self ? DEST_LOCAL : d;
  }
  __device__ __forceinline__ P rank_of(u64 i) const {
    const u64 w = i >> 6;
    const int b = (int) (i & 63);
    const u64 below = b == 63 ? ~0ull : ((2ull << b) - 1ull);
    const u64 z = ~tiebits[w] & below;
    const u64 head = z ? w * 64 + (u64) (63 - __clzll((long long) z)) : (u64) carry[w];
    return (P) (index_offset + head);
  }
  __device__ __forceinline__ void local(u64 j) const {
    u64 off;
    (void) tl.owner((u64) sa[c0 + j], &off);
    isa[off] = rank_of(c0 + j);
  }
  __device__ __forceinline__ void emit(u64 j, u32 at) const {
    u64 off;
    (void) tl.owner((u64) sa[c0 + j], &off);
    const u64 r = (u64) rank_of(c0 + j);
    u32 *rec = srec + (u64) at * (1 + sizeof(P) / 4);
    rec[0] = (u32) off;
    rec[1] = (u32) r;
    if (sizeof(P) == 8) rec[2] = (u32) (r >> 32);
  }
};

// the same for the entries k_win_filter has listed (position, head of the tie group):
// only the suffixes of the windows whose ranks travel are looked at, once.  The
// ranks leave as records (offset in the owner's tile, rank) -- one exchange.
template <typename P> struct FilteredRanks {
  const P *fpos;
  const u32 *fhead;
  u64 index_offset;
  Tiles tl;
  P *isa;
  u32 *srec;           // records of RANKREC words: offset, rank (one or two words)
  static constexpr int RANKREC = 1 + (int) (sizeof(P) / 4);
  __device__ __forceinline__ u32 dest(u64 j) const {
    u64 off;
    const u32 d = tl.owner((u64) fpos[j], &off);
    return d == tl.self ? DEST_LOCAL : d;
  }
  __device__ __forceinline__ void local(u64 j) const {
    u64 off;
    (void) tl.owner((u64) fpos[j], &off);
    isa[off] = (P) (index_offset + fhead[j]);
  }
  __device__ __forceinline__ void emit(u64 j, u32 at) const {
    u64 off;
    (void) tl.owner((u64) fpos[j], &off);
    const u64 r = index_offset + fhead[j];
    u32 *rec = srec + (u64) at * RANKREC;
    rec[0] = (u32) off;
    rec[1] = (u32) r;
    if (sizeof(P) == 8) rec[2] = (u32) (r >> 32);
  }
};

// rank queries of a round: k2[j] = rank of suffix upos[j] + h
template <typename P> struct RankQueries {
  const P *upos;
  u64 h, n;
  Tiles tl;
  const P *isa;
  P *k2;
  u32 *sendq, *order;
  __device__ __forceinline__ u64 target(u64 j) const {
    const u64 q = (u64) upos[j] + h;
    return q > n ? n : q;   // cannot happen for a tied suffix; keeps the access in range
  }
  __device__ __forceinline__ u32 dest(u64 j) const {
    u64 off;
    const u32 d = tl.owner(target(j), &off);
    return d == tl.self ? DEST_LOCAL : d;
  }
  // (the look-ups in the own tile: by k_local_lookups, once the new ranks the
  // other parts send with their queries are stored -- not here)
  __device__ __forceinline__ void local(u64) const {}
  __device__ __forceinline__ void lookup_local(u64 j) const {
    u64 off;
    if (tl.owner(target(j), &off) == tl.self) k2[j] = isa[off];
  }
  __device__ __forceinline__ void emit(u64 j, u32 at) const {
    u64 off;
    (void) tl.owner(target(j), &off);
    sendq[at] = (u32) off;
    order[at] = (u32) j;
  }
};
template <typename P>
__global__ __launch_bounds__(256) void k_local_lookups(RankQueries<P> rq, u64 m) {
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  if (j < m) rq.lookup_local(j);
}

// new ranks of the suffixes a round has refined (slot j moved to a new group)
template <typename P> struct RankUpdates {
  const P *cval;         // position now in slot j
  const u32 *gnew;       // its new group head
  const u32 *ugrp;       // the slot's old group head
  u64 index_offset;
  Tiles tl;
  P *isa;
  u32 *soff;
  P *srank;
  __device__ __forceinline__ u32 dest(u64 j) const {
    if (gnew[j] == ugrp[j]) return DEST_NONE;   // the leading subgroup keeps its rank
    u64 off;
    const u32 d = tl.owner((u64) cval[j], &off);
    return d == tl.self ? DEST_LOCAL : d;
  }
  __device__ __forceinline__ void local(u64 j) const {
    u64 off;
    (void) tl.owner((u64) cval[j], &off);
    isa[off] = (P) (index_offset + gnew[j]);
  }
  __device__ __forceinline__ void emit(u64 j, u32 at) const {
    u64 off;
    (void) tl.owner((u64) cval[j], &off);
    soff[at] = (u32) off;
    srank[at] = (P) (index_offset + gnew[j]);
  }
};

template <typename P>
__global__ __launch_bounds__(256) void k_isa_store(const u32 *__restrict__ off,
                                                   const P *__restrict__ rank, u64 cnt,
                                                   P *__restrict__ isa) {
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  if (i < cnt) isa[off[i]] = rank[i];
}

// records (offset, rank) as FilteredRanks / the fused round exchange send them
template <typename P>
__global__ __launch_bounds__(256) void k_isa_store_rec(const u32 *__restrict__ rec, u64 cnt,
                                                       P *__restrict__ isa) {
  constexpr int W = 1 + (int) (sizeof(P) / 4);
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt) return;
  const u32 *r = rec + i * W;
  u64 v = r[1];
  if (sizeof(P) == 8) v |= (u64) r[2] << 32;
  isa[r[0]] = (P) v;
}

// ---- one exchange per round for the new ranks of the last round AND the queries
// of this one.  The message for part d: its update records (offset, rank), then
// its query offsets, 32-bit words.  FuseTab (by value): for the SEND side the
// words before part d's message, its updates, its queries; the same for the
// RECEIVE side per source part.
struct FuseTab {
  u32 n;                          // parts
  u32 upre[DEST_MAXPARTS + 1];    // updates for / from the parts before
  u32 qpre[DEST_MAXPARTS + 1];    // queries ...
  u64 wpre[DEST_MAXPARTS + 1];    // message words ...
};
__device__ __forceinline__ u32 fuse_part(const u32 *pre, u32 n, u32 i) {
  u32 lo = 0, hi = n;             // largest d with pre[d] <= i
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (pre[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}
// soff/srank: the bucketed updates (part-major), sq: the bucketed query offsets
template <typename P>
__global__ __launch_bounds__(256) void k_fuse_pack(FuseTab tab, const u32 *__restrict__ soff,
                                                   const P *__restrict__ srank,
                                                   const u32 *__restrict__ sq,
                                                   u32 *__restrict__ msg) {
  constexpr int W = 1 + (int) (sizeof(P) / 4);
  const u32 nu = tab.upre[tab.n], nq = tab.qpre[tab.n];
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  if (i < nu) {
    const u32 d = fuse_part(tab.upre, tab.n, (u32) i);
    u32 *rec = msg + tab.wpre[d] + (u64) ((u32) i - tab.upre[d]) * W;
    const u64 r = (u64) srank[i];
    rec[0] = soff[i];
    rec[1] = (u32) r;
    if (sizeof(P) == 8) rec[2] = (u32) (r >> 32);
  } else if (i < (u64) nu + nq) {
    const u32 k = (u32) (i - nu);
    const u32 d = fuse_part(tab.qpre, tab.n, k);
    const u32 ud = tab.upre[d + 1] - tab.upre[d];
    msg[tab.wpre[d] + (u64) ud * W + (k - tab.qpre[d])] = sq[k];
  }
}
// received messages: first all updates ...
template <typename P>
__global__ __launch_bounds__(256) void k_fuse_updates(FuseTab tab, const u32 *__restrict__ msg,
                                                      P *__restrict__ isa) {
  constexpr int W = 1 + (int) (sizeof(P) / 4);
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  if (i >= tab.upre[tab.n]) return;
  const u32 s = fuse_part(tab.upre, tab.n, (u32) i);
  const u32 *rec = msg + tab.wpre[s] + (u64) ((u32) i - tab.upre[s]) * W;
  u64 v = rec[1];
  if (sizeof(P) == 8) v |= (u64) rec[2] << 32;
  isa[rec[0]] = (P) v;
}
// ... then the answers to the queries, source by source
template <typename P>
__global__ __launch_bounds__(256) void k_fuse_answers(FuseTab tab, const u32 *__restrict__ msg,
                                                      const P *__restrict__ isa,
                                                      P *__restrict__ ans) {
  constexpr int W = 1 + (int) (sizeof(P) / 4);
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  if (i >= tab.qpre[tab.n]) return;
  const u32 s = fuse_part(tab.qpre, tab.n, (u32) i);
  const u32 us = tab.upre[s + 1] - tab.upre[s];
  ans[i] = isa[msg[tab.wpre[s] + (u64) us * W + ((u32) i - tab.qpre[s])]];
}

template <typename P>
__global__ __launch_bounds__(256) void k_answer(const u32 *__restrict__ q, u64 cnt,
                                                const P *__restrict__ isa,
                                                P *__restrict__ ans) {
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  if (i < cnt) ans[i] = isa[q[i]];
}

template <typename P>
__global__ __launch_bounds__(256) void k_k2_scatter(const P *__restrict__ ans,
                                                    const u32 *__restrict__ order, u64 m,
                                                    P *__restrict__ k2) {
  const u64 s = (u64) blockIdx.x * 256 + threadIdx.x;
  if (s < m) k2[order[s]] = ans[s];
}

// positions of the slice at full width: low half from the sort's values, high
// half from the spare bits of the sorted keys
template <int BITS>
__global__ __launch_bounds__(256) void k_wide_positions(const u64 *__restrict__ keys,
                                                        const u32 *__restrict__ lo, u64 n,
                                                        u64 *__restrict__ out) {
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (Key<BITS>::poshi(keys[i]) << 32) | lo[i];
}

// ---------------------------------------------------------------------------
// bucket table (.bck): bcktab.c:55-81,519-577, sfx-suffixer.c:379-383,476-516
// ---------------------------------------------------------------------------
// code of the first k symbols in base sigma; a prefix shorter than k letters is
// padded with the largest letter (the key pads with 1-bits)
template <int BITS>
__device__ __forceinline__ u64 bck_code(u64 key, u32 k, u32 sigma) {
  if (BITS == 2) return k == 0 ? 0 : key >> (64 - 2 * k);
  u64 code = 0;
  for (u32 j = 0; j < k; j++) {
    u32 d = (u32) (key >> (64 - BITS * (j + 1))) & ((1u << BITS) - 1u);
    if (d >= sigma) d = sigma - 1;
    code = code * sigma + d;
  }
  return code;
}

// hist[c] = entries whose padded k-code is c (suffixes that start with a
// special are not in any bucket); the keys are sorted, so a wave adds one count
// per run of equal codes.  Suffixes with fewer than k letters in front of a
// special are counted per padded (k-1)-prefix and, for 1..k-2 letters, per
// prefix of exactly that length.
template <int BITS>
__global__ __launch_bounds__(256) void k_bck_count(
    const u64 *__restrict__ keys, u64 N, u32 k, u32 sigma, u32 *__restrict__ hist,
    u32 *__restrict__ countspecial, u32 *__restrict__ distpfx) {
  using K = Key<BITS>;
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = i < N;
  const u64 key = valid ? keys[i] : ~0ull;
  const u32 dc = K::dcode(key);
  const bool inbucket = valid && dc != K::DMAX;
  const u64 code = inbucket ? bck_code<BITS>(key, k, sigma) : ~0ull;
  const u64 prev = __shfl_up(code, 1, 64);
  const bool head = inbucket && (lane == 0 || prev != code);
  const u64 heads = __ballot(head), live = __ballot(inbucket);
  if (head) {
    // entries of this run inside the wave: up to the next head
    const u64 above = lane == 63 ? 0 : (heads >> (lane + 1)) << (lane + 1);
    const int end = above ? __ffsll((unsigned long long) above) - 1 : 64;
    u64 mask = end == 64 ? ~0ull : ((1ull << end) - 1ull);
    mask &= ~((1ull << lane) - 1ull);
    atomicAdd(&hist[code], (u32) __popcll(live & mask));
  }
  if (inbucket && dc != 0) {
    const u32 letters = K::letters(key);
    if (letters < k) {
      atomicAdd(&countspecial[bck_code<BITS>(key, k - 1, sigma)], 1u);
      if (letters >= 1 && letters + 2 <= k) {
        u64 off = 0, pw = sigma;
        for (u32 l = 1; l < letters; l++) { off += pw; pw *= sigma; }
        atomicAdd(&distpfx[off + bck_code<BITS>(key, letters, sigma)], 1u);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// finalize
// ---------------------------------------------------------------------------
struct Stats {          // device-side accumulators
  unsigned long long numties;     // entries tied with their predecessor
  unsigned long long lcpsum;      // masked sum
  unsigned long long numlarge;    // lcp >= 255
  unsigned long long longest;     // index of suffix 0
  u32 maxlcp;
  u32 count;                      // generic counter (compaction totals)
  u32 count2;                     // second counter (deferred elements)
  u32 dmax;                       // direct tie path: max lcp
  unsigned long long dsum;        // direct tie path: lcp sum
  u32 dfallback;                  // direct tie path gave up on some group
  u32 count3;                     // third counter (small groups)
  unsigned long long smalldone;   // entries of the small groups settled directly
  unsigned long long crowded;     // MSD sort: entries of the runs k_msd_local read and left to
                                  // k_msd_local_radix (a crowded bin): read, not written by it
};

constexpr int FIN_THREADS = 256;
constexpr int FIN_PER_THREAD = 4;
constexpr int FIN_TILE = FIN_THREADS * FIN_PER_THREAD * 4;  // 4096 entries

// "entry i has the same sorted key bits as entry i-1" for every entry, and the
// number of such entries: all the refinement needs from the sorted keys, so the
// table emission (k_finalize) can run beside it on a second stream
template <int BITS>
__global__ __launch_bounds__(256) void k_tiebits(const u64 *__restrict__ keys,
                                                 u64 N, u64 *__restrict__ tiebits,
                                                 Stats *stats) {
  using K = Key<BITS>;
  __shared__ u32 s_cnt[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // a wave walks 16 consecutive groups of 64 entries: all loads are issued up
  // front, and the key in front of a group is the last lane of the group before.
  // The workgroups stride over the tiles and add to the one global counter once
  // at their end: an atomic per tile (732 K at 3 Gbp, ~12 ns each on one
  // address) used to cost more than the 24 GB this kernel reads.
  u32 cnt = 0;
  const u64 ntiles = (N + 4095) / 4096;
  for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const u64 wbase = tile * 4096 + (u64) w * 1024;
  u64 k[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const u64 i = wbase + (u64) r * 64 + lane;
    k[r] = i < N ? keys[i] : ~0ull;
  }
  u64 before = (wbase > 0 && wbase < N) ? keys[wbase - 1] : ~k[0];   // wave-uniform
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const u64 i = wbase + (u64) r * 64 + lane;
    u64 prev = __shfl_up(k[r], 1, 64);
    if (lane == 0) prev = before;
    before = __shfl(k[r], 63, 64);
    const bool tie = i > 0 && i < N && (k[r] >> K::DSHIFT) == (prev >> K::DSHIFT) &&
                     K::dcode(k[r]) == 0;
    const u64 m = __ballot(tie);
    if (lane == 0 && i < N) { tiebits[i >> 6] = m; cnt += (u32) __popcll(m); }
  }
  }
  if (lane == 0) s_cnt[w] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (t) atomicAdd(&stats->numties, (unsigned long long) t);
  }
}

// Each thread owns 4 consecutive entries so that .lcp/.bwt leave as one
// 32-bit store and .suf as two 16-byte stores per lane.
template <int BITS, typename P>
__global__ __launch_bounds__(FIN_THREADS) void k_finalize(
    const u64 *__restrict__ keys, const P *__restrict__ pos, u64 N,
    u32 prefixlength, u64 *__restrict__ suf, u8 *__restrict__ lcp,
    u8 *__restrict__ bwt, u64 *__restrict__ tiebits, Stats *stats,
    u64 prev_key, int has_prev, u64 index_offset) {
  // prev_key: key of the entry in front of this slice (part builds): the last
  // key of the preceding lexicographic range, needed for the LCP of entry 0
  using K = Key<BITS>;
  __shared__ unsigned long long s_sum[FIN_THREADS / 64];
  __shared__ unsigned long long s_ties[FIN_THREADS / 64];
  __shared__ u32 s_max[FIN_THREADS / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long sum = 0, ties = 0;
  u32 mx = 0;
  // (workgroups stride over the tiles: one set of atomics per workgroup, not
  // per tile -- see k_tiebits)
  const u64 ntiles = (N + FIN_TILE - 1) / FIN_TILE;
  for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
  for (int r = 0; r < FIN_PER_THREAD; r++) {
    const u64 i0 = tile * FIN_TILE + ((u64) r * FIN_THREADS + threadIdx.x) * 4;
    u64 k[4];
    P p[4];
    if (i0 + 4 <= N) {
      const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(keys + i0);
      const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(keys + i0 + 2);
      k[0] = a.x; k[1] = a.y; k[2] = b.x; k[3] = b.y;
      if (sizeof(P) == 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(pos + i0);
        p[0] = (P) q.x; p[1] = (P) q.y; p[2] = (P) q.z; p[3] = (P) q.w;
      } else {
        const ulonglong2 q0 = *reinterpret_cast<const ulonglong2 *>(pos + i0);
        const ulonglong2 q1 = *reinterpret_cast<const ulonglong2 *>(pos + i0 + 2);
        p[0] = (P) q0.x; p[1] = (P) q0.y; p[2] = (P) q1.x; p[3] = (P) q1.y;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        k[c] = i0 + c < N ? keys[i0 + c] : ~0ull;
        p[c] = i0 + c < N ? pos[i0 + c] : (P) 0;
      }
    }
    // predecessor key: previous lane's last key, or a global load at the
    // wave's left edge
    u64 prevk = __shfl_up(k[3], 1, 64);
    if (lane == 0) prevk = i0 == 0 ? prev_key : (i0 < N + 4 ? keys[i0 - 1] : ~0ull);
    u32 lcpv[4], tiemask = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const u64 i = i0 + c;
      const u64 a = c == 0 ? prevk : k[c - 1], b = k[c];
      const u32 da = K::letters(a), db = K::letters(b);
      const u64 x = (a ^ b) >> K::LOW_BITS;
      u32 m = x ? (u32) (__clzll((long long) x) - K::LOW_BITS) / BITS
                : (u32) K::SYMS;
      u32 l = m < da ? m : da;
      l = l < db ? l : db;
      bool tie = (m == (u32) K::SYMS) && (K::dcode(a) == 0) && (K::dcode(b) == 0);
      if ((i == 0 && !has_prev) || i >= N) l = 0;
      if (i == 0 || i >= N) tie = false;   // equal keys never straddle a range border
      lcpv[c] = l;
      if (tie) { tiemask |= 1u << c; ties++; }
      else if (i < N) {
        mx = l > mx ? l : mx;
        if (db >= prefixlength) sum += l;
        if (p[c] == 0) stats->longest = index_offset + i;
      }
    }
    if (i0 + 4 <= N) {
      if (suf != nullptr) {
        ulonglong2 s0, s1;
        s0.x = p[0]; s0.y = p[1]; s1.x = p[2]; s1.y = p[3];
        *reinterpret_cast<ulonglong2 *>(suf + i0) = s0;
        *reinterpret_cast<ulonglong2 *>(suf + i0 + 2) = s1;
      }
      if (lcp != nullptr)
        *reinterpret_cast<u32 *>(lcp + i0) =
            lcpv[0] | (lcpv[1] << 8) | (lcpv[2] << 16) | (lcpv[3] << 24);
      if (bwt != nullptr) {
        u32 bw = 0;
#pragma unroll
        for (int c = 0; c < 4; c++)
          bw |= (u32) Pay<BITS>::to_bwt((u32) (k[c] & K::PAY_MASK)) << (8 * c);
        *reinterpret_cast<u32 *>(bwt + i0) = bw;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if (i0 + c < N) {
          if (suf != nullptr) suf[i0 + c] = p[c];
          if (lcp != nullptr) lcp[i0 + c] = (u8) lcpv[c];
          if (bwt != nullptr)
            bwt[i0 + c] = Pay<BITS>::to_bwt((u32) (k[c] & K::PAY_MASK));
        }
      }
    }
    // tie bits: 4 per lane, 16 lanes per 64-bit word
    u64 tw = (u64) tiemask << (4 * (lane & 15));
    tw |= __shfl_xor(tw, 1, 64);
    tw |= __shfl_xor(tw, 2, 64);
    tw |= __shfl_xor(tw, 4, 64);
    tw |= __shfl_xor(tw, 8, 64);
    if (tiebits != nullptr && (lane & 15) == 0 && i0 < N) tiebits[i0 >> 6] = tw;
  }
  }
  // block reduction of the statistics, one atomic each per block
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    sum += __shfl_xor(sum, d, 64);
    ties += __shfl_xor(ties, d, 64);
    const u32 o = __shfl_xor(mx, d, 64);
    mx = o > mx ? o : mx;
  }
  if (lane == 0) { s_sum[w] = sum; s_ties[w] = ties; s_max[w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long S = 0, T = 0;
    u32 M = 0;
    for (int i = 0; i < FIN_THREADS / 64; i++) {
      S += s_sum[i]; T += s_ties[i]; M = s_max[i] > M ? s_max[i] : M;
    }
    if (S) atomicAdd(&stats->lcpsum, S);
    if (T && tiebits != nullptr) atomicAdd(&stats->numties, T);
    if (M) atomicMax(&stats->maxlcp, M);
  }
}

// ---------------------------------------------------------------------------
// tied suffixes: unresolved list, group heads, rank table
// ---------------------------------------------------------------------------
// P is the type of text positions and of ranks: u32, or u64 in a part build of
// a sequence with n >= 2^32 (GTAMD_FORCE_WIDE=1 takes it at any size).  Indices
// into a part's slice stay 32-bit (a slice has fewer than 2^32 entries).

// a workgroup of 256 threads owns 256 words (16384 entries) of the bitmap: it
// delivers ONE count (the offsets inside are a block scan away for the kernels
// that place things -- scans over one counter per word were 0.65 ms each, five
// of them per build) and, per word, the highest "not tied" index
__device__ __forceinline__ u32 block_total_256(u32 v, u32 *s4) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if (lane == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  const u32 t = s4[0] + s4[1] + s4[2] + s4[3];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void k_tie_words(
    const u64 *__restrict__ tiebits, u64 nwords, u32 *__restrict__ blkcnt,
    u32 *__restrict__ headw) {
  __shared__ u32 s4[4];
  const u64 w = (u64) blockIdx.x * 256 + threadIdx.x;
  u32 c = 0;
  if (w < nwords) {
    const u64 t = tiebits[w];
    const u64 nx = w + 1 < nwords ? tiebits[w + 1] : 0ull;
    c = (u32) __popcll(t | (t >> 1) | (nx << 63));
    const u64 z = ~t;
    headw[w] = z ? (u32) (w * 64 + (63 - __clzll((long long) z))) : 0u;
  }
  const u32 tot = block_total_256(c, s4);
  if (threadIdx.x == 0) blkcnt[blockIdx.x] = tot;
}

// head (first index) of the tie group that entry i belongs to
__device__ __forceinline__ u32 group_head(const u64 *tiebits, const u32 *carry,
                                          u64 i) {
  const u64 w = i >> 6;
  const int b = (int) (i & 63);
  const u64 below = b == 63 ? ~0ull : ((2ull << b) - 1ull);
  const u64 z = ~tiebits[w] & below;
  return z ? (u32) (w * 64 + (63 - __clzll((long long) z))) : carry[w];
}

// one thread per 64-entry word of the bitmap (few words have an unresolved
// entry at all: a lane per entry left most lanes idle -- 4.6 ms for 50 M
// entries of 3 G); the threads of a wave write neighbouring stretches of the
// list.  blkoff: exclusive scan of k_tie_words' workgroup counts.
template <typename P>
__global__ __launch_bounds__(256) void k_unres_emit(
    const u64 *__restrict__ tiebits, u64 nwords, const u32 *__restrict__ blkoff,
    const u32 *__restrict__ carry, const P *__restrict__ sa,
    u32 *__restrict__ uidx0, u32 *__restrict__ uidx, P *__restrict__ upos,
    u32 *__restrict__ ugrp) {
  __shared__ u32 s_scan[4];
  const u64 w = (u64) blockIdx.x * 256 + threadIdx.x;
  u64 t = 0, u = 0;
  if (w < nwords) {
    t = tiebits[w];
    const u64 nx = w + 1 < nwords ? tiebits[w + 1] : 0ull;
    u = t | (t >> 1) | (nx << 63);
  }
  u32 tot;
  u32 j = blkoff[blockIdx.x] + block_scan_excl_sum((u32) __popcll(u), &tot, s_scan);
  if (u == 0) return;
  const u32 carryw = carry[w];
  while (u) {
    const int b = __ffsll((unsigned long long) u) - 1;
    u &= u - 1;
    const u64 i = w * 64 + b;
    // head of i's group: highest "not tied" entry at or below i
    const u64 below = b == 63 ? ~0ull : ((2ull << b) - 1ull);
    const u64 z = ~t & below;
    uidx0[j] = (u32) i;
    uidx[j] = (u32) i;
    upos[j] = sa[i];
    ugrp[j] = z ? (u32) (w * 64 + (63 - __clzll((long long) z))) : carryw;
    j++;
  }
}

// The rank table ("ISA" of the first sort, refined in place by the doubling
// rounds): rank[sa[i]] = rank_offset + head of i's group.  Written directly it
// is N random 4-byte stores, each costing a 32-byte HBM write (measured: 94 GB
// for 12 GB of payload).  So: (1) heads in suffix order, streaming;
// (2) one radix pass partitions the (position, head) pairs by the leading
// position bits; (3) the scatter then walks one position window after the
// other and its stores meet in cache before they reach HBM.
__global__ __launch_bounds__(256) void k_heads(
    const u64 *__restrict__ tiebits, const u32 *__restrict__ carry, u64 N,
    u32 rank_offset, u32 *__restrict__ heads) {
  // four consecutive entries per thread (one 16-byte store): they share their
  // bitmap word, the head of an entry is the head of the one before it unless
  // it starts a group itself
  const u64 i0 = ((u64) blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= N) return;
  const u64 t = tiebits[i0 >> 6];
  u32 h[4];
  h[0] = group_head(tiebits, carry, i0);
#pragma unroll
  for (int g = 1; g < 4; g++)
    h[g] = ((t >> ((i0 + g) & 63)) & 1ull) ? h[g - 1] : (u32) (i0 + g);
  if (i0 + 4 <= N) {
    *reinterpret_cast<uint4 *>(heads + i0) =
        make_uint4(rank_offset + h[0], rank_offset + h[1], rank_offset + h[2],
                   rank_offset + h[3]);
  } else {
    for (int g = 0; g < 4 && i0 + g < N; g++) heads[i0 + g] = rank_offset + h[g];
  }
}

// The rank table through LDS, for a build that owns ALL positions: after the
// pairs are partitioned on their leading position bits, bucket b holds exactly
// the positions [b << wb, (b + 1) << wb) -- and, every position occurring once,
// exactly the pairs with these indices.  A workgroup scatters one window of
// heads inside LDS (4-byte LDS stores cost nothing against 4-byte global
// stores, each of which is a memory transaction of its own: 47 G/s measured)
// and writes the window out in whole lines.  With wb = 16 (N >= 2^31) a
// window is twice the LDS; two workgroups then read the bucket and keep one
// half each (they are eight workgroups apart, i.e. on the same XCD, so that
// the second read of the pairs can hit that L2).
constexpr int RW_BITS = 15;
constexpr int RW_THREADS = 1024;

// winlist != nullptr: only some windows are built (see below) -- windows of 2^fb
// positions, renumbered in ascending order to "compact positions" (k_win_filter):
// bucket k of the partitioned pairs holds the compact positions [k << wb, ...), i.e.
// the selected windows k << (wb - fb) ... of the list, and every one of them goes
// to its own place in the table
__global__ __launch_bounds__(RW_THREADS) void k_rank_window(
    const u32 *__restrict__ pos, const u32 *__restrict__ heads, u64 N, int wb,
    int split, u32 nbuckets, u32 *__restrict__ rank, const u32 *__restrict__ winlist,
    int fb, u32 nfine, u64 M) {
  __shared__ u32 s_win[1 << RW_BITS];
  u32 bucket = blockIdx.x, half = 0;
  if (split == 2) {
    bucket = (blockIdx.x >> 4) * 8u + (blockIdx.x & 7u);
    half = (blockIdx.x >> 3) & 1u;
  }
  if (bucket >= nbuckets) return;   // whole workgroup
  const int sb = wb - (split == 2 ? 1 : 0);   // position bits inside a window
  const u32 smask = (1u << sb) - 1u;
  const u64 first = (u64) bucket << wb;                    // where the window's pairs are
  const u64 total = winlist != nullptr ? M : N;
  const u64 end = first + (1ull << wb) < total ? first + (1ull << wb) : total;
  const u64 cnt = end - first;
  const u32 *bp = pos + first, *bh = heads + first;
  // four pairs per lane and step while whole groups are left (first is a
  // multiple of 4 whenever wb >= 2; both arrays are 16-byte aligned then)
  const u64 cnt4 = (wb >= 2) ? (cnt & ~3ull) : 0;
  for (u64 i = (u64) threadIdx.x * 4; i < cnt4; i += (u64) RW_THREADS * 4) {
    const uint4 p4 = *reinterpret_cast<const uint4 *>(bp + i);
    const uint4 h4 = *reinterpret_cast<const uint4 *>(bh + i);
    if (split == 1 || ((p4.x >> sb) & 1u) == half) s_win[p4.x & smask] = h4.x;
    if (split == 1 || ((p4.y >> sb) & 1u) == half) s_win[p4.y & smask] = h4.y;
    if (split == 1 || ((p4.z >> sb) & 1u) == half) s_win[p4.z & smask] = h4.z;
    if (split == 1 || ((p4.w >> sb) & 1u) == half) s_win[p4.w & smask] = h4.w;
  }
  for (u64 i = cnt4 + threadIdx.x; i < cnt; i += RW_THREADS) {
    const u32 p = bp[i];
    if (split == 1 || ((p >> sb) & 1u) == half) s_win[p & smask] = bh[i];
  }
  __syncthreads();
  const u64 wfirst = first + ((u64) half << sb);
  if (wfirst >= total) return;
  if (winlist != nullptr) {
    // (fb <= sb: whole selected windows; the last window of the text may be short)
    const u32 f0 = (u32) (wfirst >> fb), fmask = (1u << fb) - 1u;
    if (fb >= 2) {
      for (u32 i = threadIdx.x * 4u; i < (1u << sb); i += RW_THREADS * 4u) {
        const u32 fw = f0 + (i >> fb);
        if (fw >= nfine) break;
        const u64 dst = ((u64) winlist[fw] << fb) + (i & fmask);
        if (dst + 4 <= N) *reinterpret_cast<uint4 *>(rank + dst) = *reinterpret_cast<const uint4 *>(s_win + i);
        else
          for (int g = 0; g < 4 && dst + g < N; g++) rank[dst + g] = s_win[i + g];
      }
    } else {
      for (u32 i = threadIdx.x; i < (1u << sb); i += RW_THREADS) {
        const u32 fw = f0 + (i >> fb);
        if (fw >= nfine) break;
        const u64 dst = ((u64) winlist[fw] << fb) + (i & fmask);
        if (dst < N) rank[dst] = s_win[i];
      }
    }
    return;
  }
  const u64 wend = wfirst + (1ull << sb) < N ? wfirst + (1ull << sb) : N;
  const u64 wcnt = wend - wfirst;
  u32 *out = rank + wfirst;
  const u64 wcnt4 = (sb >= 2) ? (wcnt & ~3ull) : 0;
  for (u64 i = (u64) threadIdx.x * 4; i < wcnt4; i += (u64) RW_THREADS * 4)
    *reinterpret_cast<uint4 *>(out + i) = *reinterpret_cast<const uint4 *>(s_win + i);
  for (u64 i = wcnt4 + threadIdx.x; i < wcnt; i += RW_THREADS) out[i] = s_win[i];
}

// Only the windows the doubling rounds can touch are built.  After the pair
// path the suffixes still tied are a small part of the text (1.7 % of the
// human-like 3 Gbp: high-copy repeat families), and a round reads rank[p + h]
// and writes rank[p] for those suffixes p only -- positions that cluster: in
// half of the windows of 64 K positions, but in a fourteenth of the windows of
// 8 K (RW_FINE).  So: mark the windows of p .. p + H for the first rounds'
// offsets, keep the (position, head) pairs of marked windows (one streaming
// pass over the suffix array), partition and scatter only those.  The
// granularity of the selection is independent of the size of the LDS window:
// the positions of the selected windows are renumbered densely ("compact
// positions": number of the window in the list << fb | offset inside), the
// partition and the LDS scatter work on those, and k_rank_window sends every
// window of 2^fb entries to its place in the table.  A later round whose offset
// reaches an unbuilt window has it built first (k_win_check; never happens
// while h <= H).  Whole table 38 ms -> windows of 64 K 20 ms -> of 8 K 6 ms at
// 3 Gbp.
constexpr int RW_FINE = 13;
template <typename P>
__global__ __launch_bounds__(256) void k_win_mark(const P *__restrict__ upos, u64 m, u64 lo,
                                                  u64 hi, int wb, u64 nwin,
                                                  u32 *__restrict__ need) {
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const u64 p = (u64) upos[j];
  u64 w0 = (p + lo) >> wb, w1 = (p + hi) >> wb;
  if (w1 >= nwin) w1 = nwin - 1;
  for (u64 w = w0; w <= w1; w++)
    if (!((need[w >> 5] >> (w & 31)) & 1u)) atomicOr(&need[w >> 5], 1u << (w & 31));
}

// does this round's offset reach a window that has not been built?
template <typename P>
__global__ __launch_bounds__(256) void k_win_check(const P *__restrict__ upos, u64 m, u64 h,
                                                   int wb, u64 nwin,
                                                   const u32 *__restrict__ built,
                                                   u32 *__restrict__ need, Stats *stats) {
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  u64 w = ((u64) upos[j] + h) >> wb;
  if (w >= nwin) w = nwin - 1;
  if (!((built[w >> 5] >> (w & 31)) & 1u)) {
    if (!((need[w >> 5] >> (w & 31)) & 1u)) atomicOr(&need[w >> 5], 1u << (w & 31));
    stats->count2 = 1;
  }
}

// one workgroup: the windows that are needed and not built yet, in ascending
// order (winlist), as a bitmap (sel) with the number of selected windows before
// every word (pref), their number; they count as built from here on
__global__ __launch_bounds__(1024) void k_win_select(const u32 *__restrict__ need,
                                                     u32 *__restrict__ built, u64 nwords,
                                                     u32 *__restrict__ sel,
                                                     u32 *__restrict__ winlist,
                                                     u32 *__restrict__ pref, Stats *stats) {
  __shared__ u32 s_scan[16];
  const u64 per = (nwords + 1023) / 1024;
  const u64 w0 = (u64) threadIdx.x * per;
  const u64 w1 = w0 + per < nwords ? w0 + per : nwords;
  u32 cnt = 0;
  for (u64 w = w0; w < w1; w++) cnt += (u32) __popc(need[w] & ~built[w]);
  u32 tot;
  u32 at = block_scan_excl<SCAN_SUM, 1024>(cnt, &tot, s_scan);
  for (u64 w = w0; w < w1; w++) {
    u32 x = need[w] & ~built[w];
    sel[w] = x;
    pref[w] = at;        // selected windows before this word
    built[w] |= x;
    while (x) {
      const int b = __ffs(x) - 1;
      x &= x - 1;
      winlist[at++] = (u32) (w * 32 + b);
    }
  }
  if (threadIdx.x == 0) stats->count = tot;
}

// entries of the suffix array whose position lies in a selected window, together
// with the heads of their tie groups, in any order (they are partitioned by
// position next).  A workgroup walks WF_ITER spans of 16384 entries:
//  * every lane loads four consecutive entries per step (the wave reads 1 KB in
//    one piece), all loads of a span are issued before the first is used;
//  * the bitmap of the selected windows is in LDS (LSEL; 45 KB for the windows of
//    8 K positions of a 3 Gbp text): 3 G look-ups at random words through the
//    vector memory path cost 6 ms at one lane per cycle, from LDS they disappear
//    behind the loads (tools/microbench/winfilter.hip: read only 1.8 ms, with
//    global look-ups 7.7, with LDS look-ups 1.8);
//  * the selected entries of a span go through a queue in LDS and leave it one per
//    thread: dense stores, no idle lanes, nothing loaded twice -- with the entries
//    written where the threads found them (a lane has one selected entry of sixteen)
//    every 4-byte store went to memory as a write of its own (14 GB written for
//    1.7 GB of entries, PMC) and the kernel took 9-12 ms instead of 3.5;
//  * one atomic on the list's cursor per span reserves the room.
// P: positions of 32 bits, or of 64 (part builds of n >= 2^32); cap: room in the
// list -- a workgroup that would write behind it stops and sets stats->count2 (the
// caller takes another way then); pref != nullptr: the list gets compact positions
// (see k_rank_window) instead of positions.
constexpr int WF_THREADS = 1024;
constexpr int WF_Q = 4;                                   // uint4 loads per thread
constexpr u64 WF_SPAN = (u64) WF_THREADS * 4 * WF_Q;      // 16384 entries
constexpr int WF_ITER = 8;
constexpr int WF_QUEUE = 2048;                            // entries of the queue
constexpr u64 WF_LDS_MAX = 60 * 1024;      // bitmap + prefix counts a workgroup takes to LDS
template <typename P>
__device__ __forceinline__ void wf_load4(const P *__restrict__ sa, u64 i0, u64 NL, P v[4]) {
  if (i0 + 4 <= NL) {
    if (sizeof(P) == 4) {
      const uint4 x = *reinterpret_cast<const uint4 *>(sa + i0);
      v[0] = (P) x.x; v[1] = (P) x.y; v[2] = (P) x.z; v[3] = (P) x.w;
    } else {
      const ulonglong2 v0 = *reinterpret_cast<const ulonglong2 *>(sa + i0);
      const ulonglong2 v1 = *reinterpret_cast<const ulonglong2 *>(sa + i0 + 2);
      v[0] = (P) v0.x; v[1] = (P) v0.y; v[2] = (P) v1.x; v[3] = (P) v1.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = i0 + k < NL ? sa[i0 + k] : (P) 0;
  }
}
// dynamic LDS: [LSEL: bitmap padded to groups of four words | selected windows before
// every group] | queue of positions | queue of indices inside the span
template <typename P>
static size_t wf_lds_bytes(u64 nww, bool lsel) {
  const u64 nww4 = (nww + 3) & ~3ull;
  return (size_t) ((lsel ? nww4 * 4 + nww4 : 0) + 16 + WF_QUEUE * (sizeof(P) + 2));
}
template <typename P, bool LSEL>
__global__ __launch_bounds__(WF_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_win_filter(
    const P *__restrict__ sa, u64 NL, int wb, const u32 *__restrict__ sel, u32 nww,
    const u32 *__restrict__ pref, const u64 *__restrict__ tiebits, const u32 *__restrict__ carry,
    P *__restrict__ fpos, u32 *__restrict__ fhead, u64 cap, Stats *stats) {
  extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];
  __shared__ u32 s_scan[WF_THREADS / 64];
  __shared__ u32 s_base;
  const u32 nww4 = (nww + 3u) & ~3u;
  u32 *s_sel = s_dyn, *s_pref4 = s_dyn + nww4;
  P *s_qpos = reinterpret_cast<P *>(s_dyn + (LSEL ? nww4 + nww4 / 4 : 0u) + ((LSEL ? nww4 / 4 : 0u) & 1u));
  unsigned short *s_qidx = reinterpret_cast<unsigned short *>(s_qpos + WF_QUEUE);
  if (LSEL) {
    for (u32 i = threadIdx.x; i < nww4; i += WF_THREADS) s_sel[i] = i < nww ? sel[i] : 0u;
    if (pref != nullptr)
      for (u32 i = threadIdx.x; i < nww4 / 4; i += WF_THREADS) s_pref4[i] = pref[4 * i];
    __syncthreads();
  }
  const u32 *selw = LSEL ? s_sel : sel;
  for (int it = 0; it < WF_ITER; it++) {
    const u64 first = ((u64) blockIdx.x * WF_ITER + it) * WF_SPAN;
    if (first >= NL) break;       // (whole workgroup)
    P p[WF_Q][4];
#pragma unroll
    for (int q = 0; q < WF_Q; q++)
      wf_load4(sa, first + (u64) q * (WF_THREADS * 4) + (u64) threadIdx.x * 4, NL, p[q]);
    u32 m = 0;
#pragma unroll
    for (int q = 0; q < WF_Q; q++) {
      const u64 i0 = first + (u64) q * (WF_THREADS * 4) + (u64) threadIdx.x * 4;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 w = (u32) (p[q][k] >> wb);      // (2^40 positions, windows of >= 2^13)
        if (i0 + k < NL) m |= ((selw[w >> 5] >> (w & 31)) & 1u) << (4 * q + k);
      }
    }
    u32 tot;
    const u32 excl = block_scan_excl<SCAN_SUM, WF_THREADS>((u32) __popc(m), &tot, s_scan);
    if (threadIdx.x == 0) {
      u32 b = tot ? atomicAdd(&stats->count, tot) : 0u;
      if ((u64) b + tot > cap) { stats->count2 = 1u; b = ~0u; }
      s_base = b;
    }
    for (u32 r0 = 0; r0 < tot; r0 += WF_QUEUE) {      // (one round unless a span has many)
      u32 o = excl - r0;          // (wraps for the entries of earlier rounds: then o >= WF_QUEUE)
#pragma unroll
      for (int q = 0; q < WF_Q; q++) {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if ((m >> (4 * q + k)) & 1u) {
            if (o < (u32) WF_QUEUE) {
              s_qpos[o] = p[q][k];
              s_qidx[o] = (unsigned short) (q * (WF_THREADS * 4) + threadIdx.x * 4 + k);
            }
            o++;
          }
      }
      __syncthreads();
      if (s_base == ~0u) return;          // (whole workgroup)
      const u32 nq = tot - r0 < (u32) WF_QUEUE ? tot - r0 : (u32) WF_QUEUE;
      const u64 base = (u64) s_base + r0;
      for (u32 j = threadIdx.x; j < nq; j += WF_THREADS) {
        P v = s_qpos[j];
        if (pref != nullptr) {
          // the compact position: windows before this one in the list | offset
          const u32 w = (u32) (v >> wb), x = w >> 5;
          u32 d;
          if (LSEL) {
            // (prefix counts of every fourth word; the words between by popcount)
            const uint4 g = *reinterpret_cast<const uint4 *>(s_sel + (x & ~3u));
            const u32 low = (1u << (w & 31)) - 1u, jj = x & 3u;
            d = s_pref4[x >> 2] + (u32) __popc(g.x & (jj == 0 ? low : ~0u)) +
                (u32) __popc(g.y & (jj == 1 ? low : (jj > 1 ? ~0u : 0u))) +
                (u32) __popc(g.z & (jj == 2 ? low : (jj > 2 ? ~0u : 0u))) +
                (u32) __popc(g.w & (jj == 3 ? low : 0u));
          } else {
            d = pref[x] + (u32) __popc(sel[x] & ((1u << (w & 31)) - 1u));
          }
          v = (P) (((u64) d << wb) | ((u64) v & ((1ull << wb) - 1ull)));
        }
        fpos[base + j] = v;
        fhead[base + j] = group_head(tiebits, carry, first + s_qidx[j]);
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------
// tie groups of exactly two suffixes ("pairs"): 87 % of the tied suffixes of
// the human-like workload -- every low-copy repeat makes them -- and the ones
// that stay tied longest under prefix doubling (an exact 8 K copy takes nine
// rounds).  Two suffixes need no rank table: ONE comparison on the packed
// text gives their order and their LCP, and walked in TEXT order the next pair
// on the same diagonal (a+1, b+1) starts from the previous LCP minus one
// (Kasai et al.; src/match/sfx-linlcp.c:74-129), so only the first pair of a
// chunk pays for a long match.  The pairs leave the tie bitmap before the
// unresolved list and the rank table are built: everything behind sees them
// as settled suffixes.
// ---------------------------------------------------------------------------
constexpr int LCP_CHUNK = 32;
constexpr u32 PAIR_SWAP = 1u << 31;

// pair heads of a bitmap word: entry i not tied, i+1 tied, i+2 not tied
__device__ __forceinline__ u64 pair_heads(u64 t, u64 nx) {
  return ~t & ((t >> 1) | (nx << 63)) & ~((t >> 2) | (nx << 62));
}

// heads of the tie groups of exactly g = 3 or 4 entries ("small groups")
__device__ __forceinline__ u64 small_heads(u64 t, u64 nx, int g) {
  u64 m = ~t & ~((t >> g) | (nx << (64 - g)));
  for (int k = 1; k < g; k++) m &= (t >> k) | (nx << (64 - k));
  return m;
}

// per workgroup of 256 words: number of pair heads, of small-group heads and of
// the small groups' records; per word: the bitmap without the pairs
__global__ __launch_bounds__(256) void k_pair_words(
    const u64 *__restrict__ tiebits, u64 nwords, u32 *__restrict__ cnt,
    u32 *__restrict__ scnt, u32 *__restrict__ rcnt, u64 *__restrict__ tiebits2) {
  __shared__ u32 s4[4];
  const u64 w = (u64) blockIdx.x * 256 + threadIdx.x;
  u32 np = 0, n3 = 0, n4 = 0;
  if (w < nwords) {
    const u64 t = tiebits[w];
    const u64 nx = w + 1 < nwords ? tiebits[w + 1] : 0ull;
    const u64 ph = pair_heads(t, nx);
    // a pair head in bit 63 of the word before clears this word's bit 0
    u64 prevhead = 0;
    if (w > 0) prevhead = pair_heads(tiebits[w - 1], t) >> 63;
    np = (u32) __popcll(ph);
    n3 = (u32) __popcll(small_heads(t, nx, 3));
    n4 = (u32) __popcll(small_heads(t, nx, 4));
    tiebits2[w] = t & ~((ph << 1) | prevhead);
  }
  const u32 tp = block_total_256(np, s4);
  const u32 ts = block_total_256(n3 + n4, s4);
  const u32 tr = block_total_256(3u * n3 + 6u * n4, s4);
  if (threadIdx.x == 0) { cnt[blockIdx.x] = tp; scnt[blockIdx.x] = ts; rcnt[blockIdx.x] = tr; }
}

// ---------------------------------------------------------------------------
// tie groups of three or four suffixes.  Most of them are a repeat pair that a
// third suffix happens to share 20 symbols with (n / 4^20 = 0.3 % of all
// 20-mers at 3 Gbp): spread over the whole text, they alone would make the
// rounds need every window of the rank table.  Every pair of members of such a
// group joins the list of the pair path (3 or 6 records per group), where the
// comparisons are amortised along the text; k_small_combine then sorts the
// group from the pairwise results.  (Comparing inside the group directly, one
// thread per group, cost 29 ms at 3 Gbp: every deep pair paid its full LCP.)
// ---------------------------------------------------------------------------
// members (x, y) of record q of a group: (0,1) (0,2) (1,2) (0,3) (1,3) (2,3)
__device__ __forceinline__ void small_pair(int q, int *x, int *y) {
  const int xs = (0x210100 >> (4 * q)) & 15, ys = (0x333221 >> (4 * q)) & 15;
  *x = xs; *y = ys;
}

// per group: index of its first entry, size, ordinal of its first record; per
// record: (smaller position, other position | ordinal << 32) appended to the
// pair list behind the np pairs; one thread per bitmap word
template <typename P>
__global__ __launch_bounds__(256) void k_small_emit(
    const u64 *__restrict__ tiebits, u64 nwords, const u32 *__restrict__ soff,
    const u32 *__restrict__ roff, const P *__restrict__ sa, u64 np,
    u32 *__restrict__ sidx, u8 *__restrict__ ssize, u32 *__restrict__ srec,
    P *__restrict__ pkey, u64 *__restrict__ pval) {
  // soff / roff: exclusive scans of k_pair_words' workgroup counts
  __shared__ u32 s_scan[4];
  const u64 w = (u64) blockIdx.x * 256 + threadIdx.x;
  u64 t = 0, h3 = 0, h4 = 0;
  if (w < nwords) {
    t = tiebits[w];
    const u64 nx = w + 1 < nwords ? tiebits[w + 1] : 0ull;
    h3 = small_heads(t, nx, 3);
    h4 = small_heads(t, nx, 4);
  }
  u64 h = h3 | h4;
  u32 tot;
  u32 j = soff[blockIdx.x] + block_scan_excl_sum((u32) __popcll(h), &tot, s_scan);
  u64 r = np + roff[blockIdx.x] +
          block_scan_excl_sum(3u * (u32) __popcll(h3) + 6u * (u32) __popcll(h4), &tot, s_scan);
  if (h == 0) return;
  while (h) {
    const int b = __ffsll((unsigned long long) h) - 1;
    h &= h - 1;
    const u64 i = w * 64 + b;
    const int g = ((h4 >> b) & 1ull) ? 4 : 3;
    sidx[j] = (u32) i;
    ssize[j] = (u8) g;
    srec[j] = (u32) r;
    u64 m[4];
    for (int k = 0; k < g; k++) m[k] = sa[i + k];   // position order (stable sort)
    const int nrec = g == 3 ? 3 : 6;
    for (int q = 0; q < nrec; q++) {
      int x, y;
      small_pair(q, &x, &y);
      pkey[r] = (P) m[x];
      // (64-bit positions: the value is the index of member x | y << 30 ... not
      // needed: the partner is looked up through sidx, see k_pair_resolve)
      pval[r] = (sizeof(P) == 4 ? m[y] : (i + y - 1)) | (r << 32);
      r++;
    }
    j++;
  }
}

// sorts the group from the pairwise results (res[ordinal]: LCP | PAIR_SWAP if
// the member with the larger position is the smaller suffix), writes it back
// to the suffix array, takes it out of the bitmap; per group: the permutation
// (2 bits per place: which member went there) and the LCPs of places 1 .. g-1
template <typename P>
__global__ __launch_bounds__(256) void k_small_combine(
    const u32 *__restrict__ sidx, const u8 *__restrict__ ssize, const u32 *__restrict__ srec,
    const u32 *__restrict__ res, u64 ns, P *__restrict__ sa, u64 *__restrict__ tiebits2,
    u32 *__restrict__ sres, u32 *__restrict__ slcp, Stats *stats) {
  __shared__ unsigned long long s_sum[4], s_large[4];
  __shared__ u32 s_max[4], s_cnt[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  unsigned long long sum = 0, nlarge = 0;
  u32 mx = 0, done = 0;
  if (j < ns) {
    const u64 i = sidx[j];
    const int g = ssize[j];
    const u32 r0 = srec[j];
    // less[x][y] for x < y: member x is the smaller suffix; lc: their LCP
    u32 lc[4][4];
    bool xfirst[4][4];
    const int nrec = g == 3 ? 3 : 6;
    for (int q = 0; q < nrec; q++) {
      int x, y;
      small_pair(q, &x, &y);
      const u32 r = res[r0 + q];
      lc[x][y] = lc[y][x] = r & ~PAIR_SWAP;
      xfirst[x][y] = !(r & PAIR_SWAP);
      xfirst[y][x] = (r & PAIR_SWAP) != 0;
    }
    // place of a member = number of members that are smaller suffixes
    u64 pos[4];
    int who[4] = {0, 0, 0, 0};
    for (int k = 0; k < g; k++) pos[k] = sa[i + k];
    for (int k = 0; k < g; k++) {
      int place = 0;
      for (int o = 0; o < g; o++)
        if (o != k && xfirst[o][k]) place++;
      who[place] = k;
    }
    u32 perm = 0;
    for (int k = 0; k < g; k++) {
      sa[i + k] = (P) pos[who[k]];
      perm |= (u32) who[k] << (2 * k);
      if (k > 0) {
        const u32 lv = lc[who[k - 1]][who[k]];
        slcp[3 * j + (k - 1)] = lv;
        sum += lv;
        nlarge += lv >= GTAMD_LCPOVERFLOW;
        mx = lv > mx ? lv : mx;
      }
    }
    sres[j] = perm | ((u32) g << 8);
    // entries i+1 .. i+g-1 are no longer tied with their predecessor
    for (int k = 1; k < g; k++) {
      const u64 e = i + k;
      atomicAnd(reinterpret_cast<unsigned long long *>(&tiebits2[e >> 6]), ~(1ull << (e & 63)));
    }
    done = (u32) g;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    sum += __shfl_xor(sum, d, 64);
    nlarge += __shfl_xor(nlarge, d, 64);
    done += __shfl_xor(done, d, 64);
    const u32 o = __shfl_xor(mx, d, 64);
    mx = o > mx ? o : mx;
  }
  if (lane == 0) { s_sum[w] = sum; s_large[w] = nlarge; s_max[w] = mx; s_cnt[w] = done; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long S = 0, Lg = 0;
    u32 M = 0, D = 0;
    for (int x = 0; x < 4; x++) {
      S += s_sum[x]; Lg += s_large[x]; D += s_cnt[x]; M = s_max[x] > M ? s_max[x] : M;
    }
    if (S) atomicAdd(&stats->lcpsum, S);
    if (Lg) atomicAdd(&stats->numlarge, Lg);
    if (M) atomicMax(&stats->maxlcp, M);
    if (D) atomicAdd(&stats->smalldone, (unsigned long long) D);
  }
}

// table entries of the small groups (after the emission of the other entries)
template <typename P>
__global__ __launch_bounds__(256) void k_small_apply(
    const u32 *__restrict__ sidx, const u32 *__restrict__ sres, const u32 *__restrict__ slcp,
    u64 ns, const P *__restrict__ sa, u64 *__restrict__ suf, u8 *__restrict__ lcp,
    u8 *__restrict__ bwt, u32 *__restrict__ lcpfull, u64 index_offset, Stats *stats) {
  // (grid-stride: beside the refinement the grid is kept small, see apply_grid)
  for (u64 j = (u64) blockIdx.x * 256 + threadIdx.x; j < ns; j += (u64) gridDim.x * 256) {
    const u32 r = sres[j];
    const u64 i = sidx[j];
    const int g = (int) (r >> 8) & 7;
    u8 old[4] = {0, 0, 0, 0};
    if (bwt != nullptr)
      for (int k = 0; k < g; k++) old[k] = bwt[i + k];
    for (int k = 0; k < g; k++) {
      const u64 p = sa[i + k];
      if (suf != nullptr) suf[i + k] = p;
      if (bwt != nullptr) bwt[i + k] = old[(r >> (2 * k)) & 3u];   // (the symbols of the keys, in key order)
      if (p == 0) stats->longest = index_offset + i + k;
      if (k > 0 && lcp != nullptr) {
        const u32 lv = slcp[3 * j + (k - 1)];
        lcp[i + k] = (u8) (lv < GTAMD_LCPOVERFLOW ? lv : GTAMD_LCPOVERFLOW);
        if (lv >= GTAMD_LCPOVERFLOW) lcpfull[i + k] = lv;
      }
    }
  }
}

// (smaller position, value | ordinal of the pair << 32) with value = the other
// position (32-bit positions) or the index of the pair's first entry (64-bit
// positions: the partner is looked up), and that index again by ordinal; one
// thread per bitmap word.  The workgroup's pairs are staged in LDS and leave in
// whole lines (a thread's three or four pairs written one by one cost 12.7 GB
// of HBM writes for 2.7 GB of pairs).  off: exclusive scan of the workgroup counts.
constexpr int PE_STAGE = 2048;
template <typename P>
__global__ __launch_bounds__(256) void k_pair_emit(
    const u64 *__restrict__ tiebits, u64 nwords, const u32 *__restrict__ off,
    const P *__restrict__ sa, P *__restrict__ pkey, u64 *__restrict__ pval,
    u32 *__restrict__ pidx) {
  __shared__ u32 s_scan[4];
  __shared__ P s_pk[PE_STAGE];
  __shared__ u64 s_pv[PE_STAGE];
  __shared__ u32 s_pi[PE_STAGE];
  const u64 w = (u64) blockIdx.x * 256 + threadIdx.x;
  u64 ph = 0;
  if (w < nwords) {
    const u64 t = tiebits[w];
    const u64 nx = w + 1 < nwords ? tiebits[w + 1] : 0ull;
    ph = pair_heads(t, nx);
  }
  u32 tot;
  u32 local = block_scan_excl_sum((u32) __popcll(ph), &tot, s_scan);
  const u32 base = off[blockIdx.x];
  const bool staged = tot <= (u32) PE_STAGE;
  while (ph) {
    const int b = __ffsll((unsigned long long) ph) - 1;
    ph &= ph - 1;
    const u64 i = w * 64 + b;
    const u64 j = (u64) base + local;
    const P a = sa[i];    // the stable sort left equal keys in position order
    // 32-bit positions: the partner travels with the pair (no look-up later)
    const u64 v = (sizeof(P) == 4 ? (u64) sa[i + 1] : i) | (j << 32);
    if (staged) { s_pk[local] = a; s_pv[local] = v; s_pi[local] = (u32) i; }
    else { pkey[j] = a; pval[j] = v; pidx[j] = (u32) i; }
    local++;
  }
  if (!staged) return;   // (whole workgroup)
  __syncthreads();
  for (u32 k = threadIdx.x; k < tot; k += 256) {
    pkey[base + k] = s_pk[k];
    pval[base + k] = s_pv[k];
    pidx[base + k] = s_pi[k];
  }
}

// order and LCP of every pair, LCP_CHUNK consecutive pairs (by text position)
// per thread; a pair in the wrong order changes places in the suffix array
// here, the tables get their entries from k_pair_apply, which walks the pairs
// in TABLE order (in text order its five accesses per pair were five random
// lines: 18 ms for 170 M pairs)
constexpr int PR_LINE_MAX = 16; // records a thread loads at once at most (a chunk is a multiple of it)
template <int BITS, typename P, int PR_LINE>
__global__ __launch_bounds__(256) void k_pair_resolve(
    Text t, const P *__restrict__ pkey, const u64 *__restrict__ pval, u64 nrec, u64 np,
    const u32 *__restrict__ pidx, P *__restrict__ sa, u32 *__restrict__ res, Stats *stats,
    int chunk, int longext) {
  // records with ordinal < np are pairs (their LCP is a table entry: counted in
  // the statistics); the others are pairs of members of small groups
  __shared__ unsigned long long s_sum[4], s_large[4];
  __shared__ u32 s_max[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long sum = 0, nlarge = 0;
  u32 mx = 0;
  const u64 nchunks = (nrec + chunk - 1) / chunk;
  for (u64 c = (u64) blockIdx.x * 256 + threadIdx.x; c < nchunks;
       c += (u64) gridDim.x * 256) {
    // the last records of this chunk: the pairs of a small group's members share
    // their smaller position, so the record one diagonal step back is up to
    // three records back
    constexpr int RING = 4;
    u64 ra[RING] = {0, 0, 0, 0}, rb[RING] = {0, 0, 0, 0}, rl[RING] = {0, 0, 0, 0};
    bool rf[RING] = {true, true, true, true};
    int filled = 0;
    // A thread's records are consecutive in the list: it takes PR_LINE of them at a
    // time as whole lines (the keys of one step are one 64-byte line, the values
    // two).  Loaded one record per step, every line was touched 16 times over the
    // life of a chunk -- long, when a comparison comes between -- and the 3 million
    // chunks' lines did not stay in the L2 that long: the kernel fetched 62 GB for
    // its 4.8 GB of records (PMC) and was bound by exactly that.
    for (int e0 = 0; e0 < chunk; e0 += PR_LINE) {
      const u64 s0 = c * chunk + e0;
      if (s0 >= nrec) break;
      P ak[PR_LINE];
      u64 ivk[PR_LINE];
      if (s0 + PR_LINE <= nrec) {
        if (sizeof(P) == 4) {
#pragma unroll
          for (int q = 0; q < PR_LINE / 4; q++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(pkey + s0 + 4 * q);
            ak[4 * q] = (P) v.x; ak[4 * q + 1] = (P) v.y; ak[4 * q + 2] = (P) v.z; ak[4 * q + 3] = (P) v.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < PR_LINE / 2; q++) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(pkey + s0 + 2 * q);
            ak[2 * q] = (P) v.x; ak[2 * q + 1] = (P) v.y;
          }
        }
#pragma unroll
        for (int q = 0; q < PR_LINE / 2; q++) {
          const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(pval + s0 + 2 * q);
          ivk[2 * q] = v.x; ivk[2 * q + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int q = 0; q < PR_LINE; q++) {
          ak[q] = s0 + q < nrec ? pkey[s0 + q] : (P) 0;
          ivk[q] = s0 + q < nrec ? pval[s0 + q] : 0ull;
        }
      }
#pragma unroll
    for (int e = 0; e < PR_LINE; e++) {
      const u64 s = s0 + e;
      if (s >= nrec || e0 + e >= chunk) continue;
      const u64 a = ak[e];
      const u64 iv = ivk[e];
      const u64 j = iv >> 32;
      const u64 b = sizeof(P) == 4 ? (iv & 0xFFFFFFFFull) : (u64) sa[(iv & 0xFFFFFFFFull) + 1];
      u64 l = 0;
      bool a_first = true, known = false;
#pragma unroll
      for (int k = 0; k < RING; k++) {
        if (k < filled && !known) {
          const u64 d = a - ra[k];
          if (b >= rb[k] && b - rb[k] == d && rl[k] >= (u64) Key<BITS>::KNOWN + d) {
            // a record on the same diagonal: the same first difference decides,
            // d symbols nearer -- nothing to read
            l = rl[k] - d;
            a_first = rf[k];
            known = true;
          }
        }
      }
      if (!known) {
        if (longext) {
          l = lcp_extend_long<BITS, 4>(t, a, b, (u64) Key<BITS>::KNOWN, &a_first);
        } else {                                     // (A/B: GTAMD_PAIR_LONG=0)
          l = lcp_extend<BITS>(t, a, b, (u64) Key<BITS>::KNOWN);
          // the first difference decides: a special is larger than every letter,
          // two specials compare by position
          const bool spa = is_special(t, a + l), spb = is_special(t, b + l);
          a_first = (spa || spb) ? ((spa && spb) ? a < b : spb)
                                 : Sym<BITS>::at(t, a + l) < Sym<BITS>::at(t, b + l);
        }
      }
#pragma unroll
      for (int k = RING - 1; k > 0; k--) { ra[k] = ra[k - 1]; rb[k] = rb[k - 1]; rl[k] = rl[k - 1]; rf[k] = rf[k - 1]; }
      ra[0] = a; rb[0] = b; rl[0] = l; rf[0] = a_first;
      if (filled < RING) filled++;
      const u32 lv = l < 0x7FFFFFFFull ? (u32) l : 0x7FFFFFFFu;
      res[j] = lv | (a_first ? 0u : PAIR_SWAP);   // by ordinal: the later steps walk the table
      if (j < np) {
        // a pair in the wrong order changes places here (both positions are at
        // hand; this kernel waits for memory anyway -- as a kernel of its own
        // the swap was 5.4 ms of read-modify-write); the small groups are
        // sorted by k_small_combine
        if (!a_first) {
          const u64 i = sizeof(P) == 4 ? (u64) pidx[j] : (iv & 0xFFFFFFFFull);
          sa[i] = (P) b;
          sa[i + 1] = (P) a;
        }
        sum += lv;     // tied suffixes have >= KEY_SYMS >= prefixlength letters
        nlarge += lv >= GTAMD_LCPOVERFLOW;
        mx = lv > mx ? lv : mx;
      }
    }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    sum += __shfl_xor(sum, d, 64);
    nlarge += __shfl_xor(nlarge, d, 64);
    const u32 o = __shfl_xor(mx, d, 64);
    mx = o > mx ? o : mx;
  }
  if (lane == 0) { s_sum[w] = sum; s_large[w] = nlarge; s_max[w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long S = 0, Lg = 0;
    u32 M = 0;
    for (int i = 0; i < 4; i++) {
      S += s_sum[i]; Lg += s_large[i]; M = s_max[i] > M ? s_max[i] : M;
    }
    if (S) atomicAdd(&stats->lcpsum, S);
    if (Lg) atomicAdd(&stats->numlarge, Lg);
    if (M) atomicMax(&stats->maxlcp, M);
  }
}

// table entries of the pairs (after the emission of the other entries), pair by
// pair in table order: the LCP of the second entry; .suf and .bwt of both
// change places with the suffixes (the emission wrote the BWT symbols of the
// keys, in key order)
template <typename P>
__global__ __launch_bounds__(256) void k_pair_apply(
    const u32 *__restrict__ pidx, const u32 *__restrict__ res, u64 np,
    const P *__restrict__ sa, u64 *__restrict__ suf, u8 *__restrict__ lcp,
    u8 *__restrict__ bwt, u32 *__restrict__ lcpfull, u64 index_offset, Stats *stats) {
  // (grid-stride: beside the refinement the grid is kept small, see apply_grid)
  for (u64 j = (u64) blockIdx.x * 256 + threadIdx.x; j < np; j += (u64) gridDim.x * 256) {
    const u64 i = pidx[j];
    const u32 r = res[j], lv = r & ~PAIR_SWAP;
    if (lcp != nullptr) {
      lcp[i + 1] = (u8) (lv < GTAMD_LCPOVERFLOW ? lv : GTAMD_LCPOVERFLOW);
      if (lv >= GTAMD_LCPOVERFLOW) lcpfull[i + 1] = lv;
    }
    // (a pair in its old order has its .suf / .bwt entries from the emission;
    // suffix 0, whose place the statistics want, is always the first of its pair)
    if (r & PAIR_SWAP) {
      const u64 x = sa[i], y = sa[i + 1];
      if (y == 0) stats->longest = index_offset + i + 1;
      if (suf != nullptr) { suf[i] = x; suf[i + 1] = y; }
      if (bwt != nullptr) {
        const u8 b0 = bwt[i], b1 = bwt[i + 1];
        bwt[i] = b1;
        bwt[i + 1] = b0;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// One doubling round for all tie groups that lie inside one tile of the
// unresolved list: sort each group by k2 in LDS (bitonic network on the packed
// key  local group | k2 | slot), derive the new group heads.  Groups that
// reach across a tile border (or are larger than a tile) are left in place
// and flagged for the global radix path.
// ---------------------------------------------------------------------------
constexpr int RT_TILE = 2048;
constexpr int RT_THREADS = 256;
constexpr int RT_PER = RT_TILE / RT_THREADS;   // 8

// sort key of the tile: local group (12 bits) | deferred | k2 | slot (11 bits);
// ranks of a wide build have up to 40 bits
template <typename P> struct RoundKey {
  static constexpr int KK = sizeof(P) == 8 ? 40 : 32;
  static __device__ __forceinline__ u64 pack(u32 lg, bool open, u64 kk, u32 e) {
    return ((u64) lg << (12 + KK)) | ((u64) open << (11 + KK)) | (kk << 11) | (u64) e;
  }
  static __device__ __forceinline__ bool open(u64 key) { return (key >> (11 + KK)) & 1ull; }
};

template <typename P>
__global__ __launch_bounds__(RT_THREADS) void k_round_tile(
    const u32 *__restrict__ uidx, const P *__restrict__ upos,
    const u32 *__restrict__ ugrp, P *__restrict__ k2, u64 m,
    P *__restrict__ cv, u32 *__restrict__ hv, const u32 *__restrict__ tstart,
    u32 *__restrict__ flagbits, const P *__restrict__ rank, u64 h, u64 n) {
  // tstart: the tiles start at group borders (k_tile_starts), so that a group
  // no larger than the tiles' slack never reaches across a border; a tile
  // that is larger than the LDS arrays is worked off in chunks, with the groups
  // across chunk borders left to the global path (flagbits: one bit per slot)
  // rank != nullptr: look the ranks up here (k2[j] = rank[upos[j] + h]) instead
  // of reading a k2 array (which a part build fills through the exchange)
  // 28 KB of LDS (32-bit positions), so that five workgroups share a CU (the
  // rank lookups below are latency-bound random reads: more waves in flight
  // hide more of it).  The group numbers and second keys are only needed until
  // the sort keys are built, so they are staged in the memory of the keys
  // themselves (64-bit ranks have a staging area of their own).
  using RK = RoundKey<P>;
  constexpr bool WIDE = sizeof(P) == 8;
  __shared__ u64 s_key[RT_TILE];
  __shared__ P s_pos[RT_TILE];
  __shared__ u64 s_k2w[WIDE ? RT_TILE : 1];
  __shared__ u16 s_start[RT_TILE + 2];   // first slot of each local group
  __shared__ u32 s_scan[4];
  u32 *s_grp = reinterpret_cast<u32 *>(s_key);
  u32 *s_k2n = s_grp + RT_TILE;
  const int tid = threadIdx.x;
  const u64 tile_first = tstart[blockIdx.x], tile_end = tstart[blockIdx.x + 1];
  for (u64 base = tile_first; base < tile_end; base += RT_TILE) {
  const u32 cnt = (u32) ((tile_end - base) < (u64) RT_TILE ? (tile_end - base) : (u64) RT_TILE);
  // groups that continue in a neighbouring tile
  const u32 g_first = ugrp[base], g_last = ugrp[base + cnt - 1];
  const bool first_open = base > 0 && ugrp[base - 1] == g_first;
  const bool last_open = base + cnt < m && ugrp[base + cnt] == g_last;
#pragma unroll
  for (int c = 0; c < RT_PER; c++) {
    const u32 e = (u32) c * RT_THREADS + tid;
    P kv = 0;
    if (e < cnt) {
      const P p = upos[base + e];
      s_grp[e] = ugrp[base + e];
      s_pos[e] = p;
      if (rank != nullptr) {
        u64 q = (u64) p + h;
        if (q > n) q = n;  // cannot happen for a tied suffix; keeps the load in range
        kv = rank[q];
      } else
        kv = k2[base + e];
    } else {
      s_grp[e] = 0xFFFFFFFFu;   // padding: one trailing pseudo group
      s_pos[e] = 0;
    }
    if (WIDE) s_k2w[e] = (u64) kv; else s_k2n[e] = (u32) kv;
  }
  __syncthreads();
  // from here on a thread owns 8 consecutive slots; it takes their group
  // numbers and second keys into registers before the keys overwrite them
  u32 gr[RT_PER];
  u64 kr[RT_PER];
  const u32 e0 = (u32) tid * RT_PER;
  u64 kprev = tid > 0 ? (WIDE ? s_k2w[e0 - 1] : (u64) s_k2n[e0 - 1]) : 0ull;
  u32 gprev = tid > 0 ? s_grp[e0 - 1] : 0u;
#pragma unroll
  for (int c = 0; c < RT_PER; c++) {
    gr[c] = s_grp[e0 + c];
    kr[c] = WIDE ? s_k2w[e0 + c] : (u64) s_k2n[e0 + c];
  }
  // local group numbers: inclusive count of group starts
  u32 startflags = 0, nstart = 0;
#pragma unroll
  for (int c = 0; c < RT_PER; c++) {
    const u32 e = e0 + c;
    const bool st = e == 0 || gr[c] != gprev;
    gprev = gr[c];
    startflags |= (u32) st << c;
    nstart += st;
  }
  u32 tot;
  // (the barriers inside the scan also end the staging area's first life)
  u32 lg = block_scan_excl_sum(nstart, &tot, s_scan);
  u32 nflag = 0;
  // does any group of this tile split in this round?  (inside a long repeat
  // most rounds leave a group as it is: every member's k2 is the same)
  int splits = 0;
  u32 openbits = 0;   // deferred slots of this thread: one byte per thread in flg
#pragma unroll
  for (int c = 0; c < RT_PER; c++) {
    const u32 e = e0 + c;
    const bool start = (startflags >> c) & 1u;
    lg += start;
    const u32 g = gr[c];
    const bool open = e < cnt && ((first_open && g == g_first) ||
                                  (last_open && g == g_last));
    const u64 kraw = kr[c];
    const u64 kk = open ? 0ull : kraw;
    splits |= (e < cnt && !open && !start && kraw != kprev);
    kprev = kraw;
    // local group | deferred | k2 | slot: a group is deferred as a whole, so
    // the flag bit never reorders anything
    s_key[e] = RK::pack(lg, open, kk, e);
    openbits |= (u32) open << c;
    if (open && rank != nullptr) k2[base + e] = (P) kraw;   // for the global path
    nflag += open;
  }
  if (openbits) {   // rare: a group larger than the tiles' slack, or than a tile
#pragma unroll
    for (int c = 0; c < RT_PER; c++)
      if ((openbits >> c) & 1u) {
        const u64 slot = base + e0 + c;
        atomicOr(&flagbits[slot >> 5], 1u << (slot & 31));
      }
  }
  // (the barrier also orders the s_key writes before the network's reads)
  const int any_split = __syncthreads_or(splits);
  if (any_split) {
    // largest group that has to be sorted here (open and padding groups are
    // already in order)
    {
      u32 lgw = lg;   // local group number of this thread's last slot
#pragma unroll
      for (int c = RT_PER - 1; c >= 0; c--) {
        const u32 e = e0 + c;
        if ((startflags >> c) & 1u) { s_start[lgw] = (u16) e; lgw--; }
      }
      if (tid == 0) s_start[tot + 1] = (u16) RT_TILE;
    }
    __syncthreads();
    u32 gmax = 0;
    {
      u32 lgw = lg;
#pragma unroll
      for (int c = RT_PER - 1; c >= 0; c--) {
        const u32 e = e0 + c;
        if ((startflags >> c) & 1u) {
          const bool open = RK::open(s_key[e]);
          if (e < cnt && !open) {
            const u32 size = (u32) s_start[lgw + 1] - e;
            gmax = size > gmax ? size : gmax;
          }
          lgw--;
        }
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const u32 o = __shfl_xor(gmax, d, 64);
      gmax = o > gmax ? o : gmax;
    }
    if ((tid & 63) == 0) s_scan[tid >> 6] = gmax;
    __syncthreads();
    gmax = s_scan[0];
    for (int i = 1; i < RT_THREADS / 64; i++) gmax = s_scan[i] > gmax ? s_scan[i] : gmax;
    __syncthreads();
    if (gmax <= 32u) {
      // small groups: odd-even transposition, gmax + 1 phases; the group
      // number in the top key bits keeps every exchange inside its group
      for (u32 ph = 0; ph <= gmax; ph++) {
        const u32 odd = ph & 1u;
#pragma unroll
        for (int c = 0; c < RT_PER / 2; c++) {
          const u32 lo = 2u * ((u32) c * RT_THREADS + tid) + odd;
          if (lo + 1 < (u32) RT_TILE) {
            const u64 a = s_key[lo], b = s_key[lo + 1];
            if (a > b) { s_key[lo] = b; s_key[lo + 1] = a; }
          }
        }
        __syncthreads();
      }
    } else {
      // bitonic network, ascending
      for (u32 k = 2; k <= (u32) RT_TILE; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
          for (int c = 0; c < RT_PER / 2; c++) {
            const u32 idx = (u32) c * RT_THREADS + tid;
            const u32 lo = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
            const u32 hi = lo + j;
            const bool up = (lo & k) == 0;
            const u64 a = s_key[lo], b = s_key[hi];
            if ((a > b) == up) { s_key[lo] = b; s_key[hi] = a; }
          }
          __syncthreads();
        }
      }
    }
  }
  // outputs; new group head = first slot of every (group, k2) class
  u32 hval[RT_PER], hmax = 0;
#pragma unroll
  for (int c = 0; c < RT_PER; c++) {
    const u32 e = e0 + c;
    const u64 key = s_key[e];
    const bool head = e == 0 || (key >> 11) != (s_key[e - 1] >> 11);
    const u32 v = (head && e < cnt) ? uidx[base + e] : 0u;
    hval[c] = v;
    hmax = v > hmax ? v : hmax;
  }
  u32 carry = block_scan_excl_max(hmax, &tot, s_scan);
#pragma unroll
  for (int c = 0; c < RT_PER; c++) {
    const u32 e = e0 + c;
    carry = hval[c] > carry ? hval[c] : carry;
    if (e < cnt) {
      const u64 key = s_key[e];
      const u32 src = (u32) (key & 2047u);
      if (!RK::open(key)) {
        cv[base + e] = s_pos[src];
        hv[base + e] = carry;
      }
    }
  }
  (void) nflag;
  __syncthreads();   // the LDS arrays are filled again by the next chunk
  }
}

// where the tiles of a round start: at the multiples of RT_STRIDE slots, moved
// back to the first slot of the group that lies there (a group that starts
// more than RT_TILE slots before stays cut: it is larger than a tile anyway)
constexpr int RT_STRIDE_MIN = 512;
__global__ __launch_bounds__(256) void k_tile_starts(const u32 *__restrict__ ugrp, u64 m,
                                                     u32 ntiles, u32 stride,
                                                     u32 *__restrict__ tstart) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t > ntiles) return;
  if (t == ntiles) { tstart[t] = (u32) m; return; }
  const u64 s = (u64) t * stride;
  const u32 g = ugrp[s];
  // the slots of a group are contiguous and the groups ascend: first slot of
  // g in [s - RT_TILE, s] by bisection (walking back slot by slot was a chain
  // of dependent loads: 0.1 ms per round for groups of a few hundred)
  u64 lo = s > (u64) RT_TILE ? s - RT_TILE : 0, hi = s;     // ugrp[hi] == g
  if (ugrp[lo] == g) {
    tstart[t] = (u32) (lo == 0 ? 0 : s);   // starts further back than a tile: stays cut
    return;
  }
  while (hi - lo > 1) {                    // ugrp[lo] != g, ugrp[hi] == g
    const u64 mid = (lo + hi) >> 1;
    if (ugrp[mid] == g) hi = mid; else lo = mid;
  }
  tstart[t] = (u32) hi;
}

// deferred slots per RT_TILE slots, from the bitmap
__global__ __launch_bounds__(256) void k_flag_count(const u32 *__restrict__ flagbits, u64 m,
                                                    u32 nblocks, u32 *__restrict__ blkcnt) {
  const u32 b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nblocks) return;
  u32 c = 0;
  for (int w = 0; w < RT_TILE / 32; w++) c += (u32) __popc(flagbits[(u64) b * (RT_TILE / 32) + w]);
  blkcnt[b] = c;
  (void) m;
}

// global path for the deferred elements: order by (group, k2) with two stable
// sorts -- on k2, then on the group -- of an index into the deferred list.  One
// workgroup per RT_TILE slots; tileoff = exclusive scan of k_flag_count's counts.
template <typename P>
__global__ __launch_bounds__(RT_THREADS) void k_flag_gather(
    const u32 *__restrict__ flagbits, const u32 *__restrict__ tileoff,
    const u32 *__restrict__ ugrp, const P *__restrict__ k2,
    const P *__restrict__ upos, u64 m, P *__restrict__ fk2, P *__restrict__ fk2_sort,
    u32 *__restrict__ fgrp, P *__restrict__ fpos, u32 *__restrict__ fj,
    u32 *__restrict__ perm) {
  __shared__ u32 s_scan[4];
  const u64 base = (u64) blockIdx.x * RT_TILE + (u64) threadIdx.x * RT_PER;
  // (the tile kernel sets no bit at or behind m)
  // (slot s is bit s & 31 of word s >> 5: byte s >> 3, bit s & 7)
  const u32 f = base < m ? (u32) reinterpret_cast<const u8 *>(flagbits)[(u64) blockIdx.x * RT_THREADS + threadIdx.x] : 0u;
  const u32 cnt = (u32) __popc(f);
  u32 tot;
  u32 o = tileoff[blockIdx.x] + block_scan_excl_sum(cnt, &tot, s_scan);
  if (f == 0) return;
#pragma unroll
  for (int c = 0; c < RT_PER; c++)
    if ((f >> c) & 1u) {
      const u64 j = base + c;
      const P k = k2[j];
      fk2[o] = k;
      fk2_sort[o] = k;
      fgrp[o] = ugrp[j];
      fpos[o] = upos[j];
      fj[o] = (u32) j;
      perm[o] = o;
      o++;
    }
}

__global__ __launch_bounds__(256) void k_gather_u32(const u32 *__restrict__ src,
                                                    const u32 *__restrict__ idx, u64 n,
                                                    u32 *__restrict__ dst) {
  const u64 r = (u64) blockIdx.x * 256 + threadIdx.x;
  if (r < n) dst[r] = src[idx[r]];
}

template <typename P>
__global__ __launch_bounds__(256) void k_flag_heads(
    const u32 *__restrict__ perm, const u32 *__restrict__ fgrp, const P *__restrict__ fk2,
    const P *__restrict__ fpos, const u32 *__restrict__ uidx, const u32 *__restrict__ fj,
    u64 nf, u32 *__restrict__ fhv, P *__restrict__ cvs) {
  const u64 r = (u64) blockIdx.x * 256 + threadIdx.x;
  if (r >= nf) return;
  const u32 a = perm[r];
  bool head = r == 0;
  if (!head) {
    const u32 b = perm[r - 1];
    head = fgrp[a] != fgrp[b] || fk2[a] != fk2[b];
  }
  fhv[r] = head ? uidx[fj[r]] : 0u;
  cvs[r] = fpos[a];
}

template <typename P>
__global__ __launch_bounds__(256) void k_flag_scatter(
    const P *__restrict__ cvs, const u32 *__restrict__ fhv,
    const u32 *__restrict__ fj, u64 nf, P *__restrict__ cv,
    u32 *__restrict__ hv) {
  const u64 r = (u64) blockIdx.x * 256 + threadIdx.x;
  if (r >= nf) return;
  const u32 j = fj[r];
  cv[j] = cvs[r];
  hv[j] = fhv[r];
}

// write the round's result back: positions into the suffix array, new group
// heads into the rank table (a build that holds the whole table; a part build
// sends them to the owners of the positions, RankUpdates); flag what is still
// tied
template <typename P>
__global__ __launch_bounds__(256) void k_round_apply(
    const P *__restrict__ cval, const u32 *__restrict__ gnew,
    const u32 *__restrict__ uidx, const u32 *__restrict__ ugrp, u64 m,
    u64 rank_offset, P *__restrict__ sa, P *__restrict__ rank,
    u64 *__restrict__ keep, u32 *__restrict__ blockcnt) {
  __shared__ u32 s_cnt[4];
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  bool kept = false;
  if (j < m) {
    const P p = cval[j];
    const u32 g = gnew[j], i = uidx[j];
    // slot j stays inside its old group's range, so ugrp[j] is the old group of
    // whichever suffix now sits here; the leading subgroup keeps that id
    if (rank != nullptr && g != ugrp[j]) rank[p] = (P) (rank_offset + g);
    const bool head = g == i;
    const bool nexthead = j + 1 == m || gnew[j + 1] == uidx[j + 1];
    const bool resolved = head && nexthead;
    if (resolved) sa[i] = p;   // final place; unresolved ones move again
    kept = !resolved;
  }
  // survivors: one bit each (a word per wave) for the compaction, which scans
  // the counts of the blocks, not the flags
  const u64 b = __ballot(kept);
  if ((threadIdx.x & 63) == 0) {
    keep[((u64) blockIdx.x * 256 + threadIdx.x) >> 6] = b;
    s_cnt[threadIdx.x >> 6] = (u32) __popcll(b);
  }
  __syncthreads();
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// boff: exclusive scan of the block counts
template <typename P>
__global__ __launch_bounds__(256) void k_round_compact(
    const u64 *__restrict__ keep, const u32 *__restrict__ boff,
    const u32 *__restrict__ blockcnt, const u32 *__restrict__ uidx,
    const P *__restrict__ cval, const u32 *__restrict__ gnew, u64 m,
    u32 *__restrict__ uidx2, P *__restrict__ upos2, u32 *__restrict__ ugrp2,
    Stats *stats) {
  __shared__ u32 s_scan[4];
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  const u32 k = (u32) (keep[j >> 6] >> (j & 63)) & 1u;   // (no bit at or behind m)
  u32 tot;
  const u32 o = boff[blockIdx.x] + block_scan_excl_sum(k, &tot, s_scan);
  if (k) {
    uidx2[o] = uidx[j];
    upos2[o] = cval[j];
    ugrp2[o] = gnew[j];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
    stats->count = boff[blockIdx.x] + blockcnt[blockIdx.x];
}

// total of an exclusively scanned count array (last offset + last count)
__global__ void k_total(const u32 *__restrict__ off, const u32 *__restrict__ cnt,
                        u64 n, Stats *stats) {
  stats->count = n ? off[n - 1] + cnt[n - 1] : 0u;
}

__global__ void k_total2(const u32 *__restrict__ off, const u32 *__restrict__ cnt,
                         u64 n, Stats *stats) {
  stats->count2 = n ? off[n - 1] + cnt[n - 1] : 0u;
}
__global__ void k_total3(const u32 *__restrict__ off, const u32 *__restrict__ cnt,
                         u64 n, Stats *stats) {
  stats->count3 = n ? off[n - 1] + cnt[n - 1] : 0u;
}

// ---------------------------------------------------------------------------
// few, shallow ties (random coincidences on non-repetitive input): resolve
// each tie group by comparing the suffixes on the packed text directly, no
// rank table needed.  Gives up (dfallback) on big groups or deep matches.
// ---------------------------------------------------------------------------
constexpr int DIRECT_MAX_GROUP = 16;
constexpr u64 DIRECT_MAX_LCP = 250;   // below LCPOVERFLOW: never an .llv entry

template <int BITS>
__device__ bool suffix_less(const Text &t, u64 p, u64 q, bool *deep) {
  const u64 l = lcp_extend<BITS>(t, p, q, (u64) Key<BITS>::KNOWN, DIRECT_MAX_LCP);
  if (l >= DIRECT_MAX_LCP) { *deep = true; return p < q; }
  const bool sp = is_special(t, p + l), sq = is_special(t, q + l);
  if (sp || sq) return (sp && sq) ? p < q : sq;   // a special is the larger one
  return Sym<BITS>::at(t, p + l) < Sym<BITS>::at(t, q + l);
}

// (works on a private copy of nothing: a group that gives up has not been
// touched, so the other paths can take it over)
template <int BITS, typename P>
__global__ __launch_bounds__(256) void k_direct_ties(
    Text t, const u32 *__restrict__ uidx0, const u32 *__restrict__ ugrp, u64 m0,
    P *__restrict__ sa, u64 *__restrict__ suf, u8 *__restrict__ lcp,
    u8 *__restrict__ bwt, bool want_lcp, u64 index_offset, Stats *stats) {
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  if (j >= m0) return;
  const u32 i0 = uidx0[j];
  if (ugrp[j] != i0) return;          // one thread per group, at its head
  int g = 1;
  while (j + g < m0 && ugrp[j + g] == i0 && g <= DIRECT_MAX_GROUP) g++;
  if (g > DIRECT_MAX_GROUP) { stats->dfallback = 1; return; }
  u64 pos[DIRECT_MAX_GROUP];
  bool deep = false;
  for (int k = 0; k < g; k++) {       // insertion sort
    const u64 p = sa[i0 + k];
    int a = k;
    while (a > 0 && suffix_less<BITS>(t, p, pos[a - 1], &deep)) {
      pos[a] = pos[a - 1];
      a--;
    }
    pos[a] = p;
  }
  if (deep) { stats->dfallback = 1; return; }
  for (int k = 0; k < g; k++) {
    const u64 i = (u64) i0 + k;
    const u64 p = pos[k];
    sa[i] = (P) p;
    if (suf != nullptr) suf[i] = p;
    if (bwt != nullptr) bwt[i] = Pay<BITS>::to_bwt(Pay<BITS>::before(t, p));
    if (p == 0) stats->longest = index_offset + i;
    if (k > 0 && want_lcp) {
      const u32 l = (u32) lcp_extend<BITS>(t, pos[k - 1], p, (u64) Key<BITS>::KNOWN);
      if (lcp != nullptr) lcp[i] = (u8) l;
      atomicAdd(&stats->dsum, (unsigned long long) l);
      atomicMax(&stats->dmax, l);
    }
  }
}

// ---------------------------------------------------------------------------
// tied suffixes: final LCP / BWT / .suf entries and .llv
// ---------------------------------------------------------------------------
// .suf and .bwt entries of every suffix that took part in the refinement
template <int BITS, typename P>
__global__ __launch_bounds__(256) void k_fix_basic(
    Text t, const u32 *__restrict__ uidx0, u64 m0,
    const P *__restrict__ sa, u64 *__restrict__ suf, u8 *__restrict__ bwt,
    Stats *stats, u64 index_offset) {
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  if (j >= m0) return;
  const u64 i = uidx0[j];
  const u64 p = sa[i];
  if (suf != nullptr) suf[i] = p;
  if (bwt != nullptr) bwt[i] = Pay<BITS>::to_bwt(Pay<BITS>::before(t, p));
  if (p == 0) stats->longest = index_offset + i;
}

// Which entries of the unresolved list are tied with their predecessor (need
// a real LCP)?  Counted per workgroup first; after a scan of the counts the
// same workgroups place (text position, index in the table) of these entries.
__device__ __forceinline__ bool tied_with_pred(const u64 *__restrict__ tiebits, u64 i) {
  return (tiebits[i >> 6] >> (i & 63)) & 1ull;
}

__device__ __forceinline__ void block_count_256(bool flag, u32 *__restrict__ blockcnt) {
  __shared__ u32 s_cnt[4];
  const u64 b = __ballot(flag);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = (u32) __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ __launch_bounds__(256) void k_tied_counts(
    const u32 *__restrict__ uidx0, u64 m0, const u64 *__restrict__ tiebits,
    u32 *__restrict__ blockcnt) {
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  block_count_256(j < m0 && tied_with_pred(tiebits, uidx0[j]), blockcnt);
}

template <typename P>
__global__ __launch_bounds__(256) void k_lcp_pairs(
    const u32 *__restrict__ boff, const u64 *__restrict__ tiebits,
    const u32 *__restrict__ uidx0, const P *__restrict__ sa, u64 m0,
    P *__restrict__ pkey, u32 *__restrict__ pval) {
  __shared__ u32 s_scan[4];
  const u64 j = (u64) blockIdx.x * 256 + threadIdx.x;
  u32 i = 0;
  bool tied = false;
  if (j < m0) {
    i = uidx0[j];
    tied = tied_with_pred(tiebits, i);
  }
  u32 tot;
  const u32 o = boff[blockIdx.x] + block_scan_excl_sum(tied ? 1u : 0u, &tot, s_scan);
  if (tied) {
    pkey[o] = sa[i];
    pval[o] = i;
  }
}

// LCP of the tied entries in TEXT order (the pairs are sorted by position):
// lcp(p+1, pred(p+1)) >= lcp(p, pred(p)) - 1 (Kasai et al.; the reference's
// src/match/sfx-linlcp.c:74-129 walks the whole text this way), hence
// lcp(p+d, pred(p+d)) >= lcp(p, pred(p)) - d for any distance d -- which is
// what a part build needs, where a part sees only every R-th position of a
// repeat.  Inside a repeat only the first position of a chunk pays for the
// full extension.  One thread walks LCP_CHUNK consecutive pairs.  Every
// access by table index is a random line; values that do not fit the byte
// (and only those) also go to a 32-bit side table by index, from which the
// .llv pairs are collected in index order afterwards.
template <int BITS, typename P>
__global__ __launch_bounds__(256) void k_lcp_chunks(
    Text t, const P *__restrict__ pkey, const u32 *__restrict__ pval, u64 m1,
    const P *__restrict__ sa, u8 *__restrict__ lcp, u32 *__restrict__ lcpfull,
    Stats *stats) {
  __shared__ unsigned long long s_sum[4], s_large[4];
  __shared__ u32 s_max[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long sum = 0, nlarge = 0;
  u32 mx = 0;
  const u64 nchunks = (m1 + LCP_CHUNK - 1) / LCP_CHUNK;
  for (u64 c = (u64) blockIdx.x * 256 + threadIdx.x; c < nchunks;
       c += (u64) gridDim.x * 256) {
  u64 prevp = 0, l = 0;
  for (int e = 0; e < LCP_CHUNK; e++) {
    const u64 s = c * LCP_CHUNK + e;
    if (s >= m1) break;
    const u64 p = pkey[s];
    const u64 i = pval[s];
    const u64 q = sa[i - 1];
    u64 from = (u64) Key<BITS>::KNOWN;
    if (e > 0 && l > from + (p - prevp)) from = l - (p - prevp);
    l = lcp_extend_long<BITS>(t, q, p, from);
    prevp = p;
    const u32 lv = l < 0x7FFFFFFFull ? (u32) l : 0x7FFFFFFFu;
    lcp[i] = (u8) (lv < GTAMD_LCPOVERFLOW ? lv : GTAMD_LCPOVERFLOW);
    if (lv >= GTAMD_LCPOVERFLOW) lcpfull[i] = lv;
    sum += lv;     // tied suffixes have >= KEY_SYMS >= prefixlength letters
    nlarge += lv >= GTAMD_LCPOVERFLOW;
    mx = lv > mx ? lv : mx;
  }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    sum += __shfl_xor(sum, d, 64);
    nlarge += __shfl_xor(nlarge, d, 64);
    const u32 o = __shfl_xor(mx, d, 64);
    mx = o > mx ? o : mx;
  }
  if (lane == 0) { s_sum[w] = sum; s_large[w] = nlarge; s_max[w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long S = 0, Lg = 0;
    u32 M = 0;
    for (int i = 0; i < 4; i++) {
      S += s_sum[i]; Lg += s_large[i]; M = s_max[i] > M ? s_max[i] : M;
    }
    if (S) atomicAdd(&stats->lcpsum, S);
    if (Lg) atomicAdd(&stats->numlarge, Lg);
    if (M) atomicMax(&stats->maxlcp, M);
  }
}

// .llv pairs in index order: (index into the lcp table, value),
// src/match/sfx-lcpvalues.c:402-411.  The byte table itself says where they
// are (an entry of 255; the first sort's own LCPs are below the key length):
// 16 entries per thread, counted per workgroup, then placed.
constexpr int LLV_PER = 16;
constexpr int LLV_TILE = 256 * LLV_PER;

__device__ __forceinline__ u32 llv_mask(const u8 *__restrict__ lcp, u64 i0, u64 N) {
  u32 mask = 0;
  if (i0 + LLV_PER <= N && ((reinterpret_cast<uintptr_t>(lcp + i0)) & 15) == 0) {
    const uint4 q = *reinterpret_cast<const uint4 *>(lcp + i0);
    const u32 v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
        if (((v[a] >> (8 * b)) & 255u) == GTAMD_LCPOVERFLOW) mask |= 1u << (4 * a + b);
  } else {
    for (int k = 0; k < LLV_PER; k++)
      if (i0 + k < N && lcp[i0 + k] == GTAMD_LCPOVERFLOW) mask |= 1u << k;
  }
  return mask;
}

__global__ __launch_bounds__(256) void k_large_counts(const u8 *__restrict__ lcp, u64 N,
                                                      u32 *__restrict__ blockcnt) {
  __shared__ u32 s_scan[4];
  const u64 i0 = ((u64) blockIdx.x * 256 + threadIdx.x) * LLV_PER;
  const u32 c = i0 < N ? (u32) __popc(llv_mask(lcp, i0, N)) : 0u;
  u32 tot;
  (void) block_scan_excl_sum(c, &tot, s_scan);
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_llv_emit(
    const u8 *__restrict__ lcp, u64 N, const u32 *__restrict__ lcpfull,
    const u32 *__restrict__ boff, u64 index_offset, u64 *__restrict__ llv) {
  __shared__ u32 s_scan[4];
  const u64 i0 = ((u64) blockIdx.x * 256 + threadIdx.x) * LLV_PER;
  u32 mask = i0 < N ? llv_mask(lcp, i0, N) : 0u;
  u32 tot;
  u64 o = boff[blockIdx.x] + block_scan_excl_sum((u32) __popc(mask), &tot, s_scan);
  while (mask) {
    const int k = __ffs(mask) - 1;
    mask &= mask - 1;
    llv[2 * o] = index_offset + i0 + k;
    llv[2 * o + 1] = lcpfull[i0 + k];
    o++;
  }
}

// grid of a kernel whose workgroups stride over `tiles` tiles: enough
// workgroups to fill the 256 CUs several times over, few enough that their
// closing atomics do not queue up on one address
static inline u32 stride_grid(u64 tiles) {
  // (3 Gbp, 2048 / 4096 / 8192 / 16384 workgroups alternating in one process:
  // 140.4 / 139.4 / 139.15 / 139.2 ms)
  u64 cap = 256 * 32;
  if (const char *e = getenv("GTAMD_STRIDE_WGS")) { const long v = atol(e); if (v >= 256 && v <= (1 << 20)) cap = (u64) v; }
  return (u32) (tiles < cap ? (tiles ? tiles : 1) : cap);
}

#include "esa_msd.h"

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
// device buffer that only grows; growing loses the contents
struct DevBuf {
  void *p;
  u64 bytes;
  template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct gtamd_esa_ctx {
  int device;
  u32 sigma;
  int bits;                // 2 or 5
  u64 max_n, n, N;         // N = n + 1 entries
  int readmode;            // GtReadmode of the sequence handed in as bytes
  hipStream_t st, st2;     // st2: table emission beside the refinement
  hipEvent_t ev_sorted, ev_emitted, ev_applied;
  // resident sequence
  DevBuf tb_own, sp_own;
  Text text;
  bool have_text;
  // workspace, allocated by the first run that needs it (what a run needs
  // depends on the tables wanted and, in a part build, on the slice size)
  DevBuf k0, k1, v0, v1;   // ping-pong (key, position) pairs of the radix sort
  DevBuf isa_tmp;          // 8 B per entry: partitioned pairs of the rank build /
                           // bucketing scratch and 64-bit positions of a part build
  DevBuf rws;              // radix / scan workspace
  DevBuf msd;              // tables of the most-significant-digit-first sort (esa_msd.h)
  DevBuf dig0, dig1;       // digit side arrays of the first sort (experiment)
  DevBuf suf, lcp, bwt;    // outputs at on-disk width
  DevBuf tiebits, tiebits2;
  DevBuf arena;            // unresolved list, round buffers
  DevBuf arena_p;          // lists of the pairs and of the small groups
  DevBuf xrecv;            // part builds: receive side of the exchanges
  DevBuf winbuf;           // bitmaps and list of the rank-table windows
  DevBuf lcpfull_buf;      // LCP values beyond the byte, by table index (when the pairs'
                           // entries are written beside the refinement)
  DevBuf partws;           // part builds: suffixes kept per text tile, and their scan
  DevBuf posw;             // part builds with 64-bit positions: positions of the kept
                           // suffixes in text order
  u64 *llv;
  u64 llv_pairs, llv_cap;
  u32 *bck;                      // .bck sections, back to back
  u64 bck_codes, bck_special, bck_dist;
  Stats *d_stats, *h_stats;   // h_stats: pinned host mirror
  u32 *h_counts;              // pinned: per-part counters read back per round
  u32 *h_hist;                // pinned: key-bin histogram of the own tile
  // (every asynchronous device-to-host copy of the engine lands in pinned memory
  // the context owns, never on a stack frame or in a container that goes away)
  u32 user_prefixlength;   // 0 = automatic
  // part build (lexicographic range `part` of `numparts`)
  u32 part, numparts;
  gtamd_allgather_fn comm_allgather;
  gtamd_alltoallv_fn comm_alltoallv;
  void (*comm_abort)(void *);   // called when a part build fails: the other parts must not wait
  void *comm_abort_user;
  void *comm_user;
  u64 NL, index_offset;    // entries and offset of this part's slice
  u32 *d_parthist;         // PART_BINS counters
  u8 *d_owner;             // bin -> owning part
  u32 *d_counts;           // 4 x DEST_MAXPARTS per-part counters
  // results
  u32 want;
  bool ran;
  gtamd_esa_stats stats;
  gtamd_esa_timing timing;
  u64 alloc_bytes;         // device memory held by the context
  float alloc_ms;          // host time of the allocations since the last run started
  // events
  hipEvent_t ev[8];
  hipEvent_t ev_scatter[2 * 16];
};

static void free_dev(void *p) { if (p != nullptr) (void) hipFree(p); }
static void free_buf(DevBuf &b) { free_dev(b.p); b.p = nullptr; b.bytes = 0; }

// grow-only; whoever calls this knows that the old contents are dead
static int ensure_buf(gtamd_esa_ctx *c, DevBuf &b, u64 bytes, const char *what) {
  if (bytes <= b.bytes) return 0;
  HIP_TRY(hipStreamSynchronize(c->st));
  HIP_TRY(hipStreamSynchronize(c->st2));
  const auto t0 = std::chrono::steady_clock::now();
  c->alloc_bytes -= b.bytes;
  free_buf(b);
  bytes = (bytes + 255) & ~255ull;
  const hipError_t me = hipMalloc(&b.p, bytes);
  c->alloc_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (me != hipSuccess) {
    (void) hipGetLastError();
    b.p = nullptr;
    size_t mfree = 0, mtotal = 0;
    (void) hipMemGetInfo(&mfree, &mtotal);
    gtamd_set_error("cannot allocate %llu bytes of device memory for %s: the "
                    "context holds %llu bytes already, %llu of the device's %llu "
                    "are free",
                    (unsigned long long) bytes, what,
                    (unsigned long long) c->alloc_bytes, (unsigned long long) mfree,
                    (unsigned long long) mtotal);
    return -1;
  }
  b.bytes = bytes;
  c->alloc_bytes += bytes;
  return 0;
}

extern "C" void gtamd_esa_destroy(gtamd_esa_ctx *c) {
  if (c == nullptr) return;
  (void) hipSetDevice(c->device);
  if (c->st != nullptr) (void) hipStreamSynchronize(c->st);
  if (c->st2 != nullptr) (void) hipStreamSynchronize(c->st2);
  DevBuf *bufs[] = {&c->tb_own, &c->sp_own, &c->k0, &c->k1, &c->v0, &c->v1, &c->isa_tmp,
                    &c->rws, &c->dig0, &c->dig1, &c->suf, &c->lcp, &c->bwt, &c->tiebits,
                    &c->tiebits2, &c->arena, &c->arena_p, &c->xrecv, &c->winbuf, &c->msd,
                    &c->lcpfull_buf, &c->partws, &c->posw};
  for (DevBuf *b : bufs) free_buf(*b);
  free_dev(c->llv); free_dev(c->bck); free_dev(c->d_stats);
  free_dev(c->d_parthist); free_dev(c->d_owner); free_dev(c->d_counts);
  if (c->h_stats != nullptr) (void) hipHostFree(c->h_stats);
  if (c->h_counts != nullptr) (void) hipHostFree(c->h_counts);
  if (c->h_hist != nullptr) (void) hipHostFree(c->h_hist);
  for (auto &e : c->ev) if (e != nullptr) (void) hipEventDestroy(e);
  for (auto &e : c->ev_scatter) if (e != nullptr) (void) hipEventDestroy(e);
  if (c->ev_sorted != nullptr) (void) hipEventDestroy(c->ev_sorted);
  if (c->ev_emitted != nullptr) (void) hipEventDestroy(c->ev_emitted);
  if (c->ev_applied != nullptr) (void) hipEventDestroy(c->ev_applied);
  if (c->st2 != nullptr) (void) hipStreamDestroy(c->st2);
  if (c->st != nullptr) (void) hipStreamDestroy(c->st);
  delete c;
}

#define CTX_TRY(expr)                                                         \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      gtamd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                      __FILE__, __LINE__);                                    \
      gtamd_esa_destroy(c);                                                   \
      return nullptr;                                                         \
    }                                                                         \
  } while (0)

// positions a build can address: 32 bits in the sort's value plus the spare
// key bits (part builds; a single build keeps everything 32-bit)
static u64 max_positions(int bits) {
  const int spare = bits == 2 ? Key<2>::SPARE_BITS : Key<5>::SPARE_BITS;
  const int total = 32 + spare > 40 ? 40 : 32 + spare;   // ranks travel in 40 bits
  return 1ull << total;
}
constexpr u64 SINGLE_LIMIT = (1ull << 32) - 4096;   // entries of one slice / single build

extern "C" gtamd_esa_ctx *gtamd_esa_create(int device, uint64_t max_n,
                                           uint32_t numofchars) {
  GTAMD_ABI_BEGIN
  if (gtamd_device_count() <= device || device < 0) {
    gtamd_set_error("no HIP device %d available (this library has no CPU "
                    "fallback)", device);
    return nullptr;
  }
  if (numofchars < 2 || numofchars > 28) {
    gtamd_set_error("alphabet size %u not supported (2..28)", numofchars);
    return nullptr;
  }
  const int bits = numofchars <= 4 ? 2 : 5;
  if (max_n + 1 >= max_positions(bits) - 4096) {
    gtamd_set_error("sequence of %llu symbols exceeds the position range of "
                    "this engine (%llu)", (unsigned long long) max_n,
                    (unsigned long long) max_positions(bits));
    return nullptr;
  }
  gtamd_esa_ctx *c = new gtamd_esa_ctx();
  memset((void *) c, 0, sizeof *c);
  c->device = device;
  c->sigma = numofchars;
  c->bits = bits;
  c->max_n = max_n;
  CTX_TRY(hipSetDevice(device));
  CTX_TRY(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
  CTX_TRY(hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking));
  CTX_TRY(hipEventCreateWithFlags(&c->ev_sorted, hipEventDisableTiming));
  CTX_TRY(hipEventCreateWithFlags(&c->ev_emitted, hipEventDisableTiming));
  CTX_TRY(hipEventCreateWithFlags(&c->ev_applied, hipEventDisableTiming));
  CTX_TRY(hipMalloc(&c->d_stats, sizeof(Stats)));
  CTX_TRY(hipHostMalloc(&c->h_stats, sizeof(Stats), hipHostMallocDefault));
  CTX_TRY(hipHostMalloc(&c->h_counts, 4 * DEST_MAXPARTS * 4, hipHostMallocDefault));
  CTX_TRY(hipHostMalloc(&c->h_hist, PART_BINS * 4, hipHostMallocDefault));
  CTX_TRY(hipMalloc(&c->d_parthist, PART_BINS * 4));
  CTX_TRY(hipMalloc(&c->d_owner, PART_BINS));
  CTX_TRY(hipMalloc(&c->d_counts, 4 * DEST_MAXPARTS * 4));
  c->numparts = 1;
  for (auto &e : c->ev) CTX_TRY(hipEventCreate(&e));
  for (auto &e : c->ev_scatter) CTX_TRY(hipEventCreate(&e));
  return c;
  GTAMD_ABI_END(nullptr)
}

extern "C" int gtamd_esa_set_part(gtamd_esa_ctx *c, uint32_t part,
                                  uint32_t numparts) {
  GTAMD_ABI_BEGIN
  if (c == nullptr) { gtamd_set_error("null context"); return -1; }
  if (numparts == 0 || numparts > (u32) DEST_MAXPARTS || part >= numparts) {
    gtamd_set_error("invalid part %u of %u (1..%d parts)", part, numparts, DEST_MAXPARTS);
    return -1;
  }
  c->part = part;
  c->numparts = numparts;
  c->ran = false;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_esa_set_comm(gtamd_esa_ctx *c, gtamd_allgather_fn ag,
                                  gtamd_alltoallv_fn a2a, void *user) {
  GTAMD_ABI_BEGIN
  if (c == nullptr) { gtamd_set_error("null context"); return -1; }
  c->comm_allgather = ag;
  c->comm_alltoallv = a2a;
  c->comm_user = user;
  c->comm_abort = nullptr;
  c->comm_abort_user = nullptr;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_esa_set_comm_abort(gtamd_esa_ctx *c, void (*abort_fn)(void *), void *user) {
  GTAMD_ABI_BEGIN
  if (c == nullptr) { gtamd_set_error("null context"); return -1; }
  c->comm_abort = abort_fn;
  c->comm_abort_user = user;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_esa_set_prefixlength(gtamd_esa_ctx *c, uint32_t k) {
  GTAMD_ABI_BEGIN
  if (c == nullptr) { gtamd_set_error("null context"); return -1; }
  const u32 maxk = c->bits == 2 ? (u32) KeyLayout<2>::KEY_SYMS : (u32) KeyLayout<5>::KEY_SYMS;
  if (k > maxk) {
    gtamd_set_error("prefix length %u is too large: the maximal prefix length "
                    "for this engine is %u", k, maxk);
    return -1;
  }
  c->user_prefixlength = k;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_esa_set_readmode(gtamd_esa_ctx *c, int readmode) {
  GTAMD_ABI_BEGIN
  if (c == nullptr) { gtamd_set_error("null context"); return -1; }
  if (readmode < 0 || readmode > 3) {
    gtamd_set_error("invalid readmode %d (0 forward, 1 reverse, 2 complement, "
                    "3 reverse complement)", readmode);
    return -1;
  }
  // complementing needs a=0 c=1 g=2 t=3 (gt_alphabet_is_dna, src/core/encseq.c:4960)
  if (readmode >= 2 && c->sigma != 4) {
    gtamd_set_error("readmode %s is only defined for DNA alphabets",
                    readmode == 2 ? "cpl" : "rcl");
    return -1;
  }
  c->readmode = readmode;
  c->have_text = false;   // applies to the next gtamd_esa_set_sequence_bytes
  return 0;
  GTAMD_ABI_END(-1)
}

static int set_n(gtamd_esa_ctx *c, u64 n) {
  if (n > c->max_n) {
    gtamd_set_error("sequence length %llu exceeds the context capacity %llu",
                    (unsigned long long) n, (unsigned long long) c->max_n);
    return -1;
  }
  c->n = n;
  c->N = n + 1;
  c->ran = false;
  return 0;
}

extern "C" int gtamd_esa_set_sequence_bytes(gtamd_esa_ctx *c,
                                            const uint8_t *enc, uint64_t n,
                                            int is_device) {
  GTAMD_ABI_BEGIN
  if (c == nullptr) { gtamd_set_error("null context"); return -1; }
  TRY(set_n(c, n));
  HIP_TRY(hipSetDevice(c->device));
  const u64 spw = c->bits == 2 ? 32 : 12;
  const u64 nw_tb = div_up(c->N, spw), nw_sp = div_up(c->N, 64);
  TRY(ensure_buf(c, c->tb_own, (nw_tb + 2) * 8, "the packed sequence"));
  TRY(ensure_buf(c, c->sp_own, (nw_sp + 2) * 8, "the special bitmap"));
  const u8 *d_enc = enc;
  u8 *staged = nullptr;
  if (!is_device && n > 0) {
    HIP_TRY(hipMalloc(&staged, n));
    if (hipMemcpyAsync(staged, enc, n, hipMemcpyHostToDevice, c->st) != hipSuccess) {
      free_dev(staged);
      gtamd_set_error("copying the sequence to the device failed");
      return -1;
    }
    d_enc = staged;
  }
  u64 *tb = c->tb_own.as<u64>(), *sp = c->sp_own.as<u64>();
  const int rev = c->readmode & 1, cpl = c->readmode >> 1;
  if (c->bits == 2)
    k_pack_symbols<2><<<(u32) div_up(nw_tb, 256), 256, 0, c->st>>>(d_enc, n, tb, nw_tb, rev, cpl);
  else
    k_pack_symbols<5><<<(u32) div_up(nw_tb, 256), 256, 0, c->st>>>(d_enc, n, tb, nw_tb, rev, cpl);
  hipError_t e1 = hipGetLastError();
  k_pack_specials<<<(u32) div_up(nw_sp, 256), 256, 0, c->st>>>(d_enc, n, sp, nw_sp, rev);
  hipError_t e2 = hipGetLastError();
  hipError_t e3 = hipStreamSynchronize(c->st);
  free_dev(staged);
  HIP_TRY(e1); HIP_TRY(e2); HIP_TRY(e3);
  c->text.tb = tb; c->text.sp = sp; c->text.n = n;
  c->text.nw_tb = nw_tb; c->text.nw_sp = nw_sp;
  c->have_text = true;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_esa_set_sequence_packed(gtamd_esa_ctx *c,
                                             const uint64_t *twobit,
                                             const uint64_t *specialbits,
                                             uint64_t n) {
  GTAMD_ABI_BEGIN
  if (c == nullptr) { gtamd_set_error("null context"); return -1; }
  if (c->bits != 2) {
    gtamd_set_error("packed input is defined for the 2-bit DNA layout only");
    return -1;
  }
  if (c->readmode != 0) {
    gtamd_set_error("packed input is read in place: readmode %d cannot be applied "
                    "(hand the sequence in as bytes)", c->readmode);
    return -1;
  }
  TRY(set_n(c, n));
  c->text.tb = twobit; c->text.sp = specialbits; c->text.n = n;
  c->text.nw_tb = div_up(n, 32); c->text.nw_sp = div_up(n + 1, 64);
  c->have_text = true;
  return 0;
  GTAMD_ABI_END(-1)
}

// ---------------------------------------------------------------------------
// the run
// ---------------------------------------------------------------------------
static int fetch_stats(gtamd_esa_ctx *c) {
  HIP_TRY(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(Stats),
                         hipMemcpyDeviceToHost, c->st));
  HIP_TRY(hipStreamSynchronize(c->st));
  return 0;
}

static int bits_for(u64 maxvalue) {
  int b = 1;
  while (b < 64 && (maxvalue >> b) != 0) b++;
  return b;
}

// bump allocation inside the arena; a first walk with base == nullptr only
// adds up the size
struct Bump {
  u8 *base;
  u64 off;
  template <typename T> T *take(u64 count) {
    off = (off + 255) & ~255ull;
    T *p = base != nullptr ? reinterpret_cast<T *>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

// ---- collectives of a part build ----------------------------------------------
// Every allgather carries a status word in front of the payload: a part that
// has failed locally (out of memory, a HIP error) says so in the NEXT allgather
// and all parts leave together -- nobody is left waiting in a collective.
// `failed` != 0: this part has failed; returns -1 on every part if any has.
static int comm_allgather(gtamd_esa_ctx *c, int failed, const void *send, void *recv,
                          u32 bytes) {
  const u32 R = c->numparts;
  if (R == 1 && c->comm_allgather == nullptr) {   // one part, no transport given
    if (bytes) memcpy(recv, send, bytes);
    return failed ? -1 : 0;
  }
  std::vector<u8> mine(8 + (size_t) bytes), all((size_t) R * (8 + bytes));
  const u64 status = failed ? 1 : 0;
  memcpy(mine.data(), &status, 8);
  if (bytes) memcpy(mine.data() + 8, send, bytes);
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = c->comm_allgather(c->comm_user, mine.data(), all.data(), 8 + bytes);
  c->timing.comm_ms += std::chrono::duration<float, std::milli>(
                           std::chrono::steady_clock::now() - t0).count();
  c->timing.comm_calls++;
  if (rc != 0) {
    gtamd_set_error("allgather callback failed");
    return -1;
  }
  int bad = -1;
  for (u32 r = 0; r < R; r++) {
    u64 s;
    memcpy(&s, all.data() + (size_t) r * (8 + bytes), 8);
    if (s != 0 && bad < 0) bad = (int) r;
    if (bytes) memcpy((u8 *) recv + (size_t) r * bytes, all.data() + (size_t) r * (8 + bytes) + 8, bytes);
  }
  if (bad >= 0) {
    if (!failed)
      gtamd_set_error("part %d of the build failed (see its own message)", bad);
    return -1;
  }
  return 0;
}

// device buffers; ordered on the engine's stream (the callback enqueues the
// exchange on it or synchronises it, see include/gtamd_esa.h)
static int comm_alltoallv(gtamd_esa_ctx *c, const void *send, const u64 *sc, void *recv,
                          const u64 *rc, u32 elem, const char *what) {
  const u32 R = c->numparts;
  if (R == 1 && c->comm_alltoallv == nullptr) {
    if (sc[0] != rc[0]) { gtamd_set_error("exchange (%s): count mismatch", what); return -1; }
    if (sc[0])
      HIP_TRY(hipMemcpyAsync(recv, send, sc[0] * elem, hipMemcpyDeviceToDevice, c->st));
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const int crc = c->comm_alltoallv(c->comm_user, send, sc, recv, rc, elem, (void *) c->st);
  c->timing.comm_ms += std::chrono::duration<float, std::milli>(
                           std::chrono::steady_clock::now() - t0).count();
  c->timing.comm_calls++;
  for (u32 r = 0; r < R; r++)
    if (r != c->part) c->timing.comm_bytes += sc[r] * elem;
  if (crc != 0) {
    gtamd_set_error("alltoallv callback failed (%s)", what);
    return -1;
  }
  return 0;
}

// bucket m items by destination part: dest bytes, per-block counts and their
// scan, per-part totals into c->d_counts + slot * DEST_MAXPARTS
template <typename F>
static int dest_count(gtamd_esa_ctx *c, const F &f, u64 m, u8 *dest, u32 *bcount, u32 *boff,
                      u32 *scanws, int slot) {
  const u32 R = c->numparts;
  hipStream_t st = c->st;
  u32 *counts = c->d_counts + slot * DEST_MAXPARTS;
  HIP_TRY(hipMemsetAsync(counts, 0, DEST_MAXPARTS * 4, st));
  if (m == 0) return 0;
  const u64 nb = div_up(m, 256);
  k_dest_count<F><<<(u32) nb, 256, 0, st>>>(f, m, R, nb, dest, bcount);
  HIP_TRY(hipGetLastError());
  TRY(scan_u32(SCAN_SUM, bcount, boff, (u64) R * nb, false, scanws, st));
  k_dest_totals<<<1, DEST_MAXPARTS, 0, st>>>(boff, bcount, R, nb, counts);
  HIP_TRY(hipGetLastError());
  return 0;
}
template <typename F>
static int dest_place(gtamd_esa_ctx *c, const F &f, u64 m, const u8 *dest, const u32 *boff) {
  if (m == 0) return 0;
  const u64 nb = div_up(m, 256);
  k_dest_place<F><<<(u32) nb, 256, 0, c->st>>>(f, m, c->numparts, nb, dest, boff);
  HIP_TRY(hipGetLastError());
  return 0;
}
static u64 dest_words(u32 R, u64 m) { return (u64) R * div_up(m ? m : 1, 256) + 16; }

// the three sections of the bucket table from the sorted keys
template <int BITS>
static int build_bcktab(gtamd_esa_ctx *c, const u64 *skey, u64 NL, u32 k,
                        hipStream_t st) {
  u64 codes = 1, special = 1, dist = 0, pw = 1;
  for (u32 j = 0; j < k; j++) {
    if (codes > (1ull << 31) / c->sigma) {
      gtamd_set_error("bucket table for prefixlength %u over %u letters is too "
                      "large", k, c->sigma);
      return -1;
    }
    codes *= c->sigma;
  }
  for (u32 j = 0; j + 1 < k; j++) special *= c->sigma;
  for (u32 j = 1; j + 1 < k; j++) { pw *= c->sigma; dist += pw; }
  const u64 total = codes + 1 + special + dist;
  free_dev(c->bck);
  c->bck = nullptr;
  HIP_TRY(hipMalloc(&c->bck, total * 4));
  HIP_TRY(hipMemsetAsync(c->bck, 0, total * 4, st));
  if (NL > 0) {
    k_bck_count<BITS><<<(u32) div_up(NL, 256), 256, 0, st>>>(
        skey, NL, k, c->sigma, c->bck, c->bck + codes + 1,
        c->bck + codes + 1 + special);
    HIP_TRY(hipGetLastError());
  }
  // left borders: exclusive prefix sums; the last entry becomes the number of
  // suffixes that are in a bucket
  u32 *ws = c->rws.as<u32>();
  u32 *own = nullptr;
  if (scan_workspace_words(codes + 1) * 4 > c->rws.bytes) {
    HIP_TRY(hipMalloc(&own, scan_workspace_words(codes + 1) * 4));
    ws = own;
  }
  const int rc = scan_u32(SCAN_SUM, c->bck, c->bck, codes + 1, false, ws, st);
  if (own != nullptr) { (void) hipStreamSynchronize(st); free_dev(own); }
  TRY(rc);
  c->bck_codes = codes; c->bck_special = special; c->bck_dist = dist;
  return 0;
}

// workspace of a run over `cap` entries
static u64 rws_words_for(u64 cap) {
  return radix_workspace_words(cap) + 10 * (div_up(cap, 64) + 64) +
         scan_workspace_words(div_up(cap, 64)) + 64;
}
static int ensure_workspace(gtamd_esa_ctx *c, u64 cap, u32 want, bool dist) {
  const u64 pad = cap + 8;
  TRY(ensure_buf(c, c->k0, pad * 8, "sort keys"));
  TRY(ensure_buf(c, c->k1, pad * 8, "sort keys"));
  TRY(ensure_buf(c, c->v0, pad * 4, "sort values"));
  TRY(ensure_buf(c, c->v1, pad * 4, "sort values"));
  TRY(ensure_buf(c, c->rws, rws_words_for(cap) * 4, "the radix workspace"));
  TRY(ensure_buf(c, c->tiebits, (div_up(cap, 64) + 2) * 8, "the tie bitmap"));
  if (want & GTAMD_WANT_SUF) TRY(ensure_buf(c, c->suf, pad * 8, "the suffix table"));
  if (want & GTAMD_WANT_LCP) TRY(ensure_buf(c, c->lcp, pad, "the lcp table"));
  if (want & GTAMD_WANT_BWT) TRY(ensure_buf(c, c->bwt, pad, "the bwt table"));
  if (dist) TRY(ensure_buf(c, c->isa_tmp, pad * 8, "the exchange scratch"));
  {
    // digit-byte side arrays of the sort: an experiment switch (esa_prims.hip)
    const char *db = getenv("GTAMD_DIGBYTES");
    if (db != nullptr && db[0] == '1') {
      TRY(ensure_buf(c, c->dig0, pad, "digit bytes"));
      TRY(ensure_buf(c, c->dig1, pad, "digit bytes"));
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------
// the first sort of a DNA whole-table build, most significant digit first
// (esa_msd.h): keygen, sort, tie bitmap and table emission in one go
// ---------------------------------------------------------------------------
// Bits of level C.  The deepest it goes: ranges of 128-256 entries.  Fewer bits
// are better when the ranges stay small enough for level D: a 256-way level C
// writes runs of 16 entries where a 64-way one writes runs of 64 (3 Gbp
// human-like, k_msd_scatter_lvl<2> 11.8 -> 8.9 ms and 30.2 -> 24.8 GB written;
// the build with 8 / 7 / 6 bits alternating in one process: 140.6 / 140.1 /
// 138.9 ms) -- but level D packs whole ranges into tiles of 4096 entries: a text
// with 70 % A + T, whose 8-mer ranges differ in size by a factor of 15, takes
// 136.1 / 136.3 / 149.4 ms with 8 / 7 / 6 bits.  So the depth is chosen per build,
// from the sizes of the 65536 ranges level B leaves (k_msd_skew).
static int msd_cbits_max(u64 N) {
  const int c = bits_for(N > 1 ? N - 1 : 1) - 24;
  return c < 0 ? 0 : (c > 8 ? 8 : c);
}
// forced depth (tests: every depth at small N), or -1
static int msd_cbits_forced() {
  if (const char *e = getenv("GTAMD_MSD_CBITS")) {
    const int v = atoi(e);
    if (v >= 0 && v <= 8) return v;
  }
  return -1;
}
// penalty[k]: what level D pays with cmax - 2 + k bits, summed over the entries
// (k_msd_skew, ms per 10^9 entries).  Per 10^9 entries one bit less than the
// deepest is worth ~0.17 ms, two ~0.57 ms (3 Gbp, alternating in one process).
// Below 64 children per parent nothing more is gained (1 Gbp, 6 / 5 / 4 bits:
// 48.6 / 49.3 / 49.0 ms).
static int msd_choose_cbits(u64 N, int cmax, const float *penalty) {
  int best = cmax;
  float best_cost = 0.0f;
  for (int k = 1; k >= 0; k--) {
    const int c = cmax - 2 + k;
    if (c < 6) continue;
    const float gain = k == 1 ? 0.17f : 0.57f;
    const float cost = (penalty[k] - penalty[2]) / (float) N - gain;
    if (cost < best_cost) { best = c; best_cost = cost; }
  }
  return best;
}
static u32 msd_big_max() {
  if (const char *e = getenv("GTAMD_MSD_BIG_MAX")) {   // tests: the giant path at small N
    const long v = atol(e);
    if (v >= MS_TILE && v <= (long) MSD_BIG_MAX) return (u32) v;
  }
  return MSD_BIG_MAX;
}
struct MsdWs {
  u32 *hist, *scanws, *tcnt, *tfirst, *dtcnt, *dtfirst, *startA, *startB, *F, *scan2, *counters,
      *biglist, *giantlist, *crowdlist;
  MsTile *desc;
  MdTile *dtiles;
  u64 *firstkey, *lastkey;
  u32 rows, tilesB_ub, tilesC_ub, tilesD_ub;
  u64 nf;
};
static u64 msd_carve(u64 N, int cb, u8 *base, MsdWs *w) {
  const u32 ntA = (u32) div_up(N, MS_TILE);
  w->tilesB_ub = ntA + 256;
  w->tilesC_ub = ntA + MSD_PARENTS;
  // (packed tiles: two consecutive ones hold more than MS_TILE entries together)
  w->tilesD_ub = (u32) (2 * div_up(N, MS_TILE)) + MSD_PARENTS + 8;
  w->rows = w->tilesC_ub + 2;
  w->nf = (u64) MSD_PARENTS << cb;
  Bump b;
  b.base = base;
  b.off = 0;
  w->hist = b.take<u32>((u64) w->rows * 256);
  w->scanws = b.take<u32>(radix_rows_workspace_words(w->rows));
  w->desc = b.take<MsTile>(w->tilesC_ub);
  w->tcnt = b.take<u32>(MSD_PARENTS + 8);
  w->tfirst = b.take<u32>(MSD_PARENTS + 8);
  w->dtcnt = b.take<u32>(MSD_PARENTS + 8);
  w->dtfirst = b.take<u32>(MSD_PARENTS + 8);
  w->startA = b.take<u32>(256 + 8);
  w->startB = b.take<u32>(MSD_PARENTS + 8);
  w->F = b.take<u32>(w->nf + 8);
  w->scan2 = b.take<u32>(scan_workspace_words(w->nf + 1) + 64);
  w->counters = b.take<u32>(16);
  w->dtiles = b.take<MdTile>(w->tilesD_ub);
  w->firstkey = b.take<u64>(w->tilesD_ub);
  w->lastkey = b.take<u64>(w->tilesD_ub);
  w->biglist = b.take<u32>(w->tilesD_ub);
  w->crowdlist = b.take<u32>(w->tilesD_ub);
  w->giantlist = b.take<u32>((u64) (N / MS_TILE) + 8);
  return b.off + 256;
}

// One level below A: parents pstart[0 .. np], digit = the top `cb` bits of the
// key word.  prepare: tiles, histogram, scan, child starts (cstart, (np << cb)
// + 1 entries); scatter: the move.
static int msd_level_prepare(gtamd_esa_ctx *c, const MsdWs &w, const u32 *pstart, u32 np, int cb,
                             u32 tiles_ub, const u32 *kin, u32 *cstart) {
  hipStream_t st = c->st;
  k_msd_tilecount<<<(np + 1 + 255) / 256, 256, 0, st>>>(pstart, 0, np, (u32) MS_TILE, w.tcnt);
  HIP_TRY(hipGetLastError());
  TRY(scan_u32(SCAN_SUM, w.tcnt, w.tfirst, (u64) np + 1, false, w.scan2, st));
  k_msd_tiledesc<<<(tiles_ub + 255) / 256, 256, 0, st>>>(pstart, w.tfirst, np, tiles_ub, w.desc);
  HIP_TRY(hipGetLastError());
  k_msd_hist_lvl<<<tiles_ub + 1, MS_THREADS, 0, st>>>(kin, w.desc, tiles_ub, 32 - cb, w.hist);
  HIP_TRY(hipGetLastError());
  TRY(radix_scan_tile_rows(w.hist, tiles_ub + 1, w.scanws, st));
  const u64 nchild = (u64) np << cb;
  k_msd_tot<<<(u32) div_up(nchild + 1, 256), 256, 0, st>>>(w.hist, w.tfirst, np, cb, cstart);
  HIP_TRY(hipGetLastError());
  TRY(scan_u32(SCAN_SUM, cstart, cstart, nchild + 1, false, w.scan2, st));
  return 0;
}

// on return the tables are emitted (provisional for tied entries, as after
// k_finalize): *sa_out holds the positions in suffix order, (*fkey, *fval) are
// free buffers of 8 / 4 bytes per entry, the tie bitmap and stats->numties are set
//
// A part build (src != nullptr): the entries are the N keys the part has filtered
// from the text (k_part_filter), not all suffixes of the text; density_n is the
// length of the whole table (the depth of level C goes by how crowded the ranges
// are, which is a property of the text, not of the slice).
struct MsdPartSrc {
  const u64 *ck;       // the kept keys, text order
  const u32 *cp32;     // their positions, or nullptr: the value is the entry's number
  u64 index_offset;    // of the slice
  const unsigned long long *prev_key;   // device: largest key of the ranges below
  int has_prev;
};
template <int FMT>
static int msd_sort_emit(gtamd_esa_ctx *c, u32 want, u32 prefixlength, u64 N, u64 density_n,
                         const MsdPartSrc *src, u32 **sa_out,
                         u64 **fkey, u32 **fval, u64 *local_entries) {
  hipStream_t st = c->st;
  const int cmax = msd_cbits_max(density_n);
  int cb = msd_cbits_forced() >= 0 ? msd_cbits_forced() : cmax;
  MsdWs w;
  const u64 bytes = msd_carve(N, cb > cmax ? cb : cmax, nullptr, &w);    // room for the deepest level C
  TRY(ensure_buf(c, c->msd, bytes, "the tables of the first sort"));
  (void) msd_carve(N, cb > cmax ? cb : cmax, c->msd.as<u8>(), &w);
  const u32 ntA = (u32) div_up(N, MS_TILE);
  const u32 last_valid = (u32) (N - (u64) (ntA - 1) * MS_TILE);
  const u64 pad = N + 8;
  // buffer pairs: (k0, v0) and (k1, v1) as 32-bit keys and positions; the bytes
  // of level A in the upper half of k1
  u32 *ka = c->k0.as<u32>(), *pa = c->v0.as<u32>();
  u32 *kb = c->k1.as<u32>(), *pb = c->v1.as<u32>();
  u8 *xa = c->k1.as<u8>() + pad * 4;
  // ---- level A
  if (src != nullptr)
    k_msd_hist_a_keys<<<ntA, MS_THREADS, 0, st>>>(src->ck, N, w.hist);
  else if (FMT == 1)
    k_msd_hist_a5<<<ntA, MS_THREADS, 0, st>>>(c->text, N, w.hist);
  else
    k_msd_hist_a<<<ntA, MS_THREADS, 0, st>>>(c->text, N, w.hist);
  HIP_TRY(hipGetLastError());
  TRY(radix_scan_tile_rows(w.hist, ntA, w.scanws, st));
  k_msd_starts_a<<<1, 256, 0, st>>>(w.hist, (u32) N, w.startA);
  HIP_TRY(hipGetLastError());
  if (src != nullptr)
    k_msd_scatter_a<1><<<((ntA + 7u) >> 3) * 8u, MS_THREADS, 0, st>>>(
        c->text, N, last_valid, w.hist, ntA, src->ck, src->cp32, ka, xa, pa);
  else if (FMT == 1)
    k_msd_scatter_a<2><<<((ntA + 7u) >> 3) * 8u, MS_THREADS, 0, st>>>(
        c->text, N, last_valid, w.hist, ntA, nullptr, nullptr, ka, xa, pa);
  else
    k_msd_scatter_a<0><<<((ntA + 7u) >> 3) * 8u, MS_THREADS, 0, st>>>(
        c->text, N, last_valid, w.hist, ntA, nullptr, nullptr, ka, xa, pa);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev[1], st));
  // ---- level B: (ka, xa, pa) -> (kb, pb)
  TRY(msd_level_prepare(c, w, w.startA, 256, 8, w.tilesB_ub, ka, w.startB));
  k_msd_scatter_lvl<1><<<((w.tilesB_ub + 7u) >> 3) * 8u, MS_THREADS, 0, st>>>(
      ka, xa, pa, w.desc, w.tilesB_ub, w.hist, w.tfirst, w.startB, 8, 24, kb, pb);
  HIP_TRY(hipGetLastError());
  if (msd_cbits_forced() < 0 && cmax >= 1) {
    // how deep level C has to cut, from the ranges level B leaves (the host
    // waits here while the device moves the entries of level B)
    float *d_exp = reinterpret_cast<float *>(w.counters);
    float *h_exp = reinterpret_cast<float *>(c->h_counts);
    HIP_TRY(hipMemsetAsync(d_exp, 0, 16, st));
    k_msd_skew<<<MSD_PARENTS / 256, 256, 0, st>>>(w.startB, cmax, d_exp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_exp, d_exp, 12, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    cb = msd_choose_cbits(N, cmax, h_exp);
  }
  w.nf = (u64) MSD_PARENTS << cb;
  // ---- level C: (kb, pb) -> (ka, pa); its scatter waits for the look at the tiles
  u32 *kf = kb, *pf = pb, *ko = ka, *po = pa;   // (kf, pf): where the finest level ends up
  const u32 *F = w.startB;
  if (cb > 0) {
    TRY(msd_level_prepare(c, w, w.startB, MSD_PARENTS, cb, w.tilesC_ub, kb, w.F));
    F = w.F;
    kf = ka; pf = pa; ko = kb; po = pb;
  }
  // ---- level D tiles, and a look at them
  HIP_TRY(hipMemsetAsync(w.counters, 0, 64, st));
  // tiles of whole ranges: packed (default), or cut by the stride rule
  // (GTAMD_MSD_PACK=0, kept for comparison)
  const char *pk = getenv("GTAMD_MSD_PACK");
  const bool packed = pk == nullptr || atoi(pk) != 0;
  u32 pack_cap = MD_CAP;
  if (const char *e = getenv("GTAMD_MSD_PACK_CAP")) { const long v = atol(e); if (v >= 1024 && v <= (long) MD_CAP) pack_cap = (u32) v; }
  if (packed) {
    k_msd_pack<false><<<(MSD_PARENTS + 1 + 255) / 256, 256, 0, st>>>(F, cb, nullptr, pack_cap, w.dtcnt, nullptr,
                                                                nullptr, nullptr, nullptr, 0);
    HIP_TRY(hipGetLastError());
    TRY(scan_u32(SCAN_SUM, w.dtcnt, w.dtfirst, (u64) MSD_PARENTS + 1, false, w.scan2, st));
    k_msd_pack<true><<<(MSD_PARENTS + 1 + 255) / 256, 256, 0, st>>>(F, cb, w.dtfirst, pack_cap, nullptr, w.dtiles,
                                                               w.biglist, w.giantlist, w.counters,
                                                               msd_big_max());
    HIP_TRY(hipGetLastError());
  } else {
  k_msd_tilecount<<<(MSD_PARENTS + 1 + 255) / 256, 256, 0, st>>>(F, cb, MSD_PARENTS, MSD_STRIDE,
                                                               w.dtcnt);
  HIP_TRY(hipGetLastError());
  TRY(scan_u32(SCAN_SUM, w.dtcnt, w.dtfirst, (u64) MSD_PARENTS + 1, false, w.scan2, st));
  k_msd_dtiles<<<(w.tilesD_ub + 255) / 256, 256, 0, st>>>(F, cb, w.dtfirst, w.tilesD_ub, w.dtiles,
                                                        w.biglist, w.giantlist, w.crowdlist,
                                                        w.counters, msd_big_max());
  HIP_TRY(hipGetLastError());
  }
  u32 *hc = c->h_counts;
  HIP_TRY(hipMemcpyAsync(hc, w.counters, 16, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(hc + 4, w.dtfirst + MSD_PARENTS, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const u32 nbig = hc[0], maxrun = hc[1], nbigentries = hc[2], ngiant = hc[3], ntD = hc[4];
  if (getenv("GTAMD_DEBUG") != nullptr)
    fprintf(stderr, "gtamd: msd sort: %d bits at level C, %u runs, %u big (largest %u, %u entries "
            "in all), %u giant\n", cb, ntD, nbig, maxrun, nbigentries, ngiant);
  std::vector<MdTile> giants(ngiant);
  std::vector<u32> giant_t(ngiant);
  if (ngiant > 0) {
    // (blocking copies: the destinations are containers of this frame)
    HIP_TRY(hipMemcpy(giant_t.data(), w.giantlist, (size_t) ngiant * 4, hipMemcpyDeviceToHost));
    for (u32 i = 0; i < ngiant; i++)
      HIP_TRY(hipMemcpy(&giants[i], w.dtiles + giant_t[i], sizeof(MdTile), hipMemcpyDeviceToHost));
  }
  if (cb > 0) {
    k_msd_scatter_lvl<2><<<((w.tilesC_ub + 7u) >> 3) * 8u, MS_THREADS, 0, st>>>(
        kb, nullptr, pb, w.desc, w.tilesC_ub, w.hist, w.tfirst, w.F, cb, 32 - cb, ka, pa);
    HIP_TRY(hipGetLastError());
  }
  // ---- level D
  MsdOut o;
  o.suf = (want & GTAMD_WANT_SUF) ? c->suf.as<u64>() : nullptr;
  o.lcp = (want & GTAMD_WANT_LCP) ? c->lcp.as<u8>() : nullptr;
  o.bwt = (want & GTAMD_WANT_BWT) ? c->bwt.as<u8>() : nullptr;
  o.sa = pf;
  o.tiebits = c->tiebits.as<u64>();
  o.firstkey = w.firstkey;
  o.lastkey = w.lastkey;
  o.stats = c->d_stats;
  o.prefixlength = prefixlength;
  o.index_offset = src != nullptr ? src->index_offset : 0;
  o.val_is_index = src != nullptr && src->cp32 == nullptr;
  if (o.val_is_index) o.suf = nullptr;     // (k_part_positions writes it)
  HIP_TRY(hipMemsetAsync(o.tiebits, 0, (div_up(N, 64) + 2) * 8, st));
  if (ntD > 0) {
    // (GTAMD_MSD_RADIX=1: every run takes the LSD passes a run with a crowded
    // bin of the counting pass is left to; tests)
    const char *fr = getenv("GTAMD_MSD_RADIX");
    HIP_TRY(hipEventRecord(c->ev_scatter[0], st));
    u32 bin_limit = MD_BIN_LIMIT_DEFAULT;
    if (const char *e = getenv("GTAMD_MSD_BIN_LIMIT")) { const long v = atol(e); if (v >= 2 && v <= 4096) bin_limit = (u32) v; }
    k_msd_local<FMT><<<stride_grid(ntD), MS_THREADS, 0, st>>>(kf, pf, w.dtiles, ntD, cb,
                                                        fr != nullptr && fr[0] == '1', bin_limit,
                                                        w.crowdlist, w.counters, o);
    HIP_TRY(hipEventRecord(c->ev_scatter[1], st));
    HIP_TRY(hipGetLastError());
    k_msd_local_radix<FMT><<<ntD < 2048u ? ntD : 2048u, MS_THREADS, 0, st>>>(kf, pf, w.dtiles,
                                                                        w.crowdlist, w.counters,
                                                                        cb, o);
    HIP_TRY(hipGetLastError());
  }
  if (nbig > 0) {
    k_msd_big<FMT><<<nbig < 2048u ? nbig : 2048u, MS_THREADS, 0, st>>>(kf, pf, ko, po, w.dtiles,
                                                                w.biglist, w.counters, cb, o);
    HIP_TRY(hipGetLastError());
  }
  for (u32 i = 0; i < ngiant; i++) {
    // the bits of K2 above the payload (29, or the code's 24), least significant digit first
    const MdTile g = giants[i];
    const u64 cnt = g.end - g.begin;
    // (an even number of passes: the run ends where it began, which is where
    // k_msd_emit_run reads it)
    const int shifts[4] = {FMT == 1 ? 8 : 3, FMT == 1 ? 14 : 11, FMT == 1 ? 20 : 19, FMT == 1 ? 26 : 27};
    const int widths[4] = {FMT == 1 ? 6 : 8, FMT == 1 ? 6 : 8, FMT == 1 ? 6 : 8, FMT == 1 ? 6 : 5};
    int nev = 0;
    TRY(radix_sort_pairs<u32, u32>(kf + g.begin, pf + g.begin, ko + g.begin, po + g.begin, cnt,
                                   shifts, widths, 4, c->rws.as<u32>(), st, nullptr, &nev));
    k_msd_emit_run<FMT><<<(u32) div_up(cnt, MS_TILE), MS_THREADS, 0, st>>>(kf, pf, giant_t[i], g.begin,
                                                                     (u32) cnt, g.s16, o);
    HIP_TRY(hipGetLastError());
  }
  if (ntD > 0) {
    k_msd_seams<FMT><<<(ntD + 255) / 256, 256, 0, st>>>(w.dtiles, ntD, src != nullptr ? src->prev_key : nullptr,
                                                   src != nullptr ? src->has_prev : 0, o);
    HIP_TRY(hipGetLastError());
  }
  *sa_out = pf;
  *fkey = reinterpret_cast<u64 *>(ko);
  *fval = po;
  *local_entries = N - nbigentries;
  c->stats.msd_big_entries = nbigentries;
  return 0;
}

// BITS: symbol width; WIDE: positions and ranks are 64-bit (part builds of
// sequences with n >= 2^32)
template <int BITS, bool WIDE> static int run_impl(gtamd_esa_ctx *c, u32 want, bool dist) {
  using K = Key<BITS>;
  using P = typename std::conditional<WIDE, u64, u32>::type;
  const u64 N = c->N, n = c->n;
  const u32 R = c->numparts;
  hipStream_t st = c->st;
  const bool want_suf = want & GTAMD_WANT_SUF, want_lcp = want & GTAMD_WANT_LCP,
             want_bwt = want & GTAMD_WANT_BWT;
  const u32 prefixlength = c->user_prefixlength
                               ? c->user_prefixlength
                               : gtamd_recommended_prefixlength(c->sigma, n);
  if (prefixlength > (u32) K::SYMS) {
    gtamd_set_error("prefixlength %u exceeds the key width %d", prefixlength,
                    K::SYMS);
    return -1;
  }
  if (dist && (want & GTAMD_WANT_BCK)) {
    // as the reference: no bucket table from a run in parts
    // (gt_Sfxiterator_bcktab2file, src/match/sfx-suffixer.c:2206-2217)
    gtamd_set_error("the bucket table is not available from a part build");
    return -1;
  }
  if (R > 1 && (c->comm_allgather == nullptr || c->comm_alltoallv == nullptr)) {
    gtamd_set_error("a part build needs the collective callbacks "
                    "(gtamd_esa_set_comm)");
    return -1;
  }
  memset(&c->timing, 0, sizeof c->timing);
  memset(&c->stats, 0, sizeof c->stats);
  c->alloc_ms = 0;
  c->llv_pairs = 0;
  const bool debug = getenv("GTAMD_DEBUG") != nullptr;

  // ---- keygen (whole table, or the pairs of this part's key range)
  u64 NL = N, index_offset = 0;
  Tiles tl;
  tl.T = N;
  tl.self = c->part;
  u64 Tn = N;                       // positions of the own text tile
  int fail = 0;                     // local failure, reported in the next allgather
  // DNA whole-table builds: the keygen also does the sort's first pass (the
  // dcode digit); GTAMD_FUSED_PASS0=0 takes the plain keygen + full sort
  bool pass0_done = false;
  // DNA whole-table builds with 32-bit positions: keygen, sort and emission
  // most significant digit first (esa_msd.h).  GTAMD_MSD=0 takes the LSD sort,
  // GTAMD_MSD=1 the MSD sort at any size (tests; by default from 2^25 entries,
  // where it starts to win: 2.3 against 2.6 ms at 50 M, 1.8 against 1.2 ms at 20 M)
  bool msd = false;
  u32 *msd_sa = nullptr, *msd_fval = nullptr;
  u64 *msd_fkey = nullptr, msd_local = 0;
  // (the 5-bit alphabets: by the 40-bit code of nine symbols, esa_msd.h FMT 1; the
  // statistics mask with prefixlength, which the code must be able to tell: <= 9)
  if (!dist && !WIDE && !(want & GTAMD_WANT_BCK) && N >= 64 && (BITS == 2 || (prefixlength <= 9 && c->sigma <= 20))) {
    const char *e = getenv("GTAMD_MSD");
    msd = e != nullptr ? e[0] == '1' : N >= (1ull << 25);
  }
  // DNA part builds: the part's suffixes are filtered from the replicated text by
  // their key range and sorted most significant digit first -- no pair is
  // exchanged (GTAMD_MSD=0: tile keygen + alltoallv of the pairs + LSD sort, which
  // is what the 5-bit alphabets still take)
  bool msd_part = false;
  u64 prev_key = 0;
  int has_prev = 0;
  if (dist && BITS == 2) {
    const char *e = getenv("GTAMD_MSD");
    msd_part = !(e != nullptr && e[0] == '0');
  }
  if (!dist) {
    TRY(ensure_workspace(c, N, want, false));
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(Stats), st));
    HIP_TRY(hipEventRecord(c->ev[0], st));
    const char *fz = getenv("GTAMD_FUSED_PASS0");
    if (msd) {
      TRY(msd_sort_emit<BITS == 5 ? 1 : 0>(c, want, prefixlength, N, N, nullptr, &msd_sa, &msd_fkey, &msd_fval,
                                           &msd_local));
      HIP_TRY(hipEventRecord(c->ev_emitted, st));   // (what the joins below wait for)
    } else if (BITS == 2 && !(fz != nullptr && fz[0] == '0')) {
      const u32 ntiles = (u32) div_up(N, KP_TILE);
      k_dc_hist_dna<<<ntiles, KP_THREADS, 0, st>>>(c->text, N, c->rws.as<u32>());
      HIP_TRY(hipGetLastError());
      TRY(radix_scan_tile_hist(c->rws.as<u32>(), N, st));
      // (into the second buffer pair: the five passes left then end in the
      // first, where the six passes of the plain path end too)
      k_keygen_pass0_dna<<<ntiles, KP_THREADS, 0, st>>>(c->text, N, c->rws.as<u32>(),
                                                      c->k1.as<u64>(), c->v1.as<u32>());
      pass0_done = true;
    } else if (BITS == 2)
      k_keygen_dna<<<(u32) div_up(N, 1024), 256, 0, st>>>(c->text, 0, N, c->k0.as<u64>(),
                                                          c->v0.as<u32>());
    else
      k_keygen<BITS><<<(u32) div_up(N, 1024), 256, 0, st>>>(c->text, 0, N, c->k0.as<u64>(),
                                                            c->v0.as<u32>());
    HIP_TRY(hipGetLastError());
  } else if (msd_part) {
    msd = true;
    // tile geometry of the rank table (cut by text position, see below)
    tl.T = div_up(div_up(N, R), 4096) * 4096;
    {
      const u64 first = (u64) c->part * tl.T < N ? (u64) c->part * tl.T : N;
      const u64 end = first + tl.T < N ? first + tl.T : N;
      Tn = end - first;
    }
    auto part_sort = [&]() -> int {
      HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(Stats), st));
      HIP_TRY(hipEventRecord(c->ev[0], st));
      // ---- range cuts from a histogram of the key bins over every stride-th
      // suffix of the WHOLE text: every part computes the same counts (integer
      // sums) and so the same cuts -- nothing to agree on
      u32 *hist = c->h_hist;
      // (a sample of 10^7 suffixes puts the cuts within 0.1 % of where all of them
      // would; every workgroup flushes its 16 K counters with atomics: few workgroups)
      const u64 stride = N > (1ull << 30) ? 256 : (N > (1u << 24) ? 16 : 1);
      HIP_TRY(hipMemsetAsync(c->d_parthist, 0, PART_BINS * 4, st));
      k_key_hist<BITS><<<N > (1ull << 30) ? 512 : 1024, 256, 0, st>>>(c->text, 0, N, stride, c->d_parthist);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(hist, c->d_parthist, PART_BINS * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      u64 nsamp = 0;
      for (int b = 0; b < PART_BINS; b++) nsamp += hist[b];
      std::vector<u32> cut(R + 1);
      cut[0] = 0;
      cut[R] = PART_BINS;
      {
        u64 run = 0;
        u32 b = 0;
        for (u32 r = 1; r < R; r++) {
          const u64 target = (u64) (((unsigned __int128) nsamp * r) / R);
          while (b < (u32) PART_BINS && run < target) run += hist[b++];
          cut[r] = b;
        }
      }
      const u32 lo = cut[c->part], hi = cut[c->part + 1];
      // ---- what the part keeps: per text tile, below its range, in all
      const u64 ntT64 = div_up(N, MS_TILE);
      if (ntT64 >= (1ull << 31)) { gtamd_set_error("text of %llu tiles", (unsigned long long) ntT64); return -1; }
      const u32 ntT = (u32) ntT64;
      TRY(ensure_buf(c, c->partws, ((u64) ntT + 64 + scan_workspace_words(ntT)) * 4, "the tile counts of the part"));
      u32 *tkeep = c->partws.as<u32>(), *tscan = tkeep + ntT + 32;
      unsigned long long *acc = reinterpret_cast<unsigned long long *>(c->d_counts);
      unsigned long long *hacc = reinterpret_cast<unsigned long long *>(c->h_counts);
      HIP_TRY(hipMemsetAsync(acc, 0, 32, st));
      HIP_TRY(hipMemsetAsync(tkeep, 0, (u64) ntT * 4, st));
      k_part_count<<<ntT < 4096u ? ntT : 4096u, MS_THREADS, 0, st>>>(c->text, N, ntT, lo, hi, tkeep, acc);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(hacc, acc, 24, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      index_offset = hacc[0];
      const u32 binbelow = hacc[0] > 0 ? (u32) hacc[1] : (u32) PART_BINS;
      has_prev = hacc[0] > 0;
      NL = hacc[2];
      if (NL >= SINGLE_LIMIT) {
        gtamd_set_error("slice of %llu entries exceeds the 32-bit index range of one "
                        "part: use more parts", (unsigned long long) NL);
        return -1;
      }
      // (everything behind the sort is sized for the slice or the text tile,
      // whichever is larger: the rank table of the tile lives in the sort's buffers)
      const u64 cap = NL > Tn ? NL : Tn;
      TRY(ensure_workspace(c, cap, want, true));
      if (WIDE) TRY(ensure_buf(c, c->posw, (NL + 8) * 8, "the positions of the part's suffixes"));
      msd_sa = c->v0.as<u32>();
      msd_fkey = c->k1.as<u64>();
      msd_fval = c->v1.as<u32>();
      if (NL == 0) {
        HIP_TRY(hipMemsetAsync(c->tiebits.p, 0, 16, st));
        HIP_TRY(hipEventRecord(c->ev[1], st));
        return 0;
      }
      // ---- its keys and positions, in text order
      TRY(scan_u32(SCAN_SUM, tkeep, tkeep, ntT, false, tscan, st));
      u64 *ck = c->isa_tmp.as<u64>();
      HIP_TRY(hipMemsetAsync(acc + 3, 0, 8, st));     // (the largest key below, made beside the filter)
      if (WIDE)
        k_part_filter<u64><<<ntT, MS_THREADS, 0, st>>>(c->text, N, lo, hi, binbelow, tkeep, ck,
                                                      c->posw.as<u64>(), acc + 3);
      else
        k_part_filter<u32><<<ntT, MS_THREADS, 0, st>>>(c->text, N, lo, hi, binbelow, tkeep, ck,
                                                      c->v1.as<u32>(), acc + 3);
      HIP_TRY(hipGetLastError());
      MsdPartSrc src;
      src.ck = ck;
      src.cp32 = WIDE ? nullptr : c->v1.as<u32>();
      src.index_offset = index_offset;
      src.prev_key = acc + 3;
      src.has_prev = has_prev;
      TRY(msd_sort_emit<0>(c, want, prefixlength, NL, N, &src, &msd_sa, &msd_fkey, &msd_fval, &msd_local));
      return 0;
    };
    fail = part_sort() != 0;
    HIP_TRY(hipEventRecord(c->ev_emitted, st));
  } else {
    // the own text tile: T positions per part, a multiple of the keygen tile
    tl.T = div_up(div_up(N, R), 4096) * 4096;
    const u64 first = (u64) c->part * tl.T < N ? (u64) c->part * tl.T : N;
    const u64 end = first + tl.T < N ? first + tl.T : N;
    Tn = end - first;
    // room for the slice: the tile plus a margin for uneven ranges (grown
    // below, once the real slice size is known, if that is not enough)
    u64 cap = Tn + Tn / 8 + 65536;
    if (cap >= SINGLE_LIMIT) cap = SINGLE_LIMIT - 1;
    fail |= ensure_workspace(c, cap, want, true) != 0;
    if (!fail) {
      HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(Stats), st));
      HIP_TRY(hipEventRecord(c->ev[0], st));
    }
    // range cuts from a histogram of the key bins over every 16th suffix: every
    // part counts its own tile, the sum is the same on every part
    std::vector<u32> allhist((size_t) PART_BINS * R);
    u32 *hist = c->h_hist;
    memset(hist, 0, PART_BINS * 4);
    if (!fail) {
      const u64 stride = N > (1u << 24) ? 16 : 1;
      HIP_TRY(hipMemsetAsync(c->d_parthist, 0, PART_BINS * 4, st));
      if (Tn > 0) {
        k_key_hist<BITS><<<1024, 256, 0, st>>>(c->text, first, end, stride, c->d_parthist);
        HIP_TRY(hipGetLastError());
      }
      HIP_TRY(hipMemcpyAsync(hist, c->d_parthist, PART_BINS * 4, hipMemcpyDeviceToHost, st));
      // the keys of the tile, meanwhile
      if (Tn > 0) {
        if (BITS == 2)
          k_keygen_dna<<<(u32) div_up(Tn, 1024), 256, 0, st>>>(c->text, first, end,
                                                               c->k0.as<u64>(), c->v0.as<u32>());
        else
          k_keygen<BITS><<<(u32) div_up(Tn, 1024), 256, 0, st>>>(c->text, first, end,
                                                                 c->k0.as<u64>(), c->v0.as<u32>());
        HIP_TRY(hipGetLastError());
      }
      HIP_TRY(hipStreamSynchronize(st));
    }
    TRY(comm_allgather(c, fail, hist, allhist.data(), PART_BINS * 4));
    std::vector<u64> start(PART_BINS + 1);
    start[0] = 0;
    for (int b = 0; b < PART_BINS; b++) {
      u64 s = 0;
      for (u32 r = 0; r < R; r++) s += allhist[(size_t) r * PART_BINS + b];
      start[b + 1] = start[b] + s;
    }
    const u64 nsamp = start[PART_BINS];
    std::vector<u32> cut(R + 1);
    std::vector<u8> owner(PART_BINS);
    cut[0] = 0;
    cut[R] = PART_BINS;
    for (u32 r = 1; r < R; r++) {
      const u64 target = (u64) (((unsigned __int128) nsamp * r) / R);
      u32 b = cut[r - 1];
      while (b < (u32) PART_BINS && start[b] < target) b++;
      cut[r] = b;
    }
    for (u32 r = 0; r < R; r++)
      for (u32 b = cut[r]; b < cut[r + 1]; b++) owner[b] = (u8) r;
    HIP_TRY(hipMemcpyAsync(c->d_owner, owner.data(), PART_BINS,
                           hipMemcpyHostToDevice, st));
    // bucket the tile's pairs by the owner of their key range (stable)
    u8 *dest = c->isa_tmp.as<u8>();
    u32 *bcount = reinterpret_cast<u32 *>(dest + ((Tn + 255) & ~255ull));
    u32 *boff = bcount + dest_words(R, Tn);
    OwnerOfKey own;
    own.keys = c->k0.as<u64>(); own.vals = c->v0.as<u32>(); own.owner = c->d_owner;
    own.keys_out = c->k1.as<u64>(); own.vals_out = c->v1.as<u32>();
    if (((Tn + 255) & ~255ull) + 2 * dest_words(R, Tn) * 4 > c->isa_tmp.bytes ||
        scan_workspace_words((u64) R * div_up(Tn ? Tn : 1, 256)) * 4 > c->rws.bytes) {
      gtamd_set_error("%u parts are too many for a tile of %llu positions", R,
                      (unsigned long long) Tn);
      fail = 1;
    }
    if (!fail) {
      TRY(dest_count(c, own, Tn, dest, bcount, boff, c->rws.as<u32>(), 0));
      HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, DEST_MAXPARTS * 4,
                             hipMemcpyDeviceToHost, st));
      TRY(dest_place(c, own, Tn, dest, boff));
      HIP_TRY(hipStreamSynchronize(st));
    }
    std::vector<u64> sendcounts(R), recvcounts(R), matrix((size_t) R * R);
    for (u32 r = 0; r < R; r++) sendcounts[r] = fail ? 0 : c->h_counts[r];
    TRY(comm_allgather(c, fail, sendcounts.data(), matrix.data(), R * 8));
    NL = 0;
    u64 allparts = 0;
    for (u32 s = 0; s < R; s++) {
      recvcounts[s] = matrix[(size_t) s * R + c->part];
      NL += recvcounts[s];
      for (u32 q = 0; q < R; q++) {
        allparts += matrix[(size_t) s * R + q];
        if (q < c->part) index_offset += matrix[(size_t) s * R + q];
      }
    }
    if (allparts != N) {
      gtamd_set_error("range partition counts %llu suffixes, expected %llu",
                      (unsigned long long) allparts, (unsigned long long) N);
      fail = 1;
    }
    if (NL >= SINGLE_LIMIT) {
      gtamd_set_error("slice of %llu entries exceeds the 32-bit index range of one "
                      "part: use more parts", (unsigned long long) NL);
      fail = 1;
    }
    // the receive side first (its old contents -- the unbucketed pairs -- are
    // dead); the rest of the workspace after the exchange
    if (!fail && NL > cap) {
      fail |= ensure_buf(c, c->k0, (NL + 8) * 8, "sort keys") != 0;
      fail |= ensure_buf(c, c->v0, (NL + 8) * 4, "sort values") != 0;
    }
    TRY(comm_allgather(c, fail, nullptr, nullptr, 0));
    TRY(comm_alltoallv(c, c->k1.p, sendcounts.data(), c->k0.p, recvcounts.data(), 8,
                       "keys"));
    TRY(comm_alltoallv(c, c->v1.p, sendcounts.data(), c->v0.p, recvcounts.data(), 4,
                       "positions"));
    if (NL > cap) {
      HIP_TRY(hipStreamSynchronize(st));   // k1/v1 have been sent
      fail |= ensure_workspace(c, NL, want, true) != 0;
      // (a failure here is reported by the next allgather; nothing is launched
      // on the missing buffers, see `fail` below)
    }
  }
  c->NL = NL;
  c->index_offset = index_offset;
  if (debug)
    fprintf(stderr, "gtamd: part %u/%u: tile %llu positions, slice %llu entries at %llu%s\n",
            c->part, R, (unsigned long long) Tn, (unsigned long long) NL,
            (unsigned long long) index_offset, WIDE ? " (64-bit positions)" : "");
  // (a part that filters its suffixes from the text reports a failure with the
  // count of its ties, further down: no exchange of its own for it)
  if (dist && !msd_part) TRY(comm_allgather(c, fail, nullptr, nullptr, 0));
  if (!msd) HIP_TRY(hipEventRecord(c->ev[1], st));   // (the MSD sort: after its level A)

  // ---- first sort: all key bits above the payload
  int shifts[16], widths[16], np = 0;
  if (pass0_done) {   // the dcode digit is sorted: the prefix bits are left
    static_assert((64 - Key<2>::LOW_BITS) % 8 == 0, "whole passes over the DNA prefix");
    for (int b = K::LOW_BITS; b < 64; b += 8) {
      shifts[np] = b;
      widths[np] = 64 - b < 8 ? 64 - b : 8;
      np++;
    }
  } else
  for (int b = K::DSHIFT; b < 64; b += 8) {   // dcode and prefix, contiguous
    shifts[np] = b;
    widths[np] = 64 - b < 8 ? 64 - b : 8;
    np++;
  }
  int nev = 0;
  u64 *ka = pass0_done ? c->k1.as<u64>() : c->k0.as<u64>(),
      *kb = pass0_done ? c->k0.as<u64>() : c->k1.as<u64>();
  u32 *va = pass0_done ? c->v1.as<u32>() : c->v0.as<u32>(),
      *vb = pass0_done ? c->v0.as<u32>() : c->v1.as<u32>();
  if (!msd)
    TRY(radix_sort_pairs<u64, u32>(ka, va, kb, vb, NL, shifts, widths, np,
                              c->rws.as<u32>(), st, c->ev_scatter, &nev, c->dig0.as<u8>(),
                              c->dig1.as<u8>()));
  u64 *skey = (np & 1) ? kb : ka;   // sorted keys
  u32 *sa32 = (np & 1) ? vb : va;   // positions in suffix order (low half)
  u64 *fkey = (np & 1) ? ka : kb;   // free key-sized buffer
  u32 *fval = (np & 1) ? va : vb;   // free value-sized buffer
  if (msd) {
    skey = nullptr;                 // (no sorted keys: the tables are out already)
    sa32 = msd_sa;
    fkey = msd_fkey;
    fval = msd_fval;
  }
  HIP_TRY(hipEventRecord(c->ev[2], st));

  if (want & GTAMD_WANT_BCK) TRY(build_bcktab<BITS>(c, skey, NL, prefixlength, st));

  // positions at the width the refinement works with
  P *sa;
  if (WIDE) {
    u64 *sa64 = c->isa_tmp.as<u64>();   // (the bucketing scratch / the kept keys are dead)
    if (NL > 0 && msd_part && !fail) {
      k_part_positions<<<(u32) div_up(NL, 256), 256, 0, st>>>(
          sa32, c->posw.as<u64>(), NL, index_offset, sa64, want_suf ? c->suf.as<u64>() : nullptr,
          c->d_stats);
      HIP_TRY(hipGetLastError());
    } else if (NL > 0 && !msd_part) {
      k_wide_positions<BITS><<<(u32) div_up(NL, 256), 256, 0, st>>>(skey, sa32, NL, sa64);
      HIP_TRY(hipGetLastError());
    }
    sa = reinterpret_cast<P *>(sa64);
  } else
    sa = reinterpret_cast<P *>(sa32);
  u64 *d_suf = want_suf ? c->suf.as<u64>() : nullptr;
  u8 *d_lcp = want_lcp ? c->lcp.as<u8>() : nullptr;
  u8 *d_bwt = want_bwt ? c->bwt.as<u8>() : nullptr;

  // ---- finalize; a part needs the last key of the preceding range (the MSD
  // sort of a part has it from its count of the text)
  if (R > 1 && !msd_part) {
    u64 mine[2] = {NL, 0};
    HIP_TRY(hipStreamSynchronize(st));
    if (NL > 0)   // (blocking copy: the destination is on this stack frame)
      HIP_TRY(hipMemcpy(&mine[1], skey + (NL - 1), 8, hipMemcpyDeviceToHost));
    std::vector<u64> all(2 * (size_t) R);
    TRY(comm_allgather(c, 0, mine, all.data(), 16));
    for (u32 r = 0; r < c->part; r++)
      if (all[2 * r] > 0) { prev_key = all[2 * r + 1]; has_prev = 1; }
  }
  // Table emission (k_finalize, bandwidth-bound) runs on the second stream.
  // It is started where the first stream turns latency-bound (the comparisons
  // of the pair path, the rounds), so that the two actually overlap; whatever
  // it writes for tied entries is provisional and overwritten after the join.
  bool emitted = msd;
  auto launch_emission = [&]() -> int {
    if (emitted) return 0;
    emitted = true;
    HIP_TRY(hipEventRecord(c->ev_sorted, st));
    HIP_TRY(hipStreamWaitEvent(c->st2, c->ev_sorted, 0));
    if (NL > 0) {
      k_finalize<BITS, P><<<stride_grid(div_up(NL, FIN_TILE)), FIN_THREADS, 0, c->st2>>>(
          skey, sa, NL, prefixlength, d_suf, d_lcp, d_bwt, nullptr,
          c->d_stats, prev_key, has_prev, index_offset);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(c->ev_emitted, c->st2));
    return 0;
  };
  u64 *tiebits = c->tiebits.as<u64>();
  if (NL > 0 && !msd) {
    k_tiebits<BITS><<<stride_grid(div_up(NL, 4096)), 256, 0, st>>>(skey, NL, tiebits,
                                                                   c->d_stats);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(c->ev[3], st));
  if (!(msd_part && fail)) TRY(fetch_stats(c));
  const u64 numties = (msd_part && fail) ? 0 : c->h_stats->numties;
  u64 anyties = numties;
  if (msd_part && fail && R == 1) return -1;
  if (R > 1) {
    std::vector<u64> all(R);
    TRY(comm_allgather(c, msd_part ? fail : 0, &numties, all.data(), 8));
    anyties = 0;
    for (u32 r = 0; r < R; r++) anyties += all[r];
  }
  u32 rounds = 0;
  u64 m0 = 0, npairs = 0, m0_tied_all = 0, rank_built = 0;
  HIP_TRY(hipEventRecord(c->ev[4], st));
  HIP_TRY(hipEventRecord(c->ev[5], st));
  if (anyties > 0) {
    // ---- how many suffixes are tied with a neighbour
    const u64 nwords = div_up(NL, 64);
    u32 *cntw = c->rws.as<u32>();          // radix workspace is idle now
    u32 *headw = cntw + nwords + 16;
    u32 *offw = headw + nwords + 16;
    u32 *carry = offw + nwords + 16;
    u32 *pcnt = carry + nwords + 16;       // pair heads per word, and their scan
    u32 *poff = pcnt + nwords + 16;
    u32 *scnt = poff + nwords + 16;        // small-group heads per word, and their scan
    u32 *soff = scnt + nwords + 16;
    u32 *rcnt = soff + nwords + 16;        // pairs of members of the small groups per word
    u32 *roff = rcnt + nwords + 16;
    u32 *scanws = roff + nwords + 16;
    u32 *pws = scanws + scan_workspace_words(nwords) + 64;   // radix workspace (NL pairs)
    const u64 nwb = div_up(nwords, 256);   // workgroups of 256 bitmap words
    auto tie_words = [&](const u64 *bits) -> int {
      if (NL > 0) {
        k_tie_words<<<(u32) nwb, 256, 0, st>>>(bits, nwords, cntw, headw);
        HIP_TRY(hipGetLastError());
        TRY(scan_u32(SCAN_SUM, cntw, offw, nwb, false, scanws, st));
        TRY(scan_u32(SCAN_MAX, headw, carry, nwords, false, scanws, st));
      }
      k_total<<<1, 1, 0, st>>>(offw, cntw, NL > 0 ? nwb : 0, c->d_stats);
      HIP_TRY(hipGetLastError());
      TRY(fetch_stats(c));
      return 0;
    };
    TRY(tie_words(tiebits));
    m0 = c->h_stats->count;
    // few shallow ties: settle them by direct comparison, no rank table
    bool settled = false;
    {
      const u64 thresh = NL / 512 > 4096 ? NL / 512 : 4096;
      u64 small = m0 <= thresh;
      if (R > 1) {
        std::vector<u64> all(R);
        TRY(comm_allgather(c, 0, &small, all.data(), 8));
        for (u32 r = 0; r < R; r++) small &= all[r];
      }
      if (small) {
        const u64 mp = (m0 + 64 + 3) & ~3ull;
        int bad = ensure_buf(c, c->arena, mp * (12 + sizeof(P)) + 4096, "the tied suffixes") != 0;
        if (!bad) {
          Bump a = {c->arena.as<u8>(), 0};
          u32 *uidx0 = a.take<u32>(mp), *uidx = a.take<u32>(mp), *ugrp = a.take<u32>(mp);
          P *upos = a.take<P>(mp);
          if (m0 > 0) {
            k_unres_emit<P><<<(u32) div_up(nwords, 256), 256, 0, st>>>(
                tiebits, nwords, offw, carry, sa, uidx0, uidx, upos, ugrp);
            HIP_TRY(hipGetLastError());
          }
          TRY(launch_emission());
          HIP_TRY(hipStreamWaitEvent(st, c->ev_emitted, 0));   // join the emission
          if (m0 > 0) {
            k_direct_ties<BITS, P><<<(u32) div_up(m0, 256), 256, 0, st>>>(
                c->text, uidx0, ugrp, m0, sa, d_suf, d_lcp, d_bwt, want_lcp,
                index_offset, c->d_stats);
            HIP_TRY(hipGetLastError());
          }
          TRY(fetch_stats(c));
        }
        u64 gaveup = bad ? 1 : c->h_stats->dfallback;
        if (R > 1) {
          std::vector<u64> all(R);
          TRY(comm_allgather(c, bad, &gaveup, all.data(), 8));
          gaveup = 0;
          for (u32 r = 0; r < R; r++) gaveup |= all[r];
        } else if (bad)
          return -1;
        settled = gaveup == 0;
        if (!settled) {
          // discard the partial statistics of the direct attempt
          HIP_TRY(hipMemsetAsync(&c->d_stats->dsum, 0, 8, st));
          HIP_TRY(hipMemsetAsync(&c->d_stats->dmax, 0, 4, st));
        }
      }
    }
    if (!settled) {
    // ---- pairs and small groups leave the bitmap; what is left goes through
    // prefix doubling
    TRY(ensure_buf(c, c->tiebits2, (nwords + 2) * 8, "the tie bitmap"));
    u64 *tiebits2 = c->tiebits2.as<u64>();
    const char *np_env = getenv("GTAMD_NO_PAIRS");                // A/B switches
    const bool no_pairs = np_env != nullptr && np_env[0] == '1';
    const char *ns_env = getenv("GTAMD_NO_SMALL_GROUPS");
    const bool no_small = no_pairs || (ns_env != nullptr && ns_env[0] == '1');
    u64 nsmall = 0, nsrec = 0;
    if (NL > 0) {
      if (no_pairs) {
        HIP_TRY(hipMemcpyAsync(tiebits2, tiebits, nwords * 8, hipMemcpyDeviceToDevice, st));
      } else {
        k_pair_words<<<(u32) nwb, 256, 0, st>>>(tiebits, nwords, pcnt, scnt, rcnt, tiebits2);
        HIP_TRY(hipGetLastError());
        TRY(scan_u32(SCAN_SUM, pcnt, poff, nwb, false, scanws, st));
        TRY(scan_u32(SCAN_SUM, scnt, soff, nwb, false, scanws, st));
        TRY(scan_u32(SCAN_SUM, rcnt, roff, nwb, false, scanws, st));
        k_total2<<<1, 1, 0, st>>>(poff, pcnt, nwb, c->d_stats);
        k_total3<<<1, 1, 0, st>>>(soff, scnt, nwb, c->d_stats);
        k_total<<<1, 1, 0, st>>>(roff, rcnt, nwb, c->d_stats);
        HIP_TRY(hipGetLastError());
        TRY(fetch_stats(c));
        npairs = c->h_stats->count2;
        nsmall = c->h_stats->count3;
        nsrec = c->h_stats->count;
      }
    }
    // small groups cost three or six records each: only while that stays a
    // small part of the table (a sequence set with three or four near-identical
    // members has them everywhere: prefix doubling is the better tool then)
    if (no_small || nsrec > NL / 8 || npairs + nsrec >= SINGLE_LIMIT) { nsmall = 0; nsrec = 0; }
    const u64 nrec = npairs + nsrec;          // records of the pair list
    const u64 pp = (nrec + 64 + 3) & ~3ull;
    const u64 sp = (nsmall + 64 + 3) & ~3ull;
    P *pk_a = nullptr, *pk_b = nullptr;
    u64 *pv_a = nullptr, *pv_b = nullptr;
    u32 *pidx = nullptr, *pres = nullptr, *prws = nullptr, *sidx = nullptr, *sres = nullptr,
        *slcp = nullptr, *srec = nullptr;
    u8 *ssize = nullptr;
    auto layout_p = [&](Bump &a) {
      pk_a = a.take<P>(pp); pk_b = a.take<P>(pp);
      pv_a = a.take<u64>(pp); pv_b = a.take<u64>(pp);
      pidx = a.take<u32>(pp); pres = a.take<u32>(pp);
      prws = a.take<u32>(radix_workspace_words(nrec));
      sidx = a.take<u32>(sp); sres = a.take<u32>(sp); slcp = a.take<u32>(3 * sp);
      srec = a.take<u32>(sp);
      ssize = a.take<u8>(sp);
    };
    // (GTAMD_APPLY_EARLY, see below: the table entries of the pairs beside the
    // rounds need a buffer of their own for the LCP values beyond the byte; both
    // buffers of this step are agreed on in one exchange)
    int apply_early = 2;
    if (const char *e = getenv("GTAMD_APPLY_EARLY")) apply_early = atoi(e);
    if (apply_early < 0 || apply_early > 2) apply_early = 0;
    {
      Bump sz = {nullptr, 0};
      layout_p(sz);
      fail = ensure_buf(c, c->arena_p, sz.off + 4096, "the pairs of tied suffixes") != 0;
      if (!fail && want_lcp && apply_early && nrec > 0)
        fail = ensure_buf(c, c->lcpfull_buf, (NL + 8) * 4, "the LCP values beyond the byte") != 0;
      if (R > 1) TRY(comm_allgather(c, fail, nullptr, nullptr, 0));
      else if (fail) return -1;
      Bump a = {c->arena_p.as<u8>(), 0};
      layout_p(a);
    }
    const int nb = bits_for(N - 1);      // bits of a position / of a rank
    const int nbl = bits_for(NL ? NL - 1 : 0);   // bits of an index into the slice
    auto passes_for = [](int bits, int *ps, int *pw) -> int {
      int cnt = 0;
      for (int b = 0; b < bits; b += 8) {
        ps[cnt] = b;
        pw[cnt] = bits - b < 8 ? bits - b : 8;
        cnt++;
      }
      return cnt;
    };
    int ps[8], pw[8];
    const int pn = passes_for(nb, ps, pw);
    // ---- the pairs (and the pairs of members of the small groups): sorted by
    // text position, compared
    if (nrec > 0) {
      if (npairs > 0) {
        k_pair_emit<P><<<(u32) div_up(nwords, 256), 256, 0, st>>>(tiebits, nwords, poff, sa, pk_a,
                                                                 pv_a, pidx);
        HIP_TRY(hipGetLastError());
      }
      if (nsmall > 0) {
        k_small_emit<P><<<(u32) div_up(nwords, 256), 256, 0, st>>>(
            tiebits, nwords, soff, roff, sa, npairs, sidx, ssize, srec, pk_a, pv_a);
        HIP_TRY(hipGetLastError());
      }
      TRY(radix_sort_pairs<P, u64>(pk_a, pv_a, pk_b, pv_b, nrec, ps, pw, pn, prws, st,
                                   nullptr, nullptr));
      const P *pk_sorted = (pn & 1) ? pk_b : pk_a;
      const u64 *pv_sorted = (pn & 1) ? pv_b : pv_a;
      TRY(launch_emission());   // bandwidth-bound, beside the comparisons
      // pairs per thread: a chunk's first pair pays its whole comparison, the
      // others ride on the diagonal (3 Gbp, alternating in one process, 16 / 32 /
      // 64 / 128 pairs: 145.0 / 144.4 / 144.0 / 143.6 ms)
      // -- with a million threads at least: the kernel waits for its loads, and a
      // part of eight has an eighth of the records (3 Gbp, 43 M records a part, 8 /
      // 32 / 128 / 512 pairs per thread: 4.7 / 4.7 / 5.6 / 11.5 ms)
      int pair_chunk = (int) (nrec >> 20 < 16 ? 16 : (nrec >> 20 > 128 ? 128 : nrec >> 20));
      if (const char *e = getenv("GTAMD_PAIR_CHUNK")) { const int v = atoi(e); if (v >= 4 && v <= 1024) pair_chunk = v; }
      pair_chunk = (pair_chunk + PR_LINE_MAX - 1) / PR_LINE_MAX * PR_LINE_MAX;    // (whole lines of records per thread)
      {
        int line = PR_LINE_MAX, longext = 1;
        if (const char *e = getenv("GTAMD_PAIR_LONG")) longext = e[0] != '0';
        if (const char *e = getenv("GTAMD_PAIR_LINE")) { const int v = atoi(e); if (v == 4 || v == 8 || v == 16) line = v; }
        const u32 grid = stride_grid(div_up(div_up(nrec, (u64) pair_chunk), 256));
        if (line == 4)
          k_pair_resolve<BITS, P, 4><<<grid, 256, 0, st>>>(
              c->text, pk_sorted, pv_sorted, nrec, npairs, pidx, sa, pres, c->d_stats, pair_chunk, longext);
        else if (line == 8)
          k_pair_resolve<BITS, P, 8><<<grid, 256, 0, st>>>(
              c->text, pk_sorted, pv_sorted, nrec, npairs, pidx, sa, pres, c->d_stats, pair_chunk, longext);
        else
          k_pair_resolve<BITS, P, 16><<<grid, 256, 0, st>>>(
              c->text, pk_sorted, pv_sorted, nrec, npairs, pidx, sa, pres, c->d_stats, pair_chunk, longext);
      }
      HIP_TRY(hipGetLastError());
      if (nsmall > 0) {
        k_small_combine<P><<<(u32) div_up(nsmall, 256), 256, 0, st>>>(
            sidx, ssize, srec, pres, nsmall, sa, tiebits2, sres, slcp, c->d_stats);
        HIP_TRY(hipGetLastError());
      }
    }
    // ---- table entries of the pairs and of the small groups (random lines,
    // 11 ms at 3 Gbp): on the second stream, beside the rank table and the
    // doubling rounds (streaming kernels, then short ones that leave most of
    // the device idle).  Nothing on this stream touches those entries or the
    // pair lists until the join before the walk over the LCP table; the LCP
    // values beyond the byte need a buffer of their own for that -- behind the
    // rounds they went where the rank table had been.  GTAMD_APPLY_EARLY: 0 =
    // behind the rounds on this stream, 1 = from here on, 2 = from the first
    // round on.  The rank table's kernels are bound by HBM bandwidth, where every
    // byte of the entries costs its time again (k_win_filter 7.8 -> 13 ms beside
    // them); the rounds wait for latency and launches, and there the entries
    // are nearly free (3 Gbp, one process: 142.5 / 137.5 / 133.0 ms).
    u32 *lcpfull = nullptr;
    if (want_lcp && apply_early && nrec > 0) lcpfull = c->lcpfull_buf.as<u32>();   // (allocated above)
    if (nrec == 0) apply_early = 0;
    // Beside the refinement the two kernels get a grid of one workgroup per CU:
    // with a workgroup per 256 pairs the dispatcher kept every wave slot filled
    // with their waves and the short kernels of this stream -- the copy of the
    // statistics among them, which the host waits for -- queued behind them
    // for milliseconds (kernel trace: a 6 ms copy).  64 / 128 / 192 / 256 /
    // 320 / 384 / 512 / 1024 workgroups: 146.0 / 135.5 / 133.3 / 133.0 / 133.8 /
    // 134.3 / 136.4 / 138.0 ms -- fewer do not finish before the join.
    u64 apply_wgs = 256;
    {
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0)
        apply_wgs = (u64) cus;
    }
    if (const char *e = getenv("GTAMD_APPLY_WGS")) { const long v = atol(e); if (v >= 64 && v <= (1 << 22)) apply_wgs = (u64) v; }
    const bool apply_wgs_given = getenv("GTAMD_APPLY_WGS") != nullptr;
    auto launch_apply = [&](hipStream_t s) -> int {
      // One workgroup per CU is right while the rounds last as long as the
      // entries take that way (3 Gbp human-like: 13 + 1.6 ms of entries under
      // 12 ms of rounds; repeat-heavy: 35 under 195).  Many pairs and few
      // suffixes left for the rounds -- low-copy repeats only -- would leave the
      // small grid working alone behind the last round: the grid grows with what
      // the entries need over what the rounds give (measured rates: 77 ns per
      // 1000 pairs and 285 ns per 1000 small groups with one workgroup per CU,
      // 400 ns per 1000 suffixes in the rounds), up to the full one.
      u64 cap = ~0ull;
      if (apply_early == 1) cap = apply_wgs;
      else if (apply_early == 2 && m0 > 0) {
        cap = apply_wgs;
        if (!apply_wgs_given) {
          const double need = 7.7e-8 * (double) npairs + 2.85e-7 * (double) nsmall;   // ms
          const double have = 1.25 * 4.0e-7 * (double) m0 + 0.5;
          if (need > have) cap = (u64) ((double) apply_wgs * (need / have));
        }
      }
      if (npairs > 0) {
        const u64 g = div_up(npairs, 256);
        k_pair_apply<P><<<(u32) (g < cap ? g : cap), 256, 0, s>>>(
            pidx, pres, npairs, sa, d_suf, d_lcp, d_bwt, lcpfull, index_offset, c->d_stats);
        HIP_TRY(hipGetLastError());
      }
      if (nsmall > 0) {
        const u64 g = div_up(nsmall, 256);
        k_small_apply<P><<<(u32) (g < cap ? g : cap), 256, 0, s>>>(
            sidx, sres, slcp, nsmall, sa, d_suf, d_lcp, d_bwt, lcpfull, index_offset, c->d_stats);
        HIP_TRY(hipGetLastError());
      }
      return 0;
    };
    auto apply_beside = [&]() -> int {
      // (ev_sorted is free: the emission, if it is a kernel of its own, has been
      // launched by the pair path; st2 runs the entries behind it)
      // (a stream of the lowest priority for them changes nothing: 137.6 ms
      // against 136.9 -- the rounds lose 5 ms to the memory system, not to the
      // dispatcher)
      HIP_TRY(hipEventRecord(c->ev_sorted, st));
      HIP_TRY(hipStreamWaitEvent(c->st2, c->ev_sorted, 0));
      TRY(launch_apply(c->st2));
      HIP_TRY(hipEventRecord(c->ev_applied, c->st2));
      return 0;
    };
    if (apply_early == 1) TRY(apply_beside());
    TRY(tie_words(tiebits2));
    m0 = c->h_stats->count;
    const u64 smalldone = c->h_stats->smalldone;
    u64 anyleft = m0;
    if (R > 1) {
      std::vector<u64> all(R);
      TRY(comm_allgather(c, 0, &m0, all.data(), 8));
      anyleft = 0;
      for (u32 r = 0; r < R; r++) anyleft += all[r];
    }
    if (debug)
      fprintf(stderr, "gtamd: part %u: %llu tied with a neighbour, %llu pairs, %llu small groups "
              "(%llu entries settled), %llu left\n",
              c->part, (unsigned long long) numties, (unsigned long long) npairs,
              (unsigned long long) nsmall, (unsigned long long) smalldone,
              (unsigned long long) m0);
    // arena: unresolved list and round buffers | exchange buffers
    const u64 mp = (m0 + 64 + 3) & ~3ull;   // every array of the arena 16-byte aligned
    const u64 ISA_CHUNK = 1ull << 27;
    const u64 ichunk = NL < ISA_CHUNK ? NL : ISA_CHUNK;
    const u64 xm = dist ? (ichunk > mp ? ichunk : mp) : 0;   // items bucketed at a time
    // what one exchange can bring in: a position of the own tile is asked for /
    // updated at most once per round and no part sends more than it has tied
    // suffixes (or one chunk of first ranks)
    u64 xrecv_n = (u64) R * ISA_CHUNK > anyleft ? (u64) R * ISA_CHUNK : anyleft;
    if (xrecv_n > Tn) xrecv_n = Tn;
    xrecv_n += 64;
    P *upos = nullptr, *upos2 = nullptr, *cvo = nullptr,
      *k2 = nullptr, *fk2 = nullptr, *fk2s_a = nullptr, *fk2s_b = nullptr, *fpos = nullptr,
      *cvs = nullptr, *lk_a = nullptr, *lk_b = nullptr, *xrank = nullptr, *xans = nullptr;
    u32 *uidx0 = nullptr,
        *uidx = nullptr, *ugrp = nullptr, *uidx2 = nullptr, *ugrp2 = nullptr, *hv = nullptr,
        *koff = nullptr, *fgrp = nullptr, *fj = nullptr, *perm_a = nullptr, *perm_b = nullptr,
        *gk_a = nullptr, *gk_b = nullptr, *fhv = nullptr, *lv_a = nullptr, *lv_b = nullptr,
        *rws2 = nullptr, *scanws2 = nullptr, *xoff = nullptr, *xorder = nullptr,
        *xbc_q = nullptr, *xbo_q = nullptr, *xbc_u = nullptr, *xbo_u = nullptr, *xqoff = nullptr,
        *xmsg = nullptr;
    constexpr u64 RANKREC = 1 + sizeof(P) / 4;     // words of an (offset, rank) record
    u64 *keep = nullptr;
    u8 *xdest_q = nullptr, *xdest_u = nullptr;
    u32 *flagbits = nullptr, *tstart = nullptr;
    auto layout = [&](Bump &a) {
      uidx0 = a.take<u32>(mp); uidx = a.take<u32>(mp); ugrp = a.take<u32>(mp);
      uidx2 = a.take<u32>(mp); ugrp2 = a.take<u32>(mp);
      upos = a.take<P>(mp); upos2 = a.take<P>(mp);
      cvo = a.take<P>(mp);           // positions in the round's new order
      hv = a.take<u32>(mp);          // head values -> new group ids
      keep = a.take<u64>(mp / 64 + 4);   // 1 bit per slot
      koff = a.take<u32>(mp);
      k2 = a.take<P>(mp);            // rank of the suffix h further on
      flagbits = a.take<u32>(mp / 32 + RT_TILE / 32 + 64);   // deferred to the global path (1 bit per slot)
      tstart = a.take<u32>(mp / RT_STRIDE_MIN + 64);          // where the tiles of a round start
      // global path of a round (groups across tile borders)
      fk2 = a.take<P>(mp); fk2s_a = a.take<P>(mp); fk2s_b = a.take<P>(mp);
      fpos = a.take<P>(mp); cvs = a.take<P>(mp);
      fgrp = a.take<u32>(mp); fj = a.take<u32>(mp);
      perm_a = a.take<u32>(mp); perm_b = a.take<u32>(mp);
      gk_a = a.take<u32>(mp); gk_b = a.take<u32>(mp);
      fhv = a.take<u32>(mp);
      // LCP pairs of the tie fix
      lk_a = a.take<P>(mp); lk_b = a.take<P>(mp);
      lv_a = a.take<u32>(mp); lv_b = a.take<u32>(mp);
      rws2 = a.take<u32>(radix_workspace_words(m0));   // radix + scan workspace for the rounds
      scanws2 = a.take<u32>(scan_workspace_words(mp > dest_words(R, xm) ? mp : dest_words(R, xm)));
      if (dist) {
        xoff = a.take<u32>(xm + 64); xorder = a.take<u32>(xm + 64);
        xrank = a.take<P>(xm + 64); xans = a.take<P>(xm + 64);
        xqoff = a.take<u32>(xm + 64);
        // one message per exchange: the first ranks as records; a round's new
        // ranks (records) and queries (offsets) together
        xmsg = a.take<u32>((xm + 64) * (RANKREC + 1));
        xdest_q = a.take<u8>(xm + 64); xdest_u = a.take<u8>(xm + 64);
        xbc_q = a.take<u32>(dest_words(R, xm)); xbo_q = a.take<u32>(dest_words(R, xm));
        xbc_u = a.take<u32>(dest_words(R, xm)); xbo_u = a.take<u32>(dest_words(R, xm));
      }
    };
    {
      Bump sz = {nullptr, 0};
      layout(sz);
      fail = ensure_buf(c, c->arena, sz.off + 4096, "the refinement of tied suffixes") != 0;
      if (dist && !fail)
        fail = ensure_buf(c, c->xrecv, xrecv_n * (8 + 2 * sizeof(P)) + 1024,
                          "the exchange of ranks") != 0;
      if (dist && !fail)      // (bitmaps of the windows of 2^16 positions, see below)
        fail = ensure_buf(c, c->winbuf, (3 * (div_up(N, 1ull << 16) / 32 + 2) + 16) * 4,
                          "the rank windows") != 0;
      if (R > 1) TRY(comm_allgather(c, fail, nullptr, nullptr, 0));
      else if (fail) return -1;
      Bump a = {c->arena.as<u8>(), 0};
      layout(a);
    }
    // received: a message of up to xrecv_n records and xrecv_n offsets, and behind
    // it the answers to the offsets
    u32 *xrecv_msg = c->xrecv.as<u32>();
    P *xrecv_ans = reinterpret_cast<P *>(c->xrecv.as<u8>() + ((xrecv_n * (4 + 4 * RANKREC) + 255) & ~255ull));
    P *rank = nullptr;       // whole table (single build) ...
    P *isa = nullptr;        // ... or the ranks of the own text tile (part build)
    // ---- unresolved list of what is left
    if (m0 > 0) {
      k_unres_emit<P><<<(u32) div_up(nwords, 256), 256, 0, st>>>(
          tiebits2, nwords, offw, carry, sa, uidx0, uidx, upos, ugrp);
      HIP_TRY(hipGetLastError());
    }
    // ---- few suffixes left behind the pair path (random coincidences of three or
    // more: a protein set, a uniform text -- 410 K of 10^9 residues): settled by
    // direct comparison like the few ties of a small input; no rank table (its
    // filter alone walks the whole suffix array: 2 ms at 10^9), no round
    m0_tied_all = m0;                        // (what the statistics report)
    if (!dist && m0 > 0 && m0 <= (NL / 512 > 4096 ? NL / 512 : 4096)) {
      TRY(launch_emission());
      HIP_TRY(hipStreamWaitEvent(st, c->ev_emitted, 0));
      HIP_TRY(hipMemsetAsync(&c->d_stats->dfallback, 0, 4, st));
      k_direct_ties<BITS, P><<<(u32) div_up(m0, 256), 256, 0, st>>>(
          c->text, uidx0, ugrp, m0, sa, d_suf, d_lcp, d_bwt, want_lcp, index_offset, c->d_stats);
      HIP_TRY(hipGetLastError());
      TRY(fetch_stats(c));
      if (c->h_stats->dfallback == 0) {
        m0 = 0;
        anyleft = 0;
      } else {
        // (a deep or a big group: the rounds take them all; the partial statistics
        // of this attempt are dropped)
        HIP_TRY(hipMemsetAsync(&c->d_stats->dsum, 0, 8, st));
        HIP_TRY(hipMemsetAsync(&c->d_stats->dmax, 0, 4, st));
      }
    }
    // ---- rank table of a single build: the windows of positions the rounds can
    // touch (all of them when that is most of the text)
    int rk_wb = 0;
    u64 rk_nwin = 0, rk_h0 = 0;
    // (tests: the window bitmap of k_win_filter read from global memory, as a text of more
    // than 3.9 G symbols has it)
    const char *wfe = getenv("GTAMD_WIN_FILTER_LDS");
    const bool wf_global = wfe != nullptr && wfe[0] == '0';
    bool rk_windows = false;          // only some windows are built
    u32 *w_need = nullptr, *w_built = nullptr, *w_sel = nullptr, *w_list = nullptr;
    // builds the windows that are needed and not built (all == the whole table)
    std::function<int(bool)> build_rank = [](bool) -> int { return 0; };
    if (anyleft > 0 && !dist) {
      TRY(ensure_buf(c, c->isa_tmp, (NL + 8) * 8, "the rank table build"));
      u32 *rank32 = fval;
      rank = reinterpret_cast<P *>(rank32);
      u32 *heads = reinterpret_cast<u32 *>(fkey);          // free key buffer
      u32 *ppos = c->isa_tmp.as<u32>(), *phead = ppos + ((NL + 3) & ~3ull);  // (skey is still
                                                           // being read by the emission)
      u32 *qhead = heads, *qpos = heads + ((NL + 3) & ~3ull);
      int wmax = RW_BITS;
      if (const char *e = getenv("GTAMD_RANK_WINDOW_BITS")) {
        const int v = atoi(e);
        if (v >= 2 && v <= RW_BITS) wmax = v;
      }
      const u32 *spos = reinterpret_cast<const u32 *>(sa);
      const bool heads_array = bits_for(N - 1) <= wmax;
      // Partition down to windows that fit the LDS (one or two passes), then
      // k_rank_window.  (Measured at 3 Gbp: direct scatter 120 ms; one 8-bit
      // pass + global scatter 84 ms; two passes + LDS window 30 ms.)
      // GTAMD_RANK_WINDOW_BITS shrinks the window so that tests reach every
      // shape at small N.
      int pb = nb > wmax ? nb - wmax : 0;
      if (pb > 16) pb = 16;
      const int wb = nb - pb;             // wmax, or wmax + 1 with two halves
      if (wb > wmax + 1) {
        gtamd_set_error("rank table: %d position bits do not fit two passes and a "
                        "%d-bit window", nb, wmax);
        return -1;
      }
      const int split = wb > wmax ? 2 : 1;
      // the windows that are selected are finer than the windows the LDS takes
      const int fb = wmax < RW_FINE ? wmax : RW_FINE;
      rk_wb = fb;
      rk_nwin = div_up(N, 1ull << fb);
      const u64 nww = rk_nwin / 32 + 2;
      const char *we = getenv("GTAMD_RANK_ALL_WINDOWS");           // A/B switch
      const bool all_windows = (we != nullptr && we[0] == '1') || pb == 0;
      u32 *w_pref = nullptr;
      if (!all_windows) {
        TRY(ensure_buf(c, c->winbuf, (4 * nww + rk_nwin + 16) * 4, "the rank windows"));
        w_need = c->winbuf.as<u32>(); w_built = w_need + nww; w_sel = w_built + nww;
        w_pref = w_sel + nww;
        w_list = w_pref + nww;
        HIP_TRY(hipMemsetAsync(w_need, 0, 2 * nww * 4, st));
        // offsets of the first rounds
        rk_h0 = (u64) K::KNOWN << 9;
        if (rk_h0 > (3ull << wb)) rk_h0 = 3ull << wb;
        if (m0 > 0) {
          k_win_mark<P><<<(u32) div_up(m0, 256), 256, 0, st>>>(upos, m0, 0, rk_h0, fb, rk_nwin,
                                                              w_need);
          HIP_TRY(hipGetLastError());
        }
        rk_windows = true;
      }
      const GroupHeadValues headgen = {tiebits2, carry, nwords, 0u};
      build_rank = [=, &rk_windows, &rank_built](bool first) -> int {
        u64 nsel = 0;
        if (rk_windows) {
          k_win_select<<<1, 1024, 0, st>>>(w_need, w_built, nww, w_sel, w_list, w_pref, c->d_stats);
          HIP_TRY(hipGetLastError());
          TRY(fetch_stats(c));
          nsel = c->h_stats->count;
          if (nsel == 0) return 0;
          if (first && nsel * 2 > rk_nwin) rk_windows = false;   // most of the text: all of it
          if (debug)
            fprintf(stderr, "gtamd: rank table: %llu of %llu windows of 2^%d positions%s\n",
                    (unsigned long long) nsel, (unsigned long long) rk_nwin, fb,
                    rk_windows ? "" : " -> whole table");
        }
        if (!rk_windows) {
          if (heads_array) {
            k_heads<<<(u32) div_up(NL, 1024), 256, 0, st>>>(tiebits2, carry, NL, 0u, heads);
            HIP_TRY(hipGetLastError());
          }
          const u32 *wpos = spos, *whead = heads;
          if (pb > 8) {
            const int s0 = nb - pb, w0 = pb - 8, s1 = nb - 8, w1 = 8;   // heads is dead by then
            if (heads_array)
              TRY(radix_partition_u32(spos, heads, ppos, phead, NL, s0, w0, pws, st));
            else
              TRY(radix_pass_group_heads(spos, headgen, ppos, phead, NL, s0, w0, pws, st));
            // (the second pass has to keep the first one's grouping: the sort's stable pass)
            TRY(radix_sort_pairs<u32, u32>(ppos, phead, qpos, qhead, NL, &s1, &w1, 1, pws, st,
                                           nullptr, nullptr));
            wpos = qpos; whead = qhead;
          } else if (pb > 0) {
            const int s0 = nb - pb, w0 = pb;
            if (heads_array)
              TRY(radix_partition_u32(spos, heads, ppos, phead, NL, s0, w0, pws, st));
            else
              TRY(radix_pass_group_heads(spos, headgen, ppos, phead, NL, s0, pb, pws, st));
            wpos = ppos; whead = phead;
          }
          const u32 nbuckets = (u32) div_up(NL, 1ull << wb);
          const u32 grid = split == 2 ? ((nbuckets + 7u) / 8u) * 16u : nbuckets;
          k_rank_window<<<grid, RW_THREADS, 0, st>>>(wpos, whead, NL, wb, split, nbuckets, rank32,
                                                     nullptr, 0, 0u, 0);
          HIP_TRY(hipGetLastError());
          rank_built = NL;
          return 0;
        }
        // the pairs of the selected windows, with compact positions, partitioned down
        // to the windows the LDS takes
        HIP_TRY(hipMemsetAsync(&c->d_stats->count, 0, 4, st));
        if ((nww + 4) * 5 <= WF_LDS_MAX && !wf_global) {
          // (more than the 64 KB a kernel gets without asking)
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_win_filter<u32, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int) wf_lds_bytes<u32>(nww, true)));
          k_win_filter<u32, true><<<(u32) div_up(NL, WF_SPAN * WF_ITER), WF_THREADS, wf_lds_bytes<u32>(nww, true), st>>>(
              spos, NL, fb, w_sel, (u32) nww, w_pref, tiebits2, carry, ppos, phead, ~0ull, c->d_stats);
        } else
          k_win_filter<u32, false><<<(u32) div_up(NL, WF_SPAN * WF_ITER), WF_THREADS, wf_lds_bytes<u32>(nww, false), st>>>(
              spos, NL, fb, w_sel, (u32) nww, w_pref, tiebits2, carry, ppos, phead, ~0ull, c->d_stats);
        HIP_TRY(hipGetLastError());
        TRY(fetch_stats(c));
        const u64 M = c->h_stats->count;   // nsel windows (the last one of the text is short)
        if (M == 0) return 0;
        rank_built += M;
        int mb = bits_for(((u64) nsel << fb) - 1);
        if (mb < fb) mb = fb;
        int pbm = mb > wmax ? mb - wmax : 0;
        if (pbm > 16) pbm = 16;
        const int wbm = mb - pbm;          // (<= wmax + 1, as mb <= nb)
        const int splitm = wbm > wmax ? 2 : 1;
        const u32 *wpos = ppos, *whead = phead;
        if (pbm > 8) {
          const int s0 = mb - pbm, w0 = pbm - 8, s1 = mb - 8, w1 = 8;
          TRY(radix_partition_u32(ppos, phead, qpos, qhead, M, s0, w0, pws, st));
          TRY(radix_sort_pairs<u32, u32>(qpos, qhead, ppos, phead, M, &s1, &w1, 1, pws, st,
                                         nullptr, nullptr));
        } else if (pbm > 0) {
          const int s0 = mb - pbm, w0 = pbm;
          TRY(radix_partition_u32(ppos, phead, qpos, qhead, M, s0, w0, pws, st));
          wpos = qpos; whead = qhead;
        }
        const u32 nbuckets = (u32) div_up(M, 1ull << wbm);
        const u32 grid = splitm == 2 ? ((nbuckets + 7u) / 8u) * 16u : nbuckets;
        k_rank_window<<<grid, RW_THREADS, 0, st>>>(wpos, whead, N, wbm, splitm, nbuckets, rank32,
                                                   w_list, fb, (u32) nsel, M);
        HIP_TRY(hipGetLastError());
        return 0;
      };
      TRY(build_rank(true));
    }
    std::vector<u64> qcounts(R), ucounts(R), zero(R, 0);
    const size_t GW = 2 * (size_t) R + 2;    // a part's row in a round's allgather
    std::vector<u64> gathered((size_t) R * GW), mine(GW);
    // part builds: the first ranks go to the owners of the positions, ISA_CHUNK
    // entries at a time -- only for the windows of positions the rounds can
    // reach (the union over the parts; see k_win_mark), which is a fifth of the
    // text for the human-like workload: the exchange shrinks accordingly
    std::vector<u32> h_need, h_built, h_all;
    u64 dw_nww = 0;
    std::function<int()> send_ranks = []() -> int { return 0; };
    if (anyleft > 0 && dist) {
      isa = WIDE ? reinterpret_cast<P *>(fkey) : reinterpret_cast<P *>(fval);
      rk_wb = 16;
      rk_nwin = div_up(N, 1ull << rk_wb);
      dw_nww = rk_nwin / 32 + 2;
      const char *we = getenv("GTAMD_RANK_ALL_WINDOWS");           // A/B switch
      rk_windows = !(we != nullptr && we[0] == '1');
      // (a part that cannot get the buffers of this step says so in the first
      // allgather of send_ranks)
      fail |= ensure_buf(c, c->winbuf, (3 * dw_nww + 16) * 4, "the rank windows") != 0;
      if (fail && R == 1) return -1;
      w_need = c->winbuf.as<u32>(); w_built = w_need + dw_nww; w_sel = w_built + dw_nww;
      h_need.assign(dw_nww, 0u); h_built.assign(dw_nww, 0u); h_all.assign((size_t) dw_nww * R, 0u);
      if (!fail) HIP_TRY(hipMemsetAsync(w_need, 0, 3 * dw_nww * 4, st));
      rk_h0 = (u64) K::KNOWN << 9;
      if (rk_h0 > (3ull << rk_wb)) rk_h0 = 3ull << rk_wb;
      if (!fail && rk_windows && m0 > 0) {
        k_win_mark<P><<<(u32) div_up(m0, 256), 256, 0, st>>>(upos, m0, 0, rk_h0, rk_wb, rk_nwin,
                                                            w_need);
        HIP_TRY(hipGetLastError());
      }
      // where the entries of the windows that travel are listed: a buffer that has
      // done its work (the keys the part filtered from the text; their positions)
      P *fpos = nullptr;
      u32 *fhead = nullptr;
      u64 list_cap = 0;
      if (!WIDE) {
        list_cap = NL;
        fpos = c->isa_tmp.as<P>();
        fhead = c->isa_tmp.as<u32>() + ((NL + 3) & ~3ull);
      } else if (msd_part && c->posw.bytes >= 4096) {
        list_cap = (c->posw.bytes - 256) / 12;
        fpos = c->posw.as<P>();
        fhead = reinterpret_cast<u32 *>(c->posw.as<u8>() + ((list_cap * 8 + 255) & ~255ull));
        list_cap -= 64;
      }
      send_ranks = [&]() -> int {
        // the windows any part needs and nobody has sent yet (the first allgather of
        // this step also says whether a part could not get its buffers)
        const u32 *d_sel = nullptr;
        int agreed = 0;
        if (rk_windows) {
          if (!fail) {
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipMemcpy(h_need.data(), w_need, dw_nww * 4, hipMemcpyDeviceToHost));
          }
          TRY(comm_allgather(c, fail, h_need.data(), h_all.data(), (u32) (dw_nww * 4)));
          agreed = 1;
          u64 fresh = 0, all = 0;
          for (u64 w = 0; w < dw_nww; w++) {
            u32 x = 0;
            for (u32 r = 0; r < R; r++) x |= h_all[(size_t) r * dw_nww + w];
            x &= ~h_built[w];
            h_need[w] = x;
            h_built[w] |= x;
            fresh += (u64) __builtin_popcount(x);
            all += (u64) __builtin_popcount(h_built[w]);
          }
          if (debug)
            fprintf(stderr, "gtamd: part %u: ranks of %llu more windows of 2^%d positions travel "
                    "(%llu of %llu so far)\n", c->part, (unsigned long long) fresh, rk_wb,
                    (unsigned long long) all, (unsigned long long) rk_nwin);
          if (fresh == 0) return 0;
          HIP_TRY(hipMemcpyAsync(w_sel, h_need.data(), dw_nww * 4, hipMemcpyHostToDevice, st));
          HIP_TRY(hipMemcpyAsync(w_built, h_built.data(), dw_nww * 4, hipMemcpyHostToDevice, st));
          d_sel = w_sel;
        }
        // the entries of those windows, listed in one walk over the slice (a fifth of
        // it for the human-like workload): what is bucketed and sent is the list.  A
        // list that does not fit its buffer (or all windows): the slice itself, chunk
        // by chunk.
        bool listed = false;
        u64 M = 0;
        if (!fail && d_sel != nullptr && NL > 0 && list_cap > 0) {
          HIP_TRY(hipMemsetAsync(&c->d_stats->count, 0, 8, st));     // count, count2
          // (the bitmap of the windows of 64 K positions: 6 KB for 3 Gbp -- in LDS whenever it fits)
          if ((dw_nww + 4) * 5 <= WF_LDS_MAX && !wf_global) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_win_filter<P, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int) wf_lds_bytes<P>(dw_nww, true)));
            k_win_filter<P, true><<<(u32) div_up(NL, WF_SPAN * WF_ITER), WF_THREADS, wf_lds_bytes<P>(dw_nww, true), st>>>(
                sa, NL, rk_wb, d_sel, (u32) dw_nww, nullptr, tiebits2, carry, fpos, fhead, list_cap, c->d_stats);
          } else
            k_win_filter<P, false><<<(u32) div_up(NL, WF_SPAN * WF_ITER), WF_THREADS, wf_lds_bytes<P>(0, false), st>>>(
                sa, NL, rk_wb, d_sel, 0u, nullptr, tiebits2, carry, fpos, fhead, list_cap, c->d_stats);
          HIP_TRY(hipGetLastError());
          TRY(fetch_stats(c));
          listed = c->h_stats->count2 == 0;
          M = c->h_stats->count;
        }
        const u64 items = listed ? M : NL, per = listed ? (xm ? xm : 1) : ISA_CHUNK;
        u64 chunks = div_up(items, per);
        if (R > 1 || !agreed) {
          std::vector<u64> all(R);
          TRY(comm_allgather(c, agreed ? 0 : fail, &chunks, all.data(), 8));
          for (u32 r = 0; r < R; r++) chunks = all[r] > chunks ? all[r] : chunks;
        }
        for (u64 ch = 0; ch < chunks; ch++) {
          const u64 c0 = ch * per < items ? ch * per : items;
          const u64 cm = items - c0 < per ? items - c0 : per;
          if (listed) {
            FilteredRanks<P> fr;
            fr.fpos = fpos + c0; fr.fhead = fhead + c0; fr.index_offset = index_offset; fr.tl = tl;
            fr.isa = isa; fr.srec = xmsg;
            TRY(dest_count(c, fr, cm, xdest_q, xbc_q, xbo_q, scanws2, 0));
            HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, DEST_MAXPARTS * 4,
                                   hipMemcpyDeviceToHost, st));
            TRY(dest_place(c, fr, cm, xdest_q, xbo_q));
          } else {
            InitialRanks<P> ir;
            ir.sa = sa; ir.tiebits = tiebits2; ir.carry = carry; ir.c0 = c0;
            ir.index_offset = index_offset; ir.tl = tl; ir.isa = isa; ir.srec = xmsg;
            ir.sel = d_sel; ir.wb = rk_wb;
            TRY(dest_count(c, ir, cm, xdest_q, xbc_q, xbo_q, scanws2, 0));
            HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, DEST_MAXPARTS * 4,
                                   hipMemcpyDeviceToHost, st));
            TRY(dest_place(c, ir, cm, xdest_q, xbo_q));
          }
          HIP_TRY(hipStreamSynchronize(st));
          std::vector<u64> sc(R), rc(R), mat((size_t) R * R);
          for (u32 r = 0; r < R; r++) sc[r] = c->h_counts[r];
          TRY(comm_allgather(c, 0, sc.data(), mat.data(), R * 8));
          u64 nrecv = 0;
          for (u32 s = 0; s < R; s++) { rc[s] = mat[(size_t) s * R + c->part]; nrecv += rc[s]; }
          // (every part sees the whole matrix: all of them leave here, or none)
          int toobig = 0;
          for (u32 q = 0; q < R && !toobig; q++) {
            u64 tot = 0;
            for (u32 s = 0; s < R; s++) tot += mat[(size_t) s * R + q];
            const u64 first_q = (u64) q * tl.T < N ? (u64) q * tl.T : N;
            const u64 tile_q = first_q + tl.T < N ? tl.T : N - first_q;
            if (tot > tile_q) toobig = 1;
          }
          if (toobig || nrecv + 64 > xrecv_n) {
            gtamd_set_error("rank exchange: %llu ranks for a tile of %llu positions",
                            (unsigned long long) nrecv, (unsigned long long) Tn);
            return -1;
          }
          TRY(comm_alltoallv(c, xmsg, sc.data(), xrecv_msg, rc.data(), (u32) (4 * RANKREC), "first ranks"));
          if (nrecv > 0) {
            k_isa_store_rec<P><<<(u32) div_up(nrecv, 256), 256, 0, st>>>(xrecv_msg, nrecv, isa);
            HIP_TRY(hipGetLastError());
          }
        }
        return 0;
      };
      TRY(send_ranks());
    }
    TRY(launch_emission());   // (if the pair path has not started it)
    if (apply_early == 2) TRY(apply_beside());
    // ---- doubling rounds
    int gs[8], gw[8];
    const int gn = passes_for(nbl, gs, gw);
    u64 m = m0, h = (u64) K::KNOWN;
    // nominal distance of the round tiles' starts: the rest of a tile is the
    // slack for the group that lies across (a group larger than the slack goes
    // through the global path, forty launches per round)
    u32 rt_stride = 1536;
    if (const char *e = getenv("GTAMD_ROUND_STRIDE")) {
      const int v = atoi(e);
      if (v >= RT_STRIDE_MIN && v <= RT_TILE) rt_stride = (u32) v;
    }
    // part builds: the queries of the coming round and the rank updates of the
    // last one are bucketed by owner before the parts agree on the counts
    RankQueries<P> rq;
    RankUpdates<P> ru;
    u64 m_upd = 0;            // slots of the last round (its updates are pending)
    if (anyleft > 0 && dist) {
      rq.upos = upos; rq.h = h; rq.n = n; rq.tl = tl; rq.isa = isa; rq.k2 = k2;
      rq.sendq = xqoff; rq.order = xorder;
      TRY(dest_count(c, rq, m, xdest_q, xbc_q, xbo_q, scanws2, 0));
      HIP_TRY(hipMemsetAsync(c->d_counts + DEST_MAXPARTS, 0, DEST_MAXPARTS * 4, st));
    }
    for (;;) {
      if (anyleft == 0) break;
      if (!dist && m == 0) break;
      if (rounds >= 64) {
        gtamd_set_error("prefix doubling did not converge after 64 rounds");
        return -1;
      }
      if (dist) {
        // does this round's offset reach windows whose ranks have not travelled?
        // (never while h <= rk_h0; the parts decide together, in the allgather below)
        u64 more = 0;
        if (rk_windows && h > rk_h0) {
          HIP_TRY(hipMemsetAsync(&c->d_stats->count2, 0, 4, st));
          if (m > 0) {
            k_win_check<P><<<(u32) div_up(m, 256), 256, 0, st>>>(upos, m, h, rk_wb, rk_nwin, w_built,
                                                               w_need, c->d_stats);
            HIP_TRY(hipGetLastError());
          }
          HIP_TRY(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(Stats), hipMemcpyDeviceToHost, st));
        }
        // ONE allgather per round: pending updates, queries, who is left, and
        // whether more first ranks have to travel
        HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, 2 * DEST_MAXPARTS * 4,
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (rk_windows && h > rk_h0) more = c->h_stats->count2 != 0;
        for (u32 r = 0; r < R; r++) {
          mine[r] = c->h_counts[r];                        // queries to r
          mine[R + r] = c->h_counts[DEST_MAXPARTS + r];    // updates to r
        }
        mine[2 * R] = m;
        mine[2 * R + 1] = more;
        TRY(comm_allgather(c, 0, mine.data(), gathered.data(), (u32) (GW * 8)));
        u64 left = 0;
        for (u32 s = 0; s < R; s++) {
          left += gathered[(size_t) s * GW + 2 * R];
          more |= gathered[(size_t) s * GW + 2 * R + 1];
        }
        if (left == 0) break;   // (ranks nobody will ask for need not travel)
        if (more) {
          TRY(send_ranks());
          // (the exchange used the bucketing buffers of this round's queries; the
          // counts come out the same)
          TRY(dest_count(c, rq, m, xdest_q, xbc_q, xbo_q, scanws2, 0));
        }
        // the new ranks of the last round and the queries of this one: bucketed by
        // owner (own tile: stored / answered in place) ...
        if (m_upd > 0) TRY(dest_place(c, ru, m_upd, xdest_u, xbo_u));
        m_upd = 0;
        rq.sendq = xqoff; rq.order = xorder;
        TRY(dest_place(c, rq, m, xdest_q, xbo_q));
        if (R > 1) {
          // ... and sent as ONE message per part: its update records, then its query
          // offsets; the owners store the updates before they look anything up
          // (k_fuse_updates, then k_fuse_answers) and send the answers back
          constexpr u32 W = (u32) RANKREC;
          FuseTab ts, tr;
          ts.n = tr.n = R;
          std::vector<u64> sw(R), rw(R), sq(R), rqc(R);
          ts.upre[0] = ts.qpre[0] = tr.upre[0] = tr.qpre[0] = 0;
          ts.wpre[0] = tr.wpre[0] = 0;
          for (u32 r = 0; r < R; r++) {
            const u64 us = mine[R + r], qs = mine[r];
            const u64 ur = gathered[(size_t) r * GW + R + c->part], qr = gathered[(size_t) r * GW + c->part];
            sw[r] = us * W + qs; rw[r] = ur * W + qr;
            sq[r] = qs; rqc[r] = qr;
            ts.upre[r + 1] = ts.upre[r] + (u32) us; ts.qpre[r + 1] = ts.qpre[r] + (u32) qs;
            ts.wpre[r + 1] = ts.wpre[r] + sw[r];
            tr.upre[r + 1] = tr.upre[r] + (u32) ur; tr.qpre[r + 1] = tr.qpre[r] + (u32) qr;
            tr.wpre[r + 1] = tr.wpre[r] + rw[r];
          }
          // (a position of a tile is updated and asked for at most once per round;
          // every part sees the whole matrix, so all of them leave here or none)
          int toobig = 0;
          for (u32 q = 0; q < R; q++) {
            u64 tu = 0, tq = 0;
            for (u32 s = 0; s < R; s++) {
              tu += gathered[(size_t) s * GW + R + q];
              tq += gathered[(size_t) s * GW + q];
            }
            const u64 first_q = (u64) q * tl.T < N ? (u64) q * tl.T : N;
            const u64 tile_q = first_q + tl.T < N ? tl.T : N - first_q;
            if (tu > tile_q || tq > tile_q) toobig = 1;
          }
          if (toobig || (u64) tr.upre[R] + 64 > xrecv_n || (u64) tr.qpre[R] + 64 > xrecv_n) {
            gtamd_set_error("rank exchange: %u updates and %u queries for a tile of %llu positions",
                            tr.upre[R], tr.qpre[R], (unsigned long long) Tn);
            return -1;
          }
          const u64 nsend = (u64) ts.upre[R] + ts.qpre[R];
          if (nsend > 0) {
            k_fuse_pack<P><<<(u32) div_up(nsend, 256), 256, 0, st>>>(ts, xoff, xrank, xqoff, xmsg);
            HIP_TRY(hipGetLastError());
          }
          TRY(comm_alltoallv(c, xmsg, sw.data(), xrecv_msg, rw.data(), 4, "new ranks and queries"));
          if (tr.upre[R] > 0) {
            k_fuse_updates<P><<<(u32) div_up(tr.upre[R], 256), 256, 0, st>>>(tr, xrecv_msg, isa);
            HIP_TRY(hipGetLastError());
          }
          if (tr.qpre[R] > 0) {
            k_fuse_answers<P><<<(u32) div_up(tr.qpre[R], 256), 256, 0, st>>>(tr, xrecv_msg, isa, xrecv_ans);
            HIP_TRY(hipGetLastError());
          }
          // (the look-ups in the own tile, now that everybody's new ranks are in)
          if (m > 0) {
            k_local_lookups<P><<<(u32) div_up(m, 256), 256, 0, st>>>(rq, m);
            HIP_TRY(hipGetLastError());
          }
          TRY(comm_alltoallv(c, xrecv_ans, rqc.data(), xans, sq.data(), sizeof(P), "answers"));
          if (ts.qpre[R] > 0) {
            k_k2_scatter<P><<<(u32) div_up(ts.qpre[R], 256), 256, 0, st>>>(xans, xorder, ts.qpre[R], k2);
            HIP_TRY(hipGetLastError());
          }
        } else if (m > 0) {
          k_local_lookups<P><<<(u32) div_up(m, 256), 256, 0, st>>>(rq, m);
          HIP_TRY(hipGetLastError());
        }
      }
      rounds++;
      if (debug)
        fprintf(stderr, "gtamd: part %u round %u h=%llu tied=%llu\n", c->part, rounds,
                (unsigned long long) h, (unsigned long long) m);
      if (m == 0) {   // only serving other parts' queries
        h *= 2;
        HIP_TRY(hipMemsetAsync(c->d_counts, 0, 2 * DEST_MAXPARTS * 4, st));
        continue;
      }
      if (!dist && rk_windows && h > rk_h0) {
        // an offset beyond the windows built so far?  then build what it reaches
        HIP_TRY(hipMemsetAsync(&c->d_stats->count2, 0, 4, st));
        k_win_check<P><<<(u32) div_up(m, 256), 256, 0, st>>>(upos, m, h, rk_wb, rk_nwin, w_built,
                                                           w_need, c->d_stats);
        HIP_TRY(hipGetLastError());
        TRY(fetch_stats(c));
        if (c->h_stats->count2 != 0) TRY(build_rank(false));
      }
      const u32 g = (u32) div_up(m, 256);
      const u32 ntiles = (u32) div_up(m, rt_stride);   // tiles that start at group borders
      const u32 nfb = (u32) div_up(m, RT_TILE);         // blocks of the deferred-slot bitmap
      u32 *tilecnt = koff, *tileoff = koff + nfb + 16;  // (free until the apply step)
      HIP_TRY(hipMemsetAsync(flagbits, 0, (u64) nfb * (RT_TILE / 8), st));
      k_tile_starts<<<ntiles / 256 + 1, 256, 0, st>>>(ugrp, m, ntiles, rt_stride, tstart);
      HIP_TRY(hipGetLastError());
      k_round_tile<P><<<ntiles, RT_THREADS, 0, st>>>(
          uidx, upos, ugrp, k2, m, cvo, hv, tstart, flagbits, dist ? nullptr : rank, h, n);
      HIP_TRY(hipGetLastError());
      k_flag_count<<<nfb / 256 + 1, 256, 0, st>>>(flagbits, m, nfb, tilecnt);
      HIP_TRY(hipGetLastError());
      TRY(scan_u32(SCAN_SUM, tilecnt, tileoff, nfb, false, scanws2, st));
      k_total<<<1, 1, 0, st>>>(tileoff, tilecnt, nfb, c->d_stats);
      HIP_TRY(hipGetLastError());
      TRY(fetch_stats(c));
      const u64 nf = c->h_stats->count;
      if (debug)
        fprintf(stderr, "gtamd: part %u round %u: %llu entries in groups across tile borders\n",
                c->part, rounds, (unsigned long long) nf);
      if (nf > 0) {
        // groups crossing a tile border / larger than a tile: ordered by
        // (group, k2) through two stable sorts of an index
        k_flag_gather<P><<<nfb, RT_THREADS, 0, st>>>(flagbits, tileoff, ugrp, k2, upos, m, fk2,
                                                       fk2s_a, fgrp, fpos, fj, perm_a);
        HIP_TRY(hipGetLastError());
        TRY(radix_sort_pairs<P, u32>(fk2s_a, perm_a, fk2s_b, perm_b, nf, ps, pw, pn, rws2, st,
                                     nullptr, nullptr));
        u32 *perm1 = (pn & 1) ? perm_b : perm_a, *permx = (pn & 1) ? perm_a : perm_b;
        const u32 gf = (u32) div_up(nf, 256);
        k_gather_u32<<<gf, 256, 0, st>>>(fgrp, perm1, nf, gk_a);
        HIP_TRY(hipGetLastError());
        TRY(radix_sort_pairs<u32, u32>(gk_a, perm1, gk_b, permx, nf, gs, gw, gn, rws2, st,
                                       nullptr, nullptr));
        const u32 *perm2 = (gn & 1) ? permx : perm1;
        k_flag_heads<P><<<gf, 256, 0, st>>>(perm2, fgrp, fk2, fpos, uidx, fj, nf, fhv, cvs);
        HIP_TRY(hipGetLastError());
        TRY(scan_u32(SCAN_MAX, fhv, fhv, nf, true, scanws2, st));
        k_flag_scatter<P><<<gf, 256, 0, st>>>(cvs, fhv, fj, nf, cvo, hv);
        HIP_TRY(hipGetLastError());
      }
      // (koff doubles as the per-block survivor counts and their scan)
      u32 *bcnt = koff, *boffs = koff + g + 16;
      k_round_apply<P><<<g, 256, 0, st>>>(cvo, hv, uidx, ugrp, m, index_offset, sa,
                                          dist ? nullptr : rank, keep, bcnt);
      HIP_TRY(hipGetLastError());
      TRY(scan_u32(SCAN_SUM, bcnt, boffs, g, false, scanws2, st));
      k_round_compact<P><<<g, 256, 0, st>>>(keep, boffs, bcnt, uidx, cvo, hv, m, uidx2,
                                            upos2, ugrp2, c->d_stats);
      HIP_TRY(hipGetLastError());
      TRY(fetch_stats(c));
      const u64 m_old = m;
      m = c->h_stats->count;
      h *= 2;
      if (dist) {
        // the new ranks of this round (slots of the old list) and the queries
        // of the next (the compacted list): counted now, exchanged at the top
        ru.cval = cvo; ru.gnew = hv; ru.ugrp = ugrp; ru.index_offset = index_offset;
        ru.tl = tl; ru.isa = isa; ru.soff = xoff; ru.srank = xrank;
        m_upd = m_old;
        TRY(dest_count(c, ru, m_upd, xdest_u, xbc_u, xbo_u, scanws2, 1));
        rq.upos = upos2; rq.h = h;
        TRY(dest_count(c, rq, m, xdest_q, xbc_q, xbo_q, scanws2, 0));
      }
      u32 *t;
      t = uidx; uidx = uidx2; uidx2 = t;
      t = ugrp; ugrp = ugrp2; ugrp2 = t;
      P *tp = upos; upos = upos2; upos2 = tp;
      if (dist) ru.ugrp = ugrp2;   // (the old list's groups, after the swap)
    }
    HIP_TRY(hipEventRecord(c->ev[5], st));
    // ---- final entries of the tied suffixes (after the emission has
    // written its provisional values)
    TRY(launch_emission());
    HIP_TRY(hipStreamWaitEvent(st, c->ev_emitted, 0));
    // 32-bit LCP values that do not fit the byte, by table index: in a buffer
    // that has done its work (the rank table; the bucketing scratch)
    if (want_lcp && !apply_early) {
      if (!dist) lcpfull = fval;
      else if (WIDE) lcpfull = fval;
      else lcpfull = c->isa_tmp.as<u32>();
    }
    if (!apply_early) TRY(launch_apply(st));
    const u32 g0 = (u32) div_up(m0, 256);
    u32 *bcnt0 = koff, *boff0 = koff + g0 + 16;   // per-workgroup counts and their scan
    if (m0 > 0) {
      // .suf/.bwt of the tied entries: random accesses, on the second stream
      // (behind the emission there) while this stream sorts the LCP pairs
      HIP_TRY(hipEventRecord(c->ev_sorted, st));          // rounds are done
      HIP_TRY(hipStreamWaitEvent(c->st2, c->ev_sorted, 0));
      k_fix_basic<BITS, P><<<g0, 256, 0, c->st2>>>(c->text, uidx0, m0, sa, d_suf, d_bwt,
                                                   c->d_stats, index_offset);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(c->ev_emitted, c->st2));
    }
    if (want_lcp && m0 > 0) {
      // entries tied with their predecessor, sorted by text position
      k_tied_counts<<<g0, 256, 0, st>>>(uidx0, m0, tiebits2, bcnt0);
      HIP_TRY(hipGetLastError());
      TRY(scan_u32(SCAN_SUM, bcnt0, boff0, g0, false, scanws2, st));
      k_total<<<1, 1, 0, st>>>(boff0, bcnt0, g0, c->d_stats);
      HIP_TRY(hipGetLastError());
      TRY(fetch_stats(c));
      const u64 m1 = c->h_stats->count;
      k_lcp_pairs<P><<<g0, 256, 0, st>>>(boff0, tiebits2, uidx0, sa, m0, lk_a, lv_a);
      HIP_TRY(hipGetLastError());
      TRY(radix_sort_pairs<P, u32>(lk_a, lv_a, lk_b, lv_b, m1, ps, pw, pn, rws2, st,
                                   nullptr, nullptr));
      const P *pk = (pn & 1) ? lk_b : lk_a;
      const u32 *pv = (pn & 1) ? lv_b : lv_a;
      k_lcp_chunks<BITS, P><<<stride_grid(div_up(div_up(m1, LCP_CHUNK), 256)), 256, 0, st>>>(
          c->text, pk, pv, m1, sa, d_lcp, lcpfull, c->d_stats);
      HIP_TRY(hipGetLastError());
    }
    // (the pairs' entries written beside all this: the tie fix above touches
    // other entries; the walk over the whole LCP table below needs them)
    if (apply_early) HIP_TRY(hipStreamWaitEvent(st, c->ev_applied, 0));
    if (want_lcp) {
      // .llv from the byte table and the side table
      TRY(fetch_stats(c));
      const u64 pairs = c->h_stats->numlarge;
      if (pairs > c->llv_cap) {
        free_dev(c->llv);
        c->llv = nullptr;
        c->llv_cap = 0;
        HIP_TRY(hipMalloc(&c->llv, (pairs + pairs / 4 + 1024) * 16));
        c->llv_cap = pairs + pairs / 4 + 1024;
      }
      if (pairs > 0) {
        const u32 gl = (u32) div_up(NL, LLV_TILE);
        // (counts and their scan: the round buffers are free, but sized for the
        // tied suffixes -- the per-word arrays of the radix workspace hold
        // NL / 64 + 16 words each, more than the NL / 4096 needed here)
        u32 *lc = cntw, *lo = offw;
        k_large_counts<<<gl, 256, 0, st>>>(d_lcp, NL, lc);
        HIP_TRY(hipGetLastError());
        TRY(scan_u32(SCAN_SUM, lc, lo, gl, false, scanws, st));
        k_llv_emit<<<gl, 256, 0, st>>>(d_lcp, NL, lcpfull, lo, index_offset, c->llv);
        HIP_TRY(hipGetLastError());
      }
      c->llv_pairs = pairs;
    }
    }  // !settled
  }
  TRY(launch_emission());
  HIP_TRY(hipStreamWaitEvent(st, c->ev_emitted, 0));
  HIP_TRY(hipEventRecord(c->ev[6], st));
  TRY(fetch_stats(c));

  // results (of this part; the caller combines parts: sums, max, and the one
  // part that holds suffix 0 reports `longest`)
  c->stats.totallength = n;
  c->stats.numberofallsortedsuffixes = N;
  c->stats.longest = c->h_stats->longest;
  c->stats.largelcpvalues = want_lcp ? c->h_stats->numlarge : 0;
  c->stats.maxbranchdepth =
      want_lcp ? (c->h_stats->maxlcp > c->h_stats->dmax ? c->h_stats->maxlcp
                                                         : c->h_stats->dmax) : 0;
  c->stats.lcptabsum = want_lcp ? c->h_stats->lcpsum + c->h_stats->dsum : 0;
  c->stats.prefixlength = prefixlength;
  c->stats.refine_rounds = rounds;
  c->stats.tied_suffixes = (m0_tied_all ? m0_tied_all : m0) + 2 * npairs + c->h_stats->smalldone;
  c->stats.pair_suffixes = 2 * npairs;
  c->stats.device_bytes = c->alloc_bytes;
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[6])); c->timing.total_ms = ms;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->timing.keygen_ms = ms;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->timing.sort_ms = ms;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); c->timing.finalize_ms = ms;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev[4], c->ev[5])); c->timing.refine_ms = ms;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev[5], c->ev[6])); c->timing.tie_fix_ms = ms;
  float sc = 0;
  for (int i = 0; i < nev; i++) {
    HIP_TRY(hipEventElapsedTime(&ms, c->ev_scatter[2 * i], c->ev_scatter[2 * i + 1]));
    sc += ms;
  }
  if (msd && msd_local > 0) {
    HIP_TRY(hipEventElapsedTime(&sc, c->ev_scatter[0], c->ev_scatter[1]));
    nev = 1;
  }
  c->timing.scatter_ms = sc;
  c->timing.scatter_launches = (u32) nev;
  c->timing.scatter_items = msd ? msd_local : NL;
  // what the dominant kernel moves per launch: k_msd_local reads every run that fits
  // its tile and writes the tables of those it sorts itself -- the runs with a crowded
  // bin it only reads (k_msd_local_radix sorts and writes them)
  c->timing.scatter_read_items = c->timing.scatter_items;
  c->timing.scatter_written_items = msd ? msd_local - c->h_stats->crowded : NL;
  c->stats.msd_crowded_entries = msd ? c->h_stats->crowded : 0;
  c->stats.rank_entries_built = rank_built;
  c->timing.dominant_kernel = msd ? 1u : 0u;
  c->timing.alloc_ms = c->alloc_ms;
  c->want = want;
  c->ran = true;
  return 0;
}

extern "C" int gtamd_esa_run(gtamd_esa_ctx *c, uint32_t want) {
  GTAMD_ABI_BEGIN
  if (c == nullptr) { gtamd_set_error("null context"); return -1; }
  // A part that fails between two collectives (an allocation, a launch) -- or before
  // the first -- has left the others waiting in the next one: the transport is told
  // (gtamd_esa_set_comm_abort; the library's thread transport registers itself), also
  // when an exception ends the run.
  struct AbortOnFailure {
    gtamd_esa_ctx *c;
    bool ok;
    ~AbortOnFailure() {
      if (!ok && c->numparts > 1 && c->comm_abort != nullptr) c->comm_abort(c->comm_abort_user);
    }
  } guard = {c, false};
  if (!c->have_text) { gtamd_set_error("no sequence set"); return -1; }
  if ((want & 15u) == 0) { gtamd_set_error("nothing requested"); return -1; }
  HIP_TRY(hipSetDevice(c->device));
  // 64-bit positions: whenever the sequence needs them; GTAMD_FORCE_WIDE=1
  // takes that path (and the exchange machinery of a part build) at any size
  const char *fw = getenv("GTAMD_FORCE_WIDE");
  const bool force_wide = fw != nullptr && fw[0] == '1';
  const bool wide = force_wide || c->N >= SINGLE_LIMIT;
  const bool dist = c->numparts > 1 || wide;
  if (c->numparts == 1 && c->N >= SINGLE_LIMIT) {
    gtamd_set_error("sequence of %llu symbols exceeds the 32-bit index range of a "
                    "single build: build it in parts (gtamd_esa_set_part)",
                    (unsigned long long) c->n);
    return -1;
  }
  int rc;
  if (c->bits == 2)
    rc = wide ? run_impl<2, true>(c, want, dist) : run_impl<2, false>(c, want, dist);
  else
    rc = wide ? run_impl<5, true>(c, want, dist) : run_impl<5, false>(c, want, dist);
  guard.ok = rc == 0;
  return rc;
  GTAMD_ABI_END(-1)
}

// ---------------------------------------------------------------------------
// outputs
// ---------------------------------------------------------------------------
extern "C" uint64_t gtamd_esa_table_entries(const gtamd_esa_ctx *c,
                                            gtamd_table which) {
  GTAMD_ABI_BEGIN
  if (c == nullptr || !c->ran) return 0;
  if (which == GTAMD_TAB_BCK)
    return (c->want & GTAMD_WANT_BCK) ? c->bck_codes + 1 + c->bck_special + c->bck_dist : 0;
  return which == GTAMD_TAB_LLV ? c->llv_pairs : c->NL;
  GTAMD_ABI_END(0)
}
extern "C" int gtamd_esa_bck_layout(const gtamd_esa_ctx *c, uint64_t *numofallcodes,
                                    uint64_t *numofspecialcodes,
                                    uint64_t *numofdistpfxidxcounters) {
  GTAMD_ABI_BEGIN
  if (c == nullptr || !c->ran || !(c->want & GTAMD_WANT_BCK)) {
    gtamd_set_error("no bucket table was requested");
    return -1;
  }
  *numofallcodes = c->bck_codes;
  *numofspecialcodes = c->bck_special;
  *numofdistpfxidxcounters = c->bck_dist;
  return 0;
  GTAMD_ABI_END(-1)
}
extern "C" uint64_t gtamd_esa_table_offset(const gtamd_esa_ctx *c) {
  GTAMD_ABI_BEGIN
  return (c == nullptr || !c->ran) ? 0 : c->index_offset;
  GTAMD_ABI_END(0)
}
extern "C" const void *gtamd_esa_table_device(const gtamd_esa_ctx *c,
                                              gtamd_table which) {
  GTAMD_ABI_BEGIN
  if (c == nullptr || !c->ran) return nullptr;
  switch (which) {
    case GTAMD_TAB_SUF: return (c->want & GTAMD_WANT_SUF) ? c->suf.p : nullptr;
    case GTAMD_TAB_LCP: return (c->want & GTAMD_WANT_LCP) ? c->lcp.p : nullptr;
    case GTAMD_TAB_BWT: return (c->want & GTAMD_WANT_BWT) ? c->bwt.p : nullptr;
    case GTAMD_TAB_LLV: return (c->want & GTAMD_WANT_LCP) ? c->llv : nullptr;
    case GTAMD_TAB_BCK: return (c->want & GTAMD_WANT_BCK) ? c->bck : nullptr;
  }
  return nullptr;
  GTAMD_ABI_END(nullptr)
}
extern "C" int gtamd_esa_table_copy(gtamd_esa_ctx *c, gtamd_table which,
                                    void *dst, uint64_t first, uint64_t count) {
  GTAMD_ABI_BEGIN
  if (c == nullptr) { gtamd_set_error("null context"); return -1; }
  const void *src = gtamd_esa_table_device(c, which);
  const u64 entries = gtamd_esa_table_entries(c, which);
  if (count == 0) return 0;
  if (src == nullptr || first + count > entries) {
    gtamd_set_error("table %d not available or range [%llu,+%llu) outside its "
                    "%llu entries", (int) which, (unsigned long long) first,
                    (unsigned long long) count, (unsigned long long) entries);
    return -1;
  }
  const u64 esz = which == GTAMD_TAB_SUF ? 8 : which == GTAMD_TAB_LLV ? 16
                : which == GTAMD_TAB_BCK ? 4 : 1;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy(dst, (const u8 *) src + first * esz, count * esz,
                    hipMemcpyDeviceToHost));
  return 0;
  GTAMD_ABI_END(-1)
}
// for the consumers inside the library (esa_pck.hip): where the tables live
extern "C" int gtamd_esa_internal_info(const gtamd_esa_ctx *c, int *device, u32 *sigma,
                                       u32 *numparts) {
  GTAMD_ABI_BEGIN
  if (c == nullptr || !c->ran) { gtamd_set_error("no completed run"); return -1; }
  *device = c->device; *sigma = c->sigma; *numparts = c->numparts;
  return 0;
  GTAMD_ABI_END(-1)
}
extern "C" int gtamd_esa_get_stats(const gtamd_esa_ctx *c, gtamd_esa_stats *s) {
  GTAMD_ABI_BEGIN
  if (c == nullptr || !c->ran) { gtamd_set_error("no completed run"); return -1; }
  *s = c->stats;
  return 0;
  GTAMD_ABI_END(-1)
}
extern "C" int gtamd_esa_get_timing(const gtamd_esa_ctx *c, gtamd_esa_timing *t) {
  GTAMD_ABI_BEGIN
  if (c == nullptr || !c->ran) { gtamd_set_error("no completed run"); return -1; }
  *t = c->timing;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_esa_build(const uint8_t *enc, uint64_t n,
                               uint32_t numofchars, uint32_t want,
                               uint64_t *suf, uint8_t *lcp, uint8_t *bwt,
                               uint64_t *llv, uint64_t llv_capacity,
                               uint64_t *llv_pairs, gtamd_esa_stats *st) {
  GTAMD_ABI_BEGIN
  gtamd_esa_ctx *c = gtamd_esa_create(0, n, numofchars);
  if (c == nullptr) return -1;
  int rc = gtamd_esa_set_sequence_bytes(c, enc, n, 0);
  if (rc == 0) rc = gtamd_esa_run(c, want);
  if (rc == 0 && suf != nullptr && (want & GTAMD_WANT_SUF))
    rc = gtamd_esa_table_copy(c, GTAMD_TAB_SUF, suf, 0, n + 1);
  if (rc == 0 && lcp != nullptr && (want & GTAMD_WANT_LCP))
    rc = gtamd_esa_table_copy(c, GTAMD_TAB_LCP, lcp, 0, n + 1);
  if (rc == 0 && bwt != nullptr && (want & GTAMD_WANT_BWT))
    rc = gtamd_esa_table_copy(c, GTAMD_TAB_BWT, bwt, 0, n + 1);
  if (rc == 0 && (want & GTAMD_WANT_LCP)) {
    const u64 pairs = gtamd_esa_table_entries(c, GTAMD_TAB_LLV);
    if (llv_pairs != nullptr) *llv_pairs = pairs;
    if (llv != nullptr) {
      if (pairs > llv_capacity) {
        gtamd_set_error(".llv needs %llu pairs, caller provided room for %llu",
                        (unsigned long long) pairs,
                        (unsigned long long) llv_capacity);
        rc = -1;
      } else
        rc = gtamd_esa_table_copy(c, GTAMD_TAB_LLV, llv, 0, pairs);
    }
  }
  if (rc == 0 && st != nullptr) rc = gtamd_esa_get_stats(c, st);
  gtamd_esa_destroy(c);
  return rc;
  GTAMD_ABI_END(-1)
}
