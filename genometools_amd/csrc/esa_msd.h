// esa_msd.h -- the first sort of a big DNA build, most significant digit first,
// with entries that shrink as digits are used up, and the table emission done
// by the kernel that finishes the sort.  Included by esa_engine.hip (kernels
// only; the driver is msd_sort_emit there).
//
// Why: the LSD sort moves the whole 12-byte (key, position) pair through HBM
// once per digit -- keygen 36 GB + 5 x (24 GB histogram read + 72 GB scatter)
// at 3 Gbp, then 66 GB more for tie bits and table emission: 580 GB, 145 ms of
// a 204 ms build.  Most significant digit first, a digit that has been used is
// implied by where the entry lies, so it need not be carried:
//
//   level A  (fused with the keygen, text order)   digit = symbols 0..3
//            writes  K1 = symbols 4..19 (u32), X = dcode | payload (u8), P (u32)  27 GB
//   level B  inside each of the 256 ranges         digit = symbols 4..7
//            writes  K2 = symbols 8..19 | X (u32), P                             12 + 27 + 24 GB
//   level C  inside each of the 65536 ranges       digit = the next `cbits` bits
//            writes  K2, P                                                       12 + 24 + 24 GB
//   level D  one workgroup per run of whole ranges (<= 4096 entries): stable
//            LSD sort of the rest of K2 in LDS, then .suf/.lcp/.bwt, the tie
//            bitmap and the statistics straight from LDS                         24 + 42 GB
//
// 216 GB.  Levels A-C are stable partitions (per-tile histogram, column scan,
// ballot-ranked scatter: the LSD sort's machinery), so entries with equal keys
// stay in text order, which the order of suffixes that run into a special
// relies on (esa_common.h).  The levels below A work on ragged tiles: a tile
// never straddles two parent ranges, a parent range of s entries has
// ceil(s / 4096) tiles, the last one partial.
//
// Runs of ranges that exceed the LDS tile (repeats: one 12-mer with 10^5
// occurrences) are sorted by one workgroup each in global memory
// (k_msd_big), ping-pong between the two buffer pairs of the levels; runs above
// MSD_BIG_MAX entries (the suffixes that start with a wildcard all have one
// key; poly-A; two-letter texts) by the device-wide LSD sort on the 29 bits that
// are left, run by run, followed by k_msd_emit_run.
#pragma once

constexpr int MS_TILE = 4096;
constexpr int MS_THREADS = 512;
constexpr int MS_WAVES = MS_THREADS / 64;
constexpr int MS_ITEMS = MS_TILE / MS_THREADS;     // 8
constexpr int MS_WCHUNK = MS_ITEMS * 64;           // 512 consecutive entries per wave
constexpr int MS_PAD = 520;                        // row pitch of the keygen transposition
constexpr u32 MSD_PARENTS = 65536;                 // ranges after levels A and B
constexpr u32 MSD_STRIDE = 3328;                   // level D: entries per tile before snapping
constexpr int MD_ITEMS = 8;                        // level D: entries per thread, at most
constexpr u32 MD_CAP = MD_ITEMS * MS_THREADS;      // largest run of the LDS kernels
constexpr u32 MSD_BIG_MAX = 1u << 19;              // largest run one workgroup sorts alone

struct MsTile {       // ragged tile of a level pass
  u32 start;          // index of its first entry
  u32 segvalid;       // parent range << 13 | entries (0 .. 4096)
};
struct MdTile {       // level D: a run of whole finest-level ranges
  u32 begin, end;     // entries [begin, end)
  u32 s16;            // the 16 key bits levels A and B have used
  u32 pad;            // level-C digit of the first range | number of ranges << 16
};
struct MsdOut {
  u64 *suf;
  u8 *lcp;
  u8 *bwt;
  u32 *sa;            // positions in suffix order, 32-bit (what the refinement reads)
  u64 *tiebits;       // zeroed by the driver
  u64 *firstkey, *lastkey;   // per level-D tile: its first and last key, for the seams
  Stats *stats;
  u32 prefixlength;
  u64 index_offset;   // part builds: index of the slice's first entry in the whole table
  u32 val_is_index;   // part builds with 64-bit positions: the sort's value is the entry's
                      // number in text order among the part's suffixes, not its position
                      // (k_part_positions turns it into one afterwards and writes .suf)
};
struct MsdAcc {
  unsigned long long sum, ties;
  u32 mx;
};

__device__ __forceinline__ u32 ms_xcd_tile(u32 b, u32 ntiles) {
  const u32 per = (ntiles + 7u) >> 3;
  return (b & 7u) * per + (b >> 3);
}

// ---------------------------------------------------------------------------
// Two entry formats go through the same levels (FMT, a template parameter of the
// kernels that look INTO the keys: level A and everything of level D):
//
// FMT 0, the 2-bit alphabet: the sorted part of Key<2> -- 20 symbols x 2 bit and
//   the 5-bit dcode -- cut as 8 | 8 | 24 + dcode; X = dcode << 3 | payload (3 bits).
// FMT 1, the 5-bit alphabets (protein): a 40-bit CODE of the suffix' first NINE
//   symbols made for this sort: four pairs of symbols as 9-bit numbers 21 a + b
//   (a symbol is a letter 0..19 or 20 = "behind the first special": the padding,
//   larger than every letter and never a letter, so no dcode is needed to tell
//   padded prefixes apart) and the 9th symbol's upper four bits (s >> 1: 0..10,
//   monotone in s; 10 only for the padding).  Cut as 8 | 8 | 24; X = payload (5
//   bits), K2 = low 24 code bits << 8 | X.  The 50 + 4 sorted bits of Key<5> do not
//   fit the 32-bit word of levels C and D; 8 symbols + the class of the ninth do,
//   and separate all but ~2 % of 10^9 residues with Swiss-Prot frequencies (eight
//   symbols alone: 13 %).  Suffixes the sort leaves tied share MSD_KSYMS<FMT>
//   symbols for sure: 20, or 8.
// ---------------------------------------------------------------------------
template <int FMT> struct MsdFmt;
template <> struct MsdFmt<0> {
  static constexpr int PB = 3;          // K2 bits below the sorted ones
  static constexpr u32 KSYMS = 20;
};
template <> struct MsdFmt<1> {
  static constexpr int PB = 8;
  static constexpr u32 KSYMS = 8;
};

// what the seams keep of an entry: FMT 0 the 64-bit key (layout of Key<2>) from
// what levels A-C keep of it; FMT 1 code << 8 | X
template <int FMT>
__device__ __forceinline__ u64 msd_full(u32 s16, u32 k2) {
  if (FMT == 1) return ((u64) s16 << 32) | (u64) k2;
  return ((u64) s16 << 48) | ((u64) (k2 >> 8) << 24) | ((u64) ((k2 >> 3) & 31u) << 19) |
         (u64) (k2 & 7u);
}

// ---- FMT 1: the code of the 5-bit alphabets
__device__ __forceinline__ u32 p5_first(u32 pair) { return (pair * 3121u) >> 16; }   // pair / 21, pair < 441
// letters in front of the first special, as far as the code shows them: 0..8, or 9
// for "nine or more"
__device__ __forceinline__ u32 p5_letters(u64 code) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const u32 pr = (u32) (code >> (31 - 9 * k)) & 511u;
    const u32 a = p5_first(pr);
    if (a == 20u) return 2u * k;
    if (pr - 21u * a == 20u) return 2u * k + 1u;
  }
  return ((u32) code & 15u) == 10u ? 8u : 9u;
}
// symbols two codes share (at most 8: the ninth is known by its class only)
__device__ __forceinline__ u32 p5_common(u64 a, u64 b) {
  const u64 x = a ^ b;
  if ((x >> 4) == 0) return 8u;
  const int lz = __clzll((long long) x) - 24;          // 0 .. 35: inside pair lz / 9
  const int f = lz / 9;
  const u32 pa = (u32) (a >> (31 - 9 * f)) & 511u, pb = (u32) (b >> (31 - 9 * f)) & 511u;
  return 2u * (u32) f + (p5_first(pa) == p5_first(pb) ? 1u : 0u);
}
// code and payload of suffix p (Key<5>'s padding rule: everything from the first
// special on counts as the largest symbol; a suffix that starts with a special
// has the largest code)
// (pay: the payload, and bit 5 = "the code shows a padding": fewer than nine letters)
__device__ __forceinline__ u64 p5_code(const Text &t, u64 p, u32 &pay) {
  pay = Pay<5>::before(t, p);
  const u64 win = Sym<5>::window(t, p);                 // symbols p .. p+10 in the top bits
  const u32 sp = (u32) sp_window(t, p) & 0x1FFu;
  const int d = sp ? __ffs((int) sp) - 1 : 9;
  if (sp) pay |= 32u;
  u32 sym[9];
#pragma unroll
  for (int i = 0; i < 9; i++) sym[i] = i < d ? (u32) (win >> (59 - 5 * i)) & 31u : 20u;
  u64 code = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) code = (code << 9) | (u64) (sym[2 * k] * 21u + sym[2 * k + 1]);
  return (code << 4) | (u64) (sym[8] >> 1);
}
// the same for the suffixes p0 .. p0+7 (p0 a multiple of 8) as "keys" code << 24 |
// X: away from specials the 16 symbols are taken from two windows and every pair
// of neighbours is made once for the four codes it is part of
__device__ __forceinline__ void p5_keys8(const Text &t, u64 p0, u64 (&key)[KP_PER]) {
  const u64 S = sp_window(t, p0);
  // (one division for the three places: p0, p0 + 8 and the symbol in front)
  const u64 w0 = p0 / 12;
  const int r0 = (int) (p0 - w0 * 12);
  const u64 t0 = tb_word(t, w0), t1 = tb_word(t, w0 + 1), t2 = tb_word(t, w0 + 2);
  const u64 wa = r0 <= 1 ? t0 << (5 * r0) : (t0 << (5 * r0)) | (t1 >> (5 * (12 - r0)));
  const int r8 = r0 + 8 >= 12 ? r0 - 4 : r0 + 8;
  const u64 ta = r0 + 8 >= 12 ? t1 : t0, tb = r0 + 8 >= 12 ? t2 : t1;
  const u64 wb = r8 <= 1 ? ta << (5 * r8) : (ta << (5 * r8)) | (tb >> (5 * (12 - r8)));
  // the 16 stored symbols p0 .. p0+15 (a special is stored as 0 or 1; behind the end
  // of the text whatever the words hold: the bit of position n is set in S, nothing
  // behind the first special of a suffix is looked at)
  u32 sym[16], pr[15];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    sym[i] = (u32) (wa >> (59 - 5 * i)) & 31u;
    sym[8 + i] = (u32) (wb >> (59 - 5 * i)) & 31u;
  }
#pragma unroll
  for (int i = 0; i < 15; i++) pr[i] = sym[i] * 21u + sym[i + 1];
  u32 pay = Pay<5>::UNDEF;
  if (p0 > 0) {
    const u32 c = r0 ? (u32) (t0 >> (59 - 5 * (r0 - 1))) & 31u : (u32) (tb_word(t, w0 - 1) >> 4) & 31u;
    pay = (c < 2u && is_special(t, p0 - 1)) ? ((c & 1u) ? Pay<5>::SEP : Pay<5>::WILD) : c;
  }
  const bool clean = (S & 0xFFFFull) == 0;       // no special among the 16 positions
#pragma unroll
  for (int g = 0; g < KP_PER; g++) {
    u64 code = ((u64) pr[g] << 31) | ((u64) pr[g + 2] << 22) | ((u64) pr[g + 4] << 13) |
               ((u64) pr[g + 6] << 4) | (u64) (sym[g + 8] >> 1);
    u32 x = pay;
    const u32 sp = (u32) (S >> g) & 0x1FFu;      // specials among the suffix' first nine
    if (!clean && sp != 0) {
      // (p5_code: everything from the first special on is the padding symbol 20.  One
      // lane in twenty of a protein set comes here, so nearly every wave does: the
      // symbols are the ones at hand, no window is read again)
      const int d = __ffs((int) sp) - 1;
      u32 q[9];
#pragma unroll
      for (int i = 0; i < 9; i++) q[i] = i < d ? sym[g + i] : 20u;
      code = ((u64) (q[0] * 21u + q[1]) << 31) | ((u64) (q[2] * 21u + q[3]) << 22) |
             ((u64) (q[4] * 21u + q[5]) << 13) | ((u64) (q[6] * 21u + q[7]) << 4) | (u64) (q[8] >> 1);
      x |= 32u;
    }
    key[g] = (code << 24) | (u64) x;
    // the payload of the next suffix: this symbol, or which special it is
    pay = (!clean && ((S >> g) & 1ull)) ? ((sym[g] & 1u) ? Pay<5>::SEP : Pay<5>::WILD) : sym[g];
  }
}

// keys of the suffixes p0 .. p0+7 (p0 a multiple of 8), bit for bit those of
// make_key<2>; the arithmetic of k_keygen_pass0_dna
__device__ __forceinline__ void dna_keys8(const Text &t, u64 p0, u64 (&key)[KP_PER]) {
  using K = Key<2>;
  using P = Pay<2>;
  constexpr int SYMS = K::SYMS;
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
  if (p0 == 0) {
    pay = P::UNDEF;
  } else {
    const u32 c = o ? (u32) (hi >> (64 - o)) & 3u : (u32) tb_word(t, w - 1) & 3u;
    const bool sp = c < 2u && (so ? (s0 >> (so - 1)) & 1ull : sp_word(t, sw - 1) >> 63);
    pay = sp ? ((c & 1u) ? P::SEP : P::WILD) : c;
  }
  // no special among the KP_PER + SYMS - 1 positions the keys look at (all but
  // the stretches around wildcard runs and sequence ends): a key is its window
  // and the symbol in front of it
  if ((S & ((1ull << (KP_PER + SYMS - 1)) - 1ull)) == 0) {
#pragma unroll
    for (int g = 0; g < KP_PER; g++) {
      const u64 win = g ? (a_hi << (2 * g)) | (a_lo >> (64 - 2 * g)) : a_hi;
      key[g] = (win & (~0ull << K::LOW_BITS)) | pay;
      pay = (u32) (win >> 62);
    }
    return;
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
    const u32 c = (u32) (win >> 62);
    pay = (c < 2u && (s & 1ull)) ? ((c & 1u) ? P::SEP : P::WILD) : c;
  }
}

// lanes of the wave that hold the same NB-bit digit, as a count of those below
// this lane and the size of the group (the ranking of esa_prims.hip's scatter
// kernel: one bit-field extract, one compare and one three-input bit operation
// per half and bit)
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

// barrier for kernels whose threads share LDS only: waits for this wave's LDS
// operations, not for its global loads and stores (__syncthreads() makes the
// stores of the table emission and the loads fetched ahead for the next run
// complete first: ~2 us per barrier in a kernel that has a dozen per run)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// block_scan_excl of esa_devutil.h (sum, 512 threads) on that barrier
__device__ __forceinline__ u32 ms_scan_excl(u32 v, u32 *lds8) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u32 inc = wave_scan_incl<SCAN_SUM>(v);
  if (lane == 63) lds8[w] = inc;
  lds_barrier();
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < MS_WAVES; i++) {
    const u32 x = lds8[i];
    if (i < w) carry += x;
  }
  lds_barrier();
  return carry + (inc - v);
}

// ---------------------------------------------------------------------------
// level A: keygen + partition on the first four symbols
// ---------------------------------------------------------------------------
// (5-bit alphabets: the same over the codes of FMT 1)
__global__ __launch_bounds__(MS_THREADS) void k_msd_hist_a5(Text t, u64 N, u32 *__restrict__ hist) {
  __shared__ u32 h[MS_WAVES][256];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < MS_WAVES * 256; i += MS_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const u64 p0 = (u64) blockIdx.x * MS_TILE + (u64) tid * KP_PER;
  if (p0 < N) {
    const int npos = N - p0 < KP_PER ? (int) (N - p0) : KP_PER;
    u64 key[KP_PER];
    p5_keys8(t, p0, key);
#pragma unroll
    for (int g = 0; g < KP_PER; g++)
      if (g < npos) atomicAdd(&h[w][(u32) (key[g] >> 56)], 1u);
  }
  __syncthreads();
  if (tid < 256) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < MS_WAVES; i++) c += h[i][tid];
    hist[(u64) blockIdx.x * 256 + tid] = c;
  }
}

__global__ __launch_bounds__(MS_THREADS) void k_msd_hist_a(Text t, u64 N,
                                                           u32 *__restrict__ hist) {
  __shared__ u32 h[MS_WAVES][256];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < MS_WAVES * 256; i += MS_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const u64 p0 = (u64) blockIdx.x * MS_TILE + (u64) tid * KP_PER;
  if (p0 < N) {
    const int npos = N - p0 < KP_PER ? (int) (N - p0) : KP_PER;
    u64 key[KP_PER];
    dna_keys8(t, p0, key);
#pragma unroll
    for (int g = 0; g < KP_PER; g++)
      if (g < npos) atomicAdd(&h[w][(u32) (key[g] >> 56)], 1u);
  }
  __syncthreads();
  if (tid < 256) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < MS_WAVES; i++) c += h[i][tid];
    hist[(u64) blockIdx.x * 256 + tid] = c;
  }
}

// ---------------------------------------------------------------------------
// part builds: the suffixes of this part's key range, taken from
// the replicated text by a scan -- no (key, position) pair leaves the device
// ---------------------------------------------------------------------------
// A part owns the suffixes whose PART_BITS leading key bits lie in [lo, hi).
// Both passes over the text look at those bits only (seven symbols and the
// special bits of seven positions: a few instructions per suffix where the whole
// key takes ~40); the whole key is made for the suffixes the part keeps.
struct DnaWin {       // the text around the suffixes p0 .. p0+7 (p0 a multiple of 8)
  u64 a_hi, a_lo;     // symbols p0 .. p0+31 and what the word pair holds behind them
  u64 S;              // bit j: position p0 + j is special
  u32 pay0;           // payload (symbol in front) of suffix p0
};
__device__ __forceinline__ DnaWin dna_win(const Text &t, u64 p0) {
  using P = Pay<2>;
  DnaWin r;
  const u64 w = p0 >> 5;
  const int o = (int) (p0 & 31) * 2;
  const u64 hi = tb_word(t, w), lo = tb_word(t, w + 1);
  r.a_hi = o ? (hi << o) | (lo >> (64 - o)) : hi;
  r.a_lo = lo << o;
  const u64 sw = p0 >> 6;
  const int so = (int) (p0 & 63);
  const u64 s0 = sp_word(t, sw), s1 = sp_word(t, sw + 1);
  r.S = so ? (s0 >> so) | (s1 << (64 - so)) : s0;
  if (p0 == 0) {
    r.pay0 = P::UNDEF;
  } else {
    const u32 c = o ? (u32) (hi >> (64 - o)) & 3u : (u32) tb_word(t, w - 1) & 3u;
    const bool sp = c < 2u && (so ? (s0 >> (so - 1)) & 1ull : sp_word(t, sw - 1) >> 63);
    r.pay0 = sp ? ((c & 1u) ? P::SEP : P::WILD) : c;
  }
  return r;
}
// the same from the four words a thread has loaded for its bins
__device__ __forceinline__ DnaWin dna_win_from(const Text &t, u64 p0, u64 hi, u64 lo, u64 s0, u64 s1) {
  using P = Pay<2>;
  DnaWin r;
  const u64 w = p0 >> 5;
  const int o = (int) (p0 & 31) * 2;
  r.a_hi = o ? (hi << o) | (lo >> (64 - o)) : hi;
  r.a_lo = lo << o;
  const u64 sw = p0 >> 6;
  const int so = (int) (p0 & 63);
  r.S = so ? (s0 >> so) | (s1 << (64 - so)) : s0;
  if (p0 == 0) {
    r.pay0 = P::UNDEF;
  } else {
    const u32 c = o ? (u32) (hi >> (64 - o)) & 3u : (u32) tb_word(t, w - 1) & 3u;
    const bool sp = c < 2u && (so ? (s0 >> (so - 1)) & 1ull : sp_word(t, sw - 1) >> 63);
    r.pay0 = sp ? ((c & 1u) ? P::SEP : P::WILD) : c;
  }
  return r;
}
// what the bins of p0 .. p0+7 need of it (no payload: two loads less)
__device__ __forceinline__ void dna_win_bins(const Text &t, u64 p0, u64 &a_hi, u32 &s14) {
  const u64 w = p0 >> 5;
  const int o = (int) (p0 & 31) * 2;
  const u64 hi = tb_word(t, w);
  a_hi = o ? (hi << o) | (tb_word(t, w + 1) >> (64 - o)) : hi;
  const u64 sw = p0 >> 6;
  const int so = (int) (p0 & 63);      // a multiple of 8
  u64 S = sp_word(t, sw) >> so;
  if (so > 64 - 14) S |= sp_word(t, sw + 1) << (64 - so);
  s14 = (u32) S & 0x3FFFu;             // special bits of p0 .. p0+13
}
// bit g: the key bin of suffix p0 + g lies in [lo, lo + width); also the number of
// suffixes with a bin below lo and the largest such bin
__device__ __forceinline__ u32 dna_bins8_in_range(u64 a_hi, u32 s14, int npos, u32 lo, u32 width,
                                                  u32 &nbelow, u32 &mxbin) {
  u32 mask = 0;
  if (s14 == 0 && npos == KP_PER) {
#pragma unroll
    for (int g = 0; g < KP_PER; g++) {
      const u32 bin = (u32) ((a_hi << (2 * g)) >> (64 - PART_BITS));
      if (bin - lo < width) mask |= 1u << g;
      else if (bin < lo) { nbelow++; mxbin = bin > mxbin ? bin : mxbin; }
    }
    return mask;
  }
#pragma unroll
  for (int g = 0; g < KP_PER; g++) {
    if (g < npos) {
      u32 bin = (u32) ((a_hi << (2 * g)) >> (64 - PART_BITS));
      const u32 s7 = (s14 >> g) & 0x7Fu;
      if (s7) {
        const int d = __ffs((int) s7) - 1;
        bin = d == 0 ? (u32) PART_BINS - 1u : bin | ((1u << (2 * (7 - d))) - 1u);
      }
      if (bin - lo < width) mask |= 1u << g;
      else if (bin < lo) { nbelow++; mxbin = bin > mxbin ? bin : mxbin; }
    }
  }
  return mask;
}
// the PART_BITS leading bits of the key of suffix p0 + g: seven symbols, padded
// with 1-bits behind a special (make_key<2>; key_bin<2>)
__device__ __forceinline__ u32 dna_bin(const DnaWin &w, int g) {
  static_assert(PART_BITS == 14 && KP_PER + 7 <= 32, "seven symbols of the first word");
  u32 b = (u32) ((w.a_hi << (2 * g)) >> (64 - PART_BITS));
  const u32 s7 = (u32) (w.S >> g) & 0x7Fu;
  if (s7) {
    const int d = __ffs((int) s7) - 1;
    b = d == 0 ? (u32) PART_BINS - 1u : b | ((1u << (2 * (7 - d))) - 1u);
  }
  return b;
}
// the key of suffix p0 + g, bit for bit that of make_key<2> / dna_keys8
__device__ __forceinline__ u64 dna_key_at(const DnaWin &w, int g) {
  using K = Key<2>;
  using P = Pay<2>;
  constexpr int SYMS = K::SYMS;
  u32 pay = w.pay0;
  if (g > 0) {
    const u32 c = (u32) (w.a_hi >> (64 - 2 * g)) & 3u;
    pay = (c < 2u && ((w.S >> (g - 1)) & 1ull)) ? ((c & 1u) ? P::SEP : P::WILD) : c;
  }
  const u64 win = g ? (w.a_hi << (2 * g)) | (w.a_lo >> (64 - 2 * g)) : w.a_hi;
  const u64 s = (w.S >> g) & ((1ull << SYMS) - 1ull);
  const int d = s ? __ffsll((unsigned long long) s) - 1 : SYMS;
  if (d == 0) return (~0ull << K::DSHIFT) | pay;
  u64 pre = win >> K::LOW_BITS;
  u32 dc = 0;
  if (d < SYMS) {
    pre |= (1ull << (2 * (SYMS - d))) - 1ull;
    dc = (u32) (SYMS - d);
  }
  return (pre << K::LOW_BITS) | ((u64) dc << K::DSHIFT) | pay;
}

// k_part_count: per text tile the number of suffixes the part keeps (tkeep, zeroed
// by the host; their scan is where the tile's keys go), and over the whole text
// the number of suffixes BELOW the range (the slice's offset in the table) and the
// largest key bin among them.  acc: [0] suffixes below, [1] largest bin below, [2]
// suffixes kept.  No barrier per tile: a wave adds its count to the tile's.
__global__ __launch_bounds__(MS_THREADS) void k_part_count(
    Text t, u64 N, u32 ntiles, u32 lo, u32 hi, u32 *__restrict__ tkeep,
    unsigned long long *__restrict__ acc) {
  __shared__ unsigned long long s_below[MS_WAVES], s_kept[MS_WAVES];
  __shared__ u32 s_max[MS_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned long long below = 0;
  u32 kept = 0, mxbin = 0;
  for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const u64 p0 = (u64) tile * MS_TILE + (u64) tid * KP_PER;
    u32 k = 0;
    if (p0 < N) {
      const int npos = N - p0 < KP_PER ? (int) (N - p0) : KP_PER;
      u64 a_hi;
      u32 s14, nb = 0;
      dna_win_bins(t, p0, a_hi, s14);
      k = (u32) __popc(dna_bins8_in_range(a_hi, s14, npos, lo, hi - lo, nb, mxbin));
      below += nb;
    }
    kept += k;
    const u32 wsum = wave_scan_incl<SCAN_SUM>(k);
    if (lane == 63 && wsum) atomicAdd(&tkeep[tile], wsum);
  }
  unsigned long long kept64 = kept;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    below += __shfl_xor(below, d, 64);
    kept64 += __shfl_xor(kept64, d, 64);
    const u32 o = __shfl_xor(mxbin, d, 64);
    mxbin = o > mxbin ? o : mxbin;
  }
  if (lane == 0) { s_below[w] = below; s_kept[w] = kept64; s_max[w] = mxbin; }
  __syncthreads();
  if (tid == 0) {
    unsigned long long B = 0, K = 0;
    u32 M = 0;
    for (int i = 0; i < MS_WAVES; i++) { B += s_below[i]; K += s_kept[i]; M = s_max[i] > M ? s_max[i] : M; }
    if (B) atomicAdd(&acc[0], B);
    if (M) atomicMax(&acc[1], (unsigned long long) M);
    if (K) atomicAdd(&acc[2], K);
  }
}

// k_part_filter: the keys of the suffixes the part keeps, and their positions, in
// text order (toff: exclusive scan of k_part_count's tile counts), one workgroup
// per tile.  PV: u32 positions (n < 2^32) or u64.  Beside it the largest key of
// bin `binbelow` -- the largest bin below the range that holds a suffix -- which
// is the neighbour of the slice's first entry: its LCP needs nothing else, two
// keys of different ranges differ inside their PART_BITS leading bits.
// (binbelow >= PART_BINS: none.)
// (What the kernel is bound by, measured at 3 Gbp / 8 parts: the bins and the scan
// 1.9 ms; the keys of the kept suffixes made one by one behind the compaction, with
// their six loads each, +2.5 ms; made where the bins are looked at with a function
// per key, every thread walks through eight of them for the one it keeps, +3 ms;
// stores of single keys, 12 bytes to a line, +1 ms.  So: all eight keys of a thread
// at once -- dna_keys8 is a few instructions per key away from specials --, the kept
// ones to LDS, whole lines out.)
template <typename PV>
__global__ __launch_bounds__(MS_THREADS) void k_part_filter(
    Text t, u64 N, u32 lo, u32 hi, u32 binbelow, const u32 *__restrict__ toff,
    u64 *__restrict__ ck, PV *__restrict__ cp, unsigned long long *__restrict__ prevkey) {
  __shared__ u64 s_k[MS_TILE];       // the kept keys of the tile, in text order
  __shared__ u16 s_off[MS_TILE];     // and where in the tile their suffixes start
  __shared__ u32 s_scan[MS_WAVES];
  const int tid = threadIdx.x;
  const u64 tile_base = (u64) blockIdx.x * MS_TILE;
  const u64 p0 = tile_base + (u64) tid * KP_PER;
  u64 key[KP_PER];
  u32 mask = 0;
  unsigned long long mx = 0;
  if (p0 < N) {
    const int npos = N - p0 < KP_PER ? (int) (N - p0) : KP_PER;
    dna_keys8(t, p0, key);
#pragma unroll
    for (int g = 0; g < KP_PER; g++) {
      const u32 bin = (u32) (key[g] >> (64 - PART_BITS));
      if (g < npos) {
        if (bin - lo < hi - lo) mask |= 1u << g;
        else if (bin == binbelow) mx = key[g] > mx ? key[g] : mx;
      }
    }
  }
  if (mx) atomicMax(prevkey, mx);    // (N / 2^14 suffixes of the text at most)
  u32 all;
  u32 at = block_scan_excl<SCAN_SUM, MS_THREADS>((u32) __popc(mask), &all, s_scan);
#pragma unroll
  for (int g = 0; g < KP_PER; g++) {
    if ((mask >> g) & 1u) {
      s_k[at] = key[g];
      s_off[at] = (u16) (tid * KP_PER + g);
      at++;
    }
  }
  __syncthreads();
  const u64 base = toff[blockIdx.x];
  for (u32 i = tid; i < all; i += MS_THREADS) {
    ck[base + i] = s_k[i];
    cp[base + i] = (PV) (tile_base + (u64) s_off[i]);
  }
}

// level A's histogram over a tile of the kept keys
__global__ __launch_bounds__(MS_THREADS) void k_msd_hist_a_keys(const u64 *__restrict__ ck, u64 M,
                                                                u32 *__restrict__ hist) {
  __shared__ u32 h[MS_WAVES][256];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < MS_WAVES * 256; i += MS_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const u64 base = (u64) blockIdx.x * MS_TILE;
#pragma unroll
  for (int j = 0; j < MS_ITEMS; j++) {
    const u64 e = base + (u64) j * MS_THREADS + tid;
    if (e < M) atomicAdd(&h[w][(u32) (ck[e] >> 56)], 1u);
  }
  __syncthreads();
  if (tid < 256) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < MS_WAVES; i++) c += h[i][tid];
    hist[(u64) blockIdx.x * 256 + tid] = c;
  }
}

// positions of a part build with 64-bit positions: the sort carried the entry's
// number in text order (val_is_index); its position is cp[that number].  Also .suf
// and the index of suffix 0.
__global__ __launch_bounds__(256) void k_part_positions(const u32 *__restrict__ idx,
                                                        const u64 *__restrict__ cp, u64 M,
                                                        u64 index_offset, u64 *__restrict__ sa64,
                                                        u64 *__restrict__ suf, Stats *stats) {
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const u64 p = cp[idx[i]];
  sa64[i] = p;
  if (suf != nullptr) suf[i] = p;
  if (p == 0) stats->longest = index_offset + i;
}

// starts of the 256 level-A ranges: row 0 of the scanned histogram
__global__ void k_msd_starts_a(const u32 *__restrict__ scanned, u32 N, u32 *__restrict__ start) {
  const u32 d = threadIdx.x;
  if (d < 256) start[d] = scanned[d];
  if (d == 0) start[256] = N;
}

// SRC 0: all suffixes of a 2-bit text; SRC 1 (FROMKEYS): the entries are the keys a
// part build has filtered from the text (ck, their values cp32 or -- nullptr --
// their numbers); SRC 2: all suffixes of a 5-bit text, by their FMT 1 codes (as
// "keys" code << 24 | payload: level-A digit and K1 lie where Key<2> has them)
template <int SRC>
__global__ __launch_bounds__(MS_THREADS) void k_msd_scatter_a(
    Text t, u64 N, u32 last_valid, const u32 *__restrict__ scanned, u32 ntiles,
    const u64 *__restrict__ ck, const u32 *__restrict__ cp32,
    u32 *__restrict__ k1out, u8 *__restrict__ xout, u32 *__restrict__ pout) {
  // 40 KB: first the keys in (suffix of the thread, thread) order, padded
  // against bank conflicts; then the staging area in digit order
  __shared__ u64 s_t[5120];
  __shared__ u16 s_cnt_mem[MS_WAVES * 256];
  __shared__ u32 s_obase[256];
  __shared__ u32 s_scan[MS_WAVES];
  static_assert(KP_PER * MS_PAD <= 5120 && MS_TILE * 10 <= 5120 * 8, "staging fits");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile = ms_xcd_tile(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  const u64 tile_base = (u64) tile * MS_TILE;
  const u32 valid = tile + 1u == ntiles ? last_valid : (u32) MS_TILE;
  for (int i = tid; i < MS_WAVES * 256 / 2; i += MS_THREADS)
    reinterpret_cast<u32 *>(s_cnt_mem)[i] = 0;
  u32 gbase = 0;
  if (tid < 256) gbase = scanned[(u64) tile * 256 + tid];
  constexpr bool FROMKEYS = SRC == 1;
  u64 key[MS_ITEMS];
  u32 rk[MS_ITEMS], val[MS_ITEMS];
  if (FROMKEYS) {
    // (already in the (wave, item, lane) order of the ranking: no transposition)
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) {
      const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
      key[j] = e < valid ? ck[tile_base + e] : ~0ull;
      val[j] = (e < valid && cp32 != nullptr) ? cp32[tile_base + e] : (u32) tile_base + e;
    }
  } else {
  {
    u64 key8[KP_PER];
    const u64 p0 = tile_base + (u64) tid * KP_PER;
    if (p0 < N) {
      if (SRC == 2) p5_keys8(t, p0, key8);
      else dna_keys8(t, p0, key8);
    } else {
#pragma unroll
      for (int g = 0; g < KP_PER; g++) key8[g] = ~0ull;
    }
#pragma unroll
    for (int g = 0; g < KP_PER; g++) s_t[g * MS_PAD + tid] = key8[g];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < MS_ITEMS; j++) {
    const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;   // suffix tile_base + e
    key[j] = s_t[(e & 7u) * MS_PAD + (e >> 3)];
  }
  }
  __syncthreads();   // the transposition area is free (the counters are zero)
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
      for (int i = 0; i < MS_WAVES; i++) {
        c[i] = s_cnt[i * 256 + tid];
        tot += c[i];
      }
    }
    u32 all;
    u32 dbase = block_scan_excl<SCAN_SUM, MS_THREADS>(tot, &all, s_scan);
    if (tid < 256) {
      s_obase[tid] = gbase - dbase;
#pragma unroll
      for (int i = 0; i < MS_WAVES; i++) {
        s_cnt[i * 256 + tid] = (u16) dbase;
        dbase += c[i];
      }
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
    s_p[pos] = FROMKEYS ? val[j] : (u32) tile_base + e;
    s_x[pos] = SRC == 2 ? (u8) ((u32) key[j] & 63u)
                        : (u8) ((((u32) (key[j] >> 19) & 31u) << 3) | ((u32) key[j] & 7u));
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
}

// ---------------------------------------------------------------------------
// levels B and C: stable partition inside every parent range
// ---------------------------------------------------------------------------
// tiles per parent range; entry np is 0 (its exclusive scan is the tile count).
// Parent s covers pstart[s << sh] .. pstart[(s + 1) << sh].
__global__ __launch_bounds__(256) void k_msd_tilecount(const u32 *__restrict__ pstart, int sh,
                                                       u32 np, u32 per, u32 *__restrict__ cnt) {
  const u32 s = blockIdx.x * 256u + threadIdx.x;
  if (s > np) return;
  if (s == np) { cnt[s] = 0; return; }
  const u32 size = pstart[(u64) (s + 1) << sh] - pstart[(u64) s << sh];
  cnt[s] = (size + per - 1u) / per;
}

// largest s with tfirst[s] <= t (tfirst has np + 1 entries, tfirst[np] > t)
__device__ __forceinline__ u32 ms_parent_of(const u32 *__restrict__ tfirst, u32 np, u32 t) {
  u32 lo = 0, hi = np;        // answer in [lo, hi)
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (tfirst[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_msd_tiledesc(const u32 *__restrict__ pstart,
                                                      const u32 *__restrict__ tfirst, u32 np,
                                                      u32 tiles_ub, MsTile *__restrict__ desc) {
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= tiles_ub) return;
  MsTile d;
  d.start = 0;
  d.segvalid = 0;
  if (t < tfirst[np]) {
    const u32 s = ms_parent_of(tfirst, np, t);
    const u32 k = t - tfirst[s];
    const u32 b = pstart[s] + k * (u32) MS_TILE, e = pstart[s + 1];
    const u32 valid = e - b < (u32) MS_TILE ? e - b : (u32) MS_TILE;
    d.start = b;
    d.segvalid = (s << 13) | valid;
  }
  desc[t] = d;
}

// row t: counts of the digit  key >> dsh  over tile t; rows >= tiles_ub are zero
__global__ __launch_bounds__(MS_THREADS) void k_msd_hist_lvl(
    const u32 *__restrict__ keys, const MsTile *__restrict__ desc, u32 tiles_ub, int dsh,
    u32 *__restrict__ hist) {
  __shared__ u32 h[MS_WAVES][256];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < MS_WAVES * 256; i += MS_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const u32 t = blockIdx.x;
  if (t < tiles_ub) {
    const MsTile d = desc[t];
    const u32 valid = d.segvalid & 0x1FFFu;
    // 16-byte loads from the first entry whose index is a multiple of four (the
    // order inside the tile does not matter to a histogram); the up to three
    // entries in front of it and behind the last whole quad one by one
    u32 head = (4u - (d.start & 3u)) & 3u;
    head = head < valid ? head : valid;
    const u32 nq = (valid - head) >> 2, tail = valid - head - 4u * nq;
    const u32 *kp = keys + d.start;
    uint4 q[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const u32 i = (u32) j * MS_THREADS + tid;
      if (i < nq) q[j] = *reinterpret_cast<const uint4 *>(kp + head + 4u * i);
    }
    if ((u32) tid < head) atomicAdd(&h[w][kp[tid] >> dsh], 1u);
    if ((u32) tid < tail) atomicAdd(&h[w][kp[head + 4u * nq + tid] >> dsh], 1u);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const u32 i = (u32) j * MS_THREADS + tid;
      if (i < nq) {
        atomicAdd(&h[w][q[j].x >> dsh], 1u);
        atomicAdd(&h[w][q[j].y >> dsh], 1u);
        atomicAdd(&h[w][q[j].z >> dsh], 1u);
        atomicAdd(&h[w][q[j].w >> dsh], 1u);
      }
    }
  }
  __syncthreads();
  if (tid < 256) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < MS_WAVES; i++) c += h[i][tid];
    hist[(u64) t * 256 + tid] = c;
  }
}

// entries of child (s, d): difference of the scanned rows at the parent's tile
// borders; tot[np << cb] = 0
__global__ __launch_bounds__(256) void k_msd_tot(const u32 *__restrict__ scanned,
                                                 const u32 *__restrict__ tfirst, u32 np, int cb,
                                                 u32 *__restrict__ tot) {
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  const u64 nchild = (u64) np << cb;
  if (i > nchild) return;
  if (i == nchild) { tot[i] = 0; return; }
  const u32 s = (u32) (i >> cb), d = (u32) i & ((1u << cb) - 1u);
  tot[i] = scanned[(u64) tfirst[s + 1] * 256 + d] - scanned[(u64) tfirst[s] * 256 + d];
}

// LEVEL 1 (B): (K1, X, P) -> (K2 = K1 << 8 | X, P), digit K1 >> 24
// LEVEL 2 (C): (K2, P) -> (K2, P), digit K2 >> dsh
template <int LEVEL>
__global__ __launch_bounds__(MS_THREADS) void k_msd_scatter_lvl(
    const u32 *__restrict__ kin, const u8 *__restrict__ xin, const u32 *__restrict__ pin,
    const MsTile *__restrict__ desc, u32 tiles_ub, const u32 *__restrict__ scanned,
    const u32 *__restrict__ tfirst, const u32 *__restrict__ cstart, int cb, int dsh,
    u32 *__restrict__ kout, u32 *__restrict__ pout) {
  __shared__ u32 s_key[MS_TILE];
  __shared__ u32 s_val[MS_TILE];
  __shared__ u8 s_x[LEVEL == 1 ? MS_TILE : 4];
  __shared__ u16 s_cnt_mem[MS_WAVES * 256];
  __shared__ u32 s_obase[256];
  __shared__ u32 s_scan[MS_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile = ms_xcd_tile(blockIdx.x, tiles_ub);
  if (tile >= tiles_ub) return;
  const MsTile td = desc[tile];
  const u32 valid = td.segvalid & 0x1FFFu, seg = td.segvalid >> 13;
  if (valid == 0) return;
  for (int i = tid; i < MS_WAVES * 256 / 2; i += MS_THREADS)
    reinterpret_cast<u32 *>(s_cnt_mem)[i] = 0;
  // where this tile's entries of digit d go: start of child (seg, d) + what the
  // tiles of the parent in front of this one hold of d
  u32 gbase = 0;
  if (tid < (1 << cb))
    gbase = cstart[((u64) seg << cb) + tid] + scanned[(u64) tile * 256 + tid] -
            scanned[(u64) tfirst[seg] * 256 + tid];
  u32 key[MS_ITEMS], val[MS_ITEMS], xv[MS_ITEMS], rk[MS_ITEMS];
  const u32 *kp = kin + td.start;
  const u32 *pp = pin + td.start;
  const u8 *xp = xin + td.start;
#pragma unroll
  for (int j = 0; j < MS_ITEMS; j++) {
    const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
    if (e < valid) {
      key[j] = kp[e];
      val[j] = pp[e];
      xv[j] = LEVEL == 1 ? (u32) xp[e] : 0u;
    } else {
      key[j] = ~0u;
      val[j] = 0;
      xv[j] = 0;
    }
  }
  __syncthreads();   // counters are zero
  ms_vu16 *cnt_w = (ms_vu16 *) s_cnt_mem + w * 256;
#pragma unroll
  for (int j = 0; j < MS_ITEMS; j++) {
    const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
    const u32 d = e < valid ? key[j] >> dsh : 255u;
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
      for (int i = 0; i < MS_WAVES; i++) {
        c[i] = s_cnt[i * 256 + tid];
        tot += c[i];
      }
    }
    u32 all;
    u32 dbase = block_scan_excl<SCAN_SUM, MS_THREADS>(tot, &all, s_scan);
    if (tid < 256) {
      s_obase[tid] = gbase - dbase;
#pragma unroll
      for (int i = 0; i < MS_WAVES; i++) {
        s_cnt[i * 256 + tid] = (u16) dbase;
        dbase += c[i];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < MS_ITEMS; j++) {
    const u32 d = rk[j] & 255u;
    const u32 pos = (u32) cnt_w[d] + (rk[j] >> 8);
    s_key[pos] = key[j];
    s_val[pos] = val[j];
    if (LEVEL == 1) s_x[pos] = (u8) xv[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < MS_ITEMS; j++) {
    const u32 e = (u32) j * MS_THREADS + tid;
    if (e < valid) {
      const u32 k = s_key[e];
      const u32 g = s_obase[k >> dsh] + e;
      kout[g] = LEVEL == 1 ? (k << 8) | (u32) s_x[e] : k;
      pout[g] = s_val[e];
    }
  }
}

// How deep level C has to cut: what level D pays for ranges that are larger than
// they need be, for cmax - 2 .. cmax bits.  Level D packs whole ranges into tiles
// of MD_CAP entries (k_msd_pack): a range above the tile is a big run (k_msd_big:
// a fifth of k_msd_local's rate), and ranges of r entries fill a tile to
// floor(MD_CAP / r) * r / MD_CAP only -- a half-empty tile costs k_msd_local
// what a full one costs.  pstart: the 65536 parents after level B, whose ranges
// are taken to be of equal size.  out[k]: sum over the entries of the penalty in
// ms per 10^9 entries, for cmax - 2 + k bits.
__global__ __launch_bounds__(256) void k_msd_skew(const u32 *__restrict__ pstart, int cmax,
                                                  float *__restrict__ out) {
  __shared__ float s_sum[3][4];
  const u32 s = blockIdx.x * 256u + threadIdx.x;
  const float size = s < MSD_PARENTS ? (float) (pstart[s + 1] - pstart[s]) : 0.0f;
  float e[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int c = cmax - 2 + k;
    float pen = 0.0f;
    if (c >= 0 && size > 0.0f) {
      const float r = size / (float) (1u << c), cap = (float) MD_CAP;
      if (r > 0.95f * cap) pen = 60.0f;                    // 0.6 ms per cent of the entries in big runs
      else if (r >= 1.0f) {
        const float fill = floorf(cap / r) * r / cap;
        pen = 6.0f * (0.85f / fill - 1.0f);                // k_msd_local: 6 ms per 10^9 entries at 85 %
        if (pen < 0.0f) pen = 0.0f;
      }
    }
    e[k] = size * pen;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float v = e[k];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) s_sum[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float t = s_sum[threadIdx.x][0] + s_sum[threadIdx.x][1] + s_sum[threadIdx.x][2] +
                    s_sum[threadIdx.x][3];
    if (t > 0.0f) atomicAdd(&out[threadIdx.x], t);
  }
}

// ---------------------------------------------------------------------------
// level D: tiles of whole finest-level ranges
// ---------------------------------------------------------------------------
// F: starts of the finest-level ranges (nf + 1 entries), range j of parent s16 is
// (s16 << cb) + j.  Tile k of a parent starts at the first range start that is
// >= parent start + k * MSD_STRIDE.  A tile above the LDS tile goes to the big
// list (one workgroup sorts it in global memory), one above MSD_BIG_MAX to the
// giant list (the driver sorts it with the device-wide sort).  counters: [0] big
// tiles, [1] largest tile, [2] entries in big and giant tiles, [3] giant tiles,
// [4] tiles k_msd_local leaves to k_msd_local_radix
__global__ __launch_bounds__(256) void k_msd_dtiles(const u32 *__restrict__ F, int cb,
                                                    const u32 *__restrict__ tfirst, u32 tiles_ub,
                                                    MdTile *__restrict__ tiles,
                                                    u32 *__restrict__ biglist,
                                                    u32 *__restrict__ giantlist,
                                                    u32 *__restrict__ crowdlist,
                                                    u32 *__restrict__ counters, u32 big_max) {
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= tiles_ub) return;
  MdTile d;
  d.begin = d.end = d.s16 = d.pad = 0;
  if (t < tfirst[MSD_PARENTS]) {
    const u32 s = ms_parent_of(tfirst, MSD_PARENTS, t);
    const u32 k = t - tfirst[s], count = tfirst[s + 1] - tfirst[s];
    const u64 lo = (u64) s << cb, hi = (u64) (s + 1) << cb;
    const u32 pbase = F[lo];
    u32 cut[2];
    u64 cutj[2];     // the range the cut is the start of
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const u32 kk = k + (u32) q;
      if (kk == 0) { cut[q] = pbase; cutj[q] = lo; }
      else if (kk >= count) { cut[q] = F[hi]; cutj[q] = hi; }
      else {
        // first range start >= target among F[lo .. hi]
        const u64 target = (u64) pbase + (u64) kk * MSD_STRIDE;
        u64 a = lo, b = hi;           // F[a] < target <= F[b] (F[hi] = parent end >= target)
        while (b - a > 1) {
          const u64 mid = (a + b) >> 1;
          if ((u64) F[mid] >= target) b = mid; else a = mid;
        }
        cut[q] = F[b];
        cutj[q] = b;
      }
    }
    d.begin = cut[0];
    d.end = cut[1];
    d.s16 = s;
    // level-C digit of the first range and the number of ranges: what is left
    // to sort is K2 - (digit << (32 - cb)), below (ranges << (32 - cb))
    d.pad = (u32) (cutj[0] - lo) | ((u32) (cutj[1] - cutj[0]) << 16);
    const u32 cnt = d.end - d.begin;
    if (cnt > big_max) {
      giantlist[atomicAdd(&counters[3], 1u)] = t;
      atomicAdd(&counters[2], cnt);
      atomicMax(&counters[1], cnt);
    } else if (cnt > (u32) MS_TILE) {
      biglist[atomicAdd(&counters[0], 1u)] = t;
      atomicAdd(&counters[2], cnt);
      atomicMax(&counters[1], cnt);
    }
  }
  tiles[t] = d;
}

// The same tiles by packing: a parent's ranges in order, a tile takes whole
// ranges while they fit `cap` entries (at least one).  Only a single range above
// the LDS tile makes a big run; with the stride rule above two ranges of 2500
// entries make one of 5000 (a text with 70 % A + T at 3 Gbp: 76 119 big runs, 13 %
// of the entries, 11 ms in k_msd_big).  One thread per parent; COUNT: tiles per
// parent (cnt[MSD_PARENTS] = 0), EMIT: the descriptors from tfirst[parent] on.
template <bool EMIT>
__global__ __launch_bounds__(256) void k_msd_pack(const u32 *__restrict__ F, int cb,
                                                  const u32 *__restrict__ tfirst, u32 cap,
                                                  u32 *__restrict__ cnt, MdTile *__restrict__ tiles,
                                                  u32 *__restrict__ biglist,
                                                  u32 *__restrict__ giantlist,
                                                  u32 *__restrict__ counters, u32 big_max) {
  const u32 s = blockIdx.x * 256u + threadIdx.x;
  if (s > MSD_PARENTS) return;
  if (s == MSD_PARENTS) { if (!EMIT) cnt[s] = 0; return; }
  const u64 lo = (u64) s << cb;
  const u32 nr = 1u << cb;
  const u32 out0 = EMIT ? tfirst[s] : 0u;
  u32 t = 0, jstart = 0;
  u32 begin = F[lo], prev = begin;
  auto close = [&](u32 end, u32 jend) {
    if (EMIT) {
      MdTile d;
      d.begin = begin; d.end = end; d.s16 = s;
      d.pad = jstart | ((jend - jstart) << 16);
      const u32 idx = out0 + t, n = end - begin;
      if (n > big_max) {
        giantlist[atomicAdd(&counters[3], 1u)] = idx;
        atomicAdd(&counters[2], n);
        atomicMax(&counters[1], n);
      } else if (n > (u32) MS_TILE) {
        biglist[atomicAdd(&counters[0], 1u)] = idx;
        atomicAdd(&counters[2], n);
        atomicMax(&counters[1], n);
      }
      tiles[idx] = d;
    }
    t++;
  };
  for (u32 j0 = 0; j0 < nr; j0 += 8) {
    u32 nx[8];
#pragma unroll
    for (int k = 0; k < 8; k++) nx[k] = j0 + k < nr ? F[lo + j0 + k + 1] : 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32 j = j0 + (u32) k;
      if (j < nr) {
        const u32 next = nx[k];
        if (prev > begin && next - begin > cap) {    // range j does not fit behind what the tile holds
          close(prev, j);
          begin = prev;
          jstart = j;
        }
        prev = next;
      }
    }
  }
  if (prev > begin) close(prev, nr);
  if (!EMIT) cnt[s] = t;
}

// Table entries of the sorted run [gbeg, gbeg + cnt) that lies in LDS
// (s_key[i] + base = K2 of entry i, s_val[i] its position): .suf, .lcp
// (provisional for tied entries, as k_finalize), .bwt, the 32-bit positions and
// the tie bits.  The entries of a run share their first 8 symbols, so all of
// it is 32-bit arithmetic on K2 (symbols 8..19 | dcode | payload).  The entry
// in front of the run: K2 = prevk2 if has_prev, else the run's first entry
// gets lcp 0 and k_msd_seams settles it.
__device__ __forceinline__ u32 k2_letters(u32 k2) {
  const u32 dc = (k2 >> 3) & 31u;
  return dc == 0 ? 20u : (dc == 31u ? 0u : 20u - dc);
}
// one quad of table entries (entries i0 .. i0+3 of the run, global index g0 ..);
// INTERIOR: all four and the entry in front are inside the run
template <int FMT, bool INTERIOR>
__device__ __forceinline__ u32 msd_emit_quad(const u32 *s_key, const u32 *s_val, int i0, u32 cnt,
                                             u64 g0, u32 base, bool has_prev, u32 prevk2, u32 s16,
                                             const MsdOut &o, MsdAcc &acc) {
  u32 k[4], pv[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int i = i0 + c;
    const bool ok = INTERIOR || (i >= 0 && i < (int) cnt);
    k[c] = ok ? s_key[i] + base : 0u;
    pv[c] = ok ? s_val[i] : 0u;
  }
  const u32 prevk = (INTERIOR || i0 > 0) ? s_key[i0 - 1] + base : prevk2;
  u32 lcpv[4] = {0, 0, 0, 0}, tiemask = 0;
  const u64 chi = (u64) s16 << 24;                    // FMT 1: the code bits the run shares
  // (FMT 1: X bit 5 says whether the code shows a padding at all -- 3 % of the suffixes
  // of a protein set; the others have "nine or more" letters)
  u32 da = FMT == 1 ? ((prevk & 32u) ? p5_letters(chi | (prevk >> 8)) : 9u) : k2_letters(prevk);
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int i = i0 + c;
    const bool ok = INTERIOR || (i >= 0 && i < (int) cnt);
    const bool have_a = INTERIOR || i > 0 || has_prev;
    const u32 a = (c == 0 || (!INTERIOR && i == 0)) ? prevk : k[c - 1], b = k[c];
    if (!INTERIOR && c > 0 && i == 0)
      da = FMT == 1 ? ((prevk & 32u) ? p5_letters(chi | (prevk >> 8)) : 9u) : k2_letters(prevk);
    const u32 db = FMT == 1 ? ((b & 32u) ? p5_letters(chi | (b >> 8)) : 9u) : k2_letters(b);
    const u32 x = (a ^ b) >> 8;                       // symbols 8 .. 19 / the code's low 24 bits
    u32 m;
    bool tie;
    if (FMT == 1) {
      m = p5_common(chi | (a >> 8), chi | (b >> 8));
      tie = x == 0 && db == 9u;                       // the same code, nine letters or more
    } else {
      m = x ? 8u + ((u32) __clz((int) x) - 8u) / 2u : 20u;
      tie = x == 0 && ((a | b) & 0xF8u) == 0;         // all 20 symbols, both dcodes 0
    }
    u32 l = m < da ? m : da;
    l = l < db ? l : db;
    if (!have_a) { l = 0; tie = false; }
    da = db;
    if (!ok) continue;
    lcpv[c] = l;
    if (tie) {
      tiemask |= 1u << c;
      acc.ties++;
    } else {
      if (have_a) {
        acc.mx = l > acc.mx ? l : acc.mx;
        if (db >= o.prefixlength) acc.sum += l;
      }
      if (pv[c] == 0 && !o.val_is_index) o.stats->longest = o.index_offset + g0 + (u64) c;
    }
  }
  if (INTERIOR || (i0 >= 0 && i0 + 4 <= (int) cnt)) {
    if (o.suf != nullptr) {
      *reinterpret_cast<ulonglong2 *>(o.suf + g0) = make_ulonglong2(pv[0], pv[1]);
      *reinterpret_cast<ulonglong2 *>(o.suf + g0 + 2) = make_ulonglong2(pv[2], pv[3]);
    }
    *reinterpret_cast<uint4 *>(o.sa + g0) = make_uint4(pv[0], pv[1], pv[2], pv[3]);
    if (o.lcp != nullptr)
      *reinterpret_cast<u32 *>(o.lcp + g0) =
          lcpv[0] | (lcpv[1] << 8) | (lcpv[2] << 16) | (lcpv[3] << 24);
    if (o.bwt != nullptr) {
      u32 bw = 0;
#pragma unroll
      for (int c = 0; c < 4; c++)
        bw |= (u32) (FMT == 1 ? Pay<5>::to_bwt(k[c] & 31u) : Pay<2>::to_bwt(k[c] & 7u)) << (8 * c);
      *reinterpret_cast<u32 *>(o.bwt + g0) = bw;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int i = i0 + c;
      if (i < 0 || i >= (int) cnt) continue;
      if (o.suf != nullptr) o.suf[g0 + c] = pv[c];
      o.sa[g0 + c] = pv[c];
      if (o.lcp != nullptr) o.lcp[g0 + c] = (u8) lcpv[c];
      if (o.bwt != nullptr) o.bwt[g0 + c] = FMT == 1 ? Pay<5>::to_bwt(k[c] & 31u) : Pay<2>::to_bwt(k[c] & 7u);
    }
  }
  return tiemask;
}

template <int FMT>
__device__ __forceinline__ void msd_emit(const u32 *s_key, const u32 *s_val, u32 cnt, u64 gbeg,
                                         u32 base, bool has_prev, u32 prevk2, u32 s16,
                                         const MsdOut &o, u32 *s_bits, MsdAcc &acc) {
  static_assert(Key<2>::SYMS == 20 && Key<2>::DMAX == 31, "the layout msd_full spells out");
  const int tid = threadIdx.x;
  const u32 off64 = (u32) (gbeg & 63);
  const u32 nbw = (cnt + off64 + 63u) >> 6;
  for (u32 i = tid; i < 2u * nbw; i += MS_THREADS) s_bits[i] = 0;
  lds_barrier();
  const u32 mis = (u32) (gbeg & 3);
  const u32 nquads = (cnt + mis + 3u) >> 2;
  const u64 gq = gbeg - mis;
  for (u32 q = tid; q < nquads; q += MS_THREADS) {
    const int i0 = (int) (4u * q) - (int) mis;
    const u64 g0 = gq + 4ull * q;
    u32 tiemask;
    if (i0 >= 1 && i0 + 4 <= (int) cnt)
      tiemask = msd_emit_quad<FMT, true>(s_key, s_val, i0, cnt, g0, base, has_prev, prevk2, s16, o, acc);
    else
      tiemask = msd_emit_quad<FMT, false>(s_key, s_val, i0, cnt, g0, base, has_prev, prevk2, s16, o, acc);
    if (tiemask) {
      const u32 bo = 4u * q + (off64 - mis);   // bit of quad entry 0 in the LDS bitmap
      atomicOr(&s_bits[bo >> 5], tiemask << (bo & 31u));
    }
  }
  lds_barrier();
  const u64 w0 = gbeg >> 6;
  for (u32 wi = tid; wi < nbw; wi += MS_THREADS) {
    const u64 v = (u64) s_bits[2u * wi] | ((u64) s_bits[2u * wi + 1u] << 32);
    if (v == 0) continue;
    const bool interior = (wi > 0 || off64 == 0) && (w0 + wi + 1) * 64 <= gbeg + cnt;
    if (interior) o.tiebits[w0 + wi] = v;
    else atomicOr(reinterpret_cast<unsigned long long *>(o.tiebits + w0 + wi),
                  (unsigned long long) v);
  }
  lds_barrier();
}

__device__ __forceinline__ void msd_acc_flush(MsdAcc acc, Stats *stats) {
  __shared__ unsigned long long s_sum[MS_WAVES], s_ties[MS_WAVES];
  __shared__ u32 s_max[MS_WAVES];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    acc.sum += __shfl_xor(acc.sum, d, 64);
    acc.ties += __shfl_xor(acc.ties, d, 64);
    const u32 ot = __shfl_xor(acc.mx, d, 64);
    acc.mx = ot > acc.mx ? ot : acc.mx;
  }
  if (lane == 0) { s_sum[w] = acc.sum; s_ties[w] = acc.ties; s_max[w] = acc.mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long S = 0, T = 0;
    u32 M = 0;
    for (int i = 0; i < (int) (blockDim.x >> 6); i++) {
      S += s_sum[i]; T += s_ties[i]; M = s_max[i] > M ? s_max[i] : M;
    }
    if (S) atomicAdd(&stats->lcpsum, S);
    if (T) atomicAdd(&stats->numties, T);
    if (M) atomicMax(&stats->maxlcp, M);
  }
}

// what of K2 is left to sort in the run [begin, end): K2 minus the first
// entry's level-C digit, bits [3, nbits)
__device__ __forceinline__ void msd_run_bits(const u32 *__restrict__ kin, u32 begin, u32 end,
                                             int cb, u32 &base, int &nbits) {
  if (cb == 0) { base = 0; nbits = 32; return; }
  const int csh = 32 - cb;
  const u32 f0 = kin[begin] >> csh, f1 = kin[end - 1] >> csh;
  base = f0 << csh;
  nbits = csh + (f1 > f0 ? 32 - __clz((int) (f1 - f0)) : 0);
}

constexpr int MD_BITS = 9;
constexpr int MD_RADIX = 1 << MD_BITS;
static_assert(MD_RADIX == MS_THREADS, "one thread per digit in the scan");

// The sort inside a run.  Fast path: ONE counting pass on the top 12 of the
// bits that are left (LDS atomics; a bin holds the entries that share ~18
// symbols: one, seldom more), then every entry finds its place inside its bin by
// comparing (rest of the bits, input order) with its bin mates -- exact and
// stable whatever order the atomics came back in.  A tile with a bin above
// MD_BIN_LIMIT (many copies of one 20-mer) takes the stable LSD passes instead
// (three or four 9-bit digits, ballot-ranked like the scatter kernels).
constexpr int MD_BINBITS = 12;
constexpr int MD_BINS = 1 << MD_BINBITS;
constexpr u32 MD_BIN_LIMIT_DEFAULT = 128;

__device__ __forceinline__ u32 md_base(const u32 *s_binw, u32 bin) {
  return (s_binw[bin >> 1] >> ((bin & 1u) * 16u)) & 0xFFFFu;
}

template <int FMT>
__global__ __launch_bounds__(MS_THREADS, 4) void k_msd_local(
    const u32 *__restrict__ kin, const u32 *__restrict__ pin, const MdTile *__restrict__ tiles,
    u32 ntiles, int cb, int force_radix, u32 bin_limit, u32 *__restrict__ crowdlist,
    u32 *__restrict__ counters, MsdOut o) {
  __shared__ u32 s_key[MS_TILE];
  __shared__ u32 s_val[MS_TILE];
  // bin counters (two 16-bit counters per word, one more for the end of the
  // last bin) and the flag
  __shared__ __attribute__((aligned(16))) u32 s_binw[MD_BINS / 2 + 8];
  __shared__ u32 s_scan[MS_WAVES];
  __shared__ u32 s_bits[(MS_TILE + 128) / 32];
  static_assert(MD_BINS / 2 == 4 * MS_THREADS, "four counter words per thread in the scan");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  u32 *s_c = s_key;     // (rest of the bits, input order) at the counting-sort place
  MsdAcc acc;
  acc.sum = acc.ties = 0;
  acc.mx = 0;
  // The kernel is bound by instructions (a run is ~3500 entries, seven per
  // thread; what a thread does once per run -- its share of the bin scan, the
  // loop, the barriers -- weighs as much as what it does per entry), then by
  // the latency of its dozen phases.  So: big runs, no barrier that waits for
  // global memory, the next run's entries fetched while this run's tables go
  // out, the LDS reads of a phase issued for all entries before any is used.
  u32 key[MD_ITEMS], val[MD_ITEMS], rk[MD_ITEMS];
  const MdTile none = {0u, 0u, 0u, 0u};
  MdTile td = none;
  u32 t = blockIdx.x;
  if (t < ntiles) td = tiles[t];
  auto fetch = [&](const MdTile &x) {
    const u32 cnt = x.end - x.begin;
    if (cnt == 0 || cnt > MD_CAP) return;
    const u32 items = (cnt + MS_THREADS - 1) / MS_THREADS, wchunk = items * 64u;
    const u32 *kp = kin + x.begin;
    const u32 *pp = pin + x.begin;
#pragma unroll
    for (int j = 0; j < MD_ITEMS; j++) {
      const u32 e = (u32) w * wchunk + (u32) j * 64 + lane;
      if ((u32) j < items && e < cnt) {
        key[j] = kp[e];
        val[j] = pp[e];
      }
    }
  };
  fetch(td);
  while (t < ntiles) {
    const u32 tn = t + gridDim.x;
    MdTile tdn = none;
    if (tn < ntiles) tdn = tiles[tn];
    const u32 cnt = td.end - td.begin;
    bool skip = cnt == 0 || cnt > MD_CAP;   // (longer runs: the other kernels)
    u32 base = 0;
    if (!skip) {
      // what is left to sort: K2 minus the first range's level-C digit
      const int csh = 32 - cb;
      const u32 span = td.pad >> 16;
      base = cb ? (td.pad & 0xFFFFu) << csh : 0u;
      const int nbits = cb ? csh + (span > 1u ? 32 - __clz((int) (span - 1u)) : 0) : 32;
      // items per thread and the wave's chunk, so that a part-filled tile keeps
      // all eight waves busy
      const u32 items = (cnt + MS_THREADS - 1) / MS_THREADS, wchunk = items * 64u;
      const u32 e0 = (u32) w * wchunk + lane;      // entry of item j: e0 + 64 j
      // ---- counting pass
      constexpr int PB = MsdFmt<FMT>::PB;
      const int sb = nbits - PB;
      const int binshift = sb > MD_BINBITS ? nbits - MD_BINBITS : PB;
      const u32 lowmask = sb > MD_BINBITS ? (1u << (binshift - PB)) - 1u : 0u;
      for (int i = tid; i < MD_BINS / 8 + 2; i += MS_THREADS)
        reinterpret_cast<uint4 *>(s_binw)[i] = make_uint4(0, 0, 0, 0);
      lds_barrier();
#pragma unroll
      for (int j = 0; j < MD_ITEMS; j++) {
        if ((u32) j < items && e0 + 64u * j < cnt) {
          key[j] -= base;
          const u32 bin = key[j] >> binshift;
          const u32 old = atomicAdd(&s_binw[bin >> 1], (bin & 1u) ? 65536u : 1u);
          rk[j] = (bin & 1u) ? old >> 16 : old & 0xFFFFu;
        }
      }
      lds_barrier();
      {
        // thread tid owns bins 8 tid .. 8 tid + 7: counts -> exclusive starts
        const uint4 q0 = reinterpret_cast<uint4 *>(s_binw)[tid];
        u32 wd[4] = {q0.x, q0.y, q0.z, q0.w};
        u32 run = 0, mx = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u32 c0 = wd[k] & 0xFFFFu, c1 = wd[k] >> 16;
          mx = c0 > mx ? c0 : mx;
          mx = c1 > mx ? c1 : mx;
          wd[k] = run | ((run + c0) << 16);
          run += c0 + c1;
        }
        const u32 pre = ms_scan_excl(run, s_scan) * 0x10001u;
        reinterpret_cast<uint4 *>(s_binw)[tid] =
            make_uint4(wd[0] + pre, wd[1] + pre, wd[2] + pre, wd[3] + pre);
        if (tid == 0) s_binw[MD_BINS / 2] = cnt;           // end of the last bin
        if (mx > bin_limit || force_radix) s_binw[MD_BINS / 2 + 1] = 1u;
      }
      lds_barrier();
      if (s_binw[MD_BINS / 2 + 1] != 0) {     // (the same for every thread)
        // a crowded bin: this run is left to k_msd_local_radix
        if (tid == 0) {
          crowdlist[atomicAdd(&counters[4], 1u)] = t;
          atomicAdd(&o.stats->crowded, (unsigned long long) cnt);
        }
        skip = true;
      } else {
        // every entry to its bin, in the order the atomics came back, as
        // (rest of the bits, input order)
        u32 b0[MD_ITEMS];
#pragma unroll
        for (int j = 0; j < MD_ITEMS; j++)
          b0[j] = ((u32) j < items && e0 + 64u * j < cnt) ? md_base(s_binw, key[j] >> binshift) : 0u;
#pragma unroll
        for (int j = 0; j < MD_ITEMS; j++)
          if ((u32) j < items && e0 + 64u * j < cnt)
            s_c[b0[j] + rk[j]] = (((key[j] >> PB) & lowmask) << 12) | (e0 + 64u * j);
        lds_barrier();
        // its place: the bin's start + the bin mates that are smaller.  The first
        // two mates in one go (a bin seldom has more), the rest one by one.
        u32 b1[MD_ITEMS], m0[MD_ITEMS], m1[MD_ITEMS];
#pragma unroll
        for (int j = 0; j < MD_ITEMS; j++) {
          const bool ok = (u32) j < items && e0 + 64u * j < cnt;
          b1[j] = ok ? md_base(s_binw, (key[j] >> binshift) + 1u) : 1u;
          m0[j] = s_c[b0[j]];
        }
#pragma unroll
        for (int j = 0; j < MD_ITEMS; j++) m1[j] = s_c[b0[j] + 1u < b1[j] ? b0[j] + 1u : b0[j]];
#pragma unroll
        for (int j = 0; j < MD_ITEMS; j++) {
          const u32 c = (((key[j] >> PB) & lowmask) << 12) | (e0 + 64u * j);
          u32 r = b0[j] + (m0[j] < c ? 1u : 0u) + (m1[j] < c ? 1u : 0u);   // (m1 = m0 if alone:
          if (b0[j] + 1u >= b1[j]) r = b0[j];                              //  then the place is b0)
          if (b1[j] - b0[j] > 2u)
            for (u32 q = b0[j] + 2u; q < b1[j]; q++) r += s_c[q] < c ? 1u : 0u;
          rk[j] = r;
        }
        lds_barrier();   // s_c is read, its space is s_key again
#pragma unroll
        for (int j = 0; j < MD_ITEMS; j++) {
          if ((u32) j < items && e0 + 64u * j < cnt) {
            s_key[rk[j]] = key[j];
            s_val[rk[j]] = val[j];
          }
        }
        lds_barrier();
      }
    }
    // the next run's entries are on their way while this run's tables go out
    fetch(tdn);
    if (!skip) {
      if (tid == 0) {
        o.firstkey[t] = msd_full<FMT>(td.s16, s_key[0] + base);
        o.lastkey[t] = msd_full<FMT>(td.s16, s_key[cnt - 1] + base);
      }
      msd_emit<FMT>(s_key, s_val, cnt, td.begin, base, false, 0u, td.s16, o, s_bits, acc);
    }
    t = tn;
    td = tdn;
  }
  __syncthreads();
  msd_acc_flush(acc, o.stats);
}

// the runs k_msd_local has left (a crowded bin) or never had (above its 2044
// entries): stable LSD passes on 9-bit digits, ballot-ranked like the scatter
// kernels, up to 4096 entries
template <int FMT>
__global__ __launch_bounds__(MS_THREADS) void k_msd_local_radix(
    const u32 *__restrict__ kin, const u32 *__restrict__ pin, const MdTile *__restrict__ tiles,
    const u32 *__restrict__ crowdlist, const u32 *__restrict__ counters, int cb, MsdOut o) {
  __shared__ u32 s_key[MS_TILE];
  __shared__ u32 s_val[MS_TILE];
  __shared__ u16 s_cnt_mem[MS_WAVES * MD_RADIX];
  __shared__ u32 s_scan[MS_WAVES];
  __shared__ u32 s_bits[(MS_TILE + 128) / 32];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  ms_vu16 *s_cnt = (ms_vu16 *) s_cnt_mem;
  ms_vu16 *cnt_w = s_cnt + w * MD_RADIX;
  const u32 ncrowd = counters[4];
  MsdAcc acc;
  acc.sum = acc.ties = 0;
  acc.mx = 0;
  for (u32 ci = blockIdx.x; ci < ncrowd; ci += gridDim.x) {
    const u32 t = crowdlist[ci];
    const MdTile td = tiles[t];
    const u32 cnt = td.end - td.begin;
    const int csh = 32 - cb;
    const u32 span = td.pad >> 16;
    const u32 base = cb ? (td.pad & 0xFFFFu) << csh : 0u;
    const int nbits = cb ? csh + (span > 1u ? 32 - __clz((int) (span - 1u)) : 0) : 32;
    const int npass = (nbits - MsdFmt<FMT>::PB + MD_BITS - 1) / MD_BITS;
    const u32 items = (cnt + MS_THREADS - 1) / MS_THREADS, wchunk = items * 64u;
    u32 key[MS_ITEMS], val[MS_ITEMS], rk[MS_ITEMS];
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) {
      const u32 e = (u32) w * wchunk + (u32) j * 64 + lane;
      if ((u32) j < items && e < cnt) {
        key[j] = kin[(u64) td.begin + e] - base;
        val[j] = pin[(u64) td.begin + e];
      } else {
        key[j] = ~0u;      // behind every entry, in every pass
        val[j] = 0;
      }
    }
    for (int p = 0; p < npass; p++) {
      const int shift = MsdFmt<FMT>::PB + MD_BITS * p;
      for (int i = tid; i < MS_WAVES * MD_RADIX / 2; i += MS_THREADS)
        reinterpret_cast<u32 *>(s_cnt_mem)[i] = 0;
      lds_barrier();
#pragma unroll
      for (int j = 0; j < MS_ITEMS; j++) {
        if ((u32) j < items) {
          const u32 d = (key[j] >> shift) & (u32) (MD_RADIX - 1);
          u32 intra, group;
          ms_match<MD_BITS>(d, intra, group);
          const u32 old = cnt_w[d];
          if (intra == 0) cnt_w[d] = (u16) (old + group);
          rk[j] = ((old + intra) << MD_BITS) | d;
        }
      }
      lds_barrier();
      {
        u32 c[MS_WAVES];
        u32 tot = 0;
#pragma unroll
        for (int i = 0; i < MS_WAVES; i++) {
          c[i] = s_cnt[i * MD_RADIX + tid];
          tot += c[i];
        }
        u32 dbase = ms_scan_excl(tot, s_scan);
#pragma unroll
        for (int i = 0; i < MS_WAVES; i++) {
          s_cnt[i * MD_RADIX + tid] = (u16) dbase;
          dbase += c[i];
        }
      }
      lds_barrier();
#pragma unroll
      for (int j = 0; j < MS_ITEMS; j++) {
        if ((u32) j < items) {
          const u32 d = rk[j] & (u32) (MD_RADIX - 1);
          const u32 pos = (u32) cnt_w[d] + (rk[j] >> MD_BITS);
          s_key[pos] = key[j];
          s_val[pos] = val[j];
        }
      }
      lds_barrier();
      if (p + 1 < npass) {
#pragma unroll
        for (int j = 0; j < MS_ITEMS; j++) {
          if ((u32) j < items) {
            const u32 e = (u32) w * wchunk + (u32) j * 64 + lane;
            key[j] = s_key[e];
            val[j] = s_val[e];
          }
        }
        lds_barrier();
      }
    }
    if (tid == 0) {
      o.firstkey[t] = msd_full<FMT>(td.s16, s_key[0] + base);
      o.lastkey[t] = msd_full<FMT>(td.s16, s_key[cnt - 1] + base);
    }
    msd_emit<FMT>(s_key, s_val, cnt, td.begin, base, false, 0u, td.s16, o, s_bits, acc);
  }
  __syncthreads();
  msd_acc_flush(acc, o.stats);
}

// oversize runs: one workgroup sorts a run in global memory, 8-bit digits,
// ping-pong between (ka, pa) -- where the run lies and where the 32-bit
// positions have to end up -- and the same index range of (kb, pb)
template <int FMT>
__global__ __launch_bounds__(MS_THREADS) void k_msd_big(
    u32 *__restrict__ ka, u32 *__restrict__ pa, u32 *__restrict__ kb, u32 *__restrict__ pb,
    const MdTile *__restrict__ tiles, const u32 *__restrict__ biglist,
    const u32 *__restrict__ counters, int cb, MsdOut o) {
  __shared__ u32 s_key[MS_TILE];
  __shared__ u32 s_val[MS_TILE];
  __shared__ u16 s_cnt_mem[MS_WAVES * 256];
  __shared__ u32 s_obase[256], s_cur[256];
  __shared__ u32 s_scan[MS_WAVES];
  __shared__ u32 s_bits[(MS_TILE + 128) / 32];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  ms_vu16 *s_cnt = (ms_vu16 *) s_cnt_mem;
  ms_vu16 *cnt_w = s_cnt + w * 256;
  const u32 nbig = counters[0];
  MsdAcc acc;
  acc.sum = acc.ties = 0;
  acc.mx = 0;
  for (u32 bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
    const u32 t = biglist[bi];
    const MdTile td = tiles[t];
    const u32 cnt = td.end - td.begin;
    u32 base;
    int nbits;
    msd_run_bits(ka, td.begin, td.end, cb, base, nbits);
    const int npass = nbits > MsdFmt<FMT>::PB ? (nbits - MsdFmt<FMT>::PB + 7) / 8 : 0;
    u32 *sk = ka + td.begin, *sp = pa + td.begin, *dk = kb + td.begin, *dp = pb + td.begin;
    __syncthreads();   // (base was read from ka by every thread before anyone writes)
    for (int p = 0; p < npass; p++) {
      const int shift = MsdFmt<FMT>::PB + 8 * p;
      if (tid < 256) s_cur[tid] = 0;
      __syncthreads();
      for (u32 e = tid; e < cnt; e += MS_THREADS)
        atomicAdd(&s_cur[((sk[e] - base) >> shift) & 255u], 1u);
      __syncthreads();
      {
        const u32 v = tid < 256 ? s_cur[tid] : 0u;
        u32 all;
        const u32 x = block_scan_excl<SCAN_SUM, MS_THREADS>(v, &all, s_scan);
        if (tid < 256) s_cur[tid] = x;
      }
      __syncthreads();
      for (u32 c0 = 0; c0 < cnt; c0 += (u32) MS_TILE) {
        const u32 valid = cnt - c0 < (u32) MS_TILE ? cnt - c0 : (u32) MS_TILE;
        for (int i = tid; i < MS_WAVES * 256 / 2; i += MS_THREADS)
          reinterpret_cast<u32 *>(s_cnt_mem)[i] = 0;
        u32 key[MS_ITEMS], val[MS_ITEMS], rk[MS_ITEMS];
#pragma unroll
        for (int j = 0; j < MS_ITEMS; j++) {
          const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
          if (e < valid) {
            key[j] = sk[c0 + e] - base;
            val[j] = sp[c0 + e];
          } else {
            key[j] = ~0u;
            val[j] = 0;
          }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MS_ITEMS; j++) {
          const u32 e = (u32) w * MS_WCHUNK + (u32) j * 64 + lane;
          const u32 d = e < valid ? (key[j] >> shift) & 255u : 255u;
          u32 intra, group;
          ms_match<8>(d, intra, group);
          const u32 old = cnt_w[d];
          if (intra == 0) cnt_w[d] = (u16) (old + group);
          rk[j] = ((old + intra) << 8) | d;
        }
        __syncthreads();
        {
          u32 c[MS_WAVES];
          u32 tot = 0;
          if (tid < 256) {
#pragma unroll
            for (int i = 0; i < MS_WAVES; i++) {
              c[i] = s_cnt[i * 256 + tid];
              tot += c[i];
            }
          }
          u32 all;
          u32 dbase = block_scan_excl<SCAN_SUM, MS_THREADS>(tot, &all, s_scan);
          if (tid < 256) {
            const u32 cur = s_cur[tid];
            s_obase[tid] = cur - dbase;
            // (the entries behind `valid` sit in digit 255: they are not written
            // and do not move the cursor)
            s_cur[tid] = cur + tot - (tid == 255 ? (u32) MS_TILE - valid : 0u);
#pragma unroll
            for (int i = 0; i < MS_WAVES; i++) {
              s_cnt[i * 256 + tid] = (u16) dbase;
              dbase += c[i];
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MS_ITEMS; j++) {
          const u32 d = rk[j] & 255u;
          const u32 pos = (u32) cnt_w[d] + (rk[j] >> 8);
          s_key[pos] = key[j];
          s_val[pos] = val[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MS_ITEMS; j++) {
          const u32 e = (u32) j * MS_THREADS + tid;
          if (e < valid) {
            const u32 k = s_key[e];
            const u32 g = s_obase[(k >> shift) & 255u] + e;
            dk[g] = k + base;
            dp[g] = s_val[e];
          }
        }
        __syncthreads();
      }
      u32 *x = sk; sk = dk; dk = x;
      x = sp; sp = dp; dp = x;
    }
    // the sorted run is in (sk, sp); table entries chunk by chunk
    u32 prevkey = 0;
    for (u32 c0 = 0; c0 < cnt; c0 += (u32) MS_TILE) {
      const u32 valid = cnt - c0 < (u32) MS_TILE ? cnt - c0 : (u32) MS_TILE;
#pragma unroll
      for (int j = 0; j < MS_ITEMS; j++) {
        const u32 e = (u32) j * MS_THREADS + tid;
        if (e < valid) {
          s_key[e] = sk[c0 + e] - base;
          s_val[e] = sp[c0 + e];
        }
      }
      __syncthreads();
      if (tid == 0) {
        if (c0 == 0) o.firstkey[t] = msd_full<FMT>(td.s16, s_key[0] + base);
        if (c0 + valid == cnt) o.lastkey[t] = msd_full<FMT>(td.s16, s_key[valid - 1] + base);
      }
      const u32 lastk = s_key[valid - 1] + base;
      msd_emit<FMT>(s_key, s_val, valid, (u64) td.begin + c0, base, c0 > 0, prevkey, td.s16, o, s_bits, acc);
      prevkey = lastk;
    }
  }
  msd_acc_flush(acc, o.stats);
}

// table entries of a giant run the device-wide sort has put in order
template <int FMT>
__global__ __launch_bounds__(MS_THREADS) void k_msd_emit_run(
    const u32 *__restrict__ kin, const u32 *__restrict__ pin, u32 t, u32 begin, u32 cnt, u32 s16,
    MsdOut o) {
  __shared__ u32 s_key[MS_TILE];
  __shared__ u32 s_val[MS_TILE];
  __shared__ u32 s_bits[(MS_TILE + 128) / 32];
  const int tid = threadIdx.x;
  MsdAcc acc;
  acc.sum = acc.ties = 0;
  acc.mx = 0;
  const u32 c0 = blockIdx.x * (u32) MS_TILE;
  const u32 valid = cnt - c0 < (u32) MS_TILE ? cnt - c0 : (u32) MS_TILE;
#pragma unroll
  for (int j = 0; j < MS_ITEMS; j++) {
    const u32 e = (u32) j * MS_THREADS + tid;
    if (e < valid) {
      s_key[e] = kin[(u64) begin + c0 + e];
      s_val[e] = pin[(u64) begin + c0 + e];
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (c0 == 0) o.firstkey[t] = msd_full<FMT>(s16, s_key[0]);
    if (c0 + valid == cnt) o.lastkey[t] = msd_full<FMT>(s16, s_key[valid - 1]);
  }
  const u32 prevkey = c0 > 0 ? kin[(u64) begin + c0 - 1] : 0u;
  msd_emit<FMT>(s_key, s_val, valid, (u64) begin + c0, 0u, c0 > 0, prevkey, s16, o, s_bits, acc);
  msd_acc_flush(acc, o.stats);
}

// lcp of every run's first entry with the last entry of the run in front of it;
// a part build: of the slice's first entry with the largest key of the ranges
// below (prev_key, if has_prev)
template <int FMT>
__global__ __launch_bounds__(256) void k_msd_seams(const MdTile *__restrict__ tiles, u32 ntiles,
                                                   const unsigned long long *__restrict__ prev_key_p,
                                                   int has_prev, MsdOut o) {
  using K = Key<2>;
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  MsdAcc acc;
  acc.sum = acc.ties = 0;
  acc.mx = 0;
  if (t < ntiles) {
    const MdTile td = tiles[t];
    if (td.end > td.begin && (td.begin > 0 || has_prev)) {
      u64 a = has_prev ? (u64) *prev_key_p : 0ull;
      if (td.begin > 0) {
        u32 u = t - 1;
        while (u > 0 && tiles[u].end == tiles[u].begin) u--;   // (an entry in front exists: begin > 0)
        a = o.lastkey[u];
      }
      const u64 b = o.firstkey[t];
      u32 da, db, m;
      if (FMT == 1) {        // (code << 8 | X)
        da = p5_letters(a >> 8); db = p5_letters(b >> 8);
        m = p5_common(a >> 8, b >> 8);
      } else {
        da = K::letters(a); db = K::letters(b);
        const u64 x = (a ^ b) >> K::LOW_BITS;
        m = x ? (u32) (__clzll((long long) x) - K::LOW_BITS) / 2u : (u32) K::SYMS;
      }
      u32 l = m < da ? m : da;
      l = l < db ? l : db;
      if (o.lcp != nullptr) o.lcp[td.begin] = (u8) l;
      acc.mx = l;
      if (db >= o.prefixlength) acc.sum = l;
    }
  }
  msd_acc_flush(acc, o.stats);
}
