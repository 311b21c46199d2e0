// esa_encode.hip -- device-side FASTA encoder and sequence statistics
// (C ABI: include/gtamd_encode.h).  gfx950 only.
//
// The reference's reader is a two-state machine over the input bytes
// (src/core/sequence_buffer_fasta.c:104-158): outside a description '>' opens
// one and emits a separator (except for the first record), white space is
// dropped, every other byte goes through the symbol map; inside a description
// only '\n' matters, it closes it.  The state after a byte is therefore the
// kind of the most recent '>' / '\n' at or before it, which makes the machine
// a max-scan:
//   k_fa_last    per 4096-byte tile: kind of the last '>' / '\n' in the tile
//   scan_u32     exclusive MAX over tiles -> state at the start of every tile
//   k_fa_tile<0> per tile: in-tile max-scan for the state of every thread's 16
//                bytes, then count symbols + separators and descriptions;
//                first illegal character, first description, histogram of the
//                original characters
//   scan_u32     exclusive SUM of the two per-tile counts
//   k_fa_tile<1> the same walk again, now writing symbols, separators and the
//                byte range of every description
// HBM traffic per input byte: 3 reads of the byte + 1 written symbol; the tile
// arrays are 1/512 of that.
//
// Statistics (encseq_charproc.gen, encseq.c:5061-5127) are run-length
// properties of four symbol classes (special, wildcard, non-special,
// non-separator).  k_run_summary folds every 4096-symbol tile into one
// summary per class with an associative merge (runs that touch a tile edge
// stay open); the host folds the ~n/4096 tile summaries with the same merge.
#include <stdarg.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "../../include/gtamd_encode.h"
#include "../../include/gtamd_esa.h"
#include "esa_prims.h"
#include "esa_devutil.h"

namespace {

constexpr int EN_THREADS = 256;
constexpr int EN_PER = 16;                       // bytes per thread
constexpr int EN_TILE = EN_THREADS * EN_PER;     // 4096
constexpr u8 LUT_BLANK = 252, LUT_UNDEF = 253;
constexpr u64 NONE64 = ~(u64) 0;

void build_lut(u8 lut[256], bool protein) {
  memset(lut, LUT_UNDEF, 256);
  // src/core/alphabet.c:84-91 (DNA), :345-356, :480-503 (protein)
  if (protein) {
    const char *letters = "LVIFKREDAGSTNQYWPHMC", *wild = "XUBZJO*-";
    for (int i = 0; letters[i]; i++) lut[(u8) letters[i]] = (u8) i;
    for (int i = 0; wild[i]; i++) lut[(u8) wild[i]] = GTAMD_WILDCARD;
  } else {
    const char *lower = "acgt", *upper = "ACGT", *wild = "nsywrkvbdhmNSYWRKVBDHM";
    for (int i = 0; i < 4; i++) lut[(u8) lower[i]] = lut[(u8) upper[i]] = (u8) i;
    lut[(u8) 'u'] = lut[(u8) 'U'] = 3;
    for (int i = 0; wild[i]; i++) lut[(u8) wild[i]] = GTAMD_WILDCARD;
  }
  // isspace() of the C locale
  lut[' '] = lut['\t'] = lut['\n'] = lut['\r'] = lut['\v'] = lut['\f'] = LUT_BLANK;
}

__device__ __forceinline__ void load16(const u8 *raw, u64 len, u64 off, u8 b[EN_PER]) {
  if (off + EN_PER <= len) {
    const uint4 v = *reinterpret_cast<const uint4 *>(raw + off);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < EN_PER; j++) b[j] = (u8) (w[j >> 2] >> (8 * (j & 3)));
  } else {
#pragma unroll
    for (int j = 0; j < EN_PER; j++) b[j] = off + j < len ? raw[off + j] : (u8) ' ';
  }
}

// kind of the last '>' / '\n' among 16 bytes: 0 none, else 2*(index in tile+1)+('>')
__device__ __forceinline__ u32 last_marker(const u8 b[EN_PER], u32 first) {
  u32 last = 0;
#pragma unroll
  for (int j = 0; j < EN_PER; j++) {
    if (b[j] == '\n') last = 2 * (first + j + 1);
    else if (b[j] == '>') last = 2 * (first + j + 1) + 1;
  }
  return last;
}

__global__ __launch_bounds__(EN_THREADS) void k_fa_last(const u8 *raw, u64 len,
                                                       u32 *tile_last) {
  __shared__ u32 lds[EN_THREADS / 64];
  const u64 tile = blockIdx.x;
  u8 b[EN_PER];
  load16(raw, len, tile * EN_TILE + (u64) threadIdx.x * EN_PER, b);
  u32 last = last_marker(b, threadIdx.x * EN_PER), total;
  (void) block_scan_excl<SCAN_MAX, EN_THREADS>(last, &total, lds);
  if (threadIdx.x == 0)
    tile_last[tile] = total == 0 ? 0 : (u32) (((tile + 1) << 1) | (total & 1));
}

struct FaGlobals {
  unsigned long long first_illegal;   // file offset of the first byte without a code
  unsigned long long first_desc;      // file offset of the first '>' that opens a description
  unsigned long long origdist[256];
};

struct FaEmit {
  u8 *enc;            // output symbols (already offset to this file's start)
  u64 *desc_start;    // already offset to this file's first description
  u64 *desc_end;
  u64 first_desc;     // as counted by the first walk (NONE64: none)
  int drop_first;     // the first description of all input emits no separator
};

// EMIT = 0: count; EMIT = 1: write
template <int EMIT>
__global__ __launch_bounds__(EN_THREADS) void k_fa_tile(
    const u8 *raw, u64 len, const u8 *lut_g, const u32 *tile_state,
    u32 *tile_syms, u32 *tile_descs, FaGlobals *g, const u32 *sym_off,
    const u32 *desc_off, FaEmit em) {
  __shared__ u32 lds[EN_THREADS / 64];
  __shared__ u8 lut[256];
  __shared__ u32 hist[256];
  const u64 tile = blockIdx.x;
  const u32 first = threadIdx.x * EN_PER;
  lut[threadIdx.x] = lut_g[threadIdx.x];
  if (!EMIT) hist[threadIdx.x] = 0;
  u8 b[EN_PER];
  load16(raw, len, tile * EN_TILE + first, b);
  u32 total;
  const u32 before = block_scan_excl<SCAN_MAX, EN_THREADS>(last_marker(b, first),
                                                           &total, lds);
  // (block_scan_excl synchronises: lut and hist are visible from here on)
  u32 state = before != 0 ? (before & 1) : (tile_state[tile] & 1);
  const u32 state0 = state;
  u32 nsym = 0, ndesc = 0;
  u64 illegal = NONE64, firstdesc = NONE64;
#pragma unroll
  for (int j = 0; j < EN_PER; j++) {
    const u8 c = b[j];
    if (state) {
      if (c == '\n') state = 0;
    } else if (c == '>') {
      state = 1;
      if (firstdesc == NONE64) firstdesc = tile * EN_TILE + first + j;
      ndesc++; nsym++;                     // a separator goes with it
    } else {
      const u8 code = lut[c];
      if (code == LUT_BLANK) continue;
      if (code == LUT_UNDEF) {
        if (illegal == NONE64) illegal = tile * EN_TILE + first + j;
        continue;
      }
      nsym++;
      if (!EMIT) atomicAdd(&hist[c], 1u);
    }
  }
  if (!EMIT) {
    u32 tot_s, tot_d;
    (void) block_scan_excl<SCAN_SUM, EN_THREADS>(nsym, &tot_s, lds);
    (void) block_scan_excl<SCAN_SUM, EN_THREADS>(ndesc, &tot_d, lds);
    if (threadIdx.x == 0) { tile_syms[tile] = tot_s; tile_descs[tile] = tot_d; }
    if (illegal != NONE64) atomicMin(&g->first_illegal, (unsigned long long) illegal);
    if (firstdesc != NONE64) atomicMin(&g->first_desc, (unsigned long long) firstdesc);
    __syncthreads();
    if (hist[threadIdx.x] != 0)
      atomicAdd(&g->origdist[threadIdx.x], (unsigned long long) hist[threadIdx.x]);
    return;
  }
  u32 dummy;
  u64 so = (u64) sym_off[tile] + block_scan_excl<SCAN_SUM, EN_THREADS>(nsym, &dummy, lds);
  u64 dn = (u64) desc_off[tile] + block_scan_excl<SCAN_SUM, EN_THREADS>(ndesc, &dummy, lds);
  state = state0;
#pragma unroll
  for (int j = 0; j < EN_PER; j++) {
    const u8 c = b[j];
    const u64 at = tile * EN_TILE + first + j;
    if (state) {
      // dn >= 1 here: a description is open
      if (c == '\n') { state = 0; em.desc_end[dn - 1] = at; }
    } else if (c == '>') {
      state = 1;
      em.desc_start[dn++] = at + 1;
      if (!(em.drop_first && at == em.first_desc))
        em.enc[so - (em.drop_first && at > em.first_desc ? 1 : 0)] = (u8) GTAMD_SEPARATOR;
      so++;
    } else {
      const u8 code = lut[c];
      if (code >= LUT_BLANK && code != GTAMD_WILDCARD) continue;   // blank or undefined
      em.enc[so - (em.drop_first && at > em.first_desc ? 1 : 0)] = code;
      so++;
    }
  }
}

// ---- run summaries ---------------------------------------------------------
// Summary of one symbol class over a stretch of symbols: runs that touch the
// left / right edge stay open (pre / suf), runs strictly inside are closed and
// counted.  A stretch consisting of members only has pre == suf == len.
template <typename T> struct RunSum {
  T len, pre, suf, members, runs, pieces[3], maxrun, minrun;
};

template <typename T> __host__ __device__ inline RunSum<T> rs_empty() {
  RunSum<T> r;
  r.len = r.pre = r.suf = r.members = r.runs = 0;
  r.pieces[0] = r.pieces[1] = r.pieces[2] = 0;
  r.maxrun = 0; r.minrun = ~(T) 0;
  return r;
}

template <typename T> __host__ __device__ inline RunSum<T> rs_single(bool member) {
  RunSum<T> r;
  r.len = 1; r.pre = r.suf = r.members = member ? 1 : 0;
  r.runs = 0; r.pieces[0] = r.pieces[1] = r.pieces[2] = 0;
  r.maxrun = 0; r.minrun = ~(T) 0;
  return r;
}

template <typename T> __host__ __device__ inline void rs_close(RunSum<T> &r, T len) {
  // stored ranges of a run in an 8-/16-/32-bit table (encseq.c:5061-5074)
  r.runs++;
  r.pieces[0] += (len + 255) / 256;
  r.pieces[1] += (len + 65535) / 65536;
  r.pieces[2] += 1;
  if (len > r.maxrun) r.maxrun = len;
  if (len < r.minrun) r.minrun = len;
}

template <typename T>
__host__ __device__ inline RunSum<T> rs_merge(const RunSum<T> &a, const RunSum<T> &b) {
  if (a.len == 0) return b;
  if (b.len == 0) return a;
  RunSum<T> r;
  r.len = a.len + b.len; r.members = a.members + b.members;
  r.runs = a.runs + b.runs;
  for (int k = 0; k < 3; k++) r.pieces[k] = a.pieces[k] + b.pieces[k];
  r.maxrun = a.maxrun > b.maxrun ? a.maxrun : b.maxrun;
  r.minrun = a.minrun < b.minrun ? a.minrun : b.minrun;
  const bool a_all = a.pre == a.len, b_all = b.pre == b.len;
  if (a_all && b_all) r.pre = r.suf = r.len;
  else if (a_all) { r.pre = a.len + b.pre; r.suf = b.suf; }
  else if (b_all) { r.pre = a.pre; r.suf = a.suf + b.len; }
  else {
    r.pre = a.pre; r.suf = b.suf;
    if (a.suf + b.pre > 0) rs_close<T>(r, a.suf + b.pre);
  }
  return r;
}

enum { CL_SPECIAL = 0, CL_WILDCARD, CL_NONSPECIAL, CL_NONSEPARATOR, CL_COUNT };

__device__ __forceinline__ bool in_class(int cl, u8 c) {
  return cl == CL_SPECIAL ? c >= GTAMD_WILDCARD
       : cl == CL_WILDCARD ? c == GTAMD_WILDCARD
       : cl == CL_NONSPECIAL ? c < GTAMD_WILDCARD : c != GTAMD_SEPARATOR;
}

__global__ __launch_bounds__(EN_THREADS) void k_run_summary(
    const u8 *enc, u64 n, RunSum<u32> *tiles, unsigned long long *chardist) {
  __shared__ RunSum<u32> red[EN_THREADS];
  __shared__ u32 hist[32];
  __shared__ u32 anyspecial;
  const u64 tile = blockIdx.x, base = tile * EN_TILE + (u64) threadIdx.x * EN_PER;
  if (threadIdx.x < 32) hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) anyspecial = 0;
  __syncthreads();
  u8 b[EN_PER];
  int cnt = 0;
  if (base + EN_PER <= n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(enc + base);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < EN_PER; j++) b[j] = (u8) (w[j >> 2] >> (8 * (j & 3)));
    cnt = EN_PER;
  } else {
    for (int j = 0; j < EN_PER; j++)
      if (base + j < n) { b[j] = enc[base + j]; cnt = j + 1; }
  }
  bool special = false;
  for (int j = 0; j < cnt; j++) {
    if (b[j] >= GTAMD_WILDCARD) special = true;
    else atomicAdd(&hist[b[j] & 31], 1u);
  }
  if (special) anyspecial = 1;
  __syncthreads();
  if (threadIdx.x < 32 && hist[threadIdx.x] != 0)
    atomicAdd(&chardist[threadIdx.x], (unsigned long long) hist[threadIdx.x]);
  const u64 tile_len64 = n - tile * EN_TILE < EN_TILE ? n - tile * EN_TILE : EN_TILE;
  const u32 tile_len = (u32) tile_len64;
  if (!anyspecial) {
    // the common tile: letters only -- no special, no wildcard, one open run of
    // non-specials and non-separators
    if (threadIdx.x < CL_COUNT) {
      RunSum<u32> r = rs_single<u32>(threadIdx.x >= CL_NONSPECIAL);
      r.len = tile_len;
      r.pre = r.suf = r.members = threadIdx.x >= CL_NONSPECIAL ? tile_len : 0;
      tiles[tile * CL_COUNT + threadIdx.x] = r;
    }
    return;
  }
  for (int cl = 0; cl < CL_COUNT; cl++) {
    RunSum<u32> r = rs_empty<u32>();
    for (int j = 0; j < cnt; j++) r = rs_merge<u32>(r, rs_single<u32>(in_class(cl, b[j])));
    red[threadIdx.x] = r;
    __syncthreads();
    for (int stride = 1; stride < EN_THREADS; stride <<= 1) {
      RunSum<u32> m;
      const bool act = (threadIdx.x & (2 * stride - 1)) == 0;
      if (act) m = rs_merge<u32>(red[threadIdx.x], red[threadIdx.x + stride]);
      __syncthreads();
      if (act) red[threadIdx.x] = m;
      __syncthreads();
    }
    if (threadIdx.x == 0) tiles[tile * CL_COUNT + cl] = red[0];
    __syncthreads();
  }
}

// first position of an empty sequence: a separator at either end of the
// sequence or behind another separator
__global__ void k_empty_sequence(const u8 *enc, u64 n, unsigned long long *first) {
  const u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || enc[i] != GTAMD_SEPARATOR) return;
  if (i == 0 || i + 1 == n || enc[i - 1] == GTAMD_SEPARATOR)
    atomicMin(first, (unsigned long long) i);
}

// ---- sections of INDEX.esq from the symbols in HBM -------------------------
// (layouts: src/core/encseq.c:85-99, 2594-2607, 2771-2835, 2324-2447)

// two bits per symbol, 32 symbols per word, first symbol in the top bits;
// specials: bit access stores wildcard = 0 / separator = 1, the other access
// types store the least frequent letter
__global__ void k_esq_twobit(const u8 *enc, u64 n, u64 units, int bitaccess,
                             u32 fill, u64 *words) {
  const u64 w = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= units) return;
  const u64 base = w * 32;
  u64 v = 0;
  if (base + 32 <= n) {
    const uint4 *p = reinterpret_cast<const uint4 *>(enc + base);
    const uint4 q[2] = {p[0], p[1]};
    const u32 *x = reinterpret_cast<const u32 *>(q);
#pragma unroll
    for (int j = 0; j < 32; j++) {
      const u32 c = (x[j >> 2] >> (8 * (j & 3))) & 255;
      const u64 code = c < GTAMD_WILDCARD ? c
                     : bitaccess ? (c == GTAMD_SEPARATOR ? 1u : 0u) : fill;
      v |= code << (62 - 2 * j);
    }
  } else {
    for (int j = 0; j < 32 && base + j < n; j++) {
      const u32 c = enc[base + j];
      const u64 code = c < GTAMD_WILDCARD ? c
                     : bitaccess ? (c == GTAMD_SEPARATOR ? 1u : 0u) : fill;
      v |= code << (62 - 2 * j);
    }
  }
  words[w] = v;
}

// one bit per position, first position in the top bit; the 64 positions behind
// the sequence are set as well
__global__ void k_esq_specialbits(const u8 *enc, u64 n, u64 units, u64 *words) {
  const u64 w = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= units) return;
  const u64 base = w * 64;
  u64 v = 0;
  for (int j = 0; j < 64; j++) {
    const u64 pos = base + j;
    const bool bit = pos < n ? enc[pos] >= GTAMD_WILDCARD : pos < n + 64;
    v |= (u64) bit << (63 - j);
  }
  words[w] = v;
}

// bits per symbol, most significant bit first: 8 symbols -> `bits` bytes;
// wildcard = sigma, separator = sigma + 1
__global__ void k_esq_bitpack(const u8 *enc, u64 n, u32 sigma, u32 bits, u64 nbytes,
                              u8 *out) {
  const u64 t = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (t * 8 >= n) return;
  u64 v = 0;
  for (int j = 0; j < 8; j++) {
    const u64 pos = t * 8 + j;
    u32 c = pos < n ? enc[pos] : 0;
    if (pos < n && c >= GTAMD_WILDCARD) c = c == GTAMD_WILDCARD ? sigma : sigma + 1;
    v = (v << bits) | c;
  }
  for (u32 k = 0; k < bits; k++) {
    const u64 at = t * bits + k;
    if (at < nbytes) out[at] = (u8) (v >> (8 * (bits - 1 - k)));
  }
}

enum { POS_WILDCARD_START = 0, POS_WILDCARD_END, POS_SEPARATOR };

// positions of one kind in increasing order: count per tile, then (after a scan
// of the counts) write
template <int EMIT>
__global__ __launch_bounds__(EN_THREADS) void k_positions(
    const u8 *enc, u64 n, int kind, u32 *tile_count, const u32 *tile_off, u64 *out) {
  __shared__ u32 lds[EN_THREADS / 64];
  const u64 tile = blockIdx.x, base = tile * EN_TILE + (u64) threadIdx.x * EN_PER;
  u8 b[EN_PER + 2];          // b[0] = symbol before, b[EN_PER + 1] = symbol behind
  for (int j = 0; j < EN_PER + 2; j++) {
    const u64 pos = base + j;             // position + 1
    b[j] = pos >= 1 && pos - 1 < n ? enc[pos - 1] : (u8) 0;
  }
  u32 flags = 0, cnt = 0;
  for (int j = 0; j < EN_PER; j++) {
    if (base + j >= n) break;
    const u8 c = b[j + 1];
    const bool hit = kind == POS_SEPARATOR ? c == GTAMD_SEPARATOR
                   : kind == POS_WILDCARD_START ? c == GTAMD_WILDCARD && b[j] != GTAMD_WILDCARD
                   : c == GTAMD_WILDCARD && b[j + 2] != GTAMD_WILDCARD;
    if (hit) { flags |= 1u << j; cnt++; }
  }
  u32 total;
  const u32 before = block_scan_excl<SCAN_SUM, EN_THREADS>(cnt, &total, lds);
  if (!EMIT) {
    if (threadIdx.x == 0) tile_count[tile] = total;
    return;
  }
  u64 at = (u64) tile_off[tile] + before;
  for (int j = 0; j < EN_PER; j++)
    if (flags & (1u << j)) out[at++] = base + j;
}

// ---------------------------------------------------------------------------
// FASTQ on the device (src/core/seq_iterator_fastq.c:96-305, the block grammar):
// the STRICT four-line form only -- "@name", one line of symbols, "+[name]", one
// line of as many quality characters, every line ended by '\n', no blanks in the
// symbol and quality lines.  That is what sequencers write, and it is the form
// in which a record's parts are known from the line number alone: newlines are
// counted per tile, a scan gives every byte its line, line mod 4 its role.
// Anything else -- sequences or qualities over several lines, a missing last
// newline, an illegal symbol, lengths that differ -- makes the walk say "not for
// me" (FqGlobals.bad), and the host reader, which has the reference's messages for
// all of that, takes the input (gtamd_encoder_declined).
// ---------------------------------------------------------------------------
struct FqGlobals {
  unsigned long long bad;              // != 0: not the strict form / an illegal symbol
  unsigned long long origdist[256];    // the original characters of the sequences
};

// walk of a tile: EMIT 0 newlines per tile; 1 (line numbers known) sequence symbols
// per tile, the checks, the histogram; 2 the symbols and separators
struct FqEmit {
  u8 *enc;            // where this file's output starts
  int seen_record;    // a record (of an earlier file) precedes this file's first
};
template <int EMIT>
__global__ __launch_bounds__(EN_THREADS) void k_fq_tile(
    const u8 *raw, u64 len, const u8 *lut, const u32 *t_nl, const u32 *t_sym, u32 *t_out,
    FqGlobals *g, FqEmit em) {
  __shared__ u32 lds[EN_THREADS / 64];
  __shared__ u32 s_hist[256];
  const u64 tile = blockIdx.x, first = tile * EN_TILE + (u64) threadIdx.x * EN_PER;
  u8 b[EN_PER];
  u32 nnl = 0;
  if (EMIT == 1) { s_hist[threadIdx.x] = 0; __syncthreads(); }
#pragma unroll
  for (int j = 0; j < EN_PER; j++) {
    b[j] = first + j < len ? raw[first + j] : (u8) 0;
    nnl += first + j < len && b[j] == '\n';
  }
  u32 total;
  const u32 nlbefore = block_scan_excl<SCAN_SUM, EN_THREADS>(nnl, &total, lds);
  if (EMIT == 0) {
    if (threadIdx.x == 0) t_out[tile] = total;
    return;
  }
  u64 line = (u64) t_nl[tile] + nlbefore;
  u32 nsym = 0, bad = 0;
#pragma unroll
  for (int j = 0; j < EN_PER; j++) {
    const u8 c = b[j];
    if (first + j >= len) continue;
    if (c == '\n') { line++; continue; }
    if (c == '\r') bad = 1;          // (what a carriage return means differs by line)
    const u32 role = (u32) line & 3u;
    if (role == 1u) {
      const u8 code = lut[c];
      if (code >= LUT_BLANK && code != GTAMD_WILDCARD) bad = 1;      // blank, or no symbol
      else { nsym++; if (EMIT == 1) atomicAdd(&s_hist[c], 1u); }
    } else if (role == 3u && c == ' ') bad = 1;
  }
  u32 tsym;
  const u32 symbefore = block_scan_excl<SCAN_SUM, EN_THREADS>(nsym, &tsym, lds);
  if (EMIT == 1) {
    if (threadIdx.x == 0) t_out[tile] = tsym;
    if (bad) atomicOr(&g->bad, 1ull);
    __syncthreads();
    if (s_hist[threadIdx.x]) atomicAdd(&g->origdist[threadIdx.x], (unsigned long long) s_hist[threadIdx.x]);
    return;
  }
  // the symbols of record r lie behind those of the records before and the
  // separators between them (one in front of every record but the first of all)
  u64 so = (u64) t_sym[tile] + symbefore;
  line = (u64) t_nl[tile] + nlbefore;
#pragma unroll
  for (int j = 0; j < EN_PER; j++) {
    const u8 c = b[j];
    if (first + j >= len) continue;
    const u64 r = line >> 2;
    const bool linestart = first + j == 0 || (j ? b[j - 1] : raw[first - 1]) == '\n';
    if (linestart && (line & 3u) == 0 && (r > 0 || em.seen_record))
      em.enc[so + (em.seen_record ? r : r - 1)] = (u8) GTAMD_SEPARATOR;
    if (c == '\n') { line++; continue; }
    if ((line & 3u) == 1u) {
      em.enc[so + (em.seen_record ? r + 1 : r)] = lut[c];
      so++;
    }
  }
}
// positions of the newlines, in order (counts and offsets of k_fq_tile<0>)
__global__ __launch_bounds__(EN_THREADS) void k_fq_newlines(const u8 *raw, u64 len,
                                                            const u32 *t_nl, u32 *nl) {
  __shared__ u32 lds[EN_THREADS / 64];
  const u64 tile = blockIdx.x, first = tile * EN_TILE + (u64) threadIdx.x * EN_PER;
  u32 mask = 0;
#pragma unroll
  for (int j = 0; j < EN_PER; j++)
    if (first + j < len && raw[first + j] == '\n') mask |= 1u << j;
  u32 total;
  u32 at = t_nl[tile] + block_scan_excl<SCAN_SUM, EN_THREADS>((u32) __popc(mask), &total, lds);
  for (int j = 0; j < EN_PER; j++)
    if ((mask >> j) & 1u) nl[at++] = (u32) (first + j);
}
// one thread per record: '@' and '+' where they belong, the '+' line empty or the
// name again, as many qualities as symbols, no empty sequence; the description's
// bytes, and the record's lengths (for the file length table)
__global__ __launch_bounds__(256) void k_fq_records(const u8 *raw, const u32 *nl, u64 nrec,
                                                    u64 *desc_start, u64 *desc_end,
                                                    u32 *seqlen, u32 *desclen, FqGlobals *g) {
  const u64 k = (u64) blockIdx.x * 256 + threadIdx.x;
  if (k >= nrec) return;
  const u64 s0 = k ? (u64) nl[4 * k - 1] + 1 : 0, e0 = nl[4 * k], s1 = e0 + 1, e1 = nl[4 * k + 1],
            s2 = e1 + 1, e2 = nl[4 * k + 2], s3 = e2 + 1, e3 = nl[4 * k + 3];
  bool ok = e0 > s0 && raw[s0] == '@' && e2 > s2 && raw[s2] == '+' && e1 > s1 && e3 - s3 == e1 - s1;
  if (ok && e2 - s2 > 1) {
    ok = e2 - s2 == e0 - s0;
    for (u64 i = 1; ok && i < e0 - s0; i++) ok = raw[s0 + i] == raw[s2 + i];
  }
  if (!ok) atomicOr(&g->bad, 2ull);
  desc_start[k] = s0 + 1;
  desc_end[k] = e0;
  seqlen[k] = (u32) (e1 - s1);
  desclen[k] = (u32) (e0 - s0 - 1);
}

struct InputFile {
  std::string name;
  const u8 *bytes;
  u64 length;
  u64 out_start, out_len, ndesc, desc_base;   // filled by finish
  bool fastq;
};

template <typename T> int dev_alloc(T **p, u64 count) {
  *p = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(p), (count ? count : 1) * sizeof(T)) != hipSuccess) {
    gtamd_set_error("cannot allocate %llu bytes of device memory for the encoder",
                    (unsigned long long) (count * sizeof(T)));
    return -1;
  }
  return 0;
}

}  // namespace

struct gtamd_encoder {
  int device;
  bool finished;
  u8 lut[256];             // code per input byte, LUT_UNDEF, LUT_BLANK
  u32 sigma, packbits;     // alphabet size; bits per symbol of the bit packing
  std::vector<InputFile> files;
  hipStream_t st;
  hipEvent_t ev[4];
  u8 *d_enc;
  u64 n, cap_enc;
  u64 *d_desc_start, *d_desc_end;
  u64 ndesc, cap_desc;
  gtamd_encode_summary sum;
  float total_ms, parse_ms, stats_ms;
  u64 input_bytes;
  // FASTQ input: a record's sequence and description lengths (the file length
  // table of the reference's FASTQ reader is made from them), in input order
  std::vector<u32> rec_seqlen, rec_desclen, rec_file;
  bool declined;           // the last finish met FASTQ input the device reader does not take
};

static void enc_free(gtamd_encoder *e) {
  if (e->d_enc) (void) hipFree(e->d_enc);
  if (e->d_desc_start) (void) hipFree(e->d_desc_start);
  if (e->d_desc_end) (void) hipFree(e->d_desc_end);
  e->d_enc = nullptr; e->d_desc_start = e->d_desc_end = nullptr;
  e->cap_enc = e->cap_desc = 0;
}

extern "C" gtamd_encoder *gtamd_encoder_create(int device, int protein) {
  GTAMD_ABI_BEGIN
  u8 lut[256];
  build_lut(lut, protein != 0);
  for (int c = 0; c < 256; c++) if (lut[c] == LUT_BLANK) lut[c] = LUT_UNDEF;
  return gtamd_encoder_create_map(device, lut, protein ? 20 : 4, protein ? 5 : 3);
  GTAMD_ABI_END(nullptr)
}

extern "C" gtamd_encoder *gtamd_encoder_create_map(int device, const uint8_t *symbolmap,
                                                   uint32_t numofchars,
                                                   unsigned bitspersymbol) {
  GTAMD_ABI_BEGIN
  int count = 0;
  if (symbolmap == nullptr || numofchars < 1 || numofchars > 32 || bitspersymbol < 1 ||
      bitspersymbol > 8) {
    gtamd_set_error("invalid alphabet for the device encoder (%u letters, %u bits)",
                    numofchars, bitspersymbol);
    return nullptr;
  }
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
    gtamd_set_error("no HIP device %d available (this library has no CPU fallback)",
                    device);
    return nullptr;
  }
  gtamd_encoder *e = new gtamd_encoder();
  e->device = device; e->finished = false;
  e->sigma = numofchars; e->packbits = bitspersymbol;
  for (int c = 0; c < 256; c++) {
    const u8 v = symbolmap[c];
    e->lut[c] = v < numofchars || v == GTAMD_WILDCARD ? v : LUT_UNDEF;
  }
  // isspace() of the C locale is skipped between symbols
  e->lut[' '] = e->lut['\t'] = e->lut['\n'] = e->lut['\r'] = e->lut['\v'] = e->lut['\f'] = LUT_BLANK;
  e->d_enc = nullptr; e->d_desc_start = e->d_desc_end = nullptr;
  e->n = e->cap_enc = e->ndesc = e->cap_desc = 0;
  e->total_ms = e->parse_ms = e->stats_ms = 0; e->input_bytes = 0;
  memset(&e->sum, 0, sizeof e->sum);
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&e->st) != hipSuccess) {
    gtamd_set_error("cannot create a stream on device %d", device);
    delete e;
    return nullptr;
  }
  for (auto &ev : e->ev) (void) hipEventCreate(&ev);
  return e;
  GTAMD_ABI_END(nullptr)
}

extern "C" void gtamd_encoder_destroy(gtamd_encoder *e) {
  if (e == nullptr) return;
  (void) hipSetDevice(e->device);
  enc_free(e);
  for (auto &ev : e->ev) (void) hipEventDestroy(ev);
  (void) hipStreamDestroy(e->st);
  delete e;
}

extern "C" int gtamd_encoder_add_file(gtamd_encoder *e, const char *name,
                                      const uint8_t *bytes, uint64_t length) {
  GTAMD_ABI_BEGIN
  if (e == nullptr || name == nullptr || (bytes == nullptr && length > 0)) {
    gtamd_set_error("invalid argument to gtamd_encoder_add_file");
    return -1;
  }
  if (length >= ((u64) 1 << 32) - EN_TILE) {
    gtamd_set_error("file '%s' is too large for the device encoder (%llu bytes, "
                    "limit 4 GiB per file)", name, (unsigned long long) length);
    return -1;
  }
  InputFile f;
  f.name = name; f.bytes = bytes; f.length = length;
  f.out_start = f.out_len = f.ndesc = f.desc_base = 0;
  f.fastq = length > 0 && bytes[0] == '@';
  e->files.push_back(f);
  e->finished = false;
  return 0;
  GTAMD_ABI_END(-1)
}

// descriptions seen so far do not fit: grow both arrays, keep the contents
static int grow_desc(gtamd_encoder *e, u64 need) {
  if (need <= e->cap_desc) return 0;
  const u64 cap = need + need / 2 + 1024;
  u64 *ns, *ne;
  TRY(dev_alloc(&ns, cap));
  TRY(dev_alloc(&ne, cap));
  if (e->ndesc > 0) {
    HIP_TRY(hipMemcpyAsync(ns, e->d_desc_start, e->ndesc * 8, hipMemcpyDeviceToDevice, e->st));
    HIP_TRY(hipMemcpyAsync(ne, e->d_desc_end, e->ndesc * 8, hipMemcpyDeviceToDevice, e->st));
  }
  HIP_TRY(hipStreamSynchronize(e->st));
  if (e->d_desc_start) (void) hipFree(e->d_desc_start);
  if (e->d_desc_end) (void) hipFree(e->d_desc_end);
  e->d_desc_start = ns; e->d_desc_end = ne; e->cap_desc = cap;
  return 0;
}

static u64 count_lines(const u8 *bytes, u64 upto) {
  u64 line = 1;
  for (const u8 *p = bytes, *end = bytes + upto;
       (p = static_cast<const u8 *>(memchr(p, '\n', (size_t) (end - p)))) != nullptr; p++)
    line++;
  return line;
}

template <typename T> static RunSum<u64> widen(const RunSum<T> &a) {
  RunSum<u64> r;
  r.len = a.len; r.pre = a.pre; r.suf = a.suf; r.members = a.members; r.runs = a.runs;
  for (int k = 0; k < 3; k++) r.pieces[k] = a.pieces[k];
  r.maxrun = a.maxrun; r.minrun = a.minrun == ~(T) 0 ? ~(u64) 0 : (u64) a.minrun;
  return r;
}

static int encode_files(gtamd_encoder *e, u8 *d_raw, u8 *d_lut, u32 *d_tile,
                        u32 *d_ws, FaGlobals *d_glob, u64 max_tiles) {
  bool seen_record = false;
  u32 *t_last = d_tile, *t_state = d_tile + max_tiles, *t_syms = d_tile + 2 * max_tiles,
      *t_descs = d_tile + 3 * max_tiles, *t_soff = d_tile + 4 * max_tiles,
      *t_doff = d_tile + 5 * max_tiles;
  e->n = 0; e->ndesc = 0;
  for (size_t fi = 0; fi < e->files.size(); fi++) {
    InputFile &f = e->files[fi];
    const u64 ntiles = div_up(f.length, EN_TILE);
    f.out_start = e->n; f.out_len = 0; f.ndesc = 0; f.desc_base = e->ndesc;
    if (ntiles == 0) continue;
    HIP_TRY(hipMemcpyAsync(d_raw, f.bytes, f.length, hipMemcpyHostToDevice, e->st));
    if (f.fastq) {
      // ---- the strict four-line form, or not for the device
      FqGlobals *d_fq = nullptr;
      u32 *d_nl = nullptr, *d_rs = nullptr, *d_rd = nullptr;
      auto decline = [&]() -> int {
        if (d_fq) (void) hipFree(d_fq);
        if (d_nl) (void) hipFree(d_nl);
        if (d_rs) (void) hipFree(d_rs);
        if (d_rd) (void) hipFree(d_rd);
        e->declined = true;
        gtamd_set_error("FASTQ file '%s' is not in the strict four-line form the device reader "
                        "takes (the host reader reads it)", f.name.c_str());
        return -1;
      };
      if (f.bytes[f.length - 1] != '\n') return decline();
      TRY(dev_alloc(&d_fq, 1));
      HIP_TRY(hipMemsetAsync(d_fq, 0, sizeof(FqGlobals), e->st));
      u32 *t_nlc = t_last, *t_nlo = t_state;
      k_fq_tile<0><<<(u32) ntiles, EN_THREADS, 0, e->st>>>(d_raw, f.length, d_lut, nullptr, nullptr, t_nlc,
                                                          d_fq, FqEmit());
      HIP_TRY(hipGetLastError());
      TRY(scan_u32(SCAN_SUM, t_nlc, t_nlo, ntiles, false, d_ws, e->st));
      k_fq_tile<1><<<(u32) ntiles, EN_THREADS, 0, e->st>>>(d_raw, f.length, d_lut, t_nlo, nullptr, t_syms,
                                                          d_fq, FqEmit());
      HIP_TRY(hipGetLastError());
      TRY(scan_u32(SCAN_SUM, t_syms, t_soff, ntiles, false, d_ws, e->st));
      u32 last[4];
      FqGlobals got;
      HIP_TRY(hipStreamSynchronize(e->st));
      HIP_TRY(hipMemcpy(&got, d_fq, sizeof got, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(&last[0], t_nlo + ntiles - 1, 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(&last[1], t_nlc + ntiles - 1, 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(&last[2], t_soff + ntiles - 1, 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(&last[3], t_syms + ntiles - 1, 4, hipMemcpyDeviceToHost));
      const u64 nlines = (u64) last[0] + last[1], nsym = (u64) last[2] + last[3];
      if (got.bad != 0 || nlines == 0 || (nlines & 3) != 0) return decline();
      const u64 nrec = nlines / 4;
      if (dev_alloc(&d_nl, nlines) != 0 || dev_alloc(&d_rs, nrec) != 0 || dev_alloc(&d_rd, nrec) != 0) {
        (void) decline();
        e->declined = false;
        gtamd_set_error("cannot allocate device memory for the %llu records of '%s'",
                        (unsigned long long) nrec, f.name.c_str());
        return -1;
      }
      k_fq_newlines<<<(u32) ntiles, EN_THREADS, 0, e->st>>>(d_raw, f.length, t_nlo, d_nl);
      HIP_TRY(hipGetLastError());
      TRY(grow_desc(e, e->ndesc + nrec));
      k_fq_records<<<(u32) div_up(nrec, 256), 256, 0, e->st>>>(d_raw, d_nl, nrec, e->d_desc_start + e->ndesc,
                                                             e->d_desc_end + e->ndesc, d_rs, d_rd, d_fq);
      HIP_TRY(hipGetLastError());
      FqEmit em;
      em.enc = e->d_enc + e->n;
      em.seen_record = seen_record ? 1 : 0;
      k_fq_tile<2><<<(u32) ntiles, EN_THREADS, 0, e->st>>>(d_raw, f.length, d_lut, t_nlo, t_soff, nullptr,
                                                          d_fq, em);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(e->st));
      HIP_TRY(hipMemcpy(&got, d_fq, 8, hipMemcpyDeviceToHost));
      if (got.bad != 0) return decline();
      const size_t r0 = e->rec_seqlen.size();
      e->rec_seqlen.resize(r0 + nrec); e->rec_desclen.resize(r0 + nrec); e->rec_file.resize(r0 + nrec, (u32) fi);
      HIP_TRY(hipMemcpy(e->rec_seqlen.data() + r0, d_rs, nrec * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(e->rec_desclen.data() + r0, d_rd, nrec * 4, hipMemcpyDeviceToHost));
      (void) hipFree(d_fq); (void) hipFree(d_nl); (void) hipFree(d_rs); (void) hipFree(d_rd);
      for (int c = 0; c < 256; c++) e->sum.originaldistribution[c] += got.origdist[c];
      f.out_len = nsym + nrec - (seen_record ? 0 : 1);
      f.ndesc = nrec;
      e->n += f.out_len;
      e->ndesc += nrec;
      seen_record = true;
      continue;
    }
    FaGlobals init;
    memset(&init, 0, sizeof init);
    init.first_illegal = init.first_desc = NONE64;
    HIP_TRY(hipMemcpyAsync(d_glob, &init, sizeof init, hipMemcpyHostToDevice, e->st));
    k_fa_last<<<(u32) ntiles, EN_THREADS, 0, e->st>>>(d_raw, f.length, t_last);
    HIP_TRY(hipGetLastError());
    TRY(scan_u32(SCAN_MAX, t_last, t_state, ntiles, false, d_ws, e->st));
    k_fa_tile<0><<<(u32) ntiles, EN_THREADS, 0, e->st>>>(
        d_raw, f.length, d_lut, t_state, t_syms, t_descs, d_glob, nullptr, nullptr,
        FaEmit());
    HIP_TRY(hipGetLastError());
    TRY(scan_u32(SCAN_SUM, t_syms, t_soff, ntiles, false, d_ws, e->st));
    TRY(scan_u32(SCAN_SUM, t_descs, t_doff, ntiles, false, d_ws, e->st));
    FaGlobals got;
    u32 last[4];   // last tile: offsets and counts
    // (blocking copies: the destinations are on this stack frame)
    HIP_TRY(hipStreamSynchronize(e->st));
    HIP_TRY(hipMemcpy(&got, d_glob, sizeof got, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last[0], t_soff + ntiles - 1, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last[1], t_syms + ntiles - 1, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last[2], t_doff + ntiles - 1, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last[3], t_descs + ntiles - 1, 4, hipMemcpyDeviceToHost));
    if (got.first_illegal != NONE64) {
      // wording of src/core/sequence_buffer_inline.h:37-41
      gtamd_set_error("illegal character '%c': file \"%s\", line %llu",
                      f.bytes[got.first_illegal], f.name.c_str(),
                      (unsigned long long) count_lines(f.bytes, got.first_illegal));
      return -1;
    }
    const u64 emitted = (u64) last[0] + last[1], ndesc = (u64) last[2] + last[3];
    const bool drop = !seen_record && got.first_desc != NONE64;
    for (int c = 0; c < 256; c++) e->sum.originaldistribution[c] += got.origdist[c];
    TRY(grow_desc(e, e->ndesc + ndesc));
    if (ndesc > 0)
      HIP_TRY(hipMemsetAsync(e->d_desc_end + e->ndesc, 0xff, ndesc * 8, e->st));
    FaEmit em;
    em.enc = e->d_enc + e->n;
    em.desc_start = e->d_desc_start + e->ndesc;
    em.desc_end = e->d_desc_end + e->ndesc;
    em.first_desc = got.first_desc;
    em.drop_first = drop ? 1 : 0;
    k_fa_tile<1><<<(u32) ntiles, EN_THREADS, 0, e->st>>>(
        d_raw, f.length, d_lut, t_state, nullptr, nullptr, nullptr, t_soff, t_doff, em);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->st));   // d_raw is reused by the next file
    f.out_len = emitted - (drop ? 1 : 0);
    f.ndesc = ndesc;
    e->n += f.out_len;
    e->ndesc += ndesc;
    if (ndesc > 0) seen_record = true;
  }
  if (!seen_record) {
    // src/core/sequence_buffer_fasta.c:163-168
    gtamd_set_error("no sequences in multiple fasta file(s) %s ...",
                    e->files.empty() ? "" : e->files[0].name.c_str());
    return -1;
  }
  return 0;
}

static int summarise(gtamd_encoder *e) {
  const u64 n = e->n, ntiles = div_up(n, EN_TILE);
  if (n == 0) {
    gtamd_set_error("file '%s' contains an empty sequence",
                    e->files.empty() ? "(symbols)" : e->files.back().name.c_str());
    return -1;
  }
  RunSum<u32> *d_tiles;
  unsigned long long *d_small;     // 32 counters + first empty sequence
  TRY(dev_alloc(&d_tiles, ntiles * CL_COUNT));
  if (dev_alloc(&d_small, 33) != 0) { (void) hipFree(d_tiles); return -1; }
  std::vector<RunSum<u32>> tiles(ntiles * CL_COUNT);
  unsigned long long small[33];
  memset(small, 0, sizeof small);
  small[32] = NONE64;
  int rc = 0;
  do {
    if (hipMemcpyAsync(d_small, small, sizeof small, hipMemcpyHostToDevice, e->st) != hipSuccess) { rc = -1; break; }
    k_empty_sequence<<<(u32) div_up(n, 256), 256, 0, e->st>>>(e->d_enc, n, d_small + 32);
    k_run_summary<<<(u32) ntiles, EN_THREADS, 0, e->st>>>(e->d_enc, n, d_tiles, d_small);
    if (hipGetLastError() != hipSuccess) { rc = -1; break; }
    // blocking copies: the destinations are pageable (a vector, the stack) and
    // gone when this function returns
    if (hipStreamSynchronize(e->st) != hipSuccess) { rc = -1; break; }
    if (hipMemcpy(tiles.data(), d_tiles, tiles.size() * sizeof(RunSum<u32>),
                  hipMemcpyDeviceToHost) != hipSuccess) { rc = -1; break; }
    if (hipMemcpy(small, d_small, sizeof small, hipMemcpyDeviceToHost) != hipSuccess) { rc = -1; break; }
  } while (0);
  (void) hipStreamSynchronize(e->st);   // on every exit: nothing in flight behind this frame
  (void) hipFree(d_tiles);
  (void) hipFree(d_small);
  if (rc != 0) { gtamd_set_error("device statistics of the encoded sequence failed"); return -1; }
  if (small[32] != NONE64) {
    // the file whose '>' produced the offending separator (or the last one)
    if (e->files.empty()) {
      gtamd_set_error("the sequence contains an empty sequence");
      return -1;
    }
    size_t fi = e->files.size() - 1;
    for (size_t k = 0; k < e->files.size(); k++)
      if (small[32] >= e->files[k].out_start &&
          small[32] < e->files[k].out_start + e->files[k].out_len) { fi = k; break; }
    if (small[32] + 1 == n) fi = e->files.size() - 1;
    gtamd_set_error("file '%s' contains an empty sequence", e->files[fi].name.c_str());
    return -1;
  }
  RunSum<u64> tot[CL_COUNT];
  for (int cl = 0; cl < CL_COUNT; cl++) tot[cl] = rs_empty<u64>();
  for (u64 t = 0; t < ntiles; t++)
    for (int cl = 0; cl < CL_COUNT; cl++)
      tot[cl] = rs_merge<u64>(tot[cl], widen(tiles[t * CL_COUNT + cl]));
  // the runs at the two ends of the whole sequence are closed now
  u64 pre[CL_COUNT], suf[CL_COUNT];
  for (int cl = 0; cl < CL_COUNT; cl++) {
    RunSum<u64> &r = tot[cl];
    pre[cl] = r.pre; suf[cl] = r.suf;
    if (r.len > 0 && r.pre == r.len) rs_close<u64>(r, r.len);
    else {
      if (r.pre > 0) rs_close<u64>(r, r.pre);
      if (r.suf > 0) rs_close<u64>(r, r.suf);
    }
  }
  gtamd_encode_summary &s = e->sum;
  s.totallength = n;
  s.numofsequences = tot[CL_NONSEPARATOR].runs;
  s.specialcharacters = tot[CL_SPECIAL].members;
  s.realspecialranges = tot[CL_SPECIAL].runs;
  s.wildcards = tot[CL_WILDCARD].members;
  s.realwildcardranges = tot[CL_WILDCARD].runs;
  for (int k = 0; k < 3; k++) {
    s.specialrangestab[k] = tot[CL_SPECIAL].pieces[k];
    s.wildcardrangestab[k] = tot[CL_WILDCARD].pieces[k];
  }
  s.lengthofspecialprefix = pre[CL_SPECIAL]; s.lengthofspecialsuffix = suf[CL_SPECIAL];
  s.lengthofwildcardprefix = pre[CL_WILDCARD]; s.lengthofwildcardsuffix = suf[CL_WILDCARD];
  s.lengthoflongestnonspecial = tot[CL_NONSPECIAL].maxrun;
  s.minseqlen = tot[CL_NONSEPARATOR].minrun;
  s.maxseqlen = tot[CL_NONSEPARATOR].maxrun;
  s.equallength = s.minseqlen == s.maxseqlen && s.wildcards == 0 ? 1 : 0;
  for (int c = 0; c < 32; c++) s.characterdistribution[c] = small[c];
  return 0;
}

extern "C" int gtamd_encoder_finish(gtamd_encoder *e) {
  GTAMD_ABI_BEGIN
  if (e == nullptr) { gtamd_set_error("null encoder"); return -1; }
  if (e->files.empty()) { gtamd_set_error("option \"-db\" is mandatory"); return -1; }
  HIP_TRY(hipSetDevice(e->device));
  u64 total = 0, longest = 0;
  for (const InputFile &f : e->files) {
    total += f.length;
    if (f.length > longest) longest = f.length;
  }
  enc_free(e);
  e->finished = false;
  e->declined = false;
  e->rec_seqlen.clear(); e->rec_desclen.clear(); e->rec_file.clear();
  // (the device reader takes FASTA files or FASTQ files, not both in one run: the
  // reference's rules at the seam of the two formats stay with the host reader)
  {
    size_t nq = 0;
    for (const InputFile &f : e->files) nq += f.fastq;
    if (nq != 0 && nq != e->files.size()) {
      e->declined = true;
      gtamd_set_error("FASTA and FASTQ files in one run: the host reader reads them");
      return -1;
    }
  }
  memset(&e->sum, 0, sizeof e->sum);
  e->input_bytes = total;
  const u64 max_tiles = div_up(longest, EN_TILE) + 1;
  u8 *d_raw = nullptr, *d_lut = nullptr;
  const u8 *lut = e->lut;
  u32 *d_tile = nullptr, *d_ws = nullptr;
  FaGlobals *d_glob = nullptr;
  int rc = -1;
  do {
    if (dev_alloc(&e->d_enc, total + EN_TILE) != 0) break;
    e->cap_enc = total + EN_TILE;
    if (dev_alloc(&d_raw, longest + EN_TILE) != 0 || dev_alloc(&d_lut, 256) != 0 ||
        dev_alloc(&d_tile, 6 * max_tiles) != 0 ||
        dev_alloc(&d_ws, scan_workspace_words(max_tiles)) != 0 ||
        dev_alloc(&d_glob, 1) != 0)
      break;
    if (hipMemcpyAsync(d_lut, lut, 256, hipMemcpyHostToDevice, e->st) != hipSuccess) {
      gtamd_set_error("cannot copy the symbol map to the device");
      break;
    }
    (void) hipEventRecord(e->ev[0], e->st);
    if (encode_files(e, d_raw, d_lut, d_tile, d_ws, d_glob, max_tiles) != 0) break;
    (void) hipEventRecord(e->ev[1], e->st);
    if (summarise(e) != 0) break;
    (void) hipEventRecord(e->ev[2], e->st);
    if (hipStreamSynchronize(e->st) != hipSuccess) {
      gtamd_set_error("device encoder failed: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    (void) hipEventElapsedTime(&e->parse_ms, e->ev[0], e->ev[1]);
    (void) hipEventElapsedTime(&e->stats_ms, e->ev[1], e->ev[2]);
    (void) hipEventElapsedTime(&e->total_ms, e->ev[0], e->ev[2]);
    rc = 0;
  } while (0);
  if (d_raw) (void) hipFree(d_raw);
  if (d_lut) (void) hipFree(d_lut);
  if (d_tile) (void) hipFree(d_tile);
  if (d_ws) (void) hipFree(d_ws);
  if (d_glob) (void) hipFree(d_glob);
  if (rc != 0) { enc_free(e); e->n = 0; e->ndesc = 0; return -1; }
  e->finished = true;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_encoder_set_symbols(gtamd_encoder *e, const uint8_t *symbols,
                                         uint64_t n) {
  GTAMD_ABI_BEGIN
  if (e == nullptr || (symbols == nullptr && n > 0)) {
    gtamd_set_error("invalid argument to gtamd_encoder_set_symbols");
    return -1;
  }
  HIP_TRY(hipSetDevice(e->device));
  enc_free(e);
  e->finished = false;
  e->files.clear();
  memset(&e->sum, 0, sizeof e->sum);
  e->input_bytes = n;
  e->ndesc = 0;
  if (dev_alloc(&e->d_enc, n + EN_TILE) != 0) return -1;
  e->cap_enc = n + EN_TILE;
  e->n = n;
  (void) hipEventRecord(e->ev[0], e->st);
  if (n > 0 && hipMemcpyAsync(e->d_enc, symbols, n, hipMemcpyHostToDevice, e->st) != hipSuccess) {
    gtamd_set_error("cannot copy %llu symbols to the device", (unsigned long long) n);
    enc_free(e);
    return -1;
  }
  (void) hipEventRecord(e->ev[1], e->st);
  if (summarise(e) != 0) { enc_free(e); e->n = 0; return -1; }
  (void) hipEventRecord(e->ev[2], e->st);
  HIP_TRY(hipStreamSynchronize(e->st));
  (void) hipEventElapsedTime(&e->parse_ms, e->ev[0], e->ev[1]);
  (void) hipEventElapsedTime(&e->stats_ms, e->ev[1], e->ev[2]);
  (void) hipEventElapsedTime(&e->total_ms, e->ev[0], e->ev[2]);
  e->finished = true;
  return 0;
  GTAMD_ABI_END(-1)
}

static int need_finished(const gtamd_encoder *e) {
  if (e == nullptr || !e->finished) {
    gtamd_set_error("gtamd_encoder_finish has not completed");
    return -1;
  }
  return 0;
}

extern "C" uint64_t gtamd_encoder_length(const gtamd_encoder *e) {
  GTAMD_ABI_BEGIN
  return e != nullptr && e->finished ? e->n : 0;
  GTAMD_ABI_END(0)
}

extern "C" const uint8_t *gtamd_encoder_device_symbols(const gtamd_encoder *e) {
  GTAMD_ABI_BEGIN
  return e != nullptr && e->finished ? e->d_enc : nullptr;
  GTAMD_ABI_END(nullptr)
}

extern "C" int gtamd_encoder_copy_symbols(const gtamd_encoder *e, uint8_t *dst,
                                          uint64_t first, uint64_t count) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  if (first > e->n || count > e->n - first) {
    gtamd_set_error("range [%llu, +%llu) exceeds the %llu encoded symbols",
                    (unsigned long long) first, (unsigned long long) count,
                    (unsigned long long) e->n);
    return -1;
  }
  HIP_TRY(hipSetDevice(e->device));
  if (count > 0) HIP_TRY(hipMemcpy(dst, e->d_enc + first, count, hipMemcpyDeviceToHost));
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_encoder_get_summary(const gtamd_encoder *e, gtamd_encode_summary *s) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  *s = e->sum;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_encoder_file_lengths(const gtamd_encoder *e, size_t file,
                                          uint64_t *length, uint64_t *effectivelength) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  if (file >= e->files.size()) { gtamd_set_error("no input file %zu", file); return -1; }
  const InputFile &f = e->files[file];
  // symbols plus the separators between the file's own sequences
  // (src/core/sequence_buffer_fasta.c:133-146,156)
  const u64 own_separators = f.ndesc > 0 ? f.ndesc - 1 : 0;
  u64 separators_written = f.ndesc;
  if (f.ndesc > 0 && f.desc_base == 0) separators_written--;   // first record of all
  *length = f.length;
  *effectivelength = f.out_len - separators_written + own_separators;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_encoder_declined(const gtamd_encoder *e) {
  GTAMD_ABI_BEGIN
  return e != nullptr && e->declined ? 1 : 0;
  GTAMD_ABI_END(0)
}
extern "C" uint64_t gtamd_encoder_num_fastq_records(const gtamd_encoder *e) {
  GTAMD_ABI_BEGIN
  return e != nullptr && e->finished ? (uint64_t) e->rec_seqlen.size() : 0;
  GTAMD_ABI_END(0)
}
extern "C" int gtamd_encoder_get_fastq_records(const gtamd_encoder *e, uint32_t *file, uint32_t *seqlen,
                                               uint32_t *desclen, uint64_t capacity) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  const size_t m = e->rec_seqlen.size();
  if (m > capacity) { gtamd_set_error("%zu records, room for %llu", m, (unsigned long long) capacity); return -1; }
  for (size_t k = 0; k < m; k++) { file[k] = e->rec_file[k]; seqlen[k] = e->rec_seqlen[k]; desclen[k] = e->rec_desclen[k]; }
  return 0;
  GTAMD_ABI_END(-1)
}
extern "C" uint64_t gtamd_encoder_num_descriptions(const gtamd_encoder *e) {
  GTAMD_ABI_BEGIN
  return e != nullptr && e->finished ? e->ndesc : 0;
  GTAMD_ABI_END(0)
}

// the caller's arrays hold `capacity` entries: more would be written -> error
static int check_capacity(const char *what, u64 needed, u64 capacity) {
  if (needed <= capacity) return 0;
  gtamd_set_error("%s: %llu entries do not fit the caller's %llu", what,
                  (unsigned long long) needed, (unsigned long long) capacity);
  return -1;
}

extern "C" int gtamd_encoder_get_descriptions(const gtamd_encoder *e, uint32_t *file,
                                              uint64_t *start, uint64_t *end,
                                              uint64_t capacity) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  TRY(check_capacity("descriptions", e->ndesc, capacity));
  HIP_TRY(hipSetDevice(e->device));
  if (e->ndesc == 0) return 0;
  HIP_TRY(hipMemcpy(start, e->d_desc_start, e->ndesc * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(end, e->d_desc_end, e->ndesc * 8, hipMemcpyDeviceToHost));
  for (size_t fi = 0; fi < e->files.size(); fi++) {
    const InputFile &f = e->files[fi];
    for (u64 k = 0; k < f.ndesc; k++) {
      file[f.desc_base + k] = (uint32_t) fi;
      // a description that the end of the file cuts off
      if (end[f.desc_base + k] == NONE64) end[f.desc_base + k] = f.length;
    }
  }
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_encoder_get_timing(const gtamd_encoder *e, float *total_ms,
                                        float *parse_ms, float *stats_ms,
                                        uint64_t *input_bytes) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  if (total_ms) *total_ms = e->total_ms;
  if (parse_ms) *parse_ms = e->parse_ms;
  if (stats_ms) *stats_ms = e->stats_ms;
  if (input_bytes) *input_bytes = e->input_bytes;
  return 0;
  GTAMD_ABI_END(-1)
}

// ---- INDEX.esq sections ---------------------------------------------------
static int launch_1d(u64 items, u32 *blocks) {
  const u64 b = div_up(items, 256);
  if (b == 0 || b > 0x7fffffffull) { gtamd_set_error("launch of %llu items", (unsigned long long) items); return -1; }
  *blocks = (u32) b;
  return 0;
}

extern "C" int gtamd_encoder_pack_twobit(const gtamd_encoder *e, int bitaccess,
                                         unsigned fillcode, uint64_t *words,
                                         uint64_t capacity) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  HIP_TRY(hipSetDevice(e->device));
  const u64 units = e->n < 32 ? 2 : 2 + (e->n - 1) / 32;
  TRY(check_capacity("two-bit encoding", units, capacity));
  u64 *d;
  u32 blocks;
  TRY(dev_alloc(&d, units));
  int rc = launch_1d(units, &blocks);
  if (rc == 0) {
    k_esq_twobit<<<blocks, 256, 0, e->st>>>(e->d_enc, e->n, units, bitaccess, fillcode & 3, d);
    if (hipGetLastError() != hipSuccess ||
        hipStreamSynchronize(e->st) != hipSuccess ||
        hipMemcpy(words, d, units * 8, hipMemcpyDeviceToHost) != hipSuccess) {
      gtamd_set_error("packing the two-bit encoding on the device failed");
      rc = -1;
    }
  }
  (void) hipFree(d);
  return rc;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_encoder_pack_specialbits(const gtamd_encoder *e, uint64_t *words,
                                              uint64_t capacity) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  HIP_TRY(hipSetDevice(e->device));
  const u64 units = 1 + (e->n + 63) / 64;
  TRY(check_capacity("special bits", units, capacity));
  u64 *d;
  u32 blocks;
  TRY(dev_alloc(&d, units));
  int rc = launch_1d(units, &blocks);
  if (rc == 0) {
    k_esq_specialbits<<<blocks, 256, 0, e->st>>>(e->d_enc, e->n, units, d);
    if (hipGetLastError() != hipSuccess ||
        hipStreamSynchronize(e->st) != hipSuccess ||
        hipMemcpy(words, d, units * 8, hipMemcpyDeviceToHost) != hipSuccess) {
      gtamd_set_error("packing the special bits on the device failed");
      rc = -1;
    }
  }
  (void) hipFree(d);
  return rc;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_encoder_pack_bytecompress(const gtamd_encoder *e, uint8_t *bytes,
                                               uint64_t capacity) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  HIP_TRY(hipSetDevice(e->device));
  const u32 sigma = e->sigma, bits = e->packbits;
  const u64 nbytes = ((u64) bits * e->n + 7) / 8;
  TRY(check_capacity("bit-packed symbols", nbytes, capacity));
  u8 *d;
  u32 blocks;
  TRY(dev_alloc(&d, nbytes));
  int rc = launch_1d(div_up(e->n, 8), &blocks);
  if (rc == 0) {
    k_esq_bitpack<<<blocks, 256, 0, e->st>>>(e->d_enc, e->n, sigma, bits, nbytes, d);
    if (hipGetLastError() != hipSuccess ||
        hipStreamSynchronize(e->st) != hipSuccess ||
        hipMemcpy(bytes, d, nbytes, hipMemcpyDeviceToHost) != hipSuccess) {
      gtamd_set_error("bit-packing the symbols on the device failed");
      rc = -1;
    }
  }
  (void) hipFree(d);
  return rc;
  GTAMD_ABI_END(-1)
}

// the `expected` positions of one kind, in increasing order, to host memory
static int positions_to_host(const gtamd_encoder *e, int kind, u64 expected, u64 *out) {
  if (expected == 0) return 0;
  const u64 ntiles = div_up(e->n, EN_TILE);
  u32 *d_cnt = nullptr, *d_ws = nullptr;
  u64 *d_out = nullptr;
  int rc = -1;
  do {
    if (dev_alloc(&d_cnt, 2 * ntiles) != 0 || dev_alloc(&d_out, expected) != 0 ||
        dev_alloc(&d_ws, scan_workspace_words(ntiles)) != 0)
      break;
    k_positions<0><<<(u32) ntiles, EN_THREADS, 0, e->st>>>(e->d_enc, e->n, kind, d_cnt,
                                                          nullptr, nullptr);
    if (hipGetLastError() != hipSuccess) { gtamd_set_error("k_positions launch failed"); break; }
    if (scan_u32(SCAN_SUM, d_cnt, d_cnt + ntiles, ntiles, false, d_ws, e->st) != 0) break;
    u32 last[2];
    if (hipStreamSynchronize(e->st) != hipSuccess ||
        hipMemcpy(&last[0], d_cnt + ntiles - 1, 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&last[1], d_cnt + 2 * ntiles - 1, 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gtamd_set_error("reading position counts from the device failed");
      break;
    }
    if ((u64) last[0] + last[1] != expected) {
      gtamd_set_error("found %llu positions, the sequence statistics say %llu",
                      (unsigned long long) last[0] + last[1], (unsigned long long) expected);
      break;
    }
    k_positions<1><<<(u32) ntiles, EN_THREADS, 0, e->st>>>(e->d_enc, e->n, kind, nullptr,
                                                          d_cnt + ntiles, d_out);
    if (hipGetLastError() != hipSuccess ||
        hipStreamSynchronize(e->st) != hipSuccess ||
        hipMemcpy(out, d_out, expected * 8, hipMemcpyDeviceToHost) != hipSuccess) {
      gtamd_set_error("collecting positions on the device failed");
      break;
    }
    rc = 0;
  } while (0);
  if (d_cnt) (void) hipFree(d_cnt);
  if (d_out) (void) hipFree(d_out);
  if (d_ws) (void) hipFree(d_ws);
  return rc;
}

extern "C" int gtamd_encoder_get_wildcard_runs(const gtamd_encoder *e, uint64_t *start,
                                               uint64_t *length, uint64_t capacity) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  HIP_TRY(hipSetDevice(e->device));
  const u64 runs = e->sum.realwildcardranges;
  TRY(check_capacity("wildcard runs", runs, capacity));
  TRY(positions_to_host(e, POS_WILDCARD_START, runs, start));
  TRY(positions_to_host(e, POS_WILDCARD_END, runs, length));
  for (u64 r = 0; r < runs; r++) length[r] = length[r] - start[r] + 1;
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" int gtamd_encoder_get_separators(const gtamd_encoder *e, uint64_t *pos,
                                            uint64_t capacity) {
  GTAMD_ABI_BEGIN
  TRY(need_finished(e));
  HIP_TRY(hipSetDevice(e->device));
  TRY(check_capacity("separators", e->sum.numofsequences - 1, capacity));
  return positions_to_host(e, POS_SEPARATOR, e->sum.numofsequences - 1, pos);
  GTAMD_ABI_END(-1)
}
