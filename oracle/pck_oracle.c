/*
  pck_oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or
  called from the product.  Only tests/ may use it.

  Plain-C restatement of the file `gt packedindex trsuftab INDEX` writes
  (INDEX.bdx: the block-composition compressed BWT of the reference's packed
  index) from the tables of a suffix-array project: .bwt bytes, .suf entries,
  the encoded sequence.  It follows the reference step by step -- also where the
  reference's staging buffers leave stale bits in the file -- and cites the
  file:line of each step.  Parity pinned: checked against INDEX.bdx files the
  reference itself wrote (oracle/_ref/gt_ref_pck, built from /root/reference by
  oracle/Makefile.ref) on the reference's own fixtures and option sets; their
  md5 sums are committed in tests/golden/golden_pck.json
  (tests/golden/make_golden_pck.py).

  Covered: block encoding with any block size / blocks per bucket the
  reference accepts, locate information as bitmap or as counts, no locate
  information, -sprank (reversibly sorted specials), both flavours (trsuftab:
  no sequence statistics; mkindex: with them); alphabets of the suffixerator
  (DNA, protein).  Not covered: -ctxilog (context map, a second file).
*/
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "pck_oracle.h"

/* gt_requiredUInt64Bits, src/core/bitpackstringop.c:60-79: bits needed for v, 1 for v = 0 */
static unsigned reqbits(uint64_t v)
{
  unsigned r = 1;
  while (v >>= 1) r++;
  return r;
}

/* gt_bsStoreUInt64 semantics (src/core/bitpackstringop.c): most significant
   bit first, bit 0 of the string is the top bit of byte 0, other bits kept */
static void bs_store(uint8_t *s, uint64_t off, unsigned bits, uint64_t v)
{
  unsigned i;
  for (i = 0; i < bits; i++) {
    uint64_t p = off + i;
    uint8_t m = (uint8_t) (0x80u >> (p & 7));
    if ((v >> (bits - 1 - i)) & 1) s[p >> 3] |= m; else s[p >> 3] &= (uint8_t) ~m;
  }
}

static uint64_t binom(unsigned n, unsigned k)
{
  uint64_t r = 1;
  unsigned i;
  if (k > n) return 0;
  if (k > n - k) k = n - k;
  for (i = 1; i <= k; i++) r = r * (n - k + i) / i;
  return r;
}

static uint64_t fact(unsigned n)
{
  uint64_t r = 1;
  while (n > 1) r *= n--;
  return r;
}

/* number of arrangements of a multiset with the given counts */
static uint64_t multinomial(const unsigned *cnt, unsigned sigma)
{
  unsigned s, tot = 0;
  uint64_t r;
  for (s = 0; s < sigma; s++) tot += cnt[s];
  r = fact(tot);
  for (s = 0; s < sigma; s++) r /= fact(cnt[s]);
  return r;
}

/* index of a composition in the list gt_initCompositionList builds
   (src/match/eis-seqblocktranslate.c:156-246): all compositions in ascending
   order of the string (count of symbol 0, count of symbol 1, ...), the first
   one being (0, ..., 0, blockSize) */
static uint64_t comp_index(const unsigned *cnt, unsigned sigma, unsigned bsize)
{
  uint64_t r = 0;
  unsigned i, v, left = bsize;
  for (i = 0; i + 1 < sigma; i++) {
    unsigned k = sigma - i - 1;   /* symbols after i */
    for (v = 0; v < cnt[i]; v++)
      r += binom(left - v + k - 1, k - 1);
    left -= cnt[i];
  }
  return r;
}

/* index of a block among the arrangements of its composition in ascending
   order (initPermutationsList / nextPermutation, eis-seqblocktranslate.c:319-425) */
static uint64_t perm_index(const uint8_t *block, unsigned bsize, unsigned *cnt, unsigned sigma)
{
  uint64_t r = 0;
  unsigned p, s;
  for (p = 0; p < bsize; p++) {
    for (s = 0; s < block[p]; s++)
      if (cnt[s]) {
        cnt[s]--;
        r += multinomial(cnt, sigma);
        cnt[s]++;
      }
    cnt[block[p]]--;
  }
  return r;
}

/* largest number of arrangements any composition has: counts as even as possible */
static uint64_t max_perms(unsigned sigma, unsigned bsize)
{
  unsigned cnt[256], s;
  for (s = 0; s < sigma; s++) cnt[s] = bsize / sigma + (s < bsize % sigma ? 1u : 0u);
  return multinomial(cnt, sigma);
}

typedef struct {
  uint8_t *d; size_t len, cap;
} outbuf;

/* fseeko + fwrite on a file that starts empty: holes read as zeros */
static void out_pwrite(outbuf *o, uint64_t pos, const void *src, size_t n)
{
  if (pos + n > o->cap) {
    size_t nc = (size_t) ((pos + n) * 2 + 4096);
    o->d = realloc(o->d, nc);
    memset(o->d + o->cap, 0, nc - o->cap);
    o->cap = nc;
  }
  memcpy(o->d + pos, src, n);
  if (pos + n > o->len) o->len = (size_t) (pos + n);
}

static uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

/* MRAEncGetRangeOfSymbol(MRAEncMapSymbol(...)) of the BWT alphabet
   (gt_SANewMRAEnc, eis-suffixarray-interface.c:210-218): letters are range 0,
   wildcard (= UNDEFBWTCHAR) and separator range 1 */
static int sym_range(unsigned sym) { return sym >= 254u; }

/* isSortModeTransition, src/match/eis-bwtseq-extinfo.c:343-382, for the default
   range sort modes { VALUE, UNDEFINED } */
static int sort_mode_transition(const uint8_t *seq, uint64_t seqlen, uint64_t pos)
{
  unsigned a, b;
  if (pos > 0 && pos < seqlen - 1) { a = seq[pos - 1]; b = seq[pos]; }
  else if (pos == 0) { a = 254u; b = seq[0]; }
  else { a = seq[seqlen - 2]; b = 254u; }
  return sym_range(a) != sym_range(b);
}

typedef struct {
  /* parameters and derived widths */
  unsigned sigma, bsize, bblocks, locint;
  int loc_bitmap, loc_count;
  uint64_t total_len, bucket_len;
  unsigned bits_per_ulong, comp_idx_bits, var_off_bits, cb_off_bits, bits_per_orig_pos;
  const unsigned *sym_bits, *sym_off;
  uint64_t cw_ext_bits, pre_var_idx, pre_cb_off, pre_comp_idx, pre_cw_ext;
  uint64_t cw_data_pos, var_data_pos;
  /* struct appendState, eis-blockcomp.c:142-152 */
  uint8_t *comp_cache, *perm_cache;
  uint64_t cw_mem_pos, cw_disk_off, var_mem_pos, cw_mem_old, var_disk_off, var_mem_old;
  /* symbol counts up to the current block / up to the start of the bucket */
  uint64_t *buck, *buck_last;
  /* region list of the special symbols */
  uint64_t nranges, ranges_cap, *r_start, *r_len;
  uint8_t *r_sym;
  /* inputs */
  const uint8_t *bwt, *seq;
  const uint64_t *suf;
  uint64_t suf_next;
  uint64_t *mark_bwt, *mark_orig;
  uint8_t *block;
  /* -sprank (BWTReversiblySorted): ranks of the specials of the text */
  int reversible;
  unsigned bits_per_orig_rank;
  const uint64_t *sprank;       /* sprank[q] = specials in seq[0, q) */
  uint64_t *rank_queue;
  outbuf o;
} pck_state;

/* gt_SRLAddPosition, eis-seqranges.c:183-213 (symbol of the range alphabet:
   wildcard 0, separator 1; 1 symbol bit leaves 63 bits for the length) */
static void add_region_position(pck_state *st, uint64_t pos, uint8_t rs)
{
  const uint64_t max_range_len = (((uint64_t) 1) << 63) - 1;
  uint64_t n = st->nranges;
  if (n && st->r_sym[n - 1] == rs && st->r_start[n - 1] + st->r_len[n - 1] == pos
      && st->r_len[n - 1] < max_range_len) {
    st->r_len[n - 1]++;
    return;
  }
  if (n == st->ranges_cap) {
    st->ranges_cap *= 2;
    st->r_start = realloc(st->r_start, st->ranges_cap * sizeof *st->r_start);
    st->r_len = realloc(st->r_len, st->ranges_cap * sizeof *st->r_len);
    st->r_sym = realloc(st->r_sym, st->ranges_cap);
  }
  st->r_start[n] = pos; st->r_len[n] = 1; st->r_sym[n] = rs; st->nranges = n + 1;
}

/* one block of the BWT: SDRRead + gt_MRAEncSymbolsTransform + (for the last,
   short block) the fill with symbol 0, then addBlock2OutputBuffer and
   append2IdxOutput (eis-blockcomp.c:238-267, 541-554, 574-596, 1762-1775) */
static void add_block(pck_state *st, uint64_t block_num)
{
  const unsigned sigma = st->sigma, bsize = st->bsize;
  uint64_t base = block_num * bsize;
  uint64_t take = st->total_len - base < bsize ? st->total_len - base : bsize, i;
  unsigned cnt[256];
  uint8_t *block = st->block;
  for (i = 0; i < bsize; i++) {
    unsigned c = i < take ? st->bwt[base + i] : 0;
    block[i] = (uint8_t) (c == 254u ? sigma : c == 255u ? sigma + 1 : c);
  }
  for (i = 0; i < bsize; i++) st->buck[block[i]]++;
  for (i = 0; i < bsize; i++)
    if (block[i] >= sigma) add_region_position(st, base + i, (uint8_t) (block[i] - sigma));
  /* block alphabet: symbols of the region list fall back to symbol 0 */
  for (i = 0; i < bsize; i++) if (block[i] >= sigma) block[i] = 0;
  memset(cnt, 0, sizeof cnt[0] * sigma);
  for (i = 0; i < bsize; i++) cnt[block[i]]++;
  {
    uint64_t ci = comp_index(cnt, sigma, bsize);
    uint64_t np = multinomial(cnt, sigma);
    unsigned pbits = np > 1 ? reqbits(np - 1) : 0;
    uint64_t pi = np > 1 ? perm_index(block, bsize, cnt, sigma) : 0;
    bs_store(st->comp_cache, st->cw_mem_pos, st->comp_idx_bits, ci);
    st->cw_mem_pos += st->comp_idx_bits;
    bs_store(st->perm_cache, st->var_mem_pos, pbits, pi);
    st->var_mem_pos += pbits;
  }
}

/* writeOutputBuffer, eis-blockcomp.c:269-293, for the bucket that covers `len`
   positions from the last update position */
static void flush_bucket(pck_state *st, uint64_t len)
{
  /* appendCallBackOutput, eis-blockcomp.c:1777-1802, with addLocateInfo,
     eis-bwtseq-extinfo.c:384-541 */
  if (st->locint) {
    uint64_t nmarks = 0, i, written = 0;
    unsigned bits_bwt_pos = reqbits(len - 1);
    bs_store(st->comp_cache, st->cw_mem_old + st->pre_cb_off, st->cb_off_bits,
             st->var_mem_pos - st->var_mem_old);
    uint64_t nranks = 0;
    for (i = 0; i < len; i++) {
      uint64_t v = st->suf[st->suf_next++];
      /* reversibly sorted specials: no extra marks where letters and specials
         meet, the mark stores the text position divided by the interval
         (eis-bwtseq-extinfo.c:415-441) */
      int mark = (v % st->locint) == 0 ||
                 (!st->reversible && sort_mode_transition(st->seq, st->total_len, v));
      if (mark) { st->mark_bwt[nmarks] = i; st->mark_orig[nmarks] = st->reversible ? v / st->locint : v; nmarks++; }
      if (st->bits_per_orig_rank) {
        /* eis-bwtseq-extinfo.c:452-471: the symbol before the suffix is sorted by
           rank (a special, or the undefined symbol before suffix 0): its rank
           among the specials of the text, specialsRank (eis-specialsrank.c:160-190) */
        unsigned sym = v ? st->seq[v - 1] : 254u;
        if (sym >= 254u) st->rank_queue[nranks++] = st->sprank[v ? v - 1 : st->total_len - 1];
      }
      if (st->loc_bitmap)
        bs_store(st->comp_cache, st->cw_mem_old + st->pre_cw_ext + i, 1, (uint64_t) mark);
    }
    if (st->loc_count) {
      unsigned bits_count = reqbits(len);
      bs_store(st->perm_cache, st->var_mem_pos + written, bits_count, nmarks);
      written += bits_count;
    }
    for (i = 0; i < nmarks; i++) {
      if (st->loc_count) {
        bs_store(st->perm_cache, st->var_mem_pos + written, bits_bwt_pos, st->mark_bwt[i]);
        written += bits_bwt_pos;
      }
      bs_store(st->perm_cache, st->var_mem_pos + written, st->bits_per_orig_pos, st->mark_orig[i]);
      written += st->bits_per_orig_pos;
    }
    for (i = 0; i < nranks; i++) {
      bs_store(st->perm_cache, st->var_mem_pos + written, st->bits_per_orig_rank, st->rank_queue[i]);
      written += st->bits_per_orig_rank;
    }
    st->cw_mem_pos = st->pre_cw_ext + st->cw_mem_old + st->cw_ext_bits;
    st->var_mem_pos += written;
  }
  /* updateIdxOutput, eis-blockcomp.c:1807-1886: counts up to the start of the
     bucket, offset of its variable-width part, then both buffers go to the file
     in whole bytes and the incomplete last byte moves to the front */
  {
    unsigned s;
    uint64_t nbytes, vbytes;
    for (s = 0; s < st->sigma; s++)
      bs_store(st->comp_cache, st->cw_mem_old + st->sym_off[s], st->sym_bits[s], st->buck_last[s]);
    bs_store(st->comp_cache, st->cw_mem_old + st->pre_var_idx, st->var_off_bits, st->var_disk_off);
    nbytes = st->cw_mem_pos / 8;
    out_pwrite(&st->o, st->cw_data_pos + st->cw_disk_off, st->comp_cache, (size_t) nbytes);
    if ((st->cw_mem_old = st->cw_mem_pos % 8) != 0) st->comp_cache[0] = st->comp_cache[nbytes];
    vbytes = st->var_mem_pos / 8;
    out_pwrite(&st->o, st->var_data_pos + st->var_disk_off / 8, st->perm_cache, (size_t) vbytes);
    if (st->var_mem_pos % 8) st->perm_cache[0] = st->perm_cache[vbytes];
    st->cw_disk_off += nbytes;
    st->cw_mem_pos = st->pre_comp_idx + st->cw_mem_old;
    st->var_disk_off += st->var_mem_pos - st->var_mem_old;
    st->var_mem_old = (st->var_mem_pos %= 8);
  }
}

static uint64_t last_var_bits;
/* bits of the variable-width part of the last file made (for tests of the
   product's own replay of the staging buffers) */
uint64_t ora_pck_last_var_bits(void) { return last_var_bits; }

int ora_pck_bdx(const uint8_t *bwt, const uint64_t *suf, const uint8_t *seq,
                uint64_t total_len, unsigned sigma, uint64_t longest,
                const ora_pck_params *pp, uint8_t **out, size_t *out_len)
{
  const unsigned bsize = pp->block_size, bblocks = pp->bucket_blocks;
  const unsigned locint = pp->locate_interval;
  const int toggles = pp->feature_toggles;
  const int loc_bitmap = (toggles & ORA_PCK_LOCATE_BITMAP) != 0;
  const int loc_count = (toggles & ORA_PCK_LOCATE_COUNT) != 0;
  const int reversible = locint && (toggles & ORA_PCK_REVERSIBLY_SORTED) != 0;
  const uint64_t bucket_len = (uint64_t) bsize * bblocks;
  uint64_t *sprank = NULL, total_specials = 0;
  unsigned bits_per_orig_rank = 0;
  /* numBuckets, eis-blockcomp.c:1633-1638 */
  const uint64_t nbuckets = (total_len + 1) / bucket_len + (((total_len + 1) % bucket_len) ? 1 : 0);
  unsigned bits_per_ulong, comp_idx_bits, max_perm_idx_bits;
  unsigned sym_sum_bits, cb_off_bits, var_off_bits, num_modes = 2;
  uint64_t cw_ext_bits, max_var_ext_bits_per_bucket = 0, max_var_bits_total;
  uint64_t cw_bits, header_len, cw_data_pos, var_data_pos, cw_len, range_enc_pos;
  unsigned bits_per_orig_pos = 0, state_bits_per_ulong = 0;
  unsigned sym_bits[256], sym_off[256];
  uint64_t letter_count[256], regular = 0;
  pck_state st;

  if (!bsize || !bblocks || sigma < 1 || sigma > 250 || bsize > 20 || total_len < 2) return -1;
  if (locint && !loc_bitmap && !loc_count) return -1;
  memset(&st, 0, sizeof st);

  /* gt_newGenBlockEncIdxSeq, eis-blockcomp.c:336-339 */
  bits_per_ulong = reqbits(total_len - 1);
  /* no sequence statistics on the trsuftab path (gt_initSuffixarrayFileInterface
     passes none): symSumBitsDefaultSetup, eis-blockcomp.c:757-774; on the mkindex
     path the counters are as wide as the number of occurrences of each letter
     needs (eis-blockcomp.c:385-437; newSeqStatsFromCharDist,
     eis-suffixerator-interface.c:176-206) */
  {
    unsigned s;
    uint64_t i;
    for (s = 0; s < sigma; s++) letter_count[s] = 0;
    for (i = 0; i < total_len; i++) if (bwt[i] < sigma) { letter_count[bwt[i]]++; regular++; }
    sym_sum_bits = 0;
    for (s = 0; s < sigma; s++) {
      sym_bits[s] = pp->with_statistics ? reqbits(letter_count[s]) : bits_per_ulong;
      sym_off[s] = sym_sum_bits;
      sym_sum_bits += sym_bits[s];
    }
  }
  /* gt_initCompositionList, eis-seqblocktranslate.c:168-183,244 */
  comp_idx_bits = reqbits(binom(bsize + sigma - 1, sigma - 1) - 1);
  max_perm_idx_bits = reqbits(max_perms(sigma, bsize) - 1);
  /* eis-blockcomp.c:490-493 */
  cw_ext_bits = (uint64_t) (locint && loc_bitmap ? 1 : 0) * bucket_len;
  cb_off_bits = locint ? reqbits((uint64_t) max_perm_idx_bits * bblocks) : 0;
  /* vwBits, eis-blockcomp.c:1659-1689 with locBitsUpperBounds,
     eis-bwtseq-extinfo.c:195-251 and initAddLocateInfoState, :253-337 */
  max_var_bits_total = nbuckets * ((uint64_t) max_perm_idx_bits * bblocks);
  if (locint) {
    uint64_t last_pos = total_len - 1, extra = 0, desc_len[2], desc_rep[2], max_seg = 0, tot = 0;
    int i;
    state_bits_per_ulong = reqbits(last_pos);
    bits_per_orig_pos = reversible ? reqbits(last_pos / locint) : reqbits(last_pos);
    if (reversible) {
      /* buildSpRTable / gt_createBWTSeqGeneric, eis-bwtseq-construct.c:206-229,
         eis-bwtseq-extinfo.c:585-600 */
      uint64_t q;
      sprank = malloc((size_t) (total_len + 1) * sizeof *sprank);
      sprank[0] = 0;
      for (q = 0; q + 1 < total_len; q++) sprank[q + 1] = sprank[q] + (seq[q] >= 254u ? 1 : 0);
      total_specials = sprank[total_len - 1];
      bits_per_orig_rank = reqbits(total_specials);
    }
    if (!reversible && locint > 1) {
      uint64_t std_marks = total_len / locint;
      uint64_t a = total_len / 2, b = total_len - std_marks;
      extra = a < b ? a : b;
      if (pp->with_statistics) {
        /* symbols outside the value-sorted range: wildcards, separators and the
           undefined symbol before suffix 0, as newSeqStatsFromCharDist counts
           them (eis-bwtseq-extinfo.c:302-314) */
        uint64_t nonval = total_len - regular + 1, rest = total_len - nonval;
        if (nonval < extra) extra = nonval;
        if (rest < extra) extra = rest;
      }
    }
    desc_rep[0] = (total_len + 1) / bucket_len; desc_len[0] = bucket_len;
    desc_rep[1] = ((total_len + 1) % bucket_len) ? 1 : 0; desc_len[1] = total_len % bucket_len;
    for (i = 0; i < 2; i++) {
      if (desc_len[i] > max_seg) max_seg = desc_len[i];
      if (loc_count) tot += reqbits(desc_len[i]) * desc_rep[i];
    }
    tot += (total_len / locint + extra) * ((loc_count ? reqbits(max_seg) : 0) + bits_per_orig_pos);
    /* specialsRank(seqLen): the specials and the terminator -- which the
       reference's sample table counts twice when seqLen falls on a sample
       position (the last sample then covers the terminator and
       specialsRankFromSampleTable adds it again, eis-specialsrank.c:108-128,
       160-190); sample interval 2^bits(bits(seqLen)), eis-bwtseq-construct.c:217-222 */
    if (bits_per_orig_rank) {
      uint64_t bound = total_specials + 1;
      if (total_specials && total_len % ((uint64_t) 1 << reqbits(reqbits(total_len))) == 0) bound++;
      tot += bound * bits_per_orig_rank;
    }
    max_var_ext_bits_per_bucket =
      max_seg * ((loc_count ? state_bits_per_ulong : 0) + bits_per_orig_pos + bits_per_orig_rank)
      + (loc_count ? reqbits(max_seg) : 0);
    max_var_bits_total += tot;
  }
  var_off_bits = reqbits(max_var_bits_total);
  /* superBlockCWBits, eis-blockcomp.c:1603-1610 */
  cw_bits = sym_sum_bits + var_off_bits + cb_off_bits + (uint64_t) comp_idx_bits * bblocks + cw_ext_bits;
  /* blockEncIdxSeqHeaderLength, eis-blockcomp.c:1919-1946 */
  header_len = 4 + 4 + 8 + 8 + 12 + 12 + 8 + 8 + 8 + 4 * sigma + 8 + 8 + 8 + 12 + 4 * num_modes;
  if (cb_off_bits) header_len += 8 + 12 + 12;
  cw_data_pos = round_up(header_len + (locint ? 8 + 16 : 0) + (bits_per_orig_rank ? 8 + 8 : 0),
                         8192);                     /* initOnDiskBlockCompIdx, :1715-1722 */
  cw_len = (cw_bits * nbuckets + 7) / 8;            /* cwSize, :1640-1649 */
  var_data_pos = cw_data_pos + cw_len;

  st.sigma = sigma; st.bsize = bsize; st.bblocks = bblocks; st.locint = locint;
  st.loc_bitmap = loc_bitmap; st.loc_count = loc_count;
  st.total_len = total_len; st.bucket_len = bucket_len;
  st.bits_per_ulong = bits_per_ulong; st.comp_idx_bits = comp_idx_bits;
  st.sym_bits = sym_bits; st.sym_off = sym_off;
  st.var_off_bits = var_off_bits; st.cb_off_bits = cb_off_bits;
  st.bits_per_orig_pos = bits_per_orig_pos; st.cw_ext_bits = cw_ext_bits;
  st.pre_var_idx = sym_sum_bits; st.pre_cb_off = st.pre_var_idx + var_off_bits;
  st.pre_comp_idx = st.pre_cb_off + cb_off_bits;
  st.pre_cw_ext = st.pre_comp_idx + (uint64_t) comp_idx_bits * bblocks;
  st.cw_data_pos = cw_data_pos; st.var_data_pos = var_data_pos;
  /* the staging buffers of initAppendState, eis-blockcomp.c:1735-1750: the
     reference allocates them zeroed once and never clears them again */
  st.comp_cache = calloc((size_t) ((cw_bits + 7) / 8 + 2), 1);
  st.perm_cache = calloc((size_t) (((uint64_t) max_perm_idx_bits * bblocks
                                    + max_var_ext_bits_per_bucket + 7) / 8 + 2), 1);
  st.cw_mem_pos = st.pre_comp_idx;
  st.buck = calloc(sigma + 2, sizeof *st.buck);
  st.buck_last = calloc(sigma + 2, sizeof *st.buck_last);
  st.ranges_cap = 1024;
  st.r_start = malloc(st.ranges_cap * sizeof *st.r_start);
  st.r_len = malloc(st.ranges_cap * sizeof *st.r_len);
  st.r_sym = malloc(st.ranges_cap);
  st.bwt = bwt; st.seq = seq; st.suf = suf;
  st.mark_bwt = malloc((size_t) bucket_len * sizeof *st.mark_bwt);
  st.mark_orig = malloc((size_t) bucket_len * sizeof *st.mark_orig);
  st.block = malloc(bsize);
  st.reversible = reversible; st.bits_per_orig_rank = bits_per_orig_rank; st.sprank = sprank;
  st.rank_queue = malloc((size_t) bucket_len * sizeof *st.rank_queue);

  /* the construction loop, eis-blockcomp.c:529-609 */
  {
    uint64_t num_full_blocks = total_len / bsize, block_num = 0, last_update = 0;
    while (block_num < num_full_blocks) {
      add_block(&st, block_num);
      if (!((++block_num) % bblocks)) {
        flush_bucket(&st, bucket_len);
        memcpy(st.buck_last, st.buck, (sigma + 2) * sizeof *st.buck);
        last_update = block_num * bsize;
      }
    }
    if (total_len % bsize) add_block(&st, block_num);
    flush_bucket(&st, total_len - last_update);      /* "one bucket still unfinished" */
  }
  /* finalizeIdxOutput, eis-blockcomp.c:2420-2471 */
  if (st.cw_mem_old) { out_pwrite(&st.o, cw_data_pos + st.cw_disk_off, st.comp_cache, 1); st.cw_disk_off++; }
  if (st.var_mem_old) out_pwrite(&st.o, var_data_pos + st.var_disk_off / 8, st.perm_cache, 1);
  range_enc_pos = var_data_pos + st.var_disk_off / 8 + ((st.var_disk_off % 8) ? 1 : 0);
  last_var_bits = st.var_disk_off;
  {
    /* gt_SRLSaveToStream, eis-seqranges.c:459-468, after the terminator region
       just beyond the sequence (symbol 0 of the range alphabet) */
    uint64_t i, nr = st.nranges + 1;
    uint8_t *rb = calloc((size_t) (8 + 16 * nr), 1);
    memcpy(rb, &nr, 8);
    for (i = 0; i < nr; i++) {
      uint64_t start = i < st.nranges ? st.r_start[i] : total_len + bsize;
      uint64_t len = i < st.nranges ? st.r_len[i] : 1;
      unsigned sym = i < st.nranges ? st.r_sym[i] : 0;
      memcpy(rb + 8 + 16 * i, &start, 8);
      /* struct seqRange, eis-seqranges-priv.h:25-63: 1 symbol bit, 63 length bits */
      bs_store(rb + 8 + 16 * i + 8, 0, 1, sym);
      bs_store(rb + 8 + 16 * i + 8, 1, 63, len);
    }
    out_pwrite(&st.o, range_enc_pos, rb, (size_t) (8 + 16 * nr));
    free(rb);
  }
  free(st.comp_cache); free(st.perm_cache); free(st.buck); free(st.buck_last);
  free(st.r_start); free(st.r_len); free(st.r_sym); free(st.block);
  free(st.mark_bwt); free(st.mark_orig); free(st.rank_queue); free(sprank);
  /* writeIdxHeader, eis-blockcomp.c:1984-2094 */
  {
    uint8_t *h = calloc((size_t) header_len + 128, 1);
    uint64_t off = 8, v64;
    uint32_t v32;
    unsigned i;
#define PUT32(x) do { v32 = (uint32_t) (x); memcpy(h + off, &v32, 4); off += 4; } while (0)
#define PUT64(x) do { v64 = (uint64_t) (x); memcpy(h + off, &v64, 8); off += 8; } while (0)
    memcpy(h, "BDX", 4);
    v32 = (uint32_t) round_up(header_len, 8192); memcpy(h + 4, &v32, 4);
    PUT32(0x424b535a); PUT32(bsize);
    PUT32(0x42424c4b); PUT32(bblocks);
    PUT32(0x564f4646); PUT64(var_data_pos);
    PUT32(0x524f4646); PUT64(range_enc_pos);
    PUT32(0x53454c45); PUT64(total_len);
    PUT32(0x53504254); PUT32(bits_per_ulong);
    PUT32(0x56444f42); PUT32(var_off_bits);
    PUT32(0x53534254); PUT32(sigma);
    for (i = 0; i < sigma; i++) PUT32(sym_bits[i]);
    PUT32(0x42454642); PUT32(0);
    PUT32(0x52454642); PUT32(0);
    PUT32(0x4e4d524e); PUT32(num_modes);
    PUT32(1);   /* BLOCK_COMPOSITION_INCLUDE (enum rangeStoreMode, eis-encidxseq.h) */
    PUT32(2);   /* REGIONS_LIST */
    if (cb_off_bits) {
      PUT32(0x43424d42); PUT32(cb_off_bits);
      PUT32(0x43455842); PUT64(cw_ext_bits);
      PUT32(0x4d455842); PUT64(max_var_ext_bits_per_bucket);
    }
    if (off != header_len) { free(h); free(st.o.d); return -2; }
    if (locint) {
      /* writeExtIdxHeader + writeLocateInfoHeader, eis-blockcomp.c:1964-1972,
         eis-bwtseq-extinfo.c:39-76 */
      PUT32(0x45480000u | 1111u); PUT32(16);
      PUT64(longest); PUT32(locint); PUT32((uint32_t) toggles);
      if (bits_per_orig_rank) {
        /* writeRankSortHeader, eis-bwtseq-extinfo.c:106-122: bits per rank, then the
           sort mode of the two ranges as int16: SORTMODE_VALUE 0, SORTMODE_RANK 2 */
        uint16_t m;
        PUT32(0x45480000u | 1112u); PUT32(8);
        PUT32(bits_per_orig_rank);
        m = 0; memcpy(h + off, &m, 2); off += 2;
        m = 2; memcpy(h + off, &m, 2); off += 2;
      }
    }
    out_pwrite(&st.o, 0, h, (size_t) off);
    free(h);
  }
  *out = st.o.d;
  *out_len = st.o.len;
  return 0;
}

void ora_pck_free(uint8_t *p) { free(p); }

/* estimateBestLocateTypeFeature + gt_computePackedIndexDefaults,
   src/match/eis-bwtseq-param.c:69-103; the segment length of the block
   encoding is blockSize * bucketBlocks (gt_blockEncIdxSeqSegmentLen,
   eis-blockcomp.c:2682-2686) */
int ora_pck_default_toggles(unsigned block_size, unsigned bucket_blocks,
                            unsigned locate_interval, int locbitmap)
{
  if (locbitmap >= 0) return locbitmap ? ORA_PCK_LOCATE_BITMAP : ORA_PCK_LOCATE_COUNT;
  if (!locate_interval) return 0;
  {
    unsigned seg = block_size * bucket_blocks;
    if (seg > (seg + 1) * reqbits(seg) / locate_interval) return ORA_PCK_LOCATE_COUNT;
    return ORA_PCK_LOCATE_BITMAP;
  }
}

/* The context map `gt packedindex mkindex|trsuftab -ctxilog I` / `gt packedindex
   mkctxmap` write beside the index (INDEX.<I>cxm, src/match/eis-bwtseq-context.c):
   16 bits interval log, 16 bits entry width, then for every 2^I-th text position
   q the row of the suffix that FOLLOWS it -- gt_BWTSCRFMapAdvance :158-177 maps
   origPos = (suf[row] + seqLen - 1) % seqLen -- as uniform entries of bits(seqLen - 1)
   bits, most significant bit first (readBS2Map :239-260).  The file is created
   by writing the first character of its own suffix ".<I>cxm" at its last byte
   (BWTSeqCRMapOpen :291-300 writes `buf`, not a zero), so the unused bits of the
   last byte are those of '.'.  ilog < 0: gt_requiredUIntBits(requiredUlongBits(seqLen)),
   :65-68.  Returns the interval log used, or -1. */
int ora_pck_ctxmap(const uint64_t *suf, uint64_t total_len, int ilog, uint8_t **out,
                   size_t *out_len)
{
  unsigned bits = reqbits(total_len - 1);
  uint64_t nentries, r, size;
  uint8_t *d;
  if (ilog < 0) ilog = (int) reqbits(reqbits(total_len));
  if ((unsigned) ilog >= reqbits(total_len) || ilog > 62) return -1;   /* ctxMapILogIsValid */
  nentries = (total_len + ((uint64_t) 1 << ilog) - 1) >> ilog;
  size = 4 + (bits * nentries + 7) / 8;
  d = calloc((size_t) size + 8, 1);
  d[size - 1] = '.';
  bs_store(d, 0, 16, (uint64_t) ilog);
  bs_store(d, 16, 16, bits);
  for (r = 0; r < total_len; r++) {
    uint64_t op = (suf[r] + total_len - 1) % total_len;
    if ((op & (((uint64_t) 1 << ilog) - 1)) == 0)
      bs_store(d + 4, (op >> ilog) * bits, bits, r);
  }
  *out = d;
  *out_len = (size_t) size;
  return ilog;
}
