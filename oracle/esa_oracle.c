/*
  esa_oracle.c -- TEST INFRASTRUCTURE ONLY (see esa_oracle.h).

  CPU restatement of the ordering rule and table layouts of
  `gt suffixerator`.  Plain C, single thread, no cleverness: the point is to
  be obviously right, not fast.  Reference citations are to /root/reference.
*/
#include "esa_oracle.h"
#include <ctype.h>
#include <dlfcn.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define ISSPECIAL(c) ((c) >= 254u)   /* src/core/chardef.h:60 */

/* ------------------------------------------------------------------ */
/* alphabets: src/core/alphabet.c:84-91 (domains), :345-356 (DNA map),
   :480-503 (protein map) */
static void dna_symbolmap(uint8_t *map)
{
  const char *wild = "nsywrkvbdhmNSYWRKVBDHM";
  memset(map, 253, 256);               /* UNDEFCHAR, chardef.h:37 */
  map['a'] = map['A'] = 0;
  map['c'] = map['C'] = 1;
  map['g'] = map['G'] = 2;
  map['t'] = map['T'] = 3;
  map['u'] = map['U'] = 3;
  for (; *wild; wild++) map[(unsigned char) *wild] = ORA_WILDCARD;
}

static void protein_symbolmap(uint8_t *map)
{
  const char *aa = "LVIFKREDAGSTNQYWPHMC", *wild = "XUBZJO*-";
  unsigned i;
  memset(map, 253, 256);
  for (i = 0; aa[i]; i++) map[(unsigned char) aa[i]] = (uint8_t) i;
  for (; *wild; wild++) map[(unsigned char) *wild] = ORA_WILDCARD;
}

/* FASTA scanning as in src/core/sequence_buffer_fasta.c:44-170: '>' outside a
   description starts a description that runs to the next newline; the first
   '>' emits nothing, every later one emits SEPARATOR; white space is skipped;
   symbols go through the symbol map (sequence_buffer_inline.h:26-58).
   Empty sequences are rejected as in src/core/encseq_charproc.gen:112-117. */
/* "*.gz" is read through zlib, chosen by the file name as in
   src/core/file.c:42-53; the text is spooled to a temporary file so that the
   reader below stays a plain stdio loop */
static FILE *open_input(const char *path)
{
  const size_t len = strlen(path);
  if (len >= 4 && strcmp(path + len - 3, ".gz") == 0) {
    gzFile gz = gzopen(path, "rb");
    FILE *tmp = gz != NULL ? tmpfile() : NULL;
    char buf[1 << 15];
    int got;
    if (tmp == NULL) { if (gz != NULL) gzclose(gz); return NULL; }
    while ((got = gzread(gz, buf, sizeof buf)) > 0) fwrite(buf, 1, (size_t) got, tmp);
    gzclose(gz);
    rewind(tmp);
    return tmp;
  }
  if (len >= 5 && strcmp(path + len - 4, ".bz2") == 0) {
    /* libbz2 without a header in this image: bound at run time */
    typedef void *(*open_fn)(const char *, const char *);
    typedef int (*read_fn)(void *, void *, int);
    typedef void (*close_fn)(void *);
    void *lib = dlopen("libbz2.so.1.0", RTLD_NOW);
    open_fn bzopen = lib ? (open_fn) dlsym(lib, "BZ2_bzopen") : NULL;
    read_fn bzread = lib ? (read_fn) dlsym(lib, "BZ2_bzread") : NULL;
    close_fn bzclose = lib ? (close_fn) dlsym(lib, "BZ2_bzclose") : NULL;
    void *bz = bzopen && bzread && bzclose ? bzopen(path, "rb") : NULL;
    FILE *tmp = bz != NULL ? tmpfile() : NULL;
    char buf[1 << 15];
    int got;
    if (tmp == NULL) { if (bz != NULL) bzclose(bz); return NULL; }
    while ((got = bzread(bz, buf, (int) sizeof buf)) > 0) fwrite(buf, 1, (size_t) got, tmp);
    bzclose(bz);
    rewind(tmp);
    return tmp;
  }
  return fopen(path, "rb");
}

int ora_encode_fasta(const char *path, int protein, uint8_t **encout,
                     uint64_t *nout, char *err, size_t errlen)
{
  FILE *fp = open_input(path);
  uint8_t map[256], *enc;
  uint64_t n = 0, cap = 1 << 16, line = 1, curlen = 0;
  int c, indesc = 0, first = 1;

  if (fp == NULL) {
    snprintf(err, errlen, "cannot open file '%s'", path);
    return -1;
  }
  if (protein) protein_symbolmap(map); else dna_symbolmap(map);
  enc = malloc(cap);
  c = fgetc(fp);
  if (c != EOF) ungetc(c, fp);
  if (c == '@') {
    /* FASTQ blocks, src/core/seq_iterator_fastq.c:96-305: "@name", sequence
       up to '+', "+[name]", as many quality characters as symbols */
    for (;;) {
      uint64_t nsym = 0, nq = 0;
      c = fgetc(fp);
      if (c == EOF) break;
      if (c != '@') goto fastqerr;
      while ((c = fgetc(fp)) != EOF && c != '\n') ;
      if (c == EOF) goto fastqerr;
      if (!first) { if (n + 1 >= cap) { cap *= 2; enc = realloc(enc, cap); } enc[n++] = ORA_SEPARATOR; }
      first = 0;
      while ((c = fgetc(fp)) != EOF && c != '+') {
        if (c == '\n' || c == ' ') continue;
        if (map[c & 255] == 253) {
          snprintf(err, errlen, "illegal character '%c': file \"%s\"", c, path);
          free(enc); fclose(fp);
          return -1;
        }
        if (n + 1 >= cap) { cap *= 2; enc = realloc(enc, cap); }
        enc[n++] = map[c & 255];
        nsym++;
      }
      if (c == EOF || nsym == 0) goto fastqerr;
      while ((c = fgetc(fp)) != EOF && c != '\n') ;
      while (nq < nsym && (c = fgetc(fp)) != EOF)
        if (c != '\n' && c != ' ') nq++;
      if (nq < nsym || fgetc(fp) != '\n') goto fastqerr;
    }
    fclose(fp);
    if (first) goto fastqerr2;
    *encout = enc;
    *nout = n;
    return 0;
fastqerr:
    fclose(fp);
fastqerr2:
    snprintf(err, errlen, "malformed FASTQ file '%s'", path);
    free(enc);
    return -1;
  }
  while ((c = fgetc(fp)) != EOF) {
    if (indesc) {
      if (c == '\n') { line++; indesc = 0; }
      continue;
    }
    if (c == '\n') line++;
    if (isspace(c)) continue;
    if (n + 1 >= cap) { cap *= 2; enc = realloc(enc, cap); }
    if (c == '>') {
      if (first) first = 0;
      else {
        if (curlen == 0) goto emptyseq;
        enc[n++] = ORA_SEPARATOR;
        curlen = 0;
      }
      indesc = 1;
    } else {
      uint8_t code = map[c & 255];
      if (code == 253) {
        snprintf(err, errlen, "illegal character '%c': file \"%s\", line %llu",
                 c, path, (unsigned long long) line);
        free(enc); fclose(fp);
        return -1;
      }
      enc[n++] = code;
      curlen++;
    }
  }
  fclose(fp);
  if (first) {
    snprintf(err, errlen, "no sequences in multiple fasta file(s) %s ...",
             path);
    free(enc);
    return -1;
  }
  if (curlen == 0) {
emptyseq:
    snprintf(err, errlen, "file '%s' contains an empty sequence", path);
    free(enc);
    return -1;
  }
  *encout = enc;
  *nout = n;
  return 0;
}

/* readmodes: src/core/readmode_api.h:24-27; complement of a DNA code is
   3 - code (a<->t, c<->g), specials are their own complement */
void ora_apply_readmode(uint8_t *enc, uint64_t n, int readmode)
{
  uint64_t i;
  if (readmode == 1 || readmode == 3)
    for (i = 0; i < n / 2; i++) {
      uint8_t t = enc[i]; enc[i] = enc[n - 1 - i]; enc[n - 1 - i] = t;
    }
  if (readmode == 2 || readmode == 3)
    for (i = 0; i < n; i++) if (enc[i] < 4) enc[i] = (uint8_t) (3 - enc[i]);
}

uint8_t *ora_mirror(const uint8_t *enc, uint64_t n)
{
  uint8_t *m = malloc(2 * n + 1);
  uint64_t i;
  memcpy(m, enc, n);
  m[n] = ORA_SEPARATOR;
  for (i = 0; i < n; i++) {
    uint8_t c = enc[n - 1 - i];
    m[n + 1 + i] = c < 4 ? (uint8_t) (3 - c) : c;
  }
  return m;
}

/* ------------------------------------------------------------------ */
/* range bookkeeping: src/core/encseq.c:5061-5074 */
static uint64_t stored_ranges(uint64_t len, uint64_t maxrangevalue)
{
  if (maxrangevalue == UINT32_MAX) return 1;
  if (len <= maxrangevalue + 1) return 1;
  if (len % (maxrangevalue + 1) == 0) return len / (maxrangevalue + 1);
  return 1 + len / (maxrangevalue + 1);
}

/* src/core/encseq.c:924-949 */
static uint64_t sw_table_size(int kind, uint64_t totallength, uint64_t items)
{
  static const uint64_t width[3] = {1, 2, 4},
                        maxv[3] = {UCHAR_MAX, USHRT_MAX, UINT32_MAX};
  if (items == 0) return 0;
  return 2 * width[kind] * items + 8 * (totallength / maxv[kind] + 1);
}

/* specialcharinfo as accumulated in src/core/encseq_charproc.gen and
   src/core/encseq.c:5690-5745; the choice of the stored-range variant follows
   determinesmallestrep, src/core/encseq_access_type.c:95-131 (all terms of
   gt_encseq_determine_size that do not depend on the access type cancel). */
void ora_seqstats_compute(const uint8_t *enc, uint64_t n, uint32_t numofchars,
                          uint64_t lengthofdbfilenames, uint64_t numofdbfiles,
                          ora_seqstats *st)
{
  uint64_t i, sp_tab[3] = {0, 0, 0}, wc_tab[3] = {0, 0, 0}, runsp = 0,
           runwc = 0, seqlen = 0, firstlen = 0;
  int k, spprefix = 1, wcprefix = 1, eqlen = 1, nseq = 0;
  static const uint64_t maxv[3] = {UCHAR_MAX, USHRT_MAX, UINT32_MAX};

  (void) lengthofdbfilenames; (void) numofdbfiles;
  memset(st, 0, sizeof *st);
  st->totallength = n;
  st->numofchars = numofchars;
  st->numofsequences = 1;
  for (i = 0; i <= n; i++) {
    unsigned c = i < n ? enc[i] : 0;        /* sentinel: a letter ends runs */
    int issp = i < n && ISSPECIAL(c), iswc = i < n && c == ORA_WILDCARD;
    if (issp) {
      st->specialcharacters++; runsp++;
      if (spprefix) st->lengthofspecialprefix++;
    } else {
      if (i < n) spprefix = 0;
      if (runsp > 0) {
        if (i == n) st->lengthofspecialsuffix = runsp;
        st->realspecialranges++;
        for (k = 0; k < 3; k++) sp_tab[k] += stored_ranges(runsp, maxv[k]);
        runsp = 0;
      }
    }
    if (iswc) {
      st->wildcards++; runwc++;
      if (wcprefix) st->lengthofwildcardprefix++;
    } else {
      if (i < n) wcprefix = 0;
      if (runwc > 0) {
        if (i == n) st->lengthofwildcardsuffix = runwc;
        st->realwildcardranges++;
        for (k = 0; k < 3; k++) wc_tab[k] += stored_ranges(runwc, maxv[k]);
        runwc = 0;
      }
    }
    if (i < n && c == ORA_SEPARATOR) {
      st->numofsequences++;
      if (nseq == 0) firstlen = seqlen; else if (seqlen != firstlen) eqlen = 0;
      nseq++; seqlen = 0;
    } else if (i < n) seqlen++;
  }
  (void) eqlen; (void) firstlen;
  /* the stored-range counts are those of the smallest of the three table
     widths, whatever access type the encoder picks afterwards and for every
     alphabet (doupdatesumranges, src/core/encseq.c:5215-5256) */
  {
    uint64_t cmin = 0;
    for (k = 0; k < 3; k++) {
      uint64_t tmp = sw_table_size(k, n, wc_tab[k]);
      if (k == 0 || tmp < cmin) {
        cmin = tmp;
        st->specialranges = sp_tab[k];
        st->wildcardranges = wc_tab[k];
      }
    }
  }
}

void ora_seqstats_mirror(ora_seqstats *st, int last_symbol_is_wildcard)
{
  st->totallength = 2 * st->totallength + 1;
  st->specialcharacters = 2 * st->specialcharacters + 1;
  if (last_symbol_is_wildcard) {
    st->specialranges = 2 * st->specialranges - 1;
    st->realspecialranges = 2 * st->realspecialranges - 1;
  } else {
    st->specialranges = 2 * st->specialranges + 1;
    st->realspecialranges = 2 * st->realspecialranges + 1;
  }
  st->wildcards *= 2;
  st->wildcardranges *= 2;
  st->realwildcardranges *= 2;
  st->numofsequences *= 2;
}

/* ------------------------------------------------------------------ */
/* src/match/initbasepower.c:23-34 */
static unsigned maxbasepower(unsigned numofchars)
{
  uint64_t minfailure = UINT64_MAX / numofchars, thepower = 1;
  unsigned i;
  for (i = 0; thepower < minfailure; i++) thepower *= numofchars;
  return i;
}

static uint64_t ipow(uint64_t b, unsigned e)
{
  uint64_t r = 1;
  while (e--) r *= b;
  return r;
}

/* src/match/bcktab.c:239-324 with withspecialsuffixes=true */
static uint64_t bcktab_sizeoftable(unsigned numofchars, unsigned k,
                                   uint64_t maxvalue)
{
  uint64_t w = maxvalue <= UINT_MAX ? 4 : 8, size, counters = 0;
  unsigned idx;
  size = w * (ipow(numofchars, k) + 1);
  size += w * ipow(numofchars, k - 1);
  if (k > 2)
    for (idx = 1; idx < k - 1; idx++) counters += ipow(numofchars, idx);
  size += w * counters;
  return size;
}

/* src/match/sfx-apfxlen.c:49-109, multiplier 0.25 (sfx-apfxlen.h:22) */
uint32_t ora_recommended_prefixlength(uint32_t numofchars, uint64_t n)
{
  unsigned k, mbp;
  for (k = 1; ; k++) {
    uint64_t sizeofrep = bcktab_sizeoftable(numofchars, k, n + 1);
    if ((double) sizeofrep / 0.25 > (double) (uint64_t) n) break;
  }
  k--;
  if (k == 0) return 1;
  mbp = maxbasepower(numofchars);
  return mbp >= 1 && mbp < k ? mbp : k;
}

/* ------------------------------------------------------------------ */
/* The ordering rule: a special at position p (and the virtual
   end at p=n) is the unique symbol 256+p, larger than every letter.
   src/core/encseq.h:640 GT_UNIQUEINT, src/match/sfx-bentsedg.c:42-50,75-80,
   src/core/encseq.c:6449-6530 (both special => compare positions). */
static const uint8_t *g_enc;
static uint64_t g_n;

static int suffix_cmp(const void *pa, const void *pb)
{
  uint64_t p = *(const uint64_t *) pa, q = *(const uint64_t *) pb;
  if (p == q) return 0;
  for (;; p++, q++) {
    int sp = p >= g_n || ISSPECIAL(g_enc[p]),
        sq = q >= g_n || ISSPECIAL(g_enc[q]);
    if (sp || sq) {
      if (sp && sq) return p < q ? -1 : 1;
      return sp ? 1 : -1;
    }
    if (g_enc[p] != g_enc[q]) return g_enc[p] < g_enc[q] ? -1 : 1;
  }
}

/* Layout of .suf (src/match/sfx-suffixgetset.c:586-690,
   src/match/sfx-suffixer.c:2184-2198): the rule above puts every suffix that
   starts with a special behind all others, in text order, and n last; so only
   the others need sorting. */
void ora_suffix_array(const uint8_t *enc, uint64_t n, uint64_t *sa)
{
  uint64_t i, m = 0, t;
  for (i = 0; i < n; i++) if (!ISSPECIAL(enc[i])) sa[m++] = i;
  g_enc = enc; g_n = n;
  qsort(sa, m, sizeof *sa, suffix_cmp);
  t = m;
  for (i = 0; i < n; i++) if (ISSPECIAL(enc[i])) sa[t++] = i;
  sa[t] = n;
}

/* LCP counts matching non-special symbols only (SURVEY 0.3;
   src/match/sfx-linlcp.c:74-129 `withspecial`, src/core/encseq.c:6449) */
static uint64_t lcp_from(const uint8_t *enc, uint64_t n, uint64_t p,
                         uint64_t q, uint64_t l)
{
  while (p + l < n && q + l < n && !ISSPECIAL(enc[p + l]) &&
         enc[p + l] == enc[q + l])
    l++;
  return l;
}

void ora_lcp_direct(const uint8_t *enc, uint64_t n, const uint64_t *sa,
                    uint64_t *lcp)
{
  uint64_t i;
  lcp[0] = 0;
  for (i = 1; i <= n; i++) lcp[i] = lcp_from(enc, n, sa[i - 1], sa[i], 0);
}

/* Kasai et al. as in src/match/sfx-linlcp.c:74-129 */
void ora_lcp_kasai(const uint8_t *enc, uint64_t n, const uint64_t *sa,
                   uint64_t *lcp)
{
  uint64_t *isa = malloc((n + 1) * sizeof *isa), i, h = 0;
  for (i = 0; i <= n; i++) isa[sa[i]] = i;
  lcp[0] = 0;
  for (i = 0; i <= n; i++) {
    uint64_t r = isa[i];
    if (r == 0) { h = 0; continue; }
    h = lcp_from(enc, n, i, sa[r - 1], h);
    lcp[r] = h;
    if (h > 0) h--;
  }
  free(isa);
}

/* src/match/sfx-run.c:173-210 */
void ora_bwt(const uint8_t *enc, uint64_t n, const uint64_t *sa, uint8_t *bwt)
{
  uint64_t i;
  for (i = 0; i <= n; i++)
    bwt[i] = sa[i] == 0 ? ORA_UNDEFBWT : enc[sa[i] - 1];
}

/* src/match/sfx-lcpvalues.c:371-433, src/match/lcpoverflow.h:24-30 */
uint64_t ora_lcp_to_bytes(const uint64_t *lcp, uint64_t nplus1, uint8_t *lcpb,
                          uint64_t *llv)
{
  uint64_t i, pairs = 0;
  for (i = 0; i < nplus1; i++) {
    if (lcp[i] < ORA_LCPOVERFLOW) lcpb[i] = (uint8_t) lcp[i];
    else {
      lcpb[i] = ORA_LCPOVERFLOW;
      if (llv != NULL) { llv[2 * pairs] = i; llv[2 * pairs + 1] = lcp[i]; }
      pairs++;
    }
  }
  return pairs;
}

/* .prj numbers that depend on the tables:
   longest        src/match/sfx-suffixgetset.c:241-245
   maxbranchdepth src/match/sfx-lcpvalues.c:153-156,198-201,395-398 (all
                  entries)
   lcptabsum      src/match/sfx-lcpvalues.c:414, reached only from :662, i.e.
                  only entries of suffixes with >= prefixlength leading
                  non-special symbols */
void ora_esastats_compute(const uint8_t *enc, uint64_t n, const uint64_t *sa,
                          const uint64_t *lcp, uint32_t prefixlength,
                          ora_esastats *st)
{
  uint64_t i;
  memset(st, 0, sizeof *st);
  st->numberofallsortedsuffixes = n + 1;
  st->prefixlength = prefixlength;
  for (i = 0; i <= n; i++) {
    uint64_t p = sa[i], k;
    if (p == 0) st->longest = i;
    if (lcp == NULL) continue;
    if (lcp[i] >= ORA_LCPOVERFLOW) st->largelcpvalues++;
    if (lcp[i] > st->maxbranchdepth) st->maxbranchdepth = lcp[i];
    for (k = 0; k < prefixlength; k++)
      if (p + k >= n || ISSPECIAL(enc[p + k])) break;
    if (k == prefixlength) st->lcptabsum += (double) lcp[i];
  }
}

/* Linear-time suffix array check in the spirit of the reference's
   gt_suftab_lightweightcheck (src/match/sfx-lwcheck.c:181-337): sa is a
   permutation of 0..n; neighbours are ordered by first symbol, and where the
   first symbols are equal letters, by the rank of the next suffix. */
int ora_check_suffix_array(const uint8_t *enc, uint64_t n, const uint64_t *sa,
                           uint64_t *where)
{
  uint64_t *isa = malloc((n + 1) * sizeof *isa), i;
  int rc = 0;
  for (i = 0; i <= n; i++) isa[i] = UINT64_MAX;
  for (i = 0; i <= n && !rc; i++) {
    if (sa[i] > n || isa[sa[i]] != UINT64_MAX) { rc = 1; *where = i; }
    else isa[sa[i]] = i;
  }
  for (i = 1; i <= n && !rc; i++) {
    uint64_t a = sa[i - 1], b = sa[i];
    uint64_t ka = a >= n || ISSPECIAL(enc[a]) ? 256 + a : enc[a],
             kb = b >= n || ISSPECIAL(enc[b]) ? 256 + b : enc[b];
    if (ka > kb) { rc = 2; *where = i; }
    else if (ka == kb && isa[a + 1] > isa[b + 1]) { rc = 3; *where = i; }
  }
  free(isa);
  return rc;
}

/* src/match/sfx-outprj.c:38-83 */
int ora_write_prj(const char *path, const ora_seqstats *ss,
                  const ora_esastats *es, int with_lcp, int readmode,
                  int mirrored)
{
  FILE *fp = fopen(path, "wb");
  if (fp == NULL) return -1;
#define OUT(F) fprintf(fp, #F "=%llu\n", (unsigned long long) ss->F)
  OUT(totallength); OUT(specialcharacters); OUT(specialranges);
  OUT(realspecialranges); OUT(lengthofspecialprefix);
  OUT(lengthofspecialsuffix); OUT(wildcards); OUT(wildcardranges);
  OUT(realwildcardranges); OUT(lengthofwildcardprefix);
  OUT(lengthofwildcardsuffix); OUT(numofsequences);
#undef OUT
  fprintf(fp, "numofdbsequences=%llu\n",
          (unsigned long long) ss->numofsequences);
  fprintf(fp, "numofquerysequences=0\n");
  fprintf(fp, "numberofallsortedsuffixes=%llu\n",
          (unsigned long long) es->numberofallsortedsuffixes);
  fprintf(fp, "longest=%llu\n", (unsigned long long) es->longest);
  fprintf(fp, "prefixlength=%u\n", es->prefixlength);
  fprintf(fp, "largelcpvalues=%llu\n",
          (unsigned long long) (with_lcp ? es->largelcpvalues : 0));
  fprintf(fp, "averagelcp=%.2f\n",
          with_lcp ? es->lcptabsum / (double) es->numberofallsortedsuffixes
                   : 0.0);
  fprintf(fp, "maxbranchdepth=%llu\n",
          (unsigned long long) (with_lcp ? es->maxbranchdepth : 0));
  fprintf(fp, "integersize=64\nlittleendian=1\nreadmode=%d\nmirrored=%d\n",
          readmode, mirrored);
  fclose(fp);
  return 0;
}

/* The three sections of the bucket table for prefix length k, restated from
   what the reference's counting and insertion passes leave in GtBcktab
   (src/match/bcktab.c:55-81,1274-1304; special suffixes
   src/match/sfx-suffixer.c:476-516, sfx-enumcodes.c:100-222): every suffix that
   does not start with a special belongs to the bucket of its first k symbols,
   a suffix with fewer letters before a special to the bucket of its letters
   padded with the largest letter. */
void ora_bcktab(const uint8_t *enc, uint64_t n, uint32_t sigma, uint32_t k,
                uint32_t *leftborder, uint32_t *countspecialcodes,
                uint32_t *distpfxidx)
{
  uint64_t codes = 1, special = 1, dist = 0, pw = 1, sum = 0;
  for (uint32_t j = 0; j < k; j++) codes *= sigma;
  for (uint32_t j = 0; j + 1 < k; j++) special *= sigma;
  for (uint32_t j = 1; j + 1 < k; j++) { pw *= sigma; dist += pw; }
  memset(leftborder, 0, 4 * (codes + 1));
  memset(countspecialcodes, 0, 4 * special);
  if (dist > 0) memset(distpfxidx, 0, 4 * dist);
  for (uint64_t p = 0; p < n; p++) {
    uint64_t code = 0, prefix = 0;
    uint32_t letters = 0;
    while (letters < k && p + letters < n && enc[p + letters] < ORA_WILDCARD) {
      code = code * sigma + enc[p + letters];
      letters++;
    }
    if (letters == 0) continue;                  /* starts with a special */
    prefix = code;
    for (uint32_t j = letters; j < k; j++) code = code * sigma + (sigma - 1);
    leftborder[code]++;
    if (letters < k) {
      countspecialcodes[code / sigma]++;
      if (letters + 2 <= k) {
        uint64_t off = 0, q = sigma;
        for (uint32_t l = 1; l < letters; l++) { off += q; q *= sigma; }
        distpfxidx[off + prefix]++;
      }
    }
  }
  for (uint64_t c = 0; c <= codes; c++) {        /* counts -> left borders */
    const uint32_t cnt = leftborder[c];
    leftborder[c] = (uint32_t) sum;
    sum += cnt;
  }
}
