/* esq_host.c -- INDEX.esq (the encoded sequence as GenomeTools' own tools map
   it) and INDEX.ssp (sequence separator positions), byte for byte as
   gt_encseq_encoder_encode writes them for DNA and protein input without
   -lossless:
     file layout        gt_encseq_assign_header_mapspec / _sequence_mapspec,
                        src/core/encseq.c:1195-1402; every section is padded to
                        8 bytes (src/core/mapspec.c:350-457)
     access type        src/core/encseq_access_type.c:96-162
     two bit encoding   src/core/encseq.c:85-99, 2594-2607, 2822-2835,
                        src/core/accspecialrange.gen:227-245
     special bits       src/core/encseq.c:2771-2776, 2816-2820
     wildcard tables    src/core/accspecialrange.gen:29-261
     bit packing        src/core/encseq.c:2324-2447 (protein)
     .ssp               src/core/encseq.c:1714-1910, 910-981 */
#include "host_internal.h"
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { FILE *fp; uint64_t off; int failed; } outfile;

static void put(outfile *o, const void *p, uint64_t bytes)
{
  static const uint8_t zero[8] = {0};
  if (bytes == 0) return;               /* empty sections are not padded */
  if (fwrite(p, 1, bytes, o->fp) != bytes) o->failed = 1;
  o->off += bytes;
  if (o->off % 8 != 0) {
    const uint64_t pad = 8 - o->off % 8;
    if (fwrite(zero, 1, pad, o->fp) != pad) o->failed = 1;
    o->off += pad;
  }
}

static void put_word(outfile *o, uint64_t w) { put(o, &w, sizeof w); }

/* one table of special ranges: page-relative start, length - 1, and per page
   the number of ranges that start up to its end */
typedef struct {
  int kind;                 /* 0 uchar, 1 ushort, 2 uint32 */
  uint64_t items, numofpages;
  void *positions, *rangelengths;
  uint64_t *endidxinpage;
} swtable;

static const uint64_t sw_maxv[3] = {UCHAR_MAX, USHRT_MAX, UINT32_MAX};
static const size_t sw_width[3] = {1, 2, 4};

static void sw_free(swtable *t)
{
  free(t->positions); free(t->rangelengths); free(t->endidxinpage);
  t->positions = t->rangelengths = NULL; t->endidxinpage = NULL;
}

static void sw_store(void *tab, int kind, uint64_t idx, uint64_t v)
{
  if (kind == 0) ((uint8_t *) tab)[idx] = (uint8_t) v;
  else if (kind == 1) ((uint16_t *) tab)[idx] = (uint16_t) v;
  else ((uint32_t *) tab)[idx] = (uint32_t) v;
}

/* The table of `runs` maximal runs (start, length; length NULL: single
   positions) for a sequence of n symbols, `items` stored ranges expected: a
   run longer than the length field holds is cut into pieces of maxv + 1
   (src/core/accspecialrange.gen:136-161); endidxinpage[p] counts the ranges
   that start at or before the last position of page p (:210-215, 247-251). */
static int sw_build(swtable *t, int kind, uint64_t n, const uint64_t *start,
                    const uint64_t *length, uint64_t runs, uint64_t items)
{
  const uint64_t maxv = sw_maxv[kind];
  uint64_t fill = 0, page = 0;
  memset(t, 0, sizeof *t);
  t->kind = kind; t->items = items;
  t->numofpages = n / maxv + 1;
  t->positions = malloc(sw_width[kind] * (items ? items : 1));
  t->rangelengths = length != NULL ? malloc(sw_width[kind] * (items ? items : 1)) : NULL;
  t->endidxinpage = malloc(sizeof (uint64_t) * t->numofpages);
  if (t->positions == NULL || t->endidxinpage == NULL ||
      (length != NULL && t->rangelengths == NULL)) {
    sw_free(t);
    return -1;
  }
  for (uint64_t r = 0; r < runs; r++) {
    uint64_t pos = start[r], left = length != NULL ? length[r] : 1;
    while (left > 0) {
      const uint64_t piece = left < maxv + 1 ? left : maxv + 1;
      /* pages that end before this range starts are complete */
      while (page < t->numofpages && page * (maxv + 1) + maxv < pos)
        t->endidxinpage[page++] = fill;
      if (fill < items) {
        sw_store(t->positions, kind, fill, pos & maxv);
        if (length != NULL) sw_store(t->rangelengths, kind, fill, piece - 1);
      }
      fill++;
      pos += piece; left -= piece;
    }
  }
  while (page < t->numofpages) t->endidxinpage[page++] = fill;
  if (fill != items) { sw_free(t); return -2; }
  return 0;
}

static void sw_put(outfile *o, const swtable *t, int withlengths)
{
  if (t->items == 0) return;
  put(o, t->positions, sw_width[t->kind] * t->items);
  if (withlengths) put(o, t->rangelengths, sw_width[t->kind] * t->items);
  put(o, t->endidxinpage, sizeof (uint64_t) * t->numofpages);
}

/* width of the separator table: the smallest of the three
   (src/core/encseq.c:1714-1736) */
static int ssp_kind(uint64_t n, uint64_t numofseparators)
{
  uint64_t best = gtamd_swtable_bytes(0, 0, n, numofseparators), size;
  int kind = 0;
  size = gtamd_swtable_bytes(1, 0, n, numofseparators);
  if (size < best) { best = size; kind = 1; }
  size = gtamd_swtable_bytes(2, 0, n, numofseparators);
  if (size < best) kind = 2;
  return kind;
}

/* number of distinct original characters and the size of the largest class of
   characters mapped to one code (src/core/encseq.c:5275-5358) */
static void original_classes(const uint64_t *dist, const uint8_t *enc_of_char,
                             uint64_t *numofallchars, uint8_t *maxsubalphasize)
{
  uint64_t classsize[256];
  memset(classsize, 0, sizeof classsize);
  *numofallchars = 0; *maxsubalphasize = 0;
  for (int c = 1; c < 128; c++)
    if (dist[c] > 0 && enc_of_char[c] != GTAMD_SEPARATOR) {
      classsize[enc_of_char[c]]++;
      (*numofallchars)++;
    }
  for (int k = 0; k < 255; k++)
    if (classsize[k] > *maxsubalphasize) *maxsubalphasize = (uint8_t) classsize[k];
}

unsigned gtamd_least_probable(const gtamd_seqanalysis *an)
{
  unsigned least = 0;
  for (unsigned k = 1; k < an->ss.numofchars; k++)
    if (an->chardist[k] < an->chardist[least]) least = k;
  return least;
}

int gtamd_force_sat(gtamd_seqanalysis *an, const char *satname, int notdna, char *err,
                    size_t errlen)
{
  int sat;
  if (satname == NULL) return 0;
  if (gtamd_parse_sat(satname, notdna, &sat, err, errlen) != 0) return -1;
  if (gtamd_choose_access_type(an, an->sp_tab, an->wc_tab, sat) != 0) {
    /* src/core/encseq_access_type.c:185-193 */
    snprintf(err, errlen, "illegal argument \"%s\" to option -sat: %s is only possible for "
             "DNA sequences, if all sequences are of equal length and no sequence contains "
             "a wildcard", satname, satname);
    return -1;
  }
  return 0;
}

void gtamd_esq_needs(const gtamd_seqanalysis *an, int write_ssp, int *twobit,
                     int *specialbits, int *packed, int *wildcardruns,
                     int *separators)
{
  const uint64_t numsep = an->ss.numofsequences - 1;
  const int viatables = an->sat >= GTAMD_SAT_UCHARTABLES;
  *packed = an->sat == GTAMD_SAT_BYTECOMPRESS;
  *twobit = an->sat >= GTAMD_SAT_EQUALLENGTH;
  *specialbits = an->sat == GTAMD_SAT_BITACCESS && (an->sat_wildcardranges > 0 || numsep > 0);
  *wildcardruns = viatables && an->sat_wildcardranges > 0;
  /* the separator table exists for table access types and on request */
  *separators = numsep > 0 && an->sat != GTAMD_SAT_EQUALLENGTH && (write_ssp || viatables);
}

int gtamd_write_esq_sections(const char *indexname, const char *const *paths,
                             size_t numfiles, const gtamd_alphabet *a,
                             const gtamd_seqanalysis *an, const gtamd_encinfo *info,
                             int write_ssp, const gtamd_esq_sections *sec,
                             char *err, size_t errlen)
{
  const uint32_t numofchars = a->numofchars;
  const uint64_t n = an->ss.totallength, numsep = an->ss.numofsequences - 1;
  outfile o = {NULL, 0, 0};
  char path[4096];
  uint64_t lengthofdbfilenames = 0, numofallchars;
  uint8_t maxsubalphasize, *names = NULL;
  swtable wct, sspt;
  int have_wct = 0, have_ssp = 0, rc = -1, need_tb, need_sb, need_pk, need_wc, need_sep;

  memset(&wct, 0, sizeof wct); memset(&sspt, 0, sizeof sspt);
  if (info == NULL || info->numfiles != numfiles) {
    snprintf(err, errlen, "file information of the encoder is missing");
    return -1;
  }
  gtamd_esq_needs(an, write_ssp, &need_tb, &need_sb, &need_pk, &need_wc, &need_sep);
  if ((an->sat == GTAMD_SAT_DIRECTACCESS && sec->plain == NULL && n > 0) ||
      (need_tb && sec->twobit == NULL) || (need_sb && sec->specialbits == NULL) ||
      (need_pk && sec->packed == NULL) || (need_wc && sec->wc_start == NULL) ||
      (need_sep && sec->seppos == NULL)) {
    snprintf(err, errlen, "sequence sections of the encoded sequence are missing");
    return -1;
  }
  original_classes(info->originaldistribution, a->symbolmap, &numofallchars,
                   &maxsubalphasize);
  for (size_t f = 0; f < numfiles; f++) lengthofdbfilenames += strlen(paths[f]) + 1;
  names = malloc(lengthofdbfilenames ? lengthofdbfilenames : 1);
  if (names == NULL) goto nomem;
  for (size_t f = 0, off = 0; f < numfiles; f++) {
    memcpy(names + off, paths[f], strlen(paths[f]) + 1);
    off += strlen(paths[f]) + 1;
  }
  if (need_sep) {
    const int brc = sw_build(&sspt, ssp_kind(n, numsep), n, sec->seppos, NULL, numsep, numsep);
    if (brc == -1) goto nomem;
    if (brc != 0) goto inconsistent;
    have_ssp = 1;
  }
  if (an->sat >= GTAMD_SAT_UCHARTABLES) {
    const int brc = sw_build(&wct, an->sat - GTAMD_SAT_UCHARTABLES, n, sec->wc_start,
                             sec->wc_len, need_wc ? sec->wc_runs : 0, an->sat_wildcardranges);
    if (brc == -1) goto nomem;
    if (brc != 0) goto inconsistent;
    have_wct = 1;
  }

  snprintf(path, sizeof path, "%s.esq", indexname);
  if ((o.fp = fopen(path, "wb")) == NULL) {
    snprintf(err, errlen, "cannot open file '%s' for writing", path);
    goto done;
  }
  {
    const uint8_t is64bit = 1;
    const uint64_t sci[14] = {
      an->ss.specialcharacters, an->ss.specialranges, an->ss.realspecialranges,
      an->ss.lengthofspecialprefix, an->ss.lengthofspecialsuffix, an->ss.wildcards,
      an->ss.wildcardranges, an->ss.realwildcardranges,
      an->ss.lengthofwildcardprefix, an->ss.lengthofwildcardsuffix,
      an->lengthoflongestnonspecial,
      info->exceptioncharacters, 0, info->realexceptionranges };   /* -lossless */
    put(&o, &is64bit, 1);
    put_word(&o, 3);                           /* format version */
    put_word(&o, (uint64_t) an->sat);
    put_word(&o, n);
    put_word(&o, an->ss.numofsequences);
    put_word(&o, numfiles);
    put_word(&o, lengthofdbfilenames);
    put(&o, sci, sizeof sci);
    put_word(&o, an->minseqlen);
    put_word(&o, an->maxseqlen);
    put_word(&o, (uint64_t) a->alphatype);     /* 0 DNA, 1 protein, 2 symbol map */
    put_word(&o, a->lengthofalphadef);
    put(&o, a->alphadef, a->lengthofalphadef);   /* the symbol map text, if any */
    put(&o, names, lengthofdbfilenames);
    put(&o, &maxsubalphasize, 1);
    put_word(&o, numofallchars);
    put(&o, info->filelengthtab, sizeof (gtamd_filelength) * numfiles);
    put(&o, an->chardist, sizeof (uint64_t) * numofchars);
  }
  if (an->sat == GTAMD_SAT_DIRECTACCESS) put(&o, sec->plain, n);
  else if (need_pk) put(&o, sec->packed, (a->bitspersymbol * n + 7) / 8);
  else {
    put(&o, sec->twobit, sizeof (uint64_t) * (n < 32 ? 2 : 2 + (n - 1) / 32));
    if (need_sb) put(&o, sec->specialbits, sizeof (uint64_t) * (1 + (n + 63) / 64));
    if (have_wct) sw_put(&o, &wct, 1);
  }
  if (fclose(o.fp) != 0 || o.failed) {
    snprintf(err, errlen, "cannot write file '%s'", path);
    o.fp = NULL;
    goto done;
  }
  o.fp = NULL;
  if (have_ssp) {
    snprintf(path, sizeof path, "%s.ssp", indexname);
    o.off = 0; o.failed = 0;
    if ((o.fp = fopen(path, "wb")) == NULL) {
      snprintf(err, errlen, "cannot open file '%s' for writing", path);
      goto done;
    }
    sw_put(&o, &sspt, 0);
    if (fclose(o.fp) != 0 || o.failed) {
      snprintf(err, errlen, "cannot write file '%s'", path);
      o.fp = NULL;
      goto done;
    }
    o.fp = NULL;
  }
  rc = 0;
  goto done;
inconsistent:
  snprintf(err, errlen, "range lists do not match the sequence statistics");
  goto done;
nomem:
  snprintf(err, errlen, "out of memory while writing the encoded sequence");
done:
  sw_free(&wct); sw_free(&sspt);
  free(names);
  return rc;
}

int gtamd_write_esq(const char *indexname, const char *const *paths,
                    size_t numfiles, const uint8_t *enc, uint64_t n,
                    int protein, const gtamd_encinfo *info, int write_ssp,
                    char *err, size_t errlen)
{
  return gtamd_write_esq_sat(indexname, paths, numfiles, enc, n, protein, info, write_ssp,
                             NULL, NULL, err, errlen);
}

int gtamd_write_esq_sat(const char *indexname, const char *const *paths,
                        size_t numfiles, const uint8_t *enc, uint64_t n,
                        int protein, const gtamd_encinfo *info, int write_ssp,
                        const char *sat, gtamd_seqstats *ss, char *err, size_t errlen)
{
  gtamd_alphabet a;
  gtamd_alphabet_standard(&a, protein);
  return gtamd_write_esq_alpha(indexname, paths, numfiles, enc, n, &a, info, write_ssp, sat,
                               ss, err, errlen);
}

/* the sequence sections by host loops over the symbols */
int gtamd_write_esq_alpha(const char *indexname, const char *const *paths,
                          size_t numfiles, const uint8_t *enc, uint64_t n,
                          const gtamd_alphabet *a, const gtamd_encinfo *info,
                          int write_ssp, const char *sat, gtamd_seqstats *ss,
                          char *err, size_t errlen)
{
  const uint32_t numofchars = a->numofchars, bits = a->bitspersymbol;
  gtamd_seqanalysis an;
  gtamd_esq_sections sec;
  uint64_t *twobit = NULL, *specialbits = NULL, *wc_start = NULL, *wc_len = NULL,
           *seppos = NULL, nwc = 0, nsep = 0;
  uint8_t *packed = NULL;
  unsigned least;
  int rc = -1, need_tb, need_sb, need_pk, need_wc, need_sep;

  gtamd_analyse_sequence(enc, n, numofchars, &an);
  if (gtamd_force_sat(&an, sat, numofchars != 4, err, errlen) != 0) return -1;
  if (ss != NULL) *ss = an.ss;
  gtamd_esq_needs(&an, write_ssp, &need_tb, &need_sb, &need_pk, &need_wc, &need_sep);
  least = gtamd_least_probable(&an);
  memset(&sec, 0, sizeof sec);
  sec.plain = enc;
  if (need_pk && (packed = calloc((bits * n + 7) / 8 + 2, 1)) == NULL) goto nomem;
  if (need_tb && (twobit = calloc(n < 32 ? 2 : 2 + (n - 1) / 32, 8)) == NULL) goto nomem;
  if (need_sb) {
    if ((specialbits = calloc(1 + (n + 63) / 64, 8)) == NULL) goto nomem;
    for (uint64_t p = n; p < n + 64; p++)
      specialbits[p / 64] |= (uint64_t) 1 << (63 - p % 64);
  }
  if (need_wc) {
    wc_start = malloc(8 * (an.ss.realwildcardranges + 1));
    wc_len = malloc(8 * (an.ss.realwildcardranges + 1));
    if (wc_start == NULL || wc_len == NULL) goto nomem;
  }
  if (need_sep && (seppos = malloc(8 * an.ss.numofsequences)) == NULL) goto nomem;
  for (uint64_t pos = 0; pos < n; pos++) {
    const uint8_t c = enc[pos];
    if (need_wc && c == GTAMD_WILDCARD) {
      if (pos > 0 && enc[pos - 1] == GTAMD_WILDCARD) wc_len[nwc - 1]++;
      else { wc_start[nwc] = pos; wc_len[nwc++] = 1; }
    }
    if (need_sep && c == GTAMD_SEPARATOR) seppos[nsep++] = pos;
    if (twobit == NULL && packed == NULL) continue;      /* direct access */
    if (packed != NULL) {
      /* `bits` per symbol (at most 8), most significant bit first; wildcard
         and separator are the two codes behind the alphabet */
      const unsigned v = c == GTAMD_WILDCARD ? numofchars
                       : c == GTAMD_SEPARATOR ? numofchars + 1 : c;
      const uint64_t bit = (uint64_t) bits * pos;
      const unsigned shift = 16 - bits - (unsigned) (bit % 8);
      packed[bit / 8] |= (uint8_t) ((v << shift) >> 8);
      if (shift < 8) packed[bit / 8 + 1] |= (uint8_t) (v << shift);
    } else {
      uint64_t code;
      if (c < GTAMD_WILDCARD) code = c;
      else if (an.sat == GTAMD_SAT_BITACCESS) code = c == GTAMD_SEPARATOR ? 1 : 0;
      else code = least;
      if (specialbits != NULL && c >= GTAMD_WILDCARD)
        specialbits[pos / 64] |= (uint64_t) 1 << (63 - pos % 64);
      twobit[pos / 32] |= code << (62 - 2 * (pos % 32));
    }
  }
  sec.twobit = twobit; sec.specialbits = specialbits; sec.packed = packed;
  sec.wc_start = wc_start; sec.wc_len = wc_len; sec.wc_runs = nwc;
  sec.seppos = seppos;
  rc = gtamd_write_esq_sections(indexname, paths, numfiles, a, &an, info,
                                write_ssp, &sec, err, errlen);
  goto done;
nomem:
  snprintf(err, errlen, "out of memory while writing the encoded sequence");
done:
  free(twobit); free(specialbits); free(packed); free(wc_start); free(wc_len); free(seppos);
  return rc;
}

/* ---- reading INDEX.esq (+ INDEX.ssp) back: the -ii path ---- */

typedef struct { const uint8_t *p; uint64_t len, off; int bad; } infile;

static const void *take(infile *in, uint64_t bytes)
{
  const void *q = in->p + in->off;
  if (bytes == 0) return q;
  if (in->bad || in->off + bytes > in->len) { in->bad = 1; return NULL; }
  in->off += bytes;
  if (in->off % 8 != 0) in->off += 8 - in->off % 8;
  return q;
}

static uint64_t take_word(infile *in)
{
  const uint64_t *w = take(in, 8);
  return w != NULL ? *w : 0;
}

static int slurp_file(const char *path, uint8_t **data, uint64_t *len)
{
  FILE *fp = fopen(path, "rb");
  long size;
  if (fp == NULL) return -1;
  if (fseek(fp, 0, SEEK_END) != 0 || (size = ftell(fp)) < 0 || fseek(fp, 0, SEEK_SET) != 0) {
    fclose(fp);
    return -1;
  }
  *data = malloc(size > 0 ? (size_t) size : 1);
  if (*data == NULL || fread(*data, 1, (size_t) size, fp) != (size_t) size) {
    free(*data); fclose(fp);
    return -1;
  }
  fclose(fp);
  *len = (uint64_t) size;
  return 0;
}

static uint64_t sw_load(const void *tab, int kind, uint64_t idx)
{
  return kind == 0 ? ((const uint8_t *) tab)[idx]
       : kind == 1 ? ((const uint16_t *) tab)[idx] : ((const uint32_t *) tab)[idx];
}

/* mark the members of a stored range table in enc: entry idx of page p starts
   at p * (maxv + 1) + positions[idx] and covers rangelengths[idx] + 1 symbols
   (a single one without lengths) */
static int sw_apply(infile *in, int kind, int withlengths, uint64_t n, uint64_t items,
                    uint8_t symbol, uint8_t *enc)
{
  const void *positions, *lengths = NULL;
  const uint64_t *endidx, numofpages = n / sw_maxv[kind] + 1;
  uint64_t idx = 0;
  if (items == 0) return 0;
  positions = take(in, sw_width[kind] * items);
  if (withlengths) lengths = take(in, sw_width[kind] * items);
  endidx = take(in, 8 * numofpages);
  if (in->bad) return -1;
  for (uint64_t page = 0; page < numofpages; page++) {
    if (endidx[page] > items || endidx[page] < idx) return -1;
    for (; idx < endidx[page]; idx++) {
      const uint64_t start = page * (sw_maxv[kind] + 1) + sw_load(positions, kind, idx),
                     len = withlengths ? sw_load(lengths, kind, idx) + 1 : 1;
      if (start + len > n) return -1;
      memset(enc + start, symbol, len);
    }
  }
  return idx == items ? 0 : -1;
}

int gtamd_read_esq(const char *indexname, uint8_t **enc_out, uint64_t *n_out,
                   int *protein_out, gtamd_seqstats *ss, char *err, size_t errlen)
{
  gtamd_alphabet a;
  if (gtamd_read_esq_alpha(indexname, enc_out, n_out, &a, ss, err, errlen) != 0) return -1;
  if (a.alphatype > 1) {
    free(*enc_out); *enc_out = NULL;
    gtamd_alphabet_free(&a);
    snprintf(err, errlen, "index '%s' uses a custom alphabet", indexname);
    return -1;
  }
  *protein_out = a.alphatype == 1;
  return 0;
}

int gtamd_read_esq_alpha(const char *indexname, uint8_t **enc_out, uint64_t *n_out,
                         gtamd_alphabet *alpha, gtamd_seqstats *ss, char *err,
                         size_t errlen)
{
  char path[4096];
  uint8_t *data = NULL, *sspdata = NULL, *enc = NULL;
  uint64_t len = 0, ssplen = 0, sat, n, numseq, numfiles, namelen, alphatype,
           alphadeflen, wildcardranges, version;
  const uint64_t *sci;
  infile in;
  uint32_t numofchars;
  int rc = -1;

  memset(alpha, 0, sizeof *alpha);
  snprintf(path, sizeof path, "%s.esq", indexname);
  if (slurp_file(path, &data, &len) != 0) {
    snprintf(err, errlen, "cannot open file '%s'", path);
    return -1;
  }
  in.p = data; in.len = len; in.off = 0; in.bad = 0;
  {
    const uint8_t *is64 = take(&in, 1);
    if (is64 == NULL || *is64 != 1) {
      snprintf(err, errlen, "index '%s' was not written for 64-bit integers", indexname);
      goto done;
    }
  }
  version = take_word(&in);
  sat = take_word(&in);
  n = take_word(&in);
  numseq = take_word(&in);
  numfiles = take_word(&in);
  namelen = take_word(&in);
  sci = take(&in, 14 * 8);
  (void) take_word(&in); (void) take_word(&in);       /* min/max sequence length */
  alphatype = take_word(&in);
  alphadeflen = take_word(&in);
  if (in.bad || version != 3) {
    snprintf(err, errlen, "index '%s' has an unsupported format version", indexname);
    goto done;
  }
  memset(alpha, 0, sizeof *alpha);
  if (alphatype <= 1 && alphadeflen == 0) gtamd_alphabet_standard(alpha, alphatype == 1);
  else if (alphatype == 2 && alphadeflen > 0) {
    const char *def = take(&in, alphadeflen);
    if (in.bad) goto corrupt;
    if (gtamd_alphabet_from_text(def, alphadeflen, path, alpha, err, errlen) != 0) goto done;
  } else goto corrupt;
  numofchars = alpha->numofchars;
  wildcardranges = sci[6];
  (void) take(&in, namelen);
  (void) take(&in, 1);                                 /* maxsubalphasize */
  (void) take_word(&in);                               /* numofallchars */
  (void) take(&in, 16 * numfiles);
  (void) take(&in, 8 * (uint64_t) numofchars);
  if (in.bad || numseq == 0 || n + 1 < numseq) goto corrupt;
  enc = malloc(n ? n : 1);
  if (enc == NULL) { snprintf(err, errlen, "out of memory"); goto done; }

  if (sat == GTAMD_SAT_DIRECTACCESS) {
    const uint8_t *plain = take(&in, n);
    if (in.bad) goto corrupt;
    memcpy(enc, plain, n);
  } else if (sat == GTAMD_SAT_BYTECOMPRESS) {
    const unsigned bits = alpha->bitspersymbol;
    const uint8_t *packed = take(&in, (bits * n + 7) / 8);
    if (in.bad) goto corrupt;
    for (uint64_t pos = 0; pos < n; pos++) {
      const uint64_t bit = bits * pos;
      unsigned w = (unsigned) packed[bit / 8] << 8;
      if (bit / 8 + 1 < (bits * n + 7) / 8) w |= packed[bit / 8 + 1];
      w = (w >> (16 - bits - (unsigned) (bit % 8))) & ((1u << bits) - 1);
      if (w > numofchars + 1) goto corrupt;
      enc[pos] = w == numofchars ? GTAMD_WILDCARD
               : w == numofchars + 1 ? GTAMD_SEPARATOR : (uint8_t) w;
    }
  } else if (sat <= GTAMD_SAT_UINT32TABLES) {
    const uint64_t units = n < 32 ? 2 : 2 + (n - 1) / 32,
                   *twobit = take(&in, 8 * units);
    if (numofchars != 4 || in.bad) goto corrupt;
    for (uint64_t pos = 0; pos < n; pos++)
      enc[pos] = (uint8_t) ((twobit[pos / 32] >> (62 - 2 * (pos % 32))) & 3);
    if (sat == GTAMD_SAT_EQUALLENGTH) {
      /* numseq sequences of one length, a separator after each but the last */
      const uint64_t seqlen = (n - (numseq - 1)) / numseq;
      if (seqlen * numseq + numseq - 1 != n) goto corrupt;
      for (uint64_t s = 0; s + 1 < numseq; s++) enc[s * (seqlen + 1) + seqlen] = GTAMD_SEPARATOR;
    } else if (sat == GTAMD_SAT_BITACCESS) {
      if (wildcardranges > 0 || numseq > 1) {
        const uint64_t *specialbits = take(&in, 8 * (1 + (n + 63) / 64));
        if (in.bad) goto corrupt;
        for (uint64_t pos = 0; pos < n; pos++)
          if ((specialbits[pos / 64] >> (63 - pos % 64)) & 1)
            enc[pos] = enc[pos] == 1 ? GTAMD_SEPARATOR : GTAMD_WILDCARD;
      }
    } else {
      const int kind = (int) sat - GTAMD_SAT_UCHARTABLES;
      if (sw_apply(&in, kind, 1, n, wildcardranges, GTAMD_WILDCARD, enc) != 0) goto corrupt;
      if (numseq > 1) {
        /* the separators of table access types live in INDEX.ssp */
        infile ssp;
        snprintf(path, sizeof path, "%s.ssp", indexname);
        if (slurp_file(path, &sspdata, &ssplen) != 0) {
          snprintf(err, errlen, "cannot open file '%s'", path);
          goto done;
        }
        ssp.p = sspdata; ssp.len = ssplen; ssp.off = 0; ssp.bad = 0;
        if (sw_apply(&ssp, ssp_kind(n, numseq - 1), 0, n, numseq - 1, GTAMD_SEPARATOR, enc) != 0) {
          snprintf(err, errlen, "file '%s' does not fit index '%s'", path, indexname);
          goto done;
        }
      }
    }
  } else goto corrupt;
  *enc_out = enc; enc = NULL;
  *n_out = n;
  if (ss != NULL) {
    /* GtSpecialcharinfo as stored (src/core/chardef.h:91-116) */
    memset(ss, 0, sizeof *ss);
    ss->totallength = n; ss->numofsequences = numseq; ss->numofchars = numofchars;
    ss->specialcharacters = sci[0]; ss->specialranges = sci[1];
    ss->realspecialranges = sci[2]; ss->lengthofspecialprefix = sci[3];
    ss->lengthofspecialsuffix = sci[4]; ss->wildcards = sci[5];
    ss->wildcardranges = sci[6]; ss->realwildcardranges = sci[7];
    ss->lengthofwildcardprefix = sci[8]; ss->lengthofwildcardsuffix = sci[9];
  }
  rc = 0;
  goto done;
corrupt:
  snprintf(err, errlen, "index file '%s.esq' is truncated or inconsistent", indexname);
done:
  free(enc); free(data); free(sspdata);
  if (rc != 0) gtamd_alphabet_free(alpha);
  return rc;
}
