/* encseq_host.c -- FASTA reading, symbol encoding and sequence statistics of
   the host layer (see include/gtamd_host.h for the reference interfaces). */
#include "gtamd_host.h"
#include "gtamd_md5.h"
#include "host_internal.h"
#include <ctype.h>
#include <dlfcn.h>
#include <limits.h>
#include <pthread.h>
#include <unistd.h>
#include <zlib.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { SYM_UNDEF = 253 };

static void build_symbolmap(uint8_t map[256], int protein)
{
  memset(map, SYM_UNDEF, 256);
  if (protein) {
    static const char letters[] = "LVIFKREDAGSTNQYWPHMC", wild[] = "XUBZJO*-";
    for (int i = 0; letters[i]; i++) map[(uint8_t) letters[i]] = (uint8_t) i;
    for (int i = 0; wild[i]; i++) map[(uint8_t) wild[i]] = GTAMD_WILDCARD;
  } else {
    static const char lower[] = "acgt", upper[] = "ACGT",
                      wild[] = "nsywrkvbdhmNSYWRKVBDHM";
    for (int i = 0; i < 4; i++) map[(uint8_t) lower[i]] = map[(uint8_t) upper[i]] = (uint8_t) i;
    map['u'] = map['U'] = 3;
    for (int i = 0; wild[i]; i++) map[(uint8_t) wild[i]] = GTAMD_WILDCARD;
  }
}

void gtamd_symbolmap(uint8_t map[256], int protein) { build_symbolmap(map, protein); }

typedef struct { uint8_t *p; uint64_t len, cap; } bytebuf;

static int bb_push(bytebuf *b, uint8_t c)
{
  if (b->len == b->cap) {
    uint64_t ncap = b->cap ? b->cap * 2 : (1u << 20);
    uint8_t *np = realloc(b->p, ncap);
    if (np == NULL) return -1;
    b->p = np; b->cap = ncap;
  }
  b->p[b->len++] = c;
  return 0;
}

/* room for `more` further bytes, so that a hot loop can store without checks */
static int bb_reserve(bytebuf *b, uint64_t more)
{
  if (b->cap - b->len < more) {
    uint64_t ncap = b->len + more + (more >> 3) + 64;
    uint8_t *np = realloc(b->p, ncap);
    if (np == NULL) return -1;
    b->p = np; b->cap = ncap;
  }
  return 0;
}

static int is_blank(int c)
{
  return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f';
}

/* whole file in memory (the device encoder wants it whole, the host reader
   walks it once).  The compression follows the file name, as in the reference
   (gt_file_mode_determine, src/core/file.c:42-53): "*.gz" is read through
   zlib, "*.bz2" through libbz2; the file length the encoder reports is that of
   the decompressed text. */
int gtamd_read_input_file(const char *path, uint8_t **data, uint64_t *len)
{
  const size_t plen = strlen(path);
  uint64_t cap = 1 << 20, n = 0;
  uint8_t *buf;
  if (plen >= 5 && strcmp(path + plen - 4, ".bz2") == 0) {
    /* libbz2 has no header in this image: bind the three calls at run time */
    typedef void *(*open_fn)(const char *, const char *);
    typedef int (*read_fn)(void *, void *, int);
    typedef void (*close_fn)(void *);
    void *lib = dlopen("libbz2.so.1.0", RTLD_NOW);
    open_fn bzopen;
    read_fn bzread;
    close_fn bzclose;
    void *bz;
    int got;
    if (lib == NULL) lib = dlopen("libbz2.so.1", RTLD_NOW);
    if (lib == NULL) return -3;
    bzopen = (open_fn) dlsym(lib, "BZ2_bzopen");
    bzread = (read_fn) dlsym(lib, "BZ2_bzread");
    bzclose = (close_fn) dlsym(lib, "BZ2_bzclose");
    if (bzopen == NULL || bzread == NULL || bzclose == NULL) return -3;
    if ((bz = bzopen(path, "rb")) == NULL) return -1;
    buf = malloc(cap);
    while (buf != NULL && (got = bzread(bz, buf + n, (int) (cap - n < (1u << 30) ? cap - n : (1u << 30)))) > 0) {
      n += (uint64_t) got;
      if (n == cap) {
        uint8_t *nb = realloc(buf, cap * 2);
        if (nb == NULL) { free(buf); buf = NULL; break; }
        buf = nb; cap *= 2;
      }
    }
    bzclose(bz);
    if (buf == NULL) return -2;
    *data = buf; *len = n;
    return 0;
  }
  if (plen >= 4 && strcmp(path + plen - 3, ".gz") == 0) {
    gzFile gz = gzopen(path, "rb");
    int got;
    if (gz == NULL) return -1;
    (void) gzbuffer(gz, 1 << 20);
    buf = malloc(cap);
    while (buf != NULL && (got = gzread(gz, buf + n, (unsigned) (cap - n < (1u << 30) ? cap - n : (1u << 30)))) > 0) {
      n += (uint64_t) got;
      if (n == cap) {
        uint8_t *nb = realloc(buf, cap * 2);
        if (nb == NULL) { free(buf); buf = NULL; break; }
        buf = nb; cap *= 2;
      }
    }
    if (buf != NULL && !gzeof(gz)) { free(buf); gzclose(gz); return -4; }
    gzclose(gz);
  } else {
    FILE *fp = fopen(path, "rb");
    size_t got;
    if (fp == NULL) return -1;
    if (fseek(fp, 0, SEEK_END) == 0) {
      const long size = ftell(fp);
      if (size > 0) cap = (uint64_t) size + 1;
      rewind(fp);
    }
    buf = malloc(cap);
    while (buf != NULL && (got = fread(buf + n, 1, cap - n, fp)) > 0) {
      n += got;
      if (n == cap) {
        uint8_t *nb = realloc(buf, cap * 2);
        if (nb == NULL) { free(buf); buf = NULL; break; }
        buf = nb; cap *= 2;
      }
    }
    fclose(fp);
  }
  if (buf == NULL) return -2;
  *data = buf; *len = n;
  return 0;
}

void gtamd_read_input_error(int code, const char *path, char *err, size_t errlen)
{
  snprintf(err, errlen,
           code == -1 ? "cannot open file '%s'"
           : code == -3 ? "file '%s': libbz2 is not available to read bzip2-compressed input"
           : code == -4 ? "file '%s' is not a complete gzip stream"
                        : "out of memory while reading '%s'", path);
}

typedef gtamd_fastq_record fastq_record;

typedef struct {
  const uint8_t *map;
  bytebuf *out;
  bytebuf *desc;         /* NUL-separated descriptions, or NULL */
  int seen_record;       /* any sequence so far, over all files */
  uint64_t seqlen;       /* symbols of the FASTA sequence being read */
  gtamd_encinfo *info;   /* original-character histogram and file lengths, or NULL */
  size_t file;           /* index of the file being read */
  fastq_record *rec;     /* FASTQ records so far (for the file length table) */
  size_t nrec, caprec;
  bytebuf *orig;         /* original character of every symbol (-lossless), or NULL */
} encstate;

/* (multi-)FASTA, src/core/sequence_buffer_fasta.c:44-170 */
static uint64_t line_of(const unsigned char *d, size_t upto)
{
  uint64_t line = 1;
  for (const unsigned char *p = d, *end = d + upto;
       (p = memchr(p, '\n', (size_t) (end - p))) != NULL; p++)
    line++;
  return line;
}

static int parse_fasta(encstate *st, const char *path, const unsigned char *d,
                       size_t len, char *err, size_t errlen)
{
  /* byte classes of the sequence state: symbol, blank, '>', anything else */
  enum { C_SYMBOL = 0, C_BLANK, C_HEADER, C_ILLEGAL };
  uint8_t cls[256];
  uint64_t hist[4][256], separators_counted = 0;
  const uint64_t out0 = st->out->len;
  uint64_t seqstart = st->out->len - st->seqlen;     /* first symbol of the open sequence */
  int first_in_file = 1;
  size_t i = 0;
  /* a file contributes at most one symbol per byte */
  if (bb_reserve(st->out, len + 1) != 0 || (st->orig != NULL && bb_reserve(st->orig, len + 1) != 0))
    goto nomem;
  memset(hist, 0, sizeof hist);
  for (int c = 0; c < 256; c++)
    cls[c] = is_blank(c) ? C_BLANK : c == '>' ? C_HEADER
           : st->map[c] == SYM_UNDEF ? C_ILLEGAL : C_SYMBOL;
  while (i < len) {
    const int c = d[i];
    const uint8_t k = cls[c];
    if (k == C_SYMBOL) {
      st->out->p[st->out->len++] = st->map[c];
      if (st->orig != NULL) st->orig->p[st->orig->len++] = (uint8_t) c;
      hist[i & 3][c]++;     /* four tables: no store-to-load chain on one counter */
      i++;
    } else if (k == C_BLANK) i++;
    else if (k == C_HEADER) {
      const unsigned char *nl;
      if (st->seen_record) {
        if (st->out->len == seqstart) {
          snprintf(err, errlen, "file '%s' contains an empty sequence", path);
          return -1;
        }
        st->out->p[st->out->len++] = GTAMD_SEPARATOR;
        if (st->orig != NULL) st->orig->p[st->orig->len++] = 0;
        seqstart = st->out->len;
        /* the separator in front of a file's first sequence is not counted
           for that file (sequence_buffer_fasta.c:133-146) */
        if (!first_in_file) separators_counted++;
      }
      st->seen_record = 1;
      first_in_file = 0;
      /* the description runs to the end of the line (or of the file) */
      i++;
      nl = memchr(d + i, '\n', len - i);
      {
        const size_t end = nl != NULL ? (size_t) (nl - d) : len;
        if (st->desc != NULL) {
          for (size_t q = i; q < end; q++)
            if (d[q] != '\r' && bb_push(st->desc, d[q]) != 0) goto nomem;
          if (nl != NULL && bb_push(st->desc, 0) != 0) goto nomem;
        }
        i = nl != NULL ? end + 1 : len;
      }
    } else {
      snprintf(err, errlen, "illegal character '%c': file \"%s\", line %llu", c,
               path, (unsigned long long) line_of(d, i));
      return -1;
    }
  }
  st->seqlen = st->out->len - seqstart;
  if (st->info != NULL) {
    uint64_t symbols = 0;
    for (int c = 0; c < 256; c++) {
      const uint64_t h = hist[0][c] + hist[1][c] + hist[2][c] + hist[3][c];
      st->info->originaldistribution[c] += h;
      symbols += h;
    }
    /* bytes read / symbols and separators contributed by this file
       (sequence_buffer_fasta.c:86-94,104,144,156) */
    st->info->filelengthtab[st->file].length = len;
    st->info->filelengthtab[st->file].effectivelength = symbols + separators_counted;
  }
  (void) out0;
  return 0;
nomem:
  snprintf(err, errlen, "out of memory while reading '%s'", path);
  return -1;
}

/* FASTQ, block grammar of src/core/seq_iterator_fastq.c:96-305: "@name",
   sequence lines up to a '+', "+[name]", then exactly as many quality
   characters as the sequence has symbols (over any number of lines) */
static int parse_fastq(encstate *st, const char *path, const unsigned char *d,
                       size_t len, char *err, size_t errlen)
{
  size_t i = 0;
  uint64_t line = 1, hist[4][256];
  uint8_t cls[256];                /* 0 symbol, 1 skipped ('\n', ' '), 2 illegal */
  if (bb_reserve(st->out, len + 1) != 0 || (st->orig != NULL && bb_reserve(st->orig, len + 1) != 0))
    goto nomem;
  memset(hist, 0, sizeof hist);
  for (int c = 0; c < 256; c++)
    cls[c] = c == '\n' || c == ' ' ? 1 : st->map[c] == SYM_UNDEF ? 2 : 0;
  while (i < len) {
    size_t name0, name1, nsym = 0, nq = 0, q0, q1;
    if (d[i] != '@') {
      snprintf(err, errlen, "'@' expected, '%c' encountered instead in line %llu",
               d[i], (unsigned long long) line);
      return -1;
    }
    name0 = ++i;
    while (i < len && d[i] != '\n') i++;
    if (i >= len) goto premature;
    name1 = i++; line++;
    if (st->desc != NULL) {
      for (size_t k = name0; k < name1; k++)
        if (bb_push(st->desc, d[k]) != 0) goto nomem;
      if (bb_push(st->desc, 0) != 0) goto nomem;
    }
    if (st->seen_record && bb_push(st->out, GTAMD_SEPARATOR) != 0) goto nomem;
    if (st->seen_record && st->orig != NULL && bb_push(st->orig, 0) != 0) goto nomem;
    st->seen_record = 1;
    for (; i < len && d[i] != '+'; i++) {
      const int c = d[i];
      const uint8_t k = cls[c];
      if (k == 0) {                          /* a symbol */
        st->out->p[st->out->len++] = st->map[c];
        if (st->orig != NULL) st->orig->p[st->orig->len++] = (uint8_t) c;
        hist[i & 3][c]++;
        nsym++;
      } else if (k == 1) {                   /* line end or blank */
        if (c == '\n') line++;
      } else {
        /* the reference maps the sequence after it has read it completely */
        uint64_t l2 = line;
        for (size_t q = i; q < len && d[q] != '+'; q++) l2 += d[q] == '\n';
        snprintf(err, errlen, "illegal character '%c': file \"%s\", line %llu", c,
                 path, (unsigned long long) l2);
        return -1;
      }
    }
    if (i >= len) goto premature;
    if (nsym == 0) {
      snprintf(err, errlen, "empty sequence given in file '%s', line %llu", path,
               (unsigned long long) (line - 1));
      return -1;
    }
    q0 = ++i;                                   /* behind the '+' */
    while (i < len && d[i] != '\n') i++;
    if (i >= len) goto premature;
    q1 = i++; line++;
    if (q1 > q0 && (q1 - q0 != name1 - name0 || memcmp(d + q0, d + name0, q1 - q0) != 0)) {
      snprintf(err, errlen, "sequence description '%.*s' is not equal to qualities "
               "description '%.*s' in line %llu", (int) (name1 - name0), d + name0,
               (int) (q1 - q0), d + q0, (unsigned long long) (line - 1));
      return -1;
    }
    /* the usual layout: all qualities on one line */
    if (i + nsym < len && d[i + nsym] == '\n' && memchr(d + i, '\n', nsym) == NULL &&
        memchr(d + i, ' ', nsym) == NULL) {
      i += nsym;
      nq = nsym;
    }
    while (nq < nsym) {
      if (i >= len) {
        /* the reference notices the short quality string first
           (src/core/seq_iterator_fastq.c:297-304) */
        snprintf(err, errlen, "lengths of character sequence and qualities sequence "
                 "differ (%llu <-> %llu)", (unsigned long long) nq,
                 (unsigned long long) nsym);
        return -1;
      }
      if (d[i] == '\n') line++;
      else if (d[i] != ' ') nq++;
      i++;
    }
    if (i >= len) goto premature;
    if (d[i] != '\n') {
      snprintf(err, errlen, "qualities string of sequence length %llu is not ended "
               "by newline in file '%s', line %llu -- this may be a sign for sequence "
               "and qualities strings of different length", (unsigned long long) nsym,
               path, (unsigned long long) line);
      return -1;
    }
    i++; line++;
    if (st->info != NULL) {
      if (st->nrec == st->caprec) {
        const size_t ncap = st->caprec ? 2 * st->caprec : 1024;
        fastq_record *nr = realloc(st->rec, ncap * sizeof *nr);
        if (nr == NULL) goto nomem;
        st->rec = nr; st->caprec = ncap;
      }
      st->rec[st->nrec].seqlen = nsym;
      st->rec[st->nrec].desclen = name1 - name0;
      st->rec[st->nrec].file = st->file;
      st->nrec++;
    }
  }
  if (st->info != NULL)
    for (int c = 0; c < 256; c++)
      st->info->originaldistribution[c] += hist[0][c] + hist[1][c] + hist[2][c] + hist[3][c];
  return 0;
premature:
  snprintf(err, errlen, "premature end of file '%s' in line %llu: file ended before "
           "end of block", path, (unsigned long long) (line - 1));
  return -1;
nomem:
  snprintf(err, errlen, "out of memory while reading '%s'", path);
  return -1;
}

int gtamd_encode_files(const char *const *paths, size_t numfiles, int protein,
                       uint8_t **enc, uint64_t *n, char *err, size_t errlen)
{
  return gtamd_encode_files_desc(paths, numfiles, protein, enc, n, NULL, NULL,
                                 err, errlen);
}

int gtamd_encode_files_desc(const char *const *paths, size_t numfiles,
                            int protein, uint8_t **enc, uint64_t *n,
                            char **desc, uint64_t *desclen, char *err,
                            size_t errlen)
{
  return gtamd_encode_files_info(paths, numfiles, protein, enc, n, desc, desclen,
                                 NULL, err, errlen);
}

void gtamd_encinfo_free(gtamd_encinfo *info)
{
  if (info != NULL) { free(info->filelengthtab); info->filelengthtab = NULL; }
}

/* The file length table as the reference's FASTQ reader accounts it: it fills
   an output buffer of 8192 symbols per call and books what it read per call,
   including the description lengths and the quirks at buffer boundaries
   (src/core/sequence_buffer_fastq.c:42-191, OUTBUFSIZE sequence_buffer_rep.h:30) */
void gtamd_fastq_filelengths(const gtamd_fastq_record *rec, size_t nrec, size_t lastfile,
                             gtamd_filelength *tab)
{
  const uint64_t OUTBUF = 8192;
  uint64_t overflow = 0;
  size_t r = 0, filenum = 0;
  int carry = 0, complete = 0;
  while (!complete) {
    uint64_t out = 0, add = 0, rd = 0;
    if (carry) { out++; rd++; add++; carry = 0; }
    if (overflow > 0) {
      const uint64_t k = overflow < OUTBUF - out ? overflow : OUTBUF - out;
      out += k; add += k; rd += k; overflow -= k;
      if (overflow > 0) continue;           /* counts of this call are dropped */
      out++; rd++;
    }
    for (;;) {
      const size_t newfile = r < nrec ? rec[r].file : lastfile;
      if (filenum != newfile) {
        tab[filenum].length += rd; tab[filenum].effectivelength += add;
        rd = add = 0; filenum = newfile;
      }
      if (r == nrec) { complete = 1; out--; add--; break; }
      {
        /* (symbol by symbol in the reference: what fits the buffer is booked,
           the rest overflows into the next call) */
        const uint64_t room = out < OUTBUF ? OUTBUF - out : 0;
        const uint64_t k = rec[r].seqlen < room ? rec[r].seqlen : room;
        out += k; add += k; rd += k;
        overflow += rec[r].seqlen - k;
      }
      if (overflow == 0) {
        if (out >= OUTBUF) carry = 1;
        else { out++; add++; }
      }
      rd += rec[r].desclen + 1;
      r++;
      if (out >= OUTBUF) break;
    }
    tab[filenum].length += rd; tab[filenum].effectivelength += add;
  }
}

int gtamd_encode_files_info(const char *const *paths, size_t numfiles,
                            int protein, uint8_t **enc, uint64_t *n,
                            char **desc, uint64_t *desclen,
                            gtamd_encinfo *info, char *err, size_t errlen)
{
  gtamd_alphabet a;
  gtamd_alphabet_standard(&a, protein);
  return gtamd_encode_files_alpha(paths, numfiles, &a, enc, n, desc, desclen, info, err,
                                  errlen);
}

int gtamd_encode_files_alpha(const char *const *paths, size_t numfiles,
                             const gtamd_alphabet *a, uint8_t **enc, uint64_t *n,
                             char **desc, uint64_t *desclen,
                             gtamd_encinfo *info, char *err, size_t errlen)
{
  return gtamd_encode_files_orig(paths, numfiles, a, enc, n, NULL, desc, desclen, info, err,
                                 errlen);
}

int gtamd_encode_files_orig(const char *const *paths, size_t numfiles,
                            const gtamd_alphabet *a, uint8_t **enc, uint64_t *n,
                            uint8_t **orig, char **desc, uint64_t *desclen,
                            gtamd_encinfo *info, char *err, size_t errlen)
{
  uint8_t map[256];
  bytebuf obuf = {NULL, 0, 0};
  bytebuf out = {NULL, 0, 0};
  bytebuf dbuf = {NULL, 0, 0};
  encstate st = {map, &out, desc != NULL ? &dbuf : NULL, 0, 0, info, 0, NULL, 0, 0,
                 orig != NULL ? &obuf : NULL};
  int last_was_fasta = 1, rc = 0;

  memcpy(map, a->symbolmap, 256);
  if (info != NULL) {
    memset(info, 0, sizeof *info);
    info->numfiles = numfiles;
    info->filelengthtab = calloc(numfiles ? numfiles : 1, sizeof *info->filelengthtab);
    if (info->filelengthtab == NULL) {
      snprintf(err, errlen, "out of memory");
      return -1;
    }
  }
  for (size_t f = 0; f < numfiles && rc == 0; f++) {
    uint8_t *data = NULL;
    uint64_t len = 0;
    st.file = f;
    rc = gtamd_read_input_file(paths[f], &data, &len);
    if (rc != 0) {
      gtamd_read_input_error(rc, paths[f], err, errlen);
      break;
    }
    /* format by the first character, as the reference guesses it
       (src/core/sequence_buffer.c) */
    if (len > 0 && data[0] == '@') {
      /* a FASTQ file always ends a record: the next file starts a new one */
      if (st.seen_record && last_was_fasta && st.seqlen == 0) rc = -3;
      else rc = parse_fastq(&st, paths[f], data, len, err, errlen);
      last_was_fasta = 0;
      st.seqlen = 1;
    } else {
      if (st.seen_record && !last_was_fasta) {
        /* FASTA after FASTQ: its first '>' must emit the separator */
        st.seqlen = 1;
      }
      rc = parse_fasta(&st, paths[f], data, len, err, errlen);
      last_was_fasta = 1;
    }
    free(data);
    if (rc == -3)
      snprintf(err, errlen, "file '%s' contains an empty sequence", paths[f]);
  }
  if (rc == 0 && !st.seen_record) {
    snprintf(err, errlen, "no sequences in multiple fasta file(s) %s ...",
             numfiles ? paths[0] : "");
    rc = -1;
  }
  if (rc == 0 && last_was_fasta && st.seqlen == 0) {
    snprintf(err, errlen, "file '%s' contains an empty sequence", paths[numfiles - 1]);
    rc = -1;
  }
  if (rc != 0) {
    free(out.p); free(dbuf.p); free(st.rec); free(obuf.p);
    gtamd_encinfo_free(info);
    return -1;
  }
  if (orig != NULL) *orig = obuf.p;
  if (info != NULL && st.nrec > 0)
    gtamd_fastq_filelengths(st.rec, st.nrec, numfiles - 1, info->filelengthtab);
  free(st.rec);
  /* a header that ends with the file (no newline) still counts */
  if (desc != NULL && (dbuf.len == 0 || dbuf.p[dbuf.len - 1] != 0)) (void) bb_push(&dbuf, 0);
  *enc = out.p;
  *n = out.len;
  if (desc != NULL) { *desc = (char *) dbuf.p; *desclen = dbuf.len; }
  return 0;
}

void gtamd_clip_descriptions(char *desc, uint64_t *desclen)
{
  /* -clipdesc: every description ends at its first white space
     (gt_desc_buffer_append_char, src/core/desc_buffer.c:69-80) */
  uint64_t in = 0, out = 0;
  while (in < *desclen) {
    const uint64_t l = strlen(desc + in);
    uint64_t keep = 0;
    while (keep < l && !is_blank((unsigned char) desc[in + keep])) keep++;
    memmove(desc + out, desc + in, keep);
    desc[out + keep] = 0;
    out += keep + 1;
    in += l + 1;
  }
  *desclen = out;
}

int gtamd_write_des_sds(const char *indexname, const char *desc,
                        uint64_t desclen, int write_des, int write_sds)
{
  char path[4096];
  FILE *fd = NULL, *fs = NULL;
  uint64_t off = 0, longest = 0, pos = 0;
  const uint64_t fin = ~(uint64_t) 0;
  if (write_des) {
    snprintf(path, sizeof path, "%s.des", indexname);
    if ((fd = fopen(path, "wb")) == NULL) return -1;
  }
  if (write_sds) {
    snprintf(path, sizeof path, "%s.sds", indexname);
    if ((fs = fopen(path, "wb")) == NULL) { if (fd) fclose(fd); return -1; }
  }
  while (pos < desclen) {
    const uint64_t l = strlen(desc + pos);
    const int last = pos + l + 1 >= desclen;
    if (l > longest) longest = l;
    off += l;
    if (fd) { fwrite(desc + pos, 1, l, fd); fputc('\n', fd); }
    if (fs && !last) fwrite(&off, sizeof off, 1, fs);
    off += 1;
    pos += l + 1;
  }
  if (fd) { fwrite(&longest, sizeof longest, 1, fd); fwrite(&fin, sizeof fin, 1, fd); }
  if (fd && fclose(fd) != 0) { if (fs) fclose(fs); return -1; }
  if (fs && fclose(fs) != 0) return -1;
  return 0;
}

/* MD5 of one sequence: its decoded symbols in upper case, a wildcard decodes
   to the alphabet's wildcard character (N / X) */
static void md5_of_sequence(const uint8_t *enc, uint64_t len, const uint8_t *show,
                            char hex[33])
{
  uint8_t block[4096];
  size_t fill = 0;
  gtamd_md5 st;
  gtamd_md5_init(&st);
  for (uint64_t i = 0; i < len; i++) {
    block[fill++] = show[enc[i]];
    if (fill == sizeof block) { gtamd_md5_update(&st, block, fill); fill = 0; }
  }
  gtamd_md5_update(&st, block, fill);
  gtamd_md5_hex(&st, hex);
}

typedef struct {
  const uint8_t *enc;
  const uint64_t *start, *len;     /* of the sequences of this batch */
  uint64_t count;
  const uint8_t *show;             /* upper-case character per symbol */
  char *hex;                       /* 33 bytes per sequence */
  uint64_t next;                   /* work counter, handed out under the lock */
  pthread_mutex_t lock;
} md5_batch;

static void *md5_worker(void *arg)
{
  md5_batch *b = arg;
  for (;;) {
    uint64_t k;
    pthread_mutex_lock(&b->lock);
    k = b->next++;
    pthread_mutex_unlock(&b->lock);
    if (k >= b->count) return NULL;
    md5_of_sequence(b->enc + b->start[k], b->len[k], b->show, b->hex + 33 * k);
  }
}

int gtamd_write_md5(const char *indexname, const uint8_t *enc, uint64_t n,
                    int protein)
{
  gtamd_alphabet a;
  gtamd_alphabet_standard(&a, protein);
  return gtamd_write_md5_alpha(indexname, enc, n, &a);
}

static int write_md5_show(const char *indexname, const uint8_t *enc, uint64_t n,
                          const uint8_t *show);

int gtamd_write_md5_orig(const char *indexname, const uint8_t *enc,
                         const uint8_t *orig, uint64_t n)
{
  /* the sums over the original characters in upper case: the batch code works
     on "symbols", so hand it the upper-cased originals with an identity map
     (separators keep their code) */
  uint8_t show[256], *up = malloc(n ? n : 1);
  int rc;
  if (up == NULL) return -1;
  for (int c = 0; c < 256; c++) show[c] = (uint8_t) c;
  for (uint64_t i = 0; i < n; i++)
    up[i] = enc[i] == GTAMD_SEPARATOR ? (uint8_t) GTAMD_SEPARATOR
                                      : (uint8_t) toupper(orig[i]);
  rc = write_md5_show(indexname, up, n, show);
  free(up);
  return rc;
}

int gtamd_write_md5_alpha(const char *indexname, const uint8_t *enc, uint64_t n,
                          const gtamd_alphabet *a)
{
  uint8_t show[256];
  /* decoded symbols in upper case; a wildcard decodes to the alphabet's
     wildcard character (encseq_charproc.gen:27-36,55-66) */
  memset(show, 0, sizeof show);
  for (uint32_t c = 0; c < a->numofchars; c++)
    show[c] = (uint8_t) toupper((unsigned char) a->characters[c]);
  show[GTAMD_WILDCARD] = (uint8_t) toupper((unsigned char) a->wildcardshow);
  return write_md5_show(indexname, enc, n, show);
}

static int write_md5_show(const char *indexname, const uint8_t *enc, uint64_t n,
                          const uint8_t *show)
{
  /* the sums of different sequences are independent: batches of sequences go
     to a few threads, longest-first order does not matter at this grain */
  enum { BATCH = 1 << 16, MAXTHREADS = 16 };
  char path[4096];
  uint64_t *start = malloc(sizeof *start * BATCH), *len = malloc(sizeof *len * BATCH),
           pos = 0;
  char *hex = malloc(33 * (size_t) BATCH);
  long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
  int nthreads = ncpu < 1 ? 1 : ncpu > MAXTHREADS ? MAXTHREADS : (int) ncpu, rc = -1;
  FILE *fp = NULL;
  snprintf(path, sizeof path, "%s.md5", indexname);
  if (start == NULL || len == NULL || hex == NULL || (fp = fopen(path, "wb")) == NULL) goto done;
  while (pos <= n) {
    md5_batch b;
    pthread_t th[MAXTHREADS];
    int started = 0;
    uint64_t count = 0;
    while (count < BATCH && pos <= n) {
      const uint8_t *sep = pos < n ? memchr(enc + pos, GTAMD_SEPARATOR, n - pos) : NULL;
      const uint64_t end = sep != NULL ? (uint64_t) (sep - enc) : n;
      start[count] = pos; len[count] = end - pos; count++;
      pos = end + 1;
    }
    b.enc = enc; b.start = start; b.len = len; b.count = count; b.show = show;
    b.hex = hex; b.next = 0;
    pthread_mutex_init(&b.lock, NULL);
    for (int t = 1; t < nthreads && (uint64_t) t < count; t++)
      if (pthread_create(&th[started], NULL, md5_worker, &b) == 0) started++;
    (void) md5_worker(&b);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&b.lock);
    if (fwrite(hex, 33, count, fp) != count) goto done;
  }
  rc = 0;
done:
  if (fp != NULL && fclose(fp) != 0) rc = -1;
  free(start); free(len); free(hex);
  return rc;
}

/* a run of `len` specials is stored as this many table entries when the run
   length field holds at most maxv+1 (src/core/encseq.c:5061-5074) */
static uint64_t pieces(uint64_t len, uint64_t maxv)
{
  if (maxv == UINT32_MAX || len <= maxv + 1) return 1;
  return len / (maxv + 1) + (len % (maxv + 1) != 0);
}

typedef struct { uint64_t chars, runs, tab[3], prefix, suffix, current; int at_start; } runstat;

static void run_feed(runstat *r, int member, int last)
{
  static const uint64_t maxv[3] = {UCHAR_MAX, USHRT_MAX, UINT32_MAX};
  if (member) {
    r->chars++; r->current++;
    if (r->at_start) r->prefix++;
  } else r->at_start = 0;
  if ((!member || last) && r->current > 0) {
    if (member && last) r->suffix = r->current;
    r->runs++;
    for (int k = 0; k < 3; k++) r->tab[k] += pieces(r->current, maxv[k]);
    r->current = 0;
  }
}

uint64_t gtamd_swtable_bytes(int width_kind, int withrangelengths, uint64_t n,
                             uint64_t items)
{
  static const uint64_t width[3] = {1, 2, 4}, maxv[3] = {UCHAR_MAX, USHRT_MAX, UINT32_MAX};
  if (items == 0) return 0;
  return (withrangelengths ? 2 : 1) * width[width_kind] * items
         + 8 * (n / maxv[width_kind] + 1);
}

void gtamd_analyse_sequence(const uint8_t *enc, uint64_t n, uint32_t numofchars,
                            gtamd_seqanalysis *an)
{
  runstat sp, wc;
  gtamd_seqstats *st = &an->ss;
  uint64_t seqlen = 0, nsep = 0, nonspecial = 0, eqvalue = 0;
  int equal = 1;
  memset(an, 0, sizeof *an);
  memset(&sp, 0, sizeof sp); memset(&wc, 0, sizeof wc);
  sp.at_start = wc.at_start = 1;
  an->minseqlen = an->maxseqlen = ~(uint64_t) 0;
  for (uint64_t i = 0; i <= n; i++) {
    const uint8_t c = i < n ? enc[i] : GTAMD_SEPARATOR;
    if (i < n) {
      run_feed(&sp, c >= GTAMD_WILDCARD, i + 1 == n);
      run_feed(&wc, c == GTAMD_WILDCARD, i + 1 == n);
    }
    if (c < GTAMD_WILDCARD) { an->chardist[c]++; nonspecial++; }
    else {
      if (nonspecial > an->lengthoflongestnonspecial) an->lengthoflongestnonspecial = nonspecial;
      nonspecial = 0;
    }
    if (c == GTAMD_SEPARATOR) {
      /* end of a sequence (the last one ends with the input) */
      if (an->maxseqlen == ~(uint64_t) 0 || seqlen > an->maxseqlen) an->maxseqlen = seqlen;
      if (an->minseqlen == ~(uint64_t) 0 || seqlen < an->minseqlen) an->minseqlen = seqlen;
      if (eqvalue > 0) { if (seqlen != eqvalue) equal = 0; }
      else eqvalue = seqlen;
      if (i < n) nsep++;
      seqlen = 0;
    } else seqlen++;
  }
  /* more specials than separators: some sequence holds a wildcard */
  if (sp.chars > nsep) equal = 0;
  an->equallength = equal;
  an->equallength_value = equal ? eqvalue : 0;
  st->totallength = n; st->numofchars = numofchars; st->numofsequences = nsep + 1;
  st->specialcharacters = sp.chars; st->realspecialranges = sp.runs;
  st->lengthofspecialprefix = sp.prefix; st->lengthofspecialsuffix = sp.suffix;
  st->wildcards = wc.chars; st->realwildcardranges = wc.runs;
  st->lengthofwildcardprefix = wc.prefix; st->lengthofwildcardsuffix = wc.suffix;
  for (int k = 0; k < 3; k++) { an->sp_tab[k] = sp.tab[k]; an->wc_tab[k] = wc.tab[k]; }
  (void) gtamd_choose_access_type(an, sp.tab, wc.tab, -1);
}

int gtamd_parse_sat(const char *name, int protein, int *sat, char *err, size_t errlen)
{
  static const char *const names[7] = {"direct", "bytecompress", "eqlen", "bit", "uchar",
                                       "ushort", "uint32"};
  *sat = -1;
  for (int k = 0; k < 7; k++) if (!strcmp(name, names[k])) *sat = k;
  if (*sat < 0) {
    /* src/core/encseq.c:797-807 */
    snprintf(err, errlen, "Illegal argument \"%s\" to option -sat; must be one of the "
             "following keywords: direct, bytecompress, eqlen, bit, uchar, ushort, uint32", name);
    return -1;
  }
  /* src/core/encseq_access_type.c:163-221 */
  if (!protein && *sat == GTAMD_SAT_BYTECOMPRESS) {
    snprintf(err, errlen, "illegal argument \"%s\" to option -sat: cannot use bytecompress "
             "on DNA sequences", name);
    return -1;
  }
  if (protein && *sat != GTAMD_SAT_BYTECOMPRESS && *sat != GTAMD_SAT_DIRECTACCESS) {
    snprintf(err, errlen, "illegal argument \"%s\" to option -sat: as the sequence is not "
             "DNA, you can choose bytecompress or direct", name);
    return -1;
  }
  return 0;
}

int gtamd_choose_access_type(gtamd_seqanalysis *an, const uint64_t sp_tab[3],
                             const uint64_t wc_tab[3], int forced_sat)
{
  gtamd_seqstats *st = &an->ss;
  const uint64_t n = st->totallength, nsep = st->numofsequences - 1;
  uint64_t best = 0;
  /* the "ranges" numbers are those of the smallest of the three table
     representations, whatever access type is used in the end -- or of the
     table type -sat asks for (src/core/encseq.c:5215-5256, :797-814; sizes
     encseq.c:924-949) */
  const int forcetable = forced_sat >= GTAMD_SAT_UCHARTABLES ? forced_sat - GTAMD_SAT_UCHARTABLES : 3;
  for (int k = 0; k < 3; k++) {
    const uint64_t size = gtamd_swtable_bytes(k, 1, n, wc_tab[k]);
    if (forcetable != 3 && k != forcetable) continue;
    if (k == 0 || forcetable == k || size < best) {
      best = size; st->specialranges = sp_tab[k]; st->wildcardranges = wc_tab[k];
    }
  }
  if (forced_sat >= 0) {
    /* -sat: src/core/encseq_access_type.c:163-221 */
    if (forced_sat == GTAMD_SAT_EQUALLENGTH && !an->equallength) return -1;
    an->sat = forced_sat;
    an->sat_wildcardranges = forcetable != 3 ? wc_tab[forcetable] : wc_tab[0];
    return 0;
  }
  /* access type: non-DNA alphabets are bit-packed; DNA takes the smallest of
     bit access and the three table types, or "equal length" when all sequences
     have the same length and hold no wildcard
     (src/core/encseq_access_type.c:96-162) */
  an->sat_wildcardranges = wc_tab[0];
  if (st->numofchars != 4) an->sat = GTAMD_SAT_BYTECOMPRESS;
  else if (an->equallength) an->sat = GTAMD_SAT_EQUALLENGTH;
  else {
    an->sat = GTAMD_SAT_BITACCESS;
    best = (wc_tab[0] > 0 || nsep > 0) ? 8 * (1 + (n + 63) / 64) : 0;
    for (int k = 0; k < 3; k++) {
      const uint64_t size = gtamd_swtable_bytes(k, 1, n, wc_tab[k]);
      if (size < best) {
        best = size; an->sat = GTAMD_SAT_UCHARTABLES + k;
        an->sat_wildcardranges = wc_tab[k];
      }
    }
  }
  return 0;
}

void gtamd_analysis_from_summary(const gtamd_encode_summary *s, uint32_t numofchars,
                                 gtamd_seqanalysis *an)
{
  gtamd_seqstats *st = &an->ss;
  memset(an, 0, sizeof *an);
  st->totallength = s->totallength; st->numofchars = numofchars;
  st->numofsequences = s->numofsequences;
  st->specialcharacters = s->specialcharacters; st->realspecialranges = s->realspecialranges;
  st->lengthofspecialprefix = s->lengthofspecialprefix;
  st->lengthofspecialsuffix = s->lengthofspecialsuffix;
  st->wildcards = s->wildcards; st->realwildcardranges = s->realwildcardranges;
  st->lengthofwildcardprefix = s->lengthofwildcardprefix;
  st->lengthofwildcardsuffix = s->lengthofwildcardsuffix;
  an->lengthoflongestnonspecial = s->lengthoflongestnonspecial;
  an->minseqlen = s->minseqlen; an->maxseqlen = s->maxseqlen;
  an->equallength = s->equallength != 0;
  an->equallength_value = an->equallength ? s->maxseqlen : 0;
  for (int c = 0; c < 32; c++) an->chardist[c] = s->characterdistribution[c];
  for (int k = 0; k < 3; k++) {
    an->sp_tab[k] = s->specialrangestab[k]; an->wc_tab[k] = s->wildcardrangestab[k];
  }
  (void) gtamd_choose_access_type(an, s->specialrangestab, s->wildcardrangestab, -1);
}

void gtamd_sequence_stats(const uint8_t *enc, uint64_t n, uint32_t numofchars,
                          gtamd_seqstats *st)
{
  gtamd_seqanalysis an;
  gtamd_analyse_sequence(enc, n, numofchars, &an);
  *st = an.ss;
}

void gtamd_apply_readmode(uint8_t *enc, uint64_t n, int readmode)
{
  const int reverse = readmode == 1 || readmode == 3,
            complement = readmode == 2 || readmode == 3;
  if (reverse)
    for (uint64_t a = 0, b = n; a + 1 < b; a++) {
      const uint8_t t = enc[a];
      b--;
      enc[a] = enc[b];
      enc[b] = t;
    }
  if (complement)
    for (uint64_t i = 0; i < n; i++)
      if (enc[i] < 4) enc[i] ^= 3;      /* a<->t, c<->g */
}

uint8_t *gtamd_mirror(const uint8_t *enc, uint64_t n)
{
  uint8_t *m = malloc(2 * n + 1);
  if (m == NULL) return NULL;
  memcpy(m, enc, n);
  m[n] = GTAMD_SEPARATOR;
  for (uint64_t i = 0; i < n; i++) {
    const uint8_t c = enc[n - 1 - i];
    m[n + 1 + i] = c < 4 ? (uint8_t) (c ^ 3) : c;
  }
  return m;
}

void gtamd_seqstats_mirror(gtamd_seqstats *st, int last_symbol_is_wildcard)
{
  const uint64_t central = last_symbol_is_wildcard ? 0 : 2;   /* 2r-1 or 2r+1 */
  st->totallength = 2 * st->totallength + 1;
  st->specialcharacters = 2 * st->specialcharacters + 1;
  st->specialranges = 2 * st->specialranges - 1 + central;
  st->realspecialranges = 2 * st->realspecialranges - 1 + central;
  st->wildcards *= 2;
  st->wildcardranges *= 2;
  st->realwildcardranges *= 2;
  st->numofsequences *= 2;
}

/* INDEX.prj of a `gt packedindex mkindex` run: gt_outprjfile with no suffixes
   written and `longest` undefined (src/match/sfx-run.c:600-690 with doesa false,
   src/match/sfx-outprj.c:38-83) */
int gtamd_write_prj_packedindex(const char *path, const gtamd_seqstats *ss,
                                uint32_t prefixlength, int readmode, int mirrored)
{
  FILE *fp = fopen(path, "wb");
  if (fp == NULL) return -1;
  fprintf(fp, "totallength=%llu\n", (unsigned long long) ss->totallength);
  fprintf(fp, "specialcharacters=%llu\n", (unsigned long long) ss->specialcharacters);
  fprintf(fp, "specialranges=%llu\n", (unsigned long long) ss->specialranges);
  fprintf(fp, "realspecialranges=%llu\n", (unsigned long long) ss->realspecialranges);
  fprintf(fp, "lengthofspecialprefix=%llu\n", (unsigned long long) ss->lengthofspecialprefix);
  fprintf(fp, "lengthofspecialsuffix=%llu\n", (unsigned long long) ss->lengthofspecialsuffix);
  fprintf(fp, "wildcards=%llu\n", (unsigned long long) ss->wildcards);
  fprintf(fp, "wildcardranges=%llu\n", (unsigned long long) ss->wildcardranges);
  fprintf(fp, "realwildcardranges=%llu\n", (unsigned long long) ss->realwildcardranges);
  fprintf(fp, "lengthofwildcardprefix=%llu\n", (unsigned long long) ss->lengthofwildcardprefix);
  fprintf(fp, "lengthofwildcardsuffix=%llu\n", (unsigned long long) ss->lengthofwildcardsuffix);
  fprintf(fp, "numofsequences=%llu\n", (unsigned long long) ss->numofsequences);
  fprintf(fp, "numofdbsequences=%llu\n", (unsigned long long) ss->numofsequences);
  fprintf(fp, "numofquerysequences=0\nnumberofallsortedsuffixes=0\n");
  fprintf(fp, "prefixlength=%u\n", prefixlength);
  fprintf(fp, "largelcpvalues=0\naveragelcp=0.00\nmaxbranchdepth=0\n");
  fprintf(fp, "integersize=64\nlittleendian=1\nreadmode=%d\nmirrored=%d\n",
          readmode, mirrored ? 1 : 0);
  return fclose(fp) == 0 ? 0 : -1;
}

int gtamd_write_prj(const char *path, const gtamd_seqstats *ss,
                    const gtamd_esa_stats *es, int with_lcp, int readmode,
                    int mirrored)
{
  FILE *fp = fopen(path, "wb");
  const unsigned long long n1 = es->numberofallsortedsuffixes;
  if (fp == NULL) return -1;
  fprintf(fp, "totallength=%llu\n", (unsigned long long) ss->totallength);
  fprintf(fp, "specialcharacters=%llu\n", (unsigned long long) ss->specialcharacters);
  fprintf(fp, "specialranges=%llu\n", (unsigned long long) ss->specialranges);
  fprintf(fp, "realspecialranges=%llu\n", (unsigned long long) ss->realspecialranges);
  fprintf(fp, "lengthofspecialprefix=%llu\n", (unsigned long long) ss->lengthofspecialprefix);
  fprintf(fp, "lengthofspecialsuffix=%llu\n", (unsigned long long) ss->lengthofspecialsuffix);
  fprintf(fp, "wildcards=%llu\n", (unsigned long long) ss->wildcards);
  fprintf(fp, "wildcardranges=%llu\n", (unsigned long long) ss->wildcardranges);
  fprintf(fp, "realwildcardranges=%llu\n", (unsigned long long) ss->realwildcardranges);
  fprintf(fp, "lengthofwildcardprefix=%llu\n", (unsigned long long) ss->lengthofwildcardprefix);
  fprintf(fp, "lengthofwildcardsuffix=%llu\n", (unsigned long long) ss->lengthofwildcardsuffix);
  fprintf(fp, "numofsequences=%llu\n", (unsigned long long) ss->numofsequences);
  fprintf(fp, "numofdbsequences=%llu\n", (unsigned long long) ss->numofsequences);
  fprintf(fp, "numofquerysequences=0\n");
  fprintf(fp, "numberofallsortedsuffixes=%llu\n", n1);
  fprintf(fp, "longest=%llu\n", (unsigned long long) es->longest);
  fprintf(fp, "prefixlength=%u\n", es->prefixlength);
  fprintf(fp, "largelcpvalues=%llu\n", with_lcp ? (unsigned long long) es->largelcpvalues : 0ull);
  fprintf(fp, "averagelcp=%.2f\n", with_lcp ? (double) es->lcptabsum / (double) n1 : 0.0);
  fprintf(fp, "maxbranchdepth=%llu\n", with_lcp ? (unsigned long long) es->maxbranchdepth : 0ull);
  fprintf(fp, "integersize=64\nlittleendian=1\nreadmode=%d\nmirrored=%d\n",
          readmode, mirrored ? 1 : 0);
  return fclose(fp) == 0 ? 0 : -1;
}
