/* device_host.c -- the host layer's use of the device encoder
   (include/gtamd_encode.h): whole files go to the GPU, the symbols stay there
   for the ESA engine, and what the sequence-side files need comes back as
   summaries, lists and packed sections instead of per-byte host loops. */
#include "host_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

int gtamd_input_is_fastq(const char *const *paths, size_t numfiles)
{
  /* format by the first character (src/core/sequence_buffer.c guesses the
     type from the first file) */
  for (size_t f = 0; f < numfiles; f++) {
    const size_t plen = strlen(paths[f]);
    int c;
    if (plen >= 4 && strcmp(paths[f] + plen - 3, ".gz") == 0) {
      gzFile gz = gzopen(paths[f], "rb");
      if (gz == NULL) continue;
      c = gzgetc(gz);
      gzclose(gz);
    } else {
      FILE *fp = fopen(paths[f], "rb");
      if (fp == NULL) continue;
      c = fgetc(fp);
      fclose(fp);
    }
    if (c == '@') return 1;
  }
  return 0;
}

int gtamd_device_encode_files(const char *const *paths, size_t numfiles,
                              int protein, gtamd_encoder **enc_out,
                              char **desc, uint64_t *desclen,
                              gtamd_encinfo *info, char *err, size_t errlen)
{
  gtamd_alphabet a;
  gtamd_alphabet_standard(&a, protein);
  return gtamd_device_encode_files_alpha(paths, numfiles, &a, enc_out, desc, desclen, info,
                                         err, errlen);
}

int gtamd_device_encode_files_alpha(const char *const *paths, size_t numfiles,
                                    const gtamd_alphabet *a, gtamd_encoder **enc_out,
                                    char **desc, uint64_t *desclen,
                                    gtamd_encinfo *info, char *err, size_t errlen)
{
  uint8_t **raw = calloc(numfiles ? numfiles : 1, sizeof *raw);
  uint64_t *rawlen = calloc(numfiles ? numfiles : 1, sizeof *rawlen);
  gtamd_encoder *de = NULL;
  gtamd_encode_summary sum;
  uint32_t *dfile = NULL, *rfile = NULL, *rseq = NULL, *rdesc = NULL;
  gtamd_fastq_record *rec = NULL;
  uint64_t *dstart = NULL, *dend = NULL, ndesc, total = 0, nrec = 0;
  char *dbuf = NULL;
  int rc = -1;

  *enc_out = NULL;
  if (info != NULL) memset(info, 0, sizeof *info);
  if (raw == NULL || rawlen == NULL) goto nomem;
  if ((de = gtamd_encoder_create_map(0, a->symbolmap, a->numofchars, a->bitspersymbol)) == NULL)
    goto deverr;
  for (size_t f = 0; f < numfiles; f++) {
    const int src = gtamd_read_input_file(paths[f], &raw[f], &rawlen[f]);
    if (src != 0) {
      gtamd_read_input_error(src, paths[f], err, errlen);
      goto done;
    }
    if (gtamd_encoder_add_file(de, paths[f], raw[f], rawlen[f]) != 0) goto deverr;
  }
  if (gtamd_encoder_finish(de) != 0) {
    if (gtamd_encoder_declined(de)) rc = GTAMD_DEVICE_DECLINED;
    goto deverr;
  }
  if (gtamd_encoder_get_summary(de, &sum) != 0) goto deverr;
  if (info != NULL) {
    info->numfiles = numfiles;
    info->filelengthtab = calloc(numfiles ? numfiles : 1, sizeof *info->filelengthtab);
    if (info->filelengthtab == NULL) goto nomem;
    memcpy(info->originaldistribution, sum.originaldistribution,
           sizeof info->originaldistribution);
    for (size_t f = 0; f < numfiles; f++)
      if (gtamd_encoder_file_lengths(de, f, &info->filelengthtab[f].length,
                                     &info->filelengthtab[f].effectivelength) != 0)
        goto deverr;
    if ((nrec = gtamd_encoder_num_fastq_records(de)) > 0) {
      /* FASTQ: the table is booked per buffer fill of the reference's reader */
      rfile = malloc(4 * nrec); rseq = malloc(4 * nrec); rdesc = malloc(4 * nrec);
      rec = malloc(nrec * sizeof *rec);
      if (rfile == NULL || rseq == NULL || rdesc == NULL || rec == NULL) goto nomem;
      if (gtamd_encoder_get_fastq_records(de, rfile, rseq, rdesc, nrec) != 0) goto deverr;
      for (uint64_t k = 0; k < nrec; k++) {
        rec[k].seqlen = rseq[k]; rec[k].desclen = rdesc[k]; rec[k].file = rfile[k];
      }
      memset(info->filelengthtab, 0, numfiles * sizeof *info->filelengthtab);
      gtamd_fastq_filelengths(rec, nrec, numfiles - 1, info->filelengthtab);
    }
  }
  if (desc != NULL) {
    /* the descriptions, NUL-separated, carriage returns dropped
       (src/core/sequence_buffer_fasta.c:112-121) */
    uint64_t fill = 0;
    ndesc = gtamd_encoder_num_descriptions(de);
    dfile = malloc(4 * (ndesc + 1)); dstart = malloc(8 * (ndesc + 1));
    dend = malloc(8 * (ndesc + 1));
    if (dfile == NULL || dstart == NULL || dend == NULL) goto nomem;
    if (gtamd_encoder_get_descriptions(de, dfile, dstart, dend, ndesc + 1) != 0) goto deverr;
    for (uint64_t k = 0; k < ndesc; k++) total += dend[k] - dstart[k] + 1;
    if ((dbuf = malloc(total + 1)) == NULL) goto nomem;
    for (uint64_t k = 0; k < ndesc; k++) {
      const uint8_t *p = raw[dfile[k]] + dstart[k], *end = raw[dfile[k]] + dend[k];
      for (; p < end; p++) if (*p != '\r') dbuf[fill++] = (char) *p;
      dbuf[fill++] = 0;
    }
    *desc = dbuf; *desclen = fill;
    dbuf = NULL;
  }
  *enc_out = de; de = NULL;
  rc = 0;
  goto done;
deverr:
  snprintf(err, errlen, "%s", gtamd_esa_last_error());
  goto done;
nomem:
  snprintf(err, errlen, "out of memory while reading the input files");
done:
  if (raw != NULL) for (size_t f = 0; f < numfiles; f++) free(raw[f]);
  free(raw); free(rawlen); free(dfile); free(dstart); free(dend); free(dbuf);
  free(rfile); free(rseq); free(rdesc); free(rec);
  gtamd_encoder_destroy(de);
  if (rc != 0 && info != NULL) gtamd_encinfo_free(info);
  return rc;
}

int gtamd_write_esq_device(const char *indexname, const char *const *paths,
                           size_t numfiles, const gtamd_encoder *enc,
                           int protein, const gtamd_encinfo *info, int write_ssp,
                           const char *sat, gtamd_seqstats *ss, char *err, size_t errlen)
{
  gtamd_alphabet a;
  gtamd_alphabet_standard(&a, protein);
  return gtamd_write_esq_device_alpha(indexname, paths, numfiles, enc, &a, info, write_ssp,
                                      sat, ss, err, errlen);
}

int gtamd_write_esq_device_alpha(const char *indexname, const char *const *paths,
                                 size_t numfiles, const gtamd_encoder *enc,
                                 const gtamd_alphabet *a, const gtamd_encinfo *info,
                                 int write_ssp, const char *sat, gtamd_seqstats *ss,
                                 char *err, size_t errlen)
{
  gtamd_encode_summary sum;
  gtamd_seqanalysis an;
  gtamd_esq_sections sec;
  uint64_t *twobit = NULL, *specialbits = NULL, *wc_start = NULL, *wc_len = NULL,
           *seppos = NULL, n;
  uint8_t *packed = NULL, *plain = NULL;
  int rc = -1, need_tb, need_sb, need_pk, need_wc, need_sep;

  if (gtamd_encoder_get_summary(enc, &sum) != 0) goto deverr;
  gtamd_analysis_from_summary(&sum, a->numofchars, &an);
  if (gtamd_force_sat(&an, sat, a->numofchars != 4, err, errlen) != 0) return -1;
  if (ss != NULL) *ss = an.ss;
  n = an.ss.totallength;
  gtamd_esq_needs(&an, write_ssp, &need_tb, &need_sb, &need_pk, &need_wc, &need_sep);
  memset(&sec, 0, sizeof sec);
  if (an.sat == GTAMD_SAT_DIRECTACCESS) {
    if ((plain = malloc(n ? n : 1)) == NULL) goto nomem;
    if (gtamd_encoder_copy_symbols(enc, plain, 0, n) != 0) goto deverr;
  }
  if (need_pk) {
    if ((packed = malloc((a->bitspersymbol * n + 7) / 8 + 1)) == NULL) goto nomem;
    if (gtamd_encoder_pack_bytecompress(enc, packed, (a->bitspersymbol * n + 7) / 8 + 1) != 0) goto deverr;
  }
  if (need_tb) {
    if ((twobit = malloc(8 * (n < 32 ? 2 : 2 + (n - 1) / 32))) == NULL) goto nomem;
    if (gtamd_encoder_pack_twobit(enc, an.sat == GTAMD_SAT_BITACCESS,
                                  gtamd_least_probable(&an), twobit,
                                  n < 32 ? 2 : 2 + (n - 1) / 32) != 0) goto deverr;
  }
  if (need_sb) {
    if ((specialbits = malloc(8 * (1 + (n + 63) / 64))) == NULL) goto nomem;
    if (gtamd_encoder_pack_specialbits(enc, specialbits, 1 + (n + 63) / 64) != 0) goto deverr;
  }
  if (need_wc) {
    wc_start = malloc(8 * (sum.realwildcardranges + 1));
    wc_len = malloc(8 * (sum.realwildcardranges + 1));
    if (wc_start == NULL || wc_len == NULL) goto nomem;
    if (gtamd_encoder_get_wildcard_runs(enc, wc_start, wc_len, sum.realwildcardranges + 1) != 0) goto deverr;
  }
  if (need_sep) {
    if ((seppos = malloc(8 * sum.numofsequences)) == NULL) goto nomem;
    if (gtamd_encoder_get_separators(enc, seppos, sum.numofsequences) != 0) goto deverr;
  }
  sec.twobit = twobit; sec.specialbits = specialbits; sec.packed = packed;
  sec.plain = plain;
  sec.wc_start = wc_start; sec.wc_len = wc_len; sec.wc_runs = sum.realwildcardranges;
  sec.seppos = seppos;
  rc = gtamd_write_esq_sections(indexname, paths, numfiles, a, &an, info,
                                write_ssp, &sec, err, errlen);
  goto done;
deverr:
  snprintf(err, errlen, "%s", gtamd_esa_last_error());
  goto done;
nomem:
  snprintf(err, errlen, "out of memory while writing the encoded sequence");
done:
  free(twobit); free(specialbits); free(packed); free(plain); free(wc_start); free(wc_len);
  free(seppos);
  return rc;
}
