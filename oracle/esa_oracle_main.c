/*
  esa_oracle_main.c -- TEST INFRASTRUCTURE ONLY (see esa_oracle.h).
  Command-line front end of the CPU restatement; writes IDX.suf/.lcp/.llv/
  .bwt/.prj in the reference's on-disk layouts so the files can be compared
  byte for byte with those of oracle/_ref/gt_ref_sfx.

  usage: esa_oracle (-dna|-protein) [-suf] [-lcp] [-bwt] [-kasai]
                    [-dir fwd|rev|cpl|rcl] [-mirrored] -db FASTA -indexname IDX
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "esa_oracle.h"

static void dump(const char *idx, const char *sfx, const void *p, size_t bytes)
{
  char path[4096];
  FILE *fp;
  snprintf(path, sizeof path, "%s%s", idx, sfx);
  fp = fopen(path, "wb");
  if (fp == NULL || fwrite(p, 1, bytes, fp) != bytes) {
    perror(path);
    exit(EXIT_FAILURE);
  }
  fclose(fp);
}

int main(int argc, char **argv)
{
  const char *db = NULL, *idx = NULL;
  int protein = 0, suf = 0, lcp = 0, bwt = 0, kasai = 0, i, readmode = 0,
      mirrored = 0;
  uint8_t *enc, *lcpb, *bwtb;
  uint64_t n, *sa, *lcpw = NULL, *llv = NULL, pairs = 0;
  char err[1024], path[4096];
  ora_seqstats ss;
  ora_esastats es;
  uint32_t sigma;

  for (i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-dna")) protein = 0;
    else if (!strcmp(argv[i], "-protein")) protein = 1;
    else if (!strcmp(argv[i], "-suf")) suf = 1;
    else if (!strcmp(argv[i], "-lcp")) lcp = 1;
    else if (!strcmp(argv[i], "-bwt")) bwt = 1;
    else if (!strcmp(argv[i], "-kasai")) kasai = 1;
    else if (!strcmp(argv[i], "-mirrored")) mirrored = 1;
    else if (!strcmp(argv[i], "-dir") && i + 1 < argc) {
      const char *d = argv[++i];
      readmode = !strcmp(d, "rev") ? 1 : !strcmp(d, "cpl") ? 2 : !strcmp(d, "rcl") ? 3 : 0;
    }
    else if (!strcmp(argv[i], "-db") && i + 1 < argc) db = argv[++i];
    else if (!strcmp(argv[i], "-indexname") && i + 1 < argc) idx = argv[++i];
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  if (db == NULL || idx == NULL) {
    fprintf(stderr, "need -db and -indexname\n");
    return 2;
  }
  if (ora_encode_fasta(db, protein, &enc, &n, err, sizeof err) != 0) {
    fprintf(stderr, "gt suffixerator: error: %s\n", err);
    return EXIT_FAILURE;
  }
  sigma = protein ? 20 : 4;
  /* .prj describes the sequence as stored; the tables the sequence as read */
  ora_seqstats_compute(enc, n, sigma, strlen(db) + 1, 1, &ss);
  if (mirrored) {
    uint8_t *m = ora_mirror(enc, n);
    ora_seqstats_mirror(&ss, n > 0 && enc[n - 1] == ORA_WILDCARD);
    free(enc);
    enc = m;
    n = 2 * n + 1;
  }
  ora_apply_readmode(enc, n, readmode);
  sa = malloc((n + 1) * sizeof *sa);
  ora_suffix_array(enc, n, sa);
  if (suf) dump(idx, ".suf", sa, (n + 1) * sizeof *sa);
  if (lcp) {
    lcpw = malloc((n + 1) * sizeof *lcpw);
    lcpb = malloc(n + 1);
    if (kasai) ora_lcp_kasai(enc, n, sa, lcpw);
    else ora_lcp_direct(enc, n, sa, lcpw);
    pairs = ora_lcp_to_bytes(lcpw, n + 1, lcpb, NULL);
    llv = malloc((2 * pairs + 1) * sizeof *llv);
    ora_lcp_to_bytes(lcpw, n + 1, lcpb, llv);
    dump(idx, ".lcp", lcpb, n + 1);
    dump(idx, ".llv", llv, 2 * pairs * sizeof *llv);
    free(lcpb);
  }
  if (bwt) {
    bwtb = malloc(n + 1);
    ora_bwt(enc, n, sa, bwtb);
    dump(idx, ".bwt", bwtb, n + 1);
    free(bwtb);
  }
  ora_esastats_compute(enc, n, sa, lcpw, ora_recommended_prefixlength(sigma, n),
                       &es);
  snprintf(path, sizeof path, "%s.prj", idx);
  if (ora_write_prj(path, &ss, &es, lcp, readmode, mirrored) != 0) { perror(path); return 1; }
  free(llv); free(lcpw); free(sa); free(enc);
  return EXIT_SUCCESS;
}
