/*
  ref_driver.c -- TEST INFRASTRUCTURE ONLY (oracle side, never shipped).

  A small main() of our own that drives the *reference's* enhanced-suffix-array
  engine through its library seam:
    GtEncseqEncoder / GtEncseqLoader      src/core/encseq_api.h:223-448
    gt_recommendedprefixlength            src/match/sfx-apfxlen.h
    gt_Outlcpinfo_new                     src/match/sfx-lcpvalues.h:102
    gt_Sfxiterator_new_withadditionalvalues / _next / _longest / _delete
                                          src/match/sfx-suffixer.h:32-70
    gt_suffixsortspace_to_file            src/match/sfx-suffixgetset.h
    gt_outprjfile                         src/match/sfx-outprj.h
  i.e. the same call sequence as gt_runsuffixerator (src/match/sfx-run.c:428-717)
  for the options  -dna|-protein -suf [-lcp] [-bwt] -db FILE -indexname IDX
  without the option parser (src/core/option.c needs the generated gt_config.h,
  which we neither have nor fake).

  All reference sources are compiled where they lie under /root/reference by
  oracle/Makefile.ref; the binary lands in oracle/_ref/ (git-ignored).

  Second oracle: with -sain the tables come from the
  reference's OTHER construction -- SA-IS + the Phi algorithm,
    gt_sain_encseq_sortsuffixes           src/match/sfx-sain.c:1577
    gt_plain_lcp_phialgorithm             src/match/sfx-linlcp.c:131
  (uint32 only, what `gt dev sain` runs) -- written in the same file layouts, so
  that the two constructions of the reference can be compared with each other
  and with the goldens (tests/test_oracle_golden.py).

  usage: gt_ref_sfx (-dna|-protein) [-suf] [-lcp] [-bwt] [-pl K] [-dc V] [-sain]
                    [-dir fwd|rev|cpl|rcl] [-mirrored] [-sat TYPE] [-bck] [-suftabuint] [-clipdesc] [-smap FILE] [-lossless]
                    -db FASTA... -indexname IDX [-time]
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "core/alphabet_api.h"
#include "core/class_alloc_lock.h"
#include "core/combinatorics.h"
#include "core/defined-types.h"
#include "core/encseq_api.h"
#include "core/error_api.h"
#include "core/fa.h"
#include "core/log.h"
#include "core/ma.h"
#include "core/str_array_api.h"
#include "core/symbol.h"
#include "core/chardef.h"
#include "core/yarandom.h"
#include "match/esa-fileend.h"
#include "match/sfx-apfxlen.h"
#include "match/sfx-lcpvalues.h"
#include "match/sfx-outprj.h"
#include "match/sfx-strategy.h"
#include "match/sfx-suffixer.h"
#include "match/sfx-sain.h"
#include "match/sfx-linlcp.h"

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static FILE *open_tab(const char *indexname, const char *suffix)
{
  char path[4096];
  FILE *fp;
  snprintf(path, sizeof path, "%s%s", indexname, suffix);
  fp = fopen(path, "wb");
  if (fp == NULL) { perror(path); exit(EXIT_FAILURE); }
  return fp;
}

int main(int argc, char **argv)
{
  const char *db[64], *indexname = NULL, *sat = NULL, *smap = NULL;
  int numdb = 0;
  bool dna = true, want_suf = false, want_lcp = false, want_bwt = false,
       showtime = false, haserr = false, mirrored = false, want_bck = false,
       suftabuint = false, clipdesc = false, lossless = false, sain = false;
  GtReadmode readmode = GT_READMODE_FORWARD;
  unsigned int userpl = 0, dc = 0, prefixlength, numofchars;
  int i;
  GtError *err;
  GtEncseqEncoder *ee;
  GtEncseqLoader *el;
  GtEncseq *encseq = NULL;
  GtStrArray *dbs;
  GtOutlcpinfo *outlcpinfo = NULL;
  Sfxstrategy strategy;
  Sfxiterator *sfi;
  FILE *fpsuf = NULL, *fpbwt = NULL, *fpbck = NULL;
  GtUword numberofallsortedsuffixes = 0, totallength;
  Definedunsignedlong longest;
  double t0, t_encode, t_esa;

  for (i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-dna")) dna = true;
    else if (!strcmp(argv[i], "-protein")) dna = false;
    else if (!strcmp(argv[i], "-suf")) want_suf = true;
    else if (!strcmp(argv[i], "-lcp")) want_lcp = true;
    else if (!strcmp(argv[i], "-bwt")) want_bwt = true;
    else if (!strcmp(argv[i], "-time")) showtime = true;
    else if (!strcmp(argv[i], "-bck")) want_bck = true;
    else if (!strcmp(argv[i], "-suftabuint")) suftabuint = true;
    else if (!strcmp(argv[i], "-clipdesc")) clipdesc = true;
    else if (!strcmp(argv[i], "-lossless")) lossless = true;
    else if (!strcmp(argv[i], "-mirrored")) mirrored = true;
    else if (!strcmp(argv[i], "-sain")) sain = true;
    else if (!strcmp(argv[i], "-dir") && i + 1 < argc) {
      const char *d = argv[++i];
      readmode = !strcmp(d, "rev") ? GT_READMODE_REVERSE
               : !strcmp(d, "cpl") ? GT_READMODE_COMPL
               : !strcmp(d, "rcl") ? GT_READMODE_REVCOMPL : GT_READMODE_FORWARD;
    }
    else if (!strcmp(argv[i], "-pl") && i + 1 < argc)
      userpl = (unsigned int) atoi(argv[++i]);
    else if (!strcmp(argv[i], "-dc") && i + 1 < argc)
      dc = (unsigned int) atoi(argv[++i]);
    else if (!strcmp(argv[i], "-db") && i + 1 < argc) {
      /* one or more files, up to the next option */
      while (i + 1 < argc && argv[i + 1][0] != '-' && numdb < 64)
        db[numdb++] = argv[++i];
    }
    else if (!strcmp(argv[i], "-sat") && i + 1 < argc) sat = argv[++i];
    else if (!strcmp(argv[i], "-smap") && i + 1 < argc) smap = argv[++i];
    else if (!strcmp(argv[i], "-indexname") && i + 1 < argc)
      indexname = argv[++i];
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  if (numdb == 0 || indexname == NULL) {
    fprintf(stderr, "need -db and -indexname\n");
    return 2;
  }

  /* the parts of gt_lib_init (src/core/init.c:100-123) that do not need the
     option parser */
  gt_ma_init(false);
  gt_fa_init();
  gt_log_init();
  gt_symbol_init();
  gt_class_alloc_lock_init();
  gt_ya_rand_init(0);
  gt_combinatorics_init();

  err = gt_error_new();
  t0 = now_s();
  ee = gt_encseq_encoder_new();
  if (smap != NULL) {
    /* -smap FILE (an existing file: no lookup in gtdata/trans) */
    if (gt_encseq_encoder_use_symbolmap_file(ee, smap, err) != 0) {
      fprintf(stderr, "gt suffixerator: error: %s\n", gt_error_get(err));
      return EXIT_FAILURE;
    }
  }
  else if (dna) gt_encseq_encoder_set_input_dna(ee);
  else gt_encseq_encoder_set_input_protein(ee);
  if (clipdesc) gt_encseq_encoder_clip_desc(ee);   /* -clipdesc, encseq_options.c */
  if (lossless) gt_encseq_encoder_enable_lossless_support(ee);   /* -lossless */
  if (sat != NULL) {
    /* -sat of src/core/encseq_options.c: force an access type */
    if (gt_encseq_encoder_use_representation(ee, sat, err) != 0) {
      fprintf(stderr, "gt suffixerator: error: %s\n", gt_error_get(err));
      return EXIT_FAILURE;
    }
  }
  dbs = gt_str_array_new();
  for (i = 0; i < numdb; i++) gt_str_array_add_cstr(dbs, db[i]);
  if (gt_encseq_encoder_encode(ee, dbs, indexname, err) != 0) haserr = true;
  gt_encseq_encoder_delete(ee);
  gt_str_array_delete(dbs);
  if (!haserr) {
    el = gt_encseq_loader_new();
    gt_encseq_loader_disable_autosupport(el);
    gt_encseq_loader_do_not_require_des_tab(el);
    gt_encseq_loader_do_not_require_sds_tab(el);
    gt_encseq_loader_do_not_require_ssp_tab(el);
    if (mirrored) gt_encseq_loader_mirror(el);   /* src/core/encseq_options.c -mirrored */
    encseq = gt_encseq_loader_load(el, indexname, err);
    gt_encseq_loader_delete(el);
    if (encseq == NULL) haserr = true;
  }
  t_encode = now_s() - t0;
  if (haserr) {
    fprintf(stderr, "gt suffixerator: error: %s\n", gt_error_get(err));
    return EXIT_FAILURE;
  }
  totallength = gt_encseq_total_length(encseq);
  numofchars = gt_alphabet_num_of_chars(gt_encseq_alphabet(encseq));
  if (sain) {
    /* SA-IS + Phi: .suf (GtUword), .lcp (bytes, 255 = overflow), .llv (index, value) */
    GtUword nonspecial = totallength - gt_encseq_specialcharacters(encseq), idx,
            maxlcp = 0;
    GtUsainindextype *suftab;
    FILE *fp;
    if (gt_sain_checkmaxsequencelength(totallength, true, err) != 0) {
      fprintf(stderr, "gt dev sain: error: %s\n", gt_error_get(err));
      return EXIT_FAILURE;
    }
    t0 = now_s();
    suftab = gt_sain_encseq_sortsuffixes(encseq, readmode, false, false, NULL, NULL);
    if (want_suf) {
      fp = open_tab(indexname, GT_SUFTABSUFFIX);
      for (idx = 0; idx <= totallength; idx++) {
        GtUword v = (GtUword) suftab[idx];
        fwrite(&v, sizeof v, 1, fp);
      }
      fclose(fp);
    }
    if (want_bwt) {
      fp = open_tab(indexname, GT_BWTTABSUFFIX);
      for (idx = 0; idx <= totallength; idx++) {
        GtUword startpos = (GtUword) suftab[idx];
        fputc(startpos == 0 ? (int) UNDEFBWTCHAR
                            : (int) gt_encseq_get_encoded_char(encseq, startpos - 1, readmode),
              fp);
      }
      fclose(fp);
    }
    if (want_lcp) {
      GtUchar *seq = gt_malloc(totallength + 1);
      unsigned int *lcptab;
      FILE *fpllv = open_tab(indexname, GT_LARGELCPTABSUFFIX);
      if (totallength > 0)
        gt_encseq_extract_encoded(encseq, seq, 0, totallength - 1);
      lcptab = gt_plain_lcp_phialgorithm(false, &maxlcp, seq, true, nonspecial,
                                         totallength, suftab);
      fp = open_tab(indexname, GT_LCPTABSUFFIX);
      for (idx = 0; idx <= totallength; idx++) {
        GtUword v = (idx > 0 && idx < nonspecial) ? (GtUword) lcptab[idx] : 0;
        if (v >= 255UL) {
          fwrite(&idx, sizeof idx, 1, fpllv);
          fwrite(&v, sizeof v, 1, fpllv);
          v = 255UL;
        }
        fputc((int) v, fp);
      }
      fclose(fp);
      fclose(fpllv);
      gt_free(lcptab);
      gt_free(seq);
    }
    gt_free(suftab);
    if (showtime)
      printf("# TIME totallength=" GT_WU " sain=%.3f maxlcp=" GT_WU "\n", totallength,
             now_s() - t0, maxlcp);
    gt_encseq_delete(encseq);
    gt_error_delete(err);
    return EXIT_SUCCESS;
  }
  prefixlength = userpl > 0
                   ? userpl
                   : gt_recommendedprefixlength(numofchars, totallength,
                                     GT_RECOMMENDED_MULTIPLIER_DEFAULT, true);
  defaultsfxstrategy(&strategy,
                     gt_encseq_bitwise_cmp_ok(encseq) ? false : true);
  strategy.differencecover = dc;
  strategy.suftabuint = suftabuint;     /* -suftabuint, src/match/index_options.c:475 */

  t0 = now_s();
  if (want_lcp) {
    outlcpinfo = gt_Outlcpinfo_new(indexname, numofchars, prefixlength,
                                   false, false, NULL, NULL, err);
    if (outlcpinfo == NULL) haserr = true;
  }
  if (want_suf) fpsuf = open_tab(indexname, GT_SUFTABSUFFIX);
  if (want_bwt) fpbwt = open_tab(indexname, GT_BWTTABSUFFIX);
  /* -bck, src/match/sfx-run.c:157-160: gt_fa_fopen, the iterator closes it */
  if (want_bck) {
    fpbck = gt_fa_fopen_with_suffix(indexname, GT_BCKTABSUFFIX, "wb", err);
    if (fpbck == NULL) haserr = true;
  }
  longest.defined = false;
  longest.valueunsignedlong = 0;
  sfi = haserr ? NULL
               : gt_Sfxiterator_new_withadditionalvalues(encseq,
                     readmode, prefixlength, 1U, 0UL, outlcpinfo,
                     fpbck, &strategy, NULL, false, NULL, err);
  if (sfi == NULL) haserr = true;
  while (!haserr) {
    GtUword numberofsuffixes, pos;
    bool specialsuffixes = false;
    const GtSuffixsortspace *sssp
      = gt_Sfxiterator_next(&numberofsuffixes, &specialsuffixes, sfi);
    if (sssp == NULL) break;
    if (fpsuf != NULL)
      gt_suffixsortspace_to_file(fpsuf, sssp, numberofsuffixes);
    if (fpbwt != NULL) {
      /* same rule as bwttab2file, src/match/sfx-run.c:173-210 */
      for (pos = 0; pos < numberofsuffixes; pos++) {
        GtUword startpos = gt_suffixsortspace_getdirect(sssp, pos);
        GtUchar cc = startpos == 0
                       ? (GtUchar) UNDEFBWTCHAR
                       : gt_encseq_get_encoded_char(encseq, startpos - 1,
                                                    readmode);
        fputc((int) cc, fpbwt);
      }
    }
    numberofallsortedsuffixes += numberofsuffixes;
  }
  if (!haserr) {
    longest.defined = true;
    longest.valueunsignedlong = gt_Sfxiterator_longest(sfi);
    /* src/match/sfx-run.c:303-310 */
    if (fpbck != NULL && gt_Sfxiterator_bcktab2file(fpbck, sfi, err) != 0)
      haserr = true;
  }
  if (sfi != NULL && gt_Sfxiterator_delete(sfi, err) != 0) haserr = true;
  if (fpsuf != NULL) fclose(fpsuf);
  if (fpbwt != NULL) fclose(fpbwt);
  if (!haserr) {
    GtUword numoflargelcpvalues = 0, maxbranchdepth = 0;
    double averagelcp = 0.0;
    if (outlcpinfo != NULL) {
      /* src/match/sfx-run.c:671-681 */
      numoflargelcpvalues = gt_Outlcpinfo_numoflargelcpvalues(outlcpinfo);
      maxbranchdepth = gt_Outlcpinfo_maxbranchdepth(outlcpinfo);
      averagelcp = gt_Outlcpinfo_lcptabsum(outlcpinfo) /
                   numberofallsortedsuffixes;
    }
    if (gt_outprjfile(indexname, readmode, encseq,
                      numberofallsortedsuffixes, prefixlength,
                      numoflargelcpvalues, averagelcp, maxbranchdepth,
                      &longest, err) != 0)
      haserr = true;
  }
  gt_Outlcpinfo_delete(outlcpinfo);
  t_esa = now_s() - t0;
  gt_encseq_delete(encseq);
  if (haserr) {
    fprintf(stderr, "gt suffixerator: error: %s\n", gt_error_get(err));
    return EXIT_FAILURE;
  }
  if (showtime)
    printf("# TIME totallength=" GT_WU " prefixlength=%u encode=%.3f "
           "esa=%.3f\n", totallength, prefixlength, t_encode, t_esa);
  gt_error_delete(err);
  return EXIT_SUCCESS;
}
