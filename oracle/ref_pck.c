/*
  ref_pck.c -- TEST INFRASTRUCTURE ONLY (oracle side, never shipped).

  A small main() of our own around the *reference's* packed-index construction
  from an existing suffix-array project (what `gt packedindex trsuftab INDEX`
  runs, src/tools/gt_packedindex_trsuftab.c:44-79):

    gt_trSuftab2BWTSeq                    src/match/eis-bwtseq-construct.c:64-92
      -> gt_createBWTSeqGeneric           src/match/eis-bwtseq-extinfo.c:558-676
      -> gt_newGenBlockEncIdxSeq          src/match/eis-blockcomp.c:304-655

  with the parameter block filled in the way gt_registerPackedIndexOptions and
  gt_computePackedIndexDefaults do (src/match/eis-bwtseq-param.c:25-103), but
  without the option parser (src/core/option.c needs the generated gt_config.h,
  which we neither have nor fake).  Reads INDEX.prj/.esq/.suf/.bwt (written by
  oracle/_ref/gt_ref_sfx or by gt-suffixerator-amd), writes INDEX.bdx.

  With -mkindex the index is built the way `gt packedindex mkindex` does
  (gt_runsuffixerator(false, ...) -> run_packedindexconstruction,
  src/match/sfx-run.c:369-425): the BWT comes from the suffixerator itself
  through the sfxInterface, which also supplies sequence statistics (they
  change the widths of the occurrence counters and one size bound); INDEX.esq
  must exist (written by gt_ref_sfx).

  usage: gt_ref_pck [-bsize B] [-blbuck K] [-locfreq F] [-locbitmap yes|no]
                    [-sprank] [-ctxilog I] [-mkindex [-dir fwd|rev|cpl|rcl]] INDEX
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "core/error_api.h"
#include "core/logger.h"
#include "core/str_api.h"
#include "core/class_alloc_lock.h"
#include "core/combinatorics.h"
#include "core/fa.h"
#include "core/log.h"
#include "core/ma.h"
#include "core/symbol.h"
#include "core/yarandom.h"
#include "match/eis-bwtseq.h"
#include "match/eis-bwtseq-construct.h"
#include "match/eis-bwtseq-param.h"
#include "match/eis-bwtseq-context-param.h"
#include "match/eis-encidxseq.h"
#include "match/eis-suffixerator-interface.h"
#include "match/sfx-apfxlen.h"
#include "match/sfx-strategy.h"
#include "match/sfx-outprj.h"
#include "core/defined-types.h"
#include "core/alphabet_api.h"
#include "core/encseq_api.h"
#include "core/readmode_api.h"

int main(int argc, char **argv)
{
  struct bwtParam params;
  unsigned bsize = 8, blbuck = 8, locfreq = 16;
  int locbitmap = -1, i, mkindex = 0, sprank = 0, ctxilog = CTX_MAP_ILOG_NOMAP;
  GtReadmode readmode = GT_READMODE_FORWARD;
  const char *index = NULL;
  GtError *err;
  GtLogger *logger;
  GtStr *project;
  BWTSeq *bwtSeq;

  for (i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-bsize") && i + 1 < argc) bsize = (unsigned) atoi(argv[++i]);
    else if (!strcmp(argv[i], "-blbuck") && i + 1 < argc) blbuck = (unsigned) atoi(argv[++i]);
    else if (!strcmp(argv[i], "-locfreq") && i + 1 < argc) locfreq = (unsigned) atoi(argv[++i]);
    else if (!strcmp(argv[i], "-locbitmap") && i + 1 < argc) locbitmap = !strcmp(argv[++i], "yes");
    else if (!strcmp(argv[i], "-mkindex")) mkindex = 1;
    else if (!strcmp(argv[i], "-sprank")) sprank = 1;
    else if (!strcmp(argv[i], "-ctxilog") && i + 1 < argc) ctxilog = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-dir") && i + 1 < argc) {
      /* -dir of the index options, with -mkindex only (trsuftab takes the read
         mode from INDEX.prj) */
      i++;
      readmode = !strcmp(argv[i], "rev") ? GT_READMODE_REVERSE
               : !strcmp(argv[i], "cpl") ? GT_READMODE_COMPL
               : !strcmp(argv[i], "rcl") ? GT_READMODE_REVCOMPL : GT_READMODE_FORWARD;
    }
    else if (argv[i][0] != '-') index = argv[i];
    else { fprintf(stderr, "gt_ref_pck: unknown option %s\n", argv[i]); return 2; }
  }
  if (!index) { fprintf(stderr, "usage: gt_ref_pck [options] INDEX\n"); return 2; }

  /* the parts of gt_lib_init (src/core/init.c:100-123) that do not need the
     generated configuration, as in ref_driver.c */
  gt_ma_init(false);
  gt_fa_init();
  gt_log_init();
  gt_symbol_init();
  gt_class_alloc_lock_init();
  gt_ya_rand_init(0);
  gt_combinatorics_init();
  err = gt_error_new();
  project = gt_str_new_cstr(index);
  memset(&params, 0, sizeof params);
  params.seqParams.encType = BWT_ON_BLOCK_ENC;
  params.seqParams.encParams.blockEnc.blockSize = bsize;
  params.seqParams.encParams.blockEnc.bucketBlocks = blbuck;
  params.seqParams.EISFeatureSet = gt_convertBWTOptFlags2EISFeatures(BWTDEFOPT_MULTI_QUERY);
  params.locateInterval = locfreq;
  params.sourceRankInterval = -1;
  params.ctxMapILog = ctxilog;   /* -ctxilog: INDEX.<ilog>cxm beside INDEX.bdx */
  params.projectName = project;
  params.featureToggles = BWTBaseFeatures;
  if (locbitmap >= 0)
    params.featureToggles |= locbitmap ? BWTLocateBitmap : BWTLocateCount;
  else if (locfreq) {
    /* estimateBestLocateTypeFeature, eis-bwtseq-param.c:71-87 */
    unsigned segmentLen = gt_estimateSegmentSize(&params.seqParams);
    if (segmentLen > (segmentLen + 1) * gt_requiredUIntBits(segmentLen) / locfreq)
      params.featureToggles |= BWTLocateCount;
    else
      params.featureToggles |= BWTLocateBitmap;
  }
  /* -sprank: gt_computePackedIndexDefaults, eis-bwtseq-param.c:98-100 */
  if (sprank) params.featureToggles |= BWTReversiblySorted;
  logger = gt_logger_new(false, GT_LOGGER_DEFLT_PREFIX, stdout);
  if (!mkindex)
    bwtSeq = gt_trSuftab2BWTSeq(&params, logger, err);
  else {
    /* run_packedindexconstruction, src/match/sfx-run.c:369-425 */
    GtEncseqLoader *el = gt_encseq_loader_new();
    GtEncseq *encseq;
    Sfxstrategy strategy;
    sfxInterface *si;
    unsigned int numofchars, prefixlength;
    gt_encseq_loader_disable_autosupport(el);
    gt_encseq_loader_do_not_require_des_tab(el);
    gt_encseq_loader_do_not_require_sds_tab(el);
    gt_encseq_loader_do_not_require_ssp_tab(el);
    encseq = gt_encseq_loader_load(el, index, err);
    gt_encseq_loader_delete(el);
    if (encseq == NULL) {
      fprintf(stderr, "gt_ref_pck: error: %s\n", gt_error_get(err));
      return 1;
    }
    numofchars = gt_alphabet_num_of_chars(gt_encseq_alphabet(encseq));
    if (numofchars > 10U && params.seqParams.encParams.blockEnc.blockSize > 3U)
      params.seqParams.encParams.blockEnc.blockSize = 3U;
    prefixlength = gt_recommendedprefixlength(numofchars, gt_encseq_total_length(encseq),
                                              GT_RECOMMENDED_MULTIPLIER_DEFAULT, true);
    defaultsfxstrategy(&strategy, gt_encseq_bitwise_cmp_ok(encseq) ? false : true);
    si = gt_newSfxInterface(readmode, prefixlength, 1U, 0UL, &strategy, encseq,
                            NULL, false, gt_encseq_total_length(encseq) + 1, logger, err);
    bwtSeq = si != NULL ? gt_createBWTSeqFromSfxI(&params, si, err) : NULL;
    if (bwtSeq != NULL) {
      /* the project file of such a run: gt_runsuffixerator calls gt_outprjfile
         with what its outfileinfo holds when no suffix table went through it
         (src/match/sfx-run.c:598-604 initial values, :661-690) */
      Definedunsignedlong longest;
      longest.defined = false;
      longest.valueunsignedlong = 0;
      if (gt_outprjfile(index, readmode, encseq, 0, prefixlength, 0, 0.0, 0,
                        &longest, err) != 0) {
        fprintf(stderr, "gt_ref_pck: error: %s\n", gt_error_get(err));
        return 1;
      }
      printf("featureToggles=%d\n", params.featureToggles);
      gt_deleteBWTSeq(bwtSeq);
      gt_deleteSfxInterface(si);
      gt_encseq_delete(encseq);
      return 0;
    }
  }
  if (!bwtSeq) {
    fprintf(stderr, "gt_ref_pck: error: %s\n", gt_error_is_set(err) ? gt_error_get(err) : "?");
    return 1;
  }
  printf("featureToggles=%d\n", params.featureToggles);
  gt_deleteBWTSeq(bwtSeq);
  gt_logger_delete(logger);
  gt_str_delete(project);
  gt_error_delete(err);
  return 0;
}
