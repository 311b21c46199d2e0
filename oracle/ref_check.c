/*
  ref_check.c -- TEST INFRASTRUCTURE ONLY (oracle side, never shipped).

  Loads an index with the REFERENCE's own mappers and verifies it with the
  reference's own checkers -- what `gt dev sfxmap -tis -suf -lcp -bwt -des -ssp
  -esa INDEX` does (src/tools/gt_sfxmap.c:664-800):
    gt_mapsuffixarray                  src/match/esa-map.h:32 (maps .esq .ssp
                                       .des .suf .lcp .llv .bwt, reads .prj)
    gt_suftab_lightweightcheck         src/match/sfx-lwcheck.h:28
    gt_lcptab_lightweightcheck         src/match/sfx-linlcp.h:47 (Manzini's
                                       lcp9 from the mapped suffix table vs
                                       the stored .lcp/.llv)
  plus a direct check of .bwt against the mapped sequence and a walk over the
  descriptions.  Used by the GPU tests on indexes written by
  gt-suffixerator-amd: the reference must accept them as its own.

  usage: gt_ref_check INDEX      exit 0 and "ok ..." on success
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "core/alphabet_api.h"
#include "core/chardef.h"
#include "core/class_alloc_lock.h"
#include "core/combinatorics.h"
#include "core/encseq.h"
#include "core/error_api.h"
#include "core/fa.h"
#include "core/log.h"
#include "core/ma.h"
#include "core/symbol.h"
#include "core/yarandom.h"
#include "match/esa-map.h"
#include "match/sfx-linlcp.h"
#include "match/sfx-lwcheck.h"

static GtUchar accesschar(const void *encseq, GtUword position, GtReadmode readmode)
{
  return gt_encseq_get_encoded_char((const GtEncseq *) encseq, position, readmode);
}

static GtUword charcount(const void *encseq, GtUchar idx)
{
  return gt_encseq_charcount((const GtEncseq *) encseq, idx);
}

int main(int argc, char **argv)
{
  Suffixarray sa;
  GtError *err;
  GtUword totallength, i, desctotal = 0;
  if (argc != 2) { fprintf(stderr, "usage: %s INDEX\n", argv[0]); return 2; }
  gt_ma_init(false);
  gt_fa_init();
  gt_log_init();
  gt_symbol_init();
  gt_class_alloc_lock_init();
  gt_ya_rand_init(0);
  gt_combinatorics_init();
  err = gt_error_new();
  if (gt_mapsuffixarray(&sa, SARR_ALLTAB, argv[1], NULL, err) != 0) {
    fprintf(stderr, "gt_ref_check: error: %s\n", gt_error_get(err));
    return EXIT_FAILURE;
  }
  totallength = gt_encseq_total_length(sa.encseq);
  if (sa.numberofallsortedsuffixes != totallength + 1) {
    fprintf(stderr, "gt_ref_check: error: " GT_WU " suffixes for length " GT_WU "\n",
            sa.numberofallsortedsuffixes, totallength);
    return EXIT_FAILURE;
  }
  /* both exit(GT_EXIT_PROGRAMMING_ERROR) with a message at the first fault */
  gt_suftab_lightweightcheck(accesschar, charcount, sa.encseq, sa.readmode, totallength,
                             gt_encseq_alphabetnumofchars(sa.encseq), sa.suftab,
                             sizeof *sa.suftab, NULL);
  if (gt_lcptab_lightweightcheck(argv[1], sa.encseq, sa.readmode, sa.suftab, NULL,
                                 err) != 0) {
    fprintf(stderr, "gt_ref_check: error: %s\n", gt_error_get(err));
    return EXIT_FAILURE;
  }
  for (i = 0; i <= totallength; i++) {
    const GtUword p = sa.suftab[i];
    const GtUchar want = p == 0 ? (GtUchar) UNDEFBWTCHAR
                                : gt_encseq_get_encoded_char(sa.encseq, p - 1, sa.readmode);
    if (sa.bwttab[i] != want) {
      fprintf(stderr, "gt_ref_check: error: bwt[" GT_WU "] = %u, expected %u\n", i,
              (unsigned) sa.bwttab[i], (unsigned) want);
      return EXIT_FAILURE;
    }
  }
  for (i = 0; i < gt_encseq_num_of_sequences(sa.encseq); i++) {
    GtUword len;
    (void) gt_encseq_description(sa.encseq, &len, i);
    desctotal += len;
    if (gt_encseq_seqstartpos(sa.encseq, i) + gt_encseq_seqlength(sa.encseq, i) >
        totallength) {
      fprintf(stderr, "gt_ref_check: error: sequence " GT_WU " exceeds the index\n", i);
      return EXIT_FAILURE;
    }
  }
  printf("ok totallength=" GT_WU " sequences=" GT_WU " descriptionbytes=" GT_WU
         " prefixlength=%u\n", totallength, gt_encseq_num_of_sequences(sa.encseq),
         desctotal, sa.prefixlength);
  gt_freesuffixarray(&sa);
  gt_error_delete(err);
  return EXIT_SUCCESS;
}
