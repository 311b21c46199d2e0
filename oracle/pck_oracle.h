/*
  pck_oracle.h -- TEST INFRASTRUCTURE ONLY (see pck_oracle.c).  Never linked
  into, imported by or called from the product.
*/
#ifndef PCK_ORACLE_H
#define PCK_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum BWTFeatures, src/match/eis-bwtseq-param.h:78-94 */
#define ORA_PCK_LOCATE_BITMAP 1
#define ORA_PCK_LOCATE_COUNT  2
#define ORA_PCK_REVERSIBLY_SORTED 4   /* -sprank */

typedef struct {
  unsigned block_size;       /* -bsize, default 8 */
  unsigned bucket_blocks;    /* -blbuck, default 8 */
  unsigned locate_interval;  /* -locfreq, default 16, 0 = no locate information */
  int feature_toggles;       /* ORA_PCK_LOCATE_* as gt_computePackedIndexDefaults chooses */
  int with_statistics;       /* 0: `gt packedindex trsuftab` (tables from files, no
                                sequence statistics), 1: `gt packedindex mkindex` (BWT
                                from the suffixerator, with statistics) */
} ora_pck_params;

/* the feature toggles `gt packedindex` derives from its options
   (src/match/eis-bwtseq-param.c:69-103): locbitmap -1 = option not given */
int ora_pck_default_toggles(unsigned block_size, unsigned bucket_blocks,
                            unsigned locate_interval, int locbitmap);

/* the bytes of INDEX.bdx for a project with total_len = n + 1 table entries:
   bwt[total_len] (.bwt), suf[total_len] (.suf), seq[n] (encoded symbols),
   longest = index of suffix 0.  *out is malloc'ed (ora_pck_free). */
int ora_pck_bdx(const uint8_t *bwt, const uint64_t *suf, const uint8_t *seq,
                uint64_t total_len, unsigned sigma, uint64_t longest,
                const ora_pck_params *pp, uint8_t **out, size_t *out_len);
/* the bytes of INDEX.<ilog>cxm (-ctxilog; ilog < 0: the automatic interval);
   returns the interval log used, -1 for an invalid one */
int ora_pck_ctxmap(const uint64_t *suf, uint64_t total_len, int ilog, uint8_t **out,
                   size_t *out_len);
void ora_pck_free(uint8_t *p);
uint64_t ora_pck_last_var_bits(void);

#ifdef __cplusplus
}
#endif
#endif
