/*
  esa_oracle.h -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or
  called from the product (genometools_amd/, include/, the CLI).  Only tests/,
  __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

  A plain-C, single-threaded CPU restatement of what `gt suffixerator
  -suf -lcp -bwt` computes.  Every function cites
  the reference file:line whose behaviour it restates.  Parity pinned: the
  restatement is checked against tables produced by the reference itself
  (oracle/_ref/gt_ref_sfx, built from /root/reference by oracle/Makefile.ref)
  on the reference's own suffixerator fixtures; the md5 sums of those tables
  are committed under tests/golden/ (see tests/golden/make_golden.py).
*/
#ifndef ESA_ORACLE_H
#define ESA_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_WILDCARD  254u   /* src/core/chardef.h:33 */
#define ORA_SEPARATOR 255u   /* src/core/chardef.h:34 */
#define ORA_UNDEFBWT  254u   /* src/core/chardef.h:65 UNDEFBWTCHAR */
#define ORA_LCPOVERFLOW 255u /* src/match/lcpoverflow.h:24 */

typedef struct {
  uint64_t totallength, specialcharacters, specialranges, realspecialranges,
           lengthofspecialprefix, lengthofspecialsuffix, wildcards,
           wildcardranges, realwildcardranges, lengthofwildcardprefix,
           lengthofwildcardsuffix, numofsequences;
  uint32_t numofchars;
} ora_seqstats;

typedef struct {
  uint64_t numberofallsortedsuffixes, longest, largelcpvalues, maxbranchdepth;
  double   lcptabsum;      /* the masked sum of SURVEY 0.4 */
  uint32_t prefixlength;
} ora_esastats;

/* FASTA -> encoded symbols (0..sigma-1, 254 wildcard, 255 separator).
   Returns 0, or -1 with a message in err (same wording as the reference).
   *enc is malloc'ed, caller frees. */
int ora_encode_fasta(const char *path, int protein, uint8_t **enc,
                     uint64_t *n, char *err, size_t errlen);

/* readmode 0 forward, 1 reverse, 2 complement, 3 reverse complement
   (src/core/readmode_api.h:24-27): the sequence as the reference reads it in
   that mode, in place.  Complement only for DNA (3 - code), specials stay. */
void ora_apply_readmode(uint8_t *enc, uint64_t n, int readmode);
/* -mirrored (src/core/encseq_api.h:190-198): enc + separator + reverse
   complement of enc; returns a malloc'ed sequence of 2n+1 symbols */
uint8_t *ora_mirror(const uint8_t *enc, uint64_t n);

/* statistics of the encoded sequence that go into .prj */
void ora_seqstats_compute(const uint8_t *enc, uint64_t n, uint32_t numofchars,
                          uint64_t lengthofdbfilenames, uint64_t numofdbfiles,
                          ora_seqstats *st);

/* statistics a mirrored encseq reports, from those of the original sequence
   (src/core/encseq.c:4960-5054): doubled counts, +1 for the central separator,
   and one range less when the original ends with a wildcard (it merges with
   the central separator; encseq.c:4970-4976 tests position totallength-1 of
   the unmirrored sequence); prefix/suffix lengths stay those of the original */
void ora_seqstats_mirror(ora_seqstats *st, int last_symbol_is_wildcard);

/* gt_recommendedprefixlength restated */
uint32_t ora_recommended_prefixlength(uint32_t numofchars, uint64_t n);

/* suffix array by comparison sort on the ordering rule; sa has n+1 entries */
void ora_suffix_array(const uint8_t *enc, uint64_t n, uint64_t *sa);

/* full-width LCP table (n+1 entries, lcp[0]=0) by direct comparison */
void ora_lcp_direct(const uint8_t *enc, uint64_t n, const uint64_t *sa,
                    uint64_t *lcp);
/* the same table in O(n) (Kasai et al. with "specials never match") */
void ora_lcp_kasai(const uint8_t *enc, uint64_t n, const uint64_t *sa,
                   uint64_t *lcp);

/* bwt[i] = enc[sa[i]-1], 254 when sa[i]==0 */
void ora_bwt(const uint8_t *enc, uint64_t n, const uint64_t *sa, uint8_t *bwt);

/* .lcp bytes + .llv pairs from the full-width table.  llv must have room for
   2*(#values >= 255) entries; returns the number of pairs. */
uint64_t ora_lcp_to_bytes(const uint64_t *lcp, uint64_t nplus1, uint8_t *lcpb,
                          uint64_t *llv);

/* longest, largelcpvalues, maxbranchdepth, masked lcp sum */
void ora_esastats_compute(const uint8_t *enc, uint64_t n, const uint64_t *sa,
                          const uint64_t *lcp, uint32_t prefixlength,
                          ora_esastats *st);

/* linear-time check that sa is THE suffix array of enc under the ordering
   rule (restates the idea of sfx-lwcheck.c:181): returns 0 if ok, else a
   1-based code of the failed condition; *where = offending index */
int ora_check_suffix_array(const uint8_t *enc, uint64_t n, const uint64_t *sa,
                           uint64_t *where);

/* sections of INDEX.bck (uint32 variant) for prefix length k: leftborder has
   sigma^k + 1 entries, countspecialcodes sigma^(k-1), distpfxidx
   sigma + ... + sigma^(k-2) (src/match/bcktab.c:519-558) */
void ora_bcktab(const uint8_t *enc, uint64_t n, uint32_t sigma, uint32_t k,
                uint32_t *leftborder, uint32_t *countspecialcodes,
                uint32_t *distpfxidx);

/* write NAME.prj exactly as sfx-outprj.c:38-83 does */
int ora_write_prj(const char *path, const ora_seqstats *ss,
                  const ora_esastats *es, int with_lcp, int readmode,
                  int mirrored);

#ifdef __cplusplus
}
#endif
#endif
