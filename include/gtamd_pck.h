/*
  gtamd_pck.h -- C ABI of the packed-index builder ("consumers
  on device: packed index / FM construction from the BWT").

  What it replaces in the reference: the construction of INDEX.bdx, the
  block-composition compressed BWT sequence of `gt packedindex`, from the
  tables of a suffix-array project -- `gt packedindex trsuftab INDEX`:

    gt_packedindex_trsuftab            src/tools/gt_packedindex_trsuftab.c:44-79
    gt_trSuftab2BWTSeq                 src/match/eis-bwtseq-construct.c:64-92
    gt_createBWTSeqGeneric             src/match/eis-bwtseq-extinfo.c:558-676
      addLocateInfo (locate marks)     src/match/eis-bwtseq-extinfo.c:384-541
    gt_newGenBlockEncIdxSeq            src/match/eis-blockcomp.c:304-655
      gt_block2IndexPair               src/match/eis-seqblocktranslate.c:436-540
      updateIdxOutput / writeIdxHeader src/match/eis-blockcomp.c:1807-2094
      gt_SRLSaveToStream               src/match/eis-seqranges.c:459-468
    option defaults                    src/match/eis-bwtseq-param.c:25-103,
                                       src/match/eis-blockcomp-param.c:21-36

  The reference streams BWT symbols and suffix-array entries through a
  single-threaded encoder; here the tables are already resident in HBM
  (gtamd_esa_run), and the whole file image is assembled on the device: every
  bucket of blockSize x bucketBlocks positions is independent once the prefix
  sums of the symbol counts and of the variable-width bits are known.  The
  image is byte-identical to the reference's file (also where the reference's
  staging buffers leave stale bits in the last bucket).

  Covered: block encoding, locate information as counts or as bitmap
  (-locfreq, -locbitmap), none (-locfreq 0), reversibly sorted specials
  (-sprank: the ranks of the text's specials in the var parts and the sort-mode
  extension header), and the context map of -ctxilog / `gt packedindex mkctxmap`
  (INDEX.<I>cxm, gtamd_pck_ctxmap_*).

  Conventions as in gtamd_esa.h: 0 / -1, message from gtamd_esa_last_error().
  Plain C; no CPU fallback.
*/
#ifndef GTAMD_PCK_H
#define GTAMD_PCK_H

#include <stddef.h>
#include <stdint.h>
#include "gtamd_esa.h"

#ifdef __cplusplus
extern "C" {
#endif

/* enum BWTFeatures, src/match/eis-bwtseq-param.h:78-94 */
#define GTAMD_PCK_LOCATE_BITMAP 1
#define GTAMD_PCK_LOCATE_COUNT  2
#define GTAMD_PCK_REVERSIBLY_SORTED 4  /* -sprank / -sprankilog */

typedef struct {
  uint32_t block_size;       /* -bsize   (default 8), 1..16 */
  uint32_t bucket_blocks;    /* -blbuck  (default 8); block_size * bucket_blocks <= 16384 */
  uint32_t locate_interval;  /* -locfreq (default 16), 0 = no locate information */
  int32_t feature_toggles;   /* GTAMD_PCK_LOCATE_*; gtamd_pck_default_toggles(), | GTAMD_PCK_REVERSIBLY_SORTED
                                for -sprank (gt_computePackedIndexDefaults,
                                src/match/eis-bwtseq-param.c:98-100) */
  int32_t with_statistics;   /* 0: the file `gt packedindex trsuftab INDEX` writes (tables
                                read back from files: the reference has no sequence
                                statistics there), 1: the file `gt packedindex mkindex`
                                writes (BWT straight from the suffixerator, with
                                statistics: run_packedindexconstruction,
                                src/match/sfx-run.c:369-425; they narrow the occurrence
                                counters, src/match/eis-blockcomp.c:385-437, and one size
                                bound, src/match/eis-bwtseq-extinfo.c:302-314) */
} gtamd_pck_params;

/* layout of the image, as the header fields of INDEX.bdx report it, and the
   device time of the build */
typedef struct {
  uint64_t file_bytes;       /* size of INDEX.bdx */
  uint64_t cw_data_pos;      /* constant-width records of the buckets start here */
  uint64_t var_data_pos;     /* VOFF: variable-width part */
  uint64_t range_enc_pos;    /* ROFF: region list of the special symbols */
  uint64_t num_buckets;
  uint64_t num_regions;      /* incl. the terminator region */
  uint64_t var_bits;         /* bits of the variable-width part */
  uint32_t cw_bits;          /* bits of one constant-width record */
  float build_ms;            /* device time: counting pass, scans, emission */
} gtamd_pck_info;

typedef struct gtamd_pck gtamd_pck;

/* the feature toggles `gt packedindex` derives from its options
   (gt_computePackedIndexDefaults, src/match/eis-bwtseq-param.c:89-103):
   locbitmap < 0 = option -locbitmap not given */
int gtamd_pck_default_toggles(uint32_t block_size, uint32_t bucket_blocks,
                              uint32_t locate_interval, int locbitmap);

/* a builder on HIP device `device`; NULL on failure.  A builder keeps its device
   buffers between builds (image, block table, scratch); one thread at a time
   per builder. */
gtamd_pck *gtamd_pck_create(int device);
void gtamd_pck_destroy(gtamd_pck *pck);

/* Build the image of INDEX.bdx from device-resident tables of a project with
   total_len = n + 1 entries: bwt (uint8, the .bwt table: letters, 254 wildcard
   and for the suffix 0, 255 separator), suf (uint64, the .suf table; may be
   NULL when locate_interval is 0), `numofchars` letters, `longest` = the index
   of suffix 0 (rot0Pos of the locate header).  Synchronous. */
int gtamd_pck_build(gtamd_pck *pck, const uint8_t *bwt_device,
                    const uint64_t *suf_device, uint64_t total_len,
                    uint32_t numofchars, uint64_t longest,
                    const gtamd_pck_params *params);

/* the same from an engine context whose last run produced GTAMD_WANT_SUF |
   GTAMD_WANT_BWT (whole-table build) */
int gtamd_pck_build_from_esa(gtamd_pck *pck, const gtamd_esa_ctx *esa,
                             const gtamd_pck_params *params);

/* the same from tables in HOST memory (read back from INDEX.bwt / INDEX.suf, as
   `gt packedindex trsuftab` does): uploaded, then built as above */
int gtamd_pck_build_host(gtamd_pck *pck, const uint8_t *bwt_host,
                         const uint64_t *suf_host, uint64_t total_len,
                         uint32_t numofchars, uint64_t longest,
                         const gtamd_pck_params *params);

int gtamd_pck_get_info(const gtamd_pck *pck, gtamd_pck_info *info);
/* device pointer of the image (file_bytes bytes; valid until the next build /
   destroy) */
const void *gtamd_pck_image_device(const gtamd_pck *pck);
/* copy bytes [offset, offset + count) of the image to host memory */
int gtamd_pck_image_copy(gtamd_pck *pck, void *dst, uint64_t offset, uint64_t count);

/* ---- the context map (-ctxilog I of mkindex / trsuftab, `gt packedindex
   mkctxmap`; src/match/eis-bwtseq-context.c:36-300): INDEX.<I>cxm, for every
   2^I-th text position the row of the suffix that follows it -- what lets the
   reference regenerate any stretch of the text from the index.  ilog < 0: the
   automatic interval (log of the log of the length); *ilog_used (may be NULL)
   gets the interval of the file name.  Needs only the suffix array. */
int gtamd_pck_ctxmap_build(gtamd_pck *pck, const uint64_t *suf_device, uint64_t total_len,
                           int ilog, int *ilog_used);
int gtamd_pck_ctxmap_build_from_esa(gtamd_pck *pck, const gtamd_esa_ctx *esa, int ilog,
                                    int *ilog_used);
int gtamd_pck_ctxmap_build_host(gtamd_pck *pck, const uint64_t *suf_host, uint64_t total_len,
                                int ilog, int *ilog_used);
uint64_t gtamd_pck_ctxmap_bytes(const gtamd_pck *pck);
int gtamd_pck_ctxmap_copy(gtamd_pck *pck, void *dst, uint64_t offset, uint64_t count);

#ifdef __cplusplus
}
#endif
#endif
