/* host_internal.h -- shared between the files of the host layer, not part of
   the public interface */
#ifndef GTAMD_HOST_INTERNAL_H
#define GTAMD_HOST_INTERNAL_H
#include "gtamd_host.h"

/* GtEncseqAccessType, src/core/encseq_access_type.h:24-34 */
enum {
  GTAMD_SAT_DIRECTACCESS = 0, GTAMD_SAT_BYTECOMPRESS, GTAMD_SAT_EQUALLENGTH,
  GTAMD_SAT_BITACCESS, GTAMD_SAT_UCHARTABLES, GTAMD_SAT_USHORTTABLES,
  GTAMD_SAT_UINT32TABLES
};

typedef struct {
  gtamd_seqstats ss;
  uint64_t lengthoflongestnonspecial, minseqlen, maxseqlen;
  uint64_t chardist[32];          /* occurrences of every non-special code */
  int equallength;                /* all sequences equally long, no wildcard */
  uint64_t equallength_value;
  int sat;                        /* access type the reference would choose */
  uint64_t sat_wildcardranges;    /* stored wildcard ranges for that type */
} gtamd_seqanalysis;

void gtamd_analyse_sequence(const uint8_t *enc, uint64_t n, uint32_t numofchars,
                            gtamd_seqanalysis *an);

/* symbol map of the DNA / protein alphabet; 253 marks undefined characters */
void gtamd_symbolmap(uint8_t map[256], int protein);

/* bytes of a range table of the given width (0 uchar, 1 ushort, 2 uint32),
   src/core/encseq.c:924-949 */
uint64_t gtamd_swtable_bytes(int width_kind, int withrangelengths, uint64_t n,
                             uint64_t items);
#endif
