/* host_internal.h -- shared between the files of the host layer, not part of
   the public interface */
#ifndef GTAMD_HOST_INTERNAL_H
#define GTAMD_HOST_INTERNAL_H
#include "gtamd_host.h"
#include "gtamd_encode.h"

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
  uint64_t sp_tab[3], wc_tab[3];  /* stored ranges per table width */
} gtamd_seqanalysis;

void gtamd_analyse_sequence(const uint8_t *enc, uint64_t n, uint32_t numofchars,
                            gtamd_seqanalysis *an);

/* the sections of INDEX.esq / INDEX.ssp behind the header, whichever the access
   type needs (gtamd_esq_needs): packed symbols, special bits, maximal
   wildcard runs, separator positions */
typedef struct {
  const uint64_t *twobit, *specialbits;
  const uint8_t *packed;
  const uint8_t *plain;           /* direct access: the symbols themselves */
  const uint64_t *wc_start, *wc_len;
  uint64_t wc_runs;
  const uint64_t *seppos;
} gtamd_esq_sections;

void gtamd_esq_needs(const gtamd_seqanalysis *an, int write_ssp, int *twobit,
                     int *specialbits, int *packed, int *wildcardruns,
                     int *separators);   /* direct access: none of the first three */
/* code that stands in for specials in the two-bit encoding of the equal-length
   and table access types (src/core/encseq.c:4468-4485) */
unsigned gtamd_least_probable(const gtamd_seqanalysis *an);
int gtamd_write_esq_sections(const char *indexname, const char *const *paths,
                             size_t numfiles, const gtamd_alphabet *a,
                             const gtamd_seqanalysis *an, const gtamd_encinfo *info,
                             int write_ssp, const gtamd_esq_sections *sec,
                             char *err, size_t errlen);

/* FASTQ: what the file length table needs of every record, and the table as the
   reference's FASTQ reader books it (encseq_host.c) */
typedef struct { uint64_t seqlen, desclen; size_t file; } gtamd_fastq_record;
void gtamd_fastq_filelengths(const gtamd_fastq_record *rec, size_t nrec, size_t lastfile,
                             gtamd_filelength *tab);

/* one input file, whole, decompressed when its name ends in ".gz"; 0 or a code
   for gtamd_read_input_error */
int gtamd_read_input_file(const char *path, uint8_t **data, uint64_t *len);
void gtamd_read_input_error(int code, const char *path, char *err, size_t errlen);

/* symbol map of the DNA / protein alphabet; 253 marks undefined characters */
void gtamd_symbolmap(uint8_t map[256], int protein);

/* stored-range counts and access type from the range counts per table width;
   needs ss.totallength/numofsequences/numofchars and equallength set */
int gtamd_choose_access_type(gtamd_seqanalysis *an, const uint64_t sp_tab[3],
                             const uint64_t wc_tab[3], int forced_sat);
/* -sat NAME -> access type number; the alphabet-dependent checks and messages
   of src/core/encseq.c:797-807, encseq_access_type.c:163-221 */
int gtamd_parse_sat(const char *name, int notdna, int *sat, char *err, size_t errlen);
/* apply a forced access type (-sat) to a finished analysis; -1 with the
   reference's message when "eqlen" does not fit the sequences */
int gtamd_force_sat(gtamd_seqanalysis *an, const char *satname, int notdna, char *err,
                    size_t errlen);
/* the same analysis from the device encoder's summary */
void gtamd_analysis_from_summary(const gtamd_encode_summary *s, uint32_t numofchars,
                                 gtamd_seqanalysis *an);

/* bytes of a range table of the given width (0 uchar, 1 ushort, 2 uint32),
   src/core/encseq.c:924-949 */
uint64_t gtamd_swtable_bytes(int width_kind, int withrangelengths, uint64_t n,
                             uint64_t items);
#endif
