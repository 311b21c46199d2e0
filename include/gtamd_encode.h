/*
  gtamd_encode.h -- C ABI of the device-side sequence encoder: FASTA bytes in,
  encoded symbols (0..sigma-1, 254 wildcard, 255 separator) resident in HBM
  out, together with every number the reference's encoder derives while it
  reads the input.  It replaces the read side of gt_encseq_encoder_encode:

    gt_sequence_buffer_fasta_advance    src/core/sequence_buffer_fasta.c:41-171
      (the two-state reader: '>' opens a description anywhere outside one,
       '\n' closes it, white space is skipped, consecutive sequences -- also
       across files -- are joined by one separator)
    process_char                        src/core/sequence_buffer_inline.h:26-58
      (symbol map, "illegal character" error, character distribution)
    gt_inputfiles2sequencekeyvalues     src/core/encseq.c:5421-5673 with
      encseq_charproc.gen (special/wildcard ranges, prefix/suffix lengths,
       longest non-special stretch, min/max sequence length, equal-length
       test, file length table, distribution of the original characters)

  The reference reads the input byte by byte, twice; here one file is one
  device buffer and every step is a tile-parallel kernel (tile-local state
  machine + scans over tile summaries).  FASTQ input (a file that starts with
  '@'; src/core/seq_iterator_fastq.c:96-305) is encoded on the device in its
  STRICT FOUR-LINE form -- name, symbols, '+' line, qualities, one line each,
  '\n' line ends, no blanks -- where the line number alone says what a byte is;
  everything else about FASTQ (sequences over several lines, a missing last
  newline, malformed blocks, FASTA and FASTQ mixed) the device reader DECLINES:
  gtamd_encoder_finish fails, gtamd_encoder_declined() says why, and the caller
  hands the input to the host reader (include/gtamd_host.h), which has the
  reference's behaviour and messages for all of it.

  Conventions as in gtamd_esa.h: 0 / -1, message from gtamd_esa_last_error(),
  worded like the reference's ("illegal character 'X': file \"f\", line 3",
  "file 'f' contains an empty sequence", "no sequences in multiple fasta
  file(s) f ...").  No CPU fallback.
*/
#ifndef GTAMD_ENCODE_H
#define GTAMD_ENCODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gtamd_encoder gtamd_encoder;

/* what gt_inputfiles2sequencekeyvalues returns besides the symbols; the three
   "tab" entries are the stored-range counts for 8-, 16- and 32-bit range
   tables (currentspecialrangevalue, src/core/encseq.c:5061-5074) */
typedef struct {
  uint64_t totallength, numofsequences;
  uint64_t specialcharacters, realspecialranges, specialrangestab[3],
           lengthofspecialprefix, lengthofspecialsuffix;
  uint64_t wildcards, realwildcardranges, wildcardrangestab[3],
           lengthofwildcardprefix, lengthofwildcardsuffix;
  uint64_t lengthoflongestnonspecial, minseqlen, maxseqlen;
  uint64_t equallength;                  /* 1: all sequences equally long and
                                            without wildcard */
  uint64_t characterdistribution[32];    /* per symbol code */
  uint64_t originaldistribution[256];    /* per input byte, sequences only */
} gtamd_encode_summary;

gtamd_encoder *gtamd_encoder_create(int device, int protein);
/* any alphabet (-smap): code 0..numofchars-1 or 254 (wildcard) per input byte,
   everything else is an illegal character; bitspersymbol is that of the
   bit-packed access type (src/core/alphabet.c:300-305) */
gtamd_encoder *gtamd_encoder_create_map(int device, const uint8_t *symbolmap,
                                        uint32_t numofchars, unsigned bitspersymbol);
void gtamd_encoder_destroy(gtamd_encoder *enc);

/* One input file, in -db order: its name (for messages) and its bytes in host
   memory, which must stay valid until gtamd_encoder_finish returns. */
int gtamd_encoder_add_file(gtamd_encoder *enc, const char *name,
                           const uint8_t *bytes, uint64_t length);

/* Encode all files added so far into one sequence. */
int gtamd_encoder_finish(gtamd_encoder *enc);
/* 1 if the last gtamd_encoder_finish failed because the input is FASTQ the
   device reader does not take (see above): not an error of the input, the host
   reader is to read it. */
int gtamd_encoder_declined(const gtamd_encoder *enc);
/* FASTQ input: per record, in input order, the file it came from and the
   lengths of its sequence and its description -- what the reference's FASTQ
   reader books into the file length table per buffer fill
   (src/core/sequence_buffer_fastq.c:42-191).  0 records for FASTA input. */
uint64_t gtamd_encoder_num_fastq_records(const gtamd_encoder *enc);
int gtamd_encoder_get_fastq_records(const gtamd_encoder *enc, uint32_t *file,
                                    uint32_t *seqlen, uint32_t *desclen,
                                    uint64_t capacity);

/* Instead of add_file/finish: n already encoded symbols from host memory (the
   host reader's output, e.g. of FASTQ input) become the encoder's sequence;
   summary (without the original-character histogram) and the INDEX.esq
   sections are then available as after gtamd_encoder_finish. */
int gtamd_encoder_set_symbols(gtamd_encoder *enc, const uint8_t *symbols, uint64_t n);

uint64_t gtamd_encoder_length(const gtamd_encoder *enc);
/* the symbols in device memory (valid until destroy), e.g. for
   gtamd_esa_set_sequence_bytes(ctx, ptr, n, 1) */
const uint8_t *gtamd_encoder_device_symbols(const gtamd_encoder *enc);
int gtamd_encoder_copy_symbols(const gtamd_encoder *enc, uint8_t *dst,
                               uint64_t first, uint64_t count);
int gtamd_encoder_get_summary(const gtamd_encoder *enc, gtamd_encode_summary *s);

/* GtFilelengthvalues of input file `file` */
int gtamd_encoder_file_lengths(const gtamd_encoder *enc, size_t file,
                               uint64_t *length, uint64_t *effectivelength);

/* Descriptions: one per sequence, as byte ranges [start, end) of the file they
   came from (start is behind the '>', end is the terminating newline or the
   end of the file; a carriage return inside is the caller's to drop, as the
   reference does).  The arrays hold `capacity` entries each;
   gtamd_encoder_num_descriptions() are written, -1 if that is more. */
uint64_t gtamd_encoder_num_descriptions(const gtamd_encoder *enc);
int gtamd_encoder_get_descriptions(const gtamd_encoder *enc, uint32_t *file,
                                   uint64_t *start, uint64_t *end,
                                   uint64_t capacity);

/* The sections of INDEX.esq behind its header, built from the symbols in HBM
   and delivered to host memory (layouts: src/core/encseq.c:85-99, 2594-2607,
   2771-2835 two-bit encoding and special bits; :2324-2447 bit packing):
     pack_twobit        2 + (n-1)/32 words (2 if n < 32), 32 symbols per word,
                        first symbol in the top bits; specials are stored as
                        wildcard 0 / separator 1 (bitaccess != 0) or as
                        fillcode (the least frequent letter)
     pack_specialbits   1 + (n+63)/64 words, one bit per position, first in the
                        top bit; the 64 positions behind the sequence are set
     pack_bytecompress  (bits*n+7)/8 bytes, bits per symbol as given at
                        creation (5 protein, 3 DNA), wildcard = sigma,
                        separator = sigma + 1
     wildcard_runs      the realwildcardranges maximal runs (start, length)
     separators         the numofsequences - 1 separator positions
   `capacity` is the size of the caller's array(s) in entries (words, bytes,
   runs, positions): nothing is written and -1 returned if the section is
   larger. */
int gtamd_encoder_pack_twobit(const gtamd_encoder *enc, int bitaccess,
                              unsigned fillcode, uint64_t *words,
                              uint64_t capacity);
int gtamd_encoder_pack_specialbits(const gtamd_encoder *enc, uint64_t *words,
                                   uint64_t capacity);
int gtamd_encoder_pack_bytecompress(const gtamd_encoder *enc, uint8_t *bytes,
                                    uint64_t capacity);
int gtamd_encoder_get_wildcard_runs(const gtamd_encoder *enc, uint64_t *start,
                                    uint64_t *length, uint64_t capacity);
int gtamd_encoder_get_separators(const gtamd_encoder *enc, uint64_t *pos,
                                 uint64_t capacity);

/* device time of the last finish (ms, HIP events) and the bytes it read */
int gtamd_encoder_get_timing(const gtamd_encoder *enc, float *total_ms,
                             float *parse_ms, float *stats_ms,
                             uint64_t *input_bytes);

#ifdef __cplusplus
}
#endif
#endif
