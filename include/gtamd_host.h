/*
  gtamd_host.h -- host-side C layer around the device engine: what the
  reference's driver (src/match/sfx-run.c:428-717) does before and after the
  Sfxiterator loop, reduced to the `gt suffixerator` surface of this path.

    gtamd_encode_files      GtEncseqEncoder read side: FASTA -> encoded symbols
                            (src/core/sequence_buffer_fasta.c:44-170,
                            src/core/alphabet.c:84-91,345-356,480-503)
    gtamd_sequence_stats    GtSpecialcharinfo (src/core/chardef.h:91-116,
                            src/core/encseq_charproc.gen, encseq.c:5061-5127)
    gtamd_write_prj         gt_outprjfile (src/match/sfx-outprj.c:38-118)
    gtamd_suffixerator      the tool function, GtToolfunc shape
                            (src/core/toolbox.h:32, src/tools/gt_suffixerator.c:22)

  Pure C (gcc); links against libgtamd_esa.so for the hot path.
*/
#ifndef GTAMD_HOST_H
#define GTAMD_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "gtamd_esa.h"
#include "gtamd_encode.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint64_t totallength, specialcharacters, specialranges, realspecialranges,
           lengthofspecialprefix, lengthofspecialsuffix, wildcards,
           wildcardranges, realwildcardranges, lengthofwildcardprefix,
           lengthofwildcardsuffix, numofsequences;
  uint32_t numofchars;
} gtamd_seqstats;

/* An alphabet: the built-in DNA and protein ones (src/core/alphabet.c:84-91,
   345-356, 480-503) or one read from a symbol map file (-smap; format and
   parser src/core/alphabet.c:118-330: one line per symbol class, optionally a
   blank and the character shown for the class, leading '#' lines are comments,
   the last line holds the wildcards).  numofchars <= 28 for the engine. */
typedef struct {
  uint8_t symbolmap[256];      /* code per input byte; 254 wildcard, 253 undefined */
  uint32_t numofchars;
  char characters[64];         /* character shown for every code */
  char wildcardshow;
  int alphatype;               /* 0 DNA, 1 protein, 2 from a symbol map */
  unsigned bitspersymbol;      /* of the bit-packed access type */
  char *alphadef;              /* symbol map text as INDEX.esq stores it, or NULL */
  uint64_t lengthofalphadef;
} gtamd_alphabet;

void gtamd_alphabet_standard(gtamd_alphabet *a, int protein);
int gtamd_alphabet_from_file(const char *path, gtamd_alphabet *a, char *err, size_t errlen);
int gtamd_alphabet_from_text(const char *text, uint64_t len, const char *mapfile,
                             gtamd_alphabet *a, char *err, size_t errlen);
void gtamd_alphabet_free(gtamd_alphabet *a);

/* Read one or more (multi-)FASTA files into one encoded sequence; consecutive
   sequences are joined by one separator, also across files.  protein != 0
   selects the protein alphabet.  *enc is malloc'ed.  Returns 0, or -1 with the
   reference's error text in err ("illegal character 'X': file \"f\", line 2",
   "file 'f' contains an empty sequence"). */
int gtamd_encode_files(const char *const *paths, size_t numfiles, int protein,
                       uint8_t **enc, uint64_t *n, char *err, size_t errlen);

/* The same, also returning the sequence descriptions (header lines without
   the leading '>' / '@'), NUL-separated in one malloc'ed block of *desclen
   bytes, one per sequence.  desc may be NULL. */
int gtamd_encode_files_desc(const char *const *paths, size_t numfiles,
                            int protein, uint8_t **enc, uint64_t *n,
                            char **desc, uint64_t *desclen, char *err,
                            size_t errlen);

/* What the encoder saw besides the symbols: how often every original input
   byte occurred in the sequences, and per input file the bytes read and the
   symbols (with separators) it contributed -- GtFilelengthvalues,
   src/core/filelengthvalues.h:22-26, filled in
   src/core/sequence_buffer_fasta.c:56-94 / sequence_buffer_fastq.c:113-188. */
typedef struct { uint64_t length, effectivelength; } gtamd_filelength;
typedef struct {
  uint64_t originaldistribution[256];
  gtamd_filelength *filelengthtab;      /* numfiles entries, malloc'ed */
  size_t numfiles;
  uint64_t exceptioncharacters,         /* -lossless: set by gtamd_write_ois, go */
           realexceptionranges;         /* into the header of INDEX.esq          */
} gtamd_encinfo;

int gtamd_encode_files_info(const char *const *paths, size_t numfiles,
                            int protein, uint8_t **enc, uint64_t *n,
                            char **desc, uint64_t *desclen,
                            gtamd_encinfo *info, char *err, size_t errlen);
void gtamd_encinfo_free(gtamd_encinfo *info);

/* -lossless (src/core/encseq_api.h:276-286): the reader also returns the
   original character of every symbol (0 for separators; *orig malloc'ed);
   gtamd_write_ois writes INDEX.ois -- most frequent original character per
   symbol class, the classes' character lists, and the bit-packed list of
   "exceptions" with their runs (src/core/encseq.c:1018-1078, 5275-5419) -- and
   stores the exception counts in *info for INDEX.esq; the MD5 sums are taken
   over the original characters (encseq_charproc.gen:30-36). */
int gtamd_encode_files_orig(const char *const *paths, size_t numfiles,
                            const gtamd_alphabet *a, uint8_t **enc, uint64_t *n,
                            uint8_t **orig, char **desc, uint64_t *desclen,
                            gtamd_encinfo *info, char *err, size_t errlen);
int gtamd_write_ois(const char *indexname, const uint8_t *enc, const uint8_t *orig,
                    uint64_t n, const gtamd_alphabet *a, gtamd_encinfo *info,
                    char *err, size_t errlen);
int gtamd_write_md5_orig(const char *indexname, const uint8_t *enc,
                         const uint8_t *orig, uint64_t n);

/* The functions of this header that take `int protein` have a twin ending in
   _alpha that takes any alphabet instead. */
int gtamd_encode_files_alpha(const char *const *paths, size_t numfiles,
                             const gtamd_alphabet *a, uint8_t **enc, uint64_t *n,
                             char **desc, uint64_t *desclen,
                             gtamd_encinfo *info, char *err, size_t errlen);
int gtamd_write_md5_alpha(const char *indexname, const uint8_t *enc, uint64_t n,
                          const gtamd_alphabet *a);
int gtamd_write_esq_alpha(const char *indexname, const char *const *paths,
                          size_t numfiles, const uint8_t *enc, uint64_t n,
                          const gtamd_alphabet *a, const gtamd_encinfo *info,
                          int write_ssp, const char *sat, gtamd_seqstats *ss,
                          char *err, size_t errlen);
int gtamd_read_esq_alpha(const char *indexname, uint8_t **enc, uint64_t *n,
                         gtamd_alphabet *a, gtamd_seqstats *ss, char *err,
                         size_t errlen);
int gtamd_device_encode_files_alpha(const char *const *paths, size_t numfiles,
                                    const gtamd_alphabet *a, gtamd_encoder **enc,
                                    char **desc, uint64_t *desclen,
                                    gtamd_encinfo *info, char *err, size_t errlen);
int gtamd_write_esq_device_alpha(const char *indexname, const char *const *paths,
                                 size_t numfiles, const gtamd_encoder *enc,
                                 const gtamd_alphabet *a, const gtamd_encinfo *info,
                                 int write_ssp, const char *sat, gtamd_seqstats *ss,
                                 char *err, size_t errlen);

/* INDEX.esq -- the encoded sequence in the reference's own on-disk format, so
   that an index written here can be mapped by GenomeTools' tools
   (gt_encseq_loader_load) -- and INDEX.ssp, the separator positions, when the
   reference would write it: header and sequence sections of
   src/core/encseq.c:1195-1402 with the access type the reference chooses
   (src/core/encseq_access_type.c:96-162): "equallength", "bit", "uchar",
   "ushort", "uint32" for DNA, "bytecompress" for protein.  paths are stored as
   given.  write_ssp mirrors the -ssp option.  0, or -1 with a message. */
int gtamd_write_esq(const char *indexname, const char *const *paths,
                    size_t numfiles, const uint8_t *enc, uint64_t n,
                    int protein, const gtamd_encinfo *info, int write_ssp,
                    char *err, size_t errlen);

/* The same two steps with the device encoder (include/gtamd_encode.h) for
   FASTA input and FASTQ input in the four-line form: the files are read whole and encoded on the GPU; *enc (destroy
   with gtamd_encoder_destroy) holds the symbols in HBM, ready for
   gtamd_esa_set_sequence_bytes(ctx, gtamd_encoder_device_symbols(*enc), n, 1);
   descriptions and file information come back as from gtamd_encode_files_info.
   gtamd_write_esq_device writes INDEX.esq/.ssp from sections packed on the
   device, byte-identical to gtamd_write_esq; *ss (may be NULL) receives the
   sequence statistics.  gtamd_device_encode_files returns
   GTAMD_DEVICE_DECLINED (err says why) for FASTQ the device reader does not
   take: gtamd_encode_files_info then reads it.  gtamd_input_is_fastq: 1 if a
   file starts with '@'. */
#define GTAMD_DEVICE_DECLINED (-2)
int gtamd_input_is_fastq(const char *const *paths, size_t numfiles);
int gtamd_device_encode_files(const char *const *paths, size_t numfiles,
                              int protein, gtamd_encoder **enc,
                              char **desc, uint64_t *desclen,
                              gtamd_encinfo *info, char *err, size_t errlen);
int gtamd_write_esq_device(const char *indexname, const char *const *paths,
                           size_t numfiles, const gtamd_encoder *enc,
                           int protein, const gtamd_encinfo *info, int write_ssp,
                           const char *sat, gtamd_seqstats *ss, char *err, size_t errlen);

/* The way back (option -ii, src/match/sfx-run.c:454-493 /
   gt_encseq_loader_load): the symbols of an existing INDEX.esq, written by
   GenomeTools or by gtamd_write_esq, for every access type ("direct",
   "bytecompress", "equallength", "bit", "uchar", "ushort", "uint32"; the last
   three also need INDEX.ssp when there is more than one sequence).  DNA and
   protein alphabets only.  *enc is malloc'ed; ss (may be NULL) receives the
   sequence statistics stored in the header. */
int gtamd_read_esq(const char *indexname, uint8_t **enc, uint64_t *n,
                   int *protein, gtamd_seqstats *ss, char *err, size_t errlen);

/* The same with the access type forced (-sat direct|bytecompress|eqlen|bit|
   uchar|ushort|uint32; NULL: the reference's choice) and the sequence
   statistics -- whose stored-range counts follow the forced table type --
   returned in *ss (may be NULL).  Errors of src/core/encseq.c:797-807 and
   src/core/encseq_access_type.c:163-221 with their wording. */
int gtamd_write_esq_sat(const char *indexname, const char *const *paths,
                        size_t numfiles, const uint8_t *enc, uint64_t n,
                        int protein, const gtamd_encinfo *info, int write_ssp,
                        const char *sat, gtamd_seqstats *ss, char *err, size_t errlen);

void gtamd_sequence_stats(const uint8_t *enc, uint64_t n, uint32_t numofchars,
                          gtamd_seqstats *st);

/* INDEX.des (descriptions, each followed by '\n', then the length of the
   longest one and ~0 as two 8-byte words) and INDEX.sds (8-byte end offset of
   every description but the last), src/core/encseq_charproc.gen:118-130,
   src/core/encseq.c:5613-5624.  INDEX.md5: per sequence the MD5 of its decoded
   upper-case symbols as 32 hex digits + NUL (encseq_charproc.gen:52-92). */
/* -clipdesc (src/core/desc_buffer.c:63-80): cut every description of the
   NUL-separated block at its first white space, in place */
void gtamd_clip_descriptions(char *desc, uint64_t *desclen);
int gtamd_write_des_sds(const char *indexname, const char *desc,
                        uint64_t desclen, int write_des, int write_sds);
int gtamd_write_md5(const char *indexname, const uint8_t *enc, uint64_t n,
                    int protein);

/* The sequence as the reference reads it with -dir fwd|rev|cpl|rcl (readmode
   0..3, src/core/readmode_api.h:24-27), in place; complement (3 - code) is
   defined for DNA only, specials are their own complement. */
void gtamd_apply_readmode(uint8_t *enc, uint64_t n, int readmode);

/* -mirrored (src/core/encseq_api.h:190-198, encseq_options.c): sequence +
   separator + its reverse complement, 2n+1 symbols, malloc'ed; and the
   statistics a mirrored GtEncseq reports, from those of the original
   (src/core/encseq.c:4960-5054). */
uint8_t *gtamd_mirror(const uint8_t *enc, uint64_t n);
void gtamd_seqstats_mirror(gtamd_seqstats *st, int last_symbol_is_wildcard);

/* INDEX.prj; with_lcp == 0 writes the zero LCP statistics the reference
   writes when -lcp was not requested (src/match/sfx-run.c:664-670) */
int gtamd_write_prj(const char *path, const gtamd_seqstats *ss,
                    const gtamd_esa_stats *es, int with_lcp, int readmode,
                    int mirrored);

/* `gt suffixerator` for the option subset of this path:
     -db FILE... | -ii INDEX  -indexname NAME  -dna | -protein
     -suf -lcp -bwt -bck  -suftabuint  -sat TYPE  -smap FILE  -lossless
     -pl [K]  -v  -dir fwd|rev|cpl|rcl  -mirrored  -clipdesc  and, accepted
     without effect on the tables (strategy knobs of the CPU algorithm),
     -parts N  -memlimit X  -dc V  -algbds A B C  -maxwidthrealmedian W
     -cmpcharbychar -dccheck -iterscan -kmerswithencseqreader -noshortreadsort
     -samplewithprefixlengthnull -storespecialcodes -withradixsort
     -showprogress -tis [yes|no];
     -plain -kys -lcpdist -compressedoutput -genomediff
     -sortmaxdepth -spmopt -swallow-tail -onlybucketinsertion change what is
     written and are refused ("option \"-X\" is not supported ...").
   -des -sds -md5 -ssp [yes|no] select the sequence-side files; INDEX.esq is
   always written, as the reference does.
   argv[0] is the tool name.  Returns 0, or -1 with the message in err (the
   caller prints "gt suffixerator: error: <err>" and exits 1, src/gt.c:48-52). */
int gtamd_suffixerator(int argc, const char **argv, char *err, size_t errlen);

/* `gt dev mergeesa -indexname OUT -ii INDEX1 INDEX2 ...` (tool function
   src/tools/gt_mergeesa.c:61, engine src/match/esa-merge.c:136-200, output
   src/match/test-mergeesa.c:110-190): OUT.suf / OUT.lcp / OUT.llv of the
   concatenation of the indexes' sequence sets -- byte for byte what
   `gt suffixerator` writes for all their files at once, which is what the
   reference's own test compares the merge with
   (testsuite/gt_mergeesa_include.rb:17-19).  Here the merge is a build on the
   device from the input indexes' INDEX.esq. */
int gtamd_mergeesa(int argc, const char **argv, char *err, size_t errlen);

/* `gt packedindex trsuftab [-bsize B] [-blbuck K] [-locfreq F] [-locbitmap
   [yes|no]] [-sprank [yes|no]] [-sprankilog I] [-v] INDEX` (tool function src/tools/gt_packedindex_trsuftab.c:44-79,
   construction src/match/eis-bwtseq-construct.c:64-92): INDEX.bdx, the
   block-compressed BWT of the packed index, from the project's INDEX.prj / .esq /
   .bwt / .suf -- byte for byte the reference's file; built on
   the device through include/gtamd_pck.h; with -ctxilog I also INDEX.<I>cxm. */
int gtamd_packedindex_trsuftab(int argc, const char **argv, char *err, size_t errlen);

/* `gt packedindex mkindex` (src/tools/gt_packedindex.c:33-36:
   gt_parseargsandcallsuffixerator(false, ...)): the command line of
   gtamd_suffixerator without the table switches, plus -bsize -blbuck -locfreq
   -locbitmap -sprank -sprankilog; writes the sequence-side files, INDEX.bdx as the reference's
   run_packedindexconstruction does (src/match/sfx-run.c:369-425: with sequence
   statistics, block size 3 for alphabets of more than 10 letters) and INDEX.prj
   (no suffixes written, no `longest`). */
int gtamd_packedindex_mkindex(int argc, const char **argv, char *err, size_t errlen);
/* `gt packedindex mkctxmap [-ctxilog I] INDEX` (src/tools/gt_packedindex_mkctxmap.c:40-139):
   INDEX.<I>cxm from INDEX.prj / INDEX.suf; mkindex and trsuftab take -ctxilog too */
int gtamd_packedindex_mkctxmap(int argc, const char **argv, char *err, size_t errlen);
int gtamd_write_prj_packedindex(const char *path, const gtamd_seqstats *ss,
                                uint32_t prefixlength, int readmode, int mirrored);

#ifdef __cplusplus
}
#endif
#endif
