/*
  gtamd_esa.h -- C ABI of the MI355X-native enhanced-suffix-array engine.

  Drop-in boundary for ONE path of GenomeTools: what `gt suffixerator
  -suf -lcp -bwt` computes between "encoded sequence in memory" and "tables
  written".  The reference has no plugin ABI; the seam this library replaces is
  the Sfxiterator library interface plus its two sinks:

    gt_Sfxiterator_new_withadditionalvalues   src/match/sfx-suffixer.h:48-60
    gt_Sfxiterator_next                       src/match/sfx-suffixer.h:62-64
    gt_Sfxiterator_longest                    src/match/sfx-suffixer.h:72
    gt_Sfxiterator_delete                     src/match/sfx-suffixer.h:35
    GtOutlcpinfo (LCP sink, .lcp/.llv, stats) src/match/sfx-lcpvalues.h:102-137
    bwttab2file (BWT sink)                    src/match/sfx-run.c:173-210
    gt_recommendedprefixlength (for .prj)     src/match/sfx-apfxlen.h:24-27

  The reference hands the suffix array out in memory-sized slices because it
  sorts bucket by bucket on the CPU; on a 288 GB device the whole table is
  resident, so the iterator collapses into one coarse call (gtamd_esa_run) on a
  context that owns the device workspace.  Conventions follow GtError: every
  function returns 0 on success and -1 on failure, with a message available
  from gtamd_esa_last_error() (the text a GtToolfunc shim would pass to
  gt_error_set).

  Plain C: only pointers, sizes and PODs cross this boundary.  There is no CPU
  fallback: without a HIP device every compute entry point fails with -1.
*/
#ifndef GTAMD_ESA_H
#define GTAMD_ESA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTAMD_WILDCARD   254u  /* src/core/chardef.h:33 */
#define GTAMD_SEPARATOR  255u  /* src/core/chardef.h:34 */
#define GTAMD_UNDEFBWT   254u  /* src/core/chardef.h:65 */
#define GTAMD_LCPOVERFLOW 255u /* src/match/lcpoverflow.h:24 */

/* which tables to produce: the -suf / -lcp / -bwt switches of
   src/match/index_options.c:298-338 */
#define GTAMD_WANT_SUF 1u
#define GTAMD_WANT_LCP 2u
#define GTAMD_WANT_BWT 4u
#define GTAMD_WANT_BCK 8u  /* -bck, src/match/index_options.c */

/* table selectors for gtamd_esa_table_* */
typedef enum {
  GTAMD_TAB_SUF = 0,  /* (n+1) x uint64, native endian  (.suf) */
  GTAMD_TAB_LCP = 1,  /* (n+1) x uint8                  (.lcp) */
  GTAMD_TAB_BWT = 2,  /* (n+1) x uint8                  (.bwt) */
  GTAMD_TAB_LLV = 3,  /* numlargelcp x (uint64 index, uint64 value) (.llv) */
  GTAMD_TAB_BCK = 4   /* uint32: the sections of .bck back to back, see
                         gtamd_esa_bck_layout */
} gtamd_table;

/* numbers the reference prints into .prj (src/match/sfx-outprj.c:38-83) that
   depend on the tables */
typedef struct {
  uint64_t totallength;                /* n */
  uint64_t numberofallsortedsuffixes;  /* n + 1 */
  uint64_t longest;                    /* index i with suf[i] == 0 */
  uint64_t largelcpvalues;             /* entries of .llv */
  uint64_t maxbranchdepth;             /* max lcp */
  uint64_t lcptabsum;                  /* masked sum; averagelcp = sum/(n+1) */
  uint32_t prefixlength;               /* gt_recommendedprefixlength(sigma,n) */
  uint32_t refine_rounds;              /* engine statistic: doubling rounds */
  uint64_t tied_suffixes;              /* engine statistic: suffixes that were
                                          not separated by the first sort */
  uint64_t pair_suffixes;              /* ... of these, in tie groups of two
                                          (settled by one text comparison) */
  uint64_t device_bytes;               /* device memory the context holds */
  /* MSD first sort (DNA builds from 2^25 entries): entries in runs too long for
     the LDS kernel (sorted in global memory by k_msd_big / the device-wide
     sort), and entries of runs the LDS kernel read but left to its radix
     fallback because one bin of its counting pass was crowded */
  uint64_t msd_big_entries;
  uint64_t msd_crowded_entries;
  /* rank table of the doubling rounds (single builds): entries of the table that
     were built -- the windows of positions the rounds can touch -- of n + 1; 0
     if no round was needed */
  uint64_t rank_entries_built;
} gtamd_esa_stats;

/* per-stage device time of the last run, measured with HIP events on the
   context's own stream (milliseconds) */
typedef struct {
  float total_ms;      /* text resident -> tables resident */
  float keygen_ms;
  float sort_ms;       /* all radix passes of the first sort */
  float finalize_ms;   /* SA widening + LCP/BWT emission + tie detection */
  float refine_ms;     /* prefix-doubling rounds incl. rank table build */
  float tie_fix_ms;    /* LCP/BWT of tied suffixes + .llv */
  /* the dominant kernel of the first sort, timed with its own event pairs:
     dominant_kernel 0 = k_rs_scatter (a radix pass of the LSD sort: 8 + 4 bytes
     read and written per pair), 1 = k_msd_local (last level of the MSD sort of
     a DNA whole-table build: sorts a run in LDS and writes the tables; 8 bytes
     read, 8 + 4 + 1 + 1 written per entry) */
  float scatter_ms;    /* sum over the launches of that kernel */
  uint32_t scatter_launches;
  uint64_t scatter_items;  /* pairs / entries per launch */
  /* part builds: host time spent inside the collective callbacks (waiting for
     the other parts included), their number, and the bytes this part sent */
  float comm_ms;
  uint32_t comm_calls;
  uint64_t comm_bytes;
  /* host time this run spent allocating device memory (the workspace is
     allocated by the first run that needs it: a cold run pays seconds for
     ~140 GB at 3 Gbp, the following runs on the context nothing) */
  float alloc_ms;
  uint32_t dominant_kernel;
  /* entries the dominant kernel reads / writes per launch (k_rs_scatter: both
     the pairs of the pass; k_msd_local: reads every run that fits its tile,
     writes the tables of the runs it sorts itself): the roofline's algorithmic
     bytes are 8 x read + 14.125 x written for k_msd_local, 12 x read + 12 x
     written for k_rs_scatter */
  uint64_t scatter_read_items;
  uint64_t scatter_written_items;
} gtamd_esa_timing;

typedef struct gtamd_esa_ctx gtamd_esa_ctx;

/* ---- library ---------------------------------------------------------- */
/* number of HIP devices visible (0 if none / no driver); never fails */
int gtamd_device_count(void);
/* message of the last failure on the calling thread */
const char *gtamd_esa_last_error(void);

/* arithmetic of src/match/sfx-apfxlen.c:83-109 (host only, no device) */
uint32_t gtamd_recommended_prefixlength(uint32_t numofchars, uint64_t n);
/* Host-only self test of the exception barrier every entry point of this
   library runs behind (C callers expect -1 + message, never a C++ exception
   across the ABI): sizes a host container with `host_bytes` bytes inside the
   barrier; 0 when the allocation succeeded, -1 (message set) when it threw. */
int gtamd_abi_selftest(uint64_t host_bytes);

/* ---- context ---------------------------------------------------------- */
/* Create an engine on HIP device `device` for sequences of up to max_n
   symbols over an alphabet of `numofchars` letters (4: 2-bit DNA path,
   <= 28: 5-bit path).  The device workspace is allocated by the first run that
   needs it (by the tables wanted and, in a part build, the slice size) and
   kept for the following runs; NULL on failure. */
gtamd_esa_ctx *gtamd_esa_create(int device, uint64_t max_n,
                                uint32_t numofchars);
void gtamd_esa_destroy(gtamd_esa_ctx *ctx);

/* Restrict the build to the part `part` of `numparts` (at most 128)
   equal-width (by suffix count) lexicographic ranges, the reference's -parts
   mechanism (src/match/sfx-partssuf.c:172-347) used here to shard over GPUs.
   Default 0 of 1.  Every part holds the whole packed sequence and builds the
   slice [table_offset, table_offset + table_entries) of each table.  The work
   of a part falls with the number of parts: a DNA part takes the suffixes of its
   key range from the text by a scan (two cheap passes over the packed text) and
   sorts them where they are -- no pair travels; a part over a larger alphabet
   makes the sort keys of ITS 1/R tile of the text and sends each (key, position)
   pair to the part that owns the key range.  The rank table of the
   prefix-doubling rounds is cut by text position (part t holds the ranks of the
   suffixes starting in tile t), so a round sends the tile owners the new ranks
   and this round's questions in one message and gets the answers in another --
   all through the callbacks of gtamd_esa_set_comm.  A part build addresses up to
   2^40 positions (a single build: 2^32 - 4096); one slice must stay below 2^32
   entries.  The statistics of a part cover its slice: the caller adds
   lcptabsum / largelcpvalues, takes the max of maxbranchdepth, and `longest`
   from the part whose slice holds suffix 0 (the others report 0). */
int gtamd_esa_set_part(gtamd_esa_ctx *ctx, uint32_t part, uint32_t numparts);

/* Collectives a part build needs; the caller supplies the transport (RCCL
   through torch.distributed in bench.py).  Both return 0 on success.
   allgather: every part contributes `bytes` bytes of HOST memory, `recv`
   (host) receives numparts x bytes in part order.
   alltoallv: DEVICE buffers of `elem_bytes`-sized elements; the block for part
   r has sendcounts[r] elements (the block for the caller itself included: it
   is copied), blocks are laid out in part order on both sides; recvcounts is
   known to the engine (it allgathers the count matrix).  `stream` is the
   engine's hipStream_t: the exchange must come after the work already queued
   on it, and work queued on it after the callback returns must see the
   received data -- enqueue the exchange on that stream (RCCL), or synchronise
   the stream, exchange, and return when the data has arrived.
   If a part fails (out of memory ...) it says so in its next allgather and
   gtamd_esa_run returns -1 on every part. */
typedef int (*gtamd_allgather_fn)(void *user, const void *send, void *recv,
                                  uint32_t bytes);
typedef int (*gtamd_alltoallv_fn)(void *user, const void *send,
                                  const uint64_t *sendcounts, void *recv,
                                  const uint64_t *recvcounts,
                                  uint32_t elem_bytes, void *stream);
int gtamd_esa_set_comm(gtamd_esa_ctx *ctx, gtamd_allgather_fn allgather,
                       gtamd_alltoallv_fn alltoallv, void *user);
/* What gtamd_esa_run calls when a part build fails on THIS part outside the
   agreed failures (a launch error, an exception between two collectives): the
   transport's way of telling the other parts not to wait for it -- they return -1
   from their next collective.  After gtamd_esa_set_comm (which clears it); the
   library's thread transport registers itself in gtamd_comm_attach.  Without it a
   failing part leaves the others blocked, as a failing MPI rank would. */
int gtamd_esa_set_comm_abort(gtamd_esa_ctx *ctx, void (*abort_fn)(void *user), void *user);

/* ---- transports that ship with the library ----------------------------- */
/* (genometools_amd/csrc/esa_comm.hip)  A C caller needs no Python and no torch
   for a build on several GPUs:

   THREADS -- one process, one host thread per part, as the reference runs its
   own parallel sorting (gt -j N, src/core/thread_api.h): create the transport
   once, one context per part on the device of your choice (the same device for
   all is fine), gtamd_comm_attach(comm, part, ctx, device) instead of
   gtamd_esa_set_part / gtamd_esa_set_comm, then gtamd_esa_run on every context
   from its own thread.  The alltoallv is a set of peer copies each part pulls
   onto its stream (hipMemcpyPeerAsync).  A thread that leaves early calls
   gtamd_comm_abort so that the others do not wait for it.

   RCCL -- one process per part: rank 0 makes the id (gtamd_comm_rccl_unique_id)
   and hands it to the others by whatever means the launcher has; every rank
   creates its communicator and attaches its context.  alltoallv = grouped
   ncclSend / ncclRecv on the engine's stream.  librccl is loaded when this
   transport is asked for, not before. */
typedef struct gtamd_comm gtamd_comm;
gtamd_comm *gtamd_comm_threads_create(uint32_t numparts);
int gtamd_comm_rccl_unique_id(uint8_t id[128]);
gtamd_comm *gtamd_comm_rccl_create(const uint8_t id[128], uint32_t rank,
                                   uint32_t numparts, int device);
int gtamd_comm_attach(gtamd_comm *comm, uint32_t part, gtamd_esa_ctx *ctx,
                      int device);
void gtamd_comm_abort(gtamd_comm *comm);
void gtamd_comm_destroy(gtamd_comm *comm);

/* Read mode of the sequence (the `readmode` argument of
   gt_Sfxiterator_new_withadditionalvalues, src/match/sfx-suffixer.h:48-60;
   GtReadmode of src/core/readmode.h: 0 forward, 1 reverse, 2 complement,
   3 reverse complement; complementing is defined for DNA only).  Applied while
   the next gtamd_esa_set_sequence_bytes packs the sequence: the tables are
   those of the sequence read that way. */
int gtamd_esa_set_readmode(gtamd_esa_ctx *ctx, int readmode);

/* Prefix length to report in .prj and to mask averagelcp with (option -pl K of
   src/match/index_options.c:363; 0 = gt_recommendedprefixlength).  The device
   algorithm itself does not bucket by it. */
int gtamd_esa_set_prefixlength(gtamd_esa_ctx *ctx, uint32_t prefixlength);

/* ---- input: the encoded sequence (GtEncseq read side) ------------------ */
/* One byte per symbol as the reference's encoder delivers them
   (src/core/encseq.c:249 gt_encseq_get_encoded_char): 0..sigma-1 letters,
   254 wildcard, 255 separator.  `enc` may be a host or a device pointer
   (is_device != 0).  The engine packs it on the device into its resident
   form: 2-bit words, 32 symbols per uint64, first symbol in the two most
   significant bits as in GtTwobitencoding (src/core/intbits.h:78-83), plus a
   bitmap of special positions (the reference's GT_ACCESS_TYPE_BITACCESS
   layout, src/core/encseq.c:2822-2835); 5-bit symbols for larger alphabets. */
int gtamd_esa_set_sequence_bytes(gtamd_esa_ctx *ctx, const uint8_t *enc,
                                 uint64_t n, int is_device);

/* Already packed input, device resident: `twobit` = ceil(n/32) words in the
   GtTwobitencoding layout where a special position holds 0 (wildcard) or 1
   (separator), `specialbits` = ceil((n+1)/64) words, bit (p & 63) of word
   p >> 6 set iff position p is special; bit n (the virtual end) must be set.
   The engine reads the buffers in place; they must outlive the run. */
int gtamd_esa_set_sequence_packed(gtamd_esa_ctx *ctx, const uint64_t *twobit,
                                  const uint64_t *specialbits, uint64_t n);

/* ---- the hot path ------------------------------------------------------ */
/* Build the requested tables (GTAMD_WANT_* mask) from the resident sequence;
   synchronous: returns when the tables are resident in device memory. */
int gtamd_esa_run(gtamd_esa_ctx *ctx, uint32_t want);

/* ---- output ------------------------------------------------------------ */
/* number of entries of a table after a run (n+1, or the .llv pair count; for
   a part build: the entries of this part's slice) and the slice's offset in
   the whole table */
uint64_t gtamd_esa_table_entries(const gtamd_esa_ctx *ctx, gtamd_table which);
uint64_t gtamd_esa_table_offset(const gtamd_esa_ctx *ctx);
/* The bucket table of the run's prefixlength k (GTAMD_WANT_BCK; GtBcktab,
   src/match/bcktab.c:55-81, file layout :519-558): GTAMD_TAB_BCK holds, as
   uint32 and back to back,
     leftborder[numofallcodes + 1]   first index of every bucket of k-mer codes
                                     (base sigma, first symbol most significant);
                                     suffixes with fewer than k letters before a
                                     special sit at the end of the bucket of
                                     their prefix padded with the largest letter;
                                     the last entry is the number of suffixes
                                     that do not start with a special
     countspecialcodes[sigma^(k-1)]  such short suffixes per padded (k-1)-prefix
     distpfxidx[sigma + ... + sigma^(k-2)]  suffixes with exactly l = 1..k-2
                                     letters before a special, per l-letter code
   (n + 1 <= UINT32_MAX here, so the reference writes uint32 too). */
int gtamd_esa_bck_layout(const gtamd_esa_ctx *ctx, uint64_t *numofallcodes,
                         uint64_t *numofspecialcodes,
                         uint64_t *numofdistpfxidxcounters);
/* device pointer of a table (valid until the next run / destroy) */
const void *gtamd_esa_table_device(const gtamd_esa_ctx *ctx, gtamd_table which);
/* copy entries [first, first+count) of a table to host memory */
int gtamd_esa_table_copy(gtamd_esa_ctx *ctx, gtamd_table which, void *dst,
                         uint64_t first, uint64_t count);
int gtamd_esa_get_stats(const gtamd_esa_ctx *ctx, gtamd_esa_stats *st);
int gtamd_esa_get_timing(const gtamd_esa_ctx *ctx, gtamd_esa_timing *tm);

/* ---- one-shot convenience (host buffers in, host buffers out) ---------- */
/* suf/lcp/bwt may be NULL when not wanted; llv receives up to llv_capacity
   pairs (2 x uint64 each); *llv_pairs gets the real count. */
int gtamd_esa_build(const uint8_t *enc, uint64_t n, uint32_t numofchars,
                    uint32_t want, uint64_t *suf, uint8_t *lcp, uint8_t *bwt,
                    uint64_t *llv, uint64_t llv_capacity, uint64_t *llv_pairs,
                    gtamd_esa_stats *st);

/* ---- synthetic sequences for benchmarks and parity tests --------------- */
/* Fill a device byte buffer with the synthetic sequence `model` of length n
   (see genometools_amd/synth.py for the definitions, which are pure functions
   of (model, seed, n, position)).  Returns -1 for an unknown model. */
int gtamd_synth_bytes(int device, int model, uint64_t seed, uint64_t n,
                      uint8_t *dst_device);

#ifdef __cplusplus
}
#endif
#endif
