/* suffixerator_tool.c -- `gt suffixerator` for the option subset of this path,
   in the shape of a GtToolfunc (src/core/toolbox.h:32): parse, encode, build
   on the device through the C ABI, write INDEX.suf/.lcp/.llv/.bwt/.prj.
   Option names and defaults follow src/core/encseq_options.c:181-290 and
   src/match/index_options.c:298-515; file suffixes src/match/esa-fileend.h. */
#include "gtamd_host.h"
#include "gtamd_pck.h"
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <time.h>
#include <pthread.h>

#define MAXDB 1024

static int fail(char *err, size_t errlen, const char *msg, const char *arg)
{
  snprintf(err, errlen, msg, arg);
  return -1;
}

static int failf(char *err, size_t errlen, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err, errlen, fmt, ap);
  va_end(ap);
  return -1;
}

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* INDEX.bck: the three uint32 sections, each padded to 8 bytes
   (src/match/bcktab.c:519-565, src/core/mapspec.c:350-457) */
static int write_bcktab(gtamd_esa_ctx *ctx, const char *index, char *err, size_t errlen)
{
  static const uint8_t zero[8] = {0};
  char path[4096];
  uint64_t sec[3], first = 0;
  FILE *fp;
  if (gtamd_esa_bck_layout(ctx, &sec[0], &sec[1], &sec[2]) != 0) {
    snprintf(err, errlen, "%s", gtamd_esa_last_error());
    return -1;
  }
  sec[0] += 1;
  snprintf(path, sizeof path, "%s.bck", index);
  if ((fp = fopen(path, "wb")) == NULL)
    return fail(err, errlen, "cannot open file '%s' for writing", path);
  for (int k = 0; k < 3; k++) {
    uint32_t *buf;
    if (sec[k] == 0) continue;
    if ((buf = malloc(4 * sec[k])) == NULL ||
        gtamd_esa_table_copy(ctx, GTAMD_TAB_BCK, buf, first, sec[k]) != 0 ||
        fwrite(buf, 4, sec[k], fp) != sec[k] ||
        ((4 * sec[k]) % 8 != 0 && fwrite(zero, 1, 4, fp) != 4)) {
      free(buf); fclose(fp);
      return fail(err, errlen, "cannot write file '%s'", path);
    }
    free(buf);
    first += sec[k];
  }
  return fclose(fp) == 0 ? 0 : fail(err, errlen, "cannot close file '%s'", path);
}

/* entrysize 4 for the suffix table: -suftabuint, 32-bit entries */
/* a table leaves the device in 64 MiB pieces; a second thread writes one piece
   to the file while the next one is copied (two staging buffers) */
typedef struct {
  FILE *fp;
  void *buf[2];
  size_t bytes[2];
  int full[2], done, failed;
  pthread_mutex_t mu;
  pthread_cond_t cv;
} table_writer;

static void *table_writer_main(void *arg)
{
  table_writer *w = (table_writer *) arg;
  for (int i = 0;; i ^= 1) {
    pthread_mutex_lock(&w->mu);
    while (!w->full[i] && !w->done) pthread_cond_wait(&w->cv, &w->mu);
    if (!w->full[i]) { pthread_mutex_unlock(&w->mu); return NULL; }
    pthread_mutex_unlock(&w->mu);
    if (!w->failed && fwrite(w->buf[i], 1, w->bytes[i], w->fp) != w->bytes[i]) w->failed = 1;
    pthread_mutex_lock(&w->mu);
    w->full[i] = 0;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
  }
}

/* the table `which` of a build -- of its parts in part order: the slices of a
   build in parts tile the table -- through two staging buffers to the file */
static int write_table(gtamd_esa_ctx *const *ctxs, uint32_t nctx, gtamd_table which, const char *index,
                       const char *suffix, size_t entrysize, char *err, size_t errlen)
{
  const int narrow = which == GTAMD_TAB_SUF && entrysize == 4;
  char path[4096];
  /* 64 MiB staging buffers (GTAMD_TABLE_CHUNK=BYTES: a test hook that forces
     many pieces through the two-buffer hand-over on small tables) */
  const char *chunkenv = getenv("GTAMD_TABLE_CHUNK");
  const uint64_t chunkbytes = chunkenv != NULL && atoll(chunkenv) >= 64 ? (uint64_t) atoll(chunkenv)
                                                                       : (64u << 20);
  const uint64_t chunk = chunkbytes / (narrow ? 8 : entrysize);
  table_writer w;
  pthread_t thread;
  int rc = 0, k = 0;
  memset(&w, 0, sizeof w);
  w.buf[0] = malloc(chunk * (narrow ? 8 : entrysize));
  w.buf[1] = malloc(chunk * (narrow ? 8 : entrysize));
  snprintf(path, sizeof path, "%s%s", index, suffix);
  w.fp = fopen(path, "wb");
  if (w.fp == NULL || w.buf[0] == NULL || w.buf[1] == NULL) {
    snprintf(err, errlen, "cannot open file '%s' for writing", path);
    free(w.buf[0]); free(w.buf[1]); if (w.fp) fclose(w.fp);
    return -1;
  }
  pthread_mutex_init(&w.mu, NULL);
  pthread_cond_init(&w.cv, NULL);
  if (pthread_create(&thread, NULL, table_writer_main, &w) != 0) {
    free(w.buf[0]); free(w.buf[1]); fclose(w.fp);
    return fail(err, errlen, "cannot start the writer of file '%s'", path);
  }
  for (uint32_t part = 0; part < nctx && rc == 0; part++) {
  gtamd_esa_ctx *ctx = ctxs[part];
  const uint64_t entries = gtamd_esa_table_entries(ctx, which);
  for (uint64_t first = 0; first < entries && rc == 0; first += chunk, k ^= 1) {
    const uint64_t cnt = entries - first < chunk ? entries - first : chunk;
    pthread_mutex_lock(&w.mu);
    while (w.full[k]) pthread_cond_wait(&w.cv, &w.mu);
    pthread_mutex_unlock(&w.mu);
    if (gtamd_esa_table_copy(ctx, which, w.buf[k], first, cnt) != 0) {
      snprintf(err, errlen, "%s", gtamd_esa_last_error());
      rc = -1;
      break;
    }
    if (narrow) {
      const uint64_t *wide = w.buf[k];
      uint32_t *out = w.buf[k];             /* in place, front to back */
      for (uint64_t e = 0; e < cnt; e++) out[e] = (uint32_t) wide[e];
    }
    pthread_mutex_lock(&w.mu);
    w.bytes[k] = (size_t) (cnt * entrysize);
    w.full[k] = 1;
    pthread_cond_broadcast(&w.cv);
    pthread_mutex_unlock(&w.mu);
  }
  }
  pthread_mutex_lock(&w.mu);
  while (w.full[0] || w.full[1]) pthread_cond_wait(&w.cv, &w.mu);
  w.done = 1;
  pthread_cond_broadcast(&w.cv);
  pthread_mutex_unlock(&w.mu);
  pthread_join(thread, NULL);
  pthread_mutex_destroy(&w.mu);
  pthread_cond_destroy(&w.cv);
  free(w.buf[0]); free(w.buf[1]);
  if (rc == 0 && w.failed) {
    snprintf(err, errlen, "cannot write to file '%s'", path);
    rc = -1;
  }
  if (fclose(w.fp) != 0 && rc == 0) return fail(err, errlen, "cannot close file '%s'", path);
  return rc;
}

/* ---------------------------------------------------------------------------
   -gpus R (not a reference option; the reference's -parts cuts the same way,
   src/match/sfx-partssuf.c:172-347, to save memory): R lexicographic ranges of the
   suffix table, one engine context and one host thread per range, on the GPUs of
   the node in turn (all on the one GPU when there is one: the tests).  The
   transport is the library's own (gtamd_comm_threads_create, include/gtamd_esa.h).
   --------------------------------------------------------------------------- */
typedef struct {
  gtamd_comm *comm;
  uint32_t part, numparts, userpl, want;
  int device, numofchars, rc;
  const uint8_t *enc;
  uint64_t n;
  gtamd_esa_ctx *ctx;
  gtamd_esa_stats es;
  char err[1024];
} part_job;

static void *part_main(void *arg)
{
  part_job *j = arg;
  j->rc = -1;
  j->ctx = gtamd_esa_create(j->device, j->n, (uint32_t) j->numofchars);
  if (j->ctx != NULL && gtamd_comm_attach(j->comm, j->part, j->ctx, j->device) == 0 &&
      gtamd_esa_set_prefixlength(j->ctx, j->userpl) == 0 &&
      gtamd_esa_set_sequence_bytes(j->ctx, j->enc, j->n, 0) == 0 &&
      gtamd_esa_run(j->ctx, j->want) == 0 && gtamd_esa_get_stats(j->ctx, &j->es) == 0)
    j->rc = 0;
  if (j->rc != 0) {
    snprintf(j->err, sizeof j->err, "%s", gtamd_esa_last_error());
    gtamd_comm_abort(j->comm);      /* the other parts must not wait for this one */
  }
  return NULL;
}

/* ctxs[0 .. R): the contexts holding the slices (the caller destroys them);
   es: the statistics of the whole table */
static int build_in_parts(const uint8_t *enc, uint64_t n, int numofchars, uint32_t userpl,
                          uint32_t want, uint32_t R, gtamd_esa_ctx **ctxs, gtamd_esa_stats *es,
                          char *err, size_t errlen)
{
  part_job *jobs = calloc(R, sizeof *jobs);
  pthread_t *threads = calloc(R, sizeof *threads);
  gtamd_comm *comm = gtamd_comm_threads_create(R);
  const char *same = getenv("GTAMD_GPUS_SAME_DEVICE");
  int ndev = gtamd_device_count(), rc = 0;
  uint32_t started = 0;
  uint64_t expect = 0;
  if (jobs == NULL || threads == NULL || comm == NULL) {
    free(jobs); free(threads); gtamd_comm_destroy(comm);
    return failf(err, errlen, "cannot set up a build in %u parts", R);
  }
  if (ndev < 1 || (same != NULL && same[0] == '1')) ndev = 1;
  for (uint32_t r = 0; r < R; r++) {
    jobs[r].comm = comm; jobs[r].part = r; jobs[r].numparts = R; jobs[r].userpl = userpl;
    jobs[r].want = want; jobs[r].device = (int) (r % (uint32_t) ndev);
    jobs[r].numofchars = numofchars; jobs[r].enc = enc; jobs[r].n = n;
    if (pthread_create(&threads[r], NULL, part_main, &jobs[r]) != 0) {
      gtamd_comm_abort(comm);
      rc = failf(err, errlen, "cannot start the thread of part %u", r);
      break;
    }
    started++;
  }
  for (uint32_t r = 0; r < started; r++) pthread_join(threads[r], NULL);
  memset(es, 0, sizeof *es);
  for (uint32_t r = 0; r < R; r++) {
    ctxs[r] = jobs[r].ctx;
    if (r < started && jobs[r].rc != 0 && rc == 0) {
      /* (the message of the part that failed first, not of those it took along) */
      uint32_t first = r;
      for (uint32_t q = 0; q < started; q++)
        if (jobs[q].rc != 0 && strstr(jobs[q].err, "of the build failed") == NULL &&
            strstr(jobs[q].err, "callback failed") == NULL) { first = q; break; }
      rc = failf(err, errlen, "part %u of %u: %s", first, R, jobs[first].err);
    }
  }
  for (uint32_t r = 0; r < R && rc == 0; r++) {
    const gtamd_esa_stats *p = &jobs[r].es;
    if (gtamd_esa_table_offset(ctxs[r]) != expect)
      rc = failf(err, errlen, "the slices of the parts do not tile the table (part %u)", r);
    expect += gtamd_esa_table_entries(ctxs[r], (want & GTAMD_WANT_SUF) ? GTAMD_TAB_SUF :
                                               (want & GTAMD_WANT_LCP) ? GTAMD_TAB_LCP : GTAMD_TAB_BWT);
    es->totallength = p->totallength;
    es->numberofallsortedsuffixes = p->numberofallsortedsuffixes;
    es->prefixlength = p->prefixlength;
    es->longest += p->longest;                 /* (0 from the parts without suffix 0) */
    es->largelcpvalues += p->largelcpvalues;
    es->lcptabsum += p->lcptabsum;
    es->tied_suffixes += p->tied_suffixes;
    es->pair_suffixes += p->pair_suffixes;
    es->device_bytes += p->device_bytes;
    if (p->maxbranchdepth > es->maxbranchdepth) es->maxbranchdepth = p->maxbranchdepth;
    if (p->refine_rounds > es->refine_rounds) es->refine_rounds = p->refine_rounds;
  }
  if (rc == 0 && expect != n + 1)
    rc = failf(err, errlen, "the parts hold %llu of the %llu table entries",
              (unsigned long long) expect, (unsigned long long) (n + 1));
  gtamd_comm_destroy(comm);
  free(jobs); free(threads);
  return rc;
}

static int yesno(int argc, const char **argv, int *i)
{
  /* options like -tis take an optional yes|no argument; returns the value */
  if (*i + 1 < argc && (!strcmp(argv[*i + 1], "yes") || !strcmp(argv[*i + 1], "no"))) {
    (*i)++;
    return !strcmp(argv[*i], "yes");
  }
  return 1;
}

/* (Measured and dropped: allocating the engine's workspace -- 2.4 s of hipMalloc
   at 3 Gbp -- on a second thread while the device reader runs.  The runtime
   serialises the two: the reader's stage grew by what the allocation took.) */
/* `gt packedindex mkindex` (gt_parseargsandcallsuffixerator(false, ...),
   src/tools/gt_packedindex.c:33-36): the suffixerator run below with the
   packed index as its only table (run_packedindexconstruction,
   src/match/sfx-run.c:369-425) */
typedef struct {
  gtamd_pck_params params;
  int locbitmap;                 /* -1: option not given */
  int sprank;
  int ctxilog;                   /* -2: no context map, -1: automatic interval */
} pck_request;

static int write_bdx(gtamd_esa_ctx *ctx, const pck_request *pr, uint32_t numofchars,
                     const char *indexname, int verbose, char *err, size_t errlen)
{
  gtamd_pck_params pp = pr->params;
  gtamd_pck *pck;
  gtamd_pck_info info;
  char path[4096];
  FILE *fp = NULL;
  uint8_t *buf = NULL;
  const uint64_t chunk = 64u << 20;
  int rc = -1;
  /* sfx-run.c:389-393 */
  if (numofchars > 10U && pp.block_size > 3U) pp.block_size = 3U;
  pp.feature_toggles = gtamd_pck_default_toggles(pp.block_size, pp.bucket_blocks,
                                                 pp.locate_interval, pr->locbitmap)
                       | (pr->sprank ? GTAMD_PCK_REVERSIBLY_SORTED : 0);
  pp.with_statistics = 1;
  if ((pck = gtamd_pck_create(0)) == NULL || gtamd_pck_build_from_esa(pck, ctx, &pp) != 0 ||
      gtamd_pck_get_info(pck, &info) != 0) {
    snprintf(err, errlen, "%s", gtamd_esa_last_error());
    goto done;
  }
  if (pr->ctxilog >= -1 && pp.locate_interval) {
    /* INDEX.<ilog>cxm, made beside the locate marks (eis-bwtseq-extinfo.c:473-476) */
    int used = 0;
    uint64_t n;
    uint8_t *m;
    FILE *mf;
    if (gtamd_pck_ctxmap_build_from_esa(pck, ctx, pr->ctxilog, &used) != 0) {
      snprintf(err, errlen, "%s", gtamd_esa_last_error());
      goto done;
    }
    n = gtamd_pck_ctxmap_bytes(pck);
    snprintf(path, sizeof path, "%s.%dcxm", indexname, used);
    if ((m = malloc(n ? n : 1)) == NULL || gtamd_pck_ctxmap_copy(pck, m, 0, n) != 0 ||
        (mf = fopen(path, "wb")) == NULL) {
      free(m);
      fail(err, errlen, "cannot write file '%s'", path);
      goto done;
    }
    if (fwrite(m, 1, n, mf) != n) { fclose(mf); free(m); fail(err, errlen, "cannot write file '%s'", path); goto done; }
    fclose(mf);
    free(m);
  }
  snprintf(path, sizeof path, "%s.bdx", indexname);
  if ((fp = fopen(path, "wb")) == NULL || (buf = malloc(chunk)) == NULL) {
    fail(err, errlen, "cannot open file '%s' for writing", path);
    goto done;
  }
  for (uint64_t off = 0; off < info.file_bytes; off += chunk) {
    const uint64_t cnt = info.file_bytes - off < chunk ? info.file_bytes - off : chunk;
    if (gtamd_pck_image_copy(pck, buf, off, cnt) != 0) { snprintf(err, errlen, "%s", gtamd_esa_last_error()); goto done; }
    if (fwrite(buf, 1, cnt, fp) != cnt) { fail(err, errlen, "cannot write file '%s'", path); goto done; }
  }
  if (verbose)
    printf("# packed index: blocksize=%u, blocks-per-bucket=%u, locfreq=%u: %llu bytes, %.2f ms on the device\n",
           pp.block_size, pp.bucket_blocks, pp.locate_interval,
           (unsigned long long) info.file_bytes, info.build_ms);
  rc = 0;
done:
  if (fp != NULL && fclose(fp) != 0 && rc == 0) rc = fail(err, errlen, "cannot close file '%s'", path);
  free(buf);
  gtamd_pck_destroy(pck);
  return rc;
}

static int suffixerator_run(int argc, const char **argv, char *err, size_t errlen,
                            const pck_request *pr);

int gtamd_suffixerator(int argc, const char **argv, char *err, size_t errlen)
{
  return suffixerator_run(argc, argv, err, errlen, NULL);
}

static int uint_arg(int argc, const char **argv, int *i, uint32_t *out, char *err, size_t errlen)
{
  char *end;
  unsigned long v;
  if (*i + 1 >= argc) return fail(err, errlen, "missing argument to option \"%s\"", argv[*i]);
  v = strtoul(argv[*i + 1], &end, 10);
  if (*end != 0 || argv[*i + 1][0] == '-')
    return fail(err, errlen, "argument to option \"%s\" must be a non-negative integer", argv[*i]);
  *out = (uint32_t) v;
  (*i)++;
  return 0;
}

int gtamd_packedindex_mkindex(int argc, const char **argv, char *err, size_t errlen)
{
  pck_request pr = { { 8, 8, 16, 0, 1 }, -1, 0, -2 };
  const char **rest = malloc(sizeof *rest * (size_t) (argc + 1));
  int nrest = 0, rc;
  if (rest == NULL) return fail(err, errlen, "out of memory (%s)", "packedindex mkindex");
  rest[nrest++] = argc > 0 ? argv[0] : "mkindex";
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    rc = 0;
    if (!strcmp(a, "-bsize")) rc = uint_arg(argc, argv, &i, &pr.params.block_size, err, errlen);
    else if (!strcmp(a, "-blbuck")) rc = uint_arg(argc, argv, &i, &pr.params.bucket_blocks, err, errlen);
    else if (!strcmp(a, "-locfreq")) rc = uint_arg(argc, argv, &i, &pr.params.locate_interval, err, errlen);
    else if (!strcmp(a, "-locbitmap")) pr.locbitmap = yesno(argc, argv, &i);
    else if (!strcmp(a, "-sprank")) pr.sprank = yesno(argc, argv, &i);
    else if (!strcmp(a, "-sprankilog")) {
      if (i + 1 >= argc) rc = fail(err, errlen, "missing argument to option \"%s\"", a);
      else if (atoi(argv[++i]) >= 0) pr.sprank = 1;
    } else if (!strcmp(a, "-ctxilog")) {
      if (i + 1 >= argc) rc = fail(err, errlen, "missing argument to option \"%s\"", a);
      else {
        pr.ctxilog = atoi(argv[++i]);
        if (pr.ctxilog < -2 || pr.ctxilog > 63)
          rc = fail(err, errlen, "argument to option \"%s\" must be an integer between -2 and 63", a);
      }
    }
    else if (!strcmp(a, "-suf") || !strcmp(a, "-lcp") || !strcmp(a, "-bwt") || !strcmp(a, "-bck") ||
             !strcmp(a, "-suftabuint"))
      /* the index options of the packed-index variant have no table switches
         (src/match/index_options.c: gt_index_options_register_packedidx) */
      rc = fail(err, errlen, "unknown option: %s (try -help)", a);
    else rest[nrest++] = a;
    if (rc != 0) { free(rest); return -1; }
  }
  if (pr.params.block_size < 1 || pr.params.bucket_blocks < 1) {
    free(rest);
    return fail(err, errlen, "argument to option \"-%s\" must be an integer >= 1",
                pr.params.block_size < 1 ? "bsize" : "blbuck");
  }
  rc = suffixerator_run(nrest, rest, err, errlen, &pr);
  free(rest);
  return rc;
}

static int suffixerator_run(int argc, const char **argv, char *err, size_t errlen,
                            const pck_request *pr)
{
  const char *db[MAXDB], *indexname = NULL, *inputindex = NULL, *sat = NULL, *smap = NULL;
  gtamd_alphabet alpha;
  int dnalike;
  size_t numdb = 0;
  int protein = 0, dna = 0, verbose = 0, readmode = 0, mirrored = 0,
      out_des = 1, out_sds = 1, out_md5 = 1, out_ssp = 1;   /* defaults of encseq_options.c */
  char *desc = NULL;
  uint64_t desclen = 0;
  uint32_t want = 0, userpl = 0;
  char indexbuf[4096];
  uint8_t *enc = NULL;
  uint64_t n = 0;
  gtamd_seqstats ss;
  gtamd_encinfo info;
  gtamd_esa_stats es;
  gtamd_esa_ctx *ctx = NULL, **ctxs = NULL;
  uint32_t gpus = 1, nctx = 1;
  gtamd_encoder *de = NULL;
  int rc = -1, host_encoder = 0, suftabuint = 0, clipdesc = 0, lossless = 0;
  const char *reader = "host";
  int device_rc = 0;
  uint8_t *orig = NULL;
  double t0 = now_s(), t_seq, t_build, t_create;

  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    if (!strcmp(a, "-db")) {
      while (i + 1 < argc && argv[i + 1][0] != '-') {
        if (numdb == MAXDB) return fail(err, errlen, "too many arguments to option \"-%s\"", "db");
        db[numdb++] = argv[++i];
      }
      if (numdb == 0) return fail(err, errlen, "missing argument to option \"-%s\"", "db");
    } else if (!strcmp(a, "-ii")) {
      if (i + 1 >= argc) return fail(err, errlen, "missing argument to option \"-%s\"", "ii");
      inputindex = argv[++i];
    } else if (!strcmp(a, "-indexname")) {
      if (i + 1 >= argc) return fail(err, errlen, "missing argument to option \"-%s\"", "indexname");
      indexname = argv[++i];
    } else if (!strcmp(a, "-dna")) dna = 1;
    else if (!strcmp(a, "-protein")) protein = 1;
    else if (!strcmp(a, "-smap")) {
      if (i + 1 >= argc) return fail(err, errlen, "missing argument to option \"-%s\"", "smap");
      smap = argv[++i];
    }
    else if (!strcmp(a, "-suf")) want |= GTAMD_WANT_SUF;
    else if (!strcmp(a, "-lcp")) want |= GTAMD_WANT_LCP;
    else if (!strcmp(a, "-bwt")) want |= GTAMD_WANT_BWT;
    else if (!strcmp(a, "-bck")) want |= GTAMD_WANT_BCK;
    else if (!strcmp(a, "-suftabuint")) suftabuint = 1;
    else if (!strcmp(a, "-v")) verbose = 1;
    else if (!strcmp(a, "-pl")) {
      if (i + 1 < argc && argv[i + 1][0] != '-') userpl = (uint32_t) strtoul(argv[++i], NULL, 10);
    } else if (!strcmp(a, "-dir")) {
      if (i + 1 >= argc) return fail(err, errlen, "missing argument to option \"-%s\"", "dir");
      i++;
      if (!strcmp(argv[i], "fwd")) readmode = 0;
      else if (!strcmp(argv[i], "rev")) readmode = 1;
      else if (!strcmp(argv[i], "cpl")) readmode = 2;
      else if (!strcmp(argv[i], "rcl")) readmode = 3;
      else
        return fail(err, errlen, "argument to option -dir must be fwd or rev or cpl or rcl, not %s", argv[i]);
    } else if (!strcmp(a, "-mirrored")) {
      mirrored = 1;
    } else if (!strcmp(a, "-gpus")) {
      /* not a reference option: build the tables in that many lexicographic ranges,
         one per GPU (or all on the one there is) */
      long v = i + 1 < argc ? strtol(argv[i + 1], NULL, 10) : 0;
      if (v < 1 || v > 128)
        return failf(err, errlen, "argument to option \"-%s\" must be an integer from 1 to 128", "gpus");
      gpus = (uint32_t) v;
      i++;
    } else if (!strcmp(a, "-parts") || !strcmp(a, "-memlimit") || !strcmp(a, "-dc") ||
               !strcmp(a, "-maxwidthrealmedian")) {
      /* space/strategy knobs of the CPU algorithm: the tables do not depend on
         them, the device build ignores them */
      if (i + 1 < argc && argv[i + 1][0] != '-') i++;
    } else if (!strcmp(a, "-algbds")) {
      while (i + 1 < argc && argv[i + 1][0] != '-') i++;
    } else if (!strcmp(a, "-cmpcharbychar") || !strcmp(a, "-dccheck") ||
               !strcmp(a, "-iterscan") || !strcmp(a, "-kmerswithencseqreader") ||
               !strcmp(a, "-noshortreadsort") || !strcmp(a, "-samplewithprefixlengthnull") ||
               !strcmp(a, "-storespecialcodes") || !strcmp(a, "-withradixsort")) {
      (void) yesno(argc, argv, &i);        /* more strategy switches, same tables */
    } else if (!strcmp(a, "-clipdesc")) {
      clipdesc = yesno(argc, argv, &i);
    } else if (!strcmp(a, "-sat")) {
      if (i + 1 >= argc) return fail(err, errlen, "missing argument to option \"-%s\"", "sat");
      sat = argv[++i];
    } else if (!strcmp(a, "-lossless")) {
      lossless = yesno(argc, argv, &i);
    } else if (!strcmp(a, "-plain") || !strcmp(a, "-kys") || !strcmp(a, "-lcpdist") ||
               !strcmp(a, "-compressedoutput") || !strcmp(a, "-genomediff") ||
               !strcmp(a, "-sortmaxdepth") || !strcmp(a, "-spmopt") ||
               !strcmp(a, "-swallow-tail") || !strcmp(a, "-onlybucketinsertion")) {
      /* these change what is written; not part of this path */
      return fail(err, errlen, "option \"%s\" is not supported by the MI355X engine", a);
    } else if (!strcmp(a, "-des")) out_des = yesno(argc, argv, &i);
    else if (!strcmp(a, "-sds")) out_sds = yesno(argc, argv, &i);
    else if (!strcmp(a, "-md5")) out_md5 = yesno(argc, argv, &i);
    else if (!strcmp(a, "-ssp")) out_ssp = yesno(argc, argv, &i);
    else if (!strcmp(a, "-encoder")) {
      /* not a reference option: read FASTA on the host instead of the device */
      if (i + 1 >= argc || (strcmp(argv[i + 1], "host") && strcmp(argv[i + 1], "device")))
        return fail(err, errlen, "argument to option -%s must be host or device", "encoder");
      host_encoder = !strcmp(argv[++i], "host");
    }
    else if (!strcmp(a, "-tis") || !strcmp(a, "-showprogress")) {
      (void) yesno(argc, argv, &i);
    } else
      return fail(err, errlen, "unknown option: %s (try -help)", a);
  }
  if (pr != NULL) want = GTAMD_WANT_SUF | GTAMD_WANT_BWT;   /* what the packed index is made from */
  /* src/match/sfx-opt.c:78-88 */
  if (numdb == 0 && inputindex == NULL)
    return fail(err, errlen, "either option \"-db\" or option \"-%s\" is mandatory", "ii");
  if (dna && protein)
    return fail(err, errlen, "option \"-dna\" and option \"-%s\" exclude each other", "protein");
  if (smap != NULL && (dna || protein))
    /* src/core/encseq_options.c:271-272 */
    return fail(err, errlen, "option \"-smap\" and option \"-%s\" exclude each other",
                dna ? "dna" : "protein");
  if (inputindex != NULL && (numdb > 0 || dna || protein || sat != NULL || smap != NULL))
    return fail(err, errlen, "option \"-%s\" and option \"-ii\" exclude each other",
                numdb > 0 ? "db" : smap != NULL ? "smap" : dna ? "dna" : protein ? "protein" : "sat");
  if (indexname == NULL && inputindex != NULL) {
    const char *base = strrchr(inputindex, '/');
    snprintf(indexbuf, sizeof indexbuf, "%s", base ? base + 1 : inputindex);
    indexname = indexbuf;
  }
  if (indexname == NULL) {
    /* default: basename of the single -db file (encseq_options.c:112-131) */
    const char *base;
    if (numdb > 1)
      return fail(err, errlen, "if more than one input file is given, then option -%s is mandatory", "indexname");
    base = strrchr(db[0], '/');
    snprintf(indexbuf, sizeof indexbuf, "%s", base ? base + 1 : db[0]);
    indexname = indexbuf;
  }
  if (smap != NULL) {
    if (gtamd_alphabet_from_file(smap, &alpha, err, errlen) != 0) return -1;
    if (alpha.numofchars > 28) {
      gtamd_alphabet_free(&alpha);
      return fail(err, errlen, "symbol map '%s' defines more than 28 letters", smap);
    }
  } else gtamd_alphabet_standard(&alpha, protein);
  if (inputindex != NULL) {
    /* an existing encoded sequence: nothing on the sequence side is rewritten;
       .prj repeats the statistics stored with it (src/match/sfx-run.c:454-493) */
    gtamd_alphabet_free(&alpha);
    if (gtamd_read_esq_alpha(inputindex, &enc, &n, &alpha, &ss, err, errlen) != 0) return -1;
  }
  /* complementing needs a=0 c=1 g=2 t=3 (gt_alphabet_is_dna) */
  dnalike = alpha.numofchars == 4 && alpha.symbolmap['a'] == 0 && alpha.symbolmap['c'] == 1 &&
            alpha.symbolmap['g'] == 2 && alpha.symbolmap['t'] == 3;
  if (!dnalike && (readmode >= 2 || mirrored)) {
    /* wording of src/match/sfx-run.c:566-570 */
    free(enc); gtamd_alphabet_free(&alpha);
    return fail(err, errlen, "option -%s only can be used for DNA alphabets",
                mirrored ? "mirrored" : (readmode == 2 ? "cpl" : "rcl"));
  }
  if (inputindex != NULL) {
    reader = "index";   /* (read above) */
  } else if (want != 0 && !host_encoder && !lossless &&
             (device_rc = gtamd_device_encode_files_alpha(db, numdb, &alpha, &de, &desc, &desclen, &info,
                                                          err, errlen)) != GTAMD_DEVICE_DECLINED) {
    /* tables requested: FASTA, and FASTQ in its four-line form, are read and
       encoded on the device; the symbols stay in HBM for the engine and come to
       the host only where a file needs them.  (FASTQ the device reader declines
       -- sequences over several lines and whatever is malformed -- goes to the
       host reader below, which has the reference's messages.) */
    double ts[4];
    reader = "device";
    if (device_rc != 0) return -1;
    ts[0] = now_s();
    n = gtamd_encoder_length(de);
    if (gtamd_write_esq_device_alpha(indexname, db, numdb, de, &alpha, &info, out_ssp, sat, &ss, err, errlen) != 0) {
      free(desc); gtamd_encinfo_free(&info); gtamd_encoder_destroy(de);
      return -1;
    }
    ts[1] = now_s();
    gtamd_encinfo_free(&info);
    if (clipdesc) gtamd_clip_descriptions(desc, &desclen);
    if ((out_des || out_sds) && gtamd_write_des_sds(indexname, desc, desclen, out_des, out_sds) != 0) {
      free(desc); gtamd_encoder_destroy(de);
      return fail(err, errlen, "cannot write description files of index '%s'", indexname);
    }
    free(desc);
    ts[2] = now_s();
    if (verbose)
      printf("# device reader: files read and encoded %.3f s, .esq/.ssp %.3f, .des/.sds %.3f\n",
             ts[0] - t0, ts[1] - ts[0], ts[2] - ts[1]);
    if (out_md5 || mirrored || readmode != 0) {
      /* MD5 sums and the -dir / -mirrored transforms work on host symbols */
      if ((enc = malloc(n ? n : 1)) == NULL || gtamd_encoder_copy_symbols(de, enc, 0, n) != 0) {
        free(enc); gtamd_encoder_destroy(de);
        return fail(err, errlen, "cannot copy the encoded sequence from the device (%s)",
                    gtamd_esa_last_error());
      }
      if (out_md5 && gtamd_write_md5_alpha(indexname, enc, n, &alpha) != 0) {
        free(enc); gtamd_encoder_destroy(de);
        return fail(err, errlen, "cannot write md5 file of index '%s'", indexname);
      }
      if (mirrored || readmode != 0) { gtamd_encoder_destroy(de); de = NULL; }
      else { free(enc); enc = NULL; }
    }
  } else {
    if (gtamd_encode_files_orig(db, numdb, &alpha, &enc, &n, lossless ? &orig : NULL, &desc,
                                &desclen, &info, err, errlen) != 0)
      return -1;
    /* -lossless: the exception table first, its counts go into INDEX.esq */
    if (lossless && gtamd_write_ois(indexname, enc, orig, n, &alpha, &info, err, errlen) != 0) {
      free(enc); free(orig); free(desc); gtamd_encinfo_free(&info);
      return -1;
    }
    /* the encoded sequence itself, in the reference's format (always written:
       -tis is kept for backwards compatibility only, src/match/sfx-opt.c).
       With tables to build a device is needed anyway: then the symbols of the
       host reader go there at once, and statistics and the sections of
       INDEX.esq come from the device as with the device reader. */
    if (want != 0) {
      de = gtamd_encoder_create_map(0, alpha.symbolmap, alpha.numofchars, alpha.bitspersymbol);
      if (de == NULL || gtamd_encoder_set_symbols(de, enc, n) != 0) {
        snprintf(err, errlen, "%s", gtamd_esa_last_error());
        free(enc); free(orig); free(desc); gtamd_encinfo_free(&info); gtamd_encoder_destroy(de);
        return -1;
      }
    }
    if ((de != NULL ? gtamd_write_esq_device_alpha(indexname, db, numdb, de, &alpha, &info,
                                                   out_ssp, sat, &ss, err, errlen)
                    : gtamd_write_esq_alpha(indexname, db, numdb, enc, n, &alpha, &info, out_ssp,
                                            sat, &ss, err, errlen)) != 0) {
      free(enc); free(orig); free(desc); gtamd_encinfo_free(&info); gtamd_encoder_destroy(de);
      return -1;
    }
    gtamd_encinfo_free(&info);
    /* the sequence-side files describe the sequence as stored (before -dir) */
    if (clipdesc) gtamd_clip_descriptions(desc, &desclen);
    if ((out_des || out_sds) && gtamd_write_des_sds(indexname, desc, desclen, out_des, out_sds) != 0) {
      free(enc); free(orig); free(desc); gtamd_encoder_destroy(de);
      return fail(err, errlen, "cannot write description files of index '%s'", indexname);
    }
    free(desc);
    if (out_md5 && (lossless ? gtamd_write_md5_orig(indexname, enc, orig, n)
                             : gtamd_write_md5_alpha(indexname, enc, n, &alpha)) != 0) {
      free(enc); free(orig); gtamd_encoder_destroy(de);
      return fail(err, errlen, "cannot write md5 file of index '%s'", indexname);
    }
    free(orig);
    orig = NULL;
    /* (.prj describes the sequence as stored, the tables the sequence as read)
       -dir / -mirrored transform host symbols; otherwise the engine reads the
       device copy */
    if (de != NULL) {
      if (mirrored || readmode != 0) { gtamd_encoder_destroy(de); de = NULL; }
      else { free(enc); enc = NULL; }
    }
  }
  if (mirrored) {
    uint8_t *m = gtamd_mirror(enc, n);
    if (m == NULL) { free(enc); return fail(err, errlen, "out of memory (%s)", "-mirrored"); }
    gtamd_seqstats_mirror(&ss, n > 0 && enc[n - 1] == GTAMD_WILDCARD);
    free(enc);
    enc = m;
    n = 2 * n + 1;
  }
  if (enc != NULL) gtamd_apply_readmode(enc, n, readmode);
  t_seq = now_s() - t0;
  if (verbose) {
    printf("# totallength=%llu\n# specialcharacters=%llu\n# numofsequences=%llu\n",
           (unsigned long long) ss.totallength, (unsigned long long) ss.specialcharacters,
           (unsigned long long) ss.numofsequences);
  }
  memset(&es, 0, sizeof es);
  es.totallength = n;
  es.numberofallsortedsuffixes = n + 1;
  es.prefixlength = userpl ? userpl : gtamd_recommended_prefixlength(ss.numofchars, n);
  if (want == 0) {
    /* nothing but the sequence statistics requested: the reference still
       writes the project file (src/match/sfx-run.c:664-694) */
    char path[4096];
    free(enc);
    snprintf(path, sizeof path, "%s.prj", indexname);
    gtamd_alphabet_free(&alpha);
    if (gtamd_write_prj(path, &ss, &es, 0, readmode, mirrored) != 0)
      return fail(err, errlen, "cannot open file '%s' for writing", path);
    return 0;
  }
  if (gpus > 1) {
    /* in parts: every part packs the sequence on its own device from the host copy */
    if ((want & GTAMD_WANT_BCK) || pr != NULL) {
      failf(err, errlen, "option \"-gpus\" cannot be combined with %s: that needs the whole "
           "table in one build", pr != NULL ? "the packed index" : "option \"-bck\"");
      goto done;
    }
    if (enc == NULL) {
      if ((enc = malloc(n ? n : 1)) == NULL || gtamd_encoder_copy_symbols(de, enc, 0, n) != 0) {
        failf(err, errlen, "cannot bring the encoded sequence to the host: %s", gtamd_esa_last_error());
        goto done;
      }
    }
    gtamd_encoder_destroy(de);
    de = NULL;
    if ((ctxs = calloc(gpus, sizeof *ctxs)) == NULL) { failf(err, errlen, "out of memory"); goto done; }
    nctx = gpus;
    t_create = 0;
    if (build_in_parts(enc, n, (int) ss.numofchars, userpl, want, gpus, ctxs, &es, err, errlen) != 0) goto done;
    ctx = ctxs[0];
    goto built;
  }
  t_create = now_s();
  ctx = gtamd_esa_create(0, n, ss.numofchars);
  t_create = now_s() - t_create;
  if (ctx == NULL) {
    snprintf(err, errlen, "%s", gtamd_esa_last_error());
    free(enc); gtamd_encoder_destroy(de);
    gtamd_alphabet_free(&alpha);
    return -1;
  }
  if (gtamd_esa_set_prefixlength(ctx, userpl) != 0 ||
      (de != NULL ? gtamd_esa_set_sequence_bytes(ctx, gtamd_encoder_device_symbols(de), n, 1)
                  : gtamd_esa_set_sequence_bytes(ctx, enc, n, 0)) != 0 ||
      gtamd_esa_run(ctx, want) != 0 || gtamd_esa_get_stats(ctx, &es) != 0) {
    snprintf(err, errlen, "%s", gtamd_esa_last_error());
    goto done;
  }
built:
  t_build = now_s() - t0 - t_seq;
  if (verbose && gpus > 1) printf("# tables built in %u parts\n", gpus);
  if (verbose)
    printf("# prefixlength=%u\n# tied suffixes after the first sort=%llu, refinement rounds=%u\n",
           es.prefixlength, (unsigned long long) es.tied_suffixes, es.refine_rounds);
  if (pr != NULL) {
    /* the packed index is the only table of this run; the project file says so:
       no suffixes written, no longest (src/match/sfx-run.c:600-690 with doesa false) */
    char path[4096];
    if (write_bdx(ctx, pr, ss.numofchars, indexname, verbose, err, errlen) != 0) goto done;
    snprintf(path, sizeof path, "%s.prj", indexname);
    if (gtamd_write_prj_packedindex(path, &ss, es.prefixlength, readmode, mirrored) != 0) {
      fail(err, errlen, "cannot open file '%s' for writing", path);
      goto done;
    }
    rc = 0;
    goto done;
  }
  {
    gtamd_esa_ctx *const *tabs = ctxs != NULL ? ctxs : &ctx;
    if ((want & GTAMD_WANT_SUF) &&
        write_table(tabs, nctx, GTAMD_TAB_SUF, indexname, ".suf", suftabuint ? 4 : 8, err, errlen) != 0) goto done;
    if ((want & GTAMD_WANT_BCK) && write_bcktab(ctx, indexname, err, errlen) != 0) goto done;
    if (want & GTAMD_WANT_LCP) {
      if (write_table(tabs, nctx, GTAMD_TAB_LCP, indexname, ".lcp", 1, err, errlen) != 0) goto done;
      if (write_table(tabs, nctx, GTAMD_TAB_LLV, indexname, ".llv", 16, err, errlen) != 0) goto done;
    }
    if ((want & GTAMD_WANT_BWT) && write_table(tabs, nctx, GTAMD_TAB_BWT, indexname, ".bwt", 1, err, errlen) != 0) goto done;
  }
  {
    char path[4096];
    snprintf(path, sizeof path, "%s.prj", indexname);
    if (gtamd_write_prj(path, &ss, &es, (want & GTAMD_WANT_LCP) != 0, readmode, mirrored) != 0) {
      fail(err, errlen, "cannot open file '%s' for writing", path);
      goto done;
    }
  }
  if (verbose && de != NULL) {
    float total_ms = 0, parse_ms = 0, stats_ms = 0;
    uint64_t bytes = 0;
    (void) gtamd_encoder_get_timing(de, &total_ms, &parse_ms, &stats_ms, &bytes);
    printf("# device encoder: %llu input bytes in %.2f ms (upload and parsing %.2f, "
           "statistics %.2f)\n", (unsigned long long) bytes, total_ms, parse_ms, stats_ms);
  }
  if (verbose) {
    gtamd_esa_timing tm;
    memset(&tm, 0, sizeof tm);
    (void) gtamd_esa_get_timing(ctx, &tm);
    /* a tool run is always a cold run: the context allocates its workspace
       (140 GB at 3 Gbp) inside this build; a caller that keeps the context
       pays `kernels` only from the second build on */
    printf("# seconds: input, encoding and sequence files (%s reader) %.3f; tables on the "
           "device %.3f cold (device memory allocation %.3f, kernels incl. first touch %.3f; "
           "a warm context needs the kernels only); tables to files %.3f\n",
           reader, t_seq, t_build, t_create + tm.alloc_ms / 1e3, tm.total_ms / 1e3,
           now_s() - t0 - t_seq - t_build);
    printf("# device memory held: %.1f GB\n", (double) es.device_bytes / 1e9);
  }
  rc = 0;
done:
  gtamd_alphabet_free(&alpha);
  if (ctxs != NULL) {
    for (uint32_t r = 0; r < nctx; r++) gtamd_esa_destroy(ctxs[r]);
    free(ctxs);
  } else
    gtamd_esa_destroy(ctx);
  gtamd_encoder_destroy(de);
  free(enc);
  return rc;
}

/* ---------------------------------------------------------------------------
   gt dev mergeesa (src/tools/gt_mergeesa.c, src/match/test-mergeesa.c:110-190,
   src/match/esa-merge.c:136-200): INDEX.suf / .lcp / .llv of the concatenation
   of the sequence sets of several indexes, in the order given, one separator
   between consecutive sets (src/match/encseq2offset.c) -- by the reference's
   own test the very tables `gt suffixerator` writes for all the files at once
   (testsuite/gt_mergeesa_include.rb:17-19).  The reference merges the input
   tables through a trie because sorting is what costs it minutes; on a device
   that sorts 15 Gbp/s the merge IS a build: the encoded sequences of the input
   indexes are read (INDEX.esq, any access type), joined and built.  The input
   .suf/.lcp tables only have to exist (as the reference demands of its input).
   --------------------------------------------------------------------------- */
int gtamd_mergeesa(int argc, const char **argv, char *err, size_t errlen)
{
  const char *indexname = NULL;
  const char **inputs = NULL;
  size_t numinputs = 0;
  uint8_t *all = NULL;
  uint64_t total = 0;
  int protein_all = -1, rc = -1;
  gtamd_esa_ctx *ctx = NULL;

  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-indexname")) {
      if (++i >= argc) return fail(err, errlen, "missing argument to option \"%s\"", "-indexname");
      indexname = argv[i];
    } else if (!strcmp(argv[i], "-ii")) {
      inputs = argv + i + 1;
      while (i + 1 < argc && argv[i + 1][0] != '-') { i++; numinputs++; }
    } else
      return fail(err, errlen, "unknown option: %s (try -indexname NAME -ii INDEX...)", argv[i]);
  }
  if (indexname == NULL) return fail(err, errlen, "option \"%s\" is mandatory", "-indexname");
  if (numinputs == 0) return fail(err, errlen, "option \"%s\" is mandatory", "-ii");
  printf("# storeindex=%s\n", indexname);
  for (size_t k = 0; k < numinputs; k++) {
    uint8_t *enc = NULL, *grown;
    uint64_t n = 0;
    int protein = 0;
    gtamd_seqstats ss;
    char path[4096];
    FILE *fp;
    printf("# input=%s\n", inputs[k]);
    /* the reference maps SARR_SUFTAB | SARR_LCPTAB of every input */
    snprintf(path, sizeof path, "%s.suf", inputs[k]);
    if ((fp = fopen(path, "rb")) == NULL) { fail(err, errlen, "cannot open file '%s'", path); goto done; }
    fclose(fp);
    snprintf(path, sizeof path, "%s.lcp", inputs[k]);
    if ((fp = fopen(path, "rb")) == NULL) { fail(err, errlen, "cannot open file '%s'", path); goto done; }
    fclose(fp);
    if (gtamd_read_esq(inputs[k], &enc, &n, &protein, &ss, err, errlen) != 0) goto done;
    if (protein_all >= 0 && protein != protein_all) {
      free(enc);
      fail(err, errlen, "index '%s' has another alphabet than the indexes before it", inputs[k]);
      goto done;
    }
    protein_all = protein;
    grown = realloc(all, total + n + 2);
    if (grown == NULL) { free(enc); fail(err, errlen, "out of memory (%s)", "mergeesa"); goto done; }
    all = grown;
    if (k > 0) all[total++] = (uint8_t) GTAMD_SEPARATOR;
    memcpy(all + total, enc, n);
    total += n;
    free(enc);
  }
  ctx = gtamd_esa_create(0, total, protein_all ? 20 : 4);
  if (ctx == NULL || gtamd_esa_set_sequence_bytes(ctx, all, total, 0) != 0 ||
      gtamd_esa_run(ctx, GTAMD_WANT_SUF | GTAMD_WANT_LCP) != 0) {
    snprintf(err, errlen, "%s", gtamd_esa_last_error());
    goto done;
  }
  if (write_table(&ctx, 1, GTAMD_TAB_SUF, indexname, ".suf", 8, err, errlen) != 0 ||
      write_table(&ctx, 1, GTAMD_TAB_LCP, indexname, ".lcp", 1, err, errlen) != 0 ||
      write_table(&ctx, 1, GTAMD_TAB_LLV, indexname, ".llv", 16, err, errlen) != 0)
    goto done;
  rc = 0;
done:
  gtamd_esa_destroy(ctx);
  free(all);
  return rc;
}
