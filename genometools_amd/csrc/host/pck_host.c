/* pck_host.c -- `gt packedindex trsuftab [options] INDEX` for this path
   (tool function src/tools/gt_packedindex_trsuftab.c:44-79): INDEX.bdx from
   the project's INDEX.prj / .esq / .suf / .bwt, built on the device through
   include/gtamd_pck.h.  Options and defaults: src/match/eis-bwtseq-param.c:25-67,
   src/match/eis-blockcomp-param.c:21-36. */
#include "gtamd_host.h"
#include "gtamd_pck.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int pfail(char *err, size_t errlen, const char *msg, const char *arg)
{
  snprintf(err, errlen, msg, arg);
  return -1;
}

/* one "key=value" line of INDEX.prj (src/match/esa-scanprj.c) */
static int prj_value(const char *path, const char *key, unsigned long long *value)
{
  char line[512];
  const size_t klen = strlen(key);
  FILE *fp = fopen(path, "r");
  int found = 0;
  if (fp == NULL) return -1;
  while (fgets(line, sizeof line, fp) != NULL)
    if (!strncmp(line, key, klen) && line[klen] == '=') {
      *value = strtoull(line + klen + 1, NULL, 10);
      found = 1;
    }
  fclose(fp);
  return found ? 0 : -1;
}

static void *read_whole(const char *path, uint64_t expect_bytes)
{
  FILE *fp = fopen(path, "rb");
  void *buf;
  if (fp == NULL) return NULL;
  buf = malloc(expect_bytes ? expect_bytes : 1);
  if (buf != NULL && (fread(buf, 1, expect_bytes, fp) != expect_bytes || fgetc(fp) != EOF)) {
    free(buf);
    buf = NULL;
  }
  fclose(fp);
  return buf;
}

/* INDEX.<ilog>cxm from the builder's context map image */
static int write_ctxmap(gtamd_pck *pck, const char *index, int ilog_used, char *err, size_t errlen)
{
  char path[4096];
  const uint64_t n = gtamd_pck_ctxmap_bytes(pck);
  uint8_t *buf = malloc(n ? n : 1);
  FILE *fp;
  int rc = -1;
  snprintf(path, sizeof path, "%s.%dcxm", index, ilog_used);
  if (buf == NULL) return pfail(err, errlen, "out of memory (%s)", "context map");
  if (gtamd_pck_ctxmap_copy(pck, buf, 0, n) != 0) snprintf(err, errlen, "%s", gtamd_esa_last_error());
  else if ((fp = fopen(path, "wb")) == NULL) pfail(err, errlen, "cannot open file '%s' for writing", path);
  else {
    rc = fwrite(buf, 1, n, fp) == n ? 0 : pfail(err, errlen, "cannot write file '%s'", path);
    if (fclose(fp) != 0 && rc == 0) rc = pfail(err, errlen, "cannot close file '%s'", path);
  }
  free(buf);
  return rc;
}

static int uint_option(int argc, const char **argv, int *i, uint32_t *out, char *err, size_t errlen)
{
  char *end;
  unsigned long v;
  if (*i + 1 >= argc) return pfail(err, errlen, "missing argument to option \"%s\"", argv[*i]);
  v = strtoul(argv[*i + 1], &end, 10);
  if (*end != 0 || argv[*i + 1][0] == '-')
    return pfail(err, errlen, "argument to option \"%s\" must be a non-negative integer", argv[*i]);
  *out = (uint32_t) v;
  (*i)++;
  return 0;
}

int gtamd_packedindex_trsuftab(int argc, const char **argv, char *err, size_t errlen)
{
  gtamd_pck_params pp = { 8, 8, 16, 0, 0 };
  int locbitmap = -1, verbose = 0, rc = -1, protein = 0, sprank = 0, ctxilog = -2;
  const char *index = NULL;
  char path[4096];
  unsigned long long totallength, longest, integersize = 64;
  uint8_t *enc = NULL, *bwt = NULL;
  uint64_t *suf = NULL, n = 0;
  gtamd_seqstats ss;
  gtamd_pck *pck = NULL;
  gtamd_pck_info info;
  FILE *fp = NULL;

  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    if (!strcmp(a, "-bsize")) { if (uint_option(argc, argv, &i, &pp.block_size, err, errlen)) return -1; }
    else if (!strcmp(a, "-blbuck")) { if (uint_option(argc, argv, &i, &pp.bucket_blocks, err, errlen)) return -1; }
    else if (!strcmp(a, "-locfreq")) { if (uint_option(argc, argv, &i, &pp.locate_interval, err, errlen)) return -1; }
    else if (!strcmp(a, "-locbitmap")) {
      locbitmap = 1;
      if (i + 1 < argc && (!strcmp(argv[i + 1], "yes") || !strcmp(argv[i + 1], "no")))
        locbitmap = !strcmp(argv[++i], "yes");
    } else if (!strcmp(a, "-v")) verbose = 1;
    else if (!strcmp(a, "-sprank")) {
      sprank = 1;
      if (i + 1 < argc && (!strcmp(argv[i + 1], "yes") || !strcmp(argv[i + 1], "no")))
        sprank = !strcmp(argv[++i], "yes");
    } else if (!strcmp(a, "-sprankilog")) {
      /* the sampling interval of the reference's in-memory rank table: no effect on
         the file, but a value >= 0 switches the rank sort on (eis-bwtseq-param.c:98-100) */
      if (i + 1 >= argc) return pfail(err, errlen, "missing argument to option \"%s\"", a);
      if (atoi(argv[++i]) >= 0) sprank = 1;
    } else if (!strcmp(a, "-ctxilog")) {
      /* gt_registerCtxMapOptions, src/match/eis-bwtseq-context-param.c:20-32: -1 the
         automatic interval, -2 no map */
      if (i + 1 >= argc) return pfail(err, errlen, "missing argument to option \"%s\"", a);
      ctxilog = atoi(argv[++i]);
      if (ctxilog < -2 || ctxilog > 63) return pfail(err, errlen, "argument to option \"%s\" must be an integer between -2 and 63", a);
    }
    else if (a[0] == '-') return pfail(err, errlen, "unknown option: %s (try -help)", a);
    else if (index != NULL) return pfail(err, errlen, "superfluous argument \"%s\"", a);
    else index = a;
  }
  if (index == NULL) return pfail(err, errlen, "missing argument%s", "");
  /* the option parser's minima (gt_option_new_uint_min, eis-blockcomp-param.c) */
  if (pp.block_size < 1) return pfail(err, errlen, "argument to option \"-%s\" must be an integer >= 1", "bsize");
  if (pp.bucket_blocks < 1) return pfail(err, errlen, "argument to option \"-%s\" must be an integer >= 1", "blbuck");
  pp.feature_toggles = gtamd_pck_default_toggles(pp.block_size, pp.bucket_blocks, pp.locate_interval, locbitmap)
                       | (sprank ? GTAMD_PCK_REVERSIBLY_SORTED : 0);

  snprintf(path, sizeof path, "%s.prj", index);
  if (prj_value(path, "totallength", &totallength) != 0 || prj_value(path, "longest", &longest) != 0)
    return pfail(err, errlen, "cannot read totallength / longest from file '%s'", path);
  (void) prj_value(path, "integersize", &integersize);
  if (integersize != 64)
    return pfail(err, errlen, "file '%s' describes tables of another integer size", path);
  /* the alphabet comes with the encoded sequence (the reference maps INDEX.esq) */
  if (gtamd_read_esq(index, &enc, &n, &protein, &ss, err, errlen) != 0) return -1;
  free(enc);
  if (n != totallength) return pfail(err, errlen, "INDEX.esq and INDEX.prj of '%s' disagree on the total length", index);
  snprintf(path, sizeof path, "%s.bwt", index);
  if ((bwt = read_whole(path, totallength + 1)) == NULL) {
    /* the reference would derive the BWT from .suf and .esq; this tool asks for the table */
    pfail(err, errlen, "cannot read the %s table of the project (run suffixerator with -bwt)", path);
    goto done;
  }
  if (pp.locate_interval) {   /* (-sprank without locate information stores nothing either) */
    snprintf(path, sizeof path, "%s.suf", index);
    if ((suf = read_whole(path, 8 * (totallength + 1))) == NULL) {
      pfail(err, errlen, "suffix array project %s does not hold required suffix array (.suf) "
            "and encoded sequence (.esq) information!", index);
      goto done;
    }
  }
  if ((pck = gtamd_pck_create(0)) == NULL ||
      gtamd_pck_build_host(pck, bwt, suf, totallength + 1, ss.numofchars, longest, &pp) != 0 ||
      gtamd_pck_get_info(pck, &info) != 0) {
    snprintf(err, errlen, "%s", gtamd_esa_last_error());
    goto done;
  }
  /* the context map is made beside the locate marks (addLocateInfo,
     src/match/eis-bwtseq-extinfo.c:473-476): none without locate information */
  if (ctxilog >= -1 && pp.locate_interval) {
    int used = 0;
    if (gtamd_pck_ctxmap_build_host(pck, suf, totallength + 1, ctxilog, &used) != 0) {
      snprintf(err, errlen, "%s", gtamd_esa_last_error());
      goto done;
    }
    if (write_ctxmap(pck, index, used, err, errlen) != 0) goto done;
  }
  snprintf(path, sizeof path, "%s.bdx", index);
  if ((fp = fopen(path, "wb")) == NULL) { pfail(err, errlen, "cannot open file '%s' for writing", path); goto done; }
  {
    const uint64_t chunk = 64u << 20;
    uint8_t *buf = malloc(chunk);
    if (buf == NULL) { pfail(err, errlen, "out of memory (%s)", "packedindex"); goto done; }
    for (uint64_t off = 0; off < info.file_bytes; off += chunk) {
      const uint64_t cnt = info.file_bytes - off < chunk ? info.file_bytes - off : chunk;
      if (gtamd_pck_image_copy(pck, buf, off, cnt) != 0) { snprintf(err, errlen, "%s", gtamd_esa_last_error()); free(buf); goto done; }
      if (fwrite(buf, 1, cnt, fp) != cnt) { pfail(err, errlen, "cannot write file '%s'", path); free(buf); goto done; }
    }
    free(buf);
  }
  if (verbose)
    printf("# %llu buckets of %u bits + %llu variable bits, %llu regions, %llu bytes; %.2f ms on the device\n",
           (unsigned long long) info.num_buckets, info.cw_bits, (unsigned long long) info.var_bits,
           (unsigned long long) info.num_regions, (unsigned long long) info.file_bytes, info.build_ms);
  rc = 0;
done:
  if (fp != NULL && fclose(fp) != 0 && rc == 0) rc = pfail(err, errlen, "cannot close file '%s'", path);
  gtamd_pck_destroy(pck);
  free(bwt); free(suf);
  return rc;
}

/* `gt packedindex mkctxmap [-ctxilog I] [-v] INDEX` (src/tools/gt_packedindex_mkctxmap.c:40-139):
   the context map of an existing project from its INDEX.suf */
int gtamd_packedindex_mkctxmap(int argc, const char **argv, char *err, size_t errlen)
{
  int ctxilog = -1, rc = -1, used = 0;
  const char *index = NULL;
  char path[4096];
  unsigned long long totallength;
  uint64_t *suf = NULL;
  gtamd_pck *pck = NULL;
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    if (!strcmp(a, "-ctxilog")) {
      if (i + 1 >= argc) return pfail(err, errlen, "missing argument to option \"%s\"", a);
      ctxilog = atoi(argv[++i]);
    } else if (!strcmp(a, "-v")) continue;
    else if (a[0] == '-') return pfail(err, errlen, "unknown option: %s (try -help)", a);
    else if (index != NULL) return pfail(err, errlen, "superfluous argument \"%s\"", a);
    else index = a;
  }
  if (index == NULL) return pfail(err, errlen, "missing argument%s", "");
  if (ctxilog < -1) return pfail(err, errlen, "argument to option \"%s\" must be an integer >= -1", "-ctxilog");
  snprintf(path, sizeof path, "%s.prj", index);
  if (prj_value(path, "totallength", &totallength) != 0)
    return pfail(err, errlen, "cannot read totallength from file '%s'", path);
  snprintf(path, sizeof path, "%s.suf", index);
  if ((suf = read_whole(path, 8 * (totallength + 1))) == NULL)
    return pfail(err, errlen, "The project %s does not contain sufficient information to regenerate the suffix array.", index);
  if ((pck = gtamd_pck_create(0)) == NULL ||
      gtamd_pck_ctxmap_build_host(pck, suf, totallength + 1, ctxilog, &used) != 0)
    snprintf(err, errlen, "%s", gtamd_esa_last_error());
  else
    rc = write_ctxmap(pck, index, used, err, errlen);
  gtamd_pck_destroy(pck);
  free(suf);
  return rc;
}
