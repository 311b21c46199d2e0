/* ois_host.c -- -lossless: INDEX.ois, the table of "exceptions" that lets the
   reference reproduce the original characters (case, IUPAC letters) from the
   encoded sequence.  Restated from src/core/encseq.c:
     determine_original_subdist       :5275-5359  most frequent original
                                      character per symbol class, the classes'
                                      character lists, index of a character
                                      in its class
     countnumberofexceptionranges     :5361-5419
     the exception part of the fill functions, e.g. :2780-2815, 2844-2861
     assignoistabmapspecification     :1018-1078  file layout
   An exception is a symbol whose original character is not the most frequent
   one of its class; separators are skipped (they neither belong to a run of
   exceptions nor end one). */
#include "host_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned bits_for_value(uint64_t v)
{
  unsigned bits = 0;
  while (v) { bits++; v >>= 1; }
  return bits;
}

static int put_padded(FILE *fp, const void *p, uint64_t bytes)
{
  static const uint8_t zero[8] = {0};
  if (bytes == 0) return 0;
  if (fwrite(p, 1, bytes, fp) != bytes) return -1;
  if (bytes % 8 != 0 && fwrite(zero, 1, 8 - bytes % 8, fp) != 8 - bytes % 8) return -1;
  return 0;
}

int gtamd_write_ois(const char *indexname, const uint8_t *enc, const uint8_t *orig,
                    uint64_t n, const gtamd_alphabet *a, gtamd_encinfo *info,
                    char *err, size_t errlen)
{
  uint64_t classstart[255], maxima[256], numofallchars = 0, offset = 0, nexc = 0,
           nranges = 0, run = 0, fill = 0, mappos = 0, *mappositions = NULL, endidx;
  char allchars[256], maxchars[255];
  uint8_t subsymbolmap[256], classsize[256], *exceptions = NULL;
  uint32_t *positions = NULL, *rangelengths = NULL;
  unsigned maxsub = 0, bits;
  char path[4096];
  FILE *fp = NULL;
  int rc = -1;

  memset(classstart, 0, sizeof classstart); memset(maxima, 0, sizeof maxima);
  memset(maxchars, 0, sizeof maxchars); memset(subsymbolmap, 0, sizeof subsymbolmap);
  memset(classsize, 0, sizeof classsize); memset(allchars, 0, sizeof allchars);
  for (uint32_t c = 0; c < a->numofchars; c++) maxchars[c] = a->characters[c];
  maxchars[GTAMD_WILDCARD] = a->wildcardshow;
  /* most frequent original character per class: the first one wins a tie */
  for (int ch = 1; ch < 128; ch++) {
    const uint8_t code = a->symbolmap[ch];
    if (info->originaldistribution[ch] == 0 || (code >= a->numofchars && code != GTAMD_WILDCARD))
      continue;
    if (info->originaldistribution[ch] > maxima[code]) {
      maxima[code] = info->originaldistribution[ch];
      maxchars[code] = (char) ch;
    }
    numofallchars++;
  }
  /* the classes' characters in one string, class after class, wildcards last */
  for (uint32_t k = 0; k <= a->numofchars; k++) {
    const uint32_t code = k < a->numofchars ? k : GTAMD_WILDCARD;
    classstart[code] = offset;
    for (int ch = 1; ch < 128; ch++)
      if (info->originaldistribution[ch] > 0 && a->symbolmap[ch] == code) {
        subsymbolmap[ch] = classsize[code]++;
        allchars[offset++] = (char) ch;
      }
    if (classsize[code] > maxsub) maxsub = classsize[code];
  }
  bits = maxsub > 0 ? bits_for_value(maxsub - 1) : 0;
  /* first pass: how many exceptions, how many runs */
  for (uint64_t p = 0; p < n; p++) {
    if (enc[p] == GTAMD_SEPARATOR) continue;
    if ((char) orig[p] != maxchars[enc[p]]) { nexc++; run++; }
    else if (run > 0) { nranges++; run = 0; }
  }
  if (run > 0) nranges++;
  info->exceptioncharacters = nexc;
  info->realexceptionranges = nranges;
  exceptions = calloc((bits * nexc + 7) / 8 + 2, 1);
  positions = malloc(4 * (nranges + 1)); rangelengths = malloc(4 * (nranges + 1));
  mappositions = malloc(8 * (nranges + 1));
  if (exceptions == NULL || positions == NULL || rangelengths == NULL || mappositions == NULL) {
    snprintf(err, errlen, "out of memory while writing the exception table");
    goto done;
  }
  run = 0;
  for (uint64_t p = 0; p < n; p++) {
    if (enc[p] == GTAMD_SEPARATOR) continue;
    if ((char) orig[p] == maxchars[enc[p]]) {
      if (run > 0) { rangelengths[fill - 1] = (uint32_t) (run - 1); run = 0; }
      continue;
    }
    if (run == 0) {
      positions[fill] = (uint32_t) (p & 0xFFFFFFFFu);
      mappositions[fill] = mappos;
      fill++;
      run = 1;
    } else if (run == 0xFFFFFFFFu) {
      rangelengths[fill - 1] = 0xFFFFFFFFu;      /* full: the next one opens a new run */
      run = 0;
    } else run++;
    {
      const unsigned v = subsymbolmap[orig[p]];
      const uint64_t bit = (uint64_t) bits * mappos;
      if (bits > 0) {
        const unsigned shift = 16 - bits - (unsigned) (bit % 8);
        exceptions[bit / 8] |= (uint8_t) ((v << shift) >> 8);
        exceptions[bit / 8 + 1] |= (uint8_t) (v << shift);
      }
    }
    mappos++;
  }
  if (run > 0) rangelengths[fill - 1] = (uint32_t) (run - 1);
  if (fill != nranges) {
    snprintf(err, errlen, "exception runs do not add up");
    goto done;
  }
  snprintf(path, sizeof path, "%s.ois", indexname);
  if ((fp = fopen(path, "wb")) == NULL) {
    snprintf(err, errlen, "cannot open file '%s' for writing", path);
    goto done;
  }
  if (put_padded(fp, classstart, sizeof classstart) != 0 ||
      put_padded(fp, allchars, numofallchars) != 0 ||
      put_padded(fp, maxchars, 255) != 0 || put_padded(fp, subsymbolmap, 255) != 0 ||
      put_padded(fp, exceptions, (bits * nexc + 7) / 8) != 0)
    goto werr;
  if (nranges > 0) {
    /* 32-bit range table with lengths and map positions; one page per 2^32
       positions (addswtabletomapspectable, src/core/encseq.c:879-894) */
    const uint64_t numofpages = n / 0xFFFFFFFFull + 1;
    if (put_padded(fp, positions, 4 * nranges) != 0 ||
        put_padded(fp, rangelengths, 4 * nranges) != 0)
      goto werr;
    /* every run starts on the first page: n < 2^32 in this engine */
    endidx = nranges;
    for (uint64_t pg = 0; pg < numofpages; pg++)
      if (fwrite(&endidx, 8, 1, fp) != 1) goto werr;
    if (put_padded(fp, mappositions, 8 * nranges) != 0) goto werr;
  }
  if (fclose(fp) != 0) { fp = NULL; goto werr; }
  fp = NULL;
  rc = 0;
  goto done;
werr:
  snprintf(err, errlen, "cannot write file '%s'", path);
done:
  if (fp != NULL) fclose(fp);
  free(exceptions); free(positions); free(rangelengths); free(mappositions);
  return rc;
}
