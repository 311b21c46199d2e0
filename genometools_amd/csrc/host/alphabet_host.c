/* alphabet_host.c -- the DNA and protein alphabets and alphabets from a symbol
   map file (-smap): src/core/alphabet.c:84-91,345-356,480-503 (built in) and
   :150-330 (read_symbolmap_from_lines). */
#include "host_internal.h"
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned bits_for_value(uint64_t v)
{
  unsigned bits = 0;
  while (v) { bits++; v >>= 1; }
  return bits;
}

void gtamd_alphabet_standard(gtamd_alphabet *a, int protein)
{
  memset(a, 0, sizeof *a);
  gtamd_symbolmap(a->symbolmap, protein);
  a->numofchars = protein ? 20 : 4;
  memcpy(a->characters, protein ? "LVIFKREDAGSTNQYWPHMC" : "acgt", a->numofchars);
  a->wildcardshow = protein ? 'X' : 'n';
  a->alphatype = protein ? 1 : 0;
  a->bitspersymbol = protein ? 5 : 3;
}

void gtamd_alphabet_free(gtamd_alphabet *a)
{
  if (a != NULL) { free(a->alphadef); a->alphadef = NULL; a->lengthofalphadef = 0; }
}

int gtamd_alphabet_from_text(const char *text, uint64_t len, const char *mapfile,
                             gtamd_alphabet *a, char *err, size_t errlen)
{
  /* lines: the characters of one symbol class, optionally a blank and the
     character to show for it; leading '#' lines are comments; the last line
     holds the wildcards */
  uint64_t pos = 0, linecount = 0, nlines = 0;
  unsigned mapsize = 0;
  int preamble = 1;
  memset(a, 0, sizeof *a);
  memset(a->symbolmap, 253, 256);
  a->alphatype = 2;
  for (uint64_t i = 0; i < len; i++) nlines += text[i] == '\n';
  if (len > 0 && text[len - 1] != '\n') nlines++;
  a->alphadef = malloc(len + 2);
  if (a->alphadef == NULL) { snprintf(err, errlen, "out of memory"); return -1; }
  while (pos < len) {
    const char *line = text + pos;
    uint64_t l = 0, column;
    int ignore = 0, blankfound = 0;
    char show;
    while (pos + l < len && line[l] != '\n') l++;
    /* the definition as stored in INDEX.esq: every line with a newline */
    memcpy(a->alphadef + a->lengthofalphadef, line, l);
    a->lengthofalphadef += l;
    a->alphadef[a->lengthofalphadef++] = '\n';
    if (l > 0) {
      if (preamble) { if (line[0] == '#') ignore = 1; else preamble = 0; }
      if (!ignore) {
        for (column = 0; column < l; column++) {
          const unsigned char cc = (unsigned char) line[column];
          if (ispunct(cc) || isalnum(cc)) {
            if (a->symbolmap[cc] != 253) {
              snprintf(err, errlen, "cannot map symbol '%c' to %u: it is already mapped to %u",
                       cc, mapsize, (unsigned) a->symbolmap[cc]);
              goto fail;
            }
            a->symbolmap[cc] = (uint8_t) mapsize;
          } else if (cc == ' ') {
            blankfound = 1;
            break;
          } else {
            snprintf(err, errlen, "illegal character '%c' in line %llu of mapfile %s", cc,
                     (unsigned long long) linecount, mapfile);
            goto fail;
          }
        }
        if (blankfound) {
          const unsigned char nx = column + 1 < l ? (unsigned char) line[column + 1] : 0;
          if (nx == 0 || isspace(nx)) {
            snprintf(err, errlen, "illegal character '%c' at the end of line %llu in "
                     "mapfile %s", nx, (unsigned long long) linecount, mapfile);
            goto fail;
          }
          show = (char) nx;
        } else show = line[0];
        if (mapsize >= 64) {
          snprintf(err, errlen, "mapfile %s defines too many symbol classes", mapfile);
          goto fail;
        }
        if (linecount == nlines - 1) a->wildcardshow = show;
        else a->characters[mapsize] = show;
        mapsize++;
      }
    }
    pos += l + 1;
    linecount++;
  }
  if (mapsize < 2) {
    snprintf(err, errlen, "mapfile %s does not define an alphabet", mapfile);
    goto fail;
  }
  for (int c = 0; c < 256; c++)
    if (a->symbolmap[c] == mapsize - 1) a->symbolmap[c] = GTAMD_WILDCARD;
  a->numofchars = mapsize - 1;
  a->bitspersymbol = bits_for_value(mapsize);
  return 0;
fail:
  gtamd_alphabet_free(a);
  return -1;
}

int gtamd_alphabet_from_file(const char *path, gtamd_alphabet *a, char *err, size_t errlen)
{
  FILE *fp = fopen(path, "rb");
  char *text;
  long size;
  int rc;
  if (fp == NULL) {
    snprintf(err, errlen, "cannot open file '%s'", path);
    return -1;
  }
  if (fseek(fp, 0, SEEK_END) != 0 || (size = ftell(fp)) < 0 || fseek(fp, 0, SEEK_SET) != 0 ||
      (text = malloc((size_t) size + 1)) == NULL) {
    fclose(fp);
    snprintf(err, errlen, "cannot read file '%s'", path);
    return -1;
  }
  if (fread(text, 1, (size_t) size, fp) != (size_t) size) size = 0;
  fclose(fp);
  rc = gtamd_alphabet_from_text(text, (uint64_t) size, path, a, err, errlen);
  free(text);
  return rc;
}
