/* gt-suffixerator-amd: command line entry, behaves like `gt suffixerator`
   (exit code 1 and "gt suffixerator: error: ..." on stderr, src/gt.c:48-52);
   `gt-suffixerator-amd mergeesa ...` is `gt dev mergeesa ...`,
   `gt-suffixerator-amd packedindex mkindex|trsuftab ...` is `gt packedindex ...` */
#include <stdio.h>
#include <string.h>
#include "gtamd_host.h"

int main(int argc, char **argv)
{
  char err[2048] = "";
  if (argc > 1 && !strcmp(argv[1], "mergeesa")) {
    if (gtamd_mergeesa(argc - 1, (const char **) argv + 1, err, sizeof err) != 0) {
      fprintf(stderr, "gt dev mergeesa: error: %s\n", err);
      return 1;
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "packedindex")) {
    if (argc > 2 && !strcmp(argv[2], "mkindex")) {
      if (gtamd_packedindex_mkindex(argc - 2, (const char **) argv + 2, err, sizeof err) != 0) {
        fprintf(stderr, "gt packedindex mkindex: error: %s\n", err);
        return 1;
      }
      return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "mkctxmap")) {
      if (gtamd_packedindex_mkctxmap(argc - 2, (const char **) argv + 2, err, sizeof err) != 0) {
        fprintf(stderr, "gt packedindex mkctxmap: error: %s\n", err);
        return 1;
      }
      return 0;
    }
    if (argc < 3 || strcmp(argv[2], "trsuftab")) {
      fprintf(stderr, "gt packedindex: error: tool mkindex, trsuftab or mkctxmap expected\n");
      return 1;
    }
    if (gtamd_packedindex_trsuftab(argc - 2, (const char **) argv + 2, err, sizeof err) != 0) {
      fprintf(stderr, "gt packedindex trsuftab: error: %s\n", err);
      return 1;
    }
    return 0;
  }
  if (gtamd_suffixerator(argc, (const char **) argv, err, sizeof err) != 0) {
    fprintf(stderr, "gt suffixerator: error: %s\n", err);
    return 1;
  }
  return 0;
}
