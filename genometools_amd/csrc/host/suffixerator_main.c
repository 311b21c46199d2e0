/* gt-suffixerator-amd: command line entry, behaves like `gt suffixerator`
   (exit code 1 and "gt suffixerator: error: ..." on stderr, src/gt.c:48-52) */
#include <stdio.h>
#include "gtamd_host.h"

int main(int argc, char **argv)
{
  char err[2048] = "";
  if (gtamd_suffixerator(argc, (const char **) argv, err, sizeof err) != 0) {
    fprintf(stderr, "gt suffixerator: error: %s\n", err);
    return 1;
  }
  return 0;
}
