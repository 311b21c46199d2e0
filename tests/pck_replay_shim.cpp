// Test shim (CPU): the product's host-only replay of the reference's staging
// buffers (genometools_amd/csrc/esa_pck_replay.h) on an image in host memory.
#include <cstring>
#include "../genometools_amd/csrc/esa_pck_replay.h"

namespace {
struct Img { uint8_t *d; uint64_t len; };
int rd(void *u, uint64_t off, uint64_t n, uint8_t *dst) {
  Img *i = (Img *) u;
  if (off + n > i->len) return -1;
  memcpy(dst, i->d + off, n);
  return 0;
}
int wr(void *u, uint64_t off, uint64_t n, const uint8_t *src) {
  Img *i = (Img *) u;
  if (off + n > i->len) return -1;
  memcpy(i->d + off, src, n);
  return 0;
}
}  // namespace

extern "C" int pck_replay_run(uint8_t *image, uint64_t image_len, const PckTailGeom *g,
                              uint64_t var_bits_total, const uint64_t *tail_off, uint64_t ntail) {
  Img i = { image, image_len };
  std::vector<uint64_t> t(tail_off, tail_off + ntail);
  return pck_fix_stale_bits(*g, var_bits_total, t, rd, wr, &i);
}
