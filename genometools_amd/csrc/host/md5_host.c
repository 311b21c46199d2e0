/* md5_host.c -- MD5 (RFC 1321) for the .md5 table of the host layer: one
   digest per sequence over its upper-case decoded symbols, as the reference's
   encoder computes it (src/core/encseq_charproc.gen:52-92, md5 via
   src/core/md5_encoder.c).  Written from the RFC's description. */
#include <stdint.h>
#include <string.h>
#include "gtamd_md5.h"

static const uint32_t K[64] = {
  0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a,
  0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be,
  0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340,
  0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
  0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
  0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c,
  0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa,
  0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
  0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92,
  0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
  0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
static const uint8_t S[64] = {
  7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22,
  5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
  6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};

static uint32_t rol(uint32_t x, unsigned s) { return (x << s) | (x >> (32 - s)); }

static void md5_block(uint32_t h[4], const uint8_t *p)
{
  uint32_t m[16], a = h[0], b = h[1], c = h[2], d = h[3];
  for (int i = 0; i < 16; i++)
    m[i] = (uint32_t) p[4 * i] | ((uint32_t) p[4 * i + 1] << 8) |
           ((uint32_t) p[4 * i + 2] << 16) | ((uint32_t) p[4 * i + 3] << 24);
  for (int i = 0; i < 64; i++) {
    uint32_t f;
    int g;
    if (i < 16) { f = (b & c) | (~b & d); g = i; }
    else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
    else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
    else { f = c ^ (b | ~d); g = (7 * i) & 15; }
    const uint32_t t = d;
    d = c; c = b;
    b = b + rol(a + f + K[i] + m[g], S[i]);
    a = t;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d;
}

void gtamd_md5_init(gtamd_md5 *s)
{
  s->h[0] = 0x67452301; s->h[1] = 0xefcdab89; s->h[2] = 0x98badcfe; s->h[3] = 0x10325476;
  s->len = 0; s->fill = 0;
}

void gtamd_md5_update(gtamd_md5 *s, const uint8_t *p, size_t n)
{
  s->len += n;
  while (n > 0) {
    size_t take = 64 - s->fill < n ? 64 - s->fill : n;
    memcpy(s->buf + s->fill, p, take);
    s->fill += (unsigned) take; p += take; n -= take;
    if (s->fill == 64) { md5_block(s->h, s->buf); s->fill = 0; }
  }
}

void gtamd_md5_hex(gtamd_md5 *s, char out[33])
{
  static const char hexd[] = "0123456789abcdef";
  const uint64_t bits = s->len * 8;
  uint8_t pad[72] = {0x80}, lenb[8];
  const size_t padlen = (s->fill < 56 ? 56 : 120) - s->fill;
  for (int i = 0; i < 8; i++) lenb[i] = (uint8_t) (bits >> (8 * i));
  gtamd_md5_update(s, pad, padlen);
  gtamd_md5_update(s, lenb, 8);
  for (int i = 0; i < 16; i++) {
    const uint8_t byte = (uint8_t) (s->h[i / 4] >> (8 * (i % 4)));
    out[2 * i] = hexd[byte >> 4];
    out[2 * i + 1] = hexd[byte & 15];
  }
  out[32] = '\0';
}
