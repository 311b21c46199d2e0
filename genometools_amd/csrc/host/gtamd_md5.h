/* gtamd_md5.h -- MD5 (RFC 1321), host layer internal */
#ifndef GTAMD_MD5_H
#define GTAMD_MD5_H
#include <stddef.h>
#include <stdint.h>
typedef struct { uint32_t h[4]; uint64_t len; uint8_t buf[64]; unsigned fill; } gtamd_md5;
void gtamd_md5_init(gtamd_md5 *s);
void gtamd_md5_update(gtamd_md5 *s, const uint8_t *p, size_t n);
void gtamd_md5_hex(gtamd_md5 *s, char out[33]);
#endif
