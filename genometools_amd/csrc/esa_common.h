// esa_common.h -- shared declarations of the MI355X ESA engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <exception>
#include <new>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

#define WAVE 64

// error plumbing: every launcher returns 0 / -1 and leaves a message in the
// thread-local buffer that gtamd_esa_last_error() hands out (GtError style).
void gtamd_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                         \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      gtamd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                      __FILE__, __LINE__);                                    \
      return -1;                                                              \
    }                                                                         \
  } while (0)

#define TRY(...)                                                              \
  do {                                                                        \
    if ((__VA_ARGS__) != 0) return -1;                                        \
  } while (0)

static inline u64 div_up(u64 a, u64 b) { return (a + b - 1) / b; }

// Exception barrier of the C ABI.  The callers are C (GenomeTools' GtError
// convention: -1 / NULL + message); a std::bad_alloc or std::length_error from a
// host container -- e.g. one sized from a number read back from the device --
// must not cross an extern "C" frame, where it ends in std::terminate() and
// kills the caller.  Every entry point that can reach `new` or a container runs
// its body through this: GTAMD_ABI_BEGIN ... GTAMD_ABI_END(value on failure).
#define GTAMD_ABI_BEGIN try {
#define GTAMD_ABI_END(failvalue)                                                     \
  }                                                                                  \
  catch (const std::bad_alloc &) {                                                   \
    gtamd_set_error("%s: out of host memory", __func__);                             \
    return failvalue;                                                                \
  }                                                                                  \
  catch (const std::exception &e_) {                                                 \
    gtamd_set_error("%s: %s", __func__, e_.what());                                  \
    return failvalue;                                                                \
  }                                                                                  \
  catch (...) {                                                                      \
    gtamd_set_error("%s: unexpected exception", __func__);                           \
    return failvalue;                                                                \
  }

// ---- key layout ------------------------------------------------------------
// One 64-bit sort key per suffix:
//   [ prefix: KEY_SYMS symbols x BITS ][ dcode ][ unused ][ payload: symbol
//   before the suffix ]          (prefix in the top bits, dcode right below,
//   payload in the lowest bits; only prefix and dcode are sorted, as one
//   contiguous bit range)
// dcode = 0: no special among the first KEY_SYMS symbols;
// dcode = KEY_SYMS - d (1..KEY_SYMS-1): first special after d letters (the
//   prefix is padded with 1-bits behind the d letters, so the suffix sorts
//   behind every suffix that continues with letters; a shorter run of letters
//   gives a larger dcode and sorts later, equal runs fall back to the text
//   position through the stable sort) -- this is the order the reference gets
//   from "specials are unique symbols 256+position" (src/core/encseq.h:640);
// dcode = all ones: the suffix starts with a special (tail of the table).
// The payload rides along unsorted and yields the BWT symbol for free.
template <int BITS> struct KeyLayout;
template <> struct KeyLayout<2> {
  // 20 symbols separate all suffixes of a 3 Gbp random text but ~0.3 % (those
  // go through the cheap tie paths); every symbol less is bits the radix sort
  // does not have to move: 40 + 5 sorted bits = 6 passes instead of 8
  static constexpr int KEY_SYMS = 20;       // 40 bits
  static constexpr int DCODE_BITS = 5;
  static constexpr int PAYLOAD_BITS = 3;
  static constexpr int SYMS_PER_WORD = 32;
};
template <> struct KeyLayout<5> {
  // 20^10 >> 10^9: random ties stay rare enough for the direct tie path
  static constexpr int KEY_SYMS = 10;       // 50 bits
  static constexpr int DCODE_BITS = 4;
  static constexpr int PAYLOAD_BITS = 5;
  static constexpr int SYMS_PER_WORD = 12;  // 60 bits used, 4 low bits idle
};
