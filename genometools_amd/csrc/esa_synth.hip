// esa_synth.hip -- synthetic sequences on the device (bench / parity inputs)
// and host-only arithmetic of the C ABI.  The models are specified in
// genometools_amd/synth.py; both sides must produce identical bytes
// (tests/test_synth.py).
#include "../../include/gtamd_esa.h"
#include "esa_common.h"
#include <limits.h>

namespace {

constexpr u64 GOLD = 0x9E3779B97F4A7C15ull;

__host__ __device__ __forceinline__ u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ u64 stream_key(u64 seed, u64 stream) {
  return mix64(seed + stream * GOLD + 0x1234567ull);
}
__device__ __forceinline__ u64 hx(u64 key, u64 x) {
  return mix64(key + (x + 1) * GOLD);
}
__device__ __forceinline__ u32 bg(u64 k0, u64 q) {
  return (u32) (hx(k0, q >> 5) >> (2 * (q & 31))) & 3u;
}

struct SynthParams {
  u64 key[6];
  u64 n, nblocks;
  u64 sep[23];
  int nsep;
  // MODEL_REPEAT_HEAVY: share of duplicated blocks (of 65536), satellite arrays
  u32 dup_t;
  int nsat;
  u64 sat_start[4], sat_len[4], sat_per[4];
};

__constant__ u32 c_mut_thr[4] = {0u, 4294967u, 42949673u, 214748365u};
__constant__ u32 c_nrun_len[16] = {20, 50, 90, 140, 200, 280, 370, 480, 620,
                                   800, 1000, 1300, 1700, 2300, 3300, 5200};
__constant__ u32 c_prot_cum[20] = {6340, 10848, 14740, 17273, 21106, 24735,
                                   29165, 32742, 38156, 42796, 47101, 50606,
                                   53270, 55849, 57766, 58474, 61559, 63049,
                                   64637, 65536};
const u32 h_chrom_cum[23] = {5268, 10409, 14615, 18652, 22518, 26151, 29528,
                             32609, 35540, 38387, 41255, 44080, 46502, 48775,
                             50942, 52854, 54617, 56316, 57570, 58929, 59928,
                             61011, 64325};

__device__ __forceinline__ u8 sym_uniform(const SynthParams &P, u64 p) {
  return (u8) bg(P.key[0], p);
}

__device__ u8 sym_humanlike(const SynthParams &P, u64 p) {
  for (int i = 0; i < P.nsep; i++)
    if (P.sep[i] == p) return (u8) GTAMD_SEPARATOR;
  if ((hx(P.key[5], p) & 0x3FFFFull) == 0) return (u8) GTAMD_WILDCARD;
  const u64 b = p >> 13, o = p & 8191;
  const u64 hn = hx(P.key[4], b);
  if ((hn & 0xFFFFull) < 10748ull) {
    const u64 ns = (hn >> 16) & 8191ull, nl = c_nrun_len[(hn >> 32) & 15];
    if (o >= ns && o < ns + nl) return (u8) GTAMD_WILDCARD;
  }
  for (int k = 0; k < P.nsat; k++)
    if (p >= P.sat_start[k] && p < P.sat_start[k] + P.sat_len[k])
      return (u8) bg(P.key[0], (1ull << 40) + 4096ull * k + (p - P.sat_start[k]) % P.sat_per[k]);
  const u32 kind = (u32) (hx(P.key[1], b) & 0xFFFFull);
  if (kind < P.dup_t) {
    const u64 h2 = hx(P.key[2], b);
    const u64 src = (h2 & 7) == 0 ? ((h2 >> 8) & 15) : ((h2 >> 8) % P.nblocks);
    const u32 thr = c_mut_thr[(h2 >> 3) & 3];
    u32 c = bg(P.key[0], (src << 13) | o);
    const u64 hm = hx(P.key[3], p);
    if ((u32) (hm & 0xFFFFFFFFull) < thr) c = (c + 1 + (u32) ((hm >> 32) % 3)) & 3u;
    return (u8) c;
  }
  if (kind < P.dup_t + 64u) {
    const u64 h2 = hx(P.key[2], b);
    const u64 per = 1 + ((h2 >> 40) % 6), so = (h2 >> 8) & 4095,
              tl = 64 + ((h2 >> 20) & 2047);
    if (o >= so && o < so + tl)
      return (u8) bg(P.key[0], (b << 13) + so + ((o - so) % per));
  }
  return (u8) bg(P.key[0], p);
}

__device__ __forceinline__ bool prot_raw_sep(const SynthParams &P, u64 q) {
  return (((hx(P.key[0], q) >> 40) & 0xFFFFull) < 198ull) && q > 0 &&
         q + 1 < P.n;
}
__device__ u8 sym_protein(const SynthParams &P, u64 p) {
  if (prot_raw_sep(P, p) && !prot_raw_sep(P, p - 1)) return (u8) GTAMD_SEPARATOR;
  const u64 h = hx(P.key[0], p);
  if (((h >> 20) & 0x1FFFull) == 0) return (u8) GTAMD_WILDCARD;
  const u32 r = (u32) (h & 0xFFFFull);
  u32 c = 0;
  while (c_prot_cum[c] <= r) c++;
  return (u8) c;
}

template <int MODEL>
__global__ __launch_bounds__(256) void k_synth(SynthParams P, u8 *__restrict__ dst) {
  // 4 consecutive symbols per thread -> one 32-bit store
  const u64 p0 = ((u64) blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= P.n) return;
  u32 word = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const u64 p = p0 + c;
    u8 s = 0;
    if (p < P.n)
      s = MODEL == 0 ? sym_uniform(P, p)
                     : (MODEL == 1 ? sym_humanlike(P, p) : sym_protein(P, p));   // (1: also the
                                                      // repeat-heavy parameters)
    word |= (u32) s << (8 * c);
  }
  if (p0 + 4 <= P.n) *reinterpret_cast<u32 *>(dst + p0) = word;
  else
    for (int c = 0; c < 4 && p0 + c < P.n; c++) dst[p0 + c] = (u8) (word >> (8 * c));
}

}  // namespace

extern "C" int gtamd_synth_bytes(int device, int model, uint64_t seed,
                                 uint64_t n, uint8_t *dst_device) {
  GTAMD_ABI_BEGIN
  if (model < 0 || model > 3) {
    gtamd_set_error("unknown synthetic model %d", model);
    return -1;
  }
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(device));
  SynthParams P;
  memset(&P, 0, sizeof P);
  for (int s = 0; s < 6; s++) P.key[s] = stream_key(seed, (u64) s);
  P.n = n;
  P.nblocks = (n + 8191) >> 13;
  P.nsep = 0;
  P.dup_t = model == 3 ? 32768u : 6554u;
  if (model == 3) {
    const u64 sat_len[4] = {100000, 250000, 500000, 1000000};
    const u64 sat_per[4] = {171, 5, 42, 68};
    P.nsat = n >= (1ull << 24) ? 4 : (n >= (1ull << 16) ? 2 : 0);
    for (int k = 0; k < P.nsat; k++) {
      P.sat_start[k] = (u64) (((unsigned __int128) n * (2 * k + 1)) / 9);
      P.sat_len[k] = n / 40 < sat_len[k] ? n / 40 : sat_len[k];
      P.sat_per[k] = sat_per[k];
    }
  }
  if ((model == 1 || model == 3) && n >= 65536) {
    for (int i = 0; i < 23; i++)
      P.sep[i] = (u64) (((unsigned __int128) n * h_chrom_cum[i]) >> 16);
    P.nsep = 23;
  }
  const u32 grid = (u32) div_up(div_up(n, 4), 256);
  if (model == 0) k_synth<0><<<grid, 256>>>(P, dst_device);
  else if (model == 1 || model == 3) k_synth<1><<<grid, 256>>>(P, dst_device);
  else k_synth<2><<<grid, 256>>>(P, dst_device);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return 0;
  GTAMD_ABI_END(-1)
}

// ---------------------------------------------------------------------------
// gt_recommendedprefixlength, src/match/sfx-apfxlen.c:49-109 with the table
// size model of src/match/bcktab.c:239-324 (withspecialsuffixes = true) and
// the cap gt_maxbasepower of src/match/initbasepower.c:23-34.  The engine does
// not bucket by this prefix; the number goes into .prj and masks averagelcp.
// ---------------------------------------------------------------------------
static u64 ipow64(u64 b, unsigned e) {
  u64 r = 1;
  while (e--) r *= b;
  return r;
}

extern "C" uint32_t gtamd_recommended_prefixlength(uint32_t numofchars,
                                                   uint64_t n) {
  GTAMD_ABI_BEGIN
  if (numofchars < 2) return 1;
  const u64 w = n + 1 <= (u64) UINT_MAX ? 4 : 8;
  // largest exponent that keeps numofchars^k below the code range
  unsigned mbp = 0;
  {
    const u64 minfailure = ~0ull / numofchars;
    u64 thepower = 1;
    for (mbp = 0; thepower < minfailure; mbp++) thepower *= numofchars;
  }
  unsigned k;
  for (k = 1; k <= mbp + 1; k++) {
    u64 size = w * (ipow64(numofchars, k) + 1) + w * ipow64(numofchars, k - 1);
    if (k > 2) {
      u64 counters = 0;
      for (unsigned idx = 1; idx < k - 1; idx++) counters += ipow64(numofchars, idx);
      size += w * counters;
    }
    if ((double) size / 0.25 > (double) n) break;
  }
  k--;
  if (k == 0) return 1;
  return mbp >= 1 && mbp < k ? mbp : k;
  GTAMD_ABI_END(0)
}
