// esa_pck_replay.h -- host-only part of the packed-index builder (no HIP): the
// bits the reference leaves stale in the last bucket of INDEX.bdx.
//
// The reference assembles a bucket in a staging buffer that is allocated zeroed
// once and never cleared, writes the whole bytes to the file and moves the
// incomplete last byte to the front (updateIdxOutput,
// src/match/eis-blockcomp.c:1807-1886; finalizeIdxOutput :2420-2471).  Bits of
// the LAST bucket's record that are not stored explicitly (composition indices
// of blocks that do not exist, locate bits behind the end) and the unused bits
// of the final byte of both bit strings keep what an earlier bucket left at
// that place of the buffer.  pck_fix_stale_bits replays the buffer for the last
// records of both bit strings from the image's own bytes, tracking which bits
// are known, and patches the bytes the reference wrote last.  The image is
// reached through two callbacks, so the CPU tests run this code on images made
// by the oracle (tests/test_pck_replay.py).
#pragma once
#include <stdint.h>
#include <algorithm>
#include <utility>
#include <vector>

struct PckTailGeom {
  uint64_t N, nb;                 // table entries, buckets
  uint32_t L, B;                  // positions per bucket, block size
  uint32_t locint, loc_bitmap;
  uint32_t cw_bits, pre_comp_idx, pre_cw_ext, comp_idx_bits;
  uint64_t cw_data_pos, var_data_pos;   // byte positions of the two bit strings
};
typedef int (*pck_image_read_fn)(void *user, uint64_t offset, uint64_t count, uint8_t *dst);
typedef int (*pck_image_write_fn)(void *user, uint64_t offset, uint64_t count, const uint8_t *src);

namespace pck_replay {
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

struct StaleRec { u64 gbit, nbits; };    // record: start in the stream, bits the buffer position advances by

// stream: bytes of the bit string from byte `stream_byte0` on (host copy)
// recs: consecutive records, the last one is the last of the stream;
// explicit_last: bit ranges (relative to the record start) the last record stores
// out: bytes from byte (recs.back().gbit / 8) to the end of what the reference writes
// returns false when a needed bit is unknown (window too small)
inline bool replay_tail(const std::vector<u8> &stream, u64 stream_byte0, const std::vector<StaleRec> &recs,
                        const std::vector<std::pair<u64, u64>> &explicit_last, bool from_stream_start,
                        std::vector<u8> *out) {
  u64 maxbits = 16;
  for (const StaleRec &r : recs) maxbits = std::max(maxbits, (r.gbit & 7) + r.nbits + 16);
  std::vector<u8> bit((size_t) maxbits, 0), kn((size_t) maxbits, from_stream_start ? 1 : 0);
  auto stream_bit = [&](u64 gb) -> u8 {
    const u64 byte = (gb >> 3) - stream_byte0;
    return (stream[(size_t) byte] >> (7 - (gb & 7))) & 1u;
  };
  for (size_t j = 0; j < recs.size(); j++) {
    const StaleRec &r = recs[j];
    const u64 old = r.gbit & 7;
    const bool last = j + 1 == recs.size();
    // the front byte carries the true tail of the record before
    for (u64 q = 0; q < old; q++) { bit[(size_t) q] = stream_bit(r.gbit - old + q); kn[(size_t) q] = 1; }
    if (!last) {
      for (u64 q = 0; q < r.nbits; q++) { bit[(size_t) (old + q)] = stream_bit(r.gbit + q); kn[(size_t) (old + q)] = 1; }
    } else {
      for (const auto &rg : explicit_last)
        for (u64 q = rg.first; q < rg.first + rg.second; q++) {
          bit[(size_t) (old + q)] = stream_bit(r.gbit + q);
          kn[(size_t) (old + q)] = 1;
        }
    }
    const u64 end = old + r.nbits, nbytes = end / 8;
    if (last) {
      // the whole bytes, then (finalizeIdxOutput) the incomplete byte from the
      // front of the buffer after the move: the bits of buffer byte nbytes
      const u64 total_bits = (end % 8) ? (nbytes + 1) * 8 : nbytes * 8;
      out->assign((size_t) (total_bits / 8), 0);
      for (u64 q = 0; q < total_bits; q++) {
        if (!kn[(size_t) q]) return false;
        if (bit[(size_t) q]) (*out)[(size_t) (q >> 3)] |= (u8) (0x80u >> (q & 7));
      }
      return true;
    }
    if (end % 8)
      for (u64 q = 0; q < 8; q++) {
        bit[(size_t) q] = bit[(size_t) (nbytes * 8 + q)];
        kn[(size_t) q] = kn[(size_t) (nbytes * 8 + q)];
      }
  }
  return true;
}
}  // namespace pck_replay

// tail_off: bit offsets of the var parts of the last tail_off.size() buckets
// (the field in the cw record may have lost its high bits).  Returns 0, -1 when
// an image access fails, -2 when the replay needs more than the buckets given,
// -3 when the numbers handed in cannot be right: offsets that decrease or lie
// behind var_bits_total, a var part longer than a bucket can make it.  (The
// offsets come back from the device; used unchecked, one wrapped difference
// sizes a container with ~2^64 entries -- std::length_error)
constexpr uint64_t PCK_REPLAY_MAX_BITS_PER_POSITION = 192;   // permutation index + mark + rank, generous
inline int pck_fix_stale_bits(const PckTailGeom &g, uint64_t var_bits_total,
                              const std::vector<uint64_t> &tail_off, pck_image_read_fn rd,
                              pck_image_write_fn wr, void *user) {
  using namespace pck_replay;
  const u64 last = g.nb - 1;
  const u64 last_pos = last * g.L;
  const u32 len_last = (u32) (g.N - last_pos);
  const u32 nblk_last = (len_last + g.B - 1) / g.B;
  // bits by which the buffer position advances for the last cw record: the full
  // record when the locate callback runs (appendCallBackOutput sets the position
  // behind the extension bits), else up to the last composition index stored
  const u64 last_cw_adv = g.locint ? g.cw_bits : (u64) g.pre_comp_idx + (u64) nblk_last * g.comp_idx_bits;
  if (g.nb == 0 || g.L == 0 || g.B == 0 || g.cw_bits == 0 || last_pos > g.N ||
      tail_off.empty() || tail_off.size() > g.nb || g.var_data_pos < g.cw_data_pos)
    return -3;
  {
    // what one bucket's var part can hold at most
    const u64 max_rec = (u64) g.L * PCK_REPLAY_MAX_BITS_PER_POSITION + 256;
    for (size_t k = 0; k < tail_off.size(); k++) {
      const u64 next = k + 1 < tail_off.size() ? tail_off[k + 1] : var_bits_total;
      if (tail_off[k] > next || next - tail_off[k] > max_rec) return -3;
    }
  }
  for (u64 window = 64; ; window *= 4) {
    const u64 j0 = last + 1 > window ? last + 1 - window : 0;
    if (last + 1 - j0 > tail_off.size()) return -2;
    // ---- cw records
    const u64 cw_first_byte = (j0 * g.cw_bits) / 8;
    const u64 cw_end_bit = last * g.cw_bits + g.cw_bits;
    const u64 cw_bytes = (cw_end_bit + 7) / 8 + 1 - cw_first_byte;
    std::vector<u8> cws((size_t) cw_bytes + 8, 0);
    {
      const u64 avail = std::min<u64>(cw_bytes, g.var_data_pos - (g.cw_data_pos + cw_first_byte));
      if (rd(user, g.cw_data_pos + cw_first_byte, avail, cws.data()) != 0) return -1;
    }
    std::vector<StaleRec> recs;
    for (u64 j = j0; j <= last; j++) recs.push_back({j * g.cw_bits, j == last ? last_cw_adv : (u64) g.cw_bits});
    std::vector<std::pair<u64, u64>> ex;
    ex.push_back({0, (u64) g.pre_comp_idx + (u64) nblk_last * g.comp_idx_bits});
    if (g.locint && g.loc_bitmap) ex.push_back({g.pre_cw_ext, len_last});
    std::vector<u8> tail;
    const bool ok_cw = replay_tail(cws, cw_first_byte, recs, ex, j0 == 0, &tail);
    // ---- var parts
    std::vector<StaleRec> vrecs;
    for (u64 j = j0; j <= last; j++) vrecs.push_back({tail_off[(size_t) (tail_off.size() - 1 - (last - j))], 0});
    for (size_t k = 0; k < vrecs.size(); k++)
      vrecs[k].nbits = (k + 1 < vrecs.size() ? vrecs[k + 1].gbit : var_bits_total) - vrecs[k].gbit;
    const u64 var_first_byte = vrecs[0].gbit / 8;
    const u64 var_bytes = (var_bits_total + 7) / 8 - var_first_byte;
    std::vector<u8> vs((size_t) var_bytes + 8, 0), vtail;
    if (var_bytes && rd(user, g.var_data_pos + var_first_byte, var_bytes, vs.data()) != 0) return -1;
    std::vector<std::pair<u64, u64>> vex;
    vex.push_back({0, vrecs.back().nbits});
    const bool ok_var = replay_tail(vs, var_first_byte, vrecs, vex, j0 == 0, &vtail);
    if ((!ok_cw || !ok_var) && j0 > 0) continue;
    if (!ok_cw || !ok_var) return -2;
    if (!tail.empty() && wr(user, g.cw_data_pos + (last * g.cw_bits) / 8, tail.size(), tail.data()) != 0) return -1;
    if (!vtail.empty() && wr(user, g.var_data_pos + vrecs.back().gbit / 8, vtail.size(), vtail.data()) != 0) return -1;
    return 0;
  }
}
