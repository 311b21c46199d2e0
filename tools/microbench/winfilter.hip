// Dev microbenchmark (not part of the product): variants of the rank-table filter
// (k_win_filter of csrc/esa_engine.hip) on synthetic data of the 3 Gbp shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/winfilter tools/microbench/winfilter.hip && /tmp/winfilter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64; typedef uint8_t u8;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ inline u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
constexpr int NB = 31;                       // positions of 31 bits: a bijection by an odd multiplier
constexpr u64 N = 1ull << NB;
constexpr int FB = 13;
constexpr u32 NWIN = (u32) (N >> FB);
constexpr u32 NWW = NWIN / 32;

__global__ void k_init_sa(u32 *sa) {
  const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
  if (i < N) sa[i] = (u32) ((i * 2654435761ull + 12345ull) & (N - 1));
}
__global__ void k_init_tie(u64 *tie, u32 *carry, u64 nwords) {
  const u64 w = (u64) blockIdx.x * 256 + threadIdx.x;
  if (w >= nwords) return;
  const u64 h = mix64(w);
  tie[w] = (h & 7) == 0 ? (h >> 8) & (h >> 20) : 0ull;     // sparse ties
  carry[w] = (u32) (w * 64);
}

template <int OP> __device__ __forceinline__ u32 wave_scan_incl(u32 v) {
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);
  return v;
}
template <int THREADS>
__device__ __forceinline__ u32 block_scan_excl(u32 v, u32 *total, u32 *lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u32 inc = wave_scan_incl<0>(v);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  u32 carry = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < THREADS / 64; i++) { u32 s = lds[i]; if (i < w) carry += s; tot += s; }
  __syncthreads();
  *total = tot;
  return carry + (u32) __builtin_amdgcn_update_dpp(0, (int) inc, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 group_head(const u64 *tiebits, const u32 *carry, u64 i) {
  const u64 w = i >> 6;
  const int b = (int) (i & 63);
  const u64 below = b == 63 ? ~0ull : ((2ull << b) - 1ull);
  const u64 z = ~tiebits[w] & below;
  return z ? (u32) (w * 64 + (63 - __clzll((long long) z))) : carry[w];
}
__device__ __forceinline__ void emit4(const u32 v4[4], u32 mq, u64 i0, const u64 *tiebits, const u32 *carry,
                                      const u32 *sel, const u32 *pref, u32 *fpos, u32 *fhead, u32 &o) {
  const u64 t = tiebits[i0 >> 6];
  u32 h = group_head(tiebits, carry, i0);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (k > 0 && !((t >> ((i0 + k) & 63)) & 1ull)) h = (u32) (i0 + k);
    if ((mq >> k) & 1u) {
      const u32 v = v4[k], w = v >> FB;
      const u32 d = pref[w >> 5] + (u32) __popc(sel[w >> 5] & ((1u << (w & 31)) - 1u));
      fpos[o] = (d << FB) | (v & ((1u << FB) - 1u));
      fhead[o] = h;
      o++;
    }
  }
}

// ---- variant A: the engine's kernel (1024 threads, ITER spans, one atomic, reload)
template <int ITER, bool LSEL, bool OCC8>
__global__ __launch_bounds__(1024) void k_A(const u32 *__restrict__ sa, const u32 *__restrict__ sel,
    const u32 *__restrict__ pref, const u64 *__restrict__ tiebits, const u32 *__restrict__ carry,
    u32 *__restrict__ fpos, u32 *__restrict__ fhead, u32 *cursor) {
  extern __shared__ u32 s_sel[];
  __shared__ u32 s_scan[16];
  __shared__ u32 s_base;
  if (LSEL) { for (u32 i = threadIdx.x; i < NWW; i += 1024) s_sel[i] = sel[i]; __syncthreads(); }
  const u32 *selw = LSEL ? s_sel : sel;
  const u64 wg_first = (u64) blockIdx.x * ITER * 16384;
  u32 masks[ITER];
  u32 mine = 0;
#pragma unroll
  for (int it = 0; it < ITER; it++) {
    const u64 first = wg_first + (u64) it * 16384;
    uint4 p[4];
#pragma unroll
    for (int q = 0; q < 4; q++) p[q] = *reinterpret_cast<const uint4 *>(sa + first + q * 4096 + threadIdx.x * 4);
    u32 m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const u32 e[4] = {p[q].x, p[q].y, p[q].z, p[q].w};
#pragma unroll
      for (int k = 0; k < 4; k++) { const u32 w = e[k] >> FB; m |= ((selw[w >> 5] >> (w & 31)) & 1u) << (4 * q + k); }
    }
    masks[it] = m;
    mine += __popc(m);
  }
  u32 tot;
  u32 o = block_scan_excl<1024>(mine, &tot, s_scan);
  if (threadIdx.x == 0) s_base = tot ? atomicAdd(cursor, tot) : 0u;
  __syncthreads();
  o += s_base;
#pragma unroll
  for (int it = 0; it < ITER; it++) {
    if (masks[it] == 0) continue;
    const u64 first = wg_first + (u64) it * 16384;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const u32 mq = (masks[it] >> (4 * q)) & 15u;
      if (mq == 0) continue;
      const u64 i0 = first + q * 4096 + threadIdx.x * 4;
      const uint4 x = *reinterpret_cast<const uint4 *>(sa + i0);
      const u32 v4[4] = {x.x, x.y, x.z, x.w};
      emit4(v4, mq, i0, tiebits, carry, selw, pref, fpos, fhead, o);
    }
  }
}

// ---- variant B: two kernels.  B1: masks (16 bits per thread of 16 entries) and counts per tile of
// TB1*16 entries; B2: the tiles' entries, dense
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_B1(const u32 *__restrict__ sa, const u32 *__restrict__ sel,
                                                 unsigned short *__restrict__ masks, u32 *__restrict__ tilecnt) {
  __shared__ u32 s_scan[THREADS / 64];
  const u64 first = (u64) blockIdx.x * THREADS * 16;
  uint4 p[4];
#pragma unroll
  for (int q = 0; q < 4; q++) p[q] = *reinterpret_cast<const uint4 *>(sa + first + q * (THREADS * 4) + threadIdx.x * 4);
  u32 m = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const u32 e[4] = {p[q].x, p[q].y, p[q].z, p[q].w};
#pragma unroll
    for (int k = 0; k < 4; k++) { const u32 w = e[k] >> FB; m |= ((sel[w >> 5] >> (w & 31)) & 1u) << (4 * q + k); }
  }
  masks[(u64) blockIdx.x * THREADS + threadIdx.x] = (unsigned short) m;
  u32 tot;
  (void) block_scan_excl<THREADS>((u32) __popc(m), &tot, s_scan);
  if (threadIdx.x == 0) tilecnt[blockIdx.x] = tot;
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_B2(const u32 *__restrict__ sa, const u32 *__restrict__ sel,
    const u32 *__restrict__ pref, const u64 *__restrict__ tiebits, const u32 *__restrict__ carry,
    const unsigned short *__restrict__ masks, const u32 *__restrict__ tileoff,
    u32 *__restrict__ fpos, u32 *__restrict__ fhead) {
  __shared__ u32 s_scan[THREADS / 64];
  const u64 first = (u64) blockIdx.x * THREADS * 16;
  const u32 m = masks[(u64) blockIdx.x * THREADS + threadIdx.x];
  u32 tot;
  u32 o = tileoff[blockIdx.x] + block_scan_excl<THREADS>((u32) __popc(m), &tot, s_scan);
  if (tot == 0) return;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const u32 mq = (m >> (4 * q)) & 15u;
    if (mq == 0) continue;
    const u64 i0 = first + q * (THREADS * 4) + threadIdx.x * 4;
    const uint4 x = *reinterpret_cast<const uint4 *>(sa + i0);
    const u32 v4[4] = {x.x, x.y, x.z, x.w};
    emit4(v4, mq, i0, tiebits, carry, sel, pref, fpos, fhead, o);
  }
}
// exclusive scan of tile counts (harness only: one workgroup)
__global__ __launch_bounds__(1024) void k_scan1(const u32 *cnt, u32 *off, u32 n, u32 *total) {
  __shared__ u32 s_scan[16];
  __shared__ u32 s_run;
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (u32 base = 0; base < n; base += 1024) {
    const u32 i = base + threadIdx.x;
    const u32 v = i < n ? cnt[i] : 0u;
    u32 tot;
    const u32 e = block_scan_excl<1024>(v, &tot, s_scan);
    if (i < n) off[i] = s_run + e;
    __syncthreads();
    if (threadIdx.x == 0) s_run += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_run;
}

// ---- variant C: one kernel, THREADS per workgroup, one span of THREADS*16 entries, entries kept
// in registers, one atomic per workgroup, bitmap from global memory
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_C(const u32 *__restrict__ sa, const u32 *__restrict__ sel,
    const u32 *__restrict__ pref, const u64 *__restrict__ tiebits, const u32 *__restrict__ carry,
    u32 *__restrict__ fpos, u32 *__restrict__ fhead, u32 *cursor) {
  __shared__ u32 s_scan[THREADS / 64];
  __shared__ u32 s_base;
  const u64 first = (u64) blockIdx.x * THREADS * 16;
  uint4 p[4];
#pragma unroll
  for (int q = 0; q < 4; q++) p[q] = *reinterpret_cast<const uint4 *>(sa + first + q * (THREADS * 4) + threadIdx.x * 4);
  u32 m = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const u32 e[4] = {p[q].x, p[q].y, p[q].z, p[q].w};
#pragma unroll
    for (int k = 0; k < 4; k++) { const u32 w = e[k] >> FB; m |= ((sel[w >> 5] >> (w & 31)) & 1u) << (4 * q + k); }
  }
  u32 tot;
  u32 o = block_scan_excl<THREADS>((u32) __popc(m), &tot, s_scan);
  if (threadIdx.x == 0) s_base = tot ? atomicAdd(cursor, tot) : 0u;
  __syncthreads();
  o += s_base;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const u32 mq = (m >> (4 * q)) & 15u;
    if (mq == 0) continue;
    const u64 i0 = first + q * (THREADS * 4) + threadIdx.x * 4;
    const u32 v4[4] = {p[q].x, p[q].y, p[q].z, p[q].w};
    emit4(v4, mq, i0, tiebits, carry, sel, pref, fpos, fhead, o);
  }
}

// ---- reference points: read the suffix array only / with the bitmap look-ups
__global__ __launch_bounds__(256) void k_read(const u32 *__restrict__ sa, const u32 *__restrict__ sel, u32 *out, int lookups) {
  const u64 first = (u64) blockIdx.x * 4096;
  uint4 p[4];
#pragma unroll
  for (int q = 0; q < 4; q++) p[q] = *reinterpret_cast<const uint4 *>(sa + first + q * 1024 + threadIdx.x * 4);
  u32 acc = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const u32 e[4] = {p[q].x, p[q].y, p[q].z, p[q].w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (lookups) { const u32 w = e[k] >> FB; acc += (sel[w >> 5] >> (w & 31)) & 1u; }
      else acc ^= e[k];
    }
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

// ---- variant D: entries in registers, the selected ones through a queue in LDS: the second
// phase has one entry per thread (no reload, dense stores, no idle lanes)
template <int ITER, int QC>
__global__ __launch_bounds__(1024) void k_D(const u32 *__restrict__ sa, const u32 *__restrict__ sel,
    const u32 *__restrict__ pref, const u64 *__restrict__ tiebits, const u32 *__restrict__ carry,
    u32 *__restrict__ fpos, u32 *__restrict__ fhead, u32 *cursor) {
  extern __shared__ u32 s_dyn[];
  __shared__ u32 s_scan[16];
  __shared__ u32 s_base;
  constexpr u32 NWW4 = NWW / 4;
  u32 *s_sel = s_dyn, *s_pref4 = s_dyn + NWW, *s_qpos = s_pref4 + NWW4, *s_qidx = s_qpos + QC;
  for (u32 i = threadIdx.x; i < NWW; i += 1024) s_sel[i] = sel[i];
  for (u32 i = threadIdx.x; i < NWW4; i += 1024) s_pref4[i] = pref[4 * i];
  __syncthreads();
  for (int it = 0; it < ITER; it++) {
    const u64 first = ((u64) blockIdx.x * ITER + it) * 16384;
    uint4 p[4];
#pragma unroll
    for (int q = 0; q < 4; q++) p[q] = *reinterpret_cast<const uint4 *>(sa + first + q * 4096 + threadIdx.x * 4);
    u32 m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const u32 e[4] = {p[q].x, p[q].y, p[q].z, p[q].w};
#pragma unroll
      for (int k = 0; k < 4; k++) { const u32 w = e[k] >> FB; m |= ((s_sel[w >> 5] >> (w & 31)) & 1u) << (4 * q + k); }
    }
    u32 tot;
    const u32 excl = block_scan_excl<1024>((u32) __popc(m), &tot, s_scan);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(cursor, tot) : 0u;
    for (u32 r0 = 0; r0 < tot; r0 += QC) {
      u32 o = excl - r0;          // (wraps for entries before this round: then o >= QC)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const u32 e[4] = {p[q].x, p[q].y, p[q].z, p[q].w};
#pragma unroll
        for (int k = 0; k < 4; k++)
          if ((m >> (4 * q + k)) & 1u) {
            if (o < (u32) QC) { s_qpos[o] = e[k]; s_qidx[o] = (u32) (q * 4096 + threadIdx.x * 4 + k); }
            o++;
          }
      }
      __syncthreads();
      const u32 nq = tot - r0 < (u32) QC ? tot - r0 : (u32) QC;
      const u32 base = s_base + r0;
      for (u32 j = threadIdx.x; j < nq; j += 1024) {
        const u32 v = s_qpos[j], w = v >> FB, x = w >> 5;
        const u64 i = first + s_qidx[j];
        const uint4 g = *reinterpret_cast<const uint4 *>(s_sel + (x & ~3u));
        const u32 low = (1u << (w & 31)) - 1u, jj = x & 3u;
        const u32 d = s_pref4[x >> 2] + (u32) __popc(g.x & (jj == 0 ? low : ~0u)) +
                      (u32) __popc(g.y & (jj == 1 ? low : (jj > 1 ? ~0u : 0u))) +
                      (u32) __popc(g.z & (jj == 2 ? low : (jj > 2 ? ~0u : 0u))) +
                      (u32) __popc(g.w & (jj == 3 ? low : 0u));
        fpos[base + j] = (d << FB) | (v & ((1u << FB) - 1u));
        fhead[base + j] = group_head(tiebits, carry, i);
      }
      __syncthreads();
    }
  }
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_read_lds(const u32 *__restrict__ sa, const u32 *__restrict__ sel, u32 *out, int iter) {
  extern __shared__ u32 s_dyn[];
  for (u32 i = threadIdx.x; i < NWW; i += THREADS) s_dyn[i] = sel[i];
  __syncthreads();
  u32 acc = 0;
  for (int it = 0; it < iter; it++) {
    const u64 first = ((u64) blockIdx.x * iter + it) * THREADS * 16;
    uint4 p[4];
#pragma unroll
    for (int q = 0; q < 4; q++) p[q] = *reinterpret_cast<const uint4 *>(sa + first + q * (THREADS * 4) + threadIdx.x * 4);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const u32 e[4] = {p[q].x, p[q].y, p[q].z, p[q].w};
#pragma unroll
      for (int k = 0; k < 4; k++) { const u32 w = e[k] >> FB; acc += (s_dyn[w >> 5] >> (w & 31)) & 1u; }
    }
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

int main() {
  u32 *sa, *sel, *pref, *carry, *fpos, *fhead, *cursor, *tilecnt, *tileoff;
  u64 *tie;
  unsigned short *masks;
  const u64 nwords = N / 64;
  CK(hipMalloc(&sa, N * 4)); CK(hipMalloc(&sel, NWW * 4 + 64)); CK(hipMalloc(&pref, NWW * 4 + 64));
  CK(hipMalloc(&tie, nwords * 8)); CK(hipMalloc(&carry, nwords * 4));
  CK(hipMalloc(&fpos, N * 4 / 4)); CK(hipMalloc(&fhead, N * 4 / 4)); CK(hipMalloc(&cursor, 64));
  CK(hipMalloc(&masks, N / 16 * 2)); CK(hipMalloc(&tilecnt, (N / 1024 + 64) * 4)); CK(hipMalloc(&tileoff, (N / 1024 + 64) * 4));
  k_init_sa<<<(u32) (N / 256), 256>>>(sa);
  k_init_tie<<<(u32) (nwords / 256), 256>>>(tie, carry, nwords);
  std::vector<u32> hsel(NWW), hpref(NWW);
  u32 run = 0;
  for (u32 w = 0; w < NWW; w++) {
    u32 x = 0;
    for (int b = 0; b < 32; b++) if (mix64(0x77ull + w * 32 + b) % 100 < 7) x |= 1u << b;
    hsel[w] = x; hpref[w] = run; run += __builtin_popcount(x);
  }
  printf("N = %llu entries, %u of %u windows selected (%.1f %%) -> %llu entries\n", (unsigned long long) N, run, NWIN,
         100.0 * run / NWIN, (unsigned long long) run << FB);
  CK(hipMemcpy(sel, hsel.data(), NWW * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(pref, hpref.data(), NWW * 4, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto &&fn) {
    float best = 1e9f;
    u32 got = 0;
    for (int r = 0; r < 4; r++) {
      CK(hipMemset(cursor, 0, 8));
      CK(hipEventRecord(e0));
      fn();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      CK(hipMemcpy(&got, cursor, 4, hipMemcpyDeviceToHost));
    }
    printf("%-44s %7.3f ms  (x 3e9/N = %6.2f ms)  listed %u\n", name, best, best * 3e9 / (double) N, got);
  };
  timeit("read only", [&] { k_read<<<(u32) (N / 4096), 256>>>(sa, sel, cursor + 2, 0); });
  timeit("read + bitmap look-ups (global)", [&] { k_read<<<(u32) (N / 4096), 256>>>(sa, sel, cursor + 2, 1); });
  timeit("A  1024 thr, 8 spans, LDS bitmap, occ attr", [&] { k_A<8, true, true><<<(u32) (N / (8 * 16384)), 1024, NWW * 4>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  timeit("A  1024 thr, 4 spans, LDS bitmap", [&] { k_A<4, true, true><<<(u32) (N / (4 * 16384)), 1024, NWW * 4>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  timeit("A  1024 thr, 8 spans, global bitmap", [&] { k_A<8, false, true><<<(u32) (N / (8 * 16384)), 1024, 0>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  timeit("A  1024 thr, 1 span, global bitmap", [&] { k_A<1, false, true><<<(u32) (N / 16384), 1024, 0>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  timeit("B  two kernels, 256 thr", [&] {
    k_B1<256><<<(u32) (N / 4096), 256>>>(sa, sel, masks, tilecnt);
    k_scan1<<<1, 1024>>>(tilecnt, tileoff, (u32) (N / 4096), cursor);
    k_B2<256><<<(u32) (N / 4096), 256>>>(sa, sel, pref, tie, carry, masks, tileoff, fpos, fhead);
  });
  timeit("B1 alone (masks + counts), 256 thr", [&] { k_B1<256><<<(u32) (N / 4096), 256>>>(sa, sel, masks, tilecnt); });
  timeit("B2 alone, 256 thr", [&] { k_B2<256><<<(u32) (N / 4096), 256>>>(sa, sel, pref, tie, carry, masks, tileoff, fpos, fhead); });
  timeit("B  two kernels, 1024 thr", [&] {
    k_B1<1024><<<(u32) (N / 16384), 1024>>>(sa, sel, masks, tilecnt);
    k_scan1<<<1, 1024>>>(tilecnt, tileoff, (u32) (N / 16384), cursor);
    k_B2<1024><<<(u32) (N / 16384), 1024>>>(sa, sel, pref, tie, carry, masks, tileoff, fpos, fhead);
  });
  timeit("C  256 thr, registers, atomic per 4096", [&] { k_C<256><<<(u32) (N / 4096), 256>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  timeit("C  512 thr, registers, atomic per 8192", [&] { k_C<512><<<(u32) (N / 8192), 512>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  timeit("C  1024 thr, registers, atomic per 16384", [&] { k_C<1024><<<(u32) (N / 16384), 1024>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  timeit("read + LDS look-ups, 1024 thr x 4 spans", [&] { k_read_lds<1024><<<(u32) (N / (4 * 16384)), 1024, NWW * 4>>>(sa, sel, cursor + 2, 4); });
  timeit("read + LDS look-ups, 1024 thr x 16 spans", [&] { k_read_lds<1024><<<(u32) (N / (16 * 16384)), 1024, NWW * 4>>>(sa, sel, cursor + 2, 16); });
  timeit("read + LDS look-ups, 256 thr x 16 spans", [&] { k_read_lds<256><<<(u32) (N / (16 * 4096)), 256, NWW * 4>>>(sa, sel, cursor + 2, 16); });
  timeit("read + LDS look-ups, 512 thr x 16 spans", [&] { k_read_lds<512><<<(u32) (N / (16 * 8192)), 512, NWW * 4>>>(sa, sel, cursor + 2, 16); });
  timeit("D  queue, 4 spans, QC 2048", [&] { k_D<4, 2048><<<(u32) (N / (4 * 16384)), 1024, (NWW + NWW / 4 + 2 * 2048) * 4>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  timeit("D  queue, 8 spans, QC 2048", [&] { k_D<8, 2048><<<(u32) (N / (8 * 16384)), 1024, (NWW + NWW / 4 + 2 * 2048) * 4>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  timeit("D  queue, 16 spans, QC 2048", [&] { k_D<16, 2048><<<(u32) (N / (16 * 16384)), 1024, (NWW + NWW / 4 + 2 * 2048) * 4>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  timeit("D  queue, 8 spans, QC 4096", [&] { k_D<8, 4096><<<(u32) (N / (8 * 16384)), 1024, (NWW + NWW / 4 + 2 * 4096) * 4>>>(sa, sel, pref, tie, carry, fpos, fhead, cursor); });
  return 0;
}
