// esa_comm.hip -- the two transports that ship with the library for the
// collectives of a part build (include/gtamd_esa.h: gtamd_allgather_fn,
// gtamd_alltoallv_fn), so that a C caller -- the gt-suffixerator-amd tool,
// GenomeTools itself -- needs neither Python nor torch to use more than one GPU:
//
//  * THREADS: one process, one host thread per part (the reference's own model:
//    gt -j N, src/core/thread_api.h).  The parts' contexts live on the GPUs of the
//    node -- or all on one, which is how the tests run it.  allgather through
//    host memory; alltoallv by peer copies (hipMemcpyPeerAsync) that every part
//    PULLS onto its own stream from the send buffers the others have published.
//  * RCCL: one process per part, the usual launch of a multi-GPU job.  alltoallv
//    as grouped ncclSend / ncclRecv on the engine's stream, allgather through a
//    small device buffer.  librccl is bound at run time (dlopen): the library has
//    no link-time dependency on it and a process that never asks for this
//    transport never loads it.
#include <condition_variable>
#include <mutex>
#include <vector>
#include <dlfcn.h>
#include "esa_common.h"
#include "../../include/gtamd_esa.h"

struct NcclId { char internal[128]; };      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)

struct gtamd_comm {
  int kind;               // 0 threads, 1 rccl
  u32 numparts;
  // ---- threads
  std::mutex mu;
  std::condition_variable cv;
  u32 arrived;
  u64 generation;
  bool broken;
  std::vector<const void *> send;            // per part: what it has published
  std::vector<const u64 *> counts;
  std::vector<int> device;
  std::vector<u32> elem;
  std::vector<std::vector<u8>> slot;         // allgather contributions
  struct View { gtamd_comm *g; u32 part; int device; };
  std::vector<View> views;
  // ---- rccl
  void *lib;
  void *nccl;             // ncclComm_t
  u32 rank;
  int rdevice;
  u8 *d_stage;
  u64 stage_bytes;
  hipStream_t rstream;
  int (*p_ncclCommInitRank)(void **, int, NcclId, int);
  int (*p_ncclCommDestroy)(void *);
  int (*p_ncclGroupStart)();
  int (*p_ncclGroupEnd)();
  int (*p_ncclSend)(const void *, size_t, int, int, void *, hipStream_t);
  int (*p_ncclRecv)(void *, size_t, int, int, void *, hipStream_t);
  int (*p_ncclAllGather)(const void *, void *, size_t, int, void *, hipStream_t);
};

// ---------------------------------------------------------------------------
// threads
// ---------------------------------------------------------------------------
// all parts arrive, or the transport is broken (a part has left with an error):
// nobody waits for ever
static int th_barrier(gtamd_comm *g) {
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->broken) return -1;
  const u64 gen = g->generation;
  if (++g->arrived == g->numparts) {
    g->arrived = 0;
    g->generation++;
    g->cv.notify_all();
    return 0;
  }
  g->cv.wait(lk, [&] { return g->generation != gen || g->broken; });
  return g->broken ? -1 : 0;
}
static void th_break(gtamd_comm *g) {
  std::lock_guard<std::mutex> lk(g->mu);
  g->broken = true;
  g->cv.notify_all();
}

static int th_allgather(void *user, const void *send, void *recv, uint32_t bytes) {
  gtamd_comm::View *v = (gtamd_comm::View *) user;
  gtamd_comm *g = v->g;
  try {
    g->slot[v->part].assign((const u8 *) send, (const u8 *) send + bytes);
    if (th_barrier(g) != 0) return -1;
    for (u32 r = 0; r < g->numparts; r++) {
      if (g->slot[r].size() != bytes) { th_break(g); return -1; }
      if (bytes) memcpy((u8 *) recv + (size_t) r * bytes, g->slot[r].data(), bytes);
    }
    return th_barrier(g);     // (the slots are free again)
  } catch (...) {
    th_break(g);
    return -1;
  }
}

static int th_alltoallv(void *user, const void *send, const uint64_t *sendcounts, void *recv,
                        const uint64_t *recvcounts, uint32_t elem, void *stream) {
  gtamd_comm::View *v = (gtamd_comm::View *) user;
  gtamd_comm *g = v->g;
  hipStream_t st = (hipStream_t) stream;
  // what this part sends must be complete before anybody copies out of it
  if (hipSetDevice(v->device) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { th_break(g); return -1; }
  g->send[v->part] = send;
  g->counts[v->part] = sendcounts;
  g->device[v->part] = v->device;
  g->elem[v->part] = elem;
  if (th_barrier(g) != 0) return -1;
  // pull: the block every part has for this one, in part order
  int bad = 0;
  u64 roff = 0;
  for (u32 s = 0; s < g->numparts && !bad; s++) {
    const u64 *sc = g->counts[s];
    u64 soff = 0;
    for (u32 r = 0; r < v->part; r++) soff += sc[r];
    const u64 bytes = sc[v->part] * elem;
    if (g->elem[s] != elem || sc[v->part] != recvcounts[s]) { bad = 1; break; }
    if (bytes) {
      const u8 *src = (const u8 *) g->send[s] + soff * elem;
      u8 *dst = (u8 *) recv + roff;
      const hipError_t e = g->device[s] == v->device
                               ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st)
                               : hipMemcpyPeerAsync(dst, v->device, src, g->device[s], bytes, st);
      if (e != hipSuccess) bad = 1;
    }
    roff += bytes;
  }
  // the copies are done before the sources may reuse their buffers
  if (!bad && hipStreamSynchronize(st) != hipSuccess) bad = 1;
  if (bad) { th_break(g); return -1; }
  return th_barrier(g);
}

extern "C" gtamd_comm *gtamd_comm_threads_create(uint32_t numparts) {
  GTAMD_ABI_BEGIN
  if (numparts == 0 || numparts > 128) { gtamd_set_error("invalid number of parts %u (1..128)", numparts); return nullptr; }
  gtamd_comm *g = new gtamd_comm();
  g->kind = 0;
  g->numparts = numparts;
  g->arrived = 0; g->generation = 0; g->broken = false;
  g->send.assign(numparts, nullptr); g->counts.assign(numparts, nullptr);
  g->device.assign(numparts, 0); g->elem.assign(numparts, 0);
  g->slot.resize(numparts);
  g->views.resize(numparts);
  g->lib = nullptr; g->nccl = nullptr; g->d_stage = nullptr; g->stage_bytes = 0;
  return g;
  GTAMD_ABI_END(nullptr)
}

// ---------------------------------------------------------------------------
// RCCL
// ---------------------------------------------------------------------------
static int rc_allgather(void *user, const void *send, void *recv, uint32_t bytes) {
  gtamd_comm *g = (gtamd_comm *) user;
  const u64 need = (u64) (g->numparts + 1) * (bytes ? bytes : 1);
  if (hipSetDevice(g->rdevice) != hipSuccess) return -1;
  if (need > g->stage_bytes) {
    if (g->d_stage) (void) hipFree(g->d_stage);
    g->d_stage = nullptr; g->stage_bytes = 0;
    if (hipMalloc(&g->d_stage, need + 4096) != hipSuccess) return -1;
    g->stage_bytes = need + 4096;
  }
  if (bytes == 0) {   // (an agreement without payload still is a meeting point)
    u8 z = 0;
    if (hipMemcpyAsync(g->d_stage, &z, 1, hipMemcpyHostToDevice, g->rstream) != hipSuccess) return -1;
    if (g->p_ncclAllGather(g->d_stage, g->d_stage + 1, 1, 0 /* ncclInt8 */, g->nccl, g->rstream) != 0) return -1;
    return hipStreamSynchronize(g->rstream) == hipSuccess ? 0 : -1;
  }
  if (hipMemcpyAsync(g->d_stage, send, bytes, hipMemcpyHostToDevice, g->rstream) != hipSuccess) return -1;
  if (g->p_ncclAllGather(g->d_stage, g->d_stage + bytes, bytes, 0, g->nccl, g->rstream) != 0) return -1;
  if (hipMemcpyAsync(recv, g->d_stage + bytes, (size_t) bytes * g->numparts, hipMemcpyDeviceToHost,
                     g->rstream) != hipSuccess) return -1;
  return hipStreamSynchronize(g->rstream) == hipSuccess ? 0 : -1;
}

static int rc_alltoallv(void *user, const void *send, const uint64_t *sendcounts, void *recv,
                        const uint64_t *recvcounts, uint32_t elem, void *stream) {
  gtamd_comm *g = (gtamd_comm *) user;
  hipStream_t st = (hipStream_t) stream;
  // enqueued behind the engine's kernels on the engine's stream; what the engine
  // launches next on it sees the received data
  if (g->p_ncclGroupStart() != 0) return -1;
  u64 soff = 0, roff = 0;
  int bad = 0;
  for (u32 r = 0; r < g->numparts; r++) {
    const u64 sb = sendcounts[r] * elem, rb = recvcounts[r] * elem;
    if (r == g->rank) {
      if (sb != rb) bad = 1;
      else if (sb && hipMemcpyAsync((u8 *) recv + roff, (const u8 *) send + soff, sb,
                                    hipMemcpyDeviceToDevice, st) != hipSuccess) bad = 1;
    } else {
      if (sb && g->p_ncclSend((const u8 *) send + soff, sb, 0, (int) r, g->nccl, st) != 0) bad = 1;
      if (rb && g->p_ncclRecv((u8 *) recv + roff, rb, 0, (int) r, g->nccl, st) != 0) bad = 1;
    }
    soff += sb; roff += rb;
  }
  if (g->p_ncclGroupEnd() != 0) bad = 1;
  return bad ? -1 : 0;
}

static void *rccl_open() {
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char *n : names) {
    void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h != nullptr) return h;
  }
  return nullptr;
}

extern "C" int gtamd_comm_rccl_unique_id(uint8_t id[128]) {
  GTAMD_ABI_BEGIN
  void *h = rccl_open();
  if (h == nullptr) { gtamd_set_error("cannot load librccl: %s", dlerror()); return -1; }
  int (*get)(NcclId *) = (int (*)(NcclId *)) dlsym(h, "ncclGetUniqueId");
  NcclId nid;
  if (get == nullptr || get(&nid) != 0) { gtamd_set_error("ncclGetUniqueId failed"); return -1; }
  memcpy(id, nid.internal, 128);
  return 0;
  GTAMD_ABI_END(-1)
}

extern "C" gtamd_comm *gtamd_comm_rccl_create(const uint8_t id[128], uint32_t rank, uint32_t numparts,
                                              int device) {
  GTAMD_ABI_BEGIN
  if (numparts == 0 || numparts > 128 || rank >= numparts) { gtamd_set_error("invalid rank %u of %u", rank, numparts); return nullptr; }
  void *h = rccl_open();
  if (h == nullptr) { gtamd_set_error("cannot load librccl: %s", dlerror()); return nullptr; }
  gtamd_comm *g = new gtamd_comm();
  g->kind = 1; g->numparts = numparts; g->rank = rank; g->rdevice = device;
  g->arrived = 0; g->generation = 0; g->broken = false;
  g->lib = h; g->nccl = nullptr; g->d_stage = nullptr; g->stage_bytes = 0; g->rstream = nullptr;
  g->p_ncclCommInitRank = (int (*)(void **, int, NcclId, int)) dlsym(h, "ncclCommInitRank");
  g->p_ncclCommDestroy = (int (*)(void *)) dlsym(h, "ncclCommDestroy");
  g->p_ncclGroupStart = (int (*)()) dlsym(h, "ncclGroupStart");
  g->p_ncclGroupEnd = (int (*)()) dlsym(h, "ncclGroupEnd");
  g->p_ncclSend = (int (*)(const void *, size_t, int, int, void *, hipStream_t)) dlsym(h, "ncclSend");
  g->p_ncclRecv = (int (*)(void *, size_t, int, int, void *, hipStream_t)) dlsym(h, "ncclRecv");
  g->p_ncclAllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t)) dlsym(h, "ncclAllGather");
  NcclId nid;
  memcpy(nid.internal, id, 128);
  if (g->p_ncclCommInitRank == nullptr || g->p_ncclCommDestroy == nullptr || g->p_ncclGroupStart == nullptr ||
      g->p_ncclGroupEnd == nullptr || g->p_ncclSend == nullptr || g->p_ncclRecv == nullptr ||
      g->p_ncclAllGather == nullptr) {
    gtamd_set_error("librccl lacks an entry point this transport needs");
    delete g;
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&g->rstream, hipStreamNonBlocking) != hipSuccess ||
      g->p_ncclCommInitRank(&g->nccl, (int) numparts, nid, (int) rank) != 0) {
    gtamd_set_error("cannot create the RCCL communicator of rank %u of %u on device %d", rank, numparts, device);
    if (g->rstream) (void) hipStreamDestroy(g->rstream);
    delete g;
    return nullptr;
  }
  return g;
  GTAMD_ABI_END(nullptr)
}

// ---------------------------------------------------------------------------
extern "C" int gtamd_comm_attach(gtamd_comm *g, uint32_t part, gtamd_esa_ctx *ctx, int device) {
  GTAMD_ABI_BEGIN
  if (g == nullptr || ctx == nullptr || part >= g->numparts) { gtamd_set_error("invalid argument to gtamd_comm_attach"); return -1; }
  if (gtamd_esa_set_part(ctx, part, g->numparts) != 0) return -1;
  if (g->kind == 0) {
    g->views[part].g = g; g->views[part].part = part; g->views[part].device = device;
    if (gtamd_esa_set_comm(ctx, th_allgather, th_alltoallv, &g->views[part]) != 0) return -1;
    // (a run that fails on one part breaks the barrier for all)
    return gtamd_esa_set_comm_abort(ctx, [](void *u) { th_break((gtamd_comm *) u); }, g);
  }
  if (part != g->rank) { gtamd_set_error("this process is rank %u of the RCCL transport, not part %u", g->rank, part); return -1; }
  return gtamd_esa_set_comm(ctx, rc_allgather, rc_alltoallv, g);
  GTAMD_ABI_END(-1)
}

extern "C" void gtamd_comm_abort(gtamd_comm *g) {
  if (g != nullptr && g->kind == 0) th_break(g);
}

extern "C" void gtamd_comm_destroy(gtamd_comm *g) {
  if (g == nullptr) return;
  if (g->kind == 1) {
    (void) hipSetDevice(g->rdevice);
    if (g->nccl != nullptr) (void) g->p_ncclCommDestroy(g->nccl);
    if (g->d_stage != nullptr) (void) hipFree(g->d_stage);
    if (g->rstream != nullptr) (void) hipStreamDestroy(g->rstream);
  }
  delete g;
}
