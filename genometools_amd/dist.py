"""Sharding the ESA build over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo"
in the CPU tests).  The suffix array is cut into `world` lexicographic ranges
of (almost) equal size -- the reference's `-parts` mechanism
(src/match/sfx-partssuf.c:172-347) -- and rank r builds slice r of every table
from the replicated packed sequence.  Per-rank work falls with the number of
ranks: a rank makes the sort keys of its 1/world tile of the text and sends
the (key, position) pairs to the owners of their key ranges (alltoallv, 12 B
per suffix); the rank table of the prefix-doubling rounds is cut by text
position, so a round is an alltoallv of 4-byte queries to the tile owners, one
of answers back, and one of the new ranks of refined suffixes; plus a few
small allgathers (range cuts, counts, slice border keys, round termination).
The engine calls back into `TorchComm` for these collectives; it never sees a
torch type.  On the nccl backend the exchanges are enqueued on the engine's
own HIP stream (no host synchronisation per exchange).
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_uint32)
ALLTOALLV_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p,
                                ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32,
                                ctypes.c_void_p)


class _DevMem:
    """expose raw device memory to torch without copying"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1",
                                         "data": (ptr, False), "version": 2}


def _device_bytes(ptr, nbytes, device):
    if nbytes == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(_DevMem(ptr, nbytes), device=device)


def _host_bytes(ptr, nbytes):
    buf = (ctypes.c_uint8 * nbytes).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.uint8))


class TorchComm:
    """the two collectives of include/gtamd_esa.h on a torch process group.

    `device` is the torch device the engine's buffers live on ("cuda:k"), or
    "cpu" when the buffers are host memory (gloo tests)."""

    def __init__(self, device, group=None):
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # host transport for device buffers when the backend cannot move them
        self.staged = self.device.type != "cpu" and dist.get_backend(group) == "gloo"
        self.bytes_exchanged = 0
        self.calls = 0
        # keep the ctypes thunks alive as long as the object lives
        self.allgather_cb = ALLGATHER_FN(self._allgather)
        self.alltoallv_cb = ALLTOALLV_FN(self._alltoallv)

    # host memory in, host memory out
    def _allgather(self, user, send, recv, nbytes):
        try:
            mine = _host_bytes(send, nbytes)
            out = _host_bytes(recv, nbytes * self.world)
            if self.device.type == "cpu" or self.staged:
                self._gloo_allgather(out, mine)
            else:
                d_in = mine.to(self.device)
                d_out = torch.empty(nbytes * self.world, dtype=torch.uint8, device=self.device)
                dist.all_gather_into_tensor(d_out, d_in, group=self.group)
                out.copy_(d_out.cpu())
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("allgather callback failed:", repr(e), flush=True)
            return -1

    def _p2p(self, tmp, ins, rc, sc):
        # gloo has no alltoall: pairwise non-blocking send/recv
        reqs = []
        for r in range(self.world):
            if r == self.rank:
                tmp[r].copy_(ins[r])
                continue
            if rc[r]:
                reqs.append(dist.irecv(tmp[r], src=r, group=self.group))
            if sc[r]:
                reqs.append(dist.isend(ins[r].contiguous(), dst=r, group=self.group))
        for q in reqs:
            q.wait()

    def _gloo_allgather(self, out, mine):
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine.clone(), group=self.group)
        out.copy_(torch.cat(parts))

    # device memory in, device memory out; ordered on the engine's stream
    def _alltoallv(self, user, send, sendcounts, recv, recvcounts, elem, stream):
        try:
            sc = [int(sendcounts[r]) * elem for r in range(self.world)]
            rc = [int(recvcounts[r]) * elem for r in range(self.world)]
            if self.device.type == "cpu":
                t_in = _host_bytes(send, sum(sc)) if sum(sc) else torch.empty(0, dtype=torch.uint8)
                t_out = _host_bytes(recv, sum(rc)) if sum(rc) else torch.empty(0, dtype=torch.uint8)
                outs = list(t_out.split(rc)) if sum(rc) else [torch.empty(0, dtype=torch.uint8) for _ in rc]
                ins = [x.clone() for x in t_in.split(sc)] if sum(sc) else [torch.empty(0, dtype=torch.uint8) for _ in sc]
                tmp = [torch.empty(n, dtype=torch.uint8) for n in rc]
                self._p2p(tmp, ins, rc, sc)
                for o, t in zip(outs, tmp):
                    o.copy_(t)
            elif self.staged:
                # device buffers, host transport (gloo): used to rehearse the
                # multi-process path on a box with a single GPU.  The engine's
                # stream is synchronised here, the data is back before return.
                torch.cuda.ExternalStream(stream, device=self.device).synchronize()
                t_in = _device_bytes(send, sum(sc), self.device).cpu()
                ins = list(t_in.split(sc)) if sum(sc) else [torch.empty(0, dtype=torch.uint8) for _ in sc]
                tmp = [torch.empty(n, dtype=torch.uint8) for n in rc]
                self._p2p(tmp, ins, rc, sc)
                if sum(rc):
                    _device_bytes(recv, sum(rc), self.device).copy_(torch.cat(tmp))
                torch.cuda.synchronize(self.device)
            else:
                # RCCL: enqueued behind the engine's kernels on the engine's own
                # stream; the kernels the engine launches next wait for it there
                t_in = _device_bytes(send, sum(sc), self.device)
                t_out = _device_bytes(recv, sum(rc), self.device)
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)):
                    dist.all_to_all_single(t_out, t_in, output_split_sizes=rc,
                                           input_split_sizes=sc, group=self.group)
            self.bytes_exchanged += sum(sc) - sc[self.rank]
            self.calls += 1
            return 0
        except Exception as e:
            print("alltoallv callback failed:", repr(e), flush=True)
            return -1

    def attach(self, engine):
        """make `engine` (genometools_amd.esa.EsaEngine) build part `rank` of
        `world` with this object as the transport"""
        engine.set_part(self.rank, self.world, self)


def combine_stats(stats, device, group=None):
    """whole-table statistics from the per-part ones (see gtamd_esa_set_part)"""
    dev = torch.device(device)
    if dev.type != "cpu" and dist.get_backend(group) == "gloo":
        dev = torch.device("cpu")
    s = torch.tensor([stats["lcptabsum"], stats["largelcpvalues"], stats["longest"],
                      stats["tied_suffixes"], stats.get("pair_suffixes", 0)],
                     dtype=torch.int64, device=dev)
    m = torch.tensor([stats["maxbranchdepth"], stats["refine_rounds"]],
                     dtype=torch.int64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    out = dict(stats)
    (out["lcptabsum"], out["largelcpvalues"], out["longest"], out["tied_suffixes"],
     out["pair_suffixes"]) = [int(x) for x in s.tolist()]
    out["maxbranchdepth"], out["refine_rounds"] = [int(x) for x in m.tolist()]
    return out
