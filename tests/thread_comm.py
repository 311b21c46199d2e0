"""In-process stand-in for the collectives of a part build: R engine contexts
on ONE device, one Python thread each (ctypes releases the GIL while the engine
runs), exchanging through shared state, barriers and device-to-device copies.
Lets the sharded path be checked on a single-GPU box.  Test infrastructure."""
import ctypes
import threading
import time

import numpy as np
import torch

from genometools_amd.dist import ALLGATHER_FN, ALLTOALLV_FN, _DevMem


class _View:
    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world
        self.allgather_cb = ALLGATHER_FN(self._allgather)
        self.alltoallv_cb = ALLTOALLV_FN(self._alltoallv)
        self.bytes_exchanged = 0
        self.calls = 0
        self.compute_s = 0.0      # serial mode: time this part had the device

    # serial mode (tools/parts_probe.py): only one part computes at a time, so
    # the time a part holds the device is what it would need on a GPU of its
    # own; the parts hand the device over at every collective
    def begin(self):
        if self.shared.serial:
            self.shared.gpu_lock.acquire()
            self._t = time.perf_counter()

    def end(self):
        if self.shared.serial:
            torch.cuda.synchronize(torch.device("cuda", self.shared.device))
            self.compute_s += time.perf_counter() - self._t
            self.shared.gpu_lock.release()

    def _allgather(self, user, send, recv, nbytes):
        try:
            sh = self.shared
            self.end()
            sh.slots[self.rank] = bytes((ctypes.c_uint8 * nbytes).from_address(send))
            sh.barrier.wait()
            ctypes.memmove(recv, b"".join(sh.slots), nbytes * self.world)
            sh.barrier.wait()
            self.begin()
            return 0
        except Exception as e:
            print("thread allgather failed", repr(e), flush=True)
            return -1

    def _alltoallv(self, user, send, sendcounts, recv, recvcounts, elem, stream):
        try:
            sh = self.shared
            sc = [int(sendcounts[r]) * elem for r in range(self.world)]
            rc = [int(recvcounts[r]) * elem for r in range(self.world)]
            dev = torch.device("cuda", sh.device)
            # everything the engine has queued on its stream must be done before
            # another thread copies out of this part's send buffer
            torch.cuda.ExternalStream(stream, device=dev).synchronize()
            self.end()
            sh.slots[self.rank] = (send, sc)
            sh.barrier.wait()
            out_off = 0
            for src in range(self.world):
                sptr, ssc = sh.slots[src]
                nb = ssc[self.rank]
                assert nb == rc[src], "count matrix mismatch"
                if nb:
                    s_off = sum(ssc[:self.rank])
                    s_t = torch.as_tensor(_DevMem(sptr + s_off, nb), device=dev)
                    d_t = torch.as_tensor(_DevMem(recv + out_off, nb), device=dev)
                    d_t.copy_(s_t)
                out_off += nb
            torch.cuda.synchronize(dev)
            self.bytes_exchanged += sum(sc) - sc[self.rank]
            self.calls += 1
            sh.barrier.wait()
            self.begin()
            return 0
        except Exception as e:
            print("thread alltoallv failed", repr(e), flush=True)
            sh.barrier.abort()
            return -1


class ThreadComm:
    def __init__(self, world, device=0, serial=False):
        self.world, self.device, self.serial = world, device, serial
        self.barrier = threading.Barrier(world)
        self.gpu_lock = threading.Lock()
        self.slots = [None] * world

    def view(self, rank):
        return _View(self, rank)


def _assemble(results, n):
    out = {"suf": [], "lcp": [], "bwt": [], "llv": []}
    stats = {"lcptabsum": 0, "largelcpvalues": 0, "longest": 0, "maxbranchdepth": 0,
             "tied_suffixes": 0, "pair_suffixes": 0, "refine_rounds": 0}
    expect = 0
    for off, res, *_ in results:
        assert off == expect, "slices must tile the table"
        for tab in (res.suf, res.bwt, res.lcp):
            if tab is not None:
                expect += len(tab)
                break
        for k in out:
            v = getattr(res, k)
            if v is not None:
                out[k].append(v)
        for k in ("lcptabsum", "largelcpvalues", "longest", "tied_suffixes", "pair_suffixes"):
            stats[k] += res.stats[k]
        for k in ("maxbranchdepth", "refine_rounds"):
            stats[k] = max(stats[k], res.stats[k])
        stats["prefixlength"] = res.stats["prefixlength"]
    assert expect == n + 1
    tabs = {k: (np.concatenate(v) if v else None) for k, v in out.items()}
    return tabs, stats


def build_in_parts(enc, sigma, parts, want=7, device=0, timing=False):
    """run `parts` engines concurrently; return the assembled tables and the
    combined statistics"""
    out = build_sequences_in_parts([enc], sigma, parts, want, device)
    tabs, stats, results = out[0]
    if timing:
        return tabs, stats, [(r[1].stats, r[1].timing) for r in results]
    return tabs, stats, [r[1].stats for r in results]


def build_sequences_in_parts(encs, sigma, parts, want=7, device=0, serial=False, views=None):
    """the sequences of `encs`, one after the other, through the SAME `parts`
    engine contexts (one thread per part); per sequence: assembled tables,
    combined statistics, per-part (offset, result)"""
    from genometools_amd import esa
    encs = [np.ascontiguousarray(e, dtype=np.uint8) for e in encs]
    shared = ThreadComm(parts, device, serial)
    results = [[None] * parts for _ in encs]
    errors = []
    cap = max(max(e.size for e in encs), 1)

    def worker(r):
        try:
            with esa.EsaEngine(cap, sigma, device) as eng:
                view = shared.view(r)
                if views is not None:
                    views[r] = view
                for k, enc in enumerate(encs):
                    eng.set_sequence(enc)
                    eng.set_part(r, parts, view)
                    view.compute_s, view.bytes_exchanged, view.calls = 0.0, 0, 0
                    view.begin()
                    try:
                        eng.run(want)
                    finally:
                        view.end()
                    results[k][r] = (eng.table_offset(), eng.result(), view.compute_s)
        except Exception as e:   # noqa: BLE001
            errors.append((r, repr(e)))
            shared.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(parts)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    out = []
    for enc, res in zip(encs, results):
        tabs, stats = _assemble(res, enc.size)
        out.append((tabs, stats, res))
    return out
