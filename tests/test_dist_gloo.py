"""The transport of the sharded build (genometools_amd.dist.TorchComm) on CPU:
world_size 2 and 3 over gloo, driven through the same C function-pointer
signatures the engine calls (include/gtamd_esa.h gtamd_allgather_fn /
gtamd_alltoallv_fn), with host memory standing in for device memory."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from genometools_amd.dist import TorchComm, combine_stats
        comm = TorchComm("cpu")
        # --- allgather: 16 bytes per rank, as the slice-border exchange does
        mine = np.array([rank + 1, 1000 + rank], dtype=np.uint64)
        got = np.zeros(2 * world, dtype=np.uint64)
        rc = comm.allgather_cb(None, mine.ctypes.data, got.ctypes.data, 16)
        assert rc == 0
        assert got.tolist() == [v for r in range(world) for v in (r + 1, 1000 + r)]
        # --- alltoallv of 4-byte queries: rank r sends (r + d + 1) elements to d
        sc = np.array([rank + d + 1 for d in range(world)], dtype=np.uint64)
        rcnt = np.array([s + rank + 1 for s in range(world)], dtype=np.uint64)
        send = np.concatenate([np.full(int(sc[d]), 100 * rank + d, dtype=np.uint32)
                               for d in range(world)])
        recv = np.zeros(int(rcnt.sum()), dtype=np.uint32)
        rc = comm.alltoallv_cb(None, send.ctypes.data,
                               sc.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                               recv.ctypes.data,
                               rcnt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 4, None)
        assert rc == 0
        expect = np.concatenate([np.full(int(rcnt[s]), 100 * s + rank, dtype=np.uint32)
                                 for s in range(world)])
        assert np.array_equal(recv, expect)
        # --- answers travel back with the transposed counts
        back = np.zeros(int(sc.sum()), dtype=np.uint32)
        rc = comm.alltoallv_cb(None, (recv + 7).ctypes.data,
                               rcnt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                               back.ctypes.data,
                               sc.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 4, None)
        assert rc == 0 and np.array_equal(back, send + 7)
        # --- empty blocks (a part without tied suffixes still answers queries)
        sc0 = np.zeros(world, dtype=np.uint64)
        if rank == 0:
            sc0[world - 1] = 3
        rc0 = np.zeros(world, dtype=np.uint64)
        if rank == world - 1:
            rc0[0] = 3
        send0 = np.array([5, 6, 7], dtype=np.uint32)
        recv0 = np.zeros(3, dtype=np.uint32)
        rc = comm.alltoallv_cb(None, send0.ctypes.data if rank == 0 else None,
                               sc0.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                               recv0.ctypes.data if rank == world - 1 else None,
                               rc0.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 4, None)
        assert rc == 0
        if rank == world - 1:
            assert recv0.tolist() == [5, 6, 7]
        # --- statistics of the parts -> statistics of the table
        st = combine_stats({"lcptabsum": 10 * (rank + 1), "largelcpvalues": rank,
                            "longest": 77 if rank == 1 else 0, "tied_suffixes": 2,
                            "maxbranchdepth": 5 + rank, "refine_rounds": rank}, "cpu")
        assert st["lcptabsum"] == 10 * world * (world + 1) // 2
        assert st["longest"] == 77 and st["maxbranchdepth"] == 4 + world
        assert st["tied_suffixes"] == 2 * world
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_torchcomm_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(out) == [(r, "ok") for r in range(world)], out
