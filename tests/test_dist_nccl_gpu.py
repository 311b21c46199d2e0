"""TorchComm on the real backend (nccl = RCCL) with the only world size a
one-GPU box offers (1): raw device pointers of the engine go to
torch.distributed collectives without copies, enqueued on the engine's own
stream; and the engine-level multi-process path (two processes, gloo, staged
through the host, sharing the one GPU)."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_torchcomm_nccl_single_rank(gpu):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        from genometools_amd.dist import TorchComm, combine_stats
        comm = TorchComm("cuda:0")
        mine = np.array([11, 22], dtype=np.uint64)
        got = np.zeros(2, dtype=np.uint64)
        assert comm.allgather_cb(None, mine.ctypes.data, got.ctypes.data, 16) == 0
        assert got.tolist() == [11, 22]
        src = torch.arange(1000, dtype=torch.int32, device="cuda:0")
        dst = torch.zeros(1000, dtype=torch.int32, device="cuda:0")
        cnt = np.array([1000], dtype=np.uint64)
        p = cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        # on a side stream, as the engine calls it: no host synchronisation
        # inside, the result is ordered on that stream
        side = torch.cuda.Stream(device="cuda:0")
        side.wait_stream(torch.cuda.current_stream())
        assert comm.alltoallv_cb(None, src.data_ptr(), p, dst.data_ptr(), p, 4,
                                 side.cuda_stream) == 0
        side.synchronize()
        assert torch.equal(src, dst)
        assert comm.calls == 1
        # a whole part build with one part through the nccl transport: the
        # exchange machinery (forced) with the real collectives
        os.environ["GTAMD_FORCE_WIDE"] = "1"
        try:
            import oracle_util as ou
            from genometools_amd import esa, synth
            enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 3, 300000)
            with esa.EsaEngine(enc.size, 4) as eng:
                eng.set_sequence(enc)
                comm.attach(eng)
                eng.run()
                res = eng.result()
            ora = ou.esa(enc, 4)
            for tab in ("suf", "lcp", "llv", "bwt"):
                assert np.array_equal(getattr(res, tab), ora[tab]), tab
        finally:
            del os.environ["GTAMD_FORCE_WIDE"]
        st = combine_stats({"lcptabsum": 5, "largelcpvalues": 1, "longest": 3,
                            "tied_suffixes": 2, "maxbranchdepth": 9,
                            "refine_rounds": 4}, "cuda:0")
        assert st["lcptabsum"] == 5 and st["maxbranchdepth"] == 9
    finally:
        dist.destroy_process_group()


def _engine_worker(rank, world, port, q, wide):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if wide:
        os.environ["GTAMD_FORCE_WIDE"] = "1"
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_util as ou
        from genometools_amd import esa, synth
        from genometools_amd.dist import TorchComm, combine_stats
        torch.cuda.set_device(0)
        comm = TorchComm("cuda:0")          # device buffers, host transport
        assert comm.staged
        # (tandem arrays: groups that go through the doubling rounds -- a DNA part
        # build exchanges nothing but their ranks, the sort of a part is local)
        enc = synth.generate(synth.MODEL_REPEAT_HEAVY, 17, 350000)
        ora = ou.esa(enc, 4)
        with esa.EsaEngine(enc.size, 4) as eng:
            eng.set_sequence(enc)
            comm.attach(eng)
            eng.run()
            off, res = eng.table_offset(), eng.result()
            st = combine_stats(res.stats, "cuda:0")
        cnt = len(res.suf)
        assert np.array_equal(res.suf, ora["suf"][off:off + cnt]), "suf"
        assert np.array_equal(res.lcp, ora["lcp"][off:off + cnt]), "lcp"
        assert np.array_equal(res.bwt, ora["bwt"][off:off + cnt]), "bwt"
        llv = ora["llv"]
        mine = llv[(llv[:, 0] >= off) & (llv[:, 0] < off + cnt)]
        assert np.array_equal(res.llv, mine), "llv"
        assert st["lcptabsum"] == int(ora["stats"]["lcptabsum"])
        assert st["longest"] == ora["stats"]["longest"]
        assert st["maxbranchdepth"] == ora["stats"]["maxbranchdepth"]
        assert st["refine_rounds"] > 0 and comm.calls > 0
        q.put((rank, "ok", cnt, comm.bytes_exchanged))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, repr(e) + traceback.format_exc(), 0, 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("wide", [False, True])
def test_engine_across_processes(gpu, wide):
    """engine + TorchComm in two processes (gloo; device buffers staged through
    the host because both ranks share the one GPU of the box): every rank's
    slice equals the oracle's, slices tile the table"""
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, q, wide))
             for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    assert sorted(o[:2] for o in out) == [(r, "ok") for r in range(world)], out
    assert sum(o[2] for o in out) == 350001
    assert sum(o[3] for o in out) > 0          # ranks travelled


_RCCL_WORLD1 = r"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests"))
sys.path.insert(0, %(root)r)
os.environ["GTAMD_FORCE_WIDE"] = "1"      # the exchange machinery of a part build, one part
import oracle_util as ou
from genometools_amd import _lib, esa, synth
lib = _lib.load()
ident = (ctypes.c_uint8 * 128)()
_lib.check(lib.gtamd_comm_rccl_unique_id(ident))
comm = lib.gtamd_comm_rccl_create(ident, 0, 1, 0)
assert comm, lib.gtamd_esa_last_error()
enc = synth.generate(synth.MODEL_REPEAT_HEAVY, 3, 200000)
ora = ou.esa(enc, 4)
with esa.EsaEngine(enc.size, 4) as eng:
    eng.set_sequence(enc)
    _lib.check(lib.gtamd_comm_attach(comm, 0, eng._ctx, 0))
    eng.run()
    res = eng.result()
    assert eng.timing()["comm_calls"] > 0
for tab in ("suf", "lcp", "llv", "bwt"):
    assert np.array_equal(getattr(res, tab), ora[tab]), tab
lib.gtamd_comm_destroy(comm)
print("rccl world 1 ok")
"""


def test_library_rccl_transport_world_1(gpu):
    """the RCCL transport that ships with the library (genometools_amd/csrc/esa_comm.hip:
    librccl bound at run time, ncclAllGather / grouped ncclSend-ncclRecv) as a plumbing
    test with one rank -- RCCL refuses two ranks on one device, and the box has one.
    In a process of its own: a fault inside librccl must not take the test run along."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _RCCL_WORLD1 % {"root": root}], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "rccl world 1 ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
