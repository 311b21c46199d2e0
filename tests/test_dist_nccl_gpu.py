"""TorchComm on the real backend (nccl = RCCL) with the only world size a
one-GPU box offers (1): checks that raw device pointers of the engine can be
handed to torch.distributed collectives without copies."""
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
        assert comm.alltoallv_cb(None, src.data_ptr(), p, dst.data_ptr(), p, 4) == 0
        assert torch.equal(src, dst)
        assert comm.bytes_exchanged == 4000
        st = combine_stats({"lcptabsum": 5, "largelcpvalues": 1, "longest": 3,
                            "tied_suffixes": 2, "maxbranchdepth": 9,
                            "refine_rounds": 4}, "cuda:0")
        assert st["lcptabsum"] == 5 and st["maxbranchdepth"] == 9
    finally:
        dist.destroy_process_group()
