#!/usr/bin/env python3
"""bench.py -- headline benchmark: Gbp/s of ESA construction (suf+lcp+bwt).

  python bench.py [--gpus N --steps K --warmup W]

A "step" is one complete build of the enhanced suffix array (.suf + .lcp/.llv +
.bwt, resident in HBM) from the encoded sequence already resident in HBM in
its packed form.  N=1 workload: BASELINE.json configs[2], 3 Gbp human-like DNA
(genometools_amd.synth MODEL_HUMANLIKE_DNA, seed 43), -suf -lcp -bwt.

N>1 (one process per GPU, torch.distributed / RCCL): the SAME sequence is
replicated on every GPU (750 MB packed) and the suffix array is sharded into N
lexicographic ranges (the reference's -parts idea); rank r builds slice r of
every table.  A rank filters the suffixes of its key range from the replicated
text (no exchange for the first sort) and sorts them most significant digit
first; the rank table of the doubling rounds is cut by text position (first
ranks of the windows the rounds can reach, then per round one fused alltoallv of
new ranks + queries and one of answers over xGMI, enqueued on the engine's
stream).  Total work is
fixed, so the line says "scaling": "strong" and value = n / (max over ranks of
the step time).

The JSON line carries, besides the contract fields:
  roofline      the dominant kernel (k_msd_local, level D of the MSD first sort:
                8 B read per entry of a run it takes, 14.125 B written per entry
                it sorts itself; the LSD scatter pass elsewhere: 12 in + 12 out):
                algorithmic bytes per launch
                / average launch duration measured with HIP events on the
                engine's stream; `traffic` = HBM bytes per launch from the PMC
                counters (profiles/traffic.json, only if it was measured on
                this revision of the kernel source, else null).
  job_frac      SURVEY.md 8d's own figure: whole-build compulsory bytes
                (10.25 B/bp) / step time / (8 TB/s x GPUs).
  cpu_baseline  the reference's own engine (oracle/_ref/gt_ref_sfx, kind
                "reference") or the CPU restatement (kind "port") timed on
                this host's cores on a bounded sample of the same model.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from genometools_amd import _lib, esa, synth  # noqa: E402
from genometools_amd import dist as gdist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
JOB_BYTES_PER_BP = 10.25       # 0.25 text + 8 suf + 1 lcp + 1 bwt
# algorithmic bytes of the dominant kernel per entry it (reads, writes) in a launch:
# k_rs_scatter: 8+4 read, 8+4 written per pair and radix pass; k_msd_local: K2 + position
# read for every entry of the runs that fit its tile, suf 8 + position 4 + lcp 1 + bwt 1 +
# tie bit 1/8 written for the entries of the runs it sorts itself -- the runs it leaves
# to k_msd_local_radix (a crowded bin) it only reads (DESIGN.md section 4;
# gtamd_esa_timing.scatter_read_items / scatter_written_items)
DOMINANT = {0: ("k_rs_scatter (radix scatter pass of the LSD sort)", 12.0, 12.0),
            1: ("k_msd_local (last level of the MSD sort: LDS sort of a run + table emission)",
                8.0, 14.125)}


def cpu_baseline(model, seed, sample_n):
    """time the reference engine (or the oracle port) on `sample_n` symbols"""
    enc = synth.generate(model, seed, sample_n)
    ref = os.path.join(ROOT, "oracle", "_ref", "gt_ref_sfx")
    cores = 1
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory() as tmp:
            fa = os.path.join(tmp, "sample.fna")
            synth.write_fasta(fa, enc, protein=(model == synth.MODEL_PROTEIN))
            out = subprocess.run(
                [ref, "-protein" if model == synth.MODEL_PROTEIN else "-dna",
                 "-suf", "-lcp", "-bwt", "-time", "-db", fa, "-indexname",
                 os.path.join(tmp, "idx")], check=True, capture_output=True,
                text=True).stdout
        esa_s = float(out.split("esa=")[1].split()[0])
        return {"value": sample_n / esa_s / 1e9, "unit": "Gbp/s", "cores": cores,
                "kind": "reference",
                "sample": "%d bp of the same model/seed, reference Sfxiterator "
                          "-suf -lcp -bwt incl. table writes to tmpfs, %.1f s"
                          % (sample_n, esa_s)}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_util as ou
    t0 = time.time()
    ou.esa(enc, synth.numofchars(model))
    dt = time.time() - t0
    return {"value": sample_n / dt / 1e9, "unit": "Gbp/s", "cores": cores,
            "kind": "port",
            "sample": "%d bp of the same model/seed, oracle/ restatement "
                      "(comparison sort + Kasai), %.1f s" % (sample_n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--symbols", dest="n", type=float, default=3e9, help="sequence length")
    ap.add_argument("--model", type=int, default=synth.MODEL_HUMANLIKE_DNA)
    ap.add_argument("--seed", type=int, default=43)
    ap.add_argument("--cpu-sample", type=float, default=32e6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="gloo + --same-device rehearses N>1 on one GPU")
    ap.add_argument("--same-device", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.same_device:
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend)
    dev = local_rank
    n = int(a.n)
    sigma = synth.numofchars(a.model)
    want = esa.WANT_SUF | esa.WANT_LCP | esa.WANT_BWT
    lib = _lib.load()

    # synthetic input, generated on the device, packed into its resident form
    # (every rank holds the whole sequence)
    seed = a.seed
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:%d" % dev)
    _lib.check(lib.gtamd_synth_bytes(dev, a.model, seed, n, buf.data_ptr()))
    eng = esa.EsaEngine(n, sigma, device=dev)
    eng.set_sequence_device(buf.data_ptr(), n)
    comm = None
    if world > 1:
        comm = gdist.TorchComm("cuda:%d" % dev)
        comm.attach(eng)
    del buf
    torch.cuda.empty_cache()

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        eng.run(want)
    barrier()
    t0 = time.perf_counter()
    sc_ms, sc_launches, total_dev_ms = 0.0, 0, 0.0
    for _ in range(a.steps):
        eng.run(want)   # synchronous: returns when the tables are resident
        tm = eng.timing()
        sc_ms += tm["scatter_ms"]
        sc_launches += tm["scatter_launches"]
        dominant = tm["dominant_kernel"]
        read_per_launch = tm["scatter_read_items"]
        written_per_launch = tm["scatter_written_items"]
        total_dev_ms += tm["total_ms"]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        cdev = "cuda:%d" % dev if a.backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = eng.stats()
    slice_entries = eng.entries(esa.TAB_SUF)
    if world > 1:
        st = gdist.combine_stats(st, "cuda:%d" % dev)
        t = torch.tensor([sc_ms, float(sc_launches), float(read_per_launch),
                          float(written_per_launch), float(comm.bytes_exchanged)],
                         dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        sc_ms, sc_launches = float(t[0].item()) / world, int(t[1].item()) // world
        read_per_launch = float(t[2].item()) / world
        written_per_launch = float(t[3].item()) / world
        exchanged = float(t[4].item())
    else:
        exchanged = 0.0
    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = n / (dt / a.steps) / 1e9
        sc_avg_s = sc_ms / max(sc_launches, 1) / 1e3
        kernel_name, bytes_read, bytes_written = DOMINANT[dominant]
        launch_bytes = bytes_read * float(read_per_launch) + bytes_written * float(written_per_launch)
        achieved = launch_bytes / sc_avg_s / 1e9
        # PMC traffic of that kernel: only a measurement of THIS kernel source
        # counts (tools/pmc_summary.py stamps the file with the source's hash)
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and world == 1:
            with open(tf) as f:
                tj = json.load(f)
            srcfile = os.path.join(ROOT, "genometools_amd", "csrc", tj.get("kernel_source", "esa_prims.hip"))
            with open(srcfile, "rb") as f:
                src = hashlib.sha256(f.read()).hexdigest()
            if (tj.get("n") == n and tj.get("model") == a.model
                    and kernel_name.startswith(tj.get("kernel", "?"))
                    and tj.get("kernel_source_sha256") == src):
                traffic = tj.get("hbm_bytes_per_launch")
        line = {
            "metric": "Gbp/s ESA build (suf+lcp+bwt), 3 Gbp DNA, 1/2/4/8 MI355X; bit-exact vs CPU",
            "value": value, "unit": "Gbp/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s n=%d seed=%d -suf -lcp -bwt (BASELINE.json %s)"
                                   % ({0: "uniform DNA", 1: "human-like DNA (2% N, 24 seqs, repeats)",
                                       2: "protein"}[a.model], n, a.seed,
                                      {0: "configs[1] shape", 1: "configs[2]",
                                       2: "configs[4] shape"}[a.model]),
                       "parallelism": "1 device" if world == 1 else
                                      "%d lexicographic range parts: each rank filters its key "
                                      "range from the replicated text (no exchange for the first "
                                      "sort), rank table cut by position (per round one fused "
                                      "alltoallv of new ranks + queries, one of answers)" % world,
                       "pair_suffixes": st.get("pair_suffixes", 0),
                       "xgmi_bytes_per_step": exchanged / max(a.steps + a.warmup, 1),
                       "tied_suffixes": st["tied_suffixes"],
                       "refine_rounds": st["refine_rounds"],
                       "largelcpvalues": st["largelcpvalues"],
                       "maxbranchdepth": st["maxbranchdepth"]},
            "roofline": {"bound": "hbm", "kernel": kernel_name,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launch_ms": sc_avg_s * 1e3,
                         "algorithmic_bytes_per_launch": launch_bytes,
                         "entries_read_per_launch": float(read_per_launch),
                         "entries_written_per_launch": float(written_per_launch),
                         "launches_per_step": sc_launches // max(a.steps, 1),
                         "job_frac": JOB_BYTES_PER_BP * n / (dt / a.steps) / 1e9 / (HBM_PEAK_GBS * world),
                         "job": {"achieved": JOB_BYTES_PER_BP * n / (dt / a.steps) / 1e9,
                                 "frac": JOB_BYTES_PER_BP * n / (dt / a.steps) / 1e9 / (HBM_PEAK_GBS * world),
                                 "bytes_per_bp": JOB_BYTES_PER_BP,
                                 "device_ms_per_step": total_dev_ms / a.steps}},
        }
        line["job_frac"] = line["roofline"]["job_frac"]
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.model, a.seed, int(a.cpu_sample))
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
