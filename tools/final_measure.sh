#!/bin/bash
# one gpurun call: headline bench, kernel trace of the same command, PMC passes
# (FETCH_SIZE / WRITE_SIZE separately, kernel trace only) of one 3 Gbp build,
# packed-index probe with its kernel trace.  Every step only if the one before ended.
set -o pipefail
tag="$1"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_trace.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o fetch -- python tools/scale_probe.py --n 3e9 --model 1 --seed 43 --runs 1 > gpurun_out/${tag}_pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o write -- python tools/scale_probe.py --n 3e9 --model 1 --seed 43 --runs 1 > gpurun_out/${tag}_pmc_write.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_pck_trace -o pck -- python tools/pck_probe.py --symbols 3e9 --reps 3 > gpurun_out/${tag}_pck.log 2>&1 || exit 5
tail -1 gpurun_out/${tag}_bench.json | cut -c1-400
