#!/bin/bash
# one gpurun call: headline bench, kernel trace of the same command, PMC passes
# (FETCH_SIZE / WRITE_SIZE separately, kernel trace only) of one 3 Gbp build, the
# serial probe of the part build at 3 Gbp (R = 1, 2, 4, 8) with its kernel trace at
# R = 8, the other configs.  Every step only if the one before ended.
set -o pipefail
tag="$1"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_trace.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o fetch -- python tools/scale_probe.py --n 3e9 --model 1 --seed 43 --runs 1 > gpurun_out/${tag}_pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o write -- python tools/scale_probe.py --n 3e9 --model 1 --seed 43 --runs 1 > gpurun_out/${tag}_pmc_write.log 2>&1 || exit 4
rm -f gpurun_out/${tag}_parts3g.jsonl
timeout -k 10 300 python tools/parts_probe.py --n 3e9 --parts 1,2,4,8 --serial --reps 2 --json gpurun_out/${tag}_parts3g.jsonl > gpurun_out/${tag}_parts3g.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_parts8_trace -o parts8 -- python tools/parts_probe.py --n 3e9 --parts 8 --serial --reps 2 > gpurun_out/${tag}_parts8_trace.log 2>&1 || exit 6
timeout -k 10 300 python tools/scale_probe.py --n 256e6 --model 0 --seed 42 --runs 4 > gpurun_out/${tag}_config1.log 2>&1 || exit 7
timeout -k 10 300 python tools/scale_probe.py --n 1e9 --model 2 --seed 44 --runs 4 > gpurun_out/${tag}_config4.log 2>&1 || exit 8
timeout -k 10 300 python tools/scale_probe.py --n 3e9 --model 3 --seed 43 --runs 3 > gpurun_out/${tag}_repeatheavy.log 2>&1 || exit 9
tail -1 gpurun_out/${tag}_bench.json | cut -c1-300
