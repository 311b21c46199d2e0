#!/bin/bash
# dev helper for one gpurun call: tests, then (only if the tests RAN to the end,
# whatever their verdict) the bench; never continues after a timeout or a crash
set -o pipefail
mkdir -p gpurun_out
tests="$1"; shift
tag="$1"; shift
timeout -k 10 900 python -m pytest $tests -x -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${tag}_tests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "tests ended with $rc: stopping"; exit $rc; fi
if [ -n "$1" ]; then
  timeout -k 10 600 "$@" > gpurun_out/${tag}_bench.log 2> gpurun_out/${tag}_bench.err
  brc=$?
  tail -3 gpurun_out/${tag}_bench.log; tail -5 gpurun_out/${tag}_bench.err
  exit $brc
fi
exit $rc
