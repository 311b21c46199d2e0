#!/bin/bash
# one gpurun call of round 3: a set of GPU tests, then (only if they ran to the end)
# the probes named on the command line; every step only if the one before ended
#   tools/r3_call.sh TAG "TESTS" [probe command ...]
set -o pipefail
mkdir -p gpurun_out
tag="$1"; shift
tests="$1"; shift
if [ -n "$tests" ]; then
  timeout -k 10 1000 python -m pytest $tests -x -q > gpurun_out/${tag}_tests.log 2>&1
  rc=$?
  tail -15 gpurun_out/${tag}_tests.log
  if [ $rc -ne 0 ]; then echo "tests ended with $rc: stopping"; exit $rc; fi
fi
i=0
while [ -n "$1" ]; do
  i=$((i+1))
  timeout -k 10 600 bash -c "$1" > gpurun_out/${tag}_probe$i.log 2>&1
  prc=$?
  tail -12 gpurun_out/${tag}_probe$i.log
  if [ $prc -ne 0 ]; then echo "probe $i ended with $prc: stopping"; exit $prc; fi
  shift
done
exit 0
