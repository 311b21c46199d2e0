cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 8 32 512 2048; do
GTAMD_PAIR_CHUNK=$m rocprofv3 --kernel-trace --stats -d gpurun_out/c12_trace$m -o p -- python tools/parts_probe.py --n 3e9 --parts 8 --serial --reps 2 > /dev/null 2>&1
echo "chunk $m done"
done
exit 0
