# Collect the rocprofv3 evidence for profiles/: kernel trace + stats, then the HBM
# counters in their own passes (never combined with other trace domains).
set -x
export TMPDIR=/tmp
CFG=${1:-c3}
OUT=gpurun_out/prof
[ "$CFG" != "c3" ] && OUT=gpurun_out/prof_$CFG
mkdir -p $OUT
CMD="python3 bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/trace_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/pmc_fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $CMD > $OUT/pmc_write_stdout.log 2>&1
find $OUT -type f | head -50
du -sh $OUT
