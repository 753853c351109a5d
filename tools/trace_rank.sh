# kernel timeline of one rank's cycles under rocprofv3 (developer aid):  bash tools/trace_rank.sh c5
set -x
export TMPDIR=/tmp
CFG=${1:-c5}
OUT=gpurun_out/prof_rank_$CFG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o rank -- python3 tools/rank_cycles.py $CFG 8 14 > $OUT/stdout.log 2>&1
tail -2 $OUT/stdout.log
DB=$(find $OUT -name "*.db" | head -1)
( for k in -8 -7 -6 -5 -4 -3 -2; do timeout 60 python tools/trace_cycle.py $DB $k; echo; done ) > gpurun_out/rank_timeline_$CFG.txt 2>&1
rm -rf $OUT
