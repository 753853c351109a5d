# Everything committed under profiles/ for one round, in one gpurun call (developer aid):
#   bash tools/collect_round_profiles.sh        then, in the build container, copy gpurun_out/round/* to
#   profiles/ under the round's names (rXX_* -> rNN_*)
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/round
( time python bench.py --steps 20 --warmup 5 > gpurun_out/round/bench_c3.json 2> gpurun_out/round/bench_c3.err ) 2> gpurun_out/round/bench_c3.time
python bench.py --config c5 --steps 12 --warmup 3 > gpurun_out/round/bench_c5.json 2>/dev/null
python bench.py --config c2 --steps 40 --warmup 5 > gpurun_out/round/bench_c2.json 2>/dev/null
python bench.py --config c1 --steps 400 --warmup 20 > gpurun_out/round/bench_c1.json 2>/dev/null
python tools/measure_demo_cycle.py > gpurun_out/round/demo_cycle.txt 2>/dev/null
python tools/measure_update.py 3 > gpurun_out/round/update_moments.txt 2>/dev/null
python tools/measure_update.py 10 >> gpurun_out/round/update_moments.txt 2>/dev/null
python tools/shard_cycle.py c3 8 > gpurun_out/round/shard_cycle_c4.txt 2>/dev/null
python tools/shard_cycle.py c5 8 > gpurun_out/round/shard_cycle_c5.txt 2>/dev/null
python tools/spec_cycles.py c2 60 > gpurun_out/round/spec_cycles_c2.txt 2>/dev/null
python tools/profile_host_split.py c1 2000 > gpurun_out/round/host_split_c1.txt 2>/dev/null
python tools/profile_host_split.py c2 300 > gpurun_out/round/host_split_c2.txt 2>/dev/null
bash tools/profile_rocprof.sh c3
bash tools/profile_rocprof.sh c5
bash tools/profile_rocprof.sh c2
bash tools/profile_sq.sh
for cfg in c3 c5 c2; do
  src=gpurun_out/prof; [ $cfg != c3 ] && src=gpurun_out/prof_$cfg
  ( for k in -9 -8 -7; do timeout 60 python tools/trace_cycle.py $src/trace/bench_results.db $k; echo; done ) > gpurun_out/round/cycle_timeline_$cfg.txt 2>&1
done
# the rocpd databases are too large to travel back (64 MiB limit): summarise them here
cp profiles/pmc_traffic.json gpurun_out/round/ 2>/dev/null
OBE_PROFILE_DST=gpurun_out/round python tools/summarize_profiles.py rXX c3 > /dev/null
OBE_PROFILE_DST=gpurun_out/round python tools/summarize_profiles.py rXX c5 > /dev/null
OBE_PROFILE_DST=gpurun_out/round python tools/summarize_profiles.py rXX c2 > /dev/null
rm -rf gpurun_out/prof gpurun_out/prof_c5 gpurun_out/prof_c2 gpurun_out/prof_sq
du -sh gpurun_out
