# Everything committed under profiles/ for one round, in one gpurun call (developer aid):
#   bash tools/collect_round_profiles.sh        then, in the build container, copy gpurun_out/round/* to
#   profiles/ under the round's names (rXX_* -> rNN_*)
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/round
python bench.py --steps 20 --warmup 5 > gpurun_out/round/bench_c3.json 2> gpurun_out/round/bench_c3.err
python bench.py --config c5 --steps 12 --warmup 3 > gpurun_out/round/bench_c5.json 2>/dev/null
python bench.py --config c2 --steps 40 --warmup 5 > gpurun_out/round/bench_c2.json 2>/dev/null
python bench.py --config c1 --steps 400 --warmup 20 > gpurun_out/round/bench_c1.json 2>/dev/null
python tools/measure_demo_cycle.py > gpurun_out/round/demo_cycle.txt 2>/dev/null
python tools/measure_update.py 3 > gpurun_out/round/update_moments.txt 2>/dev/null
python tools/measure_update.py 10 >> gpurun_out/round/update_moments.txt 2>/dev/null
bash tools/profile_rocprof.sh c3
bash tools/profile_rocprof.sh c5
bash tools/profile_sq.sh
python tools/trace_cycle.py gpurun_out/prof/trace/bench_results.db > gpurun_out/round/cycle_timeline_c3.txt 2>&1
python tools/trace_cycle.py gpurun_out/prof_c5/trace/bench_results.db > gpurun_out/round/cycle_timeline_c5.txt 2>&1
# the rocpd databases are too large to travel back (64 MiB limit): summarise them here
cp profiles/pmc_traffic.json gpurun_out/round/ 2>/dev/null
OBE_PROFILE_DST=gpurun_out/round python tools/summarize_profiles.py rXX c3 > /dev/null
OBE_PROFILE_DST=gpurun_out/round python tools/summarize_profiles.py rXX c5 > /dev/null
rm -rf gpurun_out/prof gpurun_out/prof_c5 gpurun_out/prof_sq
du -sh gpurun_out
