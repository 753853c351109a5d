# After tools/collect_round_profiles.sh has run on the GPU box (gpurun merges gpurun_out/round/ back): copy the set into
# profiles/ under the round's names.      bash tools/install_round_profiles.sh r05
set -e
R=${1:?round tag, e.g. r05}
cd "$(dirname "$0")/.."
S=gpurun_out/round
for f in bench_c1.json bench_c2.json bench_c3.json bench_c5.json demo_cycle.txt host_split_c1.txt host_split_c2.txt spec_cycles_c2.txt update_moments.txt; do
  [ -f $S/$f ] && [ $f != update_moments.txt ] && cp $S/$f profiles/${R}_$f
done
for f in $S/rXX_*; do cp $f profiles/${R}_$(basename ${f#$S/rXX_}); done
sed "s/rXX_/${R}_/g" $S/pmc_traffic.json > profiles/pmc_traffic.json
for c in c2 c3 c5; do
  { echo "# kernels between two sweep launches of three real cycles of: rocprofv3 --kernel-trace -- python3 bench.py --config $c --steps 5 --warmup 2 (tools/trace_cycle.py); host gaps are inflated by the profiler (every API call costs more under it): the unprofiled cycles are in ${R}_bench_$c.json and ${R}_shard_cycle_c{4,5}.txt"
    grep -v "^+" $S/cycle_timeline_$c.txt; } > profiles/${R}_cycle_timeline_$c.txt
done
echo "installed; shard cycles of this run (append to profiles/${R}_shard_cycle_c{4,5}.txt by hand, with the box's letter):"
grep -v amdgpu $S/shard_cycle_c5.txt | head -4
grep -v amdgpu $S/shard_cycle_c4.txt | head -4
