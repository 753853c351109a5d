# developer aid: K1 launch duration (back-to-back and isolated) vs the number of work items;
# with a second checkout under _ab_old/ (git archive of an older commit, built), the same for it
cfg=${1:-c3}
for b in 1536 4608; do
  OBE_SWEEP_BLOCKS=$b python tools/measure_sweep_launch.py $cfg 6
  if [ -d _ab_old ]; then OBE_AB_ROOT=_ab_old OBE_SWEEP_BLOCKS=$b python tools/measure_sweep_launch.py $cfg 6; fi
done
python tools/measure_sweep_launch.py $cfg 6
