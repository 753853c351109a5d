# developer aid: K1 launch time vs settings-per-lane and workgroup count
cfg=${1:-c3}
for spt in 4 8; do for blocks in 768 1536 2304 3072; do
OBE_SWEEP_SPT=$spt OBE_SWEEP_BLOCKS=$blocks python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg SPT', $spt, 'blocks', $blocks, 'K1 ms', round(d['roofline']['launch_ms'],3), d['roofline']['variant'], 'frac', round(d['roofline']['frac'],4), 'step ms', round(d['ms_per_step'],2))"
done; done
