# developer aid: K1 launch time vs workgroup count at one config (OBE_SWEEP_BLOCKS), SPT as planned
cfg=${1:-c2}
for blocks in 384 512 640 768 1024 1536 2304; do
OBE_SWEEP_BLOCKS=$blocks python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg blocks', $blocks, 'K1 ms', round(d['roofline']['launch_ms'],4), d['roofline']['variant'], 'frac', round(d['roofline']['frac'],4), 'plain ms', d['config']['median_ms_plain_cycle'])"
done
