# SQ issue counters of the K1 sweep kernel (own pass: --pmc only, no trace domains).
export TMPDIR=/tmp
OUT=gpurun_out/prof_sq
mkdir -p $OUT
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/sq -o bench -- $CMD > $OUT/stdout.log 2>&1
find $OUT -type f | head
