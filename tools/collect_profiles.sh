#!/bin/bash
# Collects the round's rocprofv3 evidence for bench.py's dominant kernel on the GPU box and summarises it into
# profiles/<round>/ : kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC has 4 slots,
# FETCH_SIZE needs 3 and WRITE_SIZE 2), then an SQ/TA pass.  Run through gpurun from the repo root:
#   gpurun -- 'tools/collect_profiles.sh r1'
set -e
ROUND=${1:-r1}
OUT=gpurun_out/prof_$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# LT_SHADOW_PACKETS=1 (exported by the caller) pins the walk the library picks for the bench scene, so that no launch of a
# pass is the one-off timing run of the other walk
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
B1="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B1 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/sq -- $B1 > $OUT/sq.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/cache -- $B1 > $OUT/cache.log 2>&1
python3 tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/sq $OUT/cache > $OUT/pmc_summary.txt
tail -1 $OUT/stats.log > $OUT/bench_line_under_profiler.json
echo done
