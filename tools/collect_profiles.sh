#!/bin/bash
# Collects a round's rocprofv3 evidence for bench.py's dominant kernel on the GPU box and summarises it into
# gpurun_out/prof_<round>/ (copy what is to be judged into profiles/<round>/).  One kernel-trace --stats pass, then counters in
# their own passes (--pmc with --kernel-trace only; TCC has 4 slots: FETCH_SIZE needs 3, WRITE_SIZE 2; SQ has 8):
#   fetch / write : HBM traffic                                         -> hbm_traffic.json
#   issue         : where the wave-cycles go (SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES)
#   pipes         : per-pipe active cycles (VALU, scalar, VMEM, LDS, FLAT, MISC) and instruction counts
#   insts         : instruction mix (VALU / SALU / SMEM / VMEM / LDS / branch)
#   sqc           : instruction-cache and scalar-data-cache requests / hits / misses
#   cache         : TA busy, L1 / L2 hit rates, clock                   -> issue_profile.json (tools/profile_summary.py)
# Run through gpurun from the repo root:   gpurun -- 'tools/collect_profiles.sh r2 [bench.py arguments]'
# LT_SHADOW_PACKETS (exported by the caller when wanted) pins the shadow-ray walk, so that no launch of a pass is the one-off
# timing run of the other walk.
set -e
ROUND=${1:-r2}
shift || true
EXTRA="$@"
OUT=gpurun_out/prof_$ROUND
rm -rf "$OUT"      # (a pass directory must hold ONE run: the summaries average over every csv they find)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-soup --no-e2e $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
echo "stats pass done"
B1="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-soup --no-e2e $EXTRA"
pass() {   # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $B1 > $OUT/$name.log 2>&1 || { echo "pass $name FAILED (see $OUT/$name.log)"; tail -3 $OUT/$name.log; return 0; }
  echo "pass $name done"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass issue SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM
pass pipes SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH
pass sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ
pass cache TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/issue $OUT/pipes $OUT/insts $OUT/sqc $OUT/cache > $OUT/pmc_summary.txt
python3 tools/profile_summary.py $OUT "$B1" > $OUT/profile_summary.log 2>&1 || tail -5 $OUT/profile_summary.log
grep '^{"metric"' $OUT/stats.log | tail -1 > $OUT/bench_line_under_profiler.json
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || true
echo done
