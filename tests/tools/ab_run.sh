#!/bin/bash
# Diagnostic: A/B a list of library builds (lens_trace_amd/lib/v_<tag>.so) against the default build: pixels of
# ab_dump.py's scenes must be identical, then bench.py on three scenes.  Usage: tests/tools/ab_run.sh tag1 tag2 ...
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python tests/tools/ab_dump.py base > gpurun_out/ab_base.log 2>&1
for t in "$@"; do
  LT_HIP_LIBRARY=$PWD/lens_trace_amd/lib/v_$t.so python tests/tools/ab_dump.py $t > gpurun_out/ab_$t.log 2>&1
done
: > gpurun_out/ab_cmp.log
for t in "$@"; do
  cp_args="base $t"
  python tests/tools/ab_compare.py $cp_args keep >> gpurun_out/ab_cmp.log 2>&1
done
rm -f gpurun_out/ab_*.npy
for s in wall blob soup; do
  for rep in 1 2; do
    python bench.py --scene $s --no-cpu-baseline --steps 3 > gpurun_out/bench_base_${s}_$rep.log 2>&1
    for t in "$@"; do
      LT_HIP_LIBRARY=$PWD/lens_trace_amd/lib/v_$t.so python bench.py --scene $s --no-cpu-baseline --steps 3 > gpurun_out/bench_${t}_${s}_$rep.log 2>&1
    done
  done
done
python - "$@" <<'PY'
import json, sys
for s in ("wall", "blob", "soup"):
    for t in ["base"] + sys.argv[1:]:
        v = []
        for rep in (1, 2):
            try:
                v.append(json.loads(open("gpurun_out/bench_%s_%s_%d.log" % (t, s, rep)).read().strip().splitlines()[-1])["value"])
            except Exception as e:
                v.append(str(e)[:40])
        print(s, t, v)
PY
cat gpurun_out/ab_cmp.log
