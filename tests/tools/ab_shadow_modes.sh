# shadow-ray walk: per lane (0), packets (1), chosen per wavefront (2) at several spread thresholds; frame ms of each scene
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"])'
for sc in ${SCENES:-wall soup blob colonnade mixed}; do
  line="$sc:"
  for m in 0 1; do line="$line m$m=$(LT_SHADOW_PACKETS=$m $B --scene $sc $EXTRA 2>/dev/null | python -c "$j")"; done
  for thr in ${THRS:-0.003 0.01 0.03}; do line="$line thr$thr=$(LT_SHADOW_PACKETS=2 LT_SHADOW_SPREAD=$thr $B --scene $sc $EXTRA 2>/dev/null | python -c "$j")"; done
  echo "$line"
done
