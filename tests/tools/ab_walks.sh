# camera rays only (basic), then accumulator with each shadow-ray walk forced: frame ms
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["ms_per_step"])'
for sc in ${SCENES:-wall soup}; do
  echo "$sc basic: $($B --scene $sc --program basic 2>/dev/null | python -c "$j")"
  for m in 0 1 2; do echo "$sc accumulator LT_SHADOW_PACKETS=$m: $(LT_SHADOW_PACKETS=$m $B --scene $sc 2>/dev/null | python -c "$j")"; done
done
