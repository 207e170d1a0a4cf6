# work order of a fused launch: the frames of a square side by side (LT_SQUARE_MAJOR=1) or frame after frame (=0)
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["ms_per_step"], c.get("shadow_ray_walk"))'
for sc in ${SCENES:-wall soup blob colonnade mixed}; do
  for m in 0 1; do echo "$sc LT_SQUARE_MAJOR=$m: $(LT_SQUARE_MAJOR=$m $B --scene $sc 2>/dev/null | python -c "$j")"; done
done
