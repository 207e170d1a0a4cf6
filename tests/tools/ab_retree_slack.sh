B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["ms_per_step"], c.get("shadow_ray_walk"))'
for sc in wall soup blob; do
  for sl in 0 1 2 5; do echo "$sc slack=$sl: $(LT_RETREE_SLACK=$sl $B --scene $sc 2>/dev/null | python -c "$j")"; done
done
