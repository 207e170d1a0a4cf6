# height slack of the backend's own tree (levels above ceil(log2 leaves)): frame ms
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["ms_per_step"], c.get("shadow_ray_walk"))'
for sc in ${SCENES:-wall soup blob colonnade mixed}; do
  for sl in ${SLACKS:-0 1 2 4 8}; do echo "$sc slack=$sl: $(LT_RETREE_SLACK=$sl $B --scene $sc 2>/dev/null | python -c "$j")"; done
done
