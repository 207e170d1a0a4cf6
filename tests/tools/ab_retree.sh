# the backend's own tree over the scene's leaves (LT_RETREE=1, default) against walking the caller's tree everywhere (=0)
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["value"], d["ms_per_step"], c.get("shadow_ray_walk"))'
for sc in ${SCENES:-wall soup blob colonnade mixed}; do
  for rt in 0 1; do echo "$sc LT_RETREE=$rt $EXTRA: $(LT_RETREE=$rt $B --scene $sc $EXTRA 2>/dev/null | python -c "$j")"; done
done
