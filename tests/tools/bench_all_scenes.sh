# every bench scene with the library's own choice of the shadow-ray walk: frame ms, Mrays/s, the walk chosen
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup --no-e2e"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["value"], d["ms_per_step"], c.get("shadow_ray_walk"))'
for sc in wall soup blob colonnade mixed "wall --bvh sah" "soup --bvh sah" "blob --bvh sah" "colonnade --bvh sah"; do
  echo "$sc: $($B --scene $sc 2>/dev/null | python -c "$j")"
done
