# where the soup frame's time goes: camera rays only (basic), camera + shadow rays (accumulator) with either shadow walk
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup --scene soup"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["value"], d["ms_per_step"], c.get("shadow_ray_walk"), c.get("node_visits_per_ray"), c.get("tri_tests_per_ray"), c.get("rays_per_frame"))'
echo "basic: $($B --program basic 2>/dev/null | python -c "$j")"
echo "acc per-lane: $(LT_SHADOW_PACKETS=0 $B 2>/dev/null | python -c "$j")"
echo "acc packets: $(LT_SHADOW_PACKETS=1 $B 2>/dev/null | python -c "$j")"
echo "acc sah: $($B --bvh sah 2>/dev/null | python -c "$j")"
echo "basic sah: $($B --bvh sah --program basic 2>/dev/null | python -c "$j")"
