# the other BASELINE configurations through bench.py, one GPU: value (Mrays/s), ms per step
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup --no-e2e"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["value"], d["ms_per_step"], c.get("shadow_ray_walk"), "visits/ray", round(c["node_visits_per_ray"],1))'
echo "config 2 (Cornell GI 1080p, 16 spp, 16 bounces): $($B --scene cornell --program global_illumination --width 1920 --height 1080 2>/dev/null | python -c "$j")"
echo "config 3 (blob accumulator 1080p, 64 frames): $($B --scene blob --width 1920 --height 1080 --spp 64 2>/dev/null | python -c "$j")"
echo "config 5 (colonnade accumulator 4K, 256 frames): $($B --scene colonnade --spp 256 2>/dev/null | python -c "$j")"
echo "wall GI 4K, 16 spp, 16 bounces: $($B --program global_illumination 2>/dev/null | python -c "$j")"
echo "wall basic 4K 16 spp: $($B --program basic 2>/dev/null | python -c "$j")"
echo "wall accumulator, shadow rays per lane: $(LT_SHADOW_PACKETS=0 $B 2>/dev/null | python -c "$j")"
