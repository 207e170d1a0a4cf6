# one-GPU figures of a build: headline (wall accumulator), soup, the other configs, GI on the wall, per-lane shadow walk
# usage (through gpurun, from the repo root): bash tests/tools/bench_round.sh > gpurun_out/<name>.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["value"], d["ms_per_step"], c.get("shadow_ray_walk"), "soup", c.get("soup_mrays_per_s"), c.get("soup_frame_ms"), c.get("soup_shadow_ray_walk"), "build_ms", c.get("own_hierarchy_build_ms"))'
echo "wall accumulator 4K 16 spp: $($B 2>/dev/null | python -c "$j")"
B="$B --no-soup --no-e2e"
echo "wall GI 4K, 16 spp, 16 bounces: $($B --program global_illumination 2>/dev/null | python -c "$j")"
echo "wall accumulator, shadow rays per lane: $(LT_SHADOW_PACKETS=0 $B 2>/dev/null | python -c "$j")"
echo "soup accumulator, shadow rays per lane: $(LT_SHADOW_PACKETS=0 $B --scene soup 2>/dev/null | python -c "$j")"
echo "config 2 (Cornell GI 1080p, 16 spp, 16 bounces): $($B --scene cornell --program global_illumination --width 1920 --height 1080 2>/dev/null | python -c "$j")"
echo "config 3 (blob accumulator 1080p, 64 frames): $($B --scene blob --width 1920 --height 1080 --spp 64 2>/dev/null | python -c "$j")"
echo "config 5 (colonnade accumulator 4K, 256 frames): $($B --scene colonnade --spp 256 2>/dev/null | python -c "$j")"
echo "mixed: $($B --scene mixed 2>/dev/null | python -c "$j")"
