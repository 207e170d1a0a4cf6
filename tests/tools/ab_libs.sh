# A/B of two builds of the library in ONE gpurun call (same box, same clocks): LT_HIP_LIBRARY selects the build.
# usage: bash tests/tools/ab_libs.sh <libA.so> <libB.so> [quick]
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup --no-e2e"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["value"], d["ms_per_step"], c.get("shadow_ray_walk"))'
for lib in "$1" "$2"; do
  export LT_HIP_LIBRARY=$PWD/$lib
  echo "== $lib"
  echo "wall accumulator: $($B 2>/dev/null | python -c "$j")"
  echo "wall accumulator per lane: $(LT_SHADOW_PACKETS=0 $B 2>/dev/null | python -c "$j")"
  echo "soup accumulator per lane: $(LT_SHADOW_PACKETS=0 $B --scene soup 2>/dev/null | python -c "$j")"
  echo "wall GI: $($B --program global_illumination 2>/dev/null | python -c "$j")"
  if [ -z "$3" ]; then
    echo "soup accumulator: $($B --scene soup 2>/dev/null | python -c "$j")"
    echo "blob 1080p 64: $($B --scene blob --width 1920 --height 1080 --spp 64 2>/dev/null | python -c "$j")"
    echo "colonnade: $($B --scene colonnade 2>/dev/null | python -c "$j")"
  fi
done
