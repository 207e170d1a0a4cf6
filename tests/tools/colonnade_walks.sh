B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-soup --no-e2e --scene colonnade"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["value"], d["ms_per_step"], c.get("shadow_ray_walk"))'
for spp in 256 16; do
for m in default 1 0 2 3; do
  if [ $m = default ]; then unset LT_SHADOW_PACKETS; else export LT_SHADOW_PACKETS=$m; fi
  echo "colonnade spp $spp walk $m: $($B --spp $spp 2>/dev/null | python -c "$j")"
done
done
unset LT_SHADOW_PACKETS
echo "calibration: $(LT_DEBUG_CALIBRATION=1 $B --spp 256 2>&1 | grep -i "calib\|walk" | head -12)"
