# DESIGN 5.1's table: every synthetic scene (4K, accumulator, 16 spp) with each of the four shadow-ray walks forced
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup --no-e2e"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"])'
for scene in wall soup blob colonnade mixed; do
  line="$scene:"
  for m in 1 0 2 3; do line="$line  walk $m = $(LT_SHADOW_PACKETS=$m $B --scene $scene 2>/dev/null | python -c "$j")"; done
  echo "$line   library's choice = $($B --scene $scene 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"], d["config"]["shadow_ray_walk"])')"
done
