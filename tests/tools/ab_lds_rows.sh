# occupancy cost of LDS stack rows alone (tree fixed at slack 0), then taller trees at their natural row counts
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d["config"]; print(d["ms_per_step"])'
for sc in wall blob; do
  for rows in 20 21 22 24; do echo "$sc slack0 rows=$rows: $(LT_DEBUG_LDS_ROWS=$rows $B --scene $sc 2>/dev/null | python -c "$j")"; done
  for sl in 1 2 3; do echo "$sc slack=$sl rows=24: $(LT_RETREE_SLACK=$sl LT_DEBUG_LDS_ROWS=24 $B --scene $sc 2>/dev/null | python -c "$j")"; done
done
