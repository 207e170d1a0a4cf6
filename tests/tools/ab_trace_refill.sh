# lt_trace_kernel: idle lanes that trigger a refill (LT_TRACE_REFILL) on the wall's GI frame and on the soup with queued shadow rays
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-soup --no-e2e"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"])'
for r in 8 16 24 32 40 48; do
  echo "refill $r: wall GI $(LT_TRACE_REFILL=$r $B --program global_illumination 2>/dev/null | python -c "$j") ms, soup queued $(LT_TRACE_REFILL=$r LT_SHADOW_PACKETS=3 $B --scene soup 2>/dev/null | python -c "$j") ms"
done
