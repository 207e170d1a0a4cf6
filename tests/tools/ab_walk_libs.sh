# forced shadow-ray walks (any-hit packets / per lane) on two scenes for several builds of the library (lens_trace_amd/lib/v_<tag>.so;
# LT_HIP_LIBRARY selects the build), one gpurun call: where did a walk get slower?   usage: bash tests/tools/ab_walk_libs.sh tag...
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-soup --no-e2e"
j='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["config"].get("shadow_ray_walk"))'
for tag in "$@"; do
  if [ "$tag" = current ]; then unset LT_HIP_LIBRARY; else export LT_HIP_LIBRARY=$PWD/lens_trace_amd/lib/v_$tag.so; fi
  for scene in colonnade blob; do
    for m in 1 0; do
      echo "$tag $scene walk $m: $(LT_SHADOW_PACKETS=$m $B --scene $scene 2>/dev/null | python -c "$j")"
    done
  done
done
