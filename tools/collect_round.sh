set -e
LT_SHADOW_PACKETS=1 tools/collect_profiles.sh r2 > gpurun_out/collect_r2.log 2>&1; tail -1 gpurun_out/collect_r2.log
LT_SHADOW_PACKETS=2 tools/collect_profiles.sh r2_soup --scene soup > gpurun_out/collect_soup.log 2>&1; tail -1 gpurun_out/collect_soup.log
LT_RETREE=0 LT_SHADOW_PACKETS=1 tools/collect_profiles.sh r2_callers_splits > gpurun_out/collect_callers.log 2>&1; tail -1 gpurun_out/collect_callers.log
tools/collect_profiles.sh r2_gi_wall --program global_illumination --spp 4 > gpurun_out/collect_giw.log 2>&1; tail -1 gpurun_out/collect_giw.log
tools/collect_profiles.sh r2_gi_cornell --scene cornell --program global_illumination --width 1920 --height 1080 > gpurun_out/collect_gic.log 2>&1; tail -1 gpurun_out/collect_gic.log
