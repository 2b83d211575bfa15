#!/bin/bash
# rocprofv3 kernel stats of the two secondary workloads (full SAMRoad.forward, ViT-H 256) — run via gpurun
TAG=${1:-r01d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for wl in full vith256; do
  rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$wl -o prof -- python $R/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-gpu --no-sustained --no-scene --no-workloads > $R/gpurun_out/${TAG}_prof_$wl.log 2>&1
done
cd $R
find gpurun_out/${TAG}_prof_full gpurun_out/${TAG}_prof_vith256 -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
ls gpurun_out/${TAG}_prof_full gpurun_out/${TAG}_prof_vith256
