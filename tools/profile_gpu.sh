#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then two separate PMC passes
# (FETCH_SIZE / WRITE_SIZE need different TCC slots).  Outputs under gpurun_out/<tag>_*; summarise with
# tools/summarize_profile.py <tag> gpurun_out/<tag>_prof gpurun_out/<tag>_pmc_fetch gpurun_out/<tag>_pmc_write [workload]
# usage: tools/profile_gpu.sh <tag> [encdec|full|vith256]   (outputs are tagged <tag> resp. <tag>_<workload>)
TAG=${1:-r02}
WORKLOAD=${2:-encdec}
[ "$WORKLOAD" != encdec ] && TAG=${TAG}_${WORKLOAD}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$R'); from sam_road_amd import _lib; print(_lib.build_id())" > $R/gpurun_out/${TAG}_build_id.txt
cd /tmp
BENCH="python $R/bench.py --workload $WORKLOAD --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-gpu --no-sustained --no-scene --no-workloads"
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o prof -- $BENCH > $R/gpurun_out/${TAG}_prof.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc_fetch -o pmc -- $BENCH > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pmc_write -o pmc -- $BENCH > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc_mfma -o pmc -- $BENCH > $R/gpurun_out/${TAG}_pmc_mfma.log 2>&1
cd $R
# keep only the small csv summaries (the merge-back limit is 64 MiB)
find gpurun_out/${TAG}_prof gpurun_out/${TAG}_pmc_* -type f ! -name "*.csv" -delete 2>/dev/null; find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete; find gpurun_out/${TAG}_* -type f | head -20
du -sh gpurun_out/${TAG}_* | tail -8
