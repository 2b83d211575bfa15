#!/bin/bash
# cache / TLB counters of the GEMM probe (w192 variant 61 = k-loop only, q192 variant 51) — run via gpurun
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --output-format csv --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/pmc_gemm3 -o pmc -- $R/tools/probes/gemm_probe 2 61,51 qkv,hfc2 > $R/gpurun_out/pmc_gemm3.log 2>&1
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_gemm4 -o pmc -- $R/tools/probes/gemm_probe 2 61,51 qkv,hfc2 > $R/gpurun_out/pmc_gemm4.log 2>&1
rocprofv3 --output-format csv --pmc TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCC_TAG_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum -d $R/gpurun_out/pmc_gemm5 -o pmc -- $R/tools/probes/gemm_probe 2 61,51 qkv,hfc2 > $R/gpurun_out/pmc_gemm5.log 2>&1
cd $R
find gpurun_out/pmc_gemm3 gpurun_out/pmc_gemm4 gpurun_out/pmc_gemm5 -type f ! -name "*counter_collection.csv" -delete
