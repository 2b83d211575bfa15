#!/bin/bash
# PMC counters of the GEMM probe (q192 variant 50 and pp256 variant 40) — run on the GPU box via gpurun
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_gemm1 -o pmc -- $R/tools/probes/gemm_probe 2 50,40 qkv,fc1 > $R/gpurun_out/pmc_gemm1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $R/gpurun_out/pmc_gemm2 -o pmc -- $R/tools/probes/gemm_probe 2 50,40 qkv,fc1 > $R/gpurun_out/pmc_gemm2.log 2>&1
cd $R
find gpurun_out/pmc_gemm1 gpurun_out/pmc_gemm2 -type f ! -name "*counter_collection.csv" -delete
