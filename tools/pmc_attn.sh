#!/bin/bash
# SQ counters of the two attention kernels alone (tools/attn_probe.py one) — run via gpurun; one rocprofv3 pass per counter group
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 120 rocprofv3 --output-format csv --pmc $grp -d $R/gpurun_out/pmc_attn$i -o pmc -- python $R/tools/attn_probe.py one > $R/gpurun_out/pmc_attn$i.log 2>&1
done
cd $R
find gpurun_out/pmc_attn* -type f ! -name "*counter_collection.csv" ! -name "*.log" -delete
ls gpurun_out/pmc_attn*/ 2>/dev/null | head
