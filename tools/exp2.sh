tools/probes/gemm_probe 7 50,70,71,72 qkv,fc1,hproj,hfc2 > gpurun_out/v192_probe.log 2>&1
cat gpurun_out/v192_probe.log
for v in 0 1; do SRH_GEMM_V192=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400; done > gpurun_out/v192_bench.log 2>&1
cat gpurun_out/v192_bench.log
