set -x
python -m pytest tests/test_refrun_golden.py -m gpu -q -s 2>&1 | tail -15 > gpurun_out/refrun_gpu.log
for gr in 4 8 16; do echo "== GR=$gr"; SRH_Q192_GR=$gr tools/probes/gemm_probe 7 50 qkv,fc1,hfc2 2>&1 | grep -v check; done > gpurun_out/gr_sweep.log 2>&1
for gr in 4 8; do SRH_Q192_GR=$gr python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1; done > gpurun_out/gr_bench.log 2>&1
cat gpurun_out/refrun_gpu.log gpurun_out/gr_sweep.log; cut -c1-300 gpurun_out/gr_bench.log
