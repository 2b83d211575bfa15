tools/probes/gemm_probe 9 50,58,59 qkv,fc1,hproj,hfc2 > gpurun_out/q192_sgb_probe.log 2>&1
cat gpurun_out/q192_sgb_probe.log
