SRH_PROFILE_HOST=1 python tools/scene_bench.py --iters 6 > gpurun_out/r02c_scene_full.log 2>&1
grep -c . gpurun_out/r02c_scene_full.log; tail -1 gpurun_out/r02c_scene_full.log
