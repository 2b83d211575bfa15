SRH_PROFILE_HOST=1 python tools/scene_bench.py --iters 8 > gpurun_out/r02d_scene_full.log 2>&1
tail -1 gpurun_out/r02d_scene_full.log
