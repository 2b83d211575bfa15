python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r02_gputests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench.err
bash tools/profile_gpu.sh r02 > gpurun_out/r02_profile.log 2>&1
SRH_PROFILE_HOST=1 python tools/scene_bench.py --iters 3 > gpurun_out/r02_scene.log 2>&1
cat gpurun_out/r02_gputests.log; cat gpurun_out/r02_bench_line.json | cut -c1-3000; tail -30 gpurun_out/r02_scene.log
