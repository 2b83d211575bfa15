python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r02b_gputests.log
python __graft_entry__.py smoke > gpurun_out/r02b_smoke.log 2>&1
python bench.py > gpurun_out/r02b_bench_line.json 2> gpurun_out/r02b_bench.err
SRH_PROFILE_HOST=1 python tools/scene_bench.py --iters 4 2>&1 | grep -v "queries\]" | tail -13 > gpurun_out/r02b_scene_stages.log
cat gpurun_out/r02b_gputests.log gpurun_out/r02b_smoke.log; cut -c1-400 gpurun_out/r02b_bench_line.json; echo; tail -13 gpurun_out/r02b_scene_stages.log
