SRH_PROFILE_HOST=1 python tools/scene_bench.py --iters 4 2>&1 | grep -v "queries\]" | tail -42 > gpurun_out/r02_scene_stages.log
for w in full vith256; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_$w.json 2>/dev/null
  bash tools/profile_gpu.sh r02 $w > gpurun_out/r02_profile_$w.log 2>&1
done
cat gpurun_out/r02_scene_stages.log; cut -c1-700 gpurun_out/r02_bench_full.json; echo; cut -c1-700 gpurun_out/r02_bench_vith256.json
