set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -3
python bench.py --no-scene --no-sustained --no-cpu-baseline --no-reference-gpu --steps 40 2>/dev/null | tail -1 > gpurun_out/win2_bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/win2_bench.json")); c=d["roofline"]["by_class_ms_per_step"]
print(d["value"], d["ms_per_step"], {k:round(v,3) for k,v in c.items()})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_trace -o tr -- python $GRAFT_REPO_ROOT/bench.py --no-scene --no-sustained --no-cpu-baseline --no-reference-gpu --no-check --steps 12 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r04_trace -name "*kernel_trace.csv" | head -1)
python tools/kernel_gaps.py $f > gpurun_out/r04_kernel_gaps.txt 2>&1
cat gpurun_out/r04_kernel_gaps.txt | head -40
rm -rf gpurun_out/r04_trace
