python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -5
for v in 0 1; do SRH_ATTN_WINDOW_P8=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-gpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['by_class_ms_per_step'])"; done
