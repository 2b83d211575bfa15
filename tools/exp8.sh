python -m pytest tests/test_gpu_scene.py tests/test_refrun_golden.py tests/test_gpu_fullsize.py -m gpu -q -x -k "scene or infer_one_img or cli" 2>&1 | tail -4
SRH_PROFILE_HOST=1 python tools/scene_bench.py --iters 4 2>&1 | tail -14
