import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
sys.argv = ["scene_bench.py", "--bias", "-3.0", "--wscale", "24", "--iters", "1"]
import runpy
import sam_road_amd.inferencer as inf
orig = inf.edge_votes
pr = cProfile.Profile()
calls = [0]
def wrapped(*a, **k):
    calls[0] += 1
    if calls[0] == 2:
        pr.enable(); r = orig(*a, **k); pr.disable(); return r
    return orig(*a, **k)
inf.edge_votes = wrapped
runpy.run_path("tools/scene_bench.py", run_name="__main__")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
