"""Where does pass 2 (edge_votes) spend its time on the GPU box?  Runs tools/scene_bench.py's scene once, then times the
sections of sam_road_amd.inferencer.edge_votes by monkey-patching its collaborators (host query builder, TopoNet launch,
score fetch, vote accumulation)."""
import os, sys, time, runpy
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import sam_road_amd.inferencer as inf
from sam_road_amd import _lib

acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    return w

orig_ev = inf.edge_votes
calls = [0]
def ev(net, *a, **k):
    calls[0] += 1
    if calls[0] < 2:
        return orig_ev(net, *a, **k)
    acc.clear()
    inf.build_all_patch_queries = timed("build_all_patch_queries", inf.build_all_patch_queries)
    inf._collate = timed("collate", inf._collate)
    net_topo = net.infer_toponet
    net.infer_toponet = timed("infer_toponet (launch)", net_topo)
    lib = _lib.load()
    torch.cuda.synchronize(); t = time.perf_counter()
    r = orig_ev(net, *a, **k)
    torch.cuda.synchronize(); total = time.perf_counter() - t
    net.infer_toponet = net_topo
    print("edge_votes total %.1f ms: " % (total * 1e3) + ", ".join(f"{n} {v * 1e3:.1f}" for n, v in acc.items()), flush=True)
    return r
inf.edge_votes = ev
sys.argv = ["scene_bench.py", "--iters", "1"]
runpy.run_path("tools/scene_bench.py", run_name="__main__")
