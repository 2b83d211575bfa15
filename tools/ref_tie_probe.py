"""Does the REFERENCE's own edge set depend on the order rtree.intersection returns ids in?  (build container only: imports
/root/reference verbatim under tests/golden/ref_stubs.py.)  Finding (round 2): yes, once a scene is dense enough for kNN cut-off ties —
447 nodes: 3 / 4 of 3632 edges differ between the ascending / descending / shuffled enumerations; 275 nodes: none.
    python tools/ref_tie_probe.py
"""
import io, os, sys, contextlib, tempfile
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import ref_stubs, make_golden_refrun as mg
from oracle.samroad import SAMRoadOracle, AttrDict
from oracle.synth import synth_scene
ref_stubs.install()
tmp = tempfile.mkdtemp(); ck = os.path.join(tmp, "sam.pth"); mg.fake_sam_checkpoint(ck)
sys.path.insert(0, "/root/reference"); sys.argv = ["inferencer.py", "--device", "cpu"]
with contextlib.redirect_stdout(io.StringIO()):
    import model as ref_model, inferencer as ref_inf
Dict = ref_stubs.Dict
cfg = dict(mg.SCENE_CFG)
with contextlib.redirect_stdout(io.StringIO()):
    net = ref_model.SAMRoad(Dict(dict(cfg, SAM_CKPT_PATH=ck)))
net.load_state_dict(mg.scene_state_dict(SAMRoadOracle(AttrDict(cfg)), mg.SCENE_WSEED), strict=True); net.eval()
img = synth_scene(mg.SCENE_SIZE, seed=mg.SCENE_SEED)
_, _, kp0, road0 = ref_inf.infer_one_img(net, img, Dict(cfg))
for pct in ((99.5, 98.0), (97.0, 90.0), (95.0, 80.0)):
    c = dict(cfg); c["ITSC_THRESHOLD"] = float(np.percentile(kp0[kp0 > 0], pct[0])) / 255.0; c["ROAD_THRESHOLD"] = float(np.percentile(road0[road0 > 0], pct[1])) / 255.0
    sets = {}
    for mode in ("ascending", "descending", "shuffled"):
        ref_stubs.RTREE_ORDER["mode"] = mode
        nodes, edges, _, _ = ref_inf.infer_one_img(net, img, Dict(c))
        sets[mode] = {tuple(e) for e in np.asarray(edges).tolist()}
    print(pct, "nodes", nodes.shape[0], "edges", {m: len(s) for m, s in sets.items()}, "sym diff vs ascending:", {m: len(sets[m] ^ sets["ascending"]) for m in sets})
