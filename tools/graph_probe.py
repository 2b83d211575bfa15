"""Is the encoder + decoder step launch-gap bound?  The same step (105 kernel launches on one stream) timed as plain stream launches
and as a replayed HIP graph (torch.cuda.CUDAGraph captures the library's launches: they go to torch's current stream).  Run on the
GPU box: python tools/graph_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_road_amd import Config, SAMRoad

dev = torch.device("cuda", 0)
cfg = Config(SAM_VERSION="vit_b", PATCH_SIZE=512, TOPONET_VERSION="normal", SAM_CKPT_PATH="", NO_SAM=False, USE_SAM_DECODER=False,
             ENCODER_LORA=False, NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
net = SAMRoad(cfg)
g = torch.Generator().manual_seed(1234)
sd = {k: (1.0 + 0.1 * torch.randn(v.shape, generator=g)) if (v.dim() == 1 and k.endswith("weight")) else 0.02 * torch.randn(v.shape, generator=g)
      for k, v in net.state_dict().items()}
net.load_state_dict(sd, strict=True)
net.eval().to(dev)
rgb = (torch.rand((16, 512, 512, 3), generator=torch.Generator().manual_seed(100)) * 255).round().to(dev)
step = lambda: net.infer_masks_and_img_features(rgb)

def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(5): step()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph, stream=s):
            out = step()
        ok = True
    except Exception as e:
        print("capture failed:", repr(e)[:300]); ok = False
    for rep in range(3):
        a = timed(step, 40)
        b = timed(graph.replay, 40) if ok else float("nan")
        print(f"stream launches {a:.4f} ms/step ({16e3 / a:.1f} tiles/s)   graph replay {b:.4f} ms/step ({16e3 / b:.1f} tiles/s)", flush=True)
