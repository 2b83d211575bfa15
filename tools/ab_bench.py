"""Same-box A/B of whole trees: runs `bench.py` of every tree alternately (rounds x trees) and prints tiles/s and the per-class GPU times.
    python tools/ab_bench.py --rounds 2 --workload encdec .ab/r05 .
Trees are checkouts / exports of this repository with their library built in place (git archive <commit> | tar -x -C .ab/<name>;
python -m sam_road_amd.build inside).  .ab/ is git-ignored but travels to the GPU box with gpurun."""
import argparse
import json
import os
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("trees", nargs="+")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--workload", default="encdec")
ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()
res = {t: [] for t in args.trees}
for r in range(args.rounds):
    for t in args.trees:
        env = {k: v for k, v in os.environ.items() if k != "SRH_LIB_PATH"}
        out = subprocess.run([sys.executable, "bench.py", "--workload", args.workload, "--steps", str(args.steps), "--warmup", "5",
                              "--no-cpu-baseline", "--no-reference-gpu", "--no-sustained", "--no-scene", "--no-workloads"],
                             capture_output=True, text=True, env=env, cwd=t)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"{t} round {r}: FAILED\n{out.stderr[-2000:]}", flush=True)
            continue
        js = json.loads(line[-1])
        cls = js.get("roofline", {}).get("by_class_ms_per_step", {})
        res[t].append(js["value"])
        print(f"{t} round {r}: {js['value']:.1f} tiles/s  dominant {js.get('roofline', {}).get('dominant_kernel', {}).get('frac')}  {cls}", flush=True)
for t, v in res.items():
    if v:
        print(f"{t}: mean {sum(v) / len(v):.1f} tiles/s over {len(v)} runs ({', '.join('%.1f' % x for x in v)})")
