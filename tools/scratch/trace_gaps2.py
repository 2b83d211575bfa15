#!/usr/bin/env python
"""Scratch: timeline between the end of one scene's pass 1 (scene_norm_kernel) and the start of the next (patch_im2col) in the
pipelined loop, from rocprofv3 --kernel-trace --memory-copy-trace CSVs.  usage: trace_gaps2.py <kernel_trace.csv> <memory_copy_trace.csv>"""
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:60]))
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"C {r.get('Direction', '')} {r.get('Bytes', r.get('Size', ''))}"))
ev.sort()
norms = [i for i, e in enumerate(ev) if "scene_norm_kernel" in e[2]]
print("scenes", len(norms))
for si in (len(norms) - 6, len(norms) - 5):
    i = norms[si]
    j = next(k for k in range(i + 1, len(ev)) if "patch_im2col" in ev[k][2])
    t0 = ev[i][1]
    prev_im2col = max(k for k in range(i) if "patch_im2col" in ev[k][2] and all("scene_norm" not in ev[m][2] for m in range(k, i)) and "patch_im2col" in ev[k][2]) if False else None
    print(f"--- after scene {si}: end of scene_norm -> next patch_im2col = {(ev[j][0] - t0) / 1e6:.2f} ms; events in between (ms after scene_norm end):")
    for s, e, n in ev[i + 1:j + 1]:
        if (e - s) > 20000 or n.startswith("C"):
            print(f"   +{(s - t0) / 1e6:7.3f} .. +{(e - t0) / 1e6:7.3f}  ({(e - s) / 1e3:8.1f} us)  {n}")
    busy = sum(e - s for s, e, n in ev[i + 1:j] if n.startswith("K"))
    print(f"   kernel busy in the window {busy / 1e6:.2f} ms")
# period
starts = [ev[i][1] for i in norms]
print("period (ms) between scene_norm ends:", " ".join(f"{(b - a) / 1e6:.1f}" for a, b in zip(starts, starts[1:])))
