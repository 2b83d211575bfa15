#!/usr/bin/env python
"""Inter-kernel gaps of one bench.py run from a rocprofv3 --kernel-trace CSV: for the kernels of the timed steps, the time
between one kernel's end and the next one's start on the stream (dependent launches), per kernel class of the FOLLOWING kernel.
usage: tools/kernel_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
# last third of the trace = steady-state steps
ks = ks[len(ks) // 3:]
busy = sum(e - s for s, e, _ in ks)
span = ks[-1][1] - ks[0][0]
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    gaps[n1.split("(")[0][:70]].append(s1 - e0)
allg = [g for v in gaps.values() for g in v]
print(f"kernels {len(ks)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms ({100*busy/span:.1f} %)  gaps total {sum(allg)/1e6:.3f} ms, "
      f"mean {sum(allg)/len(allg)/1e3:.2f} us, median {sorted(allg)[len(allg)//2]/1e3:.2f} us, negative (overlap) {sum(g < 0 for g in allg)}")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {len(v):5d} x mean gap {sum(v)/len(v)/1e3:7.2f} us  max {max(v)/1e3:8.2f}  before  {k}")
