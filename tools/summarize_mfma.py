#!/usr/bin/env python
"""MFMA-busy summary of a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE` pass (tools/profile_gpu.sh):
per kernel (and grid size) the per-launch means.  usage: tools/summarize_mfma.py gpurun_out/<tag>_pmc_mfma > profiles/<tag>_pmc_mfma.txt"""
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"][:r["Kernel_Name"].rfind("(")].replace("void ", "")
        if not name.startswith("srh::"):
            continue
        acc[(name, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE, bench.py --steps 6 (per-launch means)")
print("# MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)")
for (name, grid), c in sorted(acc.items()):
    m = lambda k: sum(c[k]) / max(len(c[k]), 1)
    gui, busy = m("GRBM_GUI_ACTIVE") / 8.0, m("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0
    print(f"{name + ' g=' + grid:56s} n={len(c['GRBM_GUI_ACTIVE']):4d} gui_active/xcd={gui:10.0f} clk  mfma_busy/simd={busy:9.0f} clk  mfma_busy_frac={busy / gui if gui else 0:.3f}")
