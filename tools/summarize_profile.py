#!/usr/bin/env python
"""Turn the rocprofv3 outputs merged back under gpurun_out/ into the committed summaries under profiles/.

    python tools/summarize_profile.py r01 gpurun_out/prof_r01 gpurun_out/pmc_fetch_r01 gpurun_out/pmc_write_r01

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary of `bench.py`) and
profiles/<tag>_hbm_traffic.json: per kernel, mean HBM bytes per launch from the PMC passes
(FETCH_SIZE and WRITE_SIZE collected in SEPARATE --pmc passes; both are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads, so it is doubled — MI355X_MICROARCH.md §HBM)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(path, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"][:r["Kernel_Name"].rfind("(")].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    tag, prof, fetch, write = sys.argv[1:5]
    workload = sys.argv[5] if len(sys.argv) > 5 else "encdec"
    sfx = "" if workload == "encdec" else "_" + workload          # bench.py picks the traffic summary of ITS workload by this suffix
    os.makedirs("profiles", exist_ok=True)
    for f in glob.glob(os.path.join(prof, "*kernel_stats.csv")):
        shutil.copy(f, f"profiles/{tag}_kernel_stats{sfx}.csv")
    fe, wr = mean_by_kernel(fetch, "FETCH_SIZE"), mean_by_kernel(write, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith("srh::"):
            continue
        f_kib, n = fe.get(k, (0.0, 0))
        w_kib, _ = wr.get(k, (0.0, 0))
        out[k] = {"launches_sampled": n, "fetch_bytes_per_launch": 2 * f_kib * 1024, "write_bytes_per_launch": w_kib * 1024,
                  "hbm_bytes_per_launch": 2 * f_kib * 1024 + w_kib * 1024}
    gem = [v for k, v in out.items() if "gemm" in k]
    tot_n = sum(v["launches_sampled"] for v in gem)
    if tot_n:
        out["_gemm_all"] = {"launches_sampled": tot_n,
                            "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in gem) / tot_n}
    # stamp the summary with the build it was measured on (tools/profile_gpu.sh wrote it on the GPU box): bench.py refuses a
    # summary of another build instead of quoting stale traffic
    bid_file = os.path.join(os.path.dirname(os.path.normpath(prof)), f"{tag}{sfx}_build_id.txt")
    out["_build_id"] = open(bid_file).read().strip() if os.path.exists(bid_file) else None
    json.dump(out, open(f"profiles/{tag}_hbm_traffic{sfx}.json", "w"), indent=1)
    for k, v in out.items():
        if not isinstance(v, dict):
            continue
        print(f"{k:60s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch (n={v['launches_sampled']})")


if __name__ == "__main__":
    main()
