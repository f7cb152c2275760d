"""<prof dir of profiles/device_run.sh's `prof` stage> -> profiles/<tag>_pmc_traffic.json and profiles/<tag>_rocprofv3_kernel_stats.csv

  python profiles/pmc_prof_to_json.py gpurun_out/r04/runN_prof r04

Per-dispatch means of FETCH_SIZE / WRITE_SIZE (separate --pmc passes) for every km_* kernel.  FETCH_SIZE / WRITE_SIZE are reported in KiB; on
gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads and is doubled (/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3
section).  The json records the sha256 of the kernel sources the profiled library was built from (csrc.sha256, written on the GPU box by the
prof stage): bench.py only quotes the traffic while the tree's sources still hash to it."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

prof, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(counter):
    agg = collections.defaultdict(list)
    for path in glob.glob(os.path.join(prof, f"pmc_{counter}", "*counter_collection.csv*")):
        for r in csv.DictReader(open(path)):
            if "km_" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0].strip()].append(float(r["Counter_Value"]))
    return agg


fetch, write = load("FETCH_SIZE"), load("WRITE_SIZE")
sha = open(os.path.join(prof, "csrc.sha256")).read().strip() if os.path.exists(os.path.join(prof, "csrc.sha256")) else None
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/device_run.sh prof stage) of `python bench.py --no-cpu-baseline "
               "--no-extras --steps 2 --warmup 1 --groups 1` (plus its settle and per-kernel loops), B=256x3x512x512, 3 input sets rotated; FETCH_SIZE (KiB) doubled per "
               "/opt/skills/guides/MI355X_MICROARCH.md (gfx950); per-dispatch means",
       "csrc_sha256": sha, "source": prof, "kernels": {}}
mean = lambda v: sum(v) / len(v)
for k in sorted(set(fetch) | set(write)):
    rec = {"dispatches": len(fetch.get(k, write.get(k, [])))}
    if k in fetch:
        rec["FETCH_SIZE_KiB_raw"] = round(mean(fetch[k]), 1)
        rec["hbm_read_bytes_corrected"] = int(2 * 1024 * mean(fetch[k]))
    if k in write:
        rec["WRITE_SIZE_KiB_raw"] = round(mean(write[k]), 1)
        rec["hbm_write_bytes"] = int(1024 * mean(write[k]))
    if "hbm_read_bytes_corrected" in rec and "hbm_write_bytes" in rec:
        rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_corrected"] + rec["hbm_write_bytes"]
    out["kernels"][k] = rec
path = os.path.join(root, "profiles", f"{tag}_pmc_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, "csrc", sha)
for k, r in out["kernels"].items():
    print(f"  {k[:70]:70s} {r.get('hbm_bytes_per_launch')}")
stats = glob.glob(os.path.join(prof, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(root, "profiles", f"{tag}_rocprofv3_kernel_stats.csv"))
    print("copied", stats[0])
