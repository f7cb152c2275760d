"""gpurun_out/pmc_{fetch,write,sq}/p_counter_collection.csv -> profiles/<tag>_pmc_traffic.json

Per-dispatch means for every km_* kernel.  FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts
64 B per 128-B request on wide coalesced reads and is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    path = os.path.join(root, "gpurun_out", name, "p_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].strip()
        if "km_" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


fetch, write, sq = load("pmc_fetch"), load("pmc_write"), load("pmc_sq")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes), `python bench.py` B=256x3x512x512; "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); per-dispatch means",
       "kernels": {}}
mean = lambda v: sum(v) / len(v)
for k in sorted(set(fetch) | set(write)):
    rec = {}
    if "FETCH_SIZE" in fetch.get(k, {}):
        rec["FETCH_SIZE_KiB_raw"] = round(mean(fetch[k]["FETCH_SIZE"]), 1)
        rec["hbm_read_bytes_corrected"] = int(2 * 1024 * mean(fetch[k]["FETCH_SIZE"]))
    if "WRITE_SIZE" in write.get(k, {}):
        rec["WRITE_SIZE_KiB_raw"] = round(mean(write[k]["WRITE_SIZE"]), 1)
        rec["hbm_write_bytes"] = int(1024 * mean(write[k]["WRITE_SIZE"]))
    if "hbm_read_bytes_corrected" in rec and "hbm_write_bytes" in rec:
        rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_corrected"] + rec["hbm_write_bytes"]
    for c, key in (("SQ_INSTS_VALU", "valu_wave_instr"), ("SQ_INSTS_SALU", "salu_wave_instr"), ("SQ_INSTS_LDS", "lds_wave_instr"),
                   ("SQ_INSTS_VMEM_RD", "vmem_rd_wave_instr"), ("SQ_INSTS_VMEM_WR", "vmem_wr_wave_instr"), ("SQ_WAVES", "waves")):
        if c in sq.get(k, {}):
            rec[key] = mean(sq[k][c])
    out["kernels"][k] = rec
path = os.path.join(root, "profiles", f"{tag}_pmc_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
for k, r in out["kernels"].items():
    print(k[:60], r.get("hbm_bytes_per_launch"))
