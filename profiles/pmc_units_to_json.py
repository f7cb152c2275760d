"""gpurun_out/pmc_{sq,sq2,ta,tcp,grbm}/p_counter_collection.csv (profiles/collect.sh) -> profiles/<tag>_pmc_units.json

Per-dispatch means of the SQ / TA / TCP / GRBM counters of every km_* kernel and the fractions DESIGN.md quotes from them:
  kernel_cycles                         GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)
  TA_busy_frac                          TA_TA_BUSY_sum / 256 CUs / kernel_cycles
  TA_busy_cycles_per_wave_instruction   TA_TA_BUSY_sum / TA_FLAT_WAVEFRONTS_sum
  wave_time_waiting_frac                SQ_WAIT_ANY / SQ_WAVE_CYCLES
  wave_time_issue_stalled_frac          SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  VALU_instr_per_wave                   SQ_INSTS_VALU / SQ_WAVES"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_sq", "pmc_sq2", "pmc_ta", "pmc_tcp", "pmc_grbm"):
    path = os.path.join(root, "gpurun_out", d, "p_counter_collection.csv")
    if not os.path.exists(path):
        print("missing", path)
        continue
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].strip()
        if "km_" in k and ("warp" in k or "blur" in k):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc passes (separate, profiles/collect.sh) of `python bench.py --no-cpu-baseline --no-extras`, B=256x3x512x512; per-dispatch means. "
               "SQ_* cycle counters are quad-cycles summed over waves; TA_TA_BUSY_sum / TCP_* are cycles summed over the 256 CUs; GRBM_GUI_ACTIVE is summed over "
               "the 8 XCDs (divide by 8 for the kernel's duration in shader cycles).", "kernels": {}}
for k in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    rec = {n: int(round(v)) for n, v in sorted(c.items())}
    der = {}
    if "GRBM_GUI_ACTIVE" in c:
        der["kernel_cycles"] = int(c["GRBM_GUI_ACTIVE"] / 8)
        if "TA_TA_BUSY_sum" in c:
            der["TA_busy_frac"] = round(c["TA_TA_BUSY_sum"] / 256 / (c["GRBM_GUI_ACTIVE"] / 8), 3)
    if c.get("TA_FLAT_WAVEFRONTS_sum"):
        der["TA_busy_cycles_per_wave_instruction"] = round(c.get("TA_TA_BUSY_sum", 0.0) / c["TA_FLAT_WAVEFRONTS_sum"], 1)
    if c.get("SQ_WAVE_CYCLES"):
        if "SQ_WAIT_ANY" in c:
            der["wave_time_waiting_frac"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
        if "SQ_WAIT_INST_ANY" in c:
            der["wave_time_issue_stalled_frac"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    if c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
        der["VALU_instr_per_wave"] = int(c["SQ_INSTS_VALU"] / c["SQ_WAVES"])
    rec["derived"] = der
    out["kernels"][k] = rec
path = os.path.join(root, "profiles", f"{tag}_pmc_units.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, len(out["kernels"]), "kernels")
