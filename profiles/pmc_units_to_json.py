"""Unit counters of the hot kernels -> profiles/<tag>_pmc_units.json.
  profiles/pmc_units_to_json.py <tag> [<dir>]
<dir> (default gpurun_out/<tag>... of profiles/pmc_units.sh: one sub-directory pmc_<pass>/ per rocprofv3 --pmc pass, each with the
km_* rows of its counter_collection csv) - or, without <dir>, round 2's layout gpurun_out/pmc_{sq,sq2,ta,tcp,grbm}/ of profiles/collect.sh.

Per-dispatch means of the SQ / TA / TCP / TCC / GRBM counters of every km_* kernel (template arguments kept: the blur's forward and adjoint
are different rows) and the fractions DESIGN.md quotes from them:
  kernel_cycles                         GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)
  TA_busy_frac                          TA_TA_BUSY_sum / 256 CUs / kernel_cycles
  TA_busy_cycles_per_wave_instruction   TA_TA_BUSY_sum / TA_FLAT_WAVEFRONTS_sum
  TCP_pending_stall_frac                TCP_PENDING_STALL_CYCLES_sum / 256 / kernel_cycles
  wave_time_waiting_frac                SQ_WAIT_ANY / SQ_WAVE_CYCLES
  wave_time_issue_stalled_frac          SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  wave_time_valu_frac                   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (x waves per SIMD = how busy the vector ALUs are)
  mean_waves_per_CU                     SQ_WAVE_CYCLES x 4 / kernel_cycles / 256 (occupancy actually reached, against `max_waves_per_CU`)
  VALU / SALU / LDS / VMEM instructions per wave
  HBM_GB                                FETCH_SIZE x 2 (the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = []
if len(sys.argv) > 2:
    d = sys.argv[2]
    files = sorted(glob.glob(os.path.join(d, "pmc_*", "*.km")) + glob.glob(os.path.join(d, "pmc_*", "*", "*.km")) + glob.glob(os.path.join(d, "pmc_*", "*counter_collection.csv")))
else:
    files = [os.path.join(root, "gpurun_out", d, "p_counter_collection.csv") for d in ("pmc_sq", "pmc_sq2", "pmc_ta", "pmc_tcp", "pmc_grbm")]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for path in files:
    if not os.path.exists(path):
        print("missing", path)
        continue
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0].strip()
        if "km_" in k and any(s in k for s in ("warp", "blur", "points")):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"workgroup_size": int(r["Workgroup_Size"]), "lds_bytes": int(r["LDS_Block_Size"]), "scratch_bytes": int(r["Scratch_Size"]), "grid_size": int(r["Grid_Size"])}
out = {"note": "rocprofv3 --kernel-trace --pmc passes (separate, profiles/pmc_units.sh) of profiles/pmc_step.py: BASELINE config 2's four launches (256x3x512x512 fp32) and "
               "transform_points at 2 x 1.07 GB as the control; per-dispatch means.  SQ_* cycle counters are quad-cycles summed over waves; TA_* / TCP_* / TCC_* are summed over "
               "the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Kernels run SERIALISED under counter collection and a few per cent slower than in the bench.",
       "kernels": {}}
for k in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    if c.get("GRBM_GUI_ACTIVE", 0) / 8 < 50000:  # (the small launches: chain, boxes)
        continue
    rec = {n: int(round(v)) for n, v in sorted(c.items())}
    der = dict(meta[k])
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    der["kernel_cycles"] = int(cyc)
    if "TA_TA_BUSY_sum" in c:
        der["TA_busy_frac"] = round(c["TA_TA_BUSY_sum"] / 256 / cyc, 3)
    if c.get("TA_FLAT_WAVEFRONTS_sum"):
        der["TA_busy_cycles_per_wave_instruction"] = round(c.get("TA_TA_BUSY_sum", 0.0) / c["TA_FLAT_WAVEFRONTS_sum"], 1)
    if "TCP_PENDING_STALL_CYCLES_sum" in c:
        der["TCP_pending_stall_frac"] = round(c["TCP_PENDING_STALL_CYCLES_sum"] / 256 / cyc, 3)
    if "TCP_TCP_TA_DATA_STALL_CYCLES_sum" in c:
        der["TCP_TA_data_stall_frac"] = round(c["TCP_TCP_TA_DATA_STALL_CYCLES_sum"] / 256 / cyc, 3)
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for n, key in (("SQ_WAIT_ANY", "wave_time_waiting_frac"), ("SQ_WAIT_INST_ANY", "wave_time_issue_stalled_frac"), ("SQ_ACTIVE_INST_ANY", "wave_time_issuing_frac"),
                       ("SQ_ACTIVE_INST_VALU", "wave_time_valu_frac"), ("SQ_ACTIVE_INST_LDS", "wave_time_lds_frac")):
            if n in c:
                der[key] = round(c[n] / wc, 3)
        der["mean_waves_per_CU"] = round(wc * 4 / cyc / 256, 1)
    if c.get("SQ_WAVES"):
        w = c["SQ_WAVES"]
        der["waves"] = int(w)
        for n, key in (("SQ_INSTS_VALU", "VALU_instr_per_wave"), ("SQ_INSTS_SALU", "SALU_instr_per_wave"), ("SQ_INSTS_LDS", "LDS_instr_per_wave"),
                       ("SQ_INSTS_VMEM_RD", "VMEM_RD_instr_per_wave"), ("SQ_INSTS_VMEM_WR", "VMEM_WR_instr_per_wave")):
            if n in c:
                der[key] = round(c[n] / w, 1)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        der["HBM_read_GB"] = round(c["FETCH_SIZE"] * 2 * 1024 / 1e9, 3)
        der["HBM_write_GB"] = round(c["WRITE_SIZE"] * 1024 / 1e9, 3)
    if c.get("TCC_REQ_sum"):
        der["L2_hit_frac"] = round(c.get("TCC_HIT_sum", 0.0) / c["TCC_REQ_sum"], 3)
    rec["derived"] = der
    out["kernels"][k] = rec
path = os.path.join(root, "profiles", f"{tag}_pmc_units.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, len(out["kernels"]), "kernels")
for k, r in out["kernels"].items():
    print(k[:70], {kk: vv for kk, vv in r["derived"].items() if kk not in ("grid_size", "workgroup_size")})
