"""Summarise a rocprofv3 --pmc pass with SQ counters into per-kernel per-launch averages and an MFMA-busy fraction.

    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles),   kernel cycles = duration_ns x 2.4 GHz

SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over every SIMD of the chip (MI355X_MICROARCH.md, "rocprofv3 PMC
slots": = 32 x N_mfma for v_mfma_f32_32x32x16_bf16), so dividing by the SIMD-cycles the launch lasted gives the fraction of
matrix-pipe issue slots used.  Durations come from the kernel trace of the same pass.

usage: python tools/pmc_sq.py <pmc_dir> <out.json> "<command that was profiled>"
"""
import collections
import csv
import glob
import json
import re
import sys

CLOCK_GHZ, SIMDS = 2.4, 4 * 256


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    d, out, cmd = sys.argv[1:4]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    kernels = {}
    for k, c in vals.items():
        rec = {name: round(sum(v) / len(v), 1) for name, v in c.items()}
        rec["launches"] = max(len(v) for v in c.values())
        if dur.get(k):
            ns = sum(dur[k]) / len(dur[k])
            rec["avg_duration_us_under_pmc"] = round(ns / 1e3, 2)
            # kernel cycles: GRBM_GUI_ACTIVE is summed over the 8 XCDs (it gives ~2.1-2.3 GHz under load on this part); without it the
            # nominal 2.4 GHz is assumed, which understates every fraction below
            cyc = min(rec["GRBM_GUI_ACTIVE"] / 8.0, ns * CLOCK_GHZ) if rec.get("GRBM_GUI_ACTIVE") else ns * CLOCK_GHZ     # (the counter also runs a little before / after a short kernel)
            rec["clock_ghz"] = round(cyc / ns, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in rec:
                rec["mfma_busy_frac"] = round(rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc), 4)
            if "SQ_INSTS_VALU" in rec:
                rec["valu_issue_frac"] = round(4.0 * rec["SQ_INSTS_VALU"] / (SIMDS * cyc), 4)        # a wave64 VALU instruction occupies its SIMD for 4 cycles
            if "SQ_VALU_MFMA_COEXEC_CYCLES" in rec and rec.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                rec["mfma_cycles_with_valu_beside"] = round(rec["SQ_VALU_MFMA_COEXEC_CYCLES"] / rec["SQ_VALU_MFMA_BUSY_CYCLES"], 4)
        kernels[k] = rec
    json.dump({"note": "rocprofv3 --kernel-trace --pmc <SQ counters> on `%s`; per-launch averages; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                       "(1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; valu_issue_frac = 4 x SQ_INSTS_VALU / the same" % cmd, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:14]:
        print(f"{k:40s} {v['launches']:4d}  mfma_busy={v.get('mfma_busy_frac')}  valu_issue={v.get('valu_issue_frac')}  clock={v.get('clock_ghz')} GHz  dur={v.get('avg_duration_us_under_pmc')} us")


if __name__ == "__main__":
    main()
