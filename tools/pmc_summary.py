"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, kernel-trace only) into
profiles/<name>.json: per-kernel per-launch averages and the corrected HBM traffic

    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024      [gfx950: FETCH_SIZE counts 128-B requests at 64 B, MI355X_MICROARCH.md]

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> "<command that was profiled>"
"""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            acc[m.group(1) if m else r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch_dir, write_dir, out, cmd = sys.argv[1:5]
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in fe:
        f = sum(fe[k]) / len(fe[k])
        w = sum(wr[k]) / len(wr[k]) if k in wr else 0.0
        kernels[k] = {"FETCH_SIZE_KB_avg": round(f, 1), "WRITE_SIZE_KB_avg": round(w, 1), "launches": len(fe[k]),
                      "traffic_bytes_corrected": int((2 * f + w) * 1024)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `%s`; per-launch averages; traffic = "
                       "(2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction of MI355X_MICROARCH.md (HBM section)" % cmd,
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["traffic_bytes_corrected"])[:12]:
        print(f"{k:40s} {v['launches']:4d}  {v['traffic_bytes_corrected'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
