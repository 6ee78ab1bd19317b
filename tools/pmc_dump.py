"""Per-kernel per-launch averages of every counter in one or more rocprofv3 --pmc output directories (+ kernel-trace durations).
usage: python tools/pmc_dump.py <name-filter-regex> <dir> [<dir> ...]"""
import collections, csv, glob, re, sys
flt = re.compile(sys.argv[1])
vals = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", n); return m.group(1) if m else n[:50]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)): vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)): dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(vals):
    if not flt.search(k): continue
    ns = sum(dur[k]) / max(1, len(dur[k]))
    print(f"{k}: {ns / 1e3:.1f} us avg over {len(dur[k])} launches")
    for c, v in sorted(vals[k].items()): print(f"    {c:32s} {sum(v) / len(v):16.1f}")
