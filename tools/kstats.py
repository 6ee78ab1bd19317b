"""Print the top rows of a rocprofv3 --stats kernel CSV (name, calls, average us, share).  usage: python tools/kstats.py DIR_OR_CSV [N]"""
import csv, glob, os, re, sys
p = sys.argv[1]
f = p if os.path.isfile(p) else glob.glob(p + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)[:56]
    print("%-56s %6d %9.1f us %5.1f%%" % (n, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
