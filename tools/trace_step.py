"""Kernel sequence of the LAST training step in a rocprofv3 --kernel-trace CSV (start offset, duration, queue, name)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*", "", n)[:58]
# a step starts with the collate / zero of the flat gradient buffer: find the last occurrence of the first kernel name of a step
names = [short(r["Kernel_Name"]) for r in rows]
key = "k_graph_ptr"
starts = [i for i, n in enumerate(names) if n.startswith(key)]
i0 = starts[-2] if len(starts) > 1 else 0           # two graph_ptr launches per step (radius + bond graph)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  %7.1f us  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?")[-3:], short(r["Kernel_Name"])))
    prev_end = max(prev_end, e)
