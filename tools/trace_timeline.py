"""Timeline of ONE graph-replayed step from a rocprofv3 --kernel-trace CSV of `bench.py`: every kernel with its queue, start offset and duration,
plus the time during which exactly one / several queues were busy.  Step = between two consecutive k_fgw_small_vectors / k_fgw_init launches
(the median-length one).  usage: python tools/trace_timeline.py <dir with *kernel_trace.csv> [min_us_to_list]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*", "", n)[:52]
idx = [i for i, r in enumerate(rows) if "k_fgw_init" in r["Kernel_Name"]] or [i for i, r in enumerate(rows) if "k_fgw_small_vectors" in r["Kernel_Name"]]
steps = sorted((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]), a, b) for a, b in zip(idx[:-1], idx[1:]))
L, a, b = steps[len(steps) // 4]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
print("step %.1f us, %d kernels" % (L / 1e3, len(seg)))
queues = sorted({r["Queue_Id"] for r in seg})
prev_end = {q: t0 for q in queues}
for r in seg:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    if (e - s) / 1e3 >= min_us:
        print("%8.1f  q%-2s %7.1f us  (queue idle before: %6.1f)  %s" % ((s - t0) / 1e3, q, (e - s) / 1e3, (s - prev_end[q]) / 1e3, short(r["Kernel_Name"])))
    prev_end[q] = max(prev_end[q], e)
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in seg] + [(int(r["End_Timestamp"]), -1) for r in seg])
depth, last, hist = 0, t0, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last); last = t; depth += d
print("concurrency histogram (us with k kernels in flight):", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
