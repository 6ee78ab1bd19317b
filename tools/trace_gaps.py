"""Idle time inside graph-replayed steps: from a rocprofv3 --kernel-trace CSV of `bench.py` (graph replay), take the steps between consecutive
k_fgw_init (N > 64) / k_fgw_small_vectors (N <= 64) launches in the timed region and report step length, union-of-kernels busy time and the largest gaps."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*", "", n)[:44]
idx = [i for i, r in enumerate(rows) if "k_fgw_init" in r["Kernel_Name"]]
if not idx:                                  # N <= 64: the per-solve vector kernel initialises the molecules (one launch per step)
    idx = [i for i, r in enumerate(rows) if "k_fgw_small_vectors" in r["Kernel_Name"]]
steps = []
for a, b in zip(idx[:-1], idx[1:]):
    steps.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]), a, b))
# the timed graph replays are the 20 most regular consecutive steps: take the median-length ones
steps.sort()
L, a, b = steps[len(steps) // 4]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e - t0)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("step %.1f us, %d kernels, busy (union) %.1f us, idle %.1f us in %d gaps" % (L / 1e3, len(seg), busy / 1e3, (L - busy) / 1e3, len(gaps)))
for g, at in sorted(gaps, reverse=True)[:12]:
    j = max(i for i, r in enumerate(seg) if int(r["End_Timestamp"]) - t0 <= at + 1)
    print("  gap %6.1f us at %7.1f us after %s" % (g / 1e3, at / 1e3, short(seg[j]["Kernel_Name"])))
