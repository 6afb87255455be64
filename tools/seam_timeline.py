#!/usr/bin/env python3
"""Start / end of every traversal kernel of the last seam calls (rk_acc_pot into pinned arrays) and the last device-resident steps of a
bench.py run traced with rocprofv3 --kernel-trace: usage seam_timeline.py <dir with *_kernel_trace.csv>. Prints, per call, each kernel's
start and end relative to the call's first kernel (us)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_super" in n or "k_list" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void rk::", "")[:34], r.get("Stream_Id", "?")))
rows.sort()
# group into calls: a call starts with k_super
calls, cur = [], []
for r in rows:
    if "k_super" in r[2] and cur:
        calls.append(cur); cur = []
    cur.append(r)
if cur: calls.append(cur)
print("calls traced:", len(calls))
def show(c, tag):
    t0 = c[0][0]
    print(tag, "span %.1f us" % ((max(r[1] for r in c) - t0) / 1e3))
    for r in c:
        print("   %-36s stream %-4s start %8.1f end %8.1f dur %8.1f" % (r[2], r[3], (r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3))
# bench order: ... warmup + timed seam calls come BEFORE? print a few from the middle and the end
for i in (len(calls) // 2, len(calls) // 2 + 1, len(calls) - 2, len(calls) - 1):
    if 0 <= i < len(calls): show(calls[i], "call %d" % i)
