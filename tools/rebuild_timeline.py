#!/usr/bin/env python3
"""Timeline of one time step of examples/leapfrog traced with rocprofv3 --kernel-trace: usage
rebuild_timeline.py <dir with *_kernel_trace.csv> [step index from the end, default 3].
A step starts with k_kick_drift. Prints every kernel of the step (start relative to the step's first kernel, duration, gap since the
end of the previous kernel) and the totals: busy time, idle time in gaps, gaps above 2 us."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps, cur = [], []
for r in rows:
    if "k_kick_drift" in r[2] and cur:
        steps.append(cur); cur = []
    cur.append(r)
if cur: steps.append(cur)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
def short(n):
    n = n.replace("void ", "").replace("rk::bld::", "").replace("rk::", "")
    if "rocprim" in n:
        i = n.find("detail::")
        j = n.find("trampoline_kernel<")
        if j >= 0:
            n = "rocprim:" + n[j + 18:]
            k = n.find("detail::")
            n = "rocprim:" + (n[k + 8:] if k >= 0 else n)
        elif i >= 0:
            n = "rocprim:" + n[i + 8:]
    return n.split("(")[0][:60]
def summarize(s, show):
    t0 = s[0][0]
    prev_end = None
    busy = idle = 0
    big = 0
    first_list = None
    for a, b, n in s:
        gap = 0 if prev_end is None else max(0, a - prev_end)
        if first_list is None and ("k_list" in n or "k_pc" in n or "k_super" in n):
            first_list = a
        if show:
            print("  %-62s start %8.1f dur %7.1f gap %6.1f" % (short(n), (a - t0) / 1e3, (b - a) / 1e3, gap / 1e3))
        if first_list is None:
            busy += b - a
            idle += gap
            big += gap > 2000
        prev_end = b if prev_end is None else max(prev_end, b)
    return busy / 1e3, idle / 1e3, big, ((first_list or prev_end) - t0) / 1e3, (prev_end - t0) / 1e3
i = len(steps) - 1 - back
print("steps traced:", len(steps), "showing step", i)
summarize(steps[i], True)
print("per step, up to the first traversal kernel (kick_drift + rebuild): busy us / idle-in-gaps us / gaps > 2 us / span us ; whole step us")
for j in range(max(0, i - 4), min(len(steps), i + 5)):
    print("  step %3d: %8.1f %8.1f %4d %8.1f ; %8.1f   kernels %d" % ((j,) + summarize(steps[j], False) + (len(steps[j]),)))
