"""Reads a rocprofv3 kernel trace (CSV) and prints, for the condensed CG, the gap between the end of k_trace_pre_tile and the
start of the following k_trace_post_tile, and how the deferred p / x update (k_cg_sr_update_xp, second stream) sits in it."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void hdg::", "").replace("hdg::", "")))
rows.sort()
pre = [r for r in rows if r[2].startswith("k_trace_pre_tile")]
post = [r for r in rows if r[2].startswith("k_trace_post_tile")]
xp = [r for r in rows if r[2].startswith("k_cg_sr_update_xp")]
n = min(len(pre), len(post))
skip = n // 3
gaps = [post[i][0] - pre[i][1] for i in range(skip, n) if post[i][0] > pre[i][1]]
its = [pre[i + 1][0] - pre[i][0] for i in range(skip, n - 1) if pre[i + 1][0] - pre[i][0] < 2e6]
print(f"{n} preconditioner applications; pre end -> post start: median {sorted(gaps)[len(gaps)//2]/1e3:.1f} us; pre start -> next pre start: median {sorted(its)[len(its)//2]/1e3:.1f} us")
if xp:
    d = sorted(r[1] - r[0] for r in xp[len(xp)//3:])
    print(f"{len(xp)} k_cg_sr_update_xp launches, median duration {d[len(d)//2]/1e3:.1f} us")
# one iteration in detail (the last third of the run)
i = (2 * n) // 3
t0, t1 = pre[i][0], pre[i + 1][0]
for r in rows:
    if t0 <= r[0] < t1: print(f"  {(r[0]-t0)/1e3:8.1f} .. {(r[1]-t0)/1e3:8.1f} us  {r[2]}")
