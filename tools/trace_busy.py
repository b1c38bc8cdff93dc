"""Reads a rocprofv3 kernel trace (CSV): GPU busy fraction (union of kernel intervals / span) over the last `frac` of the
run, the largest idle gaps and which kernels they sit between -- where the host leaves the queue empty."""
import csv, sys
from collections import Counter
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void hdg::", "").replace("hdg::", "")))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]
span = rows[-1][1] - rows[0][0]
busy = 0
end = rows[0][0]
gaps = []
for i, (s, e, n) in enumerate(rows):
    if s > end:
        gaps.append((s - end, rows[i - 1][2], n))
        busy += e - s
    elif e > end:
        busy += e - end
    end = max(end, e)
print(f"{len(rows)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms = {busy/span:.4f}; idle {(span-busy)/1e6:.2f} ms in {len(gaps)} gaps")
by = Counter()
cnt = Counter()
for g, a, b in gaps:
    by[(a, b)] += g
    cnt[(a, b)] += 1
for (a, b), t in by.most_common(25):
    print(f"  {t/1e3:9.1f} us in {cnt[(a,b)]:5d} gaps (avg {t/cnt[(a,b)]/1e3:6.1f})  {a[:40]:40s} -> {b[:40]}")
