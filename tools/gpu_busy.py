"""GPU busy fraction from a rocprofv3 kernel trace: union of the kernel intervals inside the busiest 1-second window of the
run (the timed region of bench.py), and the kernel time per name inside that window."""
import csv, sys
W = 1_000_000_000
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
iv = [x for x in iv if "sa_expand" not in x[2] and "bucket" not in x[2] and "radix" not in x[2].lower() and "round" not in x[2] and "flag_kernel" not in x[2]]


def union(lo, hi):
    busy, cs, ce = 0, None, None
    for s, e, _ in iv:
        s, e = max(s, lo), min(e, hi)
        if s >= e:
            continue
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce is not None:
        busy += ce - cs
    return busy


t_end = max(e for _, e, _ in iv)
best, best_lo = -1, 0
lo = iv[0][0]
while lo + W <= t_end:
    b = union(lo, lo + W)
    if b > best:
        best, best_lo = b, lo
    lo += W // 10
per = {}
for s, e, n in iv:
    s2, e2 = max(s, best_lo), min(e, best_lo + W)
    if s2 < e2:
        k = n.split("(")[0].split("::")[-1][:24]
        per[k] = per.get(k, 0) + (e2 - s2)
print("busiest 1 s window: GPU busy %.1f %%" % (100.0 * best / W))
print("kernel ms inside it:", {k: round(v / 1e6, 1) for k, v in sorted(per.items(), key=lambda x: -x[1])[:14]})
