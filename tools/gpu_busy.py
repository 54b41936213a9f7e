"""GPU busy fraction from a rocprofv3 kernel trace: union of the kernel intervals after the index build / span."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_idx = max((e for s, e, n in iv if "sa_expand" in n or "p3_build" in n), default=iv[0][0])
iv = [x for x in iv if x[0] >= t_idx]
# the timed region = the longest stretch of calls: take everything after the index build
busy, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
per = {}
for s, e, n in iv:
    k = n.split("(")[0].split("::")[-1][:24]
    per[k] = per.get(k, 0) + (e - s)
print("span %.1f ms, busy %.1f ms (%.1f %%)" % (span / 1e6, busy / 1e6, 100.0 * busy / span))
print({k: round(v / 1e6, 1) for k, v in sorted(per.items(), key=lambda x: -x[1])[:8]})
