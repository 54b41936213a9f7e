"""Key figures of the bench's JSON line(s) on stdin, one row per line."""
import json, sys
for l in sys.stdin:
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    st = d.get("stage_ms_per_step", {})
    print("value %.2f  ms/step %.1f  one-call %.2f  frac %.3f  launch %.1f ms  host %.3f s (sys %s)  busy %.2f  call %.0f ms  dev-records %.3f  parity %s %s  disturbances %s  work %s" % (
        d["value"], d["ms_per_step"], (d.get("one_call_in_flight") or {}).get("value", 0), d["roofline"]["frac"], d["roofline"].get("launch_ms", 0),
        d.get("host_cpu_s_per_step", 0), d.get("host_sys_s_per_step"), d.get("host_cpu_busy_frac", 0), st.get("total_ms", 0),
        d.get("sam_records_written_by_device_frac", 0), d.get("parity_on_sample"), d.get("all_steps_identical_to_checked_sam"), list(d.get("timed_region_disturbances", {}).values()), d.get("work_per_step")))
