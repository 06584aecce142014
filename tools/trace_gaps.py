"""Idle gaps and slow launches in a rocprofv3 --kernel-trace CSV: python tools/trace_gaps.py <dir with *kernel_trace.csv> [min gap us]"""
import csv, glob, os, sys
d, thr = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
t0 = rows[0][0]
prev_end, last_mark = rows[0][1], None
print(f"{len(rows)} dispatches; gaps > {thr} us between the end of everything before and the next start; fused mark durations (ms)")
busy_until = rows[0][1]
for s, e, k in rows:
    if s - busy_until > thr * 1e3:
        print(f"  t={1e-6 * (busy_until - t0):9.2f} ms  IDLE {1e-6 * (s - busy_until):8.3f} ms  before {k.split('(')[0][-50:]}")
    busy_until = max(busy_until, e)
    if "mark_rgb8_kernel<true, true>" in k or "mark_rgb8_kernel<false, true>" in k:
        print(f"  t={1e-6 * (s - t0):9.2f} ms  mark {1e-6 * (e - s):.4f}")
