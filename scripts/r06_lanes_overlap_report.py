"""Overlap of the lanes of a gpbo_lml_batch call, out of a rocprofv3 kernel trace of `scripts/r06_lanes_overlap.py trace`:
for the last two-lane and the last six-lane call — span, the summed kernel time per queue, and how much of the span had 0 / 1 / 2+
kernels in flight.

    python scripts/r06_lanes_overlap_report.py gpurun_out/r06_lanes/t_kernel_trace.csv
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# calls are separated by host gaps > 200 us without any kernel
calls, cur, last_end = [], [], None
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and st - last_end > 200_000 and cur:
        calls.append(cur)
        cur = []
    cur.append(r)
    last_end = en if last_end is None else max(last_end, en)
if cur:
    calls.append(cur)


def report(call):
    t0 = min(int(r["Start_Timestamp"]) for r in call)
    t1 = max(int(r["End_Timestamp"]) for r in call)
    per_q = {}
    ev = []
    for r in call:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r.get("Queue_Id", "?")
        per_q.setdefault(q, [0.0, 0, st, en])
        per_q[q][0] += (en - st) / 1e3
        per_q[q][1] += 1
        per_q[q][2] = min(per_q[q][2], st)
        per_q[q][3] = max(per_q[q][3], en)
        ev.append((st, 1))
        ev.append((en, -1))
    ev.sort()
    depth, prev, hist = 0, t0, {}
    for t, dlt in ev:
        hist[min(depth, 3)] = hist.get(min(depth, 3), 0.0) + (t - prev) / 1e3
        depth += dlt
        prev = t
    print(f"call of {len(call)} kernels on {len(per_q)} queues: span {(t1 - t0) / 1e3:.1f} us")
    for q, (busy, n, a, b) in sorted(per_q.items()):
        print(f"   queue {q}: {n:4d} kernels, {busy:8.1f} us of kernel time, active from {(a - t0) / 1e3:8.1f} to {(b - t0) / 1e3:8.1f} us")
    print("   kernels in flight (us of the span):", {("3+" if k == 3 else k): round(v, 1) for k, v in sorted(hist.items())})


sizes = sorted({len(c) for c in calls})
print(f"{len(calls)} calls, kernel counts {sizes}")
big = [c for c in calls if len(c) > 50]
if big:
    half = len(big) // 2
    print("== last two-lane call")
    report(big[half - 1])
    print("== last six-lane call")
    report(big[-1])
