"""Per-launch timeline of ONE gpbo_lml evaluation out of a rocprofv3 kernel trace (scripts/r06_lml_evidence.sh):
start (us since the evaluation's first kernel), duration, gap to the previous kernel's end, queue, kernel, grid.

    python scripts/r06_lml_timeline.py gpurun_out/r06_lml/trace_4096/t_kernel_trace.csv > profiles/r06_lml_4096_timeline.txt
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "prescale" in r["Kernel_Name"] or "mid_inputs" in r["Kernel_Name"]]
s = first[-1]
t0 = int(rows[s]["Start_Timestamp"])
prev_end = t0
groups = {}
print(f"# last evaluation of {sys.argv[1]}: start_us dur_us gap_us queue kernel grid")
for r in rows[s:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("gpbo::", "").split("(")[0].replace("void ", "")[:44]
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {(st - prev_end) / 1e3:7.1f} q{r.get('Queue_Id', '')} {name} "
          f"{r.get('Grid_Size_X', '')}x{r.get('Grid_Size_Y', '')}x{r.get('Grid_Size_Z', '')}/{r.get('Workgroup_Size_X', '')}")
    prev_end = max(prev_end, en)
    groups[name] = groups.get(name, 0.0) + (en - st) / 1e3
print(f"# total {(prev_end - t0) / 1e3:.1f} us; kernel time by name:")
for k, v in sorted(groups.items(), key=lambda kv: -kv[1]):
    print(f"#   {v:9.1f} us  {k}")
