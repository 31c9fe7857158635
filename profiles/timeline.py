"""Per-launch timeline of the LAST bench step in a rocprofv3 kernel trace (…_kernel_trace.csv):
start (µs since the step's first kernel), duration, gap to the previous kernel, name, grid."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "bbox_count" in r["Kernel_Name"]][-1]
t0 = int(rows[first]["Start_Timestamp"])
prev = t0
for r in rows[first:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("o3dr::", "")
    print(f"{(st - t0) / 1e3:9.1f}  {(en - st) / 1e3:8.1f}  gap {(st - prev) / 1e3:6.1f}  {name[:44]:44s} {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
    prev = en
    if "copyBuffer" in name and (en - st) > 200e3:
        break
