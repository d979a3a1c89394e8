"""Timeline of the last distributed step in a rocprofv3 kernel trace: python tools/dist_step_timeline.py <trace.csv>.
A step starts at the kernel named by the second argument (default: the row-info kernel, the first of the multiply that follows the exchange)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_b_rowinfo"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
lo, hi = starts[-2], starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:60]
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {name}")
    busy += e - s
    prev_end = max(prev_end, e)
t1 = int(rows[hi]["Start_Timestamp"])
print(f"step {(t1 - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {hi - lo} launches")
