"""Print a window of a rocprofv3 kernel trace (csv) around the n-th occurrence of a kernel name:
start offset, duration and gap to the previous kernel, in microseconds."""
import csv, sys
path, name, nth, before, after = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]]
c = hits[nth]
t0 = int(rows[max(0, c - before)]["Start_Timestamp"])
prev_end = None
for r in rows[max(0, c - before):c + after]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:70]))
    prev_end = e
