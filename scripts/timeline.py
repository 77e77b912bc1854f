"""Print the GPU timeline (kernels and copies) of the last scans of a rocprofv3 --kernel-trace
--memory-copy-trace --output-format csv run: python scripts/timeline.py <dir> [first_scan_from_end] [rows]"""
import csv
import glob
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:28]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[12:], None))
ev.sort()
scans = [i for i, e in enumerate(ev) if "msd_scan" in e[2]]
i0 = scans[-back]
t0 = ev[i0][0]
for e in ev[i0:i0 + nrows]:
    print("%9.1f %8.1f  %-30s q=%s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
starts = [ev[i][0] for i in scans]
print("scan-to-scan periods (us):", [round((b - a) / 1e3) for a, b in zip(starts[-9:], starts[-8:])])
