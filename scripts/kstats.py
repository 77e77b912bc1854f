"""Print kernel name (short), calls, average us from a rocprofv3 kernel_stats csv found under a directory."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
        print("%-42s %5s calls  avg %9.1f us  min %9.1f  max %9.1f" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                     float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
