"""Print the per-kernel average durations of a rocprofv3 --kernel-trace --stats csv directory."""
import csv, glob, sys
d = sys.argv[1]
for fn in glob.glob(d + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(fn)))[:12]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(4), f"{float(r['AverageNs'])/1e3:10.1f} us")
