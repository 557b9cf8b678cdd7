#!/usr/bin/env python3
"""Print the ldpc:: rows of a rocprofv3 kernel_stats.csv (name, calls, mean / min / max in us)."""
import csv, glob, sys
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*kernel_stats.csv", recursive=True) or [path]:
        for r in csv.DictReader(open(f)):
            if "ldpc::" in r["Name"]:
                name = r["Name"].replace("void ", "").split("(")[0]
                print(f"{name:48s} calls {int(r['Calls']):5d}  mean {float(r['AverageNs'])/1e3:9.1f} us  min {int(r['MinNs'])/1e3:9.1f}  max {int(r['MaxNs'])/1e3:9.1f}")
