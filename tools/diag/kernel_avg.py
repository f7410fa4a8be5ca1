"""Diagnostic: average duration of the kernels whose name contains argv[2] in a rocprofv3 kernel_stats csv (argv[1])."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(sys.argv[1].split("/")[-3], r["Name"][:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
