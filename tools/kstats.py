"""Print average kernel durations (ms) from a rocprofv3 kernel_stats.csv: python tools/kstats.py FILE [MIN_MS]"""
import csv
import re
import sys

min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "wm::" not in n:
        continue
    avg = float(r["AverageNs"]) / 1e6
    if avg < min_ms:
        continue
    short = re.sub(r"\(.*", "", n).replace("void wm::", "").replace("wm::", "")
    print("%-40s calls %3s avg %9.3f ms" % (short[:40], r["Calls"], avg))
