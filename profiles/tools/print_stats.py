"""Print a rocprofv3 *kernel_stats.csv as a compact table: python print_stats.py <dir-or-file>."""
import csv
import glob
import sys

f = sys.argv[1]
if not f.endswith(".csv"):
    f = glob.glob(f + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
