# top kernels of a rocprofv3 --stats csv
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    t = int(r["TotalDurationNs"])
    print("%-72s calls %5s total %8.2f ms %5.1f%%" % (r["Name"][:72], r["Calls"], t / 1e6, 100.0 * t / tot))
