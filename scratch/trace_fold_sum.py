import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last predict_volume = last third of the launches; group by (kernel short name, grid)
agg = collections.OrderedDict()
n = len(rows)
for r in rows[2 * n // 3:]:
    name = r["Kernel_Name"]
    short = name.split("(")[0][-70:]
    key = (short, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
print("total ms", tot)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{a[1]:9.2f} ms  n={a[0]:4d}  avg {a[1]/a[0]:8.3f}  {k}")
