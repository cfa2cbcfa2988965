"""conv launches of the last 64-chunk launch set, in launch order: kernel template, grid, duration."""
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
conv = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv_h2", "conv_first", "conv_tail", "tail_combine", "conv55"))]
# one forward = conv_first + 17 conv_h2 launches (+ more with folds); take the last forward that starts with the biggest grid of conv_first
firsts = [i for i, r in enumerate(conv) if "conv_first" in r["Kernel_Name"]]
big = max(int(conv[i].get("Grid_Size", conv[i].get("Grid_Size_X", 0))) for i in firsts)
start = [i for i in firsts if int(conv[i].get("Grid_Size", conv[i].get("Grid_Size_X", 0))) == big][-1]
end = ([i for i in firsts if i > start] + [len(conv)])[0]
tot = 0.0
for r in conv[start:end]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot += d
    name = r["Kernel_Name"].split("(")[0].replace("void isx::", "")
    print(f"{name:34s} grid {int(r.get('Grid_Size', r.get('Grid_Size_X', 0)))//int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 256))):7d} wg {d:7.3f} ms")
print(f"total {tot:.2f} ms")
