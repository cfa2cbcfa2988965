# summarise a rocprofv3 kernel trace per (kernel, grid) -> avg duration; usage: trace_layers.py <kernel_trace.csv>
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'conv_mfma' not in n: continue
    key = (n.split('(')[0].replace('void isx::', ''), r['Grid_Size_X'], r['Grid_Size_Y'], r['LDS_Block_Size'] if 'LDS_Block_Size' in r else '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(key, []).append(d)
tot = 0
for k, v in agg.items():
    print(f"{k[0]:32s} grid {int(k[1])//256:6d} x {k[2]:>3s} lds {k[3]:>6s}  n={len(v):4d} avg {sum(v)/len(v):9.1f} us  total {sum(v)/1e3:8.2f} ms")
    tot += sum(v)
print("total", tot / 1e3, "ms")
