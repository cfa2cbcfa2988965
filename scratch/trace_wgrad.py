import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void isx::', '').replace('isx::', '')
    key = (n, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(key, []).append(d)
tot = collections.Counter()
for k, v in agg.items():
    tot[k[0]] += sum(v)
    if 'wgrad_mfma' in k[0] or (len(sys.argv) > 2 and sys.argv[2] in k[0]):
        print(f"{k[0][:28]:28s} grid {int(k[1])//256:5d} x {k[2]:>3s} x {k[3]:>3s} n={len(v):3d} avg {sum(v)/len(v):9.1f} us")
steps = 7
for n, t in tot.most_common(14): print(f"{n[:40]:40s} {t/1e3/steps:8.2f} ms/step")
