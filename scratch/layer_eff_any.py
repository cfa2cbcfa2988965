# per-launch time of the wide-conv kernels (fp32 or split-f16) from a kernel trace of scratch/one_layer.py
import csv, sys, glob
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
f = glob.glob(sys.argv[1])[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r['Kernel_Name'] for k in ('conv_h2', 'conv_mfma', 'conv_fold'))]
lv = [(10,256,256),(10,129,129),(10,65,65),(10,33,33),(5,17,17)]
mods = [(1,32,0),(32,64,1),(64,128,2),(128,256,3),(256,256,4),(512,128,3),(256,64,2),(128,32,1),(64,5,0)]
layers = []
for cin, cout, l in mods:
    m = lv[l][0] * lv[l][1] * lv[l][2]
    layers.append((cin, cout, l, 2.0 * m * cout * cin * 27)); layers.append((cout, cout, l, 2.0 * m * cout * cout * 27))
mf = [x for x in layers if x[0] >= 32 and (x[1] % 32 == 0 or x[1] <= 5)]
rows = rows[-len(mf):]
tot = 0
for r, (cin, cout, l, fl) in zip(rows, mf):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; tot += d
    k = r['Kernel_Name'].split('(')[0].replace('void isx::', '')
    print(f"{cin:4d}->{cout:4d} L{l} {k[:24]:24s} grid {int(r['Grid_Size_X'])//256:7d} {d:9.1f} us {fl*B/(d*1e-6)/1e12:7.1f} TF/s")
print("total", round(tot / 1e3, 2), "ms")
