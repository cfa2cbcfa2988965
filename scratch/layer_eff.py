# per-launch efficiency of conv_mfma from a rocprofv3 kernel trace of scratch/one_layer.py (one launch set of B chunks)
import csv, sys
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lv = [(10,256,256),(10,129,129),(10,65,65),(10,33,33),(5,17,17)]
mods = [(1,32,0),(32,64,1),(64,128,2),(128,256,3),(256,256,4),(512,128,3),(256,64,2),(128,32,1),(64,5,0)]
layers = []
for cin,cout,l in mods:
    m = lv[l][0]*lv[l][1]*lv[l][2]
    layers.append((cin,cout,l,2.0*m*cout*cin*27)); layers.append((cout,cout,l,2.0*m*cout*cout*27))
mf = [x for x in layers if x[0] >= 32 and (x[1] % 32 == 0 or x[1] <= 5)]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_mfma' in r['Kernel_Name']]
rows = rows[-len(mf):]
tot_t = tot_f = 0
for r, (cin,cout,l,f) in zip(rows, mf):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tf = f * B / (d * 1e-6) / 1e12
    tot_t += d; tot_f += f * B
    print(f"{cin:4d}->{cout:4d} L{l}  grid {int(r['Grid_Size_X'])//256:7d}  {d:9.1f} us  {tf:6.1f} TF/s  {tf/157.3*100:5.1f}%")
print(f"total {tot_t/1e3:.2f} ms  {tot_f/(tot_t*1e-6)/1e12:.1f} TF/s = {tot_f/(tot_t*1e-6)/1e12/157.3*100:.1f}%")
