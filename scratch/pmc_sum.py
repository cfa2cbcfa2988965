# aggregate rocprofv3 --pmc counter_collection.csv per kernel: sum of each counter over dispatches
import csv, sys, collections, glob
for f in sys.argv[1:]:
    for path in glob.glob(f):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name'].split('(')[0].replace('void isx::', '')
            if 'conv_mfma' not in k and 'conv_h2' not in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, d in agg.items():
            print(k, {c: f"{v:.4g}" for c, v in d.items()})
