"""FETCH_SIZE / WRITE_SIZE PMC passes -> profiles/r02_pmc_traffic.json (read by bench.py as roofline.traffic).

usage: pmc_traffic_json.py <precision> <fetch_csv> <write_csv> <chunks_in_launch_set> <out_json> <source note>
FETCH_SIZE and WRITE_SIZE are in KB; FETCH_SIZE counts 64 B per 128-B request on gfx950, so it is doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section).  Only the wide-conv kernel of the precision is summed."""
import csv, json, os, sys
prec, fcsv, wcsv, chunks, out, note = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6]
kern = ("conv_mfma_kernel",) if prec == "fp32" else ("conv_h2", "conv_tail")       # every kernel of the ISX_PROF_CONV_MFMA class
def total(path, counter):
    t, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if any(k in r["Kernel_Name"] for k in kern) and r["Counter_Name"] == counter:
            t += float(r["Counter_Value"]); n += 1
    return t, n
f, nf = total(fcsv, "FETCH_SIZE"); w, nw = total(wcsv, "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
bytes_total = f * 1024 * 2 + w * 1024
d = json.load(open(out)) if os.path.exists(out) else {}
d[prec] = {"bytes_per_chunk_launch": bytes_total / nf / chunks, "launches": nf, "chunks_per_launch": chunks,
           "fetch_bytes_x2": f * 1024 * 2, "write_bytes": w * 1024, "kernel": "/".join(kern), "source": note}
json.dump(d, open(out, "w"), indent=1)
print(prec, "launches", nf, "per launch GB", bytes_total / nf / 1e9, "per chunk-launch MB", bytes_total / nf / chunks / 1e6)
