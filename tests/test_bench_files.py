"""bench.py reads its measured HBM traffic (roofline.traffic) from a committed PMC summary: the file it names must exist and
hold an entry per arithmetic mode it reports, or the bench line silently loses the field."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("isx_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_traffic_file_named_by_bench_is_committed_and_complete():
    b = _bench()
    assert os.path.exists(b.TRAFFIC_FILE), b.TRAFFIC_FILE
    d = json.load(open(b.TRAFFIC_FILE))
    for prec in ("split-f16", "fp32"):
        assert prec in d, prec
        e = d[prec]
        assert e["launches"] > 0 and e["chunks_per_launch"] == 64
        assert e["bytes_per_chunk_launch"] > 5.0e7          # more than the algorithmic 57 MB per chunk-launch
        src = e["source"].split(":")[0].replace("{fetch,write}", "fetch")
        assert os.path.exists(os.path.join(ROOT, src)), src
