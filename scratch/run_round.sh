set -e
timeout -k 10 900 python -m pytest tests/test_gpu_watershed.py -x -q -m gpu -k "plateau" 2>&1 | tail -3
ISX_FLOOD_PLATEAU=2 ISX_FLOOD_COOP_MIN=32768 ISX_FLOOD_GROUP_MIN=32768 ISX_FLOOD_STATS=1 timeout -k 10 120 python scratch/bench_ws.py blobs 256 2>&1 | grep -E "rep 1|group\] \(both" | tail -2
