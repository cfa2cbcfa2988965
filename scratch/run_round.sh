set -e
timeout -k 10 900 python -m pytest tests/test_gpu_watershed.py -x -q -m gpu 2>&1 | tail -3
ISX_FLOOD_COOP_MIN=32768 ISX_FLOOD_GROUP_MIN=32768 timeout -k 10 120 python scratch/bench_ws.py blobs 256 2>&1 | grep "rep 1"
ISX_FLOOD_COOP_MIN=32768 ISX_FLOOD_GROUP_MIN=0 timeout -k 10 120 python scratch/bench_ws.py blobs 256 2>&1 | grep "rep 1"
timeout -k 10 400 python scratch/time_structured.py "" 0,3,1,2 2>&1 | grep seed
timeout -k 10 200 python scratch/bench_c2.py 2>&1 | grep "rep 1"
