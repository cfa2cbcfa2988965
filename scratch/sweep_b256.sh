set -e
for ms in 1 2 3 4; do
    echo "== minsteps $ms"
    ISX_FLOOD_GMINSTEPS=$ms ISX_FLOOD_COOP_MIN=32768 ISX_FLOOD_GROUP_MIN=32768 ISX_FLOOD_STATS=1 timeout -k 10 120 python scratch/bench_ws.py blobs 256 2>&1 | grep -E "rep 1|flood/group" | tail -4
done
