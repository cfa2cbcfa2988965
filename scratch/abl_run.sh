# per-layer kernel trace of one 64-chunk launch set for the product library and each ablation build given: bash scratch/abl_run.sh TAG abl2 abl3 ...
set -e
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/abl_$1; mkdir -p $O; shift
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/iterseg_amd/libitersegx.so; [ $v = prod ] || L=$GRAFT_REPO_ROOT/scratch/libs/lib_$v.so
  ( cd /tmp && export TMPDIR=/tmp && ISX_LIB=$L rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/p_$v -o t -- python3 $GRAFT_REPO_ROOT/scratch/bench_unet.py 64 > $GRAFT_REPO_ROOT/$O/run_$v.log 2>&1 )
  python3 scratch/trace_layers_seq.py $O/p_$v > $O/seq_$v.txt
  rm -rf $O/p_$v
done
python3 - "$O" "$@" <<'PY'
import sys
O = sys.argv[1]; vs = sys.argv[2:]
cols = [[l.rstrip("\n") for l in open(f"{O}/seq_{v}.txt")] for v in vs]
print(" " * 52 + "".join(f"{v:>9s}" for v in vs))
for i, l in enumerate(cols[0]):
    name = l[:52] if "grid" in l else l[:10].ljust(52)
    vals = []
    for c in cols:
        t = c[i].split()
        vals.append(t[-2] if "ms" in t[-1] else t[-1])
    print(name + "".join(f"{x:>9s}" for x in vals))
PY
