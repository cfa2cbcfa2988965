# knob A/B in one call: feature maps + launch-set times (knob_ab.py) and per-layer traces: bash scratch/knob_tr.sh TAG KNOB v0 v1 [v2]
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/knob_$1; mkdir -p $O; K=$2; shift 2
python3 scratch/knob_ab.py $K $1 $2 2>&1 | grep -v amdgpu.ids | tee $O/ab.log
for v in "$@"; do
  ( cd /tmp && export TMPDIR=/tmp && env $K=$v rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/p_$v -o t -- python3 $GRAFT_REPO_ROOT/scratch/bench_unet.py 64 > $GRAFT_REPO_ROOT/$O/run_$v.log 2>&1 )
  python3 scratch/trace_layers_seq.py $O/p_$v > $O/seq_$v.txt; rm -rf $O/p_$v
done
python3 - "$O" "$@" <<'PY'
import sys
O = sys.argv[1]; vs = sys.argv[2:]
cols = [[l.rstrip("\n") for l in open(f"{O}/seq_{v}.txt")] for v in vs]
for i in range(max(len(c) for c in cols)):
    print("   ".join((c[i] if i < len(c) else "")[:62].ljust(62) for c in cols))
PY
