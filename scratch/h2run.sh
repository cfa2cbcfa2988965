# per-layer times of the split-f16 conv variants: kernel traces of one 64-chunk launch set
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for cfg in ${H2_CFGS:-"0 1" "1 1" "1 0"}; do set -- $cfg
  export ISX_H2_BG=$1 ISX_H2_MT2=$2
  rm -rf /tmp/tr_$1_$2
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$1_$2 -o t -- python3 $GRAFT_REPO_ROOT/scratch/one_layer.py 64 > /tmp/tr_$1_$2.log 2>&1 || { tail -5 /tmp/tr_$1_$2.log; exit 1; }
  echo "== BG=$1 MT2=$2"
  python3 $GRAFT_REPO_ROOT/scratch/layer_eff_any.py "/tmp/tr_$1_$2/t_kernel_trace.csv" 64 || find /tmp/tr_$1_$2 -type f
done
