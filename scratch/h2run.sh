# per-layer times of the split-f16 conv variants: kernel traces of one 64-chunk launch set
# H2_ENVS: ';'-separated env assignments per run, e.g. "ISX_H2_WIDE2=0;ISX_H2_WIDE2=1"
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
IFS=';' read -ra RUNS <<< "${H2_ENVS:-ISX_H2_WIDE2=0;ISX_H2_WIDE2=1}"
i=0
for envs in "${RUNS[@]}"; do
  i=$((i+1)); rm -rf /tmp/tr_$i
  env $envs true
  ( export $envs; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$i -o t -- python3 $GRAFT_REPO_ROOT/scratch/one_layer.py 64 > /tmp/tr_$i.log 2>&1 ) || { tail -5 /tmp/tr_$i.log; exit 1; }
  echo "== $envs"
  python3 $GRAFT_REPO_ROOT/scratch/layer_eff_any.py "/tmp/tr_$i/t_kernel_trace.csv" 64 || find /tmp/tr_$i -type f
done
