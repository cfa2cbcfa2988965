set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/planar_prof; mkdir -p $O
cd $R
for v in 0 1; do
  ISX_PLANAR=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$v -o r -- python3 scratch/one_layer.py 64 > $O/run$v.log 2>&1
done
