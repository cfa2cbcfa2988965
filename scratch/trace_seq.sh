set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ta_$1; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/p -o t -- python3 scratch/bench_unet.py 64 > $O/run.log 2>&1
python3 scratch/trace_layers_seq.py $O/p > $O/seq.txt
cat $O/seq.txt
