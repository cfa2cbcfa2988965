set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tn; mkdir -p $O
cd $R
for f in 1 0; do
  ISX_H2_NW8=$f rocprofv3 --kernel-trace --output-format csv -d $O/p$f -o t -- python3 scratch/bench_unet.py 64 > $O/run$f.log 2>&1
  python3 scratch/trace_layers_seq.py $O/p$f > $O/seq$f.txt
done
paste -d'|' $O/seq1.txt $O/seq0.txt > $O/cmp.txt
