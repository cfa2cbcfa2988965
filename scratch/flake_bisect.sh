cd $GRAFT_REPO_ROOT
for rev in "$@"; do
  d=$GRAFT_REPO_ROOT/scratch/wt_$rev; [ "$rev" = HEAD ] && d=$GRAFT_REPO_ROOT
  (cd $d && timeout -k 10 200 python scratch/unet_flake.py 200 > $GRAFT_REPO_ROOT/gpurun_out/bis_${rev}_a.log 2>&1 &
   cd $d && timeout -k 10 200 python scratch/unet_flake.py 200 > $GRAFT_REPO_ROOT/gpurun_out/bis_${rev}_b.log 2>&1; wait)
  echo "== $rev: $(grep -h done gpurun_out/bis_${rev}_a.log gpurun_out/bis_${rev}_b.log | tr '\n' ' ') $(grep -h Error gpurun_out/bis_${rev}_a.log | tail -1)"
done
