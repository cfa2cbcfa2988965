# same-box comparison of library builds: accuracy + launch-set time (scratch/acc_check.py) and per-layer trace: bash scratch/lib_ab.sh TAG prod base ...
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/lib_$1; mkdir -p $O; T=$1; shift
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/iterseg_amd/libitersegx.so; [ $v = prod ] || L=$GRAFT_REPO_ROOT/scratch/libs/lib_$v.so
  echo "== $v"; ISX_LIB=$L python3 scratch/acc_check.py 2>&1 | grep -v amdgpu.ids | tee $O/acc_$v.log
done
bash scratch/abl_run.sh ${T}_tr "$@" | grep -v "^\[" | sed 's/^/   /'
