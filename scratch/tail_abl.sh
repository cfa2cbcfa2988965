# timing-only ablations of conv_tail_kernel (dev library scratch/libs/tail_dev.so): kernel time of the tail per launch set
cd $GRAFT_REPO_ROOT
cp iterseg_amd/libitersegx.so /tmp/lib_prod.so
cp scratch/libs/tail_dev.so iterseg_amd/libitersegx.so
for a in 0 1 2 3 4; do echo "#### ISX_TAIL_ABLATE=$a"; ISX_TAIL_ABLATE=$a bash scratch/trace_seq.sh r4abl$a 2>&1 | grep -E "conv_tail|total"; done
cp /tmp/lib_prod.so iterseg_amd/libitersegx.so
