cd $GRAFT_REPO_ROOT
for g in 262144 65536 262144 65536; do echo "== fresh process, ISX_FLOOD_GROUP_MIN=$g"; ISX_FLOOD_GROUP_MIN=$g python scratch/time_structured.py "" 0 2>&1 | grep seed; done
