# Copy the evidence of one scratch/profile_r05.sh run from gpurun_out/<tag>/ into profiles/<tag>_* and rebuild the traffic summary
# bench.py reads.  usage: bash scratch/install_profiles.sh r05f
set -e
T=${1:?tag}; G=gpurun_out/$T; P=profiles
cp $G/${T}_bench.json $G/${T}_bench512_under_rocprof.json $G/${T}_bench_1024_f16.json $G/${T}_bench_1024_split-f16.json $P/
cp $G/prof_bench/r05_kernel_stats.csv $P/${T}_bench512_kernel_stats.csv
for p in split-f16 fp32; do
  cp $G/pmc_${p}_FETCH_SIZE/p_counter_collection.csv $P/${T}_pmc_fetch_size_64chunk_set_$p.csv
  cp $G/pmc_${p}_WRITE_SIZE/p_counter_collection.csv $P/${T}_pmc_write_size_64chunk_set_$p.csv
done
[ -f $G/pmc_sq/p_counter_collection.csv ] && cp $G/pmc_sq/p_counter_collection.csv $P/${T}_pmc_sq_64chunk_set_split-f16.csv
for p in fp32 bf16; do cp $G/prof_train_$p/r05_kernel_stats.csv $P/${T}_train_step_${p}_kernel_stats.csv; done
rm -f $P/r05_pmc_traffic.json
for p in split-f16 fp32; do
  python3 scratch/pmc_traffic_json.py $p $P/${T}_pmc_fetch_size_64chunk_set_$p.csv $P/${T}_pmc_write_size_64chunk_set_$p.csv 64 $P/r05_pmc_traffic.json \
    "profiles/${T}_pmc_{fetch,write}_size_64chunk_set_$p.csv: scratch/profile_r05.sh $T, separate --pmc passes over one 64-chunk launch set (scratch/one_layer.py 64)"
done
ls -la $P/${T}_*
