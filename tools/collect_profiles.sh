#!/bin/bash
# Copies the summaries of an evidence run (tools/evidence.sh <tag> ..., merged back under gpurun_out/<tag>final/) into profiles/<tag>_*:
# the files the judge reads.  usage: tools/collect_profiles.sh r06
TAG=${1:?tag}; O=gpurun_out/${TAG}final; P=profiles
cpf() { [ -s "$1" ] && cp "$1" "$2" && echo "$2"; }
cpf $O/bench_c3_full.json $P/${TAG}_bench_line.json
cpf $O/bench_line.json $P/${TAG}_bench_line_profiled_cmd.json
cpf $O/bench_kernel_stats.csv $P/${TAG}_bench_kernel_stats.csv
cpf $O/pmc_summary.txt $P/${TAG}_bench_pmc_summary.txt
cpf $O/traffic.json $P/${TAG}_traffic.json
for c in c2 c4_1gpu kmeans ransac; do cpf $O/traffic_$c.json $P/${TAG}_traffic_$c.json; cpf $O/config_${c}_kernel_stats.csv $P/${TAG}_config_${c}_kernel_stats.csv; done
cpf $O/search_pmc_summary.txt $P/${TAG}_search_pmc_summary.txt
cpf $O/bench_c2.json $P/${TAG}_config_c2_line.json
cpf $O/bench_c4_1gpu.json $P/${TAG}_config_c4_1gpu_line.json
cpf $O/bench_kmeans_profiled.json $P/${TAG}_config_kmeans_line.json
cpf $O/bench_ransac_profiled.json $P/${TAG}_config_ransac_line.json
cpf $O/config_c5_pmc_summary.txt $P/${TAG}_config_c5_pmc_summary.txt
cpf $O/config_c5_launches.txt $P/${TAG}_config_c5_launches.txt
cpf $O/independent_source_kernel_stats.csv $P/${TAG}_independent_source_kernel_stats.csv
cpf $O/affine_kernel_stats.csv $P/${TAG}_affine_kernel_stats.csv
cpf $O/directions_kernel_stats.csv $P/${TAG}_directions_kernel_stats.csv
for f in warm_trace_10m warm_trace_1m warm_trace_c4 real_cloud tie_order_build variants directions affine_forms size_sweep; do cpf $O/$f.txt $P/${TAG}_$f.txt; done
# the parity / tie reports the GPU test suite writes into gpurun_out/
for f in parity_10m variants_parity_10m parity_c4 margin_routes warm_matches_10m warm_matches_1m tie_rule tie_rule_forms tie_rule_lattice tie_rule_sharded tie_rule_directions tie_rule_target_shards tie_rule_features tie_count knn_tie_rule real_cloud_forms tie_order_device_build; do
  cpf gpurun_out/$f.json $P/${TAG}_$f.json
done
