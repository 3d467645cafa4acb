#!/bin/bash
# Copy the summaries of gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/ under the round prefix the bench reads.
#   usage: tools/publish_profiles.sh r02_mid r02
set -e
SRC=gpurun_out/$1
P=profiles/$2
cp $SRC/pmc_bench.json ${P}_pmc_bench.json
cp $SRC/valu_bench.json ${P}_valu_bench.json
cp $SRC/valu_bench.txt ${P}_valu_bench.txt
cp $SRC/pmc_bench.txt ${P}_pmc_bench.txt
cp $SRC/timeline.txt ${P}_timeline_single_proof.txt
cp $SRC/launch_hist.txt ${P}_launch_hist.txt
cp $SRC/int_rates.txt ${P}_int_rates.txt
[ -s $SRC/inflight_sweep.txt ] && cp $SRC/inflight_sweep.txt ${P}_inflight_sweep.txt
cp $SRC/host.txt ${P}_host.txt
cp "$(ls -t $SRC/stats_inflight1/*/*_kernel_stats.csv | head -1)" ${P}_kernel_stats_single_proof.csv
# (the plain bench lines are produced by tools/bench_lines.sh AFTER this script and copied from its output directory)
for f in bench_under_rocprof; do
  [ -s $SRC/$f.json ] && cp $SRC/$f.json ${P}_$f.json
done
[ -s $SRC/bench_recursion_shape_kernels.txt ] && cp $SRC/bench_recursion_shape_kernels.txt ${P}_bench_recursion_shape_kernels.txt
ls ${P}_*
