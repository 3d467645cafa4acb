#!/bin/bash
# Round profile set (run on the GPU box through gpurun).  Outputs under gpurun_out/$1/; the summaries the judge reads are
# then copied to profiles/ (tools/collect_profiles.sh only produces them).  All counter passes run ONE prover alone
# (bench.py --inflight 1) with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
#   stats_inflight1/   rocprofv3 --kernel-trace --stats of the default timed region (one WHOLE-NODE step per proof: VM batch on the
#                      device, trace build, proof)
#   pmc_fetch/, pmc_write/, pmc_valu/   FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes (separate)
#   *.json             plain bench lines for the BASELINE configs and side measurements
set -u
TAG=${1:-r06_final}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
rm -rf $OUT/stats_inflight1  # (an older trace in the same directory would be picked up by the summaries)
cd /tmp && export TMPDIR=/tmp
B1="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --no-whole-node"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_inflight1 -- $B1 > $OUT/bench_under_rocprof.json 2> $OUT/stats1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_valu -- $B1 > /dev/null 2> $OUT/pmc_valu.err
cd $ROOT
python tools/pmc_summary.py $OUT 4 $OUT/pmc_bench.json > $OUT/pmc_bench.txt
python tools/valu_summary.py $OUT 4 $OUT/valu_bench.json profiles/${TAG%%_*}_isa_mix.json > $OUT/valu_bench.txt
python tools/timeline.py $OUT/stats_inflight1 > $OUT/timeline.txt
for k in k_gkr_step k_air_round k_fold_round k_ntt k_vm_ ""; do python tools/launch_hist.py $OUT/stats_inflight1 "$k" 4; done > $OUT/launch_hist.txt
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 int_rates.hip -o int_rates 2>/dev/null && ./int_rates) > $OUT/int_rates.txt 2>&1
# the plain bench lines come AFTER the summaries above have been published (tools/publish_profiles.sh), so that bench.py finds
# counter files for the current sources: tools/bench_lines.sh
(nproc; lscpu | grep "Model name") > $OUT/host.txt
# keep the summaries, drop the raw per-launch CSVs of the counter passes (tens of MB)
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_valu
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
ls $OUT
