set -u
cd $GRAFT_REPO_ROOT
bash tools/bench_lines.sh r05_final
python tools/stress_inflight.py 8 25 0 1 > gpurun_out/r05_final/stress_inflight.txt 2>&1
tail -3 gpurun_out/r05_final/stress_inflight.txt
ls gpurun_out/r05_final
