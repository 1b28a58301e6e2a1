#!/bin/bash
# Runs ON THE GPU BOX, after tools/collect_profiles.py has stamped profiles/<round>_pmc_*.json with THIS build: the bench
# lines again, so that the committed lines carry the counter figures (traffic, valu_issue_frac) of their own build.
set -o pipefail
R=${1:-r02}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_steps20.json 2>> $OUT/bench.err || echo "bench20 failed"
timeout -k 10 300 python3 bench.py --workload c3 --steps 1000 --warmup 100 > $OUT/bench_c3.json 2> $OUT/bench_c3.err || echo "bench c3 failed"
timeout -k 10 300 python3 bench.py --workload c4 --steps 300 --warmup 30 > $OUT/bench_c4.json 2> $OUT/bench_c4.err || echo "bench c4 failed"
timeout -k 10 300 python3 bench.py --workload c5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err || echo "bench c5 failed"
ls -la $OUT/bench*.json
