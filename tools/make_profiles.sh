#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/make_profiles.sh r02'): every measurement profiles/ holds for one build.
# rocprofv3 passes are separate runs: --kernel-trace --stats for durations; --pmc passes (their own runs, --kernel-trace
# only) for HBM traffic and instruction counts.  Output: gpurun_out/prof_<round>/ ; tools/collect_profiles.py copies the
# summaries into profiles/ afterwards (in the build container).
set -o pipefail
R=${1:-r02}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# (the profiler passes time the headline path alone: --no-batched leaves the 32-agent block of the default line out)
BENCH="bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-batched --no-graph-timing"
echo "== bench lines"; date
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_steps20.json 2>> $OUT/bench.err || echo "bench20 failed"
echo "== kernel trace of the bench command"; date
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $BENCH > $OUT/kt.log 2>&1 || echo "kt failed"
echo "== PMC: instruction counts"; date
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_inst -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-batched --no-graph-timing > $OUT/pmc_inst.log 2>&1 || echo "pmc_inst failed"
echo "== PMC: HBM traffic"; date
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-batched --no-graph-timing > $OUT/pmc_fetch.log 2>&1 || echo "pmc_fetch failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-batched --no-graph-timing > $OUT/pmc_write.log 2>&1 || echo "pmc_write failed"
echo "== traversal only (HYPK kernels)"; date
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_trav -- python3 tools/traverse_only.py 100 > $OUT/kt_trav.log 2>&1 || echo "kt_trav failed"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_trav -- python3 tools/traverse_only.py 20 > $OUT/pmc_trav.log 2>&1 || echo "pmc_trav failed"
echo "== other configs"; date
timeout -k 10 400 python3 tools/bench_configs.py 3 4s 4 5 > $OUT/configs.jsonl 2> $OUT/configs.err || echo "configs failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg -- python3 tools/bench_configs.py 3 4s > $OUT/kt_cfg.log 2>&1 || echo "kt_cfg failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_cfg4 -- python3 tools/bench_configs.py 4s > $OUT/pmc_cfg4.log 2>&1 || echo "pmc_cfg4 failed"
timeout -k 10 300 python3 bench.py --workload c4 --steps 300 --warmup 30 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err || echo "bench c4 failed"
# keep what travels back small: counter CSVs are large
for d in pmc_inst pmc_fetch pmc_write pmc_trav pmc_cfg4; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_reduce.py "$f" $OUT/$d.reduced.csv && rm -rf $OUT/$d
done
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; ls $OUT
date
