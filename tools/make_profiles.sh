#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/make_profiles.sh r03'): every measurement profiles/ holds for one build.
#   1. rocprofv3 --kernel-trace --stats of every bench command (durations, launch counts)
#   2. rocprofv3 --pmc passes, each its own run with --kernel-trace only (instruction counts; FETCH_SIZE; WRITE_SIZE; the
#      matrix-pipe counters of the learned-dynamics kernel), reduced to one row per (kernel, launch size, counter)
#   3. tools/collect_profiles.py <round> --on-box: the reduced rows -> gpurun_out/prof_<round>/pmc.json, copied to
#      profiles/<round>_pmc.json ON THE BOX so that
#   4. the bench lines taken last quote the counter figures of their own build (same sources = same build id).
# Output: gpurun_out/prof_<round>/ ; `python tools/collect_profiles.py <round>` (build container) copies it into profiles/.
set -o pipefail
R=${1:-r03}
OUT="gpurun_out/prof_${R}"
mkdir -p "$OUT"
find "$OUT" -mindepth 1 -delete
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
Q="--no-cpu-baseline --no-batched --no-graph-timing"
declare -A CMD
CMD[c2]="bench.py --steps 2000 --warmup 200 $Q"
CMD[c3]="bench.py --workload c3 --steps 600 --warmup 60 $Q"
CMD[c4]="bench.py --workload c4 --steps 200 --warmup 20 $Q"
CMD[c5]="bench.py --workload c5 --steps 12 --warmup 2 $Q"
CMD[eps_c2]="bench.py --eps hbm --workload c2 --steps 200 --warmup 20"
CMD[eps_c4]="bench.py --eps hbm --workload c4 --steps 100 --warmup 10"
CMD[trav]="tools/traverse_only.py 60"
CMD[trav_c3]="tools/traverse_only.py 60 c3"
reduce() {  # <pass name>: the pass's counter CSV -> <pass name>.reduced.csv; the raw output (large) is dropped
  local d="$OUT/$1"
  local f
  f=$(find "$d" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_reduce.py "$f" "$d.reduced.csv"
  find "$d" -mindepth 1 -delete; rmdir "$d"
}
echo "== kernel traces"; date
for W in c2 c3 c4 c5 eps_c2 eps_c4 trav trav_c3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$W" -- python3 ${CMD[$W]} > "$OUT/kt_$W.log" 2>&1 || echo "kt $W failed"
  f=$(find "$OUT/kt_$W" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$W.csv"
  find "$OUT/kt_$W" -mindepth 1 -delete; rmdir "$OUT/kt_$W"
done
echo "== PMC: instruction counts"; date
for W in c2 c3 c4 eps_c2 eps_c4 trav trav_c3; do
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pmc_inst_$W" -- python3 ${CMD[$W]} > "$OUT/pmc_inst_$W.log" 2>&1 || echo "pmc_inst $W failed"
  reduce "pmc_inst_$W"
done
echo "== PMC: HBM traffic"; date
for W in c2 c3 c4 eps_c2 eps_c4; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${C}_$W" -- python3 ${CMD[$W]} > "$OUT/pmc_${C}_$W.log" 2>&1 || echo "pmc $C $W failed"
    reduce "pmc_${C}_$W"
  done
done
echo "== PMC: matrix pipe of the learned-dynamics kernel"; date
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/pmc_mfma_c5" -- python3 ${CMD[c5]} > "$OUT/pmc_mfma_c5.log" 2>&1 || echo "pmc_mfma failed"
reduce pmc_mfma_c5
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$OUT/pmc_mfma2_c5" -- python3 ${CMD[c5]} > "$OUT/pmc_mfma2_c5.log" 2>&1 || echo "pmc_mfma2 failed"
reduce pmc_mfma2_c5
echo "== counter summaries -> profiles/${R}_pmc.json (on the box: the bench lines below quote them)"; date
python3 tools/collect_profiles.py "$R" --on-box || echo "collect failed"
echo "== bench lines"; date
timeout -k 10 300 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || echo "bench failed"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_steps20.json" 2>> "$OUT/bench.err" || echo "bench20 failed"
timeout -k 10 300 python3 bench.py --workload c3 --steps 1000 --warmup 100 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err" || echo "bench c3 failed"
timeout -k 10 300 python3 bench.py --workload c4 --steps 300 --warmup 30 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err" || echo "bench c4 failed"
timeout -k 10 300 python3 bench.py --workload c5 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" || echo "bench c5 failed"
MPPI_MLP_TERMS=2 timeout -k 10 300 python3 bench.py --workload c5 --no-cpu-baseline > "$OUT/bench_c5_two_terms.json" 2> "$OUT/bench_c5_two_terms.err" || echo "bench c5 (two terms) failed"
timeout -k 10 300 python3 bench.py --eps hbm --workload c2 > "$OUT/bench_eps_c2.json" 2> "$OUT/bench_eps_c2.err" || echo "bench eps c2 failed"
timeout -k 10 300 python3 bench.py --eps hbm --workload c4 > "$OUT/bench_eps_c4.json" 2> "$OUT/bench_eps_c4.err" || echo "bench eps c4 failed"
echo "== self-launched 2-rank rehearsals (one GPU, gloo host side)"; date
for W in c2 c4; do
  MPPI_BENCH_DEVICE=0 MPPI_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --workload $W --steps 200 --warmup 20 > "$OUT/bench_2rank_$W.json" 2> "$OUT/bench_2rank_$W.err" || echo "2-rank $W failed"
done
echo "== parity margins"; date
timeout -k 10 300 python3 -m pytest tests -m gpu -q -s -k "config5_checkpoint or two_term" 2>&1 | grep PARITY_MARGIN > "$OUT/parity_margins.txt"
du -sh "$OUT"; ls "$OUT"
date
