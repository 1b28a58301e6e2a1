# Runs ON THE GPU BOX: the sharded GPU tests, then 2-rank rehearsals of the bench workloads on ONE GPU -- started the way the
# driver starts them (`python3 bench.py --gpus 2 ...`, no launcher: bench.py spawns its ranks); gloo carries the host side,
# both ranks on device 0 (the driver's real N-GPU runs use RCCL and one GPU per rank).
set -o pipefail
mkdir -p gpurun_out/r2j
timeout -k 10 300 python -m pytest tests/test_gpu_sharded.py -q 2>&1 | tail -5
for W in c2 c4 c5; do
  MPPI_BENCH_DEVICE=0 MPPI_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 $( [ $W = c5 ] && echo "--steps 10 --warmup 2" || echo "--steps 200 --warmup 20" ) --workload $W > gpurun_out/r2j/bench_2rank_$W.json 2> gpurun_out/r2j/bench_2rank_$W.err
  echo "rc $?"; tail -c 400 gpurun_out/r2j/bench_2rank_$W.err
  python3 -c "
import json; d=json.loads(open('gpurun_out/r2j/bench_2rank_$W.json').read().strip().splitlines()[-1]); print('$W', d['value'], d['n_gpus'], d['ms_per_step'], d['config']['exchange'], d['roofline']['kernel_us'])"
done
