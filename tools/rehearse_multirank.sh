# Runs ON THE GPU BOX: the sharded GPU tests, then a 2-rank rehearsal of both bench workloads on ONE GPU (gloo carries the
# host side, both ranks on device 0; the driver runs the real N-GPU scaling bench over RCCL).
set -o pipefail
mkdir -p gpurun_out/r2j
timeout -k 10 300 python -m pytest tests/test_gpu_sharded.py -q 2>&1 | tail -5
# N=2 rehearsal of both bench workloads on ONE GPU: gloo carries the host side, both ranks on device 0
for W in c2 c4 c5; do
MPPI_BENCH_DEVICE=0 MPPI_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 $( [ $W = c5 ] && echo "--steps 10 --warmup 2" || echo "--steps 200 --warmup 20" ) --workload $W > gpurun_out/r2j/bench_2rank_$W.json 2> gpurun_out/r2j/bench_2rank_$W.err; echo "rc $?"; tail -c 400 gpurun_out/r2j/bench_2rank_$W.err; python -c "
import json; d=json.loads(open('gpurun_out/r2j/bench_2rank_$W.json').read().strip().splitlines()[-1]); print('$W', d['value'], d['n_gpus'], d['ms_per_step'], d['config']['exchange'], d['roofline']['kernel_us'], d['roofline']['kernel_us_method'][:40])"
done
