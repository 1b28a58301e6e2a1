"""GPU (one box): two processes share cuda:0, each owns half of K; the product's sharded paths must equal one
process with all of K -- with the one collective per iteration (step_begin -> all-gather -> step_end; gloo carries
it here, RCCL needs one GPU per rank) and with the peer-to-peer exchange (IPC-mapped buffers, flags polled by the
finalize kernel: the same code that runs over xGMI, here between two processes on one device).  The kernels and the
ABI are the real ones."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _kwargs(K):
    from oracle import mppi_oracle
    lem = mppi_oracle.generate_lemniscate_racecar(100, 10.0)
    return dict(ref_path=lem, horizon_step_T=40, number_of_samples_K=K, param_exploration=0.1, param_alpha=0.9,
                obstacle_circles=np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]), visualize_optimal_traj=False,
                visualze_sampled_trajs=False), lem


def _worker(rank, world, port, q, exchange):
    os.environ["MPPI_EXCHANGE"] = exchange
    os.environ["MPPI_EXCHANGE_TIMEOUT_MS"] = "20000"
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dnn_mppi_mpc_amd as pkg
        torch.cuda.set_device(0)
        kw, lem = _kwargs(3001)  # odd K: shards of 1501 / 1500
        c = pkg.MPPIRacecarController(**kw, precision="f64", seed=77, process_group=dist.group.WORLD)
        us = []
        for it in range(3):  # host-driven sharded steps
            us.append(c._calc_control_input(lem[it].astype(np.float64))[1].copy())
        c._engine.set_state(lem[3].astype(np.float64))  # then the device closed loop, sharded
        c.run_closed_loop_sharded(4)
        q.put((rank, np.stack(us), c.u_prev.copy(), c._engine.get_state(), int(c.last_stats.iteration), c.exchange))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["collective", "p2p"])
def test_two_process_shards_equal_single_process(exchange):
    import dnn_mppi_mpc_amd as pkg
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kw, lem = _kwargs(3001)
    one = pkg.MPPIRacecarController(**kw, precision="f64", seed=77)
    us = np.stack([one._calc_control_input(lem[it].astype(np.float64))[1].copy() for it in range(3)])
    one._engine.set_state(lem[3].astype(np.float64))
    one._engine.run_closed_loop(4)
    u_loop, x_loop = one._engine.get_u_prev(), one._engine.get_state()
    for rank, us_r, u_r, x_r, it_r, used in outs:
        assert used == exchange
        np.testing.assert_allclose(us_r, us, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(u_r, u_loop, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(x_r, x_loop, rtol=1e-9, atol=1e-12)
        assert it_r == 7


def _nccl_worker(port, q, exchange):
    """world_size 1 over the NCCL backend (= RCCL): the sharded controller takes the collective carrier
    (`all_gather_into_tensor` on device tensors, or the library's own ncclAllGather after `mppi_comm_init`) exactly as N
    ranks on N GPUs would, with one rank."""
    os.environ["MPPI_EXCHANGE"] = exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import dnn_mppi_mpc_amd as pkg
        from dnn_mppi_mpc_amd.distributed import exchange_partials
        kw, lem = _kwargs(2048)
        c = pkg.MPPIRacecarController(**kw, precision="f64", seed=5, process_group=dist.group.WORLD)
        assert c._sharded and dist.get_backend(dist.group.WORLD) == "nccl" and c.exchange == exchange
        us = [c._calc_control_input(lem[it].astype(np.float64))[1].copy() for it in range(3)]
        c._engine.set_state(lem[3].astype(np.float64))
        c.run_closed_loop_sharded(5)  # rollout -> rank record -> all-gather (RCCL) -> finalize, per iteration
        part = torch.arange(7, dtype=torch.float64, device="cuda")
        gathered = exchange_partials(part, 1, dist.group.WORLD)
        q.put((np.stack(us), c.u_prev.copy(), c._engine.get_state(), gathered.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["collective", "rccl"])
def test_rccl_collective_path_with_one_rank(exchange):
    """The two RCCL carriers executed on the GPU -- "collective": `backend="nccl"` of distributed.exchange_partials /
    run_closed_loop_sharded; "rccl": the library's own communicator (`mppi_comm_unique_id` / `mppi_comm_init`) and the
    ncclAllGather it enqueues per iteration.  RCCL needs one GPU per rank, so one rank here (two ranks run the split
    step over gloo above): results equal the unsharded controller."""
    import dnn_mppi_mpc_amd as pkg
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q, exchange))
    p.start()
    us_r, u_r, x_r, g = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    np.testing.assert_array_equal(g, np.arange(7.0))
    kw, lem = _kwargs(2048)
    one = pkg.MPPIRacecarController(**kw, precision="f64", seed=5)
    us = np.stack([one._calc_control_input(lem[it].astype(np.float64))[1].copy() for it in range(3)])
    one._engine.set_state(lem[3].astype(np.float64))
    one._engine.run_closed_loop(5)
    np.testing.assert_allclose(us_r, us, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(u_r, one._engine.get_u_prev(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(x_r, one._engine.get_state(), rtol=1e-9, atol=1e-12)
