"""CPU, world_size 2 over gloo: the N>1 path's host logic -- shard ownership, the one all-gather of an
iteration (the product's `exchange_partials`) and the rescale merge -- reproduces the unsharded result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu
from oracle import mppi_oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dnn_mppi_mpc_amd.distributed import exchange_partials, shard_range
        fx = gu.load(name)
        o = gu.make_racecar_oracle(fx)
        ref = o.iteration(fx["x0"], fx["eps"])  # unsharded reference (every rank recomputes it)
        K = fx["eps"].shape[0]
        k0, n = shard_range(K, rank, world)
        beta = 1.0 / fx["meta"]["param_lambda"]
        rec = mppi_oracle.softmin_partial(ref["S"][k0:k0 + n], fx["eps"][k0:k0 + n], beta)
        gathered = exchange_partials(torch.from_numpy(rec), world, dist.group.WORLD)
        rho, eta, ess, w_eps = mppi_oracle.merge_partials(gathered.view(world, -1).numpy(), beta)
        w64 = ref["w"].astype(np.float64)
        q.put((rank, k0, n, float(abs(w_eps - ref["w_eps_raw"]).max()), float(rho - ref["S"].min()),
               float(ess - w64.sum() ** 2 / (w64 ** 2).sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["rc_circle_gamma", "rc_obs_default"])
def test_two_rank_shards_merge_to_the_unsharded_update(name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, k0a, na, err_a, drho_a, dess_a), (r1, k0b, nb, err_b, drho_b, dess_b) = out
    assert (k0a, k0b) == (0, na) and na + nb == gu.load(name)["eps"].shape[0]
    for err, drho, dess in ((err_a, drho_a, dess_a), (err_b, drho_b, dess_b)):
        assert err < 5e-6      # the reference sums in f32; the merge is f64
        assert abs(drho) < 1e-3
        assert abs(dess) < 1e-2


def test_shard_range_covers_all_samples():
    from dnn_mppi_mpc_amd.distributed import shard_range
    for K in (1, 7, 128, 4096, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_range(K, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (a, n), (b, _) in zip(spans, spans[1:]):
                assert a + n == b
            assert spans[-1][0] + spans[-1][1] == K
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1
