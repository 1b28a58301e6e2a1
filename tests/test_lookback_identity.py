"""The identity the one-launch resolution of the sequential waypoint index rests on (DESIGN.md section 3.2,
csrc/mppi_kernels.h LB_CAND), checked against the oracle's call-by-call scan (mppi_differential_drive.py:201-249) on the CPU:

    where a call's distances to the candidates behind c fall strictly and then never fall again, its first minimum over
    all of them is m = the number of descents, and the search entered at p returns max(p, m) -- so the threaded index is the
    running maximum of the calls' m, as long as every realised p keeps [p, p+W) inside the candidates and below W.

The kernels evaluate exactly these conditions (pass A's word per call, lb_reach) and hand an iteration that violates one to
the speculation rounds; here the prediction must equal the scan wherever the conditions hold, on straight, curved and
self-crossing paths, and the conditions must hold throughout BASELINE config 2's traversal."""
import numpy as np
import pytest

from oracle import mppi_oracle as mo

LB_CAND, W = 32, 20


def predict(px, py, ref_xy, c):
    """(idx_used_by_call, ok): the running-maximum prediction and whether the kernels' conditions hold for this iteration."""
    nc = min(LB_CAND, ref_xy.shape[0] - c)
    r = ref_xy[c:c + nc]
    d = (px[:, None] - r[None, :, 0]) ** 2 + (py[:, None] - r[None, :, 1]) ** 2
    desc = np.concatenate([np.ones((d.shape[0], 1), bool), d[:, 1:] < d[:, :-1]], axis=1)  # candidate 0 "descends"
    m = desc.sum(1) - 1
    unimodal = np.all(desc[:, :-1] | ~desc[:, 1:], axis=1)  # no descent behind a non-descent
    run = np.maximum.accumulate(m)
    leave = int(run[-1])
    reach = leave < W and (leave + W <= LB_CAND or ref_xy.shape[0] - c <= LB_CAND)
    return c + run, bool(unimodal.all() and reach)


def paths():
    x = np.linspace(0.0, 10.0, 100)
    yield "config 2's line", np.stack([x, -0.5 * x], 1), 0.113
    t = np.linspace(0.0, 2.5, 140)
    yield "a sine", np.stack([4 * t, 1.5 * np.sin(2 * t)], 1), 0.09
    lem = mo.generate_lemniscate_racecar(160, 4.0)[:, :2]
    yield "a lemniscate (crosses itself)", lem, 0.12


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_running_maximum_equals_the_sequential_scan_where_the_conditions_hold(seed):
    rng = np.random.default_rng(seed)
    checked = fell_back = 0
    detail = []
    for name, ref_xy, step in paths():
        for c in (0, 7, ref_xy.shape[0] // 2, ref_xy.shape[0] - 25, ref_xy.shape[0] - 9):
            K, T = 48, 30
            # samples that travel along the path from waypoint c at various speeds, with lateral noise
            speed = rng.uniform(0.0, 0.35, (K, 1)) * np.arange(1, T + 1)[None, :]  # waypoints ahead of c at step t
            j = np.clip(c + speed, 0, ref_xy.shape[0] - 1)
            base = np.stack([np.interp(j, np.arange(ref_xy.shape[0]), ref_xy[:, 0]),
                             np.interp(j, np.arange(ref_xy.shape[0]), ref_xy[:, 1])], -1)
            pos = base + rng.normal(0, 0.3 * step, base.shape)
            px = np.concatenate([pos[:, :, 0], pos[:, -1:, 0]], 1).reshape(-1)  # T stage calls + the terminal call, k-major
            py = np.concatenate([pos[:, :, 1], pos[:, -1:, 1]], 1).reshape(-1)
            truth, p_end = mo.sequential_waypoint_scan(px, py, ref_xy, c, W)
            pred, ok = predict(px, py, ref_xy, c)
            if ok:
                checked += 1
                np.testing.assert_array_equal(pred, truth, err_msg=f"{name}, c={c}")
                assert p_end == pred[-1]
            else:
                fell_back += 1
            detail.append((name, c, ok))
    assert checked >= 8, detail  # (the conditions hold on most of these; the rest is the fallback's business)


def test_conditions_hold_throughout_config_2s_traversal():
    from bench import config2_kwargs
    K, T = 512, 50
    kw = config2_kwargs(K=K, T=T)
    o = mo.DiffDriveOracle(**kw)
    rng = np.random.default_rng(3)
    L = np.linalg.cholesky(kw["sigma"])
    x = np.zeros(3)
    ref_xy = kw["ref_path"][:, :2]
    moved = 0
    for it in range(24):
        eps = rng.standard_normal((K, T, 2)) @ L.T
        out = o.iteration(x, eps)
        c, X = out["idx_start"], out["X"]
        if c >= ref_xy.shape[0] - 1:
            break
        px = np.concatenate([X[:, :, 0], X[:, -1:, 0]], 1).reshape(-1)
        py = np.concatenate([X[:, :, 1], X[:, -1:, 1]], 1).reshape(-1)
        pred, ok = predict(px, py, ref_xy, c)
        assert ok, it
        assert pred[-1] == out["idx_after"], it
        moved += int(pred[-1] != c)
        x = mo.diffdrive_plant_step(x, out["u0_returned"], kw["delta_t"])
    assert moved >= 10  # (the index is carried along the path iteration after iteration)


def test_a_path_that_folds_back_inside_the_candidates_is_detected():
    """A hairpin: the way back runs 0.15 m beside the way out, so a call beside both branches sees its distances fall, rise and
    fall again within the 32 candidates -- `predict` (as pass A in the kernels) must flag the iteration, and wherever it does
    not, the running maximum must still equal the scan."""
    out = np.stack([np.linspace(0.0, 1.2, 13), np.zeros(13)], 1)
    back = np.stack([np.linspace(1.2, 0.0, 13), np.full(13, 0.15)], 1)
    ref_xy = np.concatenate([out, back[1:], np.stack([np.linspace(-0.1, -3.0, 30), np.full(30, 0.15)], 1)])
    rng = np.random.default_rng(5)
    flagged = agreed = 0
    for c in (0, 3, 6, 10, 14, 20):
        K, T = 32, 20
        pos = np.stack([rng.uniform(0.2, 1.1, (K, T)), rng.uniform(-0.05, 0.2, (K, T))], -1)
        px = np.concatenate([pos[:, :, 0], pos[:, -1:, 0]], 1).reshape(-1)
        py = np.concatenate([pos[:, :, 1], pos[:, -1:, 1]], 1).reshape(-1)
        truth, _ = mo.sequential_waypoint_scan(px, py, ref_xy, c, W)
        pred, ok = predict(px, py, ref_xy, c)
        if ok:
            np.testing.assert_array_equal(pred, truth)
            agreed += 1
        else:
            flagged += 1
    assert flagged >= 3, (flagged, agreed)
