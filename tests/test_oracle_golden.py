"""CPU: the NumPy oracle reproduces what the reference itself produced (tests/golden)."""
import numpy as np
import pytest

import golden_util as gu
from oracle import mppi_oracle, philox

DD_SINGLE = [n for n in gu.names("dd_") if n != "dd_closed_loop"]
RC_SINGLE = [n for n in gu.names("rc_") if n != "rc_closed_loop"]


@pytest.mark.parametrize("name", DD_SINGLE)
def test_diffdrive_iteration_matches_reference(name):
    fx = gu.load(name)
    o = gu.make_diffdrive_oracle(fx)
    out = o.iteration(fx["x0"], gu.eps_of(fx).astype(np.float64))
    # f64 path: the oracle restates the same arithmetic, so the match is to rounding.
    np.testing.assert_allclose(out["S"], fx["S"], rtol=1e-13, atol=1e-13)
    assert int(np.argmin(out["S"])) == int(np.argmin(fx["S"]))
    np.testing.assert_allclose(out["w"], fx["w"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(out["w_eps_raw"], fx["w_eps_raw"], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(out["w_eps_filtered"], fx["w_eps_filtered"], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(out["u_returned"], fx["u_returned"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out["u0_returned"], fx["u0_returned"], rtol=1e-9, atol=1e-14)
    assert out["idx_after"] == int(fx["idx_after"])
    if "sampled_traj_list" in fx and fx["meta"]["visualze_sampled_trajs"]:
        opt, smp = o.viz_trajectories(fx["x0"], out["u_pre_shift"], out["v"])
        np.testing.assert_allclose(opt, fx["optimal_traj"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(smp, fx["sampled_traj_list"], rtol=1e-10, atol=1e-12)


def test_diffdrive_closed_loop_matches_reference():
    fx = gu.load("dd_closed_loop")
    o = mppi_oracle.DiffDriveOracle(**fx["meta"])
    state = np.zeros(3)
    for it in range(fx["x0"].shape[0]):
        np.testing.assert_allclose(state, fx["x0"][it], rtol=1e-9, atol=1e-12)
        out = o.iteration(state, gu.eps_of(fx, it).astype(np.float64))
        np.testing.assert_allclose(out["S"], fx["S"][it], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(out["u_returned"], fx["u_returned"][it], rtol=1e-8, atol=1e-12)
        assert out["idx_after"] == int(fx["idx_after"][it])
        state = mppi_oracle.diffdrive_plant_step(state, out["u0_returned"], fx["meta"]["delta_t"])
    np.testing.assert_allclose(state, fx["final_state"], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("name", RC_SINGLE)
def test_racecar_iteration_matches_reference(name):
    fx = gu.load(name)
    o = gu.make_racecar_oracle(fx)
    out = o.iteration(fx["x0"], fx["eps"])
    # f32 path: same dtype and op order as the reference, tolerance is a few f32 ulps of S
    np.testing.assert_allclose(out["S"], fx["S"], rtol=2e-6)
    np.testing.assert_allclose(out["w"], fx["w"], rtol=5e-4, atol=1e-12)
    np.testing.assert_allclose(out["u_returned"], fx["u_returned"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(out["w_eps_filtered"], fx["w_eps_filtered"], rtol=1e-4, atol=2e-6)
    assert out["idx_after"] == int(fx["idx_after"])
    if "sampled_traj_list" in fx and fx["meta"]["visualze_sampled_trajs"]:
        opt, smp = o.viz_trajectories(fx["x0"], out["u_pre_shift"], out["v"])
        np.testing.assert_allclose(opt, fx["optimal_traj"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(smp, fx["sampled_traj_list"], rtol=1e-5, atol=1e-5)


def test_racecar_closed_loop_matches_reference():
    fx = gu.load("rc_closed_loop")
    o = gu.make_racecar_oracle(fx)
    for it in range(fx["x0"].shape[0]):
        out = o.iteration(fx["x0"][it], gu.eps_of(fx, it))
        np.testing.assert_allclose(out["S"], fx["S"][it], rtol=1e-5)
        np.testing.assert_allclose(out["u_returned"], fx["u_returned"][it], rtol=1e-3, atol=1e-5)
        assert out["idx_after"] == int(fx["idx_after"][it])


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = tuple(int(x) for x in philox.philox4x32_10(*ctr, *key))
        assert got == want


def test_philox_sampler_moments():
    sigma = np.array([[0.5, 0.1], [0.1, 0.2]])
    e = philox.sample_epsilon(sigma, 7, 3, 4096, 50).reshape(-1, 2).astype(np.float64)
    assert abs(e.mean(0)).max() < 5e-3
    np.testing.assert_allclose(np.cov(e.T), sigma, atol=6e-3)
    # shard invariance: rows depend on the global sample index only
    part = philox.sample_epsilon(sigma, 7, 3, 100, 50, k_offset=1000)
    np.testing.assert_array_equal(part, philox.sample_epsilon(sigma, 7, 3, 4096, 50)[1000:1100])


@pytest.mark.parametrize("T", [10, 13, 20, 50, 75])
def test_moving_average_filters_match_reference(T):
    """The three `_moving_average_filter` implementations of the reference, called directly by
    oracle/gen_golden.py (tests/golden/filters.npz): NumPy diff-drive, NumPy race car, and the torch files' conv1d
    form (identical in mppi_differential_drive_torch.py and mppi_race_car_torch.py)."""
    fx = gu.load("filters")
    xx = fx[f"in_T{T}"]
    np.testing.assert_allclose(mppi_oracle.moving_average_diffdrive(xx, 10), fx[f"dd_T{T}"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(mppi_oracle.moving_average_racecar(xx.astype(np.float32), 10), fx[f"rc_T{T}"],
                               rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(fx[f"ddtorch_T{T}"], fx[f"rctorch_T{T}"])
    np.testing.assert_allclose(mppi_oracle.moving_average_torch(xx, 10), fx[f"rctorch_T{T}"], rtol=1e-5, atol=1e-7)
    # what the torch form is: the NumPy race-car filter delayed by window // 2 rows
    np.testing.assert_allclose(fx[f"rctorch_T{T}"][5:], fx[f"rc_T{T}"][:T - 5], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", gu.names("ddtorch_"))
def test_diffdrive_torch_variant_matches_reference(name):
    """controllers/mppi_differential_drive_torch.py run on the CPU in f32 (its x0 aliasing removed, see
    oracle/gen_golden.py): beta = lambda, no clamp in the rollout, terminal yaw wrap, conv1d filter.  The f64
    restatement with those four switches agrees to f32 rounding."""
    fx = gu.load(name)

    class TorchVariant(mppi_oracle.DiffDriveOracle):
        CLAMP_ROLLOUT, BETA_IS_LAMBDA, WRAP_YAW_TERMINAL = False, True, True
        FILTER = staticmethod(mppi_oracle.moving_average_torch)

    o = TorchVariant(**fx["meta"])
    o.u_prev[:] = fx["u_prev_in"]
    o.prev_way_point_idx = int(fx["idx_before"])
    out = o.iteration(fx["x0"], fx["eps"].astype(np.float64))
    np.testing.assert_allclose(out["S"], fx["S"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out["w_eps_raw"], fx["w_eps_raw"], rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(out["u_returned"], fx["u_returned"], rtol=1e-4, atol=5e-6)
    assert out["idx_after"] == int(fx["idx_after"])
    plain = gu.make_diffdrive_oracle(fx).iteration(fx["x0"], fx["eps"].astype(np.float64))
    assert not np.allclose(plain["u_returned"], fx["u_returned"], atol=1e-3)  # the switches matter


@pytest.mark.parametrize("name", gu.names("c5_"))
def test_config5_residual_mlp_matches_patched_reference(name):
    """BASELINE config 5 pinned through the reference itself: its MPPIAlgorithms with `_state_transition` replaced by
    x + dt (f + MLP) (the reference's MultiLayerPerceptron with saved_models/mlp_diff_300x100_3l.pth, f64), see
    oracle/gen_golden.py gen_config5.  The restatement = DiffDriveMlpOracle with the same weights."""
    fx = gu.load(name)
    w = gu.mlp_weights(name)  # (`..mlp2l..`: the reference's older two-hidden-layer checkpoint)
    o = mppi_oracle.DiffDriveMlpOracle(**fx["meta"], mlp_weights=w)
    o.u_prev[:] = fx["u_prev_in"]
    o.prev_way_point_idx = int(fx["idx_before"])
    out = o.iteration(fx["x0"], gu.eps_of(fx).astype(np.float64))
    np.testing.assert_allclose(out["S"], fx["S"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(out["u_returned"], fx["u_returned"], rtol=1e-8, atol=1e-12)
    assert out["idx_after"] == int(fx["idx_after"])
    # the f32 forward of the same network (what the engine's matrix-core path computes in) stays within the budget
    assert np.sqrt(np.mean((fx["f32mlp_u_returned"] - fx["u_returned"]) ** 2)) < 1e-5
    assert int(fx["f32mlp_idx_after"]) == int(fx["idx_after"])


def test_racecar_plant_matches_vehicle_update():
    """`Vehicle.update` (models/vehicle.py:85-114) as the race-car driver runs it (mppi_race_car.py:259-281): the
    controller is fed ref_path[i], the vehicle integrates the returned controls; and the closed loop (controller fed
    the vehicle's state)."""
    fx = gu.load("plant_rc_vehicle")
    m = fx["meta"]
    for mode in ("driver", "closed"):
        veh, u0 = fx[mode + "_vehicle"], fx[mode + "_u0"]
        for i in range(u0.shape[0]):
            nxt = mppi_oracle.racecar_plant_step(veh[i], u0[i], m["delta_t"], m["wheel_base"], m["max_steer_abs"],
                                                 m["max_accel_abs"])
            np.testing.assert_allclose(nxt, veh[i + 1], rtol=1e-12, atol=1e-12)
    # the whole closed loop through the restatement
    o = mppi_oracle.RaceCarOracle(ref_path=fx["ref_path"], **m)
    state = fx["closed_vehicle"][0]
    for i in range(fx["closed_u0"].shape[0]):
        np.testing.assert_allclose(state, fx["closed_x0"][i], rtol=1e-4, atol=1e-4)
        out = o.iteration(state, gu.eps_of(fx, i))
        np.testing.assert_allclose(out["u0_returned"], fx["closed_u0"][i], rtol=1e-3, atol=2e-5)
        assert out["idx_after"] == int(fx["closed_idx_after"][i])
        state = mppi_oracle.racecar_plant_step(state, out["u0_returned"], m["delta_t"], m["wheel_base"],
                                               m["max_steer_abs"], m["max_accel_abs"])
