"""CPU: the plain-C oracle (oracle/mppi_oracle.c) against the reference's own outputs."""
import numpy as np
import pytest

import golden_util as gu
from oracle import c_oracle

DD_SINGLE = [n for n in gu.names("dd_") if n != "dd_closed_loop"]
RC_SINGLE = [n for n in gu.names("rc_") if n != "rc_closed_loop"]


@pytest.mark.parametrize("name", DD_SINGLE)
def test_c_diffdrive_matches_reference(name):
    fx = gu.load(name)
    o = c_oracle.DiffDriveC(**fx["meta"])
    o.u_prev[:] = fx["u_prev_in"]
    o.prev_way_point_idx = int(fx["idx_before"])
    out = o.iteration(fx["x0"], gu.eps_of(fx))
    # libm cos/sin vs NumPy's vector loops differ by an ulp here and there: 1e-11 on S.
    np.testing.assert_allclose(out["S"], fx["S"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(out["u_returned"], fx["u_returned"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(out["u0_returned"], fx["u0_returned"], rtol=1e-7, atol=1e-10)
    assert out["idx_after"] == int(fx["idx_after"])


def test_c_diffdrive_closed_loop():
    fx = gu.load("dd_closed_loop")
    o = c_oracle.DiffDriveC(**fx["meta"])
    for it in range(fx["x0"].shape[0]):
        out = o.iteration(fx["x0"][it], gu.eps_of(fx, it))
        np.testing.assert_allclose(out["S"], fx["S"][it], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(out["u_returned"], fx["u_returned"][it], rtol=1e-6, atol=1e-9)
        assert out["idx_after"] == int(fx["idx_after"][it])


@pytest.mark.parametrize("name", RC_SINGLE)
def test_c_racecar_matches_reference(name):
    fx = gu.load(name)
    o = c_oracle.RaceCarC(ref_path=fx["ref_path"], **fx["meta"])
    o.u_prev[:] = fx["u_prev_in"]
    o.prev_waypoints_idx = int(fx["idx_before"])
    out = o.iteration(fx["x0"], fx["eps"])
    # f32 path, libm cosf/sinf/tanf vs NumPy's f32 loops: a few ulps through T steps
    np.testing.assert_allclose(out["S"], fx["S"], rtol=2e-5)
    np.testing.assert_allclose(out["u_returned"], fx["u_returned"], rtol=2e-3, atol=2e-5)
    assert out["idx_after"] == int(fx["idx_after"])


def test_diffdrive_rejects_short_horizon():
    fx = gu.load("dd_small_T10")
    meta = dict(fx["meta"], num_horizons_T=9)
    o = c_oracle.DiffDriveC(**meta)
    with pytest.raises(ValueError):
        o.iteration(fx["x0"], np.zeros((o.K, 9, 2), np.float32))
