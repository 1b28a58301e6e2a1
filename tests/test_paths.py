"""CPU: the host-side path generators against the reference's own outputs (tests/golden/paths.npz)."""
import numpy as np

import golden_util as gu


def test_path_generators_match_reference():
    from dnn_mppi_mpc_amd import paths
    fx = gu.load("paths")
    rx, ry, ryaw, rk, s = paths.calc_spline_course(fx["wx"], fx["wy"], ds=0.07)
    np.testing.assert_allclose(np.array([rx, ry, ryaw, rk, s]), fx["spline"], rtol=1e-9, atol=1e-10)
    p4, cp4 = paths.calc_4points_bezier_path(1.0, -2.0, 0.3, 8.0, 4.0, -1.2, 3.0)
    np.testing.assert_allclose(p4, fx["bez4"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cp4, fx["bez4_cp"], rtol=1e-14)
    np.testing.assert_allclose(paths.calc_bezier_path(fx["cp"], 64), fx["bez"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(np.array(paths.generate_lemniscate_trajectory(7.5, 80)), fx["dd_lemniscate"], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(np.array(paths.generate_point_trajectory((1.0, 2.0), (-4.0, 9.0), 37)), fx["dd_point"], rtol=1e-14)
    np.testing.assert_allclose(paths.racecar_lemniscate(90, 12.0), fx["rc_lemniscate"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(paths.racecar_circle(70, 8.0), fx["rc_circle"], rtol=1e-6, atol=1e-6)
    assert paths.ref_path_array(rx, ry, ryaw).shape == (len(rx), 3)
    assert paths.ref_path_array(rx, ry, ryaw, 5.0).shape == (len(rx), 4)
