"""GPU: the C ABI used from plain C (examples/c_caller.c, compiled here with gcc against include/mppi_hip.h and linked with
libmppi_hip.so -- no Python, no PyTorch in that process) gives what the Python mirror of the reference's class gives: the
same library behind both, so the runs agree to the rounding of the host-side plant."""
import os
import subprocess

import numpy as np
import pytest

from oracle import mppi_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_matches_the_python_mirror(tmp_path):
    import dnn_mppi_mpc_amd as pkg
    from bench import config2_kwargs
    libdir = os.path.join(ROOT, "dnn-mppi-mpc_amd", "lib")
    exe = str(tmp_path / "c_caller")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_caller.c"), "-o", exe, "-L", libdir, "-lmppi_hip",
                           "-Wl,-rpath," + libdir, "-lm"])
    K, T, n_iter = 1024, 50, 6
    out = subprocess.run([exe, str(K), str(T), str(n_iter)], check=True, capture_output=True, text=True, timeout=120).stdout
    lines = [ln.split() for ln in out.strip().splitlines()]
    host = np.array([[float(v) for v in ln] for ln in lines[:n_iter]])
    assert lines[n_iter][0] == "device"
    dev_idx, dev_x = int(lines[n_iter][1]), np.array([float(v) for v in lines[n_iter][2:5]])

    c = pkg.MPPIAlgorithms(**config2_kwargs(K=K, T=T), precision="f32", seed=2024)
    x = np.zeros(3)
    for it in range(n_iter):  # the driver's loop with the host in it
        u0 = c._calc_input_control(x)[0].copy()
        x = mppi_oracle.diffdrive_plant_step(x, u0, 0.1)
        assert int(host[it, 1]) == c.prev_way_point_idx
        np.testing.assert_allclose(host[it, 2:4], u0, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(host[it, 4:7], x, rtol=1e-7, atol=1e-9)
    c.restart_episode(np.zeros(3))  # then the same iterations resident on the device
    c._engine.set_iteration(0)
    _, st = c._engine.run_closed_loop(n_iter)
    assert dev_idx == int(st.idx_after)
    np.testing.assert_allclose(dev_x, c._engine.get_state(), rtol=1e-7, atol=1e-9)
