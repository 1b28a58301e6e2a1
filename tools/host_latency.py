"""Where the host-in-the-loop latency goes: the C ABI call with prebuilt arguments, Engine.step, the controller mirror."""
import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
import torch
import dnn_mppi_mpc_amd as pkg
from bench import config2_kwargs
from oracle import mppi_oracle
ctrl = pkg.MPPIAlgorithms(**config2_kwargs(), precision="f32", seed=1)
eng = ctrl._engine
eng.set_state(np.zeros(3)); eng.run_closed_loop(300)
x = eng.get_state()
lib, h = eng.lib, eng._h
u, u0 = np.empty((50, 2)), np.empty(2)
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
xp, up, u0p = dp(x), dp(u), dp(u0)
st = C.byref(eng.stats)
for _ in range(200): lib.mppi_step(h, xp, None, up, u0p, st, None)
ts = []
for _ in range(2000):
    t0 = time.perf_counter(); lib.mppi_step(h, xp, None, up, u0p, st, None); ts.append(time.perf_counter() - t0)
print("C ABI mppi_step via prebuilt ctypes args: median %.2f us, p10 %.2f, p90 %.2f" % tuple(1e6 * np.percentile(ts, q) for q in (50, 10, 90)))
ts = []
for _ in range(2000):
    t0 = time.perf_counter(); eng.step(x); ts.append(time.perf_counter() - t0)
print("Engine.step: median %.2f us" % (1e6 * np.median(ts)))
import contextlib, io
ts = []
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(2000):
        t0 = time.perf_counter(); ctrl._calc_input_control(x); ts.append(time.perf_counter() - t0)
print("MPPIAlgorithms._calc_input_control: median %.2f us" % (1e6 * np.median(ts)))
