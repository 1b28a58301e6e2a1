"""Diagnostic: the device sampler of a build (MPPI_LIB) against the NumPy restatement."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import dnn_mppi_mpc_amd as pkg
from oracle import philox
from bench import config2_kwargs
kw = config2_kwargs()
sigma = np.array([[0.5, 0.1], [0.1, 0.2]])
kw["sigma"] = sigma
c = pkg.MPPIAlgorithms(**kw, seed=0x1234567890ABCDEF)
got = c._engine.sample_epsilon(3).cpu().numpy()
want = philox.sample_epsilon(sigma, 0x1234567890ABCDEF, 3, 4096, 50)
d = np.abs(got - want)
print(os.environ.get("MPPI_LIB", "default")[-14:], "max abs err", d.max(), "mean", d.mean(), "cov err", np.abs(np.cov(got.reshape(-1, 2).T.astype(float)) - sigma).max())
