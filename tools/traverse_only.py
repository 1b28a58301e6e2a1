"""Diagnostic: only the traversal phase of the reference driver's run at BASELINE config 2 (the first 22 iterations of
an episode, the waypoint index moving), episode after episode -- what the HYPK kernels are profiled on
(rocprofv3 --kernel-trace --stats / --pmc ... -- python3 tools/traverse_only.py [episodes])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

import dnn_mppi_mpc_amd as pkg  # noqa: E402
from bench import config2_kwargs  # noqa: E402

ctrl = pkg.MPPIAlgorithms(**config2_kwargs(), precision="f32", seed=1)
for ep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    ctrl.restart_episode(np.zeros(3))
    ctrl._engine.run_closed_loop(22)
torch.cuda.synchronize()
print(ctrl._engine.counters(), ctrl._engine.stats.idx_after)
