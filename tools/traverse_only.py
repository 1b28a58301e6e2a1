"""Diagnostic: only the traversal phase of the reference driver's run at BASELINE config 2 (or, second argument `c3`,
config 3): the first 22 iterations of an episode, the waypoint index moving, episode after episode -- what the kernels
that resolve the sequential index in the launch (k_rollout_fused<..., HYPK>, k_rollout_dual<..., LB>) are profiled on
(rocprofv3 --kernel-trace --stats / --pmc ... -- python3 tools/traverse_only.py [episodes] [c3])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

import dnn_mppi_mpc_amd as pkg  # noqa: E402
from bench import config2_kwargs, config3_kwargs  # noqa: E402

C3 = len(sys.argv) > 2 and sys.argv[2] == "c3"
ctrl = pkg.MPPIAlgorithms(**(config3_kwargs() if C3 else config2_kwargs()), precision="f32", seed=1)
for ep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    ctrl.restart_episode(np.zeros(3))
    ctrl._engine.run_closed_loop(22)
torch.cuda.synchronize()
print(ctrl._engine.counters(), ctrl._engine.stats.idx_after)
