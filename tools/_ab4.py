import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch
import dnn_mppi_mpc_amd as pkg
from oracle import mppi_oracle as mo
lem = mo.generate_lemniscate_racecar(100, 10.0)
c = pkg.MPPIRacecarController(ref_path=lem, horizon_step_T=75, number_of_samples_K=8192, obstacle_circles=np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]),
                              visualize_optimal_traj=False, visualze_sampled_trajs=False)
e = c._engine
e.set_state(lem[0].astype(np.float64)); e.run_closed_loop(30); torch.cuda.synchronize()
res = []
for rep in range(6):
    t0 = time.perf_counter(); e.run_closed_loop(300); torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 300 * 1e6)
print(os.environ.get("MPPI_LIB", "new")[-12:], "config 4s us/iter min %.2f med %.2f" % (min(res), sorted(res)[3]))
