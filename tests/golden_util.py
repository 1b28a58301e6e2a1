"""Load tests/golden/*.npz (outputs of the reference itself, see oracle/gen_golden.py)."""
import glob
import json
import os

import numpy as np

from oracle import mppi_oracle, philox

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    fx = {k: z[k] for k in z.files}
    meta = json.loads(str(fx.pop("meta")))
    for k, v in list(meta.items()):
        if isinstance(v, list):
            meta[k] = np.array(v)
    fx["meta"] = meta
    return fx


def eps_of(fx, iteration=0):
    """Injected noise: stored for small cases, regenerated from the seed for K=4096."""
    if "eps" in fx and iteration == 0:
        return fx["eps"]
    m = fx["meta"]
    K = m.get("num_samples_K", m.get("number_of_samples_K"))
    T = m.get("num_horizons_T", m.get("horizon_step_T"))
    return philox.sample_epsilon(m["sigma"], int(fx["eps_seed"]), iteration, int(K), int(T))


def make_diffdrive_oracle(fx):
    o = mppi_oracle.DiffDriveOracle(**fx["meta"])
    if "u_prev_in" in fx:
        o.u_prev[:] = fx["u_prev_in"]
        o.prev_way_point_idx = int(fx["idx_before"])
    return o


def make_racecar_oracle(fx, raise_at_path_end=False):
    m = dict(fx["meta"])
    o = mppi_oracle.RaceCarOracle(ref_path=fx["ref_path"], raise_at_path_end=raise_at_path_end, **m)
    if "u_prev_in" in fx:
        o.u_prev[:] = fx["u_prev_in"]
        o.prev_waypoints_idx = int(fx["idx_before"])
    return o


def mlp_weights(fixture=""):
    """saved_models/mlp_diff_300x100_3l.pth as plain arrays (the checkpoint itself stays in the build container); for the
    fixtures named `..mlp2l..` the reference's older two-hidden-layer checkpoint mlp_diff_300x100.pth."""
    name = "mlp_diff_300x100_weights.npz" if "mlp2l" in fixture else "mlp_diff_300x100_3l_weights.npz"
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}
