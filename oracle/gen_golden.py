"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py dd rc paths filters c5 vehicle ddtorch

(every generator; each name selects one, no name = `dd rc paths`; all 28 committed fixtures regenerate bit for bit)

It imports ``/root/reference/controllers/mppi_*.py`` unmodified, replaces the instance's
``_calc_epsilon`` with an injected noise tensor (recipe: SURVEY.md section 8c), spies on
``_compute_weight`` / ``_moving_average_filter`` to capture S, w and the pre/post-filter
weighted noise, and stores inputs + the reference's outputs as ``.npz`` (plain arrays,
``allow_pickle=False``).  Nothing of the reference's source is copied; fixtures are data.

Library versions used for the committed fixtures are stored inside each file
(reference pins numpy 1.26.4; these were taken under the version printed below, which
matters for the race-car f32 path -- SURVEY.md H5).
"""
from __future__ import annotations

import json
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

from oracle import philox  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _spy(obj, name, log):
    orig = getattr(obj, name)

    def wrapped(*a, **kw):
        r = orig(*a, **kw)
        args = list(a) + list(kw.values())
        log.append((tuple(np.array(x, copy=True) if isinstance(x, np.ndarray) else x for x in args),
                    np.array(r, copy=True)))
        return r

    setattr(obj, name, wrapped)


def run_reference_iteration(ctrl, x0, eps, method):
    """One `_calc_*` call on a reference controller with eps injected; returns captures."""
    wlog, flog = [], []
    ctrl._calc_epsilon = lambda *a, **kw: np.array(eps, copy=True)
    if not hasattr(ctrl, "_spied"):
        ctrl._wlog, ctrl._flog = wlog, flog
        _spy(ctrl, "_compute_weight", ctrl._wlog)
        _spy(ctrl, "_moving_average_filter", ctrl._flog)
        ctrl._spied = True
    else:
        ctrl._wlog.clear()
        ctrl._flog.clear()
    idx_attr = "prev_way_point_idx" if hasattr(ctrl, "prev_way_point_idx") else "prev_waypoints_idx"
    cap = {"idx_before": int(getattr(ctrl, idx_attr)), "u_prev_in": ctrl.u_prev.copy()}
    u0, u, opt, smp = getattr(ctrl, method)(np.array(x0, copy=True))
    cap["S"] = ctrl._wlog[0][0][0]
    cap["w"] = ctrl._wlog[0][1]
    cap["w_eps_raw"] = ctrl._flog[0][0][0]
    cap["w_eps_filtered"] = ctrl._flog[0][1]
    cap["u_returned"] = np.array(u, copy=True)
    cap["u0_returned"] = np.array(u0, copy=True)
    cap["optimal_traj"] = np.array(opt, copy=True)
    cap["sampled_traj_list"] = np.array(smp, copy=True)
    cap["idx_after"] = int(getattr(ctrl, idx_attr))
    return cap


def versions():
    import scipy
    return json.dumps({"numpy": np.__version__, "scipy": scipy.__version__,
                       "python": sys.version.split()[0], "reference_tag": "2024_08_07"})


def save(name, kwargs, arrays):
    os.makedirs(OUT, exist_ok=True)
    meta = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in kwargs.items()}
    arrays = {k: np.asarray(v) for k, v in arrays.items() if v is not None}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=np.array(json.dumps(meta)),
                        versions=np.array(versions()), **arrays)
    print(f"wrote {name}.npz  ({sum(a.nbytes for a in arrays.values())/1e3:.1f} kB raw)")


# ---------------------------------------------------------------------------------------
# differential drive
# ---------------------------------------------------------------------------------------

def dd_main_kwargs(**over):
    """Parameters of the reference's `__main__` (mppi_differential_drive.py:400-410)."""
    from controllers.mppi_differential_drive import generate_point_trajectory
    cx, cy, cyaw = generate_point_trajectory(np.array([0.0, 0.0]), np.array([10.0, -5.0]))
    kw = dict(delta_t=0.1, ref_path=np.array([cx, cy, cyaw]).T, max_speed=5.0, max_omega=3.14,
              num_samples_K=128, num_horizons_T=25, param_exploration=0.0001, param_lambda=1.0,
              param_alpha=0.2, sigma=np.array([[0.1, 0.0], [0.0, 0.01]]),
              stage_cost_weight=np.array([5.0, 5.0, 10.0]), terminal_cost_weight=np.array([5.0, 5.0, 10.0]),
              visualize_optimal_traj=True, visualze_sampled_trajs=True)
    kw.update(over)
    return kw


def dd_obs_kwargs(n_circles=2, **over):
    """`_obs` `__main__` parameters (mppi_differential_drive_obs.py:436-452); the 8-circle
    set is BASELINE config 3's."""
    from controllers.mppi_differential_drive_obs import generate_point_trajectory
    cx, cy, cyaw = generate_point_trajectory(np.array([0.0, 0.0]), np.array([5.0, 5.0]))
    circles = [[2.0, 2.0, 0.4], [3.0, 3.5, 0.4]]
    rng = np.random.default_rng(1234)
    while len(circles) < n_circles:
        x, y = rng.uniform(0.5, 4.5, 2)
        if x * x + y * y > (0.4 + 0.5) ** 2:
            circles.append([float(x), float(y), 0.4])
    kw = dict(delta_t=0.1, ref_path=np.array([cx, cy, cyaw]).T, max_speed=5.0, max_omega=3.14,
              num_samples_K=128, num_horizons_T=25, param_exploration=0.05, param_lambda=10.0,
              param_alpha=0.98, sigma=np.array([[0.1, 0.0], [0.0, 0.01]]),
              stage_cost_weight=10 * np.array([5.0, 6.0, 9.0]), terminal_cost_weight=10 * np.array([5.0, 6.0, 9.0]),
              obstacle_circles=np.array(circles[:n_circles]), safety_margin_rate=0.8,
              visualize_optimal_traj=True, visualze_sampled_trajs=True)
    kw.update(over)
    return kw


def gen_diffdrive():
    from controllers.mppi_differential_drive import MPPIAlgorithms as DD
    from controllers.mppi_differential_drive_obs import MPPIAlgorithms as DDObs

    def one(name, kw, x0, seed, cls=DD, u_prev=None, idx0=0, store_eps=True, store_traj=True):
        c = cls(**kw)
        if u_prev is not None:
            c.u_prev[:] = u_prev
        c.prev_way_point_idx = idx0
        K, T = kw["num_samples_K"], kw["num_horizons_T"]
        eps = philox.sample_epsilon(kw["sigma"], seed, 0, K, T)
        cap = run_reference_iteration(c, np.array(x0, float), eps.astype(np.float64), "_calc_input_control")
        arrays = dict(x0=np.array(x0, float), eps_seed=np.array(seed), **cap)
        if store_eps:
            arrays["eps"] = eps
        if not store_traj:
            arrays.pop("sampled_traj_list")
        save(name, kw, arrays)

    one("dd_c1_default", dd_main_kwargs(), [0, 0, 0], 11)
    one("dd_c1_moderate", dd_main_kwargs(param_exploration=0.1), [0, 0, 0], 12)
    T = 25
    tt = np.arange(T)
    u_nz = np.stack([1.5 + 0.5 * np.sin(0.3 * tt), 0.2 * np.cos(0.2 * tt)], axis=1)
    one("dd_nonzero_u", dd_main_kwargs(param_exploration=0.02), [1.0, -0.4, -0.3], 13, u_prev=u_nz, idx0=5)
    one("dd_clamped", dd_main_kwargs(max_speed=0.25, max_omega=0.08, param_exploration=0.05), [0.2, 0.1, 0.1], 14,
        u_prev=u_nz * 0.2)
    one("dd_path_end", dd_main_kwargs(param_exploration=0.05), [9.95, -4.9, -0.4], 15, idx0=90)
    one("dd_viz_off", dd_main_kwargs(param_exploration=0.05, visualize_optimal_traj=False,
                                     visualze_sampled_trajs=False), [0.5, -0.2, 0.0], 16, u_prev=u_nz * 3.0)
    one("dd_small_T10", dd_main_kwargs(num_samples_K=100, num_horizons_T=10), [0, 0, 0], 17)
    # BASELINE config 2's problem on the reference (7 s per iteration): eps by seed only.
    big = dd_main_kwargs(num_samples_K=4096, num_horizons_T=50, visualize_optimal_traj=False,
                         visualze_sampled_trajs=False)
    one("dd_c2_k4096_default", big, [0, 0, 0], 21, store_eps=False, store_traj=False)
    big2 = dict(big, param_exploration=0.1)
    tt = np.arange(50)
    u50 = np.stack([2.0 + 0.5 * np.sin(0.2 * tt), -0.1 + 0.1 * np.cos(0.1 * tt)], axis=1)
    one("dd_c2_k4096_moderate", big2, [2.0, -1.1, -0.45], 22, u_prev=u50, idx0=12, store_eps=False,
        store_traj=False)

    # obstacles
    one("dd_obs_m2", dd_obs_kwargs(2), [0, 0, 0.6], 31, cls=DDObs)
    one("dd_obs_m8_collide", dd_obs_kwargs(8), [1.2, 1.3, 0.78], 32, cls=DDObs,
        u_prev=np.tile([2.0, 0.0], (25, 1)), idx0=20)
    one("dd_obs_m8_k1024", dd_obs_kwargs(8, num_samples_K=1024, num_horizons_T=50, visualze_sampled_trajs=False),
        [0.3, 0.2, 0.7], 33, cls=DDObs, store_traj=False)

    # closed loop, 12 iterations with the reference's own plant (:33-40, :305-367)
    from controllers.mppi_differential_drive import DifferentialDrive
    kw = dd_main_kwargs(param_exploration=0.05, num_samples_K=256, num_horizons_T=20,
                        visualze_sampled_trajs=False, visualize_optimal_traj=False)
    c = DD(**kw)
    plant = DifferentialDrive(np.array([0.0, 0.0, 0.0]))
    rec = {k: [] for k in ("x0", "u0_returned", "u_returned", "idx_after", "S_min", "S")}
    for it in range(12):
        state = plant.get_state()
        eps = philox.sample_epsilon(kw["sigma"], 41, it, 256, 20)
        cap = run_reference_iteration(c, state, eps.astype(np.float64), "_calc_input_control")
        rec["x0"].append(state)
        rec["u0_returned"].append(cap["u0_returned"])
        rec["u_returned"].append(cap["u_returned"])
        rec["idx_after"].append(cap["idx_after"])
        rec["S_min"].append(cap["S"].min())
        rec["S"].append(cap["S"])
        plant.update_state(kw["delta_t"], state, cap["u0_returned"])
    rec["final_state"] = plant.get_state()
    save("dd_closed_loop", kw, dict(eps_seed=np.array(41), **{k: np.array(v) for k, v in rec.items()}))


# ---------------------------------------------------------------------------------------
# race car
# ---------------------------------------------------------------------------------------

def gen_racecar():
    from controllers.mppi_race_car import MPPIRacecarController as RC
    from controllers.mppi_race_car_obstacle import MPPIRacecarController as RCObs

    base = dict(delta_t=0.05, wheel_base=2.5, max_steer_abs=0.523, max_accel_abs=2.0,
                horizon_step_T=25, number_of_samples_K=128, param_exploration=0.01, param_lambda=50.0,
                param_alpha=1.0, sigma=np.array([[0.5, 0.0], [0.0, 0.1]]),
                stage_cost_weight=np.array([50.0, 50.0, 1.0, 20.0]),
                terminal_cost_weight=np.array([50.0, 50.0, 1.0, 20.0]),
                visualize_optimal_traj=True, visualze_sampled_trajs=True)

    def one(name, cls, kw, path, x0, seed, u_prev=None, idx0=0, store_traj=True):
        c = cls(**dict(kw, ref_path=path))
        if u_prev is not None:
            c.u_prev[:] = u_prev
        c.prev_waypoints_idx = idx0
        K, T = kw["number_of_samples_K"], kw["horizon_step_T"]
        eps = philox.sample_epsilon(kw["sigma"], seed, 0, K, T)
        cap = run_reference_iteration(c, np.array(x0, np.float32), eps, "_calc_control_input")
        arrays = dict(x0=np.array(x0, np.float32), eps=eps, eps_seed=np.array(seed), ref_path=c.ref_path, **cap)
        if not store_traj:
            arrays.pop("sampled_traj_list")
        save(name, kw, arrays)

    helper = RCObs()
    lem = helper.generate_lemniscate_trajectory(100, 10.0)       # mppi_race_car_obstacle.py:288-299
    circ = RC().generate_simple_trajectory(100, 10.0)            # mppi_race_car.py:224-234
    one("rc_circle", RC, base, circ, circ[0], 51)
    one("rc_circle_gamma", RC, dict(base, param_alpha=0.9, param_lambda=20.0), circ, circ[7], 52,
        u_prev=np.tile([0.1, 0.5], (25, 1)).astype(np.float32), idx0=5)
    obs = dict(base, obstacle_circles=np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]), collision_safety_margin_rat=1.5)
    one("rc_obs_default", RCObs, obs, lem, lem[0], 53)
    one("rc_obs_none_collided", RCObs, dict(obs, obstacle_circles=np.array([[50.0, 50.0, 1.0]])), lem, lem[3], 54,
        idx0=2)
    one("rc_obs_all_collided", RCObs, dict(obs, obstacle_circles=np.array([[10.0, 0.0, 6.0], [8.0, 1.0, 3.0]])),
        lem, lem[0], 55)
    one("rc_obs_T75", RCObs, dict(obs, horizon_step_T=75, number_of_samples_K=96, visualze_sampled_trajs=False),
        lem, lem[10], 56, idx0=8, store_traj=False)

    # closed loop over the reference driver (mppi_race_car_obstacle.py:336-341: state = ref_path[i])
    kw = dict(obs, number_of_samples_K=192, horizon_step_T=20, visualize_optimal_traj=False,
              visualze_sampled_trajs=False)
    c = RCObs(**dict(kw, ref_path=lem))
    rec = {k: [] for k in ("x0", "u0_returned", "u_returned", "idx_after", "S")}
    for it in range(8):
        eps = philox.sample_epsilon(kw["sigma"], 61, it, 192, 20)
        cap = run_reference_iteration(c, lem[it], eps, "_calc_control_input")
        rec["x0"].append(lem[it])
        rec["u0_returned"].append(cap["u0_returned"])
        rec["u_returned"].append(cap["u_returned"])
        rec["idx_after"].append(cap["idx_after"])
        rec["S"].append(cap["S"])
    save("rc_closed_loop", kw, dict(eps_seed=np.array(61), ref_path=lem, **{k: np.array(v) for k, v in rec.items()}))


if __name__ == "__main__":
    print("numpy", np.__version__)
    which = sys.argv[1:] or ["dd", "rc"]
    if "dd" in which:
        gen_diffdrive()
    if "rc" in which:
        gen_racecar()


def gen_paths():
    """Outputs of the reference's path generators (path_generator/*.py, the controllers' helpers)."""
    from path_generator.cubic_spline_planner import calc_spline_course
    from path_generator.bezierPath import calc_4points_bezier_path, calc_bezier_path
    from controllers.mppi_differential_drive import generate_lemniscate_trajectory, generate_point_trajectory
    from controllers.mppi_race_car import MPPIRacecarController as RC
    from controllers.mppi_race_car_obstacle import MPPIRacecarController as RCObs
    wx = np.array([0.0, 2.5, 5.0, 7.5, 3.0, -1.0])
    wy = np.array([0.0, -3.0, 0.5, 4.0, 6.0, 2.0])
    rx, ry, ryaw, rk, s = calc_spline_course(wx, wy, ds=0.07)
    path4, cp4 = calc_4points_bezier_path(1.0, -2.0, 0.3, 8.0, 4.0, -1.2, 3.0)
    cp = np.array([[0.0, 0.0], [2.0, 5.0], [6.0, -4.0], [9.0, 1.0], [12.0, 3.0]])
    lem = generate_lemniscate_trajectory(7.5, 80)
    pt = generate_point_trajectory(np.array([1.0, 2.0]), np.array([-4.0, 9.0]), 37)
    save("paths", {}, dict(wx=wx, wy=wy, spline=np.array([rx, ry, ryaw, rk, s]), bez4=path4, bez4_cp=cp4, cp=cp,
                           bez=calc_bezier_path(cp, 64), dd_lemniscate=np.array(lem), dd_point=np.array(pt),
                           rc_lemniscate=RCObs().generate_lemniscate_trajectory(90, 12.0),
                           rc_circle=RC().generate_simple_trajectory(70, 8.0)))


if __name__ == "__main__" and "paths" in (sys.argv[1:] or ["paths"]):
    gen_paths()


def gen_filters():
    """The three `_moving_average_filter` implementations of the reference, called directly on seeded inputs:
    NumPy diff-drive (mppi_differential_drive.py:257-271), NumPy race car (mppi_race_car.py:211-222) and the
    torch one (mppi_differential_drive_torch.py:252-263 == mppi_race_car_torch.py:211-222, on the CPU)."""
    import torch
    from controllers.mppi_differential_drive import MPPIAlgorithms as DD
    from controllers.mppi_race_car import MPPIRacecarController as RC
    from controllers.mppi_differential_drive_torch import MPPIAlgorithms as DDT
    from controllers.mppi_race_car_torch import MPPIRacecarController as RCT
    rng = np.random.default_rng(20240807)
    arrays = {}
    for T in (10, 13, 20, 50, 75):
        xx = rng.normal(size=(T, 2)) * np.array([0.3, 0.05])
        arrays[f"in_T{T}"] = xx
        arrays[f"dd_T{T}"] = DD._moving_average_filter(None, xx=xx.copy(), window_size=10)
        arrays[f"rc_T{T}"] = RC._moving_average_filter(None, xx.astype(np.float32), window_size=10)
        xt = torch.from_numpy(xx.astype(np.float32))
        arrays[f"ddtorch_T{T}"] = DDT._moving_average_filter(None, xt.clone(), 10).numpy()
        arrays[f"rctorch_T{T}"] = RCT._moving_average_filter(None, xt.clone(), 10).numpy()
    save("filters", {"window_size": 10, "torch": torch.__version__}, arrays)


if __name__ == "__main__" and "filters" in sys.argv[1:]:
    gen_filters()


# ---------------------------------------------------------------------------------------
# BASELINE config 5: the reference's own MPPI class with `_state_transition` swapped for the residual model
# (SURVEY.md section 8c recipe): x' = x + dt (f(x, v) + MLP([x, v])), f = the unicycle of
# mppi_differential_drive.py:182-198, MLP = train/train_diff_mlp.py:13-36 with saved_models/mlp_diff_300x100_3l.pth
# (weights_only=True: nothing from the file is executed).  The cost / waypoint / weight / filter / shift code that
# runs is the reference's.  The weights travel as plain arrays (mlp_diff_300x100_3l_weights.npz): the checkpoint
# itself does not exist on the GPU box.
# ---------------------------------------------------------------------------------------

def gen_config5(TWO_LAYER=False):
    """TWO_LAYER: the reference's older checkpoints (saved_models/mlp_diff_300x100.pth: hidden_layer.{0,1} only) through the
    same class -- its forward walks `self.hidden_layer` (train/train_diff_mlp.py:33-34), so the module list is cut to the two
    layers the checkpoint holds; nothing else changes."""
    import types

    import torch
    from controllers.mppi_differential_drive import MPPIAlgorithms as DD
    from train.train_diff_mlp import MultiLayerPerceptron

    ckpt = "mlp_diff_300x100" if TWO_LAYER else "mlp_diff_300x100_3l"
    sd = torch.load(f"/root/reference/saved_models/{ckpt}.pth", map_location="cpu", weights_only=True)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, f"{ckpt}_weights.npz"),
                        **{k: v.numpy() for k, v in sd.items()})

    def make_net():
        n = MultiLayerPerceptron(5)
        if TWO_LAYER:
            n.hidden_layer = torch.nn.ModuleList(list(n.hidden_layer)[:2])
        n.load_state_dict(sd)
        return n
    net = make_net().double().eval()  # f64 like the NumPy controller around it (the f32 forward is recorded beside it)
    net32 = make_net().eval()

    def patched(model):
        def _state_transition(self, x_t, v_t):  # same signature as mppi_differential_drive.py:182
            x, y, yaw = x_t
            speed, omega = v_t
            z = np.array([x, y, yaw, speed, omega])
            with torch.no_grad():
                r = model(torch.as_tensor(z, dtype=next(model.parameters()).dtype)[None])[0].double().numpy()
            dt = self.delta_t
            return np.array([x + dt * (speed * np.cos(yaw) + r[0]), y + dt * (speed * np.sin(yaw) + r[1]),
                             yaw + dt * (omega + r[2])])
        return _state_transition

    def one(name, kw, x0, seed, u_prev, idx0):
        K, T = kw["num_samples_K"], kw["num_horizons_T"]
        eps = philox.sample_epsilon(kw["sigma"], seed, 0, K, T)
        caps = {}
        for tag, model in (("", net), ("f32mlp_", net32)):
            c = DD(**kw)
            c._state_transition = types.MethodType(patched(model), c)
            c.u_prev[:] = u_prev
            c.prev_way_point_idx = idx0
            cap = run_reference_iteration(c, np.array(x0, float), eps.astype(np.float64), "_calc_input_control")
            caps.update({tag + k: v for k, v in cap.items() if tag == "" or k in ("S", "u_returned", "idx_after")})
        caps.pop("sampled_traj_list")
        caps.pop("optimal_traj")
        save(name, kw, dict(x0=np.array(x0, float), eps_seed=np.array(seed), eps=eps if K <= 256 else None, **caps))

    if TWO_LAYER:
        tt = np.arange(25)
        one("c5_mlp2l_k128", dd_main_kwargs(param_exploration=0.05, visualize_optimal_traj=False, visualze_sampled_trajs=False),
            [0.4, -0.1, -0.35], 73, np.stack([1.2 + 0.3 * np.sin(0.2 * tt), 0.05 * np.cos(0.1 * tt)], axis=1), 2)
        return
    tt = np.arange(25)
    one("c5_mlp_k128", dd_main_kwargs(param_exploration=0.05, visualize_optimal_traj=False, visualze_sampled_trajs=False),
        [0.4, -0.1, -0.35], 71, np.stack([1.2 + 0.3 * np.sin(0.2 * tt), 0.05 * np.cos(0.1 * tt)], axis=1), 2)
    tt = np.arange(50)
    one("c5_mlp_k1024", dd_main_kwargs(num_samples_K=1024, num_horizons_T=50, param_exploration=0.1,
                                       visualize_optimal_traj=False, visualze_sampled_trajs=False),
        [1.0, -0.6, -0.45], 72, np.stack([1.5 + 0.5 * np.sin(0.2 * tt), -0.05 + 0.1 * np.cos(0.1 * tt)], axis=1), 6)


if __name__ == "__main__" and "c5" in sys.argv[1:]:
    gen_config5()
if __name__ == "__main__" and "c5two" in sys.argv[1:]:
    gen_config5(TWO_LAYER=True)


# ---------------------------------------------------------------------------------------
# The race-car driver's plant: Vehicle.update (models/vehicle.py:85-114) in the loop of
# controllers/mppi_race_car.py:259-281 -- (a) exactly as that `__main__` runs it (the controller is fed
# ref_path[i], the vehicle integrates the returned controls beside it), (b) closed: the controller is fed the
# vehicle's state (what `mppi_run_closed_loop` does on the device).
# ---------------------------------------------------------------------------------------

def gen_vehicle():
    from controllers.mppi_race_car import MPPIRacecarController as RC
    from models.vehicle import Vehicle
    kw = dict(delta_t=0.05, wheel_base=2.5, max_steer_abs=0.523, max_accel_abs=2.0, horizon_step_T=20,
              number_of_samples_K=192, param_exploration=0.01, param_lambda=50.0, param_alpha=1.0,
              sigma=np.array([[0.5, 0.0], [0.0, 0.1]]), stage_cost_weight=np.array([50.0, 50.0, 1.0, 20.0]),
              terminal_cost_weight=np.array([50.0, 50.0, 1.0, 20.0]), visualize_optimal_traj=False,
              visualze_sampled_trajs=False)
    # :224-234, cast as the driver does (:267).  An open arc of that circle: on the closed one the 200-candidate search
    # of the x0 call jumps to the last waypoint (= the first one) as soon as the car has left the start, and the
    # controller raises (:63-65) in its second closed-loop iteration
    path = RC().generate_simple_trajectory(100, 10.0).astype(np.float32)[:80]
    n_it, seed = 12, 81
    rec = {}
    for mode in ("driver", "closed"):
        c = RC(**dict(kw, ref_path=path))
        veh = Vehicle(ref_path=path[:, :2], visualize=False)
        veh.reset(init_state=np.array(path[0], dtype=np.float64))
        xs, us, vs, idx = [], [], [veh.get_state()], []
        for i in range(n_it):
            state = path[i] if mode == "driver" else veh.get_state()
            eps = philox.sample_epsilon(kw["sigma"], seed, i, 192, 20)
            cap = run_reference_iteration(c, state, eps, "_calc_control_input")
            veh.update(u=cap["u0_returned"], delta_t=c.delta_t, append_frame=False)
            xs.append(np.array(state, np.float64))
            us.append(cap["u0_returned"])
            idx.append(cap["idx_after"])
            vs.append(veh.get_state())
        rec.update({mode + "_x0": np.array(xs), mode + "_u0": np.array(us), mode + "_vehicle": np.array(vs),
                    mode + "_idx_after": np.array(idx), mode + "_u_final": c.u_prev.copy()})
    save("plant_rc_vehicle", kw, dict(eps_seed=np.array(seed), ref_path=path, **rec))


if __name__ == "__main__" and "vehicle" in sys.argv[1:]:
    gen_vehicle()


# ---------------------------------------------------------------------------------------
# controllers/mppi_differential_drive_torch.py run on the CPU (f32 torch): beta = lambda (:187-190), no clamp in
# the rollout (:128), terminal yaw wrap (:231), conv1d filter (:252-263).  Its `_state_transition` (:196-209) adds
# into VIEWS of its argument, so the first step of every sample advances the shared `x0` -- a defect, not a
# behaviour to keep: the fixture is taken with that one aliasing removed (the argument is
# cloned before the reference's own function body runs); every line of arithmetic is the reference's.
# ---------------------------------------------------------------------------------------

def gen_dd_torch():
    import torch
    from controllers.mppi_differential_drive_torch import MPPIAlgorithms as DDT

    orig = DDT._state_transition

    def no_alias(self, x_t, v_t):
        return orig(self, x_t.clone(), v_t)

    def one(name, x0, seed, u_prev, idx0, **over):
        cx = np.linspace(0.0, 10.0, 100)
        cy = np.linspace(0.0, -5.0, 100)
        cyaw = np.arctan2(-5.0, 10.0) * np.ones(100)
        kw = dict(delta_t=0.1, ref_path=np.array([cx, cy, cyaw]).T, max_speed=5.0, max_omega=3.14, num_samples_K=96,
                  num_horizons_T=20, param_exploration=0.05, param_lambda=1.0, param_alpha=0.2,
                  sigma=np.array([[0.1, 0.0], [0.0, 0.01]]), stage_cost_weight=np.array([5.0, 5.0, 10.0]),
                  terminal_cost_weight=np.array([5.0, 5.0, 10.0]), visualize_optimal_traj=False,
                  visualize_sampled_traj=False)
        kw.update(over)
        f = lambda v: torch.tensor(v, dtype=torch.float32)
        c = DDT(delta_t=f(kw["delta_t"]), ref_path=f(kw["ref_path"]), max_speed=f(kw["max_speed"]),
                max_omega=f(kw["max_omega"]), num_samples_K=kw["num_samples_K"], num_horizons_T=kw["num_horizons_T"],
                param_exploration=f(kw["param_exploration"]), param_lambda=f(kw["param_lambda"]),
                param_alpha=f(kw["param_alpha"]), sigma=f(kw["sigma"]), stage_cost_weight=f(kw["stage_cost_weight"]),
                terminal_cost_weight=f(kw["terminal_cost_weight"]), visualize_optimal_traj=kw["visualize_optimal_traj"],
                visualize_sampled_traj=kw["visualize_sampled_traj"])
        c._state_transition = no_alias.__get__(c)
        K, T = kw["num_samples_K"], kw["num_horizons_T"]
        eps = philox.sample_epsilon(kw["sigma"], seed, 0, K, T)
        c._calc_epsilon = lambda *a, **k: torch.tensor(eps)
        c.u_prev[:] = f(u_prev)
        c.prev_way_point_idx = idx0
        log = {}
        cw, mf = c._compute_weight, c._moving_average_filter
        c._compute_weight = lambda S: (log.__setitem__("S", S.clone().numpy()), cw(S))[1]
        c._moving_average_filter = lambda xx, window_size: (log.__setitem__("w_eps_raw", xx.clone().numpy()),
                                                            mf(xx=xx, window_size=window_size))[1]
        x0t = f(x0)
        u0, u, _, _ = c._calc_input_control(x0t)
        assert np.allclose(x0t.numpy(), np.asarray(x0, np.float32)), "x0 aliasing was not removed"
        save(name, kw, dict(x0=np.array(x0, float), eps=eps, eps_seed=np.array(seed), u_prev_in=np.array(u_prev, float),
                            idx_before=np.array(idx0), S=log["S"], w_eps_raw=log["w_eps_raw"],
                            u_returned=c.u_prev.clone().numpy(), u0_returned=c.u_prev[0].clone().numpy(),
                            idx_after=np.array(int(c.prev_way_point_idx))))

    tt = np.arange(20)
    one("ddtorch_default", [0.0, 0.0, 0.0], 91, np.zeros((20, 2)), 0)
    one("ddtorch_unclamped_wrap", [1.0, -0.4, -0.9], 92,
        np.stack([1.5 + 0.5 * np.sin(0.3 * tt), 0.2 * np.cos(0.2 * tt)], axis=1), 5, max_speed=0.6, max_omega=0.1,
        param_lambda=2.0, param_exploration=0.1, visualize_sampled_traj=True)


if __name__ == "__main__" and "ddtorch" in sys.argv[1:]:
    gen_dd_torch()
