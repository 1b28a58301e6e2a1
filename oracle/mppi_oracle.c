/*
 * Plain-C restatement of the reference's MPPI iteration.  TEST INFRASTRUCTURE ONLY.
 *
 * Scalar, single-threaded, loops in the reference's own order (k-major, t inner).  Used by
 * tests/ as the full-size checker (K=4096..65536) and by bench.py as the `cpu_baseline`
 * ("kind": "port") timed on the GPU box's host cores.  The product (libmppi_hip.so)
 * never links or calls this file.
 *
 * Parity status: PINNED -- checked against tests/golden/*.npz (outputs of the reference
 * itself, oracle/gen_golden.py) in tests/test_oracle_c.py.
 *
 * file:line citations are relative to /root/reference.
 *   diff-drive : controllers/mppi_differential_drive.py:87-289, ..._obs.py:93-313  (f64)
 *   race car   : controllers/mppi_race_car.py:55-222, ..._obstacle.py:65-274       (f32)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int K, T, n_ref, n_obs;
    int clamp_u_after_update; /* visualze_sampled_trajs (diff) / visualize_optimal_traj (race) */
    int reserved;
    double delta_t, u_max0, u_max1, wheel_base;
    double param_exploration, param_lambda, param_alpha;
    double sigma[4];
    double stage_w[4], term_w[4];
    double safety_margin; /* diff: safety_margin_rate, race: collision_safety_margin_rate */
} oracle_cfg;

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---------------------------------------------------------------- differential drive -- */

/* `_get_nearest_waypoint` mppi_differential_drive.py:201-220, SEARCH_IDX_LEN = 20 (:204) */
static int dd_nearest(const double *ref, int n_ref, int p, double x, double y) {
    int best = 0;
    double bd = 0.0;
    for (int j = 0; j < 20 && p + j < n_ref; ++j) {
        double dx = x - ref[3 * (p + j)], dy = y - ref[3 * (p + j) + 1];
        double d = dx * dx + dy * dy;
        if (j == 0 || d < bd) { bd = d; best = j; }
    }
    return p + best;
}

/* `_is_collided` mppi_differential_drive_obs.py:301-313 */
static double dd_collided(const oracle_cfg *c, const double *obs, double x, double y) {
    double rr = 0.5 * c->safety_margin;
    for (int m = 0; m < c->n_obs; ++m) {
        double dx = x - obs[3 * m], dy = y - obs[3 * m + 1], r = rr + obs[3 * m + 2];
        if (dx * dx + dy * dy < r * r) return 1.0;
    }
    return 0.0;
}

static double dd_state_cost(const oracle_cfg *c, const double *w, const double *ref, const double *obs, int i,
                            double x, double y, double yaw) {
    double ex = x - ref[3 * i], ey = y - ref[3 * i + 1], eyaw = yaw - ref[3 * i + 2];
    double cost = w[0] * (ex * ex) + w[1] * (ey * ey) + w[2] * (eyaw * eyaw);
    if (c->n_obs > 0) cost += dd_collided(c, obs, x, y) * 1.0e10;
    return cost;
}

/* `_moving_average_filter` mppi_differential_drive.py:257-271, window 10 */
static void dd_moving_average(const double *xx, double *out, int T) {
    for (int d = 0; d < 2; ++d) {
        for (int i = 0; i < T; ++i) { /* np.convolve(..., 'same'): taps i-5 .. i+4 */
            double s = 0.0;
            for (int m = i + 4; m >= i - 5; --m) /* numpy's sum runs over the kernel index */
                if (m >= 0 && m < T) s += xx[2 * m + d] * 0.1;
            out[2 * i + d] = s;
        }
        out[d] *= 10.0 / 5.0;
        for (int i = 1; i < 5; ++i) {
            out[2 * i + d] *= 10.0 / (i + 5);
            out[2 * (T - 1) + d] *= 10.0 / (i + 5);
        }
    }
}

/*
 * One `_calc_input_control` (mppi_differential_drive.py:87-165) with eps injected.
 * In/out: u_prev[T,2], idx.  Out: S[K], u0[2], stats[4] = {rho, eta, idx_start, path_end}.
 */
int oracle_diffdrive_iteration(const oracle_cfg *c, const double *ref, const double *obs, const double *x0,
                               const float *eps, double *u_prev, int *idx, double *S, double *u0_out,
                               double *stats) {
    const int K = c->K, T = c->T;
    if (T < 10) return -1; /* the reference's filter raises for T < window */
    double *u = u_prev; /* alias, :90 */
    double *w_eps = (double *)calloc((size_t)4 * T, sizeof(double)), *filt = w_eps + 2 * T;
    double *wgt = (double *)malloc(sizeof(double) * K);
    int p = dd_nearest(ref, c->n_ref, *idx, x0[0], x0[1]); /* :96 */
    int path_end = p >= c->n_ref - 1;
    if (path_end) p = c->n_ref - 1; /* :97-99 */
    const int p_start = p;
    const double gamma = c->param_lambda * (1.0 - c->param_alpha); /* :74 */
    const double det = c->sigma[0] * c->sigma[3] - c->sigma[1] * c->sigma[2];
    const double si[4] = {c->sigma[3] / det, -c->sigma[1] / det, -c->sigma[2] / det, c->sigma[0] / det};
    const double thr = (1.0 - c->param_exploration) * K; /* :116 */
    const double dt = c->delta_t;

    for (int k = 0; k < K; ++k) {
        double x = x0[0], y = x0[1], yaw = x0[2], s = 0.0;
        for (int t = 0; t < T; ++t) {
            double e0 = eps[((size_t)k * T + t) * 2], e1 = eps[((size_t)k * T + t) * 2 + 1];
            double v0 = (double)k < thr ? u[2 * t] + e0 : e0;
            double v1 = (double)k < thr ? u[2 * t + 1] + e1 : e1;
            v0 = clampd(v0, -c->u_max0, c->u_max0); /* `_g` :285-289 */
            v1 = clampd(v1, -c->u_max1, c->u_max1);
            double nx = x + v0 * cos(yaw) * dt, ny = y + v0 * sin(yaw) * dt; /* :194-196 */
            yaw = yaw + v1 * dt;
            x = nx;
            y = ny;
            p = dd_nearest(ref, c->n_ref, p, x, y); /* `_compute_cost` :228 updates the index */
            double q0 = u[2 * t] * si[0] + u[2 * t + 1] * si[2], q1 = u[2 * t] * si[1] + u[2 * t + 1] * si[3];
            s = dd_state_cost(c, c->stage_w, ref, obs, p, x, y, yaw) + gamma * (q0 * v0 + q1 * v1); /* '=' :124 */
        }
        p = dd_nearest(ref, c->n_ref, p, x, y); /* `_terminal_cost` :244 */
        S[k] = s + dd_state_cost(c, c->term_w, ref, obs, p, x, y, yaw);
    }
    *idx = p;

    double rho = S[0], eta = 0.0; /* `_compute_weight` :167-180 */
    for (int k = 1; k < K; ++k) rho = S[k] < rho ? S[k] : rho;
    for (int k = 0; k < K; ++k) eta += exp(-(1.0 / c->param_exploration) * (S[k] - rho));
    for (int k = 0; k < K; ++k) wgt[k] = (1 / eta) * exp(-(1.0 / c->param_exploration) * (S[k] - rho));
    for (int t = 0; t < T; ++t) /* :132-135 */
        for (int k = 0; k < K; ++k) {
            w_eps[2 * t] += wgt[k] * eps[((size_t)k * T + t) * 2];
            w_eps[2 * t + 1] += wgt[k] * eps[((size_t)k * T + t) * 2 + 1];
        }
    dd_moving_average(w_eps, filt, T);              /* :138 */
    for (int i = 0; i < 2 * T; ++i) u[i] += filt[i]; /* :141 */
    if (c->clamp_u_after_update)                    /* :145-149 */
        for (int t = 0; t < T; ++t) {
            u[2 * t] = clampd(u[2 * t], -c->u_max0, c->u_max0);
            u[2 * t + 1] = clampd(u[2 * t + 1], -c->u_max1, c->u_max1);
        }
    memmove(u, u + 2, sizeof(double) * 2 * (T - 1)); /* :162-163, last row repeats */
    u0_out[0] = u[0];                                /* :165 alias: pre-shift u[1] */
    u0_out[1] = u[1];
    if (stats) { stats[0] = rho; stats[1] = eta; stats[2] = p_start; stats[3] = path_end; }
    free(w_eps);
    free(wgt);
    return 0;
}

/*
 * The same iteration with a waypoint index that does not travel from sample to sample, so that samples are independent
 * and the K loop runs on all host cores (OpenMP) -- the all-core CPU figure of bench.py's `cpu_baseline` (SURVEY.md
 * section 8d-ii); the reference's own sequential index (above) cannot be parallelised over samples.
 *   per_rollout = 0  FROZEN: every call searches the window at the x0 call's index (the engine's MPPI_WAYPOINT_FROZEN,
 *                    the race-car files' semantics, mppi_race_car.py:143,152)
 *   per_rollout = 1  PER ROLLOUT: the index threads through a sample's own T stage calls and its terminal call exactly
 *                    as the reference threads it (:228, :244: every call searches from where the previous one ended) but
 *                    starts again from the x0 call's index at every sample (MPPI_WAYPOINT_PER_ROLLOUT)
 * Checked against the NumPy restatement in tests/test_oracle_c.py.
 */
int oracle_diffdrive_iteration_independent(const oracle_cfg *c, const double *ref, const double *obs, const double *x0,
                                           const float *eps, double *u_prev, int *idx, double *S, double *u0_out,
                                           double *stats, int n_threads, int per_rollout) {
    const int K = c->K, T = c->T;
    if (T < 10) return -1;
    double *u = u_prev;
    double *w_eps = (double *)calloc((size_t)4 * T, sizeof(double)), *filt = w_eps + 2 * T;
    double *wgt = (double *)malloc(sizeof(double) * K);
    int p0 = dd_nearest(ref, c->n_ref, *idx, x0[0], x0[1]);
    const int path_end = p0 >= c->n_ref - 1;
    if (path_end) p0 = c->n_ref - 1;
    const double gamma = c->param_lambda * (1.0 - c->param_alpha);
    const double det = c->sigma[0] * c->sigma[3] - c->sigma[1] * c->sigma[2];
    const double si[4] = {c->sigma[3] / det, -c->sigma[1] / det, -c->sigma[2] / det, c->sigma[0] / det};
    const double thr = (1.0 - c->param_exploration) * K, dt = c->delta_t;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int k = 0; k < K; ++k) {
        double x = x0[0], y = x0[1], yaw = x0[2], s = 0.0;
        int p = p0;
        for (int t = 0; t < T; ++t) {
            double e0 = eps[((size_t)k * T + t) * 2], e1 = eps[((size_t)k * T + t) * 2 + 1];
            double v0 = (double)k < thr ? u[2 * t] + e0 : e0, v1 = (double)k < thr ? u[2 * t + 1] + e1 : e1;
            v0 = clampd(v0, -c->u_max0, c->u_max0);
            v1 = clampd(v1, -c->u_max1, c->u_max1);
            double nx = x + v0 * cos(yaw) * dt, ny = y + v0 * sin(yaw) * dt;
            yaw = yaw + v1 * dt;
            x = nx;
            y = ny;
            p = dd_nearest(ref, c->n_ref, per_rollout ? p : p0, x, y); /* frozen: always from the x0 index */
            double q0 = u[2 * t] * si[0] + u[2 * t + 1] * si[2], q1 = u[2 * t] * si[1] + u[2 * t + 1] * si[3];
            s = dd_state_cost(c, c->stage_w, ref, obs, p, x, y, yaw) + gamma * (q0 * v0 + q1 * v1);
        }
        if (per_rollout) p = dd_nearest(ref, c->n_ref, p, x, y); /* `_terminal_cost` :244 searches once more */
        S[k] = s + dd_state_cost(c, c->term_w, ref, obs, p, x, y, yaw);
    }
    *idx = p0;
    double rho = S[0], eta = 0.0;
    for (int k = 1; k < K; ++k) rho = S[k] < rho ? S[k] : rho;
    for (int k = 0; k < K; ++k) { wgt[k] = exp(-(1.0 / c->param_exploration) * (S[k] - rho)); eta += wgt[k]; }
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int t = 0; t < T; ++t) {
        double a0 = 0.0, a1 = 0.0;
        for (int k = 0; k < K; ++k) {
            a0 += wgt[k] / eta * eps[((size_t)k * T + t) * 2];
            a1 += wgt[k] / eta * eps[((size_t)k * T + t) * 2 + 1];
        }
        w_eps[2 * t] = a0;
        w_eps[2 * t + 1] = a1;
    }
    dd_moving_average(w_eps, filt, T);
    for (int i = 0; i < 2 * T; ++i) u[i] += filt[i];
    if (c->clamp_u_after_update)
        for (int t = 0; t < T; ++t) {
            u[2 * t] = clampd(u[2 * t], -c->u_max0, c->u_max0);
            u[2 * t + 1] = clampd(u[2 * t + 1], -c->u_max1, c->u_max1);
        }
    memmove(u, u + 2, sizeof(double) * 2 * (T - 1));
    u0_out[0] = u[0];
    u0_out[1] = u[1];
    if (stats) { stats[0] = rho; stats[1] = eta; stats[2] = p0; stats[3] = path_end; }
    free(w_eps);
    free(wgt);
    return 0;
}

int oracle_diffdrive_iteration_frozen(const oracle_cfg *c, const double *ref, const double *obs, const double *x0,
                                      const float *eps, double *u_prev, int *idx, double *S, double *u0_out,
                                      double *stats, int n_threads) {
    return oracle_diffdrive_iteration_independent(c, ref, obs, x0, eps, u_prev, idx, S, u0_out, stats, n_threads, 0);
}

/* ------------------------------------------------------------------------- race car -- */

#define TWO_PI_F 6.2831855f /* float32(2.0*np.pi), mppi_race_car.py:141 under NEP 50 */

/* `get_nearest_waypoint` mppi_race_car.py:157-174, SEARCH_INDEX_LEN = 200 */
static int rc_nearest(const float *ref, int n_ref, int p, float x, float y) {
    int best = 0;
    float bd = 0.f;
    for (int j = 0; j < 200 && p + j < n_ref; ++j) {
        float dx = x - ref[4 * (p + j)], dy = y - ref[4 * (p + j) + 1];
        float d = dx * dx + dy * dy;
        if (j == 0 || d < bd) { bd = d; best = j; }
    }
    return p + best;
}

/* `_is_collided` + `_affine_transform` mppi_race_car_obstacle.py:241-274 */
static float rc_collided(const oracle_cfg *c, const double *obs, float x, float y, float yaw) {
    const double vw = 3.0 * c->safety_margin, vl = 4.0 * c->safety_margin;
    const double sx[9] = {-0.5 * vl, -0.5 * vl, 0.0, 0.5 * vl, 0.5 * vl, 0.5 * vl, 0.0, -0.5 * vl, -0.5 * vl};
    const double sy[9] = {0.0, 0.5 * vw, 0.5 * vw, 0.5 * vw, 0.0, -0.5 * vw, -0.5 * vw, -0.5 * vw, 0.0};
    const float cs = cosf(yaw), sn = sinf(yaw);
    for (int m = 0; m < c->n_obs; ++m)
        for (int q = 0; q < 9; ++q) {
            float px = (float)sx[q] * cs - (float)sy[q] * sn + x;
            float py = (float)sx[q] * sn + (float)sy[q] * cs + y;
            double dx = (double)px - obs[3 * m], dy = (double)py - obs[3 * m + 1];
            if (dx * dx + dy * dy < obs[3 * m + 2] * obs[3 * m + 2]) return 1.f;
        }
    return 0.f;
}

static float rc_state_cost(const oracle_cfg *c, const float *w, const float *ref, const double *obs, int p,
                           const float *s) {
    float yaw = fmodf(s[2] + TWO_PI_F, TWO_PI_F); /* python %: sign follows the divisor */
    if (yaw < 0.f) yaw += TWO_PI_F;
    int i = rc_nearest(ref, c->n_ref, p, s[0], s[1]);
    float ex = s[0] - ref[4 * i], ey = s[1] - ref[4 * i + 1], eyaw = yaw - ref[4 * i + 2], ev = s[3] - ref[4 * i + 3];
    float cost = w[0] * (ex * ex) + w[1] * (ey * ey) + w[2] * (eyaw * eyaw) + w[3] * (ev * ev);
    if (c->n_obs > 0) cost += rc_collided(c, obs, s[0], s[1], s[2]) * 1.0e10f;
    return cost;
}

/* `_moving_average_filter` mppi_race_car.py:211-222 */
static void rc_moving_average(const float *xx, float *out, int T) {
    float *pad = (float *)malloc(sizeof(float) * (T + 10));
    for (int d = 0; d < 2; ++d) {
        for (int j = 0; j < 5; ++j) pad[j] = xx[2 * j + d];
        for (int j = 0; j < T; ++j) pad[5 + j] = xx[2 * j + d];
        for (int j = 0; j < 5; ++j) pad[5 + T + j] = xx[2 * (T - 5 + j) + d];
        for (int i = 0; i < T; ++i) {
            float s = 0.f;
            for (int m = i + 9; m >= i; --m) s += pad[m] * 0.1f;
            out[2 * i + d] = s;
        }
    }
    free(pad);
}

/* One `_calc_control_input` (mppi_race_car.py:55-121 / _obstacle :65-131), eps injected. */
int oracle_racecar_iteration(const oracle_cfg *c, const float *ref, const double *obs, const float *x0,
                             const float *eps, float *u_prev, int *idx, float *S, float *u0_out, double *stats) {
    const int K = c->K, T = c->T;
    if (T < 5) return -1;
    float *u = u_prev;
    float *w_eps = (float *)calloc((size_t)4 * T, sizeof(float)), *filt = w_eps + 2 * T;
    float *wgt = (float *)malloc(sizeof(float) * K);
    const int p = rc_nearest(ref, c->n_ref, *idx, x0[0], x0[1]); /* :61, frozen during rollouts */
    *idx = p;
    const float gamma = (float)(c->param_lambda * (1.0 - c->param_alpha));
    const float s00 = (float)c->sigma[0], s01 = (float)c->sigma[1], s10 = (float)c->sigma[2], s11 = (float)c->sigma[3];
    const float det = s00 * s11 - s01 * s10;
    const float si[4] = {s11 / det, -s01 / det, -s10 / det, s00 / det};
    const double thr = (1.0 - c->param_exploration) * K;
    const float dt = (float)c->delta_t, l = (float)c->wheel_base;
    const float m0 = (float)c->u_max0, m1 = (float)c->u_max1;
    float sw[4], tw[4];
    for (int i = 0; i < 4; ++i) { sw[i] = (float)c->stage_w[i]; tw[i] = (float)c->term_w[i]; }

    for (int k = 0; k < K; ++k) {
        float s[4] = {x0[0], x0[1], x0[2], x0[3]}, acc = 0.f;
        for (int t = 0; t < T; ++t) {
            float e0 = eps[((size_t)k * T + t) * 2], e1 = eps[((size_t)k * T + t) * 2 + 1];
            float v0 = (double)k < thr ? u[2 * t] + e0 : e0, v1 = (double)k < thr ? u[2 * t + 1] + e1 : e1;
            v0 = clampf(v0, -m0, m0);
            v1 = clampf(v1, -m1, m1);
            float nx = s[0] + s[3] * cosf(s[2]) * dt, ny = s[1] + s[3] * sinf(s[2]) * dt; /* `_F` :183-197 */
            float nyaw = s[2] + s[3] / l * tanf(v0) * dt, nv = s[3] + v1 * dt;
            s[0] = nx; s[1] = ny; s[2] = nyaw; s[3] = nv;
            float q0 = si[0] * v0 + si[1] * v1, q1 = si[2] * v0 + si[3] * v1; /* inv(Sigma) @ v */
            acc += rc_state_cost(c, sw, ref, obs, p, s) + gamma * (u[2 * t] * q0 + u[2 * t + 1] * q1); /* '+=' :84 */
        }
        S[k] = acc + rc_state_cost(c, tw, ref, obs, p, s);
    }
    float rho = S[0], eta = 0.f; /* `_compute_weight` :199-209 */
    for (int k = 1; k < K; ++k) rho = S[k] < rho ? S[k] : rho;
    const float nb = (float)(-1.0 / c->param_lambda);
    for (int k = 0; k < K; ++k) { wgt[k] = expf(nb * (S[k] - rho)); eta += wgt[k]; }
    for (int k = 0; k < K; ++k) wgt[k] = (1.0f / eta) * wgt[k];
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < K; ++k) {
            w_eps[2 * t] += wgt[k] * eps[((size_t)k * T + t) * 2];
            w_eps[2 * t + 1] += wgt[k] * eps[((size_t)k * T + t) * 2 + 1];
        }
    rc_moving_average(w_eps, filt, T);
    for (int i = 0; i < 2 * T; ++i) u[i] += filt[i];
    if (c->clamp_u_after_update)
        for (int t = 0; t < T; ++t) {
            u[2 * t] = clampf(u[2 * t], -m0, m0);
            u[2 * t + 1] = clampf(u[2 * t + 1], -m1, m1);
        }
    memmove(u, u + 2, sizeof(float) * 2 * (T - 1));
    u0_out[0] = u[0];
    u0_out[1] = u[1];
    if (stats) { stats[0] = rho; stats[1] = eta; stats[2] = p; stats[3] = p >= c->n_ref - 1; }
    free(w_eps);
    free(wgt);
    return 0;
}
