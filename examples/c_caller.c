/*
 * A C caller of libmppi_hip.so: the reference driver's loop (controllers/mppi_differential_drive.py:394-419, `animate`
 * :305-367 without the plotting) against the C ABI alone -- no Python, no PyTorch.
 *
 *   gcc -O2 -I include examples/c_caller.c -o c_caller -L dnn-mppi-mpc_amd/lib -lmppi_hip -Wl,-rpath,$PWD/dnn-mppi-mpc_amd/lib -lm
 *   ./c_caller [K] [T] [iterations]
 *
 * Prints one line per host-in-the-loop iteration ("it idx u0_v u0_w x y yaw") and then the state after the same number of
 * device-resident closed-loop iterations; tests/test_gpu_c_caller.py holds both to the Python mirror of the classes.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mppi_hip.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != MPPI_OK) {                                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mppi_last_error(h));               \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 4096, T = argc > 2 ? atoi(argv[2]) : 50, n_iter = argc > 3 ? atoi(argv[3]) : 5;
    enum { N_REF = 100 };
    mppi_handle *h = NULL;

    /* MPPIAlgorithms(...) with the `__main__` parameters (:400-410) */
    mppi_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.model = MPPI_MODEL_DIFFDRIVE;
    cfg.precision = MPPI_PREC_F32;
    cfg.K = K;
    cfg.T = T;
    cfg.delta_t = 0.1;
    cfg.u_max[0] = 5.0;   /* max_speed */
    cfg.u_max[1] = 3.14;  /* max_omega */
    cfg.param_exploration = 0.0001;
    cfg.param_lambda = 1.0;
    cfg.param_alpha = 0.2;
    cfg.sigma[0] = 0.1;
    cfg.sigma[3] = 0.01;
    for (int i = 0; i < 3; ++i) {
        const double w[3] = {5.0, 5.0, 10.0};
        cfg.stage_cost_weight[i] = cfg.terminal_cost_weight[i] = w[i];
    }
    cfg.beta_mode = MPPI_BETA_INV_EXPLORATION;   /* exp(-(S - rho) / param_exploration), :175 */
    cfg.accumulate_stage_cost = 0;               /* S[k] = ..., :124 */
    cfg.waypoint_mode = MPPI_WAYPOINT_SEQUENTIAL;
    cfg.search_window = 20;                      /* SEARCH_IDX_LEN, :204 */
    cfg.clamp_rollout = 1;
    cfg.clamp_u_after_update = 0;                /* visualisation off */
    cfg.filter_mode = MPPI_FILTER_DIFFDRIVE;
    cfg.filter_window = 10;
    cfg.collision_penalty = 1.0e10;
    cfg.seed = 2024;
    if (mppi_create(&cfg, &h) != MPPI_OK) {
        fprintf(stderr, "mppi_create: %s\n", mppi_last_error(NULL));
        return 1;
    }

    /* generate_point_trajectory((0, 0), (10, -5), 100), :385-389 */
    double ref[N_REF][3];
    for (int i = 0; i < N_REF; ++i) {
        const double s = (double)i / (N_REF - 1);
        ref[i][0] = 10.0 * s;
        ref[i][1] = -5.0 * s;
        ref[i][2] = atan2(-5.0, 10.0);
    }
    CHECK(mppi_set_ref_path(h, &ref[0][0], N_REF, 3));

    /* the driver's loop with the host in it: observed state in, first control out, DifferentialDrive.update_state (:33-40) */
    double x[3] = {0.0, 0.0, 0.0}, u0[2];
    double *u = (double *)malloc(sizeof(double) * 2 * T);
    mppi_stats st;
    for (int it = 0; it < n_iter; ++it) {
        CHECK(mppi_step(h, x, NULL /* in-kernel Philox */, u, u0, &st, NULL /* default stream */));
        x[0] += u0[0] * cos(x[2]) * cfg.delta_t;
        x[1] += u0[0] * sin(x[2]) * cfg.delta_t;
        x[2] += u0[1] * cfg.delta_t;
        printf("%d %d %.9g %.9g %.9g %.9g %.9g\n", it, st.idx_after, u0[0], u0[1], x[0], x[1], x[2]);
    }

    /* the same loop resident on the device: restart, then n_iter iterations in one call */
    memset(u, 0, sizeof(double) * 2 * T);
    CHECK(mppi_set_u_prev(h, u));
    CHECK(mppi_set_waypoint_idx(h, 0));
    CHECK(mppi_set_iteration(h, 0));
    const double x_init[3] = {0.0, 0.0, 0.0};
    CHECK(mppi_set_state(h, x_init));
    CHECK(mppi_run_closed_loop(h, n_iter, NULL, &st, NULL));
    double xd[3];
    CHECK(mppi_get_state(h, xd));
    printf("device %d %.9g %.9g %.9g\n", st.idx_after, xd[0], xd[1], xd[2]);
    free(u);
    CHECK(mppi_destroy(h));
    return 0;
}
