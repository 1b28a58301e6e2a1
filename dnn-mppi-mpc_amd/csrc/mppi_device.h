// Device helpers shared by the analytic (mppi_kernels.hip) and learned-dynamics (mppi_mlp.hip) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

#include "mathfn.h"
#include "mppi_kernels.h"
#include "philox.h"
#include "wave_ops.h"

namespace mppi {

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------

template <typename R> __device__ __forceinline__ R dist2(const R *__restrict__ ref, int i, R x, R y) {
    const R dx = x - ref[4 * i], dy = y - ref[4 * i + 1];
    return dx * dx + dy * dy;
}

// first-minimum argmin over ref[c .. c+wlen) for this lane's (x, y); c, wlen wave-uniform
// (`get_nearest_waypoint` mppi_race_car.py:157-174, `_get_nearest_waypoint` mppi_differential_drive.py:201-220)
template <typename R>
__device__ __forceinline__ int nearest_in_window(const R *__restrict__ ref, int c, int wlen, R x, R y) {
    R best = dist2(ref, c, x, y);
    int bj = 0;
#pragma unroll 4
    for (int j = 1; j < wlen; ++j) {
        const R d = dist2(ref, c + j, x, y);
        if (d < best) { best = d; bj = j; }
    }
    return c + bj;
}

// Same search for ONE wave-uniform position with the candidates spread over the lanes.
template <typename R>
__device__ __forceinline__ int nearest_uniform(const R *__restrict__ ref, int c, int wlen, R x, R y, int lane) {
    R best = R(INFINITY);
    int bj = INT_MAX;
    for (int j = lane; j < wlen; j += 64) {
        const R d = dist2(ref, c + j, x, y);
        if (d < best) { best = d; bj = j; }
    }
    wv::argmin_first(best, bj);
    return c + bj;
}

template <typename R> __device__ __forceinline__ int window_len(int window, int n_ref, int c) {
    const int rem = n_ref - c;
    return rem < window ? rem : window;
}

// collision indicator of one state (mppi_differential_drive_obs.py:301-313,
// mppi_race_car_obstacle.py:241-274)
template <typename R> __device__ __forceinline__ bool collided(const KParams<R> &P, R x, R y, R yaw) {
    bool hit = false;
    if (P.obstacle_model == OBS_CIRCLE) {
        for (int m = 0; m < P.n_obs; ++m) {
            const R dx = x - P.obs[4 * m], dy = y - P.obs[4 * m + 1];
            hit |= dx * dx + dy * dy < P.obs[4 * m + 2];
        }
    } else if (P.obstacle_model == OBS_OUTLINE) {
        R sn, cs;
        mf::sincos_(yaw, sn, cs);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const R px = P.shape_x[q] * cs - P.shape_y[q] * sn + x;
            const R py = P.shape_x[q] * sn + P.shape_y[q] * cs + y;
            for (int m = 0; m < P.n_obs; ++m) {
                const R dx = px - P.obs[4 * m], dy = py - P.obs[4 * m + 1];
                hit |= dx * dx + dy * dy < P.obs[4 * m + 2];
            }
        }
    }
    return hit;
}

// weighted squared tracking error against waypoint i (`_compute_cost` :222-236, `_c` mppi_race_car.py:137-146)
template <typename R, int MODEL>
__device__ __forceinline__ R tracking_cost(const KParams<R> &P, const R (&w)[4], bool wrap, int i, R x, R y, R yaw,
                                           R vel) {
    const R *r = P.ref + 4 * i;
    if (wrap) yaw = mf::pymod(yaw + P.two_pi, P.two_pi);
    const R ex = x - r[0], ey = y - r[1], eyaw = yaw - r[2];
    R c = w[0] * (ex * ex) + w[1] * (ey * ey) + w[2] * (eyaw * eyaw);
    if (MODEL == MODEL_RACE) {
        const R ev = vel - r[3];
        c += w[3] * (ev * ev);
    }
    return c;
}


// controller state of this launch: *st, with the observed state / x0 index overridden by kernel arguments
template <typename R> __device__ __forceinline__ DevState load_state(const KParams<R> &P) {
    DevState sv = *P.st;  // one batch of scalar loads
    if (P.use_args) {
        sv.x0[0] = P.x0_arg[0]; sv.x0[1] = P.x0_arg[1]; sv.x0[2] = P.x0_arg[2]; sv.x0[3] = P.x0_arg[3];
        sv.c = P.c_arg;
    }
    return sv;
}

__device__ __forceinline__ bool round_unresolved(const DevState *st, int K) {
    const int fk = st->first_k;
    return fk != NO_TRIGGER && fk + 1 < K;
}

}  // namespace mppi
