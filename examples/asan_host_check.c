/* Host-side check of libmppi_hip.so under AddressSanitizer (make -C dnn-mppi-mpc_amd/csrc asan_check; no GPU needed):
 * the entry points that run on the host before any device work -- argument validation, error strings, create on a box
 * without a device -- are driven with good and bad arguments; ASan reports any out-of-bounds access or leak. */
#include <stdio.h>
#include <string.h>

#include "mppi_hip.h"

static int fails = 0;
#define EXPECT(cond)                                              \
    do {                                                          \
        if (!(cond)) { printf("FAILED: %s (line %d)\n", #cond, __LINE__); ++fails; } \
    } while (0)

int main(void) {
    EXPECT(mppi_abi_version() == MPPI_ABI_VERSION);
    mppi_config c;
    memset(&c, 0, sizeof(c));
    mppi_handle *h = NULL;
    EXPECT(mppi_create(NULL, &h) == MPPI_ERR_BAD_ARG);
    EXPECT(mppi_create(&c, NULL) == MPPI_ERR_BAD_ARG);
    EXPECT(mppi_create(&c, &h) == MPPI_ERR_BAD_ARG); /* struct_size 0 */
    EXPECT(strlen(mppi_last_error(NULL)) > 0);
    c.struct_size = (int32_t)sizeof(c);
    c.K = 0;
    c.T = 10;
    EXPECT(mppi_create(&c, &h) == MPPI_ERR_SHAPE);
    c.K = 256;
    c.model = 7;
    EXPECT(mppi_create(&c, &h) == MPPI_ERR_BAD_ARG);
    c.model = MPPI_MODEL_DIFFDRIVE;
    c.search_window = 20;
    c.filter_window = 10;
    c.sigma[0] = 0.1; c.sigma[3] = 0.01;
    c.param_exploration = 0.05; c.param_lambda = 1.0; c.param_alpha = 0.2;
    c.delta_t = 0.1;
    c.sigma[1] = c.sigma[2] = 1.0; /* not positive definite */
    EXPECT(mppi_create(&c, &h) == MPPI_ERR_BAD_ARG);
    c.sigma[1] = c.sigma[2] = 0.0;
    const int rc = mppi_create(&c, &h); /* a box without an MI355X: MPPI_ERR_NO_DEVICE, with the reason as text */
    if (rc == MPPI_OK) {
        double path[6] = {0, 0, 0, 1, 1, 0};
        EXPECT(mppi_set_ref_path(h, path, 2, 3) == MPPI_OK);
        EXPECT(mppi_set_ref_path(h, path, 0, 3) == MPPI_ERR_SHAPE);
        EXPECT(mppi_set_waypoint_idx(h, 5) == MPPI_ERR_BAD_ARG);
        EXPECT(mppi_destroy(h) == MPPI_OK);
    } else {
        EXPECT(rc == MPPI_ERR_NO_DEVICE);
        EXPECT(strstr(mppi_last_error(NULL), "HIP device") != NULL || strstr(mppi_last_error(NULL), "gfx950") != NULL);
    }
    EXPECT(mppi_destroy(NULL) == MPPI_OK);
    EXPECT(mppi_comm_unique_id(NULL) == MPPI_ERR_BAD_ARG);
    EXPECT(mppi_get_rollout_kernel(NULL, NULL, 0) == MPPI_ERR_BAD_ARG);
    mppi_cb_config cb;
    memset(&cb, 0, sizeof(cb));
    mppi_cb_handle *hc = NULL;
    EXPECT(mppi_cb_create(&cb, &hc) != MPPI_OK);
    printf(fails ? "asan_host_check: %d expectation(s) failed\n" : "asan_host_check: ok\n", fails);
    return fails ? 1 : 0;
}
