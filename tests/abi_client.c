/* abi_client.c -- a plain C99 consumer of include/mpopis.h: what a non-Python, non-Julia host links against.  Runs a short MountainCar
 * closed loop (env defaults live in the library, no track needed) with :cemppi on the device RNG and prints every control and reward with
 * full precision; tests/test_abi.py compiles it (gcc -std=c99 -pedantic: the header is valid C) and tests/test_gpu_host_api.py runs it on
 * the GPU and compares the text with the same calls made through the Python mirror.
 *   build: gcc -std=c99 -pedantic -Wall -Iinclude tests/abi_client.c -Lmpopis_amd/lib -lmpopis_hip -Wl,-rpath,$PWD/mpopis_amd/lib -o abi_client */
#include <stdio.h>
#include <string.h>
#include "mpopis.h"

static int fail(const mpopis_handle *h, const char *what, int rc) {
    fprintf(stderr, "%s failed: %d (%s)\n", what, rc, mpopis_last_error(h));
    return 1;
}

int main(void) {
    mpopis_config cfg;
    mpopis_handle *h = NULL;
    double Sigma[1] = {1.0};
    double control[2], reward[2], cost[2 * 64];
    int32_t iters[2];
    int rc, step;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.env_kind = MPOPIS_ENV_MOUNTAINCAR; cfg.num_cars = 0; cfg.policy = MPOPIS_POL_CEMPPI;
    cfg.num_samples = 64; cfg.horizon = 15; cfg.batch = 2; cfg.ais_its = 4; cfg.sigma_est = MPOPIS_SIGMA_EST_MLE;
    cfg.lambda = 0.1; cfg.alpha = 1.0; cfg.lambda_ais = 0.0; cfg.elite_threshold = 0.8; cfg.cma_sigma = 1.0; cfg.seed = 1234;
    printf("abi %d\n", mpopis_abi_version());
    if ((rc = mpopis_create(&cfg, &h)) != MPOPIS_OK) return fail(NULL, "mpopis_create", rc);
    if ((rc = mpopis_set_Sigma(h, Sigma, 1)) != MPOPIS_OK) return fail(h, "mpopis_set_Sigma", rc);
    if ((rc = mpopis_reset(h)) != MPOPIS_OK) return fail(h, "mpopis_reset", rc);
    for (step = 0; step < 5; ++step) {
        if ((rc = mpopis_policy_step(h, NULL, control, cost, NULL, NULL, NULL, iters)) != MPOPIS_OK) return fail(h, "mpopis_policy_step", rc);
        if ((rc = mpopis_env_step(h, control, reward)) != MPOPIS_OK) return fail(h, "mpopis_env_step", rc);
        printf("step %d control %.17g %.17g reward %.17g %.17g cost0 %.17g iters %d %d\n", step, control[0], control[1], reward[0], reward[1], cost[0],
               (int)iters[0], (int)iters[1]);
    }
    mpopis_destroy(h);
    return 0;
}
