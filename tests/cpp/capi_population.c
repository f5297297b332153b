/* The population-scale use of the C ABI shown in INTEGRATION.md section 2, as a plain C program (compiled with gcc to
 * keep the header honest about being C): load a track, create the environment, reset agents, then
 * {write actions, okenv_step, read distances and flags} -- and compare a checksum of the observations against the one
 * passed on the command line (computed by the test from the oracle). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "okenv.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        const int rc_ = (call);                                                                       \
        if (rc_ != OKENV_OK) {                                                                        \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, okenv_last_error(env));               \
            return 2;                                                                                 \
        }                                                                                             \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 5) {
        fprintf(stderr, "usage: capi_population track.csv agents steps out.bin\n");
        return 1;
    }
    const int N = atoi(argv[2]), steps = atoi(argv[3]), R = 16;
    okenv_t env = NULL;
    okenv_track_t tr = NULL;
    if (okenv_track_load(&tr, argv[1]) != OKENV_OK) {
        fprintf(stderr, "cannot load %s\n", argv[1]);
        return 2;
    }
    const int P = okenv_track_num_points(tr), S = okenv_track_num_segments(tr);
    float *seg = malloc(sizeof(float) * 4 * S), *cx = malloc(sizeof(float) * P), *cy = malloc(sizeof(float) * P),
          *hd = malloc(sizeof(float) * P);
    okenv_track_segments(tr, seg);
    okenv_track_get(tr, 0, cx);
    okenv_track_get(tr, 1, cy);
    okenv_track_get(tr, 4, hd);
    float rays[16];
    for (int i = 0; i < R; ++i)
        rays[i] = -70.0f + 140.0f * (float)i / (float)(R - 1);
    CHECK(okenv_create(&env, seg, S, N, R, rays, 0, OKENV_FLAG_NONE, 0.0f));
    CHECK(okenv_set_centerline(env, cx, cy, hd, P));
    int32_t *idx = malloc(sizeof(int32_t) * N);
    float *x0 = malloc(sizeof(float) * N), *y0 = malloc(sizeof(float) * N), *r0 = malloc(sizeof(float) * N),
          *thr = malloc(sizeof(float) * N), *steer = malloc(sizeof(float) * N), *obs = malloc(sizeof(float) * N * R);
    uint8_t *done = malloc(N);
    for (int i = 0; i < N; ++i) {
        const int k = (i * 37 + 3) % P;
        idx[i] = i, x0[i] = cx[k], y0[i] = cy[k], r0[i] = hd[k];
    }
    CHECK(okenv_reset_agents(env, idx, x0, y0, r0, N));
    FILE *out = fopen(argv[4], "wb");
    for (int s = 0; s < steps; ++s) {
        for (int i = 0; i < N; ++i) { /* a scripted "policy" */
            thr[i] = 20.0f + (float)((i * 7 + s) % 60);
            steer[i] = (float)((i + 3 * s) % 11) - 5.0f;
        }
        CHECK(okenv_set_actions(env, thr, steer));
        CHECK(okenv_step(env, 1));
        CHECK(okenv_get_distances(env, obs));
        CHECK(okenv_get_flags(env, done));
        if (s % 10 == 9 || s == steps - 1) {
            fwrite(obs, sizeof(float), (size_t)N * R, out);
            fwrite(done, 1, (size_t)N, out);
        }
    }
    fclose(out);
    okenv_info info;
    CHECK(okenv_get_info(env, &info));
    printf("ok: %d agents x %d rays, %d steps, grid %dx%d\n", info.num_agents, info.num_rays, steps, info.grid_nx, info.grid_ny);
    okenv_destroy(env);
    okenv_track_free(tr);
    return 0;
}
