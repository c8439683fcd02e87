/* A host that is NOT Python: plain C against include/agent0_hip.h and the HIP runtime only (tests/test_gpu_engine.py compiles and runs it on the GPU box).
 * It creates a learner handle (library-owned HBM), fills its parameters, runs three BaseLearner.train calls (agent.py:124-169) on synthetic replay rows through
 * a0_learner_update — one C call per update — and prints the status words, the mean loss of the last update and a parameter checksum as one line of JSON. */
#include <hip/hip_runtime_api.h>      /* the C API of the HIP runtime: gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "agent0_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, a0_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static unsigned lcg(unsigned* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void) {
    const int B = 32, A = 4, OBS = 4 * 84 * 84;
    a0_learner_desc d = {A, 1 /* dueling */, 1 /* double-Q */, B, 3 /* n-step */, 0.99, 5e-4, 0.0, 2 /* target sync every 2 updates */};
    a0_learner* L = NULL;
    CHECK(a0_learner_create(&d, &L));
    const long long n = a0_learner_param_floats(L);
    unsigned seed = 12345u;
    float* hp = (float*)malloc((size_t)n * sizeof(float));
    for (long long i = 0; i < n; ++i) hp[i] = ((float)(lcg(&seed) % 2001) - 1000.0f) * 2e-5f;      /* small weights in [-0.02, 0.02] */
    float* dp = NULL;
    HIP(hipMalloc((void**)&dp, (size_t)n * sizeof(float)));
    HIP(hipMemcpy(dp, hp, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    CHECK(a0_learner_set_params(L, dp, NULL, NULL));
    /* a batch of replay rows st || st_next, actions, n-step rewards, done flags, importance weights */
    unsigned char* hf = (unsigned char*)malloc((size_t)B * 2 * OBS);
    for (long long i = 0; i < (long long)B * 2 * OBS; ++i) hf[i] = (unsigned char)((lcg(&seed) & 3u) ? 0u : (lcg(&seed) & 255u));
    int ha[32]; float hr[32], hd[32], hw[32];
    for (int b = 0; b < B; ++b) { ha[b] = (int)(lcg(&seed) % (unsigned)A); hr[b] = (float)((int)(lcg(&seed) % 3u) - 1); hd[b] = (lcg(&seed) % 10u) == 0u ? 1.f : 0.f; hw[b] = 1.f; }
    unsigned char* df = NULL; int* da = NULL; float *dr = NULL, *dd = NULL, *dw = NULL, *dl = NULL;
    HIP(hipMalloc((void**)&df, (size_t)B * 2 * OBS)); HIP(hipMalloc((void**)&da, B * sizeof(int))); HIP(hipMalloc((void**)&dr, B * 4)); HIP(hipMalloc((void**)&dd, B * 4));
    HIP(hipMalloc((void**)&dw, B * 4)); HIP(hipMalloc((void**)&dl, B * 4));
    HIP(hipMemcpy(df, hf, (size_t)B * 2 * OBS, hipMemcpyHostToDevice)); HIP(hipMemcpy(da, ha, B * sizeof(int), hipMemcpyHostToDevice));
    HIP(hipMemcpy(dr, hr, B * 4, hipMemcpyHostToDevice)); HIP(hipMemcpy(dd, hd, B * 4, hipMemcpyHostToDevice)); HIP(hipMemcpy(dw, hw, B * 4, hipMemcpyHostToDevice));
    for (int step = 0; step < 3; ++step) CHECK(a0_learner_update(L, df, NULL, 2LL * OBS, da, dr, dd, dw, dl, NULL));
    float* don = NULL; float* dtg = NULL; int* dst = NULL;
    HIP(hipMalloc((void**)&don, (size_t)n * 4)); HIP(hipMalloc((void**)&dtg, (size_t)n * 4)); HIP(hipMalloc((void**)&dst, 8 * sizeof(int)));
    CHECK(a0_learner_get(L, don, dtg, NULL, NULL, dst, NULL));
    HIP(hipDeviceSynchronize());
    float hl[32]; int st[8];
    float* hon = (float*)malloc((size_t)n * 4); float* htg = (float*)malloc((size_t)n * 4);
    HIP(hipMemcpy(hl, dl, B * 4, hipMemcpyDeviceToHost)); HIP(hipMemcpy(st, dst, 8 * sizeof(int), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(hon, don, (size_t)n * 4, hipMemcpyDeviceToHost)); HIP(hipMemcpy(htg, dtg, (size_t)n * 4, hipMemcpyDeviceToHost));
    double mean = 0.0, moved = 0.0, tdiff = 0.0; int finite = 1;
    for (int b = 0; b < B; ++b) { mean += hl[b] / B; if (!isfinite(hl[b])) finite = 0; }
    for (long long i = 0; i < n; ++i) { moved += fabs((double)hon[i] - (double)hp[i]); tdiff += fabs((double)hon[i] - (double)htg[i]); if (!isfinite(hon[i])) finite = 0; }
    printf("{\"update_steps\": %d, \"nan_skipped\": %d, \"mean_loss\": %.6g, \"finite\": %d, \"param_floats\": %lld, \"moved_l1\": %.6g, \"online_minus_target_l1\": %.6g}\n", st[1], st[2], mean,
           finite, n, moved, tdiff);
    CHECK(a0_learner_destroy(L));
    return 0;
}
