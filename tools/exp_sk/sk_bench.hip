// scratch experiment (not part of the library): which part of the short-reduction kernel bounds it?
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include "short_k_fwd_exp.h"
int main() {
    const int N = 3136, K = 64;
    for (int R : {8192, 32768}) {
        float *X, *W, *b, *M, *Y, *Y2;
        hipMalloc(&X, (size_t)R * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&b, N * 4); hipMalloc(&M, (size_t)(R / 32) * N * 4);
        hipMalloc(&Y, (size_t)R * N * 4); hipMalloc(&Y2, (size_t)R * N * 4);
        hipMemset(X, 0, (size_t)R * K * 4); hipMemset(W, 0, (size_t)N * K * 4); hipMemset(b, 0, N * 4); hipMemset(M, 0, (size_t)(R / 32) * N * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 3; ++mode) {
            auto run = [&]() { a0_short_k_fwd_launch(0, X, K, W, b, mode ? M : nullptr, 32, Y, mode == 2 ? Y2 : nullptr, R, N, 1); };
            for (int i = 0; i < 3; ++i) run();
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 30; ++i) run();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("R=%d mode=%d: %.1f us\n", R, mode, ms * 1e3 / 30);
        }
        hipFree(X); hipFree(W); hipFree(b); hipFree(M); hipFree(Y); hipFree(Y2);
    }
    return 0;
}
