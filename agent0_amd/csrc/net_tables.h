// Geometry and gather tables of the Nature CNN (host side; shared by net.hip and tests/host_emul.cpp).
#pragma once
#include "a0_defs.h"
#include <vector>

struct a0_net_core {          // everything the layer orchestration needs; table pointers are device (HIP) or host (emulation)
    int C, H, W;
    int H1, W1, H2, W2, H3, W3, feat;
    int K1, K2, K3;
    const a0_i4 *ktab1, *ktab2, *ktab3, *ktab_d3, *ktab_d2;
    const int *wtab_d3, *wtab_d2[4];
};

struct a0_net_tables {
    std::vector<a0_i4> ktab1, ktab2, ktab3, ktab_d3, ktab_d2;
    std::vector<int> wtab_d3, wtab_d2[4];
};

static inline bool a0_net_core_init(a0_net_core& n, int C, int H, int W) {
    n.C = C; n.H = H; n.W = W;
    n.H1 = (H - 8) / 4 + 1; n.W1 = (W - 8) / 4 + 1;
    n.H2 = (n.H1 - 4) / 2 + 1; n.W2 = (n.W1 - 4) / 2 + 1;
    n.H3 = n.H2 - 2; n.W3 = n.W2 - 2;
    n.feat = n.H3 * n.W3 * 64;
    n.K1 = C * 64; n.K2 = 4 * 4 * 32; n.K3 = 3 * 3 * 64;
    return C >= 1 && H >= 8 && W >= 8 && n.H1 >= 4 && n.W1 >= 4 && n.H3 >= 1 && n.W3 >= 1;
}

static inline void a0_net_build_tables(const a0_net_core& n, a0_net_tables& t) {
    // conv1: k = c*64 + kh*8 + kw (reference weight order (c,kh,kw)), u8 planes [c][H][W]
    t.ktab1.resize(n.K1 / 4);
    for (int k = 0; k < n.K1; k += 4) {
        int c = k / 64, kh = (k % 64) / 8, kw = k % 8;
        t.ktab1[k / 4] = a0_i4{c * n.H * n.W + kh * n.W + kw, 0, 0, 0};
    }
    // conv2: k = (kh*4 + kw)*32 + c over act1 [H1][W1][32]
    t.ktab2.resize(n.K2 / 4);
    for (int k = 0; k < n.K2; k += 4) {
        int c = k % 32, kw = (k / 32) % 4, kh = k / 128;
        t.ktab2[k / 4] = a0_i4{(kh * n.W1 + kw) * 32 + c, kh, kw, 0};
    }
    // conv3: k = (kh*3 + kw)*64 + c over act2 [H2][W2][64]
    t.ktab3.resize(n.K3 / 4);
    for (int k = 0; k < n.K3; k += 4) {
        int c = k % 64, kw = (k / 64) % 3, kh = k / 192;
        t.ktab3[k / 4] = a0_i4{(kh * n.W2 + kw) * 64 + c, kh, kw, 0};
    }
    // conv3 data gradient: 3x3 taps over dY3 [H3][W3][64], pad 2; k = (kh'*3 + kw')*64 + oc, weight tap (2-kh', 2-kw')
    {
        const int Kd = 3 * 3 * 64;
        t.ktab_d3.resize(Kd / 4);
        t.wtab_d3.resize(Kd);
        for (int k = 0; k < Kd; ++k) {
            int oc = k % 64, kw = (k / 64) % 3, kh = k / 192;
            if (k % 4 == 0) t.ktab_d3[k / 4] = a0_i4{(kh * n.W3 + kw) * 64 + oc, kh, kw, 0};
            t.wtab_d3[k] = oc * n.K3 + ((2 - kh) * 3 + (2 - kw)) * 64;
        }
    }
    // conv2 data gradient, four stride phases: 2x2 taps over dY2 [H2][W2][64], pad 1; k = (kh'*2 + kw')*64 + oc,
    // weight tap (ph + 2(1-kh'), pw + 2(1-kw'))
    {
        const int Kd = 2 * 2 * 64;
        t.ktab_d2.resize(Kd / 4);
        for (int k = 0; k < Kd; k += 4) {
            int oc = k % 64, kw = (k / 64) % 2, kh = k / 128;
            t.ktab_d2[k / 4] = a0_i4{(kh * n.W2 + kw) * 64 + oc, kh, kw, 0};
        }
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                std::vector<int>& w = t.wtab_d2[ph * 2 + pw];
                w.resize(Kd);
                for (int k = 0; k < Kd; ++k) {
                    int oc = k % 64, kwp = (k / 64) % 2, khp = k / 128;
                    int kh = ph + 2 * (1 - khp), kw = pw + 2 * (1 - kwp);
                    w[k] = oc * n.K2 + (kh * 4 + kw) * 32;
                }
            }
    }
}
