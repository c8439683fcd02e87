"""GPU (MI355X): the dense GEMM entry points (a0_dense_fwd / _dgrad / _wgrad: model.py:112-114,144-146 and their autograd backward)
on both matrix pipes — the split-operand bf16 kernel (igemm_x9.h, default) and the fp32 fmaf-chain kernel (igemm.h) — against an
fp64 evaluation, at ragged shapes (rows / columns / reduction lengths that are not multiples of any tile; the ABI asks for N, K and ldx in multiples of 4).

Tolerance: both kernels accumulate exact products in fp32, so either one differs from fp64 by the fp32 accumulation error only:
a few 1e-7 of sum_k |a||b|.  The assert is 2e-6 of that scale, and the two kernels must agree with each other to the same bound."""
import numpy as np
import pytest
import torch

import recipe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from agent0_amd.ops import HipOps
    ops = HipOps()
    prev = ops.gemm_mode()
    yield ops
    ops.gemm_mode(prev)


def D(hip, x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(hip.device)


def _scale_close(got, want64, scale64, what, tol=2e-6):
    err = np.abs(got.astype(np.float64) - want64) / np.maximum(scale64, 1e-30)
    assert float(err.max()) < tol, f"{what}: {float(err.max()):.3e} of the accumulated magnitude"


SHAPES = [(1, 4, 512), (7, 36, 100), (256, 512, 3136), (300, 40, 64), (512, 32, 512), (513, 132, 96), (64, 512, 36), (2048, 64, 3136),
          (16384, 512, 96), (16500, 64, 256)]      # the last two reach the large-tile kernels (forward: 512 tiles of 128 x 128; data gradient: 258)


@pytest.mark.parametrize("R,N,K", SHAPES)
def test_dense_fwd_both_pipes(hip, R, N, K):
    g = recipe.gen(R * 31 + N * 7 + K)
    X = g.standard_normal((R, K)).astype(np.float32)
    W = (g.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = g.standard_normal(N).astype(np.float32)
    want = X.astype(np.float64) @ W.astype(np.float64).T + b
    scale = np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64).T + np.abs(b)
    outs = []
    for mode in (1, 0):
        hip.gemm_mode(mode)
        Y = hip.empty(R * N)
        need = hip.dense_fwd_scratch(R, N, K)
        scratch = hip.empty(max(need, 1))
        for relu in (0, 1):
            hip.dense_fwd(D(hip, X), K, D(hip, W), D(hip, b), Y, R, N, K, relu, scratch)
            got = Y.cpu().numpy().reshape(R, N)
            _scale_close(got, np.maximum(want, 0) if relu else want, scale, f"dense_fwd mode={mode} relu={relu} {R}x{N}x{K}")
        outs.append(got)
    _scale_close(outs[0], outs[1].astype(np.float64), scale, "split-operand vs fp32-chain kernel")


@pytest.mark.parametrize("R,N,K", SHAPES)
def test_dense_dgrad_both_pipes(hip, R, N, K):
    g = recipe.gen(R * 13 + N * 5 + K + 1)
    dY = g.standard_normal((R, N)).astype(np.float32)
    W = (g.standard_normal((N, K)) * 0.05).astype(np.float32)
    act = g.standard_normal((R, K)).astype(np.float32)
    want = dY.astype(np.float64) @ W.astype(np.float64)
    scale = np.abs(dY).astype(np.float64) @ np.abs(W).astype(np.float64)
    for mode in (1, 0):
        hip.gemm_mode(mode)
        dX = hip.empty(R * K)
        hip.dense_dgrad(D(hip, dY), D(hip, W), D(hip, act), dX, R, N, K)
        _scale_close(dX.cpu().numpy().reshape(R, K), want * (act > 0), scale, f"dense_dgrad masked mode={mode}")
        hip.dense_dgrad(D(hip, dY), D(hip, W), None, dX, R, N, K)
        _scale_close(dX.cpu().numpy().reshape(R, K), want, scale, f"dense_dgrad mode={mode}")


@pytest.mark.parametrize("R,N,K", SHAPES + [(8192, 512, 640)])     # the last one is deep enough for the split kernel's 128 x 128 tiles
def test_dense_wgrad_both_pipes(hip, R, N, K):
    """grad block = [dW (N x K) | db (N)]; db is the by-product row sum of the staged dY^T tiles."""
    g = recipe.gen(R * 3 + N * 11 + K + 2)
    dY = g.standard_normal((R, N)).astype(np.float32)
    X = g.standard_normal((R, K)).astype(np.float32)
    want_w = dY.astype(np.float64).T @ X.astype(np.float64)
    scale_w = np.abs(dY).astype(np.float64).T @ np.abs(X).astype(np.float64)
    want_b = dY.astype(np.float64).sum(0)
    scale_b = np.abs(dY).astype(np.float64).sum(0)
    for mode in (1, 0):
        hip.gemm_mode(mode)
        grad = hip.empty(N * K + N)
        slabs = hip.empty(max(hip.dense_wgrad_scratch(R, N, K), 1))
        hip.dense_wgrad(D(hip, dY), D(hip, X), K, grad, R, N, K, slabs)
        got = grad.cpu().numpy()
        _scale_close(got[: N * K].reshape(N, K), want_w, scale_w, f"dense_wgrad dW mode={mode}")
        _scale_close(got[N * K:], want_b, scale_b, f"dense_wgrad db mode={mode}")


def test_split_is_exact_for_extreme_operands(hip):
    """Operands spanning 30 binades, exact powers of two, negative zero and denormal-adjacent values: the three bf16 terms
    reproduce every fp32 operand exactly, so a one-term-per-row product (K = 1 live element) returns the exact fp32 product."""
    R, N, K = 64, 64, 32
    g = recipe.gen(99)
    x = (g.standard_normal(R) * np.exp2(g.integers(-15, 15, R))).astype(np.float32)
    w = (g.standard_normal(N) * np.exp2(g.integers(-15, 15, N))).astype(np.float32)
    x[:4] = [1.0, -0.0, 2.0 ** -20, 16777215.0]
    w[:4] = [1.0, 3.0, -(2.0 ** 10), 1.0 + 2.0 ** -23]
    X = np.zeros((R, K), np.float32); X[:, 5] = x
    W = np.zeros((N, K), np.float32); W[:, 5] = w
    hip.gemm_mode(1)
    Y = hip.empty(R * N)
    scratch = hip.empty(max(hip.dense_fwd_scratch(R, N, K), 1))
    hip.dense_fwd(D(hip, X), K, D(hip, W), hip.zeros(N), Y, R, N, K, 0, scratch)
    got = Y.cpu().numpy().reshape(R, N)
    want = (x.astype(np.float64)[:, None] * w.astype(np.float64)[None, :])
    # nine exact partial products summed in fp32: the result is the fp32 product up to the roundings of 8 small additions
    assert np.all(np.abs(got - want) <= 4 * np.spacing(np.abs(want).astype(np.float32)).astype(np.float64))


def test_dense_fwd_mul_equals_dense_fwd_then_hadamard(hip):
    """a0_dense_fwd_mul (embedding x state features in the GEMM's epilogue, model.py:244-247) against a0_dense_fwd + a0_hadamard_fwd:
    bit-identical, and against fp64."""
    B, n, N, K = 40, 32, 3136, 64             # 1280 rows: an unsplit GEMM
    R = B * n
    assert hip.dense_fwd_scratch(R, N, K) == 0
    g = recipe.gen(11)
    X = g.standard_normal((R, K)).astype(np.float32)
    W = (g.standard_normal((N, K)) * 0.1).astype(np.float32)
    b = g.standard_normal(N).astype(np.float32)
    M = g.standard_normal((B, N)).astype(np.float32)
    Xd, Wd, bd, Md = D(hip, X), D(hip, W), D(hip, b), D(hip, M)
    fused, emb, ref = hip.empty(R * N), hip.empty(R * N), hip.empty(R * N)
    hip.dense_fwd_mul(Xd, K, Wd, bd, Md, n, fused, R, N, K, 1)
    hip.dense_fwd(Xd, K, Wd, bd, emb, R, N, K, 1, None)
    hip.hadamard_fwd(emb, Md, ref, B, n, N)
    assert torch.equal(fused, ref)
    want = np.maximum(X.astype(np.float64) @ W.astype(np.float64).T + b, 0) * np.repeat(M.astype(np.float64), n, axis=0)
    scale = (np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64).T + np.abs(b)) * np.abs(np.repeat(M.astype(np.float64), n, axis=0)) + 1e-30
    assert float((np.abs(fused.cpu().numpy().reshape(R, N) - want) / scale).max()) < 2e-6


def test_dense_wgrad_multi_equals_single_calls(hip):
    """a0_dense_wgrad_multi (head + fc1 + cosine-embedding weight gradients, one slab reduction) against one a0_dense_wgrad per layer:
    bit-identical gradient blocks."""
    g = recipe.gen(12)
    R = 512
    shapes = [(R, 32, 512), (R, 512, 3136), (R, 3136, 64)]
    layers, singles = [], []
    for (r, N, K) in shapes:
        dY = D(hip, g.standard_normal((r, N)).astype(np.float32))
        X = D(hip, g.standard_normal((r, K)).astype(np.float32))
        G1, G2 = hip.empty(N * K + N), hip.empty(N * K + N)
        layers.append((dY, X, K, G1, r, N, K))
        singles.append((dY, X, K, G2, r, N, K))
    slabs = hip.empty(max(hip.dense_wgrad_multi_scratch(shapes), 4))
    hip.dense_wgrad_multi(layers, slabs)
    for (dY, X, ldx, G2, r, N, K) in singles:
        hip.dense_wgrad(dY, X, ldx, G2, r, N, K, hip.empty(max(hip.dense_wgrad_scratch(r, N, K), 4)))
    for a, b in zip(layers, singles):
        assert torch.equal(a[3], b[3])


# ------------------------------------------------------------------------------------------------ six vs nine cross products (a0_x9_products)
@pytest.mark.parametrize("R,N,K", [(256, 512, 3136), (513, 132, 96), (8192, 512, 3136), (32768, 512, 3136)])
def test_six_and_nine_product_forms_agree_to_fp32_rounding(hip, R, N, K):
    """The split-operand GEMMs form six (default) or nine (strict, A0_X9_PRODUCTS=9) of the cross products of the three-term splits.  The three that the default leaves
    out are each below 2^-24 of the product they belong to: both forms must meet the same bound against fp64 (2e-6 of sum |a||b|, this file's tolerance for either
    pipe) and agree with each other to a tenth of it; the switch must really select different kernels (the results differ somewhere) and restore."""
    g = recipe.gen(R + N + K)
    X = np.maximum(g.standard_normal((R, K)), 0).astype(np.float32)          # post-ReLU activations, as fc1 sees them
    W = (g.standard_normal((N, K)) * 0.03).astype(np.float32)
    b = np.zeros(N, np.float32)
    Xd, Wd, bd = D(hip, X), D(hip, W), D(hip, b)
    rows = slice(0, min(R, 2048))                                           # fp64 reference on the first rows (the whole product on the device for the cross-check)
    want = X[rows].astype(np.float64) @ W.astype(np.float64).T
    scale = np.abs(X[rows]).astype(np.float64) @ np.abs(W).astype(np.float64).T
    prev = hip.x9_products()
    assert prev in (6, 9)
    outs = {}
    try:
        for n in (9, 6):
            assert hip.x9_products(n) in (6, 9) and hip.x9_products() == n
            Y = hip.empty(R * N)
            need = hip.dense_fwd_scratch(R, N, K)
            hip.dense_fwd(Xd, K, Wd, bd, Y, R, N, K, 0, hip.empty(max(need, 1)))
            outs[n] = Y.view(R, N).clone()
            _scale_close(outs[n][rows].cpu().numpy(), want, scale, f"{n}-product dense_fwd {R}x{N}x{K}")
    finally:
        hip.x9_products(prev)
    assert hip.x9_products() == prev
    full_scale = (Xd.abs().double() @ Wd.abs().double().T).clamp_min(1e-30)
    rel = ((outs[6].double() - outs[9].double()).abs() / full_scale).max().item()
    assert rel < 2e-7, f"six- vs nine-product results differ by {rel:.3e} of the accumulated magnitude"
    assert not torch.equal(outs[6], outs[9]), "the switch selects a different kernel"


def test_six_and_nine_product_encoders_agree_to_fp32_rounding(hip):
    """The fused encoder (conv2 / conv3 forward) and its data gradient under a0_x9_products 6 and 9: features within 2e-6 relative to the largest feature, gradients
    within 2e-6 of the largest gradient; conv1 (bytes x three weight terms: always all three products) is bit-identical."""
    from agent0_amd.deepq.engine import DeviceNet, Workspace
    from agent0_amd.deepq.layout import NetLayout
    spec = recipe.NetSpec("dqn", 4)
    L = NetLayout.from_spec(spec)
    net = DeviceNet(hip, L, hip.net(4, 84, 84))
    net.load_state_dict(recipe.make_state_dict(spec, 11))
    B = 300
    frames = D(hip, recipe.make_frames(B, 5, (4, 84, 84)).reshape(-1))
    prev = hip.x9_products()
    got = {}
    try:
        for n in (9, 6):
            hip.x9_products(n)
            ws = Workspace(hip, L, B, 1)
            net.encode(ws, frames, None, 2 * 4 * 84 * 84, 0, B)
            got[n] = (ws.act1.clone(), ws.act2.clone(), ws.act3.clone())
    finally:
        hip.x9_products(prev)
    assert torch.equal(got[6][0], got[9][0]), "conv1 does not depend on the mode"
    for k, name in ((1, "conv2"), (2, "conv3")):
        a, b = got[6][k], got[9][k]
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), name
        assert not torch.equal(a, b), f"{name}: the switch selects a different kernel"


@pytest.mark.parametrize("B,n", [(512, 64), (512, 32), (64, 64), (128, 32)])
def test_data_gradient_with_the_embedding_product_in_its_epilogue(hip, B, n):
    """a0_dense_dgrad_hadamard (round 6; reference model.py:244-247 backwards: x = relu(cosine_emb) * features): dx = dh W_fc1 consumed in the GEMM's epilogue.  demb must be
    bit-identical to a0_dense_dgrad + a0_hadamard_bwd (the same accumulators times the same features); d3 sums a sample's n rows in the tile's register order instead of row by
    row: equal to 2e-6 of sum_n |dx| |emb|, and within this file's fp64 bound."""
    R, N, K = B * n, 512, 3136
    assert hip.dense_dgrad_hadamard_ok(R, N, K, n)
    g = recipe.gen(B + n)
    dh = (g.standard_normal((R, N)) * 1e-3).astype(np.float32)
    W = (g.standard_normal((N, K)) * 0.03).astype(np.float32)
    emb = np.maximum(g.standard_normal((R, K)), 0).astype(np.float32)
    feat = np.maximum(g.standard_normal((B, K)), 0).astype(np.float32)
    dhd, Wd, embd, featd = D(hip, dh), D(hip, W), D(hip, emb), D(hip, feat)
    dx, demb0, d30 = hip.empty(R * K), hip.empty(R * K), hip.empty(B * K)
    hip.dense_dgrad(dhd, Wd, None, dx, R, N, K)
    hip.hadamard_bwd(dx, embd, featd, demb0, d30, B, n, K)
    demb1, d31 = hip.empty(R * K).fill_(float("nan")), hip.empty(B * K).fill_(float("nan"))
    hip.dense_dgrad_hadamard(dhd, Wd, embd, featd, demb1, d31, R, N, K, n)
    assert torch.equal(demb0, demb1), "demb"
    scale = (dx.view(B, n, K).abs().double() * embd.view(B, n, K).abs().double()).sum(1).clamp_min(1e-30)
    err = ((d30.view(B, K).double() - d31.view(B, K).double()).abs() / scale).max().item()
    assert err < 2e-6, f"d3 differs by {err:.3e} of the accumulated magnitude"
    assert torch.equal(d30 == 0, d31 == 0) or float(((d30 == 0) != (d31 == 0)).float().mean()) < 1e-5, "the feature mask"
    rows = slice(0, 4 * n)                          # fp64 reference on the first four samples
    dx64 = dh[rows].astype(np.float64) @ W.astype(np.float64)
    want = np.where(feat[:4] > 0, (dx64 * emb[rows]).reshape(4, n, K).sum(1), 0.0)
    sc = (np.abs(dh[rows]).astype(np.float64) @ np.abs(W).astype(np.float64) * np.abs(emb[rows])).reshape(4, n, K).sum(1)
    _scale_close(d31.view(B, K)[:4].cpu().numpy(), want, sc, "d3 against fp64")
