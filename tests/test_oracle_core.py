"""CPU: the C part of the oracle (sum-tree, Philox, permutation, synthetic env) checked against
independent numpy mirrors, published known-answer vectors and size-independent properties."""
import numpy as np
import pytest

import recipe
from oracle import core


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kats = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in kats:
        assert tuple(int(x) for x in core.philox(ctr, key)) == want
        assert tuple(int(x) for x in core.philox_numpy(ctr, key)) == want


def test_rng_streams():
    u = core.rng_uniform(42, 7, 0, 1000)
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.05
    # offset addressing: element i of the stream does not depend on where a fill starts
    assert np.array_equal(core.rng_uniform(42, 7, 13, 100), u[13:113])
    w = core.rng_u32(42, 7, 0, 8)
    blk0 = core.philox((0, 0, 7, 0), (42, 0))
    assert np.array_equal(w[:4], blk0)
    z = core.rng_normal(1, 2, 0, 0.1, 20000)
    assert abs(z.mean()) < 0.005 and abs(z.std() - 0.1) < 0.005
    assert np.array_equal(core.rng_normal(1, 2, 6, 0.1, 10), z[6:16])


@pytest.mark.parametrize("size", [1, 5, 24, 64, 1000])
def test_sumtree_c_matches_numpy_mirror(size):
    g = recipe.gen(size)
    t = core.SumTree(size)
    mirror = np.zeros_like(t.tree)
    for rnd in range(6):
        n = int(g.integers(1, min(size, 40) + 1))
        idx = g.integers(0, size, n)
        val = g.uniform(0.0, 3.0, n).astype(np.float32)
        if rnd == 2:
            val[: n // 2] = 0.0
        t.set(idx, val)
        core.sumtree_set_numpy(mirror, t.cap2, idx, val)
        assert np.array_equal(t.tree, mirror)
        # invariant: every internal node is exactly left + right
        p = np.arange(1, t.cap2)
        assert np.array_equal(t.tree[p], t.tree[2 * p] + t.tree[2 * p + 1])
        xi = g.random(32).astype(np.float32)
        idx_s, p_s = t.sample(xi)
        seg = np.float32(t.total) / np.float32(32)
        for k in range(32):
            u = np.float32(np.float32(k) + xi[k]) * seg
            assert core.sumtree_find_numpy(t.tree, t.cap2, u) == idx_s[k]
        assert np.all(idx_s < size) and np.all(p_s > 0)
        assert np.array_equal(p_s, t.leaves()[idx_s])
    full = t.tree.copy()
    t.rebuild()
    assert np.array_equal(full, t.tree)


def test_sumtree_sampling_is_proportional():
    size = 256
    t = core.SumTree(size)
    g = recipe.gen(3)
    pri = g.uniform(0.1, 2.0, size).astype(np.float32)
    pri[17] = 0.0
    t.set(np.arange(size), pri)
    counts = np.zeros(size)
    for _ in range(400):
        idx, _ = t.sample(g.random(64).astype(np.float32))
        np.add.at(counts, idx, 1)
    assert counts[17] == 0
    freq = counts / counts.sum()
    want = pri / pri.sum()
    assert np.abs(freq - want).max() < 0.004


def test_sumtree_edge_of_range():
    t = core.SumTree(8)
    t.set(np.arange(8), np.array([1, 0, 0, 2, 0, 0, 0, 0], dtype=np.float32))
    assert t.find(0.0) == 0 and t.find(0.999) == 0 and t.find(1.0) == 3 and t.find(2.9999) == 3
    assert t.find(3.0) == 3 and t.find(100.0) == 3  # u >= total never lands on a zero-priority leaf
    z = core.SumTree(8)
    assert z.find(0.0) == 0  # all-zero tree: defined, leftmost


@pytest.mark.parametrize("n", [1, 2, 7, 512, 1000, 100001])
def test_perm_is_a_permutation(n):
    m = min(n, 5000)
    p = core.perm_batch(0, n, n, 1234) if n <= 5000 else None
    if p is not None:
        assert np.array_equal(np.sort(p), np.arange(n))
        assert not np.array_equal(core.perm_batch(0, n, n, 99), p) or n <= 2
    else:
        p = core.perm_batch(0, n, n, 1234)
        assert np.array_equal(np.sort(p), np.arange(n))
    assert np.array_equal(core.perm_batch(3, m - 3, n, 1234) if m > 3 else p[3:m], p[3:m])


def test_synth_env_contract():
    env = core.SynthVecEnv(6, seed=42, rank=0)
    obs, info = env.reset()
    assert obs.shape == (6, 4, 84, 84) and obs.dtype == np.uint8 and info == {}
    assert np.array_equal(obs[:, 0], obs[:, 3])
    fr = core.env_frame(42, 2, 0)
    assert np.array_equal(obs[2, 0], fr)
    assert fr.max() == 255 and 0.15 < (fr > 0).mean() < 0.35
    prev = obs
    n_term = 0
    rets = np.zeros(6)
    for t in range(1, 400):
        obs, r, term, trunc, info = env.step(np.zeros(6, dtype=np.int64))
        assert r.dtype == np.float64 and set(np.unique(r)) <= {-1.0, 0.0, 1.0}
        assert not trunc.any() and "life_loss" in info
        assert not (info["life_loss"] & term).any()
        rets += r
        for e in range(6):
            assert np.array_equal(obs[e, 3], core.env_frame(42, e, t))
            if term[e]:
                n_term += 1
                assert np.array_equal(obs[e, 0], obs[e, 3])
                assert info["_final_info"][e] and info["final_info"][e]["episode"]["r"][0] == np.float32(rets[e])
                rets[e] = 0
            else:
                assert np.array_equal(obs[e, :3], prev[e, 1:])
        prev = obs
    # same seed -> same trajectory; other rank -> different draws
    env2 = core.SynthVecEnv(6, seed=42, rank=1)
    env2.reset()
    env3 = core.SynthVecEnv(6, seed=42, rank=0)
    env3.reset()
    a = [env3.step(np.zeros(6))[1] for _ in range(50)]
    b = [env2.step(np.zeros(6))[1] for _ in range(50)]
    env4 = core.SynthVecEnv(6, seed=42, rank=0)
    env4.reset()
    c = [env4.step(np.zeros(6))[1] for _ in range(50)]
    assert np.array_equal(np.stack(a), np.stack(c)) and not np.array_equal(np.stack(a), np.stack(b))


@pytest.mark.parametrize("A", [4, 9, 18])
def test_synth_env_block_task_is_readable_from_the_pixels(A):
    """``task="block"`` (the learnable reward task): the rewarded action is the quadrant (mod A) of the bright 8x8 block in the NEWEST frame of the observation
    the action is chosen on — located here from the pixels alone — the next class costs -1, everything else 0; terminals are those of the stream task."""
    E = 6
    env = core.SynthVecEnv(E, seed=42, rank=0, action_dim=A, task="block")
    ref = core.SynthVecEnv(E, seed=42, rank=0, action_dim=A)
    obs, _ = env.reset()
    ref.reset()
    terms = core.env_terminals(42, 0, E, 80)
    for t in range(1, 81):
        tgt = []
        for e in range(E):
            fr = obs[e, 3]
            solid = [(y, x) for y in range(77) for x in range(77) if fr[y, x] == 255 and (fr[y:y + 8, x:x + 8] == 255).all()]
            assert len(solid) >= 1
            y, x = solid[0]
            tgt.append((2 * (y >= 39) + (x >= 39)) % A)
        tgt = np.array(tgt)
        assert np.array_equal(tgt, core.env_block_target(np.arange(E), np.full(E, t - 1), A))
        a = (tgt + np.arange(E) % 3) % A
        obs, r, term, trunc, info = env.step(a)
        want = np.where(np.arange(E) % 3 == 0, 1.0, np.where(np.arange(E) % 3 == 1, -1.0, 0.0))
        if A == 1:
            want[:] = 1.0
        assert np.array_equal(r, want)
        o2, r2, term2, _, info2 = ref.step(a)
        assert np.array_equal(obs, o2) and np.array_equal(term, term2) and np.array_equal(info["life_loss"], info2["life_loss"])
        assert np.array_equal(term, terms[t - 1])
