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


@pytest.mark.parametrize("A", [4, 9])
def test_synth_env_chase_task_needs_a_sequence_of_moves(A):
    """``task="chase"`` (round 5: the learnable task WITH temporal credit): the action moves the block on a 4 x 4 lattice (a % 4 = up, down, left, right, clamped at the
    walls); +1 only on arrival at cell 15, then a respawn at Manhattan distance >= 3.  Checked from the pixels alone: exactly one 8 x 8 block of 255 per frame, on the
    lattice, no other pixel at 255; the dynamics follow a plain-Python model of the rules; the shortest-path policy earns one reward per ~4 steps, a random one a tenth
    of that, a constant action nothing; terminals / life losses are those of the stream task."""
    E = 6
    env = core.SynthVecEnv(E, seed=42, rank=0, action_dim=A, task="chase")
    ref = core.SynthVecEnv(E, seed=42, rank=0, action_dim=A)
    obs, _ = env.reset()
    ref.reset()

    def cells_from_pixels(o):
        out = []
        for e in range(E):
            fr = o[e, 3]
            ys, xs = np.nonzero(fr == 255)
            assert len(ys) == 64 and ys.max() - ys.min() == 7 and xs.max() - xs.min() == 7, "exactly one 8 x 8 block of 255"
            y, x = int(ys.min()), int(xs.min())
            assert (y - 4) % 22 == 0 and (x - 4) % 22 == 0
            out.append(4 * ((y - 4) // 22) + (x - 4) // 22)
        return np.array(out)

    cells = cells_from_pixels(obs)
    assert np.array_equal(cells, (7 * np.arange(E) + 3) % 15) and np.array_equal(cells, core.env_chase_cells(obs))
    g = recipe.gen(3) if False else np.random.default_rng(3)
    total = {"greedy": 0.0, "random": 0.0}
    n_arrivals = 0
    for phase, steps in (("greedy", 240), ("random", 240)):
        for t in range(steps):
            if phase == "greedy":
                a = np.where((cells >> 2) < 3, 1, 3) + 4 * (np.arange(E) % 2) * (A > 4)        # down until the last row, then right; a + 4 is the same move when A > 4
            else:
                a = g.integers(0, A, E)
            obs, r, term, trunc, info = env.step(a.astype(np.int32))
            _, _, term2, _, info2 = ref.step(np.zeros(E, np.int32))
            assert np.array_equal(term, term2) and np.array_equal(info["life_loss"], info2["life_loss"])
            new = cells_from_pixels(obs)
            for e in range(E):
                cy, cx, m = cells[e] >> 2, cells[e] & 3, a[e] % 4
                cy, cx = (max(cy - 1, 0), cx) if m == 0 else (min(cy + 1, 3), cx) if m == 1 else (cy, max(cx - 1, 0)) if m == 2 else (cy, min(cx + 1, 3))
                if 4 * cy + cx == 15:
                    assert r[e] == 1.0 and (new[e] >> 2) + (new[e] & 3) <= 3, "arrival: reward and a respawn at distance >= 3"
                    n_arrivals += 1
                else:
                    assert r[e] == 0.0 and new[e] == 4 * cy + cx
                if term[e]:
                    assert (obs[e] == obs[e, 3]).all(), "a terminal resets the frame stack to the new frame"
            cells = new
            total[phase] += float(r.sum())
    assert n_arrivals > 300
    assert 0.22 <= total["greedy"] / (240 * E) <= 0.28 and total["random"] / (240 * E) < 0.06
