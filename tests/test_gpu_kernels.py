"""GPU (MI355X): replay / sum-tree / permutation / RNG / synthetic env / actor kernels against the C and numpy oracle.
Integer, byte and index results are compared bit-exactly."""
import numpy as np
import pytest
import torch

import recipe
from oracle import core
from oracle import actor as oactor
from oracle import replay as oreplay
from util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from agent0_amd.ops import HipOps
    return HipOps()


def D(hip, x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(hip.device)


@pytest.mark.parametrize("size", [1, 5, 24, 1000, 100000, 1_000_000])
def test_sumtree_bit_exact(hip, size):
    """Every node of the tree after priority updates, the sampled leaves and their priorities, and a bulk rebuild, bit for bit against
    oracle/sumtree.c — up to BASELINE's full replay size (1 M leaves, 2^21 nodes, 20 levels)."""
    g = recipe.gen(size)
    t = core.SumTree(size)
    tree = hip.zeros(2 * t.cap2)
    for rnd in range(5):
        n = int(g.integers(1, min(size, 512) + 1))
        idx = g.integers(0, size, n)
        val = g.uniform(0.0, 3.0, n).astype(np.float32)
        if rnd == 2:
            val[: n // 2] = 0
        t.set(idx, val)
        hip.sumtree_set(tree, t.cap2, D(hip, idx.astype(np.int64)), D(hip, val), n)
        assert np.array_equal(tree.cpu().numpy(), t.tree), "tree bytes after set"
        B = 64 if size < 100000 else 512
        xi = g.random(B).astype(np.float32)
        want_i, want_p = t.sample(xi)
        out_i, out_p = hip.zeros(B, dtype=torch.int64), hip.zeros(B)
        hip.sumtree_sample(tree, t.cap2, D(hip, xi), B, out_i, out_p)
        assert np.array_equal(out_i.cpu().numpy(), want_i) and np.array_equal(out_p.cpu().numpy(), want_p)
    # bulk fill + rebuild
    leaves = g.uniform(0.01, 1.0, size).astype(np.float32)
    t.tree[t.cap2:t.cap2 + size] = leaves
    t.rebuild()
    tree[t.cap2:t.cap2 + size] = D(hip, leaves)
    hip.sumtree_rebuild(tree, t.cap2)
    assert np.array_equal(tree.cpu().numpy(), t.tree)


def test_sumtree_full_size_properties(hip):
    """1 M leaves (BASELINE config 2): internal nodes are exactly left+right; sampled leaves are live; sampling is sorted by stratum."""
    size = 1_000_000
    cap2 = 1 << 20
    g = recipe.gen(1)
    tree = hip.zeros(2 * cap2)
    tree[cap2:cap2 + size] = D(hip, np.sqrt(g.uniform(0.01, 1.0, size)).astype(np.float32))
    hip.sumtree_rebuild(tree, cap2)
    idx = g.integers(0, size, 512)
    val = g.uniform(0, 2, 512).astype(np.float32)
    hip.sumtree_set(tree, cap2, D(hip, idx.astype(np.int64)), D(hip, val), 512)
    tc = tree.cpu().numpy()
    p = np.arange(1, cap2)
    assert np.array_equal(tc[p], tc[2 * p] + tc[2 * p + 1])
    out_i, out_p = hip.zeros(512, dtype=torch.int64), hip.zeros(512)
    hip.sumtree_sample(tree, cap2, D(hip, g.random(512).astype(np.float32)), 512, out_i, out_p)
    oi = out_i.cpu().numpy()
    assert oi.min() >= 0 and oi.max() < size and np.all(np.diff(oi) >= 0) and np.all(out_p.cpu().numpy() > 0)
    assert np.array_equal(out_p.cpu().numpy(), tc[cap2 + oi])


@pytest.mark.parametrize("n", [1, 7, 512, 100001, 1_000_000])
def test_perm_bit_exact(hip, n):
    count = min(n, 4096)
    start = 0 if n <= 4096 else n - count - 3
    out = hip.zeros(count, dtype=torch.int64)
    hip.perm_batch(start, count, n, 1234, out)
    assert np.array_equal(out.cpu().numpy(), core.perm_batch(start, count, n, 1234))


def test_rng_streams(hip):
    n = 10007
    u32 = hip.zeros(n, dtype=torch.int32)
    hip.rng_u32(42, 7, 5, u32, n)
    assert np.array_equal(u32.cpu().numpy().view(np.uint32), core.rng_u32(42, 7, 5, n))
    u = hip.zeros(n)
    hip.rng_uniform((1 << 40) + 3, 9, 1 << 33, u, n)
    assert np.array_equal(u.cpu().numpy(), core.rng_uniform((1 << 40) + 3, 9, 1 << 33, n))
    z = hip.zeros(n)
    hip.rng_normal(3, 1, 11, 0.1, z, n)
    assert_close(z, core.rng_normal(3, 1, 11, 0.1, n), 1e-4, 1e-6, "Box-Muller normals (transcendental rounding only)")
    ri = hip.zeros(n, dtype=torch.int32)
    hip.rng_randint(8, 2, 0, 18, ri, n)
    assert np.array_equal(ri.cpu().numpy(), (core.rng_u32(8, 2, 0, n) % 18).astype(np.int32))


@pytest.mark.parametrize("task,A", [("stream", 4), ("block", 4), ("block", 9), ("block", 18), ("chase", 4), ("chase", 9)])
def test_synth_env_bytes(hip, task, A):
    """a0_env_synth_step against oracle/synth_env.c, all three reward tasks; on the block task the actions cycle through right / wrong / neutral classes, on the chase task
    two thirds of the moves follow the shortest path (so that arrivals, respawns and wall clamps all occur)."""
    E = 5
    env = core.SynthVecEnv(E, seed=42, rank=3, action_dim=A, task=task)
    tid = core.SynthVecEnv.TASKS[task]
    act = hip.zeros(E, dtype=torch.int32)
    n_pos = n_neg = 0
    obs_c, _ = env.reset()
    obs = [hip.zeros(E * 4 * 84 * 84, dtype=torch.uint8), hip.zeros(E * 4 * 84 * 84, dtype=torch.uint8)]
    ep = hip.zeros(E)
    f = [hip.zeros(E) for _ in range(6)]
    hip.env_reset(42, 3, E, obs[0], ep, task=tid)
    assert np.array_equal(obs[0].cpu().numpy().reshape(E, 4, 84, 84), obs_c)
    n_term = 0
    o = obs_c
    for t in range(1, 1200):
        if task == "chase":
            c = core.env_chase_cells(o)
            a = np.where((np.arange(E) + t) % 3 == 0, (np.arange(E) * 7 + t) % A, np.where((c >> 2) < 3, 1, 3) + 4 * ((t % 2) if A > 4 else 0))
        else:
            a = (core.env_block_target(np.arange(E), np.full(E, t - 1), A) + (np.arange(E) + t) % 3) % A        # target, target + 1 (wrong), target + 2
        act.copy_(torch.from_numpy(a.astype(np.int32)))
        o, r, term, trunc, info = env.step(a)
        n_pos += int((r > 0).sum()); n_neg += int((r < 0).sum())
        hip.env_step(42, 3, E, t, obs[(t - 1) % 2], obs[t % 2], ep, *f, action=act, A=A, task=tid)
        if t < 40 or term.any() or (task == "chase" and (t % 7 == 0 or (r > 0).any())):
            assert np.array_equal(obs[t % 2].cpu().numpy().reshape(E, 4, 84, 84), o), f"obs at step {t}"
        assert np.array_equal(f[0].cpu().numpy(), r.astype(np.float32))
        assert np.array_equal(f[1].cpu().numpy() != 0, term) and not f[2].cpu().numpy().any()
        assert np.array_equal(f[3].cpu().numpy() != 0, info["life_loss"])
        if term.any():
            n_term += int(term.sum())
            fr = np.array([info["final_info"][i]["episode"]["r"][0] if term[i] else 0.0 for i in range(E)], dtype=np.float32)
            assert np.array_equal(f[5].cpu().numpy(), fr) and np.array_equal(f[4].cpu().numpy() != 0, term)
    assert n_term > 0
    assert np.array_equal(ep.cpu().numpy(), env.ep_ret)
    if task == "block":
        assert n_pos > 1500 and n_neg > 1500        # a third of the actions each (A = 4 ... 18: target + 2 is never rewarded)
    if task == "chase":
        assert n_pos > 400 and n_neg == 0           # arrivals at the target cell


@pytest.mark.parametrize("n_step", [1, 3])
def test_actor_nstep_and_egreedy(hip, n_step):
    E, T, A = 6, 25, 4
    g = recipe.gen(n_step)
    from collections import deque
    tracker = deque(maxlen=n_step)
    ring_a, ring_r, ring_d = hip.zeros(n_step * E, dtype=torch.int32), hip.zeros(n_step * E), hip.zeros(n_step * E)
    oa, orr, od = hip.zeros(E, dtype=torch.int32), hip.zeros(E), hip.zeros(E)
    for t in range(T):
        action = g.integers(0, A, E)
        reward = g.choice([-1.0, 0.0, 1.0], E)
        term, trunc, life = g.random(E) < 0.2, g.random(E) < 0.1, g.random(E) < 0.2
        done = np.logical_and(np.logical_or(term, life), np.logical_not(trunc))
        tracker.append((None, action, reward, done))
        R, Dn = oactor.nstep_scan(tracker, 0.99)
        hip.actor_nstep(E, n_step, t, 0.99, D(hip, action.astype(np.int32)), D(hip, reward.astype(np.float32)), D(hip, term.astype(np.float32)),
                        D(hip, trunc.astype(np.float32)), D(hip, life.astype(np.float32)), ring_a, ring_r, ring_d, oa, orr, od)
        assert np.array_equal(oa.cpu().numpy(), tracker[0][1].astype(np.int32))
        assert np.array_equal(orr.cpu().numpy(), R.astype(np.float32)), "fp64 n-step sum rounded once to fp32"
        assert np.array_equal(od.cpu().numpy() != 0, Dn)
    greedy, rnd, u = g.integers(0, A, E), g.integers(0, A, E), g.random(E)
    qmax = g.standard_normal(E).astype(np.float32)
    act, qs = hip.zeros(E, dtype=torch.int32), hip.zeros(1)
    hip.actor_egreedy(D(hip, greedy.astype(np.int32)), D(hip, rnd.astype(np.int32)), D(hip, u.astype(np.float32)), 0.3, E, act, D(hip, qmax), qs)
    want, _ = oactor.egreedy(np.eye(A)[greedy], rnd, u.astype(np.float32), np.float32(0.3))
    assert np.array_equal(act.cpu().numpy(), want.astype(np.int32))
    assert_close(qs, [qmax.mean()], 1e-6, 1e-7, "mean max-Q")


def test_replay_insert_lookup_gather_priorities(hip):
    cap, E, ob = 24, 4, 4 * 84 * 84
    frames = hip.zeros(cap * 2 * ob, dtype=torch.uint8)
    r_act, r_rew, r_done = hip.zeros(cap, dtype=torch.int32), hip.zeros(cap), hip.zeros(cap)
    ref = oreplay.ReferenceReplay(cap, True, total_steps=1000)
    prio = hip.zeros(cap); hip.fill_f32(prio, cap, 1.0)
    pstate = hip.zeros(1); hip.fill_f32(pstate, 1, 1.0)
    g = recipe.gen(8)
    written, uid = 0, 0
    store = {}
    for it in range(9):
        obs = g.integers(0, 256, (E, ob), dtype=np.uint8)
        nxt = g.integers(0, 256, (E, ob), dtype=np.uint8)
        a = g.integers(0, 4, E).astype(np.int32); r = g.choice([-1.0, 0.0, 1.0], E).astype(np.float32); d = (g.random(E) < 0.3).astype(np.float32)
        hip.replay_insert(frames, cap, ob, written % cap, E, D(hip, obs), D(hip, nxt), D(hip, a), D(hip, r), D(hip, d), r_act, r_rew, r_done)
        trans = []
        for i in range(E):
            store[uid] = (np.concatenate((obs[i], nxt[i])), a[i], r[i], d[i])
            trans.append((uid, a[i], r[i], d[i])); uid += 1
        ref.extend(trans)
        hip.priority_tail(prio, cap, E, pstate, 0.5)
        written += E
        top = min(written, cap); head = written % cap if written > cap else 0
        assert head == ref.head and top == ref.top
        B = 6
        idx = g.integers(0, 1000, B)
        slot, act, rew, done, pr, io = hip.zeros(B, dtype=torch.int32), hip.zeros(B, dtype=torch.int32), hip.zeros(B), hip.zeros(B), hip.zeros(B), hip.zeros(B, dtype=torch.int64)
        hip.replay_lookup(D(hip, idx.astype(np.int64)), B, top, head, cap, slot, r_act, r_rew, r_done, prio, act, rew, done, pr, io)
        out = hip.zeros(B * 2 * ob, dtype=torch.uint8)
        hip.replay_gather(frames, 2 * ob, slot, B, out, cap)
        got_rows = out.cpu().numpy().reshape(B, -1)
        for b in range(B):
            u, at, rt, dt, p, i2 = ref[int(idx[b])]
            assert np.array_equal(got_rows[b], store[u][0]) and int(act[b]) == at and float(rew[b]) == rt and float(done[b]) == dt
            assert int(io[b]) == i2 and abs(float(pr[b]) - float(p)) <= 1.2e-7 * float(p)
        if it % 2 == 1:
            ids = g.integers(0, top, 5); ids[1] = ids[0]
            losses = g.uniform(0, 3, 5).astype(np.float32)
            # oracle: torch indexed assignment with duplicates keeps the LAST write on CPU
            ref.update_priority(ids, losses)
            hip.priority_update(prio, D(hip, ids.astype(np.int64)), D(hip, losses), 5, 0.01, 0.5, pstate, None)
        # fp: the device computes the IEEE-correct sqrt; torch's CPU pow(x, 0.5) is a vectorised (Sleef) sqrt that is
        # occasionally 1 ulp off, so priorities are compared to 1 ulp and untouched slots exactly
        assert_close(prio, ref.priority, 1.2e-7, 0, f"priority vector after iteration {it}")
        assert np.array_equal(prio.cpu().numpy() == 1.0, ref.priority == 1.0)
        assert float(pstate[0]) == np.float32(ref.max_p)
        sc, ps, w = hip.zeros(256), hip.zeros(1), hip.zeros(B)
        hip.sum_f32(prio, cap, sc, ps)
        hip.is_weights(pr, B, ps, top, 0.5, w)
        want = oreplay.is_weights(pr.cpu().numpy(), float(torch.from_numpy(ref.priority).sum()), top, 0.5)
        assert_close(w, want, 2e-6, 1e-7, "importance weights")


@pytest.mark.parametrize("prioritize,sumtree", [(False, False), (True, False), (True, True)])
def test_fused_sample_gather_equals_separate_kernels(prioritize, sumtree):
    """a0_replay_sample_gather == (a0_perm_batch | a0_sumtree_sample) + a0_replay_lookup + a0_replay_gather, bit for bit."""
    from agent0_amd.deepq.config import parse_overrides
    from agent0_amd.deepq.replay import ReplayDataset, TransitionBlock
    over = ["replay.size=300", "learner.batch_size=64", "wandb=false", "tb=false"]
    if prioritize:
        over += ["replay.policy=prioritize", f"replay.sumtree={str(sumtree).lower()}"]
    reps = []
    for _ in range(2):
        cfg = parse_overrides(over)
        cfg.obs_shape, cfg.action_dim = (4, 84, 84), 4
        rp = ReplayDataset(cfg)
        g = recipe.gen(3)
        n = 420                        # wraps the ring: head != 0
        st = {"obs": torch.from_numpy(g.integers(0, 256, (n, rp.obs_bytes), dtype=np.uint8)).cuda(), "obs_next": torch.from_numpy(g.integers(0, 256, (n, rp.obs_bytes), dtype=np.uint8)).cuda(),
              "act": torch.from_numpy(g.integers(0, 4, n).astype(np.int32)).cuda(), "rew": torch.from_numpy(g.standard_normal(n).astype(np.float32)).cuda(),
              "done": torch.from_numpy((g.random(n) < 0.2).astype(np.float32)).cuda()}
        rp.extend(TransitionBlock(n, staged=st))
        if prioritize:
            rp.update_priority(torch.from_numpy(g.integers(0, 300, 64)).cuda(), torch.from_numpy(g.uniform(0, 3, 64).astype(np.float32)).cuda())
        reps.append(rp)
    a, b = reps
    rows = torch.empty(64 * a.row_bytes, dtype=torch.uint8, device="cuda")
    for it in range(6):
        ba = a.sample_gathered(rows)
        bb = b.sample()
        ref = torch.empty_like(rows)
        b.ops.replay_gather(b.frames, b.row_bytes, bb.slot, 64, ref, b.size)
        assert torch.equal(ba.slot, bb.slot) and torch.equal(ba.idx, bb.idx) and torch.equal(rows, ref)
        assert torch.equal(ba.act, bb.act) and torch.equal(ba.rew, bb.rew) and torch.equal(ba.done, bb.done) and torch.equal(ba.weights, bb.weights)


def test_replay_ring_full_size_round_trip(hip):
    """BASELINE configs[1] size — 1 M rows of 56 448 B = 56.4 GB, byte offsets far beyond 2^32 — checked through a size-independent
    property: every row carries the serial number of its transition in its first and last 8 bytes of st and of st_next; after the ring
    has wrapped, logical index i must return transition (written - size + i), through a0_replay_lookup + a0_replay_gather and through
    the fused a0_replay_sample_gather (uniform permutation), with act / rew / done following the same slot."""
    cap, ob, n = 1_000_000, 4 * 84 * 84, 8192
    frames = torch.empty(cap * 2 * ob, dtype=torch.uint8, device=hip.device)
    r_act, r_rew, r_done = hip.zeros(cap, dtype=torch.int32), hip.zeros(cap), hip.zeros(cap)
    obs = torch.zeros(n, ob, dtype=torch.uint8, device=hip.device)
    nxt = torch.zeros(n, ob, dtype=torch.uint8, device=hip.device)
    shifts = torch.arange(8, device=hip.device, dtype=torch.int64) * 8

    def tag(t):          # int64 [k] -> uint8 [k][8], little endian
        return ((t[:, None] >> shifts[None, :]) & 255).to(torch.uint8)

    written = 0
    total = cap + 5 * n + 123           # wraps, and ends off a block boundary
    while written < total:
        k = min(n, total - written)
        t = torch.arange(written, written + k, device=hip.device, dtype=torch.int64)
        obs[:k, :8] = tag(t); obs[:k, -8:] = tag(t ^ 0x5555)
        nxt[:k, :8] = tag(t + (1 << 40)); nxt[:k, -8:] = tag(t ^ 0x3333)
        c = written % cap
        k1 = min(k, cap - c)            # a0_replay_insert takes a contiguous slot range: split at the wrap like ReplayDataset.extend
        for o, kk in ((0, k1), (k1, k - k1)):
            if kk:
                hip.replay_insert(frames, cap, ob, (written + o) % cap, kk, obs[o:o + kk].reshape(-1), nxt[o:o + kk].reshape(-1),
                                  (t[o:o + kk] % 18).to(torch.int32), (t[o:o + kk] % 3 - 1).float(), (t[o:o + kk] % 2).float(), r_act, r_rew, r_done)
        written += k
    top, head = cap, written % cap
    B = 4096
    g = recipe.gen(77)
    idx_np = np.concatenate(([0, 1, cap - 1, cap - head - 1, cap - head, cap - head + 1], g.integers(0, cap, B - 6))).astype(np.int64)
    idx = D(hip, idx_np)
    want_t = idx + (written - cap)

    def check_rows(rows, act, rew, done, t):
        rows = rows.view(t.numel(), 2 * ob)
        def untag(b):
            return (b.to(torch.int64) << shifts[None, :]).sum(1)
        assert torch.equal(untag(rows[:, :8]), t) and torch.equal(untag(rows[:, ob - 8:ob]), t ^ 0x5555)
        assert torch.equal(untag(rows[:, ob:ob + 8]), t + (1 << 40)) and torch.equal(untag(rows[:, 2 * ob - 8:]), t ^ 0x3333)
        assert torch.equal(act.long(), t % 18) and torch.equal(rew, (t % 3 - 1).float()) and torch.equal(done, (t % 2).float())

    slot, act, rew, done, io = hip.zeros(B, dtype=torch.int32), hip.zeros(B, dtype=torch.int32), hip.zeros(B), hip.zeros(B), hip.zeros(B, dtype=torch.int64)
    hip.replay_lookup(idx, B, top, head, cap, slot, r_act, r_rew, r_done, None, act, rew, done, None, io)
    assert torch.equal(slot.long(), (head + idx) % cap) and torch.equal(io, idx)
    out = torch.empty(B * 2 * ob, dtype=torch.uint8, device=hip.device)
    hip.replay_gather(frames, 2 * ob, slot, B, out, cap)
    check_rows(out, act, rew, done, want_t)
    # fused sampler: one epoch position of the Feistel permutation of range(top); indices must be a set of distinct valid positions
    hip.replay_sample_gather(0, 3 * B, top, 0xC0FFEE, None, 1, None, top, head, cap, frames, 2 * ob, r_act, r_rew, r_done, None, B, out, io, slot, act, rew, done, None)
    assert int(io.min()) >= 0 and int(io.max()) < top and io.unique().numel() == B
    assert torch.equal(slot.long(), (head + io) % cap)
    check_rows(out, act, rew, done, io + (written - cap))


@pytest.mark.parametrize("size,start,n", [(1000, 0, 1000), (1000, 990, 37), (24, 20, 24), (100000, 99000, 20480), (100000, 5, 1), (1 << 20, (1 << 20) - 3, 20480)])
def test_sumtree_set_range_equals_batch_set(hip, size, start, n):
    """a0_sumtree_set_range (one launch per rollout) against the oracle's set on the same (idx, val) pairs, including ring wrap-around."""
    g = recipe.gen(size + start + n)
    t = core.SumTree(size)
    leaves = g.uniform(0.0, 2.0, size).astype(np.float32)
    t.tree[t.cap2:t.cap2 + size] = leaves
    t.rebuild()
    tree = D(hip, t.tree.copy())
    v = np.float32(g.uniform(0.5, 3.0))
    idx = (start + np.arange(n)) % size
    t.set(idx, np.full(n, v, np.float32))
    hip.sumtree_set_range(tree, t.cap2, start, n, size, D(hip, np.array([v], np.float32)))
    assert np.array_equal(tree.cpu().numpy(), t.tree)


def test_noisy_multi_equals_per_module_kernels(hip):
    """a0_noisy_multi (all NoisyLinear modules of a network in one launch) against a0_noisy_compose / a0_noisy_grad_sigma per module:
    bit-identical effective weights and sigma gradients (reference model.py:54-62,78-83)."""
    g = recipe.gen(21)
    mods_spec = [(512, 3136, 0, 512), (256, 512, 0, 204), (256, 512, 204, 255)]      # fc1; q_head and value_head rows of one packed block
    def blk(N, K):
        return D(hip, g.standard_normal(N * K + N).astype(np.float32))
    blocks = {}
    mods_c, mods_g, per = [], [], []
    for (N, K, r0, r1) in mods_spec:
        key = (N, K)
        if key not in blocks:
            blocks[key] = (blk(N, K), blk(N, K), hip.zeros(N * K + N), hip.zeros(N * K + N), hip.zeros(N * K + N), hip.zeros(N * K + N))
        mu, sg, eff_a, eff_b, gs_a, gs_b = blocks[key]
        nin = D(hip, g.standard_normal(K).astype(np.float32)); now = D(hip, g.standard_normal(r1 - r0).astype(np.float32)); nob = D(hip, g.standard_normal(r1 - r0).astype(np.float32))
        mods_c.append((mu, sg, eff_a, N, K, r0, r1, nin, now, nob))
        mods_g.append((mu, None, gs_a, N, K, r0, r1, nin, now, nob))
        per.append((mu, sg, eff_b, gs_b, N, K, r0, r1, nin, now, nob))
    hip.noisy_multi(False, mods_c)
    hip.noisy_multi(True, mods_g)
    for (mu, sg, eff_b, gs_b, N, K, r0, r1, nin, now, nob) in per:
        hip.noisy_compose(mu, sg, eff_b, N, K, r0, r1, nin, now, nob)
        hip.noisy_grad_sigma(mu, gs_b, N, K, r0, r1, nin, now, nob)
    for (mu, sg, eff_a, eff_b, gs_a, gs_b) in blocks.values():
        assert torch.equal(eff_a, eff_b) and torch.equal(gs_a, gs_b)


@pytest.mark.parametrize("n,R", [(2, 512), (3, 512), (2, 300), (3, 1024)])
def test_grouped_fc1_passes_equal_the_separate_gemms(hip, n, R):
    """a0_dense_fwd_partial_multi (round 4): two or three passes of one layer shape — own inputs, weights, slab buffers — as ONE launch with a half / a third of the splits
    each pass would take alone.  Every product is still exact (three bf16 terms per operand, nine cross products, fp32 accumulation); only the association of the
    partial sums changes with the split.  Checked: the slab sums of every pass against an fp64 GEMM (fp32-accumulation accuracy) and against the single-pass
    kernel's (a few ulp of the accumulated magnitude), no cross-talk between the passes, ragged row counts."""
    N, K = 512, 3136
    assert hip.dense_fwd_partial_multi_ok(n, R, N, K)
    g = recipe.gen(n * R)
    Xs = [D(hip, g.standard_normal((R, K)).astype(np.float32)) for _ in range(n)]
    Ws = [D(hip, (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)) for _ in range(n)]
    ns1 = hip.dense_fwd_partial_slabs(R, N, K)
    slabs = [hip.empty(ns1 * R * N).fill_(float("nan")) for _ in range(n)]
    ns = hip.dense_fwd_partial_multi(Xs, K, Ws, R, N, K, slabs)
    assert 1 <= ns < ns1, "fewer, deeper splits than a pass on its own"
    for i in range(n):
        got = slabs[i][: ns * R * N].view(ns, R, N).double().sum(0)
        want = Xs[i].double() @ Ws[i].double().t()
        ref = hip.empty(ns1 * R * N)
        hip.dense_fwd_partial(Xs[i], K, Ws[i], R, N, K, ref)
        single = ref.view(ns1, R, N).double().sum(0)
        err, err1 = float((got - want).abs().max()), float((single - want).abs().max())
        assert err < 2e-5 and err < 4 * err1 + 1e-6, f"pass {i}: {err} against fp64 (single-pass kernel: {err1})"
        assert torch.isfinite(slabs[i][: ns * R * N]).all() and (ns == ns1 or torch.isnan(slabs[i][ns * R * N:]).all())


def test_fc1_data_and_weight_gradient_in_one_launch_equal_the_two_calls(hip):
    """a0_dense_dgrad_wgrad (round 4): fc1's masked data gradient and its unsplit weight gradient of a 512-row batch — two GEMMs of 392 tiles each — as one launch
    (a0_igemm_x9_pair_kernel: blockIdx.y picks the body).  Same bodies on the same tiles: dX and the [W | b] gradient block must be bit-identical to
    a0_dense_dgrad + a0_dense_wgrad."""
    R, N, K = 512, 512, 3136
    assert hip.dense_dgrad_wgrad_ok(R, N, K) and not hip.dense_dgrad_wgrad_ok(256, 512, K) and not hip.dense_dgrad_wgrad_ok(512, 256, K)
    assert hip.dense_dgrad_wgrad2_ok(R, N, K, 32, 512) and not hip.dense_dgrad_wgrad2_ok(R, N, K, 256, 512), "the third problem rides along for scalar heads only"
    g = recipe.gen(5)
    dY = D(hip, g.standard_normal((R, N)).astype(np.float32))
    W = D(hip, (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    X = D(hip, np.maximum(g.standard_normal((R, K)), 0).astype(np.float32))
    dX0, dX1 = hip.empty(R * K), hip.empty(R * K).fill_(float("nan"))
    g0, g1 = hip.empty(N * K + N), hip.empty(N * K + N).fill_(float("nan"))
    hip.dense_dgrad(dY, W, X, dX0, R, N, K)
    hip.dense_wgrad(dY, X, K, g0, R, N, K, hip.empty(max(hip.dense_wgrad_scratch(R, N, K), 4)))
    hip.dense_dgrad_wgrad(dY, W, X, K, dX1, g1, R, N, K)
    assert torch.equal(dX0, dX1) and torch.equal(g0, g1)
    assert float((dX1.view(R, K)[X.view(R, K) <= 0]).abs().max()) == 0.0, "the ReLU mask"


@pytest.mark.parametrize("N2", [32, 256, 800, 52])
def test_fc1_gradients_and_the_head_weight_gradient_in_one_launch(hip, N2):
    """a0_dense_dgrad_wgrad2 (round 5): the pair launch above plus the HEAD's weight gradient as a third problem of the same launch (a0_igemm_x9_trio_kernel).  fc1's two
    gradients stay bit-identical to the pair's; the head's [W | b] gradient is an unsplit sum on the bf16 pipe where a0_dense_wgrad splits it into slabs on the fp32
    pipe — both within fp32 rounding of the fp64 product (head widths of dqn, c51, qr and a ragged one)."""
    R, N, K, K2 = 512, 512, 3136, 512
    g = recipe.gen(6)
    dY = D(hip, g.standard_normal((R, N)).astype(np.float32))
    W = D(hip, (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    X = D(hip, np.maximum(g.standard_normal((R, K)), 0).astype(np.float32))
    dY2n = g.standard_normal((R, N2)).astype(np.float32)
    X2n = np.maximum(g.standard_normal((R, K2)), 0).astype(np.float32)
    dY2, X2 = D(hip, dY2n), D(hip, X2n)
    dX0, dX1 = hip.empty(R * K), hip.empty(R * K).fill_(float("nan"))
    g0, g1 = hip.empty(N * K + N), hip.empty(N * K + N).fill_(float("nan"))
    h0, h1 = hip.empty(N2 * K2 + N2), hip.empty(N2 * K2 + N2).fill_(float("nan"))
    hip.dense_dgrad_wgrad(dY, W, X, K, dX0, g0, R, N, K)
    hip.dense_wgrad(dY2, X2, K2, h0, R, N2, K2, hip.empty(max(hip.dense_wgrad_scratch(R, N2, K2), 4)))
    hip.dense_dgrad_wgrad2(dY, W, X, K, dX1, g1, R, N, K, dY2, X2, K2, h1, N2, K2)
    assert torch.equal(dX0, dX1) and torch.equal(g0, g1)
    ref = np.concatenate([(dY2n.astype(np.float64).T @ X2n.astype(np.float64)).reshape(-1), dY2n.astype(np.float64).sum(0)])
    scale = np.abs(ref).max()
    e0, e1 = np.abs(h0.cpu().numpy() - ref).max() / scale, np.abs(h1.cpu().numpy() - ref).max() / scale
    assert e1 < 2e-6 and e1 < 4 * e0 + 2e-7, f"head weight gradient {e1} of the scale against fp64 (a0_dense_wgrad: {e0})"


@pytest.mark.parametrize("R,N", [(512, 256), (512, 800), (64, 256), (200, 132)])
def test_head_data_and_weight_gradient_side_by_side(hip, R, N):
    """a0_dense_dgrad_wgrad's small class (round 5): a distributional head's masked data gradient (R x 512 out, k = N) and weight gradient (N x 512 out, k = R) — unequal
    tile counts — in one launch.  The data gradient is the same body on the same tiles as a0_dense_dgrad (bit-identical); the weight gradient an unsplit sum on the bf16
    pipe, within fp32 rounding of the fp64 product like a0_dense_wgrad's."""
    K = 512
    assert hip.dense_dgrad_wgrad_ok2(R, N, K) and not hip.dense_dgrad_wgrad_ok(R, N, K) and not hip.dense_dgrad_wgrad_ok2(32, N, K)
    g = recipe.gen(8)
    dYn = g.standard_normal((R, N)).astype(np.float32)
    Xn = np.maximum(g.standard_normal((R, K)), 0).astype(np.float32)
    dY, X = D(hip, dYn), D(hip, Xn)
    W = D(hip, (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    dX0, dX1 = hip.empty(R * K), hip.empty(R * K).fill_(float("nan"))
    g0, g1 = hip.empty(N * K + N), hip.empty(N * K + N).fill_(float("nan"))
    hip.dense_dgrad(dY, W, X, dX0, R, N, K)
    hip.dense_wgrad(dY, X, K, g0, R, N, K, hip.empty(max(hip.dense_wgrad_scratch(R, N, K), 4)))
    hip.dense_dgrad_wgrad(dY, W, X, K, dX1, g1, R, N, K)
    assert torch.equal(dX0, dX1)
    ref = np.concatenate([(dYn.astype(np.float64).T @ Xn.astype(np.float64)).reshape(-1), dYn.astype(np.float64).sum(0)])
    scale = np.abs(ref).max()
    e0, e1 = np.abs(g0.cpu().numpy() - ref).max() / scale, np.abs(g1.cpu().numpy() - ref).max() / scale
    # (one fp32 chain over the R rows against a0_dense_wgrad's sum of shorter chains: a few ulp of the scale either way; measured 8.8e-7 / 1.5e-7 at R = 512, N = 800)
    assert e0 < 2e-6 and e1 < 2e-6, f"weight gradient {e1} of the scale against fp64 (a0_dense_wgrad: {e0})"


def test_sumtree_sample_batch_equals_separate_kernels(hip):
    """a0_sumtree_sample_batch (draws + descent + lookup + importance weights, one launch) against a0_rng_uniform + a0_sumtree_sample +
    a0_replay_lookup + a0_is_weights: identical indices, slots, metadata, priorities and weights."""
    size, B = 100000, 512
    g = recipe.gen(31)
    t = core.SumTree(size)
    t.tree[t.cap2:t.cap2 + size] = g.uniform(0.0, 2.0, size).astype(np.float32)
    t.tree[t.cap2 + 100:t.cap2 + 5000] = 0.0          # a dead stretch the descent must never enter
    t.rebuild()
    tree = D(hip, t.tree)
    r_act = D(hip, g.integers(0, 18, size).astype(np.int32)); r_rew = D(hip, g.standard_normal(size).astype(np.float32)); r_done = D(hip, (g.random(size) < 0.1).astype(np.float32))
    seed, stream, off, top, beta = 0x1234_0000_002A, 5, 4096, 77777, 0.45
    xi, idx0, p0 = hip.zeros(B), hip.zeros(B, dtype=torch.int64), hip.zeros(B)
    hip.rng_uniform(seed, stream, off, xi, B)
    hip.sumtree_sample(tree, t.cap2, xi, B, idx0, p0)
    slot0, act0, rew0, done0, io0, w0 = hip.zeros(B, dtype=torch.int32), hip.zeros(B, dtype=torch.int32), hip.zeros(B), hip.zeros(B), hip.zeros(B, dtype=torch.int64), hip.zeros(B)
    hip.replay_lookup(idx0, B, size, 0, size, slot0, r_act, r_rew, r_done, None, act0, rew0, done0, None, io0)
    hip.is_weights(p0, B, tree[1:2], top, beta, w0)
    idx1, slot1, act1, rew1, done1, p1, w1 = hip.zeros(B, dtype=torch.int64), hip.zeros(B, dtype=torch.int32), hip.zeros(B, dtype=torch.int32), hip.zeros(B), hip.zeros(B), hip.zeros(B), hip.zeros(B)
    hip.sumtree_sample_batch(seed, stream, off, tree, t.cap2, B, top, size, beta, r_act, r_rew, r_done, idx1, slot1, act1, rew1, done1, p1, w1)
    for a, b in ((idx0, idx1), (slot0, slot1), (act0, act1), (rew0, rew1), (done0, done1), (p0, p1), (w0, w1)):
        assert torch.equal(a, b)
    assert float(w1.max()) <= 1.0 and float(p1.min()) > 0.0


def _filled_replay(hip, size, B, extra=()):
    from agent0_amd.deepq.config import parse_overrides
    from agent0_amd.deepq.replay import ReplayDataset, TransitionBlock
    cfg = parse_overrides([f"replay.size={size}", f"learner.batch_size={B}", "replay.policy=prioritize", "wandb=false", "tb=false", *extra])
    cfg.obs_shape = (4, 36, 36)              # small rows: the priority logic does not depend on the frame size
    rp = ReplayDataset(cfg, ops=hip)
    rp.extend(TransitionBlock(size, start=0))
    return rp


def test_nan_skipped_update_leaves_the_sumtree_alone(hip):
    """A NaN loss makes the learner skip the step (agent.py:152-158) and the reference's trainer then never calls update_priority
    (trainer.py:103 sees q_loss None).  On the device the skip decision is state[3]; with it set, neither the leaves, nor the root, nor
    max_p may change — a NaN priority would poison total, the stratified segments and every later importance weight."""
    rp = _filled_replay(hip, 3000, 64)
    tree0, p0 = rp.tree.clone(), rp._pstate.clone()
    ids = D(hip, recipe.gen(3).integers(0, 3000, 64).astype(np.int64))
    loss = D(hip, np.full(64, np.nan, np.float32))
    state = hip.zeros(8, dtype=torch.int32)
    state[3] = 1
    rp.update_priority(ids, loss, state=state)
    assert torch.equal(rp.tree, tree0) and torch.equal(rp._pstate, p0) and torch.isfinite(rp.tree[1])
    # the same call with the flag down does write (and would have poisoned the tree with these losses)
    state[3] = 0
    good = D(hip, recipe.gen(4).uniform(0, 2, 64).astype(np.float32))
    rp.update_priority(ids, good, state=state)
    assert not torch.equal(rp.tree, tree0) and torch.isfinite(rp.tree).all()
    b = rp.sample()
    assert torch.isfinite(b.weights).all() and float(b.weights.max()) <= 1.0


@pytest.mark.parametrize("size", [200_000, 1_000_000])
def test_deferred_top_of_the_sumtree_changes_no_number(hip, size, monkeypatch):
    """Round 4: update_priority writes leaves and subtrees only (a0_sumtree_set_from_loss(defer_top = 1)); the levels with < 2048 nodes are recomputed by the NEXT
    batch's launch (a0_sumtree_sample_batch(rebuild_top = 1), which also walks those levels in LDS), by the rollout's range insert, or by the ``tree`` property for any
    other reader.  Batches (indices, slots, priorities, importance weights), max_p and the whole tree must be bit-identical to the three-launch sequence over rounds of
    sample / update / extend, and the tree equal to the oracle's after the same sets."""
    from agent0_amd.deepq.replay import TransitionBlock
    B = 512

    def run(defer):
        monkeypatch.setenv("A0_SUMTREE_DEFER_TOP", "1" if defer else "0")
        rp = _filled_replay(hip, size, B)
        assert rp._defer_top == defer and hip.sumtree_set_from_loss_ok(rp.cap2)
        g = recipe.gen(77)
        out, stale = [], []
        for r in range(6):
            b = rp.sample()
            out += [b.idx.clone(), b.slot.clone(), b.prio.clone(), b.weights.clone()]
            loss = g.uniform(0, 3, B).astype(np.float32)
            if r == 2:
                loss[:] = 0.0                   # a round of equal priorities
            rp.update_priority(b.idx, D(hip, loss))
            stale.append(rp._top_stale)
            if r == 3:
                rp.extend(TransitionBlock(4096, start=rp.written % size))      # the range insert recomputes the top itself
                assert not rp._top_stale
            if r == 4:
                out.append(rp._tree.clone())    # raw: stale top levels in the deferring run
                out.append(rp.tree.clone())     # through the property: up to date
                assert not rp._top_stale
        out += [rp.tree.clone(), rp._pstate.clone()]
        return rp, out, stale

    rp0, a, s0 = run(False)
    rp1, b, s1 = run(True)
    assert s0 == [False] * 6 and s1 == [True] * 6
    raw0, raw1 = a.pop(20), b.pop(20)
    assert torch.equal(raw0[2048:], raw1[2048:]) and not torch.equal(raw0[:2048], raw1[:2048]), "only the levels with < 2048 nodes were deferred"
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    t = core.SumTree(size)
    t.tree[:] = 0
    t.tree[t.cap2:] = rp1.tree[rp1.cap2:].cpu().numpy()
    t.rebuild()
    assert np.array_equal(t.tree[1:], rp1.tree.cpu().numpy()[1:])


def test_priority_update_of_a_batch_larger_than_one_workgroup(hip):
    """B = 2048 > the 1024 leaves a0_sumtree_set stages in LDS: ReplayDataset.update_priority splits the batch in order (a later duplicate
    still wins across the split); the C entry point itself refuses n > 1024.  Tree bytes against the oracle's set on the same pairs."""
    from agent0_amd._abi import A0Error
    B, size = 2048, 5000
    rp = _filled_replay(hip, size, B)
    t = core.SumTree(size)
    t.tree[:] = rp.tree.cpu().numpy()
    g = recipe.gen(9)
    ids = g.integers(0, size, B)
    ids[1500] = ids[3]                      # a duplicate that straddles the split: the later one (second chunk) must win
    loss = g.uniform(0, 3, B).astype(np.float32)
    rp.update_priority(D(hip, ids.astype(np.int64)), D(hip, loss))
    val = np.sqrt(loss + np.float32(0.01))            # fp32 add, IEEE sqrt — what a0_prio_pow computes for alpha = 0.5
    t.set(ids, val)
    got = rp.tree.cpu().numpy()
    assert np.array_equal(got[t.cap2:], t.tree[t.cap2:]), "leaves (IEEE sqrt on both sides)"
    assert np.array_equal(got, t.tree), "every ancestor"
    assert got[t.cap2 + ids[3]] == val[1500]
    with pytest.raises(A0Error, match="1024"):
        hip.sumtree_set(rp.tree, rp.cap2, D(hip, ids.astype(np.int64)), D(hip, val), B)
    b = rp.sample()                          # B > 1024 takes the unfused sampling path
    assert int(b.idx.min()) >= 0 and int(b.idx.max()) < size and torch.isfinite(b.weights).all()


@pytest.mark.parametrize("size,prioritize", [(64, False), (40, True)])
def test_replay_extend_accepts_the_reference_transition_list(hip, size, prioritize):
    """ReplayDataset.extend(list) with the reference's own transition tuples (agent.py:78-81: ``(frames blob, at, rt, dt)`` per env and
    step; replay.py:45-53) — what its Actor.sample / TrainerNode hand over.  The list is produced by the oracle actor under the G7 script
    (whose blobs are pinned to the reference's by the g7 fixture's hashes); after the insert the ring must hold exactly those rows, in the
    reference's deque order, also when the ring wraps (size 40 < 48 transitions)."""
    import hashlib
    from agent0_amd.deepq.config import parse_overrides
    from agent0_amd.deepq.replay import ReplayDataset
    from test_oracle_golden import FakeEnv, _script, params_for
    from util import golden
    n_step = 3
    g = golden(f"g7_actor_n{n_step}_life1")
    spec = recipe.SPECS["dqn"]
    E, T = 4, 12
    env = FakeEnv(E, _script(E, T, 70 + n_step, True))
    it = iter(zip(g["draws_int"], g["draws_u"]))
    act = oactor.OracleActor(env, params_for(spec, 11), spec, n_step=n_step, sample_steps=6, draw=lambda E_: next(it))
    cfg = parse_overrides([f"replay.size={size}", "learner.batch_size=8", "wandb=false", "tb=false"] + (["replay.policy=prioritize"] if prioritize else []))
    cfg.obs_shape, cfg.action_dim = (4, 84, 84), 4
    rp = ReplayDataset(cfg, ops=hip)
    rp.extend([])
    assert len(rp) == 0
    everything = []
    for call in range(2):
        data, _, _ = act.sample(float(g["eps"]))
        # the blob in every form a caller may hold it: (8, 84, 84) array, flat array, bytes, bytearray, memoryview; rewards as numpy float64
        forms = [lambda f: f, lambda f: f.reshape(-1), lambda f: f.tobytes(), lambda f: bytearray(f.tobytes()), lambda f: memoryview(f.tobytes())]
        lst = [(forms[i % 5](frames), at, rt, dt) for i, (frames, at, rt, dt) in enumerate(data)]
        rp.extend(lst)
        everything += data
        assert len(rp) == min(len(everything), size) and rp.written == len(everything)
    n = len(everything)
    assert n == 48
    hashes = [np.frombuffer(hashlib.sha256(f.tobytes()).digest()[:8], dtype=np.uint64)[0] for f, *_ in everything]
    assert np.array_equal(np.array(hashes, dtype=np.uint64), g["blob_hash"]), "the list itself is the reference's (g7 fixture)"
    first = n - len(rp)                     # the deque dropped the oldest entries
    for i in range(len(rp)):
        row, at, rt, dt, pr, idx = rp[i]
        f, a_w, r_w, d_w = everything[first + i]
        assert np.array_equal(row, f.reshape(-1)) and at == int(a_w) and rt == float(np.float32(r_w)) and dt == bool(d_w) and idx == i
    if prioritize:
        assert float(rp.priority[-1]) == 1.0 and 0.4 < rp.beta < 0.41
    with pytest.raises(ValueError, match="frame bytes|lz4"):
        rp.extend([(b"\x00" * 100, 0, 0.0, False)])
    with pytest.raises(TypeError):
        rp.extend([(1, 2, 3)])


@pytest.mark.parametrize("E,nstack,fb", [(5, 4, 64), (256, 4, 7056), (37, 2, 16)])
def test_host_env_pcie_legs_as_library_calls(hip, E, nstack, fb):
    """Round 4 (N1): a0_env_pool_upload — newest frames + scalars + the whole stacks of the envs whose `advance` scalar is 0, then the device frame stack — must leave
    exactly what the per-copy sequence leaves (the stack a host-side FrameStack would hold, atari_wrappers.py:63 / agent.py:27), report the number of whole stacks, and
    work on the caller's stream; a0_env_pool_send must put the actions and then the 8-byte step word into page-locked host memory through its device address."""
    import ctypes as C
    from agent0_amd._abi import check
    from agent0_amd.ops import _stream
    g = np.random.default_rng(E * 131 + fb)
    n_scal, adv_row = 7, 6
    prev = g.integers(0, 256, (E, nstack, fb), dtype=np.uint8)
    host_obs = g.integers(0, 256, (E, nstack, fb), dtype=np.uint8)           # what the workers hold: used for the envs uploaded whole
    newest = g.integers(0, 256, (E, fb), dtype=np.uint8)
    scal = g.standard_normal((n_scal, E)).astype(np.float32)
    adv = (g.random(E) < 0.7).astype(np.float32)
    adv[0], adv[E - 1] = 0.0, 0.0                                            # runs at both ends (E - 2 .. E - 1 adjacent when the draw says so)
    if E > 3:
        adv[1], adv[2] = 0.0, 1.0
    scal[adv_row] = adv
    want = np.where(adv[:, None, None] != 0, np.concatenate([prev[:, 1:], newest[:, None]], axis=1), host_obs)
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
    new_h, scal_h, obs_h = pin(newest), pin(scal), pin(host_obs)
    prev_d, out_d = D(hip, prev), torch.full((E, nstack, fb), 7, dtype=torch.uint8, device=hip.device)
    new_d, scal_d = torch.zeros(E * fb, dtype=torch.uint8, device=hip.device), torch.zeros(n_scal, E, device=hip.device)
    n_whole = C.c_int(-1)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        check(hip.lib.a0_env_pool_upload(new_h.data_ptr(), new_d.data_ptr(), scal_h.data_ptr(), scal_d.data_ptr(), n_scal, adv_row, obs_h.data_ptr(), prev_d.data_ptr(),
                                         out_d.data_ptr(), E, nstack, fb, C.byref(n_whole), _stream()), "a0_env_pool_upload")
    side.synchronize()
    assert n_whole.value == int((adv == 0).sum())
    assert np.array_equal(out_d.cpu().numpy(), want) and np.array_equal(scal_d.cpu().numpy(), scal) and np.array_equal(new_d.cpu().numpy().reshape(E, fb), newest)
    # the per-copy sequence of round 3 gives the same device stack
    out2 = torch.full((E, nstack, fb), 9, dtype=torch.uint8, device=hip.device)
    for e in np.flatnonzero(adv == 0):
        out2[e].copy_(obs_h[e])
    hip.env_frame_stack(prev_d.view(-1), new_d, scal_d[adv_row], out2.view(-1), E, nstack, fb)
    assert torch.equal(out2, out_d)
    # send: actions, then the word
    act = D(hip, g.integers(0, 18, E, dtype=np.int32))
    act_h, ctl_h = torch.full((E,), -1, dtype=torch.int32).pin_memory(), torch.zeros(2, dtype=torch.int64).pin_memory()
    dp = [C.c_void_p(), C.c_void_p()]
    check(hip.lib.a0_host_device_pointer(act_h.data_ptr(), C.byref(dp[0])), "a0_host_device_pointer")
    check(hip.lib.a0_host_device_pointer(ctl_h.data_ptr(), C.byref(dp[1])), "a0_host_device_pointer")
    word = (2 << 56) | 123456789
    check(hip.lib.a0_env_pool_send(act.data_ptr(), dp[0].value, E, dp[1].value + 8, word, _stream()), "a0_env_pool_send")
    torch.cuda.synchronize()
    assert torch.equal(act_h, act.cpu()) and int(ctl_h[1]) == word and int(ctl_h[0]) == 0
    from agent0_amd.ops import A0Error
    with pytest.raises(A0Error):
        check(hip.lib.a0_env_pool_upload(new_h.data_ptr(), new_d.data_ptr(), scal_h.data_ptr(), scal_d.data_ptr(), n_scal, n_scal, obs_h.data_ptr(), prev_d.data_ptr(),
                                         out_d.data_ptr(), E, nstack, fb, None, _stream()), "a0_env_pool_upload")
