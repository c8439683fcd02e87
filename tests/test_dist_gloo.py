"""CPU, world_size 2, gloo: the data-parallel path (agent0_amd.deepq.dist.GradAllReduce hooked into DeviceLearner) driven by the
emulation backend.  Property checked: a SUM all-reduce of the per-rank gradients with Adam eps = 1e-2/(W*B) reproduces ONE
reference step on the concatenated global batch (the reference reduces its loss by sum, agent.py:154), replicas stay
bit-identical, and a NaN on one rank skips the step on every rank."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, nan_rank, out, algo="dqn"):
    sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import recipe
    from agent0_amd.deepq.dist import GradAllReduce, init_process_group
    from agent0_amd.deepq.engine import DeviceLearner
    from agent0_amd.deepq.layout import NetLayout
    from cpu_ops import CpuOps

    torch.set_num_threads(2)
    init_process_group(backend="gloo")
    spec = recipe.NetSpec("dqn", 4, dueling=True, obs_shape=(4, 36, 36)) if algo == "dqn" else recipe.NetSpec(algo, 4, obs_shape=(4, 36, 36))
    L = NetLayout.from_spec(spec)
    B = 4
    ops = CpuOps()
    dev = DeviceLearner(ops, L, B, double_q=True, target_update_freq=1)
    sd = recipe.make_state_dict(spec, 11)
    dev.online.load_state_dict(sd); dev.target.load_state_dict(recipe.make_state_dict(spec, 12))
    dev.grad_hook = GradAllReduce(L.n_adam)
    dev.adam_eps = 1e-2 / (world * B)
    frames = recipe.make_frames(world * B, 61, spec.obs_shape)[rank * B:(rank + 1) * B]
    a, r, d, w = [x[rank * B:(rank + 1) * B] for x in recipe.make_transitions(world * B, 4, 62)]
    r = r.copy()
    if nan_rank == rank:
        r[0] = np.nan
    dev.update(torch.from_numpy(frames.copy()).reshape(-1), None, 2 * 4 * 36 * 36, torch.from_numpy(a.astype(np.int32)), torch.from_numpy(r),
               torch.from_numpy(d.astype(np.float32)), torch.from_numpy(w))
    out[rank] = (dev.online.flat.clone().numpy(), dev.target.flat.clone().numpy(), dev.state.clone().numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port() -> int:
    import socket

    with socket.socket() as sk:        # a fixed port can still sit in TIME_WAIT from the previous run of this suite
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(world, nan_rank, port=None, algo="dqn"):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, nan_rank, out, algo), nprocs=world, join=True)
    return dict(out)


def test_sum_allreduce_equals_global_batch_step():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import recipe
    from oracle import learner as olearner
    from oracle.losses import Hyper

    res = _run(2, -1, 29533)
    p0, t0, s0 = res[0]
    p1, t1, s1 = res[1]
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1), "replicas must stay bit-identical"
    assert s0[1] == 1 and s1[1] == 1
    spec = recipe.NetSpec("dqn", 4, dueling=True, obs_shape=(4, 36, 36))
    B = 8
    ora = olearner.OracleLearner(spec, recipe.make_state_dict(spec, 11), recipe.make_state_dict(spec, 12), Hyper(double_q=True), batch_size=B, target_update_freq=1)
    frames = recipe.make_frames(B, 61, spec.obs_shape)
    a, r, d, w = recipe.make_transitions(B, 4, 62)
    ora.train(frames.reshape(B, -1), a, r, d.astype(np.float32), w, np.arange(B))       # Adam eps = 1e-2/8 == 1e-2/(W*B)
    from agent0_amd.deepq.layout import NetLayout
    L = NetLayout.from_spec(spec)
    got = L.unpack(torch.from_numpy(p0))
    for k, v in got.items():
        assert float((v - ora.po[k].detach()).abs().max()) < 2e-5, k


def test_nan_on_one_rank_skips_everywhere():
    res = _run(2, 1, 29534)
    for rank in (0, 1):
        p, t, s = res[rank]
        assert s[1] == 0 and s[2] == 1, "update_steps unchanged, one skipped step on every rank"
    assert np.array_equal(res[0][0], res[1][0])


def test_fqf_fraction_net_is_reduced_too():
    """BASELINE configs[4] (fqf, data parallel): the fraction net is trained by its own RMSprop step (agent.py:140-147); its gradient travels
    with the dense bucket, so replicas stay bit-identical and every block — fraction net included — equals ONE reference step on the
    concatenated global batch."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import recipe
    from oracle import learner as olearner
    from oracle.losses import Hyper

    res = _run(2, -1, algo="fqf")
    p0, t0, s0 = res[0]
    p1, t1, s1 = res[1]
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1), "replicas must stay bit-identical"
    spec = recipe.NetSpec("fqf", 4, obs_shape=(4, 36, 36))
    B = 8
    ora = olearner.OracleLearner(spec, recipe.make_state_dict(spec, 11), recipe.make_state_dict(spec, 12), Hyper(double_q=True), batch_size=B, target_update_freq=1)
    frames = recipe.make_frames(B, 61, spec.obs_shape)
    a, r, d, w = recipe.make_transitions(B, 4, 62)
    ora.train(frames.reshape(B, -1), a, r, d.astype(np.float32), w, np.arange(B))
    from agent0_amd.deepq.layout import NetLayout
    L = NetLayout.from_spec(spec)
    got = L.unpack(torch.from_numpy(p0))
    assert any("fraction" in k for k in got)
    for k, v in got.items():
        ref = ora.po[k].detach()
        assert float((v - ref).abs().max()) < 2e-5 + 2e-5 * float(ref.abs().max()), k
