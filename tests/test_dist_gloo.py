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


class _FakeDpOps:
    """Stand-in for the a0_dp_* entry points (no RCCL on a CPU box): records calls, fails where told to."""

    def __init__(self, fail_id=False, fail_init=False, hang_init=False):
        self.fail_id, self.fail_init, self.hang_init, self.destroyed, self.inits = fail_id, fail_init, hang_init, [], 0

    def dp_unique_id(self):
        if self.fail_id:
            raise RuntimeError("a0_dp: cannot load librccl.so.1: no such file")
        return bytes(range(128))

    def dp_init(self, blob, rank, world):
        assert blob == bytes(range(128)), "every rank received rank 0's blob"
        self.inits += 1
        if self.hang_init:
            import time
            time.sleep(30)
        if self.fail_init:
            raise RuntimeError("ncclCommInitRank: unhandled system error")
        return 1000 + rank

    def dp_destroy(self, comm):
        self.destroyed.append(comm)


def _agreement_worker(rank, world, port, case, out):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from agent0_amd.deepq import dist as adist

    adist.init_process_group(backend="gloo")
    ops = _FakeDpOps(fail_id=(case == "id" and rank == 1), fail_init=(case == "init" and rank == 0))
    try:
        comm = adist.collective_communicator(ops)
        res = ("comm", comm)
    except adist.DpUnavailable as e:
        res = ("unavailable", str(e))
    # the very next collective pairs up on every rank (a rank that skipped the blob broadcast used to pair a
    # differently sized broadcast with it)
    x = torch.full((7,), float(rank + 1))
    dist.all_reduce(x)
    out[rank] = res + (float(x[0]), ops.inits, list(ops.destroyed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["ok", "id", "init"])
def test_rccl_hook_choice_is_collective(case):
    """RCCL failing on ONE rank (library missing on rank 1 / ncclCommInitRank failing on rank 0) must put EVERY rank on the same branch:
    all get a communicator or all raise DpUnavailable (ranks that did get one destroy it), and the group's collectives stay in step."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_agreement_worker, args=(2, _free_port(), case, out), nprocs=2, join=True)
    res = dict(out)
    kinds = {res[r][0] for r in (0, 1)}
    assert len(kinds) == 1, res
    assert res[0][2] == 3.0 and res[1][2] == 3.0
    if case == "ok":
        assert res[0][1] == 1000 and res[1][1] == 1001 and res[0][4] == [] and res[1][4] == []
    elif case == "id":
        assert kinds == {"unavailable"} and "this rank" in res[1][1] and "another rank" in res[0][1]
        assert res[0][3] == 0 and res[1][3] == 0, "nobody enters ncclCommInitRank when a rank cannot load RCCL"
    else:
        assert kinds == {"unavailable"} and res[1][4] == [1001], "the rank whose communicator was created destroys it"


def test_dp_init_timeout_raises_instead_of_hanging():
    from agent0_amd.deepq import dist as adist
    import time

    t0 = time.time()
    with pytest.raises(adist.DpInitTimeout, match="did not return within"):
        adist._call_with_timeout(lambda: time.sleep(20), 0.3, "a0_dp_init (ncclCommInitRank)")
    assert time.time() - t0 < 5
    assert adist._call_with_timeout(lambda: 41 + 1, 5, "x") == 42
    with pytest.raises(ValueError):
        adist._call_with_timeout(lambda: (_ for _ in ()).throw(ValueError("boom")), 5, "x")


def _bench_line_worker(rank, world, port, out):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import bench
    from agent0_amd.deepq.dist import GradAllReduce, init_process_group

    init_process_group(backend="gloo")
    hook = GradAllReduce(1000)
    # each rank's clock: rank 1 is the straggler of the K = 4 steps, and its barrier-to-barrier time is the larger one
    dt_local, dt = (0.040, 0.0505) if rank == 0 else (0.050, 0.051)
    dt_max, per_rank = bench.rank_clock(dist, dt, dt_local, rank, world, 4, "cpu")
    rep = bench.exchange_report(hook, world)
    out[rank] = (dt_max, per_rank, rep, bench.throughput_fields(world, 80 * 256, 20, 4, 2, dt_max))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_line_assembly_over_two_ranks():
    """bench.py's multi-rank bookkeeping with the GPU parts left out (gloo, world_size 2): the clock is the MAX over ranks of the
    barrier-to-barrier time, per_rank_ms_per_step carries every rank's own time, `value` counts the units of ALL ranks, and the "rccl" object is what the
    exchange itself reports (rank count, an all-reduce of ones) checked against WORLD_SIZE."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bench_line_worker, args=(2, port, out), nprocs=2, join=True)
    res = dict(out)
    for rank in (0, 1):
        dt_max, per_rank, rep, tp = res[rank]
        assert dt_max == pytest.approx(0.051)
        assert per_rank["ranks"] == [10.0, 12.5] and per_rank["min"] == 10.0 and per_rank["max"] == 12.5
        assert rep["nranks"] == 2 and rep["rank"] == rank and rep["allreduce_of_ones"] == 2.0 and rep["matches_world_size"] is True
        assert tp["n_gpus"] == 2 and tp["scaling"] == "weak" and tp["ms_per_step"] == pytest.approx(12.75)
        assert tp["value"] == pytest.approx(2 * 80 * 256 * 4 / 0.051, rel=1e-6)
        assert tp["updates_per_sec"] == pytest.approx(2 * 20 * 4 / 0.051, rel=1e-4)
    import bench
    assert bench.exchange_report(None, 2) is None
