"""Data parallelism over the GPUs of one node: per-rank env/replay shards, one RCCL all-reduce per update.

No reference counterpart (the reference has no torch.distributed / NCCL call anywhere; its only inter-process traffic
is Launchpad's gRPC, launch.py:166-176).  Design: every rank owns its vectorized envs, its replay ring
and sum-tree and samples locally; the flat fp32 gradient buffer (6.7-14 MB) is SUM-reduced once per update over xGMI
with ``torch.distributed`` (backend "nccl" == RCCL on ROCm), together with the NaN-skip flag so replicas stay in
lock-step.  Because the reference reduces the loss by SUM (agent.py:154), a SUM all-reduce with Adam eps = 1e-2/(W*B)
is exactly the reference step on the W*B global batch.
"""
from __future__ import annotations

import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def free_port() -> int:
    """A TCP port that is free right now on the loopback interface (for a job's rendezvous)."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def dp_forced() -> bool:
    """A0_DP_FORCE=1: run the data-parallel code path (process group, gradient buckets, graphs split around the exchange) even with
    one rank, so that a one-GPU box can exercise RCCL + hipGraph capture + the overlap streams (tests/test_gpu_trainer.py)."""
    return os.environ.get("A0_DP_FORCE") == "1"


def init_process_group(backend: str | None = None):
    import torch.distributed as dist

    rank, local_rank, world = env_world()
    if (world > 1 or dp_forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def graph_capture_kwargs() -> dict:
    """Arguments for ``torch.cuda.graph`` while a process group is alive: RCCL's watchdog thread polls events concurrently, so only the
    capturing thread's own calls may count as part of (or as errors of) the capture."""
    import torch.distributed as dist

    return dict(capture_error_mode="thread_local") if (dist.is_available() and dist.is_initialized()) else {}


class GradAllReduce:
    """``DeviceLearner.grad_hook``: SUM the flat gradient buffer and MAX the NaN flag across ranks, in two buckets.

    The flat layout puts the convolution blocks first ([0, conv_end), 0.3 MB) and the dense blocks after them ([conv_end, n), 6.4 MB
    for dqn ... 14 MB for noisy c51).  The backward pass finishes the dense blocks FIRST, so their all-reduce — 95 % of the bytes —
    is issued asynchronously (RCCL runs it on its own stream over xGMI) and overlaps the encoder backward, which is ~40 % of an
    update's kernel time; the NaN flag rides at its tail as a float (any rank's NaN makes the sum nonzero); the small convolution
    bucket follows, then the optimizer waits for both: two collectives per update."""

    bucketed = True       # DeviceLearner calls start_dense / finish around the encoder backward (a plain callable gets one call instead)

    def __init__(self, n_grad: int, group=None):
        import torch.distributed as dist

        self.dist, self.n, self.group = dist, n_grad, group
        self.world = dist.get_world_size(group)
        self.active = (self.world > 1 or dp_forced()) and os.environ.get("A0_DP_DRYRUN") != "1"      # DRYRUN: graph split without collectives (diagnostics)
        self._dense = None

    def start_dense(self, grads: torch.Tensor, conv_end: int, end: int | None = None):
        """Asynchronous SUM of grads[conv_end:end]; ``end`` may reach past the parameters to the tail slot that carries the NaN flag as
        a float (DeviceLearner), so that the flag needs no collective of its own."""
        if self.active:
            self._dense = self.dist.all_reduce(grads[conv_end: (self.n if end is None else end)], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, grads: torch.Tensor, state: torch.Tensor | None, conv_end: int):
        """The convolution bucket, the NaN flag when it did not travel with the dense bucket (``state`` given), and the join."""
        if not self.active:
            return
        self.dist.all_reduce(grads[:conv_end], op=self.dist.ReduceOp.SUM, group=self.group)
        if state is not None:
            self.dist.all_reduce(state[0:1], op=self.dist.ReduceOp.MAX, group=self.group)
        if self._dense is not None:
            self._dense.wait()          # the current stream waits for the dense bucket (no host block on RCCL)
            self._dense = None

    def report(self) -> dict:
        """The torch.distributed counterpart of RcclGradAllReduce.report (collective)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if self.dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        x = torch.ones(1024, device=dev)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM, group=self.group)
        return {"backend": f"torch.distributed ({self.dist.get_backend(self.group)})", "nranks": self.dist.get_world_size(self.group), "rank": self.dist.get_rank(self.group),
                "device": (torch.cuda.current_device() if dev.type == "cuda" else None), "allreduce_of_ones": float(x[0]), "allreduce_of_ones_last": float(x[-1]), "in_graph": False}

    def __call__(self, grads: torch.Tensor, state: torch.Tensor):
        """One-shot form (no overlap): the whole buffer, then the flag."""
        if not self.active:
            return
        self.dist.all_reduce(grads[: self.n], op=self.dist.ReduceOp.SUM, group=self.group)
        self.dist.all_reduce(state[0:1], op=self.dist.ReduceOp.MAX, group=self.group)


class DpUnavailable(RuntimeError):
    """The C-ABI exchange cannot be used by this process GROUP (agreed on by every rank): use torch.distributed instead."""


class DpInitTimeout(RuntimeError):
    """A rank did not come back from a collective initialisation call: the job cannot continue (nothing to fall back to safely)."""


def _call_with_timeout(fn, seconds: float, what: str):
    """Run ``fn`` on a helper thread bound to the caller's device; DpInitTimeout if it has not returned after ``seconds`` (the thread is a
    daemon: the process can then exit non-zero instead of hanging in a collective a peer never entered)."""
    import threading

    box, dev = {}, torch.cuda.current_device() if torch.cuda.is_available() else None

    def run():
        try:
            if dev is not None:
                torch.cuda.set_device(dev)
            box["value"] = fn()
        except BaseException as e:      # noqa: BLE001
            box["error"] = e

    t = threading.Thread(target=run, daemon=True, name="a0-dp-init")
    t.start()
    t.join(seconds)
    if t.is_alive():
        raise DpInitTimeout(f"{what} did not return within {seconds:.0f} s (a peer rank never entered it?)")
    if "error" in box:
        raise box["error"]
    return box["value"]


def _agree(group, ok: bool, what: str, err: str = "", on_fail=None):
    """MIN all-reduce of a success flag over the torch group: returns on every rank, or raises DpUnavailable on every rank."""
    import torch.distributed as dist

    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if float(flag[0]) < 0.5:
        if on_fail is not None:
            on_fail()
        raise DpUnavailable(f"{what} failed on " + (f"this rank: {err}" if not ok else "another rank"))


def collective_communicator(ops, group=None) -> int:
    """An RCCL communicator over the C-ABI for every rank of ``group``, or DpUnavailable on every rank (ranks that did get one destroy it).
    Stage 1: librccl resolves on every rank (a0_dp_unique_id is local; only rank 0's id is used).  Stage 2: the rendezvous blob (ALWAYS
    broadcast), then a0_dp_init — ncclCommInitRank is itself collective, and stage 1 has established that every rank will enter it; a
    rank that does not come back within A0_DP_INIT_TIMEOUT seconds ends the job with DpInitTimeout."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    my_id, err = bytes(128), ""
    try:
        my_id = ops.dp_unique_id()
    except Exception as e:      # noqa: BLE001
        err = str(e)
    _agree(group, not err, "loading RCCL through the C-ABI (a0_dp_unique_id)", err)
    blob = torch.frombuffer(bytearray(my_id), dtype=torch.uint8).clone().to(dev)
    dist.broadcast(blob, src=0, group=group)
    comm, err = 0, ""
    try:
        comm = _call_with_timeout(lambda: ops.dp_init(bytes(blob.cpu().numpy().tobytes()), rank, world),
                                  float(os.environ.get("A0_DP_INIT_TIMEOUT", "180")), "a0_dp_init (ncclCommInitRank)")
    except DpInitTimeout:
        raise
    except Exception as e:      # noqa: BLE001
        err = str(e)
    _agree(group, not err, "a0_dp_init", err, (lambda: ops.dp_destroy(comm)) if comm else None)
    return comm


class RcclGradAllReduce:
    """``DeviceLearner.grad_hook`` over the C-ABI's own exchange (``a0_dp_allreduce``: RCCL on a HIP stream).  Same two buckets as
    GradAllReduce, but every call is an ordinary stream-ordered launch, so the whole update — forward, dense backward, the dense bucket's
    all-reduce on a side stream WHILE the encoder backward runs on the main one, the convolution bucket's all-reduce, Adam — is captured
    into ONE hipGraph: the side stream forks from and joins the capturing stream through events, which the graph records as parallel
    branches.  One communicator, used on the side stream only, so the two collectives are totally ordered on every rank.  The
    communicator's 128-byte rendezvous blob travels through torch.distributed (broadcast from rank 0)."""

    bucketed = True
    in_graph = True       # BaseLearner._update: no eager call between graphs is needed

    def __init__(self, ops, n_grad: int, group=None):
        """Collective: every rank of ``group`` must construct this at the same point.  Each stage that can fail on ONE rank (loading
        librccl, ncclCommInitRank, the eager self-test) is followed by a MIN all-reduce of a success flag over the torch group, so either
        every rank ends up with a communicator or every rank raises ``DpUnavailable`` — and then ``make_grad_hook`` puts all of them on the
        torch.distributed exchange.  No rank skips a collective another rank is waiting in."""
        import torch.distributed as dist

        self.ops, self.n = ops, n_grad
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.world, self.comm = world, 0
        self.comm = collective_communicator(ops, group)          # stages 1 and 2
        agree = lambda ok, what, err="": _agree(group, ok, what, err, self.close)
        self.side = torch.cuda.Stream()
        self.active = os.environ.get("A0_DP_DRYRUN") != "1"
        if self.active:
            # stage 3: one eager all-reduce (checks the communicator; RCCL's lazy allocations happen outside any capture)
            err = ""
            try:
                self._eager_self_test(rank, world)
            except Exception as e:      # noqa: BLE001
                err = str(e)
            agree(not err, "the a0_dp_allreduce self-test", err)
            # stage 4: can the call be captured?  (decides in-graph vs eager; MIN over ranks inside)
            self._capture_self_test(rank, world, group)

    def _eager_self_test(self, rank: int, world: int):
        x = torch.full((1024,), float(rank + 1), device="cuda")
        want = world * (world + 1) / 2.0
        self.ops.dp_allreduce(self.comm, x, x.numel())
        torch.cuda.synchronize()
        if abs(float(x[0]) - want) > 1e-3 or abs(float(x[-1]) - want) > 1e-3:
            raise RuntimeError(f"a0_dp_allreduce self-test: got {float(x[0])}, expected {want}")

    def _capture_self_test(self, rank: int, world: int, group):
        """The same call captured on the side stream into a small hipGraph and replayed.  If the capture does not work with this RCCL
        build the hook stays usable — it then runs eagerly between three graphs (in_graph = False) — and every rank takes the same
        decision (MIN over ranks)."""
        import sys
        import torch.distributed as dist

        x = torch.full((1024,), float(rank + 1), device="cuda")
        want = world * (world + 1) / 2.0
        ok = 1.0
        try:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, **graph_capture_kwargs()):
                self.side.wait_stream(torch.cuda.current_stream())
                self.ops.dp_allreduce(self.comm, x, x.numel(), stream=self.side)
                torch.cuda.current_stream().wait_stream(self.side)
            g.replay()
            torch.cuda.synchronize()
            if abs(float(x[0]) - want) > 1e-3:
                ok = 0.0
        except Exception as e:      # noqa: BLE001
            print(f"agent0_amd.dist: a0_dp_allreduce cannot be captured into a hipGraph here ({e}); it will run eagerly between graphs", file=sys.stderr)
            ok = 0.0
            torch.cuda.synchronize()
        flag = torch.tensor([ok], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        self.in_graph = bool(float(flag[0]) > 0.5)

    def start_dense(self, grads: torch.Tensor, conv_end: int, end: int | None = None):
        end = self.n if end is None else end
        if self.active:
            self.side.wait_stream(torch.cuda.current_stream())      # the dense blocks' gradients (and the NaN flag behind them) are final
            self.ops.dp_allreduce(self.comm, grads[conv_end:end], end - conv_end, stream=self.side)

    def finish(self, grads: torch.Tensor, state, conv_end: int):
        if not self.active:
            return
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)                                   # the convolution blocks' gradients are final
        self.ops.dp_allreduce(self.comm, grads[:conv_end], conv_end, stream=self.side)
        cur.wait_stream(self.side)                                   # Adam sees both buckets reduced

    def report(self) -> dict:
        """What the exchange itself says about the job (collective: every rank calls it at the same point): RCCL's own rank count / rank / device for the communicator
        the gradients travel on (a0_dp_info) and the result of an all-reduce of ones through it — a record that reads nranks = 8 and allreduce_of_ones = 8.0 was
        produced by eight ranks that really exchanged data."""
        nranks, rank, device = self.ops.dp_info(self.comm)
        x = torch.ones(1024, device="cuda")
        self.ops.dp_allreduce(self.comm, x, x.numel())
        torch.cuda.synchronize()
        return {"backend": "rccl (a0_dp_allreduce: ncclAllReduce through the C-ABI)", "nranks": nranks, "rank": rank, "device": device,
                "allreduce_of_ones": float(x[0]), "allreduce_of_ones_last": float(x[-1]), "in_graph": bool(self.in_graph)}

    def close(self):
        if self.comm:
            self.ops.dp_destroy(self.comm)
            self.comm = 0


def make_grad_hook(ops, n_grad: int, group=None):
    """The gradient exchange for this process group: the in-graph RCCL path on GPUs (A0_DP_BACKEND=torch forces the torch.distributed
    calls, which is also what a CPU/gloo group gets — the variable must be the same on every rank); if RCCL cannot be initialised through
    the C-ABI on ANY rank, ALL ranks use the torch path and say so.  Collective: call it at the same point on every rank."""
    import sys

    if torch.cuda.is_available() and os.environ.get("A0_DP_BACKEND", "rccl") != "torch":
        try:
            return RcclGradAllReduce(ops, n_grad, group)
        except DpUnavailable as e:      # raised on EVERY rank or on none (RcclGradAllReduce.__init__): the torch path computes the same sums
            print(f"agent0_amd.dist: a0_dp_* unavailable ({e}); every rank falls back to torch.distributed all_reduce", file=sys.stderr)
        # anything else (DpInitTimeout, an error outside the agreed stages) propagates: the job exits non-zero rather than run with ranks
        # on different exchanges
    return GradAllReduce(n_grad, group)
