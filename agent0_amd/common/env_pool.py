"""Host front-end for CPU environments: worker processes step slices of the vector env and write straight into a
double-buffered, page-locked uint8 ring that the GPU reads by DMA.

What it replaces: the reference's ``make_atari`` returns a gymnasium AsyncVectorEnv (agent0/common/atari_wrappers.py:59-69) whose
``step`` pickles the (E,4,84,84) observation batch through pipes into the actor process, which then copies it to the GPU with a blocking
``torch.from_numpy(obs).to(device)`` (agent0/deepq/agent.py:27).  Here

  * every worker owns envs [e0, e0 + k) and writes observations, rewards, done flags and finished-episode records into shared memory
    that is registered with the HIP runtime as page-locked (``hipHostRegister``), one half of a two-deep ring per step — there is no
    pickling and no staging copy, and the half the GPU may still be reading is never the one being written;
  * actions travel the other way without a host-side wait: the actor's action tensor and a step sequence number are DMA-copied into the
    same shared block in stream order, and the workers poll the sequence number — the Python thread never calls ``.cpu()`` or
    ``stream.synchronize()`` on the step path; it only waits for its own workers (CPU work) before enqueueing the next upload;
  * uploads run on a dedicated copy stream; the compute stream waits for them through an event, so the n-step / replay-insert kernels
    of step t and anything else already enqueued overlap the DMA of step t+1's observations.

Interface: the one ``Actor`` consumes (DeviceSynthVecEnv): ``reset() -> (obs, info)``, ``step(action, final_mask, final_ret, ctrl) ->
(obs, reward, terminal, truncated, info)`` with device tensors; ``info["life_loss"]``, ``info["final_mask"]``, ``info["final_ret"]`` carry
what the reference reads from ``info["life_loss"]`` and ``info["final_info"]`` / ``info["_final_info"]`` (agent.py:55-62,85-88).

A slice is any object with the gymnasium VECTOR contract over its k envs (``reset() -> (obs[k,...], info)``, ``step(a[k]) -> (obs,
reward, terminated, truncated, info)`` with autoreset and the same info keys); ``VectorizedSingles`` builds one from single envs, doing
the autoreset, episode statistics and reward clipping the reference gets from gymnasium's wrappers.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import numpy as np
import torch

CMD_NONE, CMD_STEP, CMD_RESET, CMD_CLOSE = 0, 1, 2, 3
# control block (int64): [0] command sequence number, [1] command, [2 + w] sequence number worker w has completed
CTL_SEQ, CTL_CMD, CTL_DONE0 = 0, 1, 2


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n

    def __getitem__(self, i):
        return self


class VectorizedSingles:
    """k single environments behind the gymnasium vector contract: autoreset (the observation returned at the end of an episode is the
    first one of the next; the finished episode's return goes to ``info["final_info"][i]["episode"]["r"]``), episode statistics over the
    UNCLIPPED rewards and sign-clipped rewards out — the order of the reference's wrapper list (atari_wrappers.py:61-68: statistics inside,
    clipping outside) — plus the OR of the per-env ``life_loss`` flags the reference's EpisodicLifeEnv reports (atari_wrappers.py:35-51)."""

    def __init__(self, envs, clip_reward: bool = True):
        self.envs, self.clip = list(envs), clip_reward
        self.returns = np.zeros(len(self.envs), dtype=np.float64)

    def reset(self, **kw):
        obs = [e.reset(**kw)[0] for e in self.envs]
        self.returns[:] = 0
        return np.stack(obs), {}

    def step(self, actions):
        k = len(self.envs)
        obs, rew, term, trunc, life = [None] * k, np.zeros(k, np.float64), np.zeros(k, bool), np.zeros(k, bool), np.zeros(k, bool)
        final = [None] * k
        for i, (env, a) in enumerate(zip(self.envs, actions)):
            o, r, te, tr, info = env.step(int(a))
            self.returns[i] += r
            rew[i], term[i], trunc[i], life[i] = r, te, tr, bool(info.get("life_loss", False))
            if te or tr:
                final[i] = {"episode": {"r": np.array([self.returns[i]], dtype=np.float32)}}
                self.returns[i] = 0
                o, _ = env.reset()
            obs[i] = o
        info = {"life_loss": life}
        if any(f is not None for f in final):
            info["final_info"] = np.array(final, dtype=object)
            info["_final_info"] = np.array([f is not None for f in final])
        return np.stack(obs), (np.sign(rew) if self.clip else rew), term, trunc, info

    def close(self):
        for e in self.envs:
            if hasattr(e, "close"):
                e.close()


class HostSynthSlice:
    """``make_slice`` of a HOST synthetic env with the shapes and rates of the device one (84x84 frame stack, rewards P = 0.05 / 0.05 / 0.9,
    terminal 1/500, life loss 1/200; frames drawn from a small pre-generated bank): stands in for ALE when measuring the front-end's
    PCIe-inclusive throughput (tools/bench_host_env.py).  Not byte-compatible with the device env — parity tests use the oracle's twin."""

    def __init__(self, seed: int = 42, bank: int = 32):
        self.seed, self.bank = seed, bank

    def __call__(self, e0: int, k: int):
        return _HostSynthEnv(e0, k, self.seed, self.bank)


class _HostSynthEnv:
    def __init__(self, e0, k, seed, bank):
        self.k = k
        self.rng = np.random.default_rng([seed, e0])
        self.frames = self.rng.integers(0, 256, (bank, 84, 84), dtype=np.uint8) * (self.rng.random((bank, 84, 84)) < 0.25)
        self.obs = np.zeros((k, 4, 84, 84), dtype=np.uint8)
        self.ret = np.zeros(k, dtype=np.float32)

    def reset(self, **kw):
        self.obs[:] = self.frames[self.rng.integers(0, len(self.frames), self.k)][:, None]
        self.ret[:] = 0
        return self.obs.copy(), {}

    def step(self, action):
        k, rng = self.k, self.rng
        new = self.frames[rng.integers(0, len(self.frames), k)]
        u = rng.random(k)
        rew = np.where(u < 0.05, -1.0, np.where(u < 0.10, 1.0, 0.0))
        term = rng.random(k) < 1 / 500
        life = (~term) & (rng.random(k) < 1 / 200)
        self.ret += rew
        self.obs[:, :3] = self.obs[:, 1:]
        self.obs[:, 3] = new
        self.obs[term] = new[term][:, None]
        info = {"life_loss": life}
        if term.any():
            fi = np.empty(k, dtype=object)
            for i in np.nonzero(term)[0]:
                fi[i] = {"episode": {"r": np.array([self.ret[i]], dtype=np.float32)}}
            info["final_info"], info["_final_info"] = fi, term.copy()
            self.ret[term] = 0
        return self.obs.copy(), rew, term, np.zeros(k, bool), info

    def close(self):
        pass


def _record(buf, half, lo, k, obs, reward, terminated, truncated, info):
    """One slice's step result into the shared buffers (views into shared memory)."""
    buf["obs"][half, lo:lo + k] = np.asarray(obs, dtype=np.uint8).reshape(k, -1)
    sc = buf["scal"][half]
    sc[0, lo:lo + k] = np.asarray(reward, dtype=np.float32)
    sc[1, lo:lo + k] = np.asarray(terminated, dtype=np.float32)
    sc[2, lo:lo + k] = np.asarray(truncated, dtype=np.float32)
    sc[3, lo:lo + k] = np.asarray(info["life_loss"], dtype=np.float32) if "life_loss" in info else 0.0
    sc[4, lo:lo + k] = 0.0
    sc[5, lo:lo + k] = 0.0
    if "final_info" in info:
        mask = np.asarray(info["_final_info"], dtype=bool)
        sc[4, lo:lo + k] = mask.astype(np.float32)
        for i in np.nonzero(mask)[0]:
            sc[5, lo + i] = float(info["final_info"][i]["episode"]["r"][0])


def _views(obs_t, scal_t, act_t, ctl_t):
    return {"obs": obs_t.numpy(), "scal": scal_t.numpy(), "act": act_t.numpy(), "ctl": ctl_t.numpy()}


def _worker_main(w, make_slice, lo, k, obs_t, scal_t, act_t, ctl_t, spin_us):
    torch.set_num_threads(1)
    buf = _views(obs_t, scal_t, act_t, ctl_t)
    ctl = buf["ctl"]
    env = make_slice(lo, k)
    seen = 0
    try:
        while True:
            while int(ctl[CTL_SEQ]) == seen:                 # the sequence number arrives by DMA (steps) or from the parent (reset / close)
                time.sleep(spin_us * 1e-6)
            seen = int(ctl[CTL_SEQ])
            cmd = int(ctl[CTL_CMD])
            if cmd == CMD_CLOSE:
                break
            half = seen & 1
            if cmd == CMD_RESET:
                obs, _ = env.reset()
                buf["obs"][half, lo:lo + k] = np.asarray(obs, dtype=np.uint8).reshape(k, -1)
            else:
                _record(buf, half, lo, k, *env.step(buf["act"][lo:lo + k].copy()))
            ctl[CTL_DONE0 + w] = seen
    finally:
        env.close()
        ctl[CTL_DONE0 + w] = -1


class HostEnvPool:
    def __init__(self, make_slice: Callable[[int, int], object], num_envs: int, obs_shape=(4, 84, 84), action_dim: int = 4, num_workers: int = 4, ops=None,
                 start_method: str = "spawn", spin_us: float = 20.0, has_life_loss: bool = True):
        """``make_slice(e0, k)`` -> vector env over envs [e0, e0 + k) (must be picklable for worker processes).  ``num_workers = 0`` steps
        the whole vector env in this process (same buffers and copy stream; the action then has to be waited for here)."""
        if ops is None:
            from agent0_amd.ops import HipOps
            ops = HipOps()
        self.ops, self.E, self.W = ops, int(num_envs), int(num_workers)
        self.obs_shape = tuple(int(v) for v in obs_shape)
        self.obs_bytes = int(np.prod(self.obs_shape))
        self.action_dim = int(action_dim)
        self.observation_space = _Space(shape=(self.E,) + self.obs_shape)
        self.action_space = _Space(n=self.action_dim)
        self.has_life_loss = has_life_loss
        E = self.E
        # ---- shared, page-locked host block: observations and scalars double-buffered by step parity, actions, control words
        self._obs_h = torch.zeros(2, E, self.obs_bytes, dtype=torch.uint8).share_memory_()
        self._scal_h = torch.zeros(2, 6, E, dtype=torch.float32).share_memory_()
        self._act_h = torch.zeros(E, dtype=torch.int32).share_memory_()
        self._ctl_h = torch.zeros(CTL_DONE0 + max(self.W, 1), dtype=torch.int64).share_memory_()
        self._pinned = []
        rt = torch.cuda.cudart()
        for t in (self._obs_h, self._scal_h, self._act_h, self._ctl_h):
            err = rt.cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)
            if int(err) != 0:
                raise RuntimeError(f"hipHostRegister failed ({err}): the env pool needs page-locked shared memory for its DMA ring")
            self._pinned.append(t)
        self._np = _views(self._obs_h, self._scal_h, self._act_h, self._ctl_h)
        # ---- device side
        self._obs = [ops.zeros(E * self.obs_bytes, dtype=torch.uint8), ops.zeros(E * self.obs_bytes, dtype=torch.uint8)]
        self._scal_d = [ops.zeros(6, E), ops.zeros(6, E)]
        self._seq_d = torch.zeros(2, dtype=torch.int64, device=ops.device)          # [seq, CMD_STEP]: DMA-copied over ctl[0:2] behind the actions
        self._seq_d[1] = CMD_STEP
        self.copy_stream = torch.cuda.Stream()
        self._uploaded = torch.cuda.Event()
        self.seq = 0
        self.g = 0
        self.pcie_bytes_per_step = E * (self.obs_bytes + 6 * 4 + 4) + 16
        # ---- workers
        self._procs, self._local = [], None
        bounds = [round(i * E / max(self.W, 1)) for i in range(max(self.W, 1) + 1)]
        self._slices = [(bounds[i], bounds[i + 1] - bounds[i]) for i in range(max(self.W, 1))]
        if self.W == 0:
            self._local = make_slice(0, E)
        else:
            import multiprocessing as mp
            ctx = mp.get_context(start_method)
            for w, (lo, k) in enumerate(self._slices):
                p = ctx.Process(target=_worker_main, args=(w, make_slice, lo, k, self._obs_h, self._scal_h, self._act_h, self._ctl_h, spin_us), daemon=True)
                p.start()
                self._procs.append(p)

    # ------------------------------------------------------------------ host-side handshake
    def _post(self, cmd: int):
        """reset / close: the parent writes the command itself (no device work involved)."""
        self.seq += 1
        self._np["ctl"][CTL_CMD] = cmd
        self._np["ctl"][CTL_SEQ] = self.seq

    def _wait_workers(self, timeout_s: float = 120.0):
        ctl, t0 = self._np["ctl"], time.time()
        while True:
            done = ctl[CTL_DONE0:CTL_DONE0 + self.W]
            if (done == self.seq).all():
                return
            if (done < 0).any():
                raise RuntimeError("an env worker process died")
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"env workers did not finish step {self.seq} within {timeout_s} s (completed: {done.tolist()})")
            time.sleep(5e-6)

    def _upload(self, half: int, scalars: bool):
        """Page-locked half -> device buffers on the copy stream; the compute stream picks the result up through an event."""
        self.copy_stream.wait_stream(torch.cuda.current_stream())      # kernels still reading this half of the DEVICE ring (n-step / replay insert of two steps ago)
        with torch.cuda.stream(self.copy_stream):
            self._obs[half].copy_(self._obs_h[half].view(-1), non_blocking=True)
            if scalars:
                self._scal_d[half].copy_(self._scal_h[half], non_blocking=True)
            self._uploaded.record(self.copy_stream)
        torch.cuda.current_stream().wait_event(self._uploaded)

    # ------------------------------------------------------------------ env interface
    def reset(self, **kw):
        # nothing may still be reading the ring: a reset is outside the step path
        torch.cuda.current_stream().synchronize()
        if self.W == 0:
            self.seq += 1
            obs, _ = self._local.reset(**kw)
            self._np["obs"][self.seq & 1] = np.asarray(obs, dtype=np.uint8).reshape(self.E, -1)
        else:
            self._post(CMD_RESET)
            self._wait_workers()
        self._seq_d[0] = self.seq
        self.g = 0
        half = self.seq & 1
        self._upload(half, scalars=False)
        return self._obs[half], {}

    def step(self, action: torch.Tensor, final_mask: Optional[torch.Tensor] = None, final_ret: Optional[torch.Tensor] = None, ctrl=None):
        self.seq += 1
        self.g += 1
        half = self.seq & 1
        cur = torch.cuda.current_stream()
        # actions, then (sequence number, command): two DMA copies in stream order — workers that see the new number see the actions
        self._act_h.copy_(action, non_blocking=True)
        self._seq_d[0:1].add_(1)
        self._ctl_h[CTL_SEQ:CTL_CMD + 1].copy_(self._seq_d, non_blocking=True)
        if self.W == 0:
            ev = torch.cuda.Event()
            ev.record(cur)
            ev.synchronize()                     # in-process stepping has to wait for the action here (worker mode: the workers poll instead)
            _record(self._np, half, 0, self.E, *self._local.step(self._np["act"].copy()))
        else:
            self._wait_workers()                 # CPU work only: the env steps themselves
        self._upload(half, scalars=True)
        sc = self._scal_d[half]
        if final_mask is not None:
            final_mask.copy_(sc[4], non_blocking=True)
            final_ret.copy_(sc[5], non_blocking=True)
        info = {"final_mask": sc[4] if final_mask is None else final_mask, "final_ret": sc[5] if final_ret is None else final_ret}
        if self.has_life_loss:
            info["life_loss"] = sc[3]
        return self._obs[half], sc[0], sc[1], sc[2], info

    def close(self):
        if self._local is not None:
            self._local.close()
            self._local = None
        if self._procs:
            self._post(CMD_CLOSE)
            for p in self._procs:
                p.join(timeout=10)
                if p.is_alive():
                    p.terminate()
            self._procs = []
        rt = torch.cuda.cudart()
        for t in self._pinned:
            rt.cudaHostUnregister(t.data_ptr())
        self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 — interpreter shutdown
            pass
