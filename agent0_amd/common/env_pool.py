"""Host front-end for CPU environments: worker processes step slices of the vector env and write straight into a
double-buffered, page-locked uint8 ring that the GPU reads by DMA.

What it replaces: the reference's ``make_atari`` returns a gymnasium AsyncVectorEnv (agent0/common/atari_wrappers.py:59-69) whose
``step`` pickles the (E,4,84,84) observation batch through pipes into the actor process, which then copies it to the GPU with a blocking
``torch.from_numpy(obs).to(device)`` (agent0/deepq/agent.py:27).  Here

  * every worker owns envs [e0, e0 + k) and writes observations, rewards, done flags and finished-episode records into shared memory
    that is registered with the HIP runtime as page-locked (``hipHostRegister``), one half of a two-deep ring per step — there is no
    pickling and no staging copy, and the half the GPU may still be reading is never the one being written;
  * actions travel the other way without a host-side wait: the actor's action tensor and then a step sequence number are written into the
    same shared block in stream order (``a0_env_pool_send``: two tiny kernels storing through the block's device address), and the workers
    poll the sequence number — the Python thread never calls ``.cpu()`` or ``stream.synchronize()`` on the step path; it only waits for
    its own workers (CPU work) before enqueueing the next upload;
  * an upload is ONE library call (``a0_env_pool_upload``: the copies and the frame-stack kernel) on the caller's stream — the encoder waits
    for it anyway, and the bookkeeping of the previous step is enqueued behind the actions (``Actor._rollout_host``), so it runs while the
    workers step.  (The per-copy torch path — whole-stack mode, resets, ``A0_ENV_POOL_CALLS=0`` — uploads on a copy stream.)
  * the frame stack lives on the DEVICE (``newest_frame=True``, the default for stacked observations): a worker marks every env whose
    stack merely advanced — its older frames are byte-for-byte the previous observation's newer ones — and for those only the newest
    frame crosses PCIe (7 KB instead of 28 KB per env and step); ``a0_env_frame_stack`` rebuilds the stack from the previous observation
    the device already holds.  Envs whose stack changed in any other way (episode boundary, the extra presses after a lost life) are
    uploaded whole, so the device observation is bit-identical to the host one for any wrapper order.

Interface: the one ``Actor`` consumes (DeviceSynthVecEnv): ``reset() -> (obs, info)``, ``step(action, final_mask, final_ret, ctrl) ->
(obs, reward, terminal, truncated, info)`` with device tensors; ``info["life_loss"]``, ``info["final_mask"]``, ``info["final_ret"]`` carry
what the reference reads from ``info["life_loss"]`` and ``info["final_info"]`` / ``info["_final_info"]`` (agent.py:55-62,85-88).

A slice is any object with the gymnasium VECTOR contract over its k envs (``reset() -> (obs[k,...], info)``, ``step(a[k]) -> (obs,
reward, terminated, truncated, info)`` with autoreset and the same info keys); ``VectorizedSingles`` builds one from single envs, doing
the autoreset, episode statistics and reward clipping the reference gets from gymnasium's wrappers.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from multiprocessing import shared_memory
from typing import Callable, Optional

import numpy as np
import torch

from agent0_amd._abi import check
from agent0_amd.ops import _stream

from .host_envs import (CMD_CLOSE, CMD_RESET, CMD_STEP, CTL_ARG, CTL_DONE0, CTL_WORD, N_SCAL, HostSynthSlice, OffsetSlices, VectorizedSingles, block_layout,  # noqa: F401
                        block_views, ctl_word, record, worker_main)


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n

    def __getitem__(self, i):
        return self


class HostEnvPool:
    def __init__(self, make_slice: Callable[[int, int], object], num_envs: int, obs_shape=(4, 84, 84), action_dim: int = 4, num_workers: int = 4, ops=None,
                 start_method: str = "spawn", spin_us: float = 20.0, has_life_loss: bool = True, newest_frame: Optional[bool] = None,
                 busy_us: float = 500.0, inline_upload: bool = True):
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
        # device frame stack: possible when the leading axis stacks frames of a multiple of 16 bytes
        self.nstack = self.obs_shape[0] if len(self.obs_shape) > 1 else 1
        fb = self.obs_bytes // self.nstack
        can_stack = self.nstack > 1 and fb % 16 == 0
        if newest_frame and not can_stack:
            raise ValueError(f"newest_frame needs stacked observations with frames of a multiple of 16 bytes, got shape {self.obs_shape}")
        self.newest_frame = can_stack if newest_frame is None else bool(newest_frame)
        self.frame_bytes = fb if self.newest_frame else 0
        self.full_uploads = 0                 # envs whose whole stack crossed PCIe on the step path (device frame-stack mode)
        # ---- ONE shared, page-locked host block (multiprocessing.shared_memory: the workers map it by name and need no torch):
        # observations and scalars double-buffered by step parity, actions, control words
        o_obs, o_scal, o_act, o_ctl, o_new, total = block_layout(E, self.obs_bytes, self.W, self.frame_bytes)
        self._shm = shared_memory.SharedMemory(create=True, size=total)
        self._np = block_views(self._shm.buf, E, self.obs_bytes, self.W, self.frame_bytes)
        for v in self._np.values():
            v[...] = 0
        whole = torch.frombuffer(self._shm.buf, dtype=torch.uint8)
        self._whole = whole
        self._obs_h = whole[o_obs:o_obs + 2 * E * self.obs_bytes].view(2, E, self.obs_bytes)
        self._scal_h = whole[o_scal:o_scal + 2 * N_SCAL * E * 4].view(torch.float32).view(2, N_SCAL, E)
        self._new_h = whole[o_new:o_new + 2 * E * self.frame_bytes].view(2, E * self.frame_bytes)
        self._act_h = whole[o_act:o_act + 4 * E].view(torch.int32)
        self._ctl_h = whole[o_ctl:o_ctl + 8 * (CTL_DONE0 + max(self.W, 1))].view(torch.int64)
        err = torch.cuda.cudart().cudaHostRegister(whole.data_ptr(), total, 0)
        if int(err) != 0:
            raise RuntimeError(f"hipHostRegister failed ({err}): the env pool needs page-locked shared memory for its DMA ring")
        self._registered = True
        # ---- device side
        self._obs = [ops.zeros(E * self.obs_bytes, dtype=torch.uint8), ops.zeros(E * self.obs_bytes, dtype=torch.uint8)]
        self._scal_d = [ops.zeros(N_SCAL, E), ops.zeros(N_SCAL, E)]
        self._new_d = ops.zeros(max(E * self.frame_bytes, 16), dtype=torch.uint8)
        self._seq_d = torch.zeros(1, dtype=torch.int64, device=ops.device)          # (CMD_STEP << 56) | seq: ONE word, DMA-copied over ctl[0] behind the actions
        self.copy_stream = torch.cuda.Stream()
        self._uploaded = torch.cuda.Event()
        # the step path's two PCIe legs as one library call each (a0_env_pool_upload / a0_env_pool_send): raw addresses, resolved once.  ``inline_upload``:
        # the upload is enqueued on the caller's stream (the encoder waits for it anyway), False: on the pool's copy stream, whose DMA can run beside another
        # group's inference (HostEnvGroups, A0_ENV_GROUP_COPY_STREAMS=1).  A0_ENV_POOL_CALLS=0: the per-copy torch calls (same bytes; a tuning aid)
        self.library_calls = self.newest_frame and os.environ.get("A0_ENV_POOL_CALLS", "1") != "0"
        self.inline_upload = bool(inline_upload)
        self.wait_s = 0.0                     # host time spent waiting for the workers (the env's own cost as the step path sees it)
        if self.library_calls:
            dp = C.c_void_p()
            check(ops.lib.a0_host_device_pointer(whole.data_ptr(), C.byref(dp)), "a0_host_device_pointer")
            base = dp.value - whole.data_ptr()                                        # device address of a host byte = host address + base (0 on this platform)
            self._p = dict(act_dev=self._act_h.data_ptr() + base, ctl_dev=self._ctl_h[CTL_WORD:].data_ptr() + base,
                           new_h=[self._new_h[h].data_ptr() for h in (0, 1)], scal_h=[self._scal_h[h].data_ptr() for h in (0, 1)],
                           obs_h=[self._obs_h[h].data_ptr() for h in (0, 1)], obs_d=[o.data_ptr() for o in self._obs],
                           scal_d=[t.data_ptr() for t in self._scal_d], new_d=self._new_d.data_ptr())
            self._n_whole = C.c_int(0)
        self.seq = 0
        self.g = 0
        self.pcie_bytes_per_step = E * ((self.frame_bytes or self.obs_bytes) + N_SCAL * 4 + 4) + 16      # + obs_bytes per whole-stack upload (full_uploads)
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
                p = ctx.Process(target=worker_main, args=(w, make_slice, lo, k, self._shm.name, E, self.obs_bytes, self.W, spin_us, self.frame_bytes, busy_us),
                                daemon=True)
                p.start()
                self._procs.append(p)

    # ------------------------------------------------------------------ host-side handshake
    def _post(self, cmd: int, arg: int = -1):
        """reset / close: the parent writes the command itself (no device work involved) — the argument first, then the one word that
        carries command and sequence number together."""
        self.seq += 1
        self._np["ctl"][CTL_ARG] = arg
        self._np["ctl"][CTL_WORD] = ctl_word(cmd, self.seq)

    def _wait_workers(self, timeout_s: float = 120.0):
        ctl, t0 = self._np["ctl"], time.perf_counter()
        while True:
            done = ctl[CTL_DONE0:CTL_DONE0 + self.W]
            if (done == self.seq).all():
                self.wait_s += time.perf_counter() - t0
                return
            if (done < 0).any():
                raise RuntimeError("an env worker process died")
            waited = time.perf_counter() - t0
            if waited > timeout_s:
                raise TimeoutError(f"env workers did not finish step {self.seq} within {timeout_s} s (completed: {done.tolist()})")
            if waited > 2e-3:                    # a step of a real emulator takes longer than this: stop burning the core (a sleep costs ~60 us of timer slack)
                time.sleep(2e-5)

    def _upload(self, half: int, scalars: bool):
        """Page-locked half -> device buffers: one ``a0_env_pool_upload`` call on the caller's stream (or the pool's copy stream, ``inline_upload=False``) on the step
        path; the per-copy form below for resets (``scalars`` False: whole stacks, nothing else), whole-stack mode and ``A0_ENV_POOL_CALLS=0`` runs on the copy
        stream, and the compute stream picks the result up through an event."""
        if scalars and self.library_calls:
            p = self._p
            if self.inline_upload:
                stream = _stream()
            else:
                cur = torch.cuda.current_stream()
                self.copy_stream.wait_stream(cur)
                stream = self.copy_stream.cuda_stream
            check(self.ops.lib.a0_env_pool_upload(p["new_h"][half], p["new_d"], p["scal_h"][half], p["scal_d"][half], N_SCAL, 6, p["obs_h"][half],
                                                  p["obs_d"][half ^ 1], p["obs_d"][half], self.E, self.nstack, self.frame_bytes, C.byref(self._n_whole), stream),
                  "a0_env_pool_upload")
            self.full_uploads += self._n_whole.value
            if not self.inline_upload:
                self._uploaded.record(self.copy_stream)
                cur.wait_event(self._uploaded)
            return
        self.copy_stream.wait_stream(torch.cuda.current_stream())      # kernels still reading this half of the DEVICE ring (n-step / replay insert of two steps ago)
        with torch.cuda.stream(self.copy_stream):
            if scalars and self.newest_frame:
                self._new_d.copy_(self._new_h[half], non_blocking=True)
                self._scal_d[half].copy_(self._scal_h[half], non_blocking=True)
                # whole stacks for the envs that did not merely advance (host memory: the workers have finished); adjacent envs in one copy
                whole = np.flatnonzero(self._np["scal"][half, 6] == 0.0)
                self.full_uploads += len(whole)
                dst, src, ob = self._obs[half], self._obs_h[half].view(-1), self.obs_bytes
                i = 0
                while i < len(whole):
                    j = i
                    while j + 1 < len(whole) and whole[j + 1] == whole[j] + 1:
                        j += 1
                    a, b = int(whole[i]) * ob, (int(whole[j]) + 1) * ob
                    dst[a:b].copy_(src[a:b], non_blocking=True)
                    i = j + 1
                self.ops.env_frame_stack(self._obs[half ^ 1], self._new_d, self._scal_d[half][6], self._obs[half], self.E, self.nstack, self.frame_bytes)
            else:
                self._obs[half].copy_(self._obs_h[half].view(-1), non_blocking=True)
                if scalars:
                    self._scal_d[half].copy_(self._scal_h[half], non_blocking=True)
            self._uploaded.record(self.copy_stream)
        torch.cuda.current_stream().wait_event(self._uploaded)

    # ------------------------------------------------------------------ env interface
    def reset(self, **kw):
        # nothing may still be reading the ring: a reset is outside the step path
        torch.cuda.current_stream().synchronize()
        seed = kw.get("seed")
        if self.W == 0:
            self.seq += 1
            obs, _ = self._local.reset(**kw)
            self._np["obs"][self.seq & 1] = np.asarray(obs, dtype=np.uint8).reshape(self.E, -1)
        else:
            if set(kw) - {"seed"}:
                raise TypeError(f"HostEnvPool.reset: only `seed` reaches the worker processes, got {sorted(kw)}")
            self._post(CMD_RESET, -1 if seed is None else int(seed))      # worker w resets its slice [lo, lo + k) with seed + lo
            self._wait_workers()
            # the acknowledged word again, as a STEP word, written by the host in one store: from here on the DMA-written step words only ever
            # change sequence-number bytes, so a copy that lands in pieces cannot show a worker the next number beside the old CMD_RESET
            # (host_envs.accept_word)
            self._np["ctl"][CTL_WORD] = ctl_word(CMD_STEP, self.seq)
        self._seq_d[0] = ctl_word(CMD_STEP, self.seq)
        self.g = 0
        half = self.seq & 1
        self._upload(half, scalars=False)
        return self._obs[half], {}

    def step(self, action: torch.Tensor, final_mask: Optional[torch.Tensor] = None, final_ret: Optional[torch.Tensor] = None, ctrl=None):
        self.step_send(action)
        return self.step_recv(final_mask, final_ret)

    def step_send(self, action: torch.Tensor):
        """First half of ``step``: hand the actions to the workers (``a0_env_pool_send``: actions, then the step word, in stream order) and return at once — the workers step while the caller
        does something else (``HostEnvGroups``: the other group's inference).  In-process stepping (no workers) does the env step here."""
        self.seq += 1
        self.g += 1
        half = self.seq & 1
        # actions, then the command word (CMD_STEP and the sequence number in ONE 8-byte word): two kernels (or two DMA copies) in stream order — workers
        # that see the new number see the actions, and can never pair it with the previous command
        if self.library_calls and action.dtype == torch.int32 and action.is_contiguous() and action.numel() == self.E:
            check(self.ops.lib.a0_env_pool_send(action.data_ptr(), self._p["act_dev"], self.E, self._p["ctl_dev"], ctl_word(CMD_STEP, self.seq), _stream()), "a0_env_pool_send")
        else:
            self._seq_d.fill_(ctl_word(CMD_STEP, self.seq))
            self._act_h.copy_(action, non_blocking=True)
            self._ctl_h[CTL_WORD:CTL_WORD + 1].copy_(self._seq_d, non_blocking=True)
        if self.W == 0:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            ev.synchronize()                     # in-process stepping has to wait for the action here (worker mode: the workers poll instead)
            record(self._np, half, 0, self.E, *self._local.step(self._np["act"].copy()))

    def step_recv(self, final_mask: Optional[torch.Tensor] = None, final_ret: Optional[torch.Tensor] = None):
        """Second half of ``step``: wait for the workers (CPU work only), enqueue the upload, return the device tensors."""
        half = self.seq & 1
        if self.W != 0:
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
        if getattr(self, "_np", None) is None and not self._procs and self._local is None:
            return
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
        if getattr(self, "_registered", False):
            torch.cuda.current_stream().synchronize()
            self.copy_stream.synchronize()
            torch.cuda.cudart().cudaHostUnregister(self._whole.data_ptr())
            self._registered = False
            self._np = None
            del self._obs_h, self._scal_h, self._act_h, self._ctl_h, self._new_h, self._whole
            try:
                self._shm.close()
                self._shm.unlink()
            except (BufferError, FileNotFoundError):
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 — interpreter shutdown
            pass


class HostEnvGroups:
    """The vector env as ``groups`` consecutive ranges of envs, each a ``HostEnvPool`` of its own (own workers, own page-locked ring), so that
    the actor can step one group on the CPU while the GPU infers the other — what the reference gets from running ``num_actors`` actor processes beside each
    other (agent0/deepq/launch.py:30-61, 166-172), without giving up the single replay ring and its transition order: the actor writes group g's transitions of
    step t to the slots a one-group rollout would have used (``Actor._rollout_groups``), so the ring holds the same bytes in the same order.
    Worker processes are divided between the groups (at least one each)."""

    def __init__(self, make_slice: Callable[[int, int], object], num_envs: int, groups: int = 2, num_workers: int = 4, ops=None, **kw):
        if ops is None:
            from agent0_amd.ops import HipOps
            ops = HipOps()
        groups = int(groups)
        if groups < 2 or num_envs < groups:
            raise ValueError("HostEnvGroups: at least two groups of at least one env each")
        self.ops, self.E = ops, int(num_envs)
        bounds = [round(i * self.E / groups) for i in range(groups + 1)]
        per = 0 if num_workers == 0 else max(1, int(num_workers) // groups)
        self.offsets = bounds[:-1]
        # every group's upload goes on the caller's stream: the Python thread bounds a grouped rollout, and a copy stream per group costs it three stream calls per
        # group-step for a DMA overlap the GPU does not need (1.16 - 1.22 M env-frames/s against 0.94 - 1.00 M, profiles/r04_experiments.md).  A0_ENV_GROUP_COPY_STREAMS=1: copy streams
        inline = os.environ.get("A0_ENV_GROUP_COPY_STREAMS", "0") != "1"
        self.pools = [HostEnvPool(OffsetSlices(make_slice, bounds[i]), bounds[i + 1] - bounds[i], num_workers=per, ops=ops, inline_upload=inline, **kw) for i in range(groups)]
        p0 = self.pools[0]
        self.obs_shape, self.obs_bytes, self.action_dim = p0.obs_shape, p0.obs_bytes, p0.action_dim
        self.observation_space = _Space(shape=(self.E,) + self.obs_shape)
        self.action_space = _Space(n=self.action_dim)
        self.has_life_loss = p0.has_life_loss

    @property
    def full_uploads(self):
        return sum(p.full_uploads for p in self.pools)

    @property
    def pcie_bytes_per_step(self):
        return sum(p.pcie_bytes_per_step for p in self.pools)

    @property
    def newest_frame(self):
        return self.pools[0].newest_frame

    def reset(self, **kw):
        seed = kw.get("seed")
        out = []
        for p, off in zip(self.pools, self.offsets):
            out.append(p.reset(**({} if seed is None else {"seed": int(seed) + off}))[0])
        return out, {}

    def close(self):
        for p in self.pools:
            p.close()
