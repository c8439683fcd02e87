"""ctypes front-end for the C part of the oracle (oracle/_build/liba0oracle.so)
plus small numpy mirrors used to cross-check the C code itself.

TEST INFRASTRUCTURE — see oracle/__init__.py for who may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SAN = os.environ.get("A0_SANITIZE") == "1"          # tools/asan.sh: the ASan + UBSan build of the same sources (make -C oracle asan)
_SO = os.path.join(_HERE, "_build", *(("asan",) if _SAN else ()), "liba0oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("sumtree.c", "philox.c", "synth_env.c")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "asan" if _SAN else "all"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.a0o_sumtree_cap2.restype = C.c_int64
        _lib.a0o_sumtree_cap2.argtypes = [C.c_int64]
        _lib.a0o_sumtree_total.restype = C.c_float
        _lib.a0o_sumtree_find.restype = C.c_int64
        _lib.a0o_sumtree_find.argtypes = [C.c_void_p, C.c_int64, C.c_float]
        _lib.a0o_perm_index.restype = C.c_uint64
        _lib.a0o_perm_index.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------- sum-tree
class SumTree:
    def __init__(self, size: int):
        self.size = size
        self.cap2 = int(lib().a0o_sumtree_cap2(size))
        self.tree = np.zeros(2 * self.cap2, dtype=np.float32)

    def set(self, idx, val):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        val = np.ascontiguousarray(val, dtype=np.float32)
        assert idx.min() >= 0 and idx.max() < self.size
        lib().a0o_sumtree_set(_p(self.tree), C.c_int64(self.cap2), _p(idx), _p(val), C.c_int64(idx.size))

    def rebuild(self):
        lib().a0o_sumtree_rebuild(_p(self.tree), C.c_int64(self.cap2))

    @property
    def total(self) -> np.float32:
        return self.tree[1]

    def leaves(self) -> np.ndarray:
        return self.tree[self.cap2:self.cap2 + self.size]

    def find(self, u: float) -> int:
        return int(lib().a0o_sumtree_find(_p(self.tree), self.cap2, C.c_float(u)))

    def sample(self, xi):
        xi = np.ascontiguousarray(xi, dtype=np.float32)
        B = xi.size
        out_i = np.empty(B, dtype=np.int64)
        out_p = np.empty(B, dtype=np.float32)
        lib().a0o_sumtree_sample(_p(self.tree), C.c_int64(self.cap2), _p(xi), C.c_int64(B), _p(out_i), _p(out_p))
        return out_i, out_p


def sumtree_set_numpy(tree: np.ndarray, cap2: int, idx, val):
    """Level-synchronous mirror (the order the GPU kernel uses): leaves, then each level bottom-up."""
    idx = np.asarray(idx, dtype=np.int64)
    for i, v in zip(idx, np.asarray(val, dtype=np.float32)):
        tree[cap2 + i] = v
    nodes = np.unique((cap2 + idx) >> 1)
    while nodes.size and nodes[0] >= 1:
        tree[nodes] = tree[2 * nodes] + tree[2 * nodes + 1]
        if nodes[0] == 1:
            break
        nodes = np.unique(nodes >> 1)


def sumtree_find_numpy(tree: np.ndarray, cap2: int, u) -> int:
    u = np.float32(u)
    n = 1
    while n < cap2:
        left, right = tree[2 * n], tree[2 * n + 1]
        if u < left or not (right > 0):
            n = 2 * n
        else:
            u = np.float32(u - left)
            n = 2 * n + 1
    return n - cap2


def perm_batch(start: int, count: int, n: int, seed: int) -> np.ndarray:
    out = np.empty(count, dtype=np.int64)
    lib().a0o_perm_batch(C.c_uint64(start), C.c_uint64(count), C.c_uint64(n), C.c_uint32(seed), _p(out))
    return out


# ----------------------------------------------------------------------------- philox
def philox(ctr, key) -> np.ndarray:
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().a0o_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def rng_u32(seed: int, stream: int, offset: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    lib().a0o_rng_u32(C.c_uint64(seed), C.c_uint32(stream), C.c_uint64(offset), _p(out), C.c_uint64(n))
    return out


def rng_uniform(seed: int, stream: int, offset: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    lib().a0o_rng_uniform(C.c_uint64(seed), C.c_uint32(stream), C.c_uint64(offset), _p(out), C.c_uint64(n))
    return out


def rng_normal(seed: int, stream: int, offset: int, std: float, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    lib().a0o_rng_normal(C.c_uint64(seed), C.c_uint32(stream), C.c_uint64(offset), C.c_float(std), _p(out), C.c_uint64(n))
    return out


def philox_numpy(ctr, key) -> np.ndarray:
    """Independent numpy restatement (checks philox.c against itself + the KATs)."""
    c = [np.uint64(x) for x in ctr]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask, p0 & mask]
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return np.array(c, dtype=np.uint64).astype(np.uint32)


# ----------------------------------------------------------------------------- synthetic env
class SynthVecEnv:
    """CPU twin of the device synthetic env, with the vector-env API the actor consumes."""

    H = W = 84

    TASKS = {"stream": 0, "block": 1, "chase": 2}

    def __init__(self, num_envs: int, seed: int = 42, rank: int = 0, action_dim: int = 4, env_offset: int = 0, task: str = "stream"):
        """``env_offset``: this object is envs [env_offset, env_offset + num_envs) of a larger vector env (a slice owned by one worker).
        ``task``: "stream" = action-independent rewards, "block" = the learnable bandit task, "chase" = the task with temporal credit (oracle/synth_env.c)."""
        self.E, self.seed, self.rank, self.action_dim, self.e0 = num_envs, seed, rank, action_dim, env_offset
        self.task = self.TASKS[task]
        self.g = np.zeros(num_envs, dtype=np.uint32)
        self.ep_ret = np.zeros(num_envs, dtype=np.float32)
        self.obs = np.zeros((num_envs, 4, 84, 84), dtype=np.uint8)

    def reset(self):
        lib().a0o_env_reset_task_at(C.c_uint64(self.seed), C.c_uint32(self.rank), C.c_int64(self.e0), C.c_int64(self.E), C.c_int32(self.task), _p(self.g), _p(self.ep_ret),
                                    _p(self.obs))
        return self.obs.copy(), {}

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.int32)
        out = np.empty_like(self.obs)
        rew = np.empty(self.E, dtype=np.float32)
        term = np.empty(self.E, dtype=np.uint8)
        trunc = np.empty(self.E, dtype=np.uint8)
        life = np.empty(self.E, dtype=np.uint8)
        fmask = np.empty(self.E, dtype=np.uint8)
        fret = np.empty(self.E, dtype=np.float32)
        lib().a0o_env_step_task_at(C.c_uint64(self.seed), C.c_uint32(self.rank), C.c_int64(self.e0), C.c_int64(self.E), _p(a), C.c_int32(self.action_dim), C.c_int32(self.task),
                                _p(self.g), _p(self.ep_ret),
                           _p(self.obs), _p(out), _p(rew), _p(term), _p(trunc), _p(life), _p(fmask), _p(fret))
        self.obs = out
        info = {"life_loss": life.astype(bool)}
        if fmask.any():
            fi = np.empty(self.E, dtype=object)
            for i in np.nonzero(fmask)[0]:
                fi[i] = {"episode": {"r": np.array([fret[i]], dtype=np.float32)}}
            info["final_info"] = fi
            info["_final_info"] = fmask.astype(bool)
        return out.copy(), rew.astype(np.float64), term.astype(bool), trunc.astype(bool), info

    def close(self):
        pass


def env_terminals(seed: int, rank: int, E: int, steps: int, env_offset: int = 0) -> np.ndarray:
    """bool [steps, E]: env e terminates at step g = row + 1 (action-independent)."""
    out = np.empty((steps, E), dtype=np.uint8)
    lib().a0o_env_terminals(C.c_uint64(seed), C.c_uint32(rank), C.c_int64(env_offset), C.c_int64(E), C.c_int64(steps), _p(out))
    return out.astype(bool)


def env_block_target(e, g_prev, A: int):
    """The block task's rewarded action for an observation whose newest frame is frame(e, g_prev) (oracle/synth_env.c)."""
    e, g_prev = np.asarray(e, dtype=np.uint32), np.asarray(g_prev, dtype=np.uint32)
    by = (np.uint32(3) * g_prev + np.uint32(11) * e) % np.uint32(77)
    bx = (np.uint32(5) * g_prev + np.uint32(7) * e) % np.uint32(77)
    return ((2 * (by >= 39).astype(np.int64) + (bx >= 39).astype(np.int64)) % A).astype(np.int64)


def env_chase_cells(obs: np.ndarray, env_offset: int = 0) -> np.ndarray:
    """The chase task's state of every env of an observation batch [E, 4, 84, 84]: the lattice cell of the block in the newest frame (oracle/synth_env.c)."""
    obs = np.ascontiguousarray(obs, dtype=np.uint8).reshape(-1, 4, 84, 84)
    return np.array([lib().a0o_env_chase_cell(_p(np.ascontiguousarray(obs[i, 3])), C.c_uint32(env_offset + i)) for i in range(obs.shape[0])], dtype=np.int64)


def env_frame(seed: int, e: int, g: int) -> np.ndarray:
    out = np.empty(84 * 84, dtype=np.uint8)
    lib().a0o_env_frame(C.c_uint32(seed), C.c_uint32(e), C.c_uint32(g), _p(out))
    return out.reshape(84, 84)
