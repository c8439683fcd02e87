"""GPU micro-benchmark (diagnostics): the c51 learner's head stage at B = 512 — reduce_bias_act_multi, the online head GEMM over 2B rows, the target head GEMM
over B rows, a0_c51_head_loss_slabs — each alone and together (graph replays, HIP events).  Knobs: A0_FC1_VARIANT, A0_FC1_WGS (csrc/net.hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agent0_amd.ops import HipOps
hip = HipOps()


def timeit(run, n=200):
    for _ in range(10): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): run()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n // 20): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


B, A, T, ld = 512, 4, 51, 256
dev = "cuda"
h_on = torch.randn(2 * B * 512, device=dev).abs(); h_tg = torch.randn(B * 512, device=dev).abs()
Wo = torch.randn(ld * 512, device=dev) * 0.05; Wt = torch.randn(ld * 512, device=dev) * 0.05
bo = torch.zeros(ld, device=dev); bt = torch.zeros(ld, device=dev)
ns_on, ns_tg = hip.dense_fwd_partial_slabs(2 * B, ld, 512), hip.dense_fwd_partial_slabs(B, ld, 512)
s_on = torch.empty(ns_on * 2 * B * ld, device=dev); s_tg = torch.empty(ns_tg * B * ld, device=dev)
act = torch.randint(0, A, (B,), dtype=torch.int32, device=dev); rew = torch.randn(B, device=dev).sign(); done = torch.zeros(B, device=dev); wgt = torch.ones(B, device=dev)
atoms = torch.linspace(-10, 10, T, device=dev)
loss, draw, state = torch.empty(B, device=dev), torch.empty(B * ld, device=dev), torch.zeros(8, dtype=torch.int32, device=dev)
q1, q2, m, a_s = torch.empty(B * A * T, device=dev), torch.empty(B * A * T, device=dev), torch.empty(B * T, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
ns1 = hip.dense_fwd_partial_slabs(B, 512, 3136)
fc = [torch.randn(ns1 * B * 512, device=dev) for _ in range(3)]
b1 = torch.zeros(512, device=dev)


def gemm_on(): hip.dense_fwd_partial(h_on, 512, Wo, 2 * B, ld, 512, s_on)
def gemm_tg(): hip.dense_fwd_partial(h_tg, 512, Wt, B, ld, 512, s_tg)
def tail(full=True):
    hip.c51_head_loss_slabs(s_on, ns_on, 2 * B, s_tg, ns_tg, B, bo, bt, ld, A, T, True, act, rew, done, wgt, atoms, 0.97, -10.0, 10.0, B, loss, draw, state,
                            q_on=q1 if full else None, q_tg=q2 if full else None, m_out=m if full else None, a_star=a_s)
def rba(): hip.reduce_bias_act_multi([(fc[0], ns1, b1, h_on[: B * 512], B), (fc[1], ns1, b1, h_tg, B), (fc[2], ns1, b1, h_on[B * 512:], B)], 512, True)
def both():
    rba(); gemm_on(); gemm_tg(); tail()


tag = f"variant={os.environ.get('A0_FC1_VARIANT', '-')} wgs={os.environ.get('A0_FC1_WGS', '256')} slabs on/tg {ns_on}/{ns_tg}"
print(f"{tag}: reduce x3 {timeit(rba):.2f} | head GEMM 2B rows {timeit(gemm_on):.2f} | B rows {timeit(gemm_tg):.2f} | tail {timeit(tail):.2f} "
      f"(no optional outputs {timeit(lambda: tail(False)):.2f}) | all four {timeit(both):.2f} us", flush=True)
