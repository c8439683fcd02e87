"""GPU micro-benchmark: the quantile-regression learner's loss stage at B = 512, A = 4, N = N' = 200 (BASELINE's qr variant: 512 x 200 x 200 = 20.5 M quantile pairs,
reference agent.py:110-114,272-293) — the stand-alone a0_quantile_huber_kernel and a0_qr_head_loss_slabs (head slabs -> loss + head gradient), graph replays timed with
HIP events, against the fp32 vector-unit bound of the pair sweep:

    20.48 M pairs x 8 vector instructions (sub, cmp, cndmask, min, fma, mul, med3, pk_fma: csrc/loss.hip::a0_qh_sweep) / 64 lanes
    = 2.56 M wave-instructions over 1024 SIMDs, one per 2 cycles per SIMD at two or more waves per SIMD (MI355X_MICROARCH.md) at 2.4 GHz  ->  2.1 us;
    thread i owns online quantile i, so 200 of a workgroup's 256 lanes work: 2.7 us.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import json
import torch
from agent0_amd.ops import HipOps
hip = HipOps()


def timeit(run, n=400):
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


B, A, T = 512, 4, 200
dev = "cuda"
out = {}
for dueling, double_q in ((False, False), (True, True)):
    NQ = A + (1 if dueling else 0)
    ld = (NQ * T + 31) // 32 * 32
    R_on = 2 * B if double_q else B
    ns_on, ns_tg = hip.dense_fwd_partial_slabs(R_on, ld, 512), hip.dense_fwd_partial_slabs(B, ld, 512)
    s_on, s_tg = torch.randn(ns_on * R_on * ld, device=dev) * 0.5, torch.randn(ns_tg * B * ld, device=dev) * 0.5
    bo, bt = torch.zeros(ld, device=dev), torch.zeros(ld, device=dev)
    act = torch.randint(0, A, (B,), dtype=torch.int32, device=dev); rew = torch.randn(B, device=dev).sign(); done = torch.zeros(B, device=dev); wgt = torch.ones(B, device=dev)
    taus = ((2 * torch.arange(T, dtype=torch.float32) + 1) / (2.0 * T)).to(dev)
    loss, draw, state = torch.empty(B, device=dev), torch.empty(B * ld, device=dev), torch.zeros(8, dtype=torch.int32, device=dev)
    q1, q2, a_s = torch.empty(B * A * T, device=dev), torch.empty(B * A * T, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    y, dq = torch.randn(B * T, device=dev), torch.zeros(B * A * T, device=dev)
    q1.normal_()

    def fused(full=True):
        hip.qr_head_loss_slabs(s_on, ns_on, R_on, s_tg, ns_tg, B if double_q else -1, bo, bt, ld, A, T, dueling, act, rew, done, wgt, taus, 0.97, B, loss, draw, state,
                               q_on=q1 if full else None, q_tg=q2 if full else None, a_star=a_s)

    def alone():
        hip.loss_quantile_huber(q1, A * T, 1, T, y, taus, 0, act, wgt, B, T, T, loss, dq, state)

    pairs = B * T * T
    bound = pairs * 8 / 64 * 2 / 1024 / 2.4e9 * 1e6
    r = {"a0_quantile_huber_kernel_us": round(timeit(alone), 2), "a0_qr_head_loss_slabs_us": round(timeit(fused), 2), "a0_qr_head_loss_slabs_no_optional_outputs_us": round(timeit(lambda: fused(False)), 2),
         "pairs": pairs, "valu_bound_us": round(bound, 2), "valu_bound_with_200_of_256_lanes_us": round(bound * 256 / 200, 2), "head_slabs_online_target": [ns_on, ns_tg], "ld": ld}
    r["quantile_huber_fraction_of_valu_bound"] = round(r["valu_bound_with_200_of_256_lanes_us"] / r["a0_quantile_huber_kernel_us"], 3)
    out["dueling_double" if dueling else "plain"] = r
    print(("dueling + double-Q" if dueling else "plain"), json.dumps(r), flush=True)
if len(sys.argv) > 1:
    json.dump({"what": "tools/ubench_quantile.py: B = 512, A = 4, N = N' = 200 (512 x 200 x 200 quantile pairs); HIP events over hipGraph replays", "cases": out}, open(sys.argv[1], "w"), indent=1)
