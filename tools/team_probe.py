#!/usr/bin/env python3
"""Team form of the fused kernel against the classic one-wavefront-per-problem form on the C4 shard: bit identity
(same chunk count on both sides) and device time, over owners per compute unit / chunk counts / priority schemes.
   python tools/team_probe.py [P] [arm ...]     arm = owners:chunks:prio, e.g. 2:6:0   (default: a small sweep)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tinyopt_amd as ta


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
    arms = sys.argv[2:] or ["2:6:0", "2:8:0", "2:5:0", "2:12:0", "1:6:0", "3:6:0", "4:6:0", "2:6:1", "2:4:0"]
    n, m = 50, 2000
    from tinyopt_amd.api import default_context
    ctx = default_context()
    model, x0, _ = ta.DenseRow.synthetic(P, n, m, torch.float32)
    opts = ta.Options.benchmark()
    x = x0.clone()
    out = ta.Optimize(x, model, opts)

    def solve(**tune):
        with ctx.tuning(**tune):
            x.copy_(x0)
            out.counters.zero_()
            ta.Optimize(x, model, opts, out=out)
            torch.cuda.synchronize()
            res = (x.clone(), out.stop_reason.clone(), out.num_iters.clone(), out.final_cost.clone(), out.counters.clone())

            def run():
                x.copy_(x0)
                ta.Optimize(x, model, opts, out=out)
            t = timed(run)
        return res, t

    for arm in arms:
        no, k, prio = (int(v) for v in arm.split(":"))
        ref, t_ref = solve(coop_chunks=k)
        got, t_team = solve(team_on=1, team_owners=no, coop_chunks=k, team_prio=prio)
        same = all(torch.equal(a, b) for a, b in zip(ref[:4], got[:4]))
        cnt_r, cnt_t = ref[4].cpu().numpy(), got[4].cpu().numpy()
        its = int(ref[2].sum().item())
        print(f"P={P} owners/CU={no} chunks={k} prio={prio}: classic {t_ref[0]:.3f} ms (median {t_ref[1]:.3f}) = {its / t_ref[0] / 1e3:.2f} M it/s | "
              f"team {t_team[0]:.3f} ms (median {t_team[1]:.3f}) = {its / t_team[0] / 1e3:.2f} M it/s | "
              f"bit-identical={same} counters classic={cnt_r[:5].tolist()} team={cnt_t[:5].tolist()}", flush=True)
    base, t_base = solve()
    print(f"P={P} classic, its own chunking (K = 2): {t_base[0]:.3f} ms (median {t_base[1]:.3f}) = {int(base[2].sum().item()) / t_base[0] / 1e3:.2f} M it/s")


if __name__ == "__main__":
    main()
