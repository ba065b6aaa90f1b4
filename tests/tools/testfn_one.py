"""Debug helper (GPU): one test-function problem, device history vs the oracle's.  usage: testfn_one.py <name> <start-index>"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tinyopt_amd as ta
from oracle import pyoracle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_testfns import CASES, _options
name, p = sys.argv[1], int(sys.argv[2])
x0 = CASES[name][0]
rng = np.random.default_rng(11)
starts = np.array(x0)[None, :] + rng.uniform(-0.3, 0.3, (256, len(x0)))
starts[0] = x0
s = starts[p:p + 1].copy()
o = _options(ta, pyoracle, name)
hs = o.max_iters + 2
ref = pyoracle.testfn_lm(name, s, o.to_pod(), hist_stride=hs)
x = torch.from_numpy(s.copy()).cuda()
out = ta.Optimize(x, ta.TestFn(name, 1), o, history=True)
torch.cuda.synchronize()
k = max(int(out.num_iters[0]), int(ref["iters"][0])) + 1
k = min(k, 12)
np.set_printoptions(precision=10, linewidth=200)
print("start", s, "gpu x", x.cpu().numpy(), "ref x", ref["x"])
print("gpu stop/iters/fails", int(out.stop_reason[0]), int(out.num_iters[0]), int(out.num_failures[0]), "ref", ref["stop"], ref["iters"], ref["fails"])
print("gpu errs", out.errs.cpu().numpy()[0, :k]); print("ref errs", ref["errs"][0, :k])
print("gpu d2  ", out.deltas2.cpu().numpy()[0, :k]); print("ref d2  ", ref["deltas2"][0, :k])
print("gpu succ", out.successes.cpu().numpy()[0, :k]); print("ref succ", ref["succ"][0, :k])
g, H, c, _ = ta.accumulate(ta.TestFn(name, 1), torch.from_numpy(s).cuda())
print("H at start", H.cpu().numpy(), "eig", np.linalg.eigvalsh(H.cpu().numpy()[0]), "g", g.cpu().numpy())
