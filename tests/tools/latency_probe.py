"""Single-problem latency of the row-split path (BASELINE configs C2 and C5): whole solve from x0, device time by HIP events."""
import torch, time, numpy as np, sys
sys.path.insert(0,".")
import tinyopt_amd as ta
from oracle import pyoracle
def t(fn, reps=30):
    fn(); torch.cuda.synchronize(); ts=[]; ev=[]
    for _ in range(reps):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        t0=time.perf_counter(); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0); ev.append(e0.elapsed_time(e1)*1e3)
    return min(ts)*1e6, min(ev)
model,x0,_ = ta.DenseRow.synthetic(1,6,1000,torch.float64)
o=ta.Options.benchmark()
x=x0.clone(); out=ta.Optimize(x,model,o); torch.cuda.synchronize()
def run(): x.copy_(x0); ta.Optimize(x,model,o,out=out)
print("C2: host+device %.0f us, device(events) %.0f us"%t(run))
data,p0,ps=pyoracle.synth_se3_reproj(1,25000,np.float64)
m5=ta.SE3Reproj(torch.from_numpy(data).cuda(),25000); xp0=torch.from_numpy(p0).cuda(); xp=xp0.clone()
out5=ta.Optimize(xp,m5,o); torch.cuda.synchronize()
def run5(): xp.copy_(xp0); ta.Optimize(xp,m5,o,out=out5)
print("C5: host+device %.0f us, device(events) %.0f us"%t(run5))

for splits, name in ((1, "one chunk"), (None, "auto")):
    def r2(): x.copy_(x0); ta.Optimize(x, model, o, out=out, splits=splits)
    def r5(): xp.copy_(xp0); ta.Optimize(xp, m5, o, out=out5, splits=splits)
    print(name, "C2 %.0f/%.0f us" % t(r2), "C5 %.0f/%.0f us" % t(r5), "iters", int(out.num_iters[0]), int(out5.num_iters[0]))
