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
def run(): ta.Optimize(x,model,o,out=out)
print("C2: host+device %.0f us, device(events) %.0f us"%t(run))
data,p0,ps=pyoracle.synth_se3_reproj(1,25000,np.float64)
m5=ta.SE3Reproj(torch.from_numpy(data).cuda(),25000); xp=torch.from_numpy(p0).cuda()
out5=ta.Optimize(xp,m5,o); torch.cuda.synchronize()
def run5(): ta.Optimize(xp,m5,o,out=out5)
print("C5: host+device %.0f us, device(events) %.0f us"%t(run5))
