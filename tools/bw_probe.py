import torch
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
for mb in (8,17,34,50,100,200,400,800):
    n=mb*1024*1024//4
    x=torch.randn(n,device='cuda'); y=torch.empty_like(x)
    dt=t(lambda: y.copy_(x))
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): y.copy_(x)
    dg=t(lambda: g.replay(),10)/20
    print("%4d MB copy: eager %.1f us %.2f TB/s | in-graph back-to-back %.1f us %.2f TB/s"%(mb,dt*1e6,2*n*4/dt/1e12,dg*1e6,2*n*4/dg/1e12))
