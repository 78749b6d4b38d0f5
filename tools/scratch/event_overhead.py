import torch, statistics
torch.cuda.init()
x = torch.zeros(1<<20, device="cuda")
def pairs(n, fn=None):
    out=[]
    evs=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda._sleep(int(0.02*2.4e9))
    for e0,e1 in evs:
        e0.record()
        if fn: fn()
        e1.record()
    torch.cuda.synchronize()
    return [e0.elapsed_time(e1)*1e3 for e0,e1 in evs]
for _ in range(2):
    a=pairs(50); print("empty pair us: median %.2f min %.2f max %.2f" % (statistics.median(a), min(a), max(a)))
    b=pairs(50, lambda: x.add_(1.0)); print("tiny kernel pair us: median %.2f min %.2f" % (statistics.median(b), min(b)))
