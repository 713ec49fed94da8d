import torch
N=1<<26
lw=torch.randn(N,device="cuda"); idx=torch.empty(N,dtype=torch.int64,device="cuda")
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev=[torch.cuda.Event(enable_timing=True) for _ in range(n+1)]
    ev[0].record()
    for i in range(n): fn(); ev[i+1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i+1]) for i in range(n))[n//2]*1e3
print("fill int64 (8N written)", t(lambda: idx.fill_(3)), "us")
print("float->int64 copy (4N read, 8N written)", t(lambda: idx.copy_(lw)), "us")
print("float copy (4N read, 4N written)", t(lambda: lw.clone()), "us")
