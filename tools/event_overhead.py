import torch, time
x = torch.zeros(1024, device="cuda")
def pair(fn, n=200):
    ts=[]
    for _ in range(n):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); 
        ts.append((e0,e1))
    torch.cuda.synchronize()
    v=sorted(a.elapsed_time(b)*1e3 for a,b in ts)
    return v[len(v)//2], v[len(v)//10], v[-len(v)//10]
print("empty pair us (median, p10, p90):", pair(lambda: None))
print("tiny kernel pair:", pair(lambda: x.add_(1)))
big = torch.zeros(64<<20, device="cuda")
print("256MB add pair:", pair(lambda: big.add_(1), 50))
# queue saturated: launch 50 kernels then pairs interleaved
def sat():
    for _ in range(3): big.add_(1)
print("pair after 3 big kernels (incl. them):", pair(sat, 30))
