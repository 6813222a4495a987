import torch, time
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e-3
for gb in (0.5, 2.0):
    n=int(gb*2**30/4)
    a=torch.empty(n,device="cuda"); b=torch.empty(n,device="cuda")
    tw=t(lambda: a.fill_(1.0)); tc=t(lambda: b.copy_(a)); tr=t(lambda: a.sum())
    print(f"{gb} GiB: fill (write) {gb*1.0737/tw/1e3:.2f} TB/s   copy (r+w) {2*gb*1.0737/tc/1e3:.2f} TB/s   sum (read) {gb*1.0737/tr/1e3:.2f} TB/s")
