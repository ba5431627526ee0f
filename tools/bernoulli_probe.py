import torch, time
torch.manual_seed(0)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/n*1e3
k=1600
print("threads", torch.get_num_threads())
print("bernoulli only      %.2f ms" % t(lambda: torch.empty((k,256)).bernoulli_(0.5)))
print("+ to uint8          %.2f ms" % t(lambda: torch.empty((k,256)).bernoulli_(0.5).to(torch.uint8)))
print("+ to cuda           %.2f ms" % t(lambda: torch.empty((k,256)).bernoulli_(0.5).to(torch.uint8).to("cuda")))
torch.manual_seed(1); a=torch.empty((k,256)).bernoulli_(0.5)
n0=torch.get_num_threads(); torch.set_num_threads(1)
torch.manual_seed(1); b=torch.empty((k,256)).bernoulli_(0.5)
print("same values with 1 thread:", torch.equal(a,b))
print("1 thread: all three %.2f ms" % t(lambda: torch.empty((k,256)).bernoulli_(0.5).to(torch.uint8).to("cuda")))
torch.set_num_threads(n0)
pin=torch.empty((k,256),dtype=torch.uint8).pin_memory()
def pinned():
    m=torch.empty((k,256)).bernoulli_(0.5)
    pin.copy_(m)
    return pin.to("cuda", non_blocking=True)
print("pinned path         %.2f ms" % t(pinned))
