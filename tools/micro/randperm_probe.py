import time, torch
def t(n=60000, reps=20):
    t0=time.perf_counter()
    for _ in range(reps): torch.randperm(n)
    return (time.perf_counter()-t0)/reps*1e3
print('fresh', t(), 'ms; threads', torch.get_num_threads())
with torch.no_grad(): print('no_grad', t())
g=torch.Generator(); g.manual_seed(1)
t0=time.perf_counter()
for _ in range(20): torch.randperm(60000, generator=g)
print('own gen', (time.perf_counter()-t0)/20*1e3)
x=torch.randn(4000,4000); y=x@x
print('after matmul', t())
torch.set_num_threads(1)
print('1 thread', t())
torch.set_num_threads(128)
print('128 again', t())
for n in (1000, 10000, 30000, 60000, 100000): print(n, t(n))
