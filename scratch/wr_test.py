import torch, sys
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
R, V = 40960, 50000
x = torch.empty(R, V, dtype=torch.bfloat16, device='cuda')
t = timeit(lambda: x.fill_(1.0)); print('fill_ 4.1GB linear: %.0f us  %.0f GB/s' % (t, x.numel()*2/t/1e3))
y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x)); print('copy 4.1GB->4.1GB: %.0f us  %.0f GB/s (r+w)' % (t, 2*x.numel()*2/t/1e3))
t = timeit(lambda: x.sum()); print('sum (read 4.1GB): %.0f us  %.0f GB/s' % (t, x.numel()*2/t/1e3))
z = torch.empty(819200, 128, dtype=torch.bfloat16, device='cuda'); w = torch.empty_like(z)
t = timeit(lambda: w.copy_(z)); print('copy 210MB->210MB: %.0f us  %.0f GB/s (r+w)' % (t, 2*z.numel()*2/t/1e3))
