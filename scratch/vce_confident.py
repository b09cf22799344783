"""vocab_ce_fwd at the C2 size on rows without / with a dominant probability (the clipped sweep's block-uniform second path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bert4clickpath_amd import ops, _lib as L
R, V, K = 40960, 50000, 128
g = torch.Generator(device='cuda').manual_seed(3)
wt = (torch.randn(V, K, device='cuda', generator=g) * 0.12).bfloat16()
bias = torch.randn(V, device='cuda', generator=g) * 0.3
y = torch.randint(0, V, (R,), device='cuda', generator=g, dtype=torch.int32)
gs = torch.tensor([1.0 / R], device='cuda')
for name, conf in (('no dominant entry', 0.0), ('every 64th row confident', 1 / 64), ('half of the rows confident', 0.5), ('all rows confident', 1.0)):
    h = torch.randn(R, K, device='cuda', generator=g) * 2.0
    pick = torch.rand(R, device='cuda', generator=g) < conf
    j = torch.randint(0, V, (R,), device='cuda', generator=g)
    hw = wt[j].float()
    h = torch.where(pick[:, None], hw * (30.0 / (hw * hw).sum(1, keepdim=True)), h).bfloat16()
    for _ in range(3): item, dh, rs = ops.vocab_ce_fwd(h, wt, bias, y, gs, V, L.CE_TF)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): item, dh, rs = ops.vocab_ce_fwd(h, wt, bias, y, gs, V, L.CE_TF)
    b.record(); torch.cuda.synchronize()
    print('%-28s %.3f ms   rows in the clip regime %.0f %%, rows with everything outside the range %.1f %%' % (name, a.elapsed_time(b) / 10, 100 * float((rs[:, 3] > 0).float().mean()), 100 * float((rs[:, 5] > 0).float().mean())))
