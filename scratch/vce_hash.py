"""Hashes of the logits-free head's forward outputs (item loss, dh, row scalars) on seeded C2-sized operands, for
bit-identity checks between two builds of the library (B4C_LIB_PATH) -- and the sweeps' times (B4C_VCE_TIMING=1)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bert4clickpath_amd import ops, _lib as L
R, V, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (40960, 50000, 128))]
for scale in (0.3, 1.0, 2.0):
    g = torch.Generator(device='cuda').manual_seed(7)
    h = (torch.randn(R, K, device='cuda', generator=g) * scale).bfloat16()
    Vp = (V + 7) // 8 * 8
    wt = torch.zeros(Vp, K, device='cuda', dtype=torch.bfloat16); wt[:V] = (torch.randn(V, K, device='cuda', generator=g) * scale).bfloat16()
    b = torch.zeros(Vp, device='cuda'); b[:V] = torch.randn(V, device='cuda', generator=g) * 0.1
    y = torch.randint(0, V, (R,), device='cuda', generator=g).int()
    gs = torch.tensor([1.0 / R], device='cuda')
    for rep in range(4):
        item, dh, rowscal = ops.vocab_ce_fwd(h, wt, b, y, gs, V, L.CE_TF)
    torch.cuda.synchronize()
    hs = [hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16] for t in (item, dh.view(torch.int16), rowscal)]
    print('scale', scale, 'loss %.6f' % float(item.mean()), 'clipped rows', int((rowscal[:, 3] > 0).sum()), *hs, flush=True)
