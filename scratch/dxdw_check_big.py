import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops
M = 456123
g = torch.Generator().manual_seed(M)
x = torch.randint(-2, 3, (M, 128), generator=g).float().cuda().bfloat16()
G = torch.randint(-2, 3, (M, 384), generator=g).float().cuda().bfloat16()
W = torch.randint(-1, 2, (128, 384), generator=g).float().cuda().bfloat16()
res = torch.randint(-3, 4, (M, 128), generator=g).float().cuda().bfloat16()
dWs = [torch.zeros(128, 128, device='cuda') for _ in range(3)]; dbs = [torch.zeros(128, device='cuda') for _ in range(3)]
ref = ops.gemm_nt(G, W, 128, residual=res)
nores = ops.gemm_nt(G, W, 128)
for rep in range(4):
    dx = ops.gemm_dxdw(x, G, W, dWs, dbs, residual=res)
    torch.cuda.synchronize()
    badmask = (dx != ref)
    rows = badmask.any(1).nonzero().reshape(-1)
    print(rep, 'rows wrong', rows.numel())
    if rows.numel():
        r = int(rows[0])
        cols = badmask[r].nonzero().reshape(-1)
        print('  row', r, 'tile', r // 32, 'row in tile', r % 32, 'wg', (r // 32) % 256, 'local', (r // 32) // 256, 'cols wrong', cols[:16].tolist(), 'n', cols.numel())
        c = int(cols[0])
        print('  dx', float(dx[r, c]), 'ref', float(ref[r, c]), 'no-res', float(nores[r, c]), 'res', float(res[r, c]))
        # is dx equal to no-res + residual of another row / to another tile's value?
        d = (dx[r].float() - nores[r].float())
        print('  dx - GW (should be res):', d[:8].tolist(), 'res:', res[r, :8].float().tolist())
        tl = sorted(set((rows // 32).tolist()))
        print('  tiles with wrong rows:', tl[:20], 'count', len(tl), 'locals', sorted(set([(t // 256) for t in tl])))
