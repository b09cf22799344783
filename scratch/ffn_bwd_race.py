import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_ffn_bwd import _inputs, _five_kernels, _fused
from bert4clickpath_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 456123
rate, seed = 0.1, 99
a = _inputs(M, 100, seed, rate)
ref = _five_kernels(a, rate, seed)[0].float()
dz, dy, _, _ = ops.add_dropout_layernorm_bwd(a['dout'], a['z'], a['stats'], a['gamma'], rate, seed)
lim = 0.02 * float(ref.abs().max())
keep = None
for rep in range(60):
    if keep is not None and rep > 8: break
    dx = _fused(a, rate, seed)[0].float()
    off = ((dx - ref).abs() > lim)
    rows = off.any(1).nonzero().reshape(-1)
    if rows.numel() == 0:
        continue
    keep = (dx.clone(), rows.clone())
    tiles = (rows // 32)
    import collections
    per_tile = collections.Counter(tiles.tolist())
    print('rep', rep, 'rows off', rows.numel(), 'tiles', len(per_tile), 'rows/tile histogram', sorted(collections.Counter(per_tile.values()).items()))
    print('   local tile index of bad tiles (tile // 256):', sorted(collections.Counter((tiles // 256).tolist()).items())[:60])
    print('   row-in-tile of bad rows:', sorted(collections.Counter((rows % 32).tolist()).items()))
    r = int(rows[0])
    cols = off[r].nonzero().reshape(-1)
    print('   row', r, 'n cols off', cols.numel(), 'cols', cols[:16].tolist())
    gemm_f, gemm_r = dx[r] - dz[r].float(), ref[r] - dz[r].float()
    print('   (dx - dz) fused', gemm_f[cols[:6]].tolist(), 'five', gemm_r[cols[:6]].tolist())
    # is the fused row equal to the reference of ANOTHER row (stale tile)?
    d = (ref - dx[r]).abs().max(1).values
    j = int(d.argmin())
    print('   closest reference row to the fused row:', j, 'dist', float(d[j]), ' (own row dist', float(d[r]), ')')

# which stale operand explains a bad row?  candidates: the gate (h) of the row 1..4 workgroup tiles back, dy of the row 1..4 tiles back
print('--- explanation of bad rows (last bad rep)')
dx, rows = keep
F = a['F']
w2 = a['wc2'].float()[:F]            # [F][128]
w1 = a['wc1'].float()[:, :F]         # [128][F]
hh = a['h'].float()[:, :F]
dyf, dzf = dy.float(), dz.float()
def cand(r, hrow, dyrow):
    dh = ((dyf[dyrow] @ w2.T).bfloat16().float()) * (hh[hrow] > 0)
    return dh.bfloat16().float() @ w1.T + dzf[r]
for r in rows[:10].tolist():
    out = []
    for back in (0, 1, 2, 3, 4, 8):
        q = r - back * 256 * 32
        if q < 0: continue
        e_h = float((cand(r, q, r) - dx[r]).abs().max())
        e_dy = float((cand(r, r, q) - dx[r]).abs().max())
        out.append('back %d: stale-h err %.4f, stale-dy err %.4f' % (back, e_h, e_dy))
    print('  row', r, '|', ' | '.join(out))
