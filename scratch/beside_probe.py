"""What a background sweep costs the kernels beside it: N launches of an HBM-bound kernel on the main stream, timed alone and
while the side stream is kept busy with (a) the vocabulary-owned dW sweep, (b) the token-owned forward sweeps, both in the
background form (one wave per SIMD, <= 256 workgroups)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bert4clickpath_amd import ops, _lib as L
dev = 'cuda'
torch.manual_seed(0)
T, R, V, K = 229376, 20480, 50000, 128
bf = torch.bfloat16
x = torch.randn(T, 128, device=dev).to(bf)
w384 = torch.randn(384, 128, device=dev).to(bf)
w128 = torch.randn(128, 128, device=dev).to(bf)
bias = torch.zeros(384, device=dev)
y = torch.randn(T, 128, device=dev).to(bf)
gam, bet = torch.ones(128, device=dev), torch.zeros(128, device=dev)
h = (torch.randn(R, K, device=dev) * 0.5).to(bf)
wt = (torch.randn(V, K, device=dev) * 0.05).to(bf)
bv = torch.zeros(V, device=dev)
lab = torch.randint(0, V, (R,), device=dev, dtype=torch.int32)
gs = torch.tensor([1.0 / R], device=dev)
item, dh, rowscal = ops.vocab_ce_fwd(h, wt, bv, lab, gs, V, L.CE_TF)
dW, db = torch.zeros(K, V, device=dev), torch.zeros(V, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
nt, nv = (R + 127) // 128, (V + 127) // 128
# (the token-owned forward sweeps had a background form too until round 3: b4c_vocab_ce_fwd_sweep, retired with cloze_step)
bgs = {'dW sweep (vocabulary-owned)': lambda: ops.vocab_ce_dw_sweep(h, wt, bv, rowscal, V, dW, db, 0, nv, 256)}
fgs = {'gemm_nt QKV (T x 384 x 128)': lambda: ops.gemm_nt(x, w384, 384, bias),
       'gemm_nt T x 128 x 128': lambda: ops.gemm_nt(x, w128, 128, bias[:128]),
       'add_ln_fwd': lambda: ops.add_dropout_layernorm_fwd(x, y, gam, bet, 0.1, 7),
       'gemm_nt_ln': lambda: ops.gemm_nt_ln(x, w128, bias[:128], y, gam, bet, 0.1, 7) if hasattr(ops, 'gemm_nt_ln') else None}


def timed(fn, n, busy=None):
    torch.cuda.synchronize()
    if busy is not None:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                busy()
            e1.record()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3, (e0.elapsed_time(e1) / 4 * 1e3 if busy is not None else None)


for name, fn in fgs.items():
    try:
        if fn() is None:
            continue
    except Exception as e:
        print(name, 'skipped:', str(e)[:80]); continue
    alone, _ = timed(fn, 40)
    line = '%-30s alone %6.1f us' % (name, alone)
    for bn, bfn in bgs.items():
        bfn()
        t, tb = timed(fn, 40, bfn)
        line += ' | beside %s: %6.1f us (x%.2f), sweep %6.0f us' % (bn.split(' (')[0], t, t / alone, tb)
    print(line, flush=True)
for bn, bfn in bgs.items():
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        bfn()
    b.record(); torch.cuda.synchronize()
    print('%-30s alone %6.0f us per pass' % (bn, a.elapsed_time(b) / 4 * 1e3))
