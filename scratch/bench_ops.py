"""Micro-benchmarks of the individual HIP ops at C2 sizes (achieved GB/s vs algorithmic bytes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bert4clickpath_amd import ops
torch.manual_seed(0)
dev = 'cuda'
T, d, R, V = 819200, 128, 40960, 50000
bf = torch.bfloat16
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us
def report(name, us, nbytes, flops=0):
    print('%-34s %9.1f us  %7.0f GB/s  %7.1f TF/s' % (name, us, nbytes / us / 1e3, flops / us / 1e6), flush=True)
only = sys.argv[1:] 
def want(n): return not only or any(o in n for o in only)
x = torch.randn(T, d, device=dev).to(bf)
w128 = torch.randn(128, 128, device=dev).to(bf)
w384 = torch.randn(384, 128, device=dev).to(bf)
bias = torch.randn(384, device=dev)
if want('gemm_nt_128'):
    report('gemm_nt T x128 x128', timeit(lambda: ops.gemm_nt(x, w128, 128, bias[:128])), T*d*2*2, 2*T*128*128)
    res = torch.randn(T, d, device=dev).to(bf)
    report('gemm_nt T x128 x128 +residual', timeit(lambda: ops.gemm_nt(x, w128, 128, bias[:128], residual=res)), T*d*2*3, 2*T*128*128)
if want('gemm_nt_qkv'):
    report('gemm_nt T x384 x128 (QKV)', timeit(lambda: ops.gemm_nt(x, w384, 384, bias)), T*d*2*4, 2*T*128*384)
    x3 = torch.randn(T, 384, device=dev).to(bf); wc = torch.randn(128, 384, device=dev).to(bf)
    report('gemm_nt T x128 x384 (QKV dX)', timeit(lambda: ops.gemm_nt(x3, wc, 128)), T*d*2*4, 2*T*128*384)
if want('gemm_tn'):
    g = torch.randn(T, d, device=dev).to(bf)
    report('gemm_tn T: 128x128 (dW)', timeit(lambda: ops.gemm_tn(x, g, 128, 128)), T*d*2*2, 2*T*128*128)
    g3 = torch.randn(T, 384, device=dev).to(bf)
    report('gemm_tn T: 128x384 (dWqkv)', timeit(lambda: ops.gemm_tn(x, g3, 128, 384)), T*d*2*4, 2*T*128*384)
if want('vocab'):
    h = torch.randn(R, 128, device=dev).to(bf); wv = (torch.randn(V, 128, device=dev) * 0.05).to(bf); bv = torch.zeros(V, device=dev)
    report('gemm_nt R x50000 x128 (vocab fwd)', timeit(lambda: ops.gemm_nt(h, wv, V, bv), n=5), R*V*2 + R*256 + V*256, 2*R*V*128)
    dl = torch.randn(R, V, device=dev).to(bf); wcv = torch.randn(128, V, device=dev).to(bf)
    report('gemm_nt R x128 x50000 (vocab dX)', timeit(lambda: ops.gemm_nt(dl, wcv, 128), n=5), R*V*2, 2*R*V*128)
    report('gemm_tn R: 128x50000 (vocab dW)', timeit(lambda: ops.gemm_tn(h, dl, 128, V), n=5), R*V*2, 2*R*V*128)
    lab = torch.randint(0, V, (R,), device=dev, dtype=torch.int32); sc = torch.tensor([1.0 / R], device=dev)
    report('softmax_ce_fused R x50000', timeit(lambda: ops.softmax_ce_fwd_bwd_(dl, lab, sc, V, 0), n=5), 2*R*V*2)
if want('ln'):
    y = torch.randn(T, d, device=dev).to(bf); gam = torch.ones(d, device=dev); bet = torch.zeros(d, device=dev)
    report('add_ln_fwd (p=0.1)', timeit(lambda: ops.add_dropout_layernorm_fwd(x, y, gam, bet, 0.1, 7)), T*d*2*4)
    z, out, st = ops.add_dropout_layernorm_fwd(x, y, gam, bet, 0.1, 7)
    report('add_ln_bwd (p=0.1)', timeit(lambda: ops.add_dropout_layernorm_bwd(y, z, st, gam, 0.1, 7)), T*d*2*4)
if want('attn'):
    B, S, H, dh = 4096, 200, 2, 64
    qkv = (torch.randn(T, 384, device=dev) * 0.5).to(bf)
    ids_len = torch.randint(23, 201, (B,), device=dev)
    pad = (torch.arange(S, device=dev)[None, :] >= ids_len[:, None]).to(torch.uint8).contiguous()
    o, lse = ops.attn_fwd(qkv, pad, B, S, H, dh)
    report('attn_fwd (ragged lens)', timeit(lambda: ops.attn_fwd(qkv, pad, B, S, H, dh)), T*d*2*4, 4*B*S*S*d)
    do = torch.randn(T, d, device=dev).to(bf)
    report('attn_bwd (ragged lens)', timeit(lambda: ops.attn_bwd(qkv, pad, o, do, lse, B, S, H, dh)), T*d*2*8, 10*B*S*S*d)
    pad0 = torch.zeros_like(pad)
    report('attn_fwd (no padding)', timeit(lambda: ops.attn_fwd(qkv, pad0, B, S, H, dh)), T*d*2*4, 4*B*S*S*d)
    report('attn_bwd (no padding)', timeit(lambda: ops.attn_bwd(qkv, pad0, o, do, lse, B, S, H, dh)), T*d*2*8, 10*B*S*S*d)
if want('embed'):
    from bert4clickpath_amd import input_pipeline
    b = input_pipeline.synthetic_cloze_batch(4096, 200, V, seed=1)
    ids = torch.from_numpy(b['ids']).cuda(); tab = torch.randn(V + 11, d, device=dev) * 0.05
    pe = torch.randn(200, d, device=dev)
    report('embed_fwd', timeit(lambda: ops.embed_concat_pe_fwd([ids], [tab], pe, 11.3, 0.1, 5, bf)), T*d*(4+2))
    dout = torch.randn(T, d, device=dev).to(bf)
    dout[ids.reshape(-1) == 0] = 0
    report('embed_bwd (zipf ids)', timeit(lambda: ops.embed_concat_pe_bwd([ids], [tab], dout, 11.3, 0.1, 5), n=5), T*d*(2+4))
