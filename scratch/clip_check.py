import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bert4clickpath_amd import ops
class A: pass
a=A(); a.vocab=50000; a.d_model=128; a.layers=4; a.heads=2; a.dropout=0.1; a.dtype='bf16'; a.batch=4096; a.seq=200; a.n_batches=1
dev=torch.device('cuda')
model=bench.build_model(a, dev)
b=bench.make_batches(a,0,dev)[0]
orig=ops.vocab_ce_fwd
keep={}
def spy(*args, **kw):
    out=orig(*args, **kw); keep['rs']=out[2]; keep['h']=args[0]; return out
ops.vocab_ce_fwd=spy
loss=model.cloze_loss({'asin': b['items']}, b['labels'], training=True, flat_idx=b['flat_idx'])
rs=keep['rs']; h=keep['h'].float()
print('loss', float(loss), 'rows', rs.shape[0], 'clipped rows', int((rs[:,2]<0).sum()), 'h absmax', float(h.abs().max()), 'h std', float(h.std()))
