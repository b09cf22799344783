"""configs[3]-like shapes (two features, d_model 256, 6 layers, vocab 100k) for a few bf16 training steps: exercises the
unfused LayerNorm path (d > 128), 4 heads of 64, the two-feature embedding kernels and the sorted backward with 32-bit keys."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import optim, input_pipeline
from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
dev = 'cuda'
V, B, S = 100000, 512, 200
torch.manual_seed(0)
head = SoftMaxHead([1024, 512, 256, 128], V)
m = ClickstreamTransformer({'items': ['asin'], 'actions': ['act']}, {'items': ['i%d' % i for i in range(V)], 'actions': ['a%d' % i for i in range(50)]},
                           {'items': 224, 'actions': 32}, head, value_to_head='[MASK]', num_encoder_layers=6, num_attention_heads=4,
                           dropout_rate=0.1, compute_dtype=torch.bfloat16).to(dev)
opt = optim.Adam(m.parameters())
b = input_pipeline.synthetic_cloze_batch(B, S, V, seed=1, n_extra_features=1, extra_vocab=50)
ids = torch.from_numpy(b['ids'])[:, 2:S - 1].contiguous().to(dev)
act = torch.from_numpy(b['extra'][0])[:, 2:S - 1].contiguous().to(dev)
lab, fi = torch.from_numpy(b['labels']).to(dev), torch.from_numpy(b['flat_idx']).to(dev)
losses = []
for i in range(6):
    if i == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    loss = m.cloze_loss({'asin': ids, 'act': act}, lab, training=True, flat_idx=fi)
    loss.backward()
    opt.step()
    losses.append(float(loss.detach()))
torch.cuda.synchronize()
print('losses', [round(x, 4) for x in losses], 'ms/step %.1f' % ((time.perf_counter() - t0) / 4 * 1e3), 'masked', lab.numel())
assert all(x == x for x in losses) and losses[-1] < losses[0]
