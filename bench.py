#!/usr/bin/env python3
"""bench.py -- masked-items/sec of the BERT4Rec Cloze training step on MI355X.

A step = one pass of the hot path over one synthetic batch already resident in HBM:
embedding stage -> 4 encoder layers -> device-side [MASK] index generation + label compaction -> [MASK]-row gather ->
SoftMaxHead trunk + 50k-way projection -> fused softmax / masked sparse CE -> full backward -> gradient all-reduce
(N > 1) -> Adam.  Nothing is read back to the host inside the timed region.
After the timed training region rank 0 also times the SCORING leg on the same batches (forward -> materialised
(B, M, V) probabilities, as the reference's head returns them -> HitRate@10 / NDCG@10 update) and reports it under
"eval" with the roofline of the vocabulary projection.
Workload (BASELINE.json configs[1]): vocab 50,000, encoder length 200 (197 items + 3 specials),
d_model 128, 4 layers, 2 heads, dff 100 (reference-hard-coded), head [1024,512,256,128] -> V,
batch 4096 sequences per GPU (weak scaling), 10 masked items per sequence, dropout 0.1, bf16
storage / fp32 accumulate / fp32 master weights.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without WORLD_SIZE in the environment: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for roofline / cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)        # 20 warm-up + 100 timed steps
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=4096, help='sequences per GPU')
    ap.add_argument('--seq', type=int, default=200, help='encoder length (items + 3 specials)')
    ap.add_argument('--vocab', type=int, default=50000)
    ap.add_argument('--d_model', type=int, default=128)
    ap.add_argument('--layers', type=int, default=4)
    ap.add_argument('--heads', type=int, default=2)
    ap.add_argument('--dff', type=int, default=100, help='encoder FFN width: 100 = the reference (hard-coded, clickstream_transformer.py:225); '
                    '4 * d_model (512 at C2) = the BERT4Rec paper\'s width (SURVEY D4)')
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--n_batches', type=int, default=8, help='distinct resident batches cycled through')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--cpu_rows', type=int, default=256, help='sequences in the bounded CPU-baseline sample (SURVEY 8d: B=256)')
    ap.add_argument('--cpu_seconds', type=float, default=15.0, help='wall-clock bound of the CPU-baseline sample')
    ap.add_argument('--eval_steps', type=int, default=6, help='timed scoring batches (0 = skip the scoring leg)')
    ap.add_argument('--full_steps', type=int, default=8, help='extra steps, after the timed region, of the step that computes every position of every '
                    'layer (padded layout, full last layer) for the every_position_of_every_layer entry; 0 = skip')
    ap.add_argument('--dense', action='store_true', help='A/B: run the encoder on the padded (B, S) layout as the reference does '
                    '(default: padding-free layout, pad positions are not computed -- DESIGN.md section 3)')
    ap.add_argument('--full_length', action='store_true', help='SURVEY 8d no-padding variant: every sequence has 197 items (the packed '
                    'layout then equals the dense one)')
    ap.add_argument('--host_flat_idx', action='store_true', help='A/B: hand the step host-precomputed [MASK] indices (round-1 bench)')
    ap.add_argument('--record_steps', type=int, default=3,
                    help='timed steps (the LAST N of the timed region: the steady state) whose launches are bracketed by HIP events for the '
                         'roofline; -1 = all (costs ~0.4 ms/step of event overhead), 0 = none')
    ap.add_argument('--ops_flags', default='', help='A/B switches of bert4clickpath_amd.ops, e.g. "fused_ln=0,sorted_embed_bwd=0"')
    ap.add_argument('--materialised_logits', action='store_true',
                    help='A/B: vocabulary projection writes the (R x V) logits and the CE reads them (ops.flash_ce = False)')
    ap.add_argument('--config', default='c2', choices=['c2', 'c4', 'c5'], help='c2 = BASELINE.json configs[1] (default, the metric\'s workload); '
                    'c4 = configs[3]: items(192) + actions(64) concatenated -> d_model 256, 4 heads, 6 layers, vocab 100,000; '
                    'c5 = configs[4], one GPU\'s share: vocab 2,000,000, seq 512, d_model 256, 4 heads, 4 layers, batch 1024, sampled-softmax head '
                    '(8,192 shared log-uniform negatives; no reference counterpart)')
    ap.add_argument('--action_dim', type=int, default=0, help='second feature (actions) embedding dim, part of d_model; 0 = single feature')
    ap.add_argument('--action_vocab', type=int, default=1000)
    ap.add_argument('--feature_sum', action='store_true', help='config 4 as BASELINE.json words it ("dual embedding gather + sum"): items and '
                    'actions are both d_model wide and their rows are ADDED (feature_combine=\'sum\', an extension: the reference concatenates)')
    ap.add_argument('--dense_adam', action='store_true', help='config 5 A/B: the dense Adam pass over all 770 M parameters every step instead '
                    'of the row-lazy form (optim.LazyRows: same arithmetic, bit-identical tables, rows updated when they are used)')
    ap.add_argument('--traffic_json', default=None, help='optional JSON with PMC-derived HBM bytes per launch')
    a = ap.parse_args()
    if a.config == 'c4':
        a.vocab, a.d_model, a.layers, a.heads, a.action_dim = 100000, 256, 6, 4, 64
    a.sampled = 0
    if a.config == 'c5':
        a.vocab, a.seq, a.d_model, a.layers, a.heads, a.batch, a.sampled = 2000000, 512, 256, 4, 4, 1024, 8192
        a.eval_steps = 0          # scoring over 2M items materialises 41 GB of probabilities per batch: not part of this line
        a.no_cpu_baseline = True
    return a


def build_model(a, device):
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SampledSoftmaxHead, SoftMaxHead
    torch.manual_seed(1234)
    vocab = ['i%d' % i for i in range(a.vocab)]
    head = SampledSoftmaxHead([1024, 512, 256, 128], a.vocab, num_sampled=a.sampled) if a.sampled else \
        SoftMaxHead([1024, 512, 256, 128], a.vocab)
    chains, vocabs, dims = {'items': ['asin']}, {'items': vocab}, {'items': a.d_model - a.action_dim}
    if a.action_dim > 0:      # second feature, concatenated on the last axis (reference transformer.py:384-388)
        chains['actions'], vocabs['actions'], dims['actions'] = ['act'], ['a%d' % i for i in range(a.action_vocab)], a.action_dim
    if a.feature_sum:         # ... or added to the first one: both d_model wide
        dims = {f: a.d_model for f in dims}
    model = ClickstreamTransformer(chains, vocabs, dims, head,
                                   value_to_head='[MASK]', num_encoder_layers=a.layers, num_attention_heads=a.heads,
                                   dropout_rate=a.dropout, feature_combine='sum' if a.feature_sum else 'concat', encoder_ff_dim=a.dff,
                                   compute_dtype=torch.bfloat16 if a.dtype == 'bf16' else torch.float32)
    return model.to(device)


def backward_order(model):
    """Arena order = the order gradients COMPLETE in backward: head trunk, encoder layers last -> first, embedding -- and
    the vocabulary projection with the embedding when its dW sweep runs as a background job beside the encoder backward
    (ops.background_dw_expected: it is announced when backward ends), so that the buckets before it go out as they complete."""
    from bert4clickpath_amd import ops
    names = {id(p): n for n, p in model.named_parameters()}
    L = model.num_encoder_layers
    background = ops.background_dw_expected(model.transformer.d_model, model.encoder_ff_dim, model.compute_dtype)
    late = ('head.output_layer.',) if (background and ops.flash_ce) else ()

    def key(p):
        n = names[id(p)]
        if n == 'head.output_embedding':       # the sampled head's vocabulary-major projection: row-sparse gradient, kept
            return (L + 3, n)                  # at the very end of the arena next to the embedding tables (parallel.py)
        if n.startswith(late) and late:
            return (L + 2, '~' + n)            # after the embedding tables, in their bucket
        if n.startswith('head.'):
            return (0, n)
        if 'enc_layers.' in n:
            i = int(n.split('enc_layers.')[1].split('.')[0])
            return (1 + (L - 1 - i), n)
        return (L + 2, n)
    return key


def make_batches(a, rank, device):
    from bert4clickpath_amd import input_pipeline
    out = []
    for j in range(a.n_batches):
        b = input_pipeline.synthetic_cloze_batch(a.batch, a.seq, a.vocab, seed=4321 + rank + 1000 * j, full_length=a.full_length,
                                                 n_extra_features=1 if a.action_dim > 0 else 0, extra_vocab=a.action_vocab)
        ids = torch.from_numpy(b['ids'])
        feats = {'asin': ids[:, 2:a.seq - 1].contiguous().to(device)}
        if a.action_dim > 0:
            feats['act'] = torch.from_numpy(b['extra'][0])[:, 2:a.seq - 1].contiguous().to(device)
        out.append({'feats': feats, 'items': feats['asin'],
                    'flat_idx': torch.from_numpy(b['flat_idx']).to(device),
                    'labels': torch.from_numpy(b['labels']).to(device),
                    'labels_padded': torch.from_numpy(b['labels_padded']).to(device),      # (B, 10) float32, -1 = pad
                    # host-side batch metadata, as the input pipeline has it when it pads (input_pipeline.py:198-214):
                    # number of non-pad positions of the chained batch ([CLS] [SEP] items [SEP])
                    'n_real': int((b['ids'] != 0).sum()),
                    # for the launch recorder's algorithmic FLOP counts only: sum of len^2 over the sequences (len = real tokens,
                    # specials included) and sum of (masked positions x len)
                    'sum_len_sq': int(((b['lens'] + 3).astype(np.int64) ** 2).sum()),
                    'sum_q_len': int((np.minimum((2 * b['lens']) // 5, 10).astype(np.int64) * (b['lens'] + 3)).sum()),
                    # distinct table rows the batch reads (the row-lazy Adam's algorithmic bytes: rows, not id occurrences)
                    'distinct_ids': int(np.unique(b['ids']).size),
                    'R': int(b['labels'].shape[0])})
    return out


def cpu_baseline(a):
    """The oracle's torch-CPU restatement of the reference dataflow (materialised S x S attention and
    (B*M) x V probabilities), full training step, on a bounded sample of the same workload."""
    from bert4clickpath_amd import input_pipeline
    from oracle import numpy_ref as nr
    from oracle import torch_ref as tr
    # the GPU box gives one GPU's job a 16-core share of the host (more threads only oversubscribe it)
    threads = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    rng = np.random.default_rng(1234)
    P = nr.init_params(rng, {'items': a.vocab + 11}, {'items': a.d_model}, a.layers, a.dff, [1024, 512, 256, 128], a.vocab)
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in P.items()}
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    vv = {k: torch.zeros_like(v) for k, v in P.items()}
    b = input_pipeline.synthetic_cloze_batch(a.cpu_rows, a.seq, a.vocab, seed=4321)
    ids, lab = torch.from_numpy(b['ids']), torch.from_numpy(b['labels']).long()

    def step(t):
        for v in P.values():
            v.grad = None
        loss, _ = tr.model_loss(ids, lab, P, a.layers, a.heads, 4)
        loss.backward()
        with torch.no_grad():
            for k in P:
                tr.adam_step(P[k], P[k].grad, m[k], vv[k], t)
    step(1)
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0 < a.cpu_seconds and n < 50) or n < 2:
        step(n + 2)
        n += 1
    dt = (time.perf_counter() - t0) / max(n, 1)
    return {'value': float(lab.numel() / dt), 'unit': 'masked-items/s', 'cores': threads, 'kind': 'port',
            'sample': '%d sequences x S=%d (R=%d masked items), %d timed steps, fp32, torch-CPU restatement of the '
                      'reference dataflow (TF 2.3.1 not installable)' % (a.cpu_rows, a.seq, lab.numel(), n),
            'ms_per_step': dt * 1e3}


KERNEL_OF = {   # launch family (ops recorder) -> kernel symbol(s) in the rocprofv3 trace
    'gemm_nt': 'gemm_nt_kernel (dense fwd + dX on the token-sized tensors)', 'gemm_tn': 'gemm_tn_bf16_kernel (dW)',
    'gemm_nt_rows': 'gemm_nt_kernel on the [MASK] rows only (head trunk, rows-only last layer: a few us per launch)',
    'gemm_tn_rows': 'gemm_tn kernels on the [MASK] rows only', 'gemm_nt_ln_rows': 'gemm_nt_ln_kernel on the [MASK] rows only',
    'add_ln_bwd_rows': 'add_ln_bwd_kernel on the [MASK] rows only',
    'attn_fwd': 'attn_fwd_mfma_kernel', 'attn_bwd': 'attn_bwd_resident_kernel',
    'attn_mq_fwd': 'attn_mq_fwd_mfma_kernel (last layer: the [MASK] rows against all keys, one wave per (sequence, head))',
    'attn_mq_bwd': 'attn_mq_bwd_mfma_kernel', 'softmax_ce': 'softmax_ce_bf16_kernel',
    'add_ln_fwd': 'add_ln_fwd_kernel', 'gemm_nt_ln': 'gemm_nt_ln_kernel (out-proj / FFN2 GEMM + residual + dropout + LayerNorm)', 'add_ln_bwd': 'add_ln_bwd_kernel', 'embed_fwd': 'embed_fwd_kernel',
    'embed_bwd': 'embed_bwd_kernel', 'adam': 'adam_kernel',
    'vocab_ce_fwd': 'vce_token_kernel<128,1|2> + vce_combine_kernel (projection + softmax CE + dX, logits in registers)',
    'vocab_ce_dw': 'vce_dw_kernel + vce_label_kernel (projection dW / db, logits recomputed)',
    'vocab_ce_dw_bg': 'vce_dw_kernel<128,1> (the same sweep as a background kernel on a side stream: one wave per SIMD, in pieces '
                      'beside the encoder backward; its time is NOT additive to the step)',
    'vocab_proj': 'gemm_nt_wide2_kernel<true> (vocabulary projection with the softmax epilogue: probabilities R x V out)',
    'vocab_lse': 'vce_token_kernel<128,0> + vce_lse_kernel (row lse of the logits, recomputed in registers)',
    'softmax_rows': 'softmax_rows_bf16_kernel (row in registers: one read, one write)', 'topk_rows': 'topk_rows_kernel',
    'vocab_rank': 'vce_label_logit_kernel + vce_scan_kernel<128, RANK> (rank of the true item, logits recomputed in registers)',
    'vocab_topk': 'vce_scan_kernel<128, CLASSMAX | COLLECT> + vce_tau_kernel + vce_select_kernel (top-k ids, logits recomputed in registers)'}


def roofline_from(fams, steps, peak_tf, dom=None):
    """Dominant launch family of the timed region (largest summed HIP-event time; or the named one) against the HBM
    roofline: achieved = sum of the launches' ALGORITHMIC bytes / summed duration."""
    table = {}
    for fam, v in fams.items():
        ms = max(v['ms'], 1e-9)
        table[fam] = {'ms_per_step': v['ms'] / steps, 'launches_per_step': v['launches'] / steps,
                      'GB_per_s': v['bytes'] / ms / 1e6, 'TFLOP_per_s': v['flops'] / ms / 1e9}
    # background families run on a side stream BESIDE the main stream's launches (their event time overlaps the others' and
    # is stretched by design: one wave per SIMD): they are listed, but the dominant family is the main stream's largest
    fg = {f: x for f, x in fams.items() if not f.endswith('_bg')} or fams
    dom = dom or max(fg, key=lambda f: fg[f]['ms'])
    v = fams[dom]
    # which roofline bounds the family: arithmetic intensity of its ALGORITHMIC work against the machine balance
    # (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B).  The logits-free vocabulary sweeps move a few MB and do ~1 TFLOP: MFMA-bound.
    if v['bytes'] > 0 and v['flops'] / v['bytes'] > peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9):
        ach = v['flops'] / v['ms'] / 1e9
        head = {'bound': 'mfma', 'achieved': ach, 'peak': peak_tf, 'unit': 'TFLOP/s', 'frac': ach / peak_tf,
                'algorithmic_flops_per_launch': v['flops'] / v['launches'],
                'hbm_GB_per_s': v['bytes'] / v['ms'] / 1e6}
    else:
        ach = v['bytes'] / v['ms'] / 1e6
        head = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                'algorithmic_bytes_per_launch': v['bytes'] / v['launches'],
                'mfma_frac_of_%g_TF' % peak_tf: v['flops'] / v['ms'] / 1e9 / peak_tf}
    head.update({'kernel': KERNEL_OF.get(dom, dom), 'family': dom, 'avg_launch_ms': v['ms'] / v['launches'],
                 'launches_timed': v['launches'], 'traffic': None, 'families': table})
    bg = sorted(f for f in fams if f.endswith('_bg'))
    if bg:
        # kept out of `families`, whose rows add up to (at most) the step: these overlap the rows there
        head['beside'] = {'families': {f: table.pop(f) for f in bg}, 'ms_per_step': sum(fams[f]['ms'] for f in bg) / steps,
                          'note': 'launches of these families run on a side stream at the same time as the main stream\'s '
                                  '(every duration in `families` was measured with them on the CUs); their time is not '
                                  'additive to the step'}
    return head


def eval_leg(model, batches, a, peak_tf):
    """Scoring leg (reference: evaluate / predict): forward without dropout -> (B, M, V) probabilities materialised as
    SoftMaxHead returns them (head.py:36-47) -> ClozeMaskedRecall(10) / ClozeMaskedNDCG(10) update (utils.py:161-190,
    225-255).  Timed on the device timeline; the vocabulary projection's own roofline is reported (north_star: >= 60 %
    of HBM peak on the vocab-projection at batch 4096 x seq 200).
    `fused_topk`: the same batches and metrics with the scores never in memory -- model(x, scores='lazy') hands the metrics
    the rows the projection would be applied to, and they rank through b4c_vocab_rank (one sweep, logits in accumulators)."""
    from bert4clickpath_amd import ops
    from bert4clickpath_amd.cloze import ClozeMaskedNDCG, ClozeMaskedRecall

    def leg(lazy):
        rec, ndcg = ClozeMaskedRecall(10), ClozeMaskedNDCG(10)

        def score(i):
            b = batches[i % len(batches)]
            ops.set_record_hints(token_rows=a.batch * a.seq if a.dense else b['n_real'], sum_len_sq=b['sum_len_sq'], sum_q_len=b['sum_q_len'])
            with torch.no_grad():
                scores = model(b['feats'], training=False, max_matches=10, packed=False if a.dense else None,
                               n_real_tokens=None if a.dense else b['n_real'], scores='lazy' if lazy else None)   # (B, 10, V), no host sync
                rec.update_state(b['labels_padded'], scores)
                ndcg.update_state(b['labels_padded'], scores)
        for i in range(2):
            score(i)
        rec.reset_states()
        ndcg.reset_states()
        torch.cuda.synchronize()
        ops.start_recording()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.eval_steps + 1)]
        ev[0].record()
        for i in range(a.eval_steps):
            score(i)
            ev[i + 1].record()
        fams = ops.stop_recording()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.eval_steps))
        R = sum(batches[i % len(batches)]['R'] for i in range(a.eval_steps))
        dom = ('vocab_rank' if lazy else 'vocab_proj')
        roof = roofline_from(fams, a.eval_steps, peak_tf, dom=dom if dom in fams else None)
        return {'batches': a.eval_steps, 'ms_per_batch': sum(ms) / len(ms), 'ms_median': ms[len(ms) // 2],
                'masked_items_per_s': R / (sum(ms) / 1e3), 'hit_rate_at_10': float(rec.result()), 'ndcg_at_10': float(ndcg.result()),
                'roofline': roof}

    out = leg(False)
    out['what'] = 'forward -> materialised (B, 10, V) probabilities -> HitRate@10 + NDCG@10 update, batch %d' % a.batch
    if a.dtype == 'bf16' and not a.sampled:
        out['fused_topk'] = leg(True)
        out['fused_topk']['what'] = ('forward -> HitRate@10 + NDCG@10 update through the logits-free ranking sweep (scores never in memory; ranks '
                                     'the fp32 logits, where the materialised leg ranks bf16 probabilities), batch %d' % a.batch)
    return out


class Training:
    """bench.py's training step as an object (tests/test_gpu_parallel.py runs the very same step at 2 ranks): the model of
    the workload, the optimizer over an arena in BACKWARD order, the gradient reducer with the late bucket, the resident
    batches of this rank, and step(i)."""

    def __init__(self, a, rank, world, device):
        from bert4clickpath_amd import ops, optim, parallel
        self.a, self.rank, self.world, self.device = a, rank, world, device
        self.model = model = build_model(a, device)
        # config 5: the two 2M-row tables (item embedding, vocabulary-major sampled projection) take their Adam update row by
        # row, when a row is used (bit-identical to the dense update; `--dense_adam` for the A/B)
        lazy = [p for n, p in model.named_parameters() if 'embedding_layers' in n or n == 'head.output_embedding'] \
            if (a.sampled and not a.dense_adam) else []
        self.opt = optim.Adam(model.parameters(), order=backward_order(model), lazy_rows=lazy)
        arena = self.opt.arena
        emb_start = min(arena.slice_of(p)[0] for n, p in model.named_parameters() if 'embedding_layers' in n)
        # (head parameters placed behind the tables -- the projection when its sweep runs in the background -- belong to the last bucket)
        head_end = max(arena.slice_of(p)[1] for n, p in model.named_parameters()
                       if n.startswith('head.') and n != 'head.output_embedding' and arena.slice_of(p)[0] < emb_start)
        self.tables = [p for n, p in model.named_parameters() if 'embedding_layers' in n]
        # config 5: the 2M-row tables' gradients travel as (indices, rows) instead of a 2 GB dense all-reduce
        self.sparse = (self.tables + [model.head.output_embedding]) if (a.sampled and world > 1) else []
        self.reducer = parallel.GradReducer(arena, bucket_bounds=[head_end, emb_start], reduce='sum', sparse_params=self.sparse)
        self.specials = torch.tensor([3, 4], device=device)
        self.one = torch.ones((), dtype=torch.float32, device=device)   # d loss / d loss, built once (backward() would fill one per step)
        self.batches = make_batches(a, rank, device)
        self.ops = ops

    def step(self, i):
        a, b = self.a, self.batches[i % len(self.batches)]
        model, reducer = self.model, self.reducer
        # host-side facts the launch recorder's algorithmic counts use (nothing on the device depends on them)
        self.ops.set_record_hints(token_rows=a.batch * a.seq if a.dense else b['n_real'], sum_len_sq=b['sum_len_sq'],
                                  sum_q_len=b['sum_q_len'], adam_distinct_rows=b['distinct_ids'])
        self.opt.zero_grad()
        reducer.begin_backward()
        if a.host_flat_idx:
            loss = model.cloze_loss(b['feats'], b['labels'], training=True, flat_idx=b['flat_idx'], packed=False)
        else:       # device-side index generation + label compaction, cap = 10 rows per sequence, no host sync
            loss = model.cloze_loss(b['feats'], b['labels_padded'], training=True, max_masked_per_row=10,
                                    packed=False if a.dense else None, n_real_tokens=None if a.dense else b['n_real'])
        loss.backward(self.one)
        if self.sparse:
            for t, f in zip(self.tables, b['feats'].values()):
                reducer.set_touched_rows(t, torch.cat([f.reshape(-1), self.specials]))
            reducer.set_touched_rows(model.head.output_embedding, model.head.touched_rows())
        reducer.finish()
        self.opt.step(reducer.grad_mul)
        return loss


def visible_gpus():
    """Number of GPUs this process may use, WITHOUT bringing up the HIP / HSA runtime (the launcher must stay a process that has
    never touched the GPU): the KFD topology nodes that have SIMDs (CPUs are nodes too, with simd_count 0), narrowed by the
    *_VISIBLE_DEVICES variables the runtime honours.  None when the topology cannot be read."""
    import glob
    n = 0
    nodes = glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties')
    if not nodes:
        return None
    for f in nodes:
        try:
            with open(f) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get('simd_count', '0')) > 0:
            n += 1
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def self_launch(a):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: this process -- which has made no GPU call -- starts
    the N ranks (torch.distributed.run on 127.0.0.1, one process per GPU over RCCL) as a child, passes its output through
    and exits with its code.  Fewer devices than ranks (a one-GPU box): the ranks share the devices round-robin and the
    exchange goes over gloo -- a rehearsal of the N-rank path, labelled as such in the line's `config.parallelism`.
    Under a profiler's preloaded library the GPU is initialised before this program's first line: a launcher that starts
    other programs is then exactly the hop the pool forbids -- profiled runs are single-rank."""
    import socket
    import subprocess
    pre = ' '.join(os.environ.get(k, '') for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB'))
    if 'rocprof' in pre or any(k.startswith('ROCPROFILER_') or k.startswith('ROCPROF_') for k in os.environ):
        raise SystemExit('bench.py: --gpus %d under a profiler: profiled runs are single-rank (start the ranks with '
                         'torch.distributed.run yourself and profile one of them)' % a.gpus)
    ndev = visible_gpus()
    if ndev is None:                          # no KFD topology to read: let the ranks find out (they fall back to gloo themselves)
        ndev = a.gpus
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', B4C_SELF_LAUNCHED='1')
    if ndev < a.gpus:
        env.setdefault('B4C_DIST_BACKEND', 'gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(a))
    from bert4clickpath_amd import ops, parallel
    rank, local, world = parallel.init_distributed()
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (no CPU fallback)'
    ndev = torch.cuda.device_count()
    local = local % ndev      # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    if a.materialised_logits:
        ops.flash_ce = False
    for kv in filter(None, a.ops_flags.split(',')):
        k, v = kv.split('=')
        assert hasattr(ops, k), 'unknown ops flag %s' % k
        setattr(ops, k, bool(int(v)))
    tr = Training(a, rank, world, device)
    model, batches, step = tr.model, tr.batches, tr.step

    for i in range(a.warmup):
        loss = step(i)
    # HIP events around every hot-path launch of the LAST nrec timed steps (rank 0): the steady state, whatever the run's
    # length (the first steps of a run are not yet in the clip regime of the vocabulary head)
    nrec = a.steps if a.record_steps < 0 else min(a.record_steps, a.steps)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]     # one event per step boundary
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        if i == a.steps - nrec and rank == 0 and nrec > 0:
            ops.start_recording()
        loss = step(a.warmup + i)
        marks[i + 1].record()
    fams = ops.pause_recording() if (rank == 0 and nrec > 0) else None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=device, dtype=torch.float64)
    rr = torch.tensor([sum(batches[(a.warmup + i) % len(batches)]['R'] for i in range(a.steps))], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
    dt, items = float(tt[0]), float(rr[0])

    if rank == 0:
        fams = ops.stop_recording(fams) if fams is not None else None
        peak_tf = MFMA_BF16_PEAK_TF if a.dtype == 'bf16' else 157.3
        roof = roofline_from(fams, max(nrec, 1), peak_tf) if fams else None
        tj = a.traffic_json or os.path.join(ROOT, 'profiles', 'traffic.json' if a.config == 'c2' else 'traffic_%s.json' % a.config)
        traffic = None
        if roof and os.path.exists(tj):      # rocprofv3 PMC passes (separate runs), HBM bytes per launch of each family
            with open(tj) as f:
                traffic = json.load(f)
            cfg = traffic.get('_config', {})
            mine = {'vocab': a.vocab, 'batch': a.batch, 'seq': a.seq, 'd_model': a.d_model, 'layers': a.layers, 'dtype': a.dtype}
            if not (a.traffic_json or all(cfg.get(k) == v for k, v in mine.items())):     # counters are valid for that config only
                traffic = None
        def measured(fam):
            """PMC bytes of a family -- attached only when they can be a measurement of the SAME launches: HBM bytes below the
            launches' algorithmic bytes (less 5 % for counter granularity) mean the file was made from other launches."""
            t = traffic.get(fam) if traffic else None
            v = fams.get(fam) if fams else None
            if not t or not v or not v['launches']:
                return None
            alg = v['bytes'] / v['launches']
            if t.get('hbm_bytes_per_launch', 0.0) < 0.95 * alg and not t.get('on_chip_reuse'):
                return {'refused': 'PMC bytes per launch %.3g below the algorithmic %.3g: not a measurement of these launches'
                                   % (t.get('hbm_bytes_per_launch', 0.0), alg)}
            return t
        if roof and traffic:
            roof['traffic'] = measured(roof['family'])
            if 'beside' in roof:        # the background sweep's measured HBM bytes beside its algorithmic bytes
                for f, row in roof['beside']['families'].items():
                    row['traffic'] = traffic.get(f)
        # per-step durations on the device timeline (events at the step boundaries; steps with the launch recorder on are
        # ~0.4 ms longer): SURVEY 8d asks for the median and p10 / p90
        raw_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
        step_ms = sorted(raw_ms[:a.steps - nrec] or raw_ms)       # (the unrecorded steps when there are any)
        slowest = max(range(a.steps), key=lambda i: raw_ms[i])

        def pct(q):
            return step_ms[min(len(step_ms) - 1, int(round(q * (len(step_ms) - 1))))]
        shared = ndev < world
        out = {
            'metric': 'masked-items/sec (whole node)', 'value': items / dt, 'unit': 'masked-items/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if a.dtype == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': 'BERT4Rec Cloze training step: vocab=%d seq_len=%d d_model=%d layers=%d heads=%d dff=%d%s '
                                   'head=[1024,512,256,128]->V%s batch=%d seq/GPU x %d GPU, 10 masked/seq, dropout=%.2f, Zipf(1.1) ids, '
                                   '%d resident batches, [MASK] indices %s%s; encoder on %s'
                                   % (a.vocab, a.seq, a.d_model, a.layers, a.heads, a.dff,
                                      ' (reference-hard-coded)' if a.dff == 100 else ' (NOT the reference\'s 100: the paper-faithful width)',
                                      (' (sampled softmax, %d shared log-uniform negatives; Adam on the 2M-row tables %s)'
                                       % (a.sampled, 'as a dense pass' if a.dense_adam else 'row-lazy: bit-identical to the dense pass')) if a.sampled else '',
                                      a.batch, world, a.dropout, len(batches),
                                      'precomputed on the host' if a.host_flat_idx else 'generated on the device inside the step',
                                      ('; two SUMMED features items(%d)+actions(%d, vocab %d) (feature_combine=sum: an extension, the reference concatenates)'
                                       % (a.d_model, a.d_model, a.action_vocab) if a.feature_sum else
                                       '; two concatenated features items(%d)+actions(%d, vocab %d)' % (a.d_model - a.action_dim, a.action_dim, a.action_vocab))
                                      if a.action_dim > 0 else '',
                                      'the padded (B, S) layout' if a.dense else
                                      'the padding-free layout (%.0f %% of the B x S positions are real tokens)%s'
                                      % (100.0 * sum(b['n_real'] for b in batches) / (len(batches) * a.batch * a.seq),
                                         (', last layer evaluated at the [MASK] rows only' if ops.mq_last_layer else '') +
                                         ((', vocabulary dW sweep as a background kernel beside the encoder backward'
                                           if ops.background_dw_expected(a.d_model, a.dff, model.compute_dtype) else
                                           ', feed-forward and projection backward as fused single-pass kernels, vocabulary dW sweep in the foreground')
                                          if ops.flash_ce and not a.sampled else ''))),
                       'global_batch': a.batch * world, 'seq_len': a.seq,
                       'parallelism': ('dp%d' % world) + (' (REHEARSAL: %d ranks share %d device(s), gradient exchange over gloo -- not an '
                                                           'RCCL / xGMI measurement)' % (world, ndev) if shared else ''),
                       'grad_reduce': 'sum (reference semantics)'},
            'tokens_per_s': a.batch * world * a.seq * a.steps / dt,
            'step_ms': {'median': pct(0.5), 'p10': pct(0.1), 'p90': pct(0.9), 'min': step_ms[0], 'max': step_ms[-1], 'slowest_step': slowest,
                        'recorded_steps': 'the last %d of the %d timed steps carry the launch recorder (+~0.4 ms each) and are left out of '
                                          'these percentiles' % (nrec, a.steps) if 0 < nrec < a.steps else None},
            'final_loss': float(loss.detach()),
            'roofline': roof,
        }
        if roof and roof['bound'] == 'mfma':
            # the step's largest HBM-bound family next to it (the two lead the table within a few percent of each other)
            bal = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
            hb = {f: v for f, v in fams.items() if v['bytes'] > 0 and v['flops'] / v['bytes'] <= bal and not f.endswith('_bg')}
            if hb:
                r2 = roofline_from(fams, max(nrec, 1), peak_tf, dom=max(hb, key=lambda f: hb[f]['ms']))
                r2.pop('families', None)
                r2.pop('beside', None)
                if traffic:
                    r2['traffic'] = measured(r2['family'])
                out['roofline_largest_hbm_family'] = r2
        if world == 1 and a.eval_steps > 0:
            out['eval'] = eval_leg(model, batches, a, peak_tf)
        if world == 1 and a.full_steps > 0 and not a.dense:
            # the same training step with EVERY position of EVERY layer computed, as the reference's dataflow does (padded
            # layout, full last layer): same loss and gradients, the work whose results nothing reads included
            prev = ops.mq_last_layer
            ops.mq_last_layer = False
            a.dense = True
            try:
                for i in range(3):
                    step(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.full_steps):
                    step(i)
                e1.record()
                torch.cuda.synchronize()
                fms = e0.elapsed_time(e1) / a.full_steps
                Rf = sum(batches[i % len(batches)]['R'] for i in range(a.full_steps)) / a.full_steps
                out['every_position_of_every_layer'] = {
                    'what': 'the same step on the padded layout with the full last layer (what the reference computes); '
                            'identical loss and gradients', 'steps': a.full_steps, 'ms_per_step': fms,
                    'masked_items_per_s': Rf / fms * 1e3}
            finally:
                ops.mq_last_layer = prev
                a.dense = False
        if world == 1 and not a.no_cpu_baseline and a.action_dim == 0:
            out['cpu_baseline'] = cpu_baseline(a)
        print(json.dumps(out))
        ops.dump_family_log()          # (B4C_FAMILY_LOG: the launch -> family notes scratch/pmc_traffic.py pairs PMC dispatches with)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
