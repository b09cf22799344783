"""Helper of the bf16 end-to-end tests: record the ReLU on / off patterns of one HIP forward pass (the FFN of every encoder
layer and the head's trunk layers, in call order) and hand them to the oracle as its `relu` (oracle/torch_ref.py)."""
import torch

# Every end-to-end bf16 gradient test holds every parameter tensor to this L2 bound against the fp64 oracle evaluated with the
# device path's own ReLU on / off patterns (measured: worst tensor 1.1 % over three seeds, scratch/bf16_err.py; without the
# shared patterns the same comparison reads 5 - 11 %: sqrt of the ~0.5 % of units whose pre-activation changes sign under
# bf16 rounding -- a property of ReLU networks, not of the kernels).
BF16_GRAD_BOUND = 0.03


def grad_errors(named_params, ref_grads):
    """{name: L2 relative error} for every parameter whose gradient is not identically zero (the key bias: a softmax row is
    invariant to it; what either side holds there is rounding noise)"""
    out = {}
    for n, p in named_params:
        gr = ref_grads[n]
        if n.endswith('mha.wk.bias') or float(gr.abs().max()) < 1e-9:
            continue
        out[n] = float((p.grad.detach().cpu().double() - gr).norm() / gr.norm())
    return out


class GateRecorder:
    def __init__(self, ops):
        self.ops, self.patterns = ops, []
        self.rows_only_last = False       # the pass ran its last encoder layer at the [MASK] rows only (ops.MQAttnBlockFn)

    def __enter__(self):
        from bert4clickpath_amd import _lib as L
        self._orig = self.ops.gemm_nt

        def gemm_nt(a, bt, n, bias=None, act=L.ACT_NONE, **kw):
            out = self._orig(a, bt, n, bias, act, **kw)
            if act == L.ACT_RELU:
                self.patterns.append((out.detach() > 0).cpu())
            return out
        self.ops.gemm_nt = gemm_nt
        # the fused feed-forward forward (ops.ffn_fwd, bf16 / d_model 128 / >= 4,096 rows) never calls gemm_nt: its h is the pattern
        self._orig_ffn = self.ops.ffn_fwd

        def ffn_fwd(*a, **kw):
            res = self._orig_ffn(*a, **kw)
            self.patterns.append((res[0].detach() > 0).cpu())
            return res
        self.ops.ffn_fwd = ffn_fwd
        # which form the last layer took cannot be read off its pattern's row count: B x max_masked_per_row rows of the masked-query
        # form and T rows of a packed full layer can coincide (B = 4, S = 16, 40 real tokens, 10 masks per row: found by a drawn case)
        self._orig_mq = self.ops.MQAttnBlockFn
        rec = self

        class Noting(self._orig_mq):
            @staticmethod
            def apply(*a, **k):
                rec.rows_only_last = True
                return rec._orig_mq.apply(*a, **k)
        self.ops.MQAttnBlockFn = Noting
        return self

    def __exit__(self, *exc):
        self.ops.gemm_nt = self._orig
        self.ops.ffn_fwd = self._orig_ffn
        self.ops.MQAttnBlockFn = self._orig_mq

    def relu_for(self, num_layers, n_head_layers, rows_flat, B, S, token_rows=None):
        """-> relu(name, z) for the oracle.  patterns: one per FFN in layer order ([B*S or T or R rows][F padded]), then one per head
        trunk layer ([R][width]).  A rows-only last layer (masked-query form) recorded its pattern at the [MASK] rows only:
        the other positions of that layer reach no output, the oracle keeps its own there.  token_rows (packed layout): the dense
        position b*S + s of each of the T packed rows -- the pad positions, which the packed layout does not compute and whose
        gradient is zero, keep the oracle's own pattern.
        Every call also books, per layer, how far the device's pattern is from the oracle's own (self.flips[name] = (fraction of
        the compared units whose gate differs, largest |pre-activation| among those units in the oracle / rms pre-activation)):
        a kernel error that shifts pre-activations shows up here even when the loss barely moves (check_flips)."""
        pats = list(self.patterns)
        assert len(pats) == num_layers + n_head_layers, (len(pats), num_layers, n_head_layers)
        self.flips = {}

        def relu(name, z):
            kind, i = name.split('.')
            pat = pats[int(i)] if kind == 'ffn' else pats[num_layers + int(i)]
            own = z.detach() > 0
            if kind == 'ffn':
                F = z.shape[-1]
                flat = own.reshape(-1, F).clone()
                rows_only = self.rows_only_last and int(i) == num_layers - 1
                if not rows_only and pat.shape[0] == flat.shape[0]:
                    rows = None
                    flat = pat[:, :F]
                elif not rows_only and token_rows is not None and pat.shape[0] == token_rows.shape[0]:
                    rows = token_rows
                    flat[rows] = pat[:, :F]
                else:                                   # rows-only pattern of the last layer
                    # (the sync-free form runs on B x max_masked_per_row rows: the real ones first, in mask order, zero rows after)
                    assert pat.shape[0] >= rows_flat.shape[0], (pat.shape, rows_flat.shape)
                    pat = pat[:rows_flat.shape[0]]
                    rows = rows_flat
                    flat[rows] = pat[:, :F]
                mask = flat.reshape(z.shape)
                zc, oc, mc = z.detach().reshape(-1, F), own.reshape(-1, F), flat
                if rows is not None:
                    zc, oc, mc = zc[rows], oc[rows], mc[rows]
            else:
                mask = pat[:z.shape[0], :z.shape[-1]]
                zc, oc, mc = z.detach(), own, mask
            diff = oc != mc
            rms = float(zc.double().pow(2).mean().sqrt()) or 1.0
            self.flips[name] = (float(diff.double().mean()), float(zc[diff].abs().max()) / rms if bool(diff.any()) else 0.0)
            return z * mask.to(z.dtype)
        return relu

    def check_flips(self, max_fraction=0.02, max_preactivation=0.25):
        """The device pass and the fp64 oracle may disagree on a ReLU gate only where the oracle's pre-activation is close to zero
        (bf16 rounding of the layer's input moves it across): at most `max_fraction` of a layer's units (measured ~0.5 %), each
        with |pre-activation| below `max_preactivation` x the layer's rms pre-activation.  A wrong bias or a dropped K slice
        flips units far from zero and fails here, although the loss moves by less than its 2e-3 bound."""
        assert self.flips, 'relu_for(...) has not been used by an oracle pass yet'
        for name, (frac, far) in self.flips.items():
            assert frac <= max_fraction, 'layer %s: %.2f %% of the ReLU gates differ from the fp64 oracle\'s' % (name, 100 * frac)
            assert far <= max_preactivation, ('layer %s: a unit with |pre-activation| = %.3f x rms has the other gate than the '
                                              'fp64 oracle' % (name, far))
        return self.flips
