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

    def __enter__(self):
        from bert4clickpath_amd import _lib as L
        self._orig = self.ops.gemm_nt

        def gemm_nt(a, bt, n, bias=None, act=L.ACT_NONE, **kw):
            out = self._orig(a, bt, n, bias, act, **kw)
            if act == L.ACT_RELU:
                self.patterns.append((out.detach() > 0).cpu())
            return out
        self.ops.gemm_nt = gemm_nt
        return self

    def __exit__(self, *exc):
        self.ops.gemm_nt = self._orig

    def relu_for(self, num_layers, n_head_layers, rows_flat, B, S):
        """-> relu(name, z) for the oracle.  patterns: one per FFN in layer order ([B*S or T or R rows][F padded]), then one per head
        trunk layer ([R][width]).  A rows-only last layer (masked-query form) recorded its pattern at the [MASK] rows only:
        the other positions of that layer reach no output, the oracle keeps its own there."""
        pats = list(self.patterns)
        assert len(pats) == num_layers + n_head_layers, (len(pats), num_layers, n_head_layers)

        def relu(name, z):
            kind, i = name.split('.')
            pat = pats[int(i)] if kind == 'ffn' else pats[num_layers + int(i)]
            own = z.detach() > 0
            if kind == 'ffn':
                F = z.shape[-1]
                flat = own.reshape(-1, F).clone()
                if pat.shape[0] == flat.shape[0]:
                    flat = pat[:, :F]
                else:                                   # rows-only pattern of the last layer
                    assert pat.shape[0] == rows_flat.shape[0], (pat.shape, rows_flat.shape)
                    flat[rows_flat] = pat[:, :F]
                mask = flat.reshape(z.shape)
            else:
                mask = pat[:, :z.shape[-1]]
            return z * mask.to(z.dtype)
        return relu
