"""Row-lazy Adam (optim.LazyRows over b4c_adam_rows) against the dense kernel (b4c_adam_step): Keras' Adam is dense-equivalent
(main.py:87; SURVEY 8c iii: the moments of every row decay every step), and a row-sparse table under the lazy form must come out
BIT-IDENTICAL -- parameters and both moments -- once its rows are caught up, whatever the pattern of touches, repeats, long
gaps, learning-rate changes and rotation slices.  Gradients are written into the arenas directly, so the comparison is of the
optimizers alone (the training kernels' float-atomic reductions would differ between two runs before any optimizer ran)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(rows, width, n_dense, staleness, seed):
    from bert4clickpath_amd import optim
    g = torch.Generator().manual_seed(seed)
    out = []
    for lazy in (False, True):
        g.manual_seed(seed)
        dense = torch.nn.Parameter(torch.randn(n_dense, generator=g).cuda())
        table = torch.nn.Parameter((torch.randn(rows, width, generator=g) * 0.05).cuda())
        tail = torch.nn.Parameter(torch.randn(37, generator=g).cuda())
        opt = optim.Adam([dense, table, tail], lazy_rows=[table] if lazy else (), max_staleness=staleness)
        out.append((opt, dense, table, tail))
    return out


@pytest.mark.parametrize('rows,width,staleness', [(5000, 256, 7), (300, 64, 256), (70000, 128, 1000), (64, 8, 3)])
def test_lazy_rows_equal_the_dense_update_bit_for_bit(rows, width, staleness):
    (od, dd, td, ld), (ol, dl, tl, ll) = _pair(rows, width, 1000, staleness, seed=rows)
    rng = np.random.default_rng(rows + width)
    hot = rng.integers(0, rows, 5)
    for step in range(1, 26):
        if step == 12:
            od.lr = ol.lr = 3.17e-4                      # ReduceLROnPlateau in mid-run (main.py:134)
        # the rows this step reads: hot rows (every step, repeated many times), a few random ones, sometimes out-of-range ids
        n = int(rng.integers(1, 400))
        ids = np.concatenate([rng.integers(0, rows, n), np.repeat(hot, 50)])
        if step % 5 == 0:
            ids = np.concatenate([ids, [-3, rows + 9]])
        if step in (7, 8, 9):
            ids = hot[:1].copy()                          # steps that touch almost nothing
        ids_t = torch.from_numpy(ids).cuda()
        od.zero_grad()
        ol.zero_grad()
        tl._b4c_lazy.catch_up(ids_t.view(1, -1))          # what the forward pass's gather sites do
        # rows read by the "forward" must be current: compare them with the dense run's rows right now
        uniq = np.unique(np.clip(ids, 0, rows - 1))
        assert torch.equal(tl.detach()[uniq], td.detach()[uniq]), step
        # a gradient for a SUBSET of the rows read (a row that was read need not receive one), identical in both arenas
        got = uniq[rng.random(uniq.size) < 0.8]
        grad_rows = torch.from_numpy(rng.standard_normal((got.size, width)).astype(np.float32)).cuda()
        gd = torch.from_numpy(rng.standard_normal(1000).astype(np.float32)).cuda()
        for opt, d, t, l in ((od, dd, td, ld), (ol, dl, tl, ll)):
            t.grad[torch.from_numpy(got).cuda()] = grad_rows
            d.grad.copy_(gd)
            l.grad.fill_(0.25)
        mul = 0.125 if step % 4 == 0 else 1.0
        od.step(mul)
        ol.step(mul)
        assert torch.equal(dl.detach(), dd.detach()) and torch.equal(ll.detach(), ld.detach())
        assert float(tl.grad.abs().max()) == 0.0          # the step leaves the table's gradient zeroed
    lz = tl._b4c_lazy
    stale = int((lz.stamp != ol.iterations).sum())
    assert stale > 0 or staleness <= 25, 'every row is current already: the catch-up in front of state_dict is not exercised'
    sd = ol.state_dict()                                  # catches every row up
    torch.cuda.synchronize()
    assert torch.equal(tl.detach(), td.detach())
    lo, hi = ol.arena.slice_of(tl)
    lo_d, hi_d = od.arena.slice_of(td)
    assert torch.equal(sd['m'][lo:hi], od.m[lo_d:hi_d]) and torch.equal(sd['v'][lo:hi], od.v[lo_d:hi_d])
    assert torch.equal(ol.arena.flat, od.arena.flat)


def test_a_step_that_never_ran_leaves_the_gradient_to_a_full_zero_fill():
    """zero_grad() skips the lazy tables only when the last step completed (it zeroed the rows it consumed); after a step that
    failed in backward the whole gradient arena is cleared as before."""
    (od, dd, td, ld), (ol, dl, tl, ll) = _pair(200, 16, 64, 4, seed=1)
    ol.zero_grad()
    tl.grad[3] = 1.0
    ol.zero_grad()                                        # no step in between
    assert float(tl.grad.abs().max()) == 0.0
    ids = torch.tensor([3, 5], device='cuda')
    tl._b4c_lazy.catch_up(ids)
    tl.grad[3] = 1.0
    ol.step()
    assert float(tl.grad.abs().max()) == 0.0 and ol._grads_clean
    ol.zero_grad()
    assert not ol._grads_clean and not tl._b4c_lazy.touched
