"""Dropout keep masks and the one-launch weight folds against plain torch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dropout_mask_statistics_and_determinism(cuda):
    from magic_amd import _lib
    n, w = 200_003, 7          # element count not a multiple of 16: ragged tail
    m1 = _lib.dropout_mask((n, w), 0.9, 1234, cuda)
    m2 = _lib.dropout_mask((n, w), 0.9, 1234, cuda)
    m3 = _lib.dropout_mask((n, w), 0.9, 1235, cuda)
    assert m1.dtype == torch.uint8 and m1.shape == (n, w)
    assert int(m1.max()) == 1 and int(m1.min()) == 0
    assert torch.equal(m1, m2)                       # a function of (seed, index) only
    assert not torch.equal(m1, m3)
    p = float(m1.float().mean())
    sigma = (0.9 * 0.1 / (n * w)) ** 0.5
    assert abs(p - 0.9) < 6 * sigma
    # independent of the tensor shape: same seed, same flat prefix
    m4 = _lib.dropout_mask((n * w,), 0.9, 1234, cuda)
    assert torch.equal(m1.view(-1), m4)
    # neighbouring elements are uncorrelated
    a, b = m1.view(-1)[:-1].float(), m1.view(-1)[1:].float()
    corr = float(((a - p) * (b - p)).mean()) / (p * (1 - p))
    assert abs(corr) < 0.01
    for keep in (0.0, 1.0):
        m = _lib.dropout_mask((1000,), keep, 7, cuda)
        assert float(m.float().mean()) == keep


@pytest.mark.parametrize("H,D,Dp,F,e_dim,res,bias", [(4, 16, 16, 5, 17, True, True), (4, 16, 16, 64, 17, False, True),
                                                      (1, 5, 8, 64, 9, False, True), (4, 8, 8, 12, 3, True, False)])
def test_fold_matches_torch(cuda, H, D, Dp, F, e_dim, res, bias):
    from magic_amd.model.gat import _FoldFn
    g = torch.Generator().manual_seed(3)
    mk = lambda *s: torch.randn(*s, generator=g).to(cuda).requires_grad_()  # noqa: E731
    C = H * D
    w_fc, ah, at, w_efc, ae = mk(C, F), mk(1, H, D), mk(1, H, D), mk(C, e_dim), mk(1, H, D)
    w_res = mk(C, F) if res else None
    b = mk(1, H, D) if bias else None
    out = _FoldFn.apply(w_fc, ah, at, w_res, w_efc, ae, b, H, D, Dp)

    def fold(w, a):
        return (w.view(H, D, -1) * a.view(H, D, 1)).sum(1)

    def pad(w):
        wp = torch.zeros(H, Dp, w.shape[1], device=cuda)
        wp[:, :D] = w.view(H, D, -1)
        return wp.view(H * Dp, -1)

    parts = [pad(w_fc), fold(w_fc, ah), fold(w_fc, at)] + ([pad(w_res)] if res else [])
    ref = [torch.cat(parts, 0), fold(w_efc, ae).t()]
    if bias:
        ref.append(pad(b.view(C, 1)).view(-1))
    assert len(out) == len(ref)
    for o, r in zip(out, ref):
        assert o.shape == r.shape
        assert torch.allclose(o, r, rtol=1e-5, atol=1e-6)
    cot = [torch.randn(o.shape, generator=g).to(cuda) for o in out]
    params = [p for p in (w_fc, ah, at, w_res, w_efc, ae, b) if p is not None]
    g_mine = torch.autograd.grad(out, params, cot)
    g_ref = torch.autograd.grad(ref, params, cot)
    for gm, gr in zip(g_mine, g_ref):
        assert gm.shape == gr.shape
        assert torch.allclose(gm, gr, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("w", [192, 64, 384, 1024, 1100, 10, 514])   # element-wise form below 512 columns or unaligned, float4 form: four slabs, two passes
def test_rows_scatter_add_with_repeats(cuda, w):
    from magic_amd import _lib
    g = torch.Generator().manual_seed(11)
    n, s = 500, 3000
    idx = torch.randint(0, n, (s,), generator=g)
    idx[:40] = 7                                          # a long run of one row
    idx[100:330:1] = 11                                   # a hub row: a run longer than three 64-position scan steps
    src = torch.randn(s, w, generator=g)
    base = torch.randn(n, w + 8, generator=g)[:, :w]      # strided destination rows
    want = base.double().clone().index_add_(0, idx, src.double())
    dst = base.to(cuda)
    dst = torch.empty(n, w + 8, device=cuda)[:, :w].copy_(dst)
    out1 = _lib.rows_scatter_add(dst.clone(), idx.to(cuda), src.to(cuda))
    out2 = _lib.rows_scatter_add(dst.clone(), idx.to(cuda), src.to(cuda))
    assert torch.equal(out1, out2)
    assert float((out1.double().cpu() - want).abs().max()) < 1e-4
    untouched = torch.ones(n, dtype=torch.bool)
    untouched[idx] = False
    assert torch.equal(out1[untouched.to(cuda)], dst[untouched.to(cuda)])
