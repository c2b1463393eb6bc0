"""Dense projection kernels (forward, input gradient with every epilogue part, weight gradient)
against float64 torch on the same inputs.

The forward / input-gradient kernels form each fp32 product from three bf16 pieces per operand
(six of the nine cross products); the bound below -- 2e-6 of sum_k |a_k||b_k| per output element,
i.e. a few fp32 roundings of the largest partial sum -- is the grade of an fp32 fmaf chain.
"""
import pytest
import torch

from magic_amd import _lib

pytestmark = pytest.mark.gpu

TOL = 2e-6   # 32 fp32 roundings of sum |a||b|: sequential fp32 chains over K = 256 reach about half of it


def _wide(shape, gen, dev):
    """values over ~6 decades of magnitude, both signs"""
    mag = torch.exp2(torch.randint(-10, 10, shape, generator=gen, device=dev).float())
    return torch.randn(shape, generator=gen, device=dev) * mag


def _bound(a, b):
    return (a.abs().double() @ b.abs().double()) * TOL + 1e-30


@pytest.mark.parametrize("n,f,c,ns,rc,mask,strided", [
    (1000, 5, 64, 8, 64, True, False),      # layer 0 of the 64-wide entity models
    (4097, 64, 64, 8, 0, True, True),       # hidden layer, input is a column block of a wider buffer
    (333, 192, 64, 0, 0, False, False),     # encoder_to_decoder
    (2050, 64, 8, 2, 0, True, False),       # decoder (head width padded to 8)
    (257, 8, 256, 8, 256, True, False),     # layer 0 of the 256-wide batch models
    (515, 256, 256, 8, 0, True, True),
    (31, 11, 64, 8, 64, False, False),      # fewer rows than one tile
    (100, 13, 16, 8, 16, True, False),
    (64, 1024, 256, 0, 0, False, False),    # wide-layer GEMM kernel
    (2250, 384, 64, 0, 0, False, False),    # edge-reconstruction head, 64-wide models: long K against 64 outputs
    (1301, 2048, 256, 0, 0, False, False),  # ... 256-wide models
    (700, 320, 64, 0, 0, True, False),
])
def test_proj_fwd_against_float64(cuda, n, f, c, ns, rc, mask, strided):
    gen = torch.Generator(device=cuda).manual_seed(n * 7 + f)
    if strided:
        buf = _wide((n, 3 * f), gen, cuda)
        x = buf[:, f:2 * f]
    else:
        x = _wide((n, f), gen, cuda)
    w = _wide((c + ns + rc, f), gen, cuda)
    m = (torch.rand(n, f, generator=gen, device=cuda) < 0.9).to(torch.uint8) if mask else None
    scale = 1.0 / 0.9 if mask else 1.0
    hs, score, res = _lib.proj_fwd(x, w, c, ns, rc, x_mask=m, x_scale=scale)
    xe = (x * m * scale) if mask else x
    ref = xe.double() @ w.double().t()
    got = torch.cat([t for t in (hs, score, res) if t is not None], 1)
    bound = _bound(xe, w.t())
    assert got.shape == ref.shape
    err = (got.double() - ref).abs()
    assert bool((err <= bound).all()), float((err / bound).max())
    # reproducible
    hs2, score2, res2 = _lib.proj_fwd(x, w, c, ns, rc, x_mask=m, x_scale=scale)
    assert torch.equal(hs, hs2)


@pytest.mark.parametrize("n,k,f,full", [
    (1000, 72, 64, True),      # hidden layer, all epilogue parts
    (1000, 72, 64, False),
    (4099, 64, 192, False),    # encoder_to_decoder input gradient
    (700, 18, 64, False),      # decoder
    (130, 264, 256, True),     # 256-wide model: wide-layer GEMM kernel with the fused epilogue
    (130, 136, 5, False),      # layer-0 input gradient (narrow output)
    (77, 256, 1024, False),    # wide-layer GEMM kernel
    (2250, 64, 384, False),    # edge-reconstruction head, 64-wide models: many outputs from a short K
    (1301, 256, 2048, False),  # ... 256-wide models
    (333, 72, 260, False),
])
def test_proj_bwd_input_against_float64(cuda, n, k, f, full):
    gen = torch.Generator(device=cuda).manual_seed(n + k + f)
    dy = _wide((n, k), gen, cuda)
    w = _wide((k, f), gen, cuda)
    ref = dy.double() @ w.double()
    bound = _bound(dy, w)
    if not full:
        dx, cs, pg = _lib.proj_bwd_input(dy, w)
        assert cs is None and pg is None
        err = (dx.double() - ref).abs()
        assert bool((err <= bound).all()), float((err / bound).max())
        return
    add = torch.randn(n, f, generator=gen, device=cuda)
    add2 = torch.randn(n, f, generator=gen, device=cuda)
    m = (torch.rand(n, f, generator=gen, device=cuda) < 0.9).to(torch.uint8)
    scale = 1.0 / 0.9
    for slope in (0.25, -0.5):      # positive: pre-activation recovered from the output; else read
        pw = torch.tensor([slope], device=cuda)
        pre = torch.randn(n, f, generator=gen, device=cuda)
        act = torch.where(pre > 0, pre, pre * slope)
        dx, cs, pg = _lib.proj_bwd_input(dy, w, add=add, out_mask=m, out_scale=scale, add2=add2, act_out=act,
                                         act_pre=pre, prelu_w=pw, want_col_sum=True)
        v = (ref + add.double()) * m.double() * scale + add2.double()
        neg = pre <= 0
        # pre-activation as the kernel sees it (act / slope in fp32 when the slope is positive)
        pre_k = torch.where(neg, act * (1.0 / pw) if slope > 0 else pre, pre)
        pg_ref = (v * pre_k.double() * neg).sum()
        v = torch.where(neg, v * slope, v)
        b2 = bound * scale + 1e-6 * (add.abs() + add2.abs()).double() + 1e-12
        err = (dx.double() - v).abs()
        assert bool((err <= 2 * b2).all()), float((err / b2).max())
        assert torch.allclose(cs.double(), v.sum(0), rtol=1e-5, atol=1e-5 * float(v.abs().sum(0).max()))
        assert abs(float(pg) - float(pg_ref)) <= 1e-5 * float((v.abs() * pre_k.abs().double()).sum())


@pytest.mark.parametrize("n,m_,f,mask", [
    (5000, 72, 64, True), (5000, 64, 192, False), (3001, 18, 64, True), (2000, 136, 5, True),
    (1500, 264, 256, True), (999, 136, 10, False), (40, 72, 64, False),
    (1500, 256, 1024, False), (901, 256, 384, True), (77, 128, 128, True),   # block-tiled kernel of the wide layers
    (2250, 64, 384, False), (1301, 256, 2048, False),                        # edge-reconstruction head
])
def test_proj_bwd_weight_against_float64(cuda, n, m_, f, mask):
    gen = torch.Generator(device=cuda).manual_seed(n + m_ + f)
    dy = _wide((n, m_), gen, cuda)
    x = _wide((n, f), gen, cuda)
    m = (torch.rand(n, f, generator=gen, device=cuda) < 0.9).to(torch.uint8) if mask else None
    scale = 1.0 / 0.9 if mask else 1.0
    dw = _lib.proj_bwd_weight(dy, x, x_mask=m, x_scale=scale)
    xe = (x * m * scale) if mask else x
    ref = dy.double().t() @ xe.double()
    bound = _bound(dy.t(), xe)
    err = (dw.double() - ref).abs()
    assert bool((err <= bound).all()), float((err / bound).max())
    assert torch.equal(dw, _lib.proj_bwd_weight(dy, x, x_mask=m, x_scale=scale))


@pytest.mark.parametrize("n,F,H", [(5000, 256, 4), (777, 1024, 4), (3000, 128, 2)])
def test_head_scores_from_the_output_rows(cuda, n, F, H):
    """magic_proj_fwd_attn: wide layers with 64-wide heads form eh / et as (feat * attn).sum(-1) over the head's outputs
    (model/gat.py:291-292) in the projection's epilogue instead of through folded weight rows: same hs bit for bit,
    scores equal to float64 to the bound of the other projection kernels; shapes the shortcut does not serve fall back
    to the folded rows."""
    from magic_amd import _lib
    D = 64
    C = H * D
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, F, generator=g).to(cuda)
    w_fc = torch.randn(C, F, generator=g).to(cuda) / F ** 0.5
    a_h, a_t = torch.randn(1, H, D, generator=g).to(cuda), torch.randn(1, H, D, generator=g).to(cuda)
    fold = lambda a: (w_fc.view(H, D, F) * a.view(H, D, 1)).sum(1)   # noqa: E731
    w_aug = torch.cat([w_fc, fold(a_h), fold(a_t)], 0).contiguous()
    mask = (torch.rand(n, F, generator=g) < 0.9).to(torch.uint8).to(cuda)
    hs0, sc0, _ = _lib.proj_fwd(x, w_aug, C, 2 * H, 0, x_mask=mask, x_scale=1 / 0.9)
    hs1, sc1, _ = _lib.proj_fwd(x, w_aug, C, 2 * H, 0, x_mask=mask, x_scale=1 / 0.9, attn_src=a_h, attn_dst=a_t)
    assert torch.equal(hs0, hs1)
    xd = (x.double() * mask.double() / 0.9)
    hsd = xd @ w_fc.double().t()
    want = torch.cat([(hsd.view(n, H, D) * a_h.double()).sum(-1), (hsd.view(n, H, D) * a_t.double()).sum(-1)], 1)
    mag = torch.cat([(xd.abs() @ w_fc.double().abs().t()).view(n, H, D).mul(a.double().abs()).sum(-1) for a in (a_h, a_t)], 1)
    for sc in (sc0, sc1):
        assert float(((sc.double() - want).abs() / mag).max()) < 4e-6
    # a linear residual (extra output columns) is not served by the shortcut: the folded rows are used
    w_res = torch.randn(C, F, generator=g).to(cuda)
    w2 = torch.cat([w_aug, w_res], 0).contiguous()
    h2, s2, r2 = _lib.proj_fwd(x, w2, C, 2 * H, C, attn_src=a_h, attn_dst=a_t)
    h3, s3, r3 = _lib.proj_fwd(x, w2, C, 2 * H, C)
    assert torch.equal(h2, h3) and torch.equal(s2, s3) and torch.equal(r2, r3)
