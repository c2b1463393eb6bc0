"""The entry point that allocates nothing (``magic_sce_fwd_bwd_static``): called through the C ABI
and compared bit for bit with ``magic_sce_fwd_bwd`` on two calls in a row, the second with a
smaller input (the static scratch keeps the first call's contents)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib, _lib, static, x, target, ld_t, mask, alpha, want_dt):
    n, n_dim = x.shape
    dev = x.device
    loss = torch.full((1,), -7.0, device=dev)
    d_x = torch.full_like(x, 3.0)
    d_t = torch.full((n_dim,), 5.0, device=dev) if want_dt else None
    stream = torch.cuda.current_stream().cuda_stream
    common = (n, n_dim, mask.numel(), mask.data_ptr(), x.data_ptr(), x.stride(0), target.data_ptr(), ld_t,
              float(alpha), 1.0, loss.data_ptr(), d_x.data_ptr(), d_t.data_ptr() if want_dt else None)
    if static:
        st = lib.magic_sce_fwd_bwd_static(*common, stream)
    else:
        nbytes = lib.magic_sce_workspace_bytes(mask.numel(), n_dim)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        st = lib.magic_sce_fwd_bwd(*common, ws.data_ptr(), ws.numel(), stream)
    torch.cuda.synchronize()
    return st, loss, d_x, d_t


@pytest.mark.parametrize("single", [True, False])
def test_static_entry_equals_workspace_entry_twice(cuda, single):
    from magic_amd import _lib
    lib = _lib.load()
    torch.manual_seed(11)
    # first call: large enough for the full 1024-block grid; second call: smaller input, fewer blocks
    for n, n_dim in ((600_000, 13), (3_000, 5)):
        x = torch.randn(n, n_dim, device=cuda)
        mask = torch.randperm(n, device=cuda)[: n // 2].contiguous()
        if single:
            target, ld_t = torch.randn(n_dim, device=cuda) * 0.1, 0
        else:
            target = torch.nn.functional.one_hot(torch.randint(0, n_dim, (n,), device=cuda), n_dim).float()
            ld_t = target.stride(0)
        st_w, loss_w, dx_w, dt_w = _run(lib, _lib, False, x, target, ld_t, mask, 3.0, single)
        st_s, loss_s, dx_s, dt_s = _run(lib, _lib, True, x, target, ld_t, mask, 3.0, single)
        assert st_w == 0 and st_s == 0
        assert torch.equal(loss_s, loss_w) and torch.equal(dx_s, dx_w)
        if single:
            assert torch.equal(dt_s, dt_w)
        assert torch.isfinite(loss_s).all() and float(loss_s) > 0


def test_static_entry_rejects_rows_wider_than_its_scratch(cuda):
    from magic_amd import _lib
    lib = _lib.load()
    n, n_dim = 256, 17
    x = torch.randn(n, n_dim, device=cuda)
    mask = torch.arange(0, n, 2, device=cuda)
    target = torch.randn(n_dim, device=cuda)
    st, loss, d_x, _ = _run(lib, _lib, True, x, target, 0, mask, 3.0, False)
    assert st == -6 and lib.magic_status_string(st).startswith(b"input exceeds the static scratch")
    assert float(loss) == -7.0 and bool((d_x == 3.0).all())      # nothing was written
    st, *_ = _run(lib, _lib, False, x, target, 0, mask, 3.0, False)
    assert st == 0                                               # the workspace entry point serves it
