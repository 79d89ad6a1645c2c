"""csrc/wgrad.hip (dW = dY^T X and db = colsum(dY), split-K fp32 MFMA) against fp64 references."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (K, M, N)
    (2048, 288, 288), (8192, 864, 288), (640, 576, 288), (1056, 576, 288), (2048, 256, 288), (2048, 288, 256),
    (2048, 64, 288), (1, 288, 288), (63, 96, 96), (65, 100, 36), (130, 32, 32), (4096, 128, 64), (333, 292, 260),
]


@pytest.mark.parametrize("K,M,N", SHAPES)
def test_wgrad_vs_fp64(K, M, N):
    from eda_amd.nn_utils import wgrad
    torch.manual_seed(K + M + N)
    dy = torch.randn(K, M, device="cuda")
    x = torch.randn(K, N, device="cuda")
    dW, db = wgrad(dy, x)
    eW = dy.double().t() @ x.double()
    eb = dy.double().sum(0)
    tolW = 2e-5 * (dy.abs().double().t() @ x.abs().double()).max().item()
    assert (dW.double() - eW).abs().max().item() <= tolW
    assert (db.double() - eb).abs().max().item() <= 2e-5 * dy.abs().double().sum(0).max().item()
    dW2, db2 = wgrad(dy, x)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)          # fixed summation order
    dW3, none = wgrad(dy, x, want_db=False)
    assert none is None and torch.equal(dW3, dW)


def test_wgrad_strided_operands_and_slices():
    """Operands that are column slices of packed buffers, destinations that are row ranges."""
    from eda_amd.nn_utils import wgrad
    torch.manual_seed(5)
    K, d = 2048, 288
    P = torch.randn(K, 3 * d, device="cuda")
    x = torch.randn(K, d, device="cuda")
    dW = torch.full((3 * d, d), 7.0, device="cuda")
    db = torch.full((3 * d,), 7.0, device="cuda")
    wgrad(P[:, d:3 * d], x, dW=dW[d:3 * d], db=db[d:3 * d])
    eW = P[:, d:].double().t() @ x.double()
    assert (dW[d:].double() - eW).abs().max().item() <= 2e-5 * (P[:, d:].abs().double().t() @ x.abs().double()).max().item()
    assert (db[d:].double() - P[:, d:].double().sum(0)).abs().max().item() <= 1e-3
    assert (dW[:d] == 7).all() and (db[:d] == 7).all()


@pytest.mark.parametrize("K,M,N", [(2048, 3, 288), (2048, 1, 288), (8192, 4, 256), (77, 2, 64), (1, 3, 288),
                                   (100000, 3, 128)])
def test_wgrad_few_output_channels(K, M, N):
    """1-4 output channels (box centre / size / objectness heads): weighted column sums (colsum.hip)."""
    from eda_amd.nn_utils import wgrad
    torch.manual_seed(K + M)
    dy = torch.randn(K, M, device="cuda")
    x = torch.randn(K, N, device="cuda")
    for _ in range(2):                                   # second call: the ticket counters were left at zero
        dW, db = wgrad(dy, x)
        eW = dy.double().t() @ x.double()
        assert (dW.double() - eW).abs().max().item() <= 2e-5 * (dy.abs().double().t() @ x.abs().double()).max().item()
        assert (db.double() - dy.double().sum(0)).abs().max().item() <= 2e-5 * dy.abs().double().sum(0).max().item() + 1e-6
    dW2, none = wgrad(dy, x, want_db=False)
    assert none is None and torch.equal(dW2, dW)


def test_wgrad_fallback_shapes():
    """Shapes neither kernel takes (input width not a multiple of 4) go through the library."""
    from eda_amd.nn_utils import wgrad
    dy = torch.randn(2048, 3, device="cuda")
    x = torch.randn(2048, 6, device="cuda")
    dW, db = wgrad(dy, x)
    torch.testing.assert_close(dW, dy.t() @ x, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db, dy.sum(0), rtol=1e-4, atol=1e-3)
