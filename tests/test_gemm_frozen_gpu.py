"""csrc/gemm_frozen.hip: row products against a frozen weight's bf16 x 3 planes (the text encoder's 48 linear layers,
models/bdetr.py:77-80, 170-175) against fp64 -- under the SAME bound the fp32-MFMA row products meet on the same inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(640, 768, 2304), (640, 768, 768), (640, 768, 3072), (640, 3072, 768), (1040, 768, 768), (64, 64, 64), (1, 128, 64),
          (77, 192, 128), (130, 3072, 768)]


@pytest.mark.parametrize("R,K,N", SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_frozen_planes_product_vs_fp64(R, K, N, act):
    from eda_amd import gemm
    torch.manual_seed(R + K + N)
    dev = "cuda"
    x = torch.randn(R, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    planes = gemm.frozen_planes(w)
    # the planes reproduce the weight exactly: v = h + m + l
    p = planes.view(torch.bfloat16).float()
    assert torch.equal(p[0] + p[1] + p[2], w)
    y = gemm.linear_frozen(x, planes, b, act=act)
    y32 = gemm.linear_fwd(x, w, b, relu=act)
    ref = x.double() @ w.double().t() + b.double()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    scale = ref.abs().max().item()
    e_b3 = (y.double() - ref).abs().max().item() / scale
    e_32 = (y32.double() - ref).abs().max().item() / scale
    print(f"{R}x{K}x{N} act {act}: bf16x3 {e_b3:.2e}  fp32 MFMA {e_32:.2e}")
    assert e_b3 <= max(2.0 * e_32, 3e-7)
    # no bias, an output with a row stride, rows past the last 64-row block untouched
    out = torch.full((R + 3, N + 8), float("nan"), device=dev)
    gemm.linear_frozen(x, planes, None, act=0, out=out[:R, :N])
    assert torch.equal(out[:R, :N], gemm.linear_frozen(x, planes)) and torch.isnan(out[R:]).all() and torch.isnan(out[:, N:]).all()


def test_frozen_roberta_path_equals_the_fp32_mfma_path_to_fp32_rounding(monkeypatch):
    """The whole frozen encoder (12 layers) on the planes against the same module on the fp32-MFMA row products."""
    from eda_amd import roberta_fast
    from tests import model_fixtures as MF
    torch.manual_seed(0)
    m = MF.small_roberta(2).cuda().eval()
    ids = torch.randint(3, 1000, (4, 24), device="cuda")
    ids[:, 0] = 0
    ids[2, 17:] = 1
    am = ids.ne(1).long()
    outs = {}
    for mode in ("2", "1", "0"):
        monkeypatch.setenv("EDA_FROZEN_B3", mode)
        m.__dict__.pop("_eda_fast", None)
        outs[mode] = roberta_fast.encode(m, ids, am)
        assert (m.__dict__["_eda_fast"].layers[0]["planes"] is not None) == (mode != "0")
    scale = outs["0"].abs().max().item()
    assert (outs["1"] - outs["0"]).abs().max().item() <= 2e-5 * scale
    assert (outs["2"] - outs["0"]).abs().max().item() <= 2e-5 * scale
