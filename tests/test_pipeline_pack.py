"""Host logic of the pipelined step's packed hand-over buffers (eda_amd/pipeline.py): views of one byte buffer stand in for
the tensors of a batch, whatever their dtypes, and one copy of the buffer moves all of them."""
import torch

from eda_amd import pipeline


def _batch():
    g = torch.Generator().manual_seed(3)
    return {
        "point_clouds": torch.randn(2, 37, 6, generator=g),
        "tokenized": {"input_ids": torch.randint(0, 50000, (2, 11), generator=g),
                      "attention_mask": torch.rand(2, 11, generator=g) > 0.3},
        "box_mask": torch.rand(2, 5, generator=g) > 0.5,
        "odd": torch.randn(3, 5, generator=g).t(),                    # not contiguous
        "labels": torch.randint(0, 9, (2, 5), generator=g, dtype=torch.int32),
        "name": "not a tensor",
    }


def test_packed_views_keep_values_dtypes_and_alignment():
    b = _batch()
    flat = pipeline._flat(b)
    buf, views = pipeline._packed_like(flat)
    assert buf.dtype == torch.uint8 and len(views) == len(flat)
    for v, t in zip(views, flat):
        assert v.shape == t.shape and v.dtype == t.dtype and v.is_contiguous()
        assert (v.data_ptr() - buf.data_ptr()) % 256 == 0
        v.copy_(t)
    for v, t in zip(views, flat):
        assert torch.equal(v, t)
    # disjoint: writing one view leaves the others alone
    views[0].zero_()
    for v, t in zip(views[1:], flat[1:]):
        assert torch.equal(v, t)


def test_one_copy_of_the_buffer_moves_every_tensor():
    b = _batch()
    flat = pipeline._flat(b)
    src, sv = pipeline._packed_like(flat)
    dst, dv = pipeline._packed_like(flat)
    for v, t in zip(sv, flat):
        v.copy_(t)
    dst.copy_(src)
    rebuilt = pipeline._rebuild(b, dv)
    assert rebuilt["name"] == "not a tensor"
    assert sorted(rebuilt) == sorted(b) and sorted(rebuilt["tokenized"]) == sorted(b["tokenized"])
    for k in ("point_clouds", "box_mask", "odd", "labels"):
        assert torch.equal(rebuilt[k], b[k]) and rebuilt[k].dtype == b[k].dtype
    for k in b["tokenized"]:
        assert torch.equal(rebuilt["tokenized"][k], b["tokenized"][k])
    assert [t.data_ptr() for t in pipeline._flat(rebuilt)] == [v.data_ptr() for v in dv]
