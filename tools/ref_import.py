"""Import the reference's Python layers in THIS container (build container only).

Recipe of SURVEY.md Appendix B: the reference's pointnet2/_ext is CUDA-only, so
the CPU oracle façade (oracle/oracle_ext.py) is injected as ``pointnet2._ext``;
``models/__init__.py`` is bypassed (it imports termcolor/ipdb/tensorboardX, all
absent); RoBERTa is a random-init RobertaConfig model and the tokenizer a seeded
fake.  Nothing of the reference is copied into the repo -- only arrays produced
by running it (tests/golden/*.npz) are.  /root/reference does not exist on the
GPU box, so only tools/ scripts import this module, never tests or the package.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    sys.dont_write_bytecode = True
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import oracle_ext
    oracle_ext.build()
    sys.path[:0] = [REF, os.path.join(REF, "pointnet2")]
    import pointnet2  # namespace package
    sys.modules["pointnet2._ext"] = oracle_ext
    pointnet2._ext = oracle_ext
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    mods = types.SimpleNamespace()
    mods.pointnet2_utils = importlib.import_module("pointnet2_utils")
    mods.pointnet2_modules = importlib.import_module("pointnet2_modules")
    mods.edl = importlib.import_module("models.encoder_decoder_layers")
    mods.backbone = importlib.import_module("models.backbone_module")
    mods.modules = importlib.import_module("models.modules")
    mods.bdetr = importlib.import_module("models.bdetr")
    return mods


class FakeTokenized(dict):
    def __init__(self, ids, mask):
        super().__init__(input_ids=ids, attention_mask=mask)
        self.attention_mask = mask
        self.input_ids = ids

    def to(self, device):
        return FakeTokenized(self["input_ids"].to(device), self["attention_mask"].to(device))


class FakeTokenizer:
    """Maps each 'text' (a string holding an int seed) to seeded token ids."""
    max_len = 16

    @classmethod
    def from_pretrained(cls, *a, **k):
        return cls()

    def batch_encode_plus(self, texts, padding="longest", return_tensors="pt"):
        from eda_amd import synthetic
        ids, mask = synthetic.utterance_tokens(int(texts[0]), len(texts), max_len=self.max_len)
        return FakeTokenized(torch.from_numpy(ids), torch.from_numpy(mask))


def fake_roberta_factory(seed=0):
    from transformers import RobertaConfig, RobertaModel

    class FakeRoberta:
        @classmethod
        def from_pretrained(cls, *a, **k):
            torch.manual_seed(seed)
            cfg = RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1,
                                pad_token_id=1)
            m = RobertaModel(cfg)
            m.eval()
            return m
    return FakeRoberta


def build_reference_model(mods, seed=0, **kw):
    """Instantiate the reference BeaUTyDETR with fake tokenizer / random RoBERTa.
    cwd must be /root/reference so data/class_embeddings3d.npy resolves."""
    mods.bdetr.RobertaTokenizerFast = FakeTokenizer
    mods.bdetr.RobertaModel = fake_roberta_factory(seed)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        torch.manual_seed(seed)
        model = mods.bdetr.BeaUTyDETR(data_path="", **kw)
    finally:
        os.chdir(cwd)
    return model
