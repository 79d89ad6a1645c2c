"""The same model-layer parity cases on the product path: HIP kernels through the
C ABI on an MI355X, against goldens from the reference's Python layers."""
import pytest

import model_cases as MC

pytestmark = pytest.mark.gpu


def test_query_and_group():
    MC.run_query_and_group("cuda")


def test_sa_and_fp_modules():
    MC.run_fp_module("cuda")


def test_sa_module_train_mode():
    MC.run_sa_module_train("cuda")


def test_backbone():
    MC.run_backbone("cuda")


@pytest.mark.parametrize("butd", [True, False])
def test_encoder_decoder(butd):
    MC.run_encoder_decoder("cuda", butd)


@pytest.mark.parametrize("butd", [True, False])
def test_full_model(butd):
    MC.run_full_model("cuda", butd)
