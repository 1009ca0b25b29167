"""The model-level oracle (oracle/cti_models.py) against the reference's own outputs (tests/golden/g9_*, g12_*),
captured by tests/golden/make_golden_models.py.  CPU only."""
import numpy as np
import pytest

from golden_util import load, model_case
from oracle import cti_models as M
from oracle.cti_oracle import norm_max_err

TOL = 2e-5          # fp32 restatement vs the fp32 reference: summation-order noise only


@pytest.mark.parametrize("op", ["c", "none"])
def test_word_embedding(op):
    fx, p = model_case("g12_wordemb_op%s" % op)
    out = M.word_embedding(fx.i["x"], p)
    assert out.shape == fx.o["out"].shape
    assert np.array_equal(out, fx.o["out"])              # a gather: bit-exact


def test_gru_forward_all_and_last():
    fx, p = model_case("g12_gru")
    out = M.gru_forward_all(fx.i["x"], p)
    assert norm_max_err(out, fx.o["out_all"]) < TOL
    assert norm_max_err(M.question_embedding(fx.i["x"], p), fx.o["out_last"]) < TOL
    out64 = M.gru_forward_all(fx.i["x"], p, dtype=np.float64)
    assert norm_max_err(out64, fx.o["out_all"]) < TOL


@pytest.mark.parametrize("act", ["relu", "swish"])
def test_simple_classifier(act):
    fx, p = model_case("g12_classifier_%s" % act)
    assert norm_max_err(M.simple_classifier(fx.i["x"], p, activation=act), fx.o["out"]) < TOL


def test_losses():
    fx = load("g12_losses")
    bce = M.bce_with_logits_sum(fx.i["x"], fx.i["target"])
    assert abs(bce - float(fx.o["bce_sum"])) < 1e-5 * abs(bce)
    kd = M.distillation_loss(fx.i["x"], fx.i["knowledge"], fx.i["target"], fx.cfg["T"], fx.cfg["alpha"])
    assert abs(kd - float(fx.o["kd"])) < 1e-5 * abs(kd)


def test_ffoe_cti_model_logits():
    fx, p = model_case("g9_ffoe_cti")
    out = M.ffoe_cti_forward(fx.i["v"], fx.i["q"], fx.i["ans"], p, fx.cfg["args"]["gamma"])
    assert out.shape == fx.o["logits"].shape
    assert norm_max_err(out, fx.o["logits"]) < 5e-5


def test_ffoe_ban_model_logits_and_att():
    fx, p = model_case("g9_ffoe_ban")
    out, att = M.ffoe_ban_forward(fx.i["v"], fx.i["q"], p, fx.cfg["args"]["gamma"])
    assert norm_max_err(att, fx.o["att"]) < 5e-5
    assert norm_max_err(out, fx.o["logits"]) < 5e-5


def test_mc_tan_model_logits_and_att():
    fx, p = model_case("g9_mc_cti")
    out, att = M.mc_tan_forward(fx.i["v"], fx.i["q"], fx.i["ans"], p, fx.cfg["args"]["gamma"])
    assert att.shape == fx.o["att"].shape
    assert norm_max_err(att, fx.o["att"]) < 5e-5
    assert norm_max_err(out, fx.o["logits"]) < 5e-5


def test_train_step_fixture_is_consistent():
    """g10: the forward logits of the training-step fixture are the eval logits of the same model; the loss is BCE/B."""
    fx, p = model_case("g10_ffoe_cti_step")
    out = M.ffoe_cti_forward(fx.i["v"], fx.i["q"], fx.i["ans"], p, fx.cfg["args"]["gamma"])
    assert norm_max_err(out, fx.o["logits"]) < 5e-5
    loss = M.bce_with_logits_sum(fx.o["logits"], fx.i["target"]) / fx.cfg["B"]
    assert abs(loss - float(fx.o["loss"])) < 1e-5 * abs(loss)
    assert float(fx.o["loss_after"]) < float(fx.o["loss"])
