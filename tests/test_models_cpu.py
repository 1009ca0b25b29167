"""Host-side checks of the model-level mirrors (no GPU): constructor signatures and state_dict layout against the key lists
the reference produced (stored in the g9 / g12 fixtures by tests/golden/make_golden_models.py)."""
import types

import pytest
import torch

import cti_amd
from golden_util import load


class _DS:
    def __init__(self, ntoken, v_dim, num_ans):
        self.dictionary = types.SimpleNamespace(ntoken=ntoken)
        self.v_dim = v_dim
        self.num_ans_candidates = num_ans


def _keys(m):
    return [(k, list(v.shape)) for k, v in m.state_dict().items()]


@pytest.mark.parametrize("name,builder", [("g9_ffoe_cti", "build_cti"), ("g9_ffoe_ban", "build_ban"), ("g9_mc_cti", "build_mc_cti")])
def test_model_state_dict_matches_reference(name, builder):
    fx = load(name)
    c = fx.cfg
    args = types.SimpleNamespace(**c["args"])
    m = getattr(cti_amd, builder)(args, _DS(c["ntoken"], c["v_dim"], c["num_ans"]))
    ref = [(k, list(s)) for k, s in c["state_keys"]]
    ours = _keys(m)
    if builder == "build_ban":          # the reference registers c_prj only with a counter; same here
        assert not any(k.startswith("c_prj") for k, _ in ours)
    assert ours == ref
    # same seed => the same initial parameters as the reference's builder (its constructors' RNG consumption order is mirrored)
    torch.manual_seed(0)
    m0 = getattr(cti_amd, builder)(args, _DS(c["ntoken"], c["v_dim"], c["num_ans"]))
    for k, v in m0.state_dict().items():
        ref_sum = c["init_sums"][k]
        assert abs(float(v.double().sum()) - ref_sum) <= 1e-6 * max(1.0, abs(ref_sum)), k
    frozen = sorted(k for k, p in m.named_parameters() if not p.requires_grad)
    assert frozen == sorted(k for k, _ in ref if k.endswith("emb_.weight"))


def test_unit_module_state_dicts():
    for name, make in [
        ("g12_wordemb_opc", lambda c: cti_amd.WordEmbedding(c["ntoken"], 300, 0.0, c["op"])),
        ("g12_wordemb_opnone", lambda c: cti_amd.WordEmbedding(c["ntoken"], 300, 0.0, c["op"])),
        ("g12_gru", lambda c: cti_amd.QuestionEmbedding(c["in_dim"], c["num_hid"], 1, False, 0.0)),
        ("g12_classifier_relu", lambda c: cti_amd.SimpleClassifier(c["in_dim"], c["hid_dim"], c["out_dim"],
                                                                    types.SimpleNamespace(activation="relu", dropout=0.5))),
        ("g12_classifier_swish", lambda c: cti_amd.SimpleClassifier(c["in_dim"], c["hid_dim"], c["out_dim"],
                                                                     types.SimpleNamespace(activation="swish", dropout=0.5))),
    ]:
        c = load(name).cfg
        assert _keys(make(c)) == [(k, list(s)) for k, s in c["state_keys"]], name


def test_same_seed_same_initial_parameters_as_reference_recipe():
    """The builders consume torch's RNG in the reference's construction order: two builds from the same seed agree, and the
    embedding padding rows are zero like nn.Embedding(padding_idx) leaves them."""
    c = load("g9_ffoe_cti").cfg
    args = types.SimpleNamespace(**c["args"])
    torch.manual_seed(7)
    a = cti_amd.build_cti(args, _DS(c["ntoken"], c["v_dim"], c["num_ans"]))
    torch.manual_seed(7)
    b = cti_amd.build_cti(args, _DS(c["ntoken"], c["v_dim"], c["num_ans"]))
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    assert float(a.w_emb.emb.weight[c["ntoken"]].abs().sum()) == 0.0


def test_unsupported_pieces_raise():
    with pytest.raises(AssertionError):
        cti_amd.SimpleClassifier(4, 8, 2, types.SimpleNamespace(activation="gelu", dropout=0.5))
    c = load("g9_ffoe_ban").cfg
    a = dict(c["args"]); a["use_counter"] = True
    with pytest.raises(NotImplementedError):
        cti_amd.build_ban(types.SimpleNamespace(**a), _DS(c["ntoken"], c["v_dim"], c["num_ans"]))
    q = cti_amd.QuestionEmbedding(8, 4, 1, True, 0.0)
    with pytest.raises(NotImplementedError):
        q.forward_all(torch.zeros(1, 2, 8))
    with pytest.raises(cti_amd.CtiError):           # no CPU path
        cti_amd.WordEmbedding(5, 300, 0.0, "")(torch.zeros(1, 2, dtype=torch.int64))


def test_replication_factor_from_the_row_equality_bytes():
    """ops.replication_of (host side of TanModel.v_replication = 'auto'): the largest r dividing the batch with every row b, b % r != 0, equal to its predecessor."""
    import torch
    from cti_amd import ops
    f = lambda bits: ops.replication_of(torch.tensor(bits, dtype=torch.uint8))
    assert f([0, 1, 1, 1, 0, 1, 1, 1]) == 4
    assert f([0, 1, 0, 1, 0, 1, 0, 1]) == 2
    assert f([0, 1, 1, 1, 1, 1, 1, 1]) == 8                      # one image for the whole batch
    assert f([0, 1, 1, 0, 1, 1]) == 3
    assert f([0, 1, 1, 1, 0, 1, 1, 0]) == 1                      # ragged groups: no de-duplication
    assert f([0, 0, 0, 0]) == 1
    assert f([0]) == 1
