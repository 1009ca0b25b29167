"""Loading of the committed golden vectors (tests/golden/*.npz) and the frozen-stream parameter / input
regeneration used by the two BASELINE-shaped fixtures.  `rs_fill` / `rs_state` must stay identical to the
functions of the same name in tests/golden/make_golden.py (the script that ran the reference)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.cfg = json.loads(str(z["cfg"]))
        self.p, self.i, self.o, self.g = {}, {}, {}, {}
        for k in z.files:
            if k == "cfg":
                continue
            pre, rest = k.split("/", 1)
            getattr(self, pre)[rest] = z[k]


def load(name):
    return Fixture(name)


def rs_fill(rs, shape, kind):
    x = rs.standard_normal(size=shape).astype(np.float32)
    if kind == "abs":
        x = np.abs(x)
    elif kind == "tanh":
        x = np.tanh(x)
    elif kind.startswith("scale:"):
        x = x * np.float32(float(kind.split(":")[1]))
    return x


def rs_state(keys_shapes, seed):
    rs = np.random.RandomState(seed)
    shapes = {k: tuple(s) for k, s in keys_shapes}
    new = {}
    for k, shape in keys_shapes:
        shape = tuple(shape)
        if k.endswith("weight_v"):
            x = rs_fill(rs, shape, "scale:%r" % float(1.0 / np.sqrt(shape[-1])))
        elif k.endswith("weight_g"):
            n_out = shapes[k[:-1] + "v"][0]
            x = np.float32((abs(rs.standard_normal()) + 0.5) * np.sqrt(n_out))
        elif k.endswith("h_mat_g"):
            x = np.float32(abs(rs.standard_normal()) + 0.5)
        elif k.endswith("bias") and not k.endswith("h_bias"):
            x = rs_fill(rs, shape, "scale:0.1")
        else:
            x = rs_fill(rs, shape, "scale:1.0")
        new[k] = np.asarray(x, dtype=np.float32).reshape(shape)
    return new


def c1_case():
    """BASELINE config 1 inputs/params, regenerated exactly as make_golden.py:tc_c1 built them."""
    fx = load("g3_tcnet_forward_c1")
    c = fx.cfg
    params = rs_state([(k, tuple(s)) for k, s in c["state_keys"]], c["seed"])
    rs = np.random.RandomState(c["seed"] + 1)
    v = rs_fill(rs, (c["B"], c["V"], c["v_dim"]), "abs")
    q = rs_fill(rs, (c["B"], c["Q"], c["q_dim"]), "scale:1.0")
    a = rs_fill(rs, (c["B"], c["A"], c["a_dim"]), "scale:1.0")
    for b, r in c["zero_from"].items():
        v[int(b), int(r):] = 0
    return fx, params, v, q, a


def c2_sample_index(n_total, n_pick, seed):
    """Identical copy of tests/golden/make_golden.py:c2_sample_index."""
    return np.sort(np.random.RandomState(seed).choice(n_total, size=n_pick, replace=False)).astype(np.int64)


def c2_case():
    """BASELINE configs[1] widths at B=3 (fixture g3_tcnet_forward_c2): (fixture, params, v, q, a, sample index) regenerated exactly as
    make_golden.py:tc_c2 built them."""
    fx = load("g3_tcnet_forward_c2")
    c = fx.cfg
    params = rs_state([(k, tuple(s)) for k, s in c["state_keys"]], c["seed"])
    rs = np.random.RandomState(c["seed"] + 1)
    v = rs_fill(rs, (c["B"], c["V"], c["v_dim"]), "abs")
    q = rs_fill(rs, (c["B"], c["Q"], c["q_dim"]), "scale:1.0")
    a = rs_fill(rs, (c["B"], c["A"], c["a_dim"]), "scale:1.0")
    for b, r in c["zero_from"].items():
        v[int(b), int(r):] = 0
    idx = c2_sample_index(c["B"] * c["V"] * c["Q"] * c["A"] * c["glimpse"], c["n_sample"], c["seed"] + 2)
    return fx, params, v, q, a, idx


def c4_bi_case():
    fx = load("g7_biattention_c4")
    c = fx.cfg
    params = rs_state([(k, tuple(s)) for k, s in c["state_keys"]], c["seed"])
    rs = np.random.RandomState(c["seed"] + 1)
    v = rs_fill(rs, (c["B"], c["V"], c["x_dim"]), "abs")
    q = rs_fill(rs, (c["B"], c["Q"], c["y_dim"]), "tanh")
    for b, r in c["zero_from"].items():
        v[int(b), int(r):] = 0
    return fx, params, v, q


def model_state(keys_shapes, seed):
    """Parameters of the model-level fixtures (g9/g10/g12): identical copy of tests/golden/make_golden_models.py:model_state."""
    rs = np.random.RandomState(seed)
    shapes = {k: tuple(s) for k, s in keys_shapes}
    new = {}

    def normal(shape, scale):
        return (rs.standard_normal(size=shape).astype(np.float32) * np.float32(scale)).reshape(shape)

    for k, shape in keys_shapes:
        shape = tuple(shape)
        if ".rnn.weight" in k:
            x = normal(shape, 1.0 / np.sqrt(shape[-1]))
        elif ".rnn.bias" in k:
            x = normal(shape, 0.1)
        elif k.endswith("emb.weight") or k.endswith("emb_.weight"):
            x = normal(shape, 0.5)
        elif k.endswith("weight_v"):
            x = normal(shape, 1.0 / np.sqrt(shape[-1]))
        elif k.endswith("weight_g"):
            x = np.float32((abs(rs.standard_normal()) + 0.5) * np.sqrt(shapes[k[:-1] + "v"][0]))
        elif k.endswith("h_mat_g"):
            x = np.float32(abs(rs.standard_normal()) + 0.5)
        elif k.endswith("bias") and not k.endswith("h_bias"):
            x = normal(shape, 0.1)
        else:
            x = normal(shape, 1.0)
        new[k] = np.asarray(x, dtype=np.float32).reshape(shape)
    return new


def model_case(name):
    """(fixture, params) of a model-level fixture: parameters regenerated from the (key, shape) list in its cfg."""
    fx = load(name)
    params = model_state([(k, tuple(s)) for k, s in fx.cfg["state_keys"]], fx.cfg["seed"])
    return fx, params
