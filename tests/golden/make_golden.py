#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (the reference never travels to the GPU box):

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden.py

The process imports `src.*` from /root/reference (cwd) and nothing from this
repository.  Every fixture stores parameters + inputs + outputs (never an RNG
replay), except the two BASELINE-shaped cases (`*_c1`), whose parameters and
inputs are regenerated from `numpy.random.RandomState(seed)` -- a frozen legacy
stream -- by `tests/golden_util.py:rs_fill`, so that only the outputs are
stored.

Fixture keys: `p/<state_dict key>` parameters, `i/<name>` inputs, `o/<name>`
outputs, `g/<name>` gradients of `loss = sum(out * i/cot_<out>)`, `cfg` (JSON).
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
assert os.path.isdir("src") and os.path.isfile("src/tc.py"), "run with cwd=/root/reference"
sys.path.insert(0, os.getcwd())

from src.tc import TCNet  # noqa: E402
from src.bc import BCNet  # noqa: E402
from src.attention import BiAttention, TriAttention  # noqa: E402
from src.fc import FCNet  # noqa: E402
import src.Tensor as RefTensor  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def save(name, cfg, params=None, inputs=None, outputs=None, grads=None):
    d = {"cfg": np.array(json.dumps(cfg))}
    for pre, dd in (("p/", params), ("i/", inputs), ("o/", outputs), ("g/", grads)):
        for k, v in (dd or {}).items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().contiguous().numpy()
            v = np.asarray(v)
            d[pre + k] = v if v.ndim == 0 else np.ascontiguousarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-34s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


def sd(m):
    return {k: v.clone() for k, v in m.state_dict().items()}


def rs_fill(rs, shape, kind):
    """Must stay identical to tests/golden_util.py:rs_fill."""
    x = rs.standard_normal(size=shape).astype(np.float32)
    if kind == "abs":
        x = np.abs(x)
    elif kind == "tanh":
        x = np.tanh(x)
    elif kind.startswith("scale:"):
        x = x * np.float32(float(kind.split(":")[1]))
    return x


def rs_state(keys_shapes, seed):
    """Deterministic, torch-RNG-free parameters from (key, shape) pairs.  Must stay identical to
    tests/golden_util.py:rs_state.  weight_v ~ N(0, 1/fan_in); weight_g = c*sqrt(fan_out) so that the
    effective weight g*V/||V||_F has O(1/sqrt(fan_in)) entries; bias ~ 0.1*N; T_g/h_mat_v/h_bias ~ N(0,1)."""
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
        else:  # T_g, h_mat_v, h_bias
            x = rs_fill(rs, shape, "scale:1.0")
        new[k] = np.asarray(x, dtype=np.float32).reshape(shape)
    return new


def fill_state_from_rs(m, seed):
    ks = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in rs_state(ks, seed).items()})
    return m


def zero_rows(v, rows):
    for b, r in rows:
        v[b, r:] = 0
    return v


# ----------------------------------------------------------------------------------------------
# G1  ModeProduct: Kolda & Bader known answer (reference src/Tensor.py:30-35 with U_2 = U_3 = I)
# ----------------------------------------------------------------------------------------------
def g1():
    X = torch.tensor([[[1, 13], [4, 16], [7, 19], [10, 22]], [[2, 14], [5, 17], [8, 20], [11, 23]],
                      [[3, 15], [6, 18], [9, 21], [12, 24]]], dtype=torch.float32)
    U1 = torch.tensor([[1, 3, 5], [2, 4, 6]], dtype=torch.float32).unsqueeze(0)
    U2 = torch.eye(4).unsqueeze(0)
    U3 = torch.eye(2).unsqueeze(0)
    T = X.unsqueeze(0).unsqueeze(4)                       # (1,3,4,2,1)
    Y = RefTensor.ModeProduct(T, U1, U2, U3, None)
    exp0 = np.array([[22, 49, 76, 103], [28, 64, 100, 136]], np.float32)
    exp1 = np.array([[130, 157, 184, 211], [172, 208, 244, 280]], np.float32)
    y = Y.contiguous().numpy()
    assert y.shape == (1, 2, 4, 2, 1), y.shape
    assert np.array_equal(y[0, :, :, 0, 0], exp0) and np.array_equal(y[0, :, :, 1, 0], exp1)
    save("g1_modeproduct_kolda", {"ref": "src/Tensor.py:3-35"},
         inputs={"T": T, "U1": U1, "U2": U2, "U3": U3}, outputs={"Y": Y.contiguous()})
    # random, rectangular, batched, G in {1,2,3}
    g = torch.Generator().manual_seed(11)
    for G in (1, 2, 3):
        T = torch.randn(1, 5, 4, 4, G, 1, generator=g)
        M1 = torch.randn(3, 6, 5, generator=g)
        M2 = torch.randn(3, 2, 4, generator=g)
        M3 = torch.randn(3, 7, 4, generator=g)
        Y = RefTensor.ModeProduct(T, M1, M2, M3, None).contiguous()
        save("g1_modeproduct_rand_g%d" % G, {"G": G, "ref": "src/Tensor.py:3-28"},
             inputs={"T": T, "M1": M1, "M2": M2, "M3": M3}, outputs={"Y": Y})


# ----------------------------------------------------------------------------------------------
# G2  T_eff index maps: ModeProduct(arange-tensor, I, I, I) -> which T element lands where
# ----------------------------------------------------------------------------------------------
def g2():
    out = {}
    for hr, G in ((2, 1), (2, 2), (3, 2), (4, 3), (16, 1), (16, 2), (16, 4), (16, 8)):
        n = hr * hr * hr * G
        T = torch.arange(n, dtype=torch.float32).view(1, hr, hr, hr, G, 1)
        I = torch.eye(hr).unsqueeze(0)
        Y = RefTensor.ModeProduct(T, I, I, I, None).contiguous().view(hr, hr, hr, G)
        out["hr%d_g%d" % (hr, G)] = Y.numpy().astype(np.int32)
    save("g2_teff_index_maps", {"ref": "src/Tensor.py:3-28", "note": "value = flat index into T (hr,hr,hr,G)"},
         outputs=out)


# ----------------------------------------------------------------------------------------------
# G3/G4/G5/G8  TCNet.forward, TriAttention.forward, TCNet.forward_with_weights (+ grads)
# ----------------------------------------------------------------------------------------------
def tc_case(name, v_dim, q_dim, a_dim, h, R, G, k, B, V, Q, A, seed, zr):
    torch.manual_seed(seed)
    m = TriAttention(v_dim, q_dim, a_dim, h, 1, R, G, k).eval()
    g = torch.Generator().manual_seed(seed + 1)
    v = zero_rows(torch.randn(B, V, v_dim, generator=g).abs(), zr)
    q = torch.tanh(torch.randn(B, Q, q_dim, generator=g))
    a = torch.tanh(torch.randn(B, A, a_dim, generator=g))
    cfg = dict(v_dim=v_dim, q_dim=q_dim, a_dim=a_dim, h_dim=h, h_out=1, rank=R, glimpse=G, k=k,
               B=B, V=V, Q=Q, A=A, ref="src/tc.py:41-52, src/attention.py:49-59")
    # forward + grads of TCNet.forward (eval mode => dropout identity)
    v1, q1, a1 = (t.clone().requires_grad_(True) for t in (v, q, a))
    raw = m.TriAtt(v1, q1, a1)
    cot = torch.randn(raw.shape, generator=g)
    (raw * cot).sum().backward()
    grads = {"v": v1.grad, "q": q1.grad, "a": a1.grad}
    for n_, p_ in m.named_parameters():
        if p_.grad is not None:
            grads["p/" + n_] = p_.grad.clone()
    raw_c = raw.detach().contiguous()
    m.zero_grad()
    with torch.no_grad():
        p, logits = m(v, q, a)
    mask = (0 == v.abs().sum(2))
    save(name, cfg, params=sd(m), inputs={"v": v, "q": q, "a": a, "cot_raw": cot},
         outputs={"raw": raw_c, "p": p.contiguous(), "logits": logits.contiguous(), "mask": mask.numpy()},
         grads=grads)


def tc_act_case(name, act, seed):
    """TCNet with an activation other than ReLU (reference src/fc.py:24 takes any nn activation by name): forward + gradients."""
    v_dim, q_dim, a_dim, h, R, G, B, V, Q, A = 24, 20, 12, 32, 4, 2, 2, 4, 3, 3
    torch.manual_seed(seed)
    m = TCNet(v_dim, q_dim, a_dim, h, 1, R, G, act=act).eval()
    g = torch.Generator().manual_seed(seed + 1)
    v = torch.randn(B, V, v_dim, generator=g).abs()
    q = torch.tanh(torch.randn(B, Q, q_dim, generator=g))
    a = torch.tanh(torch.randn(B, A, a_dim, generator=g))
    v1, q1, a1 = (t.clone().requires_grad_(True) for t in (v, q, a))
    raw = m(v1, q1, a1)
    cot = torch.randn(raw.shape, generator=g)
    (raw * cot).sum().backward()
    grads = {"v": v1.grad, "q": q1.grad, "a": a1.grad}
    for n_, p_ in m.named_parameters():
        if p_.grad is not None:
            grads["p/" + n_] = p_.grad.clone()
    cfg = dict(v_dim=v_dim, q_dim=q_dim, a_dim=a_dim, h_dim=h, h_out=1, rank=R, glimpse=G, k=1, act=act, B=B, V=V, Q=Q, A=A,
               ref="src/tc.py:10-52 with act=%r (src/fc.py:24)" % act)
    save(name, cfg, params=sd(m), inputs={"v": v, "q": q, "a": a, "cot_raw": cot}, outputs={"raw": raw.detach().contiguous()}, grads=grads)


def tc_att_grad_case(name, seed):
    """Gradient THROUGH the masked softmax (TriAttention p), no zero-only samples."""
    v_dim, q_dim, a_dim, h, R, G, k, B, V, Q, A = 24, 20, 12, 32, 4, 2, 1, 2, 4, 3, 2
    torch.manual_seed(seed)
    m = TriAttention(v_dim, q_dim, a_dim, h, 1, R, G, k).eval()
    g = torch.Generator().manual_seed(seed + 1)
    v = zero_rows(torch.randn(B, V, v_dim, generator=g).abs(), [(1, 3)])
    q = torch.tanh(torch.randn(B, Q, q_dim, generator=g))
    a = torch.tanh(torch.randn(B, A, a_dim, generator=g))
    v1, q1, a1 = (t.clone().requires_grad_(True) for t in (v, q, a))
    p, logits = m(v1, q1, a1)
    cot = torch.randn(p.shape, generator=g)
    (p * cot).sum().backward()
    grads = {"v": v1.grad, "q": q1.grad, "a": a1.grad}
    for n_, p_ in m.named_parameters():
        grads["p/" + n_] = p_.grad.clone()
    cfg = dict(v_dim=v_dim, q_dim=q_dim, a_dim=a_dim, h_dim=h, h_out=1, rank=R, glimpse=G, k=k, B=B, V=V, Q=Q, A=A,
               ref="src/attention.py:49-59")
    save(name, cfg, params=sd(m), inputs={"v": v, "q": q, "a": a, "cot_p": cot},
         outputs={"p": p.detach().contiguous(), "logits": logits.detach().contiguous()}, grads=grads)


def tc_fww_case(name, v_dim, q_dim, a_dim, h, R, G, k, B, V, Q, A, seed):
    torch.manual_seed(seed)
    m = TCNet(v_dim, q_dim, a_dim, h, 1, R, G, dropout=[.2, .5], k=k).eval()
    g = torch.Generator().manual_seed(seed + 1)
    v = zero_rows(torch.randn(B, V, v_dim, generator=g).abs(), [(0, V - 1)])
    q = torch.tanh(torch.randn(B, Q, q_dim, generator=g))
    a = torch.tanh(torch.randn(B, A, a_dim, generator=g))
    att = torch.softmax(torch.randn(B, V * Q * A, G, generator=g), 1).view(B, V, Q, A, G)
    outs, grads = {}, {}
    v1, q1, a1, att1 = (t.clone().requires_grad_(True) for t in (v, q, a, att))
    cot = torch.randn(B, h * k, generator=g)
    o = m.forward_with_weights(v1, q1, a1, att1[:, :, :, :, 1])      # non-contiguous slice, as the callers pass it
    (o * cot).sum().backward()
    outs["out_g1"] = o.detach()
    grads.update({"v": v1.grad, "q": q1.grad, "a": a1.grad, "att": att1.grad})
    for n_, p_ in m.named_parameters():
        if p_.grad is not None:
            grads["p/" + n_] = p_.grad.clone()
    with torch.no_grad():
        outs["out_g0"] = m.forward_with_weights(v, q, a, att[:, :, :, :, 0])
    cfg = dict(v_dim=v_dim, q_dim=q_dim, a_dim=a_dim, h_dim=h, h_out=1, rank=R, glimpse=G, k=k, B=B, V=V, Q=Q, A=A,
               ref="src/tc.py:54-61")
    save(name, cfg, params=sd(m), inputs={"v": v, "q": q, "a": a, "att": att, "cot": cot}, outputs=outs, grads=grads)


def tc_c1():
    """BASELINE config 1 (B=4, V=36x2048, Q=14x600, A=4x300, rank 32, h_mm 512, glimpse 2); params from RandomState."""
    seed = 1204
    m = TriAttention(2048, 600, 300, 512, 1, 32, 2, 1).eval()
    fill_state_from_rs(m, seed)
    rs = np.random.RandomState(seed + 1)
    B, V, Q, A = 4, 36, 14, 4
    v = rs_fill(rs, (B, V, 2048), "abs")
    q = rs_fill(rs, (B, Q, 600), "scale:1.0")
    a = rs_fill(rs, (B, A, 300), "scale:1.0")
    v[0, 30:] = 0
    v[2, 17:] = 0
    with torch.no_grad():
        raw = m.TriAtt(torch.from_numpy(v), torch.from_numpy(q), torch.from_numpy(a)).contiguous()
        p, logits = m(torch.from_numpy(v), torch.from_numpy(q), torch.from_numpy(a))
    keys = [[k, list(t.shape)] for k, t in m.state_dict().items()]
    cfg = dict(v_dim=2048, q_dim=600, a_dim=300, h_dim=512, h_out=1, rank=32, glimpse=2, k=1, B=B, V=V, Q=Q, A=A,
               seed=seed, zero_from={"0": 30, "2": 17}, state_keys=keys,
               ref="BASELINE.json configs[0]; src/tc.py:41-52")
    save("g3_tcnet_forward_c1", cfg, outputs={"raw": raw, "p": p.contiguous(), "logits": logits.contiguous()})


def c2_sample_index(n_total, n_pick, seed):
    """The fixed sample of flat output positions stored by the configs[1] fixture.  Must stay identical to
    tests/golden_util.py:c2_sample_index."""
    return np.sort(np.random.RandomState(seed).choice(n_total, size=n_pick, replace=False)).astype(np.int64)


def tc_c2():
    """BASELINE config 2 widths -- the shape the headline metric is quoted on (V=36x2048, Q=14x1024, A=3129x300, rank 32,
    h_mm 512, glimpse 2) -- at B=3.  Params and inputs from RandomState (regenerated by tests/golden_util.py:c2_case); a
    sample's output is 3.15 M floats, so the fixture keeps a fixed 65 536-position sample of `raw` (TCNet.forward) and `p`
    per batch (positions over the flattened (B,V,Q,A,G) tensor), the per-(b,g) argmax of p, log-sum-exp of the masked
    logits and max|raw| -- enough to pin values, the mask and the softmax normalisation over 1.58 M positions."""
    seed = 2204
    m = TriAttention(2048, 1024, 300, 512, 1, 32, 2, 1).eval()
    fill_state_from_rs(m, seed)
    rs = np.random.RandomState(seed + 1)
    B, V, Q, A, G = 3, 36, 14, 3129, 2
    v = rs_fill(rs, (B, V, 2048), "abs")
    q = rs_fill(rs, (B, Q, 1024), "scale:1.0")
    a = rs_fill(rs, (B, A, 300), "scale:1.0")
    v[0, 29:] = 0
    v[1, 11:] = 0
    with torch.no_grad():
        raw = m.TriAtt(torch.from_numpy(v), torch.from_numpy(q), torch.from_numpy(a)).contiguous()
        p, logits = m(torch.from_numpy(v), torch.from_numpy(q), torch.from_numpy(a))
    raw, p, logits = raw.numpy(), p.contiguous().numpy(), logits.contiguous().numpy()
    assert raw.shape == (B, V, Q, A, G)
    idx = c2_sample_index(raw.size, 65536, seed + 2)
    l2 = logits.reshape(B, -1, G).astype(np.float64)
    mx = l2.max(1, keepdims=True)
    lse = (mx + np.log(np.exp(l2 - mx).sum(1, keepdims=True)))[:, 0, :]
    keys = [[k, list(t.shape)] for k, t in m.state_dict().items()]
    cfg = dict(v_dim=2048, q_dim=1024, a_dim=300, h_dim=512, h_out=1, rank=32, glimpse=G, k=1, B=B, V=V, Q=Q, A=A,
               seed=seed, zero_from={"0": 29, "1": 11}, state_keys=keys, n_sample=65536,
               ref="BASELINE.json configs[1] widths at B=3; src/tc.py:41-52, src/attention.py:49-59")
    save("g3_tcnet_forward_c2", cfg,
         outputs={"raw_s": raw.reshape(-1)[idx], "p_s": p.reshape(-1)[idx],
                  "neginf_s": np.isneginf(logits.reshape(-1)[idx]),
                  "argmax": p.reshape(B, -1, G).argmax(1).astype(np.int64),
                  "pmax": p.reshape(B, -1, G).max(1), "lse": lse.astype(np.float64),
                  "raw_absmax": np.float32(np.abs(raw).max()), "p_sum": p.reshape(B, -1, G).astype(np.float64).sum(1)})


# ----------------------------------------------------------------------------------------------
# G6/G7/G8  BCNet three branches, forward_with_weights, BiAttention (+ grads)
# ----------------------------------------------------------------------------------------------
def bc_case(name, v_dim, q_dim, h, h_out, k, B, V, Q, seed):
    torch.manual_seed(seed)
    m = BCNet(v_dim, q_dim, h, h_out, k=k).eval()
    g = torch.Generator().manual_seed(seed + 1)
    v = zero_rows(torch.randn(B, V, v_dim, generator=g).abs(), [(1, V - 2)])
    q = torch.tanh(torch.randn(B, Q, q_dim, generator=g))
    w = torch.softmax(torch.randn(B, 2, V * Q, generator=g), 2).view(B, 2, V, Q)
    outs, grads = {}, {}
    v1, q1, w1 = (t.clone().requires_grad_(True) for t in (v, q, w))
    o = m(v1, q1)
    cot = torch.randn(o.shape, generator=g)
    (o * cot).sum().backward()
    outs["fwd"] = o.detach().contiguous()
    grads.update({"fwd/v": v1.grad.clone(), "fwd/q": q1.grad.clone()})
    for n_, p_ in m.named_parameters():
        if p_.grad is not None:
            grads["fwd/p/" + n_] = p_.grad.clone()
    m.zero_grad(); v1.grad = None; q1.grad = None
    o2 = m.forward_with_weights(v1, q1, w1[:, 1])
    cot2 = torch.randn(o2.shape, generator=g)
    (o2 * cot2).sum().backward()
    outs["fww"] = o2.detach().contiguous()
    grads.update({"fww/v": v1.grad, "fww/q": q1.grad, "fww/w": w1.grad})
    for n_, p_ in m.named_parameters():
        if p_.grad is not None:
            grads["fww/p/" + n_] = p_.grad.clone()
    cfg = dict(v_dim=v_dim, q_dim=q_dim, h_dim=h, h_out=h_out, k=k, B=B, V=V, Q=Q, ref="src/bc.py:41-78")
    save(name, cfg, params=sd(m), inputs={"v": v, "q": q, "w": w, "cot_fwd": cot, "cot_fww": cot2},
         outputs=outs, grads=grads)


def bi_case(name, x_dim, y_dim, z_dim, G, B, V, Q, seed, v_mask=True):
    torch.manual_seed(seed)
    m = BiAttention(x_dim, y_dim, z_dim, G).eval()
    g = torch.Generator().manual_seed(seed + 1)
    v = zero_rows(torch.randn(B, V, x_dim, generator=g).abs(), [(0, V - 2), (B - 1, 1)])
    q = torch.tanh(torch.randn(B, Q, y_dim, generator=g))
    v1, q1 = (t.clone().requires_grad_(True) for t in (v, q))
    p, logits = m.forward_all(v1, q1, v_mask)
    cot = torch.randn(p.shape, generator=g)
    (p * cot).sum().backward()
    grads = {"v": v1.grad, "q": q1.grad}
    for n_, p_ in m.named_parameters():
        grads["p/" + n_] = p_.grad.clone()
    cfg = dict(x_dim=x_dim, y_dim=y_dim, z_dim=z_dim, glimpse=G, B=B, V=V, Q=Q, v_mask=v_mask,
               ref="src/attention.py:15-40, src/bc.py:52-58")
    save(name, cfg, params=sd(m), inputs={"v": v, "q": q, "cot_p": cot},
         outputs={"p": p.detach().contiguous(), "logits": logits.detach().contiguous()}, grads=grads)


def bi_c4():
    """BiAttention at model widths (V=36x2048, Q=14x1024, z=1024, glimpse 8), B=2; params from RandomState."""
    seed = 1205
    m = BiAttention(2048, 1024, 1024, 8).eval()
    fill_state_from_rs(m, seed)
    rs = np.random.RandomState(seed + 1)
    B, V, Q = 2, 36, 14
    v = rs_fill(rs, (B, V, 2048), "abs")
    q = rs_fill(rs, (B, Q, 1024), "tanh")
    v[1, 20:] = 0
    with torch.no_grad():
        p, logits = m.forward_all(torch.from_numpy(v), torch.from_numpy(q))
    keys = [[k, list(t.shape)] for k, t in m.state_dict().items()]
    cfg = dict(x_dim=2048, y_dim=1024, z_dim=1024, glimpse=8, B=B, V=V, Q=Q, seed=seed, zero_from={"1": 20},
               state_keys=keys, v_mask=True, ref="BASELINE.json configs[3] (BiAttention part); src/attention.py:30-40")
    save("g7_biattention_c4", cfg, outputs={"p": p.contiguous(), "logits": logits.contiguous()})


def fc_case():
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 5, 10, generator=g)
    for name, dims, act, dr in (("g0_fcnet_2layer", [10, 20, 7], "ReLU", 0.0), ("g0_fcnet_noact", [10, 12], "", 0.2),
                                ("g0_fcnet_drop", [10, 16], "ReLU", 0.5)):
        m = FCNet(dims, act=act, dropout=dr).eval()
        x1 = x.clone().requires_grad_(True)
        y = m(x1)
        cot = torch.randn(y.shape, generator=g)
        (y * cot).sum().backward()
        grads = {"x": x1.grad}
        for n_, p_ in m.named_parameters():
            grads["p/" + n_] = p_.grad.clone()
        save(name, dict(dims=dims, act=act, dropout=dr, ref="src/fc.py:13-34"), params=sd(m),
             inputs={"x": x, "cot": cot}, outputs={"y": y.detach()}, grads=grads)


def state_keys_real():
    """G11: state_dict names + shapes at the real model dims (names only)."""
    out = {}
    out["TriAttention(2048,1024,1024,512,1,32,2,1)"] = [[k, list(v.shape)] for k, v in
                                                        TriAttention(2048, 1024, 1024, 512, 1, 32, 2, 1).state_dict().items()]
    out["TCNet(2048,1024,1024,512,1,32,1,k=2)"] = [[k, list(v.shape)] for k, v in
                                                   TCNet(2048, 1024, 1024, 512, 1, 32, 1, dropout=[.2, .5], k=2).state_dict().items()]
    out["BiAttention(2048,1024,1024,8)"] = [[k, list(v.shape)] for k, v in BiAttention(2048, 1024, 1024, 8).state_dict().items()]
    out["BCNet(2048,1024,1024,None,k=1)"] = [[k, list(v.shape)] for k, v in BCNet(2048, 1024, 1024, None, k=1).state_dict().items()]
    with open(os.path.join(OUT, "g11_state_keys.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("g11_state_keys.json")


if __name__ == "__main__":
    if len(sys.argv) > 1:                                  # regenerate selected fixtures only: make_golden.py tc_c2 ...
        for fn in sys.argv[1:]:                            # "tc_c2" or "tc_act_case:g3_tcnet_act_tanh:Tanh:25" (':'-separated arguments, ints parsed)
            name, *fargs = fn.split(":")
            globals()[name](*[int(x) if x.lstrip("-").isdigit() else x for x in fargs])
        sys.exit(0)
    g1()
    g2()
    fc_case()
    tc_case("g3_tcnet_small", 64, 48, 32, 64, 4, 2, 1, 3, 5, 4, 3, seed=21, zr=[(0, 3), (2, 4)])
    tc_case("g3_tcnet_g3_odd", 40, 24, 20, 48, 3, 3, 1, 2, 7, 3, 5, seed=22, zr=[(1, 2)])
    tc_case("g3_tcnet_allzero_sample", 32, 16, 16, 32, 2, 2, 1, 2, 3, 2, 2, seed=23, zr=[(1, 0)])
    tc_att_grad_case("g8_triattention_grad", seed=24)
    tc_fww_case("g5_tcnet_fww_k2", 64, 48, 32, 64, 4, 2, 2, 3, 5, 4, 3, seed=31)
    tc_fww_case("g5_tcnet_fww_k1", 40, 24, 20, 48, 3, 2, 1, 2, 6, 3, 2, seed=32)
    tc_act_case("g3_tcnet_act_tanh", "Tanh", seed=25)
    tc_c1()
    tc_c2()
    bc_case("g6_bcnet_hnone_k1", 64, 48, 32, None, 1, 3, 5, 4, seed=41)
    bc_case("g6_bcnet_h2_k3", 64, 48, 32, 2, 3, 3, 5, 4, seed=42)
    bc_case("g6_bcnet_h40_k1", 64, 48, 32, 40, 1, 2, 5, 4, seed=43)
    bi_case("g7_biattention_g2", 64, 48, 32, 2, 3, 5, 4, seed=51)
    bi_case("g7_biattention_g8", 64, 48, 32, 8, 3, 6, 3, seed=52)
    bi_case("g7_biattention_nomask", 64, 48, 32, 2, 3, 5, 4, seed=53, v_mask=False)
    bi_c4()
    state_keys_real()
