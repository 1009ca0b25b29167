"""Gradient parity of the HIP backward path against the gradients the reference's autograd produced (tests/golden, `g/*`).
loss = sum(out * cot) with the cotangent stored in the fixture.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import cti_amd
import golden_util as gu
from oracle import cti_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


@pytest.fixture(params=["fp32", "bf16x3"], autouse=True)
def precision(request):
    old = cti_amd.get_precision()
    cti_amd.set_precision(request.param)
    yield request.param
    cti_amd.set_precision(old)


def T(x, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t.requires_grad_(True) if grad else t


def load_into(m, params):
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m.to(DEV).eval()


def check(x, ref, tol=TOL, what=""):
    x = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
    assert x.shape == tuple(ref.shape), (what, x.shape, ref.shape)
    e = O.norm_max_err(x, ref)
    assert e < tol, "%s: normalised max error %.3g >= %.3g" % (what, e, tol)


def check_param_grads(m, fx, prefix="p/", tol=TOL):
    n = 0
    for name, p in m.named_parameters():
        key = prefix + name
        if key in fx.g:
            assert p.grad is not None, name
            check(p.grad, fx.g[key], tol, "grad of " + name)
            n += 1
    assert n > 0


@pytest.mark.parametrize("name", ["g0_fcnet_2layer", "g0_fcnet_noact", "g0_fcnet_drop"])
def test_fcnet_backward(name):
    fx = gu.load(name)
    c = fx.cfg
    m = load_into(cti_amd.FCNet(c["dims"], act=c["act"], dropout=c["dropout"]), fx.p)
    x = T(fx.i["x"], grad=True)
    y = m(x)
    check(y, fx.o["y"], what=name + " forward under autograd")
    (y * T(fx.i["cot"])).sum().backward()
    check(x.grad, fx.g["x"], what=name + " dx")
    check_param_grads(m, fx)


def test_gemm_tn_split_k_vs_numpy():
    rs = np.random.RandomState(0)
    for M, N, K in ((5, 3, 4), (300, 70, 33), (5000, 130, 64), (9216, 96, 40)):
        a = rs.standard_normal((M, N)).astype(np.float32)
        b = rs.standard_normal((M, K)).astype(np.float32)
        ref = a.astype(np.float64).T @ b.astype(np.float64)
        out = cti_amd.ops.gemm_tn(T(a), T(b))
        check(out, ref, tol=2e-5, what="gemm_tn %dx%dx%d" % (M, N, K))


def test_dropout_mask_statistics_and_backward():
    torch.manual_seed(3)
    x = torch.randn(64, 1000, device=DEV, requires_grad=True)
    y = cti_amd.pkg.autograd.dropout(x, 0.3, True)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01
    assert torch.allclose(y[y != 0], (x / 0.7)[y != 0])
    y.sum().backward()
    assert torch.equal(x.grad != 0, y != 0) and torch.allclose(x.grad[y != 0], torch.full_like(x.grad[y != 0], 1 / 0.7))
    y2 = cti_amd.pkg.autograd.dropout(x, 0.3, True)                  # a fresh mask every call
    assert not torch.equal(y2 != 0, y != 0)
    assert cti_amd.pkg.autograd.dropout(x, 0.3, False) is x          # eval mode: identity
