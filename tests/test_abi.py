"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/cti_hip.h
declares (no compute without a GPU), argument errors come back as negative codes with a message, and the module
mirrors keep the reference's constructor signatures and state_dict layout."""
import inspect
import json
import os
import re

import numpy as np
import pytest
import torch

import cti_amd
import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = cti_amd.pkg._lib


def _declared():
    src = open(os.path.join(ROOT, "include", "cti_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cti_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    names = _declared()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(L.SIGNATURES) == names           # the ctypes table and the header list the same functions
    assert lib.cti_abi_version() == 1


def test_argument_errors_do_not_launch():
    lib = L.lib()
    assert lib.cti_wn_scale(None, None, None, 1, 4, None, 0, None) == -1  # CTI_E_NULL
    assert b"NULL" in lib.cti_last_error_string()
    assert lib.cti_zero_row_mask(1, 4, 1, 0, 4, None) == -2              # CTI_E_SHAPE (rows = 0) -- pointers not touched
    assert lib.cti_wn_linear_fwd(1, 4, 1, 4, None, 1, None, 1, 4, 2, 4, 4, 7, 0, None, 0, None) == -4   # bad act
    assert lib.cti_softmax_tri_workspace_bytes(4, 36, 14 * 4, 2) > 0
    assert lib.cti_softmax_tri_workspace_bytes(0, 36, 56, 2) == 0


def test_tuning_rejects_bad_values_and_defaults_to_auto():
    lib = L.lib()
    assert lib.cti_set_tuning(1, 3) == -2 and lib.cti_set_tuning(2, 63) == -2 and lib.cti_set_tuning(99, 0) == -4
    assert lib.cti_get_tuning(1) == -1 and lib.cti_get_tuning(2) == 0
    with cti_amd.ops.tuning(gemm_cfg=2, tri_chunk=64):
        assert lib.cti_get_tuning(1) == 2 and lib.cti_get_tuning(2) == 64
    assert lib.cti_get_tuning(1) == -1 and lib.cti_get_tuning(2) == 0


def test_tuning_overrides_are_thread_local():
    """include/cti_hip.h promises no process-wide mutable state: an override set on one thread is invisible to every other thread."""
    import threading
    lib = L.lib()
    seen = {}

    def other():
        seen["before"] = (lib.cti_get_tuning(1), lib.cti_get_tuning(2))
        lib.cti_set_tuning(1, 0)
        seen["own"] = lib.cti_get_tuning(1)

    with cti_amd.ops.tuning(gemm_cfg=2, tri_chunk=64):
        t = threading.Thread(target=other); t.start(); t.join()
        assert lib.cti_get_tuning(1) == 2 and lib.cti_get_tuning(2) == 64          # the other thread's override did not leak here
    assert seen == {"before": (-1, 0), "own": 0}


def test_softmax_partials_contract_on_the_host():
    """cti_tcnet_softmax_partials_bytes says where the fused partial pass exists (f16f6, glimpse 2, h % 32 == 0) and how large its block is:
    [B][tiles of 256 x 192 over (V*Q*G) x A][8 waves][G][2] floats; the consumer refuses blocks that are not a whole number of (B, G) pairs."""
    lib = L.lib()
    B, V, Q, A, h = 256, 36, 14, 3129, 512
    tiles = ((V * Q * 2 + 255) // 256) * ((A + 191) // 192)
    assert lib.cti_tcnet_softmax_partials_bytes(B, V, Q, A, h, 2, L.PREC_F16F6) == B * tiles * 8 * 2 * 2 * 4
    assert lib.cti_tcnet_softmax_partials_bytes(B, V, Q, A, h, 2, L.PREC_BF16X3) == 0          # other arithmetic: the two-pass softmax
    assert lib.cti_tcnet_softmax_partials_bytes(B, V, Q, A, h, 3, L.PREC_F16F6) == 0           # other glimpse counts
    assert lib.cti_tcnet_softmax_partials_bytes(B, V, Q, A, 48, 2, L.PREC_F16F6) == 0          # h not a multiple of the K block
    assert lib.cti_tcnet_softmax_partials_bytes(0, V, Q, A, h, 2, L.PREC_F16F6) == 0
    one = 8                                                                                   # any non-NULL addresses: nothing is dereferenced before the checks
    assert lib.cti_masked_softmax_tri_from_partials_fwd(None, one, one, 64, one, 2, 3, 4, 2, one, 64, None) == -1          # NULL logits
    assert lib.cti_masked_softmax_tri_from_partials_fwd(one, one, one, 64, one, 2, 3, 4, 3, one, 64, None) == -4           # G != 2
    assert lib.cti_masked_softmax_tri_from_partials_fwd(one, one, one, 40, one, 2, 3, 4, 2, one, 64, None) == -2           # 40 bytes: not whole (B, G) pairs
    assert lib.cti_masked_softmax_tri_from_partials_fwd(one, one, one, 64, one, 2, 3, 4, 2, one, 16, None) == -5           # workspace too small
    assert b"workspace" in lib.cti_last_error_string()


def test_f16f6_block_producers_contract_on_the_host():
    """The f16f6 mode's block-in / block-out entry points (cti_gemm_nt_f16f6_planes, cti_quantize_f16f6_scaled, cti_paralind_mbuild_f16f6_fwd) check
    their arguments before anything is launched: NULL, sizes, the output block's size / alignment, the activation code, the shapes the
    direct-encoding M build accepts."""
    lib = L.lib()
    one, al = 8, 256                                                                          # non-NULL addresses; nothing is dereferenced before the checks
    rows, M, K = 1000, 64, 96
    need = lib.cti_f16f6_planes_bytes(rows, M, 0)
    assert need == 2 * ((rows + 256) * 64 + (rows + 256) * 24 + (rows + 512) * 2) + 3 * 256   # Kb = 2 blocks of (H 64 B + FL 24 B per row, S 2 B per row) + slack rows
    assert lib.cti_gemm_nt_f16f6_planes(None, M, one, rows, al, need, 0, M, rows, K, None, 0, None) == -1                # NULL weight block
    assert lib.cti_gemm_nt_f16f6_planes(one, M, one, rows, al, need, 0, M, rows + 1, K, None, 0, None) == -2            # more rows than the operand has
    assert lib.cti_gemm_nt_f16f6_planes(one, M, one, rows, al, need, 0, M, rows, K, None, 7, None) == -4                # activation code
    assert lib.cti_gemm_nt_f16f6_planes(one, M, one, rows, al, need - 1, 0, M, rows, K, None, 0, None) == -5            # output block too small
    assert lib.cti_gemm_nt_f16f6_planes(one, M, one, rows, al + 16, need, 0, M, rows, K, None, 0, None) == -3           # output block not 256-B aligned
    assert lib.cti_gemm_nt_f16f6_planes(one, 48, one, rows, al, lib.cti_f16f6_planes_bytes(rows, 48, 0), 0, 48, rows, K, None, 0, None) == -4   # features not whole K blocks
    assert b"M % 32" in lib.cti_last_error_string()
    nq = lib.cti_f16f6_planes_bytes(rows, K, 0)
    assert lib.cti_quantize_f16f6_scaled(one, K, rows, K, 0, None, 16, al, nq, None) == -1                                # NULL scales
    assert lib.cti_quantize_f16f6_scaled(one, K, rows, K, 0, one, 0, al, nq, None) == -2                                  # scale_div = 0
    assert lib.cti_quantize_f16f6_scaled(one, K, rows, K, 0, one, 16, al, nq - 1, None) == -5
    B, V, Q, R, hr, G = 4, 36, 14, 32, 16, 2
    nm = lib.cti_f16f6_planes_bytes(B * V * Q * G, R * hr, V * Q * G)
    assert lib.cti_paralind_mbuild_f16f6_fwd(one, one, None, al, nm, B, V, Q, R, hr, G, None) == -1                      # the transposed core is required
    assert lib.cti_paralind_mbuild_f16f6_fwd(one, one, one, al, nm - 1, B, V, Q, R, hr, G, None) == -5
    assert lib.cti_paralind_mbuild_f16f6_fwd(one, one, one, al, lib.cti_f16f6_planes_bytes(B * V * Q * G, 3 * 8, V * Q * G), B, V, Q, 3, 8, G, None) == -4   # R * hr not whole K blocks
    assert lib.cti_paralind_mbuild_f16f6_fwd(al, al, al, al, lib.cti_f16f6_planes_bytes(B * 48 * 16 * G, R * hr, 48 * 16 * G), B, 48, 16, R, hr, G, None) == -4   # X + hold buffer exceed the LDS
    assert b"direct-encoding" in lib.cti_last_error_string()
    # the fused forward's workspace: the f16f6 mode holds no fp32 M where the direct-encoding build applies
    w_direct = lib.cti_tcnet_forward_workspace_bytes(8, 36, 14, 200, 256, 128, 96, 512, 32, 2, L.PREC_F16F6)
    w_fallback = lib.cti_tcnet_forward_workspace_bytes(8, 48, 16, 200, 256, 128, 96, 512, 32, 2, L.PREC_F16F6)
    assert w_fallback - w_direct > 8 * 48 * 16 * 2 * 512 * 4                                   # (more rows AND the fp32 M of the 48 x 16 shape)


def test_ops_refuse_cpu_tensors():
    with pytest.raises(cti_amd.CtiError):
        cti_amd.ops.zero_row_mask(torch.zeros(2, 3, 4))
    m = cti_amd.FCNet([4, 3]).eval()
    with torch.no_grad(), pytest.raises(cti_amd.CtiError):
        m(torch.zeros(2, 4))


def test_state_dict_layout_matches_reference_at_real_dims():
    ref = json.load(open(os.path.join(gu.GOLDEN, "g11_state_keys.json")))
    built = {
        "TriAttention(2048,1024,1024,512,1,32,2,1)": cti_amd.TriAttention(2048, 1024, 1024, 512, 1, 32, 2, 1),
        "TCNet(2048,1024,1024,512,1,32,1,k=2)": cti_amd.TCNet(2048, 1024, 1024, 512, 1, 32, 1, dropout=[.2, .5], k=2),
        "BiAttention(2048,1024,1024,8)": cti_amd.BiAttention(2048, 1024, 1024, 8),
        "BCNet(2048,1024,1024,None,k=1)": cti_amd.BCNet(2048, 1024, 1024, None, k=1),
    }
    for name, m in built.items():
        mine = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        assert mine == ref[name], name            # same names, shapes AND order (named_parameters order = flat-grad order)


@pytest.mark.parametrize("fixture,build", [
    ("g3_tcnet_small", lambda c: cti_amd.TriAttention(c["v_dim"], c["q_dim"], c["a_dim"], c["h_dim"], 1, c["rank"], c["glimpse"], c["k"])),
    ("g7_biattention_g8", lambda c: cti_amd.BiAttention(c["x_dim"], c["y_dim"], c["z_dim"], c["glimpse"])),
    ("g6_bcnet_h40_k1", lambda c: cti_amd.BCNet(c["v_dim"], c["q_dim"], c["h_dim"], c["h_out"], k=c["k"])),
    ("g6_bcnet_h2_k3", lambda c: cti_amd.BCNet(c["v_dim"], c["q_dim"], c["h_dim"], c["h_out"], k=c["k"])),
    ("g0_fcnet_2layer", lambda c: cti_amd.FCNet(c["dims"], act=c["act"], dropout=c["dropout"])),
    ("g0_fcnet_noact", lambda c: cti_amd.FCNet(c["dims"], act=c["act"], dropout=c["dropout"])),
])
def test_reference_checkpoints_load_strictly(fixture, build):
    fx = gu.load(fixture)
    m = build(fx.cfg)
    sd = {k: torch.from_numpy(v) for k, v in fx.p.items()}
    m.load_state_dict(sd, strict=True)
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), fx.p[k])


def test_constructor_signatures_match_reference():
    def names(f):
        return list(inspect.signature(f).parameters)
    assert names(cti_amd.TCNet.__init__) == ["self", "v_dim", "q_dim", "a_dim", "h_dim", "h_out", "rank", "glimpse", "act", "dropout", "k"]
    assert names(cti_amd.BCNet.__init__) == ["self", "v_dim", "q_dim", "h_dim", "h_out", "act", "dropout", "k"]
    assert names(cti_amd.BiAttention.__init__) == ["self", "x_dim", "y_dim", "z_dim", "glimpse", "dropout"]
    assert names(cti_amd.TriAttention.__init__) == ["self", "v_dim", "q_dim", "a_dim", "h_dim", "h_out", "rank", "glimpse", "k", "dropout"]
    assert names(cti_amd.FCNet.__init__) == ["self", "dims", "act", "dropout"]
    assert names(cti_amd.ModeProduct) == ["tensor", "matrix_1", "matrix_2", "matrix_3", "matrix_4", "n_way"]
    assert names(cti_amd.TCNet.forward_with_weights) == ["self", "v", "q", "a", "w"]
    assert names(cti_amd.BCNet.forward_with_weights) == ["self", "v", "q", "w"]
    assert names(cti_amd.BiAttention.forward_all) == ["self", "v", "q", "v_mask"]


def test_same_seed_same_parameters_as_reference_fixture():
    """The mirrors consume the RNG exactly like the reference constructors: make_golden.py built g7_biattention_g2
    after torch.manual_seed(51)."""
    fx = gu.load("g7_biattention_g2")
    c = fx.cfg
    torch.manual_seed(51)
    m = cti_amd.BiAttention(c["x_dim"], c["y_dim"], c["z_dim"], c["glimpse"])
    for k, v in m.state_dict().items():
        assert np.allclose(v.numpy(), fx.p[k], rtol=0, atol=1e-6), k


def test_teff_index_arithmetic_host_mirror():
    """The device scramble (csrc/cti_paralind.hip:teff_src_index) restated in Python against the G2 fixture."""
    fx = gu.load("g2_teff_index_maps")
    for key, ref in fx.o.items():
        hr, G = (int(t[2:] if t.startswith("hr") else t[1:]) for t in key.split("_"))
        J = K = hr
        for i in range(hr):
            for j in range(J):
                for k in range(K):
                    for g in range(G):
                        f = (g * K + k) * J + j
                        gs, js, ks = f % G, (f // G) % J, f // (G * J)
                        assert ref[i, j, k, g] == i * J * K * G + (js * K + ks) * G + gs


def test_kernels_with_a_private_segment_are_the_known_ones(tmp_path):
    """No kernel acquires scratch (register spills or an address-taken local) silently.  Round 4 found a kernel's scratch row corrupted while a kernel of
    ANOTHER stream was resident (tests/test_fusions_gpu.py::test_pools_beside_the_bf16x3_gru_on_another_stream): everything that may run beside another stream
    has to be scratch-free; the kernels below are the measured exceptions (main-stream or aux-stream kernels of TCNet.forward whose every launch bench.py
    checks sample by sample, and the training-only M-build backward)."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("no ROCm llvm tools")
    so = tmp_path / "lib.so"
    shutil.copy(os.path.join(ROOT, "iccv19_vqa-cti_amd", "lib", "libcti_hip.so"), so)
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    found = set()
    for f in sorted(tmp_path.glob("lib.so.*gfx950")):
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(f)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:"):
                name = line.split(":", 1)[1].strip()
            elif line.startswith(".private_segment_fixed_size:") and int(line.split(":", 1)[1]) != 0:
                found.add(name)
    allowed = ("mbuild_mfma_f6_kernel", "mbuild_bwd_staged_kernel")
    extra = sorted(n for n in found if not any(a in n for a in allowed))
    assert not extra, "kernels with a private segment (scratch) outside the known list: %s" % extra
