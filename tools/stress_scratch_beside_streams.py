#!/usr/bin/env python3
"""Round 4: do the kernels that still own a private segment (scratch) survive beside MFMA-chain kernels of ANOTHER stream?  (The bi pools did not: profiles/
r04_scratch_corruption_beside_another_stream.txt.)  The M build of the f16f6 mode (mbuild_mfma_f6_kernel: 20 B of scratch per lane) is launched 60 times on a side
stream while the main stream runs (a) nothing, (b) the fp32-grade GRU steps that corrupted the pools, (c) the f16f6 mode-3 product, (d) bf16x3 GEMMs; every launch's
planes are compared bit for bit with a launch on an idle device.     python tools/stress_scratch_beside_streams.py     (GPU box)"""
import os, sys, types
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd
ops, L = cti_amd.ops, cti_amd.pkg._lib
lib = L.lib()
dev = torch.device("cuda")
B, V, Q, R, hr, G = 256, 36, 14, 32, 16, 2
g = torch.Generator().manual_seed(1)
Vr = torch.relu(torch.randn(B, V, R * hr, generator=g)).to(dev)
Qr = torch.relu(torch.randn(B, Q, R * hr, generator=g)).to(dev)
Teff = torch.randn(R, hr, hr, hr, G, generator=g).to(dev)
Tt = ops.transpose(Teff, hr, hr * hr * G, batch=R, s_src=hr ** 3 * G, ld_src=hr * hr * G, s_dst=hr ** 3 * G, ld_dst=hr).view(R, hr * hr * G, hr)
nb = lib.cti_f16f6_planes_bytes(B * V * Q * G, R * hr, V * Q * G)


def mbuild(blk):
    rc = lib.cti_paralind_mbuild_f16f6_fwd(Vr.data_ptr(), Qr.data_ptr(), Tt.data_ptr(), blk.data_ptr(), nb, B, V, Q, R, hr, G, ops._stream())
    assert rc == 0, lib.cti_last_error_string()


ref = torch.zeros(nb, device=dev, dtype=torch.uint8)
mbuild(ref)
torch.cuda.synchronize()
# partners for the main stream
cti_amd.set_precision("bf16x3")
ds = types.SimpleNamespace(dictionary=types.SimpleNamespace(ntoken=2000), v_dim=2048, num_ans_candidates=16)
ma = types.SimpleNamespace(op="c", num_hid=1024, gamma=2, h_mm=512, rank=32, k=1, h_out=1, activation="relu", dropout=0.5, use_counter=False)
model = cti_amd.build_cti(ma, ds).to(dev).eval()
qtok = torch.randint(0, 2000, (256, 14), generator=g).to(dev)
with torch.no_grad():
    emb = model.w_emb(qtok)
Mrows = torch.randn(B * V * Q * G, 512, generator=g).to(dev)[: 64 * V * Q * G] * 4
Ar = torch.relu(torch.randn(64 * 3129, 512, generator=g)).to(dev)
pa, pb = ops.quantize_f16f6(Mrows, V * Q * G), ops.quantize_f16f6(Ar, 3129)
out3 = torch.empty((64, V * Q, 3129, G), device=dev)
x = torch.randn(9216, 2048, generator=g).to(dev); w = torch.randn(3072, 2048, generator=g).to(dev); wp = ops.split_operand(w)
big = torch.randn(6144, 6144, device=dev)


def partner(what):
    with torch.no_grad():
        if what == "gru_bf16x3":
            for _ in range(3): model.q_emb.forward_all(emb)
        elif what == "mode3_f16f6":
            for _ in range(3):
                rc = lib.cti_gemm_nt_f16f6(pa.data_ptr(), Mrows.shape[0], V * Q * G, pb.data_ptr(), Ar.shape[0], 3129, out3.data_ptr(), 3129 * G, G, V * Q * 3129 * G, G, 64,
                                           V * Q * G, 3129, 512, 0, 1, 0, 0, ops._stream())
                assert rc == 0
        elif what == "gemm_bf16x3":
            for _ in range(2): ops.wn_linear(x, w, None, 1, None, False, w_planes=wp)


side = torch.cuda.Stream()
for what in ("nothing", "gru_bf16x3", "mode3_f16f6", "gemm_bf16x3"):
    bad = 0
    for rep in range(6):
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        blks = [torch.zeros(nb, device=dev, dtype=torch.uint8) for _ in range(10)]
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            _ = big @ big
            for b_ in blks:
                mbuild(b_)
        partner(what)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(b_, ref) else 1 for b_ in blks)
    print("M build beside %-12s: %d of 60 launches differ from the idle-device planes" % (what, bad))
