"""ModeProduct -- drop-in for the reference's src/Tensor.py:3-28 (n_way = 3, the only form the model calls).

The reference runs three transpose/contiguous/matmul rounds; here the fixed index scramble its `.view` calls
imply (T_eff) is applied once to the core, modes 1+2 run in one kernel and mode 3 is one batched
MFMA GEMM writing the (B,V,Q,A,G) result directly."""
import torch

from . import ops


def ModeProduct(tensor, matrix_1, matrix_2, matrix_3, matrix_4, n_way=3):
    if n_way != 3 or matrix_4 is not None:
        raise NotImplementedError("ModeProduct: only the 3-way form (src/Tensor.py:3-20) is on the CTI path")
    if tensor.dim() == 6:
        if tensor.size(5) != 1:
            raise RuntimeError("ModeProduct: h_out must be 1 (the reference's view at src/Tensor.py:6 fails otherwise)")
        tensor = tensor[..., 0]
    if tensor.dim() != 5 or tensor.size(0) != 1:
        raise RuntimeError("ModeProduct: tensor must be (1, I, J, K, G[, 1])")
    T = tensor.float().contiguous()                         # (1,I,J,K,G) == (R=1,I,J,K,G)
    Teff = ops.teff_scramble(T)
    M = ops.paralind_mbuild(matrix_1.float(), matrix_2.float(), Teff)
    return ops.paralind_core(M, matrix_3.float())
