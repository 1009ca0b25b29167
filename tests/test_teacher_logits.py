"""Teacher-logit dump (reference src/FFOE/test.py:125-130, :184-187; read back by src/FFOE/dataset.py:265-268, :366): same keys, dtype
and values as the reference's per-question `np.float16(logits[i].detach().numpy())`, and the float32 rows the dataset hands to the loss."""
import pickle

import numpy as np
import pytest
import torch

import cti_amd


def _reference_dump(logits, qids):
    # restatement of make_json_with_logits (src/FFOE/test.py:125-130)
    assert logits.size(0) == len(qids)
    return {int(qids[i]): np.float16(logits[i].detach().numpy()) for i in range(logits.size(0))}


def test_dump_matches_the_reference_format(tmp_path):
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(7, 3129, generator=g) * 30
    logits[0, :4] = torch.tensor([65519.0, 65520.0, -1e-8, 2049.0])        # float16 rounding edges: below / at the overflow threshold, underflow, tie
    qids = [262148000, 5, 17, 393225001, 9, 12, 100]
    want = _reference_dump(logits, qids)
    got = cti_amd.pkg.make_json_with_logits(logits, qids)
    assert list(got) == list(want)
    for k in want:
        assert got[k].dtype == np.float16 and got[k].shape == (3129,)
        assert np.array_equal(got[k].view(np.uint16), want[k].view(np.uint16))
    path = tmp_path / "cti_test_logits.pkl"
    cti_amd.pkg.dump_teacher_logits(logits, qids, str(path))
    back = cti_amd.pkg.load_teacher_logits(str(path))
    assert pickle.load(open(path, "rb")).keys() == back.keys()
    rows = cti_amd.pkg.teacher_logit_batch(back, [17, 5])
    assert rows.dtype == torch.float32 and rows.shape == (2, 3129)
    assert torch.equal(rows[0], torch.from_numpy(np.float32(want[17]))) and torch.equal(rows[1], torch.from_numpy(np.float32(want[5])))
    with pytest.raises(AssertionError):
        cti_amd.pkg.make_json_with_logits(logits, qids[:3])
