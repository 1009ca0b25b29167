"""Teacher logits for the distillation loss -- the data format either side of `Distillation_Loss`.

The reference's evaluation script writes one float16 vector per question id into a pickled dict (src/FFOE/test.py:125-130, dumped at
:184-187 as `results/<model>_<split>_logits.pkl`), and its datasets read `<split>_teacher_logits.pkl` back and hand each entry to the
loss as float32 (src/FFOE/dataset.py:265-268, :366).  Host-side bookkeeping: the only arithmetic is the float32 -> float16 rounding,
done once for the whole batch on the device (round-to-nearest-even, the same as numpy's) instead of once per question on the host."""
import pickle

import numpy as np
import torch


def make_json_with_logits(logits, qIds):
    """{int(question id): float16 numpy vector of that question's logits} -- src/FFOE/test.py:125-130."""
    if logits.shape[0] != len(qIds):
        raise AssertionError("%s (true) vs %s (expected)" % (logits.shape[0], len(qIds)))      # utils.assert_eq of the reference
    half = logits.detach().to(torch.float16).cpu().numpy()
    return {int(qIds[i]): half[i].copy() for i in range(half.shape[0])}


def dump_teacher_logits(logits, qIds, path):
    """Write the dict of make_json_with_logits the way the reference does (pickle, default protocol)."""
    with open(path, "wb") as f:
        pickle.dump(make_json_with_logits(logits, qIds), f)


def load_teacher_logits(path):
    """The dict a dataset keeps as `self.teacher_logits` (src/FFOE/dataset.py:267-268)."""
    with open(path, "rb") as f:
        return pickle.load(f)


def teacher_logit_batch(teacher_logits, qIds, device=None):
    """Rows of the `knowledge` argument of Distillation_Loss for a batch of question ids: float32, as the dataset's __getitem__ produces
    them one by one (torch.from_numpy(np.float32(entry['teacher_logit'])), src/FFOE/dataset.py:366)."""
    rows = np.stack([np.float32(teacher_logits[int(q)]) for q in qIds])
    t = torch.from_numpy(rows)
    return t.to(device) if device is not None else t
