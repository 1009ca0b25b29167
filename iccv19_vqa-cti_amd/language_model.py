"""WordEmbedding / QuestionEmbedding -- drop-ins for the reference's src/language_model.py:11-98: same
constructors, state_dict keys (`emb.weight`, `emb_.weight`, `rnn.weight_ih_l0`, ...) and RNG consumption at construction.

The lookup is one gather kernel (both tables of the 'c' mode in the same pass); the GRU runs its input projection for all time
steps as ONE MFMA GEMM and each step as one small GEMM + a fused gate kernel; backward (BPTT) runs in HIP kernels too."""
import numpy as np
import torch
import torch.nn as nn

from . import autograd as AG
from . import ops

_ROWS16 = __import__("os").environ.get("CTI_EMB_ROWS16", "1") == "1"      # 0: fp32 word vectors + a split pass in front of the GRU's input product (rounds 1-5; A/B)


class WordEmbedding(nn.Module):
    """Token ids -> word vectors; row `ntoken` of each table is the padding row (src/language_model.py:13-17).  With 'c' in `op` a second,
    frozen table `emb_` is looked up in the same kernel pass and concatenated (src/language_model.py:20-22,43-44)."""

    def apply(self, fn):
        r = super().apply(fn)            # (an initialiser writing through .data moves no cache key: fc.WNLinear.apply)
        ops.invalidate_caches()
        return r

    def __init__(self, ntoken, emb_dim, dropout, op=''):
        super(WordEmbedding, self).__init__()
        self.ntoken, self.emb_dim, self.op = ntoken, emb_dim, op
        rows = ntoken + 1
        self.emb = nn.Embedding(rows, emb_dim, padding_idx=ntoken)
        if self._concat:
            self.emb_ = nn.Embedding(rows, emb_dim, padding_idx=ntoken)
            self.emb_.weight.requires_grad_(False)                      # the fixed copy
        self.dropout = nn.Dropout(dropout)

    @property
    def _concat(self):
        return 'c' in self.op

    def init_embedding(self, np_file, tfidf=None, tfidf_weights=None):
        """Host-side initialisation from a GloVe matrix (src/language_model.py:27-38): rows [0, ntoken) of `emb` take the file's
        matrix; with 'c' in op the second table takes the same matrix, or its tf-idf re-weighting (then it becomes trainable)."""
        init = torch.from_numpy(np.load(np_file))
        if tuple(init.shape) != (self.ntoken, self.emb_dim):
            raise AssertionError("embedding file has shape %s, expected %s" % (tuple(init.shape), (self.ntoken, self.emb_dim)))
        dev = self.emb.weight.device
        self.emb.weight.data[:self.ntoken] = init.to(dev)
        second = init
        if tfidf is not None:
            if tfidf_weights is not None and 0 < tfidf_weights.size:
                second = torch.cat([second, torch.from_numpy(tfidf_weights)], 0)
            second = tfidf.matmul(second)
            self.emb_.weight.requires_grad = True
        if self._concat:
            table = torch.zeros(init.shape)
            table[:second.size(0)] = second
            self.emb_.weight.data[:self.ntoken] = table.to(dev)

    def rows16(self, x):
        """Inference in the plain-bf16 mode: the word vectors as the bf16 rows QuestionEmbedding.forward_all multiplies as they stand (ops.embedding_rows16); the fp32
        vectors of forward() otherwise."""
        if (self.training or torch.is_grad_enabled() or ops.get_precision() != "bf16" or not x.is_cuda or not _ROWS16):
            return self.forward(x)
        second = self.emb_.weight if self._concat else None
        return ops.embedding_rows16(x, self.emb.weight.detach(), None if second is None else second.detach())

    def forward(self, x):
        second = self.emb_.weight if self._concat else None
        if torch.is_grad_enabled() and (self.emb.weight.requires_grad or (second is not None and second.requires_grad)):
            emb = AG.EmbeddingFn.apply(x, self.emb.weight, second, self.ntoken)
        else:
            emb = ops.embedding(x, self.emb.weight.detach(), None if second is None else second.detach())
        return AG.dropout(emb, self.dropout.p, self.training)


class QuestionEmbedding(nn.Module):
    """GRU over the word vectors (src/language_model.py:50-98).  `rnn` is a torch nn.GRU / nn.LSTM used ONLY as the parameter container
    (same keys `rnn.weight_ih_l0` ... and the same initial values as the reference); the arithmetic runs in the HIP library."""

    def apply(self, fn):
        r = super().apply(fn)            # (an initialiser writing through .data moves no cache key: fc.WNLinear.apply)
        ops.invalidate_caches()
        return r

    _CELLS = {'GRU': nn.GRU, 'LSTM': nn.LSTM}

    def __init__(self, in_dim, num_hid, nlayers, bidirect, dropout, rnn_type='GRU'):
        super(QuestionEmbedding, self).__init__()
        if rnn_type not in self._CELLS:
            raise AssertionError("rnn_type must be 'LSTM' or 'GRU', got %r" % (rnn_type,))
        self.in_dim, self.num_hid, self.nlayers, self.rnn_type = in_dim, num_hid, nlayers, rnn_type
        self.ndirections = 2 if bidirect else 1
        self.rnn = self._CELLS[rnn_type](in_dim, num_hid, nlayers, bidirectional=bool(bidirect), dropout=dropout, batch_first=True)

    def _check(self):
        if self.rnn_type != 'GRU' or self.nlayers != 1 or self.ndirections != 1:
            raise NotImplementedError("the MI355X path implements what the reference's builders construct: a 1-layer one-direction GRU "
                                      "(src/FFOE/base_model.py:141,182,186); got %s, %d layer(s), %d direction(s)"
                                      % (self.rnn_type, self.nlayers, self.ndirections))

    def forward_all(self, x):
        # x: [batch, sequence, in_dim] -> [batch, sequence, num_hid]
        self._check()
        r = self.rnn
        ps = (r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)
        if x.dtype == torch.bfloat16 and (torch.is_grad_enabled() or ops.get_precision() != "bf16"):
            x = ops.widen_bf16(x[:, :, :self.in_dim]).contiguous()         # (rows16 outside the mode it is for)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in ps)):
            return AG.GRUFn.apply(x, *ps)
        key = (ps[0].data_ptr(), ps[0]._version, ps[1].data_ptr(), ps[1]._version, ops._param_epoch[0], ops.get_precision())
        if getattr(self, "_planes_key", None) != key:                       # the two weight matrices as resident operand planes
            self._planes = (ops.split_operand(ps[0].detach()), ops.split_operand(ps[1].detach()))
            self._planes_key = key
        return ops.gru_forward(x, *[p.detach() for p in ps], w_planes=self._planes if self._planes[0] is not None else None)[0]

    def forward(self, x):
        # x: [batch, sequence, in_dim] -> the last hidden state [batch, num_hid]
        return self.forward_all(x)[:, -1]
