"""CPU ORACLE for the rows either side of the CTI path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's word embedding, GRU question embedding, classifier, losses and the three model
forwards that call the CTI / BAN modules.  Same rules as `cti_oracle.py`: only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

Parity status: PINNED against outputs of the reference captured by `tests/golden/make_golden_models.py`
(`tests/test_oracle_models.py` checks every fixture g9_*, g10_*, g12_*).

Parameters are dicts keyed like the reference modules' `state_dict()`; eval mode (every Dropout is the identity).
"""
from __future__ import annotations

import numpy as np

from . import cti_oracle as O

__all__ = ["word_embedding", "gru_forward_all", "question_embedding", "simple_classifier", "bce_with_logits_sum",
           "distillation_loss", "ffoe_cti_forward", "ffoe_ban_forward", "mc_tan_forward"]


def _sub(params, prefix):
    return {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)}


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def word_embedding(tokens, params, prefix="", dtype=np.float32):
    """WordEmbedding.forward (src/language_model.py:40-46): emb(x), concatenated with the frozen table emb_(x) when the
    module was built with 'c' in op (the key `emb_.weight` exists exactly then, :20-22); dropout 0 in every builder."""
    p = _sub(params, prefix)
    t = np.asarray(tokens)
    out = np.asarray(p["emb.weight"], dtype=dtype)[t]
    if "emb_.weight" in p:
        out = np.concatenate([out, np.asarray(p["emb_.weight"], dtype=dtype)[t]], axis=2)
    return out


def gru_forward_all(x, params, prefix="rnn.", dtype=np.float32):
    """nn.GRU(in, H, 1, batch_first=True) from a zero state (src/language_model.py:57-61,68-75,91-96).  torch gate order
    (r, z, n):  r = s(W_ir x + b_ir + W_hr h + b_hr), z likewise, n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),
    h' = (1 - z) * n + z * h.  Returns every hidden state (B, T, H)."""
    p = _sub(params, prefix)
    w_ih = np.asarray(p["weight_ih_l0"], dtype=dtype)
    w_hh = np.asarray(p["weight_hh_l0"], dtype=dtype)
    b_ih = np.asarray(p["bias_ih_l0"], dtype=dtype)
    b_hh = np.asarray(p["bias_hh_l0"], dtype=dtype)
    x = np.asarray(x, dtype=dtype)
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = np.zeros((B, H), dtype=dtype)
    out = np.empty((B, T, H), dtype=dtype)
    for t in range(T):
        gi = x[:, t] @ w_ih.T + b_ih
        gh = h @ w_hh.T + b_hh
        r = _sigmoid(gi[:, :H] + gh[:, :H])
        z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = ((1 - z) * n + z * h).astype(dtype)
        out[:, t] = h
    return out


def question_embedding(x, params, prefix="", dtype=np.float32):
    """QuestionEmbedding.forward (src/language_model.py:77-89, one direction): the last hidden state."""
    return gru_forward_all(x, params, prefix + "rnn.", dtype)[:, -1]


def simple_classifier(x, params, prefix="", activation="relu", dtype=np.float32):
    """SimpleClassifier.forward (src/classifier.py:11-28): wn-Linear, relu | swish (src/activation.py:17-22), dropout,
    wn-Linear; `main.0` and `main.3`."""
    p = _sub(params, prefix)
    h = O.wn_linear(x, p["main.0.weight_g"], p["main.0.weight_v"], p["main.0.bias"], relu=False, dtype=dtype)
    h = np.maximum(h, 0) if activation == "relu" else h * _sigmoid(h)
    return O.wn_linear(h, p["main.3.weight_g"], p["main.3.weight_v"], p["main.3.bias"], relu=False, dtype=dtype)


def bce_with_logits_sum(x, target, dtype=np.float64):
    """nn.BCEWithLogitsLoss(reduction='sum') (src/FFOE/train.py:28-33; divided by the batch size at
    src/FFOE/trainer.py:189-190): sum of max(x,0) - x*t + log(1 + exp(-|x|))."""
    x = np.asarray(x, dtype=dtype)
    t = np.asarray(target, dtype=dtype)
    return np.sum(np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x))))


def _log_softmax(x):
    m = np.max(x, axis=1, keepdims=True)
    return x - m - np.log(np.sum(np.exp(x - m), axis=1, keepdims=True))


def distillation_loss(x, knowledge, target, T, alpha, dtype=np.float64):
    """Distillation_Loss.forward (src/loss_function.py:12-26): mean_b sum_c KL(softmax(k/T) || softmax(x/T)) * alpha*T*T
    + BCE_sum/B * (1 - alpha)."""
    x = np.asarray(x, dtype=dtype)
    k = np.asarray(knowledge, dtype=dtype)
    ls = _log_softmax(x / T)
    lk = _log_softmax(k / T)
    kl = np.sum(np.exp(lk) * (lk - ls), axis=1).mean()
    return kl * (alpha * T * T) + bce_with_logits_sum(x, target, dtype) / x.shape[0] * (1.0 - alpha)


def _prj(x, params, prefix, dtype):
    """FCNet([H, H], '', .2) (q_prj / a_prj / src/FFOE/base_model.py:157,192-193): one wn-Linear, no activation."""
    return O.fcnet(x, params, prefix, act="", dtype=dtype)


def _tan_forward(v, q_tok, a_tok, params, att_prefix, glimpse, activation, dtype):
    q_emb = gru_forward_all(word_embedding(q_tok, params, "w_emb.", dtype), params, "q_emb.rnn.", dtype)
    a_emb = gru_forward_all(word_embedding(a_tok, params, "wa_emb.", dtype), params, "ans_emb.rnn.", dtype)
    att, _ = O.tri_attention(v, q_emb, a_emb, params, att_prefix, dtype=dtype)
    for g in range(glimpse):
        b_emb = O.tcnet_forward_with_weights(v, q_emb, a_emb, att[..., g], params, "t_net.%d." % g, dtype=dtype)
        q_new = _prj(b_emb[:, None, :], params, "q_prj.%d." % g, dtype) + q_emb
        a_emb = _prj(b_emb[:, None, :], params, "a_prj.%d." % g, dtype) + a_emb
        q_emb = q_new
    joint = q_emb.sum(1) + a_emb.sum(1)
    return simple_classifier(joint, params, "classifier.", activation, dtype), att


def ffoe_cti_forward(v, q_tok, a_tok, params, glimpse, activation="relu", dtype=np.float32):
    """FFOE CTIModel.forward (src/FFOE/base_model.py:112-136).  Returns logits (B, num_ans)."""
    return _tan_forward(v, q_tok, a_tok, params, "t_att.", glimpse, activation, dtype)[0]


def mc_tan_forward(v, q_tok, a_tok, params, glimpse, activation="relu", dtype=np.float32):
    """MC TanModel.forward (src/MC/base_model.py:128-152; boxes `b` are unused).  Returns (logits (B, 2), att)."""
    return _tan_forward(v, q_tok, a_tok, params, "v_att.", glimpse, activation, dtype)


def ffoe_ban_forward(v, q_tok, params, glimpse, activation="relu", dtype=np.float32):
    """FFOE BanModel.forward without the counter (src/FFOE/base_model.py:37-67, `--use_counter` off = the default,
    src/FFOE/main.py:48).  Returns (logits, att (B, G, V, Q))."""
    q_emb = gru_forward_all(word_embedding(q_tok, params, "w_emb.", dtype), params, "q_emb.rnn.", dtype)
    att, _ = O.bi_attention(v, q_emb, params, "v_att.", dtype=dtype)
    acc = 0
    for g in range(glimpse):
        b_emb = O.bcnet_forward_with_weights(v, q_emb, att[:, g], params, "b_net.%d." % g, k=1, dtype=dtype)
        q_emb = _prj(b_emb[:, None, :], params, "q_prj.%d." % g, dtype) + q_emb
        acc = acc + q_emb                                   # torch.stack(q_emb_list, 1).sum(1)
    return simple_classifier(acc.sum(1), params, "classifier.", activation, dtype), att
