"""TEST INFRASTRUCTURE ONLY — NumPy restatement of the two recurrent models on the hot path.

P1  pepper_variant TransducerGRU: 2 x bidirectional LSTM(256) + 5 x Linear(512)/SELU + Linear(3) + softmax
    /root/reference/pepper_variant/modules/python/models/simple_model.py:48-82
P2  pepper (polisher) TransducerGRU: bi-GRU(128) encoder -> bi-GRU(128) decoder -> Linear(256->5), run as
    the 19-window sliding loop with hidden carry and softmax accumulation of
    /root/reference/pepper/modules/python/models/simple_model.py:27-42 and models/predict.py:47-97

The arithmetic lives in a third-party dependency of the reference (torch.nn.LSTM/GRU/Linear/SELU/
Softmax, PyTorch 1.10.0 pinned in /root/reference/requirements.txt:3); the formulas restated here
are PyTorch's published cell definitions (gate order i,f,g,o for LSTM and r,z,n for GRU, both
biases added, reverse direction concatenated after forward). Parity pinning: this file is checked
against the reference's own model classes imported from /root/reference with identical weights
(tests/golden/make_rnn_golden.py, torch 2.10 CPU) and the resulting vectors are committed under
tests/golden/rnn_golden.npz.

Used only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import numpy as np

SELU_SCALE = 1.0507009873554805
SELU_ALPHA = 1.6732632423543772


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x [B,T,K] -> out [B,T,H]; zero initial (h,c)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = np.zeros((B, H), x.dtype)
    c = np.zeros((B, H), x.dtype)
    out = np.zeros((B, T, H), x.dtype)
    pre = x @ w_ih.T + (b_ih + b_hh)
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        g = pre[:, t] + h @ w_hh.T
        i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = _sigmoid(f) * c + _sigmoid(i) * np.tanh(gg)
        h = _sigmoid(o) * np.tanh(c)
        out[:, t] = h
    return out


def bilstm(x, w, prefix):
    outs = []
    for suffix, rev in (("", False), ("_reverse", True)):
        outs.append(lstm_direction(x, w[prefix + ".weight_ih_l0" + suffix], w[prefix + ".weight_hh_l0" + suffix],
                                   w[prefix + ".bias_ih_l0" + suffix], w[prefix + ".bias_hh_l0" + suffix], rev))
    return np.concatenate(outs, axis=2)


def selu(x):
    return SELU_SCALE * np.where(x > 0, x, SELU_ALPHA * (np.exp(np.minimum(x, 0)) - 1.0))


def softmax(x, axis):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def p1_forward(weights, images, dtype=np.float64, taps=False):
    """images int8 [B,33,26] -> probs [B,3] (dtype). taps=True also returns (enc_out, dec_out, logits)."""
    w = {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}
    x = np.asarray(images).astype(dtype)          # dataloader_predict.py:82-93: raw int8 cast, no scaling
    enc = bilstm(x, w, "encoder")
    dec = bilstm(enc, w, "decoder")
    y = dec.reshape(dec.shape[0], -1)             # torch.flatten(start_dim=1): index t*512+k
    for i in range(1, 6):
        y = selu(y @ w["linear_%d.weight" % i].T + w["linear_%d.bias" % i])
    logits = y @ w["output_layer_type.weight"].T + w["output_layer_type.bias"]
    probs = softmax(logits, 1)
    if taps:
        return probs, enc, dec, logits
    return probs


def gru_direction(x, h0, w_ih, w_hh, b_ih, b_hh, reverse):
    """x [B,T,K], h0 [B,H] -> out [B,T,H], h_final [B,H]"""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = h0.copy()
    out = np.zeros((B, T, H), x.dtype)
    pre = x @ w_ih.T + b_ih
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        gh = h @ w_hh.T + b_hh
        gi = pre[:, t]
        r = _sigmoid(gi[:, :H] + gh[:, :H])
        z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        out[:, t] = h
    return out, h


def bigru(x, hidden, w, prefix):
    """hidden [2,B,H] (0 = forward, 1 = reverse) -> out [B,T,2H], hidden_final [2,B,H]"""
    outs, hs = [], []
    for d, (suffix, rev) in enumerate((("", False), ("_reverse", True))):
        o, h = gru_direction(x, hidden[d], w[prefix + ".weight_ih_l0" + suffix], w[prefix + ".weight_hh_l0" + suffix],
                             w[prefix + ".bias_ih_l0" + suffix], w[prefix + ".bias_hh_l0" + suffix], rev)
        outs.append(o)
        hs.append(h)
    return np.concatenate(outs, axis=2), np.stack(hs)


def p2_window(w, x, hidden):
    """one TransducerGRU.forward (pepper/.../simple_model.py:27-42): x [B,100,10], hidden [2,B,H]
    -> logits [B,100,5], hidden_final [2,B,H] (the decoder's, seeds the NEXT window's encoder)"""
    enc, h_enc = bigru(x, hidden, w, "gru_encoder")
    dec, h_dec = bigru(enc, h_enc, w, "gru_decoder")
    return dec @ w["dense1.weight"].T + w["dense1.bias"], h_dec


def p2_forward(weights, images, dtype=np.float64, seq_len=1000, window=100, jump=50):
    """images uint8 [B,1000,10] -> (labels uint8 [B,1000], acc [B,1000,5]); predict.py:47-97"""
    w = {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}
    x = np.asarray(images).astype(dtype)
    B = x.shape[0]
    H = w["gru_encoder.weight_hh_l0"].shape[1]
    hidden = np.zeros((2, B, H), dtype)
    acc = np.zeros((B, seq_len, w["dense1.weight"].shape[0]), dtype)
    for i in range(0, seq_len, jump):
        if i + window > seq_len:
            break
        logits, hidden = p2_window(w, x[:, i:i + window], hidden)
        acc[:, i:i + window] += softmax(logits, 2)
    labels = acc.argmax(axis=2).astype(np.uint8)   # torch.max returns the first maximal index, as argmax
    return labels, acc
