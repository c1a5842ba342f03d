"""CPU oracle for the mutual-information heads (float64 PyTorch autograd).

TEST INFRASTRUCTURE ONLY (tests/).

Restates MINE.measure_mi / MINE.call and CLUB.call (TransformerTTS/model/MINE.py:40-149,160-200) over MineNetFirstOrder /
CLUBNet (TransformerTTS/model/layers.py:471-485,507-523), and the gradient train_models_step ascends
(TransformerTTS/model/traning_steps.py:76-81: the tape differentiates -mi with respect to the head's variables).  The random
choices of the reference (one character position, batch shuffles) are INPUTS here.

PARITY STATUS vs TensorFlow: unpinned (TensorFlow is not installable here and the reference holds no fixtures for these
heads); the formulas follow the source line by line, including tf.reduce_max carrying gradient.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _stack(x, w, p, n_hidden, out_act=None):
    for i in range(n_hidden):
        x = torch.relu(F.linear(x, w[f'{p}.fc{i}.kernel'].t(), w[f'{p}.fc{i}.bias']))
    x = F.linear(x, w[p + '.fc_proj.kernel'].t(), w[p + '.fc_proj.bias'])
    return torch.tanh(x) if out_act == 'tanh' else x


def measure_mi(joint, marginal, div_type, betas, smoothing, prev):
    """MINE.py:40-95.  prev: [n_beta, 2] array.  -> (loss tensor, exp_terms list of [term1_exp, term2_exp])."""
    cur, old = smoothing, 1 - smoothing
    exp_terms = []
    if div_type == 'KL':
        term1 = joint.mean()
        t2 = cur * torch.exp(marginal).mean() + old * prev[0][1]
        return term1 - torch.log(t2), [[0.0, float(t2.detach())]]
    loss = 0
    for i, b in enumerate(betas):
        p1, p2 = prev[i][0], prev[i][1]
        e1 = e2 = torch.tensor(0.0, dtype=joint.dtype)
        if b == 0:
            mx = ((1 - b) * marginal).max()
            term1 = joint.mean()
            e2 = cur * torch.exp((1 - b) * marginal - mx).mean() + old * p2
            term2 = (1 / (1 - b)) * (torch.log(e2) + mx)
        elif b == 1:
            mx = (-b * joint).max()
            e1 = cur * torch.exp(-b * joint - mx).mean() + old * p1
            term1 = -(1 / b) * (torch.log(e1) + mx)
            term2 = marginal.mean()
        else:
            mx1, mx2 = (-b * joint).max(), ((1 - b) * marginal).max()
            e1 = cur * torch.exp(-b * joint - mx1).mean() + old * p1
            term1 = -(1 / b) * (torch.log(e1) + mx1)
            e2 = cur * torch.exp((1 - b) * marginal - mx2).mean() + old * p2
            term2 = (1 / (1 - b)) * (torch.log(e2) + mx2)
        loss = loss + term1 - term2
        exp_terms.append([float(e1.detach()), float(e2.detach())])
    return loss, exp_terms


def _pick(text_embed, speaker_embed, draws, dt):
    text = torch.as_tensor(np.asarray(text_embed)).to(dt)[:, draws['char'], :]
    text_sh = text[torch.as_tensor(np.asarray(draws['text_perm'])).long()]
    spk = spk_sh = None
    if speaker_embed is not None:
        spk = torch.as_tensor(np.asarray(speaker_embed)).to(dt).reshape(text.shape[0], -1)
        spk_sh = spk[torch.as_tensor(np.asarray(draws['speaker_perm'])).long()]
    return text, text_sh, spk, spk_sh


def mine_call(w, n_hidden, pair_type, div_type, betas, smoothing, prev, text_embed, style_embed, speaker_embed, draws,
              dtype=torch.float64):
    """-> dict(mi, exp_terms, grads of -mi by variable name)."""
    wt = {k: torch.as_tensor(np.asarray(v)).to(dtype).clone().requires_grad_(True) for k, v in w.items()}
    text, text_sh, spk, spk_sh = _pick(text_embed, speaker_embed, draws, dtype)
    style = None if style_embed is None else torch.as_tensor(np.asarray(style_embed)).to(dtype).reshape(text.shape[0], -1)
    j, m = {'style_text': ([style, text], [style, text_sh]), 'style_speaker': ([style, spk], [style, spk_sh]),
            'text_speaker': ([text, spk], [text, spk_sh]),
            'style_text_speaker': ([style, text, spk], [style, text_sh, spk_sh])}[pair_type]
    joint = _stack(torch.cat(j, dim=1), wt, 'mine_net', n_hidden)
    marginal = _stack(torch.cat(m, dim=1), wt, 'mine_net', n_hidden)
    mi, exp_terms = measure_mi(joint, marginal, div_type, betas, smoothing, np.asarray(prev, np.float64))
    (-mi).backward()
    return {'mi': float(mi.detach()), 'exp_terms': exp_terms, 'grads': {k: t.grad.numpy() for k, t in wt.items()}}


def club_call(w, n_hidden, pair_type, text_embed, style_embed, speaker_embed, draws, dtype=torch.float64):
    """-> dict(lld, bound, grads of -lld by variable name)."""
    wt = {k: torch.as_tensor(np.asarray(v)).to(dtype).clone().requires_grad_(True) for k, v in w.items()}
    text, text_sh, spk, spk_sh = _pick(text_embed, speaker_embed, draws, dtype)
    style = None if style_embed is None else torch.as_tensor(np.asarray(style_embed)).to(dtype).reshape(text.shape[0], -1)
    x, y, ys = {'style_text': (style, text, text_sh), 'style_speaker': (style, spk, spk_sh),
                'text_speaker': (text, spk, spk_sh)}[pair_type]
    mu = _stack(x, wt, 'net_mu', n_hidden)
    log_var = _stack(x, wt, 'net_log_var', n_hidden, 'tanh')
    positive = -(mu - y) ** 2 / 2. / torch.exp(log_var)
    negative = -(mu - ys) ** 2 / 2. / torch.exp(log_var)
    lld = positive.sum(-1).mean()
    bound = (positive.sum(-1) - negative.sum(-1)).mean()
    (-lld).backward()
    return {'lld': float(lld.detach()), 'bound': float(bound.detach()), 'grads': {k: t.grad.numpy() for k, t in wt.items()}}
