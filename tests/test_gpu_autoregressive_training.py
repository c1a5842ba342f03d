"""AutoregressiveTransformer training step on the GPU (autoregressive_training.AutoregressiveTransformerTrainer, every kernel
through the C ABI) against the float64 autograd oracle (oracle/tts_train_torch.autoregressive_train_step) on the same weights,
batch and random draws.

The oracle is handed the branch the device took at every ReLU and the sign of every MAE term (trainer.gate_log), so both
differentiate the same piecewise-linear function and every gradient tensor must agree to 1e-3 of its largest entry; the
query / key projections of the self-attentions get 1e-2 (measured: median 3e-5, all others below 2e-4; the first decoder
self-attention's wq / wk reach 2e-3 .. 4e-3, varying from run to run with the order of the float atomics, when the reduction
factor leaves 7 decoder steps: its logits are large, the softmax is peaked and dS = P o (dP - delta) cancels).
Losses 2e-5 relative, outputs 1e-3 absolute (batch-statistics BatchNorm over few rows amplifies the split-bf16 error)."""
import numpy as np
import pytest
import torch

import estts  # noqa: F401
from estts import configs, weights
from oracle import tts_train_torch as OT

pytestmark = pytest.mark.gpu


def small_cfg(system_type, dropout_rate, dense_blocks=(1, 1)):
    return configs.autoregressive_config(
        system_type=system_type, encoder_model_dimension=64, decoder_model_dimension=128, encoder_num_heads=[2, 2],
        decoder_num_heads=[2, 4], encoder_dense_blocks=dense_blocks[0], decoder_dense_blocks=dense_blocks[1],
        encoder_feed_forward_dimension=128, decoder_feed_forward_dimension=160, encoder_prenet_dimension=64,
        decoder_prenet_dimension=80, encoder_attention_conv_filters=96, decoder_attention_conv_filters=112,
        postnet_conv_filters=48, postnet_conv_layers=3, ref_encoder_filters=[8, 8, 16], ref_encoder_gru_cell_units=32,
        gst_style_embed_dim=64, gst_multi_num_heads=2, gst_heads=5, speaker_embed_dim=32, max_r=3, vocab_size=30,
        encoder_max_position_encoding=64, decoder_max_position_encoding=256, dropout_rate=dropout_rate, mel_channels=80)


def batch(cfg, B, N, T, seed):
    """tokens [B, N], mel [B, T + 1, 80] with the start / end vectors, stop [B, T + 1] (1 = frame, 2 = end, 0 = padding)."""
    rng = np.random.default_rng(seed)
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    mel = np.clip(rng.normal(0, 1.5, size=(B, T + 1, 80)), -4, 4).astype(np.float32)
    stop = np.ones((B, T + 1), np.int32)
    for b in range(B):
        n_tok, n_mel = N - 2 * b, T + 1 - 3 * b
        tokens[b, n_tok:] = 0
        mel[b, 0] = cfg['mel_start_value']
        mel[b, n_mel - 1] = cfg['mel_end_value']
        mel[b, n_mel:] = 0
        stop[b, n_mel - 1] = 2
        stop[b, n_mel:] = 0
    spk = rng.normal(0, 1, size=(B, 1, cfg['speaker_embed_dim'])).astype(np.float32)
    return tokens, mel, stop, spk


def grads_by_name(model):
    """Device gradients under the oracle's names and layouts."""
    return model.gradients()


def tight_compare(model, ref, tol=1e-3, tol_qk=1e-2, floor_qk=5e-4):
    """Every tensor within tol of its largest entry; the self-attention query / key projections within tol_qk + an absolute
    floor: when a softmax saturates (the first decoder block sees prenet outputs times sqrt(d)) their true gradient vanishes
    and what the device returns is the rounding of delta = <dctx, ctx> against dP, ~1e-5 of quantities that are large
    (measured 5e-5 .. 1.2e-4 absolute from run to run, against 1e-2 .. 1e-1 for the gradients that matter)."""
    mine = grads_by_name(model)
    errs, zeros = [], []
    assert set(ref['grads']) == set(mine), set(ref['grads']) ^ set(mine)
    for k, r in ref['grads'].items():
        a = mine[k].astype(np.float64)
        r = np.asarray(r).reshape(a.shape)
        scale = np.abs(r).max()
        diff = float(np.abs(a - r).max())
        if '.sarn.mha.wq.' in k or '.sarn.mha.wk.' in k or k.startswith('decoder_prenet.'):
            # (the prenet sits behind the first decoder self-attention and inherits the rounding of its query / key gradients:
            # 1e-4 .. 2e-4 with dropout on, up to 5e-3 when the decoder runs in inference mode)
            assert diff < tol_qk * scale + (floor_qk if '.sarn.' in k else 0.0), (k, diff, scale)
            continue
        if scale < 1e-7:
            zeros.append((float(np.abs(a).max()), k))
            continue
        errs.append((diff / scale, k))
    errs.sort(reverse=True)
    print(f'tight gradient check: worst {errs[:3]}, median {errs[len(errs) // 2][0]:.1e}')
    assert errs[0][0] < tol, errs[:5]
    # gradients that vanish identically (a bias in front of a batch normalisation): what the device returns is the rounding of a
    # column sum of that layer's output gradient, so it is held against the gradient of the same layer's kernel
    for v, k in zeros:
        sibling = np.abs(np.asarray(ref['grads'].get(k.replace('.bias', '.kernel'), 0.0))).max()
        assert v < 1e-5 + 1e-3 * sibling, (k, v, sibling)
    return errs


def reference_step(model, w, cfg, tokens, mel, stop, spk, r, dropout_rate, prenet_rate, head_drop, seed, use_style_loss=False,
                   flags=(True, True, True)):
    """The oracle on the draws and branches of the device step that just ran."""
    from estts import train_ops as T
    B, N = tokens.shape
    tar_real, tar_mel, tar_stop = mel[:, 1:], mel[:, :-1][:, 0::r], stop[:, 1:]
    Tq = tar_mel.shape[1]
    de, dd = cfg['encoder_model_dimension'], cfg['decoder_model_dimension']
    seed_now = (seed * 1000003 + 0 * 7919 + 12345) & 0xFFFFFFFF
    masks = None
    if dropout_rate > 0:
        shapes = [(B * N, de)]
        for i in range(len(cfg['encoder_num_heads'])):
            shapes += [(B * N, de)] * (2 if i < cfg['encoder_dense_blocks'] else 1)
        shapes += [(B * Tq, dd)]
        for i in range(len(cfg['decoder_num_heads'])):
            shapes += [(B * Tq, dd)] * (3 if i < cfg['decoder_dense_blocks'] else 2)
        masks = []
        for site, sh in enumerate(shapes):
            ones = torch.ones(*sh, device='cuda')
            T.dropout(ones, ones, dropout_rate, seed_now, site)
            masks.append(ones.cpu().numpy().reshape(B, -1, sh[1]))
    pm = None
    if prenet_rate > 0:
        pm = []
        for site, width in ((1000, cfg['decoder_prenet_dimension']), (1001, dd)):
            ones = torch.ones(B * Tq, width, device='cuda')
            T.dropout(ones, ones, prenet_rate, seed_now, site)
            pm.append(ones.cpu().numpy().reshape(B, Tq, width))
    head_keep = None
    if head_drop:
        rng = np.random.default_rng(seed_now)
        head_keep = []
        heads = list(cfg['encoder_num_heads'])
        if 'style' in cfg['system_type']:
            heads.append(cfg['gst_multi_num_heads'])
        for H in cfg['decoder_num_heads']:
            heads += [H, H]
        if use_style_loss:
            heads.append(cfg['gst_multi_num_heads'])
        te, ts, td = flags              # the device draws head masks only where the sub-network is in training mode
        live = [te] * len(cfg['encoder_num_heads']) + ([ts] if 'style' in cfg['system_type'] else [])
        live += [td] * (2 * len(cfg['decoder_num_heads'])) + ([ts] if use_style_loss else [])
        heads = [H for H, on in zip(heads, live) if on]
        for H in heads:
            keep = np.ones((B, H))
            for b in range(B):
                keep[b, rng.permutation(H)[:head_drop]] = 0.0
            head_keep.append(keep * (H / (H - head_drop)))
    gates = [g.cpu().numpy().astype(np.float64) for g in model.gate_log]
    return OT.autoregressive_train_step(w, cfg, tokens, tar_mel, tar_real, tar_stop, spk_embed=spk, r=r, masks=masks,
                                        head_keep=head_keep, prenet_masks=pm, gates=gates, stop_scaling=8.0,
                                        use_style_loss=use_style_loss, train_text_encoder=flags[0], train_style_encoder=flags[1],
                                        train_decoder=flags[2])


def check_step(system_type, r, dropout_rate=0.0, prenet_rate=0.0, head_drop=0, dense_blocks=(1, 1), seed=3, trainer=None,
               use_style_loss=False, flags=(True, True, True)):
    from estts.autoregressive_training import AutoregressiveTransformerTrainer
    cfg = small_cfg(system_type, dropout_rate, dense_blocks)
    w = weights.init_autoregressive_transformer(cfg, seed=11)
    tokens, mel, stop, spk = batch(cfg, 3, 13, 20, seed=5)
    if 'speaker' not in system_type:
        spk = None
    model = (trainer or AutoregressiveTransformerTrainer)(w, cfg, seed=seed)
    model._compile(stop_scaling=8.0)
    model.set_constants(reduction_factor=r, decoder_prenet_dropout=prenet_rate, drop_n_heads=head_drop)
    model.gate_log = []
    out = model.train_step(mel, tokens, stop, spk, flags[0], flags[1], flags[2], apply=False, use_style_loss=use_style_loss)
    ref = reference_step(model, w, cfg, tokens, mel, stop, spk, r, dropout_rate, prenet_rate, head_drop, seed, use_style_loss, flags)
    assert abs(float(out['loss']) - ref['loss']) < 2e-5 * max(1.0, abs(ref['loss'])), (float(out['loss']), ref['loss'])
    for k in ('output', 'stop_prob', 'mel_linear'):
        assert abs(float(out['losses'][k]) - ref['losses'][k]) < 2e-5 * max(1.0, ref['losses'][k]), k
    for k in ('final_output', 'mel_linear', 'stop_prob'):
        assert np.abs(out[k].cpu().numpy() - ref[k]).max() < 1e-3, k
    tight_compare(model, ref)
    return model, ref, out


@pytest.mark.parametrize('system_type,r', [('text', 1), ('speaker_text', 3), ('speaker_text', 2), ('style_text', 1),
                                           ('speaker_style_text', 2)])
def test_train_step_gradients(system_type, r):
    check_step(system_type, r)


def test_train_step_with_conv_blocks_dropout_head_drop_and_prenet_dropout():
    check_step('speaker_text', 2, dropout_rate=0.1, prenet_rate=0.5, head_drop=1, dense_blocks=(1, 1))


def test_style_system_with_dropout_and_head_drop():
    """The reference encoder's own attention drops heads too (layers.py:577-578)."""
    model, ref, out = check_step('speaker_style_text', 1, dropout_rate=0.1, prenet_rate=0.25, head_drop=1)
    assert np.abs(out['gst_output'].cpu().numpy() - ref['gst_output']).max() < 1e-4
    state = model.store.numpy()
    for k, v in ref['new_stats'].items():
        assert np.abs(state[k] - v).max() < 1e-4 * max(1.0, np.abs(v).max()), k


@pytest.mark.parametrize('head_drop', [0, 1])
def test_style_loss(head_drop):
    """use_style_loss (traning_steps.py:41-49): the reference encoder a second time over the post-net output, mean squared
    distance of the two style embeddings added to the loss; its gradient reaches the post-net, the decoder and, twice, the
    style encoder, whose BatchNorm statistics move twice."""
    model, ref, out = check_step('style_text', 1, dropout_rate=0.1 if head_drop else 0.0, head_drop=head_drop, use_style_loss=True)
    assert ref['style_loss'] > 1e-6 and abs(float(out['style_loss']) - ref['style_loss']) < 2e-5 * max(1.0, ref['style_loss'])
    state = model.store.numpy()
    for k, v in ref['new_stats'].items():
        assert np.abs(state[k] - v).max() < 1e-4 * max(1.0, np.abs(v).max()), k
    with pytest.raises(ValueError):
        check_step('speaker_text', 1, use_style_loss=True)


@pytest.mark.parametrize('flags', [(False, True, True), (True, False, True), (True, True, False), (False, False, False)])
def test_sub_networks_in_inference_mode(flags):
    """train_text_encoder / train_style_encoder / train_decoder = False (traning_steps.py:24-29): that sub-network runs without
    dropout and head drop and with BatchNormalization on its moving statistics, and is differentiated all the same."""
    model, ref, out = check_step('speaker_style_text', 2, dropout_rate=0.1, prenet_rate=0.25, head_drop=1, flags=flags)
    state = model.store.numpy()
    moved = {k for k in state if k.endswith('moving_mean') and np.abs(state[k] - np.asarray(model_w0(model)[k])).max() > 0}
    for name, on in zip(('text_encoder', 'style_encoder', 'decoder'), flags):
        assert any(k.startswith(name) for k in moved) == on, (name, on)


def model_w0(model):
    """The variables the check_step trainer was built from (seed 11 of small_cfg)."""
    return weights.init_autoregressive_transformer(model.config, seed=11)


def test_weights_round_trip_and_adam_step():
    """weights() gives back what the trainer was built from; one applied step follows Keras Adam on the oracle's gradients;
    a few more reduce the loss; the trained variables load into the inference class."""
    from estts.autoregressive_training import AutoregressiveTransformerTrainer
    from estts.autoregressive_transformer import AutoregressiveTransformer
    cfg = small_cfg('speaker_text', 0.0)
    w = weights.init_autoregressive_transformer(cfg, seed=2)
    tokens, mel, stop, spk = batch(cfg, 2, 11, 16, seed=9)
    model = AutoregressiveTransformerTrainer(w, cfg)
    back = model.weights()
    assert set(back) == set(w)
    for k in w:
        assert np.array_equal(np.asarray(w[k], np.float32), back[k]), k
    model._compile(stop_scaling=8.0, optimizer={'learning_rate': 1e-3})
    model.set_constants(reduction_factor=2)
    model.gate_log = []
    first = model.train_step(mel, tokens, stop, spk)
    ref = reference_step(model, w, cfg, tokens, mel, stop, spk, 2, 0.0, 0.0, 0, 0)
    model.gate_log = None
    w1 = OT.adam_update(w, ref['grads'], {}, 1e-3, 1)
    mine = model.weights()
    checked = 0
    for k, g in ref['grads'].items():
        big = np.abs(g) > 0.25 * np.abs(g).max()
        if np.abs(g).max() < 1e-7 or not big.any():
            continue
        assert np.abs(np.asarray(mine[k], np.float64) - w1[k])[big].max() < 2e-5, k
        checked += 1
    assert checked > 40
    losses = [float(first['loss'])]
    for _ in range(20):
        losses.append(float(model.train_step(mel, tokens, stop, spk)['loss']))
    assert model.step == 21 and losses[-1] < 0.8 * losses[0], losses
    infer = AutoregressiveTransformer(model.weights(), cfg)
    infer._compile(stop_scaling=8.0)
    infer.set_constants(reduction_factor=2)
    val = infer.val_step(tokens, mel, stop, spk)
    assert np.isfinite(float(val['tts_loss']))
    val2 = model.val_step(tokens, mel, stop, spk)               # the trainer's own validation pass: the same model
    assert abs(float(val2['tts_loss']) - float(val['tts_loss'])) < 1e-4
    model.train_step(mel, tokens, stop, spk)
    assert abs(float(model.val_step(tokens, mel, stop, spk)['tts_loss']) - float(val['tts_loss'])) > 1e-6   # rebuilt after a step


def test_global_frames_makes_ragged_shards_equal_the_concatenated_batch():
    """Data-parallel normalisation: two shards that pad to different frame counts, each given global_frames, sum to `world`
    times the gradient of the concatenated batch (the optimizer divides by world).  Every sub-network runs in inference mode
    (no dropout, BatchNormalization on its moving statistics) and the system has no style encoder, so an utterance's
    gradient does not depend on what else is in its batch or on its padding."""
    from estts.autoregressive_training import AutoregressiveTransformerTrainer
    cfg = small_cfg('speaker_text', 0.0)
    w = weights.init_autoregressive_transformer(cfg, seed=5)
    shards = [batch(cfg, 2, 11, 18, seed=1), batch(cfg, 2, 11, 26, seed=2)]
    Tg = max(s[1].shape[1] for s in shards) - 1

    def padded(mel, stop):
        B, T1 = stop.shape
        m = np.zeros((B, Tg + 1, mel.shape[2]), np.float32)
        s = np.zeros((B, Tg + 1), np.int32)
        m[:, :T1], s[:, :T1] = mel, stop
        return m, s

    def grad_of(tokens, mel, stop, spk, **kw):
        m = AutoregressiveTransformerTrainer(w, cfg)
        m._compile(stop_scaling=8.0, optimizer={'learning_rate': 1e-3})
        m.set_constants(reduction_factor=1)
        out = m.train_step(mel, tokens, stop, spk, train_text_encoder=False, train_style_encoder=False, train_decoder=False,
                           apply=False, **kw)
        return m.store.grad.double().cpu(), float(out['loss'])

    parts = [grad_of(*s, global_frames=Tg) for s in shards]
    cat = [np.concatenate([s[0] for s in shards])]
    pm = [padded(s[1], s[2]) for s in shards]
    cat += [np.concatenate([p[0] for p in pm]), np.concatenate([p[1] for p in pm]), np.concatenate([s[3] for s in shards])]
    whole, loss = grad_of(*cat)
    summed = parts[0][0] + parts[1][0]
    scale = float(whole.norm()) * 2
    assert float((summed - 2 * whole).norm()) < 2e-3 * scale, (float((summed - 2 * whole).norm()), scale)
    assert abs((parts[0][1] + parts[1][1]) / 2 - loss) < 1e-4 * abs(loss)
    # without global_frames the short shard weighs its utterances more: the same comparison must fail clearly
    plain = grad_of(*shards[0])[0] + parts[1][0]
    assert float((plain - 2 * whole).norm()) > 2e-2 * scale


def test_gru_cell_wider_than_one_launch_takes_the_two_launch_steps(monkeypatch):
    """The one-launch backward step of the reference encoder's GRU holds K = 3 * units <= 1536 in flight.  With the limit
    lowered under this model's cell (3 * 32 = 96) the trainer must choose the two-launch steps on its own and still match the
    oracle; the inference encoder must give the same style embedding on either path."""
    from estts import ops
    from estts.autoregressive_transformer import ReferenceEncoderGST
    monkeypatch.setattr(ops, 'GRU_FUSED_MAX_H', 64)
    model, _, _ = check_step('style_text', 1)
    assert model.fused_gru is False
    cfg = small_cfg('style_text', 0.0)
    w = weights.init_autoregressive_transformer(cfg, seed=5)
    x = torch.from_numpy(np.random.default_rng(2).normal(size=(3, 45, 80)).astype(np.float32)).cuda()
    enc = ReferenceEncoderGST(w, 'style_encoder', cfg, torch.device('cuda'))
    monkeypatch.setattr(ops, 'GRU_FUSED_MAX_H', 16)
    two, _ = enc.run(x, 'bf16x3')
    monkeypatch.setattr(ops, 'GRU_FUSED_MAX_H', 1536)
    one, _ = enc.run(x, 'bf16x3')
    assert float((two - one).abs().max()) < 2e-6
