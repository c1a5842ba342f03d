"""ForwardTransformer training step on the GPU (forward_training.ForwardTransformerTrainer, every kernel through the C ABI)
against the float64 autograd oracle (oracle/tts_train_torch.py) on the same weights, batch and random draws.

Tolerances: loss 1e-5 relative, outputs 5e-4 absolute, gradients as explained in compare_grads."""
import numpy as np
import pytest
import torch

import estts  # noqa: F401
from estts import configs, weights
from oracle import tts_train_torch as OT

pytestmark = pytest.mark.gpu


def small_cfg(dropout_rate):
    return configs.forward_config(
        encoder_model_dimension=128, decoder_model_dimension=64, encoder_num_heads=[2, 2, 2], decoder_num_heads=[2, 2],
        encoder_feed_forward_dimension=256, decoder_feed_forward_dimension=192,
        encoder_attention_conv_filters=96, decoder_attention_conv_filters=160,
        postnet_conv_filters=48, postnet_conv_layers=3, mel_channels=80, vocab_size=40,
        encoder_max_position_encoding=128, decoder_max_position_encoding=512, dropout_rate=dropout_rate)


def batch(cfg, B, N, seed):
    rng = np.random.default_rng(seed)
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    durs = rng.integers(1, 5, size=(B, N)).astype(np.int32)
    for b in range(1, B):
        tokens[b, N - 2 * b:] = 0
        durs[b, N - 2 * b:] = 0
    T = int(durs.sum(1).max())
    mel = np.clip(rng.normal(0, 1.5, size=(B, T, 80)), -4, 4).astype(np.float32)
    for b in range(B):
        mel[b, durs[b].sum():] = 0
    return tokens, mel, durs


def grads_by_name(model):
    """Device gradients under the oracle's names (fused QKV split, conv kernels reshaped)."""
    st = model.store
    g = st.grad.cpu().numpy()
    out = {}
    for k, (o, sh) in st.table.items():
        n = int(np.prod(sh)) if len(sh) else 1
        v = g[o:o + n].reshape(sh)
        if k.endswith('wqkv.kernel'):
            d = sh[0]
            for j, nm in enumerate(('wq', 'wk', 'wv')):
                out[k.replace('wqkv', nm)] = v[:, j * d:(j + 1) * d]
        elif k.endswith('wqkv.bias'):
            d = sh[0] // 3
            for j, nm in enumerate(('wq', 'wk', 'wv')):
                out[k.replace('wqkv', nm)] = v[j * d:(j + 1) * d]
        else:
            out[k] = v
    return out


def compare_grads(model, ref):
    """Per-tensor and global agreement of the device gradients with the oracle's.

    The graph has kinks (ReLU gates, the sign() of the MAE loss).  The device works with ~1e-5 relative arithmetic
    noise, the oracle in float64, so in every step a few of the ~1e5 gated elements sit close enough to their kink to
    take the other branch on the device (which ones depends on the summation order inside the GEMMs).  A flipped gate
    changes the gradients of its own layer and of everything that precedes it in the forward pass, and nothing after
    it.  The oracle shows the size of the effect: perturbing its weights by 2e-6 (relative) moves single
    BatchNorm-beta gradients by 10 % and everything upstream of the flipped gate by ~1 %.  Hence: the tensors that
    follow the last ReLU (the mel projection and the tanh postnet) must agree to 5e-4 of their largest entry, every
    tensor to 35 % (a layout, sign or scale error gives >= 100 %), the median tensor to 2e-2, and the whole gradient
    vector to a cosine of 0.995.  Each backward kernel on its own is held to ~1e-4 against autograd in
    tests/test_gpu_train_ops.py."""
    mine = grads_by_name(model)
    errs, dot, n1, n2 = [], 0.0, 0.0, 0.0
    for k, r in ref['grads'].items():
        a = mine[k].reshape(-1).astype(np.float64)
        r = np.asarray(r).reshape(-1)
        dot += float(a @ r)
        n1 += float(a @ a)
        n2 += float(r @ r)
        scale = np.abs(r).max()
        if scale < 1e-7:           # e.g. conv biases in front of a BatchNorm: exactly zero in exact arithmetic
            assert np.abs(a).max() < 1e-5, k
            continue
        errs.append((float(np.abs(a - r).max() / scale), k))
    errs.sort(reverse=True)
    cosine = dot / np.sqrt(n1 * n2)
    median = errs[len(errs) // 2][0]
    after_last_gate = [(e, k) for e, k in errs if k.startswith(('decoder_postnet.', 'out.'))]
    print(f'gradient check: cosine {cosine:.6f}, worst {errs[0]}, median {median:.1e}, after the last gate {after_last_gate[0]}')
    assert errs[0][0] < 0.35, errs[:5]
    assert median < 2e-2, median
    assert after_last_gate and after_last_gate[0][0] < 5e-4, after_last_gate[:3]
    assert cosine > 0.995, cosine
    return errs[0]


def test_weights_round_trip_and_val_step():
    from estts.forward_training import ForwardTransformerTrainer
    from estts.forward_transformer import ForwardTransformer
    cfg = small_cfg(0.0)
    w = weights.init_forward_transformer(cfg, seed=3)
    model = ForwardTransformerTrainer(w, cfg)
    back = model.weights()
    assert set(back) == set(w)
    for k in w:
        assert np.array_equal(np.asarray(w[k], np.float32).reshape(back[k].shape), back[k]), k
    tokens, mel, durs = batch(cfg, 3, 17, seed=1)
    out = model.val_step(tokens, mel, durs)
    ref = ForwardTransformer(w, cfg, fold_ffn=False).call(tokens, target_durations=durs)
    assert np.abs(out['mel'].cpu().numpy() - ref['mel'].cpu().numpy()).max() < 1e-5
    l_mel = OT.masked_mae(torch.tensor(mel, dtype=torch.float64), ref['mel'].cpu().double())
    assert abs(float(out['losses']['mel']) - float(l_mel)) < 1e-5


def test_train_step_gradients_without_dropout():
    from estts.forward_training import ForwardTransformerTrainer
    cfg = small_cfg(0.0)
    w = weights.init_forward_transformer(cfg, seed=7)
    tokens, mel, durs = batch(cfg, 3, 19, seed=2)
    ref = OT.forward_train_step(w, cfg, tokens, mel, durs)
    model = ForwardTransformerTrainer(w, cfg)
    out = model.train_step(tokens, mel, durs)
    assert abs(float(out['loss']) - ref['loss']) < 1e-5 * max(1.0, abs(ref['loss']))
    assert abs(float(out['losses']['mel']) - ref['mel_loss']) < 1e-5
    assert abs(float(out['losses']['duration']) - ref['dur_loss']) < 1e-5
    assert np.abs(out['mel'].cpu().numpy() - ref['mel']).max() < 5e-4
    assert np.abs(out['duration'].cpu().numpy() - ref['duration']).max() < 3e-4
    compare_grads(model, ref)
    # BatchNorm moving statistics after one step
    state = model.store.numpy()
    for k, v in ref['new_stats'].items():
        assert np.abs(state[k] - v).max() < 1e-4 * max(1.0, np.abs(v).max()), k
    assert model.step == 1


def test_train_step_with_dropout_and_head_drop():
    from estts import train_ops as T
    from estts.forward_training import ForwardTransformerTrainer
    cfg = small_cfg(0.1)
    w = weights.init_forward_transformer(cfg, seed=9)
    B, N = 2, 15
    tokens, mel, durs = batch(cfg, B, N, seed=4)
    Tm = mel.shape[1]
    model = ForwardTransformerTrainer(w, cfg, seed=5)
    model.set_constants(drop_n_heads=1)
    seed_now = (5 * 1000003 + 0 * 7919 + 12345) & 0xFFFFFFFF
    # the same draws for the oracle: dropout masks from the device generator, head masks from the host generator
    shapes = [(B * N, 128)] + [(B * N, 128)] * 2 + [(B * N, 128)] * 2 + [(B * Tm, 64)] + [(B * Tm, 64)] * 2
    masks = []
    for site, sh in enumerate(shapes):
        ones = torch.ones(*sh, device='cuda')
        T.dropout(ones, ones, 0.1, seed_now, site)
        masks.append(ones.cpu().numpy().reshape(B, -1, sh[1]))
    rng = np.random.default_rng(seed_now)
    head_keep = []
    for H in cfg['encoder_num_heads'] + cfg['decoder_num_heads']:
        keep = np.ones((B, H))
        for b in range(B):
            keep[b, rng.permutation(H)[:1]] = 0.0
        head_keep.append(keep * (H / (H - 1)))
    ref = OT.forward_train_step(w, cfg, tokens, mel, durs, masks=masks, head_keep=head_keep)
    out = model.train_step(tokens, mel, durs)
    assert abs(float(out['loss']) - ref['loss']) < 2e-5 * max(1.0, abs(ref['loss']))
    assert np.abs(out['mel'].cpu().numpy() - ref['mel']).max() < 5e-4
    compare_grads(model, ref)


def test_train_step_with_decoder_prenet_dropout():
    """DecoderPrenet's dropout (layers.py:394-402) at rate 0.5 in the training step, against the oracle on the same masks."""
    from estts import train_ops as T
    from estts.forward_training import ForwardTransformerTrainer
    cfg = small_cfg(0.0)
    w = weights.init_forward_transformer(cfg, seed=21)
    B, N = 2, 14
    tokens, mel, durs = batch(cfg, B, N, seed=8)
    Tm = mel.shape[1]
    model = ForwardTransformerTrainer(w, cfg, seed=3)
    model.set_constants(decoder_prenet_dropout=0.5)
    seed_now = (3 * 1000003 + 0 * 7919 + 12345) & 0xFFFFFFFF
    pm = []
    for site, width in ((1000, cfg['decoder_feed_forward_dimension']), (1001, cfg['decoder_model_dimension'])):
        ones = torch.ones(B * Tm, width, device='cuda')
        T.dropout(ones, ones, 0.5, seed_now, site)
        pm.append(ones.cpu().numpy().reshape(B, Tm, width))
    assert 0.4 < (pm[0] == 0).mean() < 0.6
    ref = OT.forward_train_step(w, cfg, tokens, mel, durs, prenet_masks=pm)
    out = model.train_step(tokens, mel, durs)
    assert abs(float(out['loss']) - ref['loss']) < 2e-5 * max(1.0, abs(ref['loss']))
    err = np.abs(out['mel'].cpu().numpy() - ref['mel'])
    # batch-statistics BatchNorm over 74 rows: low-variance channels amplify the ~1e-5 error of the split-bf16 products by
    # rstd (up to 1 / sqrt(1e-3)); measured here 7e-5 mean / 1.1e-3 worst element, and 5e-5 / 4e-4 with the dropout off
    assert err.mean() < 2e-4 and err.max() < 3e-3
    compare_grads(model, ref)
    plain = OT.forward_train_step(w, cfg, tokens, mel, durs)
    assert abs(plain['loss'] - ref['loss']) > 1e-3


def test_two_adam_steps_follow_the_oracle_and_training_reduces_the_loss():
    from estts.forward_training import ForwardTransformerTrainer
    cfg = small_cfg(0.0)
    w = weights.init_forward_transformer(cfg, seed=13)
    tokens, mel, durs = batch(cfg, 2, 16, seed=6)
    model = ForwardTransformerTrainer(w, cfg)
    model.set_constants(learning_rate=1e-3, decoder_prenet_dropout=0.0, drop_n_heads=0)
    ref = OT.forward_train_step(w, cfg, tokens, mel, durs)
    state = {}
    w1 = OT.adam_update(w, ref['grads'], state, 1e-3, 1)
    first = model.train_step(tokens, mel, durs)
    mine = model.weights()
    checked = 0
    for k, g in ref['grads'].items():
        big = np.abs(g) > 0.25 * np.abs(g).max()       # entries whose sign a gate flip upstream cannot change
        if np.abs(g).max() < 1e-7 or not big.any():
            continue
        a = np.asarray(mine[k], np.float64).reshape(g.shape)
        assert np.abs(a - w1[k])[big].max() < 2e-5, k      # one Adam step moves every weight by ~lr = 1e-3
        checked += 1
    assert checked > 50
    losses = [float(first['loss'])]
    for _ in range(25):
        losses.append(float(model.train_step(tokens, mel, durs)['loss']))
    assert model.step == 26
    assert losses[-1] < 0.8 * losses[0], losses
    val = model.val_step(tokens, mel, durs)
    assert np.isfinite(float(val['loss']))


def test_val_step_and_inference_honour_decoder_prenet_dropout():
    """DecoderPrenet's dropout is on at inference in the reference (layers.py:394-402, training=True hard-coded), at the rate
    set_constants sets: val_step and ForwardTransformer.call against the float64 oracle on the device's own masks."""
    from estts import train_ops as T
    from estts.forward_training import ForwardTransformerTrainer
    from oracle import tts_numpy as O
    cfg = small_cfg(0.0)
    w = weights.init_forward_transformer(cfg, seed=23)
    B, N = 2, 13
    tokens, mel, durs = batch(cfg, B, N, seed=11)
    Tm = mel.shape[1]
    model = ForwardTransformerTrainer(w, cfg, seed=4)
    model.set_constants(decoder_prenet_dropout=0.5)
    seed_now = (4 * 1000003 + 0 * 7919 + 12345) & 0xFFFFFFFF
    pm = []
    for site, width in ((1000, cfg['decoder_feed_forward_dimension']), (1001, cfg['decoder_model_dimension'])):
        ones = torch.ones(B * Tm, width, device='cuda')
        T.dropout(ones, ones, 0.5, seed_now, site)
        pm.append(ones.cpu().numpy().astype(np.float64).reshape(B, Tm, width))
    ref = O.forward_transformer_call(w, cfg, tokens, durs, prenet_masks=pm)
    plain = O.forward_transformer_call(w, cfg, tokens, durs)
    out = model.val_step(tokens, mel, durs)
    assert np.abs(out['mel'].cpu().numpy() - ref['mel']).max() < 5e-4
    assert np.abs(plain['mel'] - ref['mel']).max() > 1e-2                 # the masks matter
    l_mel = OT.masked_mae(torch.tensor(mel, dtype=torch.float64), torch.tensor(ref['mel']))
    assert abs(float(out['losses']['mel']) - float(l_mel)) < 1e-4
    model.set_constants(decoder_prenet_dropout=0.0)
    out0 = model.val_step(tokens, mel, durs)
    assert np.abs(out0['mel'].cpu().numpy() - plain['mel']).max() < 5e-4


def test_global_lengths_rescale_losses_and_gradients():
    """train_step(global_lengths=(T, N)): the losses and their gradients are divided by the global batch's padded element
    count instead of this batch's.  The forward pass does not change, so the loss values, the gradient of the last
    BatchNorm's beta (a plain sum of d loss / d mel) and of the duration head's bias scale exactly by T_own / T and N_own / N."""
    from estts.forward_training import ForwardTransformerTrainer
    cfg = small_cfg(0.0)
    w = weights.init_forward_transformer(cfg, seed=17)
    tokens, mel, durs = batch(cfg, 3, 18, seed=5)
    Tm, N = mel.shape[1], tokens.shape[1]
    a, b = ForwardTransformerTrainer(w, cfg), ForwardTransformerTrainer(w, cfg)
    oa = a.train_step(tokens, mel, durs)
    ob = b.train_step(tokens, mel, durs, global_lengths=(Tm + 9, N + 5))
    fm, fd = Tm / (Tm + 9), N / (N + 5)
    # (the loss sums are float atomics: two runs of the same step differ in the last bits)
    assert abs(float(ob['losses']['mel']) - fm * float(oa['losses']['mel'])) < 3e-6 * max(1.0, float(oa['losses']['mel']))
    assert abs(float(ob['losses']['duration']) - fd * float(oa['losses']['duration'])) < 3e-6 * max(1.0, float(oa['losses']['duration']))
    ga, gb = grads_by_name(a), grads_by_name(b)
    last_beta = f"decoder_postnet.norm{cfg['postnet_conv_layers']}.beta"
    assert np.abs(ga[last_beta]).max() > 1e-4
    assert np.abs(gb[last_beta] - fm * ga[last_beta]).max() < 1e-6 * max(1.0, np.abs(ga[last_beta]).max())
    assert np.abs(gb['dur_pred.linear.bias'] - fd * ga['dur_pred.linear.bias']).max() < 1e-6
    with pytest.raises(ValueError):
        b.train_step(tokens, mel, durs, global_lengths=(Tm - 1, N))


def test_checkpoint_resume_equals_uninterrupted_training(tmp_path):
    """N steps + checkpoint (variables, BatchNorm statistics, Adam moments, step count, seed) + restore + N steps against 2 N
    uninterrupted steps.  Not bit-identical (float atomics order the column sums differently from run to run), so the
    weights are compared against how far one step moves them: Adam moves every weight by ~lr per step whatever its gradient's
    size, so two runs of the same four steps already differ by a few percent of lr on average (entries whose tiny gradients
    change sign between runs); resuming with zeroed moments is off by most of lr."""
    from estts.config_manager import ConfigManager
    from estts.forward_training import ForwardTransformerTrainer
    cfg = small_cfg(0.1)
    w = weights.init_forward_transformer(cfg, seed=19)
    data = [batch(cfg, 2, 15, seed=30 + i) for i in range(4)]
    lr = 1e-3

    def run(model, items):
        model.set_constants(learning_rate=lr, decoder_prenet_dropout=0.0, drop_n_heads=0)
        for d in items:
            model.train_step(*d)
        return model

    straight = run(ForwardTransformerTrainer(w, cfg, seed=6), data)
    first = run(ForwardTransformerTrainer(w, cfg, seed=6), data[:2])
    path = str(tmp_path / 'ckpt-2.npz')
    ConfigManager.save_weights(path, first.weights(), first.step, first.optimizer_state())
    w2, step, opt = ConfigManager.read_weights(path, with_optimizer=True)
    assert step == 2 and set(w2) == set(w)
    resumed = ForwardTransformerTrainer(w2, cfg, seed=999)
    resumed.load_optimizer_state(opt)
    assert resumed.step == 2 and resumed.seed == 6
    run(resumed, data[2:])
    cold = ForwardTransformerTrainer(w2, cfg, seed=6)           # what the old checkpoints gave: moments lost
    cold.iterations = 2
    run(cold, data[2:])
    ref = straight.store.flat.cpu().numpy()
    err = np.abs(resumed.store.flat.cpu().numpy() - ref)
    err_cold = np.abs(cold.store.flat.cpu().numpy() - ref)
    print(f'resume: mean |dw| {err.mean():.2e} (with the moments) vs {err_cold.mean():.2e} (without), lr {lr}')
    assert err.mean() < 0.1 * lr and err_cold.mean() > 10 * err.mean()


def tight_compare(model, ref, tol=1e-3):
    """Every gradient tensor within `tol` of its own largest entry (the oracle differentiated the device's own branches)."""
    mine = grads_by_name(model)
    errs = []
    for k, r in ref['grads'].items():
        a = mine[k].astype(np.float64)
        r = np.asarray(r).reshape(a.shape)
        scale = np.abs(r).max()
        if scale < 1e-7:
            assert np.abs(a).max() < 1e-5, k
            continue
        errs.append((float(np.abs(a - r).max() / scale), k))
    errs.sort(reverse=True)
    print(f'tight gradient check: worst {errs[:3]}, median {errs[len(errs) // 2][0]:.1e}')
    assert errs[0][0] < tol, errs[:5]


@pytest.mark.parametrize('dropout', [0.0, 0.1])
def test_train_step_gradients_tight_on_the_device_branches(dropout):
    """The oracle is given the branch the device took at every ReLU and the signs of its MAE terms (as it is given the
    dropout masks), so both differentiate the same piecewise-linear function: every gradient tensor must then agree to 1e-3
    of its largest entry (measured 1e-5..2e-4), which a wiring error upstream of any layer cannot pass."""
    from estts import train_ops as T
    from estts.forward_training import ForwardTransformerTrainer
    cfg = small_cfg(dropout)
    w = weights.init_forward_transformer(cfg, seed=27)
    B, N = 3, 17
    tokens, mel, durs = batch(cfg, B, N, seed=12)
    Tm = mel.shape[1]
    model = ForwardTransformerTrainer(w, cfg, seed=8)
    masks = head_keep = None
    if dropout > 0:
        model.set_constants(drop_n_heads=1)
        seed_now = (8 * 1000003 + 0 * 7919 + 12345) & 0xFFFFFFFF
        # dropout sites in call order: positional encoding, then per block the attention output and (dense blocks) the FFN
        shapes = [(B * N, 128)] * (1 + 2 * len(cfg['encoder_num_heads']) - (len(cfg['encoder_num_heads']) - cfg['encoder_dense_blocks'])) + \
                 [(B * Tm, 64)] * (1 + 2 * len(cfg['decoder_num_heads']) - (len(cfg['decoder_num_heads']) - cfg['decoder_dense_blocks']))
        masks = []
        for site, sh in enumerate(shapes):
            ones = torch.ones(*sh, device='cuda')
            T.dropout(ones, ones, dropout, seed_now, site)
            masks.append(ones.cpu().numpy().reshape(B, -1, sh[1]))
        rng = np.random.default_rng(seed_now)
        head_keep = []
        for H in cfg['encoder_num_heads'] + cfg['decoder_num_heads']:
            keep = np.ones((B, H))
            for b in range(B):
                keep[b, rng.permutation(H)[:1]] = 0.0
            head_keep.append(keep * (H / (H - 1)))
    model.gate_log = []
    out = model.train_step(tokens, mel, durs)
    gates = [g.float().cpu().numpy() for g in model.gate_log]
    model.gate_log = None
    ref = OT.forward_train_step(w, cfg, tokens, mel, durs, masks=masks, head_keep=head_keep, gates=gates)
    assert abs(float(out['loss']) - ref['loss']) < 2e-5 * max(1.0, abs(ref['loss']))
    assert np.abs(out['mel'].cpu().numpy() - ref['mel']).max() < 5e-4
    tight_compare(model, ref)


def test_backward_products_in_plain_bf16():
    """backward_precision='bf16': the forward pass (loss, outputs) is untouched, the gradient keeps its direction (cosine against
    the float64 oracle > 0.9995) with errors of the size of one bf16 rounding per product."""
    from estts.forward_training import ForwardTransformerTrainer
    cfg = small_cfg(0.0)
    w = weights.init_forward_transformer(cfg, seed=7)
    tokens, mel, durs = batch(cfg, 3, 19, seed=2)
    ref = OT.forward_train_step(w, cfg, tokens, mel, durs)
    model = ForwardTransformerTrainer(w, cfg)
    model.backward_precision = 'bf16'
    out = model.train_step(tokens, mel, durs)
    assert abs(float(out['loss']) - ref['loss']) < 1e-5 * max(1.0, abs(ref['loss']))
    mine = grads_by_name(model)
    dot = n1 = n2 = 0.0
    for k, r in ref['grads'].items():
        a, r = mine[k].reshape(-1).astype(np.float64), np.asarray(r).reshape(-1)
        dot, n1, n2 = dot + float(a @ r), n1 + float(a @ a), n2 + float(r @ r)
    cosine = dot / np.sqrt(n1 * n2)
    print(f'bf16 backward products: gradient cosine {cosine:.6f}')
    assert 0.9995 < cosine < 0.9999999
