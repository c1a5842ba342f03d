"""The command-line scripts end to end on the GPU (each in a process of its own, as a user runs them): duration extraction
feeding train_forward, and WaveRNN vocoding of a directory of mels and of a wav file."""
import os
import subprocess
import sys
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*args):
    r = subprocess.run([sys.executable, '-u', *args], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_extract_durations_then_train_forward(tmp_path):
    out = tmp_path / 'durations'
    log = run('scripts/extract_durations.py', '--config', 'config/wavernn', '--synthetic', '5', '--out', str(out), '--best', '--binary',
              '--fix_jumps', '--fill_mode_next')
    assert '5 utterances' in log
    files = sorted(out.glob('*.npz'))
    assert len(files) == 5
    for f in files:
        with np.load(f) as z:
            assert z['durations'].sum() == z['mel'].shape[0] and len(z['durations']) == len(z['phonemes'])
            assert z['durations'].dtype == np.int32 and (z['durations'] >= 0).all()
    run('scripts/train_forward.py', '--config', 'config/wavernn', '--data', str(out), '--max-steps', '2', '--session_name',
        f'pytest_{os.getpid()}')


def test_train_autoregressive(tmp_path):
    """scripts/train_autoregressive.py on the shipped configuration (speaker_text, reduction factor 10), then on a copy with a
    style system and the mutual-information heads switched on (MINE, a batch of their own)."""
    import shutil
    import yaml
    session = f'pytest_{os.getpid()}'
    log = run('scripts/train_autoregressive.py', '--config', 'config/wavernn', '--synthetic', '--max-steps', '2', '--session_name', session)
    cfg_dir = tmp_path / 'wavernn'
    shutil.copytree(os.path.join(ROOT, 'config', 'wavernn'), cfg_dir)
    with open(cfg_dir / 'autoregressive_config.yaml') as f:
        cfg = yaml.safe_load(f)
    cfg.update(system_type='speaker_style_text', use_mine=True, mine_sep_call=True,
               mine_batch_size_schedule=[[0, 6], [80000, 4]], tts_batch_size=4, mine_dense_hidden_units=[64, 32])
    with open(cfg_dir / 'autoregressive_config.yaml', 'w') as f:
        yaml.safe_dump(cfg, f)
    with open(cfg_dir / 'data_config.yaml') as f:
        data = yaml.safe_load(f)
    data['log_directory'] = str(tmp_path / 'logs')
    with open(cfg_dir / 'data_config.yaml', 'w') as f:
        yaml.safe_dump(data, f)
    run('scripts/train_autoregressive.py', '--config', str(cfg_dir), '--synthetic', '--max-steps', '2', '--session_name', session)
    saved = sorted(p.name for p in (tmp_path / 'logs').rglob('*.npz'))
    assert 'ckpt-2.npz' in saved and sum(n.startswith('mine') for n in saved) == 3, saved
    # a restart resumes the heads too: variables, Adam moments and step count, the smoothed exponential terms
    first = {p.name: p for p in (tmp_path / 'logs').rglob('mine*-2.npz')}
    log = run('scripts/train_autoregressive.py', '--config', str(cfg_dir), '--synthetic', '--max-steps', '4', '--session_name', session)
    assert log.count('restored head') == 3 and 'at head step 2' in log, log[-2000:]
    for name, p2 in first.items():
        p4 = p2.with_name(name.replace('-2.npz', '-4.npz'))
        with np.load(p2) as a, np.load(p4) as b:
            assert int(a['__iterations__']) == 2 and int(b['__iterations__']) == 4
            assert a['__adam_m__'].shape == b['__adam_m__'].shape and np.abs(b['__adam_m__']).max() > 0
            assert '__exp_terms__' in b.files and '__rng__' in b.files
            moved = [k for k in a.files if not k.startswith('__') and not np.array_equal(a[k], b[k])]
            assert moved, 'the resumed head did not train'


def test_gen_wavernn(tmp_path):
    rng = np.random.default_rng(0)
    mels = tmp_path / 'mels'
    mels.mkdir()
    for i, T in enumerate((40, 65)):
        np.save(mels / f'utt{i}.npy', rng.uniform(0, 1, (80, T)).astype(np.float32))
    out = tmp_path / 'gen'
    log = run('scripts/gen_wavernn.py', '--mels', str(mels), '--out', str(out), '-t', '3000', '-o', '150')
    assert '2 utterance(s)' in log
    for i, T in enumerate((40, 65)):
        with wave.open(str(out / f'0k_steps_utt{i}_gen_batched_target3000_overlap150.wav'), 'rb') as f:
            assert f.getframerate() == 16000 and f.getnframes() == (T - 1) * 200
    sr = 16000
    t = np.arange(sr) / sr
    tone = tmp_path / 'tone.wav'
    with wave.open(str(tone), 'wb') as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes((0.3 * np.sin(2 * np.pi * 220 * t) * 32767).astype('<i2').tobytes())
    run('scripts/gen_wavernn.py', '--file', str(tone), '--out', str(out), '--unbatched')
    assert (out / '__tone.wav__0k_steps_target.wav').exists() and (out / '__tone.wav__0k_steps_gen_NOT_BATCHED.wav').exists()


def test_train_wavernn_cadence(tmp_path):
    """scripts/train_wavernn.py with the reference loop's cadence (train_wavernn.py:15-64): three steps with a checkpoint in
    between (test mels vocoded + checkpoint_<k>k_steps.pyt), latest weights and a log line per epoch; a second start resumes
    from the latest weights and stops at total_steps unless forced."""
    import torch
    ck, out = tmp_path / 'ckpt', tmp_path / 'out'
    common = ['scripts/train_wavernn.py', '--synthetic', '--mode', 'RAW', '--batch_size', '4', '--checkpoints', str(ck), '--outputs', str(out),
              '--checkpoint_every', '2', '--gen_at_checkpoint', '2', '--epoch_iters', '3', '--gen_target', '1500', '--gen_overlap', '100']
    log = run(*common, '--total_steps', '3')
    assert 'New WaveRNN Training Session' in log and 'Training Complete.' in log
    assert (ck / 'latest_weights.pyt').exists() and (ck / 'checkpoint_0k_steps.pyt').exists()
    assert len((ck / 'log.txt').read_text().splitlines()) == 2          # (3 - 0) // 3 + 1 = 2 whole epochs, as the reference counts
    wavs = sorted(p.name for p in out.glob('*.wav'))
    assert wavs == ['0k_steps_1_gen_batched_target1500_overlap100.wav', '0k_steps_2_gen_batched_target1500_overlap100.wav'], wavs
    sd = torch.load(ck / 'latest_weights.pyt', map_location='cpu', weights_only=True)
    assert int(sd['step']) == 6
    mid = torch.load(ck / 'checkpoint_0k_steps.pyt', map_location='cpu', weights_only=True)
    assert int(mid['step']) == 6                   # written at steps 2, 4 and 6 under the same name (k = step // 1000)
    # a second start resumes from the latest weights: past total_steps nothing runs, with a higher total one more epoch does
    log = run(*common, '--total_steps', '3')
    assert 'Loading Weights' in log and 'Training Complete.' in log
    assert int(torch.load(ck / 'latest_weights.pyt', map_location='cpu', weights_only=True)['step']) == 6
    run(*common, '--total_steps', '8')
    assert int(torch.load(ck / 'latest_weights.pyt', map_location='cpu', weights_only=True)['step']) == 9
