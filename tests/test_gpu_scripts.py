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
