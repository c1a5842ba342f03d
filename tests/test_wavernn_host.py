"""Host-side WaveRNN logic (CPU): the composite upsampling filter and fold bookkeeping against the oracle."""
import numpy as np
import pytest

from wavernn_common import small_hp
from estts import configs, weights
from estts.wavernn import composite_upsample_filter, decode_mu_law, fold_starts, xfade_and_unfold
from oracle import wavernn_numpy as W


@pytest.mark.parametrize('hp', [small_hp('RAW'), configs.wavernn_hparams(), configs.wavernn_hparams_22k()],
                         ids=['small', 'hop200', 'hop256'])
def test_composite_filter_reproduces_the_stretch_conv_chain(hp):
    w = weights.init_wavernn(hp, seed=5)
    scales, pad, hop = hp['voc_upsample_factors'], hp['voc_pad'], hp['hop_length']
    ups = [w[f'upsample.up_layers.{2 * j + 1}.weight'] for j in range(len(scales))]
    fir, kmin = composite_upsample_filter(ups, scales)
    assert fir.shape[0] == hop and pad - kmin - (fir.shape[1] - 1) >= 0 and pad + kmin >= 0
    T = 9
    rng = np.random.default_rng(0)
    m = rng.uniform(0, 1, size=(1, hp['num_mels'], T + 2 * pad))
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    ref, _ = W.upsample_network(m, w64, scales, hp['voc_res_blocks'], pad)      # (1, T*hop, feat)
    mp = m[0].T                                                                   # padded frames [T+2pad, feat]
    got = np.zeros_like(ref[0])
    for s in range(T * hop):
        frame, phase = divmod(s, hop)
        for j in range(fir.shape[1]):
            got[s] += fir[phase, j] * mp[frame + pad - (kmin + j)]
    np.testing.assert_allclose(got, ref[0], atol=1e-6)


def test_fold_starts_match_fold_with_overlap():
    for total, target, overlap in [(10, 2, 1), (12, 2, 1), (276, 60, 12), (143360, 11000, 550), (100, 11000, 550)]:
        x = np.arange(total, dtype=np.float64).reshape(1, total, 1)
        folded = W.fold_with_overlap(x, target, overlap)
        starts = fold_starts(total, target, overlap)
        assert len(starts) == folded.shape[0]
        for i, s in enumerate(starts):
            assert folded[i, 0, 0] == (s if s < total else 0)


def test_host_unfold_and_mu_law_equal_oracle():
    rng = np.random.default_rng(1)
    y = rng.uniform(-1, 1, size=(4, 84))
    np.testing.assert_array_equal(xfade_and_unfold(y, 60, 12), W.xfade_and_unfold(y, 60, 12))
    np.testing.assert_array_equal(decode_mu_law(y, 512, False), W.decode_mu_law(y, 512, False))
