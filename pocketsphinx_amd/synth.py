"""Synthetic 16 kHz utterances for the throughput workloads (BASELINE.json configs[2] / configs[4]; SURVEY 8d
"white-noise-plus-tiled-speech at 16 kHz int16, default_rng(seed = utterance id)"): the bundled recordings
(tests/golden/speech_clips.npz) tiled with random gains and pauses over a low noise floor."""
import os

import numpy as np

_CLIPS = None


def clips():
    global _CLIPS
    if _CLIPS is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        z = np.load(os.path.join(root, "tests", "golden", "speech_clips.npz"))
        _CLIPS = [z[k].astype(np.float32) for k in sorted(z.files)]
    return _CLIPS


def utterance(utt_id, seconds=30.0, rate=16000):
    """int16 [seconds * rate]: deterministic in utt_id"""
    rng = np.random.default_rng(int(utt_id))
    n = int(round(seconds * rate))
    out = rng.integers(-12, 13, n).astype(np.float32)          # noise floor
    cl = clips()
    pos = int(rng.integers(0, rate // 2))
    while pos < n:
        c = cl[int(rng.integers(0, len(cl)))]
        g = float(rng.uniform(0.5, 1.0))
        m = min(c.size, n - pos)
        out[pos:pos + m] += g * c[:m]
        pos += m + int(rng.integers(rate // 5, rate))           # 0.2 - 1.0 s pause
    return np.clip(np.rint(out), -32768, 32767).astype(np.int16)


def batch(first_id, n_utt, seconds=30.0, rate=16000):
    """(pcm int16 [n_utt * n], samp_off int64 [n_utt + 1]) for utterances first_id .. first_id + n_utt - 1"""
    n = int(round(seconds * rate))
    pcm = np.empty(n_utt * n, np.int16)
    for u in range(n_utt):
        pcm[u * n:(u + 1) * n] = utterance(first_id + u, seconds, rate)
    return pcm, (np.arange(n_utt + 1, dtype=np.int64) * n)
