#!/usr/bin/env python3
"""TEST / BENCH INFRASTRUCTURE: tests/golden/speech_clips.npz -- the bundled 16 kHz recordings the synthetic
utterances of bench.py and tests/test_decode_pipeline_gpu.py are tiled from (reference test/data: goforward.raw,
numbers.raw, something.raw, librivox/sense_and_sensibility_01_austen_64kb-0870.wav as staged by oracle/Makefile).
Run in the build container (needs oracle/_ref/data); the fixture travels with the repository."""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "_ref", "data")
out = {}
for name in ("goforward", "numbers", "something", "librivox-0870"):
    out[name.replace("-", "_")] = np.fromfile(os.path.join(DATA, name + ".raw"), dtype=np.int16)
np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "speech_clips.npz"), **out)
print({k: v.size for k, v in out.items()})
