"""Shared fixtures for golden-vector tests."""
import functools
import os

import numpy as np

from conftest import sub

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=None)
def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


@functools.lru_cache(maxsize=None)
def sensevoice_setup(cfg_name, seed=0):
    cfg = getattr(sub("config"), cfg_name)()
    ck = sub("checkpoints").synth_sensevoice_checkpoint(cfg, seed)
    return cfg, ck


def golden_cases(g):
    for i in range(int(g["n_cases"])):
        p = f"c{i}_"
        yield i, {k[len(p):]: v for k, v in g.items() if k.startswith(p)}


def kaldi_audio(seed, n):
    return sub("checkpoints").synth_audio("kaldi", 1, int(n), seed=int(seed))[0, 0]
