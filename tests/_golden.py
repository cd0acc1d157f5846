"""Helpers for the golden fixtures written by oracle/pin_against_reference.py."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
INDEX = json.load(open(os.path.join(GOLD, 'index.json')))


def cases(kind):
    return sorted(k for k, v in INDEX.items() if v['kind'] == kind)


def load(name):
    with np.load(os.path.join(GOLD, name + '.npz')) as d:
        return {k: d[k] for k in d.files}


def has(gold, key):
    return key in gold or key + '__idx' in gold


def relerr(actual, gold, key):
    """max|actual-gold| / max|gold| for a fixture stored in full or as seeded samples."""
    a = np.asarray(actual, dtype=np.float64)
    if key in gold:
        g = gold[key].astype(np.float64)
        assert a.shape == g.shape, (key, a.shape, g.shape)
        d = np.abs(g).max()
        return float(np.abs(a - g).max() / (d if d > 0 else 1.0))
    idx, val, stat = gold[key + '__idx'], gold[key + '__val'], gold[key + '__stat']
    assert tuple(int(v) for v in stat[2:]) == a.shape, (key, a.shape, stat[2:])
    flat = a.ravel()
    d = np.abs(val).max()
    e_samp = np.abs(flat[idx] - val).max() / d
    e_norm = abs(np.sqrt((flat ** 2).sum()) - stat[0]) / stat[0]
    return float(max(e_samp, e_norm))
