"""Shared synthetic inputs for the tests."""
import numpy as np
from pwicp_amd import synth

R = 0.005


def pair(n, epoch=1, offset=(0.0, 0.0, 0.0), reduce=True):
    tgt, _ = synth.make_tile(n, R, offset=offset)
    src, Tgt = synth.make_source(n, R, epoch=epoch, offset=offset)
    if reduce:
        c = tgt.mean(axis=0)
        tgt = (tgt - c).astype(np.float32)
        src = (src - c).astype(np.float32)
    return tgt, src, Tgt


def euler(T):
    T = np.asarray(T, float).reshape(4, 4)
    ay = -np.arcsin(T[2, 0])
    return np.array([np.arctan2(T[2, 1] / np.cos(ay), T[2, 2] / np.cos(ay)), ay,
                     np.arctan2(T[1, 0] / np.cos(ay), T[0, 0] / np.cos(ay))])


def params(manual=True):
    import pwicp_amd as P
    return P.Params(R, R, 10 * R, 10 * R, 1 if manual else 0, 10 * R, 0.8 * R)
