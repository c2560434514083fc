"""Helpers for the end-to-end known-answer tests against the reference's checked-in results."""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF_ROOT = "/root/reference"


def parse_transmatrix_file(path):
    """<e>_Direct2Ref_TransMatrix.txt (layout written at src/Registration.cpp:492-539 of the reference)."""
    L = [l.strip() for l in open(path).read().split("\n")]
    T = np.array([[float(v) for v in L[i].split()] for i in range(1, 5)])
    iv = L.index("6x6 Variance-Covariance Matrix of transformation parameters:")
    V = np.array([[float(v) for v in L[i].split()] for i in range(iv + 1, iv + 7)])
    stds = [float(l.split("=")[1].split()[0]) for l in L if l.startswith("Std_")]
    return T, V, np.array(stds)


def epoch_path(e):
    p = os.path.join(GOLD, "inputs", "Epoch_%03d.pcd" % e)
    if os.path.exists(p):
        return p
    p = os.path.join(REF_ROOT, "data/data_synthetic/syntheticPC_with_transformations", "Epoch_%03d.pcd" % e)
    return p if os.path.exists(p) else None


def preprocess_4d(O, cloud, res=0.005):
    """PCpreprocessing(cloud, out, true, Res, 14, 5.0): the 4D path (Registration.cpp:415-416)."""
    return O.sor(O.voxel_grid(cloud, res), 14, 5.0)


def reduce_pair(p1, p2):
    """Registration.cpp:419-436: subtract the (float-accumulated) centroid of the preprocessed target."""
    acc = np.array([np.cumsum(p1[:, d], dtype=np.float32)[-1] for d in range(3)], np.float32)
    cen = (acc / np.float32(len(p1))).astype(np.float32)
    shift = (np.float32(-1) * cen).astype(np.float32)
    r1 = p1.copy(); r1[:, :3] = (p1[:, :3] + shift[None, :]).astype(np.float32)
    r2 = p2.copy(); r2[:, :3] = (p2[:, :3] + shift[None, :]).astype(np.float32)
    return r1, r2, shift


def final_matrix(T16, shift):
    """T_final = S^-1 * T * S in float (Registration.cpp:461)."""
    T = np.array(T16, dtype=np.float32).reshape(4, 4)
    S = np.eye(4, dtype=np.float32); S[:3, 3] = shift
    Si = np.eye(4, dtype=np.float32); Si[:3, 3] = -shift
    return ((Si @ T).astype(np.float32) @ S).astype(np.float32)


def euler(T):
    T = np.asarray(T, float)
    ay = -np.arcsin(T[2, 0])
    return np.array([np.arctan2(T[2, 1] / np.cos(ay), T[2, 2] / np.cos(ay)), ay,
                     np.arctan2(T[1, 0] / np.cos(ay), T[0, 0] / np.cos(ay))])


def printed_sigmas(V):
    """The six Std_ lines of a result file from a 6x6 VCM (Registration.cpp:524-537: mgon = 1000 * 200/pi * sqrt, mm = 1000 * sqrt)."""
    d = np.diag(np.asarray(V, float).reshape(6, 6))
    return np.concatenate([1000 * 63.6619772368 * np.sqrt(d[:3]), 1000 * np.sqrt(d[3:])])
