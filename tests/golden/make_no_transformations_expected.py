"""Generates tests/golden/no_transformations_expected.json: the oracle's result for each of the 19 Direct2Ref pairs of the
reference's SECOND input set, data/data_synthetic/syntheticPC_no_transformations (the same 20 scans before the defined
transformations were applied: expected transformation = identity up to the method's accuracy; scans kept as fixtures under
tests/golden/inputs_no_transformations/).  The reference holds no result files for this set; what pins the oracle is the 57
result files of the transformed set (tests/test_oracle_golden.py).  Run in the build container:
    python tests/golden/make_no_transformations_expected.py
Front end: the reference's own codelibrary segmentation (oracle/_ref); preprocessing / loop: oracle/pwicp_oracle.c."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(os.path.dirname(HERE)), "piecewise-icp_amd")]
import numpy as np  # noqa: E402
import _golden as G  # noqa: E402
import _oracle as O  # noqa: E402
from pwicp_amd.pcd import read_pcd  # noqa: E402


def oracle_pair(p1, e, inputs):
    p2 = G.preprocess_4d(O, read_pcd(os.path.join(inputs, "Epoch_%03d.pcd" % e)))
    r1, r2, shift = G.reduce_pair(p1, p2)
    l1, n1 = O.ref_frontend(r1, 0.05)
    l2, n2 = O.ref_frontend(r2, 0.05)
    io = O.run_loop(r1, r2, O.select_patches(r1, l1, n1), O.select_patches(r2, l2, n2), 0.005, 0.005, 0.05, 0.05, 0.05, 0.004)
    Tf = G.final_matrix(io.T16, shift)
    k = io.n_outer
    return {"n_outer": int(k), "n_inner": [int(v) for v in io.n_inner[:k]], "n_stable": [int(v) for v in io.n_stable[:k]],
            "DTseries": [float(v) for v in io.DTseries[:k + 1]], "T_final": [float(v) for v in Tf.reshape(16)],
            "euler_rad": [float(v) for v in G.euler(Tf)], "t_m": [float(v) for v in Tf[:3, 3]]}


def main():
    inputs = os.path.join(HERE, "inputs_no_transformations")
    p1 = G.preprocess_4d(O, read_pcd(os.path.join(inputs, "Epoch_001.pcd")))
    out = {"set": "syntheticPC_no_transformations", "config": "configuration_4d.txt values: Res 0.005, SV 0.05, DTinit 0.05, DTmin 0.004",
           "pairs": {str(e): oracle_pair(p1, e, inputs) for e in range(2, 21)}}
    with open(os.path.join(HERE, "no_transformations_expected.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
