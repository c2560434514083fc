"""Writes tests/golden/oracle_vs_reference.json and tests/golden/tolerance_table.json: the oracle end to end (VoxelGrid + SOR,
reduction, the reference's own compiled front end, patch selection, loop) against ALL of the reference's checked-in per-pair
results (results/4DPCReg/<e>_{Direct2Ref,Adaptive,Fixed}_TransMatrix.txt, 57 files; 46 distinct (target, source) pairs).
The tolerance of a file is twice the oracle's measured distance to it, floored at float print precision (2e-7 rad / 3e-7 m);
tests/test_oracle_golden.py and tests/test_gpu_configs.py assert against that table.  Needs /root/reference.

  python tools/golden_report.py            (about 3 minutes)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rootcause_golden as RC   # noqa: E402

AMAP = {2: 1, 3: 1, 4: 1, 5: 1, 6: 1, 7: 3, 8: 4, 9: 4, 10: 5, 11: 6, 12: 6, 13: 7, 14: 9, 15: 12, 16: 13, 17: 14, 18: 14,
        19: 14, 20: 14}                                                # recovered from the Adaptive files
FMAP = {e: (1 if e <= 4 else e - 3) for e in range(2, 21)}            # pairMode 3 (R.cpp:94-97)
DMAP = {e: 1 for e in range(2, 21)}
FLOOR = (2e-7, 3e-7)
SIGMA_FLOOR, VCM_FLOOR = 2e-5, 1.5e-12     # relative on the printed Std_ values; absolute on the 12-decimal VCM entries

if __name__ == "__main__":
    out, tol, stol, done = {}, {}, {}, {}
    for mode, M in (("Direct2Ref", DMAP), ("Adaptive", AMAP), ("Fixed", FMAP)):
        out[mode], tol[mode], stol[mode] = {}, {}, {}
        for e in range(2, 21):
            key = (M[e], e)
            if key not in done:
                c = RC.Case(M[e], e, mode)
                done[key] = (c, c.run()[0])
            c, r = done[key]
            Tg, _, _ = RC.G.parse_transmatrix_file(os.path.join(RC.G.REF_ROOT, "results/4DPCReg", "%d_%s_TransMatrix.txt" % (e, mode)))
            assert (Tg == c.Tg).all(), (mode, e)      # the same pair in another family: the reference wrote the same numbers
            out[mode][str(e)] = dict(target=M[e], d_angle_rad=r["da"], d_trans_m=r["dt"], d_sigma_rel=r["dstd"], d_vcm_abs=r["dvcm"], outer=r["outer"],
                                     inner=r["inner"], stable=r["stable"])
            tol[mode][str(e)] = [float("%.1e" % max(2 * r["da"], FLOOR[0])), float("%.1e" % max(2 * r["dt"], FLOOR[1]))]
            # a11 (calTransParaVCM, R.cpp:1273-1343): the six printed sigmas (relative) and the 6x6 matrix (absolute; the file
            # prints 12 decimals, half a step = 5e-13) of the same file, same rule
            stol[mode][str(e)] = [float("%.1e" % max(2 * r["dstd"], SIGMA_FLOOR)), float("%.1e" % max(2 * r["dvcm"], VCM_FLOOR))]
            print(mode, e, M[e], RC.fmt(r), "d_vcm %.1e" % r["dvcm"], flush=True)
    gold = os.path.join(RC.ROOT, "tests", "golden")
    with open(os.path.join(gold, "oracle_vs_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(gold, "tolerance_table.json"), "w") as f:
        json.dump(dict(unit=["rad", "m"], rule="max(2 x oracle-vs-file, [2e-7, 3e-7])", pair_map=dict(Adaptive=AMAP, Fixed=FMAP), tol=tol,
                       sigma_vcm_rule="[max(2 x rel. distance of the six Std_ values, 2e-5), max(2 x abs. distance of the VCM entries, 1.5e-12)]",
                       sigma_vcm_tol=stol), f, indent=1)
