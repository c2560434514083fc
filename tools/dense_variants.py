"""A/B of the dense 1-NN kernel variants on the real workload (first Stage-1 launch of the synthetic 1 M-point pair):
one subprocess per environment setting (the knobs are read once per process).  Run on the GPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, os, json, numpy as np
sys.path.insert(0, %r + "/piecewise-icp_amd")
import pwicp_amd as P
from pwicp_amd import synth
n = int(os.environ.get("DV_POINTS", "1000000")); r = 0.005
ctx = P.Context(0)
t, L = synth.make_tile(n, r); s, _ = synth.make_source(n, r, epoch=1); c = t.mean(0)
t = (t - c).astype(np.float32); s = (s - c).astype(np.float32)
if os.environ.get("DV_SHAPE") == "cliff":
    # cliff-like scene: steep faces (slope up to ~9) so that a column of the grid holds a tall stack of points
    for a in (t, s):
        a[:, 2] += (1.5 * np.sin(6.0 * a[:, 0])).astype(np.float32)
elif os.environ.get("DV_SHAPE") == "face_yz":
    # the tile turned into the y-z plane: a cliff looked at along x (no column layout fits)
    t = np.ascontiguousarray(t[:, [2, 0, 1]]); s = np.ascontiguousarray(s[:, [2, 0, 1]])
elif os.environ.get("DV_SHAPE") == "diagonal":
    c_, s_ = np.float32(np.cos(np.pi / 4)), np.float32(np.sin(np.pi / 4))
    for a in (t, s):
        x, z = a[:, 0].copy(), a[:, 2].copy()
        a[:, 0] = c_ * x + s_ * z; a[:, 2] = -s_ * x + c_ * z
l1, n1 = synth.grid_labels(t, 10 * r); l2, n2 = synth.grid_labels(s, 10 * r)
prm = P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r)
pair = P.Pair(ctx, t, l1, n1, s, l2, n2, prm); pair.set_profiling(1 | 4)
res = pair.run()
ms, nq, kb, edge = pair.bench_dense_nn(20)
best = 1e9
for _ in range(5):
    pair.reset(); rr = pair.run(); best = min(best, rr.t_loop_ms)
print("DV " + json.dumps({"ms": ms, "nq": nq, "kbar": kb, "edge": edge, "loop_ms": best, "d75": [res.d75[i] for i in range(res.n_outer)],
                          "dt": [float(x) for x in res.DTseries[:res.n_outer + 1]], "dense_in_run_ms": res.t_dense_nn_ms}))
''' % ROOT


def run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", WORKER], capture_output=True, text=True, env=env, timeout=900)
    for line in out.stdout.splitlines():
        if line.startswith("DV "):
            return json.loads(line[3:])
    return {"error": (out.stdout + out.stderr)[-1500:]}


if __name__ == "__main__":
    variants = [a.split(",") for a in sys.argv[1:]] or [["PWICP_DISC_CELL_FACTOR=0"], ["PWICP_DISC_CELL_FACTOR=1.5"]]
    ref = None
    for v in variants:
        env = dict(kv.split("=", 1) for kv in v if "=" in kv)
        r = run(env)
        if "error" in r:
            print(v, "ERROR", r["error"]); continue
        if ref is None:
            ref = r
        same = r["d75"] == ref["d75"] and r["dt"] == ref["dt"]
        print("%-60s dense %.2f us (in the run, with its event records: %.2f)  kbar %.1f  edge %.4f  nq %d  loop %.3f ms  same=%s" %
              (" ".join(v), r["ms"] * 1e3, r["dense_in_run_ms"] * 1e3, r["kbar"], r["edge"], r["nq"], r["loop_ms"], same))
