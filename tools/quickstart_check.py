import sys, numpy as np
sys.path.insert(0,'' + __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))) + '/piecewise-icp_amd')
import pwicp_amd as P
from pwicp_amd.pcd import read_pcd
raw_target=read_pcd('' + __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))) + '/tests/golden/inputs/Epoch_001.pcd'); raw_source=read_pcd('' + __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))) + '/tests/golden/inputs/Epoch_002.pcd')
ctx = P.Context(0)
t = ctx.preprocess(raw_target, 0.005, 14, 5.0)
s = ctx.preprocess(raw_source, 0.005, 14, 5.0)
c = t[:, :3].mean(0); t[:, :3] -= c; s[:, :3] -= c
lt, nt = ctx.frontend_segment(t, 0.05, 45, 0.005)
ls, ns = ctx.frontend_segment(s, 0.05, 45, 0.005)
pair = P.Pair(ctx, t, lt, nt, s, ls, ns, P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004))
res = pair.run()
print(res.status, res.n_outer, np.array(res.T16).reshape(4,4)[:3,3])
