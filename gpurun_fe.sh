#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python - <<'PY'
import sys,time,numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'piecewise-icp_amd')
import pwicp_amd as P, _data
from pwicp_amd import synth
ctx=P.Context(0); r=0.005
t,L=synth.make_tile(1000000,r)
t0=time.time(); nb=ctx.knn(t,45,2*r); print('GPU knn-45 on 1M pts: %.2f s'%(time.time()-t0))
t0=time.time(); lab,n=ctx.frontend_segment(t,10*r,45,r); print('GPU-kNN front end 1M: %.2f s, nsv %d'%(time.time()-t0,n))
t0=time.time(); lab2,n2=P.frontend_segment(t,10*r); print('host front end 1M: %.2f s, nsv %d equal %s'%(time.time()-t0,n2,np.array_equal(lab,lab2)))
PY
