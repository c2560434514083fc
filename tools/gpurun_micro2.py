import sys, time, numpy as np
import os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_+'/tests'); sys.path.insert(0,R_+'/piecewise-icp_amd')
import pwicp_amd as P
from pwicp_amd import synth
ctx=P.Context(0); r=0.005; n=1000000
prm=P.Params(r,r,10*r,10*r,1,10*r,0.8*r)
t,L=synth.make_tile(n,r); s,_=synth.make_source(n,r,epoch=1); c=t.mean(0); t=(t-c).astype(np.float32); s=(s-c).astype(np.float32)
l1,n1=synth.grid_labels(t,10*r); l2,n2=synth.grid_labels(s,10*r)
mode=sys.argv[1] if len(sys.argv)>1 else 'real'
if mode=='near':
    s=t.copy(); s[:,2]+=np.float32(0.2*r); l2,n2=l1,n1
pair=P.Pair(ctx,t,l1,n1,s,l2,n2,prm); pair.set_profiling(1|4)
if mode=='real': res=pair.run()
ms,nq,kb,edge=pair.bench_dense_nn(20); print(mode,'replay: %.3f ms/launch %d queries kbar %.1f'%(ms,nq,kb))
