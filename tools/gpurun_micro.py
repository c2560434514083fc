import sys, time, numpy as np
import os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_+'/tests'); sys.path.insert(0,R_+'/piecewise-icp_amd')
import pwicp_amd as P
from pwicp_amd import synth
ctx=P.Context(0); r=0.005
n=1000000
t,L=synth.make_tile(n,r); c=t.mean(0); t=(t-c).astype(np.float32)
l1,n1=synth.grid_labels(t,10*r)
prm=P.Params(r,r,10*r,10*r,1,10*r,0.8*r)
for name,shift in [('near 0.2r',0.2*r),('1.0h',2*r),('1.5h',3*r),('2.5h',5*r),('4.5h',9*r)]:
    s=t.copy(); s[:,2]+=np.float32(shift)
    l2,n2=synth.grid_labels(s,10*r)
    pair=P.Pair(ctx,t,l1,n1,s,l2,n2,prm); pair.set_profiling(1|4)
    # no run: bench over ALL source patches
    ms,nq,kb,edge=pair.bench_dense_nn(5)
    print('%-10s %.3f ms/launch  %d queries  kbar %.1f  -> %.2f Gq/s   %.1f ns/query-wave'%(name,ms,nq,kb,nq/ms/1e6, ms*1e6/(nq/64)))
    pair.close()
# real workload: first stage-1 dense launch replay
from pwicp_amd import synth
t,L=synth.make_tile(n,r); s,_=synth.make_source(n,r,epoch=1); c=t.mean(0); t=(t-c).astype(np.float32); s=(s-c).astype(np.float32)
l1,n1=synth.grid_labels(t,10*r); l2,n2=synth.grid_labels(s,10*r)
pair=P.Pair(ctx,t,l1,n1,s,l2,n2,prm); pair.set_profiling(1|4); res=pair.run()
print('loop %.3f ms, dense %.3f ms x%d, kbar %.1f, outer %d'%(res.t_loop_ms,res.t_dense_nn_ms,res.n_dense_nn_launches,res.dense_kbar,res.n_outer))
ms,nq,kb,edge=pair.bench_dense_nn(10); print('replay: %.3f ms/launch %d queries kbar %.1f'%(ms,nq,kb))
