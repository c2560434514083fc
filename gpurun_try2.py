import sys, time, numpy as np, ctypes as C
sys.path.insert(0,'tests'); sys.path.insert(0,'piecewise-icp_amd')
import _oracle as O
import pwicp_amd as P
from pwicp_amd import synth
ctx=P.Context(0)
r=0.005
def ang(T):
    T=np.asarray(T,float).reshape(4,4); ay=-np.arcsin(T[2,0]); return np.array([np.arctan2(T[2,1]/np.cos(ay),T[2,2]/np.cos(ay)),ay,np.arctan2(T[1,0]/np.cos(ay),T[0,0]/np.cos(ay))])
def cmp_run(n, use_ref=True):
    t,L=synth.make_tile(n,r); s,Tgt=synth.make_source(n,r,epoch=1)
    c=t.mean(0); t=(t-c).astype(np.float32); s=(s-c).astype(np.float32)
    t0=time.time()
    if use_ref:
        l1,n1=O.ref_frontend(t,10*r); l2,n2=O.ref_frontend(s,10*r)
    else:
        l1,n1=synth.grid_labels(t,10*r); l2,n2=synth.grid_labels(s,10*r)
    print('n',n,'labels',n1,n2,'%.1fs'%(time.time()-t0))
    P1=O.select_patches(t,l1,n1); P2=O.select_patches(s,l2,n2)
    G1=ctx.selectPatches(t,l1,n1); G2=ctx.selectPatches(s,l2,n2)
    for nm,a,b in [('off',P1.off,G1['off']),('pat',P1.pat,G1['pat']),('src',P1.src,G1['src']),('ct',P1.ct,G1['ct']),('bp',P1.bp,G1['bp']),('bpstd',P1.bpstd,G1['bpstd']),('ctstd',P1.ctstd,G1['ctstd']),('off2',P2.off,G2['off']),('pat2',P2.pat,G2['pat']),('bpstd2',P2.bpstd,G2['bpstd'])]:
        same = a.shape==b.shape and np.array_equal(a,b)
        print('  select',nm,a.shape,b.shape,'EQUAL' if same else 'DIFF', '' if same or a.shape!=b.shape else np.abs(a.astype(float)-b).max())
    nr,ok=ctx.patchNormals(P1.pat,P1.off)
    on=np.zeros((P1.m,3),np.float32); ook=np.zeros(P1.m,np.uint8)
    for i in range(P1.m):
        a,b,c3=C.c_float(),C.c_float(),C.c_float()
        seg=np.ascontiguousarray(P1.pat[P1.off[i]:P1.off[i+1]])
        ook[i]=O.lib().orc_cal_patch_normal(O._p(seg),len(seg),C.byref(a),C.byref(b),C.byref(c3)); on[i]=(a.value,b.value,c3.value)
    print('  normals equal',np.array_equal(on,nr[:,:3]),'ok equal',np.array_equal(ook,ok),'maxdiff',np.abs(on-nr[:,:3]).max())
    prm=P.Params(r,r,10*r,10*r,1,10*r,0.8*r)
    t0=time.time(); io=O.run_loop(t,s,P1,P2,r,r,10*r,10*r,10*r,0.8*r); to=time.time()-t0
    t0=time.time(); pair=P.Pair(ctx,t,l1,n1,s,l2,n2,prm); tc=time.time()-t0
    res=pair.run(check=False)
    print('  oracle: status',io.status,'outer',io.n_outer,'inner',list(io.n_inner[:io.n_outer]),'stable',list(io.n_stable[:io.n_outer]),'loop %.3fs'%io.t_loop_s)
    print('  gpu   : status',res.status,'outer',res.n_outer,'inner',list(res.n_inner[:res.n_outer]),'stable',list(res.n_stable[:res.n_outer]),'loop %.2f ms'%res.t_loop_ms,'create %.2fs'%tc)
    print('  DT o',[float(x) for x in io.DTseries[:io.n_outer+1]]); print('  DT g',[float(x) for x in res.DTseries[:res.n_outer+1]])
    To=np.array(io.T16); Tg=np.array(res.T16)
    print('  T equal',np.array_equal(To,Tg),'max|dT|',np.abs(To-Tg).max(),'dang',np.abs(ang(To)-ang(Tg)).max(),'dtr',np.abs(To.reshape(4,4)[:3,3]-Tg.reshape(4,4)[:3,3]).max())
    Vo=np.array(io.VCM); Vg=np.array(res.VCM); print('  VCM rel diff',np.abs(Vo-Vg).max()/np.abs(Vo).max())
    print('  d75 o',[io.d75[i] for i in range(io.n_outer)],'g',[res.d75[i] for i in range(res.n_outer)])
    print('  ncorr o',io.n_corr,'g',res.n_corr,'dense ms',res.t_dense_nn_ms,'launches',res.n_dense_nn_launches,'inner ms',res.t_inner_ms,'kbar',res.dense_kbar)
    Tf=Tg.reshape(4,4).astype(float); print('  vs GT: dang',np.abs(ang(Tf)-ang(Tgt)).max())
    pair.reset(); res2=pair.run(check=False); print('  rerun equal T',np.array_equal(np.array(res2.T16),Tg),'loop %.2f ms'%res2.t_loop_ms)
    ms,nq,kb,edge=pair.bench_dense_nn(10); print('  dense NN bench: %.3f ms/launch, %d queries, kbar %.1f, edge %.4f -> %.1f Mq/s'%(ms,nq,kb,edge,nq/ms/1e3))
    return pair
cmp_run(20000); cmp_run(100000)
cmp_run(1000000, use_ref=False)
