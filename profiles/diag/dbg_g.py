import sys, numpy as np
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..'))
from historymatching_amd.forward import ForwardPlan
from historymatching_amd import _lib
from tests.helpers import make_models, perms
import ctypes as C
variant=int(sys.argv[1])
_, gm = make_models(128,128)
plan=ForwardPlan(gm,1,0.025,1)
plan.set_variant(variant,0)
plan.set_inputs(perms(128,128,1,seed=3),transformed=False)
plan.pressure_only(0); plan.sync()
TX=plan.get_field("TX")[0]; TY=plan.get_field("TY")[0]; K=plan.get_field("K")[0]
G=np.empty(128*128*128); _lib.check(plan.lib.hm_fwd_get_field(plan.h,b"G",_lib.ptr(G)),"G")
NT=1024
def decode(i):
    Gi=G[i*16384:(i+1)*16384].reshape(8,NT,2)   # chunk, tid, 2
    A=np.zeros((128,128))
    tid=np.arange(NT); lane=tid&63; w=tid>>6; wr=w>>2; wc=w&3; lc=lane&15; lq=lane>>4
    for ti in range(2):
        for tj in range(2):
            for h in range(2):
                ch=((ti*2+tj)*2)+h
                for e in range(2):
                    r=2*h+e
                    rows=16*(2*wr+ti)+lq+4*r; cols=16*(2*wc+tj)+lc
                    A[rows,cols]=Gi[ch,:,e]
    return A
def Dblk(i):
    dg=TY[i,:-1]+TY[i,1:]+TX[i]+TX[i+1]
    if i==0: dg=dg.copy(); dg[0]+=2*K[0,0]
    return np.diag(dg)-np.diag(TY[i,1:128],1)-np.diag(TY[i,1:128],-1)
G0=decode(0); ref=np.linalg.inv(Dblk(0))
print('variant',variant,'block0 max rel err',np.abs(G0-ref).max()/np.abs(ref).max(), 'sym',np.abs(G0-G0.T).max())
err=np.abs(G0-ref); idx=np.unravel_index(err.argmax(),err.shape); print('worst at',idx)
bt=(err.reshape(8,16,8,16).max(axis=(1,3))/np.abs(ref).max()); np.set_printoptions(precision=1,linewidth=200); print(bt)
