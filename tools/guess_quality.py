import sys, numpy as np
sys.path.insert(0,'/root/repo')
from quadruped_ctrl_amd import workloads as W
from oracle import kron_model as K
def analyze(b, idx, name):
    tp=fp=fn=0; its=[]; v0s=[]; allv=0
    for i in idx:
        h=b["horizon"]; H,g=K.assemble(b,i)
        stance,rows=K.stance_constraints(b["gait"][i],h,b["mu"],b["f_max"])
        vi=np.array([3*k+a for k in stance for a in range(3)],int)
        Hinv=K.sweep_inverse(H[np.ix_(vi,vi)])
        x,Wf,it=K.dual_active_set(Hinv,g[vi],rows)
        n=len(vi); m=len(rows)
        Cm=np.zeros((m,n)); bv=np.zeros(m)
        for r,(a,b2,rhs) in enumerate(rows):
            Cm[r,a[0]]+=a[1]; Cm[r,b2[0]]+=b2[1]; bv[r]=rhs
        xu=-Hinv@g[vi]
        s=(Cm@xu-bv)/np.sqrt(np.einsum("ij,jk,ik->i",Cm,Hinv,Cm))
        # most violated per slot (6 rows per slot in this model: 4 pyramid, fz>=0, fz<=fmax)
        V0=set()
        for c in range(len(stance)):
            blk=s[6*c:6*c+6]; t=int(np.argmin(blk))
            if blk[t]<-1e-9: V0.add(6*c+t)
        allv+= int((s<-1e-9).sum())
        Wf=set(Wf)
        tp+=len(V0&Wf); fp+=len(V0-Wf); fn+=len(Wf-V0); its.append(it); v0s.append(len(V0))
    print(name,"robots",len(idx),"mean iters %.1f"%np.mean(its),"mean |V0| %.1f"%np.mean(v0s),"all violated %.1f"%(allv/len(idx)),
          "precision %.2f recall %.2f"%(tp/max(tp+fp,1), tp/max(tp+fn,1)))
b=W.make_config(1,batch=1024); 
# take the hard tail: robots sampled uniformly + ones with many iterations
analyze(b, range(0,1024,8), "cfg1")
b=W.make_config(2,batch=512); analyze(b, range(0,512,6), "cfg2 mixed")
b=W.make_standing(48,10); analyze(b, range(48), "standing braking")
b=W.make_config(4,batch=256); analyze(b, range(0,256,4), "cfg4 stairs")
