"""CPU prototype (numpy) of block starts for the dual active set: how many serial steps remain after adding candidate sets
at once (all violated / most violated per foot-step; PDAS-style refactorisation; add-only rounds).  DESIGN.md 5e."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from quadruped_ctrl_amd import workloads as W
from oracle import kron_model as K

def gi(Hinv, g, Cm, bv, scale, x, Wl, lam, tol=1e-9):
    """GI from a given state (x optimal on W, lam>=0). returns x, W, iterations"""
    Wl=list(Wl); lam=list(lam); it=0
    M = Hinv @ Cm[Wl].T if Wl else np.zeros((g.size,0))
    while it<2000:
        s = Cm@x-bv; sn=s/scale; 
        if Wl: sn[Wl]=0
        p=int(np.argmin(sn))
        if sn[p] >= -tol*max(1.0,np.abs(x).max()): break
        lp=0.0
        while True:
            it+=1
            hc=Hinv@Cm[p]
            if Wl:
                S=Cm[Wl]@M; d=M.T@Cm[p]; r=np.linalg.solve(S,d); z=hc-M@r
            else:
                r=np.zeros(0); z=hc
            delta=Cm[p]@z
            dep = delta <= 1e-12*(Cm[p]@hc)
            t2=np.inf if dep else -(Cm[p]@x-bv[p])/delta
            t1,l=np.inf,-1
            for k in range(len(Wl)):
                if r[k]>0 and lam[k]/r[k]<t1: t1,l=lam[k]/r[k],k
            t=min(t1,t2)
            if not np.isfinite(t): return x,Wl,-it
            if not dep: x=x+t*z
            lam=[lam[k]-t*r[k] for k in range(len(Wl))]; lp+=t
            if t==t2:
                Wl.append(p); lam.append(lp); M=np.concatenate([M,hc[:,None]],1); break
            Wl.pop(l); lam.pop(l); M=np.delete(M,l,1)
    return x,Wl,it

def block_rounds(Hinv,g,Cm,bv,scale,R,pick):
    n=g.size; xu=-Hinv@g; x=xu.copy(); Wl=[]; lam=[]; nblock=0; ndrop=0; nadded=0
    for r in range(R):
        s=(Cm@x-bv)/scale
        if Wl: s[Wl]=0
        V=pick(s)
        V=[v for v in V if v not in Wl]
        if not V: break
        # add block, skipping dependent ones (greedy rank check)
        for v in V:
            T=Wl+[v]
            S=Cm[T]@Hinv@Cm[T].T
            if np.linalg.eigvalsh(S).min() > 1e-10*np.abs(S).max(): Wl=T; nadded+=1
        nblock+=1
        while True:
            S=Cm[Wl]@Hinv@Cm[Wl].T
            lamv=np.linalg.solve(S, -(Cm[Wl]@xu-bv[Wl]))
            if (lamv>=-1e-12).all(): break
            k=int(np.argmin(lamv)); Wl.pop(k); ndrop+=1
        x=xu+Hinv@Cm[Wl].T@lamv; lam=list(lamv)
    return x,Wl,lam,nblock,ndrop,nadded

def study(b, idx, name):
    out=[]
    for i in idx:
        h=b["horizon"]; H,g=K.assemble(b,i)
        stance,rows=K.stance_constraints(b["gait"][i],h,b["mu"],b["f_max"])
        vi=np.array([3*k+a for k in stance for a in range(3)],int)
        Hinv=K.sweep_inverse(H[np.ix_(vi,vi)]); gg=g[vi]
        n=len(vi); m=len(rows)
        Cm=np.zeros((m,n)); bv=np.zeros(m)
        for r,(a,b2,rhs) in enumerate(rows):
            Cm[r,a[0]]+=a[1]; Cm[r,b2[0]]+=b2[1]; bv[r]=rhs
        scale=np.sqrt(np.einsum("ij,jk,ik->i",Cm,Hinv,Cm))
        x0,W0,it0=gi(Hinv,gg,Cm,bv,scale,-Hinv@gg,[],[])
        res=[it0,len(W0)]
        for R,pick,nm in ((1,lambda s:[int(k) for k in np.nonzero(s<-1e-9)[0]],"all1"),(2,lambda s:[int(k) for k in np.nonzero(s<-1e-9)[0]],"all2"),(3,lambda s:[int(k) for k in np.nonzero(s<-1e-9)[0]],"all3"),
                          (3,lambda s:[6*c+int(np.argmin(s[6*c:6*c+6])) for c in range(len(s)//6) if s[6*c:6*c+6].min()<-1e-9],"slot3")):
            x,Wl,lam,nb,nd,na=block_rounds(Hinv,gg,Cm,bv,scale,R,pick)
            x2,W2,it2=gi(Hinv,gg,Cm,bv,scale,x,Wl,lam)
            assert np.abs(x2-x0).max()<1e-6*max(1,np.abs(x0).max()), (np.abs(x2-x0).max())
            res+= [nb,na,nd,it2]
        out.append(res)
    o=np.array(out,float)
    print(name,"robots",len(idx),"cold iters %.1f |W| %.1f"%(o[:,0].mean(),o[:,1].mean()))
    for k,nm in enumerate(("all violated, 1 round","all violated, 2 rounds","all violated, 3 rounds","most violated per slot, 3 rounds")):
        c=o[:,2+4*k:6+4*k]
        print("   %-34s blocks %.1f added %.1f drops %.1f then GI iters %.1f (max %d)"%(nm,c[:,0].mean(),c[:,1].mean(),c[:,2].mean(),c[:,3].mean(),c[:,3].max()))
study(W.make_standing(24,10), range(24), "standing braking h10")
study(W.make_standing(12,14), range(12), "standing braking h14")
study(W.make_config(4,batch=256), range(0,256,8), "cfg4 stairs")
b=W.make_config(1,batch=1024); study(b, range(0,1024,32), "cfg1")

def pdas(Hinv,g,Cm,bv,scale,R,per_slot=True,W0=None):
    n=g.size; xu=-Hinv@g; x=xu.copy(); Wl=list(W0) if W0 else []; lamv=np.zeros(0); nfact=0; ks=[]
    first=True
    for r in range(R):
        s=(Cm@x-bv)/scale
        if Wl: s[Wl]=0
        if per_slot:
            V=[6*c+int(np.argmin(s[6*c:6*c+6])) for c in range(len(s)//6) if s[6*c:6*c+6].min()<-1e-9]
        else:
            V=[int(k) for k in np.nonzero(s<-1e-9)[0]]
        keep=[w for w,l in zip(Wl,lamv) if l>0] if len(lamv) else list(Wl)
        T=[]
        for v in keep+[v for v in V if v not in keep]:
            T2=T+[v]; S=Cm[T2]@Hinv@Cm[T2].T
            if np.linalg.eigvalsh(S).min() > 1e-10*np.abs(S).max(): T=T2
        if set(T)==set(Wl) and not first: break
        first=False
        Wl=T; nfact+=1; ks.append(len(Wl))
        if not Wl: break
        S=Cm[Wl]@Hinv@Cm[Wl].T
        lamv=np.linalg.solve(S, -(Cm[Wl]@xu-bv[Wl]))
        x=xu+Hinv@Cm[Wl].T@lamv
    # final: sequential fixneg drops
    nd=0
    while len(Wl) and (lamv<-1e-12).any():
        k=int(np.argmin(lamv)); Wl.pop(k); nd+=1
        if Wl:
            S=Cm[Wl]@Hinv@Cm[Wl].T; lamv=np.linalg.solve(S, -(Cm[Wl]@xu-bv[Wl])); x=xu+Hinv@Cm[Wl].T@lamv
        else:
            lamv=np.zeros(0); x=xu.copy()
    return x,Wl,list(lamv),nfact,nd,ks

def study2(b, idx, name, warm=False):
    out=[]
    for i in idx:
        h=b["horizon"]; H,g=K.assemble(b,i)
        stance,rows=K.stance_constraints(b["gait"][i],h,b["mu"],b["f_max"])
        vi=np.array([3*k+a for k in stance for a in range(3)],int)
        Hinv=K.sweep_inverse(H[np.ix_(vi,vi)]); gg=g[vi]
        n=len(vi); m=len(rows)
        Cm=np.zeros((m,n)); bv=np.zeros(m)
        for r,(a,b2,rhs) in enumerate(rows):
            Cm[r,a[0]]+=a[1]; Cm[r,b2[0]]+=b2[1]; bv[r]=rhs
        scale=np.sqrt(np.einsum("ij,jk,ik->i",Cm,Hinv,Cm))
        x0,W0,it0=gi(Hinv,gg,Cm,bv,scale,-Hinv@gg,[],[])
        res=[it0]
        for R in (1,2,3,4):
            x,Wl,lam,nf,nd,ks=pdas(Hinv,gg,Cm,bv,scale,R)
            x2,W2,it2=gi(Hinv,gg,Cm,bv,scale,x,Wl,lam)
            assert np.abs(x2-x0).max()<1e-6*max(1,np.abs(x0).max())
            res+=[nf,nd,it2,sum(k*k for k in ks)]
        out.append(res)
    o=np.array(out,float)
    print(name,"cold iters %.1f (max %d)"%(o[:,0].mean(),o[:,0].max()))
    for k,R in enumerate((1,2,3,4)):
        c=o[:,1+4*k:5+4*k]
        print("   PDAS per-slot R=%d: factorisations %.1f, final drops %.1f (max %d), GI iters %.1f (max %d), sum k^2 %.0f"%(R,c[:,0].mean(),c[:,1].mean(),c[:,1].max(),c[:,2].mean(),c[:,2].max(),c[:,3].mean()))
print("---- PDAS")
study2(W.make_standing(24,10), range(24), "standing braking h10")
study2(W.make_standing(12,14), range(12), "standing braking h14")
study2(W.make_standing(12,10,calm=True), range(12), "standing calm h10")
study2(W.make_config(4,batch=256), range(0,256,8), "cfg4 stairs")

def addonly(Hinv,g,Cm,bv,scale,R):
    n=g.size; xu=-Hinv@g; x=xu.copy(); Wl=[]; lamv=np.zeros(0); ks=[]
    for r in range(R):
        s=(Cm@x-bv)/scale
        if Wl: s[Wl]=0
        V=[6*c+int(np.argmin(s[6*c:6*c+6])) for c in range(len(s)//6) if s[6*c:6*c+6].min()<-1e-9]
        V=[v for v in V if v not in Wl]
        if not V: break
        for v in V:
            T=Wl+[v]; S=Cm[T]@Hinv@Cm[T].T
            if np.linalg.eigvalsh(S).min() > 1e-10*np.abs(S).max(): Wl=T
        ks.append(len(Wl))
        S=Cm[Wl]@Hinv@Cm[Wl].T
        lamv=np.linalg.solve(S, -(Cm[Wl]@xu-bv[Wl])); x=xu+Hinv@Cm[Wl].T@lamv
    nd=0
    while len(Wl) and (lamv<-1e-12).any():
        k=int(np.argmin(lamv)); Wl.pop(k); nd+=1
        if Wl:
            S=Cm[Wl]@Hinv@Cm[Wl].T; lamv=np.linalg.solve(S, -(Cm[Wl]@xu-bv[Wl])); x=xu+Hinv@Cm[Wl].T@lamv
        else: lamv=np.zeros(0); x=xu.copy()
    return x,Wl,list(lamv),len(ks),nd,ks

def study3(b, idx, name):
    out=[]
    for i in idx:
        h=b["horizon"]; H,g=K.assemble(b,i)
        stance,rows=K.stance_constraints(b["gait"][i],h,b["mu"],b["f_max"])
        vi=np.array([3*k+a for k in stance for a in range(3)],int)
        Hinv=K.sweep_inverse(H[np.ix_(vi,vi)]); gg=g[vi]
        n=len(vi); m=len(rows)
        Cm=np.zeros((m,n)); bv=np.zeros(m)
        for r,(a,b2,rhs) in enumerate(rows):
            Cm[r,a[0]]+=a[1]; Cm[r,b2[0]]+=b2[1]; bv[r]=rhs
        scale=np.sqrt(np.einsum("ij,jk,ik->i",Cm,Hinv,Cm))
        x0,W0,it0=gi(Hinv,gg,Cm,bv,scale,-Hinv@gg,[],[])
        res=[it0]
        for R in (1,2,3,4,5):
            x,Wl,lam,nf,nd,ks=addonly(Hinv,gg,Cm,bv,scale,R)
            x2,W2,it2=gi(Hinv,gg,Cm,bv,scale,x,Wl,lam)
            assert np.abs(x2-x0).max()<1e-6*max(1,np.abs(x0).max())
            res+=[nf,nd,it2,(ks[-1] if ks else 0)]
        out.append(res)
    o=np.array(out,float)
    print(name,"cold iters %.1f (max %d)"%(o[:,0].mean(),o[:,0].max()))
    for k,R in enumerate((1,2,3,4,5)):
        c=o[:,1+4*k:5+4*k]
        print("   add-only per-slot R=%d: rounds %.1f, k final %.1f (max %d), final drops %.1f (max %d), GI iters %.1f (max %d)"%(R,c[:,0].mean(),c[:,3].mean(),c[:,3].max(),c[:,1].mean(),c[:,1].max(),c[:,2].mean(),c[:,2].max()))
print("---- add-only")
study3(W.make_standing(24,10), range(24), "standing braking h10")
study3(W.make_standing(12,14), range(12), "standing braking h14")
study3(W.make_standing(12,10,calm=True), range(12), "standing calm h10")
study3(W.make_config(4,batch=256), range(0,256,8), "cfg4 stairs")
