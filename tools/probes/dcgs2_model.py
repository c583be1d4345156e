#!/usr/bin/env python3
"""numpy model of GMRES with delayed re-orthogonalisation as gmres_solve (plfx.hip) implements it, beside classical Gram-Schmidt twice:
same residual history, orthogonality of the basis to 1e-14.   python tools/probes/dcgs2_model.py"""
import numpy as np
rng=np.random.default_rng(0)
n=3000
# ill-conditioned indefinite symmetric matrix with clustered small eigenvalues of both signs, "preconditioned" ~ identity + low rank + small
Q,_=np.linalg.qr(rng.normal(size=(n,n)))
ev=np.concatenate([1+0.3*rng.normal(size=n-300), rng.uniform(-0.02,0.02,size=300)])
A=(Q*ev)@Q.T
A=A+1e-3*rng.normal(size=(n,n))/np.sqrt(n)   # nonsymmetric perturbation
b=rng.normal(size=n)
def gmres_cgs2(A,b,M,tol):
    n=len(b); V=np.zeros((M+1,n)); H=np.zeros((M+1,M)); beta=np.linalg.norm(b); V[0]=b/beta
    g=np.zeros(M+1); g[0]=beta; cs=np.zeros(M); sn=np.zeros(M); R=np.zeros((M+1,M)); res=[]
    for j in range(M):
        w=A@V[j]
        h=V[:j+1]@w; w=w-V[:j+1].T@h
        h2=V[:j+1]@w; w=w-V[:j+1].T@h2; h+=h2
        hn=np.linalg.norm(w); H[:j+1,j]=h; H[j+1,j]=hn
        col=H[:j+2,j].copy()
        for q in range(j):
            t=cs[q]*col[q]+sn[q]*col[q+1]; col[q+1]=-sn[q]*col[q]+cs[q]*col[q+1]; col[q]=t
        den=np.hypot(col[j],col[j+1]); cs[j]=col[j]/den; sn[j]=col[j+1]/den; col[j]=den; col[j+1]=0
        g[j+1]=-sn[j]*g[j]; g[j]=cs[j]*g[j]; R[:j+2,j]=col; res.append(abs(g[j+1]))
        if abs(g[j+1])<=tol: break
        V[j+1]=w/hn
    k=j+1; y=np.linalg.solve(np.triu(R[:k,:k]),g[:k]); x=V[:k].T@y
    return x,res,V[:k+1]
def gmres_dcgs2(A,b,M,tol):
    n=len(b); V=np.zeros((M+2,n)); Hraw=np.zeros((M+1,M)); beta=np.linalg.norm(b); V[0]=b/beta
    g=np.zeros(M+1); g[0]=beta; cs=np.zeros(M); sn=np.zeros(M); R=np.zeros((M+1,M)); res=[]
    w=A@V[0]; h1p=np.array([V[0]@w]); V[1]=w-V[0]*h1p[0]
    k=0
    for j in range(1,M+1):
        u=V[j].copy(); z=A@u
        s=V[:j]@u; t=V[:j]@z; uu=u@u; uz=u@z
        a2=uu-s@s; alpha=np.sqrt(a2)
        col=np.zeros(j+1); col[:j]=h1p+s; col[j]=alpha; Hraw[:j+1,j-1]=col
        for q in range(j-1):
            tt=cs[q]*col[q]+sn[q]*col[q+1]; col[q+1]=-sn[q]*col[q]+cs[q]*col[q+1]; col[q]=tt
        den=np.hypot(col[j-1],col[j]); cs[j-1]=col[j-1]/den; sn[j-1]=col[j]/den; col[j-1]=den; col[j]=0
        g[j]=-sn[j-1]*g[j-1]; g[j-1]=cs[j-1]*g[j-1]; R[:j+1,j-1]=col; res.append(abs(g[j])); k=j
        if abs(g[j])<=tol or j==M: break
        ia=1/alpha; rho=alpha*s[j-1]; d=((uz-s@t)*ia-rho)*ia; e=(rho*ia+d)*ia
        r=Hraw[:j,:j]@s
        h1p=np.concatenate([(t-r)*ia,[d]])
        f=t*ia-e*s
        V[j]=(u-V[:j].T@s)*ia
        V[j+1]=z*ia-e*u-V[:j].T@f
    y=np.linalg.solve(np.triu(R[:k,:k]),g[:k]); x=V[:k].T@y
    return x,res,V[:k]
for M in (400,):
    x1,r1,V1=gmres_cgs2(A,b,M,1e-10*np.linalg.norm(b))
    x2,r2,V2=gmres_dcgs2(A,b,M,1e-10*np.linalg.norm(b))
    print('cgs2 its',len(r1),'true res',np.linalg.norm(b-A@x1)/np.linalg.norm(b),'orth',np.linalg.norm(V1@V1.T-np.eye(len(V1))))
    print('dcgs2 its',len(r2),'true res',np.linalg.norm(b-A@x2)/np.linalg.norm(b),'orth',np.linalg.norm(V2@V2.T-np.eye(len(V2))))
    print([ '%.1e'%v for v in r1[::40]]); print(['%.1e'%v for v in r2[::40]])
