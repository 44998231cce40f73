"""Experiment (round 4), not a test: is the compact-WY factor T harmless at full size (all n - 2 reflectors of a tridiagonalisation in ONE block)?
Yes: max |T| < 2, cond(T^-1) ~ 45 at n = 1037, X = Z - V T V^T Z equals the reflector-by-reflector product to 1e-15 (random and graded matrices).
The GPU variant built on it was slower."""
import numpy as np, scipy.linalg as sl
from scipy.linalg import lapack
rng=np.random.default_rng(1)
def test(n, graded):
    A=rng.standard_normal((n,n))
    if graded:
        # density-matrix like: Psi Psi^T with decaying singular values
        U,_=np.linalg.qr(rng.standard_normal((n,n))); s=np.exp(-np.arange(n)*30.0/n)
        A=(U*s**2)@U.T
    A=(A+A.T)/2
    c,d,e,tau,info=lapack.dsytrd(A,lower=1)
    # V: column j has v_j with v[j+1]=1, v[j+2:]=c[j+2:,j]
    V=np.zeros((n,n-1))
    for j in range(n-1):
        V[j+1,j]=1.0; V[j+2:,j]=c[j+2:,j]
    tau=tau[:n-1]
    G=V.T@V
    with np.errstate(divide='ignore'):
        dinv=np.where(tau!=0,1.0/np.where(tau!=0,tau,1),1.0)
    Tinv=np.triu(G,1)+np.diag(dinv)
    # zero reflectors: tau=0 -> that row/col of T is zero; emulate by zeroing V column
    Vz=V*(tau!=0)
    Gz=Vz.T@Vz
    Tinv=np.triu(Gz,1)+np.diag(dinv)
    # recursive doubling inverse in blocks of 64
    def inv_ut(M):
        k=M.shape[0]
        if k<=64: return sl.solve_triangular(M,np.eye(k),lower=False)
        h=(k//2+63)//64*64
        A1=inv_ut(M[:h,:h]); C1=inv_ut(M[h:,h:])
        out=np.zeros_like(M); out[:h,:h]=A1; out[h:,h:]=C1; out[:h,h:]=-A1@M[:h,h:]@C1
        return out
    T=inv_ut(Tinv)
    T=T*(tau!=0)[:,None]*(tau!=0)[None,:]
    Z=np.linalg.qr(rng.standard_normal((n,n)))[0]
    X=Z-Vz@(T@(Vz.T@Z))
    Xref=Z.copy()
    for j in range(n-2,-1,-1):        # X = H_0 (H_1 ( ... H_{n-2} Z))
        v=V[:,j]; Xref-= tau[j]*np.outer(v, v@Xref)
    return np.abs(X-Xref).max(), np.abs(X.T@X-np.eye(n)).max(), np.abs(T).max(), np.linalg.cond(Tinv)
for n in (300,700,1037):
    for g in (False,True):
        print(n,'graded' if g else 'random','max|X-Xref| %.2e  |X^T X - I| %.2e  max|T| %.2e cond(Tinv) %.2e'%test(n,g))
