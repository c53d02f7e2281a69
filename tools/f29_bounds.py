# Bound propagation (units of p) for the 9x29-bit lazy field used by ec29.cuh
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
rho = Q / 2**261
LIM = 16.0
from math import ceil
class F1:  # Fq: single-product Montgomery
    @staticmethod
    def mul(A,B):
        assert A < LIM and B < LIM, (A,B)
        return A*B*rho + 1
    @staticmethod
    def sqr(A): return F1.mul(A, A)
    @staticmethod
    def ksub(B): return ceil(B)
class F2:  # Fq2: fused two-product reductions; neg inside uses K=ceil(bound)
    @staticmethod
    def mul(A,B):
        assert A < LIM and B < LIM, (A,B)
        K = 8
        assert K < LIM
        c0 = (A*B + K*B)*rho + 1
        c1 = 2*A*B*rho + 1
        return max(c0,c1)
    @staticmethod
    def sqr(A):
        # c0 = mont(a0^2 + (8p - a1)*a1), c1 = mont(a0 * 2a1): same bounds as the general product
        assert 2*A < LIM
        return F2.mul(A, A)
    @staticmethod
    def ksub(B): return ceil(B)
def sub(F, A, B, K=None):
    k = F.ksub(B) if K is None else K
    assert k >= B, (k,B)
    return A + k, k
def run(F, name):
    # fixed point iteration on acc bounds
    BX, BY, BZ = 2.0, 2.0, 1.0
    for it in range(30):
        ks = {}
        qx = qy = 2.0
        # ---- madd
        U2 = F.mul(qx, BZ); S2 = F.mul(qy, BZ)
        P, ks['madd_P'] = sub(F, U2, BX); R, ks['madd_R'] = sub(F, S2, BY)
        PP = F.sqr(P); PPP = F.mul(P,PP); Qv = F.mul(BX,PP); RR = F.sqr(R)
        s = PPP + 2*Qv
        X3, ks['X3'] = sub(F, RR, s)
        d, ks['QmX3'] = sub(F, Qv, X3)
        m1 = F.mul(R, d); m2 = F.mul(BY, PPP)
        Y3, ks['Y3'] = sub(F, m1, m2)
        ZZ3 = F.mul(BZ, PP); ZZZ3 = F.mul(BZ, PPP)
        # ---- full add (both inputs with acc bounds)
        U1 = F.mul(BX,BZ); S1 = F.mul(BY,BZ)
        Pa, ks['add_P'] = sub(F, U1, U1); Ra, ks['add_R'] = sub(F, S1, S1)
        PPa = F.sqr(Pa); PPPa = F.mul(Pa,PPa); Qa = F.mul(U1,PPa); RRa = F.sqr(Ra)
        sa = PPPa + 2*Qa
        X3a, ks['aX3'] = sub(F, RRa, sa)
        da, ks['aQmX3'] = sub(F, Qa, X3a)
        Y3a, ks['aY3'] = sub(F, F.mul(Ra,da), F.mul(S1,PPPa))
        ZZa = F.mul(F.mul(BZ,BZ),PPa); ZZZa = F.mul(F.mul(BZ,BZ),PPPa)
        # ---- dbl
        U = 2*BY; V = F.sqr(U); W = F.mul(U,V); S = F.mul(BX,V); X2 = F.sqr(BX); M = 3*X2
        X3d, ks['dX3'] = sub(F, F.sqr(M), 2*S)
        dd, ks['dSmX3'] = sub(F, S, X3d)
        Y3d, ks['dY3'] = sub(F, F.mul(M,dd), F.mul(W,BY))
        ZZd = F.mul(V,BZ); ZZZd = F.mul(W,BZ)
        nBX = max(X3, X3a, X3d, 2.0); nBY = max(Y3, Y3a, Y3d, 2.0); nBZ = max(ZZ3,ZZZ3,ZZa,ZZZa,ZZd,ZZZd,1.0)
        if abs(nBX-BX)<1e-9 and abs(nBY-BY)<1e-9 and abs(nBZ-BZ)<1e-9: break
        BX,BY,BZ = max(BX,nBX),max(BY,nBY),max(BZ,nBZ)
    print(name, 'BX=%.2f BY=%.2f BZ=%.2f'%(BX,BY,BZ), 'iters',it, ks)
    print('  max mul input seen ok; U=2BY=%.2f M=%.2f'%(2*BY, M))
run(F1,'G1'); run(F2,'G2')
