"""CPU check of the constants and flow graphs the HIP kernels hard-code (no GPU needed).

The kernels' Y transform is a SCALED 8-point DCT (csrc/common.hiph: dct8s; orthonormal X[k] = DS[k] * x[k]) whose
scales ride on the texture-feature sums.  A wrong constant there does not crash anything -- it biases the texture
mask by a few 1e-6 (this is how a mistyped cos(pi/16)/cos(3pi/16) was caught) -- so the numbers are re-derived here
from the source text and the flow graph is replayed in float32 against the oracle's float64 DCT."""
import math
import os
import re

import numpy as np

import offmark_oracle as orc
from conftest import PKG

F = np.float32
SRC = open(os.path.join(PKG, "csrc", "common.hiph")).read()


def const(name):
    m = re.search(r"\b" + name + r"\s*=\s*(-?[0-9.]+(?:e-?[0-9]+)?)f", SRC)
    assert m, name
    return float(m.group(1))


def fma(a, b, c):
    return (np.asarray(a, F).astype(np.float64) * np.float64(F(b)) + np.asarray(c, F).astype(np.float64)).astype(F)


def dct8s(x, T1, T2, T3, R13):
    """csrc/common.hiph dct8s, operation for operation, float32."""
    x = [x[..., i] for i in range(8)]
    a0, a1, a2, a3 = x[0] + x[7], x[1] + x[6], x[2] + x[5], x[3] + x[4]
    b0, b1, b2, b3 = x[0] - x[7], x[1] - x[6], x[2] - x[5], x[3] - x[4]
    c0, c1, c2, c3 = a0 + a3, a1 + a2, a1 - a2, a0 - a3
    y = [None] * 8
    y[0], y[4] = c0 + c1, c0 - c1
    y[2], y[6] = fma(c2, T2, c3), fma(c3, T2, -c2)
    p4, p7 = fma(b0, T3, b3), fma(b3, -T3, b0)
    p5, p6 = fma(b1, T1, b2), fma(b2, -T1, b1)
    u4, u6 = fma(p6, R13, p4), fma(p6, -R13, p4)
    u7, u5 = fma(p5, R13, p7), fma(p5, -R13, p7)
    y[1], y[7], y[3], y[5] = u7 + u4, u7 - u4, u5, u6
    return np.stack(y, -1).astype(F)


def test_constants_are_what_their_comments_say():
    c = lambda k: math.cos(k * math.pi / 16)            # noqa: E731
    want = dict(S0=math.sqrt(1 / 8), S1=0.5 * c(3) / math.sqrt(2), S2=0.5 * c(2), S3=0.5 * c(3),
                T1=math.tan(math.pi / 16), T2=math.tan(2 * math.pi / 16), T3=math.tan(3 * math.pi / 16), R13=c(1) / c(3),
                H1=0.5 * c(1), H2=0.5 * c(2), H3=0.5 * c(3), H4=0.5 * c(4), H5=0.5 * c(5), H6=0.5 * c(6), H7=0.5 * c(7),
                KY0=0.114, KY1=0.587, KY2=0.299, KU=0.492, KV=0.877, KDELTA=0.5, KI_B=2.032, KI_GU=-0.395, KI_GV=-0.581)
    for name, v in want.items():
        assert F(const(name)) == F(v), (name, const(name), v)      # the float32 the compiler sees is the nearest one


def test_scaled_dct_flow_graph_reproduces_the_orthonormal_dct():
    T1, T2, T3, R13 = (const(n) for n in ("T1", "T2", "T3", "R13"))
    DS = np.array([const(n) for n in ("S0", "S1", "S2", "S3", "S0", "S3", "S2", "S1")], F)
    rng = np.random.default_rng(1)
    blocks = rng.integers(0, 256, (4000, 8, 8)).astype(F)                  # Y-like data, worst case for cancellation
    blocks[:500] = rng.integers(100, 110, (500, 8, 8))                      # smooth blocks
    rows = dct8s(blocks, T1, T2, T3, R13)
    A = np.swapaxes(dct8s(np.swapaxes(rows, -1, -2), T1, T2, T3, R13), -1, -2)
    got = A.astype(np.float64) * DS[:, None].astype(np.float64) * DS[None, :].astype(np.float64)
    ref = orc.dct8x8(blocks).astype(np.float64)
    assert np.abs(got - ref).max() <= 1e-3                                   # the tests' Y-DC / coefficient tolerance
    assert np.abs(got - ref).max() <= 4e-7 * np.abs(ref).max()              # a few float32 ulp of the largest term
    # the texture-mask features from the scaled coefficients (kernel's grouping) against the oracle's
    ab = np.abs(A)
    S0, S1, S2, S3 = DS[0], DS[1], DS[2], DS[3]
    rs = fma(ab[..., 3, :] + ab[..., 5, :], S3, fma(ab[..., 2, :] + ab[..., 6, :], S2, fma(ab[..., 1, :] + ab[..., 7, :], S1,
             (ab[..., 0, :] + ab[..., 4, :]) * S0)))
    tot = fma(rs[..., 3] + rs[..., 5], S3, fma(rs[..., 2] + rs[..., 6], S2, fma(rs[..., 1] + rs[..., 7], S1, (rs[..., 0] + rs[..., 4]) * S0)))
    a00 = ab[..., 0, 0] * (S0 * S0)
    dcl = fma(ab[..., 2, 0], S2 * S0, fma(ab[..., 1, 1], S1 * S1, fma(ab[..., 1, 0] + ab[..., 0, 1], S1 * S0, fma(ab[..., 0, 2], S0 * S2, a00))))
    e = fma(ab[..., 3, 3], S3 * S3, fma(ab[..., 2, 2], S2 * S2, fma(ab[..., 2, 1] + ab[..., 1, 2], S2 * S1,
            fma(ab[..., 6, 0] + ab[..., 0, 6], S2 * S0, fma(ab[..., 5, 0] + ab[..., 0, 5], S3 * S0, fma(ab[..., 4, 0] + ab[..., 0, 4], S0 * S0,
                (ab[..., 3, 0] + ab[..., 0, 3]) * (S3 * S0)))))))
    dcl_o, eh_o, e_o = orc.texture_features(np.abs(orc.dct8x8(blocks)))
    # errors are a few float32 ulp of the block's total |coefficient| mass (eh = tot - dcl cancels on smooth blocks)
    rel = lambda x, y: np.abs(x.astype(np.float64) - y) / np.maximum(tot.astype(np.float64), 1.0)      # noqa: E731
    assert rel(tot - dcl, eh_o).max() <= 1e-6 and rel(dcl, dcl_o).max() <= 1e-6 and rel(e, e_o).max() <= 1e-6
    # no systematic bias (a mistyped constant shows up here first): mean signed relative error of eh
    assert abs(((tot - dcl).astype(np.float64) - eh_o).mean() / eh_o.mean()) <= 1e-7


def test_texture_thresholds_are_the_float32_images_of_the_float64_comparisons():
    """texture_mask compares float32 ratios with python floats, i.e. in float64 under the reference's numpy 1.23
    (dct_encoder.py:92-101).  The kernel compares in float32 against TA1/TB1/TA2/TB2 = the smallest float32 not below
    2.3 / 1.6 / 1.4 / 1.1: the same decision for every float32 input, nan and inf included."""
    for name, t in (("TA1", 2.3), ("TB1", 1.6), ("TA2", 1.4), ("TB2", 1.1)):
        m = re.search(r"\b" + name + r"\s*=\s*(0x[0-9a-fA-F.]+p[+-]?[0-9]+)f", SRC)
        assert m, name
        T = F(float.fromhex(m.group(1)))
        assert float(T) == float.fromhex(m.group(1))                       # the literal is a float32
        assert float(T) >= t > float(np.nextafter(T, F(-np.inf)))           # smallest float32 not below t
        x = T
        for _ in range(2000):                                               # 2000 float32 neighbours on each side
            x = np.nextafter(x, F(-np.inf))
        for _ in range(4000):
            assert (np.float64(x) >= t) == bool(x >= T)
            x = np.nextafter(x, F(np.inf))
        for special in (F(np.inf), F(-np.inf), F(np.nan), F(0), F(1e30)):
            with np.errstate(invalid="ignore"):
                assert (np.float64(special) >= t) == bool(special >= T)


# ---- DwtDctSvd: top singular value by Householder tridiagonalisation + Laguerre (svd_kernels.hiph) ----------------
def _svd_gram(B):
    G = np.zeros(B.shape[:1] + (4, 4), F)
    for i in range(4):
        for j in range(i, 4):
            acc = B[:, 0, i] * B[:, 0, j]
            for k in range(1, 4):
                acc = fma(B[:, k, i], B[:, k, j], acc)
            G[:, i, j] = acc
    return G


def _svd_top_eigenvalue(G, tol, cap):
    """gram_top_eigenvalue() line by line in float32 (fma = exact product, one rounding)."""
    tiny = F(1e-30)
    x0, x1, x2 = G[:, 0, 1], G[:, 0, 2], G[:, 0, 3]
    tail = fma(x2, x2, x1 * x1)
    n2 = fma(x0, x0, tail)
    nb1 = np.sqrt(n2)
    v0 = x0 + np.copysign(nb1, x0)
    vtv = fma(v0, v0, tail)
    with np.errstate(all="ignore"):
        beta = np.where(tail > tiny, F(2) / vtv, F(0)).astype(F)
    s00, s01, s02, s11, s12, s22 = (G[:, 1, 1].copy(), G[:, 1, 2].copy(), G[:, 1, 3].copy(), G[:, 2, 2].copy(), G[:, 2, 3].copy(),
                                    G[:, 3, 3].copy())
    p0 = beta * fma(s02, x2, fma(s01, x1, s00 * v0))
    p1 = beta * fma(s12, x2, fma(s11, x1, s01 * v0))
    p2 = beta * fma(s22, x2, fma(s12, x1, s02 * v0))
    K = F(0.5) * beta * fma(x2, p2, fma(x1, p1, v0 * p0))
    q0, q1, q2 = fma(-K, v0, p0), fma(-K, x1, p1), fma(-K, x2, p2)
    s00 = fma(F(-2) * v0, q0, s00)
    s01 = fma(-q0, x1, fma(-v0, q1, s01))
    s02 = fma(-q0, x2, fma(-v0, q2, s02))
    s11 = fma(F(-2) * x1, q1, s11)
    s12 = fma(-q1, x2, fma(-x1, q2, s12))
    s22 = fma(F(-2) * x2, q2, s22)
    tail2 = s02 * s02
    m2 = fma(s01, s01, tail2)
    nb2 = np.sqrt(m2)
    w0 = s01 + np.copysign(nb2, s01)
    wtw = fma(w0, w0, tail2)
    with np.errstate(all="ignore"):
        beta2 = np.where(tail2 > tiny, F(2) / wtw, F(0)).astype(F)
    p0 = beta2 * fma(s12, s02, s11 * w0)
    p1 = beta2 * fma(s22, s02, s12 * w0)
    K = F(0.5) * beta2 * fma(s02, p1, w0 * p0)
    q0, q1 = fma(-K, w0, p0), fma(-K, s02, p1)
    s11 = fma(F(-2) * w0, q0, s11)
    s12 = fma(-q0, s02, fma(-w0, q1, s12))
    s22 = fma(F(-2) * s02, q1, s22)
    a0, a1, a2, a3 = G[:, 0, 0], s00, s11, s22
    b1q, b2q, b3q, nb3 = n2, m2, s12 * s12, np.abs(s12)
    gersh = np.maximum(np.maximum(a0 + nb1, a1 + nb1 + nb2), np.maximum(a2 + nb2 + nb3, a3 + nb3))
    lam = np.minimum(gersh, (a0 + a1) + (a2 + a3)).astype(F)
    active = np.ones(len(lam), bool)
    iters = np.zeros(len(lam), int)
    for _ in range(cap):
        d0, d1, d2, d3 = a0 - lam, a1 - lam, a2 - lam, a3 - lam
        p1 = d0
        p2, e2 = fma(d1, p1, -b1q), -d1 - p1
        p3, e3, f3 = fma(d2, p2, -b2q * p1), fma(d2, e2, b2q - p2), fma(d2, F(2), F(-2) * e2)
        p4, e4, f4 = fma(d3, p3, -b3q * p2), fma(d3, e3, fma(-b3q, e2, -p3)), fma(d3, f3, fma(F(-2), e3, F(-2) * b3q))
        with np.errstate(all="ignore"):
            rp = (F(1) / p4).astype(F)
            g = e4 * rp
            h = fma(g, g, -f4 * rp)
            t = np.maximum(F(3) * fma(F(4), h, -g * g), F(0))
            step = (F(4) / (g + np.copysign(np.sqrt(t), g))).astype(F)
            step = np.where((p4 != 0) & (np.abs(step) < F(3.0e38)), step, F(0)).astype(F)
        lam = np.where(active, lam - step, lam).astype(F)
        iters += active
        active &= np.abs(step) > tol * np.abs(lam)
        if not active.any():
            break
    return np.maximum(lam, F(0)), iters


def test_svd_top_value_solver_float32_replay():
    """The DwtDctSvd kernels take the top singular value of each 4x4 LL block from the Gram matrix by two Householder
    reflections and Laguerre's iteration on the tridiagonal form (svd_kernels.hiph: gram_top_eigenvalue).  Replayed here
    in float32 against float64 LAPACK (np.linalg.svd is what the reference calls, dwt_dct_svd_encoder.py:41) with the
    tolerances and iteration caps the kernel source passes: well separated, repeated, nearly repeated, rank-1 and zero
    spectra, LL-like magnitudes."""
    src = open(os.path.join(PKG, "csrc", "svd_kernels.hiph")).read()
    m = re.search(r"kTolTight = ([0-9.eE+-]+)f, kTolLoose = ([0-9.eE+-]+)f;", src)
    c = re.search(r"kCapTight = (\d+), kCapLoose = (\d+);", src)
    min_loose = float(re.search(r"kLooseReadoutMinScale = ([0-9.]+)f;", src).group(1))
    # embed and the stand-alone read-out use the tight pair; only the verify of a block just marked reads loosely
    assert "gram_top_eigenvalue(G, kTolTight, kCapTight)" in src and "tight ? kTolTight : kTolLoose, tight ? kCapTight : kCapLoose" in src
    assert "svd_read_bit(B, a.scales[1], true)" in src and "a.scales[1] < kLooseReadoutMinScale" in src
    (tol_embed, tol_read), (cap_embed, cap_read) = (float(m.group(1)), float(m.group(2))), (int(c.group(1)), int(c.group(2)))
    rng = np.random.default_rng(5)
    sets = [rng.uniform(-200, 200, (6000, 4, 4)), rng.uniform(0, 510, (6000, 4, 4)), 1 + rng.normal(0, 1, (6000, 4, 4)),
            np.ones((1, 4, 4)) * rng.uniform(0, 300, (2000, 1, 1)) + rng.normal(0, 0.3, (2000, 4, 4))]     # near-flat LL blocks
    adv = [np.zeros((4, 4)), np.ones((4, 4)), np.eye(4) * 7, np.diag([5., 5, 1, 0]), np.diag([3., 3, 3, 3.0000001]),
           np.diag([100., 99.9999, 1e-3, 0]), np.outer([1., 2, 3, 4], [4., 3, 2, 1]), np.diag([0., 0, 0, 9]), np.diag([1e-3, 0, 0, 0])]
    for k in range(400):
        q1, _ = np.linalg.qr(rng.normal(size=(4, 4)))
        q2, _ = np.linalg.qr(rng.normal(size=(4, 4)))
        s = np.sort(rng.uniform(0, 1000, 4))[::-1]
        if k % 4 == 0:
            s[1] = s[0]
        elif k % 4 == 1:
            s[1] = s[0] * (1 - 1e-5)
        elif k % 4 == 2:
            s[1:] = 0
        adv.append(q1 @ np.diag(s) @ q2.T)
    B = np.concatenate(sets + [np.array(adv)]).astype(F)
    ref = np.linalg.svd(B.astype(np.float64), compute_uv=False)[:, 0]
    G = _svd_gram(B)
    big = ref > 1e-3
    lam, iters = _svd_top_eigenvalue(G, F(tol_embed), cap_embed)
    rel = np.abs(np.sqrt(lam.astype(np.float64)) - ref) / np.maximum(ref, 1e-30)
    assert rel[big].max() <= 6e-7, rel[big].max()                          # measured 4.6e-7; embed then refines with |B v0|
    assert np.abs(np.sqrt(lam.astype(np.float64)) - ref)[~big].max() <= 1e-3
    assert iters.max() <= cap_embed and iters.mean() <= 3.0                # cubic convergence except for repeated roots
    lam, iters = _svd_top_eigenvalue(G, F(tol_read), cap_read)
    rel = np.abs(np.sqrt(lam.astype(np.float64)) - ref) / np.maximum(ref, 1e-30)
    # loose read-out (verify of a block just marked): bit = (s0 mod scale) > scale/2 with s0 a quarter step from either
    # threshold, so the margin is scale/4 >= 1 for the scales that read loosely; s0 <= 2040 for u8 frames
    assert rel[big].max() <= 1e-4 and rel[big].max() * 2040 < 0.25 * min_loose, rel[big].max()


# ---- DwtDctSvd with blk = 8: top singular triplet of an 8x8 block (svd8_kernels.hiph) -------------------------------------
def _svd8_gram(B):
    n=8; G=np.zeros(B.shape[:1]+(n,n),F)
    for i in range(n):
        for j in range(i,n):
            acc=B[:,0,i]*B[:,0,j]
            for k in range(1,n): acc=fma(B[:,k,i],B[:,k,j],acc)
            G[:,i,j]=acc; G[:,j,i]=acc
    return G

def _svd8_solve(B, tol=F(2e-7), cap=16, want_vec=True):
    """svd8_top() of csrc/svd8_kernels.hiph step by step in float32 (fma = exact product, one rounding; the kernel's
    Newton-refined reciprocals are exact divisions here): Gram matrix, trace scaling, six Householder reflections,
    Laguerre from above on the degree-8 characteristic polynomial, twisted factorisation for the eigenvector,
    back-transformation, w = B v, s0 = |w|."""
    n=8; N=B.shape[0]
    G=_svd8_gram(B)
    tr=G[:,0,0].copy()
    for i in range(1,n): tr=tr+G[:,i,i]
    with np.errstate(all='ignore'):
        inv=np.where(tr>0, F(1)/tr, F(0)).astype(F)
    A=(G*inv[:,None,None]).astype(F)
    a=np.zeros((N,n),F); b=np.zeros((N,n-1),F)
    V=[]; BETA=[]
    for k in range(n-2):
        m=n-1-k
        x=[A[:,k,k+1+i].copy() for i in range(m)]
        tail=x[1]*x[1]
        for i in range(2,m): tail=fma(x[i],x[i],tail)
        n2=fma(x[0],x[0],tail)
        nrm=np.sqrt(n2)
        v0=x[0]+np.copysign(nrm,x[0])
        vtv=fma(v0,v0,tail)
        with np.errstate(all='ignore'):
            beta=np.where(tail>F(1e-30), F(2)/vtv, F(0)).astype(F)
        v=[v0]+x[1:]
        # trailing submatrix S (m x m) = A[k+1.., k+1..]
        p=[]
        for i in range(m):
            acc=A[:,k+1+i,k+1]*v[0]
            for j in range(1,m): acc=fma(A[:,k+1+i,k+1+j],v[j],acc)
            p.append(beta*acc)
        K=v[0]*p[0]
        for j in range(1,m): K=fma(v[j],p[j],K)
        K=F(0.5)*beta*K
        q=[fma(-K,v[i],p[i]) for i in range(m)]
        for i in range(m):
            for j in range(i,m):
                val=fma(-q[i],v[j],fma(-v[i],q[j],A[:,k+1+i,k+1+j]))
                A[:,k+1+i,k+1+j]=val; A[:,k+1+j,k+1+i]=val
        a[:,k]=A[:,k,k]
        b[:,k]=np.where(tail>F(1e-30), -np.copysign(nrm,x[0]), x[0])
        V.append(v); BETA.append(beta)
    a[:,n-2]=A[:,n-2,n-2]; a[:,n-1]=A[:,n-1,n-1]; b[:,n-2]=A[:,n-2,n-1]
    bq=(b*b).astype(F); nb=np.abs(b)
    # Gershgorin / trace (=1 after scaling) start
    g=a[:,0]+nb[:,0]
    for i in range(1,n-1): g=np.maximum(g,a[:,i]+nb[:,i-1]+nb[:,i])
    g=np.maximum(g,a[:,n-1]+nb[:,n-2])
    lam=np.minimum(g,F(1)).astype(F)
    active=np.ones(N,bool); iters=np.zeros(N,int)
    for _ in range(cap):
        d=[a[:,i]-lam for i in range(n)]
        pm2,em2,fm2=np.ones(N,F),np.zeros(N,F),np.zeros(N,F)     # p0
        pm1,em1,fm1=d[0],-np.ones(N,F),np.zeros(N,F)              # p1
        for k in range(1,n):
            pk=fma(d[k],pm1,-bq[:,k-1]*pm2)
            ek=fma(d[k],em1,fma(-bq[:,k-1],em2,-pm1))
            fk=fma(d[k],fm1,fma(-bq[:,k-1],fm2,F(-2)*em1))
            pm2,em2,fm2,pm1,em1,fm1=pm1,em1,fm1,pk,ek,fk
        with np.errstate(all='ignore'):
            rp=(F(1)/pm1).astype(F)
            gg=em1*rp
            h=fma(gg,gg,-fm1*rp)
            t=np.maximum(F(n-1)*fma(F(n),h,-gg*gg),F(0))
            step=(F(n)/(gg+np.copysign(np.sqrt(t),gg))).astype(F)
            step=np.where((pm1!=0)&(np.abs(step)<F(3e38)),step,F(0)).astype(F)
        lam=np.where(active,lam-step,lam).astype(F)
        iters+=active
        active&=np.abs(step)>tol*np.abs(lam)
        if not active.any(): break
    lam=np.maximum(lam,F(0))
    s0_val=np.sqrt(lam.astype(F)*tr).astype(F)
    if not want_vec: return s0_val,None,iters
    # twisted factorisation
    tiny=F(1e-12)
    def safe(D): return np.where(np.abs(D)<tiny, np.copysign(tiny,D), D).astype(F)
    d=[a[:,i]-lam for i in range(n)]
    Dp=[None]*n; Lp=[None]*(n-1)
    Dp[0]=d[0]
    with np.errstate(all='ignore'):
        for i in range(n-1):
            Lp[i]=(b[:,i]/safe(Dp[i])).astype(F)
            Dp[i+1]=fma(-Lp[i],b[:,i],d[i+1])
        Dm=[None]*n; Um=[None]*(n-1)
        Dm[n-1]=d[n-1]
        for i in range(n-2,-1,-1):
            Um[i]=(b[:,i]/safe(Dm[i+1])).astype(F)
            Dm[i]=fma(-Um[i],b[:,i],d[i])
    gam=np.stack([np.abs((Dp[k]+Dm[k])-d[k]) for k in range(n)],1)
    gam=np.where(np.isfinite(gam),gam,F(np.inf))
    ks=np.argmin(gam,1)
    z=np.zeros((N,n),F)
    # per k* branch (vectorised by masks)
    for kstar in range(n):
        msk=ks==kstar
        if not msk.any(): continue
        zz=np.zeros((msk.sum(),n),F); zz[:,kstar]=1
        for i in range(kstar-1,-1,-1): zz[:,i]=-Lp[i][msk]*zz[:,i+1]
        for i in range(kstar,n-1): zz[:,i+1]=-Um[i][msk]*zz[:,i]
        z[msk]=zz
    # back transform: v = H0 H1 ... H5 z
    for k in range(n-3,-1,-1):
        m=n-1-k; v=V[k]
        dot=v[0]*z[:,k+1]
        for j in range(1,m): dot=fma(v[j],z[:,k+1+j],dot)
        c=BETA[k]*dot
        for j in range(m): z[:,k+1+j]=fma(-c,v[j],z[:,k+1+j])
    n2=z[:,0]*z[:,0]
    for i in range(1,n): n2=fma(z[:,i],z[:,i],n2)
    with np.errstate(all='ignore'):
        inv=np.where((n2>0)&np.isfinite(n2), F(1)/np.sqrt(n2), F(0)).astype(F)
    vv=(z*inv[:,None]).astype(F)
    bad=~((n2>0)&np.isfinite(n2))
    vv[bad]=0; vv[bad,0]=1
    # w = B v
    w=np.zeros((N,n),F)
    for i in range(n):
        acc=B[:,i,0]*vv[:,0]
        for j in range(1,n): acc=fma(B[:,i,j],vv[:,j],acc)
        w[:,i]=acc
    s0=w[:,0]*w[:,0]
    for i in range(1,n): s0=fma(w[:,i],w[:,i],s0)
    s0=np.sqrt(s0).astype(F)
    return s0_val,(s0,w,vv),iters



def test_svd8_top_triplet_solver_float32_replay():
    """DwtDctSvd*(blk=8) needs the top singular triplet of 8x8 LL blocks (dwt_dct_svd_encoder.py:41-45 with blk=8).  The
    kernel's algorithm (svd8_kernels.hiph: svd8_top) replayed in float32 against float64 LAPACK: random, LL-like,
    flat-plus-noise, exactly / nearly repeated top singular values, rank-1 and zero blocks, and the 8x8 LL blocks of Y, U
    and V of the reference's own frame63.jpeg.  The constants are read from the kernel source."""
    from conftest import natural_frame
    src = open(os.path.join(PKG, "csrc", "svd8_kernels.hiph")).read()
    assert "svd8_top<true>(B, kTolTight, 16)" in src and "svd8_top<false>(B, kTolTight, 16)" in src
    assert "7.f * fmaf(8.f, h, -g * g)" in src and "8.f * __builtin_amdgcn_rcpf(g + copysignf" in src      # Laguerre with n = 8
    tiny = float(re.search(r"kTiny = ([0-9.eE+-]+)f;", src).group(1))
    assert tiny == 1e-12
    tol = F(float(re.search(r"kTolTight = ([0-9.eE+-]+)f", open(os.path.join(PKG, "csrc", "svd_kernels.hiph")).read()).group(1)))
    rng = np.random.default_rng(5)
    sets = [rng.uniform(-200, 200, (3000, 8, 8)), rng.uniform(0, 510, (3000, 8, 8)), 1 + rng.normal(0, 1, (3000, 8, 8)),
            np.ones((1, 8, 8)) * rng.uniform(0, 300, (2000, 1, 1)) + rng.normal(0, 0.3, (2000, 8, 8))]
    adv = [np.zeros((8, 8)), np.ones((8, 8)), np.eye(8) * 7, np.diag([5., 5, 1, 0, 0, 0, 0, 0]), np.diag([3., 3, 3, 3, 3, 3, 3, 3.0000001]),
           np.diag([100., 99.9999, 1e-3, 0, 0, 0, 0, 0]), np.outer(np.arange(1., 9), np.arange(8., 0, -1)), np.diag([0., 0, 0, 0, 0, 0, 0, 9]),
           np.diag([1e-3, 0, 0, 0, 0, 0, 0, 0])]
    for k in range(400):
        q1, _ = np.linalg.qr(rng.normal(size=(8, 8)))
        q2, _ = np.linalg.qr(rng.normal(size=(8, 8)))
        s = np.sort(rng.uniform(0, 4000, 8))[::-1]
        if k % 4 == 0:
            s[1] = s[0]
        elif k % 4 == 1:
            s[1] = s[0] * (1 - 1e-5)
        elif k % 4 == 2:
            s[1:] = 0
        adv.append(q1 @ np.diag(s) @ q2.T)
    yuv = orc.bgr2yuv_f32(natural_frame()[::2, ::2].astype(F))                 # a quarter of the frame keeps the test short
    nat = [orc.to_blocks4(np.array(orc.haar_dwt2(yuv[: yuv.shape[0] // 4 * 4, : yuv.shape[1] // 4 * 4, ch])[0]), 8).reshape(-1, 8, 8) for ch in range(3)]
    B = np.concatenate(sets + [np.array(adv)] + nat).astype(F)
    U, S, Vt = np.linalg.svd(B.astype(np.float64))
    ref, gap = S[:, 0], S[:, 1] / np.maximum(S[:, 0], 1e-30)
    sval, (s0, w, v), iters = _svd8_solve(B, tol=tol)
    big = ref > 1e-3
    rel = lambda x: np.abs(x.astype(np.float64) - ref) / np.maximum(ref, 1e-30)      # noqa: E731
    assert rel(sval)[big].max() <= 5e-7 and rel(s0)[big].max() <= 4e-7, (rel(sval)[big].max(), rel(s0)[big].max())   # measured 3.1e-7 / 2.5e-7
    assert np.abs(sval - ref)[~big].max() <= 1e-3 and np.isfinite(v).all() and np.isfinite(s0).all()
    assert iters.max() <= 16 and iters.mean() <= 3.0
    # the rank-1 direction u0 v0^T the marking moves along: (w / s0) v^T against LAPACK's, where it is defined
    R = (w[:, :, None] * v[:, None, :]) / np.maximum(s0, 1e-30)[:, None, None]
    err = np.abs(R - U[:, :, 0][:, :, None] * Vt[:, 0, :][:, None, :]).reshape(len(B), -1).max(1)
    assert err[big & (gap < 0.9)].max() <= 2e-6 and err[big & (gap < 0.99)].max() <= 2e-5          # measured 6e-7 / 5.2e-6
    assert np.abs(np.linalg.norm(v.astype(np.float64), axis=1) - 1).max() <= 1e-6                   # unit vectors, the fallbacks included


def test_fmod_shortcut_is_fmod():
    """fmod_pos (svd_kernels.hiph): trunc of an inflated quotient estimate, one fma, one fix-up == fmodf, bit for bit, for
    0 <= a < 2^20 * b.  Replayed in float32 with a 1-ulp-wrong reciprocal in both directions (v_rcp_f32 is accurate to 1 ulp)."""
    src = open(os.path.join(PKG, "csrc", "svd_kernels.hiph")).read()
    m = re.search(r"__builtin_amdgcn_rcpf\(b\) \* ([0-9.]+)f\)", src)
    infl = F(float(m.group(1)))
    assert float(infl) > 1 + 3 * 2.0 ** -23                      # more than rcp's ulp plus two roundings
    rng = np.random.default_rng(9)
    b = np.concatenate([rng.uniform(0.5, 60, 200000), np.full(50000, 15.0), np.full(50000, 36.0)]).astype(F)
    a = (rng.uniform(0, 2100, b.size)).astype(F)
    n = rng.integers(0, 60, 60000)
    a[:60000] = (n * b[:60000].astype(np.float64)).astype(F)     # exact and nearly exact multiples
    a[60000:90000] = np.nextafter(a[:30000], F(np.inf))
    a[90000:120000] = np.nextafter(a[:30000], F(-np.inf)).clip(0)
    want = np.fmod(a, b)
    for wrong in (0, 1, -1):
        rcp = (F(1) / b).astype(F)
        rcp = np.nextafter(rcp, F(np.inf)) if wrong > 0 else np.nextafter(rcp, F(0)) if wrong < 0 else rcp
        q = np.trunc(((a * rcp).astype(F) * infl).astype(F))
        r = fma(-q, b, a)
        r = np.where(r < 0, r + b, r).astype(F)
        assert np.array_equal(r, want), (wrong, np.flatnonzero(r != want)[:5])
