"""Fit and check of the GELU used in the kernels (csrc/common.h gelu_fast): erfc(t) = 2^q(t), q of degree 8 on [0, 4] (weighted
minimax fit of log2 erfc), folded to p(a) = q(a / sqrt 2); prints the fp32 coefficients and the max error of an fp32 emulation of
the kernel formula against scipy's erf in float64, next to the Abramowitz-Stegun form it replaced."""
import numpy as np
from scipy.special import erf
import numpy as np
from scipy.special import erfc, erf
np.set_printoptions(precision=17)
T=4.0
def fit(deg, iters=40):
    # minimise weighted error in erf: err_erf ~ erfc(t)*ln2*dp ; chebyshev nodes, iteratively reweighted for minimax
    n=4000
    t=0.5*T*(1-np.cos(np.pi*(np.arange(n)+0.5)/n))
    y=np.log2(erfc(t))
    w=erfc(t)*np.log(2)
    # basis: chebyshev on [0,T] for conditioning, convert later
    x=2*t/T-1
    V=np.polynomial.chebyshev.chebvander(x,deg)
    # constraint p(0)=0 not enforced; fine
    wt=np.ones(n)
    for it in range(iters):
        A=V*(w*wt)[:,None]; b=y*w*wt
        c,*_=np.linalg.lstsq(A,b,rcond=None)
        e=np.abs((V@c-y)*w)
        wt=wt*(0.5+e/e.mean()*0.5); wt/=wt.mean()
    # convert to power basis in t
    p=np.polynomial.chebyshev.cheb2poly(c)      # in x
    P=np.polynomial.Polynomial(p)
    Pt=P(np.polynomial.Polynomial([-1,2/T]))    # compose x=2t/T-1
    return Pt.coef
def evalf32(coef,z):
    z=z.astype(np.float32); a=np.minimum(np.abs(z),np.float32(T)).astype(np.float32)
    c=coef.astype(np.float32)
    p=np.full_like(a,c[-1])
    for k in range(len(c)-2,-1,-1):
        p=(p*a+c[k]).astype(np.float32)
    e=np.exp2(p.astype(np.float32)).astype(np.float32)
    r=(np.float32(1)-e).astype(np.float32)
    return np.copysign(r,z)
c=fit(8)
s=1/np.sqrt(2.0)
cp=np.array([c[k]*s**k for k in range(9)])
cf=cp.astype(np.float32)
print([float(v) for v in cf])
CL=np.float32(4.0*np.sqrt(2.0))
def gelu32(x):
    x=x.astype(np.float32); a=np.minimum(np.abs(x),CL)
    p=np.full_like(a,cf[-1])
    for k in range(7,-1,-1): p=(p*a+cf[k]).astype(np.float32)
    e=np.exp2(p).astype(np.float32)
    sgn=(x+np.abs(x)).astype(np.float32); u=(a*e).astype(np.float32)
    return (np.float32(0.5)*(sgn-u)).astype(np.float32)
x=np.linspace(-12,12,4000001)
ref=0.5*x*(1+erf(x/np.sqrt(2)))
g=gelu32(x).astype(np.float64)
err=np.abs(g-ref); print("max abs err",err.max(),"at",x[err.argmax()])
rel=err/np.maximum(np.abs(ref),1e-3); print("max err / max(|ref|,1e-3)", rel.max(), x[rel.argmax()])
# old A&S
def old(x):
    x=x.astype(np.float32); z=(x*np.float32(0.70710678118654752440)).astype(np.float32); az=np.abs(z)
    t=(np.float32(1)/(np.float32(0.3275911)*az+np.float32(1))).astype(np.float32)
    p=np.float32(1.061405429)*t+np.float32(-1.453152027); p=p*t+np.float32(1.421413741); p=p*t+np.float32(-0.284496736); p=p*t+np.float32(0.254829592); p=(p*t).astype(np.float32)
    e=np.exp2((-az*az*np.float32(1.44269504088896340736)).astype(np.float32)).astype(np.float32)
    ea=(np.float32(1)-p*e).astype(np.float32)
    return (np.float32(0.5)*x*(np.float32(1)+np.copysign(ea,z))).astype(np.float32)
eo=np.abs(old(x).astype(np.float64)-ref); print("old max abs err",eo.max())

# ---- gelu16_fast (csrc/common.h): the form for results that are rounded to 16 bits right away (MLP hidden units, fc1 outputs) ----
# erfc(t)/2 = 2^q5(t) with q5 of degree 5, fitted for the absolute error of |x| erfc(|x| / sqrt 2) / 2 (the only inexact term of
# gelu(x) = (x + |x|) / 2 - |x| erfc(|x| / sqrt 2) / 2); the 1/2 is the constant term.  Absolute error <= 5e-7 = fp32 rounding level of
# the positive side and below half an fp16 ulp wherever |gelu| >= 1e-3.
def fit16(deg, iters=80):
    n=4000
    t=0.5*T*(1-np.cos(np.pi*(np.arange(n)+0.5)/n))
    y=np.log2(erfc(t)); w=erfc(t)*np.log(2)*np.maximum(t,1e-3)
    xx=2*t/T-1
    V=np.polynomial.chebyshev.chebvander(xx,deg)
    wt=np.ones(n)
    for it in range(iters):
        A=V*(w*wt)[:,None]; b=y*w*wt
        c,*_=np.linalg.lstsq(A,b,rcond=None)
        e=np.abs((V@c-y)*w); wt=wt*(0.5+e/e.mean()*0.5); wt/=wt.mean()
    p=np.polynomial.chebyshev.cheb2poly(c); P=np.polynomial.Polynomial(p)
    return P(np.polynomial.Polynomial([-1,2/T])).coef
c5=fit16(5); h5=np.array([c5[k]*s**k for k in range(6)]); h5[0]-=1.0
h5f=h5.astype(np.float32)
print("gelu16 coefficients", [float(v) for v in h5f])
def gelu16(x):
    x=x.astype(np.float32); ax=np.abs(x); a=np.minimum(ax,CL)
    p=np.full_like(a,h5f[-1])
    for k in range(4,-1,-1): p=(p*a+h5f[k]).astype(np.float32)
    e=np.exp2(p).astype(np.float32)
    t=(np.float32(0.5)*x).astype(np.float32); t=(np.float32(0.5)*ax+t).astype(np.float32)
    return (t-a*e).astype(np.float32)
g16=gelu16(x).astype(np.float64); e16=np.abs(g16-ref)
print("gelu16 max abs err", e16.max(), "at", x[e16.argmax()])
