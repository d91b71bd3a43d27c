import numpy as np
from scipy.special import ndtr
from scipy.optimize import least_squares
VMAX=6.0
v = np.linspace(-VMAX, VMAX, 24001)
ref = v * ndtr(v)
def model(c, v):
    u = np.minimum(v*v, VMAX*VMAX)
    p = np.zeros_like(v)
    for a in c[::-1]:
        p = p*u + a
    return v/(1+np.exp(-v*p))
def lawson(deg, wt, iters=60):
    c = np.zeros(deg+1); c[0]=1.5957691216; c[1]=0.0713548
    w = np.ones_like(v)
    best=None
    for it in range(iters):
        r = least_squares(lambda c: np.sqrt(w)*wt*(model(c,v)-ref), c, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        c = r.x
        e = np.abs(wt*(model(c,v)-ref))
        if best is None or e.max()<best[0]: best=(e.max(), c.copy())
        w = w*(e/e.max()+1e-3); w/=w.sum()/len(w)
    return best
for deg in (2,3):
    for name, wt in (("abs", np.ones_like(v)), ("mixed", 1/np.maximum(np.abs(ref), 2e-3))):
        m, c = lawson(deg, wt)
        err = np.abs(model(c,v)-ref)
        vv = np.linspace(-12,12,48001); rr = vv*ndtr(vv); ee=np.abs(model(c,vv)-rr)
        print(deg, name, [float(f"{x:.10g}") for x in c], "maxabs[-6,6]", err.max(), "maxabs[-12,12]", ee.max(), "weighted", m)
