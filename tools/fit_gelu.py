import numpy as np
from scipy.special import erfc, log_ndtr
from scipy.optimize import least_squares
a = np.linspace(0, 9.0, 20001)
r_true = -log_ndtr(-a)/np.log(2.0)         # -log2(Phi(-a))
target = a*np.exp(log_ndtr(-a))            # a*Phi(-a)
for deg in (4,5,6,7):
    # fit r(a) = 1 + c1 a + ... + cdeg a^deg minimizing max abs error of a*2^-r(a)
    V = np.vander(a, deg+1, increasing=True)[:,1:]
    c0 = np.linalg.lstsq(V[a<5], (r_true-1)[a<5], rcond=None)[0]
    def res(c, p=8):
        r = 1 + V@c
        e = a*np.exp2(-np.clip(r,-50,200)) - target
        return e
    c = c0
    for it in range(30):   # iteratively reweighted LS -> approx minimax
        w = np.ones_like(a) if it==0 else (np.abs(e)/np.abs(e).max()+0.05)**1.0
        sol = least_squares(lambda cc: res(cc)*w, c, method='lm', xtol=1e-15, ftol=1e-15)
        c = sol.x; e = res(c)
    r = 1+V@c
    print(deg, 'max abs err', np.abs(e).max(), 'monotone', bool((np.diff(r)>0).all()), 'coef', [1.0]+list(c))
