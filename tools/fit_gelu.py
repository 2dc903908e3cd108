import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf
x = np.linspace(-8, 8, 40001)
Phi = 0.5 * (1 + erf(x / np.sqrt(2)))
phi = np.exp(-x * x / 2) / np.sqrt(2 * np.pi)
g = x * Phi
dg = Phi + x * phi
def model(c, x):
    x2 = x * x
    p = c[0]
    for k in range(1, len(c)): p = p + c[k] * x2 ** k
    t = x * p
    s = 1 / (1 + np.exp(-t))
    # derivative of x*s(t): s + x*s*(1-s)*t'
    dp = c[0]
    for k in range(1, len(c)): dp = dp + (2 * k + 1) * c[k] * x2 ** k
    return x * s, s + x * s * (1 - s) * dp
for deg in (2, 3, 4):
    c0 = np.zeros(deg); c0[0] = 1.5957691216; 
    if deg > 1: c0[1] = 1.5957691216 * 0.044715
    def res(c):
        a, b = model(c, x)
        return np.concatenate([a - g, 0.5 * (b - dg)])
    r = least_squares(res, c0, xtol=1e-15, ftol=1e-15, gtol=1e-15)
    a, b = model(r.x, x)
    print(deg, "coef", list(r.x), "max |gelu err| %.2e" % np.abs(a - g).max(), "max |dgelu err| %.2e" % np.abs(b - dg).max())
a, b = model(np.array([1.5957691216, 1.5957691216 * 0.044715]), x)
print("tanh-gelu: max |gelu err| %.2e  max |dgelu err| %.2e" % (np.abs(a - g).max(), np.abs(b - dg).max()))
# current A&S erf
def as_model(x):
    z = np.abs(x) * 0.70710678; t = 1 / (1 + 0.3275911 * z); e = np.exp(-x * x / 2)
    poly = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))))
    erf_abs = 1 - poly * e; cdf = 0.5 * (1 + np.copysign(erf_abs, x)); pdf = 0.3989422804 * e
    return x * cdf, cdf + x * pdf
a, b = as_model(x)
print("A&S: max |gelu err| %.2e  max |dgelu err| %.2e" % (np.abs(a - g).max(), np.abs(b - dg).max()))
