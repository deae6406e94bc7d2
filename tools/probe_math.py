import numpy as np, sys
sys.path.insert(0,'.')
import azdopt_amd as az
from azdopt_amd import _lib
rng = np.random.default_rng(0)
n = 1 << 16
x = rng.random(2 * n, dtype=np.float32)
x[: n // 2] *= np.float32(1e-38)
x[n // 2: n] = np.nextafter(x[n // 2: n], np.float32(2))
out = np.zeros(4 * n, np.float32)
_lib.check(az.lib().azd_debug_probe_math(0, _lib.ptr(x), _lib.ptr(out), n), "probe")
a, b = x[0::2], x[1::2]
d = np.abs(a-b)
ws = np.sqrt(d)
for name, off in (('azd_sqrt',0),('sqrtf (own kernel)',1),('f64 sqrt (own kernel)',2)):
    print(name, 'mismatches', (out[off::4].view(np.uint32) != ws.view(np.uint32)).sum())
g = out[0::4]
bad = np.nonzero(g.view(np.uint32) != ws.view(np.uint32))[0]
print('sqrt mismatches', len(bad), 'of', n)
for i in bad[:10]: print(i, d[i], d[i].view(np.uint32), g[i], ws[i], int(g[i].view(np.uint32)) - int(ws[i].view(np.uint32)))
# by class
sub = d < np.float32(1.17549435e-38)
print('subnormal inputs:', sub.sum(), 'bad among subnormal', np.isin(bad, np.nonzero(sub)[0]).sum())
wsub = a-(a-b); g2 = out[3::4]
print('sub mismatches', (g2.view(np.uint32) != wsub.view(np.uint32)).sum())
