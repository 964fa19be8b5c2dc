import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from clima_amd import lib
L = lib.load()
rng = np.random.default_rng(0)
x = np.concatenate([10.0 ** rng.uniform(-300, 300, 200000), rng.uniform(1.0, 2.0, 200000), -10.0 ** rng.uniform(-5, 5, 1000)])
y = np.empty(4 * len(x)); err = C.create_string_buffer(1025); dp = C.POINTER(C.c_double)
L.clima_test_device_rcp(C.byref(C.c_int(len(x))), x.ctypes.data_as(dp), y.ctypes.data_as(dp), err)
y = y.reshape(4, -1)
ref = 1.0 / x
for k in range(3):
    rel = np.abs(y[k] - ref) / np.abs(ref)
    print("newton steps %d: max rel err %.3e  (in ulps of 2^-53: %.2f)  mean %.2e" % (k, rel.max(), rel.max() / 2**-53, rel.mean()))
