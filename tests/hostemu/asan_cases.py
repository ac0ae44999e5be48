"""Run under an AddressSanitizer build of the kernel emulator (tests/test_hostemu_kernel.py::
test_emulator_under_address_sanitizer starts this file with libasan preloaded): a one-layer ACTIVE pair with 44 streams,
azimuth mode 0 only (the sanitizer's swapcontext hooks make an emulated N > 128 solve cost a minute) --
N = 132, not a multiple of 16, on the N > 128 pipeline, the shape that found the workspace overflow of round 5 -- and the
symmetric eigensolver on staging layouts whose leading dimension is not a multiple of eight.  Any out-of-bounds access of
the device code aborts the process; exit code 0 and the last line "asan cases ok" otherwise."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smrt_amd._native import PackedBatch, SmrtBatch  # noqa: E402

emu = C.CDLL(sys.argv[1])
P = C.POINTER
emu.smrt_emu_run.argtypes = [P(SmrtBatch), C.c_longlong, C.c_longlong, C.c_int, C.c_int, P(C.c_double), P(C.c_int32), P(C.c_double),
                             P(C.c_double), P(C.c_double), P(C.c_long)]
th = np.array([30.0])
b = PackedBatch([1], np.array([20.0]), np.array([380.0]) / 916.7, np.array([268.0]), np.array([0.9e-4]), None,
                [13.4e9], np.deg2rad(th), emmodel="iba", microstructure="exponential", mode="A", n_max_stream=44, m_max=0)
out = np.empty((1,) + b.out_shape())
st = np.empty(1, np.int32)
nb = C.c_long()
rc = emu.smrt_emu_run(C.byref(b.struct), 0, 1, 256, 0, out.ctypes.data_as(P(C.c_double)), st.ctypes.data_as(P(C.c_int32)), None, None, None,
                      C.byref(nb))
assert rc == 0 and st[0] == 0 and np.isfinite(out).all(), (rc, st)
emu.smrt_emu_eig_item.argtypes = [C.c_int, C.c_int, P(C.c_double), P(C.c_double), C.c_int, P(C.c_longlong)]
for NMAX, N in ((21, 21), (42, 37), (64, 64), (30, 5)):
    LD = (NMAX + 1) | 1
    rng = np.random.default_rng(N)
    A = rng.standard_normal((N, N)) @ np.diag(np.linspace(1, 9, N))
    buf = np.full((NMAX, LD), np.nan)
    buf[:N, :N] = A.T
    sig = np.zeros(NMAX)
    assert emu.smrt_emu_eig_item(NMAX, N, buf.ctypes.data_as(P(C.c_double)), sig.ctypes.data_as(P(C.c_double)), 1, None) == N
    U = buf[:N, :N].T / sig[None, :N]
    assert np.abs(U.T @ U - np.eye(N)).max() < 1e-13
print("asan cases ok")
