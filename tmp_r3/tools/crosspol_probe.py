"""Where does the device's cross-polarised backscatter noise come from?  Device vs the oracle (reference default
method) on the active random batch of tests/test_gpu_parity.py, per pair: co- and cross-pol error on their own scale and
the spread of the oracle's own methods.  Run with different SMRT_DORT_JACOBI_EXIT2 / _SKIP2 to see what the Jacobi
thresholds contribute.   python tools/crosspol_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dort_oracle as O  # noqa: E402
from smrt_amd._native import DortContext, PackedBatch  # noqa: E402

rng = np.random.default_rng(11)
S, L = 5, 6
thick = np.concatenate([rng.uniform(0.02, 0.10, (S, L - 1)), np.full((S, 1), 1000.0)], axis=1)
dens, temp, lc = rng.uniform(150, 450, (S, L)), rng.uniform(230, 270, (S, L)), rng.uniform(5e-5, 3e-4, (S, L))
freqs = np.array([5.405e9, 13.4e9])
theta = np.array([25.0, 40.0, 55.0])
nl = np.array([6, 6, 4, 6, 2], np.int32)
ctx = DortContext(0)
print("SKIP2", os.environ.get("SMRT_DORT_JACOBI_SKIP2"), "EXIT2", os.environ.get("SMRT_DORT_JACOBI_EXIT2"))
for nstream, pipeline in ((16, 1), (16, 0), (32, 1)):
    b = PackedBatch(nl, thick, dens / 916.7, temp, lc, None, freqs, np.deg2rad(theta), emmodel="iba",
                    microstructure="exponential", mode="A", n_max_stream=nstream, m_max=2)
    ctx.set_pipeline(pipeline)
    out = ctx.run(b)
    ctx.set_pipeline(1)
    worst = [0, 0]
    for f in range(2):
        for s_ in range(S):
            n = nl[s_]
            sp = dict(thickness=thick[s_, :n], density=dens[s_, :n], temperature=temp[s_, :n], microstructure="exponential",
                      corr_length=lc[s_, :n])
            kw = dict(mode="A", theta_inc_deg=theta, n_max_stream=nstream, m_max=2)
            ref = O.solve(sp, freqs[f], theta, method="schur_forcedtriu", **kw)
            alt = O.solve(sp, freqs[f], theta, method="half_rank_eig", **kw)
            own = lambda x: (np.abs(x - ref) / np.abs(ref))[:2, :2]  # noqa: E731
            e, sp_ = own(out.values[f * S + s_]), own(alt)
            co, cr = max(e[0, 0].max(), e[1, 1].max()), max(e[0, 1].max(), e[1, 0].max())
            worst = [max(worst[0], co), max(worst[1], cr)]
            print("n%d pipe%d f%d s%d  HV/VV %.1e | device co %.1e cross %.1e | oracle half_rank co %.1e cross %.1e" % (
                nstream, pipeline, f, s_, (ref[0, 1] / ref[0, 0]).max(), co, cr, max(sp_[0, 0].max(), sp_[1, 1].max()),
                max(sp_[0, 1].max(), sp_[1, 0].max())))
    print("   worst device: co-pol %.1e cross-pol %.1e" % tuple(worst))
