#!/usr/bin/env python
"""The pivot-free finish kernels (strip kernels and the register-resident one) on deliberately hard passive inputs, every pair
against the CPU oracle: layers from 0.1 mm to 100 m, ice volume fractions 0.05 ... 0.49, correlation lengths up to the
30 % renormalisation limit, 1.4 ... 183 GHz, 4 ... 32 streams, with and without a substrate / atmosphere.  Pairs the
oracle refuses must come back with the same status.
    python tools/stress_reg_extremes.py [seed] [n_cases] [big | deep]
        "big": 40 / 64 streams (the global-workspace pipeline);  "deep": up to 45 layers"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dort_oracle as O  # checker only
from smrt_amd._native import DortContext, PackedBatch



WORST_TWO = [0.0]   # the same pairs through the two-slot finish kernel (set_pipeline(4)), filled by run()
WORST_REG = [0.0]   # ... and through the register-resident finish kernel (set_pipeline(3); N <= 64 only)


def run(seed, n_cases, ctx, verbose=True, streams=(4, 7, 12, 16, 24, 32), max_layers=8):
  """(max |dTb| in K, pairs checked, pairs refused by both, status mismatches) of `n_cases` random hard cases."""
  rng = np.random.default_rng(seed)
  worst, checked, refused, mism = 0.0, 0, 0, 0
  WORST_TWO[0] = 0.0
  WORST_REG[0] = 0.0
  for case in range(n_cases):
      S, L = 6, int(rng.integers(1, max_layers + 1))
      n_str = int(rng.choice(list(streams)))
      thick = 10.0 ** rng.uniform(-4, 0.5, (S, L)); thick[:, -1] = rng.choice([0.3, 100.0], S)
      fv = rng.uniform(0.05, 0.49, (S, L)); temp = rng.uniform(200, 272.9, (S, L))
      lc = 10.0 ** rng.uniform(-5, -3.2, (S, L))
      freqs = np.sort(rng.choice([1.4e9, 6.9e9, 18.7e9, 36.5e9, 89e9, 150e9, 183e9], 3, replace=False))
      theta = np.array([rng.uniform(0, 20), rng.uniform(40, 75)])
      sub = atm = None
      if rng.random() < 0.5:
          sub = ("flat", np.full((3, S), rng.uniform(2, 30)), np.full((3, S), rng.uniform(0.01, 5)), rng.uniform(240, 273, S))
      if rng.random() < 0.4:
          atm = (rng.uniform(3, 80, 3), rng.uniform(2, 60, 3), rng.uniform(0.4, 1.0, 3))
      b = PackedBatch([L] * S, thick, fv, temp, lc, None, freqs, np.deg2rad(theta), n_max_stream=n_str, substrate=sub, atmosphere=atm)
      ctx.set_pipeline(1)   # the default: the strip finish kernels (four wavefronts for N <= 64, eight above)
      out = ctx.run(b)
      ctx.set_pipeline(3)   # the register-resident finish kernel (N <= 64; above, the strip kernel again)
      reg = ctx.run(b)
      ctx.set_pipeline(4)
      two = ctx.run(b)
      for fi, f in enumerate(freqs):
          for s in range(S):
              p = fi * S + s
              sp = dict(thickness=thick[s], frac_volume=fv[s], temperature=temp[s], microstructure="exponential", corr_length=lc[s])
              o_sub = None if sub is None else dict(kind="flat", eps=complex(sub[1][fi, s], sub[2][fi, s]), temperature=float(sub[3][s]))
              o_atm = None if atm is None else dict(tb_down=float(atm[0][fi]), tb_up=float(atm[1][fi]), transmittance=float(atm[2][fi]))
              try:
                  ref = O.solve(sp, float(f), np.rad2deg(np.deg2rad(theta)), n_max_stream=n_str, substrate=o_sub, atmosphere=o_atm)
                  st = 0
              except O.OracleError as e:
                  st = e.status
              if st != out.status[p] or st != two.status[p] or st != reg.status[p]:
                  mism += 1
                  if verbose: print("status mismatch: case %d pair %d oracle %d strip %d reg %d two-slot %d" % (case, p, st, out.status[p], reg.status[p], two.status[p]))
                  continue
              if st != 0:
                  refused += 1
                  continue
              e = float(np.abs(out.values[p] - ref).max())
              worst = max(worst, e); checked += 1
              WORST_TWO[0] = max(WORST_TWO[0], float(np.abs(two.values[p] - ref).max()))
              WORST_REG[0] = max(WORST_REG[0], float(np.abs(reg.values[p] - ref).max()))
              if e > 1e-6:
                  print("case %d pair %d: |dTb| = %.2e K (two-slot kernel: %.2e K), L = %d, n = %d, f = %.1f GHz, thinnest %.2e m" % (
                      case, p, e, float(np.abs(two.values[p] - ref).max()), L, n_str, f / 1e9, thick[s].min()))
  ctx.set_pipeline(1)
  return worst, checked, refused, mism


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    big = len(sys.argv) > 3 and sys.argv[3] == "big"     # 40 / 64 streams: the global-workspace pipeline (both runs are that one)
    deep = len(sys.argv) > 3 and sys.argv[3] == "deep"   # up to 45 layers
    worst, checked, refused, mism = run(seed, n_cases, DortContext(0), streams=(40, 64) if big else (4, 7, 12, 16, 24, 32),
                                        max_layers=45 if deep else 8)
    print("seed %d: %d pairs checked, max |dTb| = %.2e K with the strip finish kernels (register-resident kernel on the same pairs: %.2e K, "
          "two-slot / pivoted global-workspace kernel: %.2e K); %d refused by all (renormalisation / albedo); %d status mismatches"
          % (seed, checked, worst, WORST_REG[0], WORST_TWO[0], refused, mism))
