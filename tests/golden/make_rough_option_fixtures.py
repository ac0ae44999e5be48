"""Fixtures for the DORT options that interact with rough interfaces / rough substrates, generated from the REAL reference
(build container only):

  * prune_deep_snowpack cutting above a rough interface: the truncated system keeps that interface's dense reflection
    (smrt/rtsolver/dort.py:443-452 with rtsolver_utils.py:567-597) -- passive (iem_fung92) and active (geometrical_optics);
  * process_coherent_layers with a rough interface elsewhere in the snowpack (sampled on the streams of the REDUCED
    snowpack), with a rough interface ON the coherent layer (the reference's CoherentFlat takes its place,
    smrt/interface/coherent_flat.py:16-57), and with a rough substrate.

Inputs (plain arrays, the interface / substrate model by name + parameters: tests/conftest.py ROUGH_OPTION_CASES) and the
reference's result; `result_flat` is the same run with every interface Flat, so that a test can tell the two apart.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/_refstubs:/root/reference:tests python tests/golden/make_rough_option_fixtures.py
"""
import os
import warnings

import numpy as np

from smrt import make_interface, make_model, make_snowpack, make_soil, sensor_list

from conftest import ROUGH_OPTION_CASES

HERE = os.path.dirname(os.path.abspath(__file__))


def run(case, rough):
    itf = [None] * len(case["thickness"])
    if rough and case.get("interface"):
        model, kw, where = case["interface"]
        itf[where] = make_interface(model, **kw)
    substrate = None
    if case.get("substrate"):
        model, kw = case["substrate"]
        substrate = make_soil(model if rough else "flat", complex(*case["substrate_eps"]), case["substrate_temperature"],
                              **(kw if rough else {}))
    sp = make_snowpack(case["thickness"], "exponential", density=case["density"], temperature=case["temperature"],
                       corr_length=case["corr_length"], interface=itf, substrate=substrate)
    if case["mode"] == "A":
        sensor = sensor_list.active(case["frequency"], case["theta"])
    else:
        sensor = sensor_list.passive(case["frequency"], case["theta"])
    m = make_model("iba", "dort", rtsolver_options=case["options"])
    sims = list(m.prepare_simulations(sensor, sp, None, "snowpack")[0])
    res = m.run_single_simulation(sims[0], None, None)
    return np.asarray(res.data.values), res


def main():
    warnings.simplefilter("ignore")
    for name, case in ROUGH_OPTION_CASES.items():
        values, res = run(case, True)
        flat, _ = run(case, False)
        out = dict(result=values[None], result_flat=flat[None], kept_layers=len(res.other_data["ks"].values),
                   f0_ks=np.asarray(res.other_data["ks"].values), f0_thickness=np.asarray(res.other_data["thickness"].values))
        np.savez(os.path.join(HERE, name + ".npz"), **out)
        print(name, "kept layers", out["kept_layers"], "max |rough - flat| = %.3e" % np.abs(values - flat).max(),
              "rel %.3e" % (np.abs(values - flat) / np.abs(flat).clip(1e-300)).max())


if __name__ == "__main__":
    main()
