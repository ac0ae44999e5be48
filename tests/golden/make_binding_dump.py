"""Attribute dumps of REAL reference objects, for the GPU box (which has no /root/reference): what smrt's `Model.run`
hands a runner -- its Model, Sensor and Snowpack objects -- reduced to the public attributes smrt_amd/core/foreign.py
reads (class identity as [module, name], numbers, the names of the permittivity functions), plus what smrt's own
iba / dmrt + dort returned for them.  tests/conftest.py:standins_from_dump rebuilds stand-in objects of the same shape
and tests/test_gpu_model.py::test_reference_shaped_objects_through_the_runner runs them through HipBatchRunner on the
real kernels.  DATA only: no reference source text is stored.

Run in the build container:
    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/_refstubs:/root/reference python tests/golden/make_binding_dump.py
"""
import json
import os

import numpy as np

from smrt import make_model, make_snowpack, sensor_list
from smrt.atmosphere.simple_isotropic_atmosphere import SimpleIsotropicAtmosphere
from smrt.substrate.flat import Flat
from smrt.substrate.reflector import Reflector

HERE = os.path.dirname(os.path.abspath(__file__))


def cls_id(obj):
    c = obj if isinstance(obj, type) else type(obj)
    return [c.__module__, c.__name__]


def plain(v):
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, complex):
        return {"complex": [v.real, v.imag]}
    if isinstance(v, dict):
        return {"dict": [[plain(k), plain(x)] for k, x in v.items()]}
    if isinstance(v, (list, tuple)):
        return [plain(x) for x in v]
    return v


def dump_layer(layer):
    ms = layer.microstructure
    pm = layer.permittivity_model
    return dict(
        cls=cls_id(layer),
        attrs={k: plain(getattr(layer, k, None)) for k in ("thickness", "temperature", "density", "liquid_water",
                                                            "volumetric_liquid_water", "salinity", "medium",
                                                            "inclusion_shape", "emmodel", "emmodel_options")},
        frac_volume=float(layer.frac_volume),
        microstructure=dict(cls=cls_id(ms), attrs={k: plain(getattr(ms, k)) for k in list(ms.args) + list(ms.optional_args)}),
        permittivity_model=[plain(p) if not callable(p) else {"function": [p.__module__, p.__name__]} for p in pm])


def dump_snowpack(sp, frequencies):
    out = dict(cls=cls_id(sp), layers=[dump_layer(lay) for lay in sp.layers],
               interfaces=[cls_id(i) for i in sp.interfaces], substrate=None, atmosphere=None)
    if sp.substrate is not None:
        sub = sp.substrate
        out["substrate"] = dict(cls=cls_id(sub), temperature=plain(sub.temperature),
                                specular_reflection=plain(getattr(sub, "specular_reflection", None)),
                                permittivity=None if sub.permittivity_model is None else
                                [[f, plain(complex(sub.permittivity(f)))] for f in frequencies])
    if sp.atmosphere is not None:
        atm = sp.atmosphere
        out["atmosphere"] = dict(cls=cls_id(atm), attrs={k: plain(getattr(atm, k)) for k in
                                                         ("constant_tbdown", "constant_tbup", "constant_trans")})
    return out


def dump_sensor(s):
    return dict(cls=cls_id(s), mode=s.mode,
                attrs={k: plain(getattr(s, k)) for k in ("frequency", "theta_deg", "theta", "theta_inc_deg", "theta_inc",
                                                          "polarization", "polarization_inc", "phi", "channel_map", "name")})


def run_case(name, emmodel, sensor, packs, rtsolver_options, emmodel_options=None):
    m = make_model(emmodel, "dort", rtsolver_options=rtsolver_options, emmodel_options=emmodel_options)
    sims, dims = m.prepare_simulations(sensor, packs, None, "snowpack")
    sims = list(sims)
    results = [np.asarray(m.run_single_simulation(sim, None, None).data.values).tolist() for sim in sims]
    distinct_sensors, order = [], []
    for s, sp in sims:
        if not any(s is t for t in distinct_sensors):
            distinct_sensors.append(s)
        order.append([[t is s for t in distinct_sensors].index(True), [p is sp for p in packs].index(True)])
    freqs = [float(s.frequency) for s in distinct_sensors]
    return dict(name=name,
                model=dict(cls=cls_id(m), emmodel=cls_id(m.emmodel), emmodel_options=plain(m.emmodel_options),
                           rtsolver_options=plain(m.rtsolver_options)),
                sensors=[dump_sensor(s) for s in distinct_sensors],
                snowpacks=[dump_snowpack(sp, freqs) for sp in packs],
                simulations=order, results=results)


def main():
    rng = np.random.default_rng(44)

    def layers(L, last, micro="exponential"):
        kw = dict(density=rng.uniform(150, 450, L).tolist(), temperature=rng.uniform(230, 270, L).tolist())
        if micro == "exponential":
            kw["corr_length"] = rng.uniform(5e-5, 3e-4, L).tolist()
        else:
            kw["radius"] = rng.uniform(5e-5, 1.5e-4, L).tolist()
            kw["stickiness"] = 0.2
        return rng.uniform(0.05, 0.3, L - 1).tolist() + [last], kw

    cases = []
    th, kw = layers(20, 100.0)
    th2, kw2 = layers(20, 100.0)
    cases.append(run_case("headline_shape_two_snowpacks", "iba", sensor_list.amsre(),
                          [make_snowpack(th, "exponential", **kw), make_snowpack(th2, "exponential", **kw2)],
                          dict(n_max_stream=32)))
    th, kw = layers(3, 0.3)
    cases.append(run_case("flat_substrate_and_atmosphere", "iba", sensor_list.passive([18.7e9, 36.5e9], [30, 55]),
                          [make_snowpack(th, "exponential", **kw, substrate=Flat(temperature=268.0, permittivity_model=6.0 + 0.8j),
                                         atmosphere=SimpleIsotropicAtmosphere(tb_down={18.7e9: 20.0, 36.5e9: 32.0},
                                                                              tb_up={18.7e9: 6.0, 36.5e9: 11.0},
                                                                              transmittance={18.7e9: 0.95, 36.5e9: 0.9}))],
                          dict(n_max_stream=16)))
    th, kw = layers(4, 0.25, "sticky_hard_spheres")
    cases.append(run_case("dmrt_on_a_reflector", "dmrt_qca_shortrange", sensor_list.passive([10.65e9, 36.5e9], [55]),
                          [make_snowpack(th, "sticky_hard_spheres", **kw,
                                         substrate=Reflector(temperature=265.0, specular_reflection={"V": 0.2, "H": 0.35}))],
                          dict(n_max_stream=12)))
    th, kw = layers(3, 1000.0)
    cases.append(run_case("active_dense_auto", "iba", sensor_list.active(13.4e9, [30.0, 40.0]),
                          [make_snowpack(th, "exponential", density=[300.0, 700.0, 400.0], temperature=kw["temperature"],
                                         corr_length=kw["corr_length"])],
                          dict(n_max_stream=10, m_max=2), emmodel_options=dict(dense_snow_correction="auto")))
    with open(os.path.join(HERE, "reference_objects.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_binding_dump.py", cases=cases), f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
