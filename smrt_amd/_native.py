"""ctypes binding of libsmrt_dort.so (include/smrt_dort.h).

This is the ONLY way the package computes anything: there is no CPU fallback.  If the shared library is missing or no
MI355X is visible, `DortContext()` raises `SMRTError`.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from .core.error import SMRTError

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMRT_DORT_LIB: an alternative build of the same library (profiling / ablation builds made by tools/), never a fallback
LIB_PATH = os.environ.get("SMRT_DORT_LIB") or os.path.join(_HERE, "csrc", "libsmrt_dort.so")

EM_CODES = {"iba": 0, "dmrt_qca_shortrange": 1, "dmrt_qcacp_shortrange": 2, "nonscattering": 3, "host": 4,
            "iba_inverted": 5, "iba_host": 6, "rayleigh_host": 7}   # include/smrt_dort.h: SMRT_EM_*
MS_CODES = {"exponential": 0, "sticky_hard_spheres": 1, "independent_sphere": 2, "teubner_strey": 3,   # SMRT_MS_*
            # the models on the unified parameters are reparametrisations (core/layer.py: device_microstructure_params)
            "unified_scaled_exponential": 0, "unified_sticky_hard_spheres": 1, "unified_teubner_strey": 3,
            "exponential_complex_k": 4, "sticky_hard_spheres_complex_k": 5, "independent_sphere_complex_k": 6,
            "teubner_strey_complex_k": 7}   # SMRT_MS_*_COMPLEX_K: layers of the strong-contrast-expansion emmodels
SUBSTRATE_CODES = {"flat": 1, "reflector": 2, "host": 3}
NORM_CODES = {False: 0, None: 0, True: 1, "auto": 1, "forced": 2}
STATUS_MESSAGES = {
    1: "The eigen-decomposition did not converge in DORT.",
    2: "The re-normalization of the phase function exceeds the predefined threshold of 30%. This is likely because "
       "of a too large grain size or a bug in the phase function.",
    3: "The diagonalization failed in DORT: single scattering albedo >= 1 in a layer (too large grain size for the "
       "emmodel?).",
    4: "The boundary-condition system is singular.",
    5: "Invalid layer properties (temperature above the freezing point, fewer than two streams in a layer, or -- for an "
       "emmodel evaluated on the host -- a negative ka / permittivity or a stream count that differs from the device's).",
    6: "process_coherent_layers: the last layer is coherent, or two successive layers are coherent; this is not supported.",
}


class SmrtBatch(C.Structure):
    """struct smrt_batch of include/smrt_dort.h."""

    _fields_ = [
        ("n_snowpacks", C.c_int32),
        ("n_layers_max", C.c_int32),
        ("n_frequencies", C.c_int32),
        ("n_theta", C.c_int32),
        ("emmodel", C.c_int32),
        ("microstructure", C.c_int32),
        ("mode", C.c_int32),
        ("n_max_stream", C.c_int32),
        ("m_max", C.c_int32),
        ("phase_normalization", C.c_int32),
        ("rayleigh_jeans", C.c_int32),
        ("substrate_kind", C.c_int32),
        ("n_layers", C.POINTER(C.c_int32)),
        ("thickness", C.POINTER(C.c_double)),
        ("frac_volume", C.POINTER(C.c_double)),
        ("temperature", C.POINTER(C.c_double)),
        ("micro_p1", C.POINTER(C.c_double)),
        ("micro_p2", C.POINTER(C.c_double)),
        ("frequency", C.POINTER(C.c_double)),
        ("theta", C.POINTER(C.c_double)),
        ("phi", C.c_double),
        ("substrate_p1", C.POINTER(C.c_double)),
        ("substrate_p2", C.POINTER(C.c_double)),
        ("substrate_temperature", C.POINTER(C.c_double)),
        ("atm_tb_down", C.POINTER(C.c_double)),
        ("atm_tb_up", C.POINTER(C.c_double)),
        ("atm_transmittance", C.POINTER(C.c_double)),
        ("prune_optical_depth", C.c_double),
        ("layer_kind", C.POINTER(C.c_int32)),
        ("host_layer", C.POINTER(C.c_double)),
        ("host_streams", C.POINTER(C.c_int32)),
        ("host_phase", C.POINTER(C.c_double)),
        ("process_coherent_layers", C.c_int32),
        ("host_substrate", C.POINTER(C.c_double)),
        ("host_substrate_coh", C.POINTER(C.c_double)),
        ("host_interface_slot", C.POINTER(C.c_int32)),
        ("host_interface", C.POINTER(C.c_double)),
        ("host_interface_coh", C.POINTER(C.c_double)),
        ("host_interface_slots", C.c_int32),
        ("liquid_water", C.POINTER(C.c_double)),
        ("host_iba_coeff", C.POINTER(C.c_double)),
    ]


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class PackedBatch:
    """Host-side packed batch: S snowpacks x F frequencies (pair p = f * S + s, frequency-major like
    Model.prepare_simulations, smrt/core/model.py:485-502)."""

    def __init__(self, n_layers, thickness, frac_volume, temperature, micro_p1, micro_p2, frequency, theta,
                 emmodel="iba", microstructure="exponential", mode="P", n_max_stream=32, m_max=2,
                 phase_normalization="auto", rayleigh_jeans=False, phi=np.pi, substrate=None, atmosphere=None,
                 prune_deep_snowpack=None, layer_kind=None, host_emmodel=None, process_coherent_layers=False,
                 host_interfaces=None, liquid_water=None, host_scalars=None):
        """substrate: None or (kind, p1[F][S], p2[F][S], temperature[S]) with kind "flat" (p1 + i p2 = permittivity) or
        "reflector" (p1, p2 = specular reflection V, H); temperature <= 0 or NaN = no emission.
        atmosphere: None or (tb_down[F], tb_up[F], transmittance[F]).
        prune_deep_snowpack: None / False, True (= 6, smrt/rtsolver/dort.py:176-177) or the optical depth itself.
        layer_kind: None, or [S][Lmax] integer codes EM_CODES[emmodel] + 16 * MS_CODES[microstructure] for snowpacks
        that mix emmodels / microstructure models (smrt/core/model.py:529-582).
        host_interfaces: None, or (slot[F*S][Lmax] int (-1: Flat), matrices[F*S][slots][modes][4][NE][NE],
        coh[F*S][slots][4][NE]) for rough interfaces evaluated by the caller (include/smrt_dort.h: SMRT_INTERFACE_HOST).
        host_scalars: None, or (host_layer [F*S][Lmax][4] = ks, ka, Re eps_eff, Im eps_eff; iba_coeff [F*S][Lmax]) for layers of
            kind "iba_host" (SMRT_EM_IBA_HOST: IBA's phase function on the device, the layer's scalars from the caller)
        liquid_water: None (dry snow) or [S][Lmax] water / (ice + water) volume of every layer; frac_volume is then the
        volume fraction of ice + water (include/smrt_dort.h).
        host_emmodel: None, or (host_layer[F*S][Lmax][4], host_streams[F*S][Lmax], host_phase[F*S][Lmax][modes][2][NE][NE])
        for the layers of kind "host" (emmodels evaluated by the caller, include/smrt_dort.h)."""
        self.n_layers = np.ascontiguousarray(n_layers, dtype=np.int32)
        S = len(self.n_layers)
        two_d = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(S, -1))  # noqa: E731
        self.thickness = two_d(thickness)
        self.frac_volume = two_d(frac_volume)
        self.temperature = two_d(temperature)
        self.micro_p1 = two_d(micro_p1)
        self.micro_p2 = two_d(micro_p2 if micro_p2 is not None else np.zeros_like(self.micro_p1))
        Lmax = self.thickness.shape[1]
        for a in (self.frac_volume, self.temperature, self.micro_p1, self.micro_p2):
            if a.shape != (S, Lmax):
                raise SMRTError("per-layer arrays of a batch must share the shape (n_snowpacks, n_layers_max)")
        if self.n_layers.min() < 1 or self.n_layers.max() > Lmax:
            raise SMRTError("n_layers out of range")
        self.frequency = np.ascontiguousarray(np.atleast_1d(frequency), dtype=np.float64)
        self.theta = np.ascontiguousarray(np.atleast_1d(theta), dtype=np.float64)
        self.mode = mode
        s = SmrtBatch()
        s.n_snowpacks, s.n_layers_max, s.n_frequencies, s.n_theta = S, Lmax, len(self.frequency), len(self.theta)
        s.emmodel = EM_CODES[emmodel]
        s.microstructure = MS_CODES[microstructure]
        s.mode = 0 if mode == "P" else 1
        s.n_max_stream = int(n_max_stream)
        s.m_max = int(m_max)
        s.phase_normalization = NORM_CODES[phase_normalization]
        s.rayleigh_jeans = 1 if rayleigh_jeans else 0
        s.n_layers = self.n_layers.ctypes.data_as(C.POINTER(C.c_int32))
        s.thickness, s.frac_volume, s.temperature = _dptr(self.thickness), _dptr(self.frac_volume), _dptr(self.temperature)
        s.micro_p1, s.micro_p2 = _dptr(self.micro_p1), _dptr(self.micro_p2)
        s.frequency, s.theta = _dptr(self.frequency), _dptr(self.theta)
        s.phi = float(phi)
        if prune_deep_snowpack is True:
            prune_deep_snowpack = 6.0
        s.prune_optical_depth = float(prune_deep_snowpack) if prune_deep_snowpack else 0.0
        s.substrate_kind = 0
        if substrate is not None and substrate[0] == "host":
            # rough substrate, active mode: ("host", R[F*S][modes][NE][NE], Rcoh[F*S][modes][NE]) -- the dense reflection
            # matrices of the bottom boundary per azimuth mode and their specular diagonals (include/smrt_dort.h)
            # passive mode: one mode, Rcoh holds the EMISSIVITY diagonal and a fourth element the temperatures [S]
            FS, nm, ne = S * len(self.frequency), (int(m_max) + 1 if mode == "A" else 1), 3 * int(n_max_stream)
            self.host_substrate = np.ascontiguousarray(np.asarray(substrate[1], np.float64).reshape(FS, nm, ne, ne))
            self.host_substrate_coh = np.ascontiguousarray(np.asarray(substrate[2], np.float64).reshape(FS, nm, ne))
            s.substrate_kind = SUBSTRATE_CODES["host"]
            s.host_substrate, s.host_substrate_coh = _dptr(self.host_substrate), _dptr(self.host_substrate_coh)
            if len(substrate) > 3:
                self.sub_T = np.ascontiguousarray(np.nan_to_num(np.broadcast_to(np.asarray(substrate[3], np.float64), (S,)), nan=0.0))
                s.substrate_temperature = _dptr(self.sub_T)
        elif substrate is not None:
            kind, q1, q2, ts = substrate
            F = len(self.frequency)
            self.sub_p1 = np.ascontiguousarray(np.broadcast_to(np.asarray(q1, np.float64), (F, S)))
            self.sub_p2 = np.ascontiguousarray(np.broadcast_to(np.asarray(q2, np.float64), (F, S)))
            self.sub_T = np.ascontiguousarray(np.nan_to_num(np.broadcast_to(np.asarray(ts, np.float64), (S,)), nan=0.0))
            s.substrate_kind = SUBSTRATE_CODES[kind]
            s.substrate_p1, s.substrate_p2, s.substrate_temperature = _dptr(self.sub_p1), _dptr(self.sub_p2), _dptr(self.sub_T)
        if atmosphere is not None:
            F = len(self.frequency)
            self.atm = [np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float64), (F,))) for a in atmosphere]
            s.atm_tb_down, s.atm_tb_up, s.atm_transmittance = (_dptr(a) for a in self.atm)
        if layer_kind is not None:
            self.layer_kind = np.ascontiguousarray(np.asarray(layer_kind, dtype=np.int32).reshape(S, Lmax))
            s.layer_kind = self.layer_kind.ctypes.data_as(C.POINTER(C.c_int32))
        if host_emmodel is not None:
            hl, hs, hp = host_emmodel
            FS = S * len(self.frequency)
            modes = 1 if mode == "P" else int(m_max) + 1
            ne = int(n_max_stream) * (2 if mode == "P" else 3)
            self.host_layer = np.ascontiguousarray(np.asarray(hl, np.float64).reshape(FS, Lmax, 4))
            self.host_streams = np.ascontiguousarray(np.asarray(hs, np.int32).reshape(FS, Lmax))
            self.host_phase = np.ascontiguousarray(np.asarray(hp, np.float64).reshape(FS, Lmax, modes, 2, ne, ne))
            s.host_layer, s.host_phase = _dptr(self.host_layer), _dptr(self.host_phase)
            s.host_streams = self.host_streams.ctypes.data_as(C.POINTER(C.c_int32))
        if host_scalars is not None:
            if host_emmodel is not None:
                raise SMRTError("host_emmodel and host_scalars are alternatives (one host_layer array)")
            FS = S * len(self.frequency)
            self.host_layer = np.ascontiguousarray(np.asarray(host_scalars[0], np.float64).reshape(FS, Lmax, 4))
            self.host_iba_coeff = np.ascontiguousarray(np.asarray(host_scalars[1], np.float64).reshape(FS, Lmax))
            s.host_layer, s.host_iba_coeff = _dptr(self.host_layer), _dptr(self.host_iba_coeff)
        s.process_coherent_layers = 1 if process_coherent_layers else 0
        if liquid_water is not None:
            self.liquid_water = two_d(liquid_water)
            if self.liquid_water.shape != (S, Lmax):
                raise SMRTError("per-layer arrays of a batch must share the shape (n_snowpacks, n_layers_max)")
            s.liquid_water = _dptr(self.liquid_water)
        if host_interfaces is not None:
            FS, nm, ne = S * len(self.frequency), (int(m_max) + 1 if mode == "A" else 1), 3 * int(n_max_stream)
            slot = np.asarray(host_interfaces[0], dtype=np.int32).reshape(FS, Lmax)
            nslots = int(np.asarray(host_interfaces[1]).size // (FS * nm * 4 * ne * ne))
            self.host_interface_slot = np.ascontiguousarray(slot)
            self.host_interface = np.ascontiguousarray(np.asarray(host_interfaces[1], np.float64).reshape(FS, nslots, nm, 4, ne, ne))
            self.host_interface_coh = np.ascontiguousarray(np.asarray(host_interfaces[2], np.float64).reshape(FS, nslots, 4, ne))
            s.host_interface_slot = self.host_interface_slot.ctypes.data_as(C.POINTER(C.c_int32))
            s.host_interface, s.host_interface_coh = _dptr(self.host_interface), _dptr(self.host_interface_coh)
            s.host_interface_slots = nslots
        self.struct = s

    @property
    def n_pairs(self):
        return int(self.struct.n_snowpacks) * int(self.struct.n_frequencies)

    @property
    def out_stride(self):
        return (2 if self.mode == "P" else 9) * int(self.struct.n_theta)

    def out_shape(self):
        nt = int(self.struct.n_theta)
        return (2, nt) if self.mode == "P" else (3, 3, nt)


_lib = None


def load_library():
    """Load libsmrt_dort.so; fail loudly when it is not built (see __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SMRTError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
                        "smrt_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.smrt_dort_version.restype = C.c_char_p
    lib.smrt_dort_out_stride.argtypes = [P(SmrtBatch)]
    lib.smrt_dort_out_stride.restype = C.c_int32
    lib.smrt_dort_create.argtypes = [P(C.c_void_p), C.c_int32]
    lib.smrt_dort_create.restype = C.c_int32
    lib.smrt_dort_destroy.argtypes = [C.c_void_p]
    lib.smrt_dort_destroy.restype = None
    lib.smrt_dort_last_error.argtypes = [C.c_void_p]
    lib.smrt_dort_last_error.restype = C.c_char_p
    lib.smrt_dort_run.argtypes = [C.c_void_p, P(SmrtBatch), C.c_int64, C.c_int64, P(C.c_double), P(C.c_int32),
                                  P(C.c_double), P(C.c_double)]
    lib.smrt_dort_run.restype = C.c_int32
    lib.smrt_dort_upload.argtypes = [C.c_void_p, P(SmrtBatch), C.c_int64, C.c_int64]
    lib.smrt_dort_upload.restype = C.c_int32
    lib.smrt_dort_run_pairs.argtypes = [C.c_void_p, P(SmrtBatch), P(C.c_int64), C.c_int64, P(C.c_double), P(C.c_int32),
                                        P(C.c_double), P(C.c_double)]
    lib.smrt_dort_run_pairs.restype = C.c_int32
    lib.smrt_dort_upload_pairs.argtypes = [C.c_void_p, P(SmrtBatch), P(C.c_int64), C.c_int64]
    lib.smrt_dort_upload_pairs.restype = C.c_int32
    lib.smrt_dort_ft_even_phase.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                            C.c_double, P(C.c_double), C.c_int32, P(C.c_double), C.c_int32, C.c_int32, C.c_int32,
                                            P(C.c_double)]
    lib.smrt_dort_ft_even_phase.restype = C.c_int32
    lib.smrt_dort_pair_cost.argtypes = [C.c_void_p, P(C.c_double)]
    lib.smrt_dort_pair_cost.restype = C.c_int32
    lib.smrt_dort_launch_info.argtypes = [C.c_void_p, P(C.c_int64), C.c_int32]
    lib.smrt_dort_launch_info.restype = C.c_int32
    lib.smrt_dort_comm_library.argtypes = [C.c_char_p, C.c_int32, P(C.c_int32)]
    lib.smrt_dort_comm_library.restype = C.c_int32
    lib.smrt_dort_comm_unique_id.argtypes = [C.c_char_p]
    lib.smrt_dort_comm_unique_id.restype = C.c_int32
    lib.smrt_dort_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p]
    lib.smrt_dort_comm_init.restype = C.c_int32
    lib.smrt_dort_comm_init_all.argtypes = [P(C.c_void_p), C.c_int32]
    lib.smrt_dort_comm_init_all.restype = C.c_int32
    lib.smrt_dort_comm_destroy.argtypes = [C.c_void_p]
    lib.smrt_dort_comm_destroy.restype = C.c_int32
    lib.smrt_dort_gather.argtypes = [C.c_void_p, C.c_int32, P(C.c_int64), P(C.c_double), P(C.c_int32)]
    lib.smrt_dort_gather.restype = C.c_int32
    lib.smrt_dort_comm_allreduce_max.argtypes = [C.c_void_p, P(C.c_double), C.c_int32]
    lib.smrt_dort_comm_allreduce_max.restype = C.c_int32
    lib.smrt_dort_abi.argtypes = [P(C.c_int32), C.c_int32]
    lib.smrt_dort_abi.restype = C.c_int32
    lib.smrt_dort_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.smrt_dort_launch.restype = C.c_int32
    lib.smrt_dort_sync.argtypes = [C.c_void_p]
    lib.smrt_dort_sync.restype = C.c_int32
    lib.smrt_dort_download.argtypes = [C.c_void_p, P(C.c_double), P(C.c_int32), P(C.c_double), P(C.c_double)]
    lib.smrt_dort_download.restype = C.c_int32
    lib.smrt_dort_last_kernel_ms.argtypes = [C.c_void_p]
    lib.smrt_dort_last_kernel_ms.restype = C.c_double
    lib.smrt_dort_kernel_breakdown.argtypes = [C.c_void_p, C.c_int32, P(C.c_double)]
    lib.smrt_dort_kernel_breakdown.restype = C.c_int32
    lib.smrt_dort_total_kernel_ms.argtypes = [C.c_void_p, P(C.c_int64), C.c_int32]
    lib.smrt_dort_total_kernel_ms.restype = C.c_double
    lib.smrt_dort_set_block_threads.argtypes = [C.c_void_p, C.c_int32]
    lib.smrt_dort_set_block_threads.restype = C.c_int32
    lib.smrt_dort_set_pipeline.argtypes = [C.c_void_p, C.c_int32]
    lib.smrt_dort_set_pipeline.restype = C.c_int32
    lib.smrt_dort_set_diagonalisation.argtypes = [C.c_void_p, C.c_int32]
    lib.smrt_dort_set_diagonalisation.restype = C.c_int32
    lib.smrt_dort_gather_plan.argtypes = [C.c_int32, C.c_int32, C.c_int32, P(C.c_int64), C.c_void_p, C.c_int32,
                                          P(C.c_int64), P(C.c_int64)]
    lib.smrt_dort_gather_plan.restype = C.c_int32
    lib.smrt_dort_finish_reg_lds_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.smrt_dort_finish_reg_lds_bytes.restype = C.c_int32
    lib.smrt_dort_finish_strip_lds_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.smrt_dort_finish_strip_lds_bytes.restype = C.c_int32
    lib.smrt_dort_jacobi_lds_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.smrt_dort_jacobi_lds_bytes.restype = C.c_int32
    lib.smrt_dort_sum_n3.argtypes = [C.c_void_p]
    lib.smrt_dort_sum_n3.restype = C.c_double
    lib.smrt_dort_stage_cycles.argtypes = [C.c_void_p, P(C.c_double)]
    lib.smrt_dort_stage_cycles.restype = C.c_int32
    lib.smrt_dort_device_count.argtypes = []
    lib.smrt_dort_device_count.restype = C.c_int32
    lib.smrt_gauss_legendre_positive.argtypes = [C.c_int32, P(C.c_double), P(C.c_double)]
    lib.smrt_gauss_legendre_positive.restype = C.c_int32
    check_struct_layout(lib)
    _lib = lib
    return lib


def abi_layout(lib):
    """[sizeof(smrt_batch), offset of every field in declaration order] as the library was compiled (smrt_dort_abi)."""
    n = lib.smrt_dort_abi(None, 0)
    a = (C.c_int32 * n)()
    lib.smrt_dort_abi(a, n)
    return list(a)


def check_struct_layout(lib):
    """The ctypes declaration above must be the struct the library was compiled with: a stale binding would hand over
    a short or shifted struct and the library would read garbage pointers."""
    mine = [C.sizeof(SmrtBatch)] + [getattr(SmrtBatch, name).offset for name, _ in SmrtBatch._fields_]
    theirs = abi_layout(lib)
    if mine != theirs:
        raise SMRTError(f"smrt_batch layout mismatch between smrt_amd/_native.py {mine} and {LIB_PATH} {theirs}: "
                        "rebuild the library or update the binding (include/smrt_dort.h)")


EXPORTED_SYMBOLS = [
    "smrt_dort_out_stride", "smrt_dort_create", "smrt_dort_destroy", "smrt_dort_last_error", "smrt_dort_run",
    "smrt_dort_upload", "smrt_dort_upload_pairs", "smrt_dort_run_pairs", "smrt_dort_abi", "smrt_dort_pair_cost", "smrt_dort_ft_even_phase",
    "smrt_dort_launch_info", "smrt_dort_comm_library", "smrt_dort_comm_unique_id", "smrt_dort_comm_init", "smrt_dort_comm_init_all", "smrt_dort_comm_destroy", "smrt_dort_gather",
    "smrt_dort_comm_allreduce_max", "smrt_dort_launch", "smrt_dort_sync", "smrt_dort_download", "smrt_dort_last_kernel_ms", "smrt_dort_kernel_breakdown",
    "smrt_dort_total_kernel_ms", "smrt_dort_set_block_threads", "smrt_dort_set_pipeline", "smrt_dort_set_diagonalisation", "smrt_dort_sum_n3", "smrt_dort_stage_cycles", "smrt_dort_device_count", "smrt_gauss_legendre_positive",
    "smrt_dort_version", "smrt_dort_finish_reg_lds_bytes", "smrt_dort_finish_strip_lds_bytes", "smrt_dort_jacobi_lds_bytes", "smrt_dort_gather_plan",
]


class BatchOutput:
    def __init__(self, batch, pair_count):
        Lmax, nmax = int(batch.struct.n_layers_max), int(batch.struct.n_max_stream)
        self.values = np.empty((pair_count,) + batch.out_shape(), dtype=np.float64)
        self.status = np.empty(pair_count, dtype=np.int32)
        self.layers = np.empty((pair_count, Lmax, 5), dtype=np.float64)
        self.streams = np.empty((pair_count, 1 + nmax), dtype=np.float64)


class DortContext:
    """One context per GPU (smrt_dort_create / smrt_dort_destroy)."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.smrt_dort_create(C.byref(self._h), int(device))
        if rc != 0:
            self._h = C.c_void_p()
            raise SMRTError(f"smrt_dort_create(device={device}) failed with code {rc}: no usable MI355X GPU. "
                            "smrt_amd has no CPU fallback.")
        self.device = int(device)
        # a context is one set of device buffers and one stream: its calls are serialised (ctypes releases the GIL,
        # so two Python threads sharing a cached context would otherwise interleave upload / launch / download)
        self.lock = threading.RLock()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.smrt_dort_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise SMRTError(f"{what} failed: {self._lib.smrt_dort_last_error(self._h).decode()}")

    def set_block_threads(self, n):
        self._check(self._lib.smrt_dort_set_block_threads(self._h, int(n)), "smrt_dort_set_block_threads")

    def set_pipeline(self, split=1):
        """1 (default): prep / Jacobi / finish kernels (passive, Flat interfaces: the strip finish kernels); 3: the
        register-resident finish kernel instead (N <= 64); 5: the strip kernels wherever supported; 4: no pivot-free finish
        kernel; 2: the four-matrix LDS finish kernel; 0: one fused kernel per pair (include/smrt_dort.h)."""
        self._check(self._lib.smrt_dort_set_pipeline(self._h, int(split)), "smrt_dort_set_pipeline")

    DIAGONALISATIONS = ("jacobi", "symmetric")   # SMRT_DIAG_*

    def set_diagonalisation(self, mode="default"):
        """How the layer eigenproblems are diagonalised on the three-kernel pipelines: "jacobi" (one-sided Jacobi on
        B = L+^T L-), "symmetric" (tridiagonalisation + implicit QL on B B^T where it is built: N <= 64) or "default"."""
        code = -1 if mode in (None, "default") else self.DIAGONALISATIONS.index(mode)
        self._check(self._lib.smrt_dort_set_diagonalisation(self._h, code), "smrt_dort_set_diagonalisation")

    def run(self, batch: PackedBatch, pair_begin=0, pair_count=-1, pairs=None) -> BatchOutput:
        """One shot (H2D, kernels, D2H) for the pair range, or for the listed pair indices (row i = pairs[i])."""
        if pairs is not None:
            pairs = np.ascontiguousarray(pairs, dtype=np.int64)
            o = BatchOutput(batch, len(pairs))
            with self.lock:
                self._check(self._lib.smrt_dort_run_pairs(
                    self._h, C.byref(batch.struct), pairs.ctypes.data_as(C.POINTER(C.c_int64)), len(pairs),
                    _dptr(o.values), o.status.ctypes.data_as(C.POINTER(C.c_int32)), _dptr(o.layers), _dptr(o.streams)),
                    "smrt_dort_run_pairs")
            return o
        if pair_count < 0:
            pair_count = batch.n_pairs - pair_begin
        o = BatchOutput(batch, pair_count)
        with self.lock:
            self._check(self._lib.smrt_dort_run(self._h, C.byref(batch.struct), pair_begin, pair_count, _dptr(o.values),
                                                o.status.ctypes.data_as(C.POINTER(C.c_int32)), _dptr(o.layers),
                                                _dptr(o.streams)), "smrt_dort_run")
        return o

    def upload(self, batch: PackedBatch, pair_begin=0, pair_count=-1, pairs=None):
        if pairs is not None:
            pairs = np.ascontiguousarray(pairs, dtype=np.int64)
            self._check(self._lib.smrt_dort_upload_pairs(self._h, C.byref(batch.struct),
                                                         pairs.ctypes.data_as(C.POINTER(C.c_int64)), len(pairs)),
                        "smrt_dort_upload_pairs")
            self._resident = (batch, len(pairs))
            return
        if pair_count < 0:
            pair_count = batch.n_pairs - pair_begin
        self._check(self._lib.smrt_dort_upload(self._h, C.byref(batch.struct), pair_begin, pair_count), "smrt_dort_upload")
        self._resident = (batch, pair_count)

    def launch(self, out_dev=None, status_dev=None):
        self._check(self._lib.smrt_dort_launch(self._h, C.c_void_p(out_dev or 0), C.c_void_p(status_dev or 0)),
                    "smrt_dort_launch")

    def sync(self):
        self._check(self._lib.smrt_dort_sync(self._h), "smrt_dort_sync")

    def download(self) -> BatchOutput:
        batch, pair_count = self._resident
        o = BatchOutput(batch, pair_count)
        self._check(self._lib.smrt_dort_download(self._h, _dptr(o.values), o.status.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 _dptr(o.layers), _dptr(o.streams)), "smrt_dort_download")
        return o

    def ft_even_phase(self, emmodel, microstructure, frequency, frac_volume, temperature, p1, p2, mu_s, mu_i, m_max, npol):
        """Azimuthal modes of the phase matrix of one layer: array [npol, npol, m_max + 1, len(mu_s), len(mu_i)]."""
        mu_s = np.ascontiguousarray(np.atleast_1d(mu_s), dtype=np.float64)
        mu_i = np.ascontiguousarray(np.atleast_1d(mu_i), dtype=np.float64)
        out = np.empty((npol, npol, m_max + 1, len(mu_s), len(mu_i)))
        with self.lock:
            self._check(self._lib.smrt_dort_ft_even_phase(
                self._h, EM_CODES[emmodel], MS_CODES[microstructure], float(frequency), float(frac_volume), float(temperature),
                float(p1), float(p2), _dptr(mu_s), len(mu_s), _dptr(mu_i), len(mu_i), int(m_max), int(npol), _dptr(out)),
                "smrt_dort_ft_even_phase")
        return out

    def pair_cost(self):
        """Sum of N_l^3 per pair of the uploaded batch (before solving it): what the work is sharded by."""
        _, pair_count = self._resident
        cost = np.empty(pair_count)
        self._check(self._lib.smrt_dort_pair_cost(self._h, _dptr(cost)), "smrt_dort_pair_cost")
        return cost

    # ---- multi-GPU: the RCCL gather of the C ABI (smrt_dort_comm_*, smrt_dort_gather) ----------------------------
    PIPELINES = ("fused", "lds_two_slot", "lds_four_slot", "lds_reg", "fused_gmem", "gmem", "big", "gmem_strip", "lds_strip")   # SMRT_PIPELINE_*

    def launch_info(self):
        """smrt_dort_launch_info as a dict: pipeline (name), chunk_pairs, chunks, prune_rounds, staged_items (None when
        unknown), block_threads, n_max, diagonalisation (name), rayleigh_closed_form (bool)."""
        v = (C.c_int64 * 16)()
        n = self._lib.smrt_dort_launch_info(self._h, v, 16)
        self._check(0 if n > 0 else -1, "smrt_dort_launch_info")
        keys = ("pipeline", "chunk_pairs", "chunks", "prune_rounds", "staged_items", "block_threads", "n_max", "diagonalisation",
                "rayleigh_closed_form")
        d = {k: int(v[i]) for i, k in enumerate(keys[:n])}
        d["pipeline"] = self.PIPELINES[d["pipeline"]]
        d["diagonalisation"] = self.DIAGONALISATIONS[d["diagonalisation"]]
        d["rayleigh_closed_form"] = bool(d.get("rayleigh_closed_form", 0))
        if d.get("staged_items", -1) < 0:
            d["staged_items"] = None
        return d

    @staticmethod
    def comm_library():
        """(path of the RCCL library the gather runs on, its version string) -- smrt_dort_comm_library; SMRT_RCCL_LIB
        pins it."""
        lib = load_library()
        buf, version = C.create_string_buffer(1024), C.c_int32(0)
        if lib.smrt_dort_comm_library(buf, len(buf), C.byref(version)) != 0:
            raise SMRTError("RCCL is not available: " + buf.value.decode(errors="replace"))
        v = int(version.value)
        return buf.value.decode(), "%d.%d.%d" % (v // 10000, v // 100 % 100, v % 100) if v >= 10000 else str(v)

    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = C.create_string_buffer(128)
        if lib.smrt_dort_comm_unique_id(buf) != 0:
            raise SMRTError("smrt_dort_comm_unique_id failed (is librccl available?)")
        return buf.raw

    def comm_init(self, world, rank, unique_id):
        self._check(self._lib.smrt_dort_comm_init(self._h, int(world), int(rank), bytes(unique_id)), "smrt_dort_comm_init")
        self.world, self.rank = int(world), int(rank)

    @staticmethod
    def comm_init_all(contexts):
        lib = load_library()
        arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
        if lib.smrt_dort_comm_init_all(arr, len(contexts)) != 0:
            raise SMRTError("smrt_dort_comm_init_all failed: " + lib.smrt_dort_last_error(contexts[0]._h).decode())
        for r, c in enumerate(contexts):
            c.world, c.rank = len(contexts), r

    def gather(self, counts, root=0, want_host=True):
        """Collective: rows of the last launch of every rank -> root (rank order).  Returns (values, status) on the
        root (None, None elsewhere, or when want_host is False: the rows then stay on the root's device)."""
        batch, _ = self._resident
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        is_root = self.rank == root and want_host
        total = int(counts.sum())
        values = np.empty((total,) + batch.out_shape()) if is_root else None
        status = np.empty(total, np.int32) if is_root else None
        self._check(self._lib.smrt_dort_gather(self._h, int(root), counts.ctypes.data_as(C.POINTER(C.c_int64)),
                                               _dptr(values) if is_root else None,
                                               status.ctypes.data_as(C.POINTER(C.c_int32)) if is_root else None),
                    "smrt_dort_gather")
        return values, status

    def allreduce_max(self, values):
        a = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self._check(self._lib.smrt_dort_comm_allreduce_max(self._h, _dptr(a), len(a)), "smrt_dort_comm_allreduce_max")
        return a

    def barrier(self):
        self._check(self._lib.smrt_dort_comm_allreduce_max(self._h, None, 0), "smrt_dort_comm_allreduce_max")

    def last_kernel_ms(self):
        return float(self._lib.smrt_dort_last_kernel_ms(self._h))

    def total_kernel_ms(self, reset=False):
        n = C.c_int64()
        ms = float(self._lib.smrt_dort_total_kernel_ms(self._h, C.byref(n), 1 if reset else 0))
        return ms, int(n.value)

    STAGE_NAMES = ["setup", "assemble", "cholesky", "LtL", "jacobi", "triangular", "R1", "LU1", "R45", "LU2", "R78",
                   "out"]

    def kernel_breakdown(self, enable=None):
        """Per-kernel HIP-event times: kernel_breakdown(True) instruments the following launches, kernel_breakdown() returns
        {"prep", "jacobi", "finish"} in ms for the last launch ("jacobi": the diagonalisation stage, whichever kernels run
        it -- include/smrt_dort.h)."""
        if enable is not None:
            self._check(min(self._lib.smrt_dort_kernel_breakdown(self._h, 1 if enable else 0, None), 0), "smrt_dort_kernel_breakdown")
            return None
        a = np.zeros(3)
        n = self._lib.smrt_dort_kernel_breakdown(self._h, -1, _dptr(a))   # read only: the instrumentation stays as it is
        self._check(min(n, 0), "smrt_dort_kernel_breakdown")
        return {"prep": float(a[0]), "jacobi": float(a[1]), "finish": float(a[2]), "intervals": int(n)}

    def stage_cycles(self):
        a = np.zeros(16)
        self._check(self._lib.smrt_dort_stage_cycles(self._h, _dptr(a)), "smrt_dort_stage_cycles")
        d = dict(zip(self.STAGE_NAMES, a[: len(self.STAGE_NAMES)]))
        d["_jacobi_sweeps"] = a[12]
        d["_gj_panel"], d["_gj_update"], d["_gj_perm"] = a[13], a[14], a[15]
        return d

    def sum_n3(self):
        return float(self._lib.smrt_dort_sum_n3(self._h))


class GatherOp(C.Structure):
    """smrt_gather_op of include/smrt_dort.h."""
    _fields_ = [("peer", C.c_int32), ("reserved", C.c_int32), ("offset_rows", C.c_int64), ("rows", C.c_int64)]


def gather_plan(world, root, rank, counts):
    """The transfers smrt_dort_gather issues on `rank` (smrt_dort_gather_plan: host arithmetic, needs no GPU):
    ([(peer, offset_rows, rows), ...], own_offset_rows, total_rows)."""
    lib = load_library()
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    ops = (GatherOp * max(int(world), 1))()
    own, total = C.c_int64(0), C.c_int64(0)
    n = lib.smrt_dort_gather_plan(int(world), int(root), int(rank), counts.ctypes.data_as(C.POINTER(C.c_int64)),
                                  C.cast(ops, C.c_void_p), int(world), C.byref(own), C.byref(total))
    if n < 0:
        raise SMRTError("smrt_dort_gather_plan: invalid arguments")
    return [(o.peer, o.offset_rows, o.rows) for o in ops[:n]], own.value, total.value


def device_count():
    """Number of visible GPUs (smrt_dort_device_count); 0 when there is none."""
    return int(load_library().smrt_dort_device_count())


def gauss_legendre_positive(n):
    lib = load_library()
    mu = np.empty(n)
    w = np.empty(n)
    lib.smrt_gauss_legendre_positive(n, _dptr(mu), _dptr(w))
    return mu, w
