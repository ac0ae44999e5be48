"""Driver entry points: build() compiles every native piece, smoke() runs one small solve on cuda:0.

    python __graft_entry__.py          # build
    python __graft_entry__.py smoke    # build + smoke (needs an MI355X)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "smrt_amd", "csrc")
LIB = os.path.join(CSRC, "libsmrt_dort.so")
OBJ_DIR = os.path.join(CSRC, "build")
EMU_DIR = os.path.join(ROOT, "tests", "hostemu")
EMU_LIB = os.path.join(EMU_DIR, "libsmrt_emu.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=ROOT)


def _compile_objects(hipcc, sources, headers, force, extra_flags=()):
    """One object per .hip translation unit, compiled in parallel (each kernel family is its own unit: the whole
    library builds in the time of its slowest unit instead of their sum)."""
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = []
    for src in sources:
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        if force or _newer(obj, [src] + headers):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                         *extra_flags, "-c", src, "-o", obj])

    def compile_one(cmd):
        """Compile one unit; the compiler's per-kernel resource remarks (registers, scratch, waves per SIMD) go to
        <object>.resources.txt -- tests/test_host_logic.py checks the occupancy the kernels were designed for."""
        print("+", " ".join(cmd), flush=True)
        proc = subprocess.run(cmd, cwd=ROOT, stderr=subprocess.PIPE, text=True)
        remarks, other = [], []
        for line in proc.stderr.splitlines():
            (remarks if "kernel-resource-usage" in line else other).append(line)
        kept = [ln for ln in other if not ln.lstrip().startswith(("|", "^")) and "__global__" not in ln]
        if proc.returncode != 0:
            sys.stderr.write(proc.stderr)
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        if kept:
            sys.stderr.write("\n".join(kept) + "\n")
        with open(cmd[-1][:-2] + ".resources.txt", "w") as fh:
            fh.write("\n".join(ln.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").rstrip()
                               for ln in remarks if "remark:" in ln) + "\n")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, jobs))
    return [os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o") for src in sources], bool(jobs)


def build(force=False):
    """Compile the gfx950 library (hipcc cross-compiles without a GPU) and the CPU-side kernel emulator used by the
    tests, then import the package."""
    units = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))   # every header of the device code
    headers.append(os.path.join(ROOT, "include", "smrt_dort.h"))
    hip_src = [os.path.join(CSRC, "dort_hip.hip")] + headers
    hipcc = os.environ.get("HIPCC") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc")
    objects, rebuilt = _compile_objects(hipcc, units, headers, force)
    if rebuilt or not os.path.exists(LIB):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objects, "-ldl"])
    emu_src = hip_src[1:] + [os.path.join(EMU_DIR, "emu_lib.cpp"), os.path.join(EMU_DIR, "emu_runtime.hpp")]
    # the emulator library of the CPU tests, and its -DSMRT_GJ_FAST_PANEL variant (tests/test_hostemu_kernel.py would
    # compile the latter itself -- a minute of the CPU suite -- if it were missing or older than the sources)
    emu_jobs = [(lib, flags) for lib, flags in ((EMU_LIB, []), (EMU_LIB[:-3] + "_fastpanel.so", ["-DSMRT_GJ_FAST_PANEL"]))
                if force or _newer(lib, emu_src)]
    if emu_jobs:
        with ThreadPoolExecutor(max_workers=len(emu_jobs)) as pool:
            list(pool.map(lambda j: _run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", *j[1], "-I", EMU_DIR, "-o", j[0],
                                          os.path.join(EMU_DIR, "emu_lib.cpp")]), emu_jobs))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import smrt_amd  # noqa: F401
    from smrt_amd import _native

    lib = _native.load_library()
    missing = [s for s in _native.EXPORTED_SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise RuntimeError("libsmrt_dort.so lacks symbols: %s" % missing)
    print("build ok:", lib.smrt_dort_version().decode())


def smoke():
    """One small (snowpack, frequency) solve on cuda:0 through the C ABI in passive and in active mode, checked against
    the CPU oracle and the reference's own known answers (smrt/test/test_integration_iba.py:48-49,67-69)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import numpy as np

    from oracle import dort_oracle as O  # checker only
    from smrt_amd._native import DortContext, PackedBatch

    sp = dict(thickness=np.array([0.1, 100.0]), density=np.array([200.0, 400.0]),
              temperature=np.array([250.0, 250.0]), microstructure="exponential",
              corr_length=np.array([5e-5, 5e-5]))
    theta = np.deg2rad([55.0])
    batch = PackedBatch([2], sp["thickness"], sp["density"] / O.DENSITY_OF_ICE, sp["temperature"], sp["corr_length"],
                        None, [36.5e9], theta)
    ctx = DortContext(0)
    out = ctx.run(batch)
    assert out.status[0] == 0, out.status
    tb = out.values[0]
    ref = O.solve(sp, 36.5e9, [55.0])
    print("smoke: Tb(V,H) gpu =", tb[:, 0], " oracle =", ref[:, 0], " kernel ms =", ctx.last_kernel_ms())
    assert np.abs(tb - ref).max() < 1e-6
    assert np.abs(tb[:, 0] - np.array([248.09044325849692, 237.3487270223389])).max() < 1e-4
    # and the radar case of the same snowpack (smrt/test/test_integration_iba.py:55-69): 19 GHz, 55 deg, backscatter
    radar = PackedBatch([2], sp["thickness"], sp["density"] / O.DENSITY_OF_ICE, sp["temperature"], sp["corr_length"],
                        None, [19e9], theta, mode="A", m_max=2)
    out = ctx.run(radar)
    assert out.status[0] == 0, out.status
    ref = O.solve(sp, 19e9, [55.0], mode="A", theta_inc_deg=[55.0])
    sig = 10 * np.log10(4 * np.pi * np.cos(theta[0]) * np.array([out.values[0][0, 0, 0], out.values[0][1, 1, 0], out.values[0][1, 0, 0]]))
    print("smoke: sigma0 (VV, HH, HV) gpu =", sig, "dB")
    assert (np.abs(out.values[0] - ref)[:2, :2] / np.abs(ref[:2, :2]).max(axis=(0, 1))).max() < 1e-8
    assert np.abs(sig - np.array([-24.044882546524693, -24.416295329469907, -51.544272924876886])).max() < 1e-4
    ctx.close()
    print("smoke ok")


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "smoke" in sys.argv:
        smoke()
