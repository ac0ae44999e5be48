#!/usr/bin/env python
"""Build an experimental variant of libsmrt_dort.so with extra compiler flags (ablations, profiling builds):

    python tools/build_variant.py timing -DSMRT_STAGE_TIMING
    SMRT_DORT_LIB=smrt_amd/csrc/variants/libsmrt_dort_timing.so python tools/stage_profile.py

The product library is untouched; variants live under smrt_amd/csrc/variants/ (git-ignored, they travel with gpurun)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    units = sorted(os.path.join(G.CSRC, f) for f in os.listdir(G.CSRC) if f.endswith(".hip"))
    headers = sorted(os.path.join(G.CSRC, f) for f in os.listdir(G.CSRC) if f.endswith(".hpp"))
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    G.OBJ_DIR = os.path.join(G.CSRC, "build", name)
    objects, _ = G._compile_objects(hipcc, units, headers, True, extra_flags=flags)
    out_dir = os.path.join(G.CSRC, "variants")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libsmrt_dort_%s.so" % name)
    G._run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objects, "-ldl"])
    print(lib)


if __name__ == "__main__":
    main()
