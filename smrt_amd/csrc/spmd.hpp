// SPMD primitives used by the DORT device code.
//
// On the GPU (hipcc, gfx950) these map 1:1 onto the CDNA4 execution model: a workgroup of NT threads made of
// 64-lane wavefronts, LDS, s_barrier, and DPP/bpermute cross-lane moves.
//
// With -DSMRT_HOST_EMU (g++, tests only) the same source runs under a deterministic fiber emulator
// (tests/hostemu/emu_runtime.hpp): every "thread" is a ucontext fiber, barriers and shuffles are the only yield
// points.  That build exists so that kernel logic can be unit-tested and race-checked (fibers scheduled in
// forward and reverse order must give identical results) in a container without a GPU.  The product never
// loads it.
#pragma once

#if defined(SMRT_HOST_EMU)

#include <cmath>
#include "emu_runtime.hpp"
#define SMRT_DEV inline
#define SMRT_LANES 64
namespace smrt {
SMRT_DEV int tid() { return emu::tid(); }
SMRT_DEV void block_sync() { emu::block_barrier(); }
SMRT_DEV void wave_sync() { emu::wave_barrier(); }
SMRT_DEV double shfl_xor(double v, int mask) { return emu::shfl_xor(v, mask); }
SMRT_DEV int shfl_xor(int v, int mask) { return (int)emu::shfl_xor((double)v, mask); }
SMRT_DEV long long cycle_counter() { return 0; }
SMRT_DEV void lds_or(int* p, int v) { *p |= v; }
SMRT_DEV void lds_max(int* p, int v) { if (v > *p) *p = v; }
}  // namespace smrt

#else

#include <hip/hip_runtime.h>
#define SMRT_DEV __device__ __forceinline__
#define SMRT_LANES 64
namespace smrt {
SMRT_DEV int tid() { return threadIdx.x; }
SMRT_DEV void block_sync() { __syncthreads(); }
// All lanes of a wavefront execute in lockstep and their LDS operations are issued in program order; this only
// stops the compiler from moving LDS traffic across the point (and is a real rendezvous in the emulator).
SMRT_DEV void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
SMRT_DEV double shfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }
SMRT_DEV int shfl_xor(int v, int mask) { return __shfl_xor(v, mask, 64); }
SMRT_DEV long long cycle_counter() { return (long long)clock64(); }
SMRT_DEV void lds_or(int* p, int v) { atomicOr(p, v); }
SMRT_DEV void lds_max(int* p, int v) { atomicMax(p, v); }
}  // namespace smrt

#endif
