// Shared device/host helpers for libvq2 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/vq2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace vq2 {

// thread-local error message (forward runs on the main thread, backward on an autograd thread)
int set_error(int code, const char *fmt, ...);
int check_launch(const char *what);

// optional per-launch event profiling (vq2_prof_enable); see vq2_core.cpp
bool prof_enabled();
int prof_level();  // 0 off, 1 every instrumented launch, 2 only launches flagged `dominant`
const char *prof_label(const char *fmt, ...);
int prof_begin(const char *name, double flops, double bytes, hipStream_t s);
void prof_end(int id, hipStream_t s);
struct ProfScope {
    int id; hipStream_t s;
    ProfScope(const char *name, double flops, double bytes, hipStream_t st, bool dominant = false)
        : id((prof_level() == 1 || (prof_level() == 2 && dominant)) ? prof_begin(name, flops, bytes, st) : -1), s(st) {}
    ~ProfScope() { if (id >= 0) prof_end(id, s); }
};

static inline hipStream_t to_stream(vq2_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ReLU of freshly loaded values as ONE vector instruction each: fmaxf() on data straight from memory compiles to a
// canonicalising v_max_f32(x, x) plus v_max_f32(0, x) (and LLVM folds v_med3 back to that) -- and every vector-ALU
// instruction next to fp32 MFMAs costs matrix time.  On the bit pattern, max(int(x), 0) is the same function: negative
// floats (and -0) are negative integers, non-negative floats keep their bits.
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// optional ReLU without a select: floor = 0 (ReLU) or INT_MIN (identity)
__device__ __forceinline__ float relu_floor(float x, int floor_bits) { return __int_as_float(max(__float_as_int(x), floor_bits)); }
__device__ __forceinline__ float4 relu4(float4 v) {
    v.x = relu1(v.x); v.y = relu1(v.y); v.z = relu1(v.z); v.w = relu1(v.w);
    return v;
}

// Workgroups are dealt round-robin over the 8 XCDs (each with a private L2): linear id L runs on XCD
// L % 8.  Remap so that every XCD works on one CONTIGUOUS range of virtual ids -- neighbouring tiles
// (shared halo rows, shared operand panels) then hit the same L2.  Bijective for any total; only a
// speed choice, never correctness (MI355X_MICROARCH.md, workgroup dispatch).
__device__ __forceinline__ int xcd_remap(int L, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = L & 7, slot = L >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Allow > 64 KiB of dynamic LDS for a kernel (gfx950 has 160 KiB per CU).
template <typename K>
static inline void allow_big_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace vq2

#define VQ2_REQUIRE(cond, ...)                                           \
    do {                                                                 \
        if (!(cond)) return vq2::set_error(VQ2_ERR_INVALID, __VA_ARGS__); \
    } while (0)
