// Shared device helpers for the gfx950 kernels.  CDNA4 only: 64-wide wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "regtr_hip.h"      // the public C ABI: every definition below is checked against its declaration at compile time
#include "regtr_hip_experimental.h"      // ... and the opt-in experiment entry points (outside the ABI version)

#define RG_WAVE 64

// status codes of the C-ABI (include/regtr_hip.h)
#define RG_OK 0
#define RG_ERR_LAUNCH (-1)
#define RG_ERR_ARG (-2)
#define RG_ERR_WORKSPACE (-3)

#define RG_RETURN_IF_LAUNCH_FAILED()                      \
    do {                                                  \
        if (hipGetLastError() != hipSuccess) return RG_ERR_LAUNCH; \
    } while (0)

// A kernel launched with more dynamic LDS than the default limit: raise the kernel's limit on the CURRENT device, once per device
// and kernel (a process-wide flag would leave a second GPU of the process at the default), thread-safe.  false = the device refused.
template <auto Kernel>
static inline bool rg_allow_dynamic_lds(size_t bytes)
{
    static unsigned long long ok_mask = 0;       // bit d: the attribute has been set on device d (devices >= 64: set every time)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev < 64 && (__atomic_load_n(&ok_mask, __ATOMIC_ACQUIRE) >> dev & 1)) return true;
    if (hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    if (dev < 64) __atomic_fetch_or(&ok_mask, 1ull << dev, __ATOMIC_RELEASE);
    return true;
}

static inline size_t rg_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int rg_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int rg_xcd_grid(int nblocks) { return (nblocks + 7) / 8 * 8; }
static inline unsigned rg_next_pow2(unsigned x)
{
    unsigned p = 1;
    while (p < x) p <<= 1;
    return p;
}

// carve a sub-buffer out of a caller-provided workspace (256-B aligned)
struct RgCarver {
    char* base;
    size_t off;
    size_t cap;
    explicit RgCarver(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes) {}
    template <typename T>
    T* take(size_t n)
    {
        off = rg_align_up(off, 256);
        T* p = (T*)(base + off);
        off += n * sizeof(T);
        return p;
    }
    bool ok() const { return off <= cap; }
};

__device__ __forceinline__ int rg_lane() { return threadIdx.x & (RG_WAVE - 1); }

// Workgroup b of a launch runs on XCD b % 8, and every XCD has its own L2.  Kernels whose neighbouring work items touch
// overlapping memory (queries gathering the rows of their spatial neighbours) want neighbouring items on ONE L2: launch a
// grid padded to a multiple of 8 (rg_xcd_grid) and take work block rg_xcd_block(blockIdx.x, gridDim.x) -- every XCD then owns
// one contiguous eighth of the blocks instead of every eighth block.
__device__ __forceinline__ int rg_xcd_block(int b, int grid_padded) { return (b & 7) * (grid_padded >> 3) + (b >> 3); }

// index of the segment containing row i, given exclusive offsets off[0..nseg] (off[0] = 0)
__device__ __forceinline__ int rg_find_segment(const int* __restrict__ off, int nseg, int i)
{
    int lo = 0, hi = nseg;  // invariant: off[lo] <= i < off[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// The same for a WAVE-UNIFORM row i, by the whole wave at once: lanes read 64 boundaries per step (one coalesced load, one
// memory round trip) and count those <= i with a ballot -- instead of log2(nseg) DEPENDENT loads per lane.
__device__ __forceinline__ int rg_find_segment_wave(const int* __restrict__ off, int nseg, int i)
{
    int seg = 0;
    for (int base = 0; base < nseg; base += RG_WAVE) {      // boundaries off[1 .. nseg - 1]; off[b] <= i  =>  segment >= b
        const int b = base + rg_lane() + 1;
        const bool le = b < nseg && off[b] <= i;
        seg += __popcll(__ballot(le));
    }
    return seg;
}

__device__ __forceinline__ uint32_t rg_hash64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return (uint32_t)k;
}

// float <-> int with the same ordering as the floats (for atomicMin/atomicMax on floats)
__device__ __forceinline__ int rg_f2ord(float f)
{
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float rg_ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__device__ __forceinline__ float rg_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RG_WAVE);
    return v;
}
__device__ __forceinline__ double rg_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RG_WAVE);
    return v;
}
__device__ __forceinline__ float rg_wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, RG_WAVE));
    return v;
}
