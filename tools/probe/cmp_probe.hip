// Issue rate of the rank-sort inner step of the radius search (csrc/preprocess.hip: for_each_ranked) on gfx950: one wave per SIMD slot runs
// `iters` x 32 dependent-free compare + add steps against 32 register-resident keys.
//   mode 0: 64-bit unsigned compare (v_cmp_lt_u64) -- the (d2 | index) keys as they are
//   mode 1: 32-bit unsigned compare (v_cmp_lt_u32)
//   mode 2: 32-bit float compare (v_cmp_lt_f32)
//   mode 3: two 32-bit compares per key: hi <, or hi == and lo <   (the u64 order from 32-bit pieces)
//   mode 4: 64-bit float compare (v_cmp_lt_f64)
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libcmp.so cmp_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__global__ void __launch_bounds__(256) k_cmp(const uint64_t* __restrict__ keys, int* out, int iters)
{
    uint64_t k[32];
    for (int i = 0; i < 32; i++) k[i] = keys[i];          // wave-uniform loads -> SGPRs or VGPRs, either way no memory in the loop
    uint64_t mine = keys[32 + (threadIdx.x & 63)];
    int rank = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            if (MODE == 0) rank += k[i] < mine ? 1 : 0;
            if (MODE == 1) rank += (uint32_t)k[i] < (uint32_t)mine ? 1 : 0;
            if (MODE == 2) rank += __uint_as_float((uint32_t)k[i]) < __uint_as_float((uint32_t)mine) ? 1 : 0;
            if (MODE == 3) {
                const uint32_t hi = (uint32_t)(k[i] >> 32), lo = (uint32_t)k[i], mh = (uint32_t)(mine >> 32), ml = (uint32_t)mine;
                rank += (hi < mh) | ((hi == mh) & (lo < ml));
            }
            if (MODE == 4) rank += __longlong_as_double((long long)k[i]) < __longlong_as_double((long long)mine) ? 1 : 0;
        }
        mine += (uint64_t)rank << 20;                          // keeps the loop from being hoisted
    }
    out[blockIdx.x * 256 + threadIdx.x] = rank;
}

extern "C" void cmp_probe(const uint64_t* keys, int* out, int mode, int iters, int blocks, void* st)
{
    hipStream_t s = (hipStream_t)st;
    switch (mode) {
    case 0: k_cmp<0><<<blocks, 256, 0, s>>>(keys, out, iters); break;
    case 1: k_cmp<1><<<blocks, 256, 0, s>>>(keys, out, iters); break;
    case 2: k_cmp<2><<<blocks, 256, 0, s>>>(keys, out, iters); break;
    case 3: k_cmp<3><<<blocks, 256, 0, s>>>(keys, out, iters); break;
    default: k_cmp<4><<<blocks, 256, 0, s>>>(keys, out, iters); break;
    }
}
