// Does the f32-input MFMA (v_mfma_f32_16x16x4_f32) overlap with VALU work on gfx950, or do they share the FP32 ALUs?
// One workgroup of 8 waves per CU (2 waves per SIMD: waves w and w + 4).  Modes:
//   0  every wave: MFMA only (4 independent accumulators)       1  every wave: VALU FMAs only
//   2  every wave: both, interleaved                             3  waves 0-3 MFMA only, waves 4-7 VALU only (one of each per SIMD)
//   4  bf16 MFMA (v_mfma_f32_16x16x32_bf16) only                 5  waves 0-3 bf16 MFMA, waves 4-7 VALU
//   6 / 7 / 8  as 3 / 3 / 5 with only the f32-MFMA / VALU / bf16-MFMA waves running (the other four exit): the single-tenant times
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512) k_alu(float* out, int iters)
{
    const int wave = threadIdx.x >> 6;
    if ((MODE == 6 && wave >= 4) || (MODE == 7 && wave < 4) || (MODE == 8 && wave >= 4)) return;   // half the waves idle
    const bool do_mfma = MODE == 0 || MODE == 2 || ((MODE == 3 || MODE == 6) && wave < 4);
    const bool do_bf = MODE == 4 || ((MODE == 5 || MODE == 8) && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 2 || ((MODE == 3 || MODE == 5 || MODE == 7) && wave >= 4);
    f4 acc[4] = {};
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 1e-3f + i;
    const float a = threadIdx.x * 1e-4f, b = 1.0001f;
    b8 ab, bb;
    for (int i = 0; i < 8; i++) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b + i); }
    for (int it = 0; it < iters; it++) {
        if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
        if (do_bf) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[j], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = __builtin_fmaf(v[i], b, a);
        }
    }
    float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

extern "C" void alu_probe(float* out, int mode, int iters, int blocks, void* st)
{
    hipStream_t s = (hipStream_t)st;
    switch (mode) {
    case 0: k_alu<0><<<blocks, 512, 0, s>>>(out, iters); break;
    case 1: k_alu<1><<<blocks, 512, 0, s>>>(out, iters); break;
    case 2: k_alu<2><<<blocks, 512, 0, s>>>(out, iters); break;
    case 3: k_alu<3><<<blocks, 512, 0, s>>>(out, iters); break;
    case 4: k_alu<4><<<blocks, 512, 0, s>>>(out, iters); break;
    case 5: k_alu<5><<<blocks, 512, 0, s>>>(out, iters); break;
    case 6: k_alu<6><<<blocks, 512, 0, s>>>(out, iters); break;
    case 7: k_alu<7><<<blocks, 512, 0, s>>>(out, iters); break;
    default: k_alu<8><<<blocks, 512, 0, s>>>(out, iters); break;
    }
}
