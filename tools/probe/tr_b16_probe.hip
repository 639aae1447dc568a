// Development probe: what does ds_read_b64_tr_b16 return?  LDS holds v[i] = i (16-bit); every lane hands in a byte address and gets
// four 16-bit values back.  Three address patterns are dumped for lanes 0..63:
//   A  addr = 8 l                                   (lane l points at elements 4 l .. 4 l + 3)
//   B  addr = 32 ((l & 15) >> 2) + 8 (l & 3) + 128 (l >> 4)     (a row-major [4][16] block per 16-lane group, chunks in row-major lane order)
//   C  addr = 2 (l & 15) * 4 ... i.e. 8 (l & 15) + 512 (l >> 4) (sixteen contiguous chunks per group, groups 512 B apart)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probe/tr_b16_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_probe(int pattern, uint16_t* out)
{
    __shared__ __align__(16) uint16_t v[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) v[i] = (uint16_t)i;
    __syncthreads();
    const unsigned l = threadIdx.x;
    unsigned addr = 0;
    if (pattern == 0) addr = 8 * l;
    else if (pattern == 1) addr = 32 * ((l & 15) >> 2) + 8 * (l & 3) + 128 * (l >> 4);
    else addr = 8 * (l & 15) + 512 * (l >> 4);
    addr += (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)&v[0];
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[4 * l + 0] = (uint16_t)(r.x & 0xffff); out[4 * l + 1] = (uint16_t)(r.x >> 16);
    out[4 * l + 2] = (uint16_t)(r.y & 0xffff); out[4 * l + 3] = (uint16_t)(r.y >> 16);
}

int main()
{
    uint16_t* d; uint16_t h[256];
    hipMalloc(&d, sizeof(h));
    for (int p = 0; p < 3; p++) {
        k_probe<<<1, 64>>>(p, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %c\n", 'A' + p);
        for (int l = 0; l < 64; l++) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l % 4 == 3) ? "\n" : " |");
    }
    return 0;
}
