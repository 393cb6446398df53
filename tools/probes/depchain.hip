// Probe: what a dependent chain of vector instructions costs one wave (s_memtime ticks per instruction), alone on its SIMD and beside
// other waves.  hipcc --offload-arch=gfx950 -O3 tools/probes/depchain.hip -o /tmp/depchain && /tmp/depchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void chain(uint32_t *out, uint64_t *ticks, int reps, int mode)
{
    uint32_t r = 0x0C0C0C00u + (threadIdx.x & 7u), lo = 0x01020301u + out[0], hi = 0x02010302u + out[1];
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) { // min, perm, add: dependent
        for (int i = 0; i < reps; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t s = __builtin_amdgcn_perm(hi, lo, min(r, 0x0C0C0C0Cu));
                r += s;
                r = (r & 0x0C0C0C07u); // keep it inside (one more dependent op)
            }
        }
    } else if (mode == 1) { // dependent adds only
        for (int i = 0; i < reps; i++) {
#pragma unroll
            for (int k = 0; k < 64; k++)
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(lo));
        }
    } else if (mode == 2) { // independent adds (4 chains)
        uint32_t a = r, b = lo, c = hi, d = acc;
        for (int i = 0; i < reps; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(lo));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(lo));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(lo));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(lo));
            }
        }
        r = a + b + c + d;
    } else if (mode == 3) { // dependent v_perm only
        for (int i = 0; i < reps; i++) {
#pragma unroll
            for (int k = 0; k < 64; k++)
                asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(r) : "v"(hi), "v"(lo));
        }
    } else if (mode == 4) { // dependent ds_read_u8 chain
        __shared__ uint8_t tab[4096];
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = (uint8_t)(i * 7 + 3);
        __syncthreads();
        uint32_t p = threadIdx.x;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < reps; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++)
                p = (p + tab[p & 4095u]) & 4095u;
        }
        r = p;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = r + acc;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
int main()
{
    uint32_t *out; uint64_t *ticks;
    hipMalloc(&out, 4 * (2 + 4096 * 1024)); hipMemset(out, 0, 4 * (2 + 4096 * 1024));
    hipMalloc(&ticks, 8 * 65536);
    const int reps = 64;
    const char *names[] = {"min+perm+add+and (4 dependent)", "v_add dependent", "v_add 4 independent chains", "v_perm dependent", "ds_read_u8 chain (+2 valu)"};
    const int per[] = {16 * 4, 64, 64, 64, 16};
    for (int mode = 0; mode < 5; mode++)
        for (int cfg = 0; cfg < 4; cfg++) {
            const int blocks = cfg == 3 ? 256 * 8 : 256, threads = cfg == 0 ? 64 : cfg == 1 ? 256 : 1024;  // waves per CU: 1, 4, 16, 16 x 8 blocks
            chain<<<blocks, threads>>>(out, ticks, reps, mode);
            hipDeviceSynchronize();
            uint64_t h[16]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-34s threads/block %4d blocks %4d: %.1f ticks per instruction (wave 0)\n", names[mode], threads, blocks, (double)h[0] / (reps * per[mode]));
        }
    return 0;
}
