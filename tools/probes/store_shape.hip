// store_shape.hip -- what a 96-byte row piece costs the memory pipeline by the SHAPE of its stores (12-bit strip rows of k7_tiles:
// 8 lanes x 12 bytes; the alternative after a lane exchange: 6 lanes x 16 bytes, 2 lanes idle; and 10 / 14-bit rows: 8 + 2 / 12 + 2
// bytes per lane at 2-byte alignment).  Pure stores from registers, the tile kernel's geometry (256 threads, 8 workgroups per CU),
// `sc1 nt` like the product.   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_shape tools/probes/store_shape.hip && /tmp/store_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(uint8_t *out, size_t bytes_per_wave_pass, int passes)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t w = static_cast<size_t>(blockIdx.x) * 4u + wave;
    const uint32_t g = lane >> 3, k8 = lane & 7u; // eight groups of eight lanes: one row piece each
    u32x4 v = {lane, wave, blockIdx.x, 7u};
    for (int p = 0; p < passes; p++) {
        uint8_t *base = out + (w * passes + p) * bytes_per_wave_pass;
        v.x += p;
        if (MODE == 0) { // 16 bytes per lane: the plain mosaic (128 B per group)
            asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(base + g * 128u + k8 * 16u), "v"(v) : "memory");
        } else if (MODE == 1) { // 12 bytes per lane (96 B per group): the 12-bit strips today
            const u32x3 t = {v.x, v.y, v.z};
            asm volatile("global_store_dwordx3 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(base + g * 96u + k8 * 12u), "v"(t) : "memory");
        } else if (MODE == 2) { // the same 96 B per group as 6 lanes x 16 bytes, 2 lanes idle
            if (k8 < 6u)
                asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(base + g * 96u + k8 * 16u), "v"(v) : "memory");
        } else if (MODE == 3) { // 10-bit strips: 8 + 2 bytes per lane (80 B per group)
            const u32x2 t = {v.x, v.y};
            uint8_t *d = base + g * 80u + k8 * 10u;
            asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(d), "v"(t) : "memory"); // (plain, as the product's)
            asm volatile("global_store_short %0, %1, off\n\ts_nop 1" ::"v"(d + 8), "v"(v.z) : "memory");
        } else if (MODE == 4) { // the same 80 B per group as 5 lanes x 16 bytes
            if (k8 < 5u)
                asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(base + g * 80u + k8 * 16u), "v"(v) : "memory");
        } else if (MODE == 5) { // 14-bit strips: 12 + 2 bytes per lane (112 B per group)
            const u32x3 t = {v.x, v.y, v.z};
            uint8_t *d = base + g * 112u + k8 * 14u;
            asm volatile("global_store_dwordx3 %0, %1, off\n\ts_nop 1" ::"v"(d), "v"(t) : "memory");
            asm volatile("global_store_short %0, %1, off\n\ts_nop 1" ::"v"(d + 12), "v"(v.w) : "memory");
        } else { // the same 112 B per group as 7 lanes x 16 bytes
            if (k8 < 7u)
                asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(base + g * 112u + k8 * 16u), "v"(v) : "memory");
        }
    }
}

int main()
{
    const int passes = 8, nwg = 131072; // (4 GB of plain mosaic: beyond the Infinity Cache)
    const char *names[7] = {"16 B x 8 lanes (plain)", "12 B x 8 lanes (12-bit today)", "16 B x 6 lanes (12-bit, exchanged)", "8+2 B x 8 lanes (10-bit today)",
                            "16 B x 5 lanes (10-bit, exchanged)", "12+2 B x 8 lanes (14-bit today)", "16 B x 7 lanes (14-bit, exchanged)"};
    const size_t per[7] = {1024, 768, 768, 640, 640, 896, 896};
    uint8_t *buf = nullptr;
    const size_t cap = static_cast<size_t>(nwg) * 4 * passes * 1024 + 4096;
    if (hipMalloc(&buf, cap) != hipSuccess)
        return 1;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++)
        for (int m = 0; m < 7; m++) {
            float best = 1e9f;
            for (int it = 0; it < 6; it++) {
                hipEventRecord(a, 0);
                switch (m) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 0, 0, buf, per[m], passes); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 0, 0, buf, per[m], passes); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), 0, 0, buf, per[m], passes); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(nwg), dim3(256), 0, 0, buf, per[m], passes); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(nwg), dim3(256), 0, 0, buf, per[m], passes); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(nwg), dim3(256), 0, 0, buf, per[m], passes); break;
                default: hipLaunchKernelGGL(k<6>, dim3(nwg), dim3(256), 0, 0, buf, per[m], passes); break;
                }
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms = 0.f;
                hipEventElapsedTime(&ms, a, b);
                if (it >= 1 && ms < best)
                    best = ms;
            }
            const double bytes = static_cast<double>(nwg) * 4 * passes * per[m];
            if (rep == 1)
                std::printf("%-38s %7.3f ms  %6.2f TB/s  (%.2f GB)\n", names[m], best, bytes / (best * 1e-3) / 1e12, bytes / 1e9);
        }
    hipFree(buf);
    return 0;
}
