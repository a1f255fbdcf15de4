// Streaming-read rate of 16-byte-per-lane global loads at byte offsets 0 / 2 / 4 / 8 from 16-byte alignment, in two lane
// layouts: "row" (64 lanes x 16 B contiguous = what k_mmvq does) and "mfma" (16 rows x 4 lanes x 16 B, rows `stride` apart =
// what k_mmq does).  build: hipcc -O3 --offload-arch=gfx950 scripts/probe_unaligned.hip -o eagle-in-llama.cpp_amd/lib/probe_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 ld16(const void * p) { i32x4 v; __builtin_memcpy(&v, p, 16); return v; }
// every wave reads `per_wave` bytes; layout 0: contiguous 1 KB per instruction; layout 1: 16 rows x 64 B, row stride `stride`
__global__ void __launch_bounds__(512) k(const char * base, size_t per_wave, int off, int layout, int stride, int * sink) {
    const int lane = threadIdx.x & 63;
    const size_t wid = (size_t) blockIdx.x * 8 + (threadIdx.x >> 6);
    const char * p = base + wid * per_wave + off;
    i32x4 acc = {0, 0, 0, 0};
    if (layout == 0) {
        for (size_t o = 0; o + 4096 <= per_wave; o += 4096) {
            const i32x4 a = ld16(p + o + lane*16), b = ld16(p + o + 1024 + lane*16), c = ld16(p + o + 2048 + lane*16), d = ld16(p + o + 3072 + lane*16);
            acc += a ^ b ^ c ^ d;
        }
    } else {
        // the wave owns 16 rows of `stride` bytes each (per_wave = 16*stride); it walks them 64 B at a time
        const char * r = p + (size_t)(lane & 15) * stride + (lane >> 4) * 16;
        for (int o = 0; o + 256 <= stride; o += 256) {
            const i32x4 a = ld16(r + o), b = ld16(r + o + 64), c = ld16(r + o + 128), d = ld16(r + o + 192);
            acc += a ^ b ^ c ^ d;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) sink[0] = 1;
}
int main() {
    const size_t total = (size_t) 1 << 30;
    char * buf; int * sink;
    if (hipMalloc(&buf, total + 4096) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 2; }
    (void) hipMemset(buf, 1, total + 4096);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    const int grid = 512, waves = grid * 8;
    for (int layout = 0; layout < 2; ++layout) for (int stride : {8192, 9030, 9028, 9032, 6192, 3360}) for (int off : {0, 2, 4, 8}) {
        if (layout == 0 && stride != 8192) continue;
        const size_t per_wave = layout == 0 ? ((total / waves) & ~(size_t) 4095) : (size_t) 16 * stride;
        const int g = layout == 0 ? grid : (int) (total / per_wave / 8);
        k<<<g, 512>>>(buf, per_wave, off, layout, stride, sink);
        (void) hipEventRecord(e0);
        k<<<g, 512>>>(buf, per_wave, off, layout, stride, sink);
        (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1);
        const double bytes = layout == 0 ? (double) per_wave * g * 8 : (double) g * 8 * 16 * (stride / 256) * 256;
        printf("layout %s stride %5d offset %d: %.1f GB/s\n", layout ? "mfma(16 rows x 64 B)" : "row (1 KB contiguous)", stride, off, bytes / ms / 1e6);
    }
    return 0;
}
