// Pure memory-side rate of the per-unit load patterns of k_mmq (no compute): blocks of 16 waves own 16 rows; wave w reads the
// super-blocks w, w+16, ... of those rows with exactly the loads mq_frag<TYPE>::load issues.
//   pattern 0: Q4_K   (144-B blocks: hdr 16 B (all 4 kq lanes the same) + 2 x 16 B of qs per lane)
//   pattern 1: Q6_K   (210-B blocks: 2 x ql, 2 x qh (duplicated over qb), sc (x4 duplicated), d 2 B)
//   pattern 2: Q6_K', no duplicates: ql x2, qh once (16 B per lane, 64 B per row), sc|d once in lane group 0
//   pattern 3: Q6_K with dword-aligned loads (5 dwords + funnel shift)
// build: hipcc -O3 --offload-arch=gfx950 scripts/probe_pattern.hip -o eagle-in-llama.cpp_amd/lib/probe_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 ld16(const char * p) { i32x4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ i32x4 ld16_a2(const char * p) {
    const int mis = (int)((uintptr_t) p & 2); const char * q = p - mis; const int sh = mis * 8;
    const i32x4 v = ld16(q); int e = 0; if (sh) __builtin_memcpy(&e, q + 16, 4);
    i32x4 r; r.x = __builtin_amdgcn_alignbit(v.y, v.x, sh); r.y = __builtin_amdgcn_alignbit(v.z, v.y, sh); r.z = __builtin_amdgcn_alignbit(v.w, v.z, sh); r.w = __builtin_amdgcn_alignbit(e, v.w, sh);
    return r;
}
template <int P, int NW> __global__ void __launch_bounds__(NW*64) k(const char * base, int nsb, int row_bytes, int * sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
    const char * row = base + ((size_t) blockIdx.x * 16 + n) * row_bytes;
    i32x4 acc = {0, 0, 0, 0};
    for (int sb = wave; sb < nsb; sb += NW) {
        if (P == 0) {
            const char * b = row + sb*144;
            acc ^= ld16(b); acc ^= ld16(b + 16 + 64*(kq >> 1) + 16*(kq & 1)); acc ^= ld16(b + 16 + 64*(kq >> 1) + 32 + 16*(kq & 1));
        } else if (P == 1) {
            const char * b = row + sb*210;
            for (int nn = 0; nn < 2; ++nn) { acc ^= ld16(b + 64*nn + 32*(kq >> 1) + 16*(kq & 1)); acc ^= ld16(b + 128 + 32*nn + 16*(kq & 1)); }
            acc ^= ld16(b + 192); uint16_t d; __builtin_memcpy(&d, b + 208, 2); acc.x ^= d;
        } else if (P == 2) {
            const char * b = row + sb*210;
            for (int nn = 0; nn < 2; ++nn) acc ^= ld16(b + 64*nn + 32*(kq >> 1) + 16*(kq & 1));
            acc ^= ld16(b + 128 + 16*kq);
            if (kq == 0) { acc ^= ld16(b + 192); uint16_t d; __builtin_memcpy(&d, b + 208, 2); acc.x ^= d; }
        } else {
            const char * b = row + sb*210;
            for (int nn = 0; nn < 2; ++nn) { acc ^= ld16_a2(b + 64*nn + 32*(kq >> 1) + 16*(kq & 1)); acc ^= ld16_a2(b + 128 + 32*nn + 16*(kq & 1)); }
            acc ^= ld16_a2(b + 192); uint16_t d; __builtin_memcpy(&d, b + 208, 2); acc.x ^= d;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) sink[0] = 1;
}
template <int P, int NW> static void run(const char * name, char * buf, int * sink, int rows, int nsb, int blk) {
    const int row_bytes = nsb * blk; hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        // rotate through > 256 MiB so nothing stays in the Infinity Cache
        char * base = buf + (size_t)(it % 4) * ((size_t) 320 << 20);
        k<P, NW><<<rows/16, NW*64>>>(base, nsb, row_bytes, sink);
        (void) hipEventRecord(e0); k<P, NW><<<rows/16, NW*64>>>(base + ((size_t) 160 << 20), nsb, row_bytes, sink); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = (double) rows * row_bytes;
    printf("%-34s rows %6d nsb %3d NW %2d: %7.2f us  %7.1f GB/s\n", name, rows, nsb, NW, best*1e3, bytes/best/1e6);
}
int main() {
    char * buf; int * sink;
    if (hipMalloc(&buf, ((size_t) 1400 << 20)) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 2; }
    (void) hipMemset(buf, 1, ((size_t) 1400 << 20));
    run<0, 16>("Q4_K pattern", buf, sink, 4096, 43, 144);   run<1, 16>("Q6_K pattern", buf, sink, 4096, 43, 210);
    run<2, 16>("Q6_K no-duplicate pattern", buf, sink, 4096, 43, 210); run<3, 16>("Q6_K dword-aligned pattern", buf, sink, 4096, 43, 210);
    run<0, 8>("Q4_K pattern", buf, sink, 32000, 16, 144);   run<1, 8>("Q6_K pattern", buf, sink, 32000, 16, 210);
    run<2, 8>("Q6_K no-duplicate pattern", buf, sink, 32000, 16, 210); run<3, 8>("Q6_K dword-aligned pattern", buf, sink, 32000, 16, 210);
    run<0, 8>("Q4_K pattern", buf, sink, 4096, 16, 144);    run<1, 8>("Q6_K pattern", buf, sink, 4096, 16, 210);
    return 0;
}
