// Checks the lane maps assumed by k_mmq (kernels_mmq.hip) for v_mfma_i32_16x16x64_i8 with exact integer data:
//   A[i][k]: lane = i + 16*kq, byte j of the 16-byte operand <-> k = 16kq + j (any k map works as long as A and B share it
//            and lanes with the same kq cover the same k range)
//   B[k][n]: lane = n + 16*kq
//   C[i][n]: lane = n + 16*g, reg r <-> i = 4g + r
// build: hipcc --offload-arch=gfx950 scripts/probe_mfma.hip -o eagle-in-llama.cpp_amd/lib/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const i32x4 * a, const i32x4 * b, i32x4 * c) {
    const i32x4 z = {0, 0, 0, 0};
    c[threadIdx.x] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], z, 0, 0, 0);
}
#define CK(x) do { if ((x) != hipSuccess) { printf("PROBE hip error at %s\n", #x); return 2; } } while (0)
int main() {
    static int8_t A[16][64], B[64][16]; static int ref[16][16];
    srand(1);
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 64; ++kk) { A[i][kk] = (int8_t)(rand() % 255 - 127); B[kk][i] = (int8_t)(rand() % 255 - 127); }
    // the masking used by the kernel: M rows 0..7 only see k-slot groups 0,1; rows 8..15 only groups 2,3
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 64; ++kk) if ((i >> 3) != (kk >> 5)) A[i][kk] = 0;
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { int s = 0; for (int kk = 0; kk < 64; ++kk) s += A[i][kk] * B[kk][n]; ref[i][n] = s; }
    static int8_t ha[64][16], hb[64][16];
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) { const int kq = l >> 4, r = l & 15; ha[l][j] = A[r][16*kq + j]; hb[l][j] = B[16*kq + j][r]; }
    void * da, * db, * dc; CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dc, 1024));
    CK(hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice));
    k<<<1, 64>>>((const i32x4 *) da, (const i32x4 *) db, (i32x4 *) dc);
    static int hc[64][4]; CK(hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { const int n = l & 15, i = 4*(l >> 4) + r; if (hc[l][r] != ref[i][n]) ++bad; }
    printf("PROBE mfma_i32_16x16x64_i8 layout: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    if (bad) for (int l = 0; l < 64; l += 16) for (int r = 0; r < 4; ++r) { for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) if (ref[i][n] == hc[l][r]) printf("lane %d reg %d = C[%d][%d]\n", l, r, i, n); }
    return bad ? 1 : 0;
}
