// probe_icache.hip -- does once-through straight-line code run at instruction-fetch speed?  Two kernels execute the same number of
// independent VALU instructions per wave: `line` as N straight-line instructions, `loop` as a 64-instruction body N/64 times.  Per wave
// the s_memtime delta is stored; launches alternate (line, loop, other, ...) to see whether the instruction cache survives a launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))
#define R1024(x) R4(R256(x))
#define BODY "v_add_f32 %0, %0, %1\n\tv_mul_f32 %2, %2, %1\n\tv_add_f32 %3, %3, %1\n\tv_mul_f32 %4, %4, %1\n\t"
template <int REP> __global__ void __launch_bounds__(1024) k_line(float * out, unsigned long long * cyc, float s) {
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (REP == 1) { asm volatile(R256(BODY) : "+v"(a), "+v"(s), "+v"(b), "+v"(c), "+v"(d)); }            // 1024 instructions
    else          { asm volatile(R1024(BODY) : "+v"(a), "+v"(s), "+v"(b), "+v"(c), "+v"(d)); }           // 4096 instructions
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = a + b + c + d;
}
template <int N> __global__ void __launch_bounds__(1024) k_loop(float * out, unsigned long long * cyc, float s) {
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 64; ++i) { asm volatile(R16(BODY) : "+v"(a), "+v"(s), "+v"(b), "+v"(c), "+v"(d)); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = a + b + c + d;
}
__global__ void k_other(float * out) { out[blockIdx.x * blockDim.x + threadIdx.x] += 1.f; }
int main() {
    float * out; unsigned long long * cyc; const int nb = 256;
    CK(hipMalloc((void **) &out, nb * 1024 * 4)); CK(hipMalloc((void **) &cyc, nb * 16 * 8));
    std::vector<unsigned long long> h(nb * 16);
    auto report = [&](const char * name) {
        hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%-28s cycles per wave: min %6llu  median %6llu  max %6llu\n", name, h[0], h[h.size()/2], h.back());
    };
    for (int it = 0; it < 3; ++it) {
        k_line<1><<<nb, 1024>>>(out, cyc, 1.0001f); report("line 1024 instr");
        k_loop<1024><<<nb, 1024>>>(out, cyc, 1.0001f); report("loop 1024 instr");
        k_line<4><<<nb, 1024>>>(out, cyc, 1.0001f); report("line 4096 instr");
        k_loop<4096><<<nb, 1024>>>(out, cyc, 1.0001f); report("loop 4096 instr");
        k_line<1><<<nb, 1024>>>(out, cyc, 1.0001f); report("line 1024 again (after others)");
        k_line<1><<<nb, 1024>>>(out, cyc, 1.0001f); report("line 1024 back to back");
        k_other<<<nb, 256>>>(out);
    }
    // one wave per SIMD instead of four: is it issue or fetch?
    k_line<1><<<nb, 256>>>(out, cyc, 1.0001f); { hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost); std::vector<unsigned long long> v; for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) v.push_back(h[b*16+w]); std::sort(v.begin(), v.end()); printf("line 1024, 4 waves per block: min %llu median %llu max %llu\n", v[0], v[v.size()/2], v.back()); }
    k_loop<1024><<<nb, 256>>>(out, cyc, 1.0001f); { hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost); std::vector<unsigned long long> v; for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) v.push_back(h[b*16+w]); std::sort(v.begin(), v.end()); printf("loop 1024, 4 waves per block: min %llu median %llu max %llu\n", v[0], v[v.size()/2], v.back()); }
    return 0;
}
