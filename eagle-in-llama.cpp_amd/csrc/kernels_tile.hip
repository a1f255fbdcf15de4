// kernels_tile.hip -- weight re-layout: ggml's row-major quant blocks <-> the tiled layout of tile_layout.h.
//
// Reference hook points: ggml_backend_cuda_buffer_init_tensor / _set_tensor (R/ggml/src/ggml-cuda/ggml-cuda.cu:543-585) are where
// the reference GPU backend touches weights as they arrive (it only pads).  Here a weight keeps ggml's layout until the first
// MUL_MAT that reads it (uploads may come in arbitrary chunks: llama-model-loader.cpp:1033-1056 sends 1 MiB pieces), then it is
// permuted once, in place, into 16-row x 1-unit tiles; get_tensor / cpy_tensor / partial writes see the original bytes again
// (mi_untile_*).  The permutation itself is tile_src() -- one statement shared with the kernels and the CPU unit test.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "tile_layout.h"
#include <mutex>

// one thread per 16-byte chunk of the TILED image.  fwd: tiled[chunk] <- raw (gather); !fwd: raw <- tiled[chunk] (scatter).
__global__ void __launch_bounds__(256) k_tile_permute(const char * __restrict__ src, char * __restrict__ dst, int type, int64_t n_chunks, int nun, int64_t row_bytes, int fwd) {
    const int ub = mi_unit_bytes(type);                 // chunks per tile = 16*ub/16 = ub
    for (int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (int64_t) gridDim.x * blockDim.x) {
        const int64_t tile = c / ub; const int cb = (int)(c - tile * ub) * 16;
        const int64_t g = tile / nun; const int u = (int)(tile - g * nun);
        const char * rbase = nullptr; (void) rbase;
        uint16_t v[8];
        if (!fwd) { const uint4 t = *(const uint4 *)(src + c * 16); __builtin_memcpy(v, &t, 16); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int n, sb; tile_src(type, cb + 2*i, n, sb);
            const int64_t ro = (g * 16 + n) * row_bytes + (int64_t) u * ub + sb;      // 2-byte aligned: every block size and offset is even
            if (fwd) v[i] = *(const uint16_t *)(src + ro);
            else     *(uint16_t *)(dst + ro) = v[i];
        }
        if (fwd) { uint4 t; __builtin_memcpy(&t, v, 16); *(uint4 *)(dst + c * 16) = t; }
    }
}

void mi_tile_permute(hipStream_t st, const void * src, void * dst, int type, int64_t rows, int64_t k, bool fwd) {
    const int ub = mi_unit_bytes(type);
    MI_ASSERT(ub > 0 && rows % 16 == 0 && k % 256 == 0);
    const int nun = (int)(k / 256);
    const int64_t row_bytes = (int64_t) nun * ub;
    const int64_t n_chunks = rows * row_bytes / 16;
    if (n_chunks == 0) return;
    int64_t grid = (n_chunks + 255) / 256; if (grid > 65536) grid = 65536;
    k_tile_permute<<<(int) grid, 256, 0, st>>>((const char *) src, (char *) dst, type, n_chunks, nun, row_bytes, fwd ? 1 : 0);
}
