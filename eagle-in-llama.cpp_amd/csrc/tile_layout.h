// tile_layout.h -- the HBM layout of quantised weight matrices once the plugin has re-laid them out (SURVEY.md 8f-4;
// reference hook points: ggml_backend_cuda_buffer_init_tensor / set_tensor, R/ggml/src/ggml-cuda/ggml-cuda.cu:543-585).
//
// ggml stores a quantised matrix row by row, a row being k/256 "units" (K-quants: one 256-element super-block; Q8_0 / Q4_0:
// eight 32-element blocks).  The matrix-core kernel (kernels_mmq.hip) works on TILES of 16 rows x 1 unit and wants every
// wave-instruction to fetch 1 KiB of contiguous, 16-byte aligned memory.  The tiled layout is a pure byte permutation inside
// each group of 16 rows (same total size, so allocation sizes and offsets of the host do not change):
//     tensor = [row group g = 0 .. rows/16)[unit u = 0 .. k/256)[tile bytes]
// and a tile holds, piece by piece, exactly what the 64 lanes (n = lane & 15 -> row, kq = lane >> 4 -> k quarter) load:
//     Q4_K  2304 B: hdr[n] 16 B (d, dmin, scales) | qs piece ga = 0, 1: lane (n, kq) <- qs[64*(kq>>1) + 16*(kq&1) + 32*ga ..+16)
//     Q5_K  2816 B: hdr[n] | qh half hq = 0, 1: (n, hq) <- qh[16*hq ..+16) | qs pieces as Q4_K
//     Q6_K  3360 B: ql piece nn = 0, 1: lane <- ql[64*nn + 32*(kq>>1) + 16*(kq&1) ..+16) | qh: lane <- qh[16*kq ..+16) | scales[n] 16 B | d[n] 2 B
//     Q8_0  4352 B: q piece a = 0..3: lane <- block 2a + (kq>>1), quants [16*(kq&1) ..+16) | d[n][8] f16
//     Q4_0  2304 B: q piece a = 0..3: (n, kq>>1) <- block 2a + (kq>>1), all 16 nibble bytes | d[n][8] f16
// tile_src() is the single statement of that permutation: for an even byte offset b inside a tile it names the row and the byte
// offset inside that row's unit where the two bytes come from.  The re-layout kernels (kernels_tile.hip), their inverse
// (get_tensor) and the unit test (tests/test_tile_layout.py, which compiles this header for the host) all use it.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define MI_HD __host__ __device__
#else
#define MI_HD
#endif

// ggml type ids (ggml_abi.h); repeated here so that the header stands alone for the host-side unit test
#define MI_T_Q4_0 2
#define MI_T_Q8_0 8
#define MI_T_Q4_K 12
#define MI_T_Q5_K 13
#define MI_T_Q6_K 14

// bytes of one unit (256 elements) of one row
MI_HD static inline int mi_unit_bytes(int type) {
    switch (type) {
        case MI_T_Q4_K: return 144; case MI_T_Q5_K: return 176; case MI_T_Q6_K: return 210;
        case MI_T_Q8_0: return 272; case MI_T_Q4_0: return 144; default: return 0;
    }
}
MI_HD static inline int mi_tile_bytes(int type) { return 16 * mi_unit_bytes(type); }

// b: even byte offset inside a tile.  Returns the source: row n (0..15) of the group, byte offset sb inside that row's unit.
MI_HD static inline void tile_src(int type, int b, int & n, int & sb) {
    const int bb = b & 15;
    switch (type) {
        case MI_T_Q4_K: case MI_T_Q5_K: {
            if (b < 256) { n = b >> 4; sb = bb; return; }
            int o = b - 256, qs0 = 16;
            if (type == MI_T_Q5_K) {
                if (o < 512) { const int l = o >> 4; n = l & 15; sb = 16 + 16*(l >> 4) + bb; return; }
                o -= 512; qs0 = 48;
            }
            const int ga = o >> 10, l = (o & 1023) >> 4, kq = l >> 4; n = l & 15;
            sb = qs0 + 64*(kq >> 1) + 16*(kq & 1) + 32*ga + bb; return;
        }
        case MI_T_Q6_K: {
            if (b < 2048) { const int nn = b >> 10, l = (b & 1023) >> 4, kq = l >> 4; n = l & 15; sb = 64*nn + 32*(kq >> 1) + 16*(kq & 1) + bb; return; }
            if (b < 3072) { const int l = (b - 2048) >> 4, kq = l >> 4; n = l & 15; sb = 128 + 16*kq + bb; return; }
            if (b < 3328) { n = (b - 3072) >> 4; sb = 192 + bb; return; }
            n = (b - 3328) >> 1; sb = 208; return;
        }
        case MI_T_Q8_0: {
            if (b < 4096) { const int a = b >> 10, l = (b & 1023) >> 4, kq = l >> 4; n = l & 15; sb = (2*a + (kq >> 1))*34 + 2 + 16*(kq & 1) + bb; return; }
            const int o = b - 4096; n = o >> 4; sb = 34*((o & 15) >> 1); return;
        }
        case MI_T_Q4_0: {
            if (b < 2048) { const int a = b >> 9, l = (b & 511) >> 4, hk = l >> 4; n = l & 15; sb = (2*a + hk)*18 + 2 + bb; return; }
            const int o = b - 2048; n = o >> 4; sb = 18*((o & 15) >> 1); return;
        }
        default: n = 0; sb = 0; return;
    }
}
