// textcnn_b16.h -- pieces of the bf16-plane token-product GEMM shared by textcnn_prod_b16.hip (the GEMM) and
// textcnn_prod.hip (whose prepare stage writes the weight-plane image in the same launch as the token list).
#pragma once

#include "rbr_common.h"

namespace rbr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kB16Waves = 4, kB16Stages = 4;
constexpr int kB16BM = 32 * kB16Waves, kB16BN = 128, kB16KC = 16, kB16Threads = 64 * kB16Waves;
constexpr int kB16ABytes = kB16BM * kB16KC * 4;                 // 8 KiB
constexpr int kB16BFrag = 1024;                                 // one (tile, plane) fragment: 64 lanes x 16 B
constexpr int kB16BBytes = (kB16BN / 32) * 3 * kB16BFrag;       // 12 KiB
constexpr int kB16Lds = kB16Stages * (kB16ABytes + kB16BBytes); // 80 KiB: two workgroups per CU
static_assert(kB16Waves == 4 && kB16Stages == 4, "the fill schedule below (5 LDS-DMA instructions per wave and stage) assumes 4 x 4");
// "bf16 storage" form of the GEMM (RBR_PROD_BF16 only; textcnn_prod_b16.hip: prod_gemm_b16s_kernel): the token rows come from a
// compact bf16 copy [list rows][kB16sDp(D)] made once per forward, the product table T is written in bf16, a stage is 32 deep
constexpr int kB16sKC = 32;
constexpr int kB16sABytes = kB16BM * kB16sKC * 2;               // 8 KiB: [128 rows][4 x 16 B]
constexpr int kB16sBBytes = 2 * (kB16BN / 32) * kB16BFrag;      // 8 KiB: [2 k-halves][4 tiles] hi-plane fragments
constexpr int kB16sLds = kB16Stages * (kB16sABytes + kB16sBBytes);   // 64 KiB
inline int b16s_dp(int D) { return (D + kB16sKC - 1) / kB16sKC * kB16sKC; }      // row length of the compact bf16 copy

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);       // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, v);
}

// (x0, x1) -> packed bf16 pairs of the three planes; the residues x - hi and (x - hi) - mid are exact in f32
template <int NPLANES>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pack_bf16(x0, x1);
    if (NPLANES >= 2) {
        const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
        mid = pack_bf16(r0, r1);
        if (NPLANES >= 3) {
            const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
            lo = pack_bf16(s0, s1);
        }
    }
}

// Weight planes of the token-product GEMM in fragment order (see the header); element (lane l, j) of fragment
// (group, step, tile, plane) is plane(Wprod[k = step*16 + 8*(l >> 5) + j][column = group*128 + tile*32 + (l & 31)]).
struct B16Pack {
    int n_widths, D, cp_real, ngroups, nchunks;
    int kz[RBR_MAX_WIDTHS], ch[RBR_MAX_WIDTHS], poff[RBR_MAX_WIDTHS];
};

__device__ __forceinline__ float b16_prod_weight(const B16Pack& J, const PtrArray& W, int pc, int d) {
    int w = 0;
#pragma unroll
    for (int k = 1; k < RBR_MAX_WIDTHS; ++k)
        if (k < J.n_widths && pc >= J.poff[k]) w = k;
    const int rel = pc - J.poff[w];
    const int j = rel / J.ch[w], cl = rel - j * J.ch[w];
    return W.p[w][((long)cl * J.D + d) * J.kz[w] + j];
}

// one work item = the three plane fragments' 16 bytes of one lane: b16_pack_items(J) items in all
__device__ __forceinline__ long b16_pack_items(const B16Pack& J) { return (long)J.ngroups * J.nchunks * 4 * 64; }

__device__ __forceinline__ void b16_pack_item(const B16Pack& J, const PtrArray& W, unsigned char* __restrict__ bimg, long idx) {
    {
        long r = idx;
        const int l = (int)(r & 63); r >>= 6;
        const int t = (int)(r & 3); r >>= 2;
        const int c = (int)(r % J.nchunks);
        const int ng = (int)(r / J.nchunks);
        const int col = ng * kB16BN + t * 32 + (l & 31);
        const int k0 = c * kB16KC + 8 * (l >> 5);
        u32x4 ph, pm, pl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = k0 + 2 * q + e;
                x[e] = (col < J.cp_real && k < J.D) ? b16_prod_weight(J, W, col, k) : 0.f;
            }
            unsigned a, b, cc;
            split_pair<3>(x[0], x[1], a, b, cc);
            ph[q] = a; pm[q] = b; pl[q] = cc;
        }
        unsigned char* dst = bimg + (((size_t)ng * J.nchunks + c) * 4 + t) * 3 * kB16BFrag + l * 16;
        *reinterpret_cast<u32x4*>(dst) = ph;
        *reinterpret_cast<u32x4*>(dst + kB16BFrag) = pm;
        *reinterpret_cast<u32x4*>(dst + 2 * kB16BFrag) = pl;
    }
}

// The backward's row GEMM (textcnn_prod_b16.hip: prod_b16_rows_gemm) multiplies G [rows, KG] by Wprod^T [KG, D]: its weight image
// holds the planes of WT (the row-major Wprod^T the forward's pack launch left in the workspace) in the same fragment order --
// element (lane l, j) of fragment (group, step, tile, plane) is plane(WT[k = step*16 + 8*(l >> 5) + j][column = group*128 + tile*32
// + (l & 31)]), zero outside [cp_real) x [D).  One work item = one lane's 16 bytes of the three planes.
__device__ __forceinline__ void b16_pack_wt_item(const float* __restrict__ WT, int cp_real, int D, int nchunks,
                                                 unsigned char* __restrict__ bimg, long idx) {
    long r = idx;
    const int l = (int)(r & 63); r >>= 6;
    const int t = (int)(r & 3); r >>= 2;
    const int c = (int)(r % nchunks);
    const int ng = (int)(r / nchunks);
    const int col = ng * kB16BN + t * 32 + (l & 31);
    const int k0 = c * kB16KC + 8 * (l >> 5);
    u32x4 ph, pm, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = k0 + 2 * q + e;
            x[e] = (k < cp_real && col < D) ? WT[(long)k * D + col] : 0.f;
        }
        unsigned a, b, cc;
        split_pair<3>(x[0], x[1], a, b, cc);
        ph[q] = a; pm[q] = b; pl[q] = cc;
    }
    unsigned char* dst = bimg + (((size_t)ng * nchunks + c) * 4 + t) * 3 * kB16BFrag + l * 16;
    *reinterpret_cast<u32x4*>(dst) = ph;
    *reinterpret_cast<u32x4*>(dst + kB16BFrag) = pm;
    *reinterpret_cast<u32x4*>(dst + 2 * kB16BFrag) = pl;
}

}  // namespace rbr
