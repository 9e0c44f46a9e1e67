// textcnn_prod_b16.hip -- the distinct-token GEMM  T = table[tok_of_row] @ Wprod  on the bf16 MFMA pipe.
//
// gfx950 has no reduced-precision f32 MFMA (no xf32): v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate.  An f32
// value splits EXACTLY into three bf16 planes (x = hi + mid + lo up to < 2^-25 |x|: round-to-nearest residues, each
// subtraction exact), so   a*b = a_hi*b_hi + (a_hi*b_mid + a_mid*b_hi) + (a_hi*b_lo + a_lo*b_hi + a_mid*b_mid) + O(2^-26 |a b|)
// costs 6 v_mfma_f32_32x32x16_bf16 per 16 k (192 cycles) against 8 v_mfma_f32_32x32x2_f32 (512 cycles) for the same
// products, accumulated in f32: the same error class as the f32 chain (products exact, one f32 rounding per add).
//   precision 1 = "bf16x3": 6 plane products (f32-class accuracy; the headline)
//   precision 2 = "bf16x2": 3 plane products (hi*hi, hi*mid, mid*hi: ~2^-17 relative per product)
//   precision 3 = "bf16"  : 1 plane product  (operands rounded to bf16: the reduced-precision row of BASELINE configs 3/5)
// The table stays f32 in HBM (Adam's master copy); rows are gathered by LDS-DMA and split in registers by the wave that
// owns them, the weight planes are split once per step by the pack stage and stored in MFMA-fragment order.
//
// Work item = 128 token rows x 128 product channels (grid: every such pair, XCD-aware order).  4 waves, wave w owns the
// 32-row tile w and 4 channel tiles (64 accumulator registers); two workgroups per CU.  K is walked in 16-deep MFMA steps
// through a 4-stage LDS ring filled three steps ahead:
// A stage [128 rows][4 x 16 B] (row segments XOR-swizzled so the 8-float fragment reads are conflict-free), B stage
// [4 tiles][3 planes][64 lanes x 16 B] (fragment order: a straight copy).  One barrier per step.
// Measured at cfg2 (21 344 rows x 300 x 768, 1002 items): 64 us (bf16x3) / 47 (bf16x2) / 39 (bf16) against 104 us for the
// f32 MFMA kernel.  What bounds it is the CU's LDS-fill path, not the MFMA pipe and not L2: a stage is 20 one-KiB LDS-DMA
// instructions per workgroup for 4 x 24 MFMAs, and a CU retires ~1 KiB per 35 cycles whatever the source (with every fill
// reading one cached 16-byte line and no stores the bf16 kernel still takes 24 us = 381 k instructions / 256 CUs x 35
// cycles).  Tried and dropped (round 2, each correct to the same 1.3e-6):
//  * weights stationary in registers (one 32-column tile x all 19 K steps per wave = 228 VGPRs, one wave per SIMD, rows
//    split into planes once per workgroup and shared through LDS, work balanced by row ranges): 60 % fewer fills -- and
//    77 us.  With one wave per SIMD nothing hides the per-unit chain (fill issue -> 7 LDS reads -> split -> plane stores
//    -> barrier, ~570 cycles for 384 MFMA cycles) even with the MFMAs interleaved by sched_group_barrier and the second
//    step's MFMAs deferred across the barrier; two waves per SIMD do not fit beside 228 weight registers;
//  * two dedicated loader waves per workgroup issuing all 20 fills of a stage (the MFMA waves issue none): 86 us -- the
//    fills' cost is not their issue slot in the MFMA waves but the LDS side: a stage pair writes 80 KiB into LDS by DMA and
//    reads 112 KiB back per CU, which at the DMA's ~64 B/clk into LDS is as long as the stage pair's 1536 MFMA cycles;
//  * 256-row x 128-column items (8 waves, weights shared by twice the rows, two stages): 78 us.
// DESIGN.md section 4.
#include "rbr_common.h"
#include "textcnn_b16.h"

#include <cstdlib>
#include <type_traits>

namespace rbr {

__device__ float g_b16_zero[4];      // LDS-DMA source of rows / columns outside the problem (zero-initialised, never written)

struct B16Gemm {
    const int* counter;              // rows of the token list (device)
    const long long* tok_of_row;
    const float* table;
    const unsigned char* bimg;       // [group][chunk][tile][step][plane][lane][8 bf16]
    float* T;
    int cap, D, pitch, ngroups, nchunks;
    int ncols;                       // columns of T that exist (stores beyond are dropped); 0 = every column of a group (pitch-wide T)
    float* sq_part;                  // != NULL: sq_part[workgroup] = sum of squares of what this workgroup stored (fixed order)
};

__device__ __forceinline__ void b16_dma16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// NPROD plane products per f32 product: 6 (bf16x3), 3 (bf16x2), 1 (bf16)
// all but the two youngest stages' fills (2 row + NPLANES weight instructions per wave and stage) have landed
template <int NPLANES>
__device__ __forceinline__ void b16_wait_two_stages() {
    if (NPLANES == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (NPLANES == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

template <int NPROD>
__device__ __forceinline__ void prod_gemm_b16_kernel(const B16Gemm& g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int NPLANES = NPROD == 6 ? 3 : NPROD == 3 ? 2 : 1;
    const int n = min(*g.counter, g.cap);
    // blocks b and b + 8 share an XCD (round-robin dispatch): the ngroups items of a row block sit on ONE XCD, so its rows
    // are fetched into that L2 once; speed only
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int mblock = (jj / g.ngroups) * 8 + xcd, ng = jj - (jj / g.ngroups) * g.ngroups;
    const int m0 = mblock * kB16BM;
    if (m0 >= n) {
        if (g.sq_part != nullptr && threadIdx.x == 0) g.sq_part[blockIdx.x] = 0.f;
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = g.D;
    unsigned char* const Abuf = smem;
    unsigned char* const Bbuf = smem + kB16Stages * kB16ABytes;

    // gather role: instruction q of this wave moves rows wave*32 + q*16 .. +15, lane = (row in 16, 16-byte position)
    long aoff[2];
    int aseg[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rl = wave * 32 + q * 16 + (lane >> 2);
        const int row = m0 + rl;
        aoff[q] = (row < n) ? (g.tok_of_row != nullptr ? g.tok_of_row[row] : (long long)row) * (long)D : -1;      // NULL: rows in place
        aseg[q] = ((lane & 3) ^ ((rl >> 2) & 3)) * 4;          // position p of the row holds segment p ^ swz(row)
    }
    // weight fragments of a stage: 12 x 1 KiB, tile `wave`'s planes per wave -- only the planes this precision multiplies.  Every
    // wave issues kOps = 2 + NPLANES LDS-DMA instructions per stage, also past the last stage (source clamped, slot unused), so
    // the counted waits below (two younger stages in flight) hold in every step
    const unsigned char* bsrc = g.bimg + ((size_t)ng * g.nchunks) * kB16BBytes + (size_t)wave * 3 * kB16BFrag + lane * 16;
    const int nch = g.nchunks;
    auto issue_a = [&](int c) {
        const int buf = c & (kB16Stages - 1), cs = min(c, nch - 1);
        unsigned char* A = Abuf + buf * kB16ABytes + wave * 32 * 64;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = cs * kB16KC + aseg[q];
            const float* src = (aoff[q] >= 0 && col < D) ? g.table + aoff[q] + col : g_b16_zero;
            b16_dma16(src, A + q * 16 * 64);
        }
    };
    auto issue_b = [&](int c) {
        const int buf = c & (kB16Stages - 1), cs = min(c, nch - 1);
        unsigned char* B = Bbuf + buf * kB16BBytes + wave * 3 * kB16BFrag;
        const unsigned char* s = bsrc + (size_t)cs * kB16BBytes;
#pragma unroll
        for (int t = 0; t < NPLANES; ++t) b16_dma16(s + t * kB16BFrag, B + t * kB16BFrag);
    };

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int r32 = lane & 31, h = lane >> 5;
    const int swz = (r32 >> 2) & 3;
    const int arow = (wave * 32 + r32) * 64;
    const int seg0 = ((2 * h) ^ swz) * 16, seg1 = ((2 * h + 1) ^ swz) * 16;

    bf16x8 a_hi, a_mid, a_lo;
    auto split_rows = [&](const f32x4 x0, const f32x4 x1) {      // this lane's 8 floats of a step -> bf16 planes
        unsigned h4[4], m4[4] = {0, 0, 0, 0}, l4[4] = {0, 0, 0, 0};
        split_pair<NPLANES>(x0.x, x0.y, h4[0], m4[0], l4[0]);
        split_pair<NPLANES>(x0.z, x0.w, h4[1], m4[1], l4[1]);
        split_pair<NPLANES>(x1.x, x1.y, h4[2], m4[2], l4[2]);
        split_pair<NPLANES>(x1.z, x1.w, h4[3], m4[3], l4[3]);
        const u32x4 ah = {h4[0], h4[1], h4[2], h4[3]}, am = {m4[0], m4[1], m4[2], m4[3]}, al = {l4[0], l4[1], l4[2], l4[3]};
        a_hi = __builtin_bit_cast(bf16x8, ah);
        a_mid = a_lo = a_hi;
        if (NPLANES >= 2) a_mid = __builtin_bit_cast(bf16x8, am);
        if (NPLANES >= 3) a_lo = __builtin_bit_cast(bf16x8, al);
    };
    auto mma_tile = [&](const unsigned char* B, int t) {
        const unsigned char* bf = B + (t * 3) * kB16BFrag;
        const bf16x8 b_hi = *reinterpret_cast<const bf16x8*>(bf);
        if (NPROD == 6) {       // small terms first
            const bf16x8 b_mid = *reinterpret_cast<const bf16x8*>(bf + kB16BFrag);
            const bf16x8 b_lo = *reinterpret_cast<const bf16x8*>(bf + 2 * kB16BFrag);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_mid, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_hi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_mid, acc[t], 0, 0, 0);
        } else if (NPROD == 3) {
            const bf16x8 b_mid = *reinterpret_cast<const bf16x8*>(bf + kB16BFrag);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_hi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_mid, acc[t], 0, 0, 0);
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[t], 0, 0, 0);
    };

    // Software pipeline: the fills run three stages ahead; the wave's own token rows of stage c + 1 are read and split
    // between the MFMAs of stage c (a wave gathers exactly the rows of its own tile: no barrier on that path), and the
    // fill instructions of stage c + 3 are issued between the MFMA groups too -- in front of the chain they cost the wave
    // ~100 cycles each with the matrix pipe idle.
    for (int c = 0; c < 3; ++c) { issue_a(c); issue_b(c); }
    b16_wait_two_stages<NPLANES>();            // stage 0 (this wave's share)
    {
        const unsigned char* A = Abuf + arow;
        split_rows(*reinterpret_cast<const f32x4*>(A + seg0), *reinterpret_cast<const f32x4*>(A + seg1));
    }
    for (int c = 0; c < nch; ++c) {
        // stage c has landed once at most the two younger stages' fills (5 instructions each) are outstanding
        b16_wait_two_stages<NPLANES>();
        // a bare s_barrier: __syncthreads() would add a fence that drains the fills meant to stay in flight
        __builtin_amdgcn_s_barrier();         // weights of stage c landed for every wave; every wave is done with stage c - 1
        asm volatile("" ::: "memory");
        const unsigned char* B = Bbuf + (c & (kB16Stages - 1)) * kB16BBytes + lane * 16;
        const unsigned char* An = Abuf + ((c + 1) & (kB16Stages - 1)) * kB16ABytes + arow;
        mma_tile(B, 0);
        issue_a(c + 3);                       // into the slot of stage c - 1
        mma_tile(B, 1);
        issue_b(c + 3);
        mma_tile(B, 2);
        b16_wait_two_stages<NPLANES>();        // this wave's rows of stage c + 1
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(An + seg0), x1 = *reinterpret_cast<const f32x4*>(An + seg1);
        mma_tile(B, 3);
        split_rows(x0, x1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the surplus fills have drained before the LDS is released
    // C/D map of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): one store
    // instruction writes two whole 128-byte lines
    float* out = g.T + (size_t)(m0 + wave * 32 + 4 * h) * g.pitch + ng * kB16BN + r32;
    const int rows_left = n - (m0 + wave * 32 + 4 * h);
    const int cols_left = (g.ncols > 0 ? g.ncols : g.ngroups * kB16BN) - (ng * kB16BN + r32);      // this lane's column t*32 exists iff < cols_left
    float sq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (dr < rows_left) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t * 32 < cols_left) {
                    out[(size_t)dr * g.pitch + t * 32] = acc[t][r];
                    sq = fmaf(acc[t][r], acc[t][r], sq);
                }
        }
    }
    if (g.sq_part != nullptr) {          // the rows' share of the gradient norm: lanes by shuffle, waves in wave order
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        float* s_sq = reinterpret_cast<float*>(smem);
        __builtin_amdgcn_s_barrier();      // every wave is out of the K loop: the ring is free
        if (lane == 0) s_sq[wave] = sq;
        __syncthreads();
        if (tid == 0) g.sq_part[blockIdx.x] = (s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3]);
    }
}

// ---------------------------------------------------------------------------------- token rows straight into registers
// The same GEMM with the A operand (the token rows) loaded from global memory into the registers of the lane that multiplies
// them -- lane (row r32, half h) of a wave reads its 8 floats of a 16-deep step as two dwordx4, three steps ahead, through a
// ring of four register sets -- instead of through the LDS ring: the kernel above is bound by the CU's LDS-fill path, and the
// rows were 8 of the 20 KiB a stage moves through it.  Only the weight fragments still go through LDS (48 KiB).  Same values
// into the same MFMAs in the same order: the same bits.  Measured at cfg2 (1008 -> 504 workgroups): 61.0 us (rows through LDS)
// -> 59.0 (rows in registers, NT = 4) -> 56.8 (NT = 8) -> 55.4 us with the next stage's split spread between this stage's MFMAs.
// SQ counters of the NT = 8 form: the matrix pipe busy 58 % of the CUs' cycles (1.82 M MFMAs x 32 cycles), 2.5 VALU
// instructions per MFMA that co-execute with it for 6 % of its cycles only, and a clock of ~1.7 GHz under this load (what
// "2.5 PF at 2.4 GHz" prices the kernel against is not a clock the chip holds here).  Dropped: 256 rows x 256 columns per
// workgroup, one workgroup of 8 waves per CU (the stage's fragments filled once for twice the rows, 4 stages): 88 us -- as with
// round 2's 256-row item, eight waves at one barrier per step lose more than the halved fills give; and 256 rows x 128 columns
// with four waves of TWO row tiles each (every weight fragment read from LDS feeds two MFMA chains: half the LDS reads and fills
// per MFMA, rows one step ahead in a two-slot register ring): 86 us.
// NT = 32-column tiles per wave: 4 (one 128-column group per workgroup, 4 stages) or 8 (TWO adjacent groups, 3 stages: the
// rows' loads and their split into planes -- 3.7 VALU instructions per MFMA at NT = 4, by the SQ counters, with the matrix pipe
// busy 56 % of the CU's cycles -- are shared by twice the columns).
template <int NT> struct B16d {
    static constexpr int kStages = NT == 4 ? 4 : 3;
    static constexpr int kStageBytes = (NT / 4) * kB16BBytes;
    static constexpr int kLds = kStages * kStageBytes;
    static constexpr int kOps = 2 + 3 * (NT / 4);              // vector-memory instructions per wave and stage
};

template <int NPROD, int NT>
__device__ __forceinline__ void prod_gemm_b16d_kernel(const B16Gemm& g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    using Cfg = B16d<NT>;
    constexpr int S = Cfg::kStages, NG = NT / 4;
    constexpr int NPLANES = NPROD == 6 ? 3 : NPROD == 3 ? 2 : 1;
    const int n = min(*g.counter, g.cap);
    const int gpw = (g.ngroups + NG - 1) / NG;                   // workgroups per row block
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int mblock = (jj / gpw) * 8 + xcd, ng = (jj - (jj / gpw) * gpw) * NG;
    const int m0 = mblock * kB16BM;
    if (m0 >= n) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = g.D;
    unsigned char* const Bbuf = smem;
    const int r32 = lane & 31, h = lane >> 5;
    const int row = m0 + wave * 32 + r32;
    const float* const arow = (row < n) ? g.table + g.tok_of_row[row] * (long)D + 8 * h : nullptr;
    const int nch = g.nchunks;
    const unsigned char* bsrc[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q)       // a second group past the last one re-reads the last (its tiles are never stored)
        bsrc[q] = g.bimg + ((size_t)min(ng + q, g.ngroups - 1) * nch) * kB16BBytes + (size_t)wave * 3 * kB16BFrag + lane * 16;
    f32x4 ring0[S], ring1[S];                  // [slot]: floats 0..3 / 4..7 of the lane's 8 (compile-time slots only)
    auto issue_a = [&](int c, f32x4& x0, f32x4& x1) {          // ALWAYS two loads (clamped source past the end / outside the rows)
        const int cs = min(c, nch - 1), col = cs * kB16KC + 8 * h;
        const float* s0 = (arow != nullptr && col + 4 <= D) ? arow + cs * kB16KC : g_b16_zero;
        const float* s1 = (arow != nullptr && col + 8 <= D) ? arow + cs * kB16KC + 4 : g_b16_zero;
        x0 = *reinterpret_cast<const f32x4*>(s0);
        x1 = *reinterpret_cast<const f32x4*>(s1);
    };
    auto issue_b = [&](int c, int slot) {
        const int cs = min(c, nch - 1);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            unsigned char* B = Bbuf + slot * Cfg::kStageBytes + q * kB16BBytes + wave * 3 * kB16BFrag;
            const unsigned char* s_ = bsrc[q] + (size_t)cs * kB16BBytes;
#pragma unroll
            for (int t = 0; t < 3; ++t) b16_dma16(s_ + t * kB16BFrag, B + t * kB16BFrag);
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a_hi, a_mid, a_lo;                  // planes of the stage being multiplied
    bf16x8 n_hi, n_mid, n_lo;                  // ... of the next one: split between this stage's MFMAs, handed over at its end
    auto split_rows = [&](const f32x4 x0, const f32x4 x1) {
        unsigned h4[4], m4[4] = {0, 0, 0, 0}, l4[4] = {0, 0, 0, 0};
        split_pair<NPLANES>(x0.x, x0.y, h4[0], m4[0], l4[0]);
        split_pair<NPLANES>(x0.z, x0.w, h4[1], m4[1], l4[1]);
        split_pair<NPLANES>(x1.x, x1.y, h4[2], m4[2], l4[2]);
        split_pair<NPLANES>(x1.z, x1.w, h4[3], m4[3], l4[3]);
        const u32x4 ah = {h4[0], h4[1], h4[2], h4[3]}, am = {m4[0], m4[1], m4[2], m4[3]}, al = {l4[0], l4[1], l4[2], l4[3]};
        n_hi = __builtin_bit_cast(bf16x8, ah);
        n_mid = n_lo = n_hi;
        if (NPLANES >= 2) n_mid = __builtin_bit_cast(bf16x8, am);
        if (NPLANES >= 3) n_lo = __builtin_bit_cast(bf16x8, al);
    };
    auto mma_tile = [&](const unsigned char* B, int t) {       // t: tile of the workgroup's NT (group t / 4, tile t % 4 of it)
        const unsigned char* bf = B + (t >> 2) * kB16BBytes + ((t & 3) * 3) * kB16BFrag;
        const bf16x8 b_hi = *reinterpret_cast<const bf16x8*>(bf);
        if (NPROD == 6) {       // small terms first
            const bf16x8 b_mid = *reinterpret_cast<const bf16x8*>(bf + kB16BFrag);
            const bf16x8 b_lo = *reinterpret_cast<const bf16x8*>(bf + 2 * kB16BFrag);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_mid, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_hi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_mid, acc[t], 0, 0, 0);
        } else if (NPROD == 3) {
            const bf16x8 b_mid = *reinterpret_cast<const bf16x8*>(bf + kB16BFrag);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_hi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_mid, acc[t], 0, 0, 0);
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[t], 0, 0, 0);
    };
    // stage c with its ring slot Sc = c % S as a compile-time number: the rows of stage c + 1 sit in slot (Sc + 1) % S, the
    // loads of stage c + S - 1 go to slot (Sc + S - 1) % S (the slot stage c - 1 has just left)
    auto stage = [&](int c, auto Sc_) {
        constexpr int Sc = decltype(Sc_)::value;
        if (S == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // stage c landed: two younger stages x 5 instructions
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               //                  one younger stage x 8
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned char* B = Bbuf + Sc * Cfg::kStageBytes + lane * 16;
        mma_tile(B, 0);
        issue_a(c + S - 1, ring0[(Sc + S - 1) % S], ring1[(Sc + S - 1) % S]);
        mma_tile(B, 1);
        issue_b(c + S - 1, (Sc + S - 1) % S);
        split_rows(ring0[(Sc + 1) % S], ring1[(Sc + 1) % S]);   // VALU work with no MFMA of this stage depending on it ...
#pragma unroll
        for (int t = 2; t < NT; ++t) mma_tile(B, t);
        // ... which the scheduler is asked to spread between them: one MFMA, then two VALU instructions, over the stage
#pragma unroll
        for (int k = 0; k < NT * (NPROD == 6 ? 6 : NPROD == 3 ? 3 : 1); ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        a_hi = n_hi; a_mid = n_mid; a_lo = n_lo;
    };
#pragma unroll
    for (int c = 0; c < S - 1; ++c) { issue_a(c, ring0[c], ring1[c]); issue_b(c, c); }
    split_rows(ring0[0], ring1[0]);
    a_hi = n_hi; a_mid = n_mid; a_lo = n_lo;
    if (S == 4) {
        for (int c = 0; c < nch; c += 4) {                       // workgroup-uniform guards
            stage(c, std::integral_constant<int, 0>{});
            if (c + 1 < nch) stage(c + 1, std::integral_constant<int, 1>{});
            if (c + 2 < nch) stage(c + 2, std::integral_constant<int, 2>{});
            if (c + 3 < nch) stage(c + 3, std::integral_constant<int, 3 % S>{});
        }
    } else {
        for (int c = 0; c < nch; c += 3) {
            stage(c, std::integral_constant<int, 0>{});
            if (c + 1 < nch) stage(c + 1, std::integral_constant<int, 1>{});
            if (c + 2 < nch) stage(c + 2, std::integral_constant<int, 2>{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* out = g.T + (size_t)(m0 + wave * 32 + 4 * h) * g.pitch + ng * kB16BN + r32;
    const int rows_left = n - (m0 + wave * 32 + 4 * h);
    const int tiles_ok = min(NT, (g.ngroups - ng) * 4);          // the tiles of a second group that does not exist are dropped
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (dr < rows_left) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t < tiles_ok) out[(size_t)dr * g.pitch + t * 32] = acc[t][r];
        }
    }
}

// ---------------------------------------------------------------------------------- rows stationary (short K)
// The same GEMM for a SHORT contraction (D <= 128: D-ATT's 100-wide embeddings are 7 steps).  With so few steps the kernels
// above spend as long filling their ring and storing their tile as multiplying: a 128 x 128 item is ~9 us of which 4.5 are the
// K loop (D-ATT cfg4: 2 088 items in five rounds, 69 us for 18 us of MFMA time).  Here a workgroup keeps its 128 token rows --
// every K step of them, split into planes ONCE -- in registers (12 VGPRs per step) and walks the column groups with them: the
// weight fragments stream through the same 4-stage LDS ring without a break from one group to the next, and a finished group's
// tiles are stored while the next group's fills and MFMAs are already under way.
// The accumulators are held TRANSPOSED (weights as the MFMA's A operand, token rows as B: the hardware's k order is the same,
// so the bits are those of the kernels above): a lane then owns 4 adjacent columns of one token row per register quad and a tile
// leaves as 4 dwordx4 stores instead of 16 dword stores -- 16 store instructions per group and wave, which keeps the counted
// vmcnt waits of the fill pipeline (6 bits: at most 63) exact across the group boundary: in the first three stages of a group
// the previous group's 16 stores are younger than the stage's own fills and are allowed for.  A workgroup whose row block is cut
// by the end of the list (its stores are predicated, their count is not known) waits for everything instead.
// Grid: row blocks x up to kB16kSplit column splits; the kernel uses as many splits as give ~one resident round (512 slots).
// D-ATT cfg4 (29.7 k rows x 100 x 1 152 columns, 464 workgroups of 4 or 5 groups): 54 us against the ring kernel's 69.
constexpr int kB16kSplit = 4;
constexpr int kB16kPitch = 144;                                // bytes per staging row (128 + 16: the rows start 4 banks apart)
constexpr int kB16kLds = kB16Stages * kB16BBytes + kB16Waves * 32 * kB16kPitch;

template <int NPROD, int NCH>
__device__ __forceinline__ void prod_gemm_b16k_kernel(const B16Gemm& g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int NPLANES = NPROD == 6 ? 3 : NPROD == 3 ? 2 : 1;
    constexpr int S = kB16Stages;
    const int n = min(*g.counter, g.cap);
    const int smax = min(kB16kSplit, g.ngroups);
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    // the split is the SLOW index of the grid: the workgroups that have nothing to do (splits not used for this n, row blocks
    // past the list) then sit in long runs.  With the split as the fast index every other pair of consecutive workgroups left at
    // once, the dispatcher's round robin over the shader engines gave the real ones to two engines of four, and the kernel ran
    // in two rounds on half the CUs (105 us; seen in per-workgroup s_memtime stamps and HW_ID)
    const int nb8 = (int)(gridDim.x >> 3) / smax;              // octets of row blocks
    const int mblock = (jj % nb8) * 8 + xcd, sp = jj / nb8;
    const int m0 = mblock * kB16BM;
    if (m0 >= n) return;
    const int nsplit = max(1, min(smax, 512 / ((n + kB16BM - 1) / kB16BM)));
    if (sp >= nsplit) return;
    const int g0 = (g.ngroups * sp) / nsplit, g1 = (g.ngroups * (sp + 1)) / nsplit;
    const int nst = (g1 - g0) * NCH;                           // stages of this workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = g.D;
    unsigned char* const Bbuf = smem;
    const int r32 = lane & 31, h = lane >> 5;
    const int row = m0 + wave * 32 + r32;
    const bool full = m0 + kB16BM <= n;                        // workgroup-uniform
    const unsigned char* const bsrc = g.bimg + ((size_t)g0 * NCH) * kB16BBytes + (size_t)wave * 3 * kB16BFrag + lane * 16;
    auto issue_b = [&](int s) {                                // ALWAYS three fills (source clamped past the end, slot unused)
        const int ss = min(s, nst - 1);
        unsigned char* B = Bbuf + (s & (S - 1)) * kB16BBytes + wave * 3 * kB16BFrag;
        const unsigned char* s_ = bsrc + (size_t)ss * kB16BBytes;
#pragma unroll
        for (int t = 0; t < 3; ++t) b16_dma16(s_ + t * kB16BFrag, B + t * kB16BFrag);
    };
#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue_b(s);
    // the rows: every step's 8 floats of this lane, requested together, split once
    bf16x8 a_hi[NCH], a_mid[NCH], a_lo[NCH];
    {
        const float* const arow = (row < n) ? g.table + (g.tok_of_row != nullptr ? g.tok_of_row[row] : (long long)row) * (long)D + 8 * h : nullptr;
        f32x4 x0[NCH], x1[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * kB16KC + 8 * h;
            const float* s0 = (arow != nullptr && col + 4 <= D) ? arow + c * kB16KC : g_b16_zero;
            const float* s1 = (arow != nullptr && col + 8 <= D) ? arow + c * kB16KC + 4 : g_b16_zero;
            x0[c] = *reinterpret_cast<const f32x4*>(s0);
            x1[c] = *reinterpret_cast<const f32x4*>(s1);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            unsigned h4[4], m4[4] = {0, 0, 0, 0}, l4[4] = {0, 0, 0, 0};
            split_pair<NPLANES>(x0[c].x, x0[c].y, h4[0], m4[0], l4[0]);
            split_pair<NPLANES>(x0[c].z, x0[c].w, h4[1], m4[1], l4[1]);
            split_pair<NPLANES>(x1[c].x, x1[c].y, h4[2], m4[2], l4[2]);
            split_pair<NPLANES>(x1[c].z, x1[c].w, h4[3], m4[3], l4[3]);
            const u32x4 ah = {h4[0], h4[1], h4[2], h4[3]}, am = {m4[0], m4[1], m4[2], m4[3]}, al = {l4[0], l4[1], l4[2], l4[3]};
            a_hi[c] = __builtin_bit_cast(bf16x8, ah);
            a_mid[c] = a_lo[c] = a_hi[c];
            if (NPLANES >= 2) a_mid[c] = __builtin_bit_cast(bf16x8, am);
            if (NPLANES >= 3) a_lo[c] = __builtin_bit_cast(bf16x8, al);
        }
    }
    // (the split needed the rows, the rows were requested after the first three stages' fills and memory returns in order:
    // stages 0 .. 2 have landed for this wave)
    f32x16 acc[4];
    unsigned char* const stg = smem + S * kB16BBytes + wave * (32 * kB16kPitch);      // this wave's 32 staging rows
    unsigned char* const stg_w = stg + r32 * kB16kPitch + 16 * h;                       // + 32 q: quad q of row (lane & 31)
    const unsigned char* const stg_r = stg + (lane >> 3) * kB16kPitch + (lane & 7) * 16;  // + 8 i rows: 16 bytes of row 8 i + lane / 8
    const int orow0 = m0 + wave * 32 + (lane >> 3);
    float* const otile = g.T + (size_t)orow0 * g.pitch + (lane & 7) * 4;
    for (int gi = 0; gi < g1 - g0; ++gi) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int s = gi * NCH + c;
            // stage s has landed once only what was issued after its fills is outstanding: the fills of stages s + 1 and s + 2,
            // and -- in the first three stages of a group but the first -- the 16 stores of the previous group
            if (!full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (c < 3 && gi > 0) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // the stage landed for every wave; every wave is done with stage s - 1
            asm volatile("" ::: "memory");
            const unsigned char* B = Bbuf + (s & (S - 1)) * kB16BBytes + lane * 16;
            // (Measured and dropped: every tile's fragments requested ahead of the previous tile's MFMA chain, the order pinned
            // with sched_group_barrier -- 202 registers, the same 54-55 us.  Per-workgroup s_memtime stamps: ~2.5 k cycles per
            // stage with two workgroups on the CU (1 536 of them MFMA issue), ~2.9 k with one; neither the fills' latency --
            // vmcnt(0) in every stage costs 1 us -- nor the stores -- without them 48 us -- nor the fragment reads' latency.)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned char* bf = B + (t * 3) * kB16BFrag;
                const bf16x8 b_hi = *reinterpret_cast<const bf16x8*>(bf);
                if (NPROD == 6) {       // small terms first, as above
                    const bf16x8 b_mid = *reinterpret_cast<const bf16x8*>(bf + kB16BFrag);
                    const bf16x8 b_lo = *reinterpret_cast<const bf16x8*>(bf + 2 * kB16BFrag);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_hi, a_lo[c], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_lo, a_hi[c], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_mid, a_mid[c], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_hi, a_mid[c], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_mid, a_hi[c], acc[t], 0, 0, 0);
                } else if (NPROD == 3) {
                    const bf16x8 b_mid = *reinterpret_cast<const bf16x8*>(bf + kB16BFrag);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_hi, a_mid[c], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_mid, a_hi[c], acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_hi, a_hi[c], acc[t], 0, 0, 0);
                if (t == 0) issue_b(s + S - 1);                // into the slot stage s - 1 has just left
            }
        }
        // transposed C/D map: register quad q of tile t = columns t*32 + 8q + 4h .. +3 of token row (lane & 31).  Stored as it
        // lies, a dwordx4 store would be 64 separate 16-byte writes (105 us at cfg4 against the ring kernel's 69): the tile goes
        // through this wave's staging rows in LDS (144-byte pitch) and leaves row-major, 8 whole 128-byte lines per instruction
        float* out = otile + (size_t)(g0 + gi) * kB16BN;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(stg_w + 32 * q) = f32x4{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg_r + i * 8 * kB16kPitch);
                if (full || orow0 + 8 * i < n) *reinterpret_cast<f32x4*>(out + (size_t)(8 * i) * g.pitch + t * 32) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the surplus fills have drained before the LDS is released
}

// ---------------------------------------------------------------------------------- bf16 storage form (RBR_PROD_BF16)
// The reduced-precision class as BASELINE configs 3 and 5 name it: not only the MFMA operands but the byte streams are bf16.
//   rows_to_b16   : the distinct tokens' table rows, rounded once, as a compact [list rows][Dp] bf16 array (zero-padded to a
//                   multiple of 32): every channel group of the GEMM re-reads 600-byte rows instead of 1200-byte ones, by plain
//                   row index (no token indirection), and no wave converts anything in the loop;
//   prod_gemm_b16s: 32-deep stages, 2 + 2 LDS-DMA instructions per wave and stage for 8 MFMAs (the f32-row kernel: 5 for 4 --
//                   the LDS-fill path that bounds it), only the hi weight plane is staged, T is written as bf16.
// Measured at cfg2 (round 3, one box): GEMM 38.7 -> 24.4 us (+ 8.8 us for rows_to_b16), gather_pool 44.0 -> 38.9 us, step
// 0.3455 -> 0.3419 ms.  The row-reading kernels of the backward were given bf16 rows too (bf16 Wprod^T in g_times_w, the bf16
// row copy in dw_partial4) and did NOT get faster -- 64 and 76 us against 61 and 71: they issue one load instruction per lane
// and row either way and are bound by that count, not by the bytes -- so the backward keeps its f32 rows.
struct B16sGemm {
    const int* counter;
    const unsigned short* a16;       // [rows][Dp] bf16
    const unsigned char* bimg;       // as B16Gemm
    unsigned short* T;               // [rows][pitch] bf16
    int cap, Dp, pitch, ngroups, nchunks16;
};

__device__ __forceinline__ void rows_to_b16_kernel(const int* __restrict__ counter, int cap, int D, int Dp,
                                                          const long long* __restrict__ tok_of_row, const float* __restrict__ table,
                                                          unsigned short* __restrict__ a16) {
    const int n = min(*counter, cap);
    const int segs = Dp / 8;                                   // 16-byte output segments per row
    const long total = (long)n * segs;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long)gridDim.x * 256) {
        const int row = (int)(k / segs), sg = (int)(k - (long)row * segs);
        const int col = sg * 8;
        u32x4 o = {0u, 0u, 0u, 0u};
        if (col < D) {                                         // D % 4 == 0: a float4 never straddles the row end
            const float* src = table + tok_of_row[row] * (long)D + col;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(src);
            f32x4 x1 = {0.f, 0.f, 0.f, 0.f};
            if (col + 4 < D) x1 = *reinterpret_cast<const f32x4*>(src + 4);
            o = u32x4{pack_bf16(x0.x, x0.y), pack_bf16(x0.z, x0.w), pack_bf16(x1.x, x1.y), pack_bf16(x1.z, x1.w)};
        }
        *reinterpret_cast<u32x4*>(a16 + (long)row * Dp + col) = o;
    }
}

__device__ __forceinline__ void prod_gemm_b16s_kernel(const B16sGemm& g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int n = min(*g.counter, g.cap);
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int mblock = (jj / g.ngroups) * 8 + xcd, ng = jj - (jj / g.ngroups) * g.ngroups;
    const int m0 = mblock * kB16BM;
    if (m0 >= n) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const Abuf = smem;
    unsigned char* const Bbuf = smem + kB16Stages * kB16sABytes;
    const int nch = g.Dp / kB16sKC;                            // stages

    // gather role: instruction q moves rows wave*32 + q*16 .. +15 (this wave's own tile), lane = (row in 16, 16-byte position)
    long aoff[2];
    int aseg[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rl = wave * 32 + q * 16 + (lane >> 2);
        const int row = m0 + rl;
        aoff[q] = (row < n) ? (long)row * g.Dp : -1;
        aseg[q] = ((lane & 3) ^ ((rl >> 2) & 3)) * 8;          // position p of the row holds segment p ^ swz(row) (8 bf16 each)
    }
    // hi-plane weight fragments of a stage: (k-half 0 | 1) x tile `wave`; image chunk = 16 k: [group][chunk16][tile][plane][1 KiB]
    const unsigned char* bsrc = g.bimg + ((size_t)ng * g.nchunks16) * kB16BBytes + (size_t)wave * 3 * kB16BFrag + lane * 16;
    auto issue_a = [&](int c) {
        const int buf = c & (kB16Stages - 1), cs = min(c, nch - 1);
        unsigned char* A = Abuf + buf * kB16sABytes + wave * 32 * 64;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const void* src = (aoff[q] >= 0) ? static_cast<const void*>(g.a16 + aoff[q] + cs * kB16sKC + aseg[q])
                                             : static_cast<const void*>(g_b16_zero);
            b16_dma16(src, A + q * 16 * 64);
        }
    };
    auto issue_b = [&](int c) {
        const int buf = c & (kB16Stages - 1);
        unsigned char* B = Bbuf + buf * kB16sBBytes + wave * kB16BFrag;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int c16 = min(2 * min(c, nch - 1) + kh, g.nchunks16 - 1);      // past the image: any finite weights (the rows are zero there)
            b16_dma16(bsrc + (size_t)c16 * kB16BBytes, B + kh * 4 * kB16BFrag);
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int r32 = lane & 31, h = lane >> 5;
    const int swz = (r32 >> 2) & 3;
    const int arow = (wave * 32 + r32) * 64;
    const int rd0 = (h ^ swz) * 16, rd1 = ((2 + h) ^ swz) * 16;         // k-halves of the stage's two 16-deep MFMA steps

    for (int c = 0; c < 3; ++c) { issue_a(c); issue_b(c); }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                     // stage 0 (4 fills per wave and stage)
    bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Abuf + arow + rd0), a1 = *reinterpret_cast<const bf16x8*>(Abuf + arow + rd1);
    for (int c = 0; c < nch; ++c) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();         // weights of stage c landed for every wave; every wave is done with stage c - 1
        asm volatile("" ::: "memory");
        const unsigned char* B = Bbuf + (c & (kB16Stages - 1)) * kB16sBBytes + lane * 16;
        const unsigned char* An = Abuf + ((c + 1) & (kB16Stages - 1)) * kB16sABytes + arow;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, *reinterpret_cast<const bf16x8*>(B + t * kB16BFrag), acc[t], 0, 0, 0);
            if (t == 0) issue_a(c + 3);       // into the slot of stage c - 1
            if (t == 1) issue_b(c + 3);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, *reinterpret_cast<const bf16x8*>(B + (4 + t) * kB16BFrag), acc[t], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // this wave's own rows of stage c + 1 (its fills are 8 instructions back)
        a0 = *reinterpret_cast<const bf16x8*>(An + rd0);
        a1 = *reinterpret_cast<const bf16x8*>(An + rd1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // C/D map of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); neighbouring lanes pair
    // their columns so that every store moves 4 bytes (two bf16)
    unsigned short* out = g.T + (size_t)(m0 + wave * 32 + 4 * h) * g.pitch + ng * kB16BN + (r32 & ~1);
    const int rows_left = n - (m0 + wave * 32 + 4 * h);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float mine = acc[t][r];
            const float other = __shfl_xor(mine, 1);
            // even lane: (mine, other) of tile t for this row -- every lane takes part in the shuffle, the even ones store
            if (!(r32 & 1) && dr < rows_left)
                *reinterpret_cast<unsigned*>(out + (size_t)dr * g.pitch + t * 32) = pack_bf16(mine, other);
        }
    }
}

// ---------------------------------------------------------------------------------- host side
static int g_prod_precision = -1;        // -1: RBR_PROD_PRECISION env or the default (bf16x3)

int prod_precision() {
    if (g_prod_precision >= 0) return g_prod_precision;
    static const char* env = getenv("RBR_PROD_PRECISION");
    if (env) {
        if (!strcmp(env, "f32")) return RBR_PROD_F32;
        if (!strcmp(env, "bf16x3")) return RBR_PROD_BF16X3;
        if (!strcmp(env, "bf16x2")) return RBR_PROD_BF16X2;
        if (!strcmp(env, "bf16")) return RBR_PROD_BF16;
    }
    return RBR_PROD_BF16X3;
}

// the arithmetic class of a conv call: the descriptor's stamp (rbr_textcnn_desc_stamp) when it carries one, else the process-wide
// setting -- every stage of a call, and its backward, must read the SAME class (the workspace layout depends on it)
int prod_precision_of(const rbr_textcnn_desc* d) {
    return (d->flags & RBR_CONV_CLASS_STAMPED) ? RBR_CONV_CLASS_PRECISION_OF(d->flags) : prod_precision();
}
bool prod_b16_applicable(const rbr_textcnn_desc* d) { return prod_precision_of(d) != RBR_PROD_F32 && d->D % 4 == 0; }

// bf16 STORAGE (rows copy, product table): the plain-bf16 class only; RBR_B16_STORAGE=0 keeps f32 streams (an A/B switch: same
// arithmetic class either way, T's extra rounding stays inside the class's stated tolerances)
static int g_b16_storage = -1;           // -1: RBR_B16_STORAGE env (default on), 0 off, 1 on
static bool b16_storage_setting() {
    static const char* env = getenv("RBR_B16_STORAGE");
    return g_b16_storage >= 0 ? g_b16_storage != 0 : !(env && !strcmp(env, "0"));
}
bool prod_t_bf16(const rbr_textcnn_desc* d) {
    const bool on = (d->flags & RBR_CONV_CLASS_STAMPED) ? (d->flags & RBR_CONV_CLASS_T_BF16) != 0 : b16_storage_setting();
    return on && prod_b16_applicable(d) && prod_precision_of(d) == RBR_PROD_BF16;
}
size_t prod_b16_rows_bytes(const rbr_textcnn_desc* d, int cap) {
    return prod_t_bf16(d) ? (size_t)(cap + kB16BM) * b16s_dp(d->D) * 2 : 0;
}

int prod_b16_groups(int cp_real) { return (cp_real + kB16BN - 1) / kB16BN; }
size_t prod_b16_image_bytes(const rbr_textcnn_desc* d, int cp_real) {
    return (size_t)prod_b16_groups(cp_real) * ((d->D + kB16KC - 1) / kB16KC) * kB16BBytes;
}

B16Pack prod_b16_pack_job(const rbr_textcnn_desc* d) {
    B16Pack J{};
    J.n_widths = d->n_widths; J.D = d->D;
    int o = 0;
    for (int w = 0; w < d->n_widths; ++w) { J.kz[w] = d->kz[w]; J.ch[w] = d->ch[w]; J.poff[w] = o; o += d->kz[w] * d->ch[w]; }
    J.cp_real = o;
    J.ngroups = prod_b16_groups(o);
    J.nchunks = (d->D + kB16KC - 1) / kB16KC;
    return J;
}

// The bf16-storage GEMM with the token rows loaded from the f32 table into registers and rounded there (no compact bf16 copy, no
// rows_to_b16 launch), two 128-column groups per workgroup, 32-deep stages: per stage a lane loads its 16 floats (4 dwordx4), a
// wave fills one hi-plane fragment per group and k-half, and multiplies 16 MFMAs.  Same roundings, same MFMA order per
// accumulator as prod_gemm_b16s_kernel: the same bf16 T.
constexpr int kB16sdStages = 3;
constexpr int kB16sdStageBytes = 2 * kB16sBBytes;                // 16 KiB: [group][k-half][tile]
constexpr int kB16sdLds = kB16sdStages * kB16sdStageBytes;       // 48 KiB

__device__ __forceinline__ void prod_gemm_b16sd_kernel(const B16Gemm& g, unsigned short* __restrict__ T16, int nchunks16) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int S = kB16sdStages, NG = 2, NT = 8;
    const int n = min(*g.counter, g.cap);
    const int gpw = (g.ngroups + NG - 1) / NG;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int mblock = (jj / gpw) * 8 + xcd, ng = (jj - (jj / gpw) * gpw) * NG;
    const int m0 = mblock * kB16BM;
    if (m0 >= n) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = g.D;
    unsigned char* const Bbuf = smem;
    const int r32 = lane & 31, h = lane >> 5;
    const int row = m0 + wave * 32 + r32;
    const float* const arow = (row < n) ? g.table + g.tok_of_row[row] * (long)D + 8 * h : nullptr;
    const int nch = (D + kB16sKC - 1) / kB16sKC;               // 32-deep stages
    const unsigned char* bsrc[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q)
        bsrc[q] = g.bimg + ((size_t)min(ng + q, g.ngroups - 1) * nchunks16) * kB16BBytes + (size_t)wave * 3 * kB16BFrag + lane * 16;
    f32x4 ring[S][4];                      // [slot][k-half * 2 + (floats 0..3 | 4..7)]
    auto issue_a = [&](int c, f32x4 (&x)[4]) {             // ALWAYS four loads (clamped source past the row / outside the rows)
        const int cs = min(c, nch - 1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int col = cs * kB16sKC + kh * 16 + 8 * h + 4 * f;
                const float* s_ = (arow != nullptr && col + 4 <= D) ? arow + cs * kB16sKC + kh * 16 + 4 * f : g_b16_zero;
                x[kh * 2 + f] = *reinterpret_cast<const f32x4*>(s_);
            }
    };
    auto issue_b = [&](int c, int slot) {
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            unsigned char* B = Bbuf + slot * kB16sdStageBytes + q * kB16sBBytes + wave * kB16BFrag;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const int c16 = min(2 * min(c, nch - 1) + kh, nchunks16 - 1);      // past the image: finite weights, zero rows
                b16_dma16(bsrc[q] + (size_t)c16 * kB16BBytes, B + kh * 4 * kB16BFrag);
            }
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a0, a1, n0, n1;
    auto round_rows = [&](const f32x4 (&x)[4]) {
        const u32x4 p0 = {pack_bf16(x[0].x, x[0].y), pack_bf16(x[0].z, x[0].w), pack_bf16(x[1].x, x[1].y), pack_bf16(x[1].z, x[1].w)};
        const u32x4 p1 = {pack_bf16(x[2].x, x[2].y), pack_bf16(x[2].z, x[2].w), pack_bf16(x[3].x, x[3].y), pack_bf16(x[3].z, x[3].w)};
        n0 = __builtin_bit_cast(bf16x8, p0);
        n1 = __builtin_bit_cast(bf16x8, p1);
    };
    auto stage = [&](int c, auto Sc_) {
        constexpr int Sc = decltype(Sc_)::value;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // stage c landed: one younger stage x 8 instructions
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned char* B = Bbuf + Sc * kB16sdStageBytes + lane * 16;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, *reinterpret_cast<const bf16x8*>(B + (t >> 2) * kB16sBBytes + (t & 3) * kB16BFrag),
                                                             acc[t], 0, 0, 0);
            if (t == 0) issue_a(c + S - 1, ring[(Sc + S - 1) % S]);
            if (t == 1) issue_b(c + S - 1, (Sc + S - 1) % S);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, *reinterpret_cast<const bf16x8*>(B + (t >> 2) * kB16sBBytes + (4 + (t & 3)) * kB16BFrag),
                                                             acc[t], 0, 0, 0);
        round_rows(ring[(Sc + 1) % S]);
        a0 = n0; a1 = n1;
    };
#pragma unroll
    for (int c = 0; c < S - 1; ++c) { issue_a(c, ring[c]); issue_b(c, c); }
    round_rows(ring[0]);
    a0 = n0; a1 = n1;
    for (int c = 0; c < nch; c += 3) {                           // workgroup-uniform guards
        stage(c, std::integral_constant<int, 0>{});
        if (c + 1 < nch) stage(c + 1, std::integral_constant<int, 1>{});
        if (c + 2 < nch) stage(c + 2, std::integral_constant<int, 2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned short* out = T16 + (size_t)(m0 + wave * 32 + 4 * h) * g.pitch + ng * kB16BN + (r32 & ~1);
    const int rows_left = n - (m0 + wave * 32 + 4 * h);
    const int tiles_ok = min(NT, (g.ngroups - ng) * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float mine = acc[t][r];
            const float other = __shfl_xor(mine, 1);
            if (!(r32 & 1) && dr < rows_left && t < tiles_ok)
                *reinterpret_cast<unsigned*>(out + (size_t)dr * g.pitch + t * 32) = pack_bf16(mine, other);
        }
    }
}

// ---- the backward's row GEMM: rows [cap, D] = G [cap, KG] @ Wprod^T  (rbr_common.h)
bool prod_b16_rows_applicable(const rbr_textcnn_desc* d) {
    // Many short documents (NARRE: 5120 reviews of 50 tokens -> 2.3 M taps into 14.8 k x 450 cells: G is a third full) make the
    // sparse row product fetch a 1200-byte row of Wprod^T per non-zero (131 us at cfg3) where the dense product on the bf16 pipe,
    // exact three-plane split as in the forward, is 4 GFLOP.  The rule is the weight gradient's (dw_from_g): the same shapes.
    long cp = 0;
    for (int w = 0; w < d->n_widths; ++w) cp += (long)d->kz[w] * d->ch[w];
    static const bool off = getenv("RBR_ROWS_GEMM") != nullptr && atoi(getenv("RBR_ROWS_GEMM")) == 0;
    return !off && d->D % 4 == 0 && d->L <= 128 && d->n_docs >= 512 && cp <= 2048;
}
size_t prod_b16_rows_image_bytes(int KG, int D) {
    return (size_t)((D + kB16BN - 1) / kB16BN) * ((KG + kB16KC - 1) / kB16KC) * kB16BBytes;
}
long prod_b16_rows_image_items(int KG, int D) { return (long)((D + kB16BN - 1) / kB16BN) * ((KG + kB16KC - 1) / kB16KC) * 4 * 64; }
int prod_b16_rows_partials(int cap, int D) {
    return (((cap + kB16BM - 1) / kB16BM + 7) / 8 * 8) * ((D + kB16BN - 1) / kB16BN);
}
int prod_b16_rows_gemm(int KG, int D, int cap, const int* counter, const float* G, const void* bimg_t, float* rows, float* sq_part,
                       bool plain_bf16, hipStream_t st) {
    if ((((uintptr_t)G) & 15) != 0 || (KG & 3) != 0 || (D & 3) != 0) { set_error("row GEMM operands must be 16-byte aligned"); return RBR_ERR_UNSUPPORTED; }
    B16Gemm g{};
    g.counter = counter; g.tok_of_row = nullptr; g.table = G; g.bimg = static_cast<const unsigned char*>(bimg_t); g.T = rows;
    g.cap = cap; g.D = KG; g.pitch = D; g.ngroups = (D + kB16BN - 1) / kB16BN; g.nchunks = (KG + kB16KC - 1) / kB16KC;
    g.ncols = D; g.sq_part = sq_part;
    const int mblocks = ((cap + kB16BM - 1) / kB16BM + 7) / 8 * 8;
    // the bf16 class (RBR_PROD_BF16): G and Wprod^T rounded to bf16, one plane product, f32 accumulation -- what a bf16 autocast
    // backward computes; every other class keeps the exact three-plane product
    if (plain_bf16)
        return rbr::launch<prod_gemm_b16_kernel<1>, kB16Threads, 2>(dim3((unsigned)(mblocks * g.ngroups)), dim3(kB16Threads), kB16Lds, st,
                                                                   "textcnn rows GEMM launch", g);
    return rbr::launch<prod_gemm_b16_kernel<6>, kB16Threads, 2>(dim3((unsigned)(mblocks * g.ngroups)), dim3(kB16Threads), kB16Lds, st,
                                                               "textcnn rows GEMM launch", g);
}

int prod_b16_gemm(const rbr_textcnn_desc* d, int cp_real, int cap, int pitch, const int* counter, const long long* tok_of_row,
                  const float* table, const void* bimg, void* Tv, void* a16, hipStream_t st) {
    if (((uintptr_t)table & 15) != 0) { set_error("word table must be 16-byte aligned"); return RBR_ERR_UNSUPPORTED; }
    if (prod_t_bf16(d)) {
        if (!a16) { set_error("bf16 storage needs the row workspace"); return RBR_ERR_BAD_ARG; }
        B16sGemm g{};
        g.counter = counter; g.a16 = static_cast<const unsigned short*>(a16); g.bimg = static_cast<const unsigned char*>(bimg);
        g.T = static_cast<unsigned short*>(Tv); g.cap = cap; g.Dp = b16s_dp(d->D); g.pitch = pitch; g.ngroups = prod_b16_groups(cp_real);
        g.nchunks16 = (d->D + kB16KC - 1) / kB16KC;
        if (pitch < g.ngroups * kB16BN || (pitch & 1)) { set_error("product table pitch %d unusable", pitch); return RBR_ERR_BAD_ARG; }
        // six or more column groups (cfg2): the rows straight from the f32 table into registers, no compact copy, two groups per
        // workgroup -- 25.4 us against 8.4 (rows_to_b16) + 24.1, the step of the class 0.329 -> 0.313 ms.  RBR_B16_ROWS_COPY=1: the
        // two-launch form below (which smaller shapes keep: they do not fill the chip with half the workgroups).
        static const bool sd_ok = getenv("RBR_B16_ROWS_COPY") == nullptr || atoi(getenv("RBR_B16_ROWS_COPY")) == 0;
        if (sd_ok && g.ngroups >= 6) {
            B16Gemm gd{};
            gd.counter = counter; gd.tok_of_row = tok_of_row; gd.table = table; gd.bimg = g.bimg; gd.T = nullptr;
            gd.cap = cap; gd.D = d->D; gd.pitch = pitch; gd.ngroups = g.ngroups; gd.nchunks = g.nchunks16;
            const int mb = ((cap + kB16BM - 1) / kB16BM + 7) / 8 * 8;
            if (int e_ = rbr::launch<prod_gemm_b16sd_kernel, kB16Threads, 2>(dim3((unsigned)(mb * ((g.ngroups + 1) / 2))), dim3(kB16Threads), kB16sdLds, st, "textcnn prod_gemm_b16sd launch", gd, g.T, g.nchunks16)) return e_;
            return 0;
        }
        const long segs = (long)cap * (g.Dp / 8);
        if (int e_ = rbr::launch<rows_to_b16_kernel, 256>(dim3((unsigned)std::min<long>((segs + 255) / 256, 4096)), dim3(256), 0, st, "textcnn rows_to_b16 launch", counter, cap, d->D, g.Dp, tok_of_row, table, static_cast<unsigned short*>(a16))) return e_;
        const int mblocks = ((cap + kB16BM - 1) / kB16BM + 7) / 8 * 8;
        if (int e_ = rbr::launch<prod_gemm_b16s_kernel, kB16Threads, 2>(dim3((unsigned)(mblocks * g.ngroups)), dim3(kB16Threads), kB16sLds, st, "textcnn prod_gemm_b16s launch", g)) return e_;
        return 0;
    }
    float* T = static_cast<float*>(Tv);
    B16Gemm g{};
    g.counter = counter; g.tok_of_row = tok_of_row; g.table = table; g.bimg = static_cast<const unsigned char*>(bimg); g.T = T;
    g.cap = cap; g.D = d->D; g.pitch = pitch; g.ngroups = prod_b16_groups(cp_real); g.nchunks = (d->D + kB16KC - 1) / kB16KC;
    if (pitch < g.ngroups * kB16BN) { set_error("product table pitch %d < %d", pitch, g.ngroups * kB16BN); return RBR_ERR_BAD_ARG; }
    // (dynamic LDS above 64 KiB: rbr::launch raises the wrapper kernels' limit at their first launch)
    const int mblocks = ((cap + kB16BM - 1) / kB16BM + 7) / 8 * 8;           // whole XCD rounds
    const dim3 grid((unsigned)(mblocks * g.ngroups)), block(kB16Threads);
    // The form with the token rows in registers and two 128-column groups per workgroup (prod_gemm_b16d_kernel) where it was
    // measured to win: six or more groups (three workgroups per row block: cfg2's 167 row blocks fill the 512 workgroup slots
    // once) and a K of at least 12 steps.  NARRE cfg3 (4 groups: 314 workgroups) 36.5 us against 32.0, D-ATT's merged conv (9
    // groups but K = 100: 7 steps) 61 against 58 -- those keep the kernel above.
    static const bool direct_ok = getenv("RBR_GEMM_ROWS_IN_LDS") == nullptr || atoi(getenv("RBR_GEMM_ROWS_IN_LDS")) == 0;
    // Short contractions (D <= 128) with at least four column groups: the rows-stationary kernel (D-ATT's merged conv: 9 groups
    // x 7 steps).  RBR_GEMM_ROWS_STATIONARY=0: the ring kernel below.
    static const bool stationary_ok = getenv("RBR_GEMM_ROWS_STATIONARY") == nullptr || atoi(getenv("RBR_GEMM_ROWS_STATIONARY")) != 0;
    if (stationary_ok && g.nchunks >= 4 && g.nchunks <= 8 && g.ngroups >= 4 && prod_precision_of(d) == RBR_PROD_BF16X3) {
        const dim3 grid_k((unsigned)(mblocks * std::min(kB16kSplit, g.ngroups)));
        constexpr int kLdsK = kB16kLds;
        switch (g.nchunks) {
            case 4: return rbr::launch<prod_gemm_b16k_kernel<6, 4>, kB16Threads, 2>(grid_k, block, kLdsK, st, "textcnn prod_gemm_b16k launch", g);
            case 5: return rbr::launch<prod_gemm_b16k_kernel<6, 5>, kB16Threads, 2>(grid_k, block, kLdsK, st, "textcnn prod_gemm_b16k launch", g);
            case 6: return rbr::launch<prod_gemm_b16k_kernel<6, 6>, kB16Threads, 2>(grid_k, block, kLdsK, st, "textcnn prod_gemm_b16k launch", g);
            case 7: return rbr::launch<prod_gemm_b16k_kernel<6, 7>, kB16Threads, 2>(grid_k, block, kLdsK, st, "textcnn prod_gemm_b16k launch", g);
            default: return rbr::launch<prod_gemm_b16k_kernel<6, 8>, kB16Threads, 2>(grid_k, block, kLdsK, st, "textcnn prod_gemm_b16k launch", g);
        }
    }
    if (direct_ok && g.ngroups >= 6 && g.nchunks >= 12) {
        const dim3 grid_d((unsigned)(mblocks * ((g.ngroups + 1) / 2)));
        switch (prod_precision_of(d)) {
            case RBR_PROD_BF16X3: return rbr::launch<prod_gemm_b16d_kernel<6, 8>, kB16Threads, 2>(grid_d, block, B16d<8>::kLds, st, "textcnn prod_gemm_b16d launch", g);
            case RBR_PROD_BF16X2: return rbr::launch<prod_gemm_b16d_kernel<3, 8>, kB16Threads, 2>(grid_d, block, B16d<8>::kLds, st, "textcnn prod_gemm_b16d launch", g);
            case RBR_PROD_BF16: return rbr::launch<prod_gemm_b16d_kernel<1, 8>, kB16Threads, 2>(grid_d, block, B16d<8>::kLds, st, "textcnn prod_gemm_b16d launch", g);
            default: set_error("prod_b16_gemm called in f32 mode"); return RBR_ERR_BAD_ARG;
        }
    }
    switch (prod_precision_of(d)) {
        case RBR_PROD_BF16X3: return rbr::launch<prod_gemm_b16_kernel<6>, kB16Threads, 2>(grid, block, kB16Lds, st, "textcnn prod_gemm_b16 launch", g);
        case RBR_PROD_BF16X2: return rbr::launch<prod_gemm_b16_kernel<3>, kB16Threads, 2>(grid, block, kB16Lds, st, "textcnn prod_gemm_b16 launch", g);
        case RBR_PROD_BF16: return rbr::launch<prod_gemm_b16_kernel<1>, kB16Threads, 2>(grid, block, kB16Lds, st, "textcnn prod_gemm_b16 launch", g);
        default: set_error("prod_b16_gemm called in f32 mode"); return RBR_ERR_BAD_ARG;
    }
}

}  // namespace rbr

extern "C" void rbr_set_prod_precision(int32_t mode) {
    rbr::g_prod_precision = (mode >= RBR_PROD_F32 && mode <= RBR_PROD_BF16) ? mode : -1;
}
extern "C" int32_t rbr_get_prod_precision(void) { return rbr::prod_precision(); }
extern "C" void rbr_textcnn_desc_stamp(rbr_textcnn_desc* d) {
    if (!d) return;
    d->flags &= ~(RBR_CONV_CLASS_STAMPED | RBR_CONV_CLASS_T_BF16 | (0x3 << 16));
    d->flags |= RBR_CONV_CLASS_STAMPED | ((rbr::prod_precision() & 0x3) << 16) | (rbr::b16_storage_setting() ? RBR_CONV_CLASS_T_BF16 : 0);
}
extern "C" void rbr_set_b16_storage(int32_t on) { rbr::g_b16_storage = (on == 0 || on == 1) ? on : -1; }
