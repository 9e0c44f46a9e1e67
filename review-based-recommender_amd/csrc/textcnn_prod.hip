// textcnn_prod.hip -- token-product formulation of the TextCNN forward and of its table / gate gradient.
//
// The conv input is an embedding LOOKUP: x[doc, p, :] = table[ids[doc, p]].  So for every tap j and channel c
//     out[doc, l, c] = bias[c] + sum_j  T[ ids[doc, l + j - pad] ][j, c],      T[t][j, c] = <table[t, :], W[c, :, j]>
// and T needs one row per DISTINCT token of the batch, not one per position.  Review text is Zipfian: the cfg2
// batch has 262 k positions but only 21 k distinct tokens, so the contraction shrinks from 118 GFLOP (dense conv,
// what the reference executes: models/deepconn/layers.py:46-60) to 9.6 GFLOP, followed by a gather-add over
// kz rows of T per position that is L2 / Infinity-Cache row traffic.  Same fp32 arithmetic, different summation order
// (d first, taps second); masked tokens and out-of-document taps contribute exactly 0, as masked_fill and the
// conv's zero padding do.
//
// Forward stages (all on the caller's stream, nothing allocated here; rbr_textcnn_prod_prepare / _table / _pool):
//   1. zero_regions                  : token marks, row mask, work-list counters
//   2. mark_scan                     : marks of the tokens that occur + work list of the documents' 32-token slabs
//   3. compact_pack                  : marks -> list `tok_of_row` / inverse map `row_of_token`; W[c, d, j] of every bank ->
//                                      one kz = 1 bank of sum(kz*ch) "product channels" (MFMA tile image) and Wprod^T rows
//   4. conv_store_kernel             : T = table[tok_of_row] @ Wprod on the f32 MFMA pipe (the fused gather + MFMA kernel
//                                      of the dense path, run over the token list as one long document, store epilogue)
//   5. gather_pool                   : per 32-position wave-tile, sum the kz rows of T, running max / first argmax
//   then pool_finalize as for the dense path.
// Backward (rbr_textcnn_bwd_dtable_prod): zero_g_rows, build_g (G[token][(tap, channel)] += g, d(gate) from T),
//   g_times_w (dtable[token, :] = G[token, :] @ Wprod^T as a sparse row product; absent tokens' rows zeroed).
#include "rbr_common.h"
#include "textcnn_b16.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>

namespace rbr {

struct ProdArgs {
    int n_docs, L, V, cap;          // cap = rows of T that may be used (distinct tokens <= min(V, positions))
    int zrow;                       // index of the all-zero row of T (= cap)
    int pitch;                      // floats per row of T (32 * product tiles)
    int poff[RBR_MAX_WIDTHS];       // first product channel of bank w: channel (w, j, cl) = poff[w] + j*ch[w] + cl
    int rank_off[RBR_MAX_WIDTHS];   // first slot of bank w in the pooling workspace (banks ordered by kernel width, stable)
    int cp_real;                    // product channels before padding to whole launch groups
};
RBR_SHARED_KERNEL_ARG(ProdArgs);

// ---------------------------------------------------------------------------------- distinct tokens
// Launch 2 of the forward chain: blocks [0, nb_scan) build the work list of the REAL documents (which 32-token slabs
// hold an unmasked token), the remaining blocks mark the tokens that occur.  Independent jobs, one launch.
__device__ __forceinline__ void mark_scan_kernel(const ConvPlan& P, int nb_scan, long n_tok, const long long* __restrict__ ids,
                                                        const unsigned char* __restrict__ mask, int* __restrict__ sched,
                                                        unsigned char* __restrict__ used) {
    if ((int)blockIdx.x < nb_scan) {
        tile_scan_block(P, mask, sched, blockIdx.x, ids);
        return;
    }
    const long nb = gridDim.x - nb_scan;
    for (long k = (long)(blockIdx.x - nb_scan) * 256 + threadIdx.x; k < n_tok; k += nb * 256)
        if (mask == nullptr || mask[k]) used[ids[k]] = 1;      // benign race: everyone stores 1 (one byte per token id)
}

// Launch 1 of the prepare stage when the caller's raw id tensors come along (rbr_textcnn_prod_prepare_ids): the id range check of
// rbr_sanitize_ids and the zero-fill of the list state, two independent jobs that were two launches.
__global__ __launch_bounds__(256) void prep_kernel(const IdSets S, long long* __restrict__ err, const ZeroRegions R, int nb_ids) {
    if ((int)blockIdx.x < nb_ids) {
        const long long total = S.first[S.count];
        for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long long)nb_ids * 256) sanitize_id(S, k, err);
        return;
    }
    const long nz = R.n[0] + R.n[1] + R.n[2], nb = gridDim.x - nb_ids;
    for (long k = (long)(blockIdx.x - nb_ids) * 256 + threadIdx.x; k < nz; k += nb * 256) {
        if (k < R.n[0]) R.p[0][k] = 0;
        else if (k < R.n[0] + R.n[1]) R.p[1][k - R.n[0]] = 0;
        else R.p[2][k - R.n[0] - R.n[1]] = 0;
    }
}

struct PackJob {        // weight images of the product formulation, both derived from the torch-layout conv weights
    ConvPlan P;         // plan of the token pseudo-document (one kz = 1 bank of Cp channels): forward image [dc][tile][slot][dd]
    int n_widths, D, cp_real;
    int kz[RBR_MAX_WIDTHS], ch[RBR_MAX_WIDTHS], poff[RBR_MAX_WIDTHS];
};

__device__ __forceinline__ float prod_weight(const PackJob& J, const PtrArray& W, int pc, int d) {
    int w = 0;
#pragma unroll
    for (int k = 1; k < RBR_MAX_WIDTHS; ++k)
        if (k < J.n_widths && pc >= J.poff[k]) w = k;
    const int rel = pc - J.poff[w];
    const int j = rel / J.ch[w], cl = rel - j * J.ch[w];
    return W.p[w][((long)cl * J.D + d) * J.kz[w] + j];
}

// Launch 3: blocks [0, nb_compact) turn the marks into the token list (row_of_token[v] = dense row or -1,
// tok_of_row / row_mask describe the pseudo-document, *counter = rows); the remaining blocks write the product weight
// image of the forward GEMM and the row-major Wprod^T the backward's sparse product reads.
__device__ __forceinline__ void compact_pack_kernel(const PackJob& J, int nb_compact, int V, int cap,
                                                           const unsigned char* __restrict__ used, int* __restrict__ row_of_token,
                                                           long long* __restrict__ tok_of_row, unsigned char* __restrict__ row_mask,
                                                           int* __restrict__ counter, float* __restrict__ zero_row, int pitch,
                                                           const PtrArray& W, float* __restrict__ packed, float* __restrict__ WT,
                                                           const B16Pack& JB, unsigned char* __restrict__ bimg) {
    if ((int)blockIdx.x < nb_compact) {
        if (blockIdx.x == 0)      // the all-zero row of T that masked / out-of-document taps read
            for (int k = threadIdx.x; k < pitch; k += 256) zero_row[k] = 0.f;
        // dense row of token v = number of marked tokens below v: rows come out in token order, the same list for the
        // same batch on every run (no atomics).  A block first counts the marks below its 256 tokens (<= V ints, L2).
        __shared__ int s_cnt[4], s_wave[4];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int below = blockIdx.x * 256;
        int c = 0;
        const int4* u4 = reinterpret_cast<const int4*>(used);          // 16 marks (bytes of value 0 / 1) per load
        for (int q = threadIdx.x; q < below / 16; q += 256) {
            const int4 m = u4[q];
            c += __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        const int v = below + threadIdx.x;
        const int u = (v < V) ? used[v] : 0;
        const unsigned long long b = __ballot(u);
        if (lane == 0) { s_cnt[wave] = c; s_wave[wave] = __popcll(b); }
        __syncthreads();
        int base = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        for (int k = 0; k < wave; ++k) base += s_wave[k];
        if (v < V) {
            int row = -1;
            if (u) {
                row = base + __popcll(b & ((1ull << lane) - 1));
                if (row < cap) { tok_of_row[row] = v; row_mask[row] = 1; } else row = -1;   // cannot happen: cap >= distinct
            }
            row_of_token[v] = row;
        }
        if ((int)blockIdx.x == nb_compact - 1 && threadIdx.x == 0)
            *counter = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]) + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        return;
    }
    const long nb = gridDim.x - nb_compact, b0 = blockIdx.x - nb_compact;
    const ConvPlan& P = J.P;
    const int DC = P.DC;
    const long n_img = packed != nullptr ? (long)P.nchunks * P.tiles_total * kTile * DC : 0;     // bf16-plane GEMM: no f32 image
    const long n_wt = (long)J.cp_real * J.D;
    const long n_b16 = bimg != nullptr ? b16_pack_items(JB) : 0;                                 // bf16-plane GEMM: its weight planes
    for (long idx = b0 * 256 + threadIdx.x; idx < n_img + n_wt + n_b16; idx += nb * 256) {
        if (idx >= n_img + n_wt) {
            b16_pack_item(JB, W, bimg, idx - n_img - n_wt);
        } else if (idx < n_img) {
            long r = idx;
            const int dd = (int)(r % DC); r /= DC;
            const int slot = (int)(r % kTile); r /= kTile;
            const int t = (int)(r % P.tiles_total);
            const int dc = (int)(r / P.tiles_total);
            const int pc = t * kTile + slot;          // product channel
            const int d = dc * DC + dd;
            packed[idx] = (pc < J.cp_real && d < J.D) ? prod_weight(J, W, pc, d) : 0.f;
        } else {
            const long e = idx - n_img;
            const int pc = (int)(e / J.D), d = (int)(e - (long)pc * J.D);
            WT[e] = prod_weight(J, W, pc, d);         // WT[(w, j, cl)][d] = W_w[cl, d, j]
        }
    }
}

// ---------------------------------------------------------------------------------- gather + pool
// One wave per active 32-position slab (work list of the tile scan).  A lane owns FOUR consecutive channels of one
// bank at every fourth position: lane = (position mod 4, channel quad), so one float4 load instruction moves
// 4 positions x 16 quads of one (bank, tap) segment of T -- the 200-byte segments of a 50-channel bank are read whole.
// For each of its positions the lane adds its kz product rows (x gate), keeps the running max / first argmax, and the
// four position classes meet by shuffle.  Writes the same (max, first argmax) partials as the conv kernel's epilogue.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));    // segments start at any float

// (Measured and dropped in round 3: clearing the backward's G with extra workgroups of this launch -- 59.6 us against 42.7 + 10.9
// for the separate zero launch, and, worse, G is then cold when the backward reads it: g_times_w 59 -> 72 us.)
typedef unsigned u32x2u __attribute__((ext_vector_type(2), aligned(2)));  // four bf16 of a segment that starts at any element
// TB16: T holds bf16 (the plain-bf16 class with bf16 storage, textcnn_prod_b16.hip): a position's row segments are half the bytes
// -- this kernel is bound by L2 requests for exactly those -- and are widened to f32 in registers; the sums stay f32.
// (Measured and dropped, round 3: a TB16 form whose lanes own EIGHT channels -- 8 position classes x 8 octets, one 16-byte load
// where this kernel issues two 8-byte ones.  52 us against 39 us at cfg2: halving the load instructions does not help when
// the same row segments, 100 bytes at 4-byte alignment, are fetched by wider and fewer lanes.  The f32 kernel's counters
// (profiles/r03_gather_pool_pmc.json) say what bounds it instead: its L2 requests equal the distinct 128-byte lines of the
// rows a tile reads (the L1 absorbs the kz-fold re-reads), 27 % of them miss, and T (44 MB; 22 MB in bf16) fits no XCD's L2.)
// (Also measured and dropped: the tap loop unrolled by a compile-time width.  As compiled, every tap of the guarded loop below is a
// block of its own -- two loads, wait, four FMAs -- so a lane has two loads in flight; unrolled, eight, at 151 registers instead of
// 62: 53 us against 44 at cfg2, 218 against 152 for D-ATT.  Eight waves per SIMD with two loads each already keep the L2 at its
// rate; fewer waves with more loads each do not.)
template <bool TB16>
__device__ __forceinline__ void gather_pool_kernel(const ConvPlan& P, const ProdArgs& A, const long long* __restrict__ ids,
                                                          const unsigned char* __restrict__ mask, const float* __restrict__ gate,
                                                          const int* __restrict__ row_of_token, const void* __restrict__ Tv,
                                                          const int* __restrict__ sched, float* __restrict__ pval,
                                                          int* __restrict__ pidx) {
    const float* T = static_cast<const float*>(Tv);
    const unsigned short* T16 = static_cast<const unsigned short*>(Tv);
    __shared__ int s_row[kWavesPerWG][kTile + kMaxKF];
    __shared__ float s_gate[kWavesPerWG][kTile + kMaxKF];
    __shared__ float s_gate_b[kWavesPerWG][kTile + kMaxKF];       // RBR_CONV_GATE_SPLIT: the gate of the banks >= P.gate_split
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_active = sched[2 * (long)P.total_wt];
    const int slot_in_list = blockIdx.x * kWavesPerWG + wave;
    if (slot_in_list >= n_active) return;                       // wave-uniform; no block barrier below
    const int wt = sched[P.total_wt + slot_in_list];
    const int doc = wt / P.wpd, l0 = (wt % P.wpd) * kTile;
    const int L = P.L, XR = kTile + P.KF - 1;
    for (int r = lane; r < XR; r += 64) {
        const int p = l0 - P.P + r;
        int row = A.zrow;
        float gv = 1.f, gvb = 1.f;
        if (p >= 0 && p < L) {
            const long tok = (long)doc * L + p;
            if (mask == nullptr || mask[tok]) {
                const int rr = row_of_token[ids[tok]];
                row = rr >= 0 ? rr : A.zrow;
                if (gate != nullptr) {
                    gv = gate[tok];
                    if (P.gate_split > 0) gvb = gate[(long)P.n_docs * L + tok];
                }
            }
        }
        s_row[wave][r] = row;
        s_gate[wave][r] = gv;
        s_gate_b[wave][r] = gvb;
    }
    __builtin_amdgcn_wave_barrier();
    const float NEG = -__builtin_huge_valf();
    for (int w = 0; w < P.n_widths; ++w) {
        const int kz = P.kz[w], ch = P.ch[w];
        const int off = (P.pad_mode == RBR_PAD_SAME) ? (P.KF - kz) / 2 : 0;      // frame tap of the bank's tap 0
        const int Lv = (P.pad_mode == RBR_PAD_VALID) ? (L - kz + 1) : L;          // pool length of this bank
        const int nquads = (ch + 3) >> 2;
        // Lanes per position = the bank's quads rounded up to 16 / 32 / 64 (wave-uniform), so that ONE load instruction covers a
        // (row, tap) segment of the bank whole whenever it can: with a fixed 16 quads (256 bytes) per pass, a 100-channel bank's
        // 400-byte segments were fetched as a 256-byte and a 144-byte window in two passes -- each window rounded out to whole
        // 128-byte lines, 1.5x the segment's bytes in L2 requests (D-ATT cfg4: 2.34 GB of requests for 1.44 GB of segments);
        // a 50-channel bank (cfg2: 13 quads) always was one window.
        const int lq = nquads <= 16 ? 4 : (nquads <= 32 ? 5 : 6);               // log2(lanes per position)
        const int QL = 1 << lq, PS = 64 >> lq;                                  // quads per pass, position classes
        const int ps = lane >> lq, ql = lane & (QL - 1);
        const float* sg = (P.gate_split > 0 && w >= P.gate_split) ? s_gate_b[wave] : s_gate[wave];     // wave-uniform
        for (int qb = 0; qb < nquads; qb += QL) {
            const int q = qb + ql;
            const bool valid = q < nquads;
            const float* tcol = T + A.poff[w] + 4 * (valid ? q : 0);
            const unsigned short* tcol16 = T16 + A.poff[w] + 4 * (valid ? q : 0);
            float best[4] = {NEG, NEG, NEG, NEG};
            int bidx[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
            for (int p0 = 0; p0 < kTile; p0 += 2 * PS) {                  // two positions per lane and round:
                f32x4 y[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // 2 * kz independent float4 loads in flight
#pragma unroll
                for (int j = 0; j < kMaxKF; ++j) {
                    if (j < kz) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int r = p0 + PS * u + ps + j + off;
                            f32x4 v;
                            if (TB16) {
                                const u32x2u b = *reinterpret_cast<const u32x2u*>(tcol16 + (long)s_row[wave][r] * A.pitch + j * ch);
                                v = f32x4{__uint_as_float(b.x << 16), __uint_as_float(b.x & 0xffff0000u),
                                          __uint_as_float(b.y << 16), __uint_as_float(b.y & 0xffff0000u)};
                            } else {
                                v = *reinterpret_cast<const f32x4u*>(tcol + (long)s_row[wave][r] * A.pitch + j * ch);
                            }
                            const float gv = sg[r];
                            y[u].x = fmaf(v.x, gv, y[u].x); y[u].y = fmaf(v.y, gv, y[u].y);
                            y[u].z = fmaf(v.z, gv, y[u].z); y[u].w = fmaf(v.w, gv, y[u].w);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int pos = l0 + p0 + PS * u + ps;
                    if (pos < Lv) {                                       // ascending positions per lane: '>' keeps the first max
                        if (y[u].x > best[0]) { best[0] = y[u].x; bidx[0] = pos; }
                        if (y[u].y > best[1]) { best[1] = y[u].y; bidx[1] = pos; }
                        if (y[u].z > best[2]) { best[2] = y[u].z; bidx[2] = pos; }
                        if (y[u].w > best[3]) { best[3] = y[u].w; bidx[3] = pos; }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {                                 // meet the position classes
                for (int o = QL; o < 64; o <<= 1) {
                    const float ob = __shfl_xor(best[c], o);
                    const int oi = __shfl_xor(bidx[c], o);
                    if (ob > best[c] || (ob == best[c] && oi < bidx[c])) { best[c] = ob; bidx[c] = oi; }
                }
            }
            if (valid && ps == 0) {
                const long o = (long)wt * P.nslots_total + A.rank_off[w] + 4 * q;      // kz-sorted slot of the channel
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * q + c < ch) { pval[o + c] = best[c]; pidx[o + c] = bidx[c]; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------- backward: dtable = G @ Wprod^T
// The table gradient of the max-pooled conv (textcnn_bwd.hip) is  dtable[t, :] = sum_{(w,j,cl)} G[t][(w,j,cl)] * W_w[cl, :, j]
// with G[t][(w,j,cl)] = sum of g = d_feat * act'(feat) over the (doc, channel) windows whose tap j sits on token t.
// G has one row per DISTINCT token (the forward's list).  Building it costs one scalar atomic per (doc, channel, tap)
// and MERGES repeated (token, tap, channel) hits, so a Zipf-hot token costs at most sum(kz*ch) weight rows however
// often it occurs.  G is ~2 % dense: each wave compacts the non-zeros of one row into LDS and adds the matching
// rows of Wprod^T (900 KB, L2-resident) in registers, then writes the table row ONCE with plain stores -- instead of
// ~6.5 rows of f32 atomics per token in the window scatter.
struct ProdBwdArgs {
    int n_docs, L, C, KF, KG, D, cap, padding_idx, pad_mode, act;
    int t_pitch;                    // floats per row of the forward's product table T (gated convs: d(gate) reads it)
    int gate_split;                 // RBR_CONV_GATE_SPLIT: banks >= this read / write plane 1 of gate / dgate (0: one plane)
    int n_widths;
    int kz[RBR_MAX_WIDTHS], ch[RBR_MAX_WIDTHS], ch_off[RBR_MAX_WIDTHS], poff[RBR_MAX_WIDTHS];
};
RBR_SHARED_KERNEL_ARG(ProdBwdArgs);

// ... and the gate gradient build_g accumulates into (n_dgate floats, 0 for un-gated convs): one launch for both
// ... and (row-GEMM shapes) the bf16 planes of Wprod^T that GEMM reads: blocks past nb_zero, from the WT the forward's pack left
__device__ __forceinline__ void zero_g_rows_kernel(const int* __restrict__ counter, int cap, int KG4, f32x4* __restrict__ G,
                                                          float* __restrict__ dgate, long n_dgate, int nb_zero, const float* __restrict__ WT,
                                                          int cp_real, int D, long n_img, unsigned char* __restrict__ bimg_t) {
    if ((int)blockIdx.x >= nb_zero) {
        const int nchunks = (4 * KG4 + kB16KC - 1) / kB16KC;
        for (long k = (long)(blockIdx.x - nb_zero) * 256 + threadIdx.x; k < n_img; k += (long)(gridDim.x - nb_zero) * 256)
            b16_pack_wt_item(WT, cp_real, D, nchunks, bimg_t, k);
        return;
    }
    const long n = (G != nullptr) ? (long)min(*counter, cap) * KG4 : 0;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)nb_zero * 256) G[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n_dgate; k += (long)nb_zero * 256) dgate[k] = 0.f;
}

// one thread per (doc, channel, tap).  Gated convs (D-ATT: x = gate[doc,p] * table[id]): the token's share is g * gate, and
// d(gate[doc,p]) += g * <W[c,:,j], table[id]> = g * T[token][(w,j,c)], read from the forward's product table.
__device__ __forceinline__ void build_g_kernel(const ProdBwdArgs& A, const long long* __restrict__ ids,
                                                      const unsigned char* __restrict__ mask, const float* __restrict__ gate,
                                                      const int* __restrict__ row_of_token, const float* __restrict__ T,
                                                      const float* __restrict__ feat, const int* __restrict__ argmax,
                                                      const float* __restrict__ d_feat, float* __restrict__ G,
                                                      float* __restrict__ dgate, int t_bf16) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long o = e / A.KF;
    const int j = (int)(e - o * A.KF);
    if (o >= (long)A.n_docs * A.C) return;
    const int doc = (int)(o / A.C), c = (int)(o - (long)doc * A.C);
    int w = 0;
#pragma unroll
    for (int k = 1; k < RBR_MAX_WIDTHS; ++k)
        if (k < A.n_widths && c >= A.ch_off[k]) w = k;
    const int kz = A.kz[w];
    if (j >= kz) return;
    const float g = act_grad(A.act, feat[o], d_feat[o]);
    if (g == 0.f) return;
    const int padl = (A.pad_mode == RBR_PAD_SAME) ? (kz - 1) / 2 : 0;
    const int p = argmax[o] + j - padl;
    if (p < 0 || p >= A.L) return;
    const long tok = (long)doc * A.L + p;
    if (mask != nullptr && !mask[tok]) return;          // masked token: x was zeroed, no gradient
    const long long t = ids[tok];
    const int row = row_of_token[t];
    if (row < 0) return;
    const int col = A.poff[w] + j * A.ch[w] + (c - A.ch_off[w]);
    const long gtok = (A.gate_split > 0 && w >= A.gate_split) ? (long)A.n_docs * A.L + tok : tok;      // the bank's gate plane
    if (dgate != nullptr) {
        const long at = (long)row * A.t_pitch + col;
        const float tv = t_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(T)[at] << 16) : T[at];
        atomicAdd(dgate + gtok, g * tv);
    }
    if (G == nullptr) return;
    // the pad token's row stays in G (the weight gradient needs it); its TABLE row gets no gradient: g_times_w zeroes it
    atomicAdd(G + (long)row * A.KG + col, (gate != nullptr) ? g * gate[gtok] : g);
}

constexpr int kSpQ4 = 2;      // float4 columns per lane and pass: 512 floats of the table row

// One WORKGROUP per distinct-token row (grid-stride): wave w compacts and multiplies the w-th quarter of the row's
// columns, the four partial rows meet in LDS.  A Zipf-hot token ("the": every one of the sum(kz*ch) columns is
// non-zero) would otherwise keep a single wave busy longer than the rest of the kernel takes.
// Dynamic LDS: per wave KGW (int offset, float value) pairs + [4][D] partial sums.
enum { kGtwDense = 0, kGtwAccumulate = 1, kGtwRows = 2 };
constexpr int kGtwMaxBlocks = 8192;      // workgroups of g_times_w = entries of its sq_part
template <int MODE>
__device__ __forceinline__ void g_times_w_kernel(const ProdBwdArgs& A, const int KGW, const int* __restrict__ counter,
                                                        const float* __restrict__ G, const float* __restrict__ WT,
                                                        const long long* __restrict__ tok_of_row,
                                                        const int* __restrict__ row_of_token, int V, float* __restrict__ dtable,
                                                        float* __restrict__ sq_part) {
    // kGtwAccumulate: dtable is a gradient buffer shared with other producers on this stream (functional.table_fanout) -- the
    // rows of the batch's tokens are ADDED to it, nothing else is touched (no zero rows, the pad row is skipped).  A template
    // parameter: as a run-time flag it cost the plain form 12 registers and one wave per SIMD (63 -> 71 us at cfg2).
    // kGtwRows: the gradient in COMPACT form -- row r of `dtable` is the gradient of token tok_of_row[r] (the rows of absent
    // tokens are not written anywhere: rbr_clip_adam_step_rows takes them as zero), and sq_part[workgroup] = the sum of squares
    // of the rows this workgroup wrote (static row -> workgroup -> thread mapping: the same bits on every run).
    // (Measured and dropped, round 4: a WAVE per row -- the row scanned, listed and multiplied out by one wave, 8 weight-row loads in
    // flight, rows of <= 32 quads two per instruction; rows with more than 64 non-zeros marked and left to this kernel, launched
    // behind it with a filter.  Same results (191 tests).  cfg2: 43 us + 28 us for the marked rows -- the Zipf head, 750 non-zeros
    // in the first rows of the list, which THIS kernel starts at t = 0 beside everything else and the split runs as a tail --
    // against 46 us; D-ATT 93 against 88.  Four rows in flight per workgroup do not make the weight rows arrive faster.)
    // (Measured and dropped, round 4: the NEXT row's quarter of G fetched by LDS-DMA while this row is worked on -- two buffers per
    // wave, no registers, the wait placed in front of the previous row's stores.  cfg2: the step +3.5 us; D-ATT cfg4: 89 -> 86 us.
    // The ~5 us a workgroup spends per row are not the G read: they are the chain list -> weight rows -> barrier -> store.)
    constexpr bool accumulate = MODE == kGtwAccumulate;
    constexpr bool compact = MODE == kGtwRows;
    float sq = 0.f;
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int* s_pc = s_dyn + wave * 2 * KGW;
    float* s_val = reinterpret_cast<float*>(s_pc + KGW);
    float* s_part = reinterpret_cast<float*>(s_dyn + kWavesPerWG * 2 * KGW);     // [4][D]
    const int n = min(*counter, A.cap);
    const int D = A.D, nq4 = D >> 2;
    const unsigned long long lt = (1ull << lane) - 1;
    const int kbeg = wave * KGW, kend = min(A.KG, kbeg + KGW);      // this wave's columns (KGW % 4 == 0)
    for (int row = blockIdx.x; row < n; row += gridDim.x) {
        if (tok_of_row[row] == A.padding_idx) {             // nn.Embedding(padding_idx): that row gets no gradient
            float* dst = dtable + (long)(compact ? row : A.padding_idx) * D;
            if (!accumulate)
                for (int q4 = threadIdx.x; q4 < nq4; q4 += 256) *reinterpret_cast<f32x4*>(dst + 4 * q4) = f32x4{0.f, 0.f, 0.f, 0.f};
            continue;                                        // workgroup-uniform
        }
        // accumulate mode: what the shared buffer holds for this row is requested now and added at the end (a read in front
        // of the store would put one more memory latency on every row)
        const long trow = compact ? (long)row * D : (long)tok_of_row[row] * D;
        f32x4 old = {0.f, 0.f, 0.f, 0.f};
        if (accumulate && (int)threadIdx.x < nq4) old = *reinterpret_cast<const f32x4*>(dtable + trow + 4 * threadIdx.x);
        // 1. non-zeros of this wave's quarter of the row -> (weight-row offset, value) list
        int cnt = 0;
        for (int k0 = kbeg; k0 < kend; k0 += 256) {
            const int k = k0 + 4 * lane;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < kend) v = *reinterpret_cast<const f32x4*>(G + (long)row * A.KG + k);
#pragma unroll
            for (int cpt = 0; cpt < 4; ++cpt) {
                const bool nz = v[cpt] != 0.f;
                const unsigned long long b = __ballot(nz);
                if (nz) {
                    const int pos = cnt + __popcll(b & lt);
                    s_pc[pos] = (k + cpt) * D;
                    s_val[pos] = v[cpt];
                }
                cnt += __popcll(b);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // 2. partial row = sum val * WT[column, :]   (4 weight rows in flight per wave)
        for (int qblk = 0; qblk < nq4; qblk += 64 * kSpQ4) {
            f32x4 sum[kSpQ4];
            int doff[kSpQ4];
#pragma unroll
            for (int u = 0; u < kSpQ4; ++u) {
                const int q4 = qblk + lane + 64 * u;
                doff[u] = (q4 < nq4) ? 4 * q4 : -1;
                sum[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int q0 = 0; q0 < cnt; q0 += 4) {
                f32x4 wv[4][kSpQ4];
                float gv[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const bool ok = q0 + v < cnt;
                    const int it = ok ? q0 + v : q0;
                    const float* wrow = WT + s_pc[it];
                    gv[v] = ok ? s_val[it] : 0.f;
#pragma unroll
                    for (int u = 0; u < kSpQ4; ++u)
                        wv[v][u] = (doff[u] >= 0) ? *reinterpret_cast<const f32x4*>(wrow + doff[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int u = 0; u < kSpQ4; ++u) sum[u] += gv[v] * wv[v][u];
            }
#pragma unroll
            for (int u = 0; u < kSpQ4; ++u)
                if (doff[u] >= 0) *reinterpret_cast<f32x4*>(s_part + wave * D + doff[u]) = sum[u];
        }
        __syncthreads();
        // 3. the four partial rows, summed in wave order, written once
        for (int q4 = threadIdx.x; q4 < nq4; q4 += 256) {
            f32x4 r = *reinterpret_cast<const f32x4*>(s_part + 4 * q4);
#pragma unroll
            for (int w = 1; w < kWavesPerWG; ++w) r += *reinterpret_cast<const f32x4*>(s_part + w * D + 4 * q4);
            if (accumulate) r += (q4 == (int)threadIdx.x) ? old : *reinterpret_cast<const f32x4*>(dtable + trow + 4 * q4);   // D > 1024: later quads read late
            if (compact) sq += (r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w);
            *reinterpret_cast<f32x4*>(dtable + trow + 4 * q4) = r;
        }
        __syncthreads();      // lists and partial rows are rewritten for the next row
    }
    if (compact) {            // every workgroup writes its partial (zero when it had no rows): fixed-order reduction later
        __shared__ float s_sq[kWavesPerWG];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        if (lane == 0) s_sq[wave] = sq;
        __syncthreads();
        if (threadIdx.x == 0) sq_part[blockIdx.x] = (s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3]);
        return;
    }
    // rows of tokens the batch does not contain: zero (the caller need not pre-fill dtable); one wave per row
    if (row_of_token != nullptr && !accumulate) {
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        for (int v = blockIdx.x * kWavesPerWG + wv; v < V; v += gridDim.x * kWavesPerWG) {
            if (row_of_token[v] >= 0) continue;                 // wave-uniform
            float* dst = dtable + (long)v * D;
            for (int q4 = ln; q4 < nq4; q4 += 64) *reinterpret_cast<f32x4*>(dst + 4 * q4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}


// ---------------------------------------------------------------------------------- data-parallel tap exchange
// Across data-parallel ranks the table gradient is  dtable = ((1/N) sum_r G_r) @ Wprod^T  with the SAME Wprod on every rank,
// and G_r has one non-zero per (document, channel, tap): n_docs * sum(kz*ch) (token, value) pairs -- 3 MB at the cfg2 shape
// against the 60 MB dense gradient.  Ranks therefore all-gather their taps (fixed size, column implied by the position in
// the array) and each rebuilds the averaged gradient locally: marks -> union token list, 64-bit fixed-point G (integer
// atomics: order-independent, so replicas stay bit-identical), sparse product as above.
constexpr float kTapScale = 1099511627776.f;       // 2^40 units per 1.0: +-8.4e6 range, 9e-13 resolution

__global__ __launch_bounds__(256) void taps_kernel(const ProdBwdArgs A, const long long* __restrict__ ids,
                                                   const unsigned char* __restrict__ mask, const float* __restrict__ feat,
                                                   const int* __restrict__ argmax, const float* __restrict__ d_feat,
                                                   int* __restrict__ tok_out, float* __restrict__ val_out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long o = e / A.KF;
    const int j = (int)(e - o * A.KF);
    if (o >= (long)A.n_docs * A.C) return;
    const int doc = (int)(o / A.C), c = (int)(o - (long)doc * A.C);
    int w = 0;
#pragma unroll
    for (int k = 1; k < RBR_MAX_WIDTHS; ++k)
        if (k < A.n_widths && c >= A.ch_off[k]) w = k;
    const int kz = A.kz[w];
    int tok = -1;
    float val = 0.f;
    if (j < kz) {
        const float g = act_grad(A.act, feat[o], d_feat[o]);
        const int p = argmax[o] + j - ((A.pad_mode == RBR_PAD_SAME) ? (kz - 1) / 2 : 0);
        if (g != 0.f && p >= 0 && p < A.L) {
            const long pos = (long)doc * A.L + p;
            if (mask == nullptr || mask[pos]) {
                const long long t = ids[pos];
                if (t != A.padding_idx) { tok = (int)t; val = g; }
            }
        }
    }
    tok_out[e] = tok;
    val_out[e] = val;
}

// sort key = token (absent taps -> V, behind every token); payload = (product column, value)
__global__ __launch_bounds__(256) void taps_keys_kernel(const ProdBwdArgs A, long n_items, int n_sets, int V,
                                                        const int* __restrict__ tok, const float* __restrict__ val,
                                                        int* __restrict__ keys, unsigned long long* __restrict__ pay) {
    const long total = n_items * n_sets;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long)gridDim.x * 256) {
        const int t = tok[k];
        const long e = k % n_items;
        const long o = e / A.KF;
        const int j = (int)(e - o * A.KF), c = (int)(o % A.C);
        int w = 0;
#pragma unroll
        for (int q = 1; q < RBR_MAX_WIDTHS; ++q)
            if (q < A.n_widths && c >= A.ch_off[q]) w = q;
        const int col = A.poff[w] + min(j, A.kz[w] - 1) * A.ch[w] + (c - A.ch_off[w]);
        keys[k] = (t >= 0) ? t : V;
        pay[k] = ((unsigned long long)(unsigned)col << 32) | (unsigned)__float_as_int(val[k]);
    }
}

// ---- owner partition of the rebuild: rank r turns only the taps of the tokens t with t % n_sets == r into rows (local row
// t / n_sets of its slab); the slabs are exchanged afterwards (distributed.OwnerExchange).  The rank's taps are first
// compacted -- stably, in two passes over fixed chunks, so a run repeats bit for bit -- to the front of the key / payload
// arrays; what follows (sort, bounds, rows) runs on `cap_own` entries and ceil(V / n_sets) rows instead of all of them.
constexpr int kOwnChunks = 2048;
// entries the owner's sort is sized for: twice the even share (the token -> owner map is t % n_sets; the Zipf head makes the
// shares uneven) plus slack; never more than all of them
inline long taps_owner_cap(long total, int n_sets) { return std::min(total, 2 * ((total + n_sets - 1) / n_sets) + 4096); }
__device__ __forceinline__ long own_chunk(long total) { return ((total + kOwnChunks - 1) / kOwnChunks + 255) / 256 * 256; }

__global__ __launch_bounds__(256) void taps_owner_count_kernel(long total, int n_sets, int rank, const int* __restrict__ tok,
                                                               int* __restrict__ blk_cnt) {
    __shared__ int s_c[kWavesPerWG];
    const long chunk = own_chunk(total), lo = (long)blockIdx.x * chunk, hi = min(total, lo + chunk);
    int c = 0;
    for (long k = lo + threadIdx.x; k < hi; k += 256) {
        const int t = tok[k];
        c += (t >= 0 && t % n_sets == rank) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]);
}

// key = local row t / n_sets; the entries past the rank's count up to cap_own get the key v_own (behind every row).
// *overflow = 1 when the rank owns more than cap_own taps (the surplus is dropped: the caller must treat the step as failed).
__global__ __launch_bounds__(256) void taps_owner_keys_kernel(const ProdBwdArgs A, long n_items, int n_sets, int rank, int v_own,
                                                              long cap_own, const int* __restrict__ tok,
                                                              const float* __restrict__ val, const int* __restrict__ blk_cnt,
                                                              int* __restrict__ keys, unsigned long long* __restrict__ pay,
                                                              int* __restrict__ overflow) {
    __shared__ long s_red[2][kWavesPerWG];
    __shared__ int s_wcnt[kWavesPerWG];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long total = n_items * n_sets;
    long before = 0, all = 0;
    for (int i = tid; i < kOwnChunks; i += 256) {
        const int c = blk_cnt[i];
        all += c;
        if (i < (int)blockIdx.x) before += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); all += __shfl_xor(all, o); }
    if (lane == 0) { s_red[0][wave] = before; s_red[1][wave] = all; }
    __syncthreads();
    before = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
    all = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
    if (blockIdx.x == 0 && tid == 0 && all > cap_own) *overflow = 1;
    const long chunk = own_chunk(total), lo = (long)blockIdx.x * chunk, hi = min(total, lo + chunk);
    long run = before;                                         // output position of the chunk's next kept entry
    for (long k0 = lo; k0 < hi; k0 += 256) {                   // workgroup-uniform trip count
        const long k = k0 + tid;
        int t = -1;
        if (k < hi) t = tok[k];
        const bool keep = t >= 0 && t % n_sets == rank;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
        __syncthreads();
        int wbase = 0, wall = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerWG; ++w) { if (w < wave) wbase += s_wcnt[w]; wall += s_wcnt[w]; }
        if (keep) {
            const long pos = run + wbase + __popcll(m & ((1ull << lane) - 1));
            if (pos < cap_own) {
                const long e = k % n_items;
                const long o = e / A.KF;
                const int j = (int)(e - o * A.KF), c = (int)(o % A.C);
                int w = 0;
#pragma unroll
                for (int q = 1; q < RBR_MAX_WIDTHS; ++q)
                    if (q < A.n_widths && c >= A.ch_off[q]) w = q;
                const int col = A.poff[w] + min(j, A.kz[w] - 1) * A.ch[w] + (c - A.ch_off[w]);
                keys[pos] = t / n_sets;
                pay[pos] = ((unsigned long long)(unsigned)col << 32) | (unsigned)__float_as_int(val[k]);
            }
        }
        run += wall;
        __syncthreads();
    }
    for (long pos = min(all, cap_own) + (long)blockIdx.x * 256 + tid; pos < cap_own; pos += (long)gridDim.x * 256) {
        keys[pos] = v_own;
        pay[pos] = 0ull;
    }
}

// first / one-past-last sorted position of every token that has taps (start1 = first + 1, 0 = none)
__global__ __launch_bounds__(256) void taps_bounds_kernel(long total, int V, const int* __restrict__ keys, int* __restrict__ start1,
                                                          int* __restrict__ end) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = keys[i];
        if (k >= V) continue;
        if (i == 0 || keys[i - 1] != k) start1[k] = (int)i + 1;
        if (i == total - 1 || keys[i + 1] != k) end[k] = (int)i + 1;
    }
}

// Wprod^T rows for the rebuild (the forward's copy lives in a per-call workspace the exchange does not see)
// ... and the work items (token, part, parts, slot) of the tokens with more than kTapsSplit taps: each is summed by several
// workgroups of taps_rows_kernel (adjacent in the list, so they run side by side) that meet in the global row `slot`.
constexpr int kTapsCold = 16;        // wave-per-token rows: at most 4 dependent rounds of weight-row reads
constexpr int kTapsSplit = 4096;     // taps of one token summed by one workgroup
__global__ __launch_bounds__(256) void taps_wt_kernel(const PackJob J, const PtrArray W, float* __restrict__ WT, int V,
                                                      const int* __restrict__ start1, const int* __restrict__ end,
                                                      int* __restrict__ hot_count, int4* __restrict__ hot_list) {
    const long n_wt = (long)J.cp_real * J.D;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n_wt; e += (long)gridDim.x * 256) {
        const int pc = (int)(e / J.D), d = (int)(e - (long)pc * J.D);
        WT[e] = prod_weight(J, W, pc, d);
    }
    for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
        const int s1 = start1[v];
        const int cnt = (s1 == 0) ? 0 : end[v] - (s1 - 1);
        if (cnt > kTapsSplit) {
            const int parts = (cnt + kTapsSplit - 1) / kTapsSplit;
            const int base = atomicAdd(hot_count, parts);
            const int slot = atomicAdd(hot_count + 1, 1);
            for (int p = 0; p < parts; ++p) hot_list[base + p] = int4{v, p, parts, slot};
        }
    }
}

// Rows of the tokens with at most kTapsCold taps (and the zero rows of the tokens with none): one WAVE per vocabulary entry.
// Every rank holds the same sorted tap array (the all-gather delivers the same bytes everywhere and the sort is stable), so a
// sum taken in ARRAY ORDER by one wave is the same bits on every rank without fixed point: lane i keeps tap i, the wave walks
// the taps 4 at a time (4 weight rows in flight) and adds val/N * WT[col, :] in registers.  No LDS row, no compaction, no
// barrier: at 8 ranks' taps 35 k of the 45.6 k touched tokens are of this kind, and a workgroup spent ~3 us of fixed costs
// (zeroing and scanning a 750-cell row, four barriers) on each.  Runs as the second role of taps_rows_kernel's launch.
__device__ __forceinline__ void taps_cold_row(const ProdBwdArgs& A, int v, int lane, const int* __restrict__ start1,
                                              const int* __restrict__ end, const unsigned long long* __restrict__ pay,
                                              float inv_sets, const float* __restrict__ WT, float* __restrict__ dtable) {
    const int D = A.D, nq4 = D >> 2;
    float* drow = dtable + (long)v * D;
    const int s1 = start1[v];
    const int s = s1 - 1, cnt = (s1 == 0) ? 0 : end[v] - s;
    if (cnt > kTapsCold) return;                     // the workgroup role of taps_rows_kernel takes these
    const unsigned long long mine = (lane < cnt) ? pay[s + lane] : 0ull;
    const int my_off = (int)(unsigned)(mine >> 32) * D;
    const float my_val = __int_as_float((int)(unsigned)(mine & 0xffffffffu)) * inv_sets;
    for (int qblk = 0; qblk < nq4; qblk += 64 * kSpQ4) {
        f32x4 sum[kSpQ4];
        int doff[kSpQ4];
#pragma unroll
        for (int u = 0; u < kSpQ4; ++u) {
            const int q4 = qblk + lane + 64 * u;
            doff[u] = (q4 < nq4) ? 4 * q4 : -1;
            sum[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int q0 = 0; q0 < cnt; q0 += 4) {
            f32x4 wv[4][kSpQ4];
            float gv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool ok = q0 + t < cnt;
                const int it = ok ? q0 + t : q0;
                const float* wrow = WT + __shfl(my_off, it);
                gv[t] = ok ? __shfl(my_val, it) : 0.f;
#pragma unroll
                for (int u = 0; u < kSpQ4; ++u)
                    wv[t][u] = (doff[u] >= 0) ? *reinterpret_cast<const f32x4*>(wrow + doff[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < kSpQ4; ++u) sum[u] += gv[t] * wv[t][u];
        }
#pragma unroll
        for (int u = 0; u < kSpQ4; ++u)
            if (doff[u] >= 0) *reinterpret_cast<f32x4*>(drow + doff[u]) = sum[u];
    }
}

// One workgroup per listed (hot) token.  The token's taps (all ranks, adjacent after the sort)
// are summed into an LDS row of 64-bit fixed-point cells (2^-40 units, integer atomics: the sum does not depend on the
// order, so every rank gets the same bits), then the row goes through the same compaction + sparse product as
// g_times_w_kernel.  No global G, no zero-fill, no global atomics.
__global__ __launch_bounds__(256) void taps_rows_kernel(const ProdBwdArgs A, const int KGW, int V, const int* __restrict__ hot_count,
                                                        const int4* __restrict__ hot_list, const int* __restrict__ start1,
                                                        const int* __restrict__ end, const unsigned long long* __restrict__ pay,
                                                        float gscale, const float* __restrict__ WT, float* __restrict__ dtable,
                                                        unsigned long long* __restrict__ split_rows, int* __restrict__ split_arrived,
                                                        int n_row_wgs) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    long long* s_g = reinterpret_cast<long long*>(s_dyn);                          // [KG]
    int* s_lists = s_dyn + 2 * A.KG;
    int* s_pc = s_lists + wave * 2 * KGW;
    float* s_val = reinterpret_cast<float*>(s_pc + KGW);
    float* s_part = reinterpret_cast<float*>(s_lists + kWavesPerWG * 2 * KGW);     // [4][D]
    const int D = A.D, nq4 = D >> 2;
    const unsigned long long lt = (1ull << lane) - 1;
    const int kbeg = wave * KGW, kend = min(A.KG, kbeg + KGW);
    __shared__ int s_poison;      // a non-finite tap (NaN / inf gradient on some rank): the token's row becomes NaN, as the dense
                                  // all-reduce would propagate it, instead of vanishing in the float -> fixed-point conversion
    __shared__ int s_last;
    if ((int)blockIdx.x >= n_row_wgs) {                  // second role: a wave per vocabulary entry, the cold and the empty rows
        const int v = ((int)blockIdx.x - n_row_wgs) * kWavesPerWG + wave;     // (dispatched after the row workgroups: interleaving
        if (v < V) taps_cold_row(A, v, lane, start1, end, pay, gscale * kTapScale, WT, dtable);   // the roles delayed the long items, +100 us)
        return;
    }
    const int n_hot = *hot_count;
    // the parts of the split tokens first (the longest items), then the vocabulary, of which this role takes the tokens
    // with more than kTapsCold and at most kTapsSplit taps
    for (int h = blockIdx.x; h < n_hot + V; h += n_row_wgs) {
        int4 item;                                       // workgroup-uniform: token, part, parts, slot
        int s, e;
        if (h < n_hot) {
            item = hot_list[h];
            s = start1[item.x] - 1 + item.y * kTapsSplit;
            e = min(end[item.x], s + kTapsSplit);
        } else {
            item = int4{h - n_hot, 0, 1, -1};
            const int s1 = start1[item.x];
            if (s1 == 0) continue;
            s = s1 - 1; e = end[item.x];
            if (e - s <= kTapsCold || e - s > kTapsSplit) continue;
        }
        const int v = item.x;
        float* drow = dtable + (long)v * D;
        for (int k = tid; k < A.KG; k += 256) s_g[k] = 0;
        if (tid == 0) s_poison = 0;
        __syncthreads();
        for (int i0 = s; i0 < e; i0 += 256 * 8) {          // 8 coalesced payload reads in flight per thread
            unsigned long long pl[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256 + tid;
                pl[u] = (i < e) ? pay[i] : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (pl[u] != ~0ull) {
                    const float x = __int_as_float((int)(unsigned)(pl[u] & 0xffffffffu));
                    if (!(fabsf(x) <= 8.0e6f)) { s_poison = 1; continue; }     // NaN, inf, or beyond the fixed-point range
                    const long long q = __float2ll_rn(x * kTapScale);
                    atomicAdd(reinterpret_cast<unsigned long long*>(s_g) + (unsigned)(pl[u] >> 32), (unsigned long long)q);
                }
            }
        }
        __syncthreads();
        if (item.z > 1) {
            // a part of a split token: its cells go into the token's global row (64-bit integer adds: order-free, lanes on
            // adjacent cells); the part that arrives last takes the complete row back and carries on alone
            unsigned long long* grow = split_rows + (long)item.w * A.KG;
            for (int k = tid; k < A.KG; k += 256)
                if (s_g[k] != 0) atomicAdd(grow + k, (unsigned long long)s_g[k]);
            if (s_poison) atomicOr(split_arrived + 2 * item.w + 1, 1);
            __threadfence();
            __syncthreads();
            if (tid == 0) s_last = atomicAdd(split_arrived + 2 * item.w, 1) == item.z - 1;
            __syncthreads();
            if (!s_last) continue;                       // workgroup-uniform
            __threadfence();
            for (int k = tid; k < A.KG; k += 256)       // agent-scope loads: the other parts' adds were made by other XCDs
                s_g[k] = (long long)__hip_atomic_load(grow + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) s_poison = __hip_atomic_load(split_arrived + 2 * item.w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
        // non-zeros of this wave's quarter of the row -> (weight-row offset, value) list
        int cnt = 0;
        for (int k0 = kbeg; k0 < kend; k0 += 64) {
            const int k = k0 + lane;
            const float x = (k < kend) ? (float)s_g[k] * gscale : 0.f;
            const bool nz = x != 0.f;
            const unsigned long long b = __ballot(nz);
            if (nz) {
                const int pos = cnt + __popcll(b & lt);
                s_pc[pos] = k * D;
                s_val[pos] = x;
            }
            cnt += __popcll(b);
        }
        __builtin_amdgcn_wave_barrier();
        for (int qblk = 0; qblk < nq4; qblk += 64 * kSpQ4) {
            f32x4 sum[kSpQ4];
            int doff[kSpQ4];
#pragma unroll
            for (int u = 0; u < kSpQ4; ++u) {
                const int q4 = qblk + lane + 64 * u;
                doff[u] = (q4 < nq4) ? 4 * q4 : -1;
                sum[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int q0 = 0; q0 < cnt; q0 += 4) {
                f32x4 wv[4][kSpQ4];
                float gv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool ok = q0 + t < cnt;
                    const int it = ok ? q0 + t : q0;
                    const float* wrow = WT + s_pc[it];
                    gv[t] = ok ? s_val[it] : 0.f;
#pragma unroll
                    for (int u = 0; u < kSpQ4; ++u)
                        wv[t][u] = (doff[u] >= 0) ? *reinterpret_cast<const f32x4*>(wrow + doff[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < kSpQ4; ++u) sum[u] += gv[t] * wv[t][u];
            }
#pragma unroll
            for (int u = 0; u < kSpQ4; ++u)
                if (doff[u] >= 0) *reinterpret_cast<f32x4*>(s_part + wave * D + doff[u]) = sum[u];
        }
        __syncthreads();
        for (int q4 = tid; q4 < nq4; q4 += 256) {
            f32x4 r = *reinterpret_cast<const f32x4*>(s_part + 4 * q4);
#pragma unroll
            for (int w = 1; w < kWavesPerWG; ++w) r += *reinterpret_cast<const f32x4*>(s_part + w * D + 4 * q4);
            if (s_poison) { const float nan = __builtin_nanf(""); r = f32x4{nan, nan, nan, nan}; }
            *reinterpret_cast<f32x4*>(drow + 4 * q4) = r;
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------- conv weight gradient from G
// dW[(w,j,cl), :] = sum over distinct tokens t of G[t][(w,j,cl)] * table[t, :]  =  G^T @ E  (gate and mask are already
// inside G).  For MANY SHORT documents (NARRE: 5120 reviews of 50 tokens) this contraction over ~20 k distinct tokens
// (5 GFLOP on the MFMA pipe) replaces 2.3 M window-row reads.  Split over token ranges (grid.z) for parallelism; the
// partials are reduced in fixed order.  K-major operands: tiles are staged [k][m] in LDS and read with scalar ds_reads.
constexpr int kDwgSplit = 32;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void dw_from_g_kernel(int KG, int D, int cap, const int* __restrict__ counter,
                                                        const float* __restrict__ G, const long long* __restrict__ tok_of_row,
                                                        const float* __restrict__ table, float* __restrict__ part) {
    __shared__ float As[32][68];          // [k][col]
    __shared__ float Bs[32][68];          // [k][d]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int col0 = blockIdx.x * 64, d0 = blockIdx.y * 64, z = blockIdx.z;
    const int wm = wave >> 1, wn = wave & 1;
    const int n = min(*counter, cap);
    const int rps = (((n + kDwgSplit - 1) / kDwgSplit) + 31) & ~31;      // token rows per split, multiple of 32
    const int r_begin = z * rps, r_end = min(n, r_begin + rps);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int c = tid & 63, kq = tid >> 6;                                // element (k = kq + 4 q, c)
    const bool col_ok = col0 + c < KG, d_ok = d0 + c < D;
    // software pipeline: the table-row offsets run two tiles ahead of the MFMAs, the operand loads one tile ahead (in
    // registers), so a tile's 16 MFMAs per wave cover the next tile's two dependent round trips
    long off[8];
    float ga[8], gb[8];
    auto load_offs = [&](int r0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = r0 + kq + 4 * q;                              // wave-uniform: one broadcast load
            off[q] = (row < r_end) ? tok_of_row[row] * (long)D : -1;
        }
    };
    auto load_tile = [&](int r0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = r0 + kq + 4 * q;
            ga[q] = (row < r_end && col_ok) ? G[(long)row * KG + col0 + c] : 0.f;
            gb[q] = (off[q] >= 0 && d_ok) ? table[off[q] + d0 + c] : 0.f;
        }
    };
    if (r_begin < r_end) {
        load_offs(r_begin);
        load_tile(r_begin);
        load_offs(r_begin + 32);
    }
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
        __syncthreads();                    // the previous tile's MFMAs have read LDS
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            As[kq + 4 * q][c] = ga[q];
            Bs[kq + 4 * q][c] = gb[q];
        }
        __syncthreads();
        if (r0 + 32 < r_end) {
            load_tile(r0 + 32);             // uses the offsets fetched one iteration ago
            load_offs(r0 + 64);
        }
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + h][wm * 32 + i], Bs[kk + h][wn * 32 + i], acc, 0, 0, 0);
    }
    const int dcol = d0 + wn * 32 + i;
    if (dcol < D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = col0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (col < KG) part[((long)z * KG + col) * D + dcol] = acc[r];
        }
    }
}

// dbias partials: chunk b of documents -> part_b[b][c] = sum of g over the chunk (fixed order)
constexpr int kDbChunks = 256;
__device__ __forceinline__ void dbias_partial_block(int n_docs, int C, int act, const float* __restrict__ feat,
                                                    const float* __restrict__ d_feat, float* __restrict__ part_b, int cb, int b) {
    const int c = cb * 256 + threadIdx.x;
    if (c >= C) return;
    const int per = (n_docs + kDbChunks - 1) / kDbChunks;
    const int d_begin = b * per, d_end = min(n_docs, d_begin + per);
    float s = 0.f;
    for (int doc = d_begin; doc < d_end; ++doc) s += act_grad(act, feat[(long)doc * C + c], d_feat[(long)doc * C + c]);
    part_b[(long)b * C + c] = s;
}

// The same contraction on the bf16 pipe with exact three-plane operand splits (textcnn_b16.h: the forward GEMM's arithmetic, f32-class
// accuracy): 6 v_mfma_f32_32x32x16_bf16 per 16 k (192 cycles) where the kernel above issues 8 v_mfma_f32_32x32x2_f32 (512).  Both
// operands are K-major in memory (G [row][column], table [row][d]); a thread stages column c of a 32-row tile -- 8 consecutive rows,
// 8 coalesced loads -- splits its 8 values into planes ONCE and writes each plane's 16 bytes to LDS as [plane][m][32 k] (16-byte
// chunks XOR-swizzled by m: the fragment reads of 8 lanes cover all banks), so an MFMA operand is one ds_read_b128.  Same grid,
// same partials as dw_from_g_kernel (RBR_DW_FROM_G_F32=1 selects that one).
template <int NPL>      // 3: the exact split (f32-class); 1: operands rounded to bf16 (the bf16 class)
__global__ __launch_bounds__(256) void dw_from_g_b16_kernel(int KG, int D, int cap, const int* __restrict__ counter,
                                                            const float* __restrict__ G, const long long* __restrict__ tok_of_row,
                                                            const float* __restrict__ table, float* __restrict__ part,
                                                            int n_docs, int C, int act, const float* __restrict__ feat,
                                                            const float* __restrict__ d_feat, float* __restrict__ part_b) {
    if (blockIdx.z >= kDwgSplit) {
        // the bias gradient's partial sums ride in this launch (one launch less on the chain): the blocks of the z slices past the K
        // splits take one (channel block, document chunk) pair of dbias_partial_kernel each
        const int nb = gridDim.x * gridDim.y * (gridDim.z - kDwgSplit);
        const int b0 = ((blockIdx.z - kDwgSplit) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int ncb = (C + 255) / 256;
        for (int w = b0; w < ncb * kDbChunks; w += nb) dbias_partial_block(n_docs, C, act, feat, d_feat, part_b, w % ncb, w / ncb);
        return;
    }
    __shared__ __attribute__((aligned(16))) unsigned char As[NPL][64][64];      // [plane][column][32 k as bf16]
    __shared__ __attribute__((aligned(16))) unsigned char Bs[NPL][64][64];      // [plane][d][32 k as bf16]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int col0 = blockIdx.x * 64, d0 = blockIdx.y * 64, z = blockIdx.z;
    const int wm = wave >> 1, wn = wave & 1;
    const int n = min(*counter, cap);
    const int rps = (((n + kDwgSplit - 1) / kDwgSplit) + 63) & ~63;      // token rows per split, multiple of 64 (two tiles per round)
    const int r_begin = z * rps, r_end = min(n, r_begin + rps);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int c = tid & 63, kq = __builtin_amdgcn_readfirstlane(tid >> 6);      // this thread stages rows kq*8 .. kq*8+7 of column c
    const bool col_ok = col0 + c < KG, d_ok = d0 + c < D;
    const int cg = min(col0 + c, KG - 1), cd = min(d0 + c, D - 1);         // clamped: every load below is unconditional
    if (r_begin >= r_end) {                                               // an empty split (fewer rows than splits x 32): zeros
        const int dcol0 = d0 + wn * 32 + i;
        if (dcol0 < D) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = col0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (col < KG) part[((long)z * KG + col) * D + dcol0] = 0.f;
            }
        }
        return;
    }
    // Two register sets in a ring: set s holds a 32-row tile; it is refilled right after it has been staged, so a tile's loads have
    // the staging + MFMA phases of TWO tiles to arrive (one phase is ~0.4 us, a gathered table row under load well over 1 us; with
    // one set the kernel ran 63 us for 29 GFLOP of plane products).  The table-row offsets of a tile are fetched one refill earlier.
    long off[2][8];
    float ga[2][8], gb[2][8];
    // No load is predicated: rows past the split's end are clamped to its last row and their values multiplied by zero AFTER the
    // load.  A predicated load compiles to a branch around it, and behind a branch the compiler no longer counts the
    // loads in flight -- it drained them one by one (s_waitcnt vmcnt(0) after each of 16 offset loads per tile).
    auto load_offs = [&](int s_, int r0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = min(r0 + kq * 8 + q, r_end - 1);              // wave-uniform: a scalar load
            off[s_][q] = tok_of_row[row] * (long)D;
        }
    };
    auto load_tile = [&](int s_, int r0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = r0 + kq * 8 + q, rc = min(row, r_end - 1);
            const float va = G[(long)rc * KG + cg], vb = table[off[s_][q] + cd];
            // x * 1 is exact and x * 0 = 0 for the finite values here: a product, not a select -- a select on the (wave-uniform) row
            // test is turned back into a branch around the two loads
            ga[s_][q] = va * ((row < r_end && col_ok) ? 1.f : 0.f);
            gb[s_][q] = vb * ((row < r_end && d_ok) ? 1.f : 0.f);
        }
    };
    const int wchunk = (kq ^ ((c >> 1) & 3)) * 16;                        // where this thread's 8 k land in its column's row
    auto stage = [&](const float (&x)[8], unsigned char (&dst)[NPL][64][64]) {
        u32x4 ph, pm, pl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned a, b = 0, cc = 0;
            split_pair<NPL>(x[2 * q], x[2 * q + 1], a, b, cc);
            ph[q] = a; pm[q] = b; pl[q] = cc;
        }
        *reinterpret_cast<u32x4*>(&dst[0][c][wchunk]) = ph;
        if constexpr (NPL == 3) {
            *reinterpret_cast<u32x4*>(&dst[1][c][wchunk]) = pm;
            *reinterpret_cast<u32x4*>(&dst[2][c][wchunk]) = pl;
        }
    };
    const int am = wm * 32 + i, bn = wn * 32 + i;
    const int aswz = (am >> 1) & 3, bswz = (bn >> 1) & 3;
    auto mma_tile = [&]() {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {    // two 16-deep MFMA steps per 32-row tile; lane half h holds k = 8h .. 8h+7 of the step
            const int ca = ((2 * s2 + h) ^ aswz) * 16, cb = ((2 * s2 + h) ^ bswz) * 16;
            const bf16x8 a_hi = *reinterpret_cast<const bf16x8*>(&As[0][am][ca]);
            const bf16x8 b_hi = *reinterpret_cast<const bf16x8*>(&Bs[0][bn][cb]);
            if constexpr (NPL == 3) {
                const bf16x8 a_mid = *reinterpret_cast<const bf16x8*>(&As[1][am][ca]);
                const bf16x8 a_lo = *reinterpret_cast<const bf16x8*>(&As[2][am][ca]);
                const bf16x8 b_mid = *reinterpret_cast<const bf16x8*>(&Bs[1][bn][cb]);
                const bf16x8 b_lo = *reinterpret_cast<const bf16x8*>(&Bs[2][bn][cb]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc, 0, 0, 0);      // small terms first, as the forward
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_mid, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_hi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_mid, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc, 0, 0, 0);
        }
    };
    // (a set may be refilled past the end: clamped rows, zeroed values)
    load_offs(0, r_begin);
    load_offs(1, r_begin + 32);
    load_tile(0, r_begin);
    load_tile(1, r_begin + 32);
    for (int r0 = r_begin; r0 < r_end; r0 += 64) {
        __syncthreads();                    // the previous tile's MFMAs have read LDS
        stage(ga[0], As);
        stage(gb[0], Bs);
        __syncthreads();
        load_offs(0, r0 + 64);
        mma_tile();
        load_tile(0, r0 + 64);              // behind the MFMAs: its offsets were requested before them
        // the second tile of the round unconditionally (past the end it is all zeros): loads behind a branch are not counted, and
        // the wait for set 0 at the loop head would drain set 1 as well
        __syncthreads();
        stage(ga[1], As);
        stage(gb[1], Bs);
        __syncthreads();
        load_offs(1, r0 + 96);
        mma_tile();
        load_tile(1, r0 + 96);
    }
    const int dcol = d0 + wn * 32 + i;
    if (dcol < D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = col0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (col < KG) part[((long)z * KG + col) * D + dcol] = acc[r];
        }
    }
}

__global__ __launch_bounds__(256) void dbias_partial_kernel(int n_docs, int C, int act, const float* __restrict__ feat,
                                                            const float* __restrict__ d_feat, float* __restrict__ part_b) {
    dbias_partial_block(n_docs, C, act, feat, d_feat, part_b, blockIdx.x, blockIdx.y);
}

// dW_w[cl, d, j] = sum_z part[z][(w,j,cl)][d];  dbias[c] = sum_b part_b[b][c]   (torch layouts, fixed order)
__global__ __launch_bounds__(256) void dw_from_g_reduce_kernel(const ProdBwdArgs A, int cp_real, const float* __restrict__ part,
                                                               const float* __restrict__ part_b, const MutPtrArray dW,
                                                               const MutPtrArray dbias) {
    const long n_w = (long)cp_real * A.D;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n_w + A.C; idx += (long)gridDim.x * 256) {
        if (idx < n_w) {
            const int pc = (int)(idx / A.D), d = (int)(idx - (long)pc * A.D);
            float v[kDwgSplit];
#pragma unroll
            for (int z = 0; z < kDwgSplit; ++z) v[z] = part[((long)z * A.KG + pc) * A.D + d];      // all in flight, added in slice order
            float t = 0.f;
#pragma unroll
            for (int z = 0; z < kDwgSplit; ++z) t += v[z];
            int w = 0;
#pragma unroll
            for (int k = 1; k < RBR_MAX_WIDTHS; ++k)
                if (k < A.n_widths && pc >= A.poff[k]) w = k;
            const int rel = pc - A.poff[w];
            const int j = rel / A.ch[w], cl = rel - j * A.ch[w];
            dW.p[w][((long)cl * A.D + d) * A.kz[w] + j] = t;
        } else {
            const int c = (int)(idx - n_w);
            float t = 0.f;
            for (int b = 0; b < kDbChunks; ++b) t += part_b[(long)b * A.C + c];
            int w = 0;
#pragma unroll
            for (int k = 1; k < RBR_MAX_WIDTHS; ++k)
                if (k < A.n_widths && c >= A.ch_off[k]) w = k;
            dbias.p[w][c - A.ch_off[w]] = t;
        }
    }
}

}  // namespace rbr

using namespace rbr;

namespace {

struct ProdLayout {      // byte offsets inside the workspace
    size_t used, row_of_token, tok_of_row, row_mask, counter, sched, packed, wt, bimg, a16, table_T, total;
    int cap, Cp, tiles_p;
    rbr_textcnn_desc dp;
};

int g_conv_mode = -1;     // -1: unset -> RBR_CONV_MODE env ("dense" | "product") or auto; 0 auto, 1 dense, 2 product

bool prod_applicable(const rbr_textcnn_desc* d) {
    static const char* env = getenv("RBR_CONV_MODE");
    const char* mode = g_conv_mode == 1 ? "dense" : g_conv_mode == 2 ? "product" : g_conv_mode == 0 ? nullptr : env;
    if (mode && !strcmp(mode, "dense")) return false;
    if (!desc_valid(d)) return false;                      // nothing below may index d->kz / d->ch of an unchecked descriptor
    const long n_pos = (long)d->n_docs * d->L;
    long Cp = 0;
    for (int w = 0; w < d->n_widths; ++w) Cp += (long)d->kz[w] * d->ch[w];
    if (Cp > 30000) return false;                          // slot ids are 16-bit
    if ((Cp + kTile - 1) / kTile > (long)kMaxGroups * kProdGroupTiles) return false;
    if (mode && !strcmp(mode, "product")) return true;
    // auto: worth it when the vocabulary bounds the distinct tokens well below the position count
    return (long)d->V * 5 <= n_pos * 2 && n_pos >= 4096;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool prod_layout(const rbr_textcnn_desc* d, ProdLayout& Lo) {
    const long n_pos = (long)d->n_docs * d->L;
    Lo.cap = (int)std::min<long>(d->V, n_pos);
    long Cp = 0;
    for (int w = 0; w < d->n_widths; ++w) Cp += (long)d->kz[w] * d->ch[w];
    // zero-weight padding channels make all work-item groups identical: 8 tiles each (750 -> 768 = 3 x 8 tiles at cfg2:
    // 167 row blocks x 3 groups = 501 items for 512 resident workgroups), fewer when the whole bank is smaller
    const int group_tiles = (int)std::min<long>(kProdGroupTiles, (Cp + kTile - 1) / kTile);
    // the bf16-plane GEMM works on items of 128 product channels
    const int kGroupSlots = prod_b16_applicable(d) ? 128 : group_tiles * kTile;
    Cp = ((Cp + kGroupSlots - 1) / kGroupSlots) * kGroupSlots;
    Lo.Cp = (int)Cp;
    Lo.tiles_p = (int)(Cp / kTile);
    memset(&Lo.dp, 0, sizeof(Lo.dp));
    Lo.dp.n_docs = 1; Lo.dp.L = Lo.cap; Lo.dp.D = d->D; Lo.dp.V = d->V;
    Lo.dp.n_widths = 1; Lo.dp.kz[0] = 1; Lo.dp.ch[0] = Lo.Cp;
    Lo.dp.pad_mode = RBR_PAD_SAME; Lo.dp.act = RBR_ACT_RELU; Lo.dp.padding_idx = -1;
    ConvPlan plans[kMaxGroups];
    if (!build_plans(&Lo.dp, plans, kProdGroupTiles)) return false;
    const ConvPlan& p = plans[0];
    size_t o = 0;
    // [used | row_mask] are contiguous and re-zeroed every call together with the counters of `sched`
    Lo.used = o;         o += align256((size_t)d->V);                    // one byte per token id
    Lo.row_mask = o;     o += align256((size_t)Lo.cap);
    Lo.row_of_token = o; o += align256((size_t)d->V * sizeof(int));
    Lo.tok_of_row = o;   o += align256((size_t)Lo.cap * sizeof(long long));
    // flags | list (unused: the token list needs no scan) | counters.  counters[0] doubles as the distinct-token count:
    // the MFMA kernel in store mode derives its active tiles from it
    Lo.sched = o;        o += align256((2 * (size_t)p.total_wt + kSchedCounters) * sizeof(int));
    Lo.counter = Lo.sched + 2 * (size_t)p.total_wt * sizeof(int);
    Lo.packed = o;       o += align256((size_t)p.nchunks * p.tiles_total * kTile * p.DC * sizeof(float));
    long cp_real = 0;
    for (int w = 0; w < d->n_widths; ++w) cp_real += (long)d->kz[w] * d->ch[w];
    Lo.wt = o;           o += align256((size_t)cp_real * d->D * sizeof(float));
    Lo.bimg = o;         o += prod_b16_applicable(d) ? align256(prod_b16_image_bytes(d, (int)cp_real)) : 0;
    Lo.a16 = o;          o += align256(prod_b16_rows_bytes(d, Lo.cap));        // bf16 storage: compact bf16 copy of the listed rows
    // (bf16 storage: T's elements are 2 bytes; the space is reserved for f32 either way -- the layout must not depend on more
    // run-time state than it does today)
    Lo.table_T = o;      o += align256((((size_t)Lo.cap + 1) * p.nslots_total + 4) * sizeof(float));   // + one float4 of slack: quads read whole
    Lo.total = o;
    return true;
}

}  // namespace

extern "C" void rbr_set_conv_mode(int32_t mode) { g_conv_mode = (mode >= 0 && mode <= 2) ? mode : -1; }

namespace rbr {
// 0 auto, 1 dense forced, 2 product forced (rbr_set_conv_mode, else RBR_CONV_MODE): shared with the token-product gate
int forced_conv_mode() {
    if (g_conv_mode >= 0) return g_conv_mode;
    static const char* env = getenv("RBR_CONV_MODE");
    if (env && !strcmp(env, "dense")) return 1;
    if (env && !strcmp(env, "product")) return 2;
    return 0;
}
}  // namespace rbr

extern "C" size_t rbr_textcnn_fwd_ws_bytes(const rbr_textcnn_desc* d) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return 0;
    if (!prod_applicable(d)) return 0;
    ProdLayout Lo;
    if (!prod_layout(d, Lo)) return 0;
    return Lo.total;
}

namespace {

struct ProdBwdLayout {
    size_t G, bimg_t, total;         // bimg_t: weight planes of the row GEMM (prod_b16_rows_applicable shapes), else unused
    int KG, KGW, cp_real;
    bool rows_gemm;
};

bool prod_bwd_layout(const rbr_textcnn_desc* d, const ProdLayout& Lo, ProdBwdLayout& B) {
    if (d->D % 4 != 0) return false;                        // float4 rows; other widths keep the window scatter
    long cp = 0;
    for (int w = 0; w < d->n_widths; ++w) cp += (long)d->kz[w] * d->ch[w];
    B.cp_real = (int)cp;
    B.KG = (int)((cp + 3) / 4 * 4);
    B.KGW = ((B.KG / 4 + kWavesPerWG - 1) / kWavesPerWG) * 4;      // columns per wave of the sparse product
    if ((size_t)B.KGW * 8 * kWavesPerWG + (size_t)kWavesPerWG * d->D * 4 > 64 * 1024) return false;   // lists + partial rows in LDS
    size_t o = 0;
    B.G = o;  o += align256((size_t)Lo.cap * B.KG * sizeof(float));
    B.rows_gemm = prod_b16_rows_applicable(d);
    B.bimg_t = o;  o += B.rows_gemm ? align256(prod_b16_rows_image_bytes(B.KG, d->D)) : 0;
    B.total = o;
    return true;
}

}  // namespace

extern "C" size_t rbr_textcnn_bwd_prod_ws_bytes(const rbr_textcnn_desc* d) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return 0;
    if (!prod_applicable(d)) return 0;
    static const char* env = getenv("RBR_DTABLE_MODE");
    if (env && !strcmp(env, "scatter")) return 0;
    ProdLayout Lo;
    ProdBwdLayout B;
    if (!prod_layout(d, Lo) || !prod_bwd_layout(d, Lo, B)) return 0;
    return B.total;
}

// phases of dtable_through_list: build G | multiply it out | ... adding to dtable instead of overwriting it | ... into compact
// rows + sq_part (kGtwRows) | G's rows are already zero (the forward's gather_pool launch cleared them: no zero_g_rows launch)
enum { kGBuild = 1, kGProduct = 2, kGAccumulate = 4, kGRows = 8, kGZeroed = 16 };
static int dtable_through_list(const rbr_textcnn_desc* d, const ConvPlan* plans, const int64_t* ids, const uint8_t* mask, const float* gate,
                               const float* feat, const int32_t* argmax, const float* d_feat, void* fwd_ws, void* bwd_ws,
                               float* dtable, float* dgate, hipStream_t st, int phases = kGBuild | kGProduct, float* sq_part = nullptr);

extern "C" int rbr_textcnn_bwd_dtable_prod(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                                           const float* feat, const int32_t* argmax, const float* d_feat, void* fwd_ws,
                                           void* bwd_ws, float* dtable, float* dgate, void* stream) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return RBR_ERR_BAD_ARG;
    if (dgate != nullptr && gate == nullptr) dgate = nullptr;
    if (dtable == nullptr && dgate == nullptr) return 0;
    if (!ids || !feat || !argmax || !d_feat || !fwd_ws || !bwd_ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if (!prod_applicable(d)) { set_error("token-product path does not apply to this shape"); return RBR_ERR_UNSUPPORTED; }
    return dtable_through_list(d, plans, ids, mask, gate, feat, argmax, d_feat, fwd_ws, bwd_ws, dtable, dgate, (hipStream_t)stream);
}

// rbr_textcnn_bwd_dtable_prod with the table rows ADDED to `dtable` (a gradient buffer shared by several producers that run
// one after the other on the same stream: functional.table_fanout) instead of the whole [V, D] gradient being overwritten.
extern "C" int rbr_textcnn_bwd_dtable_prod_acc(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                                               const float* feat, const int32_t* argmax, const float* d_feat, void* fwd_ws,
                                               void* bwd_ws, float* dtable, float* dgate, void* stream) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return RBR_ERR_BAD_ARG;
    if (dgate != nullptr && gate == nullptr) dgate = nullptr;
    if (!ids || !feat || !argmax || !d_feat || !fwd_ws || !bwd_ws || !dtable) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if (!prod_applicable(d)) { set_error("token-product path does not apply to this shape"); return RBR_ERR_UNSUPPORTED; }
    return dtable_through_list(d, plans, ids, mask, gate, feat, argmax, d_feat, fwd_ws, bwd_ws, dtable, dgate, (hipStream_t)stream,
                               kGBuild | kGProduct | kGAccumulate);
}

// The two halves of rbr_textcnn_bwd_dtable_prod as separate calls, for callers that put work between them or beside the
// second one (functional: NARRE's dW = G^T @ table rows runs on a second stream while the product runs on the first).
extern "C" int rbr_textcnn_bwd_g_build(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                                       const float* feat, const int32_t* argmax, const float* d_feat, void* fwd_ws, void* bwd_ws,
                                       float* dgate, void* stream) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return RBR_ERR_BAD_ARG;
    if (dgate != nullptr && gate == nullptr) dgate = nullptr;
    if (!ids || !feat || !argmax || !d_feat || !fwd_ws || !bwd_ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if (!prod_applicable(d)) { set_error("token-product path does not apply to this shape"); return RBR_ERR_UNSUPPORTED; }
    return dtable_through_list(d, plans, ids, mask, gate, feat, argmax, d_feat, fwd_ws, bwd_ws, nullptr, dgate, (hipStream_t)stream,
                               kGBuild);
}

extern "C" int rbr_textcnn_bwd_g_product(const rbr_textcnn_desc* d, void* fwd_ws, void* bwd_ws, float* dtable, void* stream) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return RBR_ERR_BAD_ARG;
    if (!fwd_ws || !bwd_ws || !dtable) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if (!prod_applicable(d)) { set_error("token-product path does not apply to this shape"); return RBR_ERR_UNSUPPORTED; }
    return dtable_through_list(d, plans, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, fwd_ws, bwd_ws, dtable, nullptr,
                               (hipStream_t)stream, kGProduct);
}

// Every form of the token-product table gradient behind one entry (the calls above are its fixed-flag forms):
//   RBR_G_BUILD | RBR_G_PRODUCT   the phases to run;
//   RBR_G_ACCUMULATE              the rows are added to the dense `dtable`;
//   RBR_G_ROWS                    `dtable` is the COMPACT gradient [list rows, D] (row r = token tok_of_row[r]; absent tokens have
//                                 no row) and sq_part[rbr_textcnn_row_grad_partials(d)] receives per-workgroup sums of squares;
//   RBR_G_ZEROED                  G's rows are zero already (a caller that cleared bwd_ws itself).
extern "C" int rbr_textcnn_bwd_dtable_prod_ex(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                                              const float* feat, const int32_t* argmax, const float* d_feat, void* fwd_ws, void* bwd_ws,
                                              float* dtable, float* dgate, float* sq_part, int32_t flags, void* stream) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return RBR_ERR_BAD_ARG;
    if (dgate != nullptr && gate == nullptr) dgate = nullptr;
    if (!fwd_ws || !bwd_ws) { set_error("null workspace"); return RBR_ERR_BAD_ARG; }
    if ((flags & RBR_G_BUILD) && (!ids || !feat || !argmax || !d_feat)) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if ((flags & RBR_G_PRODUCT) && !dtable) { set_error("null dtable"); return RBR_ERR_BAD_ARG; }
    if ((flags & RBR_G_ROWS) && ((flags & RBR_G_ACCUMULATE) || !sq_part)) { set_error("RBR_G_ROWS needs sq_part and excludes RBR_G_ACCUMULATE"); return RBR_ERR_BAD_ARG; }
    if (!prod_applicable(d)) { set_error("token-product path does not apply to this shape"); return RBR_ERR_UNSUPPORTED; }
    static_assert(RBR_G_BUILD == kGBuild && RBR_G_PRODUCT == kGProduct && RBR_G_ACCUMULATE == kGAccumulate && RBR_G_ROWS == kGRows &&
                  RBR_G_ZEROED == kGZeroed, "public flags = internal phases");
    return dtable_through_list(d, plans, ids, mask, gate, feat, argmax, d_feat, fwd_ws, bwd_ws, dtable, dgate, (hipStream_t)stream,
                               flags & (kGBuild | kGProduct | kGAccumulate | kGRows | kGZeroed), sq_part);
}

// = the grid of the product kernel: one partial per workgroup, every one of them written
extern "C" size_t rbr_textcnn_row_grad_partials(const rbr_textcnn_desc* d) {
    ProdLayout Lo;
    if (!prod_applicable(d) || !prod_layout(d, Lo)) return 0;
    if (prod_b16_rows_applicable(d)) return (size_t)prod_b16_rows_partials(Lo.cap, d->D);
    return (size_t)std::min(Lo.cap, kGtwMaxBlocks);
}

// Where the forward left the token list inside `fwd_ws` (the layout is private): row_of_token [V] (dense row or -1), the
// device count of listed rows, tok_of_row [cap]; cap = rows the compact gradient must have room for.
extern "C" int rbr_textcnn_token_list(const rbr_textcnn_desc* d, void* fwd_ws, const int32_t** row_of_token, const int32_t** n_rows,
                                      const int64_t** tok_of_row, int32_t* cap) {
    ProdLayout Lo;
    if (!fwd_ws || !prod_applicable(d) || !prod_layout(d, Lo)) { set_error("no token list for this shape"); return RBR_ERR_UNSUPPORTED; }
    char* base = static_cast<char*>(fwd_ws);
    if (row_of_token) *row_of_token = reinterpret_cast<const int32_t*>(base + Lo.row_of_token);
    if (n_rows) *n_rows = reinterpret_cast<const int32_t*>(base + Lo.counter);
    if (tok_of_row) *tok_of_row = reinterpret_cast<const int64_t*>(base + Lo.tok_of_row);
    if (cap) *cap = Lo.cap;
    return 0;
}

// G over the token list in `fwd_ws` (ProdLayout), then dtable = G @ Wprod^T; shared by the token-product backward (the
// forward's list) and by the dense formulation's backward (a list built for the purpose, rbr_textcnn_bwd_dtable_list)
static int dtable_through_list(const rbr_textcnn_desc* d, const ConvPlan* plans, const int64_t* ids, const uint8_t* mask, const float* gate,
                               const float* feat, const int32_t* argmax, const float* d_feat, void* fwd_ws, void* bwd_ws,
                               float* dtable, float* dgate, hipStream_t st, int phases, float* sq_part) {
    PairSolo solo;        // zero G -> build G -> G @ Wprod^T: a producer / consumer chain over G (rbr_launch.h)
    ProdLayout Lo;
    ProdBwdLayout B;
    if (!prod_layout(d, Lo) || !prod_bwd_layout(d, Lo, B)) return RBR_ERR_BAD_ARG;
    char* fbase = static_cast<char*>(fwd_ws);
    const int* row_of_token = reinterpret_cast<const int*>(fbase + Lo.row_of_token);
    const long long* tok_of_row = reinterpret_cast<const long long*>(fbase + Lo.tok_of_row);
    const int* counter = reinterpret_cast<const int*>(fbase + Lo.counter);
    const float* WT = reinterpret_cast<const float*>(fbase + Lo.wt);     // Wprod^T, written by the forward's stage 1
    const float* T = reinterpret_cast<const float*>(fbase + Lo.table_T);  // product table, still intact
    float* G = reinterpret_cast<float*>(static_cast<char*>(bwd_ws) + B.G);

    ProdBwdArgs A{};
    A.n_docs = d->n_docs; A.L = d->L; A.C = plans[0].C; A.KF = plans[0].KF; A.KG = B.KG; A.D = d->D; A.cap = Lo.cap;
    A.padding_idx = d->padding_idx; A.pad_mode = d->pad_mode; A.act = d->act; A.n_widths = d->n_widths;
    A.gate_split = RBR_CONV_GATE_SPLIT_OF(d->flags);
    {
        ConvPlan pp[kMaxGroups];
        if (!build_plans(&Lo.dp, pp, kProdGroupTiles)) return RBR_ERR_BAD_ARG;
        A.t_pitch = pp[0].nslots_total;
    }
    int cp_real = 0;
    for (int w = 0; w < d->n_widths; ++w) {
        A.kz[w] = d->kz[w]; A.ch[w] = d->ch[w]; A.ch_off[w] = plans[0].ch_off[w];
        A.poff[w] = cp_real; cp_real += d->kz[w] * d->ch[w];
    }
    // kGBuild alone (rbr_textcnn_bwd_g_build) always builds G: its caller multiplies it out later (rbr_textcnn_bwd_g_product)
    const bool want_g = dtable != nullptr || !(phases & kGProduct);
    if (phases & kGBuild) {
        if ((want_g && !(phases & kGZeroed)) || dgate != nullptr || B.rows_gemm) {      // dgate is zeroed here: the caller hands it over uninitialised
            // row-GEMM shapes: the same launch writes the bf16 planes of Wprod^T the product phase multiplies G by
            const long n_img = B.rows_gemm ? prod_b16_rows_image_items(B.KG, d->D) : 0;
            const int nb_zero = 2048, nb_img = (int)std::min<long>((n_img + 255) / 256, 1024);
            if (int e_ = rbr::launch<zero_g_rows_kernel, 256>(dim3(nb_zero + nb_img), dim3(256), 0, st, "textcnn zero_g_rows launch", counter, Lo.cap, B.KG / 4, (want_g && !(phases & kGZeroed)) ? reinterpret_cast<f32x4*>(G) : nullptr, dgate, dgate != nullptr ? (long)d->n_docs * d->L * (RBR_CONV_GATE_SPLIT_OF(d->flags) ? 2 : 1) : 0L, nb_zero, WT, cp_real, d->D, n_img, reinterpret_cast<unsigned char*>(static_cast<char*>(bwd_ws) + B.bimg_t))) return e_;
        }
        const long n_items = (long)d->n_docs * A.C * A.KF;
        if (int e_ = rbr::launch<build_g_kernel, 256>(dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, st, "textcnn build_g launch", A, reinterpret_cast<const long long*>(ids), mask, gate, row_of_token, T, feat, argmax, d_feat, want_g ? G : nullptr, dgate, prod_t_bf16(d) ? 1 : 0)) return e_;
    }
    if (dtable == nullptr || !(phases & kGProduct)) return 0;
    const size_t lds = (size_t)B.KGW * 8 * kWavesPerWG + (size_t)kWavesPerWG * d->D * sizeof(float);
    const dim3 grid((unsigned)std::min(Lo.cap, kGtwMaxBlocks));
    if (phases & kGRows) {
        if (sq_part == nullptr) { set_error("compact row gradient needs sq_part"); return RBR_ERR_BAD_ARG; }
        if (B.rows_gemm)          // many short documents: G is dense enough for the bf16-plane GEMM (textcnn_prod_b16.hip)
            return prod_b16_rows_gemm(B.KG, d->D, Lo.cap, counter, G, static_cast<char*>(bwd_ws) + B.bimg_t, dtable, sq_part,
                                      prod_precision_of(d) == RBR_PROD_BF16, st);
        if (int e_ = rbr::launch<g_times_w_kernel<kGtwRows>, 256>(grid, dim3(256), lds, st, "textcnn g_times_w launch", A, B.KGW, counter, G, WT, tok_of_row, row_of_token, d->V, dtable, sq_part)) return e_;
    } else if (phases & kGAccumulate) {
        if (int e_ = rbr::launch<g_times_w_kernel<kGtwAccumulate>, 256>(grid, dim3(256), lds, st, "textcnn g_times_w launch", A, B.KGW, counter, G, WT, tok_of_row, row_of_token, d->V, dtable, (float*)nullptr)) return e_;
    } else {
        if (int e_ = rbr::launch<g_times_w_kernel<kGtwDense>, 256>(grid, dim3(256), lds, st, "textcnn g_times_w launch", A, B.KGW, counter, G, WT, tok_of_row, row_of_token, d->V, dtable, (float*)nullptr)) return e_;
    }
    return 0;
}

namespace {

struct ProdState {       // everything the three forward stages share, derived from (d, ws) alone
    ProdLayout Lo;
    ConvPlan pp[kMaxGroups];
    ProdArgs A;
    unsigned char* used;
    int *row_of_token, *counter, *sched_p;
    long long* tok_of_row;
    unsigned char* row_mask;
    float *packed_p, *WT, *T;
    char* base;
};

int prod_state(const rbr_textcnn_desc* d, void* ws, ProdState& S) {
    if (ws == nullptr) { set_error("null workspace"); return RBR_ERR_BAD_ARG; }
    if (!prod_applicable(d)) { set_error("token-product path does not apply to this shape"); return RBR_ERR_UNSUPPORTED; }
    if (!prod_layout(d, S.Lo)) return RBR_ERR_BAD_ARG;
    const ProdLayout& Lo = S.Lo;
    S.base = static_cast<char*>(ws);
    S.used = reinterpret_cast<unsigned char*>(S.base + Lo.used);
    S.row_of_token = reinterpret_cast<int*>(S.base + Lo.row_of_token);
    S.tok_of_row = reinterpret_cast<long long*>(S.base + Lo.tok_of_row);
    S.row_mask = reinterpret_cast<unsigned char*>(S.base + Lo.row_mask);
    S.counter = reinterpret_cast<int*>(S.base + Lo.counter);
    S.sched_p = reinterpret_cast<int*>(S.base + Lo.sched);
    S.packed_p = reinterpret_cast<float*>(S.base + Lo.packed);
    S.WT = reinterpret_cast<float*>(S.base + Lo.wt);
    S.T = reinterpret_cast<float*>(S.base + Lo.table_T);
    if (!build_plans(&S.Lo.dp, S.pp, kProdGroupTiles)) return RBR_ERR_BAD_ARG;
    S.pp[0].store_rows = 1;      // group 0's plan describes every group; the kernel folds them into one launch
    ProdArgs A{};
    A.n_docs = d->n_docs; A.L = d->L; A.V = d->V; A.cap = Lo.cap; A.zrow = Lo.cap; A.pitch = S.pp[0].nslots_total;
    int o = 0;    // product channels follow the bank order of the ORIGINAL problem (bank w, tap j, channel cl)
    for (int w = 0; w < d->n_widths; ++w) { A.poff[w] = o; o += d->kz[w] * d->ch[w]; }
    A.cp_real = o;
    for (int w = 0; w < d->n_widths; ++w) {      // as build_plans orders the channel slots
        int r = 0;
        for (int v = 0; v < d->n_widths; ++v)
            if (d->kz[v] < d->kz[w] || (d->kz[v] == d->kz[w] && v < w)) r += d->ch[v];
        A.rank_off[w] = r;
    }
    S.A = A;
    return 0;
}

}  // namespace

// Stage 1: work list of the real documents (tail of `pidx`), distinct-token list of the batch, product weight
// images.  Three launches: zero the state | mark + scan | compact + pack.
static int prod_prepare(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* const* W, int32_t* pidx,
                        void* ws, void* stream, const IdSets* id_sets, int64_t* err) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return RBR_ERR_BAD_ARG;
    if (!ids || !W || !pidx) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    ProdState S;
    if (int e = prod_state(d, ws, S)) return e;
    hipStream_t st = (hipStream_t)stream;
    int* sched = pidx + (size_t)plans[0].total_wt * plans[0].nslots_total;
    // token marks, row mask and both work-list counter blocks are re-initialised every call: graph-replay safe
    ZeroRegions zr{{reinterpret_cast<int*>(S.base + S.Lo.used), S.sched_p + 2 * (size_t)S.pp[0].total_wt,
                    sched + 2 * (size_t)plans[0].total_wt},
                   {(long)((S.Lo.row_of_token - S.Lo.used) / sizeof(int)), kSchedCounters, kSchedCounters}};
    if (id_sets != nullptr) {
        const long long total = id_sets->first[id_sets->count];
        const int nb_ids = (int)std::min<long long>((total + 255) / 256, 2048);
        const long nz = zr.n[0] + zr.n[1] + zr.n[2];
        const int nb_zero = (int)std::min<long>((nz + 255) / 256, 256);
        hipLaunchKernelGGL(prep_kernel, dim3(nb_ids + nb_zero), dim3(256), 0, st, *id_sets, reinterpret_cast<long long*>(err), zr, nb_ids);
        RBR_CHECK_LAUNCH("textcnn prep launch");
    } else if (int e = zero_regions(zr, st)) {
        return e;
    }
    const long n_tok = (long)d->n_docs * d->L;
    const int nb_scan = (plans[0].total_wt + 255) / 256;
    const int nb_mark = (int)std::min<long>((n_tok + 255) / 256, 2048);
    if (int e_ = rbr::launch<mark_scan_kernel, 256>(dim3(nb_scan + nb_mark), dim3(256), 0, st, "textcnn mark_scan launch", plans[0], nb_scan, n_tok, reinterpret_cast<const long long*>(ids), mask, sched, S.used)) return e_;
    PackJob J{};
    J.P = S.pp[0]; J.n_widths = d->n_widths; J.D = d->D; J.cp_real = S.A.cp_real;
    for (int w = 0; w < d->n_widths; ++w) { J.kz[w] = d->kz[w]; J.ch[w] = d->ch[w]; J.poff[w] = S.A.poff[w]; }
    PtrArray wp{};
    for (int w = 0; w < d->n_widths; ++w) wp.p[w] = W[w];
    const int nb_compact = (d->V + 255) / 256;
    const bool b16 = prod_b16_applicable(d);
    B16Pack JB{};
    if (b16) JB = prod_b16_pack_job(d);
    const long n_pack = (b16 ? (long)JB.ngroups * JB.nchunks * 4 * 64 : (long)S.pp[0].nchunks * S.pp[0].tiles_total * kTile * S.pp[0].DC) +
                        (long)S.A.cp_real * d->D;
    const int nb_pack = (int)std::min<long>((n_pack + 255) / 256, 2048);
    // (bf16-plane GEMM: the weight planes in MFMA-fragment order instead of the f32 tile image)
    if (int e_ = rbr::launch<compact_pack_kernel, 256>(dim3(nb_compact + nb_pack), dim3(256), 0, st, "textcnn compact_pack launch", J, nb_compact, d->V, S.Lo.cap, S.used, S.row_of_token, S.tok_of_row, S.row_mask, S.counter, prod_t_bf16(d) ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(S.T) + (size_t)S.Lo.cap * S.A.pitch)
                                      : S.T + (size_t)S.Lo.cap * S.A.pitch, prod_t_bf16(d) ? S.A.pitch / 2 : S.A.pitch, wp, b16 ? nullptr : S.packed_p, S.WT, JB, b16 ? reinterpret_cast<unsigned char*>(S.base + S.Lo.bimg) : nullptr)) return e_;
    return 0;
}

extern "C" int rbr_textcnn_prod_prepare(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask,
                                        const float* const* W, int32_t* pidx, void* ws, void* stream) {
    return prod_prepare(d, ids, mask, W, pidx, ws, stream, nullptr, nullptr);
}

// rbr_textcnn_prod_prepare with the model's id range check (rbr_sanitize_ids) folded into its first launch: `sets` are the raw
// id tensors of the forward and where their clean copies go; the conv's token ids are the clean copies of the first set(s):
// `ids` must point at sets[0].out and hold d->n_docs * d->L ids (one set, or two adjacent ones = the stacked towers).
extern "C" int rbr_textcnn_prod_prepare_ids(const rbr_textcnn_desc* d, int32_t n_sets, const rbr_id_set* sets, int64_t* err,
                                            const int64_t* ids, const uint8_t* mask, const float* const* W, int32_t* pidx, void* ws,
                                            void* stream) {
    if (!err) { set_error("null err"); return RBR_ERR_BAD_ARG; }
    IdSets S;
    if (int e = fill_id_sets(n_sets, sets, S)) return e;
    const long long n_tok = (long long)d->n_docs * d->L;
    const bool one = sets[0].out == ids && sets[0].n == n_tok;
    const bool two = n_sets >= 2 && sets[0].out == ids && sets[1].out == sets[0].out + sets[0].n && sets[0].n + sets[1].n == n_tok;
    if (!one && !two) { set_error("prod_prepare_ids: `ids` is not the clean copy of the first id set(s)"); return RBR_ERR_BAD_ARG; }
    return prod_prepare(d, ids, mask, W, pidx, ws, stream, &S, err);
}

// ---- dense formulation's table gradient through a token list built for the purpose (no product table T: un-gated convs)
namespace {
constexpr size_t kListMaxG = (size_t)256 << 20;      // the list's worst-case G (min(V, positions) rows) must stay below this
bool list_bwd_layout(const rbr_textcnn_desc* d, ProdLayout& Lo, ProdBwdLayout& B, size_t& list_bytes) {
    if (!prod_layout(d, Lo) || !prod_bwd_layout(d, Lo, B)) return false;
    if (B.total > kListMaxG) return false;
    ConvPlan pp[kMaxGroups];
    if (!build_plans(&Lo.dp, pp, kProdGroupTiles)) return false;
    list_bytes = Lo.table_T + align256((size_t)pp[0].nslots_total * sizeof(float));      // everything but T: one (zero) row of it
    return true;
}
}  // namespace

extern "C" size_t rbr_textcnn_bwd_dtable_list_ws_bytes(const rbr_textcnn_desc* d) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return 0;
    static const char* env = getenv("RBR_DTABLE_MODE");
    if (env && !strcmp(env, "scatter")) return 0;
    ProdLayout Lo; ProdBwdLayout B; size_t list_bytes;
    if (!list_bwd_layout(d, Lo, B, list_bytes)) return 0;
    return align256(list_bytes) + B.total;
}

extern "C" int rbr_textcnn_bwd_dtable_list(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* const* W,
                                           const float* feat, const int32_t* argmax, const float* d_feat, void* ws, float* dtable,
                                           void* stream) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return RBR_ERR_BAD_ARG;
    if (!ids || !W || !feat || !argmax || !d_feat || !ws || !dtable) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    ProdLayout Lo; ProdBwdLayout B; size_t list_bytes;
    if (!list_bwd_layout(d, Lo, B, list_bytes)) { set_error("token-list table gradient does not apply to this shape"); return RBR_ERR_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    char* base = static_cast<char*>(ws);
    ConvPlan pp[kMaxGroups];
    if (!build_plans(&Lo.dp, pp, kProdGroupTiles)) return RBR_ERR_BAD_ARG;
    int* sched_p = reinterpret_cast<int*>(base + Lo.sched);
    // list state: token marks, row mask, counters
    ZeroRegions zr{{reinterpret_cast<int*>(base + Lo.used), sched_p + 2 * (size_t)pp[0].total_wt, nullptr},
                   {(long)((Lo.row_of_token - Lo.used) / sizeof(int)), kSchedCounters, 0}};
    if (int e = zero_regions(zr, st)) return e;
    const long n_tok = (long)d->n_docs * d->L;
    if (int e_ = rbr::launch<mark_scan_kernel, 256>(dim3((unsigned)std::min<long>((n_tok + 255) / 256, 2048)), dim3(256), 0, st, "textcnn mark launch", plans[0], 0, n_tok, reinterpret_cast<const long long*>(ids), mask, (int*)nullptr, reinterpret_cast<unsigned char*>(base + Lo.used))) return e_;
    PackJob J{};
    J.P = pp[0]; J.n_widths = d->n_widths; J.D = d->D;
    int o = 0;
    for (int w = 0; w < d->n_widths; ++w) { J.kz[w] = d->kz[w]; J.ch[w] = d->ch[w]; J.poff[w] = o; o += d->kz[w] * d->ch[w]; }
    J.cp_real = o;
    PtrArray wp{};
    for (int w = 0; w < d->n_widths; ++w) wp.p[w] = W[w];
    const int nb_compact = (d->V + 255) / 256;
    const int nb_pack = (int)std::min<long>(((long)J.cp_real * d->D + 255) / 256, 2048);
    if (int e_ = rbr::launch<compact_pack_kernel, 256>(dim3(nb_compact + nb_pack), dim3(256), 0, st, "textcnn compact launch", J, nb_compact, d->V, Lo.cap, reinterpret_cast<unsigned char*>(base + Lo.used), reinterpret_cast<int*>(base + Lo.row_of_token), reinterpret_cast<long long*>(base + Lo.tok_of_row), reinterpret_cast<unsigned char*>(base + Lo.row_mask), reinterpret_cast<int*>(base + Lo.counter), reinterpret_cast<float*>(base + Lo.table_T), pp[0].nslots_total, wp, (float*)nullptr, reinterpret_cast<float*>(base + Lo.wt), B16Pack{}, (unsigned char*)nullptr)) return e_;
    return dtable_through_list(d, plans, ids, mask, nullptr, feat, argmax, d_feat, base, base + align256(list_bytes), dtable, nullptr, st);
}

// Stage 2 (one kernel): T = table[tok_of_row] @ Wprod on the f32 MFMA pipe.
extern "C" int rbr_textcnn_prod_table(const rbr_textcnn_desc* d, const float* table, void* ws, void* stream) {
    if (!table) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    PairSolo solo;        // the GEMM writes the product table the gather (rbr_textcnn_prod_pool) reads next: tower by tower
    ProdState S;
    if (int e = prod_state(d, ws, S)) return e;
    if (prod_b16_applicable(d))
        return prod_b16_gemm(d, S.A.cp_real, S.Lo.cap, S.A.pitch, S.counter, S.tok_of_row, table, S.base + S.Lo.bimg, S.T,
                             S.base + S.Lo.a16, (hipStream_t)stream);
    return run_conv_groups(S.pp, 1, S.tok_of_row, S.row_mask, nullptr, table, S.packed_p, S.T, nullptr, S.sched_p, (hipStream_t)stream);
}

// Stage 3 (one kernel): per active wave-tile of the real documents (work list in the tail of `pidx`, built by stage 1),
// add the kz rows of T per position, max / first argmax.
extern "C" int rbr_textcnn_prod_pool(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                                     float* pval, int32_t* pidx, void* ws, void* stream) {
    ConvPlan plans[kMaxGroups];
    const int ngroups = build_plans(d, plans);
    if (!ngroups) return RBR_ERR_BAD_ARG;
    if (!ids || !pval || !pidx) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    PairSolo solo;        // see rbr_textcnn_prod_table
    ProdState S;
    if (int e = prod_state(d, ws, S)) return e;
    hipStream_t st = (hipStream_t)stream;
    const int* sched = pidx + (size_t)plans[0].total_wt * plans[0].nslots_total;      // filled by rbr_textcnn_prod_prepare
    const int max_items = (plans[0].total_wt + kWavesPerWG - 1) / kWavesPerWG;
    if (prod_t_bf16(d))
        { if (int e_ = rbr::launch<gather_pool_kernel<true>, 256>(dim3(max_items), dim3(256), 0, st, "textcnn gather_pool launch", plans[0], S.A, reinterpret_cast<const long long*>(ids), mask, gate, S.row_of_token, static_cast<const void*>(S.T), sched, pval, pidx)) return e_; }
    else
        if (int e_ = rbr::launch<gather_pool_kernel<false>, 256>(dim3(max_items), dim3(256), 0, st, "textcnn gather_pool launch", plans[0], S.A, reinterpret_cast<const long long*>(ids), mask, gate, S.row_of_token, static_cast<const void*>(S.T), sched, pval, pidx)) return e_;
    return 0;
}

// ---- conv weight / bias gradient from G (see dw_from_g_kernel): after rbr_textcnn_bwd_dtable_prod built G in `bwd_ws`
static bool dw_from_g_applicable(const rbr_textcnn_desc* d) {
    // the shapes the document-centric dW kernel serves (textcnn_bwd.hip): many short documents
    long cp = 0;
    for (int w = 0; w < d->n_widths; ++w) cp += (long)d->kz[w] * d->ch[w];
    return prod_applicable(d) && d->D % 4 == 0 && d->L <= 128 && d->n_docs >= 512 && cp <= 2048;
}

extern "C" size_t rbr_textcnn_bwd_dw_from_g_ws_floats(const rbr_textcnn_desc* d) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans) || !dw_from_g_applicable(d)) return 0;
    ProdLayout Lo; ProdBwdLayout B;
    if (!prod_layout(d, Lo) || !prod_bwd_layout(d, Lo, B)) return 0;
    return (size_t)kDwgSplit * B.KG * d->D + (size_t)kDbChunks * plans[0].C;
}

extern "C" int rbr_textcnn_bwd_dw_from_g(const rbr_textcnn_desc* d, const float* table, const float* feat, const float* d_feat,
                                         void* fwd_ws, void* bwd_ws, float* const* dW, float* const* dbias, float* ws, void* stream) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return RBR_ERR_BAD_ARG;
    if (!dw_from_g_applicable(d)) { set_error("dW-from-G does not apply to this shape"); return RBR_ERR_UNSUPPORTED; }
    if (!table || !feat || !d_feat || !fwd_ws || !bwd_ws || !dW || !dbias || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    ProdLayout Lo; ProdBwdLayout B;
    if (!prod_layout(d, Lo) || !prod_bwd_layout(d, Lo, B)) return RBR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* fbase = static_cast<char*>(fwd_ws);
    const long long* tok_of_row = reinterpret_cast<const long long*>(fbase + Lo.tok_of_row);
    const int* counter = reinterpret_cast<const int*>(fbase + Lo.counter);
    const float* G = reinterpret_cast<const float*>(static_cast<char*>(bwd_ws) + B.G);
    ProdBwdArgs A{};
    A.n_docs = d->n_docs; A.L = d->L; A.C = plans[0].C; A.KF = plans[0].KF; A.KG = B.KG; A.D = d->D; A.cap = Lo.cap;
    A.act = d->act; A.n_widths = d->n_widths;
    int cp_real = 0;
    for (int w = 0; w < d->n_widths; ++w) {
        A.kz[w] = d->kz[w]; A.ch[w] = d->ch[w]; A.ch_off[w] = plans[0].ch_off[w];
        A.poff[w] = cp_real; cp_real += d->kz[w] * d->ch[w];
    }
    float* part = ws;
    float* part_b = ws + (size_t)kDwgSplit * B.KG * d->D;
    static const bool f32_form = getenv("RBR_DW_FROM_G_F32") != nullptr && atoi(getenv("RBR_DW_FROM_G_F32")) != 0;
    // z slices beyond the K splits: one workgroup per (channel block, document chunk) of the bias gradient's partial sums
    const int tile_wgs = ((B.KG + 63) / 64) * ((d->D + 63) / 64);
    const int db_slices = (((A.C + 255) / 256) * kDbChunks + tile_wgs - 1) / tile_wgs;
    if (f32_form)
        hipLaunchKernelGGL(dw_from_g_kernel, dim3((B.KG + 63) / 64, (d->D + 63) / 64, kDwgSplit), dim3(256), 0, st, B.KG, d->D, Lo.cap,
                           counter, G, tok_of_row, table, part);
    else      // (the bias gradient's partials in the same launch: one z slice more)
        if (prod_precision_of(d) == RBR_PROD_BF16)      // the bf16 class: one plane product of operands rounded to bf16
            hipLaunchKernelGGL(dw_from_g_b16_kernel<1>, dim3((B.KG + 63) / 64, (d->D + 63) / 64, kDwgSplit + db_slices), dim3(256), 0, st, B.KG, d->D, Lo.cap,
                               counter, G, tok_of_row, table, part, d->n_docs, A.C, d->act, feat, d_feat, part_b);
        else
            hipLaunchKernelGGL(dw_from_g_b16_kernel<3>, dim3((B.KG + 63) / 64, (d->D + 63) / 64, kDwgSplit + db_slices), dim3(256), 0, st, B.KG, d->D, Lo.cap,
                               counter, G, tok_of_row, table, part, d->n_docs, A.C, d->act, feat, d_feat, part_b);
    RBR_CHECK_LAUNCH("textcnn dw_from_g launch");
    if (f32_form) {
        hipLaunchKernelGGL(dbias_partial_kernel, dim3((A.C + 255) / 256, kDbChunks), dim3(256), 0, st, d->n_docs, A.C, d->act, feat, d_feat,
                           part_b);
        RBR_CHECK_LAUNCH("textcnn dbias partial launch");
    }
    MutPtrArray dWp{}, dbp{};
    for (int w = 0; w < d->n_widths; ++w) { dWp.p[w] = dW[w]; dbp.p[w] = dbias[w]; }
    const long total = (long)cp_real * d->D + A.C;
    hipLaunchKernelGGL(dw_from_g_reduce_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, A, cp_real,
                       part, part_b, dWp, dbp);
    RBR_CHECK_LAUNCH("textcnn dw_from_g reduce launch");
    return 0;
}

// ---- data-parallel tap exchange (see taps_kernel): taps of one rank, and the averaged table gradient from all ranks' taps
namespace {
bool taps_args(const rbr_textcnn_desc* d, ProdBwdArgs& A, int& cp_real, int& KG) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return false;
    memset(&A, 0, sizeof(A));
    A.n_docs = d->n_docs; A.L = d->L; A.C = plans[0].C; A.KF = plans[0].KF; A.D = d->D;
    A.padding_idx = d->padding_idx; A.pad_mode = d->pad_mode; A.act = d->act; A.n_widths = d->n_widths;
    cp_real = 0;
    for (int w = 0; w < d->n_widths; ++w) {
        A.kz[w] = d->kz[w]; A.ch[w] = d->ch[w]; A.ch_off[w] = plans[0].ch_off[w];
        A.poff[w] = cp_real; cp_real += d->kz[w] * d->ch[w];
    }
    KG = (cp_real + 3) / 4 * 4;
    A.KG = KG;
    return true;
}
struct TapsLayout { size_t keys_in, keys, pay_in, pay, start1, end, hot_count, hot_list, split_arrived, split_rows, WT, blk_cnt, temp, total; size_t temp_bytes; int KGW, max_split; };
bool taps_layout(const rbr_textcnn_desc* d, int n_sets, long n_items, int cp_real, int KG, TapsLayout& T) {
    if (d->D % 4 != 0 || n_sets <= 0) return false;
    T.KGW = ((KG / 4 + kWavesPerWG - 1) / kWavesPerWG) * 4;
    if ((size_t)KG * 8 + (size_t)T.KGW * 8 * kWavesPerWG + (size_t)kWavesPerWG * d->D * 4 > 64 * 1024) return false;
    const size_t total = (size_t)n_items * n_sets;
    if (total >= (size_t)1 << 31) return false;
    T.temp_bytes = 0;
    int* ni = nullptr; unsigned long long* nl = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, T.temp_bytes, ni, ni, nl, nl, total, 0, 32, (hipStream_t)0);
    size_t o = 0;
    T.keys_in = o; o += align256(total * sizeof(int));
    T.keys = o;    o += align256(total * sizeof(int));
    T.pay_in = o;  o += align256(total * sizeof(unsigned long long));
    T.pay = o;     o += align256(total * sizeof(unsigned long long));
    T.start1 = o;  o += align256((size_t)d->V * sizeof(int));
    T.end = o;     o += align256((size_t)d->V * sizeof(int));
    T.hot_count = o; o += 256;
    T.max_split = (int)(total / kTapsSplit) + 1;                 // tokens with more than kTapsSplit taps
    T.hot_list = o; o += align256(((size_t)d->V + 2 * (size_t)T.max_split) * sizeof(int4));     // parts: at most cnt / kTapsSplit + 1 per token
    T.split_arrived = o; o += align256((size_t)T.max_split * 2 * sizeof(int));                  // contiguous with split_rows: zeroed together
    T.split_rows = o; o += align256((size_t)T.max_split * KG * sizeof(unsigned long long));
    T.WT = o;      o += align256((size_t)cp_real * d->D * sizeof(float));
    T.blk_cnt = o; o += align256((size_t)(kOwnChunks + 64) * sizeof(int));      // owner partition: chunk counts, then the overflow flag
    T.temp = o;    o += align256(T.temp_bytes) + 256;
    T.total = o;
    return true;
}
}  // namespace

extern "C" size_t rbr_textcnn_taps_count(const rbr_textcnn_desc* d) {
    ProdBwdArgs A; int cp, KG;
    if (!taps_args(d, A, cp, KG)) return 0;
    return (size_t)A.n_docs * A.C * A.KF;
}

extern "C" int rbr_textcnn_bwd_taps(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* feat,
                                    const int32_t* argmax, const float* d_feat, int32_t* tok, float* val, void* stream) {
    ProdBwdArgs A; int cp, KG;
    if (!taps_args(d, A, cp, KG)) return RBR_ERR_BAD_ARG;
    if (!ids || !feat || !argmax || !d_feat || !tok || !val) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    const long n = (long)A.n_docs * A.C * A.KF;
    hipLaunchKernelGGL(taps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A,
                       reinterpret_cast<const long long*>(ids), mask, feat, argmax, d_feat, tok, val);
    RBR_CHECK_LAUNCH("textcnn taps launch");
    return 0;
}

extern "C" size_t rbr_textcnn_dtable_from_taps_ws_bytes(const rbr_textcnn_desc* d, int32_t n_sets) {
    ProdBwdArgs A; int cp, KG; TapsLayout T;
    if (!taps_args(d, A, cp, KG) || !taps_layout(d, n_sets, (long)A.n_docs * A.C * A.KF, cp, KG, T)) return 0;
    return T.total;
}

namespace {
// own_rank < 0: every token's row into out [V, D].  own_rank >= 0: the rows of the tokens t % n_sets == own_rank into
// out [ceil(V / n_sets), D] (row t / n_sets), `overflow` set when the rank's taps exceed the bound the sort is sized for.
int taps_rebuild(const rbr_textcnn_desc* d, int n_sets, int own_rank, const int32_t* tok, const float* val, const float* const* W,
                 void* ws, float* out, int32_t* overflow, hipStream_t st) {
    ProdBwdArgs A; int cp_real, KG; TapsLayout T;
    if (!taps_args(d, A, cp_real, KG)) return RBR_ERR_BAD_ARG;
    const long n_items = (long)A.n_docs * A.C * A.KF;
    if (!taps_layout(d, n_sets, n_items, cp_real, KG, T)) { set_error("tap exchange does not support this shape"); return RBR_ERR_UNSUPPORTED; }
    if (!tok || !val || !W || !ws || !out) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    const bool own = own_rank >= 0;
    if (own && (own_rank >= n_sets || !overflow)) { set_error("taps owner rebuild: rank %d of %d, overflow flag %p", own_rank, n_sets, (void*)overflow); return RBR_ERR_BAD_ARG; }
    char* base = static_cast<char*>(ws);
    int* keys_in = reinterpret_cast<int*>(base + T.keys_in);
    int* keys = reinterpret_cast<int*>(base + T.keys);
    unsigned long long* pay_in = reinterpret_cast<unsigned long long*>(base + T.pay_in);
    unsigned long long* pay = reinterpret_cast<unsigned long long*>(base + T.pay);
    int* start1 = reinterpret_cast<int*>(base + T.start1);
    int* end = reinterpret_cast<int*>(base + T.end);
    float* WT = reinterpret_cast<float*>(base + T.WT);
    const long total = n_items * n_sets;
    const int Vloc = own ? (d->V + n_sets - 1) / n_sets : d->V;                  // rows this call builds
    const long n_sort = own ? rbr::taps_owner_cap(total, n_sets) : total;        // entries that go through the sort
    A.cap = Vloc;
    if (own) {
        int* blk_cnt = reinterpret_cast<int*>(base + T.blk_cnt);
        hipLaunchKernelGGL(taps_owner_count_kernel, dim3(kOwnChunks), dim3(256), 0, st, total, n_sets, own_rank, tok, blk_cnt);
        RBR_CHECK_LAUNCH("textcnn taps owner count launch");
        hipLaunchKernelGGL(taps_owner_keys_kernel, dim3(kOwnChunks), dim3(256), 0, st, A, n_items, n_sets, own_rank, Vloc, n_sort, tok, val,
                           blk_cnt, keys_in, pay_in, overflow);
        RBR_CHECK_LAUNCH("textcnn taps owner keys launch");
    } else {
        hipLaunchKernelGGL(taps_keys_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, st, A, n_items, n_sets,
                           d->V, tok, val, keys_in, pay_in);
        RBR_CHECK_LAUNCH("textcnn taps keys launch");
    }
    int bits = 1;
    while ((1L << bits) <= Vloc) ++bits;                 // keys are in [0, Vloc]
    size_t temp_bytes = T.temp_bytes;
    if (int e = check_hip(rocprim::radix_sort_pairs(base + T.temp, temp_bytes, keys_in, keys, pay_in, pay, (size_t)n_sort, 0, bits, st),
                          "textcnn taps radix sort"))
        return e;
    int* hot_count = reinterpret_cast<int*>(base + T.hot_count);
    int4* hot_list = reinterpret_cast<int4*>(base + T.hot_list);
    int* split_arrived = reinterpret_cast<int*>(base + T.split_arrived);
    unsigned long long* split_rows = reinterpret_cast<unsigned long long*>(base + T.split_rows);
    {
        ZeroRegions z{{start1, hot_count, split_arrived},
                      {(long)(align256((size_t)d->V * sizeof(int)) / sizeof(int)), 64, (long)((T.WT - T.split_arrived) / sizeof(int))}};
        if (int e = zero_regions(z, st)) return e;
    }
    hipLaunchKernelGGL(taps_bounds_kernel, dim3((unsigned)std::min<long>((n_sort + 255) / 256, 8192)), dim3(256), 0, st, n_sort, Vloc, keys,
                       start1, end);
    RBR_CHECK_LAUNCH("textcnn taps bounds launch");
    PackJob J{};
    J.n_widths = d->n_widths; J.D = d->D; J.cp_real = cp_real;
    for (int w = 0; w < d->n_widths; ++w) { J.kz[w] = d->kz[w]; J.ch[w] = d->ch[w]; J.poff[w] = A.poff[w]; }
    PtrArray wp{};
    for (int w = 0; w < d->n_widths; ++w) wp.p[w] = W[w];
    hipLaunchKernelGGL(taps_wt_kernel, dim3((unsigned)std::min<long>(((long)cp_real * d->D + 255) / 256, 1024)), dim3(256), 0, st, J, wp, WT,
                       Vloc, start1, end, hot_count, hot_list);
    RBR_CHECK_LAUNCH("textcnn taps weights launch");
    const size_t lds = (size_t)KG * 8 + (size_t)T.KGW * 8 * kWavesPerWG + (size_t)kWavesPerWG * d->D * sizeof(float);
    const int n_row_wgs = std::min(Vloc, 8192), n_cold_wgs = (Vloc + kWavesPerWG - 1) / kWavesPerWG;
    hipLaunchKernelGGL(taps_rows_kernel, dim3((unsigned)(n_row_wgs + n_cold_wgs)), dim3(256), lds, st, A, T.KGW, Vloc, hot_count, hot_list,
                       start1, end, pay, 1.f / (kTapScale * (float)n_sets), WT, out, split_rows, split_arrived, n_row_wgs);
    RBR_CHECK_LAUNCH("textcnn taps rows launch");
    return 0;
}
}  // namespace

extern "C" int rbr_textcnn_dtable_from_taps(const rbr_textcnn_desc* d, int32_t n_sets, const int32_t* tok, const float* val,
                                            const float* const* W, void* ws, float* dtable, void* stream) {
    return taps_rebuild(d, n_sets, -1, tok, val, W, ws, dtable, nullptr, (hipStream_t)stream);
}

extern "C" int32_t rbr_textcnn_taps_owner_rows(const rbr_textcnn_desc* d, int32_t n_sets) {
    return (d && d->V > 0 && n_sets > 0) ? (int32_t)(((long)d->V + n_sets - 1) / n_sets) : 0;
}

extern "C" int rbr_textcnn_dtable_from_taps_owner(const rbr_textcnn_desc* d, int32_t n_sets, int32_t rank, const int32_t* tok,
                                                  const float* val, const float* const* W, void* ws, float* slab,
                                                  int32_t* overflow, void* stream) {
    if (rank < 0) { set_error("taps owner rebuild: rank %d", rank); return RBR_ERR_BAD_ARG; }
    return taps_rebuild(d, n_sets, rank, tok, val, W, ws, slab, overflow, (hipStream_t)stream);
}

namespace rbr {

// returns 1 when the product path ran, 0 when the caller must run the dense conv, < 0 / hip error on failure
int run_token_product(const rbr_textcnn_desc* d, const long long* ids, const unsigned char* mask, const float* gate,
                      const float* table, const float* const* W, float* pval, int* pidx, void* ws, hipStream_t st) {
    if (ws == nullptr || !prod_applicable(d)) return 0;
    const int64_t* ids64 = reinterpret_cast<const int64_t*>(ids);
    if (int e = rbr_textcnn_prod_prepare(d, ids64, mask, W, pidx, ws, st)) return e;
    if (int e = rbr_textcnn_prod_table(d, table, ws, st)) return e;
    if (int e = rbr_textcnn_prod_pool(d, ids64, mask, gate, pval, pidx, ws, st)) return e;
    return 1;
}

}  // namespace rbr
