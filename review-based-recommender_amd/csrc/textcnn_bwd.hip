// textcnn_bwd.hip -- backward of the fused TextCNN encoder, exploiting max-pool sparsity.
//
// The reference back-propagates through MaxPool1d / ReLU / conv1d / masked_fill / embedding as
// dense ops (loss.backward(), trainer/train_deepconn_pp.py:165; convolution_backward is 42 % of
// its step).  Mathematically d(out[doc,c]) reaches exactly ONE conv window -- the one at
// argmax[doc,c] -- so with g = d_feat * act'(feat):
//     dbias[c]        = sum_doc g
//     dW[c, :, j]     = sum_doc g * x[doc, l* + j - pad, :]           (x = mask * gate * table[ids])
//     dtable[id, :]  += g * gate * W[c, :, j]      for id = ids[doc, l* + j - pad]  (not padding_idx)
//     dgate[doc, p]  += g * <W[c, :, j], table[id, :]>
// which is HBM/atomic-bound row traffic, not a GEMM.  Results equal the dense backward up to
// fp32 summation order.
//
// Kernels
//   dw_partial : grid (C, NCH).  One workgroup owns channel c and a contiguous range of documents;
//                thread d keeps dW[c, d, 0..kz) in registers (no atomics), window rows are resolved
//                once per 32 documents into LDS so the table reads are independent, coalesced loads.
//   dw_reduce  : sums the NCH partial slabs in fixed order (bitwise reproducible) into torch layout.
//   dx_window  : grid (n_docs x windows of 64 positions).  The window's (channel, tap) items are bucketed by
//                token in LDS; one wave per token sums them in registers and adds one row to dtable with
//                256-byte contiguous f32 atomics.
#include "rbr_common.h"

#include <algorithm>
#include <cstdlib>

namespace rbr {

constexpr int kDocsPerBatch = 32;
constexpr int kMaxChunksBwd = 32;
constexpr int kChanBatch = 128;   // channels resolved per LDS batch in dx_scatter

struct BwdArgs {
    int n_docs, L, D, C, KF, DC, nchunks, tiles_total;
    int pad_mode, act, padding_idx;
    int n_widths;
    int kz[RBR_MAX_WIDTHS], ch[RBR_MAX_WIDTHS], ch_off[RBR_MAX_WIDTHS];
    int rank_off[RBR_MAX_WIDTHS];   // first slot (kz-sorted channel order) of bank w in the packed image
    int NCH, DPC;                   // document chunks of the dW kernels and documents per chunk
    int gate_split;                 // RBR_CONV_GATE_SPLIT: banks >= this read plane 1 of `gate` (0: one plane)
    int doc_centric;                // 1: dw_doc_kernel (short documents), 0: dw_partial_kernel
    int dev_flags;                  // tuning aid (RBR_DEV_DX_ABLATE): 1 = no atomics, 2 = no accumulation phase
};
RBR_SHARED_KERNEL_ARG(BwdArgs);

__device__ __forceinline__ int bank_of(const BwdArgs& A, int c) {
    int w = 0;
#pragma unroll
    for (int k = 1; k < RBR_MAX_WIDTHS; ++k)
        if (k < A.n_widths && c >= A.ch_off[k]) w = k;
    return w;
}

// ------------------------------------------------------------------------------------ dW
__global__ __launch_bounds__(256) void dw_partial_kernel(const BwdArgs A, const long long* __restrict__ ids,
                                                         const unsigned char* __restrict__ mask,
                                                         const float* __restrict__ gate, const float* __restrict__ table,
                                                         const float* __restrict__ feat, const int* __restrict__ argmax,
                                                         const float* __restrict__ d_feat, float* __restrict__ ws_w,
                                                         float* __restrict__ ws_b) {
    __shared__ long s_row[kDocsPerBatch * kMaxKF];
    __shared__ float s_sc[kDocsPerBatch * kMaxKF];
    __shared__ float s_g[kDocsPerBatch];
    const int c = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const int w = bank_of(A, c);
    const long gplane = (A.gate_split > 0 && w >= A.gate_split) ? (long)A.n_docs * A.L : 0;      // the bank's gate plane
    const int kz = A.kz[w];
    const int padl = (A.pad_mode == RBR_PAD_SAME) ? (kz - 1) / 2 : 0;
    const int doc_begin = chunk * A.DPC;
    const int doc_end = min(A.n_docs, doc_begin + A.DPC);
    const int L = A.L, D = A.D, C = A.C;

    float bsum = 0.f;
    for (int dbase = 0; dbase < D; dbase += 256) {
        const int d = dbase + tid;
        float acc[kMaxKF];
#pragma unroll
        for (int j = 0; j < kMaxKF; ++j) acc[j] = 0.f;

        for (int b0 = doc_begin; b0 < doc_end; b0 += kDocsPerBatch) {
            const int nb = min(kDocsPerBatch, doc_end - b0);
            __syncthreads();   // previous batch fully consumed
            for (int e = tid; e < nb * kz; e += 256) {
                const int dl = e / kz, j = e - dl * kz;
                const long o = (long)(b0 + dl) * C + c;
                const float g = act_grad(A.act, feat[o], d_feat[o]);
                const int p = argmax[o] + j - padl;
                long row = 0;      // (row 0, scale 0) = no contribution
                float sc = 0.f;
                if (g != 0.f && p >= 0 && p < L) {
                    const long tok = (long)(b0 + dl) * L + p;
                    if (mask == nullptr || mask[tok]) {
                        row = ids[tok] * (long)D;
                        sc = (gate != nullptr) ? g * gate[gplane + tok] : g;
                    }
                }
                s_row[e] = row;
                s_sc[e] = sc;
                if (j == 0) s_g[dl] = g;
            }
            __syncthreads();
            if (dbase == 0 && tid == 0)
                for (int dl = 0; dl < nb; ++dl) bsum += s_g[dl];   // fixed order
            if (d < D) {
                // branch-free: rows without a contribution were resolved to (row 0, scale 0), so all loads of
                // 4 documents x kz taps are independent and in flight together
                for (int dl = 0; dl < nb; dl += 4) {
                    float x[4][kMaxKF], sc[4][kMaxKF];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int dq = min(dl + q, nb - 1);
                        const bool ok = dl + q < nb;
#pragma unroll
                        for (int j = 0; j < kMaxKF; ++j) {
                            if (j < kz) {
                                x[q][j] = table[s_row[dq * kz + j] + d];
                                sc[q][j] = ok ? s_sc[dq * kz + j] : 0.f;
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int j = 0; j < kMaxKF; ++j)
                            if (j < kz) acc[j] = fmaf(sc[q][j], x[q][j], acc[j]);
                }
            }
        }
        if (d < D) {
#pragma unroll
            for (int j = 0; j < kMaxKF; ++j)
                if (j < kz) ws_w[(((long)chunk * C + c) * A.KF + j) * D + d] = acc[j];
        }
    }
    if (tid == 0) ws_b[(long)chunk * C + c] = bsum;
}

// (Measured and dropped in round 3: dW from the NON-ZERO LISTS of the per-token tap matrix G -- G compacted into (column, value)
// runs per 32 list rows (25 us), then one wave per 16-float slice of the embedding dim keeping dW[:, slice] in LDS and adding
// value x row slice per entry: every distinct table row is read once per slice (100 MB of L2 traffic instead of 504), but the
// work becomes 7.3 M wave-level LDS adds of 64 bytes: 957 us as first built (one wave per SIMD, every batch of entry loads a
// full memory round trip), and LDS-pipe-bound at 50-100 us on paper once pipelined -- no better than this kernel's 70.)
// float4 variant of dw_partial (D % 4 == 0, D <= 1024): thread = (slot, float4 column).  The 256 / (D/4) slots of a
// workgroup split into `jslots` taps x `dpar` documents in flight, so a 300-wide row is read by 75 lanes with one
// dwordx4 each and three window rows are read at once (the scalar kernel needs 300 + 44 lanes in two passes per row).
// Partial sums of the `dpar` document classes meet in LDS; same slabs, same dw_reduce.
typedef float f32x4 __attribute__((ext_vector_type(4)));
// kDwMaxM = taps per thread (compile-time bound: the register arrays below are sized by it)
template <int kDwMaxM>
__device__ __forceinline__ void dw_partial4_kernel(const BwdArgs& A, const long long* __restrict__ ids,
                                                          const unsigned char* __restrict__ mask,
                                                          const float* __restrict__ gate, const float* __restrict__ table,
                                                          const float* __restrict__ feat, const int* __restrict__ argmax,
                                                          const float* __restrict__ d_feat, float* __restrict__ ws_w,
                                                          float* __restrict__ ws_b) {
    __shared__ long s_row[kDocsPerBatch * kMaxKF];
    __shared__ float s_sc[kDocsPerBatch * kMaxKF];
    __shared__ float s_g[kDocsPerBatch];
    __shared__ __attribute__((aligned(16))) float s_red[512];      // [kz][D], used when dpar > 1, i.e. kz * D/4 <= 128
    const int c = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const int w = bank_of(A, c);
    const long gplane = (A.gate_split > 0 && w >= A.gate_split) ? (long)A.n_docs * A.L : 0;      // the bank's gate plane
    const int kz = A.kz[w];
    const int padl = (A.pad_mode == RBR_PAD_SAME) ? (kz - 1) / 2 : 0;
    const int doc_begin = chunk * A.DPC;
    const int doc_end = min(A.n_docs, doc_begin + A.DPC);
    const int L = A.L, D = A.D, C = A.C;
    const int nq4 = D >> 2;
    const int nslots = 256 / nq4;
    const int jslots = min(nslots, kz);
    const int dpar = nslots / jslots;
    const int q4 = tid % nq4, slot = tid / nq4;
    const bool live = slot < jslots * dpar;
    const int jsub = slot % jslots, dsub = slot / jslots;
    const int M = (kz - jsub + jslots - 1) / jslots;          // taps of this thread: jsub, jsub + jslots, ...

    f32x4 acc[kDwMaxM];
#pragma unroll
    for (int m = 0; m < kDwMaxM; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int b0 = doc_begin; b0 < doc_end; b0 += kDocsPerBatch) {
        const int nb = min(kDocsPerBatch, doc_end - b0);
        __syncthreads();   // previous batch fully consumed
        for (int e = tid; e < nb * kz; e += 256) {
            const int dl = e / kz, j = e - dl * kz;
            const long o = (long)(b0 + dl) * C + c;
            const float g = act_grad(A.act, feat[o], d_feat[o]);
            const int p = argmax[o] + j - padl;
            long row = 0;      // (row 0, scale 0) = no contribution
            float sc = 0.f;
            if (g != 0.f && p >= 0 && p < L) {
                const long tok = (long)(b0 + dl) * L + p;
                if (mask == nullptr || mask[tok]) {
                    row = ids[tok] * (long)D;
                    sc = (gate != nullptr) ? g * gate[gplane + tok] : g;
                }
            }
            s_row[e] = row;
            s_sc[e] = sc;
            if (j == 0) s_g[dl] = g;
        }
        __syncthreads();
        if (tid == 0)
            for (int dl = 0; dl < nb; ++dl) bsum += s_g[dl];   // fixed order
        if (live) {
            // branch-free: rows without a contribution were resolved to (row 0, scale 0); 4 documents x M taps of
            // independent float4 loads are in flight together
            for (int dl = dsub; dl < nb; dl += 4 * dpar) {
                f32x4 x[4][kDwMaxM];
                float sc[4][kDwMaxM];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int du = dl + u * dpar;
                    const bool ok = du < nb;
                    const int dq = ok ? du : dl;
#pragma unroll
                    for (int m = 0; m < kDwMaxM; ++m) {
                        if (m < M) {
                            const int e = dq * kz + jsub + m * jslots;
                            x[u][m] = *reinterpret_cast<const f32x4*>(table + s_row[e] + 4 * q4);
                            sc[u][m] = ok ? s_sc[e] : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int m = 0; m < kDwMaxM; ++m)
                        if (m < M) acc[m] += sc[u][m] * x[u][m];
            }
        }
    }
    // document classes meet in LDS (fixed order), class 0 writes the slab
    for (int r = 1; r < dpar; ++r) {
        __syncthreads();
        if (live && dsub == r) {
#pragma unroll
            for (int m = 0; m < kDwMaxM; ++m)
                if (m < M) *reinterpret_cast<f32x4*>(s_red + ((jsub + m * jslots) * nq4 + q4) * 4) = acc[m];
        }
        __syncthreads();
        if (live && dsub == 0) {
#pragma unroll
            for (int m = 0; m < kDwMaxM; ++m)
                if (m < M) acc[m] += *reinterpret_cast<const f32x4*>(s_red + ((jsub + m * jslots) * nq4 + q4) * 4);
        }
    }
    if (live && dsub == 0) {
#pragma unroll
        for (int m = 0; m < kDwMaxM; ++m)
            if (m < M)
                *reinterpret_cast<f32x4*>(ws_w + (((long)chunk * C + c) * A.KF + jsub + m * jslots) * D + 4 * q4) = acc[m];
    }
    if (tid == 0) ws_b[(long)chunk * C + c] = bsum;
}

// Document-centric dW for SHORT documents (NARRE reviews: L = 50, 150 channels x 3 taps = 450 windows rows drawn
// from <= 50 distinct token rows per review).  dw_partial reads every (doc, channel, tap) row from L2/Infinity Cache
// (2.76 GB at cfg3); here a workgroup owns a 32-float slice of the embedding dim, stages each document's rows for
// that slice ONCE in LDS and accumulates all (channel, tap) items in registers (thread (grp, lane) owns items
// grp, grp+8, ...).  Writes the same per-chunk slabs as dw_partial, so dw_reduce finishes it.
constexpr int kDocSlice = 32;       // floats of the embedding dim per workgroup
constexpr int kDocItems = 512;      // max C * KF handled (64 register accumulators x 8 item groups)
constexpr int kDocMaxL = 128;

__global__ __launch_bounds__(256) void dw_doc_kernel(const BwdArgs A, const long long* __restrict__ ids,
                                                     const unsigned char* __restrict__ mask, const float* __restrict__ gate,
                                                     const float* __restrict__ table, const float* __restrict__ feat,
                                                     const int* __restrict__ argmax, const float* __restrict__ d_feat,
                                                     float* __restrict__ ws_w, float* __restrict__ ws_b) {
    __shared__ float xs[kDocMaxL * kDocSlice];
    __shared__ float s_g[kDocItems];
    __shared__ short s_p[kDocItems];
    __shared__ long s_row[kDocMaxL];
    __shared__ float s_gate[kDocMaxL];
    const int slice = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const int grp = tid >> 5, lane = tid & 31;
    const int L = A.L, D = A.D, C = A.C, KF = A.KF;
    const int nitems = C * KF;
    const int d = slice * kDocSlice + lane;
    const int doc_begin = chunk * A.DPC, doc_end = min(A.n_docs, doc_begin + A.DPC);
    float acc[kDocItems / 8];
#pragma unroll
    for (int k = 0; k < kDocItems / 8; ++k) acc[k] = 0.f;
    float gb[2] = {0.f, 0.f};   // dbias partials of the items this thread resolves (tid, tid + 256)
    for (int e = tid; e < kDocItems; e += 256) { s_g[e] = 0.f; s_p[e] = 0; }   // slots >= nitems stay (0, row 0)

    for (int doc = doc_begin; doc < doc_end; ++doc) {
        __syncthreads();
        // phase A: one round of independent loads -- token rows of the document and the (channel, tap) items
        if (tid < L) {
            const long tok = (long)doc * L + tid;
            const long id = ids[tok];
            const bool ok = (mask == nullptr) || mask[tok];
            s_row[tid] = ok ? id * (long)D : -1;
            s_gate[tid] = (gate != nullptr) ? gate[tok] : 1.f;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 256 * q;
            if (e < nitems) {
                const int c = e / KF, j = e - c * KF;
                const int w = bank_of(A, c);
                const int kz = A.kz[w];
                float g = 0.f;
                int p = 0;
                if (j < kz) {
                    const long o = (long)doc * C + c;
                    g = act_grad(A.act, feat[o], d_feat[o]);
                    if (j == 0) gb[q] += g;
                    p = argmax[o] + j - ((A.pad_mode == RBR_PAD_SAME) ? (kz - 1) / 2 : 0);
                    if (p < 0 || p >= L) { g = 0.f; p = 0; }
                }
                s_g[e] = g;
                s_p[e] = (short)p;
            }
        }
        __syncthreads();
        // phase B: stage the 32-float slice of every row (all loads of a thread are independent)
#pragma unroll
        for (int q = 0; q < kDocMaxL * kDocSlice / 256; ++q) {
            const int r = q * (256 / kDocSlice) + grp;      // idx = q*256 + tid = r*32 + lane
            if (r < L) {
                const long ro = s_row[r];
                xs[r * kDocSlice + lane] = (ro >= 0 && d < D) ? table[ro + d] * s_gate[r] : 0.f;
            }
        }
        __syncthreads();
        // phase C: register accumulation, no predicates (padding slots carry g = 0)
#pragma unroll
        for (int k = 0; k < kDocItems / 8; ++k) {
            const int e = grp + 8 * k;
            acc[k] = fmaf(s_g[e], xs[s_p[e] * kDocSlice + lane], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < kDocItems / 8; ++k) {
        const int e = grp + 8 * k;
        if (e < nitems && d < D) {
            const int c = e / KF, j = e - c * KF;
            ws_w[(((long)chunk * C + c) * KF + j) * D + d] = acc[k];
        }
    }
    if (slice == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 256 * q;
            if (e < nitems && e % KF == 0) ws_b[(long)chunk * C + e / KF] = gb[q];
        }
    }
}

// One workgroup per (channel, 128 embedding columns): sums the NCH partial slabs in fixed order (reads coalesced along d,
// the NCH loads of an element independent of each other), transposes through LDS and writes the [d][tap] block of the
// torch layout as one contiguous run.
constexpr int kRedCols = 128;
__device__ __forceinline__ void dw_reduce_kernel(const BwdArgs& A, const float* __restrict__ ws_w,
                                                        const float* __restrict__ ws_b, const MutPtrArray& dW,
                                                        const MutPtrArray& dbias) {
    __shared__ float s_out[kMaxKF * kRedCols];
    const int c = blockIdx.x, d0 = blockIdx.y * kRedCols;
    const int nd = min(kRedCols, A.D - d0);
    const int w = bank_of(A, c);
    const int kz = A.kz[w];
    const long slab = (long)A.C * A.KF * A.D;
    const float* src = ws_w + ((long)c * A.KF) * A.D + d0;
    for (int e = threadIdx.x; e < kz * kRedCols; e += 256) {
        const int j = e / kRedCols, dd = e % kRedCols;
        if (dd >= nd) continue;
        const float* q = src + (long)j * A.D + dd;
        float sum = 0.f;
        int k = 0;
        for (; k + 8 <= A.NCH; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = q[(long)(k + u) * slab];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; k < A.NCH; ++k) sum += q[(long)k * slab];
        s_out[j * kRedCols + dd] = sum;
    }
    __syncthreads();
    float* dst = dW.p[w] + ((long)(c - A.ch_off[w]) * A.D + d0) * kz;
    for (int o = threadIdx.x; o < nd * kz; o += 256) dst[o] = s_out[(o % kz) * kRedCols + o / kz];
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        float b = 0.f;
        for (int k = 0; k < A.NCH; ++k) b += ws_b[(long)k * A.C + c];
        dbias.p[w][c - A.ch_off[w]] = b;
    }
}

// ------------------------------------------------------------------------------------ dtable / dgate
// One workgroup per (document, window of 64 token positions).  Every (channel, tap) whose argmax window
// row falls into the window is an ITEM: it adds  g * gate * W[c, :, j]  to the table row of that token.
// Items are bucketed in LDS by TOKEN (key = first row of the window carrying the same token id), then
// one wave per token sums its items in registers (lanes over the embedding dim, no LDS traffic) and
// issues ONE row of 256-byte contiguous f32 atomics.  Folding same-token rows before the atomics is
// what keeps Zipf-hot rows ("the", ",") from serialising the memory-side atomic units.
constexpr int kWinMax = 512;   // token positions per workgroup window (upper bound)
constexpr int kHash = 1024;    // open-addressing table: token id -> first row of the window with that id
constexpr int kMaxDI = 8;      // embedding dim handled per pass = 64 lanes * kMaxDI (scalar path)
constexpr int kMaxQ4 = 2;      // float4 columns per lane and pass (vector path): 512 floats per pass

__global__ __launch_bounds__(256) void dx_window_kernel(const BwdArgs A, const int kWin, const int nwin,
                                                        const long long* __restrict__ ids,
                                                        const unsigned char* __restrict__ mask,
                                                        const float* __restrict__ gate, const float* __restrict__ table,
                                                        const float* __restrict__ packed, const float* __restrict__ feat,
                                                        const int* __restrict__ argmax, const float* __restrict__ d_feat,
                                                        float* __restrict__ dtable, float* __restrict__ dgate) {
    constexpr int kCap = kChanBatch * kMaxKF;
    __shared__ int s_item_w[kCap];          // offset of W[c, 0, j] in the packed image
    __shared__ float s_item_g[kCap];        // g = d_feat * act'(feat)
    __shared__ short s_item_row[kCap];
    __shared__ short s_sorted[kCap];        // item indices grouped by key
    __shared__ int s_tok[kWinMax];          // token id of the row (-1: masked / outside the document)
    __shared__ short s_leader[kWinMax];
    __shared__ short s_cnt[kWinMax], s_start[kWinMax];
    __shared__ int s_fill[kWinMax];
    __shared__ int s_hkey[kHash], s_hval[kHash];
    __shared__ int s_count;
    __shared__ __attribute__((aligned(16))) float s_strip[4 * 64 * kMaxQ4 * 4];   // per-wave transpose strip
    const bool vec4 = (A.D % 4 == 0) && (A.DC % 4 == 0) && ((((uintptr_t)table) & 15) == 0);
    const int doc = blockIdx.x / nwin, p0 = (blockIdx.x % nwin) * kWin;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int L = A.L, D = A.D, C = A.C, KF = A.KF, DC = A.DC;
    const int piece = kTile * DC;                 // one channel tile of the packed image
    const int dcstride = A.tiles_total * piece;   // packed[s][dc][tile][slot][dd]

    for (int k = tid; k < kHash; k += 256) { s_hkey[k] = -1; s_hval[k] = kWinMax; }
    for (int r = tid; r < kWin; r += 256) {
        const int p = p0 + r;
        int t = -1;
        if (p < L && (mask == nullptr || mask[(long)doc * L + p])) t = (int)ids[(long)doc * L + p];
        s_tok[r] = t;
    }
    __syncthreads();
    // leader[r] = first row of the window that carries the same token (hash table, then a min per slot)
    for (int r = tid; r < kWin; r += 256) {
        const int t = s_tok[r];
        int slot = -1;
        if (t >= 0) {
            unsigned hh = ((unsigned)t * 2654435761u) >> 22;   // 10 bits
            for (;;) {
                const int old = atomicCAS(&s_hkey[hh], -1, t);
                if (old == -1 || old == t) break;
                hh = (hh + 1) & (kHash - 1);
            }
            atomicMin(&s_hval[hh], r);
            slot = (int)hh;
        }
        s_leader[r] = (short)slot;       // temporarily the hash slot
    }
    __syncthreads();
    for (int r = tid; r < kWin; r += 256) {
        const int slot = s_leader[r];
        s_leader[r] = (short)(slot >= 0 ? s_hval[slot] : r);
    }

    for (int c0 = 0; c0 < C; c0 += kChanBatch) {
        const int nc = min(kChanBatch, C - c0);
        __syncthreads();
        if (tid == 0) s_count = 0;
        for (int r = tid; r < kWin; r += 256) { s_cnt[r] = 0; s_fill[r] = 0; }
        __syncthreads();
        // ---- collect the items of this window ------------------------------------------------------
        for (int e = tid; e < nc * KF; e += 256) {
            const int cl = e / KF, j = e - cl * KF;
            const int c = c0 + cl;
            const int w = bank_of(A, c);
            const int kz = A.kz[w];
            if (j >= kz) continue;
            const int padl = (A.pad_mode == RBR_PAD_SAME) ? (kz - 1) / 2 : 0;
            const long o = (long)doc * C + c;
            const int p = argmax[o] + j - padl;
            if (p < p0 || p >= p0 + kWin || p >= L) continue;
            if (s_tok[p - p0] < 0) continue;                       // masked token: x was zeroed, no gradient
            const float g = act_grad(A.act, feat[o], d_feat[o]);
            if (g == 0.f) continue;
            const int slot = A.rank_off[w] + (c - A.ch_off[w]);    // kz-sorted slot of the channel
            const int off = (A.pad_mode == RBR_PAD_SAME) ? (KF - kz) / 2 : 0;
            const int t = slot / kTile, i = slot % kTile, s = j + off;
            const int k = atomicAdd(&s_count, 1);
            s_item_row[k] = (short)(p - p0);
            s_item_w[k] = (s * A.nchunks * A.tiles_total + t) * piece + i * DC;
            s_item_g[k] = g;
            atomicAdd(&s_fill[s_leader[p - p0]], 1);          // per-key item count (int atomics)
        }
        __syncthreads();
        if (wave == 0) {   // exclusive scan of the per-key counts by one wave
            int run = 0;
            for (int r0 = 0; r0 < kWin; r0 += 64) {
                const int r = r0 + lane;
                const int c = (r < kWin) ? s_fill[r] : 0;
                int inc = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(inc, o);
                    if (lane >= o) inc += v;
                }
                if (r < kWin) { s_start[r] = (short)(run + inc - c); s_cnt[r] = (short)c; s_fill[r] = 0; }
                run += __shfl(inc, 63);
            }
        }
        __syncthreads();
        const int n = s_count;
        for (int k = tid; k < n; k += 256) {
            const int key = s_leader[s_item_row[k]];
            s_sorted[s_start[key] + atomicAdd(&s_fill[key], 1)] = (short)k;
        }
        __syncthreads();
        // ---- one wave per token: register accumulation, then one row of atomics -------------------
#ifdef RBR_DIAG
        if (A.dev_flags & 2) continue;
#endif
        if (vec4) {
            // lanes own float4 columns (D/4 of them, kMaxQ4 per lane); 4 items are loaded together
            const int nq4 = D >> 2, qpc = DC >> 2;
            for (int key = wave; key < kWin; key += 4) {
                const int cnt = s_cnt[key];
                if (cnt == 0) continue;                               // wave-uniform
                const int base = s_start[key];
                const long trow = (long)s_tok[key] * D;
                const bool to_table = (dtable != nullptr) && (s_tok[key] != A.padding_idx);
                for (int qblk = 0; qblk < nq4; qblk += 64 * kMaxQ4) {
                    int doff[kMaxQ4];
                    f32x4 sum[kMaxQ4], tv[kMaxQ4];
#pragma unroll
                    for (int u = 0; u < kMaxQ4; ++u) {
                        const int q4 = qblk + lane + 64 * u;
                        const int dc = q4 / qpc;
                        doff[u] = (q4 < nq4) ? dc * dcstride + 4 * (q4 - dc * qpc) : -1;
                        sum[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                        tv[u] = (gate != nullptr && q4 < nq4) ? *reinterpret_cast<const f32x4*>(table + trow + 4 * q4)
                                                              : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    for (int q0 = 0; q0 < cnt; q0 += 4) {
                        f32x4 wv[4][kMaxQ4];
                        float gg[4], gv[4];
                        long tok[4];
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const bool ok = q0 + v < cnt;
                            const int it = s_sorted[base + (ok ? q0 + v : q0)];
                            const float* wrow = packed + s_item_w[it];
                            gg[v] = ok ? s_item_g[it] : 0.f;
                            tok[v] = (long)doc * L + p0 + s_item_row[it];
                            gv[v] = (gate != nullptr) ? gate[tok[v]] : 1.f;
#pragma unroll
                            for (int u = 0; u < kMaxQ4; ++u)
                                wv[v][u] = (doff[u] >= 0) ? *reinterpret_cast<const f32x4*>(wrow + doff[u])
                                                          : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            float dot = 0.f;
#pragma unroll
                            for (int u = 0; u < kMaxQ4; ++u) {
                                sum[u] += (gg[v] * gv[v]) * wv[v][u];
                                const f32x4 pr = wv[v][u] * tv[u];
                                dot += gg[v] * (pr.x + pr.y + pr.z + pr.w);
                            }
                            if (gate != nullptr && dgate != nullptr && q0 + v < cnt) {
#pragma unroll
                                for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
                                if (lane == 0) atomicAdd(dgate + tok[v], dot);
                            }
                        }
                    }
#ifdef RBR_DIAG
                    if (A.dev_flags & 1) { if (sum[0].x == 1.2345f) dtable[0] = sum[1].y; }
    #ifdef RBR_DIAG
                if (to_table && !(A.dev_flags & 1)) {
#else
                if (to_table) {
#endif
#else
                    if (to_table) {
#endif
                        // transpose through the wave's LDS strip so each atomic wave-instruction covers 256
                        // contiguous bytes of the table row
                        float* strip = s_strip + wave * (64 * kMaxQ4 * 4);
#pragma unroll
                        for (int u = 0; u < kMaxQ4; ++u)
                            if (doff[u] >= 0) *reinterpret_cast<f32x4*>(strip + 4 * (lane + 64 * u)) = sum[u];
                        const int d0 = 4 * qblk, nd = min(D - d0, 64 * kMaxQ4 * 4);
                        for (int dd = lane; dd < nd; dd += 64) atomicAdd(dtable + trow + d0 + dd, strip[dd]);
                    }
                }
            }
            continue;
        }
        for (int key = wave; key < kWin; key += 4) {
            const int cnt = s_cnt[key];
            if (cnt == 0) continue;                               // wave-uniform
            const int base = s_start[key];
            const long trow = (long)s_tok[key] * D;
            const bool to_table = (dtable != nullptr) && (s_tok[key] != A.padding_idx);
            for (int dblk = 0; dblk < D; dblk += 64 * kMaxDI) {
                int doff[kMaxDI];
                float sum[kMaxDI], tv[kMaxDI];
#pragma unroll
                for (int u = 0; u < kMaxDI; ++u) {
                    const int d = dblk + lane + 64 * u;
                    const int dc = d / DC;
                    doff[u] = (d < D) ? dc * dcstride + (d - dc * DC) : -1;
                    sum[u] = 0.f;
                    tv[u] = (gate != nullptr && d < D) ? table[trow + d] : 0.f;
                }
                for (int q = 0; q < cnt; ++q) {
                    const int it = s_sorted[base + q];
                    const float* wrow = packed + s_item_w[it];
                    const float g = s_item_g[it];
                    float gv = 1.f;
                    const long tok = (long)doc * L + p0 + s_item_row[it];
                    if (gate != nullptr) gv = gate[tok];
                    float dot = 0.f;
#pragma unroll
                    for (int u = 0; u < kMaxDI; ++u) {
                        if (doff[u] >= 0) {
                            const float wv = wrow[doff[u]];
                            sum[u] = fmaf(g * gv, wv, sum[u]);
                            dot = fmaf(g * wv, tv[u], dot);
                        }
                    }
                    if (gate != nullptr && dgate != nullptr) {
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
                        if (lane == 0) atomicAdd(dgate + tok, dot);
                    }
                }
#ifdef RBR_DIAG
                if (to_table && !(A.dev_flags & 1)) {
#else
                if (to_table) {
#endif
#pragma unroll
                    for (int u = 0; u < kMaxDI; ++u)
                        if (doff[u] >= 0) atomicAdd(dtable + trow + dblk + lane + 64 * u, sum[u]);
                }
            }
        }
    }
}

static int fill_args(const rbr_textcnn_desc* d, BwdArgs& A) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return RBR_ERR_BAD_ARG;
    const ConvPlan& p = plans[0];
    memset(&A, 0, sizeof(A));
    A.n_docs = d->n_docs; A.L = d->L; A.D = d->D; A.C = p.C; A.KF = p.KF; A.DC = p.DC; A.nchunks = p.nchunks; A.tiles_total = p.tiles_total;
    A.pad_mode = d->pad_mode; A.act = d->act; A.padding_idx = d->padding_idx;
    A.n_widths = d->n_widths;
    for (int w = 0; w < d->n_widths; ++w) { A.kz[w] = d->kz[w]; A.ch[w] = d->ch[w]; A.ch_off[w] = p.ch_off[w]; }
    // slot rank of bank w = channels of all banks sorted before it (stable by kz), as build_plans orders them
    for (int w = 0; w < d->n_widths; ++w) {
        int r = 0;
        for (int v = 0; v < d->n_widths; ++v)
            if (d->kz[v] < d->kz[w] || (d->kz[v] == d->kz[w] && v < w)) r += d->ch[v];
        A.rank_off[w] = r;
    }
    A.gate_split = RBR_CONV_GATE_SPLIT_OF(d->flags);
    // (the document-centric kernel stages gate * row once for all channels: one gate plane only)
    A.doc_centric = (d->L <= kDocMaxL && p.C * p.KF <= kDocItems && d->n_docs >= 64 && A.gate_split == 0) ? 1 : 0;
    if (A.doc_centric) {
        A.NCH = std::min(128, (d->n_docs + 15) / 16);       // many small chunks: one workgroup per (slice, chunk)
    } else {
        A.NCH = std::min(kMaxChunksBwd, (d->n_docs + kDocsPerBatch - 1) / kDocsPerBatch);
    }
    A.DPC = (d->n_docs + A.NCH - 1) / A.NCH;
    A.NCH = (d->n_docs + A.DPC - 1) / A.DPC;
#ifdef RBR_DIAG      // diagnostic build only (RBR_DIAG=1 python build.py): the product library has no wrong-result switches
    static const int flags = getenv("RBR_DEV_DX_ABLATE") ? atoi(getenv("RBR_DEV_DX_ABLATE")) : 0;
    A.dev_flags = flags;
#else
    A.dev_flags = 0;
#endif
    return 0;
}

}  // namespace rbr

using namespace rbr;

extern "C" size_t rbr_textcnn_bwd_ws_floats(const rbr_textcnn_desc* d) {
    BwdArgs A;
    if (fill_args(d, A)) return 0;
    return (size_t)A.NCH * A.C * A.KF * A.D + (size_t)A.NCH * A.C;
}

extern "C" int rbr_textcnn_bwd_dw(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                                  const float* table, const float* feat, const int32_t* argmax, const float* d_feat,
                                  float* const* dW, float* const* dbias, float* ws, void* stream) {
    BwdArgs A;
    if (int e = fill_args(d, A)) return e;
    if (!ids || !table || !feat || !argmax || !d_feat || !dW || !dbias || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const long long* ids64 = reinterpret_cast<const long long*>(ids);
    float* ws_w = ws;
    float* ws_b = ws + (size_t)A.NCH * A.C * A.KF * A.D;
    if (A.doc_centric) {
        hipLaunchKernelGGL(dw_doc_kernel, dim3((A.D + kDocSlice - 1) / kDocSlice, A.NCH), dim3(256), 0, st, A, ids64, mask,
                           gate, table, feat, argmax, d_feat, ws_w, ws_b);
        RBR_CHECK_LAUNCH("textcnn dw_doc launch");
    } else {
        // float4 rows when the layout allows it (16-byte aligned rows, one row <= 256 float4 lanes)
        const bool vec4 = (A.D % 4 == 0) && (A.D / 4 <= 256) && ((((uintptr_t)table) & 15) == 0) && ((((uintptr_t)ws_w) & 15) == 0);
        int taps_per_thread = 1;      // widest bank: ceil(kz / min(slots, kz))
        if (vec4)
            for (int w = 0; w < d->n_widths; ++w) {
                const int nslots = 256 / (A.D / 4), js = std::min(nslots, d->kz[w]);
                taps_per_thread = std::max(taps_per_thread, (d->kz[w] + js - 1) / js);
            }
        if (vec4 && taps_per_thread <= 1) {
            if (int e_ = rbr::launch<dw_partial4_kernel<1>, 256>(dim3(A.C, A.NCH), dim3(256), 0, st, "textcnn dw_partial4 launch", A, ids64, mask, gate, table, feat, argmax, d_feat, ws_w, ws_b)) return e_;
        } else if (vec4 && taps_per_thread <= 3) {
            if (int e_ = rbr::launch<dw_partial4_kernel<3>, 256>(dim3(A.C, A.NCH), dim3(256), 0, st, "textcnn dw_partial4 launch", A, ids64, mask, gate, table, feat, argmax, d_feat, ws_w, ws_b)) return e_;
        } else {      // odd widths, or rows so long that one slot would own > 3 taps: the scalar-column kernel
            hipLaunchKernelGGL(dw_partial_kernel, dim3(A.C, A.NCH), dim3(256), 0, st, A, ids64, mask, gate, table, feat, argmax,
                               d_feat, ws_w, ws_b);
            RBR_CHECK_LAUNCH("textcnn dw_partial launch");
        }
    }
    MutPtrArray dWp{}, dbp{};
    for (int w = 0; w < d->n_widths; ++w) { dWp.p[w] = dW[w]; dbp.p[w] = dbias[w]; }
    if (int e_ = rbr::launch<dw_reduce_kernel, 256>(dim3((unsigned)A.C, (unsigned)((A.D + kRedCols - 1) / kRedCols)), dim3(256), 0, st, "textcnn dw_reduce launch", A, ws_w, ws_b, dWp, dbp)) return e_;
    return 0;
}

extern "C" int rbr_textcnn_bwd_dtable(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                                      const float* table, const float* packed, const float* feat, const int32_t* argmax,
                                      const float* d_feat, float* dtable, float* dgate, void* stream) {
    if (d && gate != nullptr && RBR_CONV_GATE_SPLIT_OF(d->flags)) { set_error("rbr_textcnn_bwd_dtable: RBR_CONV_GATE_SPLIT needs the token-product formulation"); return RBR_ERR_UNSUPPORTED; }
    BwdArgs A;
    if (int e = fill_args(d, A)) return e;
    if (!ids || !table || !packed || !feat || !argmax || !d_feat) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if (dtable == nullptr && !(dgate != nullptr && gate != nullptr)) return 0;
#ifdef RBR_DIAG
    static const int win_env = getenv("RBR_DEV_DX_WIN") ? atoi(getenv("RBR_DEV_DX_WIN")) : 0;   // tuning aid
#else
    constexpr int win_env = 0;
#endif
    int kWin = win_env > 0 ? std::min(win_env, kWinMax) : 256;
    kWin = std::min(kWin, ((A.L + 63) / 64) * 64);
    const int nwin = (A.L + kWin - 1) / kWin;
    hipLaunchKernelGGL(dx_window_kernel, dim3((unsigned)(A.n_docs * nwin)), dim3(256), 0, (hipStream_t)stream, A, kWin, nwin,
                       reinterpret_cast<const long long*>(ids), mask, gate, table, packed, feat, argmax, d_feat, dtable,
                       (gate != nullptr) ? dgate : nullptr);
    RBR_CHECK_LAUNCH("textcnn dx_window launch");
    return 0;
}

extern "C" int rbr_textcnn_bwd(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                               const float* table, const float* packed, const float* feat, const int32_t* argmax,
                               const float* d_feat, float* const* dW, float* const* dbias, float* dtable, float* dgate,
                               float* ws, void* stream) {
    if (int e = rbr_textcnn_bwd_dw(d, ids, mask, gate, table, feat, argmax, d_feat, dW, dbias, ws, stream)) return e;
    return rbr_textcnn_bwd_dtable(d, ids, mask, gate, table, packed, feat, argmax, d_feat, dtable, dgate, stream);
}
