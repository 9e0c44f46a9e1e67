// textcnn_fwd.hip -- fused  embedding gather -> mask/gate -> multi-width conv1d -> max-pool
// for gfx950 (MI355X), exact fp32 on v_mfma_f32_32x32x2_f32.
//
// Replaces, for one batch of documents, the ATen sequence of the reference's NgramFeat
// (models/deepconn/layers.py:123-136): aten::embedding (layers.py:23), masked_fill
// (utils.py:60), transpose (layers.py:132), conv1d per width + cat (layers.py:55-58),
// relu + max_pool1d (layers.py:107-109).  Nothing of shape [docs, L, D] or [docs, C, L]
// is ever written to HBM: the only outputs are (max, argmax) per document and channel.
//
// Work decomposition
//   wave-tile  = 32 consecutive token positions of one document (one MFMA M-block); tiles whose tokens are all
//                masked are never computed (tile_scan_kernel builds the work list, their pooled value is exactly 0);
//   item       = 4 wave-tiles of the work list = one pass of a 4-wave workgroup; persistent workgroups (2 per CU,
//                i.e. 2 waves per SIMD) pull items from a device counter;
//   GEMM view  : out[pos, chan] = sum_{tap s} sum_{d} X[pos + s - P, d] * Wf[chan, s, d]
//                M = positions, N = channel slots (tiles of 32), K = (tap, d).
//   The K loop is cut into "pieces" = (embedding chunk dc of DC floats, tap s, one or two adjacent channel tiles):
//   [32 slots][DC] weight blocks that all 4 waves share through a 2-slot LDS ring, 60 MFMAs per wave and paired
//   piece (DC = 60).  Each wave keeps its own [32 + KF - 1][DC] slab of gathered token rows in LDS and re-reads
//   it for every tap and tile of the chunk.
//   Both the weight pieces and the token rows reach LDS by LDS-DMA (global_load_lds_dwordx4: no staging
//   registers; per-lane source address = row gather); the requests for piece q+1 are issued from inside the
//   dependent MFMA chain of piece q, where the wave has idle issue slots.
//
// MFMA operand mapping (v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5], B[k][j = l&31]):
//   both operands are fetched with ONE ds_read_b128 per 4 MFMAs: lane (i, h) reads floats
//   d0+4h .. d0+4h+3 of its row, and MFMA e (0..3) uses element e, i.e. sums the K pair
//   {d0+e, d0+4+e}.  K order inside a chunk is a permutation -- legal because A and B agree.
//   Row stride DC floats with DC/4 odd => the b128 reads are bank-conflict free.
#include "rbr_common.h"

#include <cstdlib>

namespace rbr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------ pack
// packed[s][dc][tile][slot][dd] = W_w[chan_local][d = dc*DC+dd][j = s - off]   (0 outside); tiles are global
// (all groups), so the two tiles of a paired piece are contiguous.
__global__ __launch_bounds__(256) void pack_kernel(const ConvPlan P, const PtrArray W, float* __restrict__ packed) {
    const int DC = P.DC;
    const long total = (long)P.KF * P.nchunks * P.ntiles * kTile * DC;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int dd = (int)(r % DC); r /= DC;
        const int slot = (int)(r % kTile); r /= kTile;
        const int t = (int)(r % P.ntiles); r /= P.ntiles;
        const int dc = (int)(r % P.nchunks);
        const int s = (int)(r / P.nchunks);
        const int ls = t * kTile + slot;
        const int chan = P.slot_chan[ls];
        float v = 0.f;
        if (chan >= 0) {
            const int w = P.slot_w[ls], kz = P.slot_kz[ls];
            const int j = s - (int)P.slot_off[ls];
            const int d = dc * DC + dd;
            if (j >= 0 && j < kz && d < P.D) v = W.p[w][((long)(chan - P.ch_off[w]) * P.D + d) * kz + j];
        }
        packed[((((long)s * P.nchunks + dc) * P.tiles_total + P.tile_base + t) * kTile + slot) * DC + dd] = v;
    }
}

// ------------------------------------------------------------------------------------ tile scan
// Mask-aware work list.  A wave-tile (32 positions + halo) whose tokens are ALL masked gathers only zero rows:
// every accumulator stays exactly 0.0f, so its pooled value is known (0 at the tile's first position) without
// issuing a single MFMA.  Right-padded review documents make ~1/3 of the tiles of the cfg2 batch such tiles.
// sched layout (int32, tail of the pidx workspace): flags[total_wt] | list[total_wt] | counter[kSchedCounters]
// (counter[0] = active tiles, counter[1 + g] = next work item of launch group g).
__global__ __launch_bounds__(256) void tile_scan_kernel(const ConvPlan P, const unsigned char* __restrict__ mask,
                                                        int* __restrict__ sched) {
    tile_scan_block(P, mask, sched, blockIdx.x);
}

// ------------------------------------------------------------------------------------ conv
// LDS-DMA: 16 bytes per lane straight from global memory into LDS (no VGPR staging).  The LDS destination is
// `lds_base` (wave-uniform) + lane * 16; the global source is per lane, which is what makes it a row gather.
__device__ float g_zero_row[64];   // source of masked / out-of-document rows (zero-initialised, never written)

__device__ __forceinline__ void dma16(const float* gsrc, float* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

template <int DC>
struct Frag {   // one operand of a piece: DC/8 ds_read_b128 plus a b64 tail when DC % 8 == 4
    f32x4 v[DC / 8];
    f32x2 t;
    __device__ __forceinline__ void load(const float* __restrict__ p, int h) {
        // p already includes the lane's row and the 4*h column offset
#pragma unroll
        for (int q = 0; q < DC / 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(p + 8 * q);
        t = f32x2{0.f, 0.f};
        if (DC % 8 == 4) t = *reinterpret_cast<const f32x2*>(p - 4 * h + (DC - 4) + 2 * h);   // columns DC-4+2h, +1
    }
};

// All LDS operands of a piece are requested up front: one wave alone then keeps the MFMA pipe fed (issuing each
// read pair behind the previous MFMA group left ~40 % of the pipe idle).
// `between(r)` is invoked after every second k-step (r = 0, 1, ...): the chain is one dependent accumulator, so the
// wave has ~60 idle issue cycles behind each MFMA -- the LDS-DMA requests of the next piece are slipped in there
// instead of being paid for (~100 cycles each) in front of the chain.
template <int DC, class Between>
__device__ __forceinline__ void mma_chain(f32x16& acc, const Frag<DC>& a, const Frag<DC>& b, Between&& between) {
#pragma unroll
    for (int q = 0; q < DC / 8; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].x, b.v[q].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].y, b.v[q].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].z, b.v[q].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].w, b.v[q].w, acc, 0, 0, 0);
        if (q & 1) between(q >> 1);
    }
    if (DC % 8 == 4) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.t.x, b.t.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.t.y, b.t.y, acc, 0, 0, 0);
    }
}

#ifdef RBR_DIAG
// tuning build only (tools/dev_conv_diag.py): per-workgroup cycle counts of the item-loop segments
__device__ unsigned long long g_diag[8 * 1024];
#define RBR_STAMP(slot)                                                                         \
    do {                                                                                        \
        unsigned long long _t;                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");             \
        diag[slot] += _t - t_last;                                                              \
        t_last = _t;                                                                            \
    } while (0)
#else
#define RBR_STAMP(slot) do { } while (0)
#endif

// STORE = the token-product GEMM (P.store_rows, one tap): the accumulators are written out as rows of T; the token
// fragment of a chunk is read from LDS once and kept in registers for all its pieces, and the freed slab receives the
// NEXT chunk's rows while the current chunk computes (every workgroup holds exactly one item there, so co-resident
// workgroups run in phase and cannot hide each other's gathers).
template <int NT, int DC, bool VEC, bool STORE>
__device__ __forceinline__ void conv_body(const ConvPlan& P, const long long* __restrict__ ids,
                                          const unsigned char* __restrict__ mask, const float* __restrict__ gate,
                                          const float* __restrict__ table, const float* __restrict__ packed,
                                          float* __restrict__ pval, int* __restrict__ pidx, const int* __restrict__ sched) {
    static_assert(DC % 4 == 0 && (DC / 4) % 2 == 1, "row stride must be 4*odd floats (bank-conflict-free b128 reads)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE_F = kTile * DC;             // floats of one channel tile of a piece
    constexpr int PIECE = 2 * TILE_F;              // LDS ring slot: up to two tiles
    constexpr int NLD = (PIECE / 4 + 255) / 256;   // float4 prefetch registers per thread
    const int XR = kTile + P.KF - 1;               // token rows per wave slab
    float* Ws = smem;                              // [2][2][32][DC]
    float* Xs = smem + 2 * PIECE;                  // [4 waves][XR][DC]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int L = P.L, D = P.D;
    float* Xw = Xs + wave * XR * DC;
    long* s_row = reinterpret_cast<long*>(Xs + kWavesPerWG * XR * DC) + wave * (kTile + kMaxKF);   // [XR] per wave
    int* s_item = reinterpret_cast<int*>(reinterpret_cast<long*>(Xs + kWavesPerWG * XR * DC) + kWavesPerWG * (kTile + kMaxKF));
    // store_rows launches fold all (identical) channel-tile groups into one grid: item -> (row item, group)
    const int ngroups_in_launch = STORE ? P.tiles_total / P.ntiles : 1;
    // wave-tiles with at least one unmasked token; store_rows: the slot holds the ROW count of the token list and
    // the active tiles are simply the first ceil(rows / 32)
    const int n_active = STORE ? (sched[2 * (long)P.total_wt] + kTile - 1) / kTile : sched[2 * (long)P.total_wt];
    const int* worklist = sched + P.total_wt;
    const int nrow_items = (n_active + kWavesPerWG - 1) / kWavesPerWG;
    const int nitems = nrow_items * ngroups_in_launch;

  // Persistent workgroups pull items (4 wave-tiles of the work list) from a device counter.  Items cost the same,
  // but their number per CU is fractional (e.g. 1357 items on 512 resident workgroups): with dynamic pulling the
  // last items run on CUs whose other workgroup has already drained, i.e. with the MFMA pipe to themselves.
  int* item_counter = const_cast<int*>(sched) + 2 * (long)P.total_wt + 1 + P.group;
#ifdef RBR_DIAG
  unsigned long long diag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
#endif
  // the first item of a workgroup is its own index (no round trip to the counter); later items are pulled.  When the
  // grid covers all items (the token-product GEMM: 501 items, 512 resident workgroups) nothing is ever pulled.
  bool first_item = true;
  for (;;) {
    int item;
    if (first_item) {
        item = blockIdx.x;
        first_item = false;
    } else {
        if ((int)gridDim.x >= nitems) break;
        __syncthreads();   // every wave is done with the LDS ring (and with *s_item) of the previous item
        if (tid == 0) *s_item = (int)gridDim.x + atomicAdd(item_counter, 1);
        __syncthreads();
        item = *s_item;
    }
    if (item >= nitems) break;
    RBR_STAMP(0);   // item pull
    const int grp = item / nrow_items;                 // 0 unless store_rows
    const int tile_base = P.tile_base + grp * P.ntiles;
    const float* wbase = packed + (long)tile_base * TILE_F;
    const int slot_in_list = (item - grp * nrow_items) * kWavesPerWG + wave;
    const bool active = slot_in_list < n_active;      // wave-uniform
    const int wt = active ? (STORE ? slot_in_list : worklist[slot_in_list]) : 0;   // global wave-tile
    const int doc = wt / P.wpd;
    const int l0 = (wt % P.wpd) * kTile;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Weight pieces and token rows travel global -> LDS by LDS-DMA (global_load_lds_dwordx4): piece q+1 is
    // requested into the free ring slot before piece q computes and is waited for (vmcnt(0), emitted by
    // __syncthreads) at the end of piece q.  No staging registers, no ds_write pass; the packed image is already
    // laid out exactly as the ring slot (lane-linear), and so is the token slab ([row][DC] = 16-byte granules in
    // gather order).  `st` = tap | first tile << 8 | tile count << 16; q+1, q+2 live in SGPRs, q+3 is in flight.
    const int np = P.npieces;
    const int T = P.nchunks * np;                       // pieces of one item
    auto piece_src = [&](int dcq, int st) -> const float* {
        const int s = st & 0xff, t = (st >> 8) & 0xff;
        return wbase + (long)((s * P.nchunks + dcq) * P.tiles_total + t) * TILE_F;
    };
    auto nv4_of = [&](int st) -> int { return (st >> 16) * (TILE_F / 4); };
    auto issue_round = [&](int k, float* dst, const float* src, int nv4) {   // all 4 waves: 256 float4 per round
        const int v0 = 256 * k + 64 * wave;                       // wave-uniform first granule of this instruction
        if (v0 + lane < nv4) dma16(src + 4 * (v0 + lane), dst + 4 * v0);
    };
    auto issue = [&](float* dst, const float* src, int nv4) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) issue_round(k, dst, src, nv4);
    };
    auto wrap = [&](int k) -> int {           // k mod np for 0 <= k < np + 3, without an integer division
        while (k >= np) k -= np;
        return k;
    };

    int st0 = P.piece_st[0], st1 = P.piece_st[wrap(1)], st2 = P.piece_st[wrap(2)];
    issue(Ws, piece_src(0, st0), nv4_of(st0));
    int cur = 0;
    if (active) {   // table row offset of every slab row (-1: outside the document or masked), once per item
        for (int row = lane; row < XR; row += 64) {
            const int p = l0 - P.P + row;
            long ro = -1;
            if (p >= 0 && p < L) {
                const long tok = (long)doc * L + p;
                if (mask == nullptr || mask[tok]) ro = ids[tok] * (long)D;
            }
            s_row[row] = ro;
        }
    }
    RBR_STAMP(1);   // item prologue: first piece request + row offsets

    // gather this wave's token rows for columns [dcq*DC, dcq*DC+DC) into its slab
    auto gather_chunk = [&](int dcq) {
        if (VEC) {
            constexpr int QPR = DC / 4;
            constexpr int NX = ((kTile + (STORE ? 0 : kMaxKF - 1)) * QPR + 63) / 64;
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                const int idx = lane + 64 * k;
                if (idx < XR * QPR) {
                    const int row = idx / QPR, qq = idx - row * QPR;
                    const int d = dcq * DC + 4 * qq;
                    const long ro = s_row[row];
                    const float* src = (ro >= 0 && d < D) ? table + ro + d : g_zero_row;
                    dma16(src, Xw + 256 * k);     // granule idx lands at Xw + 4*idx = row*DC + 4*qq
                }
            }
            if (!STORE && gate != nullptr) {
                // gated rows (D-ATT: x * score): scale the slab in place once the wave's own DMA has landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int idx = lane; idx < XR * QPR; idx += 64) {
                    const int row = idx / QPR;
                    const int p = l0 - P.P + row;
                    if (p >= 0 && p < L) {
                        f32x4* cell = reinterpret_cast<f32x4*>(Xw + 4 * idx);
                        *cell = *cell * gate[(long)doc * L + p];
                    }
                }
            }
        } else {
            for (int idx = lane; idx < XR * DC; idx += 64) {
                const int row = idx / DC, dd = idx - row * DC;
                float v = 0.f;
                const int d = dcq * DC + dd;
                const long ro = s_row[row];
                if (ro >= 0 && d < D) {
                    v = table[ro + d];
                    if (!STORE && gate != nullptr) v *= gate[(long)doc * L + l0 - P.P + row];
                }
                Xw[idx] = v;
            }
        }
    };

    int pi = 0, dc = 0;                       // piece q = dc * np + pi, tracked incrementally (no divisions)
    Frag<DC> a_keep;                          // STORE: token fragment of the current chunk
    for (int q = 0; q < T; ++q) {
        if (pi == 0 && (!STORE || dc == 0)) {
            if (active) gather_chunk(dc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA requests have landed ...
            __syncthreads();                                   // ... and so have everyone else's
            RBR_STAMP(2);   // row gather of the chunk (+ its barrier)
        }
        if (STORE && pi == 0 && active) {
            // chunk dc's rows were requested while chunk dc-1 computed and have landed (every piece ends in vmcnt(0));
            // read them once, then hand the slab to the next chunk's gather
            a_keep.load(Xw + i * DC + 4 * h, h);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (dc + 1 < P.nchunks) gather_chunk(dc + 1);
        }
        const int st3 = P.piece_st[wrap(pi + 3)];         // consumed three pieces from now
        // piece q+1 goes into the other ring slot (free since the last barrier); its DMA requests are issued from
        // inside the MFMA chain of the first tile (inactive waves issue them up front)
        const bool more = q + 1 < T;
        float* nxt_dst = Ws + (cur ^ 1) * PIECE;
        const float* nxt_src = piece_src((pi + 1 == np) ? dc + 1 : dc, st1);
        const int nxt_nv4 = more ? nv4_of(st1) : 0;
        constexpr int RC = (DC / 8) / 2;                  // DMA rounds that fit inside one chain
        RBR_STAMP(3);   // prefetch address setup
        if (active) {
            const int s = st0 & 0xff, t = (st0 >> 8) & 0xff, nt = st0 >> 16;
            const float* wb = Ws + cur * PIECE + i * DC + 4 * h;
            Frag<DC> a_tap, b;
            if (!STORE) a_tap.load(Xw + (i + s) * DC + 4 * h, h);
            const Frag<DC>& a = STORE ? a_keep : a_tap;
            b.load(wb, h);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                if (tt == t) {
                    mma_chain<DC>(acc[tt], a, b, [&](int r) { if (r < NLD) issue_round(r, nxt_dst, nxt_src, nxt_nv4); });
#pragma unroll
                    for (int r = RC; r < NLD; ++r) issue_round(r, nxt_dst, nxt_src, nxt_nv4);
                    if (tt + 1 < NT && nt == 2) {     // second tile of the pair: same token rows
                        b.load(wb + TILE_F, h);
                        mma_chain<DC>(acc[tt + 1], a, b, [](int) {});
                    }
                }
            }
        } else {
            issue(nxt_dst, nxt_src, nxt_nv4);
        }
        RBR_STAMP(4);   // LDS operand reads + MFMA chains
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // piece q+1: this wave's share has landed (explicit: LDS-DMA
        __syncthreads();                                   // is ordered for readers only by vmcnt + a barrier)
        RBR_STAMP(6);   // wait + barrier
        cur ^= 1;
        st0 = st1; st1 = st2; st2 = st3;
        if (++pi == np) { pi = 0; ++dc; }
    }

    // ---- epilogue -----------------------------------------------------------------------------------------
    if (STORE) {
        // token-product table: out[row, slot] = accumulator (rows of the pseudo-document = distinct tokens).  Each
        // 32 x 32 tile goes through the wave's own (now idle) slab so that a store instruction writes 8 rows x 128
        // contiguous bytes as float4 -- 4 instructions per tile instead of 16 half-coalesced dword stores.
        constexpr int TS = 36;                         // slab row stride: 16-byte aligned rows, DC >= 20 floats -> 32*36 <= 32*DC? see static_assert
        static_assert(32 * TS <= kTile * DC || DC < TS, "transpose staging must fit the wave's slab");
        if (DC >= TS) {
            const int trow = lane >> 3, tcq = lane & 7;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                if (active && tt < P.ntiles) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) Xw[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + i] = acc[tt][r];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass) {
                        const int rr = trow + 8 * pass;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(Xw + rr * TS + 4 * tcq);
                        const int row = l0 + rr;
                        if (row < L) *reinterpret_cast<f32x4*>(pval + (long)row * P.nslots_total + (long)(tile_base + tt) * kTile + 4 * tcq) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        } else {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                if (active && tt < P.ntiles) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = l0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (row < L) pval[(long)row * P.nslots_total + (long)(tile_base + tt) * kTile + i] = acc[tt][r];
                    }
                }
            }
        }
        RBR_STAMP(7);   // epilogue
        __syncthreads();   // the slab is rewritten by the next item's gather
        continue;
    }
    // max + first argmax over this wave's 32 positions, per channel slot
    const float NEG = -__builtin_huge_valf();
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        if (active && tt < P.ntiles) {
            const int ls = tt * kTile + i;
            const int kz = P.slot_kz[ls];
            const int Lv = (P.pad_mode == RBR_PAD_VALID) ? (L - kz + 1) : L;   // pool length of this channel
            float best = NEG;
            int bidx = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {   // ascending position order; strict '>' keeps the first maximum
                const int pos = l0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = acc[tt][r];
                if (pos < Lv && v > best) { best = v; bidx = pos; }
            }
            const float ob = __shfl_xor(best, 32);
            const int oi = __shfl_xor(bidx, 32);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
            if (h == 0) {
                const long o = (long)wt * P.nslots_total + (long)(tile_base + tt) * kTile + i;
                pval[o] = best;
                pidx[o] = bidx;
            }
        }
    }
    RBR_STAMP(7);   // epilogue
  }  // item loop
#ifdef RBR_DIAG
    if (tid == 0 && blockIdx.x < 1024) {
#pragma unroll
        for (int k = 0; k < 8; ++k) g_diag[8 * blockIdx.x + k] = diag[k];
    }
#endif
}

// ------------------------------------------------------------------------------------ token-product GEMM, static form
// The same computation as conv_body<8, DC, true, true> for the common shape (one tap, groups of exactly 8 channel
// tiles, 16-byte aligned rows), written with everything static: the four tile pairs of a chunk are unrolled, so the
// accumulator tiles have fixed registers; the weight operands of a pair stream through a three-quad register window
// (requested two k-steps ahead) and the two accumulation chains are interleaved -- no LDS latency and no dependent-
// issue gap between MFMAs, and no second whole weight fragment in registers.
template <int DC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void prod_gemm_kernel(const ConvPlan P, const long long* __restrict__ ids, const unsigned char* __restrict__ mask,
                      const float* __restrict__ table, const float* __restrict__ packed, float* __restrict__ out,
                      const int* __restrict__ sched) {
    static_assert(DC % 4 == 0 && (DC / 4) % 2 == 1, "row stride must be 4*odd floats");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = 8, TILE_F = kTile * DC, PIECE = 2 * TILE_F, NLD = (PIECE / 4 + 255) / 256, Q = DC / 8;
    constexpr int XR = kTile, QPR = DC / 4, NX = (XR * QPR + 63) / 64;
    float* Ws = smem;                              // [2 slots][2 tiles][32][DC]
    float* Xs = smem + 2 * PIECE;                  // [4 waves][32][DC]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int L = P.L, D = P.D;
    float* Xw = Xs + wave * XR * DC;
    long* s_row = reinterpret_cast<long*>(Xs + kWavesPerWG * XR * DC) + wave * XR;
    int* s_item = reinterpret_cast<int*>(reinterpret_cast<long*>(Xs + kWavesPerWG * XR * DC) + kWavesPerWG * XR);
    const int ngroups = P.tiles_total / NT;
    const int n_active = (sched[2 * (long)P.total_wt] + kTile - 1) / kTile;
    const int nrow_items = (n_active + kWavesPerWG - 1) / kWavesPerWG;
    const int nitems = nrow_items * ngroups;
    int* item_counter = const_cast<int*>(sched) + 2 * (long)P.total_wt + 1 + P.group;

    auto issue_round = [&](int k, float* dst, const float* src) {      // one LDS-DMA instruction of a 2-tile piece
        const int v0 = 256 * k + 64 * wave;
        if (v0 + lane < PIECE / 4) dma16(src + 4 * (v0 + lane), dst + 4 * v0);
    };

#ifdef RBR_DIAG
    unsigned long long diag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
    diag[5] = t_last;                                    // absolute start
#endif
    bool first_item = true;
    for (;;) {
        int item;
        if (first_item) {
            item = blockIdx.x;
            first_item = false;
        } else {
            if ((int)gridDim.x >= nitems) break;
            __syncthreads();
            if (tid == 0) *s_item = (int)gridDim.x + atomicAdd(item_counter, 1);
            __syncthreads();
            item = *s_item;
        }
        if (item >= nitems) break;
        const int grp = item / nrow_items;
        const int tile_base = grp * NT;
        const float* wbase = packed + (long)tile_base * TILE_F;
        const int slot_in_list = (item - grp * nrow_items) * kWavesPerWG + wave;
        const bool active = slot_in_list < n_active;
        const int l0 = (active ? slot_in_list : 0) * kTile;

        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        auto piece_src = [&](int dcq, int pair) -> const float* { return wbase + ((long)dcq * P.tiles_total + 2 * pair) * TILE_F; };
#pragma unroll
        for (int k = 0; k < NLD; ++k) issue_round(k, Ws, piece_src(0, 0));
        if (active) {
            for (int row = lane; row < XR; row += 64) {
                const int p = l0 + row;
                long ro = -1;
                if (p < L && (mask == nullptr || mask[p])) ro = ids[p] * (long)D;
                s_row[row] = ro;
            }
        }
        auto gather_chunk = [&](int dcq) {
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                const int idx = lane + 64 * k;
                if (idx < XR * QPR) {
                    const int row = idx / QPR, qq = idx - row * QPR;
                    const int d = dcq * DC + 4 * qq;
                    const long ro = s_row[row];
                    dma16((ro >= 0 && d < D) ? table + ro + d : g_zero_row, Xw + 256 * k);
                }
            }
        };
        if (active) gather_chunk(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        RBR_STAMP(2);

        int cur = 0;
        Frag<DC> a;
        for (int dc = 0; dc < P.nchunks; ++dc) {
            if (active) {
                a.load(Xw + i * DC + 4 * h, h);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (dc + 1 < P.nchunks) gather_chunk(dc + 1);       // the slab is free again: next chunk's rows behind the MFMAs
            }
#pragma unroll
            for (int pair = 0; pair < NT / 2; ++pair) {
                const bool last = (dc + 1 == P.nchunks) && (pair == NT / 2 - 1);
                float* nxt_dst = Ws + (cur ^ 1) * PIECE;
                const float* nxt_src = (pair + 1 < NT / 2) ? piece_src(dc, pair + 1) : piece_src(dc + 1, 0);
                if (active) {
                    const float* w0p = Ws + cur * PIECE + i * DC + 4 * h;
                    const float* w1p = w0p + TILE_F;
                    f32x4 w0[3], w1[3];
                    w0[0] = *reinterpret_cast<const f32x4*>(w0p);
                    w1[0] = *reinterpret_cast<const f32x4*>(w1p);
                    w0[1] = *reinterpret_cast<const f32x4*>(w0p + 8);
                    w1[1] = *reinterpret_cast<const f32x4*>(w1p + 8);
                    f32x2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
#if defined(RBR_DIAG_VARIANT) && RBR_DIAG_VARIANT == 2
                        if (false) {
#else
                        if (q + 2 < Q) {
#endif
                            w0[(q + 2) % 3] = *reinterpret_cast<const f32x4*>(w0p + 8 * (q + 2));
                            w1[(q + 2) % 3] = *reinterpret_cast<const f32x4*>(w1p + 8 * (q + 2));
                        } else if (q + 2 == Q && DC % 8 == 4) {
                            t0 = *reinterpret_cast<const f32x2*>(w0p - 4 * h + (DC - 4) + 2 * h);
                            t1 = *reinterpret_cast<const f32x2*>(w1p - 4 * h + (DC - 4) + 2 * h);
                        }
                        const f32x4 x0 = w0[q % 3], x1 = w1[q % 3];
                        f32x16& c0 = acc[2 * pair];
                        f32x16& c1 = acc[2 * pair + 1];
                        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].x, x0.x, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].x, x1.x, c1, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].y, x0.y, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].y, x1.y, c1, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].z, x0.z, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].z, x1.z, c1, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].w, x0.w, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q].w, x1.w, c1, 0, 0, 0);
#if !defined(RBR_DIAG_VARIANT) || RBR_DIAG_VARIANT != 1
                        if (!last && q < NLD) issue_round(q, nxt_dst, nxt_src);
#endif
                    }
                    if (DC % 8 == 4) {
                        f32x16& c0 = acc[2 * pair];
                        f32x16& c1 = acc[2 * pair + 1];
                        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.t.x, t0.x, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.t.x, t1.x, c1, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.t.y, t0.y, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.t.y, t1.y, c1, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = Q; r < NLD; ++r)
                        if (!last) issue_round(r, nxt_dst, nxt_src);
                } else if (!last) {
#pragma unroll
                    for (int k = 0; k < NLD; ++k) issue_round(k, nxt_dst, nxt_src);
                }
                RBR_STAMP(4);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !defined(RBR_DIAG_VARIANT) || RBR_DIAG_VARIANT != 3
                __syncthreads();
#endif
                RBR_STAMP(6);
                cur ^= 1;
            }
        }
        // epilogue: each 32 x 32 tile through the wave's idle slab, 8 rows x 128 contiguous bytes per store instruction
        constexpr int TS = 36;
        static_assert(32 * TS <= kTile * DC, "transpose staging must fit the wave's slab");
        const int trow = lane >> 3, tcq = lane & 7;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            if (active) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Xw[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + i] = acc[tt][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int rr = trow + 8 * pass;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Xw + rr * TS + 4 * tcq);
                    const int row = l0 + rr;
                    if (row < L) *reinterpret_cast<f32x4*>(out + (long)row * P.nslots_total + (long)(tile_base + tt) * kTile + 4 * tcq) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        RBR_STAMP(7);
    }
#ifdef RBR_DIAG
    if (tid == 0 && blockIdx.x < 1024) {
        diag[3] = t_last;                                // absolute end
#pragma unroll
        for (int k = 0; k < 8; ++k) g_diag[8 * blockIdx.x + k] = diag[k];
    }
#endif
}

// pooling conv: up to 5 channel tiles per launch group (2 waves/SIMD with the default register split)
template <int NT, int DC, bool VEC>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvPlan P, const long long* __restrict__ ids,
                                                       const unsigned char* __restrict__ mask,
                                                       const float* __restrict__ gate, const float* __restrict__ table,
                                                       const float* __restrict__ packed, float* __restrict__ pval,
                                                       int* __restrict__ pidx, const int* __restrict__ sched) {
    conv_body<NT, DC, VEC, false>(P, ids, mask, gate, table, packed, pval, pidx, sched);
}

// token-product GEMM (P.store_rows): compiled for exactly 2 waves/SIMD, i.e. the unified 512-entry register file is
// split 256 / 256 -- 8 accumulator tiles (128 registers) fit without spills
template <int NT, int DC, bool VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_store_kernel(const ConvPlan P, const long long* __restrict__ ids, const unsigned char* __restrict__ mask,
                       const float* __restrict__ table, const float* __restrict__ packed, float* __restrict__ out,
                       const int* __restrict__ sched) {
    conv_body<NT, DC, VEC, true>(P, ids, mask, nullptr, table, packed, out, nullptr, sched);
}

// ------------------------------------------------------------------------------------ finalize
// one thread per (doc, slot): reduce the wpd slabs in position order, bias + activation.
__device__ __forceinline__ void pool_finalize_kernel(const ConvPlan& P, const float* __restrict__ pval,
                                                            const int* __restrict__ pidx, const PtrArray& bias,
                                                            float* __restrict__ feat, int* __restrict__ argmax) {
    const int* flags = pidx + (long)P.total_wt * P.nslots_total;   // sched region written by tile_scan_kernel
    const int nslots = P.ntiles * kTile;
    const long total = (long)P.n_docs * nslots;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int doc = (int)(idx / nslots), ls = (int)(idx % nslots);
        const int chan = P.slot_chan[ls];
        if (chan < 0) continue;
        const long base = (long)doc * P.wpd * P.nslots_total + (long)P.tile_base * kTile + ls;
        float best = -__builtin_huge_valf();
        int bw_tile = -1;        // tile holding the maximum; -2 - w for an all-masked tile w (index = its first position)
        const int Lv = (P.pad_mode == RBR_PAD_VALID) ? (P.L - (int)P.slot_kz[ls] + 1) : P.L;
        // tiles in position order, first maximum wins.  The loads of a batch of 8 tiles are independent (issued together);
        // the position index is fetched once, for the winning tile only.
        for (int w0 = 0; w0 < P.wpd; w0 += 8) {
            int fl[8];
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) fl[q] = (w0 + q < P.wpd) ? flags[doc * P.wpd + w0 + q] : 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (fl[q] == 1) ? pval[base + (long)(w0 + q) * P.nslots_total] : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int w = w0 + q;
                if (w >= P.wpd) break;
                if (fl[q] == kSlabDup) continue;                // padding run behind an identical slab: changes nothing
                if (fl[q]) {
                    if (v[q] > best) { best = v[q]; bw_tile = w; }
                } else if (w * kTile < Lv && 0.f > best) {      // all-masked tile: conv sum is exactly 0 everywhere
                    best = 0.f;
                    bw_tile = -2 - w;
                }
            }
        }
        int bidx = 0;
        if (bw_tile >= 0) bidx = pidx[base + (long)bw_tile * P.nslots_total];
        else if (bw_tile <= -2) bidx = (-2 - bw_tile) * kTile;
        const int bw = P.slot_w[ls];
        const float y = best + bias.p[bw][chan - P.ch_off[bw]];
        const float f = (P.act == RBR_ACT_RELU) ? fmaxf(y, 0.f) : tanhf(y);
        feat[(long)doc * P.C + chan] = f;
        argmax[(long)doc * P.C + chan] = bidx;
    }
}

static int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

template <int NT, int DC, bool VEC>
static int launch_conv_nt(const ConvPlan& p, const long long* ids, const unsigned char* mask, const float* gate,
                          const float* table, const float* packed, float* pval, int* pidx, const int* sched, hipStream_t st) {
    const int XR = kTile + p.KF - 1;
#ifdef RBR_DIAG
    static const size_t extra_lds = getenv("RBR_DEV_CONV_EXTRA_LDS") ? (size_t)atol(getenv("RBR_DEV_CONV_EXTRA_LDS")) : 0;  // tuning aid
#else
    constexpr size_t extra_lds = 0;
#endif
    const size_t smem = (size_t)(4 * kTile * DC + kWavesPerWG * XR * DC) * sizeof(float) +
                        (size_t)kWavesPerWG * (kTile + kMaxKF) * sizeof(long) + 16 + extra_lds;
    static int occ[2] = {0, 0};   // per instantiation; resident workgroups per CU for this LDS/VGPR footprint
    const int sm = p.store_rows ? 1 : 0;
    if (occ[sm] == 0) {
        int nb = 0;
        const hipError_t e = sm ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_store_kernel<NT, DC, VEC>, 256, smem)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_fwd_kernel<NT, DC, VEC>, 256, smem);
        if (e != hipSuccess || nb <= 0) nb = 2;
        occ[sm] = nb;
    }
    const int nitems = ((p.total_wt + kWavesPerWG - 1) / kWavesPerWG) * (p.store_rows ? p.tiles_total / p.ntiles : 1);
    const dim3 grid(std::min(nitems, num_cus() * occ[sm])), block(256);
    if (p.store_rows) {
        if (gate != nullptr) { set_error("store mode takes no gate"); return RBR_ERR_UNSUPPORTED; }
#ifdef RBR_DIAG
        static const bool generic_only = getenv("RBR_DEV_GENERIC_GEMM") != nullptr;      // tuning aid
#else
        constexpr bool generic_only = false;
#endif
        if constexpr (NT == 8 && VEC && DC >= 36 && (DC / 4) % 2 == 1) if (p.KF == 1 && p.n_docs == 1 && !generic_only) {
            const size_t smem2 = (size_t)(4 * kTile * DC + kWavesPerWG * kTile * DC) * sizeof(float) +
                                 (size_t)kWavesPerWG * kTile * sizeof(long) + 16;
            static int occ2 = 0;
            if (occ2 == 0) {
                int nb = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, prod_gemm_kernel<DC>, 256, smem2) != hipSuccess || nb <= 0) nb = 2;
                occ2 = nb;
            }
            hipLaunchKernelGGL((prod_gemm_kernel<DC>), dim3(std::min(nitems, num_cus() * occ2)), block, smem2, st, p, ids, mask, table,
                               packed, pval, sched);
            RBR_CHECK_LAUNCH("textcnn prod_gemm launch");
            return 0;
        }
        hipLaunchKernelGGL((conv_store_kernel<NT, DC, VEC>), grid, block, smem, st, p, ids, mask, table, packed, pval, sched);
    } else {
        hipLaunchKernelGGL((conv_fwd_kernel<NT, DC, VEC>), grid, block, smem, st, p, ids, mask, gate, table, packed, pval,
                           pidx, sched);
    }
    RBR_CHECK_LAUNCH("textcnn conv_fwd launch");
    return 0;
}

template <int DC, bool VEC>
static int launch_conv(const ConvPlan& p, const long long* ids, const unsigned char* mask, const float* gate,
                       const float* table, const float* packed, float* pval, int* pidx, const int* sched, hipStream_t st) {
#define RBR_LAUNCH(NT) \
    case NT: return launch_conv_nt<NT, DC, VEC>(p, ids, mask, gate, table, packed, pval, pidx, sched, st);
    switch (p.ntiles) {
        RBR_LAUNCH(1) RBR_LAUNCH(2) RBR_LAUNCH(3) RBR_LAUNCH(4) RBR_LAUNCH(5) RBR_LAUNCH(6) RBR_LAUNCH(7) RBR_LAUNCH(8)
        default: set_error("ntiles=%d", p.ntiles); return RBR_ERR_UNSUPPORTED;
    }
#undef RBR_LAUNCH
}

static size_t packed_floats(const ConvPlan& p0) {
    return (size_t)p0.tiles_total * p0.KF * p0.nchunks * kTile * p0.DC;
}

}  // namespace rbr

using namespace rbr;

#ifdef RBR_DIAG
extern "C" int rbr_diag_fetch(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_diag), (size_t)n * sizeof(unsigned long long));
}
#endif

extern "C" size_t rbr_textcnn_packed_floats(const rbr_textcnn_desc* d) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return 0;
    return packed_floats(plans[0]);
}

extern "C" size_t rbr_textcnn_partial_elems(const rbr_textcnn_desc* d) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return 0;
    // partial (max, argmax) pairs + the scheduling region flags[total_wt] | list[total_wt] | counters
    return (size_t)plans[0].total_wt * plans[0].nslots_total + 2 * (size_t)plans[0].total_wt + kSchedCounters;
}

extern "C" int rbr_textcnn_pack(const rbr_textcnn_desc* d, const float* const* W, float* packed, void* stream) {
    ConvPlan plans[kMaxGroups];
    const int ng = build_plans(d, plans);
    if (!ng) return RBR_ERR_BAD_ARG;
    if (!W || !packed) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    PtrArray wp{};
    for (int w = 0; w < d->n_widths; ++w) wp.p[w] = W[w];
    for (int g = 0; g < ng; ++g) {
        const long total = (long)plans[g].ntiles * plans[g].KF * plans[g].nchunks * kTile * plans[g].DC;
        const int blocks = (int)std::min<long>((total + 255) / 256, 2048);
        hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, plans[g], wp, packed);
        RBR_CHECK_LAUNCH("textcnn pack launch");
    }
    return 0;
}

extern "C" int rbr_textcnn_conv_fwd(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask,
                                    const float* gate, const float* table, const float* const* W, const float* packed,
                                    float* pval, int32_t* pidx, void* ws, void* stream) {
    ConvPlan plans[kMaxGroups];
    const int ng = build_plans(d, plans);
    if (!ng) return RBR_ERR_BAD_ARG;
    if (!ids || !table || !packed || !pval || !pidx) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const long long* ids64 = reinterpret_cast<const long long*>(ids);
    int* sched = pidx + (size_t)plans[0].total_wt * plans[0].nslots_total;
    if (ws != nullptr && W != nullptr) {
        const int r = run_token_product(d, ids64, mask, gate, table, W, pval, pidx, ws, st);
        if (r != 0) return r == 1 ? 0 : r;
    }
    if (gate != nullptr && plans[0].gate_split > 0) { set_error("rbr_textcnn_conv_fwd: RBR_CONV_GATE_SPLIT needs the token-product formulation"); return RBR_ERR_UNSUPPORTED; }
    if (int e = scan_tiles(plans[0], mask, sched, st)) return e;
    return run_conv_groups(plans, ng, ids64, mask, gate, table, packed, pval, pidx, sched, st);
}

namespace rbr {

__device__ __forceinline__ void zero_regions_kernel(const ZeroRegions& R) {
    const long total = R.n[0] + R.n[1] + R.n[2];
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long)gridDim.x * 256) {
        if (k < R.n[0]) R.p[0][k] = 0;
        else if (k < R.n[0] + R.n[1]) R.p[1][k - R.n[0]] = 0;
        else R.p[2][k - R.n[0] - R.n[1]] = 0;
    }
}

int zero_regions(const ZeroRegions& r, hipStream_t st) {
    const long n = r.n[0] + r.n[1] + r.n[2];
    if (int e_ = rbr::launch<zero_regions_kernel, 256>(dim3((unsigned)std::min<long>((n + 255) / 256, 1024)), dim3(256), 0, st, "zero_regions launch", r)) return e_;
    return 0;
}

int scan_tiles(const ConvPlan& p, const unsigned char* mask, int* sched, hipStream_t st) {
    if (int e = zero_words(sched + 2 * (size_t)p.total_wt, kSchedCounters * sizeof(int), st)) return e;
    hipLaunchKernelGGL(tile_scan_kernel, dim3((p.total_wt + 255) / 256), dim3(256), 0, st, p, mask, sched);
    RBR_CHECK_LAUNCH("textcnn tile_scan launch");
    return 0;
}

int run_conv_groups(const ConvPlan* plans, int ngroups, const long long* ids, const unsigned char* mask, const float* gate,
                    const float* table, const float* packed, float* pval, int* pidx, const int* sched, hipStream_t st) {
    const bool vec = (plans[0].D % 4 == 0) && (((uintptr_t)table & 15) == 0);
    for (int g = 0; g < ngroups; ++g) {
        int e;
        const ConvPlan& p = plans[g];
        if (p.DC != 20 && !vec) { set_error("word table must be 16-byte aligned"); return RBR_ERR_UNSUPPORTED; }
        if (p.DC == 60) {
            e = launch_conv<60, true>(p, ids, mask, gate, table, packed, pval, pidx, sched, st);
        } else if (p.DC == 52) {
            e = launch_conv<52, true>(p, ids, mask, gate, table, packed, pval, pidx, sched, st);
        } else {
            e = vec ? launch_conv<20, true>(p, ids, mask, gate, table, packed, pval, pidx, sched, st)
                    : launch_conv<20, false>(p, ids, mask, gate, table, packed, pval, pidx, sched, st);
        }
        if (e) return e;
    }
    return 0;
}

}  // namespace rbr

extern "C" int rbr_textcnn_pool_finalize(const rbr_textcnn_desc* d, const float* pval, const int32_t* pidx,
                                         const float* const* bias, float* feat, int32_t* argmax, void* stream) {
    ConvPlan plans[kMaxGroups];
    // the widest groups a plan can describe (the slot order and the partials' layout do not depend on the grouping): one launch
    // for up to 256 channel slots instead of one per 160
    const int ng = build_plans(d, plans, kMaxTiles);
    if (!ng) return RBR_ERR_BAD_ARG;
    if (!pval || !pidx || !bias || !feat || !argmax) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    PtrArray bp{};
    for (int w = 0; w < d->n_widths; ++w) bp.p[w] = bias[w];
    for (int g = 0; g < ng; ++g) {
        const long total = (long)plans[g].n_docs * plans[g].ntiles * kTile;
        const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
        if (int e_ = rbr::launch<pool_finalize_kernel, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, "textcnn pool_finalize launch", plans[g], pval, pidx, bp, feat, argmax)) return e_;
    }
    return 0;
}
