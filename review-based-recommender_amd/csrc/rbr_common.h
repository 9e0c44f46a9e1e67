// rbr_common.h -- shared host/device definitions of librbr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/rbr_hip.h"
#include "rbr_launch.h"

namespace rbr {

constexpr int kWave = 64;        // CDNA4 wavefront
constexpr int kTile = 32;        // MFMA 32x32x2 f32: 32 token positions x 32 channel slots
constexpr int kMaxTiles = 8;     // channel tiles per launch (8 x 16 accumulator VGPRs)
constexpr int kMaxGroups = 32;   // launches per conv (<= 5 tiles = 160 channel slots each)
constexpr int kMaxSlots = kMaxTiles * kTile;
constexpr int kMaxKF = 9;        // widest conv window
constexpr int kMaxDim = 1 << 14;      // embedding dimension a descriptor may name
constexpr int kMaxChannels = kMaxGroups * kMaxTiles * kTile;      // output channels of one conv call (8192)
constexpr int kMaxPieces = kMaxTiles * kMaxKF;
constexpr int kWavesPerWG = 4;
constexpr int kSchedCounters = 64;   // ints behind the work list: [0] active tiles, [1 + g] item counter of group g

// One launch of the conv kernel: <= kMaxTiles channel tiles over all wave-tiles of the batch.
// Passed BY VALUE as a kernel argument (lives in SGPRs / the scalar cache).
struct ConvPlan {
    int n_docs, L, D, V;
    int C;              // output channels of the whole conv (all groups)
    int KF;             // frame taps: max kz
    int P;              // frame left pad: row = l + s - P
    int DC;             // embedding-dim chunk staged per piece
    int nchunks;        // ceil(D / DC)
    int wpd;            // wave-tiles (32-token slabs) per document
    int total_wt;       // n_docs * wpd
    int nslots_total;   // 32 * (tiles over all groups): row pitch of the partial-max workspace
    int tile_base;      // first tile of this group in the packed image / workspace
    int group;          // launch index of this plan (selects its work counter)
    int store_rows;     // 1: write the [32 x 32] accumulator tiles to out[row, slot] (token-product table) instead of pooling
    int ntiles;         // tiles in this group
    int npieces;        // (tap, tile) pairs streamed per embedding chunk
    int pad_mode, act;
    int tiles_total;    // channel tiles over all groups (tile pitch of the packed image)
    int piece_st[kMaxPieces];          // tap | first tile << 8 | tile count (1..2) << 16.  Dword entries: fetched with
                                       // s_load, not a vector load whose vmcnt wait would drain the weight prefetch
    short slot_chan[kMaxSlots];        // global output channel of slot, -1 = padding slot
    unsigned char slot_w[kMaxSlots];   // conv bank of the slot
    unsigned char slot_off[kMaxSlots]; // frame tap of the channel's tap 0
    unsigned char slot_kz[kMaxSlots];  // kernel width (0 = padding slot)
    int n_widths;
    int kz[RBR_MAX_WIDTHS], ch[RBR_MAX_WIDTHS], ch_off[RBR_MAX_WIDTHS];
    int gate_split;     // RBR_CONV_GATE_SPLIT: first bank of the second gate plane (0: one gate for every bank)
    int pad_runs;       // RBR_CONV_PAD_RUNS in force (un-masked, valid-padded or width-1 conv): token id of the padding, else -1
};

struct PtrArray {
    const float* p[RBR_MAX_WIDTHS];
};
struct MutPtrArray {
    float* p[RBR_MAX_WIDTHS];
};

// g = d_feat * act'(feat): gradient entering the conv output at the max-pool's argmax
__device__ __forceinline__ float act_grad(int act, float f, float d) {
    return (act == RBR_ACT_RELU) ? (f > 0.f ? d : 0.f) : d * (1.f - f * f);
}

void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// Builds the launch plans for `d`.  Returns the number of groups (0 on error, see rbr_last_error()).
bool desc_valid(const rbr_textcnn_desc* d);      // shape / range validation of a descriptor (sets the error text)
int build_plans(const rbr_textcnn_desc* d, ConvPlan* plans /* [kMaxGroups] */, int tiles_per_group = 5);
constexpr int kProdGroupTiles = 8;   // channel tiles per work item of the token-product GEMM (textcnn_prod.hip)

// Launches the fused gather + conv kernel for every group of `plans` (textcnn_fwd.hip).  `sched` must have been
// filled by scan_tiles() for the same document set.  In store_rows mode `pval` is the output table.
int run_conv_groups(const ConvPlan* plans, int ngroups, const long long* ids, const unsigned char* mask, const float* gate,
                    const float* table, const float* packed, float* pval, int* pidx, const int* sched, hipStream_t st);
// Token-product forward (textcnn_prod.hip): returns 1 when it produced pval/pidx, 0 when the dense conv must run.
int run_token_product(const rbr_textcnn_desc* d, const long long* ids, const unsigned char* mask, const float* gate,
                      const float* table, const float* const* W, float* pval, int* pidx, void* ws, hipStream_t st);
// 0 auto, 1 dense forced, 2 token-product forced (rbr_set_conv_mode / RBR_CONV_MODE)
int forced_conv_mode();
// bf16-plane form of the token-product GEMM (textcnn_prod_b16.hip): RBR_PROD_* in force, whether it serves `d`, the
// bytes of its weight-plane image, the pack launch and the GEMM launch (T pitch must be >= 128 * prod_b16_groups)
int prod_precision();
int prod_precision_of(const rbr_textcnn_desc* d);      // the descriptor's stamped class, else the process-wide setting
bool prod_b16_applicable(const rbr_textcnn_desc* d);
int prod_b16_groups(int cp_real);
size_t prod_b16_image_bytes(const rbr_textcnn_desc* d, int cp_real);
struct B16Pack;
B16Pack prod_b16_pack_job(const rbr_textcnn_desc* d);     // textcnn_b16.h
int prod_b16_gemm(const rbr_textcnn_desc* d, int cp_real, int cap, int pitch, const int* counter, const long long* tok_of_row,
                  const float* table, const void* bimg, void* T, void* a16, hipStream_t st);
// The backward's table-gradient rows as a GEMM on the same kernel (textcnn_prod_b16.hip): rows [cap, D] = G [cap, KG] @ Wprod^T, with
// the per-workgroup sums of squares the row-form optimizer wants; applicable: whether the shape takes it (many short documents: G
// is dense enough that the sparse product loses), image bytes / work items of its weight planes, partial count, launch
bool prod_b16_rows_applicable(const rbr_textcnn_desc* d);
size_t prod_b16_rows_image_bytes(int KG, int D);
long prod_b16_rows_image_items(int KG, int D);
int prod_b16_rows_partials(int cap, int D);
int prod_b16_rows_gemm(int KG, int D, int cap, const int* counter, const float* G, const void* bimg_t, float* rows, float* sq_part, bool plain_bf16,
                       hipStream_t st);
// bf16 STORAGE of the plain-bf16 class (textcnn_prod_b16.hip): T holds bf16, and `a16` (prod_b16_rows_bytes) the compact bf16 rows
bool prod_t_bf16(const rbr_textcnn_desc* d);
size_t prod_b16_rows_bytes(const rbr_textcnn_desc* d, int cap);
// Zero-fills up to three regions (4-byte aligned, sizes multiples of 4) with ONE kernel launch.  A kernel, not
// hipMemsetAsync: the launch sequence is recorded into hipGraphs (train_step.GraphedTrainStep) and stays a chain of
// plain kernel nodes (memset nodes of these shapes faulted on replay with ROCm 7.2).
struct ZeroRegions {
    int* p[3];
    long n[3];      // words
};
int zero_regions(const ZeroRegions& r, hipStream_t st);
inline int zero_words(void* p, size_t bytes, hipStream_t st) {
    ZeroRegions r{{static_cast<int*>(p), nullptr, nullptr}, {(long)(bytes / sizeof(int)), 0, 0}};
    return zero_regions(r, st);
}

// Id range check (dense_misc.hip: rbr_sanitize_ids; textcnn_prod.hip: the same job inside the prepare stage's first launch).
struct IdSets {
    const long long* in[RBR_MAX_ID_SETS];
    long long* out[RBR_MAX_ID_SETS];
    long long n[RBR_MAX_ID_SETS], limit[RBR_MAX_ID_SETS], replace[RBR_MAX_ID_SETS];
    long long first[RBR_MAX_ID_SETS + 1];     // prefix of n: element k of the launch belongs to the set with first[s] <= k < first[s+1]
    int count;
};
// host: rbr_id_set[] -> IdSets (validated); returns 0 or RBR_ERR_BAD_ARG
int fill_id_sets(int n_sets, const rbr_id_set* sets, IdSets& S);
// element k of the launch: out = in where 0 <= in < limit, else `replace` + a record in err (err[0] count, err[1] value, err[2] set)
__device__ __forceinline__ void sanitize_id(const IdSets& S, long long k, long long* __restrict__ err) {
    int s = 0;
#pragma unroll
    for (int q = 1; q < RBR_MAX_ID_SETS; ++q)
        if (q < S.count && k >= S.first[q]) s = q;
    const long long e = k - S.first[s];
    long long v = S.in[s][e];
    const bool bad = (unsigned long long)v >= (unsigned long long)S.limit[s];
    if (bad) {
        err[1] = v; err[2] = s;                       // any one offender (benign race)
        atomicAdd(reinterpret_cast<unsigned long long*>(err), 1ull);
        v = S.replace[s];
    }
    S.out[s][e] = v;
}

// Work-list scan of 256 wave-tiles (block `blk` of the scan grid): flags[wt], and the active tiles appended to the
// list in any order.  sched = flags[total_wt] | list[total_wt] | counters; the counters must be zero beforehand.
// flags: 1 = computed; 0 = all tokens masked (its conv sums are exactly 0: the finalize kernels count it as that);
// kSlabDup = un-masked slab of pure padding behind another one (P.pad_runs): position for position the values of its
// predecessor, so it can change neither a maximum nor a FIRST argmax -- neither computed nor looked at.
constexpr int kSlabDup = 2;
constexpr int kPadRunHalo = 8;
__device__ __forceinline__ void tile_scan_block(const ConvPlan& P, const unsigned char* __restrict__ mask,
                                                int* __restrict__ sched, int blk, const long long* __restrict__ ids = nullptr) {
    const int wt = blk * 256 + threadIdx.x;
    int act = 0;
    if (wt < P.total_wt) {
        act = 1;
        if (mask != nullptr) {
            const int doc = wt / P.wpd, l0 = (wt % P.wpd) * kTile;
            const int lo = max(0, l0 - P.P), hi = min(P.L, l0 + kTile + P.KF - 1 - P.P);
            act = 0;
            const unsigned char* row = mask + (long)doc * P.L;
            if (l0 + kTile <= P.L && ((((uintptr_t)(row + l0)) & 15) == 0)) {
                // the 32 tokens of the slab as two 16-byte loads, the halo bytes beside them: one round of independent loads
                const uint4 a = *reinterpret_cast<const uint4*>(row + l0), c = *reinterpret_cast<const uint4*>(row + l0 + 16);
                unsigned halo = 0;
                for (int p = lo; p < l0; ++p) halo |= row[p];
                for (int p = l0 + kTile; p < hi; ++p) halo |= row[p];
                act = (a.x | a.y | a.z | a.w | c.x | c.y | c.z | c.w | halo) != 0;
            } else {
                for (int p = lo; p < hi; ++p) act |= mask[(long)doc * P.L + p];
            }
            act = act ? 1 : 0;
        }
        sched[wt] = act;
    }
    if (mask == nullptr && ids != nullptr && P.pad_runs >= 0) {      // workgroup-uniform
        // own(w): slab w, frame and gate halo included, lies inside its document and is all padding (16-byte loads, 8 per round);
        // slab w is a duplicate when own(w) and own(w - 1) -- the predecessor's answer comes from the neighbouring thread
        __shared__ unsigned char s_own[256];
        bool own = false;
        if (wt < P.total_wt) {
            const int doc = wt / P.wpd, l0 = (wt % P.wpd) * kTile;
            const int lo = l0 - P.P - kPadRunHalo, hi = l0 + kTile + P.KF - 1 - P.P + kPadRunHalo;
            if (lo >= 0 && hi <= P.L) {
                const long long* row = ids + (long)doc * P.L;
                const long long pad = (long long)P.pad_runs;
                own = true;
                int p = lo;
                if ((((uintptr_t)(row + p)) & 15) == 0) {
                    for (; p + 16 <= hi && own; p += 16) {
                        longlong2 t[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const longlong2*>(row + p + 2 * u);
#pragma unroll
                        for (int u = 0; u < 8; ++u) own = own && t[u].x == pad && t[u].y == pad;
                    }
                }
                for (; p < hi && own; ++p) own = row[p] == pad;
            }
        }
        s_own[threadIdx.x] = own ? 1 : 0;
        __syncthreads();
        if (wt < P.total_wt && own && (wt % P.wpd) != 0) {
            bool prev;
            if (threadIdx.x > 0) {
                prev = s_own[threadIdx.x - 1] != 0;
            } else {                                             // first thread of the block: its predecessor belongs to another block
                const int doc = wt / P.wpd, l0 = (wt % P.wpd - 1) * kTile;
                const int lo = l0 - P.P - kPadRunHalo, hi = l0 + kTile + P.KF - 1 - P.P + kPadRunHalo;
                prev = lo >= 0 && hi <= P.L;
                for (int q = lo; q < hi && prev; ++q) prev = ids[(long)doc * P.L + q] == (long long)P.pad_runs;
            }
            if (prev) { act = kSlabDup; sched[wt] = act; }
        }
    }
    const unsigned long long b = __ballot(act == 1);
    act = act == 1;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && b) base = atomicAdd(sched + 2 * (long)P.total_wt, __popcll(b));
    base = __shfl(base, 0);
    if (act) sched[(long)P.total_wt + base + __popcll(b & ((1ull << lane) - 1))] = wt;
}
// Zeroes the counters of `sched` and builds flags | work list | counter for the documents of `p`.
int scan_tiles(const ConvPlan& p, const unsigned char* mask, int* sched, hipStream_t st);

// Embedding-dim chunk per weight piece: 60, 52 or 20 floats (all 4*odd: conflict-free b128 reads).  The largest
// chunk whose zero padding of D stays within 8 % of the best candidate wins (10-MFMA pieces of the 20-float chunk
// spend a third of their time in barriers).  D % 4 != 0 uses the scalar-gather instantiation (20 only).
inline int choose_dc(int D) {
    if (D % 4 != 0) return 20;
    const int cands[3] = {60, 52, 20};
    int best_pad = 1 << 30;
    for (int c : cands) best_pad = std::min(best_pad, ((D + c - 1) / c) * c);
    for (int c : cands)
        if (((D + c - 1) / c) * c * 100 <= best_pad * 108) return c;
    return 20;
}


// ---- Philox4x32-10 for the dropout draws (dense_misc.hip: rbr_dropout_multiplier; pair_head.hip: the fused head forward).
// Element i of call number `call` under `seed` is word (i & 3) of the block with counter (i >> 2, call).
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ void philox4x32_10(unsigned long long quad, unsigned long long call, unsigned long long seed,
                                              unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3) {
    c0 = (unsigned)quad; c1 = (unsigned)(quad >> 32); c2 = (unsigned)call; c3 = (unsigned)(call >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ bool dropout_keep(unsigned word, float p) {      // u in [0,1) from the top 24 bits; keep iff u >= p
    return (float)(word >> 8) * (1.f / 16777216.f) >= p;
}

}  // namespace rbr

// after a DIRECT launch (hipLaunchKernelGGL): inside a pair region (rbr_launch.h) such a launch would have run ahead of the
// recorded ones, so the entry point is refused there
#define RBR_CHECK_LAUNCH(what)                                   \
    do {                                                         \
        int _e = rbr::check_hip(hipGetLastError(), what);        \
        if (_e) return _e;                                       \
        if (rbr::pair_region_open()) {                           \
            rbr::set_error("%s: this entry point launches directly and cannot be used inside rbr_pair_begin / rbr_pair_end", what); \
            return RBR_ERR_UNSUPPORTED;                          \
        }                                                        \
    } while (0)
