// textcnn_prod.hip -- token-product formulation of the TextCNN forward.
//
// The conv input is an embedding LOOKUP: x[doc, p, :] = table[ids[doc, p]].  So for every tap j and channel c
//     out[doc, l, c] = bias[c] + sum_j  T[ ids[doc, l + j - pad] ][j, c],      T[t][j, c] = <table[t, :], W[c, :, j]>
// and T needs one row per DISTINCT token of the batch, not one per position.  Review text is Zipfian: the cfg2
// batch has 262 k positions but only ~28 k distinct tokens, so the contraction shrinks from 118 GFLOP (dense conv,
// what the reference executes: models/deepconn/layers.py:46-60) to ~13 GFLOP, followed by a gather-add over
// kz rows of T per position that is L2/HBM-bound row traffic.  Same fp32 arithmetic, different summation order
// (d first, taps second); masked tokens and out-of-document taps contribute exactly 0, as masked_fill and the
// conv's zero padding do.
//
// Stages (all on the caller's stream, nothing allocated here):
//   1. mark_tokens / compact_tokens : distinct unmasked tokens -> list `tok_of_row`, inverse map `row_of_token`
//   2. pack_prod                    : W[c, d, j] of every bank -> one kz = 1 bank of sum(kz*ch) "product channels"
//   3. conv_fwd_kernel (store_rows) : T = table[tok_of_row] @ Wprod on the f32 MFMA pipe (the SAME fused gather +
//                                     MFMA kernel as the dense path, run over the token list as one long document)
//   4. gather_pool                  : per 32-position wave-tile, sum the kz rows of T, running max / first argmax
//   then pool_finalize as for the dense path.
#include "rbr_common.h"

#include <algorithm>
#include <cstdlib>

namespace rbr {

struct ProdArgs {
    int n_docs, L, V, cap;          // cap = rows of T that may be used (distinct tokens <= min(V, positions))
    int zrow;                       // index of the all-zero row of T (= cap)
    int pitch;                      // floats per row of T (32 * product tiles)
    int poff[RBR_MAX_WIDTHS];       // first product channel of bank w: channel (w, j, cl) = poff[w] + j*ch[w] + cl
    int cp_real;                    // product channels before padding to whole launch groups
};

// ---------------------------------------------------------------------------------- distinct tokens
__global__ __launch_bounds__(256) void mark_tokens_kernel(long n_tok, const long long* __restrict__ ids,
                                                          const unsigned char* __restrict__ mask, int* __restrict__ used) {
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n_tok; k += (long)gridDim.x * 256)
        if (mask == nullptr || mask[k]) used[ids[k]] = 1;      // benign race: everyone stores 1
}

// row_of_token[v] = dense row index (any order) or -1; tok_of_row / row_mask describe the pseudo-document
__global__ __launch_bounds__(256) void compact_tokens_kernel(int V, int cap, const int* __restrict__ used,
                                                             int* __restrict__ row_of_token, long long* __restrict__ tok_of_row,
                                                             unsigned char* __restrict__ row_mask, int* __restrict__ counter,
                                                             float* __restrict__ zero_row, int pitch) {
    if (blockIdx.x == 0)      // the all-zero row of T that masked / out-of-document taps read
        for (int k = threadIdx.x; k < pitch; k += 256) zero_row[k] = 0.f;
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int u = (v < V) ? used[v] : 0;
    const unsigned long long b = __ballot(u);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && b) base = atomicAdd(counter, __popcll(b));
    base = __shfl(base, 0);
    if (v < V) {
        int row = -1;
        if (u) {
            row = base + __popcll(b & ((1ull << lane) - 1));
            if (row < cap) { tok_of_row[row] = v; row_mask[row] = 1; } else row = -1;   // cannot happen: cap >= distinct
        }
        row_of_token[v] = row;
    }
}

// ---------------------------------------------------------------------------------- product weights
// packed layout of the conv kernel for ONE bank of Cp channels with kz = 1:  [dc][tile][slot][dd]
__global__ __launch_bounds__(256) void pack_prod_kernel(const ConvPlan P, const ConvPlan D0, const ProdArgs A, const PtrArray W,
                                                        float* __restrict__ packed) {
    const int DC = P.DC;
    const long total = (long)P.nchunks * P.tiles_total * kTile * DC;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int dd = (int)(r % DC); r /= DC;
        const int slot = (int)(r % kTile); r /= kTile;
        const int t = (int)(r % P.tiles_total);
        const int dc = (int)(r / P.tiles_total);
        const int pc = t * kTile + slot;          // product channel
        const int d = dc * DC + dd;
        float v = 0.f;
        if (pc < A.cp_real && d < P.D) {
            int w = 0;
#pragma unroll
            for (int k = 1; k < RBR_MAX_WIDTHS; ++k)
                if (k < D0.n_widths && pc >= A.poff[k]) w = k;
            const int rel = pc - A.poff[w];
            const int j = rel / D0.ch[w], cl = rel - j * D0.ch[w];
            v = W.p[w][((long)cl * P.D + d) * D0.kz[w] + j];
        }
        packed[idx] = v;
    }
}

// ---------------------------------------------------------------------------------- gather + pool
// One wave per active 32-position wave-tile (work list of the dense path's tile scan).  Lanes = channel slots
// (64 per pass); for each position the lane adds its kz product rows.  Writes the same (max, first argmax)
// partials as the conv kernel's pooling epilogue.
__global__ __launch_bounds__(256) void gather_pool_kernel(const ConvPlan P, const ProdArgs A, const long long* __restrict__ ids,
                                                          const unsigned char* __restrict__ mask, const float* __restrict__ gate,
                                                          const int* __restrict__ row_of_token, const float* __restrict__ T,
                                                          const int* __restrict__ sched, float* __restrict__ pval,
                                                          int* __restrict__ pidx) {
    __shared__ int s_row[kWavesPerWG][kTile + kMaxKF];
    __shared__ float s_gate[kWavesPerWG][kTile + kMaxKF];
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
        float gv = 1.f;
        if (p >= 0 && p < L) {
            const long tok = (long)doc * L + p;
            if (mask == nullptr || mask[tok]) {
                const int rr = row_of_token[ids[tok]];
                row = rr >= 0 ? rr : A.zrow;
                if (gate != nullptr) gv = gate[tok];
            }
        }
        s_row[wave][r] = row;
        s_gate[wave][r] = gv;
    }
    __builtin_amdgcn_wave_barrier();
    const int nslots = P.ntiles * kTile;            // slots of this launch group
    for (int ls0 = 0; ls0 < nslots; ls0 += 64) {
        const int ls = ls0 + lane;
        // slot descriptors of ALL groups live in plan order; this kernel is launched with the group-0 plan of a
        // single-group problem or once per group (tile_base selects the slot range)
        const int chan = (ls < nslots) ? P.slot_chan[ls] : -1;
        int kz = 0, off = 0, col = 0, cstride = 0;
        if (chan >= 0) {
            const int w = P.slot_w[ls];
            kz = P.slot_kz[ls];
            off = P.slot_off[ls];
            cstride = P.ch[w];
            col = A.poff[w] + (chan - P.ch_off[w]);
        }
        const int Lv = (P.pad_mode == RBR_PAD_VALID) ? (L - kz + 1) : L;
        float best = -__builtin_huge_valf();
        int bidx = 0x7fffffff;
        for (int q0 = 0; q0 < kTile; q0 += 4) {          // 4 positions x kz rows of independent loads in flight
            float y[4] = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < P.KF; ++j) {
                if (j < kz) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = q0 + u + j + off;
                        const float v = T[(long)s_row[wave][r] * A.pitch + col + j * cstride];
                        y[u] = fmaf(v, s_gate[wave][r], y[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pos = l0 + q0 + u;
                if (chan >= 0 && pos < Lv && y[u] > best) { best = y[u]; bidx = pos; }
            }
        }
        if (chan >= 0) {
            const long o = (long)wt * P.nslots_total + (long)P.tile_base * kTile + ls;
            pval[o] = best;
            pidx[o] = bidx;
        }
    }
}

}  // namespace rbr

using namespace rbr;

namespace {

struct ProdLayout {      // byte offsets inside the workspace
    size_t used, row_of_token, tok_of_row, row_mask, counter, sched, packed, table_T, total;
    int cap, Cp, tiles_p;
    rbr_textcnn_desc dp;
};

int g_conv_mode = -1;     // -1: unset -> RBR_CONV_MODE env ("dense" | "product") or auto; 0 auto, 1 dense, 2 product

bool prod_applicable(const rbr_textcnn_desc* d) {
    static const char* env = getenv("RBR_CONV_MODE");
    const char* mode = g_conv_mode == 1 ? "dense" : g_conv_mode == 2 ? "product" : g_conv_mode == 0 ? nullptr : env;
    if (mode && !strcmp(mode, "dense")) return false;
    const long n_pos = (long)d->n_docs * d->L;
    long Cp = 0;
    for (int w = 0; w < d->n_widths; ++w) Cp += (long)d->kz[w] * d->ch[w];
    if (Cp > 30000) return false;                          // slot ids are 16-bit
    if ((Cp + kTile - 1) / kTile > (long)kMaxGroups * 5) return false;
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
    constexpr int kGroupSlots = 5 * kTile;            // one launch group = 5 tiles (rbr_plan.hip)
    Cp = ((Cp + kGroupSlots - 1) / kGroupSlots) * kGroupSlots;   // zero-weight padding channels: all groups identical
    Lo.Cp = (int)Cp;
    Lo.tiles_p = (int)(Cp / kTile);
    memset(&Lo.dp, 0, sizeof(Lo.dp));
    Lo.dp.n_docs = 1; Lo.dp.L = Lo.cap; Lo.dp.D = d->D; Lo.dp.V = d->V;
    Lo.dp.n_widths = 1; Lo.dp.kz[0] = 1; Lo.dp.ch[0] = Lo.Cp;
    Lo.dp.pad_mode = RBR_PAD_SAME; Lo.dp.act = RBR_ACT_RELU; Lo.dp.padding_idx = -1;
    ConvPlan plans[kMaxGroups];
    if (!build_plans(&Lo.dp, plans)) return false;
    const ConvPlan& p = plans[0];
    size_t o = 0;
    // [used | row_mask | counter] are contiguous: ONE memset node re-initialises them every call
    Lo.used = o;         o += align256((size_t)d->V * sizeof(int));
    Lo.row_mask = o;     o += align256((size_t)Lo.cap);
    Lo.counter = o;      o += 256;
    Lo.row_of_token = o; o += align256((size_t)d->V * sizeof(int));
    Lo.tok_of_row = o;   o += align256((size_t)Lo.cap * sizeof(long long));
    Lo.sched = o;        o += align256((2 * (size_t)p.total_wt + kSchedCounters) * sizeof(int));
    Lo.packed = o;       o += align256((size_t)p.nchunks * p.tiles_total * kTile * p.DC * sizeof(float));
    Lo.table_T = o;      o += align256(((size_t)Lo.cap + 1) * p.nslots_total * sizeof(float));
    Lo.total = o;
    return true;
}

}  // namespace

extern "C" void rbr_set_conv_mode(int32_t mode) { g_conv_mode = (mode >= 0 && mode <= 2) ? mode : -1; }

extern "C" size_t rbr_textcnn_fwd_ws_bytes(const rbr_textcnn_desc* d) {
    ConvPlan plans[kMaxGroups];
    if (!build_plans(d, plans)) return 0;
    if (!prod_applicable(d)) return 0;
    ProdLayout Lo;
    if (!prod_layout(d, Lo)) return 0;
    return Lo.total;
}

namespace rbr {

// returns 1 when the product path ran, 0 when the caller must run the dense conv, < 0 / hip error on failure
int run_token_product(const rbr_textcnn_desc* d, const ConvPlan* plans, int ngroups, const long long* ids,
                      const unsigned char* mask, const float* gate, const float* table, const float* const* W, float* pval,
                      int* pidx, const int* sched, void* ws, hipStream_t st) {
    if (ws == nullptr || !prod_applicable(d)) return 0;
    ProdLayout Lo;
    if (!prod_layout(d, Lo)) return RBR_ERR_BAD_ARG;
    char* base = static_cast<char*>(ws);
    int* used = reinterpret_cast<int*>(base + Lo.used);
    int* row_of_token = reinterpret_cast<int*>(base + Lo.row_of_token);
    long long* tok_of_row = reinterpret_cast<long long*>(base + Lo.tok_of_row);
    unsigned char* row_mask = reinterpret_cast<unsigned char*>(base + Lo.row_mask);
    int* counter = reinterpret_cast<int*>(base + Lo.counter);
    int* sched_p = reinterpret_cast<int*>(base + Lo.sched);
    float* packed_p = reinterpret_cast<float*>(base + Lo.packed);
    float* T = reinterpret_cast<float*>(base + Lo.table_T);

    ConvPlan pp[kMaxGroups];
    const int ngp = build_plans(&Lo.dp, pp);
    if (!ngp) return RBR_ERR_BAD_ARG;
    pp[0].store_rows = 1;      // group 0's plan describes every group; the kernel folds them into one launch
    (void)ngp;

    ProdArgs A{};
    A.n_docs = d->n_docs; A.L = d->L; A.V = d->V; A.cap = Lo.cap; A.zrow = Lo.cap; A.pitch = pp[0].nslots_total;
    {   // product channels follow the plan's bank order of the ORIGINAL problem (bank w, tap j, channel cl)
        int o = 0;
        for (int w = 0; w < d->n_widths; ++w) { A.poff[w] = o; o += d->kz[w] * d->ch[w]; }
        A.cp_real = o;
    }
    // 1. distinct tokens  (used / row_mask / counter / zero row are re-initialised every call: graph-replay safe)
    if (int e = check_hip(hipMemsetAsync(base + Lo.used, 0, Lo.row_of_token - Lo.used, st), "token-list state memset")) return e;
    const long n_tok = (long)d->n_docs * d->L;
    hipLaunchKernelGGL(mark_tokens_kernel, dim3((unsigned)std::min<long>((n_tok + 255) / 256, 2048)), dim3(256), 0, st, n_tok,
                       ids, mask, used);
    RBR_CHECK_LAUNCH("textcnn mark_tokens launch");
    hipLaunchKernelGGL(compact_tokens_kernel, dim3((d->V + 255) / 256), dim3(256), 0, st, d->V, Lo.cap, used, row_of_token,
                       tok_of_row, row_mask, counter, T + (size_t)Lo.cap * A.pitch, A.pitch);
    RBR_CHECK_LAUNCH("textcnn compact_tokens launch");
    // 2. product weights
    PtrArray wp{};
    for (int w = 0; w < d->n_widths; ++w) wp.p[w] = W[w];
    {
        const long total = (long)pp[0].nchunks * pp[0].tiles_total * kTile * pp[0].DC;
        hipLaunchKernelGGL(pack_prod_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, st, pp[0],
                           plans[0], A, wp, packed_p);
        RBR_CHECK_LAUNCH("textcnn pack_prod launch");
    }
    // 3. T = table[tok_of_row] @ Wprod  (rows beyond the distinct count are masked -> their tiles are skipped)
    if (int e = scan_tiles(pp[0], row_mask, sched_p, st)) return e;
    if (int e = run_conv_groups(pp, 1, tok_of_row, row_mask, nullptr, table, packed_p, T, nullptr, sched_p, st)) return e;
    // 4. gather + pool per active wave-tile of the real documents
    const int max_items = (plans[0].total_wt + kWavesPerWG - 1) / kWavesPerWG;
    for (int g = 0; g < ngroups; ++g) {
        hipLaunchKernelGGL(gather_pool_kernel, dim3(max_items), dim3(256), 0, st, plans[g], A, ids, mask, gate, row_of_token, T,
                           sched, pval, pidx);
        RBR_CHECK_LAUNCH("textcnn gather_pool launch");
    }
    return 1;
}

}  // namespace rbr
