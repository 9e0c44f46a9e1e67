// rbr_plan.hip -- error plumbing and the channel-tile planner shared by forward and backward.
#include "rbr_common.h"

#include <algorithm>
#include <vector>

namespace rbr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

// Every entry point that takes a descriptor goes through this before it reads d->kz / d->ch or multiplies shapes: the limits
// keep all index arithmetic of plans, layouts and grids inside int / long (tests/asan_fuzz_host.py throws hostile descriptors at
// a sanitizer build of this code).
bool desc_valid(const rbr_textcnn_desc* d) {
    if (!d) { set_error("null descriptor"); return false; }
    if (d->n_docs <= 0 || d->L <= 0 || d->D <= 0 || d->V <= 0) {
        set_error("bad shape n_docs=%d L=%d D=%d V=%d", d->n_docs, d->L, d->D, d->V);
        return false;
    }
    if (d->D > kMaxDim || (long)d->n_docs * d->L >= (1L << 31)) {
        set_error("shape out of range: D=%d (max %d), n_docs * L = %ld (max 2^31 - 1)", d->D, kMaxDim, (long)d->n_docs * d->L);
        return false;
    }
    if (d->n_widths <= 0 || d->n_widths > RBR_MAX_WIDTHS) { set_error("n_widths=%d out of range", d->n_widths); return false; }
    long c_total = 0;
    for (int w = 0; w < d->n_widths; ++w) {
        if (d->kz[w] <= 0 || d->kz[w] > kMaxKF) { set_error("kernel width %d unsupported (1..%d)", d->kz[w], kMaxKF); return false; }
        if (d->ch[w] <= 0 || d->ch[w] > kMaxChannels) { set_error("bank %d has %d channels (1..%d)", w, d->ch[w], kMaxChannels); return false; }
        c_total += d->ch[w];
        // the reference asserts odd widths for 'same' convs (deepconn/layers.py:39)
        if (d->pad_mode == RBR_PAD_SAME && d->kz[w] % 2 == 0) { set_error("'same' conv needs odd kernel width, got %d", d->kz[w]); return false; }
        if (d->pad_mode == RBR_PAD_VALID && d->kz[w] > d->L) { set_error("valid conv wider than the document"); return false; }
    }
    if (c_total > kMaxChannels) { set_error("%ld output channels (max %d)", c_total, kMaxChannels); return false; }
    if (d->pad_mode != RBR_PAD_SAME && d->pad_mode != RBR_PAD_VALID) { set_error("pad_mode %d", d->pad_mode); return false; }
    if (d->act != RBR_ACT_RELU && d->act != RBR_ACT_TANH) { set_error("act %d", d->act); return false; }
    if (d->padding_idx >= d->V) { set_error("padding_idx %d outside a table of %d rows", d->padding_idx, d->V); return false; }
    if (RBR_CONV_GATE_SPLIT_OF(d->flags) >= d->n_widths) {
        set_error("RBR_CONV_GATE_SPLIT(%d) with %d banks", RBR_CONV_GATE_SPLIT_OF(d->flags), d->n_widths);
        return false;
    }
    return true;
}

// Channel slots are ordered by ascending kernel width and cut into tiles of 32: a tile then only
// streams the taps its widest member needs (3/5/7 x 50 channels -> 27 tap-tiles instead of 30).
int build_plans(const rbr_textcnn_desc* d, ConvPlan* plans, int tiles_per_group) {
    if (!desc_valid(d)) return 0;
    int C = 0, KF = 0;
    int ch_off[RBR_MAX_WIDTHS];
    for (int w = 0; w < d->n_widths; ++w) { ch_off[w] = C; C += d->ch[w]; KF = std::max(KF, d->kz[w]); }
    const int tiles_total = (C + kTile - 1) / kTile;
    // <= 5 tiles per launch keeps the pooling kernel at 2 waves/SIMD without asking the compiler for it; the
    // store-mode kernel is compiled for 2 waves/SIMD and takes 8.  Groups are balanced (10 tiles -> 5 + 5, not 8 + 2)
    const int kPreferredTiles = std::min(std::max(tiles_per_group, 1), kMaxTiles);
    const int ngroups = (tiles_total + kPreferredTiles - 1) / kPreferredTiles;
    const int per_group = (tiles_total + ngroups - 1) / ngroups;
    if (ngroups > kMaxGroups) { set_error("%d output channels exceed the supported %d", C, kMaxGroups * kPreferredTiles * kTile); return 0; }

    std::vector<int> order(d->n_widths);
    for (int w = 0; w < d->n_widths; ++w) order[w] = w;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d->kz[a] < d->kz[b]; });
    struct Slot { int chan, w; };
    std::vector<Slot> slots;
    for (int w : order)
        for (int c = 0; c < d->ch[w]; ++c) slots.push_back({ch_off[w] + c, w});

    const int DC = choose_dc(d->D);
    const int P = (d->pad_mode == RBR_PAD_SAME) ? (KF - 1) / 2 : 0;
    for (int g = 0; g < ngroups; ++g) {
        ConvPlan& p = plans[g];
        memset(&p, 0, sizeof(p));
        p.n_docs = d->n_docs; p.L = d->L; p.D = d->D; p.V = d->V;
        p.C = C; p.KF = KF; p.P = P; p.DC = DC; p.nchunks = (d->D + DC - 1) / DC;
        p.wpd = (d->L + kTile - 1) / kTile;
        p.total_wt = d->n_docs * p.wpd;
        p.nslots_total = tiles_total * kTile;
        p.tile_base = g * per_group;
        p.group = g;
        p.ntiles = std::min(per_group, tiles_total - p.tile_base);
        p.pad_mode = d->pad_mode; p.act = d->act;
        p.gate_split = RBR_CONV_GATE_SPLIT_OF(d->flags);
        p.pad_runs = ((d->flags & RBR_CONV_PAD_RUNS) && d->padding_idx >= 0 && (d->pad_mode == RBR_PAD_VALID || KF == 1)) ? d->padding_idx : -1;
        p.n_widths = d->n_widths;
        for (int w = 0; w < d->n_widths; ++w) { p.kz[w] = d->kz[w]; p.ch[w] = d->ch[w]; p.ch_off[w] = ch_off[w]; }
        int lo[kMaxTiles], hi[kMaxTiles];
        for (int t = 0; t < p.ntiles; ++t) {
            lo[t] = KF; hi[t] = -1;
            for (int i = 0; i < kTile; ++i) {
                const int gs = (p.tile_base + t) * kTile + i;
                const int ls = t * kTile + i;
                if (gs < (int)slots.size()) {
                    const int w = slots[gs].w, kz = d->kz[w];
                    const int off = (d->pad_mode == RBR_PAD_SAME) ? (KF - kz) / 2 : 0;
                    p.slot_chan[ls] = (short)slots[gs].chan;
                    p.slot_w[ls] = (unsigned char)w;
                    p.slot_off[ls] = (unsigned char)off;
                    p.slot_kz[ls] = (unsigned char)kz;
                    lo[t] = std::min(lo[t], off);
                    hi[t] = std::max(hi[t], off + kz - 1);
                } else {
                    p.slot_chan[ls] = -1;
                }
            }
        }
        // tap-major piece order; two adjacent active tiles of one tap form ONE piece (they are contiguous in
        // the packed image and share the token-row operand), which halves the barriers per MFMA
        int np = 0;
        for (int s = 0; s < KF; ++s) {
            int t = 0;
            while (t < p.ntiles) {
                if (!(s >= lo[t] && s <= hi[t])) { ++t; continue; }
                const bool pair = (t + 1 < p.ntiles) && (s >= lo[t + 1] && s <= hi[t + 1]);
                p.piece_st[np++] = s | (t << 8) | ((pair ? 2 : 1) << 16);
                t += pair ? 2 : 1;
            }
        }
        p.npieces = np;
        p.tiles_total = tiles_total;
    }
    return ngroups;
}

}  // namespace rbr

namespace rbr {
PairState& pair_state() {
    static thread_local PairState S;
    return S;
}
}  // namespace rbr

// ---- pair regions (rbr_launch.h): the launches of two independent problems of equal shapes, zipped into shared launches
extern "C" int rbr_pair_begin(void) {
    rbr::PairState& S = rbr::pair_state();
    if (S.mode != 0) { rbr::set_error("rbr_pair_begin: a pair region is already open on this thread"); return RBR_ERR_BAD_ARG; }
    S.rec[0].clear(); S.rec[1].clear();
    S.mode = 1;
    return 0;
}
extern "C" int rbr_pair_next(void) {
    rbr::PairState& S = rbr::pair_state();
    if (S.mode != 1) { rbr::set_error("rbr_pair_next: expected after rbr_pair_begin, once"); return RBR_ERR_BAD_ARG; }
    S.mode = 2;
    return 0;
}
extern "C" void rbr_pair_abort(void) {
    rbr::PairState& S = rbr::pair_state();
    S.rec[0].clear(); S.rec[1].clear();
    S.mode = 0;
}
extern "C" int rbr_pair_end(int32_t* n_paired, int32_t* n_single) {
    rbr::PairState& S = rbr::pair_state();
    if (S.mode == 0) { rbr::set_error("rbr_pair_end: no pair region is open"); return RBR_ERR_BAD_ARG; }
    S.mode = 0;                              // from here on launches run
    S.paired = S.singles = 0;
    int rc = 0;
    const size_t n0 = S.rec[0].size(), n1 = S.rec[1].size(), nz = std::min(n0, n1);
    size_t i = 0;
    // zip while the two lists name the same kernels; a pair that does not form (grids, shared arguments) goes out as two
    // launches and the zip carries on -- the lists still line up.  Each problem's launches keep their order either way.
    while (i < nz && rc == 0) {
        const rbr::PairRec &a = S.rec[0][i], &b = S.rec[1][i];
        if (a.pair != b.pair || a.solo != b.solo) break;
        if (a.solo) {
            // a run of solo records: problem 0's launches of the run, then problem 1's (PairSolo)
            size_t j = i;
            while (j < nz && S.rec[0][j].solo && S.rec[1][j].solo && S.rec[0][j].pair == S.rec[1][j].pair) ++j;
            for (int t = 0; t < 2 && rc == 0; ++t)
                for (size_t k = i; k < j && rc == 0; ++k, ++S.singles) rc = S.rec[t][k].single(S.rec[t][k]);
            i = j;
            continue;
        }
        rc = a.pair(a, b);
        if (rc == rbr::kPairNoMatch) {
            rc = a.single(a);
            if (rc == 0) rc = b.single(b);
            S.singles += 2;
        } else if (rc == 0) {
            S.paired += 1;
        }
        ++i;
    }
    for (size_t k = i; k < n0 && rc == 0; ++k, ++S.singles) rc = S.rec[0][k].single(S.rec[0][k]);
    for (size_t k = i; k < n1 && rc == 0; ++k, ++S.singles) rc = S.rec[1][k].single(S.rec[1][k]);
    S.rec[0].clear(); S.rec[1].clear();
    if (n_paired) *n_paired = S.paired;
    if (n_single) *n_single = S.singles;
    return rc;
}

extern "C" int rbr_version(void) { return 1; }
extern "C" const char* rbr_last_error(void) { return rbr::g_err; }
