// pair_head.hip -- rating head: LastFeat (user) + LastFeat (item) + FM, forward and backward.
//
// Replaces models/deepconn/layers.py:156-165 (text_feat @ W + b + ebd[id]) for both towers and
// layers.py:189-209 (relu(u*i) -> dropout -> @h + user_bias[uid] + item_bias[iid] + g_bias), and the
// identical classes of models/narre/narre.py:66-137.  The "FM" here is an elementwise product and a
// K-long dot (K = latent_dim = 32), not an outer product: it is latency/HBM-bound VALU work, so it
// runs one wave per (user,item) pair with shuffle reductions -- no MFMA.
#include "rbr_common.h"

#include <algorithm>

namespace rbr {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct HeadTrain {               // rbr_pair_head_fwd_train extras; all zero for the plain forward
    float p_drop;
    unsigned long long seed;
    unsigned long long* rng_state;      // [call number, ticket], as rbr_dropout_multiplier
    float* drop_out;                    // [B,K] multiplier, written for the backward
    float* zero_buf;
    long zero_n;
};

// the last workgroup of the launch advances the call number (every pair block has read it by then)
__device__ __forceinline__ void head_rng_ticket(const HeadTrain& tr, unsigned long long call) {
    if (tr.rng_state == nullptr) return;
    // no fence: the call number was consumed (its value used) before this atomic is issued, and nothing else is published
    if (atomicAdd(tr.rng_state + 1, 1ull) == (unsigned long long)gridDim.x - 1) {
        tr.rng_state[1] = 0;
        tr.rng_state[0] = tr.rng_state[0] + 1;
    }
    (void)call;
}

// MSE of the trainers folded into the head's training launch (rbr_pair_head_fwd_pool): the pair block that arrives LAST at
// the ticket sees every prediction and computes nn.MSELoss(mean) and d loss / d pred exactly as mse_fwd_kernel does (same
// thread -> element mapping, same reduction order: the same bits), then re-arms the ticket.
struct HeadMse {
    const float* target;       // [B] or NULL: no loss in this launch
    float* loss;               // [1]
    float* d_unit;             // [B] d loss / d pred for an upstream gradient of 1 (may be NULL)
    int* ticket;               // zero between launches
};

// one workgroup per pair: threads [0,128) run the user tower, [128,256) the item tower; inside a tower 4 thread groups
// split the H-long dots by h mod 4 and 32 lanes cover k (looped for K > 32).  A thread's loads are issued 8 at a time.
// ft0 / ft1: this pair's user / item feature vectors (global memory, or the LDS copy the fused pool epilogue leaves).
__device__ __forceinline__ void head_pair(int b, int B, int H, int K, const float* ft0, const float* ft1,
                                          const long long* __restrict__ uid, const long long* __restrict__ iid,
                                          const rbr_head_params& p, const float* __restrict__ drop, float* __restrict__ ul,
                                          float* __restrict__ il, float* __restrict__ pred, const HeadTrain& tr,
                                          float (*s_part)[4][32], float (*s_l)[32]) {
    const int t = threadIdx.x;
    const unsigned long long call = (tr.p_drop > 0.f) ? tr.rng_state[0] : 0;
    const int side = t >> 7, hp = (t >> 5) & 3, kk = t & 31;
    const long id = side ? iid[b] : uid[b];
    const float* ft = side ? ft1 : ft0;
    const float* W = side ? p.Wi : p.Wu;
    const float* bias = side ? p.bi : p.bu;
    const float* E = side ? p.Ei : p.Eu;
    float* outl = (side ? il : ul) + (long)b * K;
    float part = 0.f;
    for (int k0 = 0; k0 < K; k0 += 32) {
        const int k = k0 + kk;
        float s = 0.f;
        if (k < K) {
            if (hp == 0) s = bias[k] + E[id * K + k];
            int hh = hp;
            for (; hh + 28 < H; hh += 32) {
                float f[8], wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { f[u] = ft[hh + 4 * u]; wv[u] = W[(long)(hh + 4 * u) * K + k]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) s = fmaf(f[u], wv[u], s);
            }
            for (; hh < H; hh += 4) s = fmaf(ft[hh], W[(long)hh * K + k], s);
        }
        s_part[side][hp][kk] = s;
        __syncthreads();
        if (hp == 0) {
            const float tot = (s_part[side][0][kk] + s_part[side][1][kk]) + (s_part[side][2][kk] + s_part[side][3][kk]);
            if (k < K) outl[k] = tot;
            s_l[side][kk] = tot;
        }
        __syncthreads();
        if (t < 32 && k < K) {
            float z = fmaxf(s_l[0][kk] * s_l[1][kk], 0.f);
            if (tr.p_drop > 0.f) {       // the draw rbr_dropout_multiplier would make for element (b, k) of this call
                const unsigned long long e = (unsigned long long)b * K + k;
                unsigned w[4];
                philox4x32_10(e >> 2, call, tr.seed, w[0], w[1], w[2], w[3]);
                const float m = dropout_keep(w[e & 3], tr.p_drop) ? 1.f / (1.f - tr.p_drop) : 0.f;
                tr.drop_out[e] = m;
                z *= m;
            } else if (drop != nullptr) {
                z *= drop[(long)b * K + k];
            }
            part = fmaf(z, p.h[k], part);
        }
    }
    if (t < 64) {
        part = wave_sum(part);
        if (t == 0) pred[b] = part + p.ub[uid[b]] + p.ib[iid[b]] + p.g[0];
    }
    if (t == 0 && tr.rng_state != nullptr) head_rng_ticket(tr, call);
    (void)B;
}

__global__ __launch_bounds__(256) void head_fwd_kernel(int B, int H, int K, const float* __restrict__ uf,
                                                       const float* __restrict__ itf, const long long* __restrict__ uid,
                                                       const long long* __restrict__ iid, const rbr_head_params p,
                                                       const float* __restrict__ drop, float* __restrict__ ul,
                                                       float* __restrict__ il, float* __restrict__ pred,
                                                       const HeadTrain tr) {
    __shared__ float s_part[2][4][32];
    __shared__ float s_l[2][32];
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= B) {                 // training launch only: blocks past the pairs clear the buffer the backward accumulates into
        for (long k = (long)(b - B) * 256 + t; k < tr.zero_n; k += (long)(gridDim.x - B) * 256) tr.zero_buf[k] = 0.f;
        if (t == 0) head_rng_ticket(tr, 0);
        return;
    }
    head_pair(b, B, H, K, uf + (long)b * H, itf + (long)b * H, uid, iid, p, drop, ul, il, pred, tr, s_part, s_l);
}

// The head forward with the encoder's pool epilogue in front and the trainer's loss behind (one launch instead of
// pool_finalize + head_fwd + mse_fwd): block b first reduces the slab partials of ITS two documents (rows b and B + b of the
// stacked encoder batch) exactly as pool_finalize_kernel does -- slabs in position order, first maximum wins, bias,
// activation -- writes feat / argmax for the backward and keeps the two feature vectors in LDS for the head.
struct HeadPool {
    const float* pval;
    const int* pidx;
    PtrArray bias;
    float* feat;               // [2B, C]
    int* argmax;               // [2B, C]
    const long long* first;    // [2B] or NULL: document r takes the pool partials of document first[r] (in-batch dedup by id: the
                               // repeated documents were blanked and not encoded; rbr_dedup_rows)
};

__global__ __launch_bounds__(256) void head_fwd_pool_kernel(const ConvPlan P, const HeadPool hp, int B, int K,
                                                            const long long* __restrict__ uid, const long long* __restrict__ iid,
                                                            const rbr_head_params p, const float* __restrict__ drop,
                                                            float* __restrict__ ul, float* __restrict__ il, float* __restrict__ pred,
                                                            const HeadTrain tr, const HeadMse mse) {
    __shared__ float s_part[2][4][32];
    __shared__ float s_l[2][32];
    __shared__ int s_last;
    extern __shared__ __attribute__((aligned(16))) float s_feat[];      // [2][C]
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= B) {
        for (long k = (long)(b - B) * 256 + t; k < tr.zero_n; k += (long)(gridDim.x - B) * 256) tr.zero_buf[k] = 0.f;
        if (t == 0) head_rng_ticket(tr, 0);
        return;
    }
    // Pool epilogue of this pair's two documents in three memory round trips: (1) the slabs' activity flags, (2) EVERY slab
    // partial of every channel slot at once (2 x wpd x nslots independent loads, ~20 per thread) into LDS, (3) the position of
    // each slot's winning slab.  An inactive slab (all tokens masked) stands for a conv sum of exactly 0 at its first position.
    const int* flags = hp.pidx + (long)P.total_wt * P.nslots_total;
    const int nslots = P.ntiles * kTile, wpd = P.wpd;
    float* s_val = s_feat + 2 * P.C;                                   // [2][wpd][nslots]
    int* s_flag = reinterpret_cast<int*>(s_val + 2 * wpd * nslots);    // [2][wpd]
    // (Measured and dropped, round 3: a two-round-trip form of this kernel -- every slab partial AND its position loaded
    // unconditionally beside the flags, the ids and the thread's column of W, the id-dependent operands in flight during the
    // winner scan, the dropout call number prefetched: 25.0 us against 24.6 us for this form.  By elimination the launch is
    // 11.6 us up to the first barrier, +4.8 for the winner scan and its stores, +7 for the head, +3 for the loss tail; none of
    // it follows the count of dependent loads.)
    // the plan's slot tables go through LDS with the flags: the plan is a kernel ARGUMENT, and a per-lane index into an argument
    // array is a vector load from the kernarg segment at every use
    __shared__ short s_chan[kMaxSlots];
    __shared__ unsigned char s_sw[kMaxSlots], s_skz[kMaxSlots];
    __shared__ int s_choff[RBR_MAX_WIDTHS];
    __shared__ const float* s_biasp[RBR_MAX_WIDTHS];
    static_assert(kMaxSlots <= 256, "one thread per slot below");
    if (t < kMaxSlots) { s_chan[t] = P.slot_chan[t]; s_sw[t] = P.slot_w[t]; s_skz[t] = P.slot_kz[t]; }
    if (t < RBR_MAX_WIDTHS) { s_choff[t] = P.ch_off[t]; s_biasp[t] = hp.bias.p[t]; }
    __shared__ int s_src[2];                                           // the documents whose partials this pair reads
    if (t < 2) s_src[t] = hp.first != nullptr ? (int)hp.first[t * B + b] : t * B + b;
    __syncthreads();
    for (int i = t; i < 2 * wpd; i += 256) {
        const int side = i >= wpd, w = side ? i - wpd : i;
        s_flag[i] = flags[s_src[side] * wpd + w];
    }
    __syncthreads();
    const float NEG = -__builtin_huge_valf();
    for (int i = t; i < 2 * wpd * nslots; i += 256) {
        const int side = i >= wpd * nslots, r = side ? i - wpd * nslots : i;
        const int w = r / nslots, ls = r - w * nslots;
        const long base = ((long)s_src[side] * wpd + w) * P.nslots_total + (long)P.tile_base * kTile + ls;
        float v = NEG;
        if (s_chan[ls] >= 0) {
            const int Lv = (P.pad_mode == RBR_PAD_VALID) ? (P.L - (int)s_skz[ls] + 1) : P.L;
            const int fl = s_flag[side * wpd + w];          // 1 computed, 0 all masked (sum exactly 0), kSlabDup: changes nothing
            v = (fl == 1) ? hp.pval[base] : ((fl == 0 && w * kTile < Lv) ? 0.f : NEG);
        }
        s_val[i] = v;
    }
    __syncthreads();
    for (int idx = t; idx < 2 * nslots; idx += 256) {
        const int side = idx >= nslots, ls = side ? idx - nslots : idx;
        const int doc = side * B + b;
        const int chan = s_chan[ls];
        if (chan < 0) continue;
        float best = NEG;
        int bw_tile = -1;
        for (int w = 0; w < wpd; ++w) {                                // slabs in position order, first maximum wins
            const float v = s_val[(side * wpd + w) * nslots + ls];
            if (v > best) { best = v; bw_tile = w; }
        }
        int bidx = 0;
        if (bw_tile >= 0) {
            if (s_flag[side * wpd + bw_tile] == 1) bidx = hp.pidx[((long)s_src[side] * wpd + bw_tile) * P.nslots_total + (long)P.tile_base * kTile + ls];
            else bidx = bw_tile * kTile;
        }
        const int bw = s_sw[ls];
        const float y = best + s_biasp[bw][chan - s_choff[bw]];
        const float f = (P.act == RBR_ACT_RELU) ? fmaxf(y, 0.f) : tanhf(y);
        hp.feat[(long)doc * P.C + chan] = f;
        hp.argmax[(long)doc * P.C + chan] = bidx;
        s_feat[side * P.C + chan] = f;
    }
    __syncthreads();
    head_pair(b, B, P.C, K, s_feat, s_feat + P.C, uid, iid, p, drop, ul, il, pred, tr, s_part, s_l);
    if (mse.target == nullptr) return;
    if (t == 0) {
        __threadfence();                                        // pred[b] is visible before the ticket is taken
        s_last = atomicAdd(mse.ticket, 1) == B - 1;
    }
    __syncthreads();
    if (!s_last) return;                                        // workgroup-uniform
    __threadfence();
    float acc = 0.f;
    for (int i = t; i < B; i += 256) {                          // as mse_fwd_kernel; the other blocks' preds via agent-scope loads
        const float d = __hip_atomic_load(pred + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - mse.target[i];
        acc += d * d;
        if (mse.d_unit != nullptr) mse.d_unit[i] = d * (2.f / (float)B);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __syncthreads();
    if ((t & 63) == 0) s_l[0][t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        *mse.loss = (s_l[0][0] + s_l[0][1] + s_l[0][2] + s_l[0][3]) / (float)B;
        *mse.ticket = 0;
    }
}

// The backward in ONE launch.  Blocks [0, B): one workgroup per pair -- embedding-row grads (atomics), then d_feat with all
// 256 threads over the 2*H outputs (rows of Wu / Wi read as float4 when K allows).  Blocks [B, B + 2(H+1)): the batch
// reductions dW = feat^T @ d_l, db, dh, dg, one workgroup per output row (side, r), 8 thread groups striding the batch in a
// fixed partition and order (bitwise reproducible); they recompute d_l from ul / il / drop / d_pred (a few flops) instead
// of waiting for the pair blocks to publish it, so both halves run side by side.
__device__ __forceinline__ void head_dl(float su, float si, float dr, float dp, float hk, float& dul, float& dil, float& zdp) {
    const float prod = su * si;
    const float dz = (prod > 0.f) ? dp * hk * dr : 0.f;
    dul = dz * si;
    dil = dz * su;
    zdp = fmaxf(prod, 0.f) * dr * dp;
}

__global__ __launch_bounds__(256) void head_bwd_kernel(int B, int H, int K, const float* __restrict__ uf,
                                                       const float* __restrict__ itf, const long long* __restrict__ uid,
                                                       const long long* __restrict__ iid, const rbr_head_params p,
                                                       const float* __restrict__ drop, const float* __restrict__ ul,
                                                       const float* __restrict__ il, const float* __restrict__ d_pred,
                                                       int pad_u, int pad_i, const rbr_head_grads g,
                                                       float* __restrict__ d_uf, float* __restrict__ d_if) {
    // (Measured and dropped in round 3: clearing the conv backward's G with extra workgroups of this launch instead of the
    // zero_g_rows launch -- the step got 19 us SLOWER: the weight-gradient branch then starts beside the L2-bound sparse product
    // instead of beside the zero fill and the atomics-bound G build, and both crawl.  Round 4, the same 64 MB cleared by extra
    // workgroups of the FORWARD's head launch, which has the chip to itself for 25 us: that launch 26 -> 37 us -- its pair blocks'
    // dependent round trips wait behind the write burst -- and the step +25 us: a G cleared 130 us before build_g has left the
    // Infinity Cache again by then, cleared right in front of it the atomics and g_times_w's reads find it there.)
    extern __shared__ __attribute__((aligned(16))) float sm[];   // pair role: [2][K4]; reduce role: [3][8][32]
    const int t = threadIdx.x;
    if ((int)blockIdx.x < B) {
        const int K4 = (K + 3) & ~3;
        const int b = blockIdx.x;
        float* s_dul = sm;
        float* s_dil = sm + K4;
        const long u = uid[b], it = iid[b];
        const float dp = d_pred[b];
        for (int k = t; k < K; k += 256) {
            const float dr = (drop != nullptr) ? drop[(long)b * K + k] : 1.f;
            float dul, dil, zdp;
            head_dl(ul[(long)b * K + k], il[(long)b * K + k], dr, dp, p.h[k], dul, dil, zdp);
            s_dul[k] = dul;
            s_dil[k] = dil;
            if (u != pad_u) atomicAdd(g.dEu + u * K + k, dul);
            if (it != pad_i) atomicAdd(g.dEi + it * K + k, dil);
        }
        if (t == 0) {
            if (u != pad_u) atomicAdd(g.dub + u, dp);
            if (it != pad_i) atomicAdd(g.dib + it, dp);
        }
        __syncthreads();
        const bool vec = (K & 3) == 0 && ((((uintptr_t)p.Wu) | ((uintptr_t)p.Wi)) & 15) == 0;
        for (int e = t; e < 2 * H; e += 256) {
            const int side = e >= H, hh = side ? e - H : e;
            const float* wrow = (side ? p.Wi : p.Wu) + (long)hh * K;
            const float* sd = side ? s_dil : s_dul;
            float a = 0.f;
            if (vec) {
                for (int k = 0; k < K; k += 4) {
                    const float4 wv = *reinterpret_cast<const float4*>(wrow + k);
                    const float4 dv = *reinterpret_cast<const float4*>(sd + k);
                    a = fmaf(dv.x, wv.x, a); a = fmaf(dv.y, wv.y, a); a = fmaf(dv.z, wv.z, a); a = fmaf(dv.w, wv.w, a);
                }
            } else {
                for (int k = 0; k < K; ++k) a = fmaf(sd[k], wrow[k], a);
            }
            (side ? d_if : d_uf)[(long)b * H + hh] = a;
        }
        return;
    }
    float (*red)[8][32] = reinterpret_cast<float (*)[8][32]>(sm);
    const int rows = H + 1;                 // row H holds the bias / h / g reductions
    const int rb = blockIdx.x - B;
    const int side = rb / rows, r = rb % rows;
    const int kk = t & 31, grp = t >> 5;
    const float* ft = side ? itf : uf;
    for (int k0 = 0; k0 < K; k0 += 32) {
        const int k = k0 + kk;
        float s = 0.f, hs = 0.f, gs = 0.f;
        if (k < K) {
            const float hk = p.h[k];
            // 8 batch rows per round: their 40 loads are issued together (a row past the batch repeats row `grp` with
            // d_pred = 0, which contributes exact zeros)
            for (int b0 = grp; b0 < B; b0 += 64) {
                float vu[8], vi[8], vd[8], vp[8], vf[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = b0 + 8 * u;
                    const bool ok = b < B;
                    const long bb = ok ? b : grp;
                    vu[u] = ul[bb * K + k];
                    vi[u] = il[bb * K + k];
                    vd[u] = (drop != nullptr) ? drop[bb * K + k] : 1.f;
                    vp[u] = ok ? d_pred[bb] : 0.f;
                    vf[u] = (r < H) ? ft[bb * H + r] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float dul, dil, zdp;
                    head_dl(vu[u], vi[u], vd[u], vp[u], hk, dul, dil, zdp);
                    const float dl = side ? dil : dul;
                    if (r < H) {
                        s = fmaf(vf[u], dl, s);
                    } else {
                        s += dl;
                        if (side == 0) {
                            hs += zdp;
                            if (k == 0) gs += vp[u];
                        }
                    }
                }
            }
        }
        __syncthreads();
        red[0][grp][kk] = s; red[1][grp][kk] = hs; red[2][grp][kk] = gs;
        __syncthreads();
        if (grp == 0 && k < K) {
            float a = 0.f, bsum = 0.f, c = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) { a += red[0][q][kk]; bsum += red[1][q][kk]; c += red[2][q][kk]; }
            if (r < H) {
                (side ? g.dWi : g.dWu)[(long)r * K + k] = a;
            } else {
                (side ? g.dbi : g.dbu)[k] = a;
                if (side == 0) {
                    g.dh[k] = bsum;
                    if (k == 0) g.dg[0] = c;
                }
            }
        }
    }
}

}  // namespace rbr

using namespace rbr;

static bool head_args_ok(int B, int H, int K) {
    if (B <= 0 || H <= 0 || K <= 0) { set_error("bad head shape B=%d H=%d K=%d", B, H, K); return false; }
    return true;
}

namespace rbr {
// ---- nn.MSELoss (mean) of the trainers.  One workgroup, fixed summation order: the loss is bitwise reproducible.
__global__ __launch_bounds__(256) void mse_fwd_kernel(long n, const float* __restrict__ pred, const float* __restrict__ target,
                                                      float* __restrict__ loss, float* __restrict__ d_unit) {
    __shared__ float s_red[4];
    float acc = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float d = pred[i] - target[i];
        acc += d * d;
        if (d_unit != nullptr) d_unit[i] = d * (2.f / (float)n);       // d loss / d pred for an upstream gradient of 1
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)n;
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(long n, const float* __restrict__ pred, const float* __restrict__ target,
                                                      const float* __restrict__ d_loss, float* __restrict__ d_pred) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d_pred[i] = (pred[i] - target[i]) * (2.f / (float)n) * d_loss[0];
}

}  // namespace rbr

extern "C" int rbr_mse_loss_fwd(int64_t n, const float* pred, const float* target, float* loss, float* d_pred_unit, void* stream) {
    if (n <= 0 || !pred || !target || !loss) { rbr::set_error("mse_loss_fwd: n=%lld or null pointer", (long long)n); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(rbr::mse_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (long)n, pred, target, loss, d_pred_unit);
    RBR_CHECK_LAUNCH("mse_fwd launch");
    return 0;
}

extern "C" int rbr_mse_loss_bwd(int64_t n, const float* pred, const float* target, const float* d_loss, float* d_pred,
                                void* stream) {
    if (n <= 0 || !pred || !target || !d_loss || !d_pred) { rbr::set_error("mse_loss_bwd: n=%lld or null pointer", (long long)n); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(rbr::mse_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n, pred,
                       target, d_loss, d_pred);
    RBR_CHECK_LAUNCH("mse_bwd launch");
    return 0;
}

extern "C" int rbr_pair_head_fwd(int32_t B, int32_t H, int32_t K, const float* u_feat, const float* i_feat,
                                 const int64_t* u_id, const int64_t* i_id, const rbr_head_params* p, const float* drop,
                                 float* ul, float* il, float* pred, void* stream) {
    if (!head_args_ok(B, H, K)) return RBR_ERR_BAD_ARG;
    if (!u_feat || !i_feat || !u_id || !i_id || !p || !ul || !il || !pred) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, B, H, K, u_feat, i_feat,
                       reinterpret_cast<const long long*>(u_id), reinterpret_cast<const long long*>(i_id), *p, drop, ul,
                       il, pred, HeadTrain{});
    RBR_CHECK_LAUNCH("pair_head_fwd launch");
    return 0;
}

extern "C" int rbr_pair_head_fwd_train(int32_t B, int32_t H, int32_t K, const float* u_feat, const float* i_feat,
                                       const int64_t* u_id, const int64_t* i_id, const rbr_head_params* p, float p_drop,
                                       uint64_t seed, uint64_t* rng_state, float* drop_out, float* zero_buf, int64_t zero_n,
                                       float* ul, float* il, float* pred, void* stream) {
    if (!head_args_ok(B, H, K)) return RBR_ERR_BAD_ARG;
    if (!u_feat || !i_feat || !u_id || !i_id || !p || !ul || !il || !pred) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if (!(p_drop >= 0.f && p_drop < 1.f)) { set_error("pair_head_fwd_train: p_drop=%f outside [0,1)", (double)p_drop); return RBR_ERR_BAD_ARG; }
    if (p_drop > 0.f && (!rng_state || !drop_out)) { set_error("pair_head_fwd_train: dropout needs rng_state and drop_out"); return RBR_ERR_BAD_ARG; }
    if (zero_n < 0 || (zero_n > 0 && !zero_buf)) { set_error("pair_head_fwd_train: bad zero buffer"); return RBR_ERR_BAD_ARG; }
    HeadTrain tr{};
    tr.p_drop = p_drop; tr.seed = seed;
    tr.rng_state = (p_drop > 0.f) ? reinterpret_cast<unsigned long long*>(rng_state) : nullptr;
    tr.drop_out = drop_out; tr.zero_buf = zero_buf; tr.zero_n = zero_n;
    const int zblocks = zero_n > 0 ? (int)std::min<long>((zero_n + 1023) / 1024, 256) : 0;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B + zblocks), dim3(256), 0, (hipStream_t)stream, B, H, K, u_feat, i_feat,
                       reinterpret_cast<const long long*>(u_id), reinterpret_cast<const long long*>(i_id), *p,
                       static_cast<const float*>(nullptr), ul, il, pred, tr);
    RBR_CHECK_LAUNCH("pair_head_fwd_train launch");
    return 0;
}

// Head forward with the encoder's pool epilogue (rbr_textcnn_pool_finalize) in front and, optionally, the trainers' MSELoss
// behind: one launch.  The encoder batch holds the B user documents, then the B item documents.
extern "C" int rbr_pair_head_fwd_pool(const rbr_textcnn_desc* d, const float* pval, const int32_t* pidx, const float* const* bias,
                                      float* feat, int32_t* argmax, const int64_t* first, int32_t K, const int64_t* u_id,
                                      const int64_t* i_id, const rbr_head_params* p, const float* drop, float p_drop, uint64_t seed,
                                      uint64_t* rng_state, float* drop_out, float* zero_buf, int64_t zero_n, float* ul, float* il,
                                      float* pred, const float* target, float* loss, float* d_pred_unit, int32_t* ticket, void* stream) {
    ConvPlan plans[kMaxGroups];
    const int ng = build_plans(d, plans, kMaxTiles);
    if (!ng) return RBR_ERR_BAD_ARG;
    if (ng != 1) { set_error("pair_head_fwd_pool: %d channel groups (one launch covers %d slots)", ng, kMaxSlots); return RBR_ERR_UNSUPPORTED; }
    if (d->n_docs % 2) { set_error("pair_head_fwd_pool: odd document count %d", d->n_docs); return RBR_ERR_BAD_ARG; }
    const int B = d->n_docs / 2, C = plans[0].C;
    if (!head_args_ok(B, C, K)) return RBR_ERR_BAD_ARG;
    if (!pval || !pidx || !bias || !feat || !argmax || !u_id || !i_id || !p || !ul || !il || !pred) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if (!(p_drop >= 0.f && p_drop < 1.f)) { set_error("pair_head_fwd_pool: p_drop=%f outside [0,1)", (double)p_drop); return RBR_ERR_BAD_ARG; }
    if (p_drop > 0.f && (!rng_state || !drop_out)) { set_error("pair_head_fwd_pool: dropout needs rng_state and drop_out"); return RBR_ERR_BAD_ARG; }
    if (zero_n < 0 || (zero_n > 0 && !zero_buf)) { set_error("pair_head_fwd_pool: bad zero buffer"); return RBR_ERR_BAD_ARG; }
    if (target != nullptr && (!loss || !ticket)) { set_error("pair_head_fwd_pool: the loss needs loss and ticket"); return RBR_ERR_BAD_ARG; }
    HeadTrain tr{};
    tr.p_drop = p_drop; tr.seed = seed;
    tr.rng_state = (p_drop > 0.f) ? reinterpret_cast<unsigned long long*>(rng_state) : nullptr;
    tr.drop_out = drop_out; tr.zero_buf = zero_buf; tr.zero_n = zero_n;
    HeadPool hp{};
    hp.pval = pval; hp.pidx = pidx; hp.feat = feat; hp.argmax = argmax; hp.first = reinterpret_cast<const long long*>(first);
    for (int w = 0; w < d->n_widths; ++w) hp.bias.p[w] = bias[w];
    HeadMse mse{target, loss, d_pred_unit, ticket};
    const int zblocks = zero_n > 0 ? (int)std::min<long>((zero_n + 1023) / 1024, 256) : 0;
    const size_t lds = ((size_t)2 * C + (size_t)2 * plans[0].wpd * plans[0].ntiles * kTile + (size_t)2 * plans[0].wpd) * sizeof(float);
    if (lds > 60 * 1024) { set_error("pair_head_fwd_pool: %zu bytes of LDS for the slab partials (documents too long)", lds); return RBR_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(head_fwd_pool_kernel, dim3(B + zblocks), dim3(256), lds, (hipStream_t)stream,
                       plans[0], hp, B, K, reinterpret_cast<const long long*>(u_id), reinterpret_cast<const long long*>(i_id), *p,
                       (p_drop > 0.f) ? nullptr : drop, ul, il, pred, tr, mse);
    RBR_CHECK_LAUNCH("pair_head_fwd_pool launch");
    return 0;
}

extern "C" size_t rbr_pair_head_bwd_ws_floats(int32_t B, int32_t K) { (void)B; (void)K; return 0; }

extern "C" int rbr_pair_head_bwd(int32_t B, int32_t H, int32_t K, const float* u_feat, const float* i_feat,
                                 const int64_t* u_id, const int64_t* i_id, const rbr_head_params* p, const float* drop,
                                 const float* ul, const float* il, const float* d_pred, int32_t pad_u, int32_t pad_i,
                                 const rbr_head_grads* g, float* d_ufeat, float* d_ifeat, float* ws, void* stream) {
    if (!head_args_ok(B, H, K)) return RBR_ERR_BAD_ARG;
    if (!u_feat || !i_feat || !u_id || !i_id || !p || !ul || !il || !d_pred || !g || !d_ufeat || !d_ifeat) {
        set_error("null pointer");
        return RBR_ERR_BAD_ARG;
    }
    (void)ws;      // no longer needed: the reduction blocks recompute d_l instead of reading it back
    const size_t lds = std::max((size_t)2 * ((K + 3) & ~3), (size_t)3 * 8 * 32) * sizeof(float);
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)(B + 2 * (H + 1))), dim3(256), lds, (hipStream_t)stream, B, H, K, u_feat,
                       i_feat, reinterpret_cast<const long long*>(u_id), reinterpret_cast<const long long*>(i_id), *p, drop,
                       ul, il, d_pred, pad_u, pad_i, *g, d_ufeat, d_ifeat);
    RBR_CHECK_LAUNCH("pair_head_bwd launch");
    return 0;
}

// ---- D-ATT's rating: ratings[b] = sum_k u[b,k] * i[b,k] (dual_att.py:58) over a STACKED [2B, K] feature block (user rows
// first, the shared fc's output), and its backward straight into the stacked gradient -- two launches instead of mul, sum,
// two elementwise products and a stack.  16 lanes per pair.
namespace rbr {
__global__ __launch_bounds__(256) void pair_dot_fwd_kernel(int B, int K, const float* __restrict__ x, float* __restrict__ out) {
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
    float s = 0.f;
    if (b < B)
        for (int k = l; k < K; k += 16) s = fmaf(x[(long)b * K + k], x[(long)(B + b) * K + k], s);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (b < B && l == 0) out[b] = s;
}

__global__ __launch_bounds__(256) void pair_dot_bwd_kernel(int B, int K, const float* __restrict__ x, const float* __restrict__ d_out,
                                                           float* __restrict__ d_x) {
    const long n = (long)2 * B * K;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int row = (int)(e / K), k = (int)(e - (long)row * K);
        const int b = row < B ? row : row - B, other = row < B ? row + B : row - B;
        d_x[e] = d_out[b] * x[(long)other * K + k];
    }
}
}  // namespace rbr

extern "C" int rbr_pair_dot_fwd(int32_t B, int32_t K, const float* x, float* out, void* stream) {
    if (B <= 0 || K <= 0 || !x || !out) { set_error("bad pair_dot arguments"); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(pair_dot_fwd_kernel, dim3((B + 15) / 16), dim3(256), 0, (hipStream_t)stream, B, K, x, out);
    RBR_CHECK_LAUNCH("pair_dot fwd launch");
    return 0;
}

extern "C" int rbr_pair_dot_bwd(int32_t B, int32_t K, const float* x, const float* d_out, float* d_x, void* stream) {
    if (B <= 0 || K <= 0 || !x || !d_out || !d_x) { set_error("bad pair_dot arguments"); return RBR_ERR_BAD_ARG; }
    const long n = (long)2 * B * K;
    hipLaunchKernelGGL(pair_dot_bwd_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, B, K,
                       x, d_out, d_x);
    RBR_CHECK_LAUNCH("pair_dot bwd launch");
    return 0;
}

