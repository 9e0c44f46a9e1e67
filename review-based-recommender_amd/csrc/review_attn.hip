// review_attn.hip -- NARRE review-level attention pool, forward and backward.
//
// Replaces models/narre/narre.py:40-64 (LinearAttention.forward):
//     e      = ebd_vals[other_id]                                   [B,R,A]
//     logit  = relu(feat @ W_rv + e @ W_id + b_1) @ h + b_2         [B,R,1]
//     att    = exp(logit) / (sum_R exp(logit) + 1e-8)               (no mask, no max subtraction)
//     out    = sum_R att * feat                                     [B,H]
// ~12 kFLOP per (sample, side): latency-bound VALU work, one workgroup per sample; the R x A
// hidden tile, the logits and the softmax live in LDS.  No MFMA: the contractions are 150- and
// 32-long dots on 10 rows.
#include "rbr_common.h"

namespace rbr {

// Every kernel of this file takes the arguments of TWO sides (blockIdx.y = 0 / 1): NARRE's user and item attention pools have
// the same shapes and different parameters, and one launch per stage for both replaces two launches on two streams (and the
// fork / join and the stack of the two gradients around them).  A one-sided call launches grid.y = 1.
struct AttnFwdSide {
    const float* feat; const long long* oid; rbr_attn_params p; float* out; float* att; float* hid; const float* drop;
};

__global__ __launch_bounds__(256) void attn_fwd_kernel(int B, int R, int H, int A, const AttnFwdSide side0, const AttnFwdSide side1) {
    const AttnFwdSide& SD = blockIdx.y ? side1 : side0;
    const float* __restrict__ feat = SD.feat;
    const long long* __restrict__ oid = SD.oid;
    const rbr_attn_params p = SD.p;
    float* __restrict__ out = SD.out;
    float* __restrict__ att = SD.att;
    float* __restrict__ hid = SD.hid;
    const float* __restrict__ drop = SD.drop;
    extern __shared__ float sm[];
    float* s_hid = sm;                  // [R*A]
    float* s_e = s_hid + R * A;         // [R] exp(logit), then att
    float* s_f = s_e + R;               // [R*H] this sample's review features
    float* s_eb = s_f + R * H;          // [R*A] counterpart-id embedding rows
    float* s_wr = s_eb + R * A;         // [H*A] W_rv
    float* s_wi = s_wr + H * A;         // [A*A] W_id
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* fb = feat + (long)b * R * H;
    for (int e = tid; e < R * H; e += 256) s_f[e] = fb[e];
    for (int e = tid; e < R * A; e += 256) s_eb[e] = p.ebd[oid[(long)b * R + e / A] * A + (e % A)];
    for (int e = tid; e < H * A; e += 256) s_wr[e] = p.W_rv[e];
    for (int e = tid; e < A * A; e += 256) s_wi[e] = p.W_id[e];
    __syncthreads();
    for (int idx = tid; idx < R * A; idx += 256) {
        const int r = idx / A, a = idx - r * A;
        float acc = p.b1[a];
        for (int a2 = 0; a2 < A; ++a2) acc = fmaf(s_eb[r * A + a2], s_wi[a2 * A + a], acc);
        for (int hh = 0; hh < H; ++hh) acc = fmaf(s_f[r * H + hh], s_wr[hh * A + a], acc);
        const float hv = fmaxf(acc, 0.f);
        hid[((long)b * R + r) * A + a] = hv;
        s_hid[idx] = hv;
    }
    __syncthreads();
    for (int r = tid; r < R; r += 256) {
        float lg = p.b2[0];
        for (int a = 0; a < A; ++a) lg = fmaf(s_hid[r * A + a], p.h[a], lg);
        s_e[r] = expf(lg);
    }
    __syncthreads();
    float denom = 1e-8f;
    for (int r = 0; r < R; ++r) denom += s_e[r];      // every thread, same order
    __syncthreads();
    for (int r = tid; r < R; r += 256) {
        const float v = s_e[r] / denom;
        s_e[r] = v;
        att[(long)b * R + r] = v;
    }
    __syncthreads();
    for (int hh = tid; hh < H; hh += 256) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s = fmaf(s_e[r], s_f[r * H + hh], s);
        out[(long)b * H + hh] = (drop != nullptr) ? s * drop[(long)b * H + hh] : s;     // nn.Dropout on the pooled feature (narre.py:62)
    }
}

// per-sample backward: d_feat, embedding-row gradient, and the per-row d_pre / d_logit the reduction needs
struct AttnBwdSide {
    const float* feat; const long long* oid; rbr_attn_params p; const float* att; const float* hid; const float* d_out;
    const float* drop; const float* d_att; int pad_idx; float* debd; float* d_feat; float* ws_dpre; float* ws_dl; float* part;
    rbr_attn_grads g;
};

__global__ __launch_bounds__(256) void attn_bwd_sample_kernel(int B, int R, int H, int A, const AttnBwdSide side0,
                                                              const AttnBwdSide side1) {
    const AttnBwdSide& SD = blockIdx.y ? side1 : side0;
    const float* __restrict__ feat = SD.feat;
    const long long* __restrict__ oid = SD.oid;
    const rbr_attn_params p = SD.p;
    const float* __restrict__ att = SD.att;
    const float* __restrict__ hid = SD.hid;
    const float* __restrict__ d_out = SD.d_out;
    const float* __restrict__ drop = SD.drop;
    const float* __restrict__ d_att = SD.d_att;
    const int pad_idx = SD.pad_idx;
    float* __restrict__ debd = SD.debd;
    float* __restrict__ d_feat = SD.d_feat;
    float* __restrict__ ws_dpre = SD.ws_dpre;
    float* __restrict__ ws_dl = SD.ws_dl;
    extern __shared__ float sm[];
    float* s_dpre = sm;            // [R*A]
    float* s_da = sm + R * A;      // [R] d(att)
    float* s_dl = s_da + R;        // [R] d(logit)
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* fb = feat + (long)b * R * H;
    const float* ab = att + (long)b * R;
    {   // d att_r = d_att_r + <d_out, feat_r>: one wave per review, lanes over the hidden dim (R is ~10: a thread per review
        // left 246 threads idle behind ten H-long chains of dependent loads)
        const int wave = tid >> 6, lane = tid & 63;
        for (int r = wave; r < R; r += 4) {
            float s = 0.f;
            for (int hh = lane; hh < H; hh += 64) {
                const float g = (drop != nullptr) ? d_out[(long)b * H + hh] * drop[(long)b * H + hh] : d_out[(long)b * H + hh];
                s = fmaf(g, fb[(long)r * H + hh], s);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) s_da[r] = s + ((d_att != nullptr) ? d_att[(long)b * R + r] : 0.f);
        }
    }
    __syncthreads();
    float S = 0.f;
    for (int r = 0; r < R; ++r) S = fmaf(s_da[r], ab[r], S);
    for (int r = tid; r < R; r += 256) {
        // att = e / (sum e + eps)  =>  d logit_r = att_r * (d att_r - sum_q d att_q * att_q)   (exact with eps)
        const float dl = ab[r] * (s_da[r] - S);
        s_dl[r] = dl;
        ws_dl[(long)b * R + r] = dl;
    }
    __syncthreads();
    for (int idx = tid; idx < R * A; idx += 256) {
        const int r = idx / A, a = idx - r * A;
        const float hv = hid[((long)b * R + r) * A + a];
        const float v = (hv > 0.f) ? s_dl[r] * p.h[a] : 0.f;
        s_dpre[idx] = v;
        ws_dpre[((long)b * R + r) * A + a] = v;
    }
    __syncthreads();
    const bool vec = (A & 3) == 0 && ((((uintptr_t)p.W_rv) | ((uintptr_t)p.W_id)) & 15) == 0;     // rows of A floats as float4
    for (int idx = tid; idx < R * H; idx += 256) {
        const int r = idx / H, hh = idx - r * H;
        float s = ab[r] * ((drop != nullptr) ? d_out[(long)b * H + hh] * drop[(long)b * H + hh] : d_out[(long)b * H + hh]);
        const float* wrow = p.W_rv + (long)hh * A;
        if (vec) {
            for (int a = 0; a < A; a += 4) {
                const float4 wv = *reinterpret_cast<const float4*>(wrow + a);
                s = fmaf(s_dpre[r * A + a], wv.x, s); s = fmaf(s_dpre[r * A + a + 1], wv.y, s);
                s = fmaf(s_dpre[r * A + a + 2], wv.z, s); s = fmaf(s_dpre[r * A + a + 3], wv.w, s);
            }
        } else {
            for (int a = 0; a < A; ++a) s = fmaf(s_dpre[r * A + a], wrow[a], s);
        }
        d_feat[((long)b * R + r) * H + hh] = s;
    }
    for (int idx = tid; idx < R * A; idx += 256) {
        const int r = idx / A, a2 = idx - r * A;
        const long id = oid[(long)b * R + r];
        if (id == pad_idx) continue;                 // nn.Embedding(padding_idx): no gradient for the pad row
        float s = 0.f;
        const float* wrow = p.W_id + (long)a2 * A;
        if (vec) {
            for (int a = 0; a < A; a += 4) {
                const float4 wv = *reinterpret_cast<const float4*>(wrow + a);
                s = fmaf(s_dpre[r * A + a], wv.x, s); s = fmaf(s_dpre[r * A + a + 1], wv.y, s);
                s = fmaf(s_dpre[r * A + a + 2], wv.z, s); s = fmaf(s_dpre[r * A + a + 3], wv.w, s);
            }
        } else {
            for (int a = 0; a < A; ++a) s = fmaf(s_dpre[r * A + a], wrow[a], s);
        }
        atomicAdd(debd + id * A + a2, s);
    }
}

// reductions over the N = B*R rows in two stages, fixed partition and order (bitwise reproducible):
//   stage 1: one workgroup per chunk of kRedRows rows stages d_pre / feat / embedding rows / hid / d_logit in LDS and writes the
//            chunk's partial of every output;   stage 2: one thread per output sums the chunks in order.
// output index: [0, H*A) dW_rv | [H*A, H*A + A*A) dW_id | + A db1 | + A dh | + 1 db2
constexpr int kRedRows = 16;      // rows per chunk: B*R = 2560 rows give 160 workgroups (64-row chunks left 5/6 of the CUs idle)

__device__ __forceinline__ int attn_n_out(int H, int A) { return H * A + A * A + 2 * A + 1; }

__global__ __launch_bounds__(256) void attn_bwd_partial_kernel(int N, int H, int A, const AttnBwdSide side0, const AttnBwdSide side1) {
    const AttnBwdSide& SD = blockIdx.y ? side1 : side0;
    const float* __restrict__ feat = SD.feat;
    const long long* __restrict__ oid = SD.oid;
    const float* __restrict__ ebd = SD.p.ebd;
    const float* __restrict__ hid = SD.hid;
    const float* __restrict__ ws_dpre = SD.ws_dpre;
    const float* __restrict__ ws_dl = SD.ws_dl;
    float* __restrict__ part = SD.part;
    extern __shared__ float sm[];
    float* s_dp = sm;                       // [rows][A]
    float* s_f = s_dp + kRedRows * A;       // [rows][H]
    float* s_eb = s_f + kRedRows * H;       // [rows][A]
    float* s_hd = s_eb + kRedRows * A;      // [rows][A]
    float* s_dl = s_hd + kRedRows * A;      // [rows]
    const int n0 = blockIdx.x * kRedRows, rows = min(kRedRows, N - n0), tid = threadIdx.x;
    for (int e = tid; e < rows * A; e += 256) {
        s_dp[e] = ws_dpre[(long)n0 * A + e];
        s_hd[e] = hid[(long)n0 * A + e];
        s_eb[e] = ebd[oid[n0 + e / A] * A + (e % A)];
    }
    for (int e = tid; e < rows * H; e += 256) s_f[e] = feat[(long)n0 * H + e];
    for (int e = tid; e < rows; e += 256) s_dl[e] = ws_dl[n0 + e];
    __syncthreads();
    const int n_out = attn_n_out(H, A);
    float* my = part + (long)blockIdx.x * n_out;
    for (int o = tid; o < n_out; o += 256) {
        float acc = 0.f;
        if (o < H * A) {
            const int hh = o / A, a = o - hh * A;
            for (int n = 0; n < rows; ++n) acc = fmaf(s_f[n * H + hh], s_dp[n * A + a], acc);
        } else if (o < H * A + A * A) {
            const int q = o - H * A, a2 = q / A, a = q - a2 * A;
            for (int n = 0; n < rows; ++n) acc = fmaf(s_eb[n * A + a2], s_dp[n * A + a], acc);
        } else if (o < H * A + A * A + A) {
            const int a = o - H * A - A * A;
            for (int n = 0; n < rows; ++n) acc += s_dp[n * A + a];
        } else if (o < H * A + A * A + 2 * A) {
            const int a = o - H * A - A * A - A;
            for (int n = 0; n < rows; ++n) acc = fmaf(s_dl[n], s_hd[n * A + a], acc);
        } else {
            for (int n = 0; n < rows; ++n) acc += s_dl[n];
        }
        my[o] = acc;
    }
}

__global__ __launch_bounds__(256) void attn_bwd_final_kernel(int n_chunks, int H, int A, const AttnBwdSide side0,
                                                             const AttnBwdSide side1) {
    const AttnBwdSide& SD = blockIdx.y ? side1 : side0;
    const float* __restrict__ part = SD.part;
    const rbr_attn_grads g = SD.g;
    const int n_out = attn_n_out(H, A);
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    float t = 0.f;
    int c = 0;
    for (; c + 8 <= n_chunks; c += 8) {        // 8 independent loads in flight, summed in chunk order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long)(c + u) * n_out + o];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; c < n_chunks; ++c) t += part[(long)c * n_out + o];
    if (o < H * A) g.dW_rv[o] = t;
    else if (o < H * A + A * A) g.dW_id[o - H * A] = t;
    else if (o < H * A + A * A + A) g.db1[o - H * A - A * A] = t;
    else if (o < H * A + A * A + 2 * A) g.dh[o - H * A - A * A - A] = t;
    else g.db2[0] = t;
}

}  // namespace rbr

using namespace rbr;

static bool attn_args_ok(int B, int R, int H, int A) {
    if (B <= 0 || R <= 0 || H <= 0 || A <= 0) { set_error("bad attention shape B=%d R=%d H=%d A=%d", B, R, H, A); return false; }
    if ((size_t)(2 * R * A + R + R * H + H * A + A * A) * sizeof(float) > 64 * 1024 ||
        (size_t)kRedRows * (3 * A + H + 1) * sizeof(float) > 64 * 1024) {
        set_error("R=%d H=%d A=%d too large for the LDS tiles", R, H, A);
        return false;
    }
    return true;
}

static size_t attn_fwd_lds(int R, int H, int A) { return (size_t)(2 * R * A + R + R * H + H * A + A * A) * sizeof(float); }

extern "C" int rbr_review_attn_fwd(int32_t B, int32_t R, int32_t H, int32_t A, const float* feat, const int64_t* other_id,
                                   const rbr_attn_params* p, const float* drop, float* out, float* att, float* hid, void* stream) {
    if (!attn_args_ok(B, R, H, A)) return RBR_ERR_BAD_ARG;
    if (!feat || !other_id || !p || !out || !att || !hid) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    const AttnFwdSide S{feat, reinterpret_cast<const long long*>(other_id), *p, out, att, hid, drop};
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B, 1), dim3(256), attn_fwd_lds(R, H, A), (hipStream_t)stream, B, R, H, A, S, S);
    RBR_CHECK_LAUNCH("review_attn_fwd launch");
    return 0;
}

// Both attention pools of a two-tower model in one launch: every per-side tensor is a stacked block with side 0 first --
// feat [2,B,R,H], other_id [2,B,R], drop [2,B,H] (or NULL), out [2,B,H], att [2,B,R], hid [2,B,R,A]; p0 / p1 the sides' parameters.
extern "C" int rbr_review_attn2_fwd(int32_t B, int32_t R, int32_t H, int32_t A, const float* feat, const int64_t* other_id,
                                    const rbr_attn_params* p0, const rbr_attn_params* p1, const float* drop, float* out, float* att,
                                    float* hid, void* stream) {
    if (!attn_args_ok(B, R, H, A)) return RBR_ERR_BAD_ARG;
    if (!feat || !other_id || !p0 || !p1 || !out || !att || !hid) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    const long long* oid = reinterpret_cast<const long long*>(other_id);
    const size_t nf = (size_t)B * R * H, ni = (size_t)B * R, no = (size_t)B * H, nh = (size_t)B * R * A;
    const AttnFwdSide S0{feat, oid, *p0, out, att, hid, drop};
    const AttnFwdSide S1{feat + nf, oid + ni, *p1, out + no, att + ni, hid + nh, drop ? drop + no : nullptr};
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B, 2), dim3(256), attn_fwd_lds(R, H, A), (hipStream_t)stream, B, R, H, A, S0, S1);
    RBR_CHECK_LAUNCH("review_attn2_fwd launch");
    return 0;
}

extern "C" size_t rbr_review_attn_bwd_ws_floats(int32_t B, int32_t R, int32_t H, int32_t A) {
    if (B <= 0 || R <= 0 || H <= 0 || A <= 0) return 0;
    const size_t chunks = ((size_t)B * R + kRedRows - 1) / kRedRows;
    return (size_t)B * R * A + (size_t)B * R + chunks * ((size_t)H * A + (size_t)A * A + 2 * (size_t)A + 1);
}

// the three stages of the backward for `sides` (1 or 2) sides
static int attn_bwd_launch(int B, int R, int H, int A, const AttnBwdSide& S0, const AttnBwdSide& S1, int sides, hipStream_t st) {
    hipLaunchKernelGGL(attn_bwd_sample_kernel, dim3(B, sides), dim3(256), (size_t)(R * A + 2 * R) * sizeof(float), st, B, R, H, A, S0, S1);
    RBR_CHECK_LAUNCH("review_attn_bwd sample launch");
    const int N = B * R, chunks = (N + kRedRows - 1) / kRedRows, n_out = H * A + A * A + 2 * A + 1;
    hipLaunchKernelGGL(attn_bwd_partial_kernel, dim3(chunks, sides), dim3(256), (size_t)kRedRows * (3 * A + H + 1) * sizeof(float), st, N,
                       H, A, S0, S1);
    RBR_CHECK_LAUNCH("review_attn_bwd partial launch");
    hipLaunchKernelGGL(attn_bwd_final_kernel, dim3((n_out + 255) / 256, sides), dim3(256), 0, st, chunks, H, A, S0, S1);
    RBR_CHECK_LAUNCH("review_attn_bwd final launch");
    return 0;
}

static AttnBwdSide attn_bwd_side(int B, int R, int A, const float* feat, const long long* oid, const rbr_attn_params& p, const float* att,
                                 const float* hid, const float* d_out, const float* drop, const float* d_att, int pad_idx,
                                 const rbr_attn_grads& g, float* d_feat, float* ws) {
    AttnBwdSide S{};
    S.feat = feat; S.oid = oid; S.p = p; S.att = att; S.hid = hid; S.d_out = d_out; S.drop = drop; S.d_att = d_att;
    S.pad_idx = pad_idx; S.debd = g.debd; S.d_feat = d_feat; S.g = g;
    S.ws_dpre = ws; S.ws_dl = ws + (size_t)B * R * A; S.part = S.ws_dl + (size_t)B * R;
    return S;
}

extern "C" int rbr_review_attn_bwd(int32_t B, int32_t R, int32_t H, int32_t A, const float* feat, const int64_t* other_id,
                                   const rbr_attn_params* p, const float* drop, const float* att, const float* hid,
                                   const float* d_out, const float* d_att, int32_t pad_idx, const rbr_attn_grads* g, float* d_feat,
                                   float* ws, void* stream) {
    if (!attn_args_ok(B, R, H, A)) return RBR_ERR_BAD_ARG;
    if (!feat || !other_id || !p || !att || !hid || !d_out || !g || !d_feat || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    const AttnBwdSide S = attn_bwd_side(B, R, A, feat, reinterpret_cast<const long long*>(other_id), *p, att, hid, d_out, drop, d_att,
                                        pad_idx, *g, d_feat, ws);
    return attn_bwd_launch(B, R, H, A, S, S, 1, (hipStream_t)stream);
}

// Backward of rbr_review_attn2_fwd (stacked blocks as there; d_out [2,B,H], d_att [2,B,R] or NULL, d_feat [2,B,R,H]).  The
// embedding-row gradients g0->debd / g1->debd ([n0, A] / [n1, A] rows) are ZEROED here and then accumulated; every other
// gradient is overwritten.  ws: 2 * rbr_review_attn_bwd_ws_floats(B, R, H, A) floats.
extern "C" int rbr_review_attn2_bwd(int32_t B, int32_t R, int32_t H, int32_t A, const float* feat, const int64_t* other_id,
                                    const rbr_attn_params* p0, const rbr_attn_params* p1, const float* drop, const float* att,
                                    const float* hid, const float* d_out, const float* d_att, int32_t pad_idx0, int32_t pad_idx1,
                                    const rbr_attn_grads* g0, const rbr_attn_grads* g1, int64_t n_ebd0, int64_t n_ebd1,
                                    float* d_feat, float* ws, void* stream) {
    if (!attn_args_ok(B, R, H, A)) return RBR_ERR_BAD_ARG;
    if (!feat || !other_id || !p0 || !p1 || !att || !hid || !d_out || !g0 || !g1 || !d_feat || !ws || n_ebd0 <= 0 || n_ebd1 <= 0) {
        set_error("null pointer");
        return RBR_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long* oid = reinterpret_cast<const long long*>(other_id);
    const size_t nf = (size_t)B * R * H, ni = (size_t)B * R, no = (size_t)B * H, nh = (size_t)B * R * A;
    const size_t wsn = rbr_review_attn_bwd_ws_floats(B, R, H, A);
    ZeroRegions z{{reinterpret_cast<int*>(g0->debd), reinterpret_cast<int*>(g1->debd), nullptr}, {(long)n_ebd0 * A, (long)n_ebd1 * A, 0}};
    if (int e = zero_regions(z, st)) return e;
    const AttnBwdSide S0 = attn_bwd_side(B, R, A, feat, oid, *p0, att, hid, d_out, drop, d_att, pad_idx0, *g0, d_feat, ws);
    const AttnBwdSide S1 = attn_bwd_side(B, R, A, feat + nf, oid + ni, *p1, att + ni, hid + nh, d_out + no, drop ? drop + no : nullptr,
                                         d_att ? d_att + ni : nullptr, pad_idx1, *g1, d_feat + nf, ws + wsn);
    return attn_bwd_launch(B, R, H, A, S0, S1, 2, st);
}
