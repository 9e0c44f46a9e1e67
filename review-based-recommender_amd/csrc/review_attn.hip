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

__global__ __launch_bounds__(256) void attn_fwd_kernel(int B, int R, int H, int A, const float* __restrict__ feat,
                                                       const long long* __restrict__ oid, const rbr_attn_params p,
                                                       float* __restrict__ out, float* __restrict__ att,
                                                       float* __restrict__ hid) {
    extern __shared__ float sm[];
    float* s_hid = sm;            // [R*A]
    float* s_e = sm + R * A;      // [R] exp(logit), then att
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* fb = feat + (long)b * R * H;
    for (int idx = tid; idx < R * A; idx += 256) {
        const int r = idx / A, a = idx - r * A;
        const long id = oid[(long)b * R + r];
        float acc = p.b1[a];
        for (int a2 = 0; a2 < A; ++a2) acc = fmaf(p.ebd[id * A + a2], p.W_id[(long)a2 * A + a], acc);
        for (int hh = 0; hh < H; ++hh) acc = fmaf(fb[(long)r * H + hh], p.W_rv[(long)hh * A + a], acc);
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
        for (int r = 0; r < R; ++r) s = fmaf(s_e[r], fb[(long)r * H + hh], s);
        out[(long)b * H + hh] = s;
    }
}

// per-sample backward: d_feat, embedding-row gradient, and the per-row d_pre / d_logit the reduction needs
__global__ __launch_bounds__(256) void attn_bwd_sample_kernel(int B, int R, int H, int A, const float* __restrict__ feat,
                                                              const long long* __restrict__ oid, const rbr_attn_params p,
                                                              const float* __restrict__ att, const float* __restrict__ hid,
                                                              const float* __restrict__ d_out,
                                                              const float* __restrict__ d_att, int pad_idx,
                                                              float* __restrict__ debd, float* __restrict__ d_feat,
                                                              float* __restrict__ ws_dpre, float* __restrict__ ws_dl) {
    extern __shared__ float sm[];
    float* s_dpre = sm;            // [R*A]
    float* s_da = sm + R * A;      // [R] d(att)
    float* s_dl = s_da + R;        // [R] d(logit)
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* fb = feat + (long)b * R * H;
    const float* ab = att + (long)b * R;
    for (int r = tid; r < R; r += 256) {
        float s = (d_att != nullptr) ? d_att[(long)b * R + r] : 0.f;
        for (int hh = 0; hh < H; ++hh) s = fmaf(d_out[(long)b * H + hh], fb[(long)r * H + hh], s);
        s_da[r] = s;
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
    for (int idx = tid; idx < R * H; idx += 256) {
        const int r = idx / H, hh = idx - r * H;
        float s = ab[r] * d_out[(long)b * H + hh];
        for (int a = 0; a < A; ++a) s = fmaf(s_dpre[r * A + a], p.W_rv[(long)hh * A + a], s);
        d_feat[((long)b * R + r) * H + hh] = s;
    }
    for (int idx = tid; idx < R * A; idx += 256) {
        const int r = idx / A, a2 = idx - r * A;
        const long id = oid[(long)b * R + r];
        if (id == pad_idx) continue;                 // nn.Embedding(padding_idx): no gradient for the pad row
        float s = 0.f;
        for (int a = 0; a < A; ++a) s = fmaf(s_dpre[r * A + a], p.W_id[(long)a2 * A + a], s);
        atomicAdd(debd + id * A + a2, s);
    }
}

// reductions over the N = B*R rows, fixed partition and order (bitwise reproducible).
// block row:  [0,H) dW_rv | [H,H+A) dW_id | H+A db1 | H+A+1 dh | H+A+2 db2
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(int N, int H, int A, const float* __restrict__ feat,
                                                              const long long* __restrict__ oid, const float* __restrict__ ebd,
                                                              const float* __restrict__ hid, const float* __restrict__ ws_dpre,
                                                              const float* __restrict__ ws_dl, const rbr_attn_grads g) {
    __shared__ float red[8][32];
    const int row = blockIdx.x;
    const int kk = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (int a0 = 0; a0 < A; a0 += 32) {
        const int a = a0 + kk;
        float s = 0.f;
        if (a < A) {
            if (row < H) {
                for (int n = grp; n < N; n += 8) s = fmaf(feat[(long)n * H + row], ws_dpre[(long)n * A + a], s);
            } else if (row < H + A) {
                const int a2 = row - H;
                for (int n = grp; n < N; n += 8) s = fmaf(ebd[oid[n] * A + a2], ws_dpre[(long)n * A + a], s);
            } else if (row == H + A) {
                for (int n = grp; n < N; n += 8) s += ws_dpre[(long)n * A + a];
            } else if (row == H + A + 1) {
                for (int n = grp; n < N; n += 8) s = fmaf(ws_dl[n], hid[(long)n * A + a], s);
            } else if (a == 0) {
                for (int n = grp; n < N; n += 8) s += ws_dl[n];
            }
        }
        __syncthreads();
        red[grp][kk] = s;
        __syncthreads();
        if (grp == 0 && a < A) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[q][kk];
            if (row < H) g.dW_rv[(long)row * A + a] = t;
            else if (row < H + A) g.dW_id[(long)(row - H) * A + a] = t;
            else if (row == H + A) g.db1[a] = t;
            else if (row == H + A + 1) g.dh[a] = t;
            else if (a == 0) g.db2[0] = t;
        }
    }
}

}  // namespace rbr

using namespace rbr;

static bool attn_args_ok(int B, int R, int H, int A) {
    if (B <= 0 || R <= 0 || H <= 0 || A <= 0) { set_error("bad attention shape B=%d R=%d H=%d A=%d", B, R, H, A); return false; }
    if ((size_t)(R * A + 2 * R) * sizeof(float) > 60 * 1024) { set_error("R*A=%d too large for the LDS tile", R * A); return false; }
    return true;
}

extern "C" int rbr_review_attn_fwd(int32_t B, int32_t R, int32_t H, int32_t A, const float* feat, const int64_t* other_id,
                                   const rbr_attn_params* p, float* out, float* att, float* hid, void* stream) {
    if (!attn_args_ok(B, R, H, A)) return RBR_ERR_BAD_ARG;
    if (!feat || !other_id || !p || !out || !att || !hid) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B), dim3(256), (size_t)(R * A + R) * sizeof(float), (hipStream_t)stream, B, R,
                       H, A, feat, reinterpret_cast<const long long*>(other_id), *p, out, att, hid);
    RBR_CHECK_LAUNCH("review_attn_fwd launch");
    return 0;
}

extern "C" size_t rbr_review_attn_bwd_ws_floats(int32_t B, int32_t R, int32_t H, int32_t A) {
    (void)H;
    return (size_t)B * R * A + (size_t)B * R;
}

extern "C" int rbr_review_attn_bwd(int32_t B, int32_t R, int32_t H, int32_t A, const float* feat, const int64_t* other_id,
                                   const rbr_attn_params* p, const float* att, const float* hid, const float* d_out,
                                   const float* d_att, int32_t pad_idx, const rbr_attn_grads* g, float* d_feat, float* ws,
                                   void* stream) {
    if (!attn_args_ok(B, R, H, A)) return RBR_ERR_BAD_ARG;
    if (!feat || !other_id || !p || !att || !hid || !d_out || !g || !d_feat || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipStream_t st = (hipStream_t)stream;
    float* ws_dpre = ws;
    float* ws_dl = ws + (size_t)B * R * A;
    const long long* oid = reinterpret_cast<const long long*>(other_id);
    hipLaunchKernelGGL(attn_bwd_sample_kernel, dim3(B), dim3(256), (size_t)(R * A + 2 * R) * sizeof(float), st, B, R, H, A,
                       feat, oid, *p, att, hid, d_out, d_att, pad_idx, g->debd, d_feat, ws_dpre, ws_dl);
    RBR_CHECK_LAUNCH("review_attn_bwd sample launch");
    hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3(H + A + 3), dim3(256), 0, st, B * R, H, A, feat, oid, p->ebd, hid, ws_dpre,
                       ws_dl, *g);
    RBR_CHECK_LAUNCH("review_attn_bwd reduce launch");
    return 0;
}
