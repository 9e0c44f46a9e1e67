// datt_gates.hip -- D-ATT attention gates (forward and backward).
//
// Replaces the `attn` branches of models/dual_att/layers.py:
//   LocalAttention  (layers.py:34-36,50): gate[b,l] = sigmoid(conv1d(E -> 1, win, pad (win-1)/2)(x)[b,l])
//   GlobalAttention (layers.py:65-67,84): gate[b,:] = sigmoid(conv1d(E -> 1, kernel = doc_len)(x)[b])
// with x = table[ids] (D-ATT has no masks: pad tokens read the table's pad row).  The products
// `score * x` (layers.py:51,85) are never materialised: the gate vector is handed to the fused conv
// kernel, which scales the token rows while gathering them (csrc/textcnn_fwd.hip, `gate`).
// One output channel => no GEMM shape; these are HBM/L2-bound row dot products on the VALU.
#include "rbr_common.h"

namespace rbr {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------------- local gate
// one wave per token; lanes stride the embedding dim; w is the Conv1d weight [1, E, win]
__global__ __launch_bounds__(256) void local_gate_fwd_kernel(int B, int L, int E, int win, const long long* __restrict__ ids,
                                                             const float* __restrict__ table, const float* __restrict__ w,
                                                             const float* __restrict__ b0, float* __restrict__ gate) {
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tok >= (long)B * L) return;
    const int b = (int)(tok / L), l = (int)(tok % L);
    const int pad = (win - 1) / 2;
    float s = 0.f;
    for (int j = 0; j < win; ++j) {
        const int p = l + j - pad;
        if (p < 0 || p >= L) continue;                     // zero padding
        const float* row = table + ids[(long)b * L + p] * (long)E;
        for (int e = lane; e < E; e += 64) s = fmaf(row[e], w[(long)e * win + j], s);
    }
    s = wsum(s);
    if (lane == 0) gate[tok] = sigmoidf_(s + b0[0]);
}

// dtable[id(b,p), e] += sum_j dpre[b, p - j + pad] * w[e, j]      (one wave per token p)
__global__ __launch_bounds__(256) void local_gate_bwd_dx_kernel(int B, int L, int E, int win,
                                                                const long long* __restrict__ ids, const float* __restrict__ w,
                                                                const float* __restrict__ gate, const float* __restrict__ dgate,
                                                                int pad_idx, float* __restrict__ dtable) {
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tok >= (long)B * L) return;
    const int b = (int)(tok / L), p = (int)(tok % L);
    const long id = ids[tok];
    if (id == pad_idx) return;
    const int pad = (win - 1) / 2;
    for (int e = lane; e < E; e += 64) {
        float s = 0.f;
        for (int j = 0; j < win; ++j) {
            const int l = p - j + pad;
            if (l < 0 || l >= L) continue;
            const float gv = gate[(long)b * L + l];
            s = fmaf(dgate[(long)b * L + l] * gv * (1.f - gv), w[(long)e * win + j], s);
        }
        atomicAdd(dtable + id * E + e, s);
    }
}

// per-document partial of dw[e, j] = sum_p x[b,p,e] * dpre[b, p - j + pad]  and of db0; thread e owns column e
__global__ __launch_bounds__(256) void local_gate_bwd_dw_kernel(int B, int L, int E, int win,
                                                                const long long* __restrict__ ids, const float* __restrict__ table,
                                                                const float* __restrict__ gate, const float* __restrict__ dgate,
                                                                float* __restrict__ ws /* [B][win*E + 1] */) {
    extern __shared__ float s_dpre[];   // [L]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int pad = (win - 1) / 2;
    float part = 0.f;
    for (int l = tid; l < L; l += 256) {
        const float gv = gate[(long)b * L + l];
        const float v = dgate[(long)b * L + l] * gv * (1.f - gv);
        s_dpre[l] = v;
        part += v;
    }
    __shared__ float s_red[256];
    s_red[tid] = part;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int k = 0; k < 256; ++k) t += s_red[k];
        ws[(long)b * (win * E + 1) + win * E] = t;
    }
    for (int e = tid; e < E; e += 256) {
        for (int j0 = 0; j0 < win; j0 += kMaxKF) {
            float acc[kMaxKF];
#pragma unroll
            for (int j = 0; j < kMaxKF; ++j) acc[j] = 0.f;
            for (int p = 0; p < L; ++p) {
                const float x = table[ids[(long)b * L + p] * (long)E + e];
#pragma unroll
                for (int j = 0; j < kMaxKF; ++j) {
                    const int l = p - (j0 + j) + pad;
                    if (j0 + j < win && l >= 0 && l < L) acc[j] = fmaf(x, s_dpre[l], acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < kMaxKF; ++j)
                if (j0 + j < win) ws[(long)b * (win * E + 1) + (long)(j0 + j) * E + e] = acc[j];
        }
    }
}

// dw[e*win + j] = sum_b ws[b][j*E + e];  db0 = sum_b ws[b][win*E]   (fixed order)
__global__ __launch_bounds__(256) void local_gate_bwd_reduce_kernel(int B, int E, int win, const float* __restrict__ ws,
                                                                    float* __restrict__ dw, float* __restrict__ db0) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n = win * E + 1;
    if (idx >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += ws[(long)b * n + idx];
    if (idx == win * E) db0[0] = s;
    else { const int j = idx / E, e = idx - j * E; dw[(long)e * win + j] = s; }
}

// ---------------------------------------------------------------------------------- global gate
// one workgroup per document: thread l walks its token's row; w is the Conv1d weight [1, E, L]
__global__ __launch_bounds__(256) void global_gate_fwd_kernel(int B, int L, int E, const long long* __restrict__ ids,
                                                              const float* __restrict__ table, const float* __restrict__ w,
                                                              const float* __restrict__ b0, float* __restrict__ gate) {
    __shared__ float s_red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float part = 0.f;
    for (int l = tid; l < L; l += 256) {
        const float* row = table + ids[(long)b * L + l] * (long)E;
        float s = 0.f;
        for (int e = 0; e < E; ++e) s = fmaf(row[e], w[(long)e * L + l], s);   // w reads coalesced across threads
        part += s;
    }
    s_red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] += s_red[tid + o];
        __syncthreads();
    }
    const float g = sigmoidf_(s_red[0] + b0[0]);
    for (int l = tid; l < L; l += 256) gate[(long)b * L + l] = g;   // broadcast: the conv kernel takes a per-token gate
}

// dpre[b] = (sum_l dgate[b,l]) * g (1-g)
__global__ __launch_bounds__(256) void global_gate_bwd_dpre_kernel(int B, int L, const float* __restrict__ gate,
                                                                   const float* __restrict__ dgate, float* __restrict__ dpre) {
    __shared__ float s_red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float part = 0.f;
    for (int l = tid; l < L; l += 256) part += dgate[(long)b * L + l];
    s_red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] += s_red[tid + o];
        __syncthreads();
    }
    if (tid == 0) { const float g = gate[(long)b * L]; dpre[b] = s_red[0] * g * (1.f - g); }
}

// dw[e, l] = sum_b dpre[b] * x[b,l,e]  (one workgroup per position l, fixed order);  block L computes db0
__global__ __launch_bounds__(256) void global_gate_bwd_dw_kernel(int B, int L, int E, const long long* __restrict__ ids,
                                                                 const float* __restrict__ table, const float* __restrict__ dpre,
                                                                 float* __restrict__ dw, float* __restrict__ db0) {
    const int l = blockIdx.x, tid = threadIdx.x;
    if (l == L) {
        if (tid == 0) { float s = 0.f; for (int b = 0; b < B; ++b) s += dpre[b]; db0[0] = s; }
        return;
    }
    for (int e = tid; e < E; e += 256) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(dpre[b], table[ids[(long)b * L + l] * (long)E + e], s);
        dw[(long)e * L + l] = s;
    }
}

// dtable[id(b,l), e] += dpre[b] * w[e, l]      (one wave per token)
__global__ __launch_bounds__(256) void global_gate_bwd_dx_kernel(int B, int L, int E, const long long* __restrict__ ids,
                                                                 const float* __restrict__ w, const float* __restrict__ dpre,
                                                                 int pad_idx, float* __restrict__ dtable) {
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tok >= (long)B * L) return;
    const int b = (int)(tok / L), l = (int)(tok % L);
    const long id = ids[tok];
    if (id == pad_idx) return;
    const float dp = dpre[b];
    for (int e = lane; e < E; e += 64) atomicAdd(dtable + id * E + e, dp * w[(long)e * L + l]);
}

}  // namespace rbr

using namespace rbr;

static bool gate_args_ok(int B, int L, int E, int win) {
    if (B <= 0 || L <= 0 || E <= 0 || win <= 0) { set_error("bad gate shape B=%d L=%d E=%d win=%d", B, L, E, win); return false; }
    return true;
}

extern "C" int rbr_datt_local_gate_fwd(int32_t B, int32_t L, int32_t E, int32_t win, const int64_t* ids, const float* table,
                                       const float* w, const float* b0, float* gate, void* stream) {
    if (!gate_args_ok(B, L, E, win)) return RBR_ERR_BAD_ARG;
    if (win % 2 == 0) { set_error("local attention window must be odd, got %d", win); return RBR_ERR_BAD_ARG; }
    if (!ids || !table || !w || !b0 || !gate) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    const long ntok = (long)B * L;
    hipLaunchKernelGGL(local_gate_fwd_kernel, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, (hipStream_t)stream, B, L, E,
                       win, reinterpret_cast<const long long*>(ids), table, w, b0, gate);
    RBR_CHECK_LAUNCH("datt local gate fwd launch");
    return 0;
}

extern "C" int rbr_datt_global_gate_fwd(int32_t B, int32_t L, int32_t E, const int64_t* ids, const float* table,
                                        const float* w, const float* b0, float* gate, void* stream) {
    if (!gate_args_ok(B, L, E, 1)) return RBR_ERR_BAD_ARG;
    if (!ids || !table || !w || !b0 || !gate) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(global_gate_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, B, L, E,
                       reinterpret_cast<const long long*>(ids), table, w, b0, gate);
    RBR_CHECK_LAUNCH("datt global gate fwd launch");
    return 0;
}

extern "C" size_t rbr_datt_gate_bwd_ws_floats(int32_t B, int32_t L, int32_t E, int32_t win, int32_t is_global) {
    (void)L;
    return is_global ? (size_t)B : (size_t)B * ((size_t)win * E + 1);
}

extern "C" int rbr_datt_local_gate_bwd(int32_t B, int32_t L, int32_t E, int32_t win, const int64_t* ids, const float* table,
                                       const float* w, const float* gate, const float* dgate, int32_t pad_idx, float* dw,
                                       float* db0, float* dtable, float* ws, void* stream) {
    if (!gate_args_ok(B, L, E, win)) return RBR_ERR_BAD_ARG;
    if (!ids || !table || !w || !gate || !dgate || !dw || !db0 || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if ((size_t)L * sizeof(float) > 60 * 1024) { set_error("doc_len %d too large for the LDS strip", L); return RBR_ERR_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    const long long* ids64 = reinterpret_cast<const long long*>(ids);
    hipLaunchKernelGGL(local_gate_bwd_dw_kernel, dim3(B), dim3(256), (size_t)L * sizeof(float), st, B, L, E, win, ids64, table,
                       gate, dgate, ws);
    RBR_CHECK_LAUNCH("datt local gate bwd dw launch");
    hipLaunchKernelGGL(local_gate_bwd_reduce_kernel, dim3((unsigned)((win * E + 1 + 255) / 256)), dim3(256), 0, st, B, E, win,
                       ws, dw, db0);
    RBR_CHECK_LAUNCH("datt local gate bwd reduce launch");
    if (dtable != nullptr) {
        const long ntok = (long)B * L;
        hipLaunchKernelGGL(local_gate_bwd_dx_kernel, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, st, B, L, E, win, ids64,
                           w, gate, dgate, pad_idx, dtable);
        RBR_CHECK_LAUNCH("datt local gate bwd dx launch");
    }
    return 0;
}

extern "C" int rbr_datt_global_gate_bwd(int32_t B, int32_t L, int32_t E, const int64_t* ids, const float* table,
                                        const float* w, const float* gate, const float* dgate, int32_t pad_idx, float* dw,
                                        float* db0, float* dtable, float* ws, void* stream) {
    if (!gate_args_ok(B, L, E, 1)) return RBR_ERR_BAD_ARG;
    if (!ids || !table || !w || !gate || !dgate || !dw || !db0 || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const long long* ids64 = reinterpret_cast<const long long*>(ids);
    hipLaunchKernelGGL(global_gate_bwd_dpre_kernel, dim3(B), dim3(256), 0, st, B, L, gate, dgate, ws);
    RBR_CHECK_LAUNCH("datt global gate bwd dpre launch");
    hipLaunchKernelGGL(global_gate_bwd_dw_kernel, dim3(L + 1), dim3(256), 0, st, B, L, E, ids64, table, ws, dw, db0);
    RBR_CHECK_LAUNCH("datt global gate bwd dw launch");
    if (dtable != nullptr) {
        const long ntok = (long)B * L;
        hipLaunchKernelGGL(global_gate_bwd_dx_kernel, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, st, B, L, E, ids64, w,
                           ws, pad_idx, dtable);
        RBR_CHECK_LAUNCH("datt global gate bwd dx launch");
    }
    return 0;
}
