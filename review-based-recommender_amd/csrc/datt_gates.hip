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

#include <algorithm>

namespace rbr {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------------- local gate
// One workgroup per (document, tile of 64 tokens): the 64 + win - 1 token rows are staged ONCE in LDS (each row is
// needed by `win` neighbouring tokens), then every wave reduces 16 tokens: lanes stride the embedding dim, win taps
// per lane, shuffle reduction.  w is the Conv1d weight [1, E, win].
constexpr int kGTile = 64;

__global__ __launch_bounds__(256) void local_gate_fwd_kernel(int B, int L, int E, int win, int ntile,
                                                             const long long* __restrict__ ids,
                                                             const float* __restrict__ table, const float* __restrict__ w,
                                                             const float* __restrict__ b0, float* __restrict__ gate) {
    extern __shared__ float s_rows[];                  // [kGTile + win - 1][E]
    __shared__ long s_off[kGTile + 2 * kMaxKF + 2];
    const int b = blockIdx.x / ntile, l0 = (blockIdx.x % ntile) * kGTile;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pad = (win - 1) / 2, R = kGTile + win - 1;
    for (int r = tid; r < R; r += 256) {
        const int p = l0 - pad + r;
        s_off[r] = (p >= 0 && p < L) ? ids[(long)b * L + p] * (long)E : -1;     // zero padding outside the document
    }
    __syncthreads();
    for (int idx = tid; idx < R * E; idx += 256) {
        const int r = idx / E, e = idx - r * E;
        const long o = s_off[r];
        s_rows[idx] = (o >= 0) ? table[o + e] : 0.f;
    }
    __syncthreads();
    const float bias = b0[0];
    for (int t = wave; t < kGTile; t += 4) {
        if (l0 + t >= L) break;
        float s = 0.f;
        for (int e = lane; e < E; e += 64)
            for (int j = 0; j < win; ++j) s = fmaf(s_rows[(t + j) * E + e], w[(long)e * win + j], s);
        s = wsum(s);
        if (lane == 0) gate[(long)b * L + l0 + t] = sigmoidf_(s + bias);
    }
}

// ---- token-folded embedding gradient of a gate -----------------------------------------------------------
// dtable[id, :] += sum over the positions p of the window that carry token `id` of  row_p[:], with
//   local : row_p[e] = sum_j dpre[b, p - j + pad] * w[e, j]   ==>  sum_p row_p[e] = sum_j c_j * w[e, j],
//           c_j = sum_p dpre[b, p - j + pad]                        (win scalars per token)
//   global: row_p[e] = dpre[b] * w[e, p]                        ==>  dpre[b] * sum_p wT[p, e]
// One workgroup per (document, window of <= 256 positions): positions are bucketed by token through an LDS
// hash table, one wave per distinct token builds the folded row and issues ONE row of contiguous f32 atomics
// (a per-token atomic row for each of the 1 M tokens of cfg4 ran 1.3 ms: Zipf-hot rows serialise).
constexpr int kGWin = 256, kGHash = 1024;

__global__ __launch_bounds__(256) void gate_bwd_dx_kernel(int B, int L, int E, int win, int is_global, int nwin,
                                                          const long long* __restrict__ ids, const float* __restrict__ w,
                                                          const float* __restrict__ wT, const float* __restrict__ gate,
                                                          const float* __restrict__ dgate, const float* __restrict__ dpre_g,
                                                          int pad_idx, float* __restrict__ dtable) {
    __shared__ int s_tok[kGWin];
    __shared__ short s_leader[kGWin], s_cnt[kGWin], s_start[kGWin], s_sorted[kGWin];
    __shared__ int s_fill[kGWin];
    __shared__ int s_hkey[kGHash], s_hval[kGHash];
    __shared__ float s_dpre[kGWin + 2 * kMaxKF];     // local: dpre of the window plus a halo of `pad` each side
    const int b = blockIdx.x / nwin, p0 = (blockIdx.x % nwin) * kGWin;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pad = (win - 1) / 2;
    const int nrow = min(kGWin, L - p0);

    for (int k = tid; k < kGHash; k += 256) { s_hkey[k] = -1; s_hval[k] = kGWin; }
    for (int r = tid; r < kGWin; r += 256) {
        int t = -1;
        if (r < nrow) { t = (int)ids[(long)b * L + p0 + r]; if (t == pad_idx) t = -1; }
        s_tok[r] = t;
        s_fill[r] = 0;
    }
    if (!is_global) {
        for (int k = tid; k < kGWin + 2 * pad; k += 256) {
            const int l = p0 - pad + k;
            float v = 0.f;
            if (l >= 0 && l < L) { const float gv = gate[(long)b * L + l]; v = dgate[(long)b * L + l] * gv * (1.f - gv); }
            s_dpre[k] = v;
        }
    }
    __syncthreads();
    for (int r = tid; r < kGWin; r += 256) {
        const int t = s_tok[r];
        int slot = -1;
        if (t >= 0) {
            unsigned hh = ((unsigned)t * 2654435761u) >> 22;
            for (;;) {
                const int old = atomicCAS(&s_hkey[hh], -1, t);
                if (old == -1 || old == t) break;
                hh = (hh + 1) & (kGHash - 1);
            }
            atomicMin(&s_hval[hh], r);
            slot = (int)hh;
        }
        s_leader[r] = (short)slot;
    }
    __syncthreads();
    for (int r = tid; r < kGWin; r += 256) {
        const int slot = s_leader[r];
        const int lead = slot >= 0 ? s_hval[slot] : -1;
        s_leader[r] = (short)lead;
        if (lead >= 0) atomicAdd(&s_fill[lead], 1);
    }
    __syncthreads();
    if (wave == 0) {   // exclusive scan of the per-token counts
        int run = 0;
        for (int r0 = 0; r0 < kGWin; r0 += 64) {
            const int r = r0 + lane;
            const int c = s_fill[r];
            int inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o);
                if (lane >= o) inc += v;
            }
            s_start[r] = (short)(run + inc - c);
            s_cnt[r] = (short)c;
            s_fill[r] = 0;
            run += __shfl(inc, 63);
        }
    }
    __syncthreads();
    for (int r = tid; r < kGWin; r += 256) {
        const int lead = s_leader[r];
        if (lead >= 0) s_sorted[s_start[lead] + atomicAdd(&s_fill[lead], 1)] = (short)r;
    }
    __syncthreads();
    const float dp = is_global ? dpre_g[b] : 0.f;
    for (int key = wave; key < kGWin; key += 4) {
        const int cnt = s_cnt[key];
        if (cnt == 0) continue;                      // wave-uniform
        const int base = s_start[key];
        float* drow = dtable + (long)s_tok[key] * E;
        if (is_global) {
            for (int e = lane; e < E; e += 64) {
                float s = 0.f;
                for (int q = 0; q < cnt; ++q) s += wT[(long)(p0 + s_sorted[base + q]) * E + e];
                atomicAdd(drow + e, dp * s);
            }
        } else {
            for (int j0 = 0; j0 < win; j0 += kMaxKF) {
                float c[kMaxKF];
#pragma unroll
                for (int j = 0; j < kMaxKF; ++j) {
                    c[j] = 0.f;
                    if (j0 + j < win)
                        for (int q = 0; q < cnt; ++q) c[j] += s_dpre[s_sorted[base + q] - (j0 + j) + 2 * pad];
                }
                for (int e = lane; e < E; e += 64) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < kMaxKF; ++j)
                        if (j0 + j < win) s = fmaf(c[j], w[(long)e * win + j0 + j], s);
                    atomicAdd(drow + e, s);
                }
            }
        }
    }
}

// wT[l, e] = w[e, l]   (global gate: coalesced reads of the per-position weight columns)
__global__ __launch_bounds__(256) void transpose_w_kernel(int E, int L, const float* __restrict__ w, float* __restrict__ wT) {
    const long n = (long)E * L;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) {
        const int l = (int)(k / E), e = (int)(k % E);
        wT[k] = w[(long)e * L + l];
    }
}

// per-document partial of dw[e, j] = sum_p x[b,p,e] * dpre[b, p - j + pad] and of db0.  The document is walked in
// tiles of 64 tokens staged in LDS; thread (e, half) owns column e for 32 tokens of each tile and keeps the win
// partial sums in registers; the two halves meet in LDS at the end.
__global__ __launch_bounds__(256) void local_gate_bwd_dw_kernel(int B, int L, int E, int win,
                                                                const long long* __restrict__ ids, const float* __restrict__ table,
                                                                const float* __restrict__ gate, const float* __restrict__ dgate,
                                                                float* __restrict__ ws /* [B][win*E + 1] */) {
    extern __shared__ float sm[];                      // dpre [L + 2*pad] | rows [kGTile][E]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int pad = (win - 1) / 2;
    float* s_dpre = sm;                                // index l + pad; zero halo
    float* s_rows = sm + L + 2 * pad;
    __shared__ long s_off[kGTile];
    __shared__ float s_red[256];
    float part = 0.f;
    for (int k = tid; k < L + 2 * pad; k += 256) {
        const int l = k - pad;
        float v = 0.f;
        if (l >= 0 && l < L) {
            const float gv = gate[(long)b * L + l];
            v = dgate[(long)b * L + l] * gv * (1.f - gv);
            part += v;
        }
        s_dpre[k] = v;
    }
    s_red[tid] = part;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int k = 0; k < 256; ++k) t += s_red[k];
        ws[(long)b * (win * E + 1) + win * E] = t;
    }
    const int half = tid >> 7, e0 = tid & 127;          // two halves x 128 column lanes
    for (int ebase = 0; ebase < E; ebase += 128) {
        const int e = ebase + e0;
        for (int j0 = 0; j0 < win; j0 += kMaxKF) {
            float acc[kMaxKF];
#pragma unroll
            for (int j = 0; j < kMaxKF; ++j) acc[j] = 0.f;
            for (int l0 = 0; l0 < L; l0 += kGTile) {
                __syncthreads();
                if (tid < kGTile) s_off[tid] = (l0 + tid < L) ? ids[(long)b * L + l0 + tid] * (long)E : -1;
                __syncthreads();
                for (int idx = tid; idx < kGTile * E; idx += 256) {
                    const int r = idx / E, ee = idx - r * E;
                    const long o = s_off[r];
                    s_rows[idx] = (o >= 0) ? table[o + ee] : 0.f;
                }
                __syncthreads();
                if (e < E) {
                    for (int t = half * 32; t < half * 32 + 32; ++t) {
                        const float x = s_rows[t * E + e];
                        const int p = l0 + t;
#pragma unroll
                        for (int j = 0; j < kMaxKF; ++j)
                            if (j0 + j < win) acc[j] = fmaf(x, s_dpre[p - (j0 + j) + 2 * pad], acc[j]);
                    }
                }
            }
            // combine the two halves (fixed order) and store
            __syncthreads();
            float* s_acc = s_rows;                                // reuse: [2][kMaxKF][128]
#pragma unroll
            for (int j = 0; j < kMaxKF; ++j) s_acc[(half * kMaxKF + j) * 128 + e0] = acc[j];
            __syncthreads();
            if (half == 0 && e < E) {
#pragma unroll
                for (int j = 0; j < kMaxKF; ++j)
                    if (j0 + j < win)
                        ws[(long)b * (win * E + 1) + (long)(j0 + j) * E + e] = s_acc[j * 128 + e0] + s_acc[(kMaxKF + j) * 128 + e0];
            }
        }
    }
}

// dw[e*win + j] = sum_b ws[b][j*E + e];  db0 = sum_b ws[b][win*E]   (fixed partition and order).
// One workgroup per 32 outputs: 8 thread groups stride the documents.
__global__ __launch_bounds__(256) void local_gate_bwd_reduce_kernel(int B, int E, int win, const float* __restrict__ ws,
                                                                    float* __restrict__ dw, float* __restrict__ db0) {
    __shared__ float red[8][32];
    const int n = win * E + 1;
    const int kk = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + kk;
    float s = 0.f;
    if (idx < n)
        for (int b = grp; b < B; b += 8) s += ws[(long)b * n + idx];
    red[grp][kk] = s;
    __syncthreads();
    if (grp == 0 && idx < n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][kk];
        if (idx == win * E) db0[0] = t;
        else { const int j = idx / E, e = idx - j * E; dw[(long)e * win + j] = t; }
    }
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
    if (win > 2 * kMaxKF + 1) { set_error("local attention window %d exceeds the supported %d", win, 2 * kMaxKF + 1); return RBR_ERR_UNSUPPORTED; }
    const size_t smem = (size_t)(kGTile + win - 1) * E * sizeof(float);
    if (smem > 120 * 1024) { set_error("embedding dim %d too large for the local-gate LDS tile", E); return RBR_ERR_UNSUPPORTED; }
    const int ntile = (L + kGTile - 1) / kGTile;
    hipLaunchKernelGGL(local_gate_fwd_kernel, dim3((unsigned)(B * ntile)), dim3(256), smem, (hipStream_t)stream, B, L, E, win,
                       ntile, reinterpret_cast<const long long*>(ids), table, w, b0, gate);
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
    return is_global ? (size_t)B + (size_t)E * L : (size_t)B * ((size_t)win * E + 1);
}

extern "C" int rbr_datt_local_gate_bwd(int32_t B, int32_t L, int32_t E, int32_t win, const int64_t* ids, const float* table,
                                       const float* w, const float* gate, const float* dgate, int32_t pad_idx, float* dw,
                                       float* db0, float* dtable, float* ws, void* stream) {
    if (!gate_args_ok(B, L, E, win)) return RBR_ERR_BAD_ARG;
    if (!ids || !table || !w || !gate || !dgate || !dw || !db0 || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if ((size_t)L * sizeof(float) > 60 * 1024) { set_error("doc_len %d too large for the LDS strip", L); return RBR_ERR_UNSUPPORTED; }
    if (win > 2 * kMaxKF + 1) { set_error("local attention window %d exceeds the supported %d", win, 2 * kMaxKF + 1); return RBR_ERR_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    const long long* ids64 = reinterpret_cast<const long long*>(ids);
    const size_t smem_dw = (size_t)(L + 2 * ((win - 1) / 2) + std::max(kGTile * E, 2 * kMaxKF * 128)) * sizeof(float);
    if (smem_dw > 140 * 1024) { set_error("doc_len %d x emb %d too large for the local-gate LDS tiles", L, E); return RBR_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(local_gate_bwd_dw_kernel, dim3(B), dim3(256), smem_dw, st, B, L, E, win, ids64, table, gate, dgate, ws);
    RBR_CHECK_LAUNCH("datt local gate bwd dw launch");
    hipLaunchKernelGGL(local_gate_bwd_reduce_kernel, dim3((unsigned)((win * E + 1 + 31) / 32)), dim3(256), 0, st, B, E, win,
                       ws, dw, db0);
    RBR_CHECK_LAUNCH("datt local gate bwd reduce launch");
    if (dtable != nullptr) {
        const int nwin = (L + kGWin - 1) / kGWin;
        hipLaunchKernelGGL(gate_bwd_dx_kernel, dim3((unsigned)(B * nwin)), dim3(256), 0, st, B, L, E, win, 0, nwin, ids64, w,
                           (const float*)nullptr, gate, dgate, (const float*)nullptr, pad_idx, dtable);
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
        float* wT = ws + B;
        hipLaunchKernelGGL(transpose_w_kernel, dim3((unsigned)std::min<long>(((long)E * L + 255) / 256, 2048)), dim3(256), 0,
                           st, E, L, w, wT);
        RBR_CHECK_LAUNCH("datt global gate transpose launch");
        const int nwin = (L + kGWin - 1) / kGWin;
        hipLaunchKernelGGL(gate_bwd_dx_kernel, dim3((unsigned)(B * nwin)), dim3(256), 0, st, B, L, E, 1, 1, nwin, ids64, w,
                           (const float*)wT, gate, dgate, (const float*)ws, pad_idx, dtable);
        RBR_CHECK_LAUNCH("datt global gate bwd dx launch");
    }
    return 0;
}
