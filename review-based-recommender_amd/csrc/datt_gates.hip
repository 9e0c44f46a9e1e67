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
typedef float f32x4g __attribute__((ext_vector_type(4)));

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
constexpr int kGateWP = 8;       // floats per token row of the tap tables S / c of the token-product local gate (win <= 8)
constexpr int kGateCopies = 16;  // private copies of the tap-sum table c (summed by gp_csum_kernel)

__device__ __forceinline__ void gate_bwd_dx_kernel(int B, int L, int E, int win, int is_global, int nwin,
                                                          const long long* __restrict__ ids, const float* __restrict__ w,
                                                          const float* __restrict__ wT, const float* __restrict__ gate,
                                                          const float* __restrict__ dgate, const float* __restrict__ dpre_g,
                                                          int pad_idx, float* __restrict__ dtable,
                                                          const int* __restrict__ row_of_token, float* __restrict__ c_out,
                                                          long c_stride) {
    __shared__ int s_tok[kGWin];
    __shared__ short s_leader[kGWin], s_cnt[kGWin], s_start[kGWin], s_sorted[kGWin];
    __shared__ int s_fill[kGWin];
    __shared__ int s_hkey[kGHash], s_hval[kGHash];
    __shared__ float s_dpre[kGWin + 2 * kMaxKF];     // local: dpre of the window plus a halo of `pad` each side
    __shared__ short s_keys[kGWin];                  // the leaders (first position of each distinct token), dense
    __shared__ int s_nkeys;
    extern __shared__ float s_cj[];                  // dense local mode: [kGWin][win] tap sums per distinct token
    const int b = blockIdx.x / nwin, p0 = (blockIdx.x % nwin) * kGWin;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pad = (win - 1) / 2;
    const int nrow = min(kGWin, L - p0);

    for (int k = tid; k < kGHash; k += 256) { s_hkey[k] = -1; s_hval[k] = kGWin; }
    for (int r = tid; r < kGWin; r += 256) {
        int t = -1;
        // c_out mode keeps the pad token: its row feeds the weight gradient (only its TABLE row gets no gradient)
        if (r < nrow) { t = (int)ids[(long)b * L + p0 + r]; if (t == pad_idx && c_out == nullptr) t = -1; }
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
    if (wave == 0) {   // exclusive scan of the per-token counts, and of the leaders themselves (dense key list)
        int run = 0, krun = 0;
        for (int r0 = 0; r0 < kGWin; r0 += 64) {
            const int r = r0 + lane;
            const int c = s_fill[r];
            int inc = c, kinc = c > 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o), kv = __shfl_up(kinc, o);
                if (lane >= o) { inc += v; kinc += kv; }
            }
            s_start[r] = (short)(run + inc - c);
            s_cnt[r] = (short)c;
            s_fill[r] = 0;
            if (c > 0) s_keys[krun + kinc - 1] = (short)r;
            run += __shfl(inc, 63);
            krun += __shfl(kinc, 63);
        }
        if (lane == 0) s_nkeys = krun;
    }
    __syncthreads();
    for (int r = tid; r < kGWin; r += 256) {
        const int lead = s_leader[r];
        if (lead >= 0) s_sorted[s_start[lead] + atomicAdd(&s_fill[lead], 1)] = (short)r;
    }
    __syncthreads();
    // The folded rows leave the workgroup thread-parallel: a wave per token walked <= 64 tokens one after the other, each
    // with a dependent L2 read in front of its atomics (latency-bound: 88 us for the tap sums alone at cfg4).
    const int nkeys = s_nkeys;
    if (c_out != nullptr) {
        // token-product backward of the local gate: only the win tap sums c_j of the token leave the workgroup.  8 lanes
        // per token (one per tap), so that a token's row is ONE 32-byte atomic request; the workgroup adds into copy
        // blockIdx % kGateCopies of the tap-sum table: requests to the same line are served one after the other, and the row
        // of a Zipf-hot token would otherwise receive one from every window of every document.
        float* mine = c_out + (long)(blockIdx.x % kGateCopies) * c_stride;
        const int j = tid & (kGateWP - 1);
        for (int k = tid / kGateWP; k < nkeys; k += 256 / kGateWP) {
            const int key = s_keys[k], cnt = s_cnt[key], base = s_start[key];
            const int row = row_of_token[s_tok[key]];
            if (row < 0 || j >= win) continue;
            float cj = 0.f;
            for (int q = 0; q < cnt; ++q) cj += s_dpre[s_sorted[base + q] - j + 2 * pad];
            atomicAdd(mine + (long)row * kGateWP + j, cj);
        }
        return;
    }
    if (!is_global) {
        // the win tap sums of every token of the window, once, into LDS
        for (int k = tid; k < nkeys; k += 256) {
            const int key = s_keys[k], cnt = s_cnt[key], base = s_start[key];
            for (int j = 0; j < win; ++j) {
                float c = 0.f;
                for (int q = 0; q < cnt; ++q) c += s_dpre[s_sorted[base + q] - j + 2 * pad];
                s_cj[k * win + j] = c;
            }
        }
        __syncthreads();
    }
    const float dp = is_global ? dpre_g[b] : 0.f;
    const int total = nkeys * E;
    constexpr int kU = 4;                            // rows in flight per thread: the reads of 4 items before the first atomic
    for (int it = tid; it < total; it += 256 * kU) {
        float s[kU];
        float* dst[kU];
        int cnt[kU], base[kU], ee[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int item = it + u * 256;
            const bool ok = item < total;
            const int k = ok ? item / E : 0, e = ok ? item - k * E : 0;
            const int key = s_keys[k];
            cnt[u] = ok ? s_cnt[key] : 0;
            base[u] = s_start[key];
            ee[u] = e;
            dst[u] = dtable + (long)s_tok[key] * E + e;
            if (is_global) {
                s[u] = ok ? wT[(long)(p0 + s_sorted[base[u]]) * E + e] : 0.f;
            } else {
                float a = 0.f;
                for (int j = 0; j < win; ++j) a = fmaf(s_cj[k * win + j], w[(long)e * win + j], a);
                s[u] = a;
            }
        }
        if (is_global) {
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                for (int q = 1; q < cnt[u]; ++q) s[u] += wT[(long)(p0 + s_sorted[base[u] + q]) * E + ee[u]];
                s[u] *= dp;
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (cnt[u] > 0) atomicAdd(dst[u], s[u]);
    }
}

// wT[l, e] = w[e, l]   (global gate: coalesced reads of the per-position weight columns)
__device__ __forceinline__ void transpose_w_kernel(int E, int L, const float* __restrict__ w, float* __restrict__ wT) {
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
__device__ __forceinline__ void global_gate_fwd_kernel(int B, int L, int E, const long long* __restrict__ ids,
                                                              const float* __restrict__ table, const float* __restrict__ w,
                                                              const float* __restrict__ b0, float* __restrict__ gate) {
    __shared__ float s_red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool vec = (E % 4 == 0) && ((((uintptr_t)table) & 15) == 0);
    float part = 0.f;
    for (int l = tid; l < L; l += 256) {
        const float* row = table + ids[(long)b * L + l] * (long)E;
        float s = 0.f;
        if (vec) {      // the thread's own row in 16-byte pieces (a quarter of the cache-line touches); w coalesced across threads
            int e = 0;
            for (; e + 16 <= E; e += 16) {      // 4 row quads and their 16 weight scalars requested before the first use
                f32x4g x[4];
                float wv[16];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const f32x4g*>(row + e + 4 * u);
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = w[(long)(e + u) * L + l];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s = fmaf(x[u].x, wv[4 * u], s); s = fmaf(x[u].y, wv[4 * u + 1], s);
                    s = fmaf(x[u].z, wv[4 * u + 2], s); s = fmaf(x[u].w, wv[4 * u + 3], s);
                }
            }
            for (; e < E; e += 4) {
                const f32x4g x = *reinterpret_cast<const f32x4g*>(row + e);
                s = fmaf(x.x, w[(long)e * L + l], s);
                s = fmaf(x.y, w[(long)(e + 1) * L + l], s);
                s = fmaf(x.z, w[(long)(e + 2) * L + l], s);
                s = fmaf(x.w, w[(long)(e + 3) * L + l], s);
            }
        } else {
            for (int e = 0; e < E; ++e) s = fmaf(row[e], w[(long)e * L + l], s);   // w reads coalesced across threads
        }
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
__device__ __forceinline__ void global_gate_bwd_dpre_kernel(int B, int L, const float* __restrict__ gate,
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

// dw[e, l] = sum_b dpre[b] * x[b,l,e]  (one workgroup per position l);  block L computes db0.  The 256 threads split
// into document classes x embedding columns (E = 100: two classes), 8 independent row reads in flight per thread; the
// classes meet in LDS in fixed order.
__device__ __forceinline__ void global_gate_bwd_dw_kernel(int B, int L, int E, const long long* __restrict__ ids,
                                                                 const float* __restrict__ table, const float* __restrict__ dpre,
                                                                 float* __restrict__ dw, float* __restrict__ db0) {
    __shared__ float s_part[256];
    constexpr int kStage = 2048;                      // documents whose row offsets and dpre are staged per pass
    __shared__ long s_off[kStage];
    __shared__ float s_dp[kStage];
    const int l = blockIdx.x, tid = threadIdx.x;
    if (l == L) {
        if (tid == 0) { float s = 0.f; for (int b = 0; b < B; ++b) s += dpre[b]; db0[0] = s; }
        return;
    }
    const int ncls = max(1, 256 / E);                 // document classes that fit beside E columns
    for (int e0 = 0; e0 < E; e0 += 256) {
        const int cls = (E >= 256) ? 0 : tid / E, e = e0 + ((E >= 256) ? tid : tid - cls * E);
        const bool live = cls < ncls && e < E;
        float s = 0.f;
        // the token ids of this position (one per document, a strided column of `ids`) are fetched once per pass into LDS, all
        // at the same time: the row reads below then depend on nothing but LDS (they were a chain id -> row, 8 at a time)
        for (int bs = 0; bs < B; bs += kStage) {
            const int nb = min(kStage, B - bs);
            __syncthreads();
            for (int k = tid; k < nb; k += 256) {
                s_off[k] = ids[(long)(bs + k) * L + l] * (long)E;
                s_dp[k] = dpre[bs + k];
            }
            __syncthreads();
            if (live) {
                for (int b0i = cls; b0i < nb; b0i += 8 * ncls) {
                    float x[8], dp[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int b = b0i + u * ncls;
                        const bool ok = b < nb;
                        dp[u] = ok ? s_dp[b] : 0.f;
                        x[u] = ok ? table[s_off[b] + e] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) s = fmaf(dp[u], x[u], s);
                }
            }
        }
        __syncthreads();
        s_part[tid] = live ? s : 0.f;
        __syncthreads();
        if (live && cls == 0) {
            float t = s_part[tid];
            for (int c2 = 1; c2 < ncls; ++c2) t += s_part[tid + c2 * E];
            dw[(long)e * L + l] = t;
        }
    }
}


// ---------------------------------------------------------------------------------- local gate, token-product form
// The gate's input is an embedding LOOKUP too: pre[b,l] = b0 + sum_j S[ids[b, l+j-pad]][j] with S[t][j] = <table[t,:], w[:,j]>,
// so S is needed once per DISTINCT token of the batch (21 k of 1 M positions at cfg4) and the gate itself is a gather of
// `win` scalars per position from a table that lives in L2.  Backward: c[t][j] = sum of dpre over the positions whose tap j
// lands on token t (LDS-bucketed per 256-position window, gate_bwd_dx_kernel's c_out mode), then
//   dtable[t,:] = sum_j c[t][j] w[:,j]   (every row written once),   dw[:,j] = sum_t c[t][j] table[t,:],   db0 = sum_t c[t][pad].
__device__ __forceinline__ void gp_mark_kernel(long n, const long long* __restrict__ ids, int* __restrict__ used) {
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) used[ids[k]] = 1;
}

__device__ __forceinline__ void gp_compact_kernel(int V, int cap, const int* __restrict__ used, int* __restrict__ row_of_token,
                                                         int* __restrict__ tok_of_row, int* __restrict__ counter) {
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
            if (row < cap) tok_of_row[row] = v; else row = -1;
        }
        row_of_token[v] = row;
    }
}

// S[row][j] = <table[tok_of_row[row], :], w[:, j]>; 16 lanes per row (four rows per wave): a wave per row spent its time in
// the 48 shuffles that fold eight accumulators over 64 lanes (22 us for 21 k rows of 400 bytes)
__device__ __forceinline__ void gp_taps_kernel(int E, int win, int cap, const int* __restrict__ counter,
                                                      const int* __restrict__ tok_of_row, const float* __restrict__ table,
                                                      const float* __restrict__ w, float* __restrict__ S) {
    const int n = min(*counter, cap);
    const int sub = threadIdx.x >> 4, l = threadIdx.x & 15;                  // 16 row slots per workgroup
    for (int row0 = blockIdx.x * 16; row0 < n; row0 += gridDim.x * 16) {
        const int row = row0 + sub;
        const bool live = row < n;
        const float* trow = table + (long)(live ? tok_of_row[row] : 0) * E;
        float acc[kGateWP];
#pragma unroll
        for (int j = 0; j < kGateWP; ++j) acc[j] = 0.f;
        if (live)
            for (int e = l; e < E; e += 16) {
                const float x = trow[e];
#pragma unroll
                for (int j = 0; j < kGateWP; ++j)
                    if (j < win) acc[j] = fmaf(x, w[(long)e * win + j], acc[j]);
            }
#pragma unroll
        for (int j = 0; j < kGateWP; ++j)
            if (j < win) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
            }
        if (live && l < kGateWP) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < kGateWP; ++j) if (j == l) v = acc[j];
            S[(long)row * kGateWP + l] = v;
        }
    }
}

__device__ __forceinline__ void gp_gate_kernel(int B, int L, int win, const long long* __restrict__ ids,
                                                      const int* __restrict__ row_of_token, const float* __restrict__ S,
                                                      const float* __restrict__ b0, float* __restrict__ gate) {
    const long n = (long)B * L;
    const int pad = (win - 1) / 2;
    const float bias = b0[0];
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) {
        const int l = (int)(k % L);
        float s = 0.f;
        for (int j = 0; j < win; ++j) {
            const int p = l + j - pad;
            if (p < 0 || p >= L) continue;                    // zero padding outside the document
            const int row = row_of_token[ids[k + j - pad]];
            s += S[(long)row * kGateWP + j];
        }
        gate[k] = sigmoidf_(s + bias);
    }
}

// c[0][k] += c[1..copies-1][k]   (fixed order)
__device__ __forceinline__ void gp_csum_kernel(int cap, const int* __restrict__ counter, float* __restrict__ c) {
    const long n = (long)min(*counter, cap) * kGateWP, stride = (long)cap * kGateWP;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) {
        float s = c[k];
#pragma unroll
        for (int g = 1; g < kGateCopies; ++g) s += c[g * stride + k];
        c[k] = s;
    }
}

// ---- global gate, table gradient through the occurrence matrix ------------------------------------------------------
// dtable[t, :] = sum_p A[t, p] * wT[p, :]  with  A[t, p] = sum over the documents b that carry token t at position p of dpre[b].
// A is kept per distinct-token ROW ([n_rows, L] f32): building it costs ONE scalar atomic per position instead of a row of E
// (the f32 atomics are priced per 64-byte request, ~25 G/s chip-wide: the per-window rows of gate_bwd_dx_kernel are 2.9 M
// requests at cfg4, the scalars 0.5 M), and every table row is then written once, from its non-zeros.
__device__ __forceinline__ void gg_zero_kernel(int cap, int L, const int* __restrict__ counter, f32x4g* __restrict__ A,
                                                      int* __restrict__ dense_count) {
    const long n = (long)min(*counter, cap) * L / 4;                    // L % 4 == 0 (checked by the caller)
    const f32x4g z = {0.f, 0.f, 0.f, 0.f};
    if (blockIdx.x == 0 && threadIdx.x == 0) *dense_count = 0;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) A[k] = z;
}

__device__ __forceinline__ void gg_scatter_kernel(int B, int L, int pad_idx, const long long* __restrict__ ids,
                                                         const int* __restrict__ row_of_token, const float* __restrict__ dpre,
                                                         float* __restrict__ A) {
    const long n = (long)B * L;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) {
        const int t = (int)ids[k];
        if (t == pad_idx) continue;
        const int b = (int)(k / L), p = (int)(k - (long)b * L);
        atomicAdd(A + (long)row_of_token[t] * L + p, dpre[b]);
    }
}

// acc[c] += sum over the non-zeros of the 64-position stretch at p0 (lane j holds a = A[row, p0 + j]) of a * wT[p0 + j, c*64 + lane]:
// the non-zeros are visited kGgU at a time, their weight rows requested before the first use.
constexpr int kGgChunks = 4;                         // embedding columns per lane: E <= 256
constexpr int kGgU = 8;
__device__ __forceinline__ void gg_stretch(float a, int p0, int E, int lane, const float* __restrict__ wT, float (&acc)[kGgChunks]) {
    unsigned long long m = __ballot(a != 0.f);
    while (m) {
        int j[kGgU];
        float av[kGgU];
#pragma unroll
        for (int u = 0; u < kGgU; ++u) {
            j[u] = m ? __builtin_ctzll(m) : -1;
            if (m) m &= m - 1;
            av[u] = (j[u] >= 0) ? __shfl(a, j[u]) : 0.f;
        }
#pragma unroll
        for (int c = 0; c < kGgChunks; ++c) {
            const int e = c * 64 + lane;
            if (c * 64 >= E) break;                  // wave-uniform
            float x[kGgU];
#pragma unroll
            for (int u = 0; u < kGgU; ++u) x[u] = (j[u] >= 0 && e < E) ? wT[(long)(p0 + j[u]) * E + e] : 0.f;
#pragma unroll
            for (int u = 0; u < kGgU; ++u) acc[c] = fmaf(av[u], x[u], acc[c]);
        }
    }
}

// One WAVE per vocabulary entry (absent tokens and the pad row: zeros).  The wave reads its row of A (all stretches requested
// at once), counts the non-zeros, and either builds the table row from them or -- a row with more than kGgDense of them, a
// Zipf-hot token -- leaves it on the dense list for gg_dense_rows_kernel (a workgroup of 16 waves per row): a workgroup per
// entry spent its time on the fixed latencies of 50 k mostly near-empty rows, a wave per entry would walk the hot rows' 1024
// non-zeros one after the other.
constexpr int kGgDense = 96, kGgStretch = 16;        // stretches (of 64 positions) of a row held in registers at a time
__device__ __forceinline__ void gg_rows_kernel(int V, int L, int E, int pad_idx, const int* __restrict__ row_of_token,
                                                      const float* __restrict__ A, const float* __restrict__ wT,
                                                      float* __restrict__ dtable, int* __restrict__ dense_count,
                                                      int* __restrict__ dense_list, int accumulate) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= V) return;
    const int row = row_of_token[v];
    if (row < 0 || v == pad_idx) {
        if (!accumulate)
            for (int e = lane; e < E; e += 64) dtable[(long)v * E + e] = 0.f;
        return;
    }
    const float* arow = A + (long)row * L;
    int nnz = 0;
    float acc[kGgChunks];
#pragma unroll
    for (int c = 0; c < kGgChunks; ++c) acc[c] = 0.f;
    for (int pb = 0; pb < L; pb += 64 * kGgStretch) {
        float a[kGgStretch];
#pragma unroll
        for (int q = 0; q < kGgStretch; ++q) { const int p = pb + q * 64 + lane; a[q] = (p < L) ? arow[p] : 0.f; }   // one latency
#pragma unroll
        for (int q = 0; q < kGgStretch; ++q) nnz += __popcll(__ballot(a[q] != 0.f));
        if (nnz > kGgDense) {                        // wave-uniform; what was summed so far is dropped
            if (lane == 0) dense_list[atomicAdd(dense_count, 1)] = v;
            return;
        }
#pragma unroll
        for (int q = 0; q < kGgStretch; ++q) gg_stretch(a[q], pb + q * 64, E, lane, wT, acc);
    }
#pragma unroll
    for (int c = 0; c < kGgChunks; ++c) {
        const int e = c * 64 + lane;
        if (e < E) dtable[(long)v * E + e] = accumulate ? dtable[(long)v * E + e] + acc[c] : acc[c];
    }
}

// dense rows: one workgroup of 16 waves per listed token, wave w walks the stretches w, w + 16, ...; partial rows meet in LDS
__device__ __forceinline__ void gg_dense_rows_kernel(int L, int E, const int* __restrict__ row_of_token,
                                                             const float* __restrict__ A, const float* __restrict__ wT,
                                                             float* __restrict__ dtable, const int* __restrict__ dense_count,
                                                             const int* __restrict__ dense_list, int accumulate) {
    __shared__ float s_part[16][kGgChunks * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = *dense_count;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int v = dense_list[i];
        const float* arow = A + (long)row_of_token[v] * L;
        float acc[kGgChunks];
#pragma unroll
        for (int c = 0; c < kGgChunks; ++c) acc[c] = 0.f;
        for (int p0 = wave * 64; p0 < L; p0 += 16 * 64) gg_stretch(p0 + lane < L ? arow[p0 + lane] : 0.f, p0, E, lane, wT, acc);
#pragma unroll
        for (int c = 0; c < kGgChunks; ++c) s_part[wave][c * 64 + lane] = acc[c];
        __syncthreads();
        for (int e = tid; e < E; e += 1024) {
            float t = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 16; ++w2) t += s_part[w2][e];
            dtable[(long)v * E + e] = accumulate ? dtable[(long)v * E + e] + t : t;
        }
        __syncthreads();
    }
}

// dtable[v, :] = sum_j c[row(v)][j] w[:, j]  (0 for absent tokens and the pad row): the whole [V, E] gradient is overwritten
__device__ __forceinline__ void gp_dtable_kernel(int V, int E, int win, int pad_idx, const int* __restrict__ row_of_token,
                                                        const float* __restrict__ c, const float* __restrict__ w,
                                                        float* __restrict__ dtable, int accumulate) {
    const long n = (long)V * E;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) {
        const int v = (int)(k / E), e = (int)(k - (long)v * E);
        const int row = row_of_token[v];
        float s = 0.f;
        const bool live = row >= 0 && v != pad_idx;
        if (live)
            for (int j = 0; j < win; ++j) s = fmaf(c[(long)row * kGateWP + j], w[(long)e * win + j], s);
        if (!accumulate) dtable[k] = s;
        else if (live) dtable[k] += s;                  // shared gradient buffer (functional.table_fanout): rows of the batch only
    }
}

// accumulate form: dtable[tok_of_row[row], :] += sum_j c[row][j] w[:, j] over the batch's rows only (the shared gradient buffer of
// functional.table_fanout): 2.1 M elements at cfg4 instead of the vocabulary's 5 M
__device__ __forceinline__ void gp_dtable_rows_kernel(int E, int win, int cap, int pad_idx, const int* __restrict__ counter,
                                                             const int* __restrict__ tok_of_row, const float* __restrict__ c,
                                                             const float* __restrict__ w, float* __restrict__ dtable) {
    const long n = (long)min(*counter, cap) * E;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) {
        const int row = (int)(k / E), e = (int)(k - (long)row * E);
        const int v = tok_of_row[row];
        if (v == pad_idx) continue;
        float* dst = dtable + (long)v * E + e;
        const float old = *dst;
        float s = 0.f;
        for (int j = 0; j < win; ++j) s = fmaf(c[(long)row * kGateWP + j], w[(long)e * win + j], s);
        *dst = old + s;
    }
}

// per-chunk partial of dw[e, j] = sum_rows c[row][j] * table[tok][e]  and  db0 = sum_rows c[row][pad]; 128 rows per workgroup
constexpr int kGpRows = 128;
__device__ __forceinline__ void gp_dw_partial_kernel(int E, int win, int cap, const int* __restrict__ counter,
                                                            const int* __restrict__ tok_of_row, const float* __restrict__ table,
                                                            const float* __restrict__ c, float* __restrict__ part) {
    __shared__ float s_c[kGpRows * kGateWP];
    __shared__ long s_off[kGpRows];
    const int n = min(*counter, cap);
    const int r0 = blockIdx.x * kGpRows, rows = max(0, min(kGpRows, n - r0));
    const int n_out = win * E + 1;
    float* my = part + (long)blockIdx.x * n_out;
    for (int e = threadIdx.x; e < rows * kGateWP; e += 256) s_c[e] = c[(long)r0 * kGateWP + e];
    for (int r = threadIdx.x; r < rows; r += 256) s_off[r] = (long)tok_of_row[r0 + r] * E;
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += 256) {
        float acc[kGateWP];
#pragma unroll
        for (int j = 0; j < kGateWP; ++j) acc[j] = 0.f;
        for (int r = 0; r < rows; ++r) {
            const float x = table[s_off[r] + e];
#pragma unroll
            for (int j = 0; j < kGateWP; ++j) acc[j] = fmaf(s_c[r * kGateWP + j], x, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < kGateWP; ++j)
            if (j < win) my[(long)e * win + j] = acc[j];
    }
    if (threadIdx.x == 0) {
        float sb = 0.f;
        const int pad = (win - 1) / 2;
        for (int r = 0; r < rows; ++r) sb += s_c[r * kGateWP + pad];
        my[n_out - 1] = sb;
    }
}

// one wave per output: lanes stride the chunks that hold rows (ceil(n / 128) of them), shuffle reduction, fixed order
__device__ __forceinline__ void gp_dw_final_kernel(int cap, int n_out, const int* __restrict__ counter,
                                                          const float* __restrict__ part, float* __restrict__ dw,
                                                          float* __restrict__ db0) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= n_out) return;
    const int n_chunks = (min(*counter, cap) + kGpRows - 1) / kGpRows;
    float t = 0.f;
    for (int k = lane; k < n_chunks; k += 64) t += part[(long)k * n_out + o];
    t = wsum(t);
    if (lane == 0) { if (o < n_out - 1) dw[o] = t; else db0[0] = t; }
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
    if (int e_ = rbr::launch<global_gate_fwd_kernel, 256>(dim3(B), dim3(256), 0, (hipStream_t)stream, "datt global gate fwd launch", B, L, E, reinterpret_cast<const long long*>(ids), table, w, b0, gate)) return e_;
    return 0;
}

extern "C" size_t rbr_datt_gate_bwd_ws_floats(int32_t B, int32_t L, int32_t E, int32_t win, int32_t is_global) {
    if (B <= 0 || L <= 0 || E <= 0 || win <= 0) return 0;
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
        if (int e_ = rbr::launch<gate_bwd_dx_kernel, 256>(dim3((unsigned)(B * nwin)), dim3(256), (size_t)kGWin * win * sizeof(float), st, "datt local gate bwd dx launch", B, L, E, win, 0, nwin, ids64, w, (const float*)nullptr, gate, dgate, (const float*)nullptr, pad_idx, dtable, (const int*)nullptr, (float*)nullptr, 0L)) return e_;
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
    if (int e_ = rbr::launch<global_gate_bwd_dpre_kernel, 256>(dim3(B), dim3(256), 0, st, "datt global gate bwd dpre launch", B, L, gate, dgate, ws)) return e_;
    if (int e_ = rbr::launch<global_gate_bwd_dw_kernel, 256>(dim3(L + 1), dim3(256), 0, st, "datt global gate bwd dw launch", B, L, E, ids64, table, ws, dw, db0)) return e_;
    if (dtable != nullptr) {
        float* wT = ws + B;
        if (int e_ = rbr::launch<transpose_w_kernel, 256>(dim3((unsigned)std::min<long>(((long)E * L + 255) / 256, 2048)), dim3(256), 0, st, "datt global gate transpose launch", E, L, w, wT)) return e_;
        const int nwin = (L + kGWin - 1) / kGWin;
        if (int e_ = rbr::launch<gate_bwd_dx_kernel, 256>(dim3((unsigned)(B * nwin)), dim3(256), 0, st, "datt global gate bwd dx launch", B, L, E, 1, 1, nwin, ids64, w, (const float*)wT, gate, dgate, (const float*)ws, pad_idx, dtable, (const int*)nullptr, (float*)nullptr, 0L)) return e_;
    }
    return 0;
}

// ---- token-product local gate (same results as rbr_datt_local_gate_fwd / _bwd; `ws` must survive from forward to backward)
namespace {
struct GateProdLayout { size_t used, counter, row_of_token, tok_of_row, S, c, part, total; int cap, n_chunks; };
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
bool gate_prod_layout(int B, int L, int E, int win, int V, GateProdLayout& G) {
    if (B <= 0 || L <= 0 || E <= 0 || V <= 0 || win <= 0 || win > kGateWP || win % 2 == 0) return false;
    const long n_pos = (long)B * L;
    G.cap = (int)std::min<long>(V, n_pos);
    G.n_chunks = (G.cap + kGpRows - 1) / kGpRows;
    size_t o = 0;
    G.used = o;         o += al256((size_t)V * sizeof(int));
    G.counter = o;      o += 256;                                   // contiguous with `used`: zeroed together
    G.row_of_token = o; o += al256((size_t)V * sizeof(int));
    G.tok_of_row = o;   o += al256((size_t)G.cap * sizeof(int));
    G.S = o;            o += al256((size_t)G.cap * kGateWP * sizeof(float));
    G.c = o;            o += al256((size_t)G.cap * kGateWP * sizeof(float) * kGateCopies);
    G.part = o;         o += al256((size_t)G.n_chunks * ((size_t)win * E + 1) * sizeof(float));
    G.total = o;
    return true;
}
}  // namespace

// ---- distinct-token rows of a document batch, shared by the gates of one tower (layout: counter | used | row_of_token | tok_of_row)
namespace {
struct TokenRowsLayout { size_t counter, used, row_of_token, tok_of_row, total; int cap; };
bool token_rows_layout(int B, int L, int V, TokenRowsLayout& T) {
    if (B <= 0 || L <= 0 || V <= 0) return false;
    T.cap = (int)std::min<long>(V, (long)B * L);
    size_t o = 0;
    T.counter = o;      o += 256;
    T.used = o;         o += al256((size_t)V * sizeof(int));      // contiguous with the counter: zeroed together
    T.row_of_token = o; o += al256((size_t)V * sizeof(int));
    T.tok_of_row = o;   o += al256((size_t)T.cap * sizeof(int));
    T.total = o;
    return true;
}
}  // namespace


namespace {
struct GateRowMaps { const int *counter, *row_of_token, *tok_of_row; };
// the distinct-token maps a token-product gate call works on: the tower's shared ones (`rows`) or the private ones in its ws
bool gate_row_maps(int B, int L, int V, const GateProdLayout& G, const char* ws_base, const void* rows, GateRowMaps& M) {
    if (rows == nullptr) {
        M.counter = reinterpret_cast<const int*>(ws_base + G.counter);
        M.row_of_token = reinterpret_cast<const int*>(ws_base + G.row_of_token);
        M.tok_of_row = reinterpret_cast<const int*>(ws_base + G.tok_of_row);
        return true;
    }
    TokenRowsLayout T;
    if (!token_rows_layout(B, L, V, T) || T.cap != G.cap) { set_error("token rows do not fit the gate shape"); return false; }
    const char* r = static_cast<const char*>(rows);
    M.counter = reinterpret_cast<const int*>(r + T.counter);
    M.row_of_token = reinterpret_cast<const int*>(r + T.row_of_token);
    M.tok_of_row = reinterpret_cast<const int*>(r + T.tok_of_row);
    return true;
}
}  // namespace

extern "C" size_t rbr_datt_local_gate_prod_ws_bytes(int32_t B, int32_t L, int32_t E, int32_t win, int32_t V) {
    GateProdLayout G;
    if (!gate_prod_layout(B, L, E, win, V, G)) return 0;
    // worth it when the vocabulary bounds the distinct tokens well below the position count (as rbr_textcnn_fwd_ws_bytes)
    const int mode = forced_conv_mode();
    if (mode == 1) return 0;
    if (mode != 2 && ((long)V * 5 > (long)B * L * 2 || (long)B * L < 4096)) return 0;
    return G.total;
}

extern "C" int rbr_datt_local_gate_fwd_prod(int32_t B, int32_t L, int32_t E, int32_t win, int32_t V, const int64_t* ids,
                                            const float* table, const float* w, const float* b0, float* gate, void* ws,
                                            const void* rows, void* stream) {
    GateProdLayout G;
    if (!gate_prod_layout(B, L, E, win, V, G)) { set_error("bad local gate shape B=%d L=%d E=%d win=%d V=%d", B, L, E, win, V); return RBR_ERR_BAD_ARG; }
    if (!ids || !table || !w || !b0 || !gate || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipStream_t st = (hipStream_t)stream;
    char* base = static_cast<char*>(ws);
    int* used = reinterpret_cast<int*>(base + G.used);
    float* S = reinterpret_cast<float*>(base + G.S);
    const long long* ids64 = reinterpret_cast<const long long*>(ids);
    const long n_pos = (long)B * L;
    GateRowMaps M;
    if (!gate_row_maps(B, L, V, G, base, rows, M)) return RBR_ERR_BAD_ARG;
    const int *counter = M.counter, *row_of_token = M.row_of_token, *tok_of_row = M.tok_of_row;
    if (rows == nullptr) {               // the tower's shared maps (rbr_datt_token_rows) were not handed in: build private ones
        if (int e = zero_words(used, G.row_of_token - G.used, st)) return e;
        if (int e_ = rbr::launch<gp_mark_kernel, 256>(dim3((unsigned)std::min<long>((n_pos + 255) / 256, 2048)), dim3(256), 0, st, "datt gate mark launch", n_pos, ids64, used)) return e_;
        if (int e_ = rbr::launch<gp_compact_kernel, 256>(dim3((V + 255) / 256), dim3(256), 0, st, "datt gate compact launch", V, G.cap, used, reinterpret_cast<int*>(base + G.row_of_token), reinterpret_cast<int*>(base + G.tok_of_row), reinterpret_cast<int*>(base + G.counter))) return e_;
    }
    if (int e_ = rbr::launch<gp_taps_kernel, 256>(dim3((unsigned)std::min((G.cap + 15) / 16, 4096)), dim3(256), 0, st, "datt gate taps launch", E, win, G.cap, counter, tok_of_row, table, w, S)) return e_;
    if (int e_ = rbr::launch<gp_gate_kernel, 256>(dim3((unsigned)std::min<long>((n_pos + 255) / 256, 8192)), dim3(256), 0, st, "datt gate gather launch", B, L, win, ids64, row_of_token, S, b0, gate)) return e_;
    return 0;
}

extern "C" int rbr_datt_local_gate_bwd_prod(int32_t B, int32_t L, int32_t E, int32_t win, int32_t V, const int64_t* ids,
                                            const float* table, const float* w, const float* gate, const float* dgate,
                                            int32_t pad_idx, float* dw, float* db0, float* dtable, void* ws, const void* rows,
                                            int32_t accumulate, void* stream) {
    GateProdLayout G;
    if (!gate_prod_layout(B, L, E, win, V, G)) { set_error("bad local gate shape B=%d L=%d E=%d win=%d V=%d", B, L, E, win, V); return RBR_ERR_BAD_ARG; }
    if (!ids || !table || !w || !gate || !dgate || !dw || !db0 || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipStream_t st = (hipStream_t)stream;
    char* base = static_cast<char*>(ws);
    GateRowMaps M;
    if (!gate_row_maps(B, L, V, G, base, rows, M)) return RBR_ERR_BAD_ARG;
    const int *counter = M.counter, *row_of_token = M.row_of_token, *tok_of_row = M.tok_of_row;
    float* c = reinterpret_cast<float*>(base + G.c);
    float* part = reinterpret_cast<float*>(base + G.part);
    const long long* ids64 = reinterpret_cast<const long long*>(ids);
    if (int e = zero_words(c, (size_t)G.cap * kGateWP * sizeof(float) * kGateCopies, st)) return e;
    const int nwin = (L + kGWin - 1) / kGWin;
    if (int e_ = rbr::launch<gate_bwd_dx_kernel, 256>(dim3((unsigned)(B * nwin)), dim3(256), 0, st, "datt gate tap-sum launch", B, L, E, win, 0, nwin, ids64, w, (const float*)nullptr, gate, dgate, (const float*)nullptr, pad_idx, (float*)nullptr, row_of_token, c, (long)G.cap * kGateWP)) return e_;
    if (int e_ = rbr::launch<gp_csum_kernel, 256>(dim3((unsigned)std::min((G.cap * kGateWP + 255) / 256, 2048)), dim3(256), 0, st, "datt gate tap-sum fold launch", G.cap, counter, c)) return e_;
    const int n_out = win * E + 1;
    if (int e_ = rbr::launch<gp_dw_partial_kernel, 256>(dim3(G.n_chunks), dim3(256), 0, st, "datt gate dw partial launch", E, win, G.cap, counter, tok_of_row, table, c, part)) return e_;
    if (int e_ = rbr::launch<gp_dw_final_kernel, 256>(dim3((n_out + 3) / 4), dim3(256), 0, st, "datt gate dw final launch", G.cap, n_out, counter, part, dw, db0)) return e_;
    if (dtable != nullptr) {
        const long n = (long)V * E;
        if (accumulate) {
            if (int e_ = rbr::launch<gp_dtable_rows_kernel, 256>(dim3((unsigned)std::min<long>(((long)G.cap * E + 255) / 256, 8192)), dim3(256), 0, st, "gp_dtable_rows_kernel launch", E, win, G.cap, pad_idx, counter, tok_of_row, c, w, dtable)) return e_;
        } else {
            if (int e_ = rbr::launch<gp_dtable_kernel, 256>(dim3((unsigned)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, st, "datt gate dtable launch", V, E, win, pad_idx, row_of_token, c, w, dtable, 0)) return e_;
        }
    }
    return 0;
}

extern "C" size_t rbr_datt_token_rows_ws_bytes(int32_t B, int32_t L, int32_t V) {
    TokenRowsLayout T;
    return token_rows_layout(B, L, V, T) ? T.total : 0;
}

extern "C" int rbr_datt_token_rows(int32_t B, int32_t L, int32_t V, const int64_t* ids, void* rows, void* stream) {
    TokenRowsLayout T;
    if (!token_rows_layout(B, L, V, T)) { set_error("bad token-rows shape B=%d L=%d V=%d", B, L, V); return RBR_ERR_BAD_ARG; }
    if (!ids || !rows) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipStream_t st = (hipStream_t)stream;
    char* base = static_cast<char*>(rows);
    int* counter = reinterpret_cast<int*>(base + T.counter);
    int* used = reinterpret_cast<int*>(base + T.used);
    if (int e = zero_words(counter, T.row_of_token - T.counter, st)) return e;
    const long n_pos = (long)B * L;
    if (int e_ = rbr::launch<gp_mark_kernel, 256>(dim3((unsigned)std::min<long>((n_pos + 255) / 256, 2048)), dim3(256), 0, st, "token rows mark launch", n_pos, reinterpret_cast<const long long*>(ids), used)) return e_;
    if (int e_ = rbr::launch<gp_compact_kernel, 256>(dim3((V + 255) / 256), dim3(256), 0, st, "token rows compact launch", V, T.cap, used, reinterpret_cast<int*>(base + T.row_of_token), reinterpret_cast<int*>(base + T.tok_of_row), counter)) return e_;
    return 0;
}

extern "C" size_t rbr_datt_global_gate_bwd_rows_ws_floats(int32_t B, int32_t L, int32_t E, int32_t V) {
    TokenRowsLayout T;
    if (!token_rows_layout(B, L, V, T) || E <= 0 || E > kGgChunks * 64 || L % 4 != 0) return 0;   // 0: use rbr_datt_global_gate_bwd
    return (size_t)B + (size_t)E * L + 8 + (size_t)T.cap * L + 4 + (size_t)V;       // dpre | wT | A | dense count + list
}

extern "C" int rbr_datt_global_gate_bwd_rows(int32_t B, int32_t L, int32_t E, int32_t V, const int64_t* ids, const float* table,
                                             const float* w, const float* gate, const float* dgate, int32_t pad_idx, float* dw,
                                             float* db0, float* dtable, float* ws, const void* rows, int32_t accumulate, void* stream) {
    TokenRowsLayout T;
    if (!gate_args_ok(B, L, E, 1) || !token_rows_layout(B, L, V, T)) return RBR_ERR_BAD_ARG;
    if (E > kGgChunks * 64 || L % 4 != 0) { set_error("global gate over token rows needs E <= %d and L %% 4 == 0 (E=%d L=%d)", kGgChunks * 64, E, L); return RBR_ERR_UNSUPPORTED; }
    // accumulate & 2: the table-gradient phase alone -- dpre is in `ws` already (an earlier call with dtable == NULL on the same
    // workspace did the weight phase); a caller that puts the two phases of its towers on different streams (functional._DattTowers)
    const bool rows_only = (accumulate & 2) != 0;
    accumulate &= 1;
    if (!ids || !w || !ws || !rows || (!rows_only && (!table || !gate || !dgate || !dw || !db0)) || (rows_only && !dtable)) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const long long* ids64 = reinterpret_cast<const long long*>(ids);
    const char* base = static_cast<const char*>(rows);
    const int* counter = reinterpret_cast<const int*>(base + T.counter);
    const int* row_of_token = reinterpret_cast<const int*>(base + T.row_of_token);
    if (!rows_only) {
        if (int e_ = rbr::launch<global_gate_bwd_dpre_kernel, 256>(dim3(B), dim3(256), 0, st, "datt global gate bwd dpre launch", B, L, gate, dgate, ws)) return e_;
        if (int e_ = rbr::launch<global_gate_bwd_dw_kernel, 256>(dim3(L + 1), dim3(256), 0, st, "datt global gate bwd dw launch", B, L, E, ids64, table, ws, dw, db0)) return e_;
    }
    if (dtable != nullptr) {
        float* wT = ws + B;
        float* A = wT + (((size_t)E * L + 3) & ~(size_t)3);             // 16-byte aligned behind wT when ws is
        if ((reinterpret_cast<uintptr_t>(A) & 15) != 0) A += 4 - ((reinterpret_cast<uintptr_t>(A) & 15) >> 2);
        if (int e_ = rbr::launch<transpose_w_kernel, 256>(dim3((unsigned)std::min<long>(((long)E * L + 255) / 256, 2048)), dim3(256), 0, st, "datt global gate transpose launch", E, L, w, wT)) return e_;
        int* dense_count = reinterpret_cast<int*>(A + (size_t)T.cap * L);
        int* dense_list = dense_count + 4;
        PairSolo solo;        // zero A -> scatter into A -> rows from A: a chain over the 160 MB occurrence matrix (rbr_launch.h)
        if (int e_ = rbr::launch<gg_zero_kernel, 256>(dim3((unsigned)std::min<long>(((long)T.cap * L / 4 + 255) / 256, 8192)), dim3(256), 0, st, "datt global gate occurrence zero launch", T.cap, L, counter, reinterpret_cast<f32x4g*>(A), dense_count)) return e_;
        const long n_pos = (long)B * L;
        if (int e_ = rbr::launch<gg_scatter_kernel, 256>(dim3((unsigned)std::min<long>((n_pos + 255) / 256, 8192)), dim3(256), 0, st, "datt global gate occurrence scatter launch", B, L, pad_idx, ids64, row_of_token, (const float*)ws, A)) return e_;
        if (int e_ = rbr::launch<gg_rows_kernel, 256>(dim3((unsigned)((V + 3) / 4)), dim3(256), 0, st, "datt global gate rows launch", V, L, E, pad_idx, row_of_token, (const float*)A, (const float*)wT, dtable, dense_count, dense_list, accumulate)) return e_;
        if (int e_ = rbr::launch<gg_dense_rows_kernel, 1024>(dim3((unsigned)std::min(V, 512)), dim3(1024), 0, st, "datt global gate dense rows launch", L, E, row_of_token, (const float*)A, (const float*)wT, dtable, (const int*)dense_count, (const int*)dense_list, accumulate)) return e_;
    }
    return 0;
}
