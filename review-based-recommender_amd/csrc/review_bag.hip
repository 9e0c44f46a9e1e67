// review_bag.hip -- the SimpleSiamese encoder (models/simple_siamese/simple_siamese.py:57-74):
//   * bag of embeddings per review: WordEmbedding -> VariationalDropout (one mask per (review, dim)) -> MaskedAvgPooling1d
//     (layers.py:24-50, 53-68, 90-110): out[r, :] = drop[r, :] * sum_l mask[r,l] * table[ids[r,l], :] / (sum_l mask[r,l] + 1e-8)
//     -- a pure HBM-bound row gather, no contraction;
//   * AddictiveAttention over the reviews of a user / item (layers.py:171-197), with NodeDropout (layers.py:7-22) folded in:
//     x = node_drop[b,r] * rev[b,r,:];  t = tanh(x Wp^T + bp);  logit = <t, wi>;  s = softmax_r(masked_fill(logit, -1e8));
//     out[b,:] = sum_r s[r] x[r,:].
// Nothing of shape [reviews, tokens, D] is ever written to HBM (the reference materialises it twice per tower).
#include "rbr_common.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace rbr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------ bag forward
// one wave per review; lanes own float4 columns of the embedding (scalar columns when D % 4 != 0)
__global__ __launch_bounds__(256) void bag_fwd_kernel(int n_rev, int T, int D, const long long* __restrict__ ids,
                                                      const unsigned char* __restrict__ mask, const float* __restrict__ table,
                                                      const float* __restrict__ drop, float* __restrict__ out,
                                                      float* __restrict__ inv_len) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kWavesPerWG + wave;
    if (r >= n_rev) return;
    const long long* rid = ids + (long)r * T;
    const unsigned char* rm = mask ? mask + (long)r * T : nullptr;
    int cnt = 0;
    for (int l = lane; l < T; l += 64) cnt += (rm == nullptr || rm[l]) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    const float inv = 1.f / ((float)cnt + 1e-8f);
    if (lane == 0) inv_len[r] = inv;
    const bool vec = (D % 4 == 0) && ((((uintptr_t)table) & 15) == 0);
    if (vec) {
        const int nq4 = D >> 2;
        for (int q0 = 0; q0 < nq4; q0 += 64) {
            const int q4 = q0 + lane;
            const bool lane_ok = q4 < nq4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // the review's token rows: every lane fetches ONE row offset (-1 = masked) of a 64-token stretch, then the rows
            // are read 8 at a time with the offsets broadcast from the lanes -- the row reads no longer wait for an id load each
            for (int t0 = 0; t0 < T; t0 += 64) {
                const int lt = t0 + lane;
                long my_off = -1;
                if (lt < T && (rm == nullptr || rm[lt])) my_off = rid[lt] * (long)D;
                const int nt = min(64, T - t0);
                for (int l0 = 0; l0 < nt; l0 += 8) {
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const long off = __shfl(my_off, (l0 + u) & 63);      // (lanes past nt hold -1)
                        v[u] = (lane_ok && l0 + u < nt && off >= 0) ? *reinterpret_cast<const f32x4*>(table + off + 4 * q4)
                                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                }
            }
            if (lane_ok) {
                f32x4 o = acc * inv;
                if (drop != nullptr) o = o * *reinterpret_cast<const f32x4*>(drop + (long)r * D + 4 * q4);
                *reinterpret_cast<f32x4*>(out + (long)r * D + 4 * q4) = o;
            }
        }
    } else {
        for (int d = lane; d < D; d += 64) {
            float acc = 0.f;
            for (int l = 0; l < T; ++l)
                if (rm == nullptr || rm[l]) acc += table[rid[l] * (long)D + d];
            float o = acc * inv;
            if (drop != nullptr) o *= drop[(long)r * D + d];
            out[(long)r * D + d] = o;
        }
    }
}

// ------------------------------------------------------------------------------------ bag backward
// dtable[ids[r,l], :] += mask * drop[r,:] * d_out[r,:] * inv_len[r]   (rows of padding_idx get nothing).
// One atomic row per occurrence would hammer the rows of frequent words (a Zipf-hot token sits in nearly every review).
// Instead the occurrences are SORTED by token (rocPRIM radix sort of (token, position) pairs), each wave walks a chunk
// of the sorted list, sums the runs of equal tokens in registers and adds one row per run: rows added ~ distinct tokens
// + chunks, not positions.
constexpr int kBagChunk = 64;       // sorted occurrences per wave (one per lane)

__global__ __launch_bounds__(256) void bag_keys_kernel(long n_pos, const long long* __restrict__ ids,
                                                       const unsigned char* __restrict__ mask, int padding_idx, int sentinel,
                                                       int* __restrict__ keys, int* __restrict__ vals) {
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n_pos; k += (long)gridDim.x * 256) {
        const long long t = ids[k];
        const bool ok = (mask == nullptr || mask[k]) && t != padding_idx;
        keys[k] = ok ? (int)t : sentinel;       // masked / padding occurrences sort to the end and are skipped
        vals[k] = (int)k;
    }
}

__global__ __launch_bounds__(256) void bag_bwd_sorted_kernel(long n_pos, int T, int D, int sentinel, const int* __restrict__ keys,
                                                             const int* __restrict__ vals, const float* __restrict__ drop,
                                                             const float* __restrict__ inv_len, const float* __restrict__ d_out,
                                                             float* __restrict__ dtable) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long e0 = ((long)blockIdx.x * kWavesPerWG + wave) * kBagChunk;
    if (e0 >= n_pos) return;
    const int n = (int)min((long)kBagChunk, n_pos - e0);
    // lane j holds occurrence e0 + j: token, review, 1 / len -- read once, broadcast with readlane below
    const int my_key = (lane < n) ? keys[e0 + lane] : sentinel;
    const int my_rev = (lane < n) ? vals[e0 + lane] / T : 0;
    const float my_inv = (lane < n && my_key != sentinel) ? inv_len[my_rev] : 0.f;
    // lanes = embedding columns, two 64-column stretches per pass; 8 occurrences' gradient rows in flight (16 loads per lane):
    // a wave walks its 64 occurrences in 8 dependent rounds instead of 16 per 64 columns (75 -> 35 us at the toys shape)
    for (int d0 = 0; d0 < D; d0 += 128) {
        const int da = d0 + lane, db = d0 + 64 + lane;
        const bool ona = da < D, onb = db < D;
        float acca = 0.f, accb = 0.f;
        int cur = __builtin_amdgcn_readfirstlane(my_key);
        for (int j0 = 0; j0 < n; j0 += 8) {
            int t[8];
            float ga[8], gb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = min(j0 + u, n - 1);
                t[u] = (j0 + u < n) ? __builtin_amdgcn_readlane(my_key, j) : sentinel;
                const int r = __builtin_amdgcn_readlane(my_rev, j);
                const float inv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_inv), j));
                ga[u] = gb[u] = 0.f;
                if (t[u] != sentinel) {          // wave-uniform
                    if (ona) { ga[u] = d_out[(long)r * D + da] * inv; if (drop != nullptr) ga[u] *= drop[(long)r * D + da]; }
                    if (onb) { gb[u] = d_out[(long)r * D + db] * inv; if (drop != nullptr) gb[u] *= drop[(long)r * D + db]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t[u] != cur) {                  // wave-uniform
                    if (cur != sentinel) {
                        if (ona) atomicAdd(dtable + (long)cur * D + da, acca);
                        if (onb) atomicAdd(dtable + (long)cur * D + db, accb);
                    }
                    acca = accb = 0.f;
                    cur = t[u];
                }
                acca += ga[u];
                accb += gb[u];
            }
        }
        if (cur != sentinel) {
            if (ona) atomicAdd(dtable + (long)cur * D + da, acca);
            if (onb) atomicAdd(dtable + (long)cur * D + db, accb);
        }
    }
}

// Sort-free form (the default): one workgroup per window of kBagWin consecutive token positions buckets them by token in LDS
// (hash table, as datt_gates.hip's gate backward), sums per distinct token the scaled gradient rows of the reviews it occurs
// in, and adds ONE row per distinct token of the window.  f32 atomics are served per 64-byte request (~25 G/s) and requests
// to one line queue behind each other (~35 ns): a Zipf-hot token now sends one row per window (~550 of them at the toys
// shape), not one per occurrence -- and the 18 launches of the capture-safe merge sort are gone.
constexpr int kBagWin = 512, kBagHash = 2048;

__global__ __launch_bounds__(256) void bag_bwd_bucket_kernel(long n_pos, int T, int D, const long long* __restrict__ ids,
                                                             const unsigned char* __restrict__ mask, int padding_idx,
                                                             const float* __restrict__ drop, const float* __restrict__ inv_len,
                                                             const float* __restrict__ d_out, float* __restrict__ dtable) {
    __shared__ int s_tok[kBagWin], s_rev[kBagWin], s_fill[kBagWin];
    __shared__ float s_inv[kBagWin];
    __shared__ short s_leader[kBagWin], s_cnt[kBagWin], s_start[kBagWin], s_sorted[kBagWin], s_keys[kBagWin];
    __shared__ int s_hkey[kBagHash], s_hval[kBagHash];
    __shared__ int s_nkeys;
    const long p0 = (long)blockIdx.x * kBagWin;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int k = tid; k < kBagHash; k += 256) { s_hkey[k] = -1; s_hval[k] = kBagWin; }
    for (int r = tid; r < kBagWin; r += 256) {
        const long k = p0 + r;
        int t = -1, rev = 0;
        float inv = 0.f;
        if (k < n_pos) {
            const long long id = ids[k];
            if ((mask == nullptr || mask[k]) && id != padding_idx) { t = (int)id; rev = (int)(k / T); inv = inv_len[rev]; }
        }
        s_tok[r] = t; s_rev[r] = rev; s_inv[r] = inv; s_fill[r] = 0;
    }
    __syncthreads();
    for (int r = tid; r < kBagWin; r += 256) {
        const int t = s_tok[r];
        int slot = -1;
        if (t >= 0) {
            unsigned hh = ((unsigned)t * 2654435761u) >> 21;              // kBagHash = 2^11
            for (;;) {
                const int old = atomicCAS(&s_hkey[hh], -1, t);
                if (old == -1 || old == t) break;
                hh = (hh + 1) & (kBagHash - 1);
            }
            atomicMin(&s_hval[hh], r);
            slot = (int)hh;
        }
        s_leader[r] = (short)slot;
    }
    __syncthreads();
    for (int r = tid; r < kBagWin; r += 256) {
        const int slot = s_leader[r];
        const int lead = slot >= 0 ? s_hval[slot] : -1;
        s_leader[r] = (short)lead;
        if (lead >= 0) atomicAdd(&s_fill[lead], 1);
    }
    __syncthreads();
    if (tid < 64) {      // exclusive scan of the per-token counts and of the leaders (dense key list)
        int run = 0, krun = 0;
        for (int r0 = 0; r0 < kBagWin; r0 += 64) {
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
    for (int r = tid; r < kBagWin; r += 256) {
        const int lead = s_leader[r];
        if (lead >= 0) s_sorted[s_start[lead] + atomicAdd(&s_fill[lead], 1)] = (short)r;
    }
    __syncthreads();
    const int total = s_nkeys * D;
    constexpr int kU = 4;                            // items in flight per thread: their first gradient rows before the first atomic
    for (int it = tid; it < total; it += 256 * kU) {
        float sum[kU];
        float* dst[kU];
        int cnt[kU], base[kU], dd[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int item = it + u * 256;
            const bool ok = item < total;
            const int k = ok ? item / D : 0, d = ok ? item - k * D : 0;
            const int key = s_keys[k];
            cnt[u] = ok ? s_cnt[key] : 0;
            base[u] = s_start[key];
            dd[u] = d;
            dst[u] = dtable + (long)s_tok[key] * D + d;
            sum[u] = 0.f;
            if (ok) {
                const int r = s_sorted[base[u]];
                const long o = (long)s_rev[r] * D + d;
                sum[u] = d_out[o] * s_inv[r] * (drop != nullptr ? drop[o] : 1.f);
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            for (int q = 1; q < cnt[u]; ++q) {
                const int r = s_sorted[base[u] + q];
                const long o = (long)s_rev[r] * D + dd[u];
                sum[u] += d_out[o] * s_inv[r] * (drop != nullptr ? drop[o] : 1.f);
            }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (cnt[u] > 0) atomicAdd(dst[u], sum[u]);
    }
}

// ------------------------------------------------------------------------------------ additive attention
constexpr int kAttMaxR = 64;      // reviews per user / item

// one workgroup per (user | item) row b.  LDS: x [R,H] (node dropout applied), t [R,K], logits / scores [R]
__global__ __launch_bounds__(256) void addatt_fwd_kernel(int B, int R, int H, int K, const float* __restrict__ rev,
                                                         const unsigned char* __restrict__ mask, const float* __restrict__ node_drop,
                                                         const float* __restrict__ Wp, const float* __restrict__ bp,
                                                         const float* __restrict__ wi, float* __restrict__ out,
                                                         float* __restrict__ scores, float* __restrict__ t_out) {
    extern __shared__ float sm[];
    float* x = sm;                   // [R][H]
    float* t = x + R * H;            // [R][K]
    float* lg = t + R * K;           // [R]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < R * H; e += 256) {
        const int r = e / H;
        float v = rev[(long)b * R * H + e];
        if (node_drop != nullptr) v *= node_drop[(long)b * R + r];
        x[e] = v;
    }
    __syncthreads();
    for (int e = tid; e < R * K; e += 256) {
        const int r = e / K, k = e - r * K;
        float acc = bp[k];
        const float* wrow = Wp + (long)k * H;
        for (int h = 0; h < H; ++h) acc = fmaf(x[r * H + h], wrow[h], acc);
        const float tv = tanhf(acc);
        t[e] = tv;
        t_out[(long)b * R * K + e] = tv;
    }
    __syncthreads();
    if (tid < R) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(t[tid * K + k], wi[k], acc);
        lg[tid] = (mask == nullptr || mask[(long)b * R + tid]) ? acc : -1e8f;       // masked_fill(~mask, -1e8)
    }
    __syncthreads();
    if (tid == 0) {      // R <= 64: a serial softmax in the reference's order of operations (max, exp, sum, divide)
        float mx = lg[0];
        for (int r = 1; r < R; ++r) mx = fmaxf(mx, lg[r]);
        float s = 0.f;
        for (int r = 0; r < R; ++r) { lg[r] = expf(lg[r] - mx); s += lg[r]; }
        const float is = 1.f / s;
        for (int r = 0; r < R; ++r) { lg[r] *= is; scores[(long)b * R + r] = lg[r]; }
    }
    __syncthreads();
    for (int h = tid; h < H; h += 256) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(lg[r], x[r * H + h], acc);
        out[(long)b * H + h] = acc;
    }
}

// d_rev [B,R,H], d_pre [B,R,K] (pre-activation gradient, consumed by the dWp GEMM), d_bp / d_wi accumulated by atomics
__global__ __launch_bounds__(256) void addatt_bwd_kernel(int B, int R, int H, int K, const float* __restrict__ rev,
                                                         const unsigned char* __restrict__ mask, const float* __restrict__ node_drop,
                                                         const float* __restrict__ Wp, const float* __restrict__ wi,
                                                         const float* __restrict__ scores, const float* __restrict__ t_in,
                                                         const float* __restrict__ d_out, float* __restrict__ d_rev,
                                                         float* __restrict__ d_pre, float* __restrict__ x_eff,
                                                         float* __restrict__ d_bp, float* __restrict__ d_wi) {
    extern __shared__ float sm[];
    float* x = sm;                   // [R][H]
    float* dp = x + R * H;           // [R][K]
    float* dl = dp + R * K;          // [R] d_logit
    float* dsr = dl + R;             // [R] d_score
    float* dy = dsr + R;             // [H]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < R * H; e += 256) {
        const int r = e / H;
        float v = rev[(long)b * R * H + e];
        if (node_drop != nullptr) v *= node_drop[(long)b * R + r];
        x[e] = v;
        x_eff[(long)b * R * H + e] = v;          // operand of the dWp GEMM
    }
    for (int h = tid; h < H; h += 256) dy[h] = d_out[(long)b * H + h];
    __syncthreads();
    if (tid < R) {
        float acc = 0.f;
        for (int h = 0; h < H; ++h) acc = fmaf(dy[h], x[tid * H + h], acc);
        dsr[tid] = acc;
    }
    __syncthreads();
    if (tid < R) {
        float dot = 0.f;
        for (int r = 0; r < R; ++r) dot = fmaf(scores[(long)b * R + r], dsr[r], dot);
        const float s = scores[(long)b * R + tid];
        // masked_fill blocks the gradient of masked logits (also when every review is masked and s is uniform)
        dl[tid] = (mask == nullptr || mask[(long)b * R + tid]) ? s * (dsr[tid] - dot) : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < R * K; e += 256) {
        const int r = e / K, k = e - r * K;
        const float tv = t_in[(long)b * R * K + e];
        const float g = dl[r] * wi[k] * (1.f - tv * tv);
        dp[e] = g;
        d_pre[(long)b * R * K + e] = g;
    }
    for (int k = tid; k < K; k += 256) {
        float gw = 0.f;
        for (int r = 0; r < R; ++r) gw = fmaf(dl[r], t_in[((long)b * R + r) * K + k], gw);
        atomicAdd(d_wi + k, gw);
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        float gb = 0.f;
        for (int r = 0; r < R; ++r) gb += dp[r * K + k];
        atomicAdd(d_bp + k, gb);
    }
    for (int e = tid; e < R * H; e += 256) {
        const int r = e / H, h = e - r * H;
        float acc = scores[(long)b * R + r] * dy[h];
        for (int k = 0; k < K; ++k) acc = fmaf(dp[r * K + k], Wp[(long)k * H + h], acc);
        if (node_drop != nullptr) acc *= node_drop[(long)b * R + r];
        d_rev[(long)b * R * H + e] = acc;
    }
}

// dWp[k, h] += sum over a chunk of 64 rows n = (b, r) of d_pre[n, k] * x[n, h]   (dWp zeroed beforehand; one workgroup per
// chunk stages its rows in LDS, every thread owns a few (k, h) outputs and adds them with one atomic each)
constexpr int kDwRows = 64;
__global__ __launch_bounds__(256) void addatt_dw_kernel(int N, int H, int K, const float* __restrict__ d_pre,
                                                        const float* __restrict__ x, float* __restrict__ d_Wp) {
    extern __shared__ float sm[];
    float* sp = sm;                    // [rows][K]
    float* sx = sm + kDwRows * K;      // [rows][H]
    const int n0 = blockIdx.x * kDwRows, rows = min(kDwRows, N - n0);
    for (int e = threadIdx.x; e < rows * K; e += 256) sp[e] = d_pre[(long)n0 * K + e];
    for (int e = threadIdx.x; e < rows * H; e += 256) sx[e] = x[(long)n0 * H + e];
    __syncthreads();
    for (int e = threadIdx.x; e < K * H; e += 256) {
        const int k = e / H, h = e - k * H;
        float acc = 0.f;
        for (int n = 0; n < rows; ++n) acc = fmaf(sp[n * K + k], sx[n * H + h], acc);
        atomicAdd(d_Wp + e, acc);
    }
}

}  // namespace rbr

using namespace rbr;

extern "C" int rbr_review_bag_fwd(int32_t n_rev, int32_t T, int32_t D, const int64_t* ids, const uint8_t* mask, const float* table,
                                  const float* drop, float* out, float* inv_len, void* stream) {
    if (n_rev <= 0 || T <= 0 || D <= 0) { set_error("bad shape n_rev=%d T=%d D=%d", n_rev, T, D); return RBR_ERR_BAD_ARG; }
    if (!ids || !table || !out || !inv_len) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(bag_fwd_kernel, dim3((n_rev + kWavesPerWG - 1) / kWavesPerWG), dim3(256), 0, (hipStream_t)stream, n_rev, T, D,
                       reinterpret_cast<const long long*>(ids), mask, table, drop, out, inv_len);
    RBR_CHECK_LAUNCH("review_bag fwd launch");
    return 0;
}

// workspace of the backward: keys | vals (in) | keys | vals (sorted) | rocPRIM temporary storage
// (the larger of the radix sort's and the merge sort's: which of the two runs is decided per call, see rbr_review_bag_bwd)
static size_t bag_sort_temp_bytes(long n_pos) {
    size_t radix = 0, merge = 0;
    int* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, radix, nul, nul, nul, nul, (size_t)n_pos, 0, 32, (hipStream_t)0);
    (void)rocprim::merge_sort(nullptr, merge, nul, nul, nul, nul, (size_t)n_pos, rocprim::less<int>(), (hipStream_t)0);
    return std::max(radix, merge);
}

extern "C" size_t rbr_review_bag_bwd_ws_bytes(int32_t n_rev, int32_t T) {
    if (n_rev <= 0 || T <= 0) return 0;
    const long n_pos = (long)n_rev * T;
    const size_t arr = (((size_t)n_pos * sizeof(int)) + 255) & ~(size_t)255;
    return 4 * arr + bag_sort_temp_bytes(n_pos) + 256;
}

extern "C" int rbr_review_bag_bwd(int32_t n_rev, int32_t T, int32_t D, int32_t V, const int64_t* ids, const uint8_t* mask,
                                  const float* drop, const float* inv_len, const float* d_out, int32_t padding_idx, float* dtable,
                                  void* ws, void* stream) {
    if (n_rev <= 0 || T <= 0 || D <= 0 || V <= 0) { set_error("bad shape n_rev=%d T=%d D=%d V=%d", n_rev, T, D, V); return RBR_ERR_BAD_ARG; }
    if (!ids || !inv_len || !d_out || !dtable || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    const long n_pos = (long)n_rev * T;
    if (n_pos >= (1L << 31)) { set_error("too many token positions"); return RBR_ERR_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    const char* bag_env = getenv("RBR_BAG_BWD");            // read per call: "sort" selects the sorted-run form below (kept as a cross-check)
    if (!(bag_env && !strcmp(bag_env, "sort"))) {          // default: LDS-bucketed windows, no sort
        hipLaunchKernelGGL(bag_bwd_bucket_kernel, dim3((unsigned)((n_pos + kBagWin - 1) / kBagWin)), dim3(256), 0, st, n_pos, T, D,
                           reinterpret_cast<const long long*>(ids), mask, padding_idx, drop, inv_len, d_out, dtable);
        RBR_CHECK_LAUNCH("review_bag bwd (bucketed) launch");
        return 0;
    }
    // rocPRIM's radix sort re-initialises its state with hipMemsetAsync (device_radix_sort.hpp:122,251) whenever it takes
    // its onesweep path, and memset NODES fault when a recorded hipGraph is replayed on this stack (DESIGN.md section 4,
    // hipGraph).  While the stream is being captured the occurrences are therefore sorted with rocprim::merge_sort, which
    // issues kernels only -- a choice that does not depend on where rocPRIM switches algorithms.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    const size_t arr = (((size_t)n_pos * sizeof(int)) + 255) & ~(size_t)255;
    char* base = static_cast<char*>(ws);
    int* keys_in = reinterpret_cast<int*>(base);
    int* vals_in = reinterpret_cast<int*>(base + arr);
    int* keys = reinterpret_cast<int*>(base + 2 * arr);
    int* vals = reinterpret_cast<int*>(base + 3 * arr);
    void* temp = base + 4 * arr;
    size_t temp_bytes = bag_sort_temp_bytes(n_pos);
    const int sentinel = V;              // sorts behind every real token
    hipLaunchKernelGGL(bag_keys_kernel, dim3((unsigned)std::min<long>((n_pos + 255) / 256, 2048)), dim3(256), 0, st, n_pos,
                       reinterpret_cast<const long long*>(ids), mask, padding_idx, sentinel, keys_in, vals_in);
    RBR_CHECK_LAUNCH("review_bag keys launch");
    int bits = 1;
    while ((1L << bits) <= V) ++bits;    // keys are in [0, V]
    if (capturing) {
        if (int e = check_hip(rocprim::merge_sort(temp, temp_bytes, keys_in, keys, vals_in, vals, (size_t)n_pos, rocprim::less<int>(), st),
                              "review_bag merge sort"))
            return e;
    } else if (int e = check_hip(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys, vals_in, vals, (size_t)n_pos, 0, bits, st),
                                 "review_bag radix sort"))
        return e;
    const long n_waves = (n_pos + kBagChunk - 1) / kBagChunk;
    hipLaunchKernelGGL(bag_bwd_sorted_kernel, dim3((unsigned)((n_waves + kWavesPerWG - 1) / kWavesPerWG)), dim3(256), 0, st, n_pos, T,
                       D, sentinel, keys, vals, drop, inv_len, d_out, dtable);
    RBR_CHECK_LAUNCH("review_bag bwd launch");
    return 0;
}

static bool addatt_ok(int B, int R, int H, int K) {
    if (B <= 0 || R <= 0 || R > kAttMaxR || H <= 0 || K <= 0) { set_error("bad shape B=%d R=%d (<= %d) H=%d K=%d", B, R, kAttMaxR, H, K); return false; }
    if ((size_t)(R * H + R * K + 2 * R + H) * sizeof(float) > 64 * 1024 || (size_t)kDwRows * (K + H) * sizeof(float) > 64 * 1024) {
        set_error("R*H + R*K (or 64*(K+H)) exceeds the LDS budget");
        return false;
    }
    return true;
}

extern "C" int rbr_additive_attn_fwd(int32_t B, int32_t R, int32_t H, int32_t K, const float* rev, const uint8_t* mask,
                                     const float* node_drop, const float* Wp, const float* bp, const float* wi, float* out,
                                     float* scores, float* t_out, void* stream) {
    if (!addatt_ok(B, R, H, K)) return RBR_ERR_BAD_ARG;
    if (!rev || !Wp || !bp || !wi || !out || !scores || !t_out) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    const size_t lds = (size_t)(R * H + R * K + R) * sizeof(float);
    hipLaunchKernelGGL(addatt_fwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, B, R, H, K, rev, mask, node_drop, Wp, bp, wi,
                       out, scores, t_out);
    RBR_CHECK_LAUNCH("additive_attn fwd launch");
    return 0;
}

extern "C" size_t rbr_additive_attn_bwd_ws_floats(int32_t B, int32_t R, int32_t H, int32_t K) {
    if (B <= 0 || R <= 0 || H <= 0 || K <= 0) return 0;
    return (size_t)B * R * K + (size_t)B * R * H;          // d_pre | x (node dropout applied)
}

extern "C" int rbr_additive_attn_bwd(int32_t B, int32_t R, int32_t H, int32_t K, const float* rev, const uint8_t* mask,
                                     const float* node_drop, const float* Wp, const float* wi, const float* scores,
                                     const float* t_in, const float* d_out, float* d_rev, float* d_Wp, float* d_bp, float* d_wi,
                                     float* ws, void* stream) {
    if (!addatt_ok(B, R, H, K)) return RBR_ERR_BAD_ARG;
    if (!rev || !Wp || !wi || !scores || !t_in || !d_out || !d_rev || !d_Wp || !d_bp || !d_wi || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipStream_t st = (hipStream_t)stream;
    float* d_pre = ws;
    float* x_eff = ws + (size_t)B * R * K;
    ZeroRegions zr{{reinterpret_cast<int*>(d_bp), reinterpret_cast<int*>(d_wi), reinterpret_cast<int*>(d_Wp)}, {K, K, (long)K * H}};
    if (int e = zero_regions(zr, st)) return e;
    const size_t lds = (size_t)(R * H + R * K + 2 * R + H) * sizeof(float);
    hipLaunchKernelGGL(addatt_bwd_kernel, dim3(B), dim3(256), lds, st, B, R, H, K, rev, mask, node_drop, Wp, wi, scores, t_in, d_out,
                       d_rev, d_pre, x_eff, d_bp, d_wi);
    RBR_CHECK_LAUNCH("additive_attn bwd launch");
    hipLaunchKernelGGL(addatt_dw_kernel, dim3((B * R + kDwRows - 1) / kDwRows), dim3(256), (size_t)kDwRows * (K + H) * sizeof(float), st,
                       B * R, H, K, d_pre, x_eff, d_Wp);
    RBR_CHECK_LAUNCH("additive_attn dW launch");
    return 0;
}
