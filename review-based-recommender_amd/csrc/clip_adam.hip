// clip_adam.hip -- the trainer's  clip_grad_norm_(params, max_norm); Adam.step()  (trainer/train_deepconn_pp.py:166-167,
// optimizer built at :135) as two launches over all parameter tensors.
//
// torch runs it as ~10 kernels: per-tensor norms, stack, norm of norms, +eps, reciprocal, clamp, one multiply pass over
// every gradient, the fused Adam pass, the step counters.  At the cfg2 shape the word table makes every full pass
// 60 MB, so the sequence moves 60 (norm) + 120 (scale in place) + 420 (Adam) MB.  Here:
//   grad_sqnorm : sum of squares of every gradient, fixed-order partials (no atomics: bitwise reproducible)
//   clip_adam   : every workgroup reduces the partials to the total norm, derives the clip coefficient, and applies
//                 g*coef -> exp_avg, exp_avg_sq, param in ONE pass (the clipped gradient is written back, as
//                 clip_grad_norm_ leaves it, unless the coefficient is exactly 1): 60 + 420..480 MB.
// Math as torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False):
//   m = m + (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// The step count t lives in device memory (a float, as torch's capturable Adam keeps it) and is advanced by the
// first kernel, so the pair can be recorded into a hipGraph.
#include "rbr_common.h"

#include <algorithm>

namespace rbr {

constexpr int kOptChunk = 4096;          // elements per work chunk
constexpr int kOptMaxPartials = 1024;

struct OptTable {
    float* p[RBR_OPT_MAX_TENSORS];
    float* g[RBR_OPT_MAX_TENSORS];
    float* m[RBR_OPT_MAX_TENSORS];
    float* v[RBR_OPT_MAX_TENSORS];
    long n[RBR_OPT_MAX_TENSORS];
    long chunk0[RBR_OPT_MAX_TENSORS + 1];      // first global chunk of tensor k (prefix sums)
    int count;
    // Tensor `rows_k` (-1: none) is an embedding table [V, rows_D] whose gradient arrives in COMPACT form (the token-product
    // conv's backward, rbr_textcnn_bwd_dtable_prod_ex with RBR_G_ROWS): g[v, :] = grows[row_of_token[v], :] when the batch holds
    // token v (row_of_token[v] >= 0), and exactly 0 otherwise -- the same update formula with g = 0, so the result is the dense
    // path's bit for bit, without reading (or, when clipping, re-writing) the ~57 % of the dense gradient that is zeros.
    // sq_part[n_sq]: the producer's partial sums of squares of grows (the table's share of the norm).
    int rows_k, rows_D, n_sq;
    const int* row_of_token;
    float* grows;
    const float* sq_part;
};

// One element's update, every operation individually rounded (no contraction): the dense path and the row-gradient path (g = 0
// for the rows of absent tokens) must produce the same bits for the same inputs, whatever code surrounds the call.
struct AdamK { float coef, w1, beta2, w2, step_size, bc2_sqrt, eps; };
__device__ __forceinline__ void adam1(const AdamK& K, float& g, float& m, float& v, float& p) {
    // plain operators under contract(off): every operation rounds on its own at every call site (HIP's __fmul_rn / __fadd_rn are
    // ordinary inline functions whose operators carry the header's contraction permission -- they still fuse after inlining)
#pragma clang fp contract(off)
    g = g * K.coef;
    m = m + K.w1 * (g - m);
    v = K.beta2 * v + (K.w2 * g) * g;
    p = p - (K.step_size * m) / (sqrtf(v) / K.bc2_sqrt + K.eps);
}
__device__ __forceinline__ AdamK adam_constants(float coef, float lr, float beta1, float beta2, float eps, double t) {
    const float bc1 = (float)(1.0 - pow((double)beta1, t));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, t));
    return AdamK{coef, 1.f - beta1, beta2, 1.f - beta2, lr / bc1, bc2_sqrt, eps};
}

// Adam's moments are touched once per step and by nobody else: streamed past the caches (nontemporal), so that they do not evict
// what the next step's forward wants to find (the parameters -- read again by the forward -- keep ordinary accesses)
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* base, int i) {
    const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(base) + i);
    return float4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void nt_store4(float* base, int i, const float4& x) {
    __builtin_nontemporal_store(f32x4_nt{x.x, x.y, x.z, x.w}, reinterpret_cast<f32x4_nt*>(base) + i);
}

__device__ __forceinline__ int tensor_of_chunk(const OptTable& T, long chunk) {
    int k = 0;
    for (int i = 1; i < T.count; ++i)        // <= 64 tensors: a short scalar scan
        if (chunk >= T.chunk0[i]) k = i;
    return k;
}

__device__ __forceinline__ float block_sum(float x, float* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) s_red[wave] = x;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];        // fixed order
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const OptTable T, long nchunks, float* __restrict__ partials,
                                                          float* __restrict__ step) {
    __shared__ float s_red[4];
    float acc = 0.f;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int k = tensor_of_chunk(T, c);
        if (k == T.rows_k) {                 // compact-gradient table: its sum of squares comes as partials (below)
            continue;
        }
        const long e0 = (c - T.chunk0[k]) * kOptChunk;
        const long n = min((long)kOptChunk, T.n[k] - e0);
        const float* g = T.g[k] + e0;
        if ((((uintptr_t)g) & 15) == 0) {
            if (n == kOptChunk) {            // whole chunk: the four loads of a thread are issued together
                float4 x[kOptChunk / 1024];
#pragma unroll
                for (int u = 0; u < kOptChunk / 1024; ++u) x[u] = reinterpret_cast<const float4*>(g)[threadIdx.x + 256 * u];
#pragma unroll
                for (int u = 0; u < kOptChunk / 1024; ++u) acc += x[u].x * x[u].x + x[u].y * x[u].y + x[u].z * x[u].z + x[u].w * x[u].w;
                continue;
            }
            const long n4 = n >> 2;
            for (long i = threadIdx.x; i < n4; i += 256) {
                const float4 x = reinterpret_cast<const float4*>(g)[i];
                acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            }
            for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
        } else {
            for (long i = threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
        }
    }
    if (T.rows_k >= 0)                       // fixed partition of the producer's partials over the grid: reproducible
        for (int i = blockIdx.x * 256 + threadIdx.x; i < T.n_sq; i += gridDim.x * 256) acc += T.sq_part[i];
    const float s = block_sum(acc, s_red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = s;
        if (blockIdx.x == 0) *step += 1.f;      // consumed by clip_adam_kernel, which runs after this grid
    }
}

// MODE: 0 no row-gradient tensor (every tensor dense), 1 one table with a compact row gradient.  A template parameter: as a
// run-time branch of one kernel the row form cost the dense one half its occupancy (226 VGPRs against 104).
// (Measured and dropped in round 3: updating the rows of the batch's ABSENT tokens early -- their gradient is zero whatever the
// backward computes -- on another stream beside the GEMM or beside the head kernels, and walking only the listed rows here
// (52 us instead of 77): the 41-55 us stream of the early kernel slowed whatever it ran beside by as much as it saved
// (GEMM 61 -> 89 us; head_fwd 24 -> 46 us): step 0.394 / 0.402 ms against 0.385.)
template <int MODE>
__global__ __launch_bounds__(256) void clip_adam_kernel(const OptTable T, long nchunks, const float* __restrict__ partials,
                                                        int npartials, float max_norm, float lr, float beta1, float beta2,
                                                        float eps, const float* __restrict__ step, float* __restrict__ gnorm_out) {
    __shared__ float s_red[4];
    float ps = 0.f;
    for (int i = threadIdx.x; i < npartials; i += 256) ps += partials[i];
    const float total = sqrtf(block_sum(ps, s_red));
    if (blockIdx.x == 0 && threadIdx.x == 0 && gnorm_out != nullptr) *gnorm_out = total;
    // clip_grad_norm_: coef = clamp(max_norm / (total + 1e-6), max = 1)
    const float coef = (max_norm > 0.f) ? fminf(max_norm / (total + 1e-6f), 1.f) : 1.f;
    const bool clipped = coef != 1.f;          // workgroup-uniform
    const AdamK K = adam_constants(coef, lr, beta1, beta2, eps, (double)*step);
#define RBR_ADAM1(G, M, V, P, c) adam1(K, G.c, M.c, V.c, P.c);
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int k = tensor_of_chunk(T, c);
        const long e0 = (c - T.chunk0[k]) * kOptChunk;
        const long n = min((long)kOptChunk, T.n[k] - e0);
        float* p = T.p[k] + e0;
        float* g = T.g[k] + e0;
        float* m = T.m[k] + e0;
        float* v = T.v[k] + e0;
        if (MODE == 1 && k == T.rows_k && n == kOptChunk && T.rows_D >= 256) {      // (shorter rows: a wave spans more than two)
            // Compact-gradient table, whole chunk (rows_D % 4 == 0, V * D < 2^32 and 16-byte aligned pointers: checked on the host).
            // The 64 float4 of a wave and round lie in at most TWO vocabulary rows, so the two list rows are WAVE-UNIFORM: they
            // come through the scalar cache (s_load), not as a per-lane gather, and a wave whose two rows are both absent from
            // the batch issues no gradient load at all -- this form issues fewer vector-memory instructions than the dense one
            // (the dense kernel's 16 loads + 16 stores per thread are what its 77 us are made of).
            const int D = T.rows_D;
            constexpr int U = 4;
            float4 G[U], M[U], V[U], P[U];
            const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = threadIdx.x + 256 * u;
                M[u] = nt_load4(m, i); V[u] = nt_load4(v, i); P[u] = reinterpret_cast<float4*>(p)[i];
            }
            int rl[U];
            unsigned ofs[U];
            bool any[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned eA = (unsigned)(e0 + 4 * (wbase + 256 * u));            // first element of the wave's 256 (uniform)
                const unsigned tokA = (unsigned)__builtin_amdgcn_readfirstlane((int)(eA / (unsigned)D));     // uniform -> SGPR
                const unsigned nextrow = (tokA + 1) * (unsigned)D;                      // first element of the following row
                const int rA = T.row_of_token[tokA];                                    // scalar load
                const unsigned tokB = (eA + 255u >= nextrow) ? tokA + 1 : tokA;        // (tokB < V: the chunk lies inside the tensor)
                const int rB = T.row_of_token[tokB];
                const unsigned e = eA + 4u * (unsigned)lane;
                const bool second = e >= nextrow;
                rl[u] = second ? rB : rA;
                ofs[u] = second ? e - nextrow : e - tokA * (unsigned)D;
                any[u] = (rA >= 0) | (rB >= 0);                                         // wave-uniform
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                G[u] = float4{0.f, 0.f, 0.f, 0.f};
                if (any[u]) {                                                           // scalar branch: whole waves skip the load
                    const float4 gq = *reinterpret_cast<const float4*>(T.grows + (long)max(rl[u], 0) * D + ofs[u]);
                    if (rl[u] >= 0) G[u] = gq;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = threadIdx.x + 256 * u;
                RBR_ADAM1(G[u], M[u], V[u], P[u], x) RBR_ADAM1(G[u], M[u], V[u], P[u], y)
                RBR_ADAM1(G[u], M[u], V[u], P[u], z) RBR_ADAM1(G[u], M[u], V[u], P[u], w)
                if (clipped && rl[u] >= 0) *reinterpret_cast<float4*>(T.grows + (long)rl[u] * D + ofs[u]) = G[u];
                nt_store4(m, i, M[u]);
                nt_store4(v, i, V[u]);
                reinterpret_cast<float4*>(p)[i] = P[u];
            }
            continue;
        }
        if (MODE == 1 && k == T.rows_k) {                 // the table's last, partial chunk (and short rows): one float4 at a time
            const int D = T.rows_D;
            const int n4r = (int)(n >> 2);               // numel = V * D is a multiple of 4
            for (int i = threadIdx.x; i < n4r; i += 256) {
                const unsigned e = (unsigned)(e0 + 4 * i);
                const unsigned tok = e / (unsigned)D;
                const int r = T.row_of_token[tok];
                const long go = (long)max(r, 0) * D + (e - tok * (unsigned)D);
                float4 G = *reinterpret_cast<const float4*>(T.grows + go);
                if (r < 0) G = float4{0.f, 0.f, 0.f, 0.f};
                float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i], P = reinterpret_cast<float4*>(p)[i];
                RBR_ADAM1(G, M, V, P, x) RBR_ADAM1(G, M, V, P, y) RBR_ADAM1(G, M, V, P, z) RBR_ADAM1(G, M, V, P, w)
                if (clipped && r >= 0) *reinterpret_cast<float4*>(T.grows + go) = G;
                reinterpret_cast<float4*>(m)[i] = M;
                reinterpret_cast<float4*>(v)[i] = V;
                reinterpret_cast<float4*>(p)[i] = P;
            }
            continue;
        }
        const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
        const long n4 = al ? (n >> 2) : 0;
        if (al && n == kOptChunk) {          // whole chunk: all 16 loads of a thread in flight before the first use
            constexpr int U = kOptChunk / 1024;
            float4 G[U], M[U], V[U], P[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = threadIdx.x + 256 * u;
                G[u] = reinterpret_cast<float4*>(g)[i]; M[u] = nt_load4(m, i);
                V[u] = nt_load4(v, i); P[u] = reinterpret_cast<float4*>(p)[i];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = threadIdx.x + 256 * u;
                RBR_ADAM1(G[u], M[u], V[u], P[u], x) RBR_ADAM1(G[u], M[u], V[u], P[u], y)
                RBR_ADAM1(G[u], M[u], V[u], P[u], z) RBR_ADAM1(G[u], M[u], V[u], P[u], w)
                if (clipped) reinterpret_cast<float4*>(g)[i] = G[u];
                nt_store4(m, i, M[u]);
                nt_store4(v, i, V[u]);
                reinterpret_cast<float4*>(p)[i] = P[u];
            }
            continue;
        }
        for (long i = threadIdx.x; i < n4; i += 256) {
            float4 G = reinterpret_cast<float4*>(g)[i], M = reinterpret_cast<float4*>(m)[i];
            float4 V = reinterpret_cast<float4*>(v)[i], P = reinterpret_cast<float4*>(p)[i];
            RBR_ADAM1(G, M, V, P, x) RBR_ADAM1(G, M, V, P, y) RBR_ADAM1(G, M, V, P, z) RBR_ADAM1(G, M, V, P, w)
            if (clipped) reinterpret_cast<float4*>(g)[i] = G;       // coef == 1: .grad already holds the "clipped" gradient
            reinterpret_cast<float4*>(m)[i] = M;
            reinterpret_cast<float4*>(v)[i] = V;
            reinterpret_cast<float4*>(p)[i] = P;
        }
#undef RBR_ADAM1
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            float gg = g[i], mm = m[i], vv = v[i], pp = p[i];
            adam1(K, gg, mm, vv, pp);
            if (clipped) g[i] = gg;
            m[i] = mm; v[i] = vv; p[i] = pp;
        }
    }
}

}  // namespace rbr

using namespace rbr;

extern "C" size_t rbr_clip_adam_ws_floats(void) { return kOptMaxPartials; }

static int clip_adam_step(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                          float* const* exp_avg_sq, const int64_t* numel, float max_norm, float lr, float beta1,
                          float beta2, float eps, float* step, float* gnorm_out, float* ws, void* stream, const rbr_row_grad* rg) {
    if (n_tensors <= 0 || n_tensors > RBR_OPT_MAX_TENSORS) { set_error("n_tensors=%d (1..%d)", n_tensors, RBR_OPT_MAX_TENSORS); return RBR_ERR_BAD_ARG; }
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !step || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    OptTable T{};
    T.count = n_tensors;
    T.rows_k = -1;
    if (rg != nullptr) {
        const int k = rg->tensor;
        if (k < 0 || k >= n_tensors || rg->D <= 0 || rg->D % 4 != 0 || rg->V <= 0 || (int64_t)rg->V * rg->D != numel[k] || numel[k] >= ((int64_t)1 << 32) || !rg->row_of_token ||
            !rg->rows || !rg->sq_part || rg->n_sq <= 0 || ((((uintptr_t)rg->rows) | ((uintptr_t)params[k]) | ((uintptr_t)exp_avg[k]) |
                                                            ((uintptr_t)exp_avg_sq[k])) & 15) != 0) {
            set_error("clip_adam_step_rows: malformed row gradient (tensor %d, V=%d, D=%d)", k, rg->V, rg->D);
            return RBR_ERR_BAD_ARG;
        }
        T.rows_k = k; T.rows_D = rg->D; T.n_sq = rg->n_sq; T.row_of_token = rg->row_of_token; T.grows = rg->rows; T.sq_part = rg->sq_part;
    }
    long chunks = 0;
    for (int k = 0; k < n_tensors; ++k) {
        if (!params[k] || (!grads[k] && k != T.rows_k) || !exp_avg[k] || !exp_avg_sq[k] || numel[k] <= 0) { set_error("tensor %d: null pointer or empty", k); return RBR_ERR_BAD_ARG; }
        T.p[k] = params[k]; T.g[k] = grads[k]; T.m[k] = exp_avg[k]; T.v[k] = exp_avg_sq[k]; T.n[k] = numel[k];
        T.chunk0[k] = chunks;
        chunks += (numel[k] + kOptChunk - 1) / kOptChunk;
    }
    T.chunk0[n_tensors] = chunks;
    hipStream_t st = (hipStream_t)stream;
    const int nb1 = (int)std::min<long>(chunks, kOptMaxPartials);
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(nb1), dim3(256), 0, st, T, chunks, ws, step);
    RBR_CHECK_LAUNCH("grad_sqnorm launch");
    const int nb2 = (int)std::min<long>(chunks, 4096);
    if (T.rows_k < 0)
        hipLaunchKernelGGL(clip_adam_kernel<0>, dim3(nb2), dim3(256), 0, st, T, chunks, ws, nb1, max_norm, lr, beta1, beta2, eps, step, gnorm_out);
    else
        hipLaunchKernelGGL(clip_adam_kernel<1>, dim3(nb2), dim3(256), 0, st, T, chunks, ws, nb1, max_norm, lr, beta1, beta2, eps, step, gnorm_out);
    RBR_CHECK_LAUNCH("clip_adam launch");
    return 0;
}

extern "C" int rbr_clip_adam_step(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const int64_t* numel, float max_norm, float lr, float beta1,
                                  float beta2, float eps, float* step, float* gnorm_out, float* ws, void* stream) {
    return clip_adam_step(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, max_norm, lr, beta1, beta2, eps, step, gnorm_out, ws,
                          stream, nullptr);
}

extern "C" int rbr_clip_adam_step_rows(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                                       float* const* exp_avg_sq, const int64_t* numel, float max_norm, float lr, float beta1,
                                       float beta2, float eps, float* step, float* gnorm_out, float* ws, const rbr_row_grad* rg,
                                       void* stream) {
    if (!rg) { set_error("clip_adam_step_rows: null row gradient"); return RBR_ERR_BAD_ARG; }
    return clip_adam_step(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, max_norm, lr, beta1, beta2, eps, step, gnorm_out, ws,
                          stream, rg);
}

// dense[v, :] = rows[row_of_token[v], :] for the tokens the batch holds, 0 elsewhere: the [V, D] gradient nn.Embedding's backward
// would have produced, for whoever wants it outside the fused optimizer step (tests, logging, a dense all-reduce).
namespace rbr {
__global__ __launch_bounds__(256) void rows_to_dense_kernel(int V, int D4, const int* __restrict__ row_of_token,
                                                            const float4* __restrict__ rows, float4* __restrict__ dense) {
    const long n = (long)V * D4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int v = (int)(i / D4);
        const int r = row_of_token[v];
        dense[i] = (r >= 0) ? rows[(long)r * D4 + (i - (long)v * D4)] : float4{0.f, 0.f, 0.f, 0.f};
    }
}
}  // namespace rbr

extern "C" int rbr_row_grad_to_dense(int32_t V, int32_t D, const int32_t* row_of_token, const float* rows, float* dense, void* stream) {
    if (V <= 0 || D <= 0 || D % 4 != 0 || !row_of_token || !rows || !dense || ((((uintptr_t)rows) | ((uintptr_t)dense)) & 15) != 0) {
        set_error("row_grad_to_dense: bad arguments (V=%d, D=%d)", V, D);
        return RBR_ERR_BAD_ARG;
    }
    const long n = (long)V * (D / 4);
    hipLaunchKernelGGL(rows_to_dense_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, V,
                       D / 4, row_of_token, reinterpret_cast<const float4*>(rows), reinterpret_cast<float4*>(dense));
    RBR_CHECK_LAUNCH("rows_to_dense launch");
    return 0;
}
