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
};

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
    const float s = block_sum(acc, s_red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = s;
        if (blockIdx.x == 0) *step += 1.f;      // consumed by clip_adam_kernel, which runs after this grid
    }
}

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
    const double t = (double)*step;
    const float bc1 = (float)(1.0 - pow((double)beta1, t));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, t));
    const float step_size = lr / bc1;
    const float w1 = 1.f - beta1, w2 = 1.f - beta2;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int k = tensor_of_chunk(T, c);
        const long e0 = (c - T.chunk0[k]) * kOptChunk;
        const long n = min((long)kOptChunk, T.n[k] - e0);
        float* p = T.p[k] + e0;
        float* g = T.g[k] + e0;
        float* m = T.m[k] + e0;
        float* v = T.v[k] + e0;
        const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
        const long n4 = al ? (n >> 2) : 0;
#define RBR_ADAM1(G, M, V, P, c)                                       \
            G.c *= coef;                                               \
            M.c = M.c + w1 * (G.c - M.c);                              \
            V.c = beta2 * V.c + w2 * G.c * G.c;                        \
            P.c -= step_size * M.c / (sqrtf(V.c) / bc2_sqrt + eps);
        if (al && n == kOptChunk) {          // whole chunk: all 16 loads of a thread in flight before the first use
            constexpr int U = kOptChunk / 1024;
            float4 G[U], M[U], V[U], P[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = threadIdx.x + 256 * u;
                G[u] = reinterpret_cast<float4*>(g)[i]; M[u] = reinterpret_cast<float4*>(m)[i];
                V[u] = reinterpret_cast<float4*>(v)[i]; P[u] = reinterpret_cast<float4*>(p)[i];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = threadIdx.x + 256 * u;
                RBR_ADAM1(G[u], M[u], V[u], P[u], x) RBR_ADAM1(G[u], M[u], V[u], P[u], y)
                RBR_ADAM1(G[u], M[u], V[u], P[u], z) RBR_ADAM1(G[u], M[u], V[u], P[u], w)
                if (clipped) reinterpret_cast<float4*>(g)[i] = G[u];
                reinterpret_cast<float4*>(m)[i] = M[u];
                reinterpret_cast<float4*>(v)[i] = V[u];
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
            const float gg = g[i] * coef;
            const float mm = m[i] + w1 * (gg - m[i]);
            const float vv = beta2 * v[i] + w2 * gg * gg;
            if (clipped) g[i] = gg;
            m[i] = mm; v[i] = vv;
            p[i] -= step_size * mm / (sqrtf(vv) / bc2_sqrt + eps);
        }
    }
}

}  // namespace rbr

using namespace rbr;

extern "C" size_t rbr_clip_adam_ws_floats(void) { return kOptMaxPartials; }

extern "C" int rbr_clip_adam_step(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const int64_t* numel, float max_norm, float lr, float beta1,
                                  float beta2, float eps, float* step, float* gnorm_out, float* ws, void* stream) {
    if (n_tensors <= 0 || n_tensors > RBR_OPT_MAX_TENSORS) { set_error("n_tensors=%d (1..%d)", n_tensors, RBR_OPT_MAX_TENSORS); return RBR_ERR_BAD_ARG; }
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !step || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    OptTable T{};
    T.count = n_tensors;
    long chunks = 0;
    for (int k = 0; k < n_tensors; ++k) {
        if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || numel[k] <= 0) { set_error("tensor %d: null pointer or empty", k); return RBR_ERR_BAD_ARG; }
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
    hipLaunchKernelGGL(clip_adam_kernel, dim3(nb2), dim3(256), 0, st, T, chunks, ws, nb1, max_norm, lr, beta1, beta2, eps, step,
                       gnorm_out);
    RBR_CHECK_LAUNCH("clip_adam launch");
    return 0;
}
