// dense_misc.hip -- the remaining small layers of the path:
//   * nn.Linear (+ReLU, +dropout multiplier) forward/backward on the f32 MFMA pipe:
//       D-ATT's shared fc (dual_att.py:31-35: Linear 500->500, ReLU, Dropout, Linear 500->50) and
//       HierPooling's optional projection (deepconn/layers.py:76-79,96);
//   * standalone embedding row gather / scatter-add (WordEmbedding.forward, layers.py:22-24), for
//     callers that want the materialised rows;
//   * NgramFeat arch="HierPooling" (layers.py:62-98,110-114): sliding-window mean + global max.
#include "rbr_common.h"

#include <type_traits>

namespace rbr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------ GEMM
// C[M,N] = A . B^T-like with arbitrary element strides: A(m,k) = A[m*sam + k*sak], B(n,k) = B[n*sbn + k*sbk].
// 64x64 output tile per workgroup (4 waves x 32x32 MFMA tile), K in chunks of 32 staged through LDS
// (row stride 36 floats = 4*odd: conflict-free ds_read_b128).  Sizes here are tiny (<= 1 GFLOP), so the
// kernel is written for generality (any strides / bounds), not for the last 20 % of the MFMA roofline.
// (Measured and dropped, round 3: 32x32 tiles with K split over the workgroup's four waves, each wave through an LDS region of its
// own and the partial tiles meeting in wave order, for the shapes that leave the chip idle under 64x64 tiles -- D-ATT's fc is 128
// tiles, its weight gradients 64.  18.2 us against 17.8 us per launch: a wave then issues the loads and LDS writes of a whole
// 32x32 tile per chunk, one dword at a time, and that instruction stream costs what the shorter MFMA chain saves.)
constexpr int GK = 32, GS = 36;

struct Epi {
    const float* bias;   // [N] or null
    const float* mul;    // [M,N] multiplier (dropout) or null
    int relu;            // activation: 0 none, 1 ReLU, 2 Tanh
};

// Split K (gridDim.z = slices > 1): D-ATT's shared fc is 128 output tiles, its weight gradients 64 and 8 -- a fraction of the
// chip's 256 CUs, each tile a chain of K / 32 barrier-separated chunks on ONE wave per SIMD.  Slice z multiplies K range
// [z * kper, (z + 1) * kper) into a partial tile in `part` ([slice][tile][4 waves][16][64] floats); gemm_reduce_kernel adds the
// partial tiles in slice order and applies the epilogue.  (Measured and dropped: ONE launch in which the last slice of a tile
// to arrive -- a ticket per tile -- does the reduction: the release / acquire fences that make the partial tiles visible across
// the 8 XCDs write back and invalidate whole L2s, and the 1024 x 500 x 500 fc took 84 us instead of 24.)
// One product as a kernel argument (the launcher fills it; tx x ty output tiles, `slices` K slices)
struct GemmJob {
    int M, N, K_all, kper, tx, ty, slices;
    long sam, sak, sbn, sbk, ldc, bytes_a, bytes_b;
    const float* A;
    const float* B;
    float* C;
    float* part;
    Epi ep;
};

// workgroup (bx, by, bz) of product J
__device__ __forceinline__ void gemm_body(const GemmJob& J, const int bx, const int by, const int bz, float* __restrict__ As,
                                          float* __restrict__ Bs) {
    const int M = J.M, N = J.N, K_all = J.K_all, kper = J.kper;
    const long sam = J.sam, sak = J.sak, sbn = J.sbn, sbk = J.sbk, ldc = J.ldc, bytes_a = J.bytes_a, bytes_b = J.bytes_b;
    const float* __restrict__ A = J.A;
    const float* __restrict__ B = J.B;
    float* __restrict__ C = J.C;
    float* __restrict__ part = J.part;
    const Epi ep = J.ep;
    const bool sliced = J.slices > 1;
    const int kb = sliced ? bz * kper : 0;
    const int K = sliced ? min(K_all - kb, kper) : K_all;          // this slice's K extent (kper is a multiple of GK)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int m0 = by * 64, n0 = bx * 64;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // element e = tid + 256 q of a 64 x GK chunk: lanes run along the contiguous dimension of the operand
    int arow[8], akk[8], brow[8], bkk[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 256 * q;
        arow[q] = (sak == 1) ? e / GK : e % 64; akk[q] = (sak == 1) ? e % GK : e / 64;
        brow[q] = (sbk == 1) ? e / GK : e % 64; bkk[q] = (sbk == 1) ? e % GK : e / 64;
    }
    // two K chunks of global loads in flight (register sets 0 / 1) while a third is multiplied out of LDS.
    // Buffer loads: the per-lane byte offset of an element at k0 = 0 is formed once, a chunk only moves the scalar offset, and
    // rows outside the operand carry an offset past its end (the range check returns 0): no per-chunk address arithmetic and no
    // predicated loads.  (Flat loads with per-element 64-bit multiplies kept the VALU as busy as the matrix pipe, and a
    // predicated load compiles to a branch, behind which the compiler drains every load in flight before the LDS writes.)
    float ra[2][8], rb[2][8];
    const __amdgpu_buffer_rsrc_t ra_desc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)bytes_a, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_desc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, (int)bytes_b, 0x00020000);
    constexpr int kOutside = 0x7f000000;             // beyond every operand (launch_gemm: < 2^30 bytes), no wrap with the chunk offset
    int offa[8], offb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int m = m0 + arow[q], n = n0 + brow[q];
        offa[q] = (m < M) ? (int)((m * sam + (kb + akk[q]) * sak) * 4) : kOutside;
        offb[q] = (n < N) ? (int)((n * sbn + (kb + bkk[q]) * sbk) * 4) : kOutside;
    }
    auto fetch = [&](int k0, float (&fa)[8], float (&fb)[8]) {       // one straight-line path: 16 loads
        const int sa = (int)(k0 * sak * 4), sb = (int)(k0 * sbk * 4);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            fa[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra_desc, offa[q], sa, 0));
            fb[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb_desc, offb[q], sb, 0));
        }
    };
    // `masked`: the K tail (past K a row-major operand continues into its next row).  The full chunks of the main loop write
    // the loaded values as they are: a select in front of the LDS write is a use the scheduler hoists into the previous
    // chunk's MFMAs, waiting for the loads a whole chunk early.
    auto chunk = [&](int k0, auto masked, float (&fa)[8], float (&fb)[8]) {
        __syncthreads();       // previous chunk's LDS reads are done
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if constexpr (decltype(masked)::value) {
                As[arow[q] * GS + akk[q]] = (k0 + akk[q] < K) ? fa[q] : 0.f;
                Bs[brow[q] * GS + bkk[q]] = (k0 + bkk[q] < K) ? fb[q] : 0.f;
            } else {
                As[arow[q] * GS + akk[q]] = fa[q];
                Bs[brow[q] * GS + bkk[q]] = fb[q];
            }
        }
        __syncthreads();
        fetch(k0 + 2 * GK, fa, fb);   // ALWAYS 16 loads (past the end they are never used): a conditional fetch leaves the number
                                      // of loads in flight unknown at the next LDS write, and the compiler then drains them all
        const float* pa = As + (wm * 32 + i) * GS + 4 * h;
        const float* pb = Bs + (wn * 32 + i) * GS + 4 * h;
#pragma unroll
        for (int q = 0; q < GK / 8; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(pa + 8 * q);
            const f32x4 b = *reinterpret_cast<const f32x4*>(pb + 8 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    };
    fetch(0, ra[0], rb[0]);
    __builtin_amdgcn_sched_barrier(0);               // set 0's loads strictly before set 1's, as on the loop's back edge: the
    fetch(GK, ra[1], rb[1]);                         // waits at the loop head are counted against the worse of the two orders
    __builtin_amdgcn_sched_barrier(0);
    const int k_full = (K / GK) * GK;                // [0, k_full): whole chunks
    int k0 = 0;
    for (; k0 + 2 * GK <= k_full; k0 += 2 * GK) {    // pairs of whole chunks: one straight-line body, register sets 0 / 1 never merge
        chunk(k0, std::false_type{}, ra[0], rb[0]);
        chunk(k0 + GK, std::false_type{}, ra[1], rb[1]);
    }
    if (k0 < K) {                                    // at most one whole chunk and the tail are left
        chunk(k0, std::true_type{}, ra[0], rb[0]);
        if (k0 + GK < K) chunk(k0 + GK, std::true_type{}, ra[1], rb[1]);
    }
    if (sliced) {                                     // a K slice: the partial tile goes to `part`, the reduce kernel finishes
        const int tile = by * J.tx + bx, ntiles = J.tx * J.ty;
        float* mine = part + (((long)bz * ntiles + tile) * 4 + wave) * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[r * 64] = acc[r];
        return;
    }
    const int n = n0 + wn * 32 + i;
    if (n >= N) return;
    const float bv = ep.bias ? ep.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) {
            float v = acc[r] + bv;
            if (ep.relu == 1) v = fmaxf(v, 0.f);
            else if (ep.relu == 2) v = tanhf(v);
            if (ep.mul) v *= ep.mul[(long)m * N + n];
            C[(long)m * ldc + n] = v;
        }
    }
}

__global__ __launch_bounds__(256) void gemm_kernel(const GemmJob J) {
    __shared__ __attribute__((aligned(16))) float As[64 * GS];
    __shared__ __attribute__((aligned(16))) float Bs[64 * GS];
    gemm_body(J, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Two independent products in one launch (a layer's weight gradient and its input gradient: same g, nothing else shared):
// the workgroups of J0 first, then J1's, in a flat grid.  Two inlined bodies behind a workgroup-uniform branch -- a select
// between the two argument structs would copy them to scratch (rbr_launch.h).
__global__ __launch_bounds__(256) void gemm2_kernel(const GemmJob J0, const GemmJob J1) {
    __shared__ __attribute__((aligned(16))) float As[64 * GS];
    __shared__ __attribute__((aligned(16))) float Bs[64 * GS];
    const int n0 = J0.tx * J0.ty * J0.slices;
    int b = blockIdx.x;
    if (b < n0) {
        const int t = J0.tx * J0.ty;
        gemm_body(J0, (b % t) % J0.tx, (b % t) / J0.tx, b / t, As, Bs);
    } else {
        b -= n0;
        const int t = J1.tx * J1.ty;
        gemm_body(J1, (b % t) % J1.tx, (b % t) / J1.tx, b / t, As, Bs);
    }
}

// C tile = epilogue(sum over the slices, in slice order); same (wave, register, lane) -> (m, n) map as gemm_kernel's epilogue
// One WAVE per 32 x 32 quadrant (grid z = quadrant): four times the workgroups of the product's grid, every load of a lane in flight
// at once -- the kernel is two memory round trips long.
struct ReduceJob {                // the reduction of one sliced product: tx x ty tiles (0 tiles: nothing to reduce)
    int M, N, slices, tx, ty;
    const float* part;
    float* C;
    long ldc;
    Epi ep;
};

// quadrant `wave` of tile (bx, by)
__device__ __forceinline__ void gemm_reduce_body(const ReduceJob& J, const int bx, const int by, const int wave, const int lane) {
    const int M = J.M, N = J.N, slices = J.slices;
    const float* __restrict__ part = J.part;
    float* __restrict__ C = J.C;
    const long ldc = J.ldc;
    const Epi ep = J.ep;
    const int i = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int tile = by * J.tx + bx, ntiles = J.tx * J.ty;
    const int m0 = by * 64, n0 = bx * 64;
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int n = n0 + wn * 32 + i;
    const float bv = (ep.bias && n < N) ? ep.bias[n] : 0.f;
    float mulv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        mulv[r] = (ep.mul && m < M && n < N) ? ep.mul[(long)m * N + n] : 1.f;
    }
    for (int z = 0; z < slices; z += 2) {            // two slices (32 loads) per round, added in slice order
        const float* p0 = part + (((long)z * ntiles + tile) * 4 + wave) * 1024 + lane;
        const float* p1 = part + (((long)min(z + 1, slices - 1) * ntiles + tile) * 4 + wave) * 1024 + lane;
        float v0[16], v1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { v0[r] = p0[r * 64]; v1[r] = p1[r * 64]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] += v0[r];
            if (z + 1 < slices) acc[r] += v1[r];
        }
    }
    if (n >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) {
            float v = acc[r] + bv;
            if (ep.relu == 1) v = fmaxf(v, 0.f);
            else if (ep.relu == 2) v = tanhf(v);
            C[(long)m * ldc + n] = v * mulv[r];
        }
    }
}

__global__ __launch_bounds__(64) void gemm_reduce_kernel(const ReduceJob J) {
    gemm_reduce_body(J, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x);
}

// slices and K per slice for an M x N x K product: enough workgroups for two per CU, at least two chunks of K per slice
static void gemm_split(int M, int N, int K, int& slices, int& kper) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    int s = (int)std::min<long>(8, std::max<long>(1, 512 / std::max<long>(tiles, 1)));
    s = std::min(s, std::max(1, K / (2 * GK)));
    kper = ((K + s - 1) / s + GK - 1) / GK * GK;
    slices = (K + kper - 1) / kper;
    if (slices <= 1) { slices = 1; kper = K; }
}
static size_t gemm_split_floats(int M, int N, int K) {
    int s, kper;
    gemm_split(M, N, K, s, kper);
    return s > 1 ? (size_t)s * ((M + 63) / 64) * ((N + 63) / 64) * 4096 : 0;
}

// part: split-K scratch (gemm_split_floats floats) or NULL = one slice
static bool make_job(int M, int N, int K, const float* A, long sam, long sak, const float* B, long sbn, long sbk, float* C, long ldc,
                     Epi ep, float* part, GemmJob& J) {
    const long bytes_a = ((long)(M - 1) * sam + (long)(K - 1) * sak + 1) * 4, bytes_b = ((long)(N - 1) * sbn + (long)(K - 1) * sbk + 1) * 4;
    if (bytes_a >= (1L << 30) || bytes_b >= (1L << 30)) {
        set_error("gemm operand of %ld / %ld bytes exceeds the 1 GiB the buffer addressing of this kernel covers", bytes_a, bytes_b);
        return false;
    }
    int slices = 1, kper = K;
    if (part != nullptr) gemm_split(M, N, K, slices, kper);
    J = GemmJob{M, N, K, kper, (N + 63) / 64, (M + 63) / 64, slices, sam, sak, sbn, sbk, ldc, bytes_a, bytes_b, A, B, C, part, ep};
    return true;
}
static ReduceJob reduce_of(const GemmJob& J) {
    if (J.slices <= 1) return ReduceJob{0, 0, 0, 0, 0, nullptr, nullptr, 0, Epi{nullptr, nullptr, 0}};
    return ReduceJob{J.M, J.N, J.slices, J.tx, J.ty, J.part, J.C, J.ldc, J.ep};
}
static int launch_gemm(int M, int N, int K, const float* A, long sam, long sak, const float* B, long sbn, long sbk, float* C,
                       long ldc, Epi ep, hipStream_t st, float* part = nullptr) {
    GemmJob J;
    if (!make_job(M, N, K, A, sam, sak, B, sbn, sbk, C, ldc, ep, part, J)) return RBR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gemm_kernel, dim3(J.tx, J.ty, J.slices), dim3(256), 0, st, J);
    RBR_CHECK_LAUNCH("gemm launch");
    if (J.slices > 1) {
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3(J.tx, J.ty, 4), dim3(64), 0, st, reduce_of(J));
        RBR_CHECK_LAUNCH("gemm reduce launch");
    }
    return 0;
}

// dyeff = dy * mul * (y > 0 when relu);  y is the layer OUTPUT (post relu, post mul)
__global__ __launch_bounds__(256) void act_bwd_kernel(long n, const float* __restrict__ dy, const float* __restrict__ y,
                                                      const float* __restrict__ mul, int relu, float* __restrict__ out) {
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) {
        float v = dy[k];
        if (mul) v *= mul[k];
        if (relu == 1) {
            // y = relu(z) * mul: z > 0  <=>  y != 0 unless mul == 0 (then v is already 0)
            const bool pos = mul ? (y[k] != 0.f) : (y[k] > 0.f);
            if (!pos) v = 0.f;
        } else if (relu == 2) {
            // y = tanh(z) * mul: tanh(z) = y / mul where mul != 0 (mul == 0: v is already 0)
            const float t = mul ? (mul[k] != 0.f ? y[k] / mul[k] : 0.f) : y[k];
            v *= 1.f - t * t;
        }
        out[k] = v;
    }
}

// db[n] = sum_m g[m, n]   (fixed order).  64 columns per workgroup, the rows split over its four waves (8 loads in
// flight per thread), partial sums met in LDS in wave order.
__device__ __forceinline__ void colsum_body(int M, int N, const float* __restrict__ g, float* __restrict__ db, const int cb,
                                            float (*s_part)[64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = cb * 64 + lane;
    float s = 0.f;
    if (n < N) {
        int m = wave;
        for (; m + 28 < M; m += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = g[(long)(m + 4 * u) * N + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; m < M; m += 4) s += g[(long)m * N + n];
    }
    s_part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && n < N) db[n] = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
}

__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const float* __restrict__ g, float* __restrict__ db) {
    __shared__ float s_part[4][64];
    colsum_body(M, N, g, db, blockIdx.x, s_part);
}

// What follows a layer's two backward products (gemm2_kernel) in ONE launch: the reductions of the sliced ones -- a workgroup per
// tile, wave = quadrant: the waves of gemm_reduce_kernel's grid -- and the bias gradient's column sums (cs_n > 0: blocks of 64
// columns of g [cs_m, cs_n]).  Flat grid: R0's tiles, R1's tiles, column blocks.
__global__ __launch_bounds__(256) void gemm_reduce2_kernel(const ReduceJob R0, const ReduceJob R1, int cs_m, int cs_n,
                                                           const float* __restrict__ g, float* __restrict__ db) {
    __shared__ float s_part[4][64];
    const int n0 = R0.tx * R0.ty, n1 = R1.tx * R1.ty;
    int b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (b < n0) { gemm_reduce_body(R0, b % R0.tx, b / R0.tx, wave, lane); return; }
    b -= n0;
    if (b < n1) { gemm_reduce_body(R1, b % R1.tx, b / R1.tx, wave, lane); return; }
    colsum_body(cs_m, cs_n, g, db, b - n1, s_part);
}

// ------------------------------------------------------------------------------------ embedding
__global__ __launch_bounds__(256) void emb_fwd_kernel(long n_tok, int D, const long long* __restrict__ ids,
                                                      const float* __restrict__ table, float* __restrict__ out) {
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tok) return;
    const float* row = table + ids[tok] * (long)D;
    for (int d = threadIdx.x & 63; d < D; d += 64) out[tok * D + d] = row[d];
}

__global__ __launch_bounds__(256) void emb_bwd_kernel(long n_tok, int D, const long long* __restrict__ ids,
                                                      const float* __restrict__ d_out, int pad_idx, float* __restrict__ dtable) {
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tok) return;
    const long id = ids[tok];
    if (id == pad_idx) return;
    for (int d = threadIdx.x & 63; d < D; d += 64) atomicAdd(dtable + id * D + d, d_out[tok * D + d]);
}

// ------------------------------------------------------------------------------------ HierPooling
// pooled[doc, d] = max_{l in [0, L-k]} (sum_{j<k} x[doc, l+j, d]) / k, x = mask * table[ids]; first argmax kept.
// The window sum is re-formed in the reference's order (j ascending) at every l, so values match avg_pool1d.
__global__ __launch_bounds__(256) void hier_fwd_kernel(int n_docs, int L, int D, int k, const long long* __restrict__ ids,
                                                       const unsigned char* __restrict__ mask, const float* __restrict__ table,
                                                       int relu, float* __restrict__ pooled, int* __restrict__ argmax) {
    extern __shared__ long s_row[];   // [L] table row offset or -1
    const int doc = blockIdx.x, tid = threadIdx.x;
    for (int l = tid; l < L; l += 256) {
        const long tok = (long)doc * L + l;
        s_row[l] = (mask == nullptr || mask[tok]) ? ids[tok] * (long)D : -1;
    }
    __syncthreads();
    const float inv = 1.f;  // division below, as ATen does (sum / k)
    (void)inv;
    for (int d = tid; d < D; d += 256) {
        float best = -__builtin_huge_valf();
        int bidx = 0;
        for (int l = 0; l + k <= L; ++l) {
            float s = 0.f;
            for (int j = 0; j < k; ++j) {
                const long row = s_row[l + j];
                s += (row >= 0) ? table[row + d] : 0.f;
            }
            const float v = s / (float)k;
            if (v > best) { best = v; bidx = l; }
        }
        pooled[(long)doc * D + d] = relu ? fmaxf(best, 0.f) : best;
        argmax[(long)doc * D + d] = bidx;
    }
}

__global__ __launch_bounds__(256) void hier_bwd_kernel(int n_docs, int L, int D, int k, const long long* __restrict__ ids,
                                                       const unsigned char* __restrict__ mask, const int* __restrict__ argmax,
                                                       const float* __restrict__ pooled, const float* __restrict__ d_pooled,
                                                       int relu, int pad_idx, float* __restrict__ dtable) {
    const int doc = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) {
        float g = d_pooled[(long)doc * D + d];
        if (relu && !(pooled[(long)doc * D + d] > 0.f)) g = 0.f;
        if (g == 0.f) continue;
        g /= (float)k;
        const int l0 = argmax[(long)doc * D + d];
        for (int j = 0; j < k; ++j) {
            const long tok = (long)doc * L + l0 + j;
            if (mask != nullptr && !mask[tok]) continue;
            const long id = ids[tok];
            if (id == pad_idx) continue;
            atomicAdd(dtable + id * D + d, g);
        }
    }
}

}  // namespace rbr

using namespace rbr;

static int linear_fwd(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* b, int32_t relu,
                      const float* drop, float* y, float* ws, void* stream) {
    if (N <= 0 || IN <= 0 || OUT <= 0) { set_error("bad linear shape"); return RBR_ERR_BAD_ARG; }
    if (!x || !W || !y) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    Epi ep{b, drop, relu};
    return launch_gemm(N, OUT, IN, x, IN, 1, W, IN, 1, y, OUT, ep, (hipStream_t)stream, ws);
}
extern "C" int rbr_linear_fwd(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* b,
                              int32_t relu, const float* drop, float* y, void* stream) {
    return linear_fwd(N, IN, OUT, x, W, b, relu, drop, y, nullptr, stream);
}
extern "C" size_t rbr_linear_fwd_ws_floats(int32_t N, int32_t IN, int32_t OUT) {
    return (N > 0 && IN > 0 && OUT > 0) ? gemm_split_floats(N, OUT, IN) : 0;
}
extern "C" int rbr_linear_fwd_ex(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* b,
                                 int32_t relu, const float* drop, float* y, float* ws, void* stream) {
    return linear_fwd(N, IN, OUT, x, W, b, relu, drop, y, ws, stream);
}

extern "C" size_t rbr_linear_bwd_ws_floats(int32_t N, int32_t OUT) { return (N > 0 && OUT > 0) ? (size_t)N * OUT : 0; }
extern "C" size_t rbr_linear_bwd_ex_ws_floats(int32_t N, int32_t IN, int32_t OUT) {
    if (N <= 0 || IN <= 0 || OUT <= 0) return 0;
    return (size_t)N * OUT + gemm_split_floats(OUT, IN, N) + gemm_split_floats(N, IN, OUT);      // g | dW's partials | d_x's
}

static int linear_bwd(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* y, const float* d_y,
                      int32_t relu, const float* drop, float* d_x, float* dW, float* db, float* ws, bool split, void* stream);
extern "C" int rbr_linear_bwd(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* y,
                              const float* d_y, int32_t relu, const float* drop, float* d_x, float* dW, float* db, float* ws,
                              void* stream) {
    return linear_bwd(N, IN, OUT, x, W, y, d_y, relu, drop, d_x, dW, db, ws, false, stream);
}
extern "C" int rbr_linear_bwd_ex(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* y,
                                 const float* d_y, int32_t relu, const float* drop, float* d_x, float* dW, float* db, float* ws,
                                 void* stream) {
    return linear_bwd(N, IN, OUT, x, W, y, d_y, relu, drop, d_x, dW, db, ws, true, stream);
}

// split: ws holds rbr_linear_bwd_ex_ws_floats (g, then the split-K partial tiles); else rbr_linear_bwd_ws_floats
static int linear_bwd(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* y, const float* d_y,
                      int32_t relu, const float* drop, float* d_x, float* dW, float* db, float* ws, bool split, void* stream) {
    if (N <= 0 || IN <= 0 || OUT <= 0) { set_error("bad linear shape"); return RBR_ERR_BAD_ARG; }
    if (!x || !W || !y || !d_y || !dW || !ws) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    float* part = split ? ws + (size_t)N * OUT : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const float* g = d_y;
    if (relu || drop) {
        const long n = (long)N * OUT;
        hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, st, n, d_y, y,
                           drop, relu, ws);
        RBR_CHECK_LAUNCH("linear act_bwd launch");
        g = ws;
    }
    Epi none{nullptr, nullptr, 0};
    if (split) {
        // the weight gradient and the input gradient start from the same g and share nothing else: ONE launch for both products
        // (gemm2_kernel), ONE for what follows them -- their split-K reductions and the bias gradient's column sums -- instead of
        // up to five (D-ATT's shared fc: 11 launches of 5-10 us in the step's backward, now 5)
        GemmJob Jw, Jx;
        float* part_w = part;
        float* part_x = part + gemm_split_floats(OUT, IN, N);
        // dW[OUT, IN] = g^T[OUT, N] . x[N, IN]:  A(m=o, k=n) = g[n*OUT + o], B(n=i, k=n) = x[n*IN + i]
        if (!make_job(OUT, IN, N, g, 1, OUT, x, 1, IN, dW, IN, none, part_w, Jw)) return RBR_ERR_UNSUPPORTED;
        if (d_x) {
            // d_x[N, IN] = g[N, OUT] . W[OUT, IN]:  A(m=n, k=o) = g[n*OUT + o], B(n=i, k=o) = W[o*IN + i]
            if (!make_job(N, IN, OUT, g, OUT, 1, W, 1, IN, d_x, IN, none, part_x, Jx)) return RBR_ERR_UNSUPPORTED;
            hipLaunchKernelGGL(gemm2_kernel, dim3((unsigned)(Jw.tx * Jw.ty * Jw.slices + Jx.tx * Jx.ty * Jx.slices)), dim3(256), 0, st, Jw, Jx);
            RBR_CHECK_LAUNCH("linear backward products launch");
        } else {
            Jx = GemmJob{};
            Jx.slices = 1;
            hipLaunchKernelGGL(gemm_kernel, dim3(Jw.tx, Jw.ty, Jw.slices), dim3(256), 0, st, Jw);
            RBR_CHECK_LAUNCH("gemm launch");
        }
        const ReduceJob Rw = reduce_of(Jw), Rx = reduce_of(Jx);
        const int nb = Rw.tx * Rw.ty + Rx.tx * Rx.ty + (db ? (OUT + 63) / 64 : 0);
        if (nb > 0) {
            hipLaunchKernelGGL(gemm_reduce2_kernel, dim3((unsigned)nb), dim3(256), 0, st, Rw, Rx, db ? N : 0, db ? OUT : 0, g, db);
            RBR_CHECK_LAUNCH("linear backward reductions launch");
        }
        return 0;
    }
    // dW[OUT, IN] = g^T[OUT, N] . x[N, IN]:  A(m=o, k=n) = g[n*OUT + o], B(n=i, k=n) = x[n*IN + i]
    if (int e = launch_gemm(OUT, IN, N, g, 1, OUT, x, 1, IN, dW, IN, none, st, nullptr)) return e;
    if (db) {
        hipLaunchKernelGGL(colsum_kernel, dim3((OUT + 63) / 64), dim3(256), 0, st, N, OUT, g, db);
        RBR_CHECK_LAUNCH("linear colsum launch");
    }
    // d_x[N, IN] = g[N, OUT] . W[OUT, IN]:  A(m=n, k=o) = g[n*OUT + o], B(n=i, k=o) = W[o*IN + i]
    if (d_x) return launch_gemm(N, IN, OUT, g, OUT, 1, W, 1, IN, d_x, IN, none, st, nullptr);
    return 0;
}

extern "C" int rbr_embedding_fwd(int64_t n_tok, int32_t D, const int64_t* ids, const float* table, float* out, void* stream) {
    if (n_tok <= 0 || D <= 0 || !ids || !table || !out) { set_error("bad embedding arguments"); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(emb_fwd_kernel, dim3((unsigned)((n_tok + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (long)n_tok, D,
                       reinterpret_cast<const long long*>(ids), table, out);
    RBR_CHECK_LAUNCH("embedding fwd launch");
    return 0;
}

extern "C" int rbr_embedding_bwd(int64_t n_tok, int32_t D, const int64_t* ids, const float* d_out, int32_t pad_idx,
                                 float* dtable, void* stream) {
    if (n_tok <= 0 || D <= 0 || !ids || !d_out || !dtable) { set_error("bad embedding arguments"); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(emb_bwd_kernel, dim3((unsigned)((n_tok + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (long)n_tok, D,
                       reinterpret_cast<const long long*>(ids), d_out, pad_idx, dtable);
    RBR_CHECK_LAUNCH("embedding bwd launch");
    return 0;
}

extern "C" int rbr_hier_pool_fwd(int32_t n_docs, int32_t L, int32_t D, int32_t k, const int64_t* ids, const uint8_t* mask,
                                 const float* table, int32_t relu, float* pooled, int32_t* argmax, void* stream) {
    if (n_docs <= 0 || L <= 0 || D <= 0 || k <= 0 || k > L) { set_error("bad hier_pool shape"); return RBR_ERR_BAD_ARG; }
    if (!ids || !table || !pooled || !argmax) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    if ((size_t)L * sizeof(long) > 60 * 1024) { set_error("doc_len %d too large", L); return RBR_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(hier_fwd_kernel, dim3(n_docs), dim3(256), (size_t)L * sizeof(long), (hipStream_t)stream, n_docs, L, D, k,
                       reinterpret_cast<const long long*>(ids), mask, table, relu, pooled, argmax);
    RBR_CHECK_LAUNCH("hier_pool fwd launch");
    return 0;
}

extern "C" int rbr_hier_pool_bwd(int32_t n_docs, int32_t L, int32_t D, int32_t k, const int64_t* ids, const uint8_t* mask,
                                 const int32_t* argmax, const float* pooled, const float* d_pooled, int32_t relu,
                                 int32_t pad_idx, float* dtable, void* stream) {
    if (n_docs <= 0 || L <= 0 || D <= 0 || k <= 0) { set_error("bad hier_pool shape"); return RBR_ERR_BAD_ARG; }
    if (!ids || !argmax || !pooled || !d_pooled || !dtable) { set_error("null pointer"); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(hier_bwd_kernel, dim3(n_docs), dim3(256), 0, (hipStream_t)stream, n_docs, L, D, k,
                       reinterpret_cast<const long long*>(ids), mask, argmax, pooled, d_pooled, relu, pad_idx, dtable);
    RBR_CHECK_LAUNCH("hier_pool bwd launch");
    return 0;
}

namespace rbr {

// ---------------------------------------------------------------------------------- id range check
// nn.Embedding raises IndexError for an id outside its table (reference models/deepconn/layers.py:23); a kernel cannot, and
// an unchecked id would read -- or, in the backward, WRITE -- outside the table.  The modules therefore pass every id tensor
// through this launch first: out = in where 0 <= in < limit, else `replace` (a valid row) plus a record in `err`
// (err[0] bad ids so far, err[1] one offending value, err[2] its set), which the host turns into the IndexError at its
// next synchronisation point (functional.check_id_errors).  Up to 8 tensors per launch; the outputs may be adjacent slices
// of one buffer, which also stacks the two towers' token ids for free.
__global__ __launch_bounds__(256) void sanitize_ids_kernel(const IdSets S, long long* __restrict__ err) {
    const long long total = S.first[S.count];
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long long)gridDim.x * 256) sanitize_id(S, k, err);
}

int fill_id_sets(int n_sets, const rbr_id_set* sets, IdSets& S) {
    if (n_sets <= 0 || n_sets > RBR_MAX_ID_SETS || !sets) { set_error("rbr_sanitize_ids: n_sets=%d", n_sets); return RBR_ERR_BAD_ARG; }
    memset(&S, 0, sizeof(S));
    S.count = n_sets;
    long long total = 0;
    for (int s = 0; s < n_sets; ++s) {
        if (sets[s].n < 0 || (sets[s].n > 0 && (!sets[s].in || !sets[s].out)) || sets[s].limit <= 0 ||
            sets[s].replace < 0 || sets[s].replace >= sets[s].limit) {
            set_error("rbr_sanitize_ids: set %d is malformed", s);
            return RBR_ERR_BAD_ARG;
        }
        S.in[s] = reinterpret_cast<const long long*>(sets[s].in); S.out[s] = reinterpret_cast<long long*>(sets[s].out);
        S.n[s] = sets[s].n; S.limit[s] = sets[s].limit; S.replace[s] = sets[s].replace;
        S.first[s] = total;
        total += sets[s].n;
    }
    S.first[n_sets] = total;
    return 0;
}

}  // namespace rbr

extern "C" int rbr_sanitize_ids(int32_t n_sets, const rbr_id_set* sets, int64_t* err, void* stream) {
    using namespace rbr;
    if (!err) { set_error("rbr_sanitize_ids: null err"); return RBR_ERR_BAD_ARG; }
    IdSets S;
    if (int e = fill_id_sets(n_sets, sets, S)) return e;
    const long long total = S.first[n_sets];
    if (total == 0) return 0;
    hipLaunchKernelGGL(sanitize_ids_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, S, reinterpret_cast<long long*>(err));
    RBR_CHECK_LAUNCH("sanitize_ids launch");
    return 0;
}

namespace rbr {

// ---------------------------------------------------------------------------------- in-batch document dedup
// A document is a function of its user / item id (the doc split builds one per id), and the reference re-encodes it for
// every pair it appears in (models/deepconn/deepconn.py:46-47).  first[b] = the first row of the batch with the same id;
// rows that are not their own first occurrence get an all-zero mask, so the encoder skips them (their 32-token slabs are
// inactive) and the caller gathers their features from row first[b].  Shape-static and sync-free: graph-capturable.
// scratch[id] holds n - (first row) (0 = id absent; zeroed by the launch in front), rows [0, B) are the user side, rows
// [B, 2B) the item side with its own scratch range.
__global__ __launch_bounds__(256) void dedup_mark_kernel(int B, const long long* __restrict__ u_ids, const long long* __restrict__ i_ids,
                                                         int U, int I, int* __restrict__ scratch) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= 2 * B) return;
    const bool item = r >= B;
    const int b = item ? r - B : r;
    const long long id = item ? i_ids[b] : u_ids[b];
    if ((unsigned long long)id >= (unsigned long long)(item ? I : U)) return;         // out of range: its own first occurrence
    atomicMax(scratch + (item ? U : 0) + id, B - b);
}

__global__ __launch_bounds__(256) void dedup_apply_kernel(int B, int L, const long long* __restrict__ u_ids,
                                                          const long long* __restrict__ i_ids, int U, int I,
                                                          const int* __restrict__ scratch, const unsigned char* __restrict__ mask_in,
                                                          long long* __restrict__ first, unsigned char* __restrict__ mask_out) {
    const int r = blockIdx.x;                      // one workgroup per row
    const bool item = r >= B;
    const int b = item ? r - B : r;
    const long long id = item ? i_ids[b] : u_ids[b];
    int f = b;
    if ((unsigned long long)id < (unsigned long long)(item ? I : U)) f = B - scratch[(item ? U : 0) + id];
    if (threadIdx.x == 0) first[r] = (item ? B : 0) + f;
    const bool keep = f == b;
    for (int l = threadIdx.x; l < L; l += 256)
        mask_out[(long)r * L + l] = keep ? (mask_in != nullptr ? mask_in[(long)r * L + l] : (unsigned char)1) : (unsigned char)0;
}

}  // namespace rbr

// The backward's side of the dedup: the gradient rows of the repeated documents are added onto their first occurrence's row (which
// is the one the encoder's backward will see: the repeated rows' masks are blank) and cleared.  One wave per row.
namespace rbr {
__global__ __launch_bounds__(256) void dedup_fold_kernel(int n_rows, int H, const long long* __restrict__ first, float* __restrict__ d_rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_rows) return;
    const int f = (int)first[r];
    if (f == r) return;
    for (int h = lane; h < H; h += 64) {
        atomicAdd(d_rows + (long)f * H + h, d_rows[(long)r * H + h]);
        d_rows[(long)r * H + h] = 0.f;
    }
}
}  // namespace rbr
extern "C" int rbr_dedup_fold_rows(int32_t n_rows, int32_t H, const int64_t* first, float* d_rows, void* stream) {
    if (n_rows <= 0 || H <= 0 || !first || !d_rows) { rbr::set_error("rbr_dedup_fold_rows: bad argument"); return RBR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(rbr::dedup_fold_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, n_rows, H,
                       reinterpret_cast<const long long*>(first), d_rows);
    RBR_CHECK_LAUNCH("dedup fold launch");
    return 0;
}

extern "C" size_t rbr_dedup_ws_bytes(int32_t U, int32_t I) { return (U > 0 && I > 0) ? ((size_t)U + (size_t)I) * sizeof(int) : 0; }

extern "C" int rbr_dedup_rows(int32_t B, int32_t L, const int64_t* u_ids, const int64_t* i_ids, int32_t U, int32_t I,
                              const uint8_t* mask_in, void* ws, int64_t* first, uint8_t* mask_out, void* stream) {
    using namespace rbr;
    if (B <= 0 || L <= 0 || U <= 0 || I <= 0 || !u_ids || !i_ids || !ws || !first || !mask_out) { set_error("rbr_dedup_rows: bad argument"); return RBR_ERR_BAD_ARG; }
    hipStream_t st = (hipStream_t)stream;
    int* scratch = static_cast<int*>(ws);
    if (int e = zero_words(scratch, ((size_t)U + (size_t)I) * sizeof(int), st)) return e;
    hipLaunchKernelGGL(dedup_mark_kernel, dim3((2 * B + 255) / 256), dim3(256), 0, st, B, reinterpret_cast<const long long*>(u_ids),
                       reinterpret_cast<const long long*>(i_ids), U, I, scratch);
    RBR_CHECK_LAUNCH("dedup mark launch");
    hipLaunchKernelGGL(dedup_apply_kernel, dim3(2 * B), dim3(256), 0, st, B, L, reinterpret_cast<const long long*>(u_ids),
                       reinterpret_cast<const long long*>(i_ids), U, I, scratch, mask_in, reinterpret_cast<long long*>(first), mask_out);
    RBR_CHECK_LAUNCH("dedup apply launch");
    return 0;
}

// ------------------------------------------------------------------------------------ MyConv1d.forward's epilogue
// The layer-level conv (deepconn/layers.py:46-60: per width Conv1d(padding = (kz-1)/2), cat on the channel dim) as a product +
// shifted adds: T[(b, n), (w, j, c)] = <x[b, :, n], W_w[c, :, j]> comes from the GEMM, and
//     out[b, ch_off[w] + c, l] = bias_w[c] + sum_j T[(b, l + j - pad_w), (w, j, c)]      (rows outside the document are zero)
// in the reference's N x C x L layout -- one launch instead of a pad, kz adds, a cat and a transpose per width.  Backward:
// dT[(b, n), (w, j, c)] = d_out[b, ch_off[w] + c, n - j + pad_w] (0 outside), dbias_w[c] = sum_{b, l} d_out (fixed order).
namespace rbr {
struct ShiftAdd {
    int bz, L, n_widths, C, cp;
    int kz[RBR_MAX_WIDTHS], ch[RBR_MAX_WIDTHS], ch_off[RBR_MAX_WIDTHS], poff[RBR_MAX_WIDTHS];
};
__device__ __forceinline__ int shift_bank(const ShiftAdd& S, int c) {
    int w = 0;
#pragma unroll
    for (int k = 1; k < RBR_MAX_WIDTHS; ++k)
        if (k < S.n_widths && c >= S.ch_off[k]) w = k;
    return w;
}
__global__ __launch_bounds__(256) void conv_shift_add_fwd_kernel(const ShiftAdd S, const float* __restrict__ T, const PtrArray bias,
                                                                 float* __restrict__ out) {
    const long n = (long)S.bz * S.C * S.L;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int l = (int)(e % S.L);
        const int c = (int)((e / S.L) % S.C), b = (int)(e / ((long)S.L * S.C));
        const int w = shift_bank(S, c), cl = c - S.ch_off[w], kz = S.kz[w], pad = (kz - 1) / 2;
        float acc = bias.p[w][cl];
        for (int j = 0; j < kz; ++j) {
            const int row = l + j - pad;
            if (row >= 0 && row < S.L) acc += T[((long)b * S.L + row) * S.cp + S.poff[w] + j * S.ch[w] + cl];
        }
        out[e] = acc;
    }
}
__global__ __launch_bounds__(256) void conv_shift_add_bwd_kernel(const ShiftAdd S, const float* __restrict__ d_out, float* __restrict__ dT) {
    const long n = (long)S.bz * S.L * S.cp;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int pc = (int)(e % S.cp);
        const int row = (int)((e / S.cp) % S.L), b = (int)(e / ((long)S.cp * S.L));
        int w = 0;
#pragma unroll
        for (int k = 1; k < RBR_MAX_WIDTHS; ++k)
            if (k < S.n_widths && pc >= S.poff[k]) w = k;
        const int rel = pc - S.poff[w], j = rel / S.ch[w], cl = rel - j * S.ch[w];
        const int l = row - j + (S.kz[w] - 1) / 2;
        dT[e] = (l >= 0 && l < S.L) ? d_out[((long)b * S.C + S.ch_off[w] + cl) * S.L + l] : 0.f;
    }
}
// dbias[c] = sum over (b, l) of d_out[b, c, l]: one workgroup per channel, fixed order
__global__ __launch_bounds__(256) void conv_shift_add_dbias_kernel(const ShiftAdd S, const float* __restrict__ d_out, const MutPtrArray dbias) {
    __shared__ float s_red[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    for (long k = threadIdx.x; k < (long)S.bz * S.L; k += 256) {
        const int b = (int)(k / S.L), l = (int)(k - (long)b * S.L);
        acc += d_out[((long)b * S.C + c) * S.L + l];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int w = shift_bank(S, c);
        dbias.p[w][c - S.ch_off[w]] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}
static bool shift_add_job(int32_t bz, int32_t L, int32_t n_widths, const int32_t* kz, const int32_t* ch, ShiftAdd& S) {
    if (bz <= 0 || L <= 0 || n_widths <= 0 || n_widths > RBR_MAX_WIDTHS || !kz || !ch || (long)bz * L >= (1L << 31)) return false;
    memset(&S, 0, sizeof(S));
    S.bz = bz; S.L = L; S.n_widths = n_widths;
    for (int w = 0; w < n_widths; ++w) {
        if (kz[w] <= 0 || kz[w] % 2 == 0 || kz[w] > kMaxKF || ch[w] <= 0 || ch[w] > kMaxChannels) return false;
        S.kz[w] = kz[w]; S.ch[w] = ch[w]; S.ch_off[w] = S.C; S.poff[w] = S.cp;
        S.C += ch[w]; S.cp += kz[w] * ch[w];
    }
    return S.C <= kMaxChannels;
}
}  // namespace rbr

extern "C" int rbr_conv_shift_add_fwd(int32_t bz, int32_t L, int32_t n_widths, const int32_t* kz, const int32_t* ch, const float* T,
                                      const float* const* bias, float* out, void* stream) {
    rbr::ShiftAdd S;
    if (!rbr::shift_add_job(bz, L, n_widths, kz, ch, S) || !T || !bias || !out) { rbr::set_error("rbr_conv_shift_add_fwd: bad argument"); return RBR_ERR_BAD_ARG; }
    rbr::PtrArray bp{};
    for (int w = 0; w < n_widths; ++w) { if (!bias[w]) { rbr::set_error("null bias"); return RBR_ERR_BAD_ARG; } bp.p[w] = bias[w]; }
    const long n = (long)bz * S.C * L;
    hipLaunchKernelGGL(rbr::conv_shift_add_fwd_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, S, T, bp, out);
    RBR_CHECK_LAUNCH("conv shift-add fwd launch");
    return 0;
}
extern "C" int rbr_conv_shift_add_bwd(int32_t bz, int32_t L, int32_t n_widths, const int32_t* kz, const int32_t* ch, const float* d_out,
                                      float* dT, float* const* dbias, void* stream) {
    rbr::ShiftAdd S;
    if (!rbr::shift_add_job(bz, L, n_widths, kz, ch, S) || !d_out || !dT || !dbias) { rbr::set_error("rbr_conv_shift_add_bwd: bad argument"); return RBR_ERR_BAD_ARG; }
    rbr::MutPtrArray dbp{};
    for (int w = 0; w < n_widths; ++w) { if (!dbias[w]) { rbr::set_error("null dbias"); return RBR_ERR_BAD_ARG; } dbp.p[w] = dbias[w]; }
    const long n = (long)bz * L * S.cp;
    hipLaunchKernelGGL(rbr::conv_shift_add_bwd_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, S, d_out, dT);
    RBR_CHECK_LAUNCH("conv shift-add bwd launch");
    hipLaunchKernelGGL(rbr::conv_shift_add_dbias_kernel, dim3(S.C), dim3(256), 0, (hipStream_t)stream, S, d_out, dbp);
    RBR_CHECK_LAUNCH("conv shift-add dbias launch");
    return 0;
}

// ------------------------------------------------------------------------------------ dropout multiplier
// out[i] = 0 with probability p, else 1/(1-p): the multiplier nn.Dropout / F.dropout applies (deepconn/layers.py:202,
// narre.py:73, dual_att.py:33).  Philox4x32-10 keyed by `seed`, counter = (element quad, call number); the call number
// lives in device memory (state[0]) and is advanced by the last workgroup to finish (ticket in state[1]), so the launch
// can be recorded into a hipGraph and still draws a fresh mask on every replay -- without the host-side generator
// bookkeeping (two fill launches per replay) a torch RNG op costs inside a captured step.
namespace rbr {

__global__ __launch_bounds__(256) void dropout_mult_kernel(long n, float p, unsigned long long seed,
                                                           unsigned long long* __restrict__ state, float* __restrict__ out) {
    const unsigned long long call = state[0];
    const float keep = 1.f / (1.f - p);
    const long nquads = (n + 3) >> 2;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (long)gridDim.x * 256) {
        unsigned c0, c1, c2, c3;
        philox4x32_10((unsigned long long)q, call, seed, c0, c1, c2, c3);
        const unsigned w[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long i = 4 * q + e;
            if (i < n) out[i] = dropout_keep(w[e], p) ? keep : 0.f;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {       // (no fence needed: the call number was consumed before this atomic is issued)
        if (atomicAdd(state + 1, 1ull) == (unsigned long long)gridDim.x - 1) {      // every workgroup has read state[0]
            state[1] = 0;
            state[0] = call + 1;
        }
    }
}

}  // namespace rbr

extern "C" int rbr_dropout_multiplier(int64_t n, float p, uint64_t seed, uint64_t* state, float* out, void* stream) {
    if (n <= 0 || !state || !out) { rbr::set_error("dropout_multiplier: n=%lld or null pointer", (long long)n); return RBR_ERR_BAD_ARG; }
    if (!(p >= 0.f && p < 1.f)) { rbr::set_error("dropout_multiplier: p=%f outside [0,1)", (double)p); return RBR_ERR_BAD_ARG; }
    const long nquads = (n + 3) / 4;
    const int blocks = (int)std::min<long>((nquads + 255) / 256, 1024);
    hipLaunchKernelGGL(rbr::dropout_mult_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long)n, p,
                       (unsigned long long)seed, reinterpret_cast<unsigned long long*>(state), out);
    RBR_CHECK_LAUNCH("dropout_multiplier launch");
    return 0;
}

// ---- 2 x 2 block concatenation: out [2B, C1 + C2] = [[a, b], [c, d]] with a, c [B, C1] and b, d [B, C2] (D-ATT: local | global
// features of the user tower over those of the item tower, the input of the shared fc) -- one launch instead of three
// torch.cat, and one for the backward's four pieces instead of four strided copies.
namespace rbr {
__global__ __launch_bounds__(256) void block_cat_kernel(int B, int C1, int C2, const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ c, const float* __restrict__ d,
                                                        float* __restrict__ out) {
    const int C = C1 + C2;
    const long n = (long)2 * B * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int r = (int)(e / C), col = (int)(e - (long)r * C);
        const int rr = r < B ? r : r - B;
        const float* src = col < C1 ? (r < B ? a : c) + (long)rr * C1 + col : (r < B ? b : d) + (long)rr * C2 + (col - C1);
        out[e] = *src;
    }
}
__global__ __launch_bounds__(256) void block_split_kernel(int B, int C1, int C2, const float* __restrict__ g, float* __restrict__ a,
                                                          float* __restrict__ b, float* __restrict__ c, float* __restrict__ d) {
    const int C = C1 + C2;
    const long n = (long)2 * B * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int r = (int)(e / C), col = (int)(e - (long)r * C);
        const int rr = r < B ? r : r - B;
        float* dst = col < C1 ? (r < B ? a : c) + (long)rr * C1 + col : (r < B ? b : d) + (long)rr * C2 + (col - C1);
        *dst = g[e];
    }
}
}  // namespace rbr

extern "C" int rbr_block_cat(int32_t B, int32_t C1, int32_t C2, const float* a, const float* b, const float* c, const float* d,
                             float* out, void* stream) {
    if (B <= 0 || C1 <= 0 || C2 <= 0 || !a || !b || !c || !d || !out) { set_error("bad block_cat arguments"); return RBR_ERR_BAD_ARG; }
    const long n = (long)2 * B * (C1 + C2);
    hipLaunchKernelGGL(rbr::block_cat_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, B,
                       C1, C2, a, b, c, d, out);
    RBR_CHECK_LAUNCH("block_cat launch");
    return 0;
}

extern "C" int rbr_block_split(int32_t B, int32_t C1, int32_t C2, const float* g, float* a, float* b, float* c, float* d,
                               void* stream) {
    if (B <= 0 || C1 <= 0 || C2 <= 0 || !g || !a || !b || !c || !d) { set_error("bad block_split arguments"); return RBR_ERR_BAD_ARG; }
    const long n = (long)2 * B * (C1 + C2);
    hipLaunchKernelGGL(rbr::block_split_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       B, C1, C2, g, a, b, c, d);
    RBR_CHECK_LAUNCH("block_split launch");
    return 0;
}

