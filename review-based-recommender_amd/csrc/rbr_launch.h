// rbr_launch.h -- one launch path for kernels that may run for TWO independent problems in one launch.
//
// D-ATT (reference models/dual_att/dual_att.py:45-57) runs the same chain of ~40 kernels once per tower, on separate
// parameters and documents of identical shapes.  Most of those kernels are short (5-30 us: a few dependent memory round
// trips), so a tower pair costs two latencies where one would do, and a recorded step carries twice the launches.  A kernel
// written as
//
//     __device__ __forceinline__ void foo_kernel(args...)            (body unchanged; uses blockIdx.x / .y only; struct
//                                                                      arguments as `const T&`: see call_pack)
//
// and launched with  rbr::launch<foo_kernel, 256>(grid, block, lds, stream, "what", args...)  runs as before -- through the
// __global__ wrapper single_k<foo_kernel, ...> -- unless the calling thread is inside a PAIR REGION
// (rbr_pair_begin / rbr_pair_next / rbr_pair_end of rbr_hip.h): then the launch is RECORDED instead (kernel, grid, arguments),
// first for problem 0, after rbr_pair_next for problem 1, and rbr_pair_end zips the two records lists: record i of problem 0 and
// record i of problem 1 -- same kernel, same grid -- leave as ONE launch of pair_k<foo_kernel, ...> with gridDim.z = 2, whose
// workgroups take their argument set by blockIdx.z.  Records that do not match (another kernel, another grid, a grid that uses
// z itself, argument sets too large for one kernarg segment) are launched one after the other, in order; results never depend
// on whether a pair formed.  Arguments larger than 256 bytes (ConvPlan, PackJob: shape-only plans) are passed once and must be
// bytewise equal in both records -- they are for towers of equal shapes -- else the pair does not form.
//
// Inside a pair region every launch must come through rbr::launch (an entry point that still launches directly would run
// out of order): RBR_CHECK_LAUNCH refuses with RBR_ERR_UNSUPPORTED when a region is open.
#pragma once

#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>
#include <vector>

namespace rbr {

void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// ---------------------------------------------------------------------------------------------- argument packs (kernarg PODs)
template <class... A> struct Pack;
template <> struct Pack<> {};
template <class H, class... T> struct Pack<H, T...> {
    H h;
    Pack<T...> t;
};
inline Pack<> make_pack() { return {}; }
template <class H, class... T> inline Pack<H, T...> make_pack(H h, T... t) { return Pack<H, T...>{h, make_pack(t...)}; }

// Arguments passed ONCE per pair (they must be bytewise equal in both records, else the pair does not form): the shape-only
// structs -- plans and dimension blocks, which hold no pointers and are the same for towers of equal shapes.  A struct that is
// indexed dynamically by the body cannot be selected between two kernarg copies without being copied to scratch memory first,
// so every such struct either is shape-only (marked here: RBR_SHARED_KERNEL_ARG) or holds nothing but pointers the body
// indexes with constants.  Anything above 256 bytes counts as shared as well (kernarg space: 4 KB per launch).
template <class T> struct SharedKernelArg : std::bool_constant<(sizeof(T) > 256)> {};
template <class T> constexpr bool kSharedArg = SharedKernelArg<T>::value;
#define RBR_SHARED_KERNEL_ARG(T) \
    template <> struct SharedKernelArg<T> : std::true_type {}
struct NoArg {};
// problem 1's arguments: the shared ones left out
template <class... A> struct Pack1;
template <> struct Pack1<> {};
template <class H, class... T> struct Pack1<H, T...> {
    std::conditional_t<kSharedArg<H>, NoArg, H> h;
    Pack1<T...> t;
};
inline void fill_pack1(Pack1<>&, const Pack<>&) {}
template <class H, class... T> inline void fill_pack1(Pack1<H, T...>& d, const Pack<H, T...>& s) {
    if constexpr (!kSharedArg<H>) d.h = s.h;
    fill_pack1(d.t, s.t);
}
inline bool shared_equal(const Pack<>&, const Pack<>&) { return true; }
template <class H, class... T> inline bool shared_equal(const Pack<H, T...>& a, const Pack<H, T...>& b) {
    if constexpr (kSharedArg<H>)
        if (memcmp(&a.h, &b.h, sizeof(H)) != 0) return false;
    return shared_equal(a.t, b.t);
}

// The members are handed to the body BY REFERENCE into the kernarg segment: a body that takes its struct arguments as
// `const T&` indexes them where they lie (scalar loads), exactly as a __global__ kernel indexes its by-value parameters; a
// by-value struct parameter would be copied to scratch memory first whenever the body indexes it dynamically.
template <class Tag, class... Done> __device__ __forceinline__ void call_pack(const Pack<>&, const Done&... d) { Tag::call(d...); }
template <class Tag, class H, class... T, class... Done>
__device__ __forceinline__ void call_pack(const Pack<H, T...>& p, const Done&... d) {
    call_pack<Tag>(p.t, d..., p.h);
}
// problem 1 / problem 0 by a (uniform) flag: ONE copy of the body, every non-shared argument selected between the two
// kernarg packs (scalars and pointers, by value).  One body keeps the register allocation of the single launch: with a body
// per argument set the allocator needed 2-4 more VGPRs, enough to cost gather_pool (64 -> 66) a wave per SIMD.
template <class Tag, class... Done> __device__ __forceinline__ void call_sel(bool, const Pack<>&, const Pack1<>&, const Done&... d) { Tag::call(d...); }
template <class Tag, class H, class... T, class... Done>
__device__ __forceinline__ void call_sel(bool second, const Pack<H, T...>& p0, const Pack1<H, T...>& p1, const Done&... d) {
    if constexpr (kSharedArg<H>) {
        call_sel<Tag>(second, p0.t, p1.t, d..., p0.h);
    } else {
        static_assert(!std::is_class_v<H>, "struct arguments that differ between the problems take the branch form");
        const H h = second ? p1.h : p0.h;
        call_sel<Tag>(second, p0.t, p1.t, d..., h);
    }
}

// The kernel body travels into the __global__ wrappers as a TYPE (KTag<decltype(&body), &body>), not as a non-type template
// argument of the wrapper itself: a function template with a dependent-typed non-type parameter is mangled with the newer
// "Tn" production, which the demanglers of rocprofv3 and binutils do not know -- the profiles showed raw _ZN3rbr6pair_kI... names.
// As a class template argument the same pointer mangles the classic way:
//     rbr::pair_k<rbr::KTag<void (*)(rbr::ConvPlan const&, ...), &rbr::gather_pool_kernel<false>>, 256, 1, ...>
template <class F, F Fn> struct KTag {
    template <class... A> static __device__ __forceinline__ void call(const A&... a) { Fn(a...); }
};
template <class Tag, int BOUNDS, int MINB, class... A> __global__ __launch_bounds__(BOUNDS, MINB) void single_k(const Pack<A...> p) {
    call_pack<Tag>(p);
}
// A struct argument that differs between the problems (PtrArray: per-tower weight pointers, indexed dynamically) cannot be
// selected by address -- the compiler would copy the kernarg packs to scratch memory -- so kernels that take one get a copy of
// the body per argument set behind a branch on the (uniform) blockIdx.z instead; they are short, un-pressured kernels
// (pool_finalize, dw_reduce, compact_pack, the GEMM whose occupancy LDS decides).
template <class Tag, class... Done> __device__ __forceinline__ void call_pack1(const Pack<>&, const Pack1<>&, const Done&... d) { Tag::call(d...); }
template <class Tag, class H, class... T, class... Done>
__device__ __forceinline__ void call_pack1(const Pack<H, T...>& p0, const Pack1<H, T...>& p1, const Done&... d) {
    if constexpr (kSharedArg<H>)
        call_pack1<Tag>(p0.t, p1.t, d..., p0.h);
    else
        call_pack1<Tag>(p0.t, p1.t, d..., p1.h);
}
template <class... A> constexpr bool kPairByBranch = ((!kSharedArg<A> && std::is_class_v<A>) || ...);
template <class Tag, int BOUNDS, int MINB, class... A>
__global__ __launch_bounds__(BOUNDS, MINB) void pair_k(const Pack<A...> p0, const Pack1<A...> p1) {
    if constexpr (kPairByBranch<A...>) {
        if (blockIdx.z == 0)
            call_pack<Tag>(p0);
        else
            call_pack1<Tag>(p0, p1);
    } else {
        call_sel<Tag>(blockIdx.z != 0, p0, p1);
    }
}

// ---------------------------------------------------------------------------------------------- the pair region (per thread)
constexpr size_t kPairArgBytes = 2304;
constexpr size_t kKernargLimit = 4000;       // bytes of kernel arguments one launch may carry (HIP: 4 KB)
struct PairRec {
    int (*pair)(const PairRec& a, const PairRec& b);     // 0: launched as one; kPairNoMatch: launch them one by one; else an error
    int (*single)(const PairRec& a);
    dim3 grid, block;
    size_t lds;
    hipStream_t st;
    const char* what;
    int solo;                        // recorded inside a PairSolo scope: never shares a launch (see PairSolo)
    alignas(16) unsigned char args[kPairArgBytes];
};
constexpr int kPairNoMatch = -1000;
struct PairState {
    int mode = 0;                    // 0: launches run at once; 1 / 2: recording problem 0 / 1
    int solo = 0;                    // depth of PairSolo scopes
    std::vector<PairRec> rec[2];
    int paired = 0, singles = 0;     // what the last rbr_pair_end launched
};
PairState& pair_state();             // thread-local (rbr_plan.hip)
inline bool pair_region_open() { return pair_state().mode != 0; }
// Launches recorded while a PairSolo lives do not pair: a run of consecutive solo records leaves problem by problem (all of
// problem 0's, then all of problem 1's).  For producer -> consumer chains over a working set of Infinity-Cache size: the
// distinct-token GEMM writes the product table T (176 MB per D-ATT tower at cfg4) that the gather reads next, build_g writes the
// G that the sparse product reads -- run tower by tower the consumer finds its input in the 256 MB cache, paired it finds the
// OTHER tower's (measured, cfg4: gather 2 x 153 us tower by tower, 366 us as a pair; g_times_w 2 x 87 against 206).
struct PairSolo {
    PairSolo() { ++pair_state().solo; }
    ~PairSolo() { --pair_state().solo; }
    PairSolo(const PairSolo&) = delete;
    PairSolo& operator=(const PairSolo&) = delete;
};

template <class K> inline int raise_lds_limit(K kernel, size_t lds) {
    // above the default 64 KB of dynamic LDS a kernel needs its limit raised once (per process; the attribute is the
    // function's, set for the current device at first use)
    return check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                     "hipFuncSetAttribute(max dynamic LDS)");
}

template <auto Fn, int BOUNDS, int MINB, class... A> struct KernelOf {
    using P0 = Pack<A...>;
    using P1 = Pack1<A...>;
    static_assert(std::is_trivially_copyable_v<P0>, "kernel arguments must be trivially copyable");
    static_assert(sizeof(P0) <= kPairArgBytes, "argument pack larger than a pair record");

    static int run_single(const P0& p, dim3 g, dim3 b, size_t lds, hipStream_t st, const char* what) {
        if (lds > 64 * 1024) {
            static bool raised = false;
            if (!raised) {
                if (int e = raise_lds_limit(single_k<KTag<decltype(Fn), Fn>, BOUNDS, MINB, A...>, 160 * 1024)) return e;
                raised = true;
            }
        }
        hipLaunchKernelGGL((single_k<KTag<decltype(Fn), Fn>, BOUNDS, MINB, A...>), g, b, lds, st, p);
        return check_hip(hipGetLastError(), what);
    }
    static int single_fn(const PairRec& r) {
        P0 p;
        memcpy(&p, r.args, sizeof(P0));
        return run_single(p, r.grid, r.block, r.lds, r.st, r.what);
    }
    static int pair_fn(const PairRec& a, const PairRec& b) {
        if (a.grid.x != b.grid.x || a.grid.y != b.grid.y || a.grid.z != 1 || b.grid.z != 1 || a.block.x != b.block.x ||
            a.block.y != b.block.y || a.block.z != b.block.z || a.lds != b.lds || a.st != b.st)
            return kPairNoMatch;
        if (sizeof(P0) + sizeof(P1) > kKernargLimit) return kPairNoMatch;
        P0 p0, pb;
        memcpy(&p0, a.args, sizeof(P0));
        memcpy(&pb, b.args, sizeof(P0));
        if (!shared_equal(p0, pb)) return kPairNoMatch;
        P1 p1;
        fill_pack1(p1, pb);
        if (a.lds > 64 * 1024) {
            static bool raised = false;
            if (!raised) {
                if (int e = raise_lds_limit(pair_k<KTag<decltype(Fn), Fn>, BOUNDS, MINB, A...>, 160 * 1024)) return e;
                raised = true;
            }
        }
        dim3 g = a.grid;
        g.z = 2;
        hipLaunchKernelGGL((pair_k<KTag<decltype(Fn), Fn>, BOUNDS, MINB, A...>), g, a.block, a.lds, a.st, p0, p1);
        return check_hip(hipGetLastError(), a.what);
    }
    static int go(dim3 g, dim3 b, size_t lds, hipStream_t st, const char* what, A... a) {
        const P0 p = make_pack<A...>(a...);
        PairState& S = pair_state();
        if (S.mode == 0) return run_single(p, g, b, lds, st, what);
        PairRec r;
        r.pair = &pair_fn;
        r.single = &single_fn;
        r.grid = g; r.block = b; r.lds = lds; r.st = st; r.what = what; r.solo = S.solo > 0;
        memcpy(r.args, &p, sizeof(P0));
        S.rec[S.mode - 1].push_back(r);
        return 0;
    }
};
// the argument pack stores values whatever the body's parameters are (`const ConvPlan&` -> ConvPlan)
template <auto Fn, int BOUNDS, int MINB, class... A> KernelOf<Fn, BOUNDS, MINB, std::decay_t<A>...> kernel_of(void (*)(A...));

// rbr::launch<kernel body, launch bound[, min workgroups per CU]>(grid, block, dynamic LDS bytes, stream, "what", args...)
// -> 0 or an error code (the text is in rbr_last_error()).  Arguments convert to the body's parameter types as in a call.
template <auto Fn, int BOUNDS, int MINB = 1, class... Args>
inline int launch(dim3 g, dim3 b, size_t lds, hipStream_t st, const char* what, Args... args) {
    using K = decltype(kernel_of<Fn, BOUNDS, MINB>(Fn));
    return K::go(g, b, lds, st, what, args...);
}

}  // namespace rbr
