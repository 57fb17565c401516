// vp_device.hpp -- gfx950 device primitives for the batched variable-projection kernels.
//
// Execution model: ONE 64-lane wavefront owns ONE separable problem.
// The m observations are spread over the lanes, R = ceil(m/64) rows per lane, and every
// column the algorithm touches (the basis columns of Phi, the data column y, the derivative
// columns dPhi) lives in VGPRs for the whole evaluation -- Phi never exists in memory.
// All reductions (column norms, Householder dot products) are wave-level and PACKED (wave_allreduce: gfx950
// v_permlane32/16_swap for the two row-level steps, DPP inside the 16-lane rows, v_readlane to broadcast): no
// LDS traffic, no barriers.  Problems whose columns do not fit one wave use a group of W waves (Grp<W>), whose
// reductions add one LDS exchange + barrier.
// Developer A/B switches (off in product builds): VP_NO_PACKED (one DPP butterfly per value), VP_NO_USQRT (IEEE sqrt
// expansion), VP_NO_SWEEP_FENCE (vp_core.hpp), VP_FIT_CLOCKS (per-section cycle accounting, tools/fit_clocks.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// LDS-typed pointers (address space 3): accesses through them stay ds_* instructions in out-of-line functions
#define VP_LDS __attribute__((address_space(3)))

namespace vp {

template <typename T> struct num;
template <> struct num<double> {
    static constexpr double eps = 2.220446049250313e-16;
    static constexpr double tiny = 2.2250738585072014e-308; // min positive normal
    static constexpr double huge = 1.7976931348623157e+308;
    // smallest squared column norm a Householder reflector is formed for: below it the unnormalised reflector's scalar
    // g = -1/(sigma (|alpha| + sigma)) ~ 1/sigma^2 overflows; such a column is numerically zero for every consumer (its
    // singular value is far below any svd_epsilon; a Jacobian column of that size does not move the step) and gets H = I
    static constexpr double norm2_min = 1e-290;
};
template <> struct num<float> {
    static constexpr float eps = 1.1920929e-07f;
    static constexpr float tiny = 1.17549435e-38f;
    static constexpr float huge = 3.4028235e+38f;
    static constexpr float norm2_min = 1e-30f;
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// the lane index recomputed on the spot (two instructions) and opaque to the optimiser: a lane-dependent address that is
// needed once per loop iteration is otherwise kept -- i.e. SPILLED and reloaded -- across the register-critical part of the
// iteration, and a scratch reload is a VMEM wait (vmcnt) the asynchronous refill of vp_fit2.hpp cannot afford
#ifndef VP_LANE_FRESH
#define VP_LANE_FRESH 1
#endif
__device__ __forceinline__ int lane_fresh() {
    if (!VP_LANE_FRESH) return lane_id();
    unsigned z = 0u; // (the count starts from an opaque zero: the mbcnt pair itself must not be hoisted out of the caller's loop)
    asm volatile("" : "+v"(z));
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}

// make a wave-uniform predicate visible to the compiler as scalar so that branches on it are
// s_cbranch (no exec-mask juggling around the DPP/readlane code below)
#ifdef VP_UNI_READFIRSTLANE
__device__ __forceinline__ bool uni(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
#else
// the compare already writes its lane mask to an SGPR pair: testing that mask (ballot) costs no VALU instruction,
// where materialising the bool in a VGPR and reading lane 0 costs two plus the VALU->SALU round trip
__device__ __forceinline__ bool uni(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }
#endif
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- optional per-section cycle accounting of the fit kernel (tools/fit_clocks.py; -DVP_FIT_CLOCKS) -------------
struct SectionClock {
    long long last;
    long long acc[12];
    __device__ __forceinline__ void start() {
        for (int k = 0; k < 12; ++k) acc[k] = 0;
        last = (long long)__builtin_readcyclecounter();
    }
    __device__ __forceinline__ void tick(int k) {
        const long long now = (long long)__builtin_readcyclecounter();
        acc[k] += now - last;
        last = now;
    }
};
#ifdef VP_FIT_CLOCKS
#define VP_TICK(clk, k) do { if (clk) (clk)->tick(k); } while (0)
#else
#define VP_TICK(clk, k) do { } while (0)
#endif

// ---- lane <-> row mapping ---------------------------------------------------------------------
// A problem is owned by a GROUP of W wavefronts (W = 1: one wavefront; W > 1: one workgroup of W waves for
// problems whose columns do not fit the registers of one wave).  Rows are dealt to the 64*W group lanes in
// pairs so that a lane's global/LDS accesses are 16 B wide for fp64: register r of group lane gl holds
// row ((r/VW)*64*W + gl)*VW + r%VW.
template <int R, int W = 1> struct Layout {
    static constexpr int VW = (R >= 2) ? 2 : 1;
    static constexpr int W_ = W;
    static_assert(R == 1 || R % 2 == 0, "R must be 1 or even");
    __device__ __forceinline__ static int row_of(int r, int gl) { return ((r / VW) * (64 * W) + gl) * VW + (r % VW); }
    // rows < 64*W*VW (all pivot rows) live in the first VW registers
    __host__ __device__ static constexpr int reg_of_row(int row) { return row % VW; }
    __host__ __device__ static constexpr int lane_of_row(int row) { return row / VW; } // group lane
};

// exchange area of a multi-wave group in LDS: two phases of all-reduce slots + two phases of broadcast slots
constexpr int VP_XV = 24; // max values per group all-reduce (Gram round of jac_qrfac: Q(Q+1)/2 + Q)
constexpr int VP_XB = 16; // max values per group broadcast
template <int W> constexpr int group_xch_bytes() { return W > 1 ? (2 * W * VP_XV + 2 * VP_XB) * 8 : 0; }

// the group a lane belongs to
template <int W_> struct Grp {
    static constexpr int W = W_;
    int lane;           // lane within the wave (0..63)
    int wave;           // wave within the group (0..W-1)
    int gl;             // group lane = wave*64 + lane: the index used by Layout::row_of
    unsigned char *xch; // LDS exchange area (W > 1), group_xch_bytes<W>() bytes, 8-byte aligned
    int phase;          // toggles between the two buffers of each area on every exchange
    __device__ __forceinline__ static Grp make(unsigned char *xch_) {
        Grp g;
        g.lane = (int)(threadIdx.x & 63u);
        g.wave = (W_ > 1) ? (int)(threadIdx.x >> 6) : 0;
        g.gl = g.wave * 64 + g.lane;
        g.xch = xch_;
        g.phase = 0;
        return g;
    }
};

// ---- cross-lane data movement -----------------------------------------------------------------
// DPP move: lanes of rows enabled in ROW_MASK receive the permuted value, all other lanes receive 0
// (old = 0 and a fresh destination register: no tied-operand copies, 1 v_mov_b32_dpp per dword).
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ double dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp(float x) {
    int v = __float_as_int(x);
    v = __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    return __int_as_float(v);
}
__device__ __forceinline__ double readlane(double x, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane(float x, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}

// DPP controls (GFX9): quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror, row_mirror,
// row_bcast:15 (lane 15 of each row -> the next row), row_bcast:31 (lane 31 -> rows 2 and 3)
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
constexpr int DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143;

constexpr int DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;

// ---- packed all-reduce ---------------------------------------------------------------------------
// V values are reduced TOGETHER: at each of the first four levels two values are merged into one register
// whose two lane halves (with respect to one lane-index bit) carry the partial sums of the two values, so
// the number of live values halves per level instead of every value paying for every level:
//   level 0  lane bit 5   v_permlane32_swap (gfx950): a'=[a.lo|b.lo], b'=[a.hi|b.hi], a'+b' -> 3 instr / fp64 pair
//   level 1  lane bit 4   v_permlane16_swap (gfx950): same with the odd/even 16-lane rows
//   level 2  lane bit 0   DPP quad_perm xor 1 with keep/send selects                         -> 7 instr / pair
//   level 3  lane bit 1   DPP quad_perm xor 2
// and the (at most ceil(V/16)) survivors finish with plain row_ror:4 / row_ror:8 steps.  The total of value
// v (v < 16) then sits in lane 32*b0(v) + 16*b1(v) + b2(v) + 2*b3(v) and is broadcast with v_readlane.
// V = 6 costs 45 instructions instead of the 120 of six independent butterflies.
template <typename T> struct SwapPair {
    T a, b;
};
template <int LEVEL> __device__ __forceinline__ SwapPair<double> lane_swap(double a, double b) {
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if constexpr (LEVEL == 0) {
        auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        return {__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1])};
    } else {
        auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return {__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1])};
    }
}
template <int LEVEL> __device__ __forceinline__ SwapPair<float> lane_swap(float a, float b) {
    unsigned ua = (unsigned)__float_as_int(a), ub = (unsigned)__float_as_int(b);
    if constexpr (LEVEL == 0) {
        auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
        return {__int_as_float((int)r[0]), __int_as_float((int)r[1])};
    } else {
        auto r = __builtin_amdgcn_permlane16_swap(ua, ub, false, false);
        return {__int_as_float((int)r[0]), __int_as_float((int)r[1])};
    }
}
// merge two values at one level: lanes whose level bit is 0 end with a's partial sum, the others with b's
template <int LEVEL, typename T> __device__ __forceinline__ T merge_level(T a, T b) {
    if constexpr (LEVEL <= 1) {
        const SwapPair<T> s = lane_swap<LEVEL>(a, b);
        return s.a + s.b;
    } else {
        const bool hi = (lane_id() & (LEVEL == 2 ? 1 : 2)) != 0;
        const T keep = hi ? b : a, send = hi ? a : b;
        if constexpr (LEVEL == 2) return keep + dpp<DPP_XOR1>(send);
        else return keep + dpp<DPP_XOR2>(send);
    }
}
// a value without a partner at this level: only the bit-0 half of the lanes needs its sum
template <int LEVEL, typename T> __device__ __forceinline__ T single_level(T a) {
    if constexpr (LEVEL <= 1) return merge_level<LEVEL>(a, T(0));
    else if constexpr (LEVEL == 2) return a + dpp<DPP_XOR1>(a);
    else return a + dpp<DPP_XOR2>(a);
}
template <int LEVEL, int V, typename T> __device__ __forceinline__ void pack_level(const T (&x)[V], T (&y)[(V + 1) / 2]) {
#pragma unroll
    for (int i = 0; i < V / 2; ++i) y[i] = merge_level<LEVEL>(x[2 * i], x[2 * i + 1]);
    if constexpr (V % 2 == 1) y[V / 2] = single_level<LEVEL>(x[V - 1]);
}
__host__ __device__ constexpr int packed_lane_of(int v) {
    return 32 * (v & 1) + 16 * ((v >> 1) & 1) + ((v >> 2) & 1) + 2 * ((v >> 3) & 1);
}

// All-reduce (sum) of V independent values across the 64 lanes; every lane ends with the totals,
// delivered through v_readlane (SGPRs), i.e. as genuinely scalar values.
template <int V, typename T> __device__ __forceinline__ void wave_allreduce(T (&x)[V]) {
#ifdef VP_NO_PACKED
    {
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] += dpp<DPP_XOR1>(x[v]);
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] += dpp<DPP_XOR2>(x[v]);
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] += dpp<DPP_HALF_MIRROR>(x[v]);
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] += dpp<DPP_MIRROR>(x[v]);
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] += dpp<DPP_BCAST15, 0xA>(x[v]); // rows 1,3 += rows 0,2
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] += dpp<DPP_BCAST31, 0xC>(x[v]); // rows 2,3 += row 1 (= rows 0+1)
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] = readlane(x[v], 63);
    }
#else
    {
        constexpr int V1 = (V + 1) / 2, V2 = (V1 + 1) / 2, V3 = (V2 + 1) / 2, V4 = (V3 + 1) / 2;
        T y1[V1], y2[V2], y3[V3], y4[V4];
        pack_level<0>(x, y1);
        pack_level<1>(y1, y2);
        pack_level<2>(y2, y3);
        pack_level<3>(y3, y4);
#pragma unroll
        for (int i = 0; i < V4; ++i) y4[i] += dpp<DPP_ROR4>(y4[i]);
#pragma unroll
        for (int i = 0; i < V4; ++i) y4[i] += dpp<DPP_ROR8>(y4[i]);
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] = readlane(y4[v >> 4], packed_lane_of(v & 15));
    }
#endif
}
template <typename T> __device__ __forceinline__ T wave_sum(T x) {
    T a[1] = {x};
    wave_allreduce(a);
    return a[0];
}

// All-reduce over the whole group.  W == 1: the wave all-reduce.  W > 1: wave all-reduce, lane 0 of every
// wave posts its V sums in LDS, ONE barrier, every lane adds the W partials in the same fixed order (so all
// waves hold bit-identical totals and take identical control-flow decisions).  Double-buffered by g.phase:
// one barrier per exchange is enough.
template <int V, typename T, class G> __device__ __forceinline__ void group_allreduce(G &g, T (&x)[V]) {
    wave_allreduce(x);
    if constexpr (G::W > 1) {
        static_assert(V <= VP_XV, "too many values in one group all-reduce");
        T *buf = reinterpret_cast<T *>(g.xch + (size_t)g.phase * (G::W * VP_XV * 8));
        if (g.lane == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v) buf[g.wave * VP_XV + v] = x[v];
        }
        __syncthreads();
        if constexpr (G::W <= 4) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                T s = buf[v];
#pragma unroll
                for (int w = 1; w < G::W; ++w) s += buf[w * VP_XV + v];
                x[v] = s;
            }
        } else {
            // (8 / 16 waves: unrolled, all V * W loads are hoisted above the additions -- V * W live registers in kernels that
            // have none to spare; the partials are added wave by wave in the same fixed order)
#pragma unroll
            for (int v = 0; v < V; ++v) x[v] = buf[v];
#pragma nounroll
            for (int w = 1; w < G::W; ++w) {
#pragma unroll
                for (int v = 0; v < V; ++v) x[v] += buf[w * VP_XV + v];
            }
        }
        g.phase ^= 1;
    }
}
template <typename T, class G> __device__ __forceinline__ T group_sum(G &g, T x) {
    T a[1] = {x};
    group_allreduce(g, a);
    return a[0];
}

// Broadcast NV values held by group lane `owner_gl` (compile-time after unrolling) to the whole group.
template <int NV, typename T, class G> __device__ __forceinline__ void group_bcast(G &g, T (&v)[NV], int owner_gl) {
    if constexpr (G::W == 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = readlane(v[i], owner_gl);
    } else {
        static_assert(NV <= VP_XB, "too many values in one group broadcast");
        T *buf = reinterpret_cast<T *>(g.xch + (size_t)(2 * G::W * VP_XV * 8) + (size_t)g.phase * (VP_XB * 8));
        if (g.gl == owner_gl) {
#pragma unroll
            for (int i = 0; i < NV; ++i) buf[i] = v[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = buf[i];
        g.phase ^= 1;
    }
}

// element at (compile-time) row `row` of a register-resident column, broadcast to the group
template <int R, typename T, class G> __device__ __forceinline__ T group_row(G &g, const T (&col)[R], int row) {
    using L = Layout<R, G::W>;
    T v[1] = {col[L::reg_of_row(row)]};
    group_bcast<1>(g, v, L::lane_of_row(row));
    return v[0];
}

// ---- tiny dynamically-indexed uniform arrays without scratch -----------------------------------
template <int N, typename T> __device__ __forceinline__ T dyn_get(const T (&a)[N], int idx) {
    T v = a[0];
#pragma unroll
    for (int j = 1; j < N; ++j) v = (idx == j) ? a[j] : v;
    return v;
}
template <int N, typename T> __device__ __forceinline__ void dyn_set(T (&a)[N], int idx, T val) {
#pragma unroll
    for (int j = 0; j < N; ++j) a[j] = (idx == j) ? val : a[j];
}
// Opaque variants (O = true): every element passes through an empty asm, so that the optimiser cannot fold the select chain
// back into a dynamically indexed load / store on the array (select of loads -> load of a selected address), which pins the
// array -- and any struct it is a member of -- in scratch memory.  Needed where the arrays are members of a state struct
// (LmVars, vp_lm_core.hpp: 512 B of scratch and ~260 scratch accesses in mrhs_lm_kernel without it); the single-RHS fit
// kernels keep the plain form (their arrays are small locals and end up in registers either way).
__device__ __forceinline__ double dyn_opq(double x) {
    asm("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float dyn_opq(float x) {
    asm("" : "+v"(x));
    return x;
}
__device__ __forceinline__ int dyn_opq(int x) {
    asm("" : "+v"(x));
    return x;
}
template <int N, bool O, typename T> __device__ __forceinline__ T dyn_get_o(const T (&a)[N], int idx) {
    if constexpr (!O) return dyn_get<N>(a, idx);
    T v = dyn_opq(a[0]);
#pragma unroll
    for (int j = 1; j < N; ++j) v = (idx == j) ? dyn_opq(a[j]) : v;
    return v;
}
template <int N, bool O, typename T> __device__ __forceinline__ void dyn_set_o(T (&a)[N], int idx, T val) {
    if constexpr (!O) {
        dyn_set<N>(a, idx, val);
        return;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) a[j] = (idx == j) ? val : dyn_opq(a[j]);
}

template <typename T> __device__ __forceinline__ bool is_finite(T x) { return (x - x) == T(0); }
template <typename T> __device__ __forceinline__ T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> __device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }
__device__ __forceinline__ double tsqrt(double x) { return __builtin_sqrt(x); }
__device__ __forceinline__ float tsqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double tabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float tabs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double tcopysign(double x, double s) { return __builtin_copysign(x, s); }
__device__ __forceinline__ float tcopysign(float x, float s) { return __builtin_copysignf(x, s); }
__device__ __forceinline__ double tldexp(double x, int e) { return __builtin_ldexp(x, e); }
__device__ __forceinline__ float tldexp(float x, int e) { return __builtin_ldexpf(x, e); }
__device__ __forceinline__ double tfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float tfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Fast wave-uniform scalar helpers for the LM bookkeeping (trust region, lmpar, Givens rotations).
// v_rcp_f64 / v_rsq_f64 deliver ~26 good bits; two Newton steps bring them to the last 1-2 ulp at
// 5-8 instructions instead of the 12-14 of the IEEE division / sqrt expansions.  They are NOT used
// where parity is priced (Householder betas, the triangular solve for c, exp arguments).
__device__ __forceinline__ double frcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float frcp(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
}
__device__ __forceinline__ double frsqrt(double h) {
    double y = __builtin_amdgcn_rsq(h);
    double e = __builtin_fma(-h * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    e = __builtin_fma(-h * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    return y;
}
__device__ __forceinline__ float frsqrt(float h) {
    float y = __builtin_amdgcn_rsqf(h);
    float e = __builtin_fmaf(-h * y, y, 1.0f);
    return __builtin_fmaf(0.5f * y, e, y);
}
// sqrt for moderate-magnitude uniform scalars (0 -> 0; no denormal/huge rescaling)
template <typename T> __device__ __forceinline__ T fsqrt(T h) {
    if (!(h > T(0))) return tsqrt(h); // 0, negative or NaN: IEEE path
    const T y = frsqrt(h);
    T s = h * y;
    s = tfma(tfma(-s, s, h), T(0.5) * y, s);
    return s;
}

// sqrt for the bookkeeping scalars (Householder betas, column norms): the rsq + Newton form, faithfully rounded,
// BRANCH-FREE (a branch per call chops the surrounding code into tiny scheduling regions): 0 and +inf, for
// which rsq * h is 0 * inf, are patched by one class test + select; negative / NaN arguments give NaN like
// the IEEE expansion; fp64 denormals are handled by v_rsq_f64 itself.  12 instructions instead of ~25.
__device__ __forceinline__ double usqrt(double h) {
#ifdef VP_NO_USQRT
    return tsqrt(h);
#endif
    const double y = frsqrt(h);
    const double s0 = h * y;
    const double s = tfma(tfma(-s0, s0, h), 0.5 * y, s0);
    return (h == 0.0 || h == __builtin_inf()) ? h : s;
}
__device__ __forceinline__ float usqrt(float h) { return tsqrt(h); }

// correctly rounded (to within the last bit in rare ties) quotient a/b given rb ~= 1/b:
// one Newton correction of the product.  Replaces the 10+ instruction IEEE division sequence
// for the per-row argument -t/tau; rb is computed once per problem.
template <typename T> __device__ __forceinline__ T div_refined(T a, T b, T rb) {
    T q0 = a * rb;
    T rem = tfma(-q0, b, a);
    return tfma(rem, rb, q0);
}

} // namespace vp
