// Streaming-store patterns on gfx950: what shape of a pure write kernel reaches the memset rate?  (ceiling for basis_kernel)
//   hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o tools/store_pattern.bin ; tools/store_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr long B = 65536, M = 1024, COLS = 2;          // per buffer: [B][COLS][M] doubles
__global__ void k_linear(double2 *p, long n2, double v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) p[i] = make_double2(v, v);
}
// one wave per problem, rows outermost, both buffers (the shape of basis_kernel); WPB waves per block
template <int WPB, bool COL_OUTER, bool ONE_BUF>
__global__ void __launch_bounds__(64 * WPB) k_problem(double *ph, double *dp, double v) {
    const int lane = threadIdx.x & 63;
    const long b = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (b >= B) return;
    double *a = ph + b * COLS * M, *c = (ONE_BUF ? ph + (B + b) * COLS * M : dp + b * COLS * M);
    if constexpr (COL_OUTER) {
#pragma unroll
        for (int k = 0; k < COLS; ++k)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = 2 * lane + 128 * r;
                *reinterpret_cast<double2 *>(a + k * M + i) = make_double2(v, v + r);
                *reinterpret_cast<double2 *>(c + k * M + i) = make_double2(v, v - r);
            }
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = 2 * lane + 128 * r;
#pragma unroll
            for (int k = 0; k < COLS; ++k) {
                *reinterpret_cast<double2 *>(a + k * M + i) = make_double2(v, v + r);
                *reinterpret_cast<double2 *>(c + k * M + i) = make_double2(v, v - r);
            }
        }
    }
}
// persistent: each wave walks over problems with a grid stride
template <int WPB> __global__ void __launch_bounds__(64 * WPB) k_persistent(double *ph, double *dp, double v) {
    const int lane = threadIdx.x & 63;
    const long nw = (long)gridDim.x * WPB;
    for (long b = (long)blockIdx.x * WPB + (threadIdx.x >> 6); b < B; b += nw) {
        double *a = ph + b * COLS * M, *c = dp + b * COLS * M;
#pragma unroll
        for (int k = 0; k < COLS; ++k)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = 2 * lane + 128 * r;
                *reinterpret_cast<double2 *>(a + k * M + i) = make_double2(v, v + r);
                *reinterpret_cast<double2 *>(c + k * M + i) = make_double2(v, v - r);
            }
    }
}
// a workgroup owns a CONTIGUOUS span of problems and its threads sweep it linearly (what a memset does), buffer after buffer
__global__ void __launch_bounds__(256) k_span(double *ph, double *dp, double v, int per_block) {
    const long b0 = (long)blockIdx.x * per_block;
    double2 *a = reinterpret_cast<double2 *>(ph + b0 * COLS * M), *c = reinterpret_cast<double2 *>(dp + b0 * COLS * M);
    const long n2 = (long)per_block * COLS * M / 2;
    for (long i = threadIdx.x; i < n2; i += 256) a[i] = make_double2(v, v);
    for (long i = threadIdx.x; i < n2; i += 256) c[i] = make_double2(v, v);
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t;
    for (int i = 0; i < 24; ++i) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (i >= 4) t.push_back(ms); }
    std::sort(t.begin(), t.end()); return t[t.size() / 2];
}
int main() {
    const size_t n = (size_t)B * COLS * M;
    double *ph, *dp; hipMalloc(&ph, 2 * n * sizeof(double)); hipMalloc(&dp, n * sizeof(double));
    const double gb = 2.0 * n * 8 / 1e9;
    auto rep = [&](const char *name, float ms) { printf("%-46s %.3f ms  %6.0f GB/s  %.3f of 8 TB/s\n", name, ms, gb / ms * 1e3, gb / ms * 1e3 / 8000); };
    rep("hipMemsetAsync (2 x n)", timeit([&] { hipMemsetAsync(ph, 0, 2 * n * 8, 0); }));
    rep("linear grid-stride, 2048 x 256", timeit([&] { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, (double2 *)ph, (long)n, 1.0); }));
    rep("linear one pass, n2/256 blocks", timeit([&] { hipLaunchKernelGGL(k_linear, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (double2 *)ph, (long)n, 1.0); }));
    rep("problem/wave rows-outer 2 buffers WPB=4", timeit([&] { hipLaunchKernelGGL((k_problem<4, false, false>), dim3(B / 4), dim3(256), 0, 0, ph, dp, 1.0); }));
    rep("problem/wave cols-outer 2 buffers WPB=4", timeit([&] { hipLaunchKernelGGL((k_problem<4, true, false>), dim3(B / 4), dim3(256), 0, 0, ph, dp, 1.0); }));
    rep("problem/wave cols-outer 1 buffer  WPB=4", timeit([&] { hipLaunchKernelGGL((k_problem<4, true, true>), dim3(B / 4), dim3(256), 0, 0, ph, dp, 1.0); }));
    rep("problem/wave cols-outer 2 buffers WPB=1", timeit([&] { hipLaunchKernelGGL((k_problem<1, true, false>), dim3(B), dim3(64), 0, 0, ph, dp, 1.0); }));
    rep("problem/wave cols-outer 2 buffers WPB=8", timeit([&] { hipLaunchKernelGGL((k_problem<8, true, false>), dim3(B / 8), dim3(512), 0, 0, ph, dp, 1.0); }));
    rep("persistent 2048 waves (512 x 4)", timeit([&] { hipLaunchKernelGGL((k_persistent<4>), dim3(512), dim3(256), 0, 0, ph, dp, 1.0); }));
    rep("persistent 8192 waves (2048 x 4)", timeit([&] { hipLaunchKernelGGL((k_persistent<4>), dim3(2048), dim3(256), 0, 0, ph, dp, 1.0); }));
    rep("contiguous span per block, 16 problems", timeit([&] { hipLaunchKernelGGL(k_span, dim3(B / 16), dim3(256), 0, 0, ph, dp, 1.0, 16); }));
    rep("contiguous span per block, 4 problems", timeit([&] { hipLaunchKernelGGL(k_span, dim3(B / 4), dim3(256), 0, 0, ph, dp, 1.0, 4); }));
    return 0;
}
