// Streaming-READ patterns on gfx950: what shape of a pure read kernel reaches the HBM rate?  (ceiling for the MRHS stream
// kernels, which read 268 MB of Y per pass at BASELINE configs[2])
//   hipcc --offload-arch=gfx950 -O3 tools/read_pattern.hip -o tools/read_pattern.bin ; tools/read_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr long NBYTES = 268435456L;               // S = 16384 columns x 16 KiB
constexpr long N2 = NBYTES / 16;                  // double2 elements
// grid-stride linear sweep, U independent 16-byte loads per thread in flight
template <int U> __global__ void __launch_bounds__(512) k_linear(const double2 *p, double *out, long n2) {
    double acc = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n2; i += stride * U) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (i + u * stride < n2) ? p[i + u * stride] : make_double2(0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 12345.678) out[0] = acc;
}
// the cooperative MRHS shape: workgroup b walks batches b, b + G, ... of NB columns (16 KiB each); 512 threads, every
// thread loads RW/2 row pairs of every column of the batch; PF batches prefetched in registers
template <int NB, int PF> __global__ void __launch_bounds__(512) k_coop(const double2 *p, double *out, long ncol) {
    constexpr int NPAIR = 2;                       // 2048 rows / 512 lanes / 2
    double acc = 0.0;
    const long nbatch = ncol / NB;
    double2 buf[PF + 1][NB][NPAIR];
    auto load = [&](int slot, long bt) {
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int k = 0; k < NPAIR; ++k) buf[slot][c][k] = p[(bt * NB + c) * 1024 + k * 512 + threadIdx.x];
    };
    long bt = blockIdx.x;
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (bt + s * (long)gridDim.x < nbatch) load(s, bt + s * (long)gridDim.x);
    int it = 0;
    for (; bt < nbatch; bt += gridDim.x, ++it) {
        // rotate: consume slot 0 (compile-time indices only: shift the window)
        double2 cur[NB][NPAIR];
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int k = 0; k < NPAIR; ++k) cur[c][k] = buf[0][c][k];
#pragma unroll
        for (int s = 0; s + 1 < PF; ++s)
#pragma unroll
            for (int c = 0; c < NB; ++c)
#pragma unroll
                for (int k = 0; k < NPAIR; ++k) buf[s][c][k] = buf[s + 1][c][k];
        if (bt + PF * (long)gridDim.x < nbatch) load(PF - 1, bt + PF * (long)gridDim.x);
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int k = 0; k < NPAIR; ++k) acc += cur[c][k].x * 1.0000001 + cur[c][k].y;
    }
    if (acc == 12345.678) out[0] = acc;
}
template <class F> float time_it(F f, int reps = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
int main() {
    double2 *p; double *out;
    hipMalloc(&p, NBYTES); hipMalloc(&out, 8);
    hipMemset(p, 0, NBYTES);
    auto rep = [&](const char *name, float ms) { printf("%-44s %8.1f us  %7.2f TB/s\n", name, ms * 1e3, NBYTES / (ms * 1e-3) / 1e12); };
    for (int blocks : {256, 512, 1024, 2048}) {
        char nm[96];
        snprintf(nm, 96, "linear U=1 blocks=%d x512", blocks); rep(nm, time_it([&] { hipLaunchKernelGGL(k_linear<1>, dim3(blocks), dim3(512), 0, 0, p, out, N2); }));
        snprintf(nm, 96, "linear U=4 blocks=%d x512", blocks); rep(nm, time_it([&] { hipLaunchKernelGGL(k_linear<4>, dim3(blocks), dim3(512), 0, 0, p, out, N2); }));
        snprintf(nm, 96, "linear U=8 blocks=%d x512", blocks); rep(nm, time_it([&] { hipLaunchKernelGGL(k_linear<8>, dim3(blocks), dim3(512), 0, 0, p, out, N2); }));
    }
    for (int blocks : {256, 512}) {
        char nm[96];
        snprintf(nm, 96, "coop NB=4 PF=1 blocks=%d", blocks); rep(nm, time_it([&] { hipLaunchKernelGGL((k_coop<4, 1>), dim3(blocks), dim3(512), 0, 0, p, out, 16384L); }));
        snprintf(nm, 96, "coop NB=4 PF=2 blocks=%d", blocks); rep(nm, time_it([&] { hipLaunchKernelGGL((k_coop<4, 2>), dim3(blocks), dim3(512), 0, 0, p, out, 16384L); }));
        snprintf(nm, 96, "coop NB=4 PF=4 blocks=%d", blocks); rep(nm, time_it([&] { hipLaunchKernelGGL((k_coop<4, 4>), dim3(blocks), dim3(512), 0, 0, p, out, 16384L); }));
        snprintf(nm, 96, "coop NB=8 PF=2 blocks=%d", blocks); rep(nm, time_it([&] { hipLaunchKernelGGL((k_coop<8, 2>), dim3(blocks), dim3(512), 0, 0, p, out, 16384L); }));
    }
    rep("hipMemcpy D2D (read+write, 2x bytes)", time_it([&] { hipMemcpyAsync((char *)p + NBYTES / 2, p, NBYTES / 2, hipMemcpyDeviceToDevice, 0); }) * 1.0f);
    return 0;
}
