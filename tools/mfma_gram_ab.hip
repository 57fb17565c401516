// A/B microbenchmark for BASELINE.json north_star's "MFMA only for the J^T J / Phi^T Phi contractions where m is large":
// the Gram round of the Jacobian QR at configs[4] geometry (fp32, q = 5 Jacobian columns + the residual = 6 columns,
// m = 4096 rows over a group of 4 waves x 64 lanes x 16 rows per lane, columns resident in registers).
//   A  (what the fit kernels do, vp_fit.hpp:jac_qrfac_scaled): 21 dot products as v_fma_f32 chains over the lane's
//      16 rows, ONE packed wave all-reduce of the 21 values (DPP / permlane), LDS exchange + barrier over the 4 waves.
//   B  (MFMA): the 6 columns padded to 16, G = Z^T Z through v_mfma_f32_16x16x4_f32.  The MFMA A/B operands want lane
//      (i + 16 k) to hold column i of row k, the resident layout has a lane hold 16 ROWS of one column per register:
//      the columns are re-staged through LDS ([row][16] floats, 128 rows per wave at a time) and read back in operand
//      layout, 4 rows per MFMA, 256 MFMAs per wave, then the 16x16 accumulators of the 4 waves are summed through LDS.
// Build + run: hipcc -O3 --offload-arch=gfx950 -I varpro_amd/csrc tools/mfma_gram_ab.hip -o /tmp/mfma_gram_ab && /tmp/mfma_gram_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "vp_device.hpp"

constexpr int NCOL = 6, R = 16, W = 4, NG = NCOL * (NCOL + 1) / 2; // 21 Gram entries

__device__ __forceinline__ void make_columns(float (&Z)[NCOL][R], int gl, int it) {
#pragma unroll
    for (int c = 0; c < NCOL; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r) Z[c][r] = __sinf(0.001f * (float)((r * 256 + gl) * (c + 1) + it));
}

__global__ void __launch_bounds__(256, 2) gram_valu(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char xch[vp::group_xch_bytes<W>()];
    using G = vp::Grp<W>;
    G grp = G::make(xch);
    float Z[NCOL][R];
    float acc_out = 0.f;
    for (int it = 0; it < iters; ++it) {
        make_columns(Z, grp.gl, it);
        float g[NG];
        int idx = 0;
#pragma unroll
        for (int a = 0; a < NCOL; ++a)
#pragma unroll
            for (int b = a; b < NCOL; ++b) {
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) acc = __builtin_fmaf(Z[a][r], Z[b][r], acc);
                g[idx++] = acc;
            }
        vp::group_allreduce(grp, g);
#pragma unroll
        for (int i = 0; i < NG; ++i) acc_out += g[i];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc_out;
}

typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) gram_mfma(float *out, int iters) {
    // per wave: 128 rows (one register pair of every column) x 16 (padded) columns staged in LDS at a time = 8 KiB
    __shared__ __attribute__((aligned(16))) float lds[4 * 128 * 16 + 4 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x;
    float *mine = lds + wave * 128 * 16;
    float *red = lds + 4 * 128 * 16; // [4][256] partial Gram tiles
    float Z[NCOL][R];
    float acc_out = 0.f;
    for (int i = lane; i < 128; i += 64)
        for (int c = NCOL; c < 16; ++c) mine[i * 16 + c] = 0.f; // pad columns stay zero
    for (int it = 0; it < iters; ++it) {
        make_columns(Z, gl, it);
        float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 2) {
            // re-stage this register pair: the lane's rows lane*2, lane*2+1 of the 128-row chunk -> [row][col]
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int c = 0; c < NCOL; ++c) mine[(lane * 2 + e) * 16 + c] = Z[c][r0 + e];
            __builtin_amdgcn_wave_barrier();
            // 4 rows per MFMA: operand element of lane l = Z[row0 + l/16][l%16] for both A (16x4) and B (4x16)
#pragma unroll
            for (int row0 = 0; row0 < 128; row0 += 4) {
                const float v = mine[(row0 + (lane >> 4)) * 16 + (lane & 15)];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v, v, acc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // sum the 4 waves' tiles through LDS
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave * 256 + k * 64 + lane] = acc[k];
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += red[k * 256 + gl];
        acc_out += s;
        __syncthreads();
    }
    atomicAdd(&out[blockIdx.x], acc_out);
}

int main() {
    const int blocks = 512, iters = 200;
    float *out;
    hipMalloc(&out, blocks * sizeof(float));
    hipMemset(out, 0, blocks * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms_a = 0, ms_b = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(gram_valu, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_a, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(gram_mfma, dim3(blocks), dim3(256), 0, 0, out, iters);
        if (hipGetLastError() != hipSuccess) { printf("{\"error\": \"mfma launch failed\"}\n"); return 1; }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_b, e0, e1);
    }
    // both kernels also spend the same time generating the columns (6 x 16 sinf per lane and iteration)
    const double rounds = (double)blocks * iters;
    const double useful = 2.0 * NG * 4096;                          // flops of the 21 dot products over 4096 rows
    const double mfma_issued = 4.0 * 256 * (16 * 16 * 4 * 2);       // 4 waves x 256 MFMAs x 2048 flop
    printf("{\"geometry\": \"fp32, 6 columns x 4096 rows per problem, 4 waves x 16 rows per lane, %d workgroups x %d rounds\", "
           "\"valu_ms\": %.4f, \"mfma_ms\": %.4f, \"valu_us_per_round_512_problems\": %.4f, \"mfma_us_per_round_512_problems\": %.4f, "
           "\"mfma_over_valu\": %.2f, \"mfma_useful_flop_fraction\": %.4f, "
           "\"mfma_pipe_utilisation_of_157TF\": %.4f, \"note\": \"both variants include the identical column generation; "
           "useful flops = 21 dot products x 2 x 4096; the MFMA variant issues 16x16x4 tiles for a 6x6 Gram and must "
           "re-stage the register-resident columns through LDS into operand layout\"}\n",
           blocks, iters, ms_a, ms_b, ms_a * 1e3 / iters, ms_b * 1e3 / iters,
           ms_b / ms_a, useful / mfma_issued, (rounds * mfma_issued / (ms_b * 1e-3)) / 157.3e12);
    return 0;
}
