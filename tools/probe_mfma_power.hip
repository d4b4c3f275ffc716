// Probe: sustained matrix-core throughput of gfx950 AT THE PACKAGE POWER CAP as a function of the MFMA shape and of the
// operand data (register-resident operands, no LDS / HBM traffic).  Run under tools/smi_watch.sh; see
// profiles/r02_power_cap.md for the reading.   usage: probe_mfma_power <mode> [seconds]
//   0  v_mfma_f32_32x32x16_f16, operands U[-1,1)          1  v_mfma_f32_16x16x32_f16, same data
//   2  32x32x16, all-zero operands                        3  32x32x16, B half zeros (post-ReLU), A = weights (+-1/38)
//   4  32x32x16, B = "lo plane" data (2^-11 x U[-1,1))    5  16x16x32, all-zero operands
//   6  32x32x16, both operands change on every MFMA (mode 0 keeps B for 4 consecutive MFMAs)
//   7  32x32x16, A half zeros (post-ReLU), B = weights (+-1/38): mode 3 with the operands swapped
//   8  32x32x16, both operands half zeros
//   9  v_mfma_i32_32x32x32_i8, random int8 operands (ops counted like flops)      10  the same, all-zero operands
//  11  i8, B half zeros (post-ReLU magnitudes), A = small signed "weight" bytes
//  12 / 13 / 15  32x32x16, B = lo-plane data whose mantissa keeps only its top 5 / 3 / 0 bits (round 6: does a coarser lo plane cost fewer joules?)
//  16  both operands U[-1,1) with 5-bit mantissas        17  B = lo-plane data, 7-bit mantissa
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float u01(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(x >> 8) * (1.f / 16777216.f);
}
__device__ __forceinline__ half8 frag(unsigned seed, int mode, bool is_b) {
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float r = 2.f * u01(seed * 8u + e) - 1.f;
        if (mode == 2 || mode == 5) r = 0.f;
        if (mode == 3) r = is_b ? fmaxf(r, 0.f) : r * (1.f / 38.f);
        if (mode == 7) r = is_b ? r * (1.f / 38.f) : fmaxf(r, 0.f);
        if (mode == 8) r = fmaxf(r, 0.f);
        if ((mode == 4 || mode == 12 || mode == 13 || mode == 15 || mode == 17) && is_b) r *= (1.f / 2048.f);
        v[e] = (_Float16)r;
        const int keep = mode == 12 ? 5 : mode == 13 ? 3 : mode == 15 ? 0 : mode == 16 ? 5 : mode == 17 ? 7 : 10;
        if (keep < 10 && (is_b || mode == 16)) {
            union { _Float16 h; unsigned short u; } c;
            c.h = v[e];
            c.u &= (unsigned short)(0xFFFFu << (10 - keep));
            v[e] = c.h;
        }
    }
    return v;
}

template <int MODE>
__global__ void __launch_bounds__(256) probe(int iters, float* out) {
    const unsigned lane_seed = (blockIdx.x * 256u + threadIdx.x) * 64u;
    half8 A[4], B[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A[i] = frag(lane_seed + i, MODE, false);
        B[i] = frag(lane_seed + 16 + i, MODE, true);
    }
    float sum = 0.f;
    if constexpr (MODE >= 9 && MODE <= 11) {
        intx4 Ai[4], Bi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned ra = 0, rb = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int va = (int)(255.f * u01((lane_seed + i) * 64u + e * 4 + b)) - 128;
                    int vb = (int)(255.f * u01((lane_seed + 16 + i) * 64u + e * 4 + b)) - 128;
                    if (MODE == 10) va = vb = 0;
                    if (MODE == 11) { vb = vb < 0 ? 0 : vb; va = va / 8; }
                    ra |= (unsigned)(va & 0xff) << (8 * b);
                    rb |= (unsigned)(vb & 0xff) << (8 * b);
                }
                Ai[i][e] = (int)ra; Bi[i][e] = (int)rb;
            }
        intx16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            // 16 MFMAs of 32x32x32 = twice the multiply-adds of 16 MFMAs of 32x32x16
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ai[i], Bi[j], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += (float)(acc[i][0] + acc[i][15]);
    } else if constexpr (MODE == 1 || MODE == 5) {
        floatx4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            // 32 MFMAs of 16x16x32 = the flops of 16 MFMAs of 32x32x16
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i], B[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
    } else {
        floatx16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], B[MODE == 6 ? ((i + j) & 3) : j], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][15];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 4.0;
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 2, iters = 20000;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    auto launch = [&]() {
        switch (mode) {
            case 0: probe<0><<<blocks, 256>>>(iters, out); break;
            case 1: probe<1><<<blocks, 256>>>(iters, out); break;
            case 2: probe<2><<<blocks, 256>>>(iters, out); break;
            case 3: probe<3><<<blocks, 256>>>(iters, out); break;
            case 4: probe<4><<<blocks, 256>>>(iters, out); break;
            case 5: probe<5><<<blocks, 256>>>(iters, out); break;
            case 6: probe<6><<<blocks, 256>>>(iters, out); break;
            case 7: probe<7><<<blocks, 256>>>(iters, out); break;
            case 9: probe<9><<<blocks, 256>>>(iters, out); break;
            case 10: probe<10><<<blocks, 256>>>(iters, out); break;
            case 11: probe<11><<<blocks, 256>>>(iters, out); break;
            case 12: probe<12><<<blocks, 256>>>(iters, out); break;
            case 13: probe<13><<<blocks, 256>>>(iters, out); break;
            case 15: probe<15><<<blocks, 256>>>(iters, out); break;
            case 16: probe<16><<<blocks, 256>>>(iters, out); break;
            case 17: probe<17><<<blocks, 256>>>(iters, out); break;
            default: probe<8><<<blocks, 256>>>(iters, out); break;
        }
    };
    launch();
    hipDeviceSynchronize();
    const double flop_per_launch = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16 * (mode >= 9 && mode <= 11 ? 2 : 1);
    auto t0 = std::chrono::steady_clock::now();
    double t_half = 0; long n = 0, n_half = 0;
    for (;;) {
        for (int k = 0; k < 4; ++k) launch();
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        n += 4;
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (t < secs / 2) { t_half = t; n_half = n; }
        if (t >= secs) {
            printf("mode %d: %ld launches in %.2f s; steady second half: %.1f TFLOP/s (%.2f ms per launch)\n", mode, n, t,
                   flop_per_launch * (n - n_half) / (t - t_half) * 1e-12, (t - t_half) / (n - n_half) * 1e3);
            break;
        }
    }
    return 0;
}
