// Probe: what the LDS-DMA path (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) delivers chip-wide, as a function
// of the source footprint (L2 / Infinity-Cache resident vs HBM streaming), the pieces a wave keeps in flight and the
// workgroups per CU.  The byte budgets of DESIGN.md's kernels are priced against these numbers (profiles/r02_wgrad_study.md).
// usage: probe_ldsdma   (prints one line per configuration)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void lds_void_t;

template <int INFLIGHT>
__global__ void __launch_bounds__(256) stream(const char* src, unsigned long long footprint, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // every wave streams its own contiguous pieces; workgroups are spread over the footprint
    unsigned long long per_wg = footprint / gridDim.x;
    unsigned long long region = blockIdx.x;
    if (per_wg < (128ull << 10)) {                      // small footprints: workgroups share 128 KiB regions
        per_wg = 128ull << 10;
        region = blockIdx.x % (footprint / per_wg);
    }
    const char* base = src + region * per_wg;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)(per_wg > 0xfffffff0ull ? 0xfffffff0u : per_wg), 0x00020000);
    const unsigned span = (unsigned)per_wg;
    unsigned off = (unsigned)(wave * INFLIGHT * 1024 + lane * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + (wave * INFLIGHT + k) * 1024), 16, off + k * 1024, 0, 0, 0);
        }
        off += 4 * INFLIGHT * 1024;
        if (off + INFLIGHT * 1024 + 1024 > span) off = (unsigned)(wave * INFLIGHT * 1024 + lane * 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0 && iters < 0) out[blockIdx.x] = ((float*)smem)[0];
}

template <int INFLIGHT>
static double run(const char* src, unsigned long long footprint, int blocks, double secs) {
    const int iters = 2000;
    const size_t lds = 4 * INFLIGHT * 1024;
    (void)hipFuncSetAttribute((const void*)&stream<INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float* out;
    (void)hipMalloc(&out, blocks * sizeof(float));
    stream<INFLIGHT><<<blocks, 256, lds>>>(src, footprint, 10, out);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double t = 0;
    do {
        stream<INFLIGHT><<<blocks, 256, lds>>>(src, footprint, iters, out);
        (void)hipDeviceSynchronize();
        ++n;
        t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (t < secs);
    (void)hipFree(out);
    return (double)n * blocks * 4.0 * INFLIGHT * 1024.0 * iters / t * 1e-12;      // TB/s
}

int main() {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const unsigned long long big = 4ull << 30;
    char* src;
    if (hipMalloc(&src, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 1, big);
    const unsigned long long fps[3] = {16ull << 20, 192ull << 20, 4ull << 30};
    const char* names[3] = {"16 MiB (L2)", "192 MiB (Infinity Cache)", "4 GiB (HBM)"};
    for (int f = 0; f < 3; ++f)
        for (int wgs = 1; wgs <= 2; ++wgs) {
            const int blocks = cus * wgs;
            printf("%-26s %d wg/CU x 4 waves: in flight per wave 4 KiB %.2f TB/s, 8 KiB %.2f, 16 KiB %.2f\n", names[f], wgs,
                   run<4>(src, fps[f], blocks, 0.5), run<8>(src, fps[f], blocks, 0.5), run<16>(src, fps[f], blocks, 0.5));
            fflush(stdout);
        }
    return 0;
}
