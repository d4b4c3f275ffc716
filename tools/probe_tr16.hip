// Probe: empirical lane mapping of ds_read_b64_tr_b16 on gfx950 for a row-major [row][16] fp16 LDS image
// (32 B rows = our chunk-plane pixel rows).  Used once to pin the wgrad fragment layout; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short short4_ __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    int l = threadIdx.x;
    int t = l & 15, g = l >> 4;
    int off = (g * 4 + (t >> 2)) * 16 + (t & 3) * 4;   // lane t of group g: row g*4 + t/4, 8-byte piece t%4
    short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_*)(lds + off));
    for (int j = 0; j < 4; j++) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    std::vector<unsigned short> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = (unsigned short)i;   // value = row*16 + col
    unsigned short *din, *dout;
    hipMalloc(&din, 8192); hipMalloc(&dout, 512);
    hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
    k<<<1, 64>>>(din, dout);
    std::vector<unsigned short> o(256);
    hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int l = 0; l < 64; l++) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; j++) {
            int v = o[l * 4 + j];
            printf(" (r%2d,c%2d)", v / 16, v % 16);
            if (v / 16 != (l >> 4) * 4 + j || v % 16 != (l & 15)) ok = false;
        }
        printf("\n");
    }
    printf("HYPOTHESIS lane l elem j = M[4*(l>>4)+j][l&15]: %s\n", ok ? "CONFIRMED" : "REFUTED");
    return 0;
}
