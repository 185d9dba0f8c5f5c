#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../ts-asr_amd/csrc/common.h"
__global__ void k(const float *in, float *out) {
    float v[4];
    for (int q = 0; q < 4; ++q) v[q] = in[q * 64 + threadIdx.x];
    wave_sum4(v);
    for (int q = 0; q < 4; ++q) out[q * 64 + threadIdx.x] = v[q];
}
int main() {
    float h[256], o[256], *d, *e;
    for (int i = 0; i < 256; ++i) h[i] = (float)((i * 37) % 101) + (i / 64) * 1000.f;
    hipMalloc(&d, 1024); hipMalloc(&e, 1024);
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, e);
    hipMemcpy(o, e, 1024, hipMemcpyDeviceToHost);
    for (int q = 0; q < 4; ++q) {
        float s = 0; for (int i = 0; i < 64; ++i) s += h[q * 64 + i];
        printf("value %d: expect %.0f, lanes 0,1,2,3,17,34,63 -> %.0f %.0f %.0f %.0f %.0f %.0f %.0f\n", q, s, o[q*64], o[q*64+1], o[q*64+2], o[q*64+3], o[q*64+17], o[q*64+34], o[q*64+63]);
    }
    return 0;
}
