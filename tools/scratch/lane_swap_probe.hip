#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../ts-asr_amd/csrc/common.h"
__global__ void k(float *out) {
    const int lane = threadIdx.x;
    float c = (float)(1 << (lane & 15)) ;   // bit per lane in row
    float a = c + dpp_mov<0x124>(c);
    float b = a + dpp_mov<0x128>(a);
    out[lane] = a; out[64 + lane] = b;
    float rowv = (float)(lane >> 4) + 1.f;    // 1,2,3,4 per row
    unsigned u = __builtin_bit_cast(unsigned, rowv);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    out[128 + lane] = __builtin_bit_cast(float, r[0]); out[192 + lane] = __builtin_bit_cast(float, r[1]);
    auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    out[256 + lane] = __builtin_bit_cast(float, r2[0]); out[320 + lane] = __builtin_bit_cast(float, r2[1]);
}
int main() {
    float o[384], *e;
    hipMalloc(&e, 384 * 4);
    k<<<1, 64>>>(e);
    hipMemcpy(o, e, 384 * 4, hipMemcpyDeviceToHost);
    printf("after ror4 (lane: sum of bits): "); for (int i = 0; i < 16; ++i) printf("%d:%04x ", i, (int)o[i]); printf("\n");
    printf("after ror8: "); for (int i = 0; i < 16; ++i) printf("%d:%04x ", i, (int)o[64 + i]); printf("\n");
    printf("p16 r0 rows: %g %g %g %g   r1 rows: %g %g %g %g\n", o[128], o[128+16], o[128+32], o[128+48], o[192], o[192+16], o[192+32], o[192+48]);
    printf("p32 r0 rows: %g %g %g %g   r1 rows: %g %g %g %g\n", o[256], o[256+16], o[256+32], o[256+48], o[320], o[320+16], o[320+32], o[320+48]);
    return 0;
}
