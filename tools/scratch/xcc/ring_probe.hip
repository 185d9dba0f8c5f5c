// Scratch probe: the exchange skeleton of the one-utterance LSTM without its arithmetic. NW workgroups; per step every workgroup publishes its
// slice of a ROWB-byte row (sentinel-prefilled) and one wave of it polls the whole previous row until complete. Cycles per step for variants.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool unwritten(u32x4 v) { return v.x == 0xFFFFFFFFu || v.y == 0xFFFFFFFFu || v.z == 0xFFFFFFFFu || v.w == 0xFFFFFFFFu; }
template <int LAUX, int SAUX, int DEPTH, int NLD>
__global__ void ring_kernel(unsigned *buf, int steps, int nw, int stride8, long long *cycles, unsigned *fail) {
    // workers: blockIdx.x % stride8 == 0, first nw of them; row = NLD * 1024 bytes; worker w publishes pieces [w * P, (w + 1) * P) of 16 bytes, P = NLD * 64 / nw
    if (blockIdx.x % stride8) return;
    const int w = blockIdx.x / stride8;
    if (w >= nw) return;
    const int lane = threadIdx.x;
    const int rowb = NLD * 1024, P = NLD * 64 / nw;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, steps * rowb, 0x00020000);
    const long long t0 = clock64();
    for (int t = 0; t < steps; ++t) {
        if (t > 0) {
            u32x4 row[NLD];
            unsigned spins = 0;
            for (;;) {
                asm volatile("" ::: "memory");
                bool bad = false;
#pragma unroll
                for (int q = 0; q < NLD; ++q) { row[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (t - 1) * rowb + (64 * q + lane) * 16, 0, LAUX); }
#pragma unroll
                for (int q = 0; q < NLD; ++q) bad = bad || unwritten(row[q]);
                if (__builtin_amdgcn_ballot_w64(bad) == 0) break;
                if (++spins > (1u << 20)) { *fail = 1; return; }
            }
        }
        if (lane < P) {
            const u32x4 v = {(unsigned)t, (unsigned)w, (unsigned)lane, 7u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, t * rowb + (w * P + lane) * 16, 0, SAUX);
        }
    }
    if (w == 0 && lane == 0) *cycles = clock64() - t0;
}
template <int LAUX, int SAUX, int NLD>
static void run(const char *name, unsigned *buf, size_t bytes, int steps, int nw, int stride8, long long *cyc, unsigned *fail) {
    hipMemset(buf, 0xFF, bytes); hipMemset(fail, 0, 4); hipMemset(cyc, 0, 8);
    hipDeviceSynchronize();
    ring_kernel<LAUX, SAUX, 1, NLD><<<nw * stride8, 64>>>(buf, steps, nw, stride8, cyc, fail);
    hipDeviceSynchronize();
    unsigned f; long long cy; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost); hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s row %d B, %2d workgroups on %s: fail %u, %.0f cycles per step\n", name, NLD * 1024, nw, stride8 == 8 ? "one XCC " : "all XCCs", f, (double)cy / steps);
}
int main() {
    const int steps = 2000;
    const size_t bytes = (size_t)steps * 4096;
    unsigned *buf, *fail; long long *cyc;
    hipMalloc(&buf, bytes); hipMalloc(&fail, 4); hipMalloc(&cyc, 8);
    for (int stride8 : {8, 1})
        for (int nw : {2, 8, 16}) {
            run<16, 16, 1>("load sc1, store sc1", buf, bytes, steps, nw, stride8, cyc, fail);
            run<16, 0, 1>("load sc1, store plain", buf, bytes, steps, nw, stride8, cyc, fail);
            run<17, 17, 1>("load sc0 sc1, store sc0 sc1", buf, bytes, steps, nw, stride8, cyc, fail);
            run<16, 16, 4>("load sc1, store sc1", buf, bytes, steps, nw, stride8, cyc, fail);
        }
    return 0;
}
