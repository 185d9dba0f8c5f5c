// Scratch probe: (1) which XCC a workgroup of a 1-D grid lands on; (2) latency of a one-way flag hand-off between two workgroups on the SAME
// XCC with L2-scope (sc0) accesses vs agent-scope (sc1) accesses, ping-pong of N rounds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void xcc_kernel(unsigned *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xF;
}
template <int AUX>
__global__ void pingpong_kernel(unsigned *buf, int a, int b, int rounds, long long *cycles, unsigned *fail) {
    // workgroups a and b of the grid play; everybody else leaves. buf[0]: a -> b, buf[64]: b -> a (separate lines)
    const int me = blockIdx.x == a ? 0 : blockIdx.x == b ? 1 : -1;
    if (me < 0 || threadIdx.x != 0) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1024, 0x00020000);
    const long long t0 = clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)r, rs, 0, 0, AUX);
        unsigned spins = 0;
        for (;;) {
            asm volatile("" ::: "memory");
            const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, me == 0 ? 256 : 0, 0, AUX);
            if (v == (unsigned)r) break;
            if (++spins > (1u << 22)) { *fail = 1; return; }
        }
        if (me == 1) __builtin_amdgcn_raw_buffer_store_b32((unsigned)r, rs, 256, 0, AUX);
    }
    if (me == 0) *cycles = clock64() - t0;
}
int main() {
    const int G = 128;
    unsigned *d; hipMalloc(&d, G * 4);
    xcc_kernel<<<G, 64>>>(d);
    std::vector<unsigned> h(G); hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost);
    printf("xcc of workgroups 0..%d:", G - 1); for (int i = 0; i < G; ++i) printf(" %u", h[i]); printf("\n");
    int a = 0, b = -1, c = -1;
    for (int i = 1; i < G; ++i) { if (b < 0 && h[i] == h[a]) b = i; if (c < 0 && h[i] != h[a]) c = i; }
    unsigned *buf, *fail; long long *cyc;
    hipMalloc(&buf, 1024); hipMalloc(&fail, 4); hipMalloc(&cyc, 8);
    const int rounds = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        const int peer = (mode & 1) ? c : b;
        hipMemset(buf, 0, 1024); hipMemset(fail, 0, 4); hipMemset(cyc, 0, 8);
        if (mode < 2) pingpong_kernel<1><<<G, 64>>>(buf, a, peer, rounds, cyc, fail);
        else pingpong_kernel<16><<<G, 64>>>(buf, a, peer, rounds, cyc, fail);
        hipDeviceSynchronize();
        unsigned f; long long cy; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost); hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost);
        printf("%s, peer on %s XCC (wg %d <-> %d): fail %u, %.0f cycles per round trip (two hand-offs)\n", mode < 2 ? "sc0" : "sc1", (mode & 1) ? "another" : "the same", a, peer, f,
               (double)cy / rounds);
    }
    return 0;
}
