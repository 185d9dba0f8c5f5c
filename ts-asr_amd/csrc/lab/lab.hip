// libtsasr_lab.so: lab equipment (include/tsasr_lab.h) - LDS / memory fills and a wall-clock stamp. Not linked into the product library.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

#include "../../../include/tsasr_lab.h"

__global__ __launch_bounds__(256) void lab_fill_lds_kernel(unsigned pattern, int words, unsigned *sink) {
    extern __shared__ unsigned fill_lds[];
    for (int i = threadIdx.x; i < words; i += 256) fill_lds[i] = pattern;
    __syncthreads();
    if (sink && fill_lds[(threadIdx.x * 97) % words] != pattern) *sink = 1;   // keeps the stores alive
}

__global__ void lab_fill_words_kernel(unsigned *p, unsigned pattern, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = pattern;
}

__global__ void lab_stamp_kernel(unsigned long long *out) { *out = __builtin_amdgcn_s_memrealtime(); }

static int launched(void) { return hipGetLastError() == hipSuccess ? 0 : -2; }

extern "C" {

int tsasr_lab_fill_lds(unsigned pattern, void *stream) {
    const int bytes = 160 * 1024;
    (void)hipFuncSetAttribute((const void *)lab_fill_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    lab_fill_lds_kernel<<<4 * cus, 256, bytes, (hipStream_t)stream>>>(pattern, bytes / 4, nullptr);
    return launched();
}

int tsasr_lab_fill(void *p, unsigned pattern, size_t nwords, void *stream) {
    if (!p || ((uintptr_t)p & 3)) return -1;
    if (nwords == 0) return 0;
    lab_fill_words_kernel<<<(unsigned)std::min<size_t>(4096, (nwords + 255) / 256), 256, 0, (hipStream_t)stream>>>((unsigned *)p, pattern, nwords);
    return launched();
}

int tsasr_lab_stamp(void *out, void *stream) {
    if (!out || ((uintptr_t)out & 7)) return -1;
    lab_stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>((unsigned long long *)out);
    return launched();
}

}  // extern "C"
