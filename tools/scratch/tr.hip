#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* out) {
  __shared__ __bf16 lds[64*64];
  for (int i = threadIdx.x; i < 64*64; i += 64) lds[i] = (__bf16)in[i];
  __syncthreads();
  const int l = threadIdx.x;
  // each 16-lane group g reads the 4x16 block at rows 4g..4g+3 (row stride 64), cols 0..15
  const int g = l / 16, q = (l % 16) / 4, p = l % 4;
  auto v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds + (4*g + q)*64 + 4*p));
  for (int j=0;j<4;++j) out[l*4+j] = (float)v[j];
}
int main() {
  float *in, *out; hipMallocManaged(&in, 4096*4); hipMallocManaged(&out, 256*4);
  for (int i=0;i<4096;++i) in[i] = (float)((i/64)*100 + (i%64));  // value = row*100+col (exact in bf16? up to 6363 -> not exact; use small)
  for (int i=0;i<4096;++i) in[i] = (float)((i/64)*16 + (i%64)%16);  // row*16+col for col<16: max 63*16+15=1023 -> needs 10 bits; bf16 has 8 -> use rows<16 only
  hipLaunchKernelGGL(k, 1, 64, 0, 0, in, out); hipDeviceSynchronize();
  for (int l=0;l<64;++l) { printf("lane %2d:", l); for (int j=0;j<4;++j) { int v=(int)out[l*4+j]; printf(" (r%d,c%d)", v/16, v%16);} printf("\n"); }
  return 0;
}
