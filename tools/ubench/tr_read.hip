// Semantics of ds_read_b64_tr_b16 on gfx950: every lane supplies the LDS address of 4 contiguous 16-bit elements; what does
// each lane get back?  LDS holds its own element index; lane l points at elements [4 (l & 15) + 64 (l >> 4), +4) — i.e. the
// 16 lanes of a group cover one [4 rows][16 cols] row-major block — and the four returned values are printed per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds + (l & 15) * 4 + (l >> 4) * 64));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d   expected (block[j][l&15] of group l>>4): %3d %3d %3d %3d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3],
        (l>>4)*64 + (l&15), (l>>4)*64 + 16 + (l&15), (l>>4)*64 + 32 + (l&15), (l>>4)*64 + 48 + (l&15));
    return 0;
}
