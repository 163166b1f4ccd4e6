// Does an LDS-DMA load issued AFTER a burst of global stores land in LDS before those stores are acknowledged?
// (vmcnt retires in order, so s_waitcnt cannot tell; the landing is observed by polling the LDS destination.)
// Every block: 8 waves; each wave issues NST 1-KiB streaming stores to distinct lines, then one global_load_lds of a
// 1-KiB "flag" chunk, then polls LDS for the flag, then waits vmcnt(0).  Prints cycles: issue->flag, issue->vmcnt(0).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NST>
__global__ __launch_bounds__(512) void k(uint4* out, const uint32_t* flagsrc, long long* res, int iters) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[8 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long t_flag = 0, t_all = 0;
    for (int it = 0; it < iters; ++it) {
        lds[wave * 256 + lane * 4] = 0;
        __syncthreads();
        uint4 v = {1u, 2u, 3u, (unsigned)it};
        uint4* dst = out + ((size_t)(blockIdx.x * 8 + wave) * iters + it) * NST * 64 + lane;
        const long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int s = 0; s < NST; ++s) __builtin_nontemporal_store(v.x + s, &dst[s * 64].x), __builtin_nontemporal_store(v.y, &dst[s * 64].y),
            __builtin_nontemporal_store(v.z, &dst[s * 64].z), __builtin_nontemporal_store(v.w, &dst[s * 64].w);
        __builtin_amdgcn_global_load_lds((gptr_t)(flagsrc + (it + 1) * 256 + lane * 4), (lptr_t)(lds + wave * 256), 16, 0, 0);
        volatile uint32_t* f = lds + wave * 256 + lane * 4;
        int spin = 0;
        while (*f != (uint32_t)(it + 1) && spin < (1 << 22)) ++spin;
        const long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_readcyclecounter();
        t_flag += t1 - t0; t_all += t2 - t0;
        __syncthreads();
    }
    if (lane == 0) { res[(blockIdx.x * 8 + wave) * 2] = t_flag / iters; res[(blockIdx.x * 8 + wave) * 2 + 1] = t_all / iters; }
}

template <int NST> void run(int blocks, int iters) {
    uint4* out; uint32_t* fs; long long* res;
    size_t n = (size_t)blocks * 8 * iters * NST * 64;
    hipMalloc(&out, n * 16 + 4096); hipMalloc(&fs, (iters + 2) * 1024); hipMalloc(&res, blocks * 16 * 8);
    uint32_t* h = (uint32_t*)malloc((iters + 2) * 1024);
    for (int i = 0; i < (iters + 2) * 256; ++i) h[i] = i / 256;
    hipMemcpy(fs, h, (iters + 2) * 1024, hipMemcpyHostToDevice);
    k<NST><<<blocks, 512>>>(out, fs, res, iters);
    hipDeviceSynchronize();
    long long* r = (long long*)malloc(blocks * 16 * 8);
    hipMemcpy(r, res, blocks * 16 * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < blocks * 8; ++i) { a += r[2 * i]; b += r[2 * i + 1]; }
    printf("NST=%2d blocks=%3d: issue->flag landed %8.0f cyc   issue->vmcnt(0) %8.0f cyc  (stores %d KiB per CU)\n", NST, blocks,
           a / (blocks * 8), b / (blocks * 8), NST * 8);
    hipFree(out); hipFree(fs); hipFree(res);
}
int main() {
    for (int blocks : {1, 256}) { run<1>(blocks, 20); run<6>(blocks, 20); run<12>(blocks, 20); run<24>(blocks, 20); }
    return 0;
}
