// Round 5: does LDS-DMA (global_load_lds_dwordx4) depend on WHERE the workgroup's LDS allocation sits?
// Found while hunting a non-repeatable result in csrc/attention32.hip: with 48 KiB of LDS per 256-thread block (3 blocks per
// CU) the kernel's results differed from launch to launch by a few ulps; with the allocation padded to 56 or 64 KiB (2 blocks
// per CU) they are bit-identical.  This probe: every block fills its whole allocation by LDS-DMA from a per-block pattern in
// global memory (1 KiB per wave-instruction, like the attention ring), waits vmcnt(0) + barrier, reads everything back with
// ds_read_b128 and counts the 16-byte chunks that do not hold the pattern; HW_REG_LDS_ALLOC gives the block's LDS base.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_base.hip -o /tmp/ldsbase && /tmp/ldsbase
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(256) void k(const uint4* __restrict__ pat, int kib, int rounds, unsigned* __restrict__ bad, unsigned* __restrict__ base_of) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned alloc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(alloc));
    unsigned nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        const uint4* src = pat + ((size_t)(blockIdx.x * 7 + r) % 64) * (size_t)(kib * 64);   // kib KiB = kib * 64 chunks of 16 B
        for (int piece = wave; piece < kib; piece += 4)   // wave w moves the 1-KiB pieces w, w + 4, ...
            __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 64 + lane), (lptr_t)(smem + piece * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = tid; c < kib * 64; c += 256) {
            const uint4 v = *(const uint4*)(smem + c * 16), e = src[c];
            nbad += (v.x != e.x) | (v.y != e.y) | (v.z != e.z) | (v.w != e.w);
        }
        __syncthreads();
    }
    atomicAdd(&bad[blockIdx.x], nbad);
    if (tid == 0) base_of[blockIdx.x] = alloc;
}

int main() {
    const int nblk = 2048, rounds = 8;
    const int sizes[] = {32, 40, 48, 52, 56, 64, 80};
    uint4* pat;
    const size_t pat_chunks = (size_t)64 * 80 * 64;
    hipMalloc(&pat, pat_chunks * 16);
    uint4* h = (uint4*)malloc(pat_chunks * 16);
    for (size_t i = 0; i < pat_chunks; ++i) h[i] = uint4{(unsigned)i * 2654435761u, (unsigned)i ^ 0x5bd1e995u, (unsigned)(i >> 3), (unsigned)i + 17u};
    hipMemcpy(pat, h, pat_chunks * 16, hipMemcpyHostToDevice);
    unsigned *bad, *base_of;
    hipMalloc(&bad, nblk * 4);
    hipMalloc(&base_of, nblk * 4);
    unsigned hb[2048], ha[2048];
    for (int s : sizes) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, s * 1024);
        hipMemset(bad, 0, nblk * 4);
        hipLaunchKernelGGL(k, dim3(nblk), dim3(256), s * 1024, 0, pat, s, rounds, bad, base_of);
        hipDeviceSynchronize();
        hipMemcpy(hb, bad, nblk * 4, hipMemcpyDeviceToHost);
        hipMemcpy(ha, base_of, nblk * 4, hipMemcpyDeviceToHost);
        // histogram of LDS bases (HW_REG_LDS_ALLOC: LDS_BASE = bits 7:0 ... printed raw) and bad chunks per base
        unsigned keys[64], cnt[64], badc[64];
        int nk = 0;
        unsigned long long total = 0;
        for (int b = 0; b < nblk; ++b) {
            total += hb[b];
            int i = 0;
            while (i < nk && keys[i] != ha[b]) ++i;
            if (i == nk && nk < 64) { keys[nk] = ha[b]; cnt[nk] = 0; badc[nk] = 0; ++nk; }
            if (i < 64) { cnt[i]++; badc[i] += hb[b]; }
        }
        printf("LDS %2d KiB per block: %llu bad 16-B chunks of %llu |", s, total, (unsigned long long)nblk * rounds * s * 64);
        for (int i = 0; i < nk; ++i) printf(" alloc=0x%08x blocks=%u bad=%u |", keys[i], cnt[i], badc[i]);
        printf("\n");
    }
    return 0;
}
