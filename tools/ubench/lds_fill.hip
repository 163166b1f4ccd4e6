// Per-CU fill rate of the LDS through LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), all 256 CUs at once,
// 8 waves per CU (one 512-thread block per CU), counted vmcnt keeping DEPTH instructions in flight per wave.
// Question: is the operand stream of the GEMM kernels (52 KiB per K tile at 256x160, 72 KiB at 256x320) bound by a
// per-CU fill rate?  Patterns (what one DMA instruction reads):
//   P0: 1 KiB contiguous
//   P1: 8 rows x 128 B, rows 640 B apart (an A tile at C = 320), 16-byte chunks XOR-swizzled within a row like gemm.hip
//   P2: the same with rows 2560 B apart (C = 1280)
//   P3: P1 without the swizzle;  P4: contiguous rows WITH the swizzle;  P5: 4 rows x 256 B (rows 640 B apart), in order
//   P6 / P7: P1 with a swizzle that keeps 32-byte pairs / 64-byte halves of a row in lane order
// Working sets:  W0: every block streams the SAME 2 MiB (L2-resident, the W-panel case)
//                W1: every XCD's 32 blocks stream the same 16 MiB, disjoint between XCDs (A panels shared inside an XCD: MALL / HBM + L2)
//                W2: every block streams its own 4 MiB (no sharing at all: HBM-bound)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int PAT, int DEPTH>
__global__ __launch_bounds__(512) void fill(const char* src, size_t block_stride, size_t xcd_stride, size_t region, int iters,
                                            long long* res) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 128 KiB ring: 16 slots of 1 KiB per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const char* base = src + (size_t)(blockIdx.x & 7) * xcd_stride + (size_t)(blockIdx.x >> 3) * block_stride;
    const int rsub = lane >> 3, pos = lane & 7;
    const long long t0 = __builtin_readcyclecounter();
    size_t off = (size_t)wave * 1024 * 64;     // waves start 64 KiB apart
    for (int it = 0; it < iters; ++it) {
        const char* g;
        if (PAT == 0) g = base + off + lane * 16;
        else if (PAT == 4) g = base + off + rsub * 128 + ((pos ^ rsub) << 4);             // contiguous rows, swizzled chunks
        else if (PAT == 5) g = base + (off / 1024) * 4 * 640 % region + (size_t)(lane >> 4) * 640 + ((lane & 15) << 4);   // 4 rows x 256 B
        else {
            const size_t stride = PAT == 2 ? 2560 : 640;
            int ch = pos ^ rsub;                                                           // P1 / P2: gemm.hip's swizzle
            if (PAT == 3) ch = pos;                                                        // strided rows, chunks in order
            if (PAT == 6) ch = pos ^ ((rsub & 3) << 1);                                    // 32-byte pairs stay in order
            if (PAT == 7) ch = pos ^ ((rsub & 1) << 2);                                    // 64-byte halves stay in order
            g = base + (off / 1024) * 8 * stride % region + (size_t)rsub * stride + (ch << 4);
        }
        char* l = smem + wave * 16384 + (it & 15) * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
        off += 8 * 1024 * 64;                   // the 8 waves interleave 64-KiB strides
        if (off + 1024 > region) off -= (off / (8 * 1024 * 64)) * (8 * 1024 * 64);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) res[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PAT, int DEPTH>
void run(const char* name, const char* buf, size_t block_stride, size_t xcd_stride, size_t region, int blocks) {
    long long* res;
    hipMalloc(&res, blocks * 8 * sizeof(long long));
    const int iters = 4096;
    hipFuncSetAttribute((const void*)fill<PAT, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    fill<PAT, DEPTH><<<blocks, 512, 131072>>>(buf, block_stride, xcd_stride, region, 64, res);   // warm
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    fill<PAT, DEPTH><<<blocks, 512, 131072>>>(buf, block_stride, xcd_stride, region, iters, res);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long* r = (long long*)malloc(blocks * 8 * sizeof(long long));
    hipMemcpy(r, res, blocks * 8 * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < blocks * 8; ++i) cyc += (double)r[i];
    cyc /= blocks * 8;
    const double bytes_cu = 8.0 * iters * 1024;
    printf("%-34s pat %d depth %2d blocks %3d: %6.1f B/clk/CU (wave clock), %6.2f TB/s chip-wide by wall time\n", name, PAT, DEPTH,
           blocks, bytes_cu / cyc, bytes_cu * blocks / (ms * 1e-3) / 1e12);
    free(r); hipFree(res);
}

int main() {
    char* buf;
    const size_t total = (size_t)1536 << 20;
    hipMalloc(&buf, total);
    hipMemset(buf, 1, total);
    for (int blocks : {256}) {
        run<0, 15>("W0 shared 2 MiB (L2-resident)", buf, 0, 0, 2 << 20, blocks);
        run<1, 15>("W0 shared 2 MiB (L2-resident)", buf, 0, 0, 2 << 20, blocks);
        run<2, 15>("W0 shared 2 MiB (L2-resident)", buf, 0, 0, 2 << 20, blocks);
        run<3, 15>("W0 shared 2 MiB (L2-resident)", buf, 0, 0, 2 << 20, blocks);
        run<4, 15>("W0 shared 2 MiB (L2-resident)", buf, 0, 0, 2 << 20, blocks);
        run<5, 15>("W0 shared 2 MiB (L2-resident)", buf, 0, 0, 2 << 20, blocks);
        run<6, 15>("W0 shared 2 MiB (L2-resident)", buf, 0, 0, 2 << 20, blocks);
        run<7, 15>("W0 shared 2 MiB (L2-resident)", buf, 0, 0, 2 << 20, blocks);
        run<0, 15>("W1 16 MiB per XCD, shared in XCD", buf, 0, (size_t)16 << 20, 16 << 20, blocks);
        run<1, 15>("W1 16 MiB per XCD, shared in XCD", buf, 0, (size_t)16 << 20, 16 << 20, blocks);
        run<3, 15>("W1 16 MiB per XCD, shared in XCD", buf, 0, (size_t)16 << 20, 16 << 20, blocks);
    }
    return 0;
}
