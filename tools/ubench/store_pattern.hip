// Per-CU drain rate of a streaming store burst by address pattern (all 256 CUs bursting at once, 8 waves per CU):
//   P0: every store instruction writes 1 KiB contiguous (whole 128-B lines)
//   P1: the GEMM epilogue's pattern: 3 row segments of 320 B per instruction (rows 640 B apart), the sibling wave
//       writes the other 320 B of the same rows (lines at the 320-B seam are written half by each wave)
//   P2: like P1 but rows 1280 B apart (N = 640: the other half of a row belongs to another CU)
// nt=1 streaming stores, nt=0 plain stores.  Reports bytes / cycle / CU from issue to vmcnt(0).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int PAT, int NT>
__global__ __launch_bounds__(512) void k(char* out, long long* res, int iters, int nst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 3, wn = wave >> 2;
    long long t_all = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        uint4 v = {1u, 2u, 3u, (unsigned)it};
        char* base = out + ((size_t)blockIdx.x * iters + it) * (size_t)(8 * nst * 1024) * (PAT == 2 ? 2 : 1);
        const long long t0 = __builtin_readcyclecounter();
        for (int s = 0; s < nst; ++s) {
            char* dst;
            if (PAT == 0) {
                dst = base + (size_t)(wave * nst + s) * 1024 + lane * 16;
            } else if (PAT == 3 || PAT == 4) {
                // P3: 16 rows x 64 B per instruction (4 lanes per row), the two halves of a 128-B line come from
                //     instructions s and s+1;  P4: 8 rows x 128 B per instruction (whole lines).  Rows 640 B apart.
                const int per = PAT == 3 ? 4 : 8;                        // lanes per row
                const int rows = 64 / per;
                const int lr = lane / per, lc = lane % per;
                const int seg = PAT == 3 ? (s % 5) : (s % 3);            // column segment inside the wave's 320 B
                const int blk = PAT == 3 ? (s / 5) : (s / 3);            // row block
                const int row = wm * 64 + blk * rows + lr;
                int off = seg * per * 16 + lc * 16;
                if (off >= 320) off = 304;                                // P4's third segment: clamp (duplicates)
                dst = base + (size_t)row * 640 + wn * 320 + off;
            } else {
                const int stride = PAT == 1 ? 640 : 1280;
                const int lrow = lane / 20 < 3 ? lane / 20 : 2, lch = lane - (lane / 20) * 20;
                const int row = wm * (nst * 3) + s * 3 + lrow;          // this wave's rows, 3 per instruction
                dst = base + (size_t)row * stride + wn * 320 + lch * 16;
            }
            uint4* d = (uint4*)dst;
            if (NT) { __builtin_nontemporal_store(v.x + s, &d->x); __builtin_nontemporal_store(v.y, &d->y);
                      __builtin_nontemporal_store(v.z, &d->z); __builtin_nontemporal_store(v.w, &d->w); }
            else *d = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_all += __builtin_readcyclecounter() - t0;
    }
    if (lane == 0) res[blockIdx.x * 8 + wave] = t_all / iters;
}

template <int PAT, int NT> void run(int blocks, int iters, int nst) {
    char* out; long long* res;
    size_t n = (size_t)blocks * iters * 8 * nst * 1024 * 2 + (1 << 20);
    hipMalloc(&out, n); hipMalloc(&res, blocks * 8 * 8);
    k<PAT, NT><<<blocks, 512>>>(out, res, iters, nst);
    hipDeviceSynchronize();
    long long* r = (long long*)malloc(blocks * 8 * 8);
    hipMemcpy(r, res, blocks * 8 * 8, hipMemcpyDeviceToHost);
    double a = 0;
    for (int i = 0; i < blocks * 8; ++i) a += r[i];
    a /= blocks * 8;
    const double bytes = 8.0 * nst * (PAT == 0 || PAT == 3 ? 1024 : (PAT == 4 ? 853 : 960));
    printf("pattern %d nt=%d blocks=%3d %3d KiB/CU: %8.0f cyc  -> %5.1f B/cyc/CU\n", PAT, NT, blocks, (int)(bytes / 1024), a, bytes / a);
    hipFree(out); hipFree(res);
}
int main() {
    for (int blocks : {1, 256}) {
        run<0, 1>(blocks, 20, 20); run<1, 1>(blocks, 20, 20); run<2, 1>(blocks, 20, 20);
        run<3, 1>(blocks, 20, 20); run<4, 1>(blocks, 20, 24);
        run<0, 0>(blocks, 20, 20); run<1, 0>(blocks, 20, 20); run<3, 0>(blocks, 20, 20);
    }
    return 0;
}
