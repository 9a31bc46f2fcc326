// line_granularity_probe.hip — does the SHAPE of a wave's memory instruction matter at equal bytes?
// The low-channel 256 px conv layers (C, N <= 64) stage their input as 32-byte pieces of 128-byte pixel rows (a
// 16-channel chunk of a 64-channel NHWC pixel) and store 32-byte pieces of 128-byte output rows; every byte is
// eventually used, but one wave instruction touches 32 different cache lines.  This probe streams the same tensor with
//   mode 0: LDS-DMA loads, 32 rows x 32 B per instruction (4 passes over the rows, chunk by chunk)   [the conv's pattern]
//   mode 1: LDS-DMA loads,  8 rows x 128 B per instruction (full lines)
//   mode 2: stores, 32 rows x 32 B per instruction (4 instructions complete a row)                   [the conv's pattern]
//   mode 3: stores,  8 rows x 128 B per instruction
// build: hipcc -O3 --offload-arch=gfx950 tools/line_granularity_probe.hip -o /tmp/lgp && /tmp/lgp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(1))) const void* gl_void_ptr;

// rows of 128 bytes; each block handles ROWS_PER_BLOCK rows per outer iteration, grid-stride
template <int MODE>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, char* __restrict__ dst, long rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows_per_iter = 512;  // per block
    for (long r0 = (long)blockIdx.x * rows_per_iter; r0 < rows; r0 += (long)gridDim.x * rows_per_iter) {
        if (MODE == 0) {
            // 4 chunks x 16 pieces of 32 rows; wave w takes pieces w, w+4, ...
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int piece = wave + 4 * it;
                    const long row = r0 + piece * 32 + (lane >> 1);
                    const char* g = src + row * 128 + ch * 32 + (lane & 1) * 16;
                    __builtin_amdgcn_global_load_lds((gl_void_ptr)g, (lds_void_ptr)(smem + (ch * 16 + piece) * 1024), 16, 0, 0);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else if (MODE == 1) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int piece = wave + 4 * it;  // 64 pieces of 8 rows
                const long row = r0 + piece * 8 + (lane >> 3);
                const char* g = src + row * 128 + (lane & 7) * 16;
                __builtin_amdgcn_global_load_lds((gl_void_ptr)g, (lds_void_ptr)(smem + piece * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else if (MODE == 2) {
            const uint4 v = make_uint4(lane, wave, (unsigned)r0, 7u);
#pragma unroll
            for (int it = 0; it < 4; ++it) {  // wave: 128 rows = 4 groups of 32 rows
                const long row = r0 + wave * 128 + it * 32 + (lane & 31);
#pragma unroll
                for (int q = 0; q < 4; ++q)  // 4 instructions complete the 128-byte rows
                    *reinterpret_cast<uint4*>(dst + row * 128 + q * 32 + (lane >> 5) * 16) = v;
            }
        } else {
            const uint4 v = make_uint4(lane, wave, (unsigned)r0, 7u);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const long row = r0 + wave * 128 + it * 8 + (lane >> 3);
                *reinterpret_cast<uint4*>(dst + row * 128 + (lane & 7) * 16) = v;
            }
        }
    }
}

template <int MODE>
float run(const char* src, char* dst, long rows, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 65536, 0, src, dst, rows);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 65536, 0, src, dst, rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const long rows = 8l << 20;  // 1 GiB of 128-byte rows (beyond the 256 MiB Infinity Cache)
    char *src, *dst;
    hipMalloc(&src, rows * 128);
    hipMalloc(&dst, rows * 128);
    hipMemset(src, 1, rows * 128);
    hipMemset(dst, 0, rows * 128);
    const double gb = rows * 128 / 1e9;
    for (int blocks : {512, 1024, 2048}) {
        printf("blocks %4d | DMA load 32x32B %.0f GB/s | DMA load 8x128B %.0f GB/s | store 32x32B %.0f GB/s | store 8x128B %.0f GB/s\n", blocks,
               gb / run<0>(src, dst, rows, blocks) * 1e3, gb / run<1>(src, dst, rows, blocks) * 1e3,
               gb / run<2>(src, dst, rows, blocks) * 1e3, gb / run<3>(src, dst, rows, blocks) * 1e3);
    }
    return 0;
}
