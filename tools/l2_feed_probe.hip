// l2_feed_probe.hip — how many bytes per clock can ONE CU pull from L2, and does the path matter?
// Every block re-reads its own L2-resident window (default 1 MiB per block; stays in the 4 MiB L2 of its XCD only if few
// blocks share it, so the window is taken modulo a per-XCD budget) with
//   mode 0: LDS-DMA   buffer_load_dwordx4 ... lds, whole 128-byte lines (8 lanes per line), `depth` instructions in flight per wave
//   mode 1: VGPR loads buffer_load_dwordx4 into registers (discarded), same addresses, same depth
//   mode 2: LDS-DMA of 64-byte half lines (the 32-channel stages of the stride-2 kernels: 4 lanes per 64-byte row)
//   mode 3: LDS-DMA of 32-byte quarter lines (the 16-channel groups of conv_gather.hip / conv_halo_dma.hip: 2 lanes per row)
//   mode 4: LDS-DMA of whole lines that lie `stride` bytes apart (operand rows of a GEMM: 8 rows per instruction; all blocks of an
//           XCD read the SAME window, as the blocks of one weight tile do) — does the L2 serve some strides worse than others?
// and prints bytes / clock / CU at the measured kernel time (clock = s_memrealtime-free: elapsed ms x 2.1 GHz nominal AND
// wall-clock GB/s, so the figure can be re-based on the real clock).
// build: hipcc -O3 --offload-arch=gfx950 tools/l2_feed_probe.hip -o /tmp/l2p && /tmp/l2p [blocks_per_cu=1] [waves=8] [depth=8]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, long window, int iters, unsigned* sink, unsigned stride = 128) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000);
    // a wave instruction covers 1 KiB: 8 lines of 128 B (mode 0 / 1) or 16 half-lines of 64 B out of 16 different lines (mode 2)
    const unsigned lane_off = MODE == 2   ? (unsigned)(lane >> 2) * 128u + (unsigned)(lane & 3) * 16u
                              : MODE == 3 ? (unsigned)(lane >> 1) * 128u + (unsigned)(lane & 1) * 16u
                              : MODE == 4 ? (unsigned)(lane >> 3) * stride + (unsigned)(lane & 7) * 16u
                                          : (unsigned)lane * 16u;
    const unsigned step = MODE == 2 ? 2048u : MODE == 3 ? 4096u : MODE == 4 ? 8u * stride : 1024u;  // address space per instruction
    const unsigned base = MODE == 4 ? 0u : (unsigned)(((long)blockIdx.x * window) & 0x3fffffff);  // mode 4: one shared window
    unsigned acc = 0;
    unsigned off = (unsigned)wave * step;
    for (int it = 0; it < iters; ++it) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned o = base + (off % (unsigned)window);
            if (MODE == 1) v[d] = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, o, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)(smem + (wave * DEPTH + d) * 1024), 16, lane_off, o, 0, 0);
            off += (unsigned)nw * step;
        }
        if (MODE == 1) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) asm volatile("" ::"v"(v[d]));  // every load has to land
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* src, long window, int blocks, int waves, unsigned* sink, unsigned stride = 128) {
    const int iters = 2000 / DEPTH * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const size_t sm = (size_t)waves * DEPTH * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(blocks), dim3(waves * 64), sm, 0, src, window, 10, sink, stride);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(blocks), dim3(waves * 64), sm, 0, src, window, iters, sink, stride);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * waves * iters * DEPTH * 1024.0;
    int cus = 256;
    printf("%-34s depth %2d: %8.1f GB/s total, %6.1f B/clk/CU at 2.1 GHz (%d blocks x %d waves, %.3f ms)\n", name, DEPTH, bytes / ms / 1e6,
           bytes / (ms * 1e-3) / 2.1e9 / cus, blocks, waves, ms);
}

int main(int argc, char** argv) {
    const int bpc = argc > 1 ? atoi(argv[1]) : 1, waves = argc > 2 ? atoi(argv[2]) : 8;
    const long window = argc > 3 ? atol(argv[3]) : 96 * 1024;  // bytes per block: 256 blocks x 96 KiB = 24 MiB over 8 XCD L2s = 3 MiB each
    const int blocks = 256 * bpc;
    char* src;
    unsigned* sink;
    (void)hipMalloc(&src, (size_t)1 << 30);
    (void)hipMemset(src, 1, (size_t)1 << 30);
    (void)hipMalloc(&sink, 4);
    printf("window %ld KiB per block, %d blocks\n", window / 1024, blocks);
    run<0, 4>("LDS-DMA whole lines", src, window, blocks, waves, sink);
    run<0, 8>("LDS-DMA whole lines", src, window, blocks, waves, sink);
    run<0, 16>("LDS-DMA whole lines", src, window, blocks, waves, sink);
    run<2, 8>("LDS-DMA 64-byte half lines", src, window, blocks, waves, sink);
    run<2, 16>("LDS-DMA 64-byte half lines", src, window, blocks, waves, sink);
    run<3, 8>("LDS-DMA 32-byte quarter lines", src, window, blocks, waves, sink);
    run<3, 16>("LDS-DMA 32-byte quarter lines", src, window, blocks, waves, sink);
    {  // shared 2 MiB window read as rows `stride` apart: contiguous, pixel rows of a 512-channel tensor, weight rows [N][9][512]
        const unsigned strides[] = {128, 1024, 1024 + 128, 9216, 9216 + 128, 4608, 4608 + 128, 2048, 4096};
        for (unsigned st : strides) {
            char nm[64];
            snprintf(nm, sizeof nm, "LDS-DMA lines %u B apart (shared)", st);
            run<4, 8>(nm, src, 2 << 20, blocks, waves, sink, st);
        }
    }
    run<1, 4>("VGPR loads whole lines", src, window, blocks, waves, sink);
    run<1, 8>("VGPR loads whole lines", src, window, blocks, waves, sink);
    run<1, 16>("VGPR loads whole lines", src, window, blocks, waves, sink);
    return 0;
}
