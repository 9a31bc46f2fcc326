// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds lds[i] = i (16-bit); each pattern gives
// per-lane element offsets; prints the 4 shorts every lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* pat, short* out) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    int a = pat[threadIdx.x];
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    int *dp; short* dout;
    hipMalloc(&dp, 64 * 4); hipMalloc(&dout, 64 * 4 * 2);
    const char* names[] = {"A: all 0", "B: l*4", "C: (l&15)*64 + (l>>4)*4", "D: (l&15)*64 + (l>>4)*1024"};
    for (int pt = 0; pt < 4; ++pt) {
        std::vector<int> pat(64);
        for (int l = 0; l < 64; ++l) {
            if (pt == 0) pat[l] = 0;
            if (pt == 1) pat[l] = l * 4;
            if (pt == 2) pat[l] = (l & 15) * 64 + (l >> 4) * 4;
            if (pt == 3) pat[l] = (l & 15) * 64 + (l >> 4) * 1024;
        }
        hipMemcpy(dp, pat.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dp, dout);
        std::vector<short> out(256);
        hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("pattern %s\n", names[pt]);
        for (int l = 0; l < 64; ++l) {
            printf("  l%2d a=%5d -> %5d %5d %5d %5d", l, pat[l], out[l*4], out[l*4+1], out[l*4+2], out[l*4+3]);
            if (l % 2 == 1) printf("\n");
        }
    }
    return 0;
}
