#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, int off) {
    __shared__ __attribute__((aligned(16))) uint32_t win[2][256];
    const int lane = threadIdx.x & 63;
    // lane l: row l >> 2, quarter l & 3 of a 16 x 16-word window whose first word is src[off] with row pitch 100 words
    const uint32_t* p = src + off + (lane >> 2) * 100 + (lane & 3) * 4;
    if (lane != 5) __builtin_amdgcn_global_load_lds(p, &win[1][0], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) ... (encoding: all counters 0)
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = win[1][i];
}
int main() {
    uint32_t *s, *o; hipMalloc(&s, 100 * 32 * 4); hipMalloc(&o, 256 * 4);
    uint32_t h[3200]; for (int i = 0; i < 3200; i++) h[i] = i;
    hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemset(o, 0xff, 1024);
    for (int off = 0; off < 4; off++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, o, off + 3);
        uint32_t r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) { uint32_t want = off + 3 + y * 100 + x; bool skipped = (y * 4 + x / 4) == 5; if (!skipped && r[y * 16 + x] != want) bad++; }
        printf("off %d: %d bad, row1: %u %u %u %u | %u %u %u %u\n", off + 3, bad, r[16], r[17], r[18], r[19], r[20], r[21], r[22], r[23]);
    }
    return 0;
}
