// k_occ.hip -- wire format: ROS nav_msgs/OccupancyGrid cells -> the loader's map values, on the device (gfx950).
//
// Replaces the per-cell loop of mapCallback (LSD/main_on_linux.cpp:108-124): the int8 cell read as uint8 becomes
//   255 (-1, unknown) -> 0,   0 (free) -> 255,   anything else (1..100, occupied) -> 1,
// which is the value set {0, 1, 255} createMapCache and myLineSegmentDetector expect (SURVEY 8f "next" #3).
// Pure byte work: 16 cells per lane per access, the three cases resolved with byte-parallel integer arithmetic.
// Bound: HBM, 2 bytes per cell (1 read + 1 written).
#include "lsd_internal.h"

namespace lsdhip {

__device__ __forceinline__ uint32_t occ4(uint32_t v) {
    // 0x80 in every byte of x that is zero (exact, no carries across bytes)
    auto zero_bytes = [](uint32_t x) { return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu); };
    const uint32_t free_ = (zero_bytes(v) >> 7) * 255u;          // bytes == 0   -> 0xff
    const uint32_t unk = (zero_bytes(~v) >> 7) * 255u;           // bytes == 255 -> 0xff
    return free_ | (~(free_ | unk) & 0x01010101u);               // free -> 255, unknown -> 0, the rest -> 1
}

__global__ __launch_bounds__(256) void k_occ_to_map(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
    const size_t nv = n / 16;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4* in4 = reinterpret_cast<const uint4*>(in);
    uint4* out4 = reinterpret_cast<uint4*>(out);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        uint4 v = in4[i];
        v.x = occ4(v.x); v.y = occ4(v.y); v.z = occ4(v.z); v.w = occ4(v.w);
        out4[i] = v;
    }
    for (size_t i = nv * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {   // tail
        const uint8_t v = in[i];
        out[i] = v == 255 ? 0 : (v == 0 ? 255 : 1);
    }
}

void launch_occ_to_map(const uint8_t* in, uint8_t* out, size_t n, hipStream_t s) {
    size_t blocks = (n / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 32) blocks = 256 * 32;                    // >> 256 CUs; grid-stride beyond
    hipLaunchKernelGGL(k_occ_to_map, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n);
}

}  // namespace lsdhip
