// Does a 16-byte global load at an address that is only 8-byte aligned return the 16 bytes AT that address on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 -o unaligned_x4 unaligned_x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t* __restrict__ p, u4* out) {
    const int l = threadIdx.x;
    out[l] = *reinterpret_cast<const u4*>(reinterpret_cast<const char*>(p) + (size_t)(l * 8u));      // lane l: bytes 8 l .. 8 l + 15
}
int main() {
    uint32_t h[256];
    for (int i = 0; i < 256; ++i) h[i] = i;
    uint32_t* d; u4* o;
    hipMalloc(&d, sizeof h); hipMalloc(&o, 64 * sizeof(u4));
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o);
    u4 r[64];
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        if (r[l].x != (unsigned)(2 * l) || r[l].y != (unsigned)(2 * l + 1) || r[l].z != (unsigned)(2 * l + 2) || r[l].w != (unsigned)(2 * l + 3)) {
            if (bad < 4) printf("lane %d: got %u %u %u %u, want %d..\n", l, r[l].x, r[l].y, r[l].z, r[l].w, 2 * l);
            ++bad;
        }
    printf("unaligned 16-byte global loads: %d of 64 lanes wrong\n", bad);
    return 0;
}
