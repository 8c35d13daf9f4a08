// Which byte does the buffer unit fetch for (index, offset) under a swizzled descriptor?  The buffer holds its own
// element numbers (uint16 k at byte 2k); prints the fetched element for a grid of (index, offset) per descriptor.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/buf_swizzle.hip -o tools/ubench/buf_swizzle
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__global__ void probe(const uint16_t *buf, unsigned w1_extra, unsigned w2, unsigned w3, const unsigned *pairs, unsigned *out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const unsigned long long base = (unsigned long long)buf;
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    r.y = __builtin_amdgcn_readfirstlane((int)(((unsigned)(base >> 32) & 0xffffu) | w1_extra));
    r.z = __builtin_amdgcn_readfirstlane((int)w2);
    r.w = __builtin_amdgcn_readfirstlane((int)w3);
    const v2u io = v2u{pairs[2 * i], pairs[2 * i + 1]};
    unsigned v;
    asm volatile("buffer_load_ushort %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(io), "s"(r) : "memory");
    out[i] = v;
}

int main() {
    const int N = 1 << 16;
    std::vector<uint16_t> h(N);
    for (int k = 0; k < N; ++k) h[k] = (uint16_t)k;
    uint16_t *d; hipMalloc(&d, N * 2); hipMemcpy(d, h.data(), N * 2, hipMemcpyHostToDevice);
    std::vector<unsigned> pairs;
    const unsigned idxs[] = {0, 1, 2, 7, 8, 9, 16}, offs[] = {0, 2, 14, 16, 18, 32, 256};
    for (unsigned a : idxs) for (unsigned b : offs) { pairs.push_back(a); pairs.push_back(b); }
    const int n = (int)pairs.size() / 2;
    unsigned *dp, *dout; hipMalloc(&dp, pairs.size() * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dp, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice);
    const unsigned stride = 1024;
    struct { const char *name; unsigned w1, w2, w3; } descs[] = {
        {"swizzle, w3 = 0x00060000 (bits 18:17 = 3)", (stride << 16) | 0x80000000u, 64, 0x00060000},
        {"swizzle, w3 = 0x00040000 (bits 18:17 = 2)", (stride << 16) | 0x80000000u, 64, 0x00040000},
        {"swizzle, w3 = 0x00000000", (stride << 16) | 0x80000000u, 64, 0x00000000},
        {"swizzle, w3 = 0x00180000 (bits 20:19 = 3, bit 17 clear)", (stride << 16) | 0x80000000u, 64, 0x00180000},
    };
    std::vector<unsigned> out(n);
    for (auto &ds : descs) {
        hipMemset(dout, 0xff, n * 4);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, ds.w1, ds.w2, ds.w3, dp, dout, n);
        hipDeviceSynchronize();
        hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost);
        printf("== %s\n", ds.name);
        for (int i = 0; i < n; ++i) printf("  idx %2u off %3u -> byte %6u%s", pairs[2 * i], pairs[2 * i + 1], out[i] * 2, (i % 7 == 6) ? "\n" : "");
    }
    return 0;
}
