// Device-side scalar math of the env spec: fp32, one IEEE operation per written operator.
// Built with -ffp-contract=off so no multiply-add is fused; divisions are hipcc's
// correctly-rounded default.  Same operation order as oracle/racecar_oracle.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "racecar_spec.h"

namespace rcd {

__device__ __forceinline__ float clampf(float d, float lo, float hi) {
    return d < lo ? lo : (d > hi ? hi : d);
}

// (sin, cos) for |a| <= ~2*pi: Cody-Waite reduction by pi/2 + cephes sinf/cosf polynomials.
__device__ __forceinline__ void sincos32(float a, float &sn, float &cs) {
    const float kf = rintf(a * 0.636619772367581343f);
    const float r = ((a - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    const int q = ((int)kf) & 3;
    const float z = r * r;
    const float s = r + (r * z) * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float c = (1.0f - 0.5f * z) + (z * z) * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    sn = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
    cs = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}

// exp for the max_speed reward (baselines/racing/environment/tasks.py:13): cephes expf.
__device__ __forceinline__ float exp32(float x) {
    x = clampf(x, -80.0f, 80.0f);
    const float kf = rintf(x * 1.44269504088896341f);
    const float r = (x - kf * 0.693359375f) - kf * -2.12194440e-4f;
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    const float y = (p * z + r) + 1.0f;
    return y * __int_as_float((((int)kf) + 127) << 23);
}

// Philox4x32-10 (Salmon et al., SC'11).
struct u32x4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                             uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

}  // namespace rcd
