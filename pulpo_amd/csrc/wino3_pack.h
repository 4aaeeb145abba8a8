// Weight transform of the F(2x2x2,3x3x3) kernel (conv3d_wino3.hip), shared with the all-packs-in-one-launch refresh (conv3d_wino.hip):
//   U = (G x G x G) w,  G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]] along z, y and x
//   wp[((chunk * 64 + pz * 16 + py * 4 + px) * NPad + n) * 8 + k % 8]      (forward: K = Cin, N = Cout, w[n][k][tap]; dgrad: K = Cout, N = Cin, w[k][n][26 - tap])
#pragma once
#include "common.h"

namespace pulpo_conv {

__device__ __forceinline__ void wino3_g(const float (&g)[3], float (&u)[4]) {
    u[0] = g[0]; u[1] = 0.5f * (g[0] + g[1] + g[2]); u[2] = 0.5f * (g[0] - g[1] + g[2]); u[3] = g[2];
}

// one thread per (chunk, n, k % 8): e = (chunk * NPad + n) * 8 + kc
__device__ __forceinline__ void pack_wino3_one(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int NPad, int dgrad, long e) {
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const int kc = (int)(e % 8);
    long r = e / 8;
    const int n = (int)(r % NPad);
    const int chunk = (int)(r / NPad);
    const int k = chunk * 8 + kc;
    const bool live = k < K && n < N;
    float uy[3][4][4];                          // [dz][py][px]
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
        float ux[3][4];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            float g[3] = {0.f, 0.f, 0.f};
            if (live) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int tap = (dz * 3 + dy) * 3 + dx;
                    g[dx] = dgrad ? w[((long)k * Cin + n) * 27 + (26 - tap)] : w[((long)n * Cin + k) * 27 + tap];
                }
            }
            wino3_g(g, ux[dy]);
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const float g[3] = {ux[0][px], ux[1][px], ux[2][px]};
            float u[4];
            wino3_g(g, u);
#pragma unroll
            for (int py = 0; py < 4; ++py) uy[dz][py][px] = u[py];
        }
    }
    float* o = wp + (((long)chunk * 64) * NPad + n) * 8 + kc;
#pragma unroll
    for (int py = 0; py < 4; ++py)
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const float g[3] = {uy[0][py][px], uy[1][py][px], uy[2][py][px]};
            float u[4];
            wino3_g(g, u);
#pragma unroll
            for (int pz = 0; pz < 4; ++pz) o[(long)(pz * 16 + py * 4 + px) * NPad * 8] = u[pz];
        }
}

}  // namespace pulpo_conv
