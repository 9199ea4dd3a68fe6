// Winograd F(2x2, 3x3) for the 3x3 convolutions of the head towers / FPN: what the two TRANSFORM passes cost on the MI355X.
// F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input patch with 16 multiplies per (input channel, output channel) instead of 36: the 3x3
// implicit GEMM (K = 9 Cin) becomes 16 independent GEMMs of K = Cin over M / 4 tiles -- 2.25x fewer MFMAs -- plus
//   input transform  V = B^T d B   per 4x4 patch and channel (adds only), written as 16 operand planes [16][M/4][Cin] in the engine's f16x2 format
//   output transform Y = A^T m A   per tile and output channel from the 16 fp32 product planes [16][M/4][Cout] (+ bias), fp32 map out.
// The engine's GEMMs take their A operand by LDS-DMA (no register stage), so the transforms cannot ride inside the GEMM: they are separate
// HBM-bound passes.  This program times both on the head-tower shape (16 frames x 100 x 160, 256 -> 256) and the FPN shape (16 x 50 x 80, 384 -> 384);
// tools/winograd_probe.py adds the GEMM halves through the library.  Standalone: no library code is used here.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {      // fp32 -> f16 hi + f16 lo (the engine's operand split, saturating)
    __half h[4], l[4];
    for (int e = 0; e < 4; ++e) {
        float x = fminf(fmaxf(v[e], -65504.f), 65504.f);
        h[e] = __float2half_rn(x);
        l[e] = __float2half_rn(x - __half2float(h[e]));
    }
    hi = u32x2{(unsigned)__half_as_ushort(h[0]) | ((unsigned)__half_as_ushort(h[1]) << 16), (unsigned)__half_as_ushort(h[2]) | ((unsigned)__half_as_ushort(h[3]) << 16)};
    lo = u32x2{(unsigned)__half_as_ushort(l[0]) | ((unsigned)__half_as_ushort(l[1]) << 16), (unsigned)__half_as_ushort(l[2]) | ((unsigned)__half_as_ushort(l[3]) << 16)};
}

// x: [B][H][W][C] fp32 (NHWC), V: [16][B * H/2 * W/2][C] in 8-channel groups of [8 hi][8 lo] f16 (4 B per element).  thread = (tile, 4 channels)
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, unsigned* __restrict__ V, int B, int H, int W, int C) {
    const int CG = C / 4, TH = H / 2, TW = W / 2;
    const long ntile = (long)B * TH * TW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ntile * CG; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        const long t = i / CG;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((long)TW * TH));
        f32x4 d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int iy = 2 * ty - 1 + r, ix = 2 * tx - 1 + c;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4*>(x + (((long)b * H + iy) * W + ix) * C + cg * 4);
                d[r][c] = v;
            }
        f32x4 u[4][4];                                             // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
#pragma unroll
        for (int c = 0; c < 4; ++c) { u[0][c] = d[0][c] - d[2][c]; u[1][c] = d[1][c] + d[2][c]; u[2][c] = d[2][c] - d[1][c]; u[3][c] = d[1][c] - d[3][c]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 v0 = u[r][0] - u[r][2], v1 = u[r][1] + u[r][2], v2 = u[r][2] - u[r][1], v3 = u[r][1] - u[r][3];
            const f32x4 vv[4] = {v0, v1, v2, v3};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                u32x2 hi, lo;
                split4(vv[c], hi, lo);
                // plane (r * 4 + c), row t, 8-channel group cg / 2: [8 hi][8 lo] halves; this lane owns 4 of the 8 channels
                unsigned* dst = V + (((long)(r * 4 + c) * ntile + t) * C + (long)(cg >> 1) * 8) ;      // 4-byte units: 8 per group of 8 channels
                *reinterpret_cast<u32x2*>(dst + (cg & 1) * 2) = hi;
                *reinterpret_cast<u32x2*>(dst + 4 + (cg & 1) * 2) = lo;
            }
        }
    }
}

// Mo: [16][ntile][C] fp32 product planes -> y: [B][H][W][C] fp32 (+ bias).  thread = (tile, 4 channels)
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mo, const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W, int C) {
    const int CG = C / 4, TH = H / 2, TW = W / 2;
    const long ntile = (long)B * TH * TW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ntile * CG; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        const long t = i / CG;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((long)TW * TH));
        f32x4 m[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) m[r][c] = *reinterpret_cast<const f32x4*>(Mo + ((long)(r * 4 + c) * ntile + t) * C + cg * 4);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + cg * 4);
        f32x4 s[2][4];                                             // A^T m: rows (m0 + m1 + m2, m1 - m2 - m3)
#pragma unroll
        for (int c = 0; c < 4; ++c) { s[0][c] = m[0][c] + m[1][c] + m[2][c]; s[1][c] = m[1][c] - m[2][c] - m[3][c]; }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const f32x4 o0 = s[r][0] + s[r][1] + s[r][2] + bb, o1 = s[r][1] - s[r][2] - s[r][3] + bb;
            float* dst = y + (((long)b * H + 2 * ty + r) * W + 2 * tx) * C + cg * 4;
            *reinterpret_cast<f32x4*>(dst) = o0;
            *reinterpret_cast<f32x4*>(dst + C) = o1;
        }
    }
}

static void run(const char* name, int B, int H, int W, int Cin, int Cout) {
    const long M = (long)B * H * W, ntile = M / 4;
    float *x, *Mo, *bias, *y;
    unsigned* V;
    CK(hipMalloc(&x, M * Cin * 4));
    CK(hipMalloc(&V, 16 * ntile * Cin * 4));
    CK(hipMalloc(&Mo, 16 * ntile * Cout * 4));
    CK(hipMalloc(&bias, Cout * 4));
    CK(hipMalloc(&y, M * Cout * 4));
    CK(hipMemset(x, 0x3c, M * Cin * 4));
    CK(hipMemset(Mo, 0x3c, 16 * ntile * Cout * 4));
    CK(hipMemset(bias, 0, Cout * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best_in = 1e9f, best_out = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        float ms;
        CK(hipEventRecord(e0));
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(wino_input_kernel, dim3(4096), dim3(256), 0, 0, x, V, B, H, W, Cin);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        best_in = ms / 5 < best_in ? ms / 5 : best_in;
        CK(hipEventRecord(e0));
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(wino_output_kernel, dim3(4096), dim3(256), 0, 0, Mo, bias, y, B, H, W, Cout);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        best_out = ms / 5 < best_out ? ms / 5 : best_out;
    }
    const double in_bytes = (double)M * Cin * 4 + 16.0 * ntile * Cin * 4, out_bytes = 16.0 * ntile * Cout * 4 + (double)M * Cout * 4;
    printf("%-22s B %2d %3dx%3d %3d->%3d: input transform %7.1f us (%.0f MB compulsory, %.2f TB/s) | output transform %7.1f us (%.0f MB, %.2f TB/s)\n", name, B, H, W, Cin,
           Cout, best_in * 1e3, in_bytes / 1e6, in_bytes / (best_in * 1e-3) / 1e12, best_out * 1e3, out_bytes / 1e6, out_bytes / (best_out * 1e-3) / 1e12);
    CK(hipFree(x)); CK(hipFree(V)); CK(hipFree(Mo)); CK(hipFree(bias)); CK(hipFree(y));
}

int main() {
    run("head tower 3x3 (s8)", 16, 100, 160, 256, 256);
    run("FPN 3x3 (s16)", 16, 50, 80, 384, 384);
    run("head tower 3x3, 1 frame", 1, 100, 160, 256, 256);
    return 0;
}
