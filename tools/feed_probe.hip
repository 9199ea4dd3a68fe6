// Which path feeds a CU faster from L2: LDS-DMA (`buffer_load_dwordx4 ... lds`), plain vector loads into VGPRs, or both at once?
// Decides whether a GEMM tile with ONE operand on each path can beat the ~30 B/clk/CU the all-DMA deep tiles are bound by
// (DESIGN.md "Single-frame path").  Standalone:  hipcc --offload-arch=gfx950 -O3 tools/feed_probe.hip -o tools/build/feed_probe
//
// Every block (4 waves, one block per CU, 256 blocks) streams ITER x PIECES KiB out of an L2-resident window (the blocks of an XCD
// share a 2 MiB window, as the tiles of one GEMM round share their panels).  A "request" is one wave-wide 16-B-per-lane load
// (1 KiB).  mode: d = requests per wave and step that go to LDS, v = requests that go to VGPRs (xor-folded so they are not dead).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int ND, int NV, int DEPTH>
__global__ __launch_bounds__(256, 1) void feed_kernel(const char* __restrict__ src, unsigned* __restrict__ sink, int iters, unsigned win_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const char* base = src + (size_t)xcd * win_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, win_bytes, 0x00020000);
    constexpr int PER = ND + NV;
    // a step of the block covers 4 waves x PER KiB, contiguous; blocks start at staggered offsets; DEPTH steps stay in flight
    unsigned off = (unsigned)(slot * 61 * 1024 * 4 * PER) % win_bytes;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 buf[DEPTH][NV > 0 ? NV : 1];
    const int voff = lane * 16;
    auto issue = [&](u32x4* dstv, int ring) __attribute__((always_inline)) {
        const unsigned o = off + wave * PER * 1024;
#pragma unroll
        for (int i = 0; i < ND; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + ((ring * 4 + wave) * ND + i) * 1024), 16, voff, o + i * 1024, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            dstv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, o + (ND + i) * 1024, 0));
        off += 4 * PER * 1024;
        if (off >= win_bytes) off -= win_bytes;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(buf[d], d);
    for (int it = DEPTH; it < iters; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (NV == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND * (DEPTH - 1)) : "memory");     // the oldest step landed
#pragma unroll
            for (int i = 0; i < NV; ++i) acc ^= buf[d][i];
            issue(buf[d], d);
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < NV; ++i) acc ^= buf[d][i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ND > 0) acc[0] ^= *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int ND, int NV, int DEPTH>
static void run(const char* src, unsigned* sink, unsigned win, int nblk) {
    static_assert(ND * DEPTH + NV * DEPTH <= 63, "vmcnt");
    const int iters = 4096 / (ND + NV) / DEPTH * DEPTH;                    // ~16 MiB per block
    const size_t lds = (size_t)DEPTH * 4 * (ND > 0 ? ND : 1) * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&feed_kernel<ND, NV, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((feed_kernel<ND, NV, DEPTH>), dim3(nblk), dim3(256), lds, 0, src, sink, iters, win);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)nblk * iters * 4 * (ND + NV) * 1024;
    printf("dma %2d + vgpr %2d requests / wave / step, %d steps (%3d KiB per CU) in flight: %8.1f us  %6.2f TB/s  %6.1f GB/s per CU  (%.1f B/clk/CU at 2.1 GHz)\n", ND, NV, DEPTH,
           4 * (ND + NV) * DEPTH, best * 1e3, bytes / best / 1e9, bytes / best / 1e6 / nblk, bytes / best / 1e6 / nblk / 2.1);
}

// ---- GEMM-like feed: the request stream of gemm_h2d_kernel without MFMAs, fragment reads or epilogue.  A tile = BM rows of A + BN rows of
// W, every K step reads 128 B of each row (row stride = the operand's leading dimension in bytes: the lines of one step are `stride`
// apart), tiles mapped to blocks exactly like the kernel (XCD-aware, chunks of 8 N tiles), DEPTH steps in flight behind a counted vmcnt,
// optional block barrier per step (the lockstep of the real K loop).
template <int BM, int BN, int DEPTH, bool BARRIER>
__global__ __launch_bounds__(256, 1) void gemm_feed_kernel(const char* __restrict__ A, const char* __restrict__ W, unsigned* __restrict__ sink, int M, int N,
                                                          int nk, int sa, int sw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_PC = BM / 32, B_PC = BN / 32, PER = A_PC + B_PC, SLOTS = DEPTH < 5 ? DEPTH : 5;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbn = (N + BN - 1) / BN, nbm = (M + BM - 1) / BM, nwg = gridDim.x;
    int L;
    { const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7; L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3); }
    const int per_chunk = nbm * 8, c = L / per_chunk, wc = min(8, nbn - c * 8), rem = L - c * per_chunk;
    const int bm = rem / wc, bn = c * 8 + rem - bm * wc;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A) + (size_t)bm * BM * sa, 0, min(BM, M - bm * BM) * sa, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(W) + (size_t)bn * BN * sw, 0, min(BN, N - bn * BN) * sw, 0x00020000);
    int voa[8], vob[8];       // fixed bound: a template-dependent bound used by the DMA builtin inside a lambda makes hipcc drop the host stubs
#pragma unroll
    for (int i = 0; i < A_PC; ++i) voa[i] = (8 * (wave + 4 * i) + (lane >> 3)) * sa + (lane & 7) * 16;
#pragma unroll
    for (int i = 0; i < B_PC; ++i) vob[i] = (8 * (wave + 4 * i) + (lane >> 3)) * sw + (lane & 7) * 16;
    auto issue = [&](int kt, int slot) __attribute__((always_inline)) {       // kt < 0: a dead request (per-lane offset out of range: zeros; the scalar offset is NOT range-checked)
        char* dst = smem + slot * (BM + BN) * 128 + wave * 1024;
        const bool live = kt >= 0;
#pragma unroll
        for (int i = 0; i < A_PC; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, live ? voa[i] : 0x7fffffff, live ? kt * 128 : 0, 0, 0);
#pragma unroll
        for (int i = 0; i < B_PC; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(dst + BM * 128 + i * 4096), 16, live ? vob[i] : 0x7fffffff, live ? kt * 128 : 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, d % SLOTS);
    int wr = (DEPTH - 1) % SLOTS;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * PER) : "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        issue(kt + DEPTH - 1 < nk ? kt + DEPTH - 1 : -1, wr);     // past the end: dead requests (as the kernel does)
        wr = wr + 1 == SLOTS ? 0 : wr + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (*reinterpret_cast<unsigned*>(smem + threadIdx.x * 4) == 0x12345678u) sink[0] = 1;
}

template <int BM, int BN, int DEPTH, bool BARRIER>
static void run_gemm(const char* buf, unsigned* sink, int M, int N, int K, int pad_a, int pad_w) {
    static_assert((DEPTH - 1) * (BM + BN) / 32 <= 63, "vmcnt");
    const int sa = K * 4 + pad_a, sw = K * 4 + pad_w, nk = K / 32;
    const char* A = buf;
    const char* W = buf + (((size_t)M * sa + 4095) & ~(size_t)4095);
    const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const size_t lds = (size_t)(DEPTH < 5 ? DEPTH : 5) * (BM + BN) * 128;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_feed_kernel<BM, BN, DEPTH, BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((gemm_feed_kernel<BM, BN, DEPTH, BARRIER>), dim3(grid), dim3(256), lds, 0, A, W, sink, M, N, nk, sa, sw);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)grid * nk * (BM + BN) * 128;
    printf("gemm feed %5d x %4d x %4d  tile %3d x %3d (%3d blocks)  %d deep%s  pad A %3d W %3d: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU at 2.1 GHz\n", M, N, K, BM, BN, grid, DEPTH,
           BARRIER ? " +barrier" : "         ", pad_a, pad_w, best * 1e3, bytes / best / 1e9, bytes / best / 1e6 / (grid < 256 ? grid : 256) / 2.1);
}

// ---- the same request stream WITH the MFMA / fragment-read load of the 128 x 96 deep tile (TM = 1, TN = 3: 18 MFMAs + 16 ds_read_b128 per
// wave and K step), two ways:  SPEC = false: every wave issues its DMA requests between its own MFMAs (gemm_h2d_kernel's schedule, 4 waves);
// SPEC = true: 8 waves, waves 0-3 only read fragments and multiply, waves 4-7 only issue the DMA requests (one producer + one consumer per
// SIMD) -- does a second wave per SIMD take the VMEM issue stalls off the MFMA stream?
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// WB = 1: the weight panel is stored K-BLOCKED in granules of 32 rows x one 32-k slice (4 KiB contiguous; a 1-KiB request = 8 consecutive
// 128-B rows of one granule is ONE contiguous KiB instead of 8 lines a row stride apart): granule (n / 32, kt) at ((n / 32) * nk + kt) * 4096.
// AB = 1: the same for the activation panel (what a producing kernel would have to write).
template <bool SPEC, int DEPTH, int WB = 0, int AB = 0>
__global__ __launch_bounds__(SPEC ? 512 : 256, 1) void gemm_mix_kernel(const char* __restrict__ A, const char* __restrict__ W, float* __restrict__ sink, int M, int N,
                                                                       int nk, int sa, int sw) {
    constexpr int BM = 128, BN = 96, A_PC = 4, B_PC = 3, PER = 7;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = !SPEC || wave >= 4, consumer = !SPEC || wave < 4;
    const int pw = wave & 3;
    const int nbn = (N + BN - 1) / BN, nbm = (M + BM - 1) / BM, nwg = gridDim.x;
    int L;
    { const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7; L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3); }
    const int per_chunk = nbm * 8, c = L / per_chunk, wc = min(8, nbn - c * 8), rem = L - c * per_chunk;
    const int bm = rem / wc, bn = c * 8 + rem - bm * wc;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A) + (size_t)bm * BM * sa, 0, min(BM, M - bm * BM) * sa, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(W) + (size_t)bn * BN * sw, 0, min(BN, N - bn * BN) * sw, 0x00020000);
    int voa[8], vob[8];
#pragma unroll
    for (int i = 0; i < A_PC; ++i) {
        const int r = 8 * (pw + 4 * i) + (lane >> 3);
        voa[i] = AB ? (r >> 5) * (nk * 4096) + (r & 31) * 128 + (lane & 7) * 16 : r * sa + (lane & 7) * 16;
    }
#pragma unroll
    for (int i = 0; i < B_PC; ++i) {
        const int r = 8 * (pw + 4 * i) + (lane >> 3);
        vob[i] = WB ? (r >> 5) * (nk * 4096) + (r & 31) * 128 + (lane & 7) * 16 : r * sw + (lane & 7) * 16;
    }
    auto issue = [&](int kt, int slot, bool live) __attribute__((always_inline)) {
        char* dst = smem + slot * (BM + BN) * 128 + pw * 1024;
#pragma unroll
        for (int i = 0; i < A_PC; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, live ? voa[i] : 0x7fffffff, kt * (AB ? 4096 : 128), 0, 0);
#pragma unroll
        for (int i = 0; i < B_PC; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(dst + BM * 128 + i * 4096), 16, live ? vob[i] : 0x7fffffff, kt * (WB ? 4096 : 128), 0, 0);
    };
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8 fa[2][2], fb[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i) { fa[1][i] = f16x8{}; }
#pragma unroll
    for (int i = 0; i < 6; ++i) { fb[1][i] = f16x8{}; }
    auto ldfrag = [&](int sbase, int half, int set) __attribute__((always_inline)) {
        const int row = pw * 32 + (lane & 31), ch = 2 * (2 * half + (lane >> 5));
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[set][i] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>((size_t)(sbase + row * 128 + (((ch + i) ^ ((row >> 1) & 7)) << 4)));
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rb = BM + j * 32 + (lane & 31);
                fb[set][2 * j + i] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>((size_t)(sbase + rb * 128 + (((ch + i) ^ ((rb >> 1) & 7)) << 4)));
            }
    };
    auto mma = [&](int set, int term) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? fb[set][2 * j + 1] : fb[set][2 * j], term == 1 ? fa[set][1] : fa[set][0], acc[j], 0, 0, 0);
    };
    if (producer)
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) issue(d, d, d < nk);
    int rd = 0, wr = DEPTH - 1;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int kt = 0; kt < nk; ++kt) {
        if (producer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * PER) : "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int sbase = lds0 + rd * (BM + BN) * 128;
        rd = rd + 1 == DEPTH ? 0 : rd + 1;
        if (SPEC) {
            if (consumer) {
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 0);
                __builtin_amdgcn_sched_barrier(0);
                ldfrag(sbase, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 1); mma(1, 2);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 0);
                __builtin_amdgcn_sched_barrier(0);
                ldfrag(sbase, 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 1); mma(0, 2);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                issue(kt + DEPTH - 1, wr, kt + DEPTH - 1 < nk);
            }
        } else {
            __builtin_amdgcn_sched_barrier(0);
            mma(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            ldfrag(sbase, 0, 0);
            issue(kt + DEPTH - 1, wr, kt + DEPTH - 1 < nk);
            __builtin_amdgcn_sched_barrier(0);
            mma(1, 1); mma(1, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ldfrag(sbase, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(0, 1); mma(0, 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        wr = wr + 1 == DEPTH ? 0 : wr + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[j][r];
    if (t == 1234.5f) sink[0] = t;
}

template <bool SPEC, int DEPTH, int WB = 0, int AB = 0>
static void run_mix(const char* buf, float* sink, int M, int N, int K) {
    const int sa = K * 4, sw = K * 4, nk = K / 32;
    const char* A = buf;
    const char* W = buf + (((size_t)M * sa + 4095) & ~(size_t)4095);
    const int grid = ((M + 127) / 128) * ((N + 95) / 96);
    const size_t lds = (size_t)DEPTH * (128 + 96) * 128;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mix_kernel<SPEC, DEPTH, WB, AB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((gemm_mix_kernel<SPEC, DEPTH, WB, AB>), dim3(grid), dim3(SPEC ? 512 : 256), lds, 0, A, W, sink, M, N, nk, sa, sw);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("gemm feed + MFMA %5d x %4d x %4d  tile 128 x 96 (%3d blocks)  %d deep  W %s A %s  %s: %7.1f us  (%.0f TF-eq of f16x2 products)\n", M, N, K, grid, DEPTH,
           WB ? "K-blocked" : "row-major", AB ? "K-blocked" : "row-major", SPEC ? "8 waves: 4 consumers + 4 producers" : "4 waves, DMA issued between the MFMAs  ", best * 1e3, 2.0 * M * N * K / best / 1e9);
}

int main(int argc, char** argv) {
    const unsigned win = (argc > 1 ? atoi(argv[1]) : 2) << 20;
    const int nblk = argc > 2 ? atoi(argv[2]) : 256;
    char* src; unsigned* sink;
    hipMalloc(&src, (size_t)win * 8);
    hipMalloc(&sink, 4);
    std::vector<unsigned> h((size_t)win * 8 / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    hipMemcpy(src, h.data(), (size_t)win * 8, hipMemcpyHostToDevice);
    printf("window %u MiB per XCD, %d blocks of 4 waves\n", win >> 20, nblk);
    run<8, 0, 2>(src, sink, win, nblk);
    run<8, 0, 4>(src, sink, win, nblk);
    run<4, 0, 8>(src, sink, win, nblk);
    run<0, 8, 2>(src, sink, win, nblk);
    run<0, 8, 4>(src, sink, win, nblk);
    run<0, 12, 4>(src, sink, win, nblk);
    run<4, 4, 2>(src, sink, win, nblk);
    run<4, 4, 4>(src, sink, win, nblk);
    run<6, 6, 4>(src, sink, win, nblk);
    run<7, 8, 4>(src, sink, win, nblk);
    run<3, 4, 4>(src, sink, win, nblk);
    run<3, 4, 8>(src, sink, win, nblk);
    if (argc > 3) {
        char* big;
        hipMalloc(&big, (size_t)1088 << 20);      // the 64000 x 3072 f16x2 panel is 786 MB
        hipMemset(big, 1, (size_t)1088 << 20);
        run_mix<false, 4>(big, (float*)sink, 4000, 768, 3072);
        run_mix<false, 4, 1>(big, (float*)sink, 4000, 768, 3072);
        run_mix<false, 4, 1, 1>(big, (float*)sink, 4000, 768, 3072);
        run_mix<true, 4, 1, 1>(big, (float*)sink, 4000, 768, 3072);
        run_mix<true, 4>(big, (float*)sink, 4000, 768, 3072);
        run_mix<false, 3>(big, (float*)sink, 4000, 768, 3072);
        run_mix<true, 5>(big, (float*)sink, 4000, 768, 3072);
        run_mix<false, 4>(big, (float*)sink, 4000, 768, 1536);
        run_mix<true, 4>(big, (float*)sink, 4000, 768, 1536);
        const int pads[][2] = {{0, 0}, {128, 128}, {256, 256}, {128, 0}, {0, 128}, {512, 512}};
        for (auto& pd : pads) {
            run_gemm<128, 96, 4, true>(big, sink, 4000, 768, 3072, pd[0], pd[1]);      // stage-2 pwconv2 of one frame
            run_gemm<128, 96, 4, false>(big, sink, 4000, 768, 3072, pd[0], pd[1]);
        }
        run_gemm<128, 96, 3, true>(big, sink, 4000, 768, 3072, 0, 0);
        run_gemm<128, 96, 6, true>(big, sink, 4000, 768, 3072, 0, 0);
        run_gemm<128, 96, 8, true>(big, sink, 4000, 768, 3072, 0, 0);
        for (auto& pd : pads) run_gemm<256, 192, 2, true>(big, sink, 4000, 3072, 768, pd[0], pd[1]);   // stage-2 pwconv1 of one frame (cfg 346)
        for (auto& pd : pads) run_gemm<64, 64, 3, true>(big, sink, 4000, 768, 3072, pd[0], pd[1]);
        run_gemm<64, 64, 8, true>(big, sink, 4000, 768, 3072, 0, 0);
        // 16 frames: stage-2 pwconv1 / pwconv2 on 256 x 256 tiles (gemm_h2q: 2 stages)
        for (auto& pd : pads) run_gemm<256, 256, 2, true>(big, sink, 64000, 3072, 768, pd[0], pd[1]);
        for (auto& pd : pads) run_gemm<256, 256, 2, true>(big, sink, 64000, 768, 3072, pd[0], pd[1]);
        hipFree(big);
    }
    hipDeviceSynchronize();
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
