// What does a persistent per-stage kernel have to gain at one frame per call?  (measured before building.)
// A ConvNeXt stage at one frame per call is a chain of ~81 DEPENDENT one-round kernels (dw7x7+LN -> pwconv1 -> pwconv2, x27).  A persistent kernel
// replaces every kernel boundary by a grid barrier among co-resident blocks.  This standalone program measures both on the MI355X:
//   (1) chain:   N dependent launches of a phase kernel on one stream (what the engine does today),
//   (2) coop:    ONE hipLaunchCooperativeKernel whose blocks run the same N phases with a grid barrier (atomic counter + spin) between them,
//   (3) plain:   the same persistent kernel through a plain launch (co-residency by construction: 256 blocks, one per CU) -- isolates what the
//                cooperative launch path itself costs,
// for three phase bodies: "empty" (one store per block: pure boundary / barrier cost), "stream" (every block reads 256 KiB that ANOTHER block
// wrote in the previous phase and writes 256 KiB: a small memory-bound layer, 64 MiB each way per phase) and "mfma" (~20 us of MFMA work per
// block on registers + the 256 KiB hand-over: a one-round GEMM-like phase).  Prints us per phase.  Build: csrc/build.sh (tools/build/persist_probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CHUNK = 256 * 1024;      // bytes a block hands to the next phase

template <int BODY>
__device__ __forceinline__ void phase_body(const char* src, char* dst, int phase, int nblk, int mfma_iters) {
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    if (BODY == 0) {
        if (tid == 0) reinterpret_cast<float*>(dst)[b * 64] = (float)phase;
        return;
    }
    const int from = (b + 1 + phase) % nblk;                       // a chunk another block (another XCD: b -> b + 1) wrote in the previous phase
    const f32x4* s = reinterpret_cast<const f32x4*>(src + (size_t)from * CHUNK);
    f32x4* d = reinterpret_cast<f32x4*>(dst + (size_t)b * CHUNK);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (BODY == 2) {
        f32x16 c = {};
        f16x8 a = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)2.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
        f16x8 w = a;
        for (int i = 0; i < mfma_iters; ++i) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, w, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, w, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, c, 0, 0, 0);
        }
        acc[0] = c[0] * 1e-30f;
    }
    for (int i = tid; i < CHUNK / 16; i += nt) {
        f32x4 v = s[i];
        v[0] += acc[0] + 1.f;
        d[i] = v;
    }
}

template <int BODY>
__global__ __launch_bounds__(256) void phase_kernel(const char* src, char* dst, int phase, int nblk, int mfma_iters) {
    phase_body<BODY>(src, dst, phase, nblk, mfma_iters);
}

// mode 0: every block's thread 0 adds to ONE counter and polls it (agent-scope acquire loads, s_sleep between polls);
// mode 1: the same with a long sleep between polls (fewer requests queued in front of the last arrivals' atomics);
// mode 2: TWO LEVELS -- the blocks of an XCD (blockIdx & 7: round-robin dispatch) arrive on their XCD's counter (own cache line), the last arrival of
//         an XCD adds to the global counter, polls it and then releases its XCD through a per-XCD flag the other blocks of the XCD poll: 8 pollers on
//         the global line instead of 256;
// mode 3: one counter, RELAXED add and polls between the two fences (no acquire -- cache invalidation -- per poll).
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned phase, unsigned nblk, int mode) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                               // release the phase's stores to the device
        if (mode == 3) {                                               // relaxed add + relaxed polls, fences only around them
            const unsigned target = (phase + 1) * nblk;
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
        } else if (mode < 2) {
            const unsigned target = (phase + 1) * nblk;
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (mode == 0) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(32);
            }
        } else {
            const unsigned x = blockIdx.x & 7, per = nblk >> 3;        // (nblk a multiple of 8)
            unsigned* xc = counter + 64 + x * 64;                      // per-XCD arrival counter, 256-byte apart
            unsigned* xf = counter + 64 + 8 * 64 + x * 64;             // per-XCD release flag (phase number + 1)
            const unsigned prev = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == (phase + 1) * per - 1) {                       // last block of this XCD
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (phase + 1) * 8u) __builtin_amdgcn_s_sleep(1);
                __hip_atomic_store(xf, phase + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(xf, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < phase + 1) __builtin_amdgcn_s_sleep(2);
            }
        }
        __threadfence();
    }
    __syncthreads();
}

template <int BODY>
__global__ __launch_bounds__(256) void persistent_kernel(char* buf0, char* buf1, int nphase, int nblk, int mfma_iters, unsigned* counter, int mode) {
    for (int p = 0; p < nphase; ++p) {
        const char* src = (p & 1) ? buf1 : buf0;
        char* dst = (p & 1) ? buf0 : buf1;
        phase_body<BODY>(src, dst, p, nblk, mfma_iters);
        grid_barrier(counter, (unsigned)p, (unsigned)nblk, mode);
    }
}

template <int BODY>
static void run(const char* name, int nblk, int nphase, int mfma_iters) {
    char *b0, *b1;
    unsigned* counter;
    const size_t cbytes = 4096 * 4;
    CK(hipMalloc(&b0, (size_t)nblk * CHUNK));
    CK(hipMalloc(&b1, (size_t)nblk * CHUNK));
    CK(hipMalloc(&counter, cbytes));
    CK(hipMemset(b0, 0, (size_t)nblk * CHUNK));
    CK(hipMemset(b1, 0, (size_t)nblk * CHUNK));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    float best_graph = 1e9f;
    float best_chain = 1e9f, best_coop[4] = {1e9f, 1e9f, 1e9f, 1e9f}, best_plain[4] = {1e9f, 1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 5; ++rep) {
        // (1) chain of dependent launches
        CK(hipEventRecord(e0, s));
        for (int p = 0; p < nphase; ++p)
            hipLaunchKernelGGL((phase_kernel<BODY>), dim3(nblk), dim3(256), 0, s, (p & 1) ? b1 : b0, (p & 1) ? b0 : b1, p, nblk, mfma_iters);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best_chain = ms < best_chain ? ms : best_chain;
        // (1b) the same chain captured ONCE into a hipGraph and replayed (does a graph shorten the boundary?)
        if (rep == 0) {
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int p = 0; p < nphase; ++p)
                hipLaunchKernelGGL((phase_kernel<BODY>), dim3(nblk), dim3(256), 0, s, (p & 1) ? b1 : b0, (p & 1) ? b0 : b1, p, nblk, mfma_iters);
            CK(hipStreamEndCapture(s, &graph));
            CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
            CK(hipGraphLaunch(gexec, s));
            CK(hipStreamSynchronize(s));
        }
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(gexec, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        best_graph = ms < best_graph ? ms : best_graph;
        for (int mode = 0; mode < 4; ++mode) {
            // (2) one cooperative launch
            CK(hipMemsetAsync(counter, 0, cbytes, s));
            void* args[] = {&b0, &b1, &nphase, &nblk, &mfma_iters, &counter, &mode};
            CK(hipEventRecord(e0, s));
            hipError_t ce = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&persistent_kernel<BODY>), dim3(nblk), dim3(256), args, 0, s);
            if (ce != hipSuccess) { fprintf(stderr, "cooperative launch refused: %s\n", hipGetErrorString(ce)); best_coop[mode] = -1.f; (void)hipGetLastError(); }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (best_coop[mode] >= 0.f) best_coop[mode] = ms < best_coop[mode] ? ms : best_coop[mode];
            // (3) the same kernel through a plain launch (one block per CU, co-resident by construction on an idle device)
            CK(hipMemsetAsync(counter, 0, cbytes, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL((persistent_kernel<BODY>), dim3(nblk), dim3(256), 0, s, b0, b1, nphase, nblk, mfma_iters, counter, mode);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            best_plain[mode] = ms < best_plain[mode] ? ms : best_plain[mode];
        }
    }
    const float k = 1e3f / nphase;
    printf("%-7s blocks %3d phases %3d: chain %7.2f (as a replayed hipGraph %7.2f) | cooperative: one counter %7.2f, long sleep %7.2f, two-level %7.2f, relaxed polls %7.2f | plain launch: %7.2f %7.2f %7.2f %7.2f   (us per phase)\n",
           name, nblk, nphase, best_chain * k, best_graph * k, best_coop[0] * k, best_coop[1] * k, best_coop[2] * k, best_coop[3] * k, best_plain[0] * k, best_plain[1] * k, best_plain[2] * k,
           best_plain[3] * k);
    CK(hipGraphExecDestroy(gexec)); CK(hipGraphDestroy(graph));
    CK(hipFree(b0)); CK(hipFree(b1)); CK(hipFree(counter));
    CK(hipStreamDestroy(s));
}

int main(int argc, char** argv) {
    const int nphase = argc > 1 ? atoi(argv[1]) : 81;
    int dev = 0, coop = 0, ncu = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent_kernel<1>, 256, 0));
    printf("device %d: %d CUs, cooperative launch %s, persistent_kernel<stream> co-resident blocks per CU %d\n", dev, ncu, coop ? "supported" : "NOT supported", occ);
    for (int nblk : {256}) {
        run<0>("empty", nblk, nphase, 0);
        run<1>("stream", nblk, nphase, 0);
        run<2>("mfma", nblk, nphase, 1200);       // 4800 MFMAs of 32x32x16 per wave, 4 waves per block: ~20 us
    }
    // one phase only: what a single launch costs either way (launch path overhead)
    run<0>("empty", 256, 1, 0);
    run<1>("stream", 256, 1, 0);
    return 0;
}
