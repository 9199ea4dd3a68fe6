// What would an IN-KERNEL split-K reduction cost at one frame per call?  ("what would still pay".)
// The stage-2 pwconv2 of one frame (4000 x 768 x 3072) runs as 256 small tiles because a larger tile leaves CUs idle; cutting K in ranges instead needs the partial
// tiles of a pair / quad of blocks to be summed.  The slab + reduce-kernel form costs a launch and a round trip (12.5 us, round 3); this program measures the
// alternative: blocks hand their partial tile to a partner INSIDE the kernel -- producer: write 192 KiB, release fence, flag; consumer: spin on the flag, acquire
// fence, read + add + write 192 KiB -- after ~20 us of MFMA work each, against the same kernel without the hand-over (every block writes its own 192 KiB).
// Partners are blocks b and b + 8 (same XCD under round-robin dispatch) or b and b + 1 (different XCDs); consumers carry the HIGHER block ids so that a
// producer can never wait behind a spinning consumer.  Prints us per kernel.  Standalone; built by csrc/build.sh (tools/build/handoff_probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TILE = 192 * 1024;      // bytes of a 256 x 192 fp32 partial tile

__device__ __forceinline__ float mfma_work(int iters) {
    f32x16 c = {};
    f16x8 a = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)2.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    for (int i = 0; i < iters; ++i) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c, 0, 0, 0);
    }
    return c[0] * 1e-30f;
}

// mode 0: no hand-over (every block writes its tile); 1: partner = b + 8 (same XCD); 2: partner = b + 1 (neighbouring XCD)
__global__ __launch_bounds__(256) void handoff_kernel(char* slab, char* out, unsigned* flags, int mode, int iters, unsigned epoch) {
    const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    const float w = mfma_work(iters);
    f32x4 v = {w + 1.f, 2.f, 3.f, 4.f};
    if (mode == 0) {
        f32x4* d = reinterpret_cast<f32x4*>(out + (size_t)b * TILE);
        for (int i = tid; i < TILE / 16; i += 256) d[i] = v;
        return;
    }
    const int half = nb / 2;
    const bool consumer = b >= half;                                  // the HIGHER ids wait: every producer is dispatched before any consumer
    const int lb = consumer ? b - half : b;
    // pair index -> the two block ids: mode 1 keeps both on XCD (lb % 8); mode 2 puts the producer on the next XCD
    const int pid = lb;
    if (!consumer) {
        const int slot = mode == 1 ? pid : (pid + 1) % half;          // (mode 2: write the tile a consumer on ANOTHER XCD will read)
        f32x4* d = reinterpret_cast<f32x4*>(slab + (size_t)slot * TILE);
        for (int i = tid; i < TILE / 16; i += 256) d[i] = v;
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            __hip_atomic_store(flags + slot * 32, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        if (tid == 0) {
            while (__hip_atomic_load(flags + pid * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(2);
            __threadfence();
        }
        __syncthreads();
        const f32x4* s = reinterpret_cast<const f32x4*>(slab + (size_t)pid * TILE);
        f32x4* d = reinterpret_cast<f32x4*>(out + (size_t)pid * TILE);
        for (int i = tid; i < TILE / 16; i += 256) d[i] = s[i] + v;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2400;               // 4800 MFMAs per wave: ~20 us
    const int nb = 256;
    char *slab, *out;
    unsigned* flags;
    CK(hipMalloc(&slab, (size_t)nb * TILE));
    CK(hipMalloc(&out, (size_t)nb * TILE));
    CK(hipMalloc(&flags, nb * 32 * 4));
    CK(hipMemset(flags, 0, nb * 32 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned epoch = 0;
    const char* names[3] = {"no hand-over (256 tiles written)", "partner on the same XCD (b, b + 8)", "partner on the next XCD"};
    for (int it : {0, iters}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 7; ++rep) {
                CK(hipEventRecord(e0));
                for (int k = 0; k < 10; ++k) {
                    ++epoch;
                    hipLaunchKernelGGL(handoff_kernel, dim3(nb), dim3(256), 0, 0, slab, out, flags, mode, it, epoch);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms / 10 < best ? ms / 10 : best;
            }
            printf("mfma iters %5d  %-40s %7.2f us per kernel\n", it, names[mode], best * 1e3f);
        }
    }
    return 0;
}
