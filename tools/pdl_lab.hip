// pdl_lab.hip -- standalone gfx950 lab (round 5): can a chain of DEPENDENT small-batch GEMVs hide its kernel boundaries by
// launching kernel n + 1 while kernel n still runs ("programmatic dependent launch", emulated)?
// (not part of the product library; build: hipcc --offload-arch=gfx950 -O3 tools/pdl_lab.hip -o gpurun_out/pdl_lab)
//
// A decoder position at <= 16 rows is ~226 dependent launches of 4-7 us; ~1.55 us of each is the kernel boundary of an in-order
// queue and ~1 us the first memory round trip of a kernel whose weights do not depend on its predecessor at all.  Mode B puts
// consecutive kernels on TWO alternating streams of one captured graph (same-stream order = the edge K(n-1) -> K(n+1), so at most
// two kernels are in flight), and replaces the edge K(n) -> K(n+1) by a flag: K(n+1) requests its weights, THEN waits until all
// workgroups of K(n) have released their outputs (agent-scope release / acquire), then loads its activations.
//   mode A: one stream, plain graph chain (the product's structure)
//   mode B: two streams + flags
// The kernel is the product's small-batch GEMV in miniature: 16 x 16 x 32 bf16 MFMA, fragment-tiled weights and activations,
// K split over the waves of a workgroup, LDS reduction, bf16 activations of the next kernel written in its tiled order.
// Reported: us per kernel of a graph-replayed chain over rotating weight matrices, and whether B's final activations are
// bit-identical to A's (a stale read would show).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

struct ChainArgs {
    const bf16_t *W;     // [N/16 tiles][K/32 steps][64 lanes][8]
    const bf16_t *x;     // [K/32 steps][64][8]  (one block of 16 rows)
    bf16_t *y;           // [N/32 steps][64][8]
    int K, N;
    unsigned *wait_flag;   // PDL: counter the PREVIOUS kernel's workgroups release into (null: first kernel / mode A)
    unsigned wait_count;
    unsigned *signal_flag; // PDL: this kernel's counter
    unsigned *reset_flag;  // PDL: the counter of the kernel before the previous one (reset by workgroup 0 after its wait)
    int *err;
};

// SPW k-steps per wave; waves = K / 32 / SPW
template <int SPW, bool PDL>
__global__ __launch_bounds__(512) void chain_gemv(ChainArgs p) {
    extern __shared__ float red[];   // [waves][64][4]
    const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (p.K >> 5) / SPW;
    // weights first: they do not depend on the previous kernel
    u32x4 wf[SPW];
    const bf16_t *wp = p.W + (((long)tile * (p.K >> 5) + (long)wave * SPW) * 64 + lane) * 8;
#pragma unroll
    for (int u = 0; u < SPW; ++u) wf[u] = *(const u32x4 *)(wp + u * 512);
    if (PDL && p.wait_flag) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            // relaxed polls, ONE acquire afterwards: an acquire per poll would invalidate the L2's non-local lines under the
            // feet of the kernel that is still running
            while (__hip_atomic_load(p.wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.wait_count) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 100000ull) { *p.err = 1; break; }   // 1 ms: give up, flag the run
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (blockIdx.x == 0 && p.reset_flag) __hip_atomic_store(p.reset_flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    u32x4 af[SPW];
    const bf16_t *ap = p.x + (((long)wave * SPW) * 64 + lane) * 8;
#pragma unroll
    for (int u = 0; u < SPW; ++u) af[u] = *(const u32x4 *)(ap + u * 512);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < SPW; ++u)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[u]), __builtin_bit_cast(bf16x8, wf[u]), acc, 0, 0, 0);
    *(f32x4 *)(red + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (wave == 0) {
        f32x4 s = *(const f32x4 *)(red + lane * 4);
        for (int w = 1; w < nw; ++w) s += *(const f32x4 *)(red + (w * 64 + lane) * 4);
        // lane holds column n = tile * 16 + (lane & 15), rows (lane >> 4) * 4 + r: write the next kernel's A fragments
        const int n = tile * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = (lane >> 4) * 4 + r;
            const float v = s[r] * 0.03125f + 0.01f * (float)((b + n) & 7);   // keep the chain's values bounded and alive
            const long o = (((long)(n >> 5) * 64) + (b + 16 * ((n & 31) >> 3))) * 8 + (n & 7);
            p.y[o] = f2bf(v);
        }
    }
    if (PDL && p.signal_flag) {
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(p.signal_flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void zero_flags(unsigned *f) { if (threadIdx.x < 4) f[threadIdx.x] = 0u; }
__global__ void trivial_kernel(float *p) { if (threadIdx.x == 999) p[0] = 1.f; }

static float replay(hipGraphExec_t ge, hipStream_t s, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best;
}

template <int SPW>
static void run(const char *name, int d, int n_kernels, int n_mats) {
    const int K = d, N = d, nw = K / 32 / SPW, tiles = N / 16;
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    bf16_t *W, *x[3];
    const size_t wel = (size_t)N * K;
    CK(hipMalloc((void **)&W, wel * 2 * n_mats));
    std::vector<bf16_t> hw(wel * n_mats);
    uint32_t st = 12345u;
    for (auto &v : hw) {
        st = st * 1664525u + 1013904223u;
        const float f = ((int)(st >> 16) % 2001 - 1000) * 1e-3f * 0.06f;
        unsigned u;
        memcpy(&u, &f, 4);
        v = (bf16_t)(u >> 16);
    }
    CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const size_t xel = (size_t)16 * K;
    std::vector<bf16_t> hx(xel);
    for (size_t i = 0; i < xel; ++i) {
        const float f = 0.25f + 0.001f * (float)(i % 97);
        unsigned u;
        memcpy(&u, &f, 4);
        hx[i] = (bf16_t)(u >> 16);
    }
    for (auto &p : x) CK(hipMalloc((void **)&p, xel * 2));
    unsigned *flags;
    int *err;
    CK(hipMalloc((void **)&flags, 64));
    CK(hipMalloc((void **)&err, 4));
    CK(hipMemset(flags, 0, 64));
    CK(hipMemset(err, 0, 4));
    const size_t lds = (size_t)nw * 1024;
    std::vector<bf16_t> outA(xel), outB(xel);
    float usA = 0.f, usB = 0.f;
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemcpy(x[0], hx.data(), xel * 2, hipMemcpyHostToDevice));
        hipGraph_t g;
        hipGraphExec_t ge;
        hipEvent_t fork, join, evA, evB;
        CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&evA, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&evB, hipEventDisableTiming));
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
        if (mode == 1) {
            CK(hipEventRecord(fork, sa));
            CK(hipStreamWaitEvent(sb, fork, 0));
        }
        for (int i = 0; i < n_kernels; ++i) {
            ChainArgs a;
            a.W = W + (size_t)(i % n_mats) * wel;
            a.x = x[i % 3];
            a.y = x[(i + 1) % 3];
            a.K = K; a.N = N; a.err = err;
            a.wait_flag = (mode == 1 && i > 0) ? flags + ((i - 1) % 3) : nullptr;
            a.wait_count = (unsigned)tiles;
            a.signal_flag = mode == 1 ? flags + (i % 3) : nullptr;
            a.reset_flag = (mode == 1 && i > 1) ? flags + ((i - 2) % 3) : nullptr;
            hipStream_t s = (mode == 1 && (i & 1)) ? sb : sa;
            if (mode == 1) chain_gemv<SPW, true><<<tiles, nw * 64, lds, s>>>(a);
            else chain_gemv<SPW, false><<<tiles, nw * 64, lds, s>>>(a);
        }
        if (mode == 1) {
            CK(hipEventRecord(join, sb));
            CK(hipStreamWaitEvent(sa, join, 0));
            zero_flags<<<1, 64, 0, sa>>>(flags);
        }
        CK(hipStreamEndCapture(sa, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        // correctness pass from the same start
        CK(hipGraphLaunch(ge, sa));
        CK(hipStreamSynchronize(sa));
        CK(hipMemcpy((mode ? outB : outA).data(), x[n_kernels % 3], xel * 2, hipMemcpyDeviceToHost));
        const float ms = replay(ge, sa);
        (mode ? usB : usA) = ms * 1e3f / n_kernels;
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    int herr = 0;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(outA.data(), outB.data(), xel * 2) == 0;
    double cs = 0;
    for (size_t i = 0; i < xel; ++i) {
        unsigned u = (unsigned)outA[i] << 16;
        float f;
        memcpy(&f, &u, 4);
        cs += f;
    }
    printf("%-28s d=%4d %3d workgroups x %2d waves, %3d kernels over %3d matrices: plain %.2f us/kernel   two streams + flags %.2f us/kernel   "
           "bit-identical %s  timeouts %s  (checksum %.4f)\n",
           name, d, tiles, nw, n_kernels, n_mats, usA, usB, same ? "yes" : "NO", herr ? "YES" : "no", cs);
    fflush(stdout);
    CK(hipFree(W));
    for (auto &p : x) CK(hipFree(p));
    CK(hipFree(flags));
    CK(hipFree(err));
    CK(hipStreamDestroy(sa));
    CK(hipStreamDestroy(sb));
}

int main() {
    {   // launch floors: a chain of trivial kernels on one stream / alternating on two streams of one graph
        hipStream_t sa, sb;
        CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        float *p;
        CK(hipMalloc((void **)&p, 64));
        for (int mode = 0; mode < 2; ++mode) {
            hipGraph_t g;
            hipGraphExec_t ge;
            hipEvent_t fork, join;
            CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
            CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
            if (mode) { CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0)); }
            for (int i = 0; i < 256; ++i) trivial_kernel<<<80, 512, 0, (mode && (i & 1)) ? sb : sa>>>(p);
            if (mode) { CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0)); }
            CK(hipStreamEndCapture(sa, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            printf("trivial kernels (80 x 512 threads), graph of 256: %s %.2f us/kernel\n", mode ? "two alternating streams" : "one stream", replay(ge, sa) * 1e3f / 256);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
        // eager launches, with and without hipExtAnyOrderLaunch (no barrier bit between the packets of one queue; hip_ext.h says
        // "not supported on GFX9xx": measured here)
        for (int any = 0; any < 2; ++any) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipStreamSynchronize(sa));
                CK(hipEventRecord(e0, sa));
                for (int i = 0; i < 2000; ++i)
                    hipExtLaunchKernelGGL(trivial_kernel, dim3(80), dim3(512), 0, sa, nullptr, nullptr, any ? hipExtAnyOrderLaunch : 0, p);
                CK(hipEventRecord(e1, sa));
                CK(hipStreamSynchronize(sa));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("trivial kernels, 2000 eager launches on one stream, %s: %.2f us/kernel\n", any ? "hipExtAnyOrderLaunch" : "in order", best * 1e3f / 2000);
        }
        CK(hipFree(p));
    }
    run<5>("large-v2 d x d (attn-out)", 1280, 256, 120);
    run<5>("large-v2 d x d, L2-resident", 1280, 256, 2);
    run<4>("medium d x d", 1024, 256, 120);
    run<6>("small d x d", 768, 256, 120);
    run<6>("tiny d x d (384)", 384, 256, 120);
    return 0;
}
