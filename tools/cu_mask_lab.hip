// cu_mask_lab.hip -- standalone gfx950 lab (round 6): what does a CU-masked stream (hipExtStreamCreateWithCUMask) buy two
// SMALL decode groups that today either share one latency-bound launch chain or trample each other's CUs?
// (not part of the product library; build: hipcc --offload-arch=gfx950 -O3 tools/cu_mask_lab.hip -o gpurun_out/cu_mask_lab)
//
//   1. census    which (XCC, SE, CU) does mask bit i enable?  (the bit -> CU map is not documented for 8-XCD parts)
//   2. graph     does a hipGraph captured from / replayed into a masked stream stay inside the mask?
//   3. pull      HBM read rate of a pure streaming kernel on: the whole chip; half the CUs of every XCD; four whole XCDs;
//                and of TWO such kernels at once on complementary halves (each reads its own buffer)
//   4. chain     us per kernel of a graph-replayed chain of DEPENDENT small streaming kernels (the decode regime: ~3-13 MB
//                per launch, cold), one chain vs two chains at once, unmasked vs complementary masks
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <set>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// hwreg(id, offset, size) immediate of s_getreg_b32: id | offset << 6 | (size - 1) << 11
#define HWREG(id, off, sz) ((id) | ((off) << 6) | (((sz)-1) << 11))

__global__ __launch_bounds__(1024) void census_kernel(unsigned *out, int spin_cycles) {
    extern __shared__ char pad[];   // large dynamic LDS: one workgroup per CU
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(HWREG(4, 0, 32));    // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg(HWREG(20, 0, 32));  // HW_REG_XCC_ID
        out[blockIdx.x] = ((xcc & 0xfu) << 16) | ((hw >> 8) & 0xffu);      // cu_id[3:0], sh_id, se_id[2:0]
    }
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin_cycles) __builtin_amdgcn_s_sleep(8);
    if (pad[threadIdx.x] == 77) out[0] = 0;
}

// pure pull: workgroup w of G streams the 16-KiB pieces w, w + G, ... of [bytes]; 4 loads in flight per lane
__global__ __launch_bounds__(256) void pull_kernel(const u32x4 *__restrict__ src, size_t n16, unsigned *sink) {
    u32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 1024;   // 1024 x 16 B = 16 KiB per workgroup pass (4 x 256 lanes)
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i + 768 < n16; i += stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256);
        const u32x4 c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
        acc ^= a ^ b ^ c ^ d;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// chain link: depends on the previous link through `dep` (read at the start, written at the end), pulls `n16` x 16 B
__global__ __launch_bounds__(256) void link_kernel(const u32x4 *__restrict__ src, size_t n16, unsigned *dep, unsigned *sink) {
    const unsigned d0 = __hip_atomic_load(dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32x4 acc = {d0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) acc ^= __builtin_nontemporal_load(src + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(dep, d0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Mask {
    uint32_t w[8];
    int count() const {
        int n = 0;
        for (int i = 0; i < 8; ++i) n += __builtin_popcount(w[i]);
        return n;
    }
};
static Mask mask_none() { Mask m; memset(&m, 0, sizeof(m)); return m; }
static Mask mask_full() { Mask m; memset(&m, 0xff, sizeof(m)); return m; }
static Mask mask_bits(int lo, int hi) {   // bits lo .. hi - 1
    Mask m = mask_none();
    for (int i = lo; i < hi; ++i) m.w[i / 32] |= 1u << (i % 32);
    return m;
}
static Mask mask_mod8(int xlo, int xhi) {   // bits i with xlo <= i % 8 < xhi
    Mask m = mask_none();
    for (int i = 0; i < 256; ++i)
        if (i % 8 >= xlo && i % 8 < xhi) m.w[i / 32] |= 1u << (i % 32);
    return m;
}
static hipStream_t masked_stream(const Mask *m) {
    hipStream_t s;
    if (!m) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else CK(hipExtStreamCreateWithCUMask(&s, 8, m->w));
    return s;
}

static std::set<unsigned> run_census(hipStream_t s, unsigned *d_out, int n_wg) {
    CK(hipMemsetAsync(d_out, 0xff, (size_t)n_wg * 4, s));
    census_kernel<<<n_wg, 1024, 96 * 1024, s>>>(d_out, 20000);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(n_wg);
    CK(hipMemcpy(h.data(), d_out, (size_t)n_wg * 4, hipMemcpyDeviceToHost));
    return std::set<unsigned>(h.begin(), h.end());
}
static void print_set(const char *name, const std::set<unsigned> &st) {
    int per_xcc[16] = {0};
    for (unsigned v : st) per_xcc[(v >> 16) & 15]++;
    printf("%-34s %3zu CUs; per XCC:", name, st.size());
    for (int x = 0; x < 8; ++x) printf(" %2d", per_xcc[x]);
    printf("\n");
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const bool skip_census = argc > 1 && !strcmp(argv[1], "nocensus");
    CK(hipSetDevice(0));
    CK(hipFuncSetAttribute((const void *)census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    unsigned *d_out, *d_sink;
    CK(hipMalloc(&d_out, 4096 * 4));
    CK(hipMalloc(&d_sink, 64));
    CK(hipMemset(d_sink, 0, 64));

    // ------------------------------------------------------------------ 1. census
    printf("== 1. census: which CUs does a mask enable (1024 one-per-CU workgroups, (XCC, SE/SH/CU id) recorded)\n");
    {
        hipStream_t s = masked_stream(nullptr);
        print_set("no mask", run_census(s, d_out, 1024));
        CK(hipStreamDestroy(s));
    }
    struct Named { const char *name; Mask m; };
    const Named sets[] = {{"bits 0..127   (words 0-3)", mask_bits(0, 128)}, {"bits 128..255 (words 4-7)", mask_bits(128, 256)},
                          {"bits i%8 < 4  (0x0f bytes)", mask_mod8(0, 4)},   {"bits i%8 >= 4 (0xf0 bytes)", mask_mod8(4, 8)},
                          {"bits 0..84", mask_bits(0, 85)},                  {"bits 0..7", mask_bits(0, 8)},
                          {"bits 0..31", mask_bits(0, 32)}};
    for (const Named &n : sets) {
        hipStream_t s = masked_stream(&n.m);
        print_set(n.name, run_census(s, d_out, 1024));
        CK(hipStreamDestroy(s));
    }
    if (!skip_census) {
        printf("single bits: bit -> (xcc, se_sh_cu id)\n");
        int ok_mod8 = 0, n_single = 0;
        for (int b = 0; b < 256; b += (b < 32 ? 1 : 13)) {
            const Mask m = mask_bits(b, b + 1);
            hipStream_t s = masked_stream(&m);
            const std::set<unsigned> st = run_census(s, d_out, 64);
            printf("  bit %3d ->", b);
            for (unsigned v : st) {
                printf(" (xcc %u, id 0x%02x)", (v >> 16) & 15, v & 0xff);
                if (st.size() == 1) { ++n_single; ok_mod8 += (int)((v >> 16) & 15) == b % 8; }
            }
            printf("\n");
            CK(hipStreamDestroy(s));
        }
        printf("single-bit masks whose XCC == bit %% 8: %d of %d\n", ok_mod8, n_single);
    }

    // ------------------------------------------------------------------ 2. graph replay inside the mask?
    printf("== 2. graph captured on / replayed into a masked stream\n");
    {
        const Mask m = mask_bits(0, 128);
        hipStream_t s = masked_stream(&m), s_plain = masked_stream(nullptr);
        const std::set<unsigned> direct = run_census(s, d_out, 1024);
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        census_kernel<<<1024, 1024, 96 * 1024, s>>>(d_out, 20000);
        census_kernel<<<1024, 1024, 96 * 1024, s>>>(d_out, 20000);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int which = 0; which < 2; ++which) {
            hipStream_t ls = which ? s_plain : s;
            CK(hipGraphLaunch(ge, ls));
            CK(hipStreamSynchronize(ls));
            std::vector<unsigned> h(1024);
            CK(hipMemcpy(h.data(), d_out, 4096, hipMemcpyDeviceToHost));
            std::set<unsigned> st(h.begin(), h.end());
            size_t inside = 0;
            for (unsigned v : st) inside += direct.count(v);
            printf("  graph captured on the masked stream, launched into the %s stream: %zu CUs, %zu of them inside the mask's %zu\n",
                   which ? "UNMASKED" : "masked", st.size(), inside, direct.size());
        }
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        CK(hipStreamDestroy(s));
        CK(hipStreamDestroy(s_plain));
    }

    // ------------------------------------------------------------------ 3. pull rate
    printf("== 3. HBM pull rate (2 GiB per kernel, nt loads, 4 x 16 B in flight per lane; GB/s = bytes / event time)\n");
    const size_t BYTES = (size_t)2 << 30;
    u32x4 *bufA, *bufB;
    CK(hipMalloc(&bufA, BYTES));
    CK(hipMalloc(&bufB, BYTES));
    CK(hipMemset(bufA, 1, BYTES));
    CK(hipMemset(bufB, 2, BYTES));
    hipEvent_t e0, e1, e2, e3;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
    auto pull_one = [&](const char *name, const Mask *m, int wg_per_cu) {
        hipStream_t s = masked_stream(m);
        const int cus = m ? m->count() : 256;
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, s));
            pull_kernel<<<cus * wg_per_cu, 256, 0, s>>>(bufA, BYTES / 16, d_sink);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        printf("  %-40s %3d CUs x %d WG: %7.1f GB/s\n", name, cus, wg_per_cu, BYTES / 1e6 / best);
        CK(hipStreamDestroy(s));
    };
    auto pull_two = [&](const char *name, const Mask *ma, const Mask *mb, int wg_per_cu) {
        hipStream_t sa = masked_stream(ma), sb = masked_stream(mb);
        const int ca = ma ? ma->count() : 256, cb = mb ? mb->count() : 256;
        double best = 1e30;
        for (int r = 0; r < 4; ++r) {
            CK(hipDeviceSynchronize());
            const double t0 = now_us();
            pull_kernel<<<ca * wg_per_cu, 256, 0, sa>>>(bufA, BYTES / 16, d_sink);
            pull_kernel<<<cb * wg_per_cu, 256, 0, sb>>>(bufB, BYTES / 16, d_sink);
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sb));
            const double dt = now_us() - t0;
            if (r && dt < best) best = dt;
        }
        printf("  %-40s 2 kernels at once: %7.1f GB/s total (host clock)\n", name, 2.0 * BYTES / 1e3 / best);
        CK(hipStreamDestroy(sa));
        CK(hipStreamDestroy(sb));
    };
    const Mask lo = mask_bits(0, 128), hi = mask_bits(128, 256), xl = mask_mod8(0, 4), xh = mask_mod8(4, 8);
    const Mask t0m = mask_bits(0, 88), t1m = mask_bits(88, 176), t2m = mask_bits(176, 256);
    for (int w : {2, 4, 8}) {
        pull_one("whole chip", nullptr, w);
        pull_one("half the CUs of every XCD", &lo, w);
        pull_one("four whole XCDs", &xl, w);
        pull_one("a third (bits 0..87)", &t0m, w);
    }
    pull_two("unmasked + unmasked", nullptr, nullptr, 4);
    pull_two("half of every XCD + the other half", &lo, &hi, 4);
    pull_two("XCDs 0-3 + XCDs 4-7", &xl, &xh, 4);

    // ------------------------------------------------------------------ 4. dependent chains
    printf("== 4. chains of DEPENDENT streaming kernels, graph-replayed (us per link; links rotate over 512 MiB so every read is cold)\n");
    unsigned *depA, *depB;
    CK(hipMalloc(&depA, 64)); CK(hipMalloc(&depB, 64));
    CK(hipMemset(depA, 0, 64)); CK(hipMemset(depB, 0, 64));
    const int LINKS = 256;
    auto build_chain = [&](hipStream_t s, const u32x4 *buf, size_t link_bytes, int wgs, unsigned *dep, hipGraphExec_t *ge) {
        hipGraph_t g;
        const size_t rot = ((size_t)512 << 20) / link_bytes;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < LINKS; ++i)
            link_kernel<<<wgs, 256, 0, s>>>(buf + (i % rot) * (link_bytes / 16), link_bytes / 16, dep, d_sink);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(ge, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    };
    auto chain_case = [&](const char *name, const Mask *ma, const Mask *mb, bool two, size_t link_bytes, int wg_a, int wg_b) {
        hipStream_t sa = masked_stream(ma), sb = masked_stream(mb);
        hipGraphExec_t ga, gb;
        build_chain(sa, bufA, link_bytes, wg_a, depA, &ga);
        build_chain(sb, bufB, link_bytes, wg_b, depB, &gb);
        double best = 1e30;
        for (int r = 0; r < 4; ++r) {
            CK(hipDeviceSynchronize());
            const double t0 = now_us();
            CK(hipGraphLaunch(ga, sa));
            if (two) CK(hipGraphLaunch(gb, sb));
            CK(hipGraphLaunch(ga, sa));
            if (two) CK(hipGraphLaunch(gb, sb));
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sb));
            const double dt = now_us() - t0;
            if (r && dt < best) best = dt;
        }
        const double per = best / (2.0 * LINKS);
        printf("  %-46s link %5.1f MB x %3d/%3d WG: %6.2f us per link%s  -> %7.1f GB/s total\n", name, link_bytes / 1e6, wg_a, wg_b, per,
               two ? " (both chains)" : "              ", (two ? 2.0 : 1.0) * link_bytes / 1e3 / per);
        CK(hipGraphExecDestroy(ga)); CK(hipGraphExecDestroy(gb));
        CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
    };
    for (size_t mb : {(size_t)0, (size_t)3, (size_t)13, (size_t)64}) {
        const size_t lb = mb ? (mb << 20) : 65536;
        chain_case("one chain, whole chip", nullptr, nullptr, false, lb, 256, 256);
        chain_case("one chain, half of every XCD", &lo, &lo, false, lb, 128, 128);
        chain_case("one chain, half of every XCD, 256 WGs", &lo, &lo, false, lb, 256, 256);
        chain_case("one chain, four whole XCDs", &xl, &xl, false, lb, 128, 128);
        chain_case("two chains, both unmasked", nullptr, nullptr, true, lb, 256, 256);
        chain_case("two chains, both unmasked, 128 WGs each", nullptr, nullptr, true, lb, 128, 128);
        chain_case("two chains, half of every XCD each", &lo, &hi, true, lb, 128, 128);
        chain_case("two chains, half of every XCD each, 256 WGs", &lo, &hi, true, lb, 256, 256);
        chain_case("two chains, four whole XCDs each", &xl, &xh, true, lb, 128, 128);
        chain_case("two chains, four whole XCDs each, 256 WGs", &xl, &xh, true, lb, 256, 256);
    }
    printf("done\n");
    return 0;
}
