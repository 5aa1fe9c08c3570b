// Micro-benchmark (GPU box): (1) which XCD a workgroup runs on (HW_REG_XCC_ID)
// against blockIdx.x % 8; (2) the round-trip latency of an 8-byte {tag, value}
// granule between two workgroups - agent scope (sc1: what pm_fargan.h's
// exchanges use, coherent across XCDs) against an L2-scope exchange between
// workgroups ON THE SAME XCD (plain store: the vector L1 writes through to the
// XCD's L2; sc0 load: bypasses the reader's L1, served by that L2).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/xcd_exchange.hip -o promonet_amd/lib/xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__global__ void where_kernel(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// MODE 0: agent scope (sc1 store / sc1 load); 1: plain store / sc0 load;
// 2: plain store / sc0 sc1 load; 3: plain store, plain load behind buffer_inv sc1
template <int MODE>
__device__ __forceinline__ void put(unsigned long long* p, unsigned long long v) {
    if constexpr (MODE == 0)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)"
                     :: "v"(p), "v"(v) : "memory");
}
template <int MODE>
__device__ __forceinline__ unsigned long long get(unsigned long long* p) {
    unsigned long long v;
    if constexpr (MODE == 0) {
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (MODE == 1) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    } else if constexpr (MODE == 2) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    } else {
        asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    }
    return v;
}

// pairs (a, b): workgroup `a` and `b` bounce a counter `rounds` times through
// two granules; out[pair] = cycles per round trip, or 0 when a bounded spin
// gave up (the partner's store never became visible)
template <int MODE>
__global__ void pingpong_kernel(unsigned long long* cells, const int* partner,
                                unsigned long long* cycles, unsigned* xcc,
                                int rounds) {
    const int me = blockIdx.x, other = partner[me];
    if (threadIdx.x != 0 || other < 0) return;
    xcc[me] = xcc_id();
    const bool first = me < other;
    const int pair = first ? me : other;
    unsigned long long* mine = cells + 2 * pair * 16 + (first ? 0 : 16);
    unsigned long long* theirs = cells + 2 * pair * 16 + (first ? 16 : 0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bool ok = true;
    for (int r = 1; r <= rounds && ok; ++r) {
        if (first) put<MODE>(mine, (unsigned long long)r);
        unsigned spins = 0;
        while (get<MODE>(theirs) != (unsigned long long)r) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) { ok = false; break; }
        }
        if (!first && ok) put<MODE>(mine, (unsigned long long)r);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (first) cycles[pair] = ok ? (t1 - t0) / rounds : 0ull;
}

int main() {
    const int G = 64;
    unsigned* where; hipMalloc(&where, G * 4);
    where_kernel<<<G, 64>>>(where);
    std::vector<unsigned> h(G);
    hipMemcpy(h.data(), where, G * 4, hipMemcpyDeviceToHost);
    int agree = 0;
    printf("XCC_ID of workgroups 0..%d:", G - 1);
    for (int i = 0; i < G; ++i) { printf(" %u", h[i]); agree += (int)h[i] == i % 8; }
    printf("\n%d of %d equal blockIdx %% 8\n", agree, G);
    // a second launch right behind an odd-sized one: does the round robin restart?
    where_kernel<<<3, 64>>>(where);
    where_kernel<<<G, 64>>>(where);
    hipMemcpy(h.data(), where, G * 4, hipMemcpyDeviceToHost);
    printf("after a 3-workgroup launch:");
    for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
    printf("\n");

    unsigned long long* cells; hipMalloc(&cells, G * 2 * 16 * 8);
    int* partner; hipMalloc(&partner, G * 4);
    unsigned long long* cycles; hipMalloc(&cycles, G * 8);
    unsigned* xcc; hipMalloc(&xcc, G * 4);
    const char* names[4] = {"agent scope (sc1 store, sc1 load)",
                            "L2 scope (plain store, sc0 load)",
                            "plain store, sc0 sc1 load",
                            "plain store, buffer_inv sc1 + plain load"};
    for (int same = 1; same >= 0; --same) {
        // 16 workgroups: same XCD -> pairs (i, i + 8); different -> (2i, 2i + 1)
        std::vector<int> p(G, -1);
        for (int i = 0; i < 8; ++i) {
            if (same) { p[i] = i + 8; p[i + 8] = i; }
            else { p[2 * i] = 2 * i + 1; p[2 * i + 1] = 2 * i; }
        }
        hipMemcpy(partner, p.data(), G * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 4; ++mode) {
            hipMemset(cells, 0, G * 2 * 16 * 8);
            hipMemset(cycles, 0, G * 8);
            const int rounds = 2000;
            switch (mode) {
                case 0: pingpong_kernel<0><<<16, 64>>>(cells, partner, cycles, xcc, rounds); break;
                case 1: pingpong_kernel<1><<<16, 64>>>(cells, partner, cycles, xcc, rounds); break;
                case 2: pingpong_kernel<2><<<16, 64>>>(cells, partner, cycles, xcc, rounds); break;
                case 3: pingpong_kernel<3><<<16, 64>>>(cells, partner, cycles, xcc, rounds); break;
            }
            hipDeviceSynchronize();
            std::vector<unsigned long long> c(G);
            std::vector<unsigned> x(G);
            hipMemcpy(c.data(), cycles, G * 8, hipMemcpyDeviceToHost);
            hipMemcpy(x.data(), xcc, G * 4, hipMemcpyDeviceToHost);
            printf("%s, partners on %s XCD: s_memtime ticks per round trip:",
                   names[mode], same ? "the SAME" : "DIFFERENT");
            for (int i = 0; i < 16; ++i) {
                const int first = same ? (i < 8 ? i : -1) : (i % 2 == 0 ? i : -1);
                if (first >= 0)
                    printf(" %llu(x%u,%u)", c[first], x[first], x[p[first]]);
            }
            printf("\n");
        }
    }

    return 0;
}
