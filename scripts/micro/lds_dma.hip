// Probe: buffer_load_dwordx4 ... lds on gfx950 - where does lane l's 16 bytes land, do out-of-range lanes write
// zeros, and does an instruction offset move the LDS destination? (scripts/micro, built by `make micro`)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* dst, int n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * 4, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __attribute__((address_space(3))) char* base =
        (__attribute__((address_space(3))) char*)smem + wave * 2048;
    // instruction 0: rows of 16 B per lane, second instruction 1 KB further in memory and in LDS
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, base, 16, (wave * 128 + lane) * 16, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, base + 1024, 16, (wave * 128 + 64 + lane) * 16, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) dst[i] = reinterpret_cast<float*>(smem)[i];
}
int main() {
    const int n = 1000;            // floats in range: the last 24 of the 1024 read are out of range
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, 4096); hipMalloc(&d, 4096);
    hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 4096, 0, s, d, n);
    std::vector<float> o(1024);
    hipMemcpy(o.data(), d, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) {
        const float want = i < n ? (float)i : 0.f;
        if (o[i] != want) { if (bad < 8) printf("lds[%d] = %g, want %g\n", i, o[i], want); ++bad; }
    }
    printf("lds_dma: %d mismatches (lane l -> base + inst offset + 16 l; out-of-range lanes write 0)\n", bad);
    return bad != 0;
}
