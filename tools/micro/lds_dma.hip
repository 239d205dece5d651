// micro-test: semantics of buffer_load_dwordx4 ... lds on gfx950 (the direct global -> LDS path of the conv kernels):
// lane L of a wave lands at M0 + 16 L?  out-of-range lanes (offset beyond num_records / bit 31 set) write zeros?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_dma.hip -o /tmp/lds_dma && /tmp/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4 * 256];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = 0xffffffffu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000u);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // wave w loads 1 KB: lanes in REVERSED order of global address (so the LDS order shows the lane order); lanes 60..63 out of range
    unsigned off = (unsigned)(wave * 1024 + (63 - lane) * 16);
    if (lane >= 62) off = 0x80000000u;
    if (lane == 60 || lane == 61) off = nbytes + 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + wave * 256), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = i;
    unsigned *src, *out;
    hipMalloc(&src, 4096); hipMalloc(&out, 4096);
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, src, 4096u, out);
    std::vector<unsigned> o(1024);
    hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                unsigned want = l >= 60 ? 0u : (unsigned)(w * 256 + (63 - l) * 4 + e);
                unsigned got = o[w * 256 + l * 4 + e];
                if (got != want && bad++ < 8) printf("wave %d lane %d elem %d: got %u want %u\n", w, l, e, got, want);
            }
    printf("lds dma: %s (%d mismatches)\n", bad ? "DIFFERENT from the assumed semantics" : "lane L -> M0 + 16 L, out-of-range lanes write zeros", bad);
    return 0;
}
