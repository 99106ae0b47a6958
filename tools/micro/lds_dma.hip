// micro-test: semantics of buffer_load ... lds with 16 bytes per lane on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* dst, int rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, rows * 1024, 0x00020000);
    // wave w loads rows w, w+4, ...: 1 KiB per row (64 lanes x 16 B) into lds[row * 256 floats]
    for (int row = wave; row < rows; row += 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + row * 256), 16,
                                             lane * 16, row * 1024, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);   // everything
    __syncthreads();
    for (int i = tid; i < rows * 256; i += blockDim.x) dst[i] = lds[i];
}
int main() {
    const int rows = 16;
    std::vector<float> h(rows * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *s, *d;
    (void)hipMalloc(&s, h.size() * 4); (void)hipMalloc(&d, h.size() * 4);
    (void)hipMemcpy(s, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(d, 0, h.size() * 4);
    k<<<1, 256, rows * 1024>>>(s, d, rows);
    std::vector<float> o(h.size());
    (void)hipMemcpy(o.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < o.size(); ++i) if (o[i] != h[i]) { if (bad < 8) printf("mismatch at %zu: %g\n", i, o[i]); ++bad; }
    printf("lds dma b128: %d mismatches of %zu (%s)\n", bad, o.size(), hipGetErrorString(hipGetLastError()));
    return 0;
}
