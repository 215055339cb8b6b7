#include <hip/hip_runtime.h>
__global__ void k(const float* src, float* dst, int n) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* base = lds + wv * 256;   // 1 KB per wave
    // each lane loads 16 bytes from src + (blockIdx.x*blockDim.x + threadIdx.x)*4 floats
    const float* g = src + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)base, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0)
    __syncthreads();
    float4 v = *reinterpret_cast<float4*>(base + lane * 4);
    *reinterpret_cast<float4*>(dst + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4) = v;
}
int main() {
    const int n = 1024 * 4;
    float *h = new float[n], *s, *d;
    for (int i = 0; i < n; ++i) h[i] = i;
    hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
    hipMemcpy(s, h, n * 4, hipMemcpyHostToDevice);
    k<<<4, 256, 4096>>>(s, d, n);
    float *o = new float[n];
    hipMemcpy(o, d, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) if (o[i] != h[i]) ++bad;
    printf("bad %d of %d (o[5]=%f o[1030]=%f)\n", bad, n, o[5], o[1030]);
    return bad != 0;
}
