#include <hip/hip_runtime.h>
__global__ void k(unsigned* o) {
    unsigned x = threadIdx.x, y = 1000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main(){ unsigned* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); unsigned h[128]; hipMemcpy(h,d,512,hipMemcpyDeviceToHost); for(int i=0;i<128;i++) printf("%u ", h[i]); printf("\n"); }
