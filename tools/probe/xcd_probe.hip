// which XCD does workgroup i run on?  (HW_REG_XCC_ID, gfx940+: id 20, bits 3:0)
#include <hip/hip_runtime.h>
__global__ void xcd_probe_kernel(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15;
}
extern "C" int xcd_probe(int* out, int gx, int gy, int gz, int threads, void* stream) {
    xcd_probe_kernel<<<dim3(gx, gy, gz), threads, 0, (hipStream_t)stream>>>(out);
    return (int)hipGetLastError();
}
