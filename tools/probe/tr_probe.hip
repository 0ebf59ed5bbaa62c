// probe of ds_read_b64_tr_b16 (gfx950): LDS holds lds[i] = i (u16); every lane passes its own byte address; the 4 returned elements are dumped
#include <hip/hip_runtime.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void tr_probe_kernel(unsigned short* out, const int* addr) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
extern "C" int tr_probe(unsigned short* out, const int* addr, void* stream) {
    tr_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out, addr);
    return (int)hipGetLastError();
}
