// Is the SCALAR offset of a raw buffer access part of the descriptor's bounds check on gfx950?  (tools only)
// A buffer of 1024 bytes inside a larger allocation; loads / stores at vector offset 0 + scalar offset 2048, and at vector offset 2048.
#include <cstdio>
#include <hip/hip_runtime.h>
__global__ void k(float* base, float* out) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 1024, 0x00020000);
    int so = 2048; asm volatile("" : "+s"(so));
    int vo = 2048; asm volatile("" : "+v"(vo));
    out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 0, so, 0));     // beyond num_records through the scalar offset
    out[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, 0, 0));     // beyond through the vector offset
    out[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0));      // in range
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 7.0f), rs, 4, so, 0);    // store beyond through the scalar offset
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 9.0f), rs, vo + 8, 0, 0); // store beyond through the vector offset
}
int main() {
    float *d, *o; hipMalloc(&d, 8192); hipMalloc(&o, 64);
    float h[2048]; for (int i = 0; i < 2048; ++i) h[i] = 100.f + i;
    hipMemcpy(d, h, 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
    float r[3]; hipMemcpy(r, o, 12, hipMemcpyDeviceToHost); hipMemcpy(h, d, 8192, hipMemcpyDeviceToHost);
    printf("load  voffset 0 + soffset 2048 (num_records 1024): %g  (612 = the data beyond, 0 = caught by the bounds check)\n", r[0]);
    printf("load  voffset 2048: %g   in range: %g\n", r[1], r[2]);
    printf("store voffset 4 + soffset 2048 wrote: %g (7 = not caught, 613 = caught);  store voffset 2056 wrote: %g (9 = not caught, 614 = caught)\n", h[513], h[514]);
    return 0;
}
