// Does a raw buffer load's SCALAR offset take part in the descriptor's range check on this GPU?
// (crf_band.hip's row loads of a block rely on it: a row past the end of the score tensor must read 0, not memory.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/bufrange_probe tools/bufrange_probe.hip && tools/bufrange_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const float *p, float *out) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 64, 0x00027000);
    const unsigned lane4 = 4u * threadIdx.x;
    out[threadIdx.x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, lane4, 0, 0));            // in range for lanes 0..15
    out[64 + threadIdx.x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 0, 128, 0));         // scalar offset past the end
    out[128 + threadIdx.x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 128, 0, 0));        // vector offset past the end
    out[192 + threadIdx.x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, lane4, 32, 0));     // mixed: in range iff lane4 + 32 < 64
}
int main() {
    float *d, *o, h[4096], r[256];
    for (int i = 0; i < 4096; ++i) h[i] = 1.0f + i;
    hipMalloc(&d, sizeof(h));
    hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("num_records = 64 bytes (16 floats); memory holds 1 + index\n");
    printf("voffset lane*4, soffset 0:   lane 0 %g lane 15 %g lane 16 %g (expect 1, 16, 0)\n", r[0], r[15], r[16]);
    printf("voffset 0, soffset 128:      %g  (0: the scalar offset is range-checked; 33: it is not)\n", r[64]);
    printf("voffset 128, soffset 0:      %g  (expect 0)\n", r[128]);
    printf("voffset lane*4, soffset 32:  lane 7 %g lane 8 %g (checked: 16, 0)\n", r[192 + 7], r[192 + 8]);
    return 0;
}
