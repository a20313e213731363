// Does the raw-buffer range check on gfx950 include the SGPR soffset?  (decides how rows past the
// frame end may be addressed in sesrq_mfma.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* buf, int nrec_bytes, int soff) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, nrec_bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(0x1234 + threadIdx.x, rs, threadIdx.x * 4, soff, 0);
}
__global__ void kl(int* buf, int* out, int nrec_bytes, int soff) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, nrec_bytes, 0x00020000);
    out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x * 4, soff, 0);
}
int main() {
    int *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 4096); hipMemset(d, 0, 4096);
    // records = 256 bytes; voffset 0..252 in range; soffset = 1024 pushes the address outside the records
    k<<<1, 64>>>(d, 256, 1024);
    int h[1024]; hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    printf("store with in-range voffset + soffset beyond num_records: word at byte 1024 = 0x%x (%s)\n", h[256], h[256] ? "WRITTEN: soffset NOT range-checked" : "dropped: soffset is range-checked");
    hipMemset(d, 0x11, 4096);
    kl<<<1, 64>>>(d, o, 256, 1024);
    hipMemcpy(h, o, 256, hipMemcpyDeviceToHost);
    printf("load  same addressing: 0x%x (%s)\n", h[0], h[0] ? "READ" : "zero");
    // negative total via voffset wrap
    hipMemset(d, 0, 4096);
    k<<<1, 64>>>(d + 256, 256, -1024);
    hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    printf("store with soffset = -1024: word at base-1024 = 0x%x\n", h[0]);
    return 0;
}
