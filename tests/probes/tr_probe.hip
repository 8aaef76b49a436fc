// probe of ds_read_b64_tr_b16 lane semantics on gfx950 (test infrastructure; prints raw results for 3 address patterns)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(unsigned short* out, int pattern) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    int l = threadIdx.x;
    int elem;
    if (pattern == 0) elem = l * 4;                                                  // canonical contiguous
    else if (pattern == 1) elem = (l >> 4) * 64 + (((l & 3) << 2) | ((l >> 2) & 3)) * 4;   // permuted inside each 16-lane group
    else elem = (l & 15) * 100 + (l >> 4) * 2000;                                    // arbitrary 8B-aligned strides (100 elems = 200 B)
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int p = 0; p < 3; ++p) {
        k<<<1, 64>>>(d, p);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; ++l) printf("%d: %d %d %d %d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
