#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
// LDS tile: 64 rows x 64 cols of int16, value = row*64+col, row stride 144 bytes.
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) char lds[64*144];
  for (int i = threadIdx.x; i < 64*64; i += 64) { int r=i/64,c=i%64; *(short*)(lds + r*144 + c*2) = (short)(r*64+c); }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  // group g: rows 8*g .. 8*g+3, cols 16*(g&1) .. +15  (row base distinct per group to see grouping)
  const int rowbase = 8*g, colbase = 16*(g&1);
  const char* addr = lds + (rowbase + q)*144 + (colbase + 4*p)*2;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
  for (int j=0;j<4;++j) out[lane*4+j] = v[j];
}
int main(){ short* d; hipMalloc(&d, 64*4*2); k<<<1,64>>>(d); short h[256]; hipMemcpy(h,d,512,hipMemcpyDeviceToHost);
  for (int l=0;l<64;++l){ printf("lane %2d:",l); for(int j=0;j<4;++j) printf(" (%d,%d)", h[l*4+j]/64, h[l*4+j]%64); printf("\n"); } return 0; }
