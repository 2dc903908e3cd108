// Hardware layout probe (gfx950): MFMA fragment maps, ds_read_tr16_b64 gather, buffer->LDS OOB.
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_layouts tools/probe_layouts.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ __bf16 f2bf(float f){ return (__bf16)f; }

// A: [16][32] row-major float, B: [32][16] row-major float (B[k][n]); C out [16][16]
__global__ void mfma16(const float* A, const float* B, float* C){
  int l = threadIdx.x;
  bf16x8 a, b;
  for(int j=0;j<8;j++){ a[j] = f2bf(A[(l&15)*32 + (l>>4)*8 + j]); b[j] = f2bf(B[((l>>4)*8+j)*16 + (l&15)]); }
  f32x4 acc = {0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,acc,0,0,0);
  for(int r=0;r<4;r++) C[((l>>4)*4+r)*16 + (l&15)] = acc[r];
}
// A: [32][16], B: [16][32] (B[k][n]); C [32][32]
__global__ void mfma32(const float* A, const float* B, float* C){
  int l = threadIdx.x;
  bf16x8 a, b;
  for(int j=0;j<8;j++){ a[j] = f2bf(A[(l&31)*16 + (l>>5)*8 + j]); b[j] = f2bf(B[((l>>5)*8+j)*32 + (l&31)]); }
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,acc,0,0,0);
  for(int r=0;r<16;r++) C[((r&3)+8*(r>>2)+4*(l>>5))*32 + (l&31)] = acc[r];
}
// tr16 read: LDS shorts = index; lane address = base + l*8 bytes
__global__ void trprobe(short* out){
  __shared__ __attribute__((aligned(16))) short lds[1024];
  int l = threadIdx.x;
  for(int i=l;i<1024;i+=64) lds[i] = (short)i;
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l*4));
  for(int j=0;j<4;j++) out[l*4+j] = t[j];
}
// tr16 read with strided rows: 16-lane group g reads a [4 rows][16 cols] block with row stride 64 shorts:
// lane p in group supplies address of row (p>>2), cols (p&3)*4 ; group g block starts at col g*16
__global__ void trprobe2(short* out){
  __shared__ __attribute__((aligned(16))) short lds[1024];
  int l = threadIdx.x;
  for(int i=l;i<1024;i+=64) lds[i] = (short)i;   // value = row*64 + col for [16][64]
  __syncthreads();
  int p = l&15, g = l>>4;
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (p>>2)*64 + g*16 + (p&3)*4));
  for(int j=0;j<4;j++) out[l*4+j] = t[j];
}
// buffer_load ... lds with OOB lanes: does LDS get zeros?
__global__ void oobprobe(const float* src, int valid_bytes, float* out){
  __shared__ __attribute__((aligned(16))) float lds[256];
  int l = threadIdx.x;
  for(int i=l;i<256;i+=64) lds[i] = -7.0f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, valid_bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(lds), 16, l*16, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for(int i=l;i<256;i+=64) out[i] = lds[i];
  // also register path
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, l*16, 0, 0));
  out[256 + l] = v[0];
}

int main(){
  {
    std::vector<float> A(16*32), B(32*16), C(256), R(256,0.f);
    for(auto&x:A) x = (float)((rand()%17)-8); for(auto&x:B) x=(float)((rand()%13)-6);
    for(int i=0;i<16;i++)for(int n=0;n<16;n++){float s=0;for(int k=0;k<32;k++)s+=A[i*32+k]*B[k*16+n];R[i*16+n]=s;}
    float *dA,*dB,*dC; hipMalloc(&dA,A.size()*4);hipMalloc(&dB,B.size()*4);hipMalloc(&dC,1024);
    hipMemcpy(dA,A.data(),A.size()*4,hipMemcpyHostToDevice);hipMemcpy(dB,B.data(),B.size()*4,hipMemcpyHostToDevice);
    mfma16<<<1,64>>>(dA,dB,dC); hipMemcpy(C.data(),dC,1024,hipMemcpyDeviceToHost);
    float e=0;for(int i=0;i<256;i++)e=fmaxf(e,fabsf(C[i]-R[i])); printf("MFMA16x16x32 hypothesis maxerr=%g\n",e);
  }
  {
    std::vector<float> A(32*16), B(16*32), C(1024), R(1024,0.f);
    for(auto&x:A) x = (float)((rand()%17)-8); for(auto&x:B) x=(float)((rand()%13)-6);
    for(int i=0;i<32;i++)for(int n=0;n<32;n++){float s=0;for(int k=0;k<16;k++)s+=A[i*16+k]*B[k*32+n];R[i*32+n]=s;}
    float *dA,*dB,*dC; hipMalloc(&dA,A.size()*4);hipMalloc(&dB,B.size()*4);hipMalloc(&dC,4096);
    hipMemcpy(dA,A.data(),A.size()*4,hipMemcpyHostToDevice);hipMemcpy(dB,B.data(),B.size()*4,hipMemcpyHostToDevice);
    mfma32<<<1,64>>>(dA,dB,dC); hipMemcpy(C.data(),dC,4096,hipMemcpyDeviceToHost);
    float e=0;for(int i=0;i<1024;i++)e=fmaxf(e,fabsf(C[i]-R[i])); printf("MFMA32x32x16 hypothesis maxerr=%g\n",e);
  }
  {
    short* d; hipMalloc(&d,512); std::vector<short> h(256);
    trprobe<<<1,64>>>(d); hipMemcpy(h.data(),d,512,hipMemcpyDeviceToHost);
    printf("TR16 dense (lane addr = l*4 shorts): lane: v0 v1 v2 v3\n");
    for(int l=0;l<64;l++) printf("  %2d: %4d %4d %4d %4d\n",l,h[l*4],h[l*4+1],h[l*4+2],h[l*4+3]);
    trprobe2<<<1,64>>>(d); hipMemcpy(h.data(),d,512,hipMemcpyDeviceToHost);
    printf("TR16 strided [16][64] (value=row*64+col): lane: v0..v3\n");
    for(int l=0;l<64;l++) printf("  %2d: %4d %4d %4d %4d\n",l,h[l*4],h[l*4+1],h[l*4+2],h[l*4+3]);
  }
  {
    std::vector<float> src(256); for(int i=0;i<256;i++) src[i]=(float)(i+1);
    float *ds,*dout; hipMalloc(&ds,1024); hipMalloc(&dout,(256+64)*4);
    hipMemcpy(ds,src.data(),1024,hipMemcpyHostToDevice);
    oobprobe<<<1,64>>>(ds, 512, dout); // only first 128 floats valid
    std::vector<float> o(320); hipMemcpy(o.data(),dout,320*4,hipMemcpyDeviceToHost);
    printf("buffer_load lds OOB: lds[124..131]= "); for(int i=124;i<132;i++) printf("%g ",o[i]); printf(" lds[255]=%g\n",o[255]);
    printf("buffer_load reg OOB: lane31=%g lane32=%g lane63=%g\n",o[256+31],o[256+32],o[256+63]);
  }
  hipDeviceSynchronize();
  printf("hipGetLastError=%d\n",(int)hipGetLastError());
  return 0;
}
