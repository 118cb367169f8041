// How fast does ONE lane's dependent fp32 add chain run on gfx950?  (sec_mean_pc_kernel's serial part.)
// hipcc --offload-arch=gfx950 -O3 tools/probes/addchain.hip -o /tmp/addchain && /tmp/addchain
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void chain_reg(float *out, const float *in, int n, long long *cyc) {
    float x[32];
    for (int j = 0; j < 32; j++) x[j] = in[j];
    float m = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i += 32) {
#pragma unroll
        for (int j = 0; j < 32; j++) m = __fadd_rn(m, x[j]);
        asm volatile("" : "+v"(m));
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = m; cyc[0] = t1 - t0; }
}
__global__ void chain_lds(float *out, const float *in, int n, long long *cyc) {
    __shared__ __attribute__((aligned(16))) float st[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) st[i] = in[i & 31];
    __syncthreads();
    float m = 0.f;
    long long t0 = __builtin_readcyclecounter();
    if (threadIdx.x < 3) {
        const float *s = st + threadIdx.x * 1024;
        for (int rep = 0; rep < n / 1024; rep++) {
            float4 w0[8], w1[8];
#pragma unroll
            for (int j = 0; j < 8; j++) w0[j] = *(const float4 *)(s + j * 4);
            for (int r = 0; r + 64 <= 1024; r += 64) {
#pragma unroll
                for (int j = 0; j < 8; j++) w1[j] = *(const float4 *)(s + r + 32 + j * 4);
#pragma unroll
                for (int j = 0; j < 8; j++) { m = __fadd_rn(m, w0[j].x); m = __fadd_rn(m, w0[j].y); m = __fadd_rn(m, w0[j].z); m = __fadd_rn(m, w0[j].w); }
                if (r + 96 <= 1024) {
#pragma unroll
                    for (int j = 0; j < 8; j++) w0[j] = *(const float4 *)(s + r + 64 + j * 4);
                }
#pragma unroll
                for (int j = 0; j < 8; j++) { m = __fadd_rn(m, w1[j].x); m = __fadd_rn(m, w1[j].y); m = __fadd_rn(m, w1[j].z); m = __fadd_rn(m, w1[j].w); }
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = m; cyc[0] = t1 - t0; }
}
int main() {
    float *in, *out; long long *cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 64); hipMalloc(&cyc, 64);
    float h[4096]; for (int i = 0; i < 4096; i++) h[i] = 1e-4f * (i % 7 + 1);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int n = 32768;
    for (int v = 0; v < 2; v++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            if (v == 0) chain_reg<<<1, 64>>>(out, in, n, cyc); else chain_lds<<<1, 256>>>(out, in, n, cyc);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s: %d dependent adds: %.1f us by events, %lld counter ticks = %.2f ticks/add, %.2f ns/add\n", v == 0 ? "registers" : "lds      ", n, ms * 1e3, c,
                   (double)c / n, ms * 1e6 / n);
        }
    }
    return 0;
}
