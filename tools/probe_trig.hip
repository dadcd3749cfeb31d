// Accuracy probe: hardware v_sin_f32 / v_cos_f32 / v_exp_f32 / v_log_f32 against fp64 libm over the ranges the
// positional encoding and Softplus(beta=100) use.  hipcc --offload-arch=gfx950 tools/probe_trig.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* s, float* c, float* sp, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float dz = x[i];                       // metres, |dz| < 0.5
    float rev = dz * 0.5f;                 // sin(pi*dz) = sin(2 pi * dz/2)
    s[i] = __builtin_amdgcn_sinf(rev);
    c[i] = __builtin_amdgcn_cosf(rev);
    float t = dz * 100.0f;                 // softplus argument range +-50
    float e = __builtin_amdgcn_exp2f(-fabsf(t) * 1.44269504f);
    sp[i] = fmaxf(dz, 0.0f) + __builtin_amdgcn_logf(1.0f + e) * (0.693147181f * 0.01f);
}
int main()
{
    const int n = 1 << 20;
    std::vector<float> hx(n), hs(n), hc(n), hp(n);
    for (int i = 0; i < n; ++i) hx[i] = -0.5f + (float)i / n;
    float *dx, *ds, *dc, *dp;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dp, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, dp, n);
    hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hp.data(), dp, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, ep = 0;
    for (int i = 0; i < n; ++i) {
        double a = (double)(hx[i] * 3.14159274f);
        es = fmax(es, fabs(hs[i] - sin(a)));
        ec = fmax(ec, fabs(hc[i] - cos(a)));
        double t = (double)hx[i] * 100.0;
        double ref = t > 20 ? (double)hx[i] : log1p(exp(t)) / 100.0;
        ep = fmax(ep, fabs(hp[i] - ref));
    }
    printf("max abs err: v_sin %.3e  v_cos %.3e  softplus100(fast) %.3e\n", es, ec, ep);
    return 0;
}
