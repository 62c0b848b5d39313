// Generic direct 3-D convolution / transposed convolution for gfx950: any Cin/Cout/k, stride 1|2.
// One thread per output element, fp32 FMA chain in (kd,kh,kw,ci) order.  This is the correctness
// anchor on the GPU (it covers the filter counts 1/2/4 the reference's shape tests use,
// src/test_model_transforms.py:27-73) and the fallback for shapes the MFMA path does not tile.
#include "common.h"

namespace {

struct Geo {
    int N, D, H, W, Cin, Cout, k, stride;
    int OD, OH, OW;
    int pd, ph, pw;
    int flags, ocs, oco;
};

__device__ __forceinline__ float epilogue(float v, int co, size_t vox, const Geo& g, const float* bias,
                                          const float* residual) {
    if (g.flags & PCC_CONV_BIAS) v += bias[co];
    if (g.flags & PCC_CONV_RELU) v = fmaxf(v, 0.f);
    if (g.flags & PCC_CONV_ADD) v += residual[vox * g.Cout + co];
    if (g.flags & PCC_CONV_CLIP01) v = fminf(fmaxf(v, 0.f), 1.f);
    return v;
}

__global__ void __launch_bounds__(256) conv_fwd_generic(Geo g, const float* __restrict__ in,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ residual,
                                                        float* __restrict__ out) {
    const size_t total = (size_t)g.N * g.OD * g.OH * g.OW * g.Cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % g.Cout);
        const size_t vox = i / g.Cout;
        size_t t = vox;
        const int ow = (int)(t % g.OW); t /= g.OW;
        const int oh = (int)(t % g.OH); t /= g.OH;
        const int od = (int)(t % g.OD);
        const int n = (int)(t / g.OD);
        float acc = 0.f;
        for (int kd = 0; kd < g.k; ++kd) {
            const int id = od * g.stride + kd - g.pd;
            if (id < 0 || id >= g.D) continue;
            for (int kh = 0; kh < g.k; ++kh) {
                const int ih = oh * g.stride + kh - g.ph;
                if (ih < 0 || ih >= g.H) continue;
                for (int kw = 0; kw < g.k; ++kw) {
                    const int iw = ow * g.stride + kw - g.pw;
                    if (iw < 0 || iw >= g.W) continue;
                    const float* ip = in + ((((size_t)n * g.D + id) * g.H + ih) * g.W + iw) * g.Cin;
                    const float* wp = w + ((((size_t)kd * g.k + kh) * g.k + kw) * g.Cin) * g.Cout + co;
                    for (int ci = 0; ci < g.Cin; ++ci) acc = fmaf(ip[ci], wp[(size_t)ci * g.Cout], acc);
                }
            }
        }
        out[vox * g.ocs + g.oco + co] = epilogue(acc, co, vox, g, bias, residual);
    }
}

// gather form of the SAME transposed conv: out[o] = sum_{i,kappa: o = i*s + kappa - pad} in[i] w[kappa]
__global__ void __launch_bounds__(256) conv_tr_generic(Geo g, const float* __restrict__ in,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ residual,
                                                       float* __restrict__ out) {
    const size_t total = (size_t)g.N * g.OD * g.OH * g.OW * g.Cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % g.Cout);
        const size_t vox = i / g.Cout;
        size_t t = vox;
        const int ow = (int)(t % g.OW); t /= g.OW;
        const int oh = (int)(t % g.OH); t /= g.OH;
        const int od = (int)(t % g.OD);
        const int n = (int)(t / g.OD);
        float acc = 0.f;
        for (int kd = 0; kd < g.k; ++kd) {
            const int td = od + g.pd - kd;
            if (td < 0 || (td % g.stride) != 0) continue;
            const int id = td / g.stride;
            if (id >= g.D) continue;
            for (int kh = 0; kh < g.k; ++kh) {
                const int th = oh + g.ph - kh;
                if (th < 0 || (th % g.stride) != 0) continue;
                const int ih = th / g.stride;
                if (ih >= g.H) continue;
                for (int kw = 0; kw < g.k; ++kw) {
                    const int tw = ow + g.pw - kw;
                    if (tw < 0 || (tw % g.stride) != 0) continue;
                    const int iw = tw / g.stride;
                    if (iw >= g.W) continue;
                    const float* ip = in + ((((size_t)n * g.D + id) * g.H + ih) * g.W + iw) * g.Cin;
                    // Keras Conv3DTranspose kernel: (kd,kh,kw,Cout,Cin)
                    const float* wp = w + (((((size_t)kd * g.k + kh) * g.k + kw) * g.Cout) + co) * g.Cin;
                    for (int ci = 0; ci < g.Cin; ++ci) acc = fmaf(ip[ci], wp[ci], acc);
                }
            }
        }
        out[vox * g.ocs + g.oco + co] = epilogue(acc, co, vox, g, bias, residual);
    }
}

}  // namespace

int pcc_conv3d_generic(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w,
                       const float* bias, const float* residual, float* out, hipStream_t st) {
    Geo g;
    g.N = d->N; g.D = d->D; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Cout = d->Cout;
    g.k = d->k; g.stride = d->stride; g.flags = d->flags;
    g.ocs = d->out_cstride ? d->out_cstride : d->Cout;
    g.oco = d->out_coffset;
    pcc_conv_out_dims(d, &g.OD, &g.OH, &g.OW);
    if (d->transposed) {
        g.pd = pcc_same_pad_low(g.OD, g.k, g.stride);
        g.ph = pcc_same_pad_low(g.OH, g.k, g.stride);
        g.pw = pcc_same_pad_low(g.OW, g.k, g.stride);
    } else {
        g.pd = pcc_same_pad_low(g.D, g.k, g.stride);
        g.ph = pcc_same_pad_low(g.H, g.k, g.stride);
        g.pw = pcc_same_pad_low(g.W, g.k, g.stride);
    }
    const size_t total = (size_t)g.N * g.OD * g.OH * g.OW * g.Cout;
    size_t blocks = (total + 255) / 256;
    const size_t cap = (size_t)ctx->num_cu * 32;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) return PCC_OK;
    if (d->transposed)
        hipLaunchKernelGGL(conv_tr_generic, dim3((unsigned)blocks), dim3(256), 0, st, g, in, w, bias, residual, out);
    else
        hipLaunchKernelGGL(conv_fwd_generic, dim3((unsigned)blocks), dim3(256), 0, st, g, in, w, bias, residual, out);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
