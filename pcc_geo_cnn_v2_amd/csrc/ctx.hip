// Context, error reporting and the conv dispatcher of libpcc_geo_hip.so.
#include <cstring>
#include <new>

#include <cstdlib>

#include "common.h"

static thread_local char g_err[512] = "";

void pcc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

PCC_API int pcc_abi_version(void) { return PCC_ABI_VERSION; }
PCC_API const char* pcc_last_error(void) { return g_err; }

PCC_API int pcc_ctx_create(int device, pcc_ctx** out) {
    PCC_REQUIRE(out != nullptr, "pcc_ctx_create: out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        pcc_set_error("pcc_ctx_create: no HIP device visible");
        return PCC_ERR_NOGPU;
    }
    PCC_REQUIRE(device >= 0 && device < n, "pcc_ctx_create: device %d out of range [0,%d)", device, n);
    pcc_ctx* c = new (std::nothrow) pcc_ctx();
    PCC_REQUIRE(c != nullptr, "pcc_ctx_create: out of memory");
    c->device = device;
    hipError_t e = hipGetDeviceProperties(&c->prop, device);
    if (e != hipSuccess) {
        delete c;
        pcc_set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return PCC_ERR_HIP;
    }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        pcc_set_error("pcc_ctx_create: device %d is %s, this library is built for gfx950 only", device,
                      c->prop.gcnArchName);
        delete c;
        return PCC_ERR_NOGPU;
    }
    c->num_cu = c->prop.multiProcessorCount;
    c->numerics = pcc_numerics_from_env();
    *out = c;
    return PCC_OK;
}

// the ONE place the numerics-affecting environment switches are read (include/pcc_geo.h, PCC_NUM_*)
uint32_t pcc_numerics_from_env() {
    static const struct { const char* name; uint32_t bit; } flags[] = {
        {"PCC_NO_SPLIT", PCC_NUM_NO_SPLIT}, {"PCC_NO_SPLIT_DIRECT", PCC_NUM_NO_SPLIT_DIRECT}, {"PCC_NO_SPLIT_TR2", PCC_NUM_NO_SPLIT_TR2},
        {"PCC_NO_WINOGRAD", PCC_NUM_NO_WINOGRAD}, {"PCC_NO_WINOGRAD32", PCC_NUM_NO_WINOGRAD32}, {"PCC_NO_WINOGRAD64", PCC_NUM_NO_WINOGRAD64},
        {"PCC_WINO_PER_GROUP", PCC_NUM_WINO_PER_GROUP}, {"PCC_NO_TR2M", PCC_NUM_NO_TR2M}, {"PCC_TR2M", PCC_NUM_TR2M},
        {"PCC_TR2_OLD", PCC_NUM_TR2_OLD}, {"PCC_NO_F16S", PCC_NUM_NO_F16S}, {"PCC_COUT1_T16", PCC_NUM_COUT1_T16}};
    uint32_t m = 0;
    for (const auto& f : flags)
        if (getenv(f.name) != nullptr) m |= f.bit;
    if (const char* v = getenv("PCC_SPLIT_MFMA")) m |= atoi(v) == 16 ? PCC_NUM_SPLIT_MFMA16 : atoi(v) == 32 ? PCC_NUM_SPLIT_MFMA32 : 0u;
    if (const char* v = getenv("PCC_SPLIT_TILE")) m |= atoi(v) == 8 ? PCC_NUM_SPLIT_TILE8 : 0u;
    if (const char* v = getenv("PCC_P16")) m |= atoi(v) ? PCC_NUM_P16 : 0u;
    return m;
}

PCC_API int pcc_ctx_get_numerics(pcc_ctx* ctx, uint32_t* family, uint32_t* switches) {
    PCC_REQUIRE(ctx != nullptr, "pcc_ctx_get_numerics: ctx is NULL");
    if (family) *family = PCC_KERNEL_FAMILY;
    if (switches) *switches = ctx->numerics;
    return PCC_OK;
}

PCC_API int pcc_ctx_set_numerics(pcc_ctx* ctx, uint32_t switches) {
    PCC_REQUIRE(ctx != nullptr, "pcc_ctx_set_numerics: ctx is NULL");
    PCC_REQUIRE((switches & ~0xffffu) == 0, "pcc_ctx_set_numerics: unknown bits 0x%x", switches & ~0xffffu);
    ctx->numerics = switches;
    return PCC_OK;
}

int pcc_ctx_scratch(pcc_ctx* ctx, size_t bytes, void** ptr) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) PCC_CHECK_HIP(hipFree(ctx->scratch));      // hipFree waits for the kernels that still use it
        ctx->scratch = nullptr; ctx->scratch_bytes = 0;
        PCC_CHECK_HIP(hipMalloc(&ctx->scratch, bytes));
        ctx->scratch_bytes = bytes;
    }
    *ptr = ctx->scratch;
    return PCC_OK;
}

int pcc_ctx_amax(pcc_ctx* ctx, int n, unsigned** ptr) {
    if (n > ctx->amax_cap) {
        if (ctx->amax) PCC_CHECK_HIP(hipFree(ctx->amax));
        ctx->amax = nullptr; ctx->amax_cap = 0;
        const int cap = n > 4096 ? n : 4096;
        PCC_CHECK_HIP(hipMalloc((void**)&ctx->amax, (size_t)cap * sizeof(unsigned)));
        ctx->amax_cap = cap;
    }
    *ptr = ctx->amax;
    return PCC_OK;
}

PCC_API int pcc_ctx_destroy(pcc_ctx* ctx) {
    if (ctx) pcc_profile_free(ctx);
    if (ctx && ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx && ctx->amax) (void)hipFree(ctx->amax);
    delete ctx;
    return PCC_OK;
}

PCC_API int pcc_ctx_num_cu(pcc_ctx* ctx) {
    PCC_REQUIRE(ctx != nullptr, "pcc_ctx_num_cu: ctx is NULL");
    return ctx->num_cu;
}

PCC_API int pcc_conv_out_dims(const pcc_conv_desc* d, int32_t* OD, int32_t* OH, int32_t* OW) {
    PCC_REQUIRE(d && OD && OH && OW, "pcc_conv_out_dims: NULL argument");
    PCC_REQUIRE(d->stride >= 1, "pcc_conv_out_dims: stride must be >= 1");
    if (d->transposed) {
        *OD = d->D * d->stride; *OH = d->H * d->stride; *OW = d->W * d->stride;
    } else {
        *OD = pcc_same_out(d->D, d->stride); *OH = pcc_same_out(d->H, d->stride); *OW = pcc_same_out(d->W, d->stride);
    }
    return PCC_OK;
}

PCC_API int pcc_conv3d(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w,
                       const float* w_packed, const float* bias, const float* residual, float* out,
                       void* stream) {
    return pcc_conv3d_ext(ctx, d, in, w, w_packed, bias, residual, out, nullptr, stream);
}

// pcc_conv3d with the side channel of the fp16-split kernels (common.h, pcc_conv_ext): what pcc_network_forward calls
int pcc_conv3d_ext(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w, const float* w_packed, const float* bias,
                   const float* residual, float* out, pcc_conv_ext* ext, void* stream) {
    if (ext) ext->out_recorded = false;
    PCC_REQUIRE(ctx && d && in && out, "pcc_conv3d: NULL argument");
    PCC_REQUIRE(d->N > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0,
                "pcc_conv3d: non-positive dimension");
    PCC_REQUIRE(d->k >= 1 && (d->stride == 1 || d->stride == 2), "pcc_conv3d: unsupported k/stride");
    PCC_REQUIRE(!(d->flags & PCC_CONV_BIAS) || bias, "pcc_conv3d: PCC_CONV_BIAS set but bias is NULL");
    PCC_REQUIRE(!(d->flags & PCC_CONV_ADD) || residual, "pcc_conv3d: PCC_CONV_ADD set but residual is NULL");
    PCC_REQUIRE(d->out_cstride == 0 || d->out_cstride >= d->Cout + d->out_coffset,
                "pcc_conv3d: out_cstride too small");
    {
        const int32_t known = PCC_CONV_BIAS | PCC_CONV_RELU | PCC_CONV_ADD | PCC_CONV_CLIP01 | PCC_CONV_F16 | PCC_CONV_IN16 | PCC_CONV_OUT16 | PCC_CONV_RES16;
        PCC_REQUIRE((d->flags & ~known) == 0, "pcc_conv3d: unknown bits in flags");
    }
    PCC_REQUIRE(!(d->flags & (PCC_CONV_IN16 | PCC_CONV_OUT16 | PCC_CONV_RES16)) || (d->flags & PCC_CONV_F16),
                "pcc_conv3d: fp16 storage (IN16 / OUT16 / RES16) exists only inside the fp16 mode (PCC_CONV_F16)");
    PCC_REQUIRE(!(d->flags & PCC_CONV_RES16) || ((d->flags & PCC_CONV_IN16) && (d->flags & PCC_CONV_ADD)),
                "pcc_conv3d: PCC_CONV_RES16 goes with PCC_CONV_IN16 | PCC_CONV_ADD");
    PCC_REQUIRE(!(d->flags & PCC_CONV_IN16) || !(d->flags & PCC_CONV_ADD) || (d->flags & PCC_CONV_RES16),
                "pcc_conv3d: an fp16-input layer takes its residual in fp16 (PCC_CONV_RES16)");
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const bool fast_ok = w_packed != nullptr && pcc_conv_mfma_supported(d) == 1;
    if (d->flags & (PCC_CONV_IN16 | PCC_CONV_OUT16)) {
        PCC_REQUIRE(fast_ok && d->impl == PCC_IMPL_AUTO, "pcc_conv3d: fp16 storage needs the packed weights and PCC_IMPL_AUTO");
        return pcc_conv3d_mfma_thr(ctx, d, in, w_packed, bias, residual, out, nullptr, nullptr, ext, st);
    }
    if (d->impl == PCC_IMPL_MFMA || d->impl == PCC_IMPL_WINOGRAD || d->impl == PCC_IMPL_SPLIT) {
        PCC_REQUIRE(fast_ok, "pcc_conv3d: PCC_IMPL_MFMA/WINOGRAD/SPLIT requested but shape not covered or w_packed NULL");
        return pcc_conv3d_mfma_thr(ctx, d, in, w_packed, bias, residual, out, nullptr, nullptr, ext, st);
    }
    if (d->impl == PCC_IMPL_AUTO && fast_ok) return pcc_conv3d_mfma_thr(ctx, d, in, w_packed, bias, residual, out, nullptr, nullptr, ext, st);
    PCC_REQUIRE(w != nullptr, "pcc_conv3d: generic path needs the Keras-layout weights `w`");
    return pcc_conv3d_generic(ctx, d, in, w, bias, residual, out, st);
}
