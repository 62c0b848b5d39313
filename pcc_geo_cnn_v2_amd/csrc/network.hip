// Whole-transform and whole-graph entry points of the C ABI (include/pcc_geo.h, "batched graph" section).
//
// The layer stacks of /root/reference/src/model_transforms.py:41-158 and the graph wiring of
// /root/reference/src/model_types.py:283-309 (V1) / :371-411 (V2) as HOST code that enqueues the kernels of this library on
// one stream: one call per transform (pcc_network_forward) or per graph phase (pcc_codec_*), instead of one ctypes call per
// layer.  All device memory belongs to the caller: the packed weight blob, the activations workspace and every output.
#include <cstring>
#include <vector>

#include "common.h"

namespace {

struct LayerSpec {
    bool transposed;
    int cout, k, stride;
    bool bias, relu;
    int res;          // 0 none, 1 output is kept as the residual of the block ("tensor1"), 2 residual added after the activation
};

// AnalysisBlock :62-70 / SynthesisBlock :73-81 with ResidualLayer.call :30-38 (mode 'add'): t1 = L0(x); t = L2(L1(t1)); t1 + t
void add_block(std::vector<LayerSpec>& v, bool tr, int f) {
    v.push_back({tr, f, 3, 2, true, true, 1});
    v.push_back({tr, f, 3, 1, true, true, 0});
    v.push_back({tr, f, 3, 1, true, true, 2});
}

bool build_layers(int transform, int F, std::vector<LayerSpec>& v, int* cin0) {
    v.clear();
    if (F <= 0) return false;
    *cin0 = F;
    switch (transform) {
        case PCC_NET_ANALYSIS_V1:             // :41-48
            *cin0 = 1;
            v.push_back({false, F, 9, 2, true, true, 0});
            v.push_back({false, F, 5, 2, true, true, 0});
            v.push_back({false, F, 5, 2, false, false, 0});
            return true;
        case PCC_NET_SYNTHESIS_V1:            // :51-59 (the final activation is ReLU, :58)
            v.push_back({true, F, 5, 2, true, true, 0});
            v.push_back({true, F, 5, 2, true, true, 0});
            v.push_back({true, 1, 9, 2, true, true, 0});
            return true;
        case PCC_NET_ANALYSIS_V2:             // :84-95
            *cin0 = 1;
            add_block(v, false, F / 2); add_block(v, false, F); add_block(v, false, F);
            v.push_back({false, F, 3, 1, false, false, 0});
            return true;
        case PCC_NET_SYNTHESIS_V2:            // :98-109
            add_block(v, true, F); add_block(v, true, F); add_block(v, true, F / 2);
            v.push_back({true, 1, 3, 1, true, true, 0});
            return true;
        case PCC_NET_ANALYSIS_PROGRESSIVE_V2: // :112-123
            *cin0 = 1;
            add_block(v, false, F / 4); add_block(v, false, F / 2); add_block(v, false, F);
            v.push_back({false, F, 3, 1, false, false, 0});
            return true;
        case PCC_NET_SYNTHESIS_PROGRESSIVE_V2:  // :126-137
            add_block(v, true, F); add_block(v, true, F / 2); add_block(v, true, F / 4);
            v.push_back({true, 1, 3, 1, true, true, 0});
            return true;
        case PCC_NET_HYPER_ANALYSIS:          // :140-147
            v.push_back({false, F, 3, 1, true, true, 0});
            v.push_back({false, F, 3, 2, true, true, 0});
            v.push_back({false, F, 3, 1, false, false, 0});
            return true;
        case PCC_NET_HYPER_SYNTHESIS:         // :150-158 (all three bias + ReLU)
            v.push_back({true, F, 3, 1, true, true, 0});
            v.push_back({true, F, 3, 2, true, true, 0});
            v.push_back({true, F, 3, 1, true, true, 0});
            return true;
        default: return false;
    }
}

inline size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }   // blob segments start on 256-byte boundaries

struct LayerImage { size_t w, pk, pk_floats, b; int cin; };            // float offsets inside the blob

pcc_conv_desc layer_desc(const LayerSpec& L, int cin, int N, int D, int H, int W, int flags) {
    pcc_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.N = N; d.D = D; d.H = H; d.W = W; d.Cin = cin; d.Cout = L.cout; d.k = L.k; d.stride = L.stride;
    d.transposed = L.transposed ? 1 : 0;
    d.flags = flags | (L.bias ? PCC_CONV_BIAS : 0) | (L.relu ? PCC_CONV_RELU : 0) | (L.res == 2 ? PCC_CONV_ADD : 0);
    d.impl = PCC_IMPL_AUTO;
    return d;
}

// The packed image of a layer depends on (Cin, Cout, k, stride, transposed) only; 64^3 is a size every fast kernel covers.
size_t blob_layout(const std::vector<LayerSpec>& v, int cin0, std::vector<LayerImage>& im) {
    size_t off = 0;
    int cin = cin0;
    im.clear();
    for (const LayerSpec& L : v) {
        LayerImage I;
        I.cin = cin;
        I.w = off; off += align64((size_t)L.k * L.k * L.k * cin * L.cout);
        const pcc_conv_desc d = layer_desc(L, cin, 1, 64, 64, 64, 0);
        I.pk_floats = pcc_conv_packed_floats(&d);
        I.pk = off; off += align64(I.pk_floats);
        I.b = off; off += L.bias ? align64((size_t)L.cout) : 0;
        im.push_back(I);
        cin = L.cout;
    }
    return off;
}

void out_dims(const LayerSpec& L, int& D, int& H, int& W) {
    if (L.transposed) { D *= L.stride; H *= L.stride; W *= L.stride; }
    else { D = pcc_same_out(D, L.stride); H = pcc_same_out(H, L.stride); W = pcc_same_out(W, L.stride); }
}

// activations: three rotating buffers (x -> t1 -> t -> t1 + t reuses x's buffer), each as large as the largest intermediate
size_t act_floats(const std::vector<LayerSpec>& v, int N, int D, int H, int W) {
    size_t mx = 0;
    for (size_t i = 0; i + 1 < v.size(); ++i) {
        out_dims(v[i], D, H, W);
        const size_t n = (size_t)N * D * H * W * v[i].cout;
        if (n > mx) mx = n;
    }
    return align64(mx);
}

// behind the three activation buffers: one row of N per-block max |x| slots per layer output (the pre-scale side channel of the
// fp16-split kernels, common.h pcc_conv_ext), zeroed by one memset per forward call
inline size_t amax_row(int N) { return (size_t)N * PCC_AMAX_SLOTS; }       // uint32 slots per layer row
inline size_t amax_bytes(size_t layers, int N) { return layers * amax_row(N) * sizeof(unsigned); }

struct Profile {
    int transform = -1, layer = -1;
    int stride = 1;                 // every stride-th call of the selected layer is timed (the event records cost the queue ~6 us each)
    unsigned calls = 0;
    std::vector<hipEvent_t> ev;     // pairs (start, stop)
    size_t used = 0;
};

}  // namespace

// per-context profiling state lives behind the context (ctx.hip owns the struct; only this file touches `profile`)
static Profile* prof_of(pcc_ctx* ctx) {
    if (!ctx->profile) ctx->profile = new Profile();
    return (Profile*)ctx->profile;
}
void pcc_profile_free(pcc_ctx* ctx) {
    Profile* p = (Profile*)ctx->profile;
    if (!p) return;
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    delete p;
    ctx->profile = nullptr;
}

PCC_API int32_t pcc_network_num_layers(int32_t transform, int32_t filters) {
    std::vector<LayerSpec> v;
    int c0;
    if (!build_layers(transform, filters, v, &c0)) { pcc_set_error("pcc_network_num_layers: unknown transform %d / filters %d", transform, filters); return PCC_ERR_ARG; }
    return (int32_t)v.size();
}

PCC_API int pcc_network_layer(int32_t transform, int32_t filters, int32_t layer, pcc_conv_desc* d, int32_t* residual_role) {
    std::vector<LayerSpec> v;
    int c0;
    PCC_REQUIRE(d && build_layers(transform, filters, v, &c0), "pcc_network_layer: bad transform / filters");
    PCC_REQUIRE(layer >= 0 && layer < (int)v.size(), "pcc_network_layer: layer %d out of range", layer);
    int cin = c0;
    for (int i = 0; i < layer; ++i) cin = v[i].cout;
    *d = layer_desc(v[layer], cin, 0, 0, 0, 0, 0);
    if (residual_role) *residual_role = v[layer].res;
    return PCC_OK;
}

PCC_API size_t pcc_weights_blob_floats(int32_t transform, int32_t filters) {
    std::vector<LayerSpec> v;
    std::vector<LayerImage> im;
    int c0;
    if (!build_layers(transform, filters, v, &c0)) return 0;
    return blob_layout(v, c0, im);
}

PCC_API int pcc_weights_pack(int32_t transform, int32_t filters, const float* const* kernels, const float* const* biases,
                             float* blob_host) {
    std::vector<LayerSpec> v;
    std::vector<LayerImage> im;
    int c0;
    PCC_REQUIRE(kernels && blob_host && build_layers(transform, filters, v, &c0), "pcc_weights_pack: bad argument");
    const size_t total = blob_layout(v, c0, im);
    memset(blob_host, 0, total * sizeof(float));
    for (size_t i = 0; i < v.size(); ++i) {
        const LayerSpec& L = v[i];
        PCC_REQUIRE(kernels[i], "pcc_weights_pack: kernel of layer %d is NULL", (int)i);
        memcpy(blob_host + im[i].w, kernels[i], (size_t)L.k * L.k * L.k * im[i].cin * L.cout * sizeof(float));
        if (im[i].pk_floats) {
            const pcc_conv_desc d = layer_desc(L, im[i].cin, 1, 64, 64, 64, 0);
            const int rc = pcc_conv_pack_weights(&d, kernels[i], blob_host + im[i].pk);
            if (rc != PCC_OK) return rc;
        }
        if (L.bias) {
            PCC_REQUIRE(biases && biases[i], "pcc_weights_pack: bias of layer %d is NULL", (int)i);
            memcpy(blob_host + im[i].b, biases[i], (size_t)L.cout * sizeof(float));
        }
    }
    return PCC_OK;
}

PCC_API int pcc_weights_upload(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* const* kernels,
                               const float* const* biases, float* blob_device, void* stream) {
    PCC_REQUIRE(ctx && blob_device, "pcc_weights_upload: NULL argument");
    const size_t n = pcc_weights_blob_floats(transform, filters);
    PCC_REQUIRE(n > 0, "pcc_weights_upload: unknown transform %d / filters %d", transform, filters);
    std::vector<float> host(n);
    const int rc = pcc_weights_pack(transform, filters, kernels, biases, host.data());
    if (rc != PCC_OK) return rc;
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    // pageable source: the copy is staged by the runtime and complete on return (model load time, not the hot path)
    PCC_CHECK_HIP(hipMemcpyAsync(blob_device, host.data(), n * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    PCC_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PCC_OK;
}

PCC_API size_t pcc_network_workspace_bytes(int32_t transform, int32_t filters, int32_t N, int32_t D, int32_t H, int32_t W) {
    std::vector<LayerSpec> v;
    int c0;
    if (!build_layers(transform, filters, v, &c0) || N <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return 3 * act_floats(v, N, D, H, W) * sizeof(float) + 256 + amax_bytes(v.size(), N);
}

PCC_API int pcc_network_out_dims(int32_t transform, int32_t filters, int32_t D, int32_t H, int32_t W, int32_t* OD, int32_t* OH,
                                 int32_t* OW, int32_t* OC) {
    std::vector<LayerSpec> v;
    int c0;
    PCC_REQUIRE(OD && OH && OW && OC && build_layers(transform, filters, v, &c0), "pcc_network_out_dims: bad argument");
    for (const LayerSpec& L : v) out_dims(L, D, H, W);
    *OD = D; *OH = H; *OW = W; *OC = v.back().cout;
    return PCC_OK;
}

static int network_forward(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* blob, const float* x, int32_t N,
                           int32_t D, int32_t H, int32_t W, float* y, void* workspace, size_t workspace_bytes,
                           int32_t layer_flags, int32_t final_flags, const pcc_thr_fuse* fuse, bool* fused, void* stream);

PCC_API int pcc_network_forward(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* blob, const float* x, int32_t N,
                                int32_t D, int32_t H, int32_t W, float* y, void* workspace, size_t workspace_bytes,
                                int32_t layer_flags, int32_t final_flags, void* stream) {
    return network_forward(ctx, transform, filters, blob, x, N, D, H, W, y, workspace, workspace_bytes, layer_flags, final_flags,
                           nullptr, nullptr, stream);
}

// fuse (synthesis transforms whose last layer is the 16 -> 1 transposed conv): that layer also writes the occupancy bits of
// the fixed-threshold extraction; *fused reports whether it did.
static int network_forward(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* blob, const float* x, int32_t N,
                           int32_t D, int32_t H, int32_t W, float* y, void* workspace, size_t workspace_bytes,
                           int32_t layer_flags, int32_t final_flags, const pcc_thr_fuse* fuse, bool* fused, void* stream) {
    if (fused) *fused = false;
    std::vector<LayerSpec> v;
    std::vector<LayerImage> im;
    int c0;
    PCC_REQUIRE(ctx && blob && x && y, "pcc_network_forward: NULL argument");
    PCC_REQUIRE(build_layers(transform, filters, v, &c0), "pcc_network_forward: unknown transform %d / filters %d", transform, filters);
    PCC_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "pcc_network_forward: non-positive dimension");
    PCC_REQUIRE((layer_flags & ~PCC_CONV_F16) == 0 && (final_flags & ~PCC_CONV_CLIP01) == 0,
                "pcc_network_forward: layer_flags may hold PCC_CONV_F16, final_flags PCC_CONV_CLIP01");
    blob_layout(v, c0, im);
    const size_t af = act_floats(v, N, D, H, W);
    PCC_REQUIRE(v.size() == 1 || (workspace && workspace_bytes >= 3 * af * sizeof(float) + amax_bytes(v.size(), N)),
                "pcc_network_forward: workspace too small (need pcc_network_workspace_bytes)");
    float* buf[3];
    unsigned* amax = nullptr;      // [layer][amax_row(N)]
    {   // 256-byte aligned start
        uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
        for (int i = 0; i < 3; ++i) buf[i] = (float*)p + (size_t)i * af;
        PCC_REQUIRE(v.size() == 1 || (uintptr_t)(buf[2] + af) + amax_bytes(v.size(), N) <= (uintptr_t)workspace + workspace_bytes,
                    "pcc_network_forward: workspace too small after alignment");
        if (v.size() > 1) amax = (unsigned*)(buf[2] + af);
    }
    Profile* prof = ctx->profile ? (Profile*)ctx->profile : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const float* in = x;
    const float* t1 = nullptr;
    int cur = 0;                 // rotating buffer that receives the next output
    int in_buf = -1, t1_buf = -1;
    // fp16 mode: inside an AnalysisBlock / SynthesisBlock (width 16, 32 or 64) the two intermediate tensors (tensor1 and the
    // output of the middle conv) live in HBM as fp16: the stride-2 (transposed) conv hands over in fp16 (PCC_CONV_OUT16), the two k3
    // stride-1 convs run on conv_f16.hip (PCC_CONV_IN16), the last one adds the fp16 residual and writes fp32 for the next block.
    int f16_block_left = 0;      // layers of the current fp16-storage block still to come
    bool final_in16 = false;
    // pre-scale side channel: does some layer take an fp16-split kernel at all (then the rows are zeroed once, here)?
    bool amax_prev = false;      // the previous layer's kernel recorded the max |x| of its output blocks into its row
    if (amax && !(layer_flags & PCC_CONV_F16)) {
        bool any = false;
        int aD = D, aH = H, aW = W;
        for (size_t i = 0; i < v.size(); ++i) {
            const pcc_conv_desc da = layer_desc(v[i], im[i].cin, N, aD, aH, aW, 0);
            if (i > 0 && pcc_conv_wants_amax(ctx, &da)) any = true;
            out_dims(v[i], aD, aH, aW);
        }
        if (any) { PCC_CHECK_HIP(hipSetDevice(ctx->device)); PCC_CHECK_HIP(hipMemsetAsync(amax, 0, amax_bytes(v.size(), N), (hipStream_t)stream)); }
        else amax = nullptr;
    } else amax = nullptr;
    for (size_t i = 0; i < v.size(); ++i) {
        const LayerSpec& L = v[i];
        const bool last = i + 1 == v.size();
        int storage = 0;
        const int oH = L.transposed ? 2 * H : H / 2, oW = L.transposed ? 2 * W : W / 2;      // (stride-2 layers)
        if ((layer_flags & PCC_CONV_F16) && L.res == 1 && L.stride == 2 && L.k == 3 && (L.cout == 16 || L.cout == 32 || L.cout == 64) &&
            oH % 16 == 0 && oW % 16 == 0 && H % 2 == 0 && W % 2 == 0 && i + 2 < v.size() && v[i + 1].res == 0 && v[i + 2].res == 2 &&
            v[i + 1].k == 3 && v[i + 1].stride == 1 && v[i + 2].k == 3 && v[i + 2].stride == 1 &&
            [&] {   // every layer of the block must be able to take its fp16 role: pcc_conv3d has no fallback for IN16 / RES16
                const pcc_conv_desc d0 = layer_desc(L, im[i].cin, N, D, H, W, layer_flags);
                if (pcc_conv_mfma_supported(&d0) != 1) return false;     // (a layer on the generic path cannot hand over in fp16)
                int bD = D, bH = H, bW = W;
                out_dims(L, bD, bH, bW);
                for (size_t q = i + 1; q <= i + 2; ++q) {
                    const pcc_conv_desc dq = layer_desc(v[q], im[q].cin, N, bD, bH, bW, layer_flags | PCC_CONV_IN16);
                    if (pcc_conv_mfma_supported(&dq) != 1 || !pcc_f16_eligible(&dq)) return false;
                }
                return true;
            }()) {
            storage = PCC_CONV_OUT16;
            f16_block_left = 2;
        } else if (f16_block_left == 2) {
            storage = PCC_CONV_IN16 | PCC_CONV_OUT16;
            f16_block_left = 1;
        } else if (f16_block_left == 1) {
            storage = PCC_CONV_IN16 | PCC_CONV_RES16;
            f16_block_left = 0;
            // the block's output stays fp16 when its only consumer is the final 16 -> 1 transposed conv (which then reads 8 B
            // per lane and contracts with one fp16 MFMA)
            if (i + 2 == v.size() && v[i + 1].transposed && v[i + 1].cout == 1 && v[i + 1].k == 3 && v[i + 1].stride == 1 && L.cout == 16 &&
                [&] { const pcc_conv_desc df = layer_desc(v[i + 1], im[i + 1].cin, N, D, H, W, layer_flags | PCC_CONV_IN16);
                      return pcc_conv_mfma_supported(&df) == 1; }()) {
                storage |= PCC_CONV_OUT16;
                final_in16 = true;
            }
        } else if (last && final_in16) {
            storage = PCC_CONV_IN16;
        }
        pcc_conv_desc d = layer_desc(L, im[i].cin, N, D, H, W, layer_flags | storage | (last ? final_flags : 0));
        float* out;
        if (last) out = y;
        else {
            // pick a buffer that holds neither the input nor the pending residual
            while (cur == in_buf || cur == t1_buf) cur = (cur + 1) % 3;
            out = buf[cur];
        }
        const bool timed = prof && prof->transform == transform && prof->layer == (int)i && (prof->calls++ % (unsigned)prof->stride) == 0;
        if (timed) {
            if (prof->used + 2 > prof->ev.size()) {
                PCC_REQUIRE(prof->ev.size() < 8192, "pcc_network_forward: profile buffer full (pcc_profile_read drains it)");
                for (int e = 0; e < 2; ++e) { hipEvent_t ev; PCC_CHECK_HIP(hipEventCreate(&ev)); prof->ev.push_back(ev); }
            }
            PCC_CHECK_HIP(hipEventRecord(prof->ev[prof->used], st));
        }
        int rc;
        // the layer's input maxima (if its producer recorded them) and where its own go (if the next layer will ask for them)
        pcc_conv_ext ext = {nullptr, nullptr, false};
        if (amax) {
            if (i > 0 && amax_prev) ext.in_amax = amax + (i - 1) * amax_row(N);
            if (!last) {
                int nD = D, nH = H, nW = W;
                out_dims(L, nD, nH, nW);
                const pcc_conv_desc dn = layer_desc(v[i + 1], im[i + 1].cin, N, nD, nH, nW, 0);
                if (pcc_conv_wants_amax(ctx, &dn)) ext.out_amax = amax + i * amax_row(N);
            }
        }
        if (last && fuse && im[i].pk_floats && d.impl == PCC_IMPL_AUTO && pcc_conv_mfma_supported(&d) == 1) {
            PCC_CHECK_HIP(hipSetDevice(ctx->device));
            rc = pcc_conv3d_mfma_thr(ctx, &d, in, blob + im[i].pk, L.bias ? blob + im[i].b : nullptr, L.res == 2 ? t1 : nullptr, out,
                                     fuse, fused, &ext, st);
        } else {
            rc = pcc_conv3d_ext(ctx, &d, in, blob + im[i].w, im[i].pk_floats ? blob + im[i].pk : nullptr,
                                L.bias ? blob + im[i].b : nullptr, L.res == 2 ? t1 : nullptr, out, &ext, stream);
        }
        if (rc != PCC_OK) return rc;
        amax_prev = ext.out_recorded;
        if (timed) { PCC_CHECK_HIP(hipEventRecord(prof->ev[prof->used + 1], st)); prof->used += 2; }
        out_dims(L, D, H, W);
        if (L.res == 1) { t1 = out; t1_buf = last ? -1 : cur; }
        if (L.res == 2) { t1 = nullptr; t1_buf = -1; }
        in = out;
        in_buf = last ? -1 : cur;
    }
    return PCC_OK;
}

// The four names of SURVEY.md §8b: the same call restricted to one family of transforms.
#define PCC_NET_FAMILY(NAME, COND, WHAT)                                                                                         \
    PCC_API int NAME(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* blob, const float* x, int32_t N, int32_t D, \
                     int32_t H, int32_t W, float* y, void* workspace, size_t workspace_bytes, int32_t layer_flags,              \
                     int32_t final_flags, void* stream) {                                                                       \
        PCC_REQUIRE(COND, #NAME ": transform %d is not " WHAT, transform);                                                      \
        return pcc_network_forward(ctx, transform, filters, blob, x, N, D, H, W, y, workspace, workspace_bytes, layer_flags,    \
                                   final_flags, stream);                                                                        \
    }
PCC_NET_FAMILY(pcc_network_forward_analysis, transform == PCC_NET_ANALYSIS_V1 || transform == PCC_NET_ANALYSIS_V2 || transform == PCC_NET_ANALYSIS_PROGRESSIVE_V2, "an analysis transform")
PCC_NET_FAMILY(pcc_network_forward_synthesis, transform == PCC_NET_SYNTHESIS_V1 || transform == PCC_NET_SYNTHESIS_V2 || transform == PCC_NET_SYNTHESIS_PROGRESSIVE_V2, "a synthesis transform")
PCC_NET_FAMILY(pcc_network_forward_hyper_a, transform == PCC_NET_HYPER_ANALYSIS, "the hyper-analysis transform")
PCC_NET_FAMILY(pcc_network_forward_hyper_s, transform == PCC_NET_HYPER_SYNTHESIS, "the hyper-synthesis transform")

// ---- live kernel timing (bench.py's roofline object) ----------------------------------------------------------------------
PCC_API int pcc_profile_select(pcc_ctx* ctx, int32_t transform, int32_t layer) {
    PCC_REQUIRE(ctx, "pcc_profile_select: ctx is NULL");
    if (transform < 0) { if (ctx->profile) { Profile* p = (Profile*)ctx->profile; p->transform = p->layer = -1; p->used = 0; } return PCC_OK; }
    Profile* p = prof_of(ctx);
    p->transform = transform; p->layer = layer & 0xffff; p->stride = (layer >> 16) > 0 ? (layer >> 16) : 1; p->calls = 0; p->used = 0;
    return PCC_OK;
}

PCC_API int pcc_profile_read(pcc_ctx* ctx, float* ms, int32_t cap, int32_t* n) {
    PCC_REQUIRE(ctx && n && (ms || cap == 0), "pcc_profile_read: NULL argument");
    *n = 0;
    Profile* p = (Profile*)ctx->profile;
    if (!p) return PCC_OK;
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    for (size_t i = 0; i + 1 < p->used && *n < cap; i += 2) {
        PCC_CHECK_HIP(hipEventSynchronize(p->ev[i + 1]));
        PCC_CHECK_HIP(hipEventElapsedTime(&ms[*n], p->ev[i], p->ev[i + 1]));
        ++*n;
    }
    p->used = 0;
    return PCC_OK;
}

// ---- graph phases (src/model_types.py:283-309 V1, :371-411 V2) ------------------------------------------------------------
static int check_io(const pcc_symbol_io* io, const char* who) {
    PCC_REQUIRE(!io || ((io->sym_bytes == 2 || io->sym_bytes == 4) && (io->idx_bytes == 1 || io->idx_bytes == 4)),
                "%s: io widths must be 2|4 (symbols) and 1|4 (indexes)", who);
    return PCC_OK;
}

static int check_codec(const pcc_codec_desc* c) {
    PCC_REQUIRE(c && (c->version == 1 || c->version == 2) && c->filters > 0, "pcc_codec: bad descriptor (version / filters)");
    PCC_REQUIRE(c->w_synthesis && c->synthesis >= 0, "pcc_codec: the synthesis transform is required");
    PCC_REQUIRE(c->version == 1 || (c->w_hyper_synthesis && c->scale_table && c->scale_levels >= 1),
                "pcc_codec: version 2 needs the hyper-synthesis weights and the scale table");
    PCC_REQUIRE(c->version == 2 || c->medians, "pcc_codec: version 1 needs the EntropyBottleneck medians");
    return PCC_OK;
}

PCC_API size_t pcc_codec_workspace_bytes(const pcc_codec_desc* c, int32_t N, int32_t D, int32_t H, int32_t W) {
    if (!c) return 0;
    size_t m = pcc_network_workspace_bytes(c->synthesis, c->filters, N, D / 8, H / 8, W / 8);
    if (c->analysis >= 0) { const size_t a = pcc_network_workspace_bytes(c->analysis, c->filters, N, D, H, W); if (a > m) m = a; }
    if (c->version == 2) {
        size_t a = pcc_network_workspace_bytes(PCC_NET_HYPER_SYNTHESIS, c->filters, N, D / 16, H / 16, W / 16);
        if (a > m) m = a;
        a = pcc_network_workspace_bytes(PCC_NET_HYPER_ANALYSIS, c->filters, N, D / 8, H / 8, W / 8);
        if (a > m) m = a;
    }
    return m;
}

// x (N,D,H,W) occupancy -> every tensor of the compress graph.  Outputs are caller-owned device buffers, NDHWC:
//   y (N,D/8..,F) float, ysym int32, y_hat float, x_hat (N,D,H,W) float (unclipped; final_flags = PCC_CONV_CLIP01 clips);
//   V2 only: z (N,D/16..,F), zsym, z_hat, sigma (N,D/8..,F) float, idx int32.
// thr != NULL additionally runs the encoder-side thresholding + compaction (clipped x_hat) like pcc_threshold_compact.
PCC_API int pcc_codec_encode(pcc_ctx* ctx, const pcc_codec_desc* c, const float* x, int32_t N, int32_t D, int32_t H, int32_t W,
                             float* y, float* z, int32_t* zsym, float* z_hat, float* sigma, int32_t* idx, int32_t* ysym,
                             float* y_hat, float* x_hat, const float* thr, float* xyz, int32_t* counts, int64_t cap,
                             int32_t* scratch, void* workspace, size_t workspace_bytes, int32_t layer_flags,
                             int32_t final_flags, const pcc_symbol_io* sink, void* symbols_ready, void* stream) {
    int rc = check_codec(c);
    if (rc != PCC_OK) return rc;
    PCC_REQUIRE(c->analysis >= 0 && c->w_analysis, "pcc_codec_encode: the analysis transform is required");
    PCC_REQUIRE(ctx && x && y && ysym && y_hat && x_hat, "pcc_codec_encode: NULL argument");
    PCC_REQUIRE(D % 8 == 0 && H % 8 == 0 && W % 8 == 0 && (c->version == 1 || (D % 16 == 0 && H % 16 == 0 && W % 16 == 0)),
                "pcc_codec_encode: block edges must be multiples of 8 (V1) / 16 (V2)");
    const int F = c->filters;
    const size_t ny = (size_t)N * (D / 8) * (H / 8) * (W / 8) * F;
    rc = pcc_network_forward(ctx, c->analysis, F, c->w_analysis, x, N, D, H, W, y, workspace, workspace_bytes, layer_flags, 0, stream);
    if (rc != PCC_OK) return rc;
    // quantisers and the scale fold; with a sink each of them also packs its result for the host coder (stream order, narrow
    // integers, tile maxima) in the same launch, so that the caller's side stream carries one copy and no kernel
    rc = check_io(sink, "pcc_codec_encode");
    if (rc != PCC_OK) return rc;
    const int64_t vy = (int64_t)(D / 8) * (H / 8) * (W / 8), vz = (int64_t)(D / 16) * (H / 16) * (W / 16);
    const int cf = sink ? sink->channels_first : 0;
    auto quantize = [&](const float* v, const float* med, int32_t* sym, float* deq, int64_t vox, void* packed, int32_t* tmax) {
        if (sink && packed) return pcc_quantize_pack(ctx, v, med, sym, deq, N, vox, F, c->round_mode, cf, packed, sink->sym_bytes, tmax, stream);
        return pcc_quantize(ctx, v, med, sym, deq, (size_t)N * vox * F, F, c->round_mode, stream);
    };
    if (c->version == 1) {
        rc = quantize(y, c->medians, ysym, y_hat, vy, sink ? sink->ysym : nullptr, sink ? sink->ysym_tile_max : nullptr);
        if (rc != PCC_OK) return rc;
    } else {
        PCC_REQUIRE(c->w_hyper_analysis && z && zsym && z_hat && sigma && idx, "pcc_codec_encode: version 2 needs the hyper tensors");
        rc = pcc_network_forward(ctx, PCC_NET_HYPER_ANALYSIS, F, c->w_hyper_analysis, y, N, D / 8, H / 8, W / 8, z, workspace,
                                 workspace_bytes, layer_flags, 0, stream);
        if (rc != PCC_OK) return rc;
        rc = quantize(z, c->medians, zsym, z_hat, vz, sink ? sink->zsym : nullptr, sink ? sink->zsym_tile_max : nullptr);
        if (rc != PCC_OK) return rc;
        rc = pcc_network_forward(ctx, PCC_NET_HYPER_SYNTHESIS, F, c->w_hyper_synthesis, z_hat, N, D / 16, H / 16, W / 16, sigma,
                                 workspace, workspace_bytes, layer_flags, 0, stream);
        if (rc != PCC_OK) return rc;
        if (sink && sink->idx)
            rc = pcc_index_pack(ctx, sigma, c->scale_table, c->scale_levels, idx, N, vy, F, cf, sink->idx, sink->idx_bytes, stream);
        else
            rc = pcc_scale_to_index(ctx, sigma, c->scale_table, c->scale_levels, idx, ny, stream);
        if (rc != PCC_OK) return rc;
        rc = quantize(y, nullptr, ysym, y_hat, vy, sink ? sink->ysym : nullptr, sink ? sink->ysym_tile_max : nullptr);
        if (rc != PCC_OK) return rc;
    }
    // everything the range coder needs is final here: the caller's copy stream / host coder can start while the synthesis
    // transform (most of the work) is still being enqueued and executed
    if (symbols_ready) PCC_CHECK_HIP(hipEventRecord((hipEvent_t)symbols_ready, (hipStream_t)stream));
    // fixed-threshold policy (model_opt.py:27-31): the encoder-side point lists in the same call; the encoder clips (:202).  The
    // last synthesis layer writes the occupancy bits itself where it can (pcc_thr_fuse); otherwise x_hat is thresholded here.
    PCC_REQUIRE(!thr || (xyz && counts && scratch), "pcc_codec_encode: thr given but xyz / counts / scratch is NULL");
    static const bool no_fuse = getenv("PCC_NO_THR_FUSE") != nullptr;
    const pcc_thr_fuse fuse = {thr, 1, thr ? pcc_threshold_mask_of(scratch, N, D) : nullptr};
    bool fused = false;
    rc = network_forward(ctx, c->synthesis, F, c->w_synthesis, y_hat, N, D / 8, H / 8, W / 8, x_hat, workspace,
                         workspace_bytes, layer_flags, final_flags, thr && !no_fuse ? &fuse : nullptr, &fused, stream);
    if (rc != PCC_OK || !thr) return rc;
    if (fused) return pcc_threshold_from_mask(ctx, N, D, H, W, xyz, counts, cap, scratch, (hipStream_t)stream);
    return pcc_threshold_compact(ctx, x_hat, N, D, H, W, thr, 1, xyz, counts, cap, scratch, stream);
}

// V2 decoder, first phase (model_types.py:403-406): z symbols -> z_hat -> sigma -> indexes.
PCC_API int pcc_codec_decode_hyper(pcc_ctx* ctx, const pcc_codec_desc* c, int32_t* zsym, int32_t N, int32_t D, int32_t H,
                                   int32_t W, float* z_hat, float* sigma, int32_t* idx, void* workspace, size_t workspace_bytes,
                                   int32_t layer_flags, const pcc_symbol_io* io, void* stream) {
    int rc = check_codec(c);
    if (rc != PCC_OK) return rc;
    PCC_REQUIRE(c->version == 2 && ctx && zsym && z_hat && sigma && idx, "pcc_codec_decode_hyper: version-2 codec and non-NULL tensors");
    PCC_REQUIRE(D % 16 == 0 && H % 16 == 0 && W % 16 == 0, "pcc_codec_decode_hyper: block edges must be multiples of 16");
    const int F = c->filters;
    const size_t nz = (size_t)N * (D / 16) * (H / 16) * (W / 16) * F, ny = (size_t)N * (D / 8) * (H / 8) * (W / 8) * F;
    rc = check_io(io, "pcc_codec_decode_hyper");
    if (rc != PCC_OK) return rc;
    if (io && io->zsym)        // stream-order symbols as the host->device copy delivered them -> NDHWC int32 + z_hat, one launch
        rc = pcc_unpack_dequantize(ctx, io->zsym, io->sym_bytes, N, (int64_t)(nz / F / N), F, io->channels_first, zsym, c->medians, z_hat, stream);
    else
        rc = pcc_dequantize(ctx, zsym, c->medians, z_hat, nz, F, stream);
    if (rc != PCC_OK) return rc;
    rc = pcc_network_forward(ctx, PCC_NET_HYPER_SYNTHESIS, F, c->w_hyper_synthesis, z_hat, N, D / 16, H / 16, W / 16, sigma,
                             workspace, workspace_bytes, layer_flags, 0, stream);
    if (rc != PCC_OK) return rc;
    if (io && io->idx)
        return pcc_index_pack(ctx, sigma, c->scale_table, c->scale_levels, idx, N, (int64_t)(ny / F / N), F, io->channels_first, io->idx,
                              io->idx_bytes, stream);
    return pcc_scale_to_index(ctx, sigma, c->scale_table, c->scale_levels, idx, ny, stream);
}

// Decoder, main phase (model_types.py:305-307 V1, :407-408 V2): y symbols -> y_hat -> x_hat, then (optionally, thr != NULL)
// the thresholding + order-preserving compaction of model_types.py:232-234 in the same call.
PCC_API int pcc_codec_decode_main(pcc_ctx* ctx, const pcc_codec_desc* c, int32_t* ysym, int32_t N, int32_t D, int32_t H,
                                  int32_t W, float* y_hat, float* x_hat, const float* thr, float* xyz, int32_t* counts,
                                  int64_t cap, int32_t* scratch, void* workspace, size_t workspace_bytes, int32_t layer_flags,
                                  const pcc_symbol_io* io, void* stream) {
    int rc = check_codec(c);
    if (rc != PCC_OK) return rc;
    PCC_REQUIRE(ctx && ysym && y_hat && x_hat, "pcc_codec_decode_main: NULL argument");
    PCC_REQUIRE(D % 8 == 0 && H % 8 == 0 && W % 8 == 0, "pcc_codec_decode_main: block edges must be multiples of 8");
    const int F = c->filters;
    const size_t ny = (size_t)N * (D / 8) * (H / 8) * (W / 8) * F;
    rc = check_io(io, "pcc_codec_decode_main");
    if (rc != PCC_OK) return rc;
    const float* ymed = c->version == 1 ? c->medians : nullptr;
    if (io && io->ysym)
        rc = pcc_unpack_dequantize(ctx, io->ysym, io->sym_bytes, N, (int64_t)(ny / F / N), F, io->channels_first, ysym, ymed, y_hat, stream);
    else
        rc = pcc_dequantize(ctx, ysym, ymed, y_hat, ny, F, stream);
    if (rc != PCC_OK) return rc;
    PCC_REQUIRE(!thr || (xyz && counts && scratch), "pcc_codec_decode_main: thr given but xyz / counts / scratch is NULL");
    static const bool no_fuse = getenv("PCC_NO_THR_FUSE") != nullptr;
    const pcc_thr_fuse fuse = {thr, 0 /* the decoder does not clip, model_types.py:232-233 */, thr ? pcc_threshold_mask_of(scratch, N, D) : nullptr};
    bool fused = false;
    rc = network_forward(ctx, c->synthesis, F, c->w_synthesis, y_hat, N, D / 8, H / 8, W / 8, x_hat, workspace,
                         workspace_bytes, layer_flags, 0, thr && !no_fuse ? &fuse : nullptr, &fused, stream);
    if (rc != PCC_OK || !thr) return rc;
    if (fused) return pcc_threshold_from_mask(ctx, N, D, H, W, xyz, counts, cap, scratch, (hipStream_t)stream);
    return pcc_threshold_compact(ctx, x_hat, N, D, H, W, thr, 0, xyz, counts, cap, scratch, stream);
}
