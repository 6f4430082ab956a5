// VAE decode stage behind the forward-level C ABI: gl_vae_create / gl_vae_load_weights / gl_vae_decode.
//
// Replaces AutoencoderKL.decode (GLIGEN/ldm/models/autoencoder.py:40-44) and Decoder.forward
// (GLIGEN/ldm/modules/diffusionmodules/model.py:535-568): z / scale_factor -> post_quant_conv (1x1, applied in fp32 while
// packing the latent) -> conv_in -> mid (ResnetBlock, single-head AttnBlock model.py:150-202, ResnetBlock) -> the up levels
// (num_res_blocks + 1 ResnetBlocks each, nearest-2x upsample + conv between levels) -> GroupNorm(eps 1e-6) + swish ->
// conv_out.  Like the UNet handle the engine owns the plan, the LIBRARY-DEFINED flat weight layout (gl_vae_weight_at: the
// host packer fills it, the same buffer travels in the multi-GPU broadcast), a grow-only activation pool with stable
// addresses and one hipGraph per (batch, latent side), captured on an engine-owned stream and replayed on the caller's.
// No new heavy kernels: every conv / 1x1 conv / GroupNorm goes through gl_conv3x3 / gl_gemm / gl_groupnorm (and therefore
// through the 8-wave deep-pipelined kernel wherever its dispatch applies).
#include "common.h"
#include "gligen_hip.h"
#include "opts.h"

#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
constexpr int VCIN_PAD = 64;
constexpr int64_t VALIGN = 256;
struct VW {
    int64_t off, bytes;
    int dtype, ndim;
    int64_t shape[4];
};
}  // namespace

struct gl_vae {
    gl_vae_config cfg;
    std::vector<std::string> names;
    std::unordered_map<std::string, VW> tab;
    int64_t total = 0;
    const char* wbase = nullptr;
    struct Buf { void* p; size_t bytes; };
    std::unordered_map<std::string, Buf> pool;
    bool pool_changed = false;
    std::map<std::pair<int, int>, hipGraphExec_t> graphs;
    hipStream_t cap_stream = nullptr;
    int launches = 0;
    int opt_epoch = 0;
    int ovr_epoch = 0;
    gl_opt_overrides ovr;              // per-handle option overrides (gl_vae_set_option)

    void add(const std::string& n, int dtype, std::initializer_list<int64_t> shp) {
        VW w{};
        w.dtype = dtype;
        w.ndim = (int)shp.size();
        int64_t numel = 1;
        int i = 0;
        for (auto s : shp) { w.shape[i++] = s; numel *= s; }
        w.bytes = numel * (dtype == 0 ? 2 : 4);
        w.off = total;
        total += (w.bytes + VALIGN - 1) / VALIGN * VALIGN;
        names.push_back(n);
        tab[n] = w;
    }
    template <typename T>
    const T* W(const std::string& n) const {
        auto it = tab.find(n);
        return it == tab.end() ? nullptr : reinterpret_cast<const T*>(wbase + it->second.off);
    }
    void* get(const std::string& tag, size_t bytes) {
        auto it = pool.find(tag);
        if (it != pool.end() && it->second.bytes >= bytes) return it->second.p;
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        if (it != pool.end()) { (void)hipFree(it->second.p); it->second = Buf{p, bytes}; }
        else pool[tag] = Buf{p, bytes};
        pool_changed = true;
        return p;
    }
    half_t* f16(const std::string& tag, size_t n) { return reinterpret_cast<half_t*>(get(tag, n * 2)); }
    float* f32(const std::string& tag, size_t n) { return reinterpret_cast<float*>(get(tag, n * 4)); }
    void drop_graphs() {
        for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
        graphs.clear();
    }
};


namespace {

#define VCK(x)                    \
    do {                          \
        const int e__ = (x);      \
        if (e__ != 0) return e__; \
    } while (0)
#define VCKP(p) \
    if ((p) == nullptr) return GL_ERR_BAD_ARG

void plan_resnet(gl_vae* v, const std::string& p, int cin, int cout) {
    v->add(p + ".norm1.g", 1, {cin}); v->add(p + ".norm1.b", 1, {cin});
    v->add(p + ".conv1.w", 0, {cout, 9 * (int64_t)cin}); v->add(p + ".conv1.b", 1, {cout});
    v->add(p + ".norm2.g", 1, {cout}); v->add(p + ".norm2.b", 1, {cout});
    v->add(p + ".conv2.w", 0, {cout, 9 * (int64_t)cout}); v->add(p + ".conv2.b", 1, {cout});
    if (cin != cout) { v->add(p + ".nin_shortcut.w", 0, {cout, cin}); v->add(p + ".nin_shortcut.b", 1, {cout}); }
}

int gn_nchunk(int HW) {
    if (HW <= 4096) { int c = HW / 4; return c < 1 ? 1 : (c > 64 ? 64 : c); }
    const int c = HW / 512;
    return c > 512 ? 512 : c;
}

struct VRun {
    gl_vae* v;
    hipStream_t st;
    int B;
    int* launches;
    void count(int n = 1) { if (launches) *launches += n; }
};

int v_gn(VRun& r, const half_t* x, int C, int HW, const std::string& p, bool silu, const std::string& tag, half_t** out) {
    gl_vae* v = r.v;
    const int nchunk = gn_nchunk(HW);
    float* partial = v->f32("gn.partial", (size_t)r.B * nchunk * 64);
    half_t* y = v->f16(tag, (size_t)r.B * HW * C);
    VCKP(partial); VCKP(y);
    VCK(gl_groupnorm(x, C, nullptr, 0, r.B, HW, v->W<float>(p + ".g"), v->W<float>(p + ".b"), 1e-6f, silu ? 1 : 0, y, partial, nchunk, r.st));
    r.count(gl_groupnorm_launches(C, HW));
    *out = y;
    return 0;
}

int v_conv(VRun& r, const half_t* x, int side, int cin, const std::string& p, int cout, int up, int epi, const void* res, void* out,
           int out_mode = GL_OUT_F16_ROWMAJOR) {
    gl_vae* v = r.v;
    gl_conv_args a;
    memset(&a, 0, sizeof(a));
    a.in = x;
    a.B = r.B; a.Hin = side; a.Win = side; a.Cin = cin;
    a.Hout = up ? 2 * side : side; a.Wout = a.Hout;
    a.stride = 1; a.upsample2x = up;
    a.g.w = v->W<half_t>(p + ".w");
    a.g.bias = v->W<float>(p + ".b");
    a.g.N = cout;
    a.g.epi = epi;
    a.g.out_mode = out_mode;
    a.g.out = out;
    a.g.ldc = out_mode == GL_OUT_F32_NCHW ? 0 : cout;
    a.g.hw = out_mode == GL_OUT_F32_NCHW ? a.Hout * a.Wout : 0;
    a.g.res = res; a.g.ldres = cout;
    float* ws = v->f32("ws", (size_t)(96ll << 20) / 4);
    VCKP(ws);
    a.g.workspace = ws; a.g.workspace_bytes = 96ll << 20;
    VCK(gl_conv3x3(&a, r.st));
    r.count();
    return 0;
}

int v_gemm(VRun& r, const half_t* a_, int lda, const half_t* w, const float* bias, int M, int N, int K, int epi, const void* res, half_t* out,
           int ldc) {
    gl_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.a = a_; g.lda = lda; g.w = w; g.bias = bias; g.M = M; g.N = N; g.K = K; g.epi = epi;
    g.out_mode = GL_OUT_F16_ROWMAJOR; g.out = out; g.ldc = ldc; g.res = res; g.ldres = N;
    float* ws = r.v->f32("ws", (size_t)(96ll << 20) / 4);
    VCKP(ws);
    g.workspace = ws; g.workspace_bytes = 96ll << 20;
    VCK(gl_gemm(&g, r.st));
    r.count();
    return 0;
}

int v_resnet(VRun& r, const std::string& p, const half_t* x, int side, int cin, int cout, const std::string& tag, half_t** out) {
    gl_vae* v = r.v;
    const int HW = side * side;
    const size_t M = (size_t)r.B * HW;
    half_t *t1, *t2;
    VCK(v_gn(r, x, cin, HW, p + ".norm1", true, "rn.gn." + std::to_string(cin) + "." + std::to_string(side), &t1));
    half_t* h = v->f16("rn.h." + std::to_string(cout) + "." + std::to_string(side), M * cout);
    VCKP(h);
    VCK(v_conv(r, t1, side, cin, p + ".conv1", cout, 0, GL_EPI_BIAS, nullptr, h));
    VCK(v_gn(r, h, cout, HW, p + ".norm2", true, "rn.gn." + std::to_string(cout) + "." + std::to_string(side), &t2));
    const half_t* sk = x;
    if (cin != cout) {
        half_t* s = v->f16("rn.sk." + std::to_string(cout) + "." + std::to_string(side), M * cout);
        VCKP(s);
        VCK(v_gemm(r, x, cin, v->W<half_t>(p + ".nin_shortcut.w"), v->W<float>(p + ".nin_shortcut.b"), (int)M, cout, cin, GL_EPI_BIAS, nullptr, s, cout));
        sk = s;
    }
    half_t* o = v->f16(tag, M * cout);
    VCKP(o);
    VCK(v_conv(r, t2, side, cout, p + ".conv2", cout, 0, GL_EPI_RES, sk, o));
    *out = o;
    return 0;
}

// single-head AttnBlock (model.py:150-202): d = C = 512 is beyond the flash kernel's register budget; Q.K^T (GEMM) -> row softmax
// -> P.V (GEMM against V^T), per sample; C^-0.5 is folded into the q weights at pack time
int v_attn(VRun& r, const std::string& p, const half_t* x, int side, int C, const std::string& tag, half_t** out) {
    gl_vae* v = r.v;
    const int N = side * side, B = r.B;
    const size_t M = (size_t)B * N;
    half_t* hn;
    VCK(v_gn(r, x, C, N, p + ".norm", false, "at.gn", &hn));
    half_t *q = v->f16("at.q", M * C), *k = v->f16("at.k", M * C), *vv = v->f16("at.v", M * C), *o = v->f16("at.o", M * C);
    VCKP(q); VCKP(k); VCKP(vv); VCKP(o);
    VCK(v_gemm(r, hn, C, v->W<half_t>(p + ".q.w"), v->W<float>(p + ".q.b"), (int)M, C, C, GL_EPI_BIAS, nullptr, q, C));
    VCK(v_gemm(r, hn, C, v->W<half_t>(p + ".k.w"), v->W<float>(p + ".k.b"), (int)M, C, C, GL_EPI_BIAS, nullptr, k, C));
    VCK(v_gemm(r, hn, C, v->W<half_t>(p + ".v.w"), v->W<float>(p + ".v.b"), (int)M, C, C, GL_EPI_BIAS, nullptr, vv, C));
    const int Np = (N + 63) / 64 * 64;
    const int Hs = 4;
    half_t* vt = v->f16("at.vt", (size_t)B * C * Np);
    half_t* s = v->f16("at.s", (size_t)B * N * Np);
    VCKP(vt); VCKP(s);
    VCK(gl_transpose_v(vv, (int64_t)N * C, C, vt, Np, B, Hs, C / Hs, N, r.st));
    r.count();
    if (Np != N) {
        if (hipMemsetAsync(s, 0, (size_t)B * N * Np * 2, r.st) != hipSuccess) return GL_ERR_BAD_ARG;
    }
    for (int b = 0; b < B; ++b) {
        half_t* sb = s + (size_t)b * N * Np;
        VCK(v_gemm(r, q + (size_t)b * N * C, C, k + (size_t)b * N * C, nullptr, N, N, C, GL_EPI_BIAS, nullptr, sb, Np));
        VCK(gl_softmax_rows(sb, N, N, Np, 1.0f, r.st));
        r.count();
        VCK(v_gemm(r, sb, Np, vt + (size_t)b * C * Np, nullptr, N, C, Np, GL_EPI_BIAS, nullptr, o + (size_t)b * N * C, C));
    }
    half_t* y = v->f16(tag, M * C);
    VCKP(y);
    VCK(v_gemm(r, o, C, v->W<half_t>(p + ".proj_out.w"), v->W<float>(p + ".proj_out.b"), (int)M, C, C, GL_EPI_RES, x, y, C));
    *out = y;
    return 0;
}

int launch_decode(gl_vae* v, int B, int side, hipStream_t st, int* launches) {
    const gl_vae_config& c = v->cfg;
    VRun r{v, st, B, launches};
    if (launches) *launches = 0;
    const int nres = c.n_mult;
    int ch = c.ch * c.ch_mult[nres - 1];
    const float* z = v->f32("in.z", (size_t)B * c.z_channels * side * side);
    VCKP(z);
    half_t* xin = v->f16("in", (size_t)B * side * side * VCIN_PAD);
    VCKP(xin);
    VCK(gl_latent_affine_pack(z, v->W<float>("post_quant_conv.w"), v->W<float>("post_quant_conv.b"), 1.0f / c.scale_factor, B, c.z_channels,
                              side * side, VCIN_PAD, xin, st));
    r.count();
    half_t* h = v->f16("conv_in", (size_t)B * side * side * ch);
    VCKP(h);
    VCK(v_conv(r, xin, side, VCIN_PAD, "decoder.conv_in", ch, 0, GL_EPI_BIAS, nullptr, h));
    VCK(v_resnet(r, "decoder.mid.block_1", h, side, ch, ch, "mid.1", &h));
    VCK(v_attn(r, "decoder.mid.attn_1", h, side, ch, "mid.a", &h));
    VCK(v_resnet(r, "decoder.mid.block_2", h, side, ch, ch, "mid.2", &h));
    for (int lvl = nres - 1; lvl >= 0; --lvl) {
        const int cout = c.ch * c.ch_mult[lvl];
        for (int i = 0; i <= c.num_res_blocks; ++i) {
            VCK(v_resnet(r, "decoder.up." + std::to_string(lvl) + ".block." + std::to_string(i), h, side, ch, cout,
                         "up." + std::to_string(lvl) + "." + std::to_string(i), &h));
            ch = cout;
        }
        if (lvl != 0) {
            half_t* u = v->f16("up." + std::to_string(lvl) + ".u", (size_t)B * 4 * side * side * ch);
            VCKP(u);
            VCK(v_conv(r, h, side, ch, "decoder.up." + std::to_string(lvl) + ".upsample.conv", ch, 1, GL_EPI_BIAS, nullptr, u));
            h = u;
            side *= 2;
        }
    }
    half_t* g;
    VCK(v_gn(r, h, ch, side * side, "decoder.norm_out", true, "fin.gn", &g));
    float* out = v->f32("out", (size_t)B * c.out_ch * side * side);
    VCKP(out);
    VCK(v_conv(r, g, side, ch, "decoder.conv_out", c.out_ch, 0, GL_EPI_BIAS, nullptr, out, GL_OUT_F32_NCHW));
    return 0;
}

}  // namespace

extern "C" int gl_vae_create(const gl_vae_config* cfg, gl_vae** out) {
    if (!cfg || !out || cfg->n_mult < 1 || cfg->n_mult > 8 || cfg->ch <= 0 || (cfg->ch % 64) || cfg->num_res_blocks < 0 || cfg->z_channels <= 0 ||
        cfg->z_channels > VCIN_PAD || cfg->out_ch <= 0 || cfg->scale_factor == 0.0f)
        return GL_ERR_BAD_ARG;
    gl_vae* v = new gl_vae();
    v->cfg = *cfg;
    const int nres = cfg->n_mult;
    int ch = cfg->ch * cfg->ch_mult[nres - 1];
    v->add("post_quant_conv.w", 1, {cfg->z_channels, cfg->embed_dim > 0 ? cfg->embed_dim : cfg->z_channels});
    v->add("post_quant_conv.b", 1, {cfg->z_channels});
    v->add("decoder.conv_in.w", 0, {ch, 9 * (int64_t)VCIN_PAD});
    v->add("decoder.conv_in.b", 1, {ch});
    plan_resnet(v, "decoder.mid.block_1", ch, ch);
    const std::string ap = "decoder.mid.attn_1";
    v->add(ap + ".norm.g", 1, {ch}); v->add(ap + ".norm.b", 1, {ch});
    for (const char* n : {"q", "k", "v", "proj_out"}) { v->add(ap + "." + n + ".w", 0, {ch, ch}); v->add(ap + "." + n + ".b", 1, {ch}); }
    plan_resnet(v, "decoder.mid.block_2", ch, ch);
    for (int lvl = nres - 1; lvl >= 0; --lvl) {
        const int cout = cfg->ch * cfg->ch_mult[lvl];
        for (int i = 0; i <= cfg->num_res_blocks; ++i) {
            plan_resnet(v, "decoder.up." + std::to_string(lvl) + ".block." + std::to_string(i), ch, cout);
            ch = cout;
        }
        if (lvl != 0) {
            v->add("decoder.up." + std::to_string(lvl) + ".upsample.conv.w", 0, {ch, 9 * (int64_t)ch});
            v->add("decoder.up." + std::to_string(lvl) + ".upsample.conv.b", 1, {ch});
        }
    }
    v->add("decoder.norm_out.g", 1, {ch}); v->add("decoder.norm_out.b", 1, {ch});
    v->add("decoder.conv_out.w", 0, {cfg->out_ch, 9 * (int64_t)ch});
    v->add("decoder.conv_out.b", 1, {cfg->out_ch});
    *out = v;
    return 0;
}

extern "C" int gl_vae_destroy(gl_vae* v) {
    if (!v) return GL_ERR_BAD_ARG;
    v->drop_graphs();
    for (auto& kv : v->pool) (void)hipFree(kv.second.p);
    if (v->cap_stream) (void)hipStreamDestroy(v->cap_stream);
    delete v;
    return 0;
}

extern "C" int gl_vae_num_weights(const gl_vae* v) { return v ? (int)v->names.size() : GL_ERR_BAD_ARG; }
extern "C" int64_t gl_vae_weights_bytes(const gl_vae* v) { return v ? v->total : -1; }
extern "C" int gl_vae_weight_at(const gl_vae* v, int32_t i, gl_weight_info* info) {
    if (!v || !info || i < 0 || i >= (int)v->names.size()) return GL_ERR_BAD_ARG;
    const std::string& n = v->names[i];
    if (n.size() >= sizeof(info->name)) return GL_ERR_BAD_ARG;
    const VW& w = v->tab.at(n);
    memset(info, 0, sizeof(*info));
    memcpy(info->name, n.c_str(), n.size());
    info->offset = w.off; info->nbytes = w.bytes; info->dtype = w.dtype; info->ndim = w.ndim;
    for (int k = 0; k < 4; ++k) info->shape[k] = w.shape[k];
    return 0;
}

extern "C" int gl_vae_load_weights(gl_vae* v, const void* packed, int64_t bytes, void* stream) {
    (void)stream;
    if (!v || !packed || bytes < v->total || (reinterpret_cast<uintptr_t>(packed) % 16)) return GL_ERR_BAD_ARG;
    v->wbase = reinterpret_cast<const char*>(packed);      // referenced, not copied: the caller keeps the buffer alive
    v->drop_graphs();
    return 0;
}

extern "C" int gl_vae_decode(gl_vae* v, const float* z, int32_t B, int32_t side, float* out, int32_t use_graph, void* stream) {
    if (!v || !z || !out || B <= 0 || side <= 0 || !v->wbase) return GL_ERR_BAD_ARG;
    gl_opts_scope opts_scope(v->ovr);
    const gl_vae_config& c = v->cfg;
    hipStream_t st = (hipStream_t)stream;
    const size_t nz = (size_t)B * c.z_channels * side * side;
    int oside = side;
    for (int l = 1; l < c.n_mult; ++l) oside *= 2;
    const size_t no = (size_t)B * c.out_ch * oside * oside;
    v->pool_changed = false;
    float* zin = v->f32("in.z", nz);
    float* obuf = v->f32("out", no);
    VCKP(zin); VCKP(obuf);
    if (hipMemcpyAsync(zin, z, nz * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return GL_ERR_BAD_ARG;
    if (v->opt_epoch != g_gl_option_epoch || v->ovr_epoch != v->ovr.epoch) {
        v->drop_graphs();
        v->opt_epoch = g_gl_option_epoch;
        v->ovr_epoch = v->ovr.epoch;
    }
    const auto key = std::make_pair((int)B, (int)side);
    auto it = v->graphs.find(key);
    if (use_graph && it == v->graphs.end()) {
        VCK(launch_decode(v, B, side, st, &v->launches));        // warm-up: allocates every pooled buffer
        if (hipStreamSynchronize(st) != hipSuccess) return GL_ERR_BAD_ARG;
        if (v->pool_changed) { v->drop_graphs(); v->pool_changed = false; }
        if (v->cap_stream == nullptr && hipStreamCreateWithFlags(&v->cap_stream, hipStreamNonBlocking) != hipSuccess) return GL_ERR_UNSUPPORTED;
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(v->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return GL_ERR_UNSUPPORTED;
        const int rc = launch_decode(v, B, side, v->cap_stream, nullptr);
        const hipError_t ec = hipStreamEndCapture(v->cap_stream, &graph);
        if (rc != 0 || ec != hipSuccess || graph == nullptr) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc != 0 ? rc : GL_ERR_UNSUPPORTED;
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) return GL_ERR_UNSUPPORTED;
        v->graphs[key] = exec;
        // the warm-up run already produced this call's result
    } else if (use_graph) {
        if (hipGraphLaunch(it->second, st) != hipSuccess) return GL_ERR_BAD_ARG;
    } else {
        VCK(launch_decode(v, B, side, st, &v->launches));
        if (v->pool_changed) { v->drop_graphs(); v->pool_changed = false; }
    }
    if (hipMemcpyAsync(out, obuf, no * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return GL_ERR_BAD_ARG;
    return 0;
}

extern "C" int gl_vae_set_option(gl_vae* v, int key, int value) {
    gl_opts probe{};
    if (!v || key < 0 || key >= GL_OPT_MAX || !gl_opts_store(probe, key, value)) return GL_ERR_BAD_ARG;
    v->ovr.mask |= (uint64_t)1 << key;
    v->ovr.v[key] = value;
    ++v->ovr.epoch;
    return 0;
}

extern "C" int gl_vae_num_launches(const gl_vae* v) { return v ? v->launches : GL_ERR_BAD_ARG; }
extern "C" int64_t gl_vae_pool_bytes(const gl_vae* v) {
    if (!v) return -1;
    int64_t s = 0;
    for (auto& kv : v->pool) s += (int64_t)kv.second.bytes;
    return s;
}
extern "C" int gl_sizeof_vae_config(void) { return (int)sizeof(gl_vae_config); }
