// ctx.cpp — vido_create / vido_destroy / error reporting.  No CPU fallback: without a gfx950
// device vido_create fails with VIDO_E_NO_DEVICE and a message.
#include "common.hpp"
#include <cstdarg>
#include <mutex>

static std::string g_create_error;
static std::mutex g_err_mu;

int vido_set_error(vido_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->err = buf;
    else { std::lock_guard<std::mutex> g(g_err_mu); g_create_error = buf; }
    return code;
}

extern "C" {

void vido_config_default(vido_config* c)
{
    memset(c, 0, sizeof *c);
    c->device = 0; c->width = 640; c->height = 480; c->max_batch = 1;
    c->n_features = 2000; c->scale_factor = 1.2f; c->n_levels = 8; c->ini_th_fast = 20; c->min_th_fast = 7;
    c->compute_descriptors = 1; c->host_threads = 0;
}

const char* vido_last_error(const vido_ctx* ctx)
{
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> g(g_err_mu);
    return g_create_error.c_str();
}

int vido_create(const vido_config* cfg, vido_ctx** out)
{
    if (!cfg || !out) return vido_set_error(nullptr, VIDO_E_INVALID, "vido_create: null argument");
    *out = nullptr;
    if (cfg->width < 64 || cfg->height < 64 || cfg->width > 4095 || cfg->height > 4095)
        return vido_set_error(nullptr, VIDO_E_INVALID, "vido_create: frame size %dx%d outside [64,4095]", cfg->width, cfg->height);
    if (cfg->n_levels < 1 || cfg->n_levels > VIDO_MAX_LEVELS || cfg->max_batch < 1 || !(cfg->scale_factor > 1.0f))
        return vido_set_error(nullptr, VIDO_E_INVALID, "vido_create: bad n_levels/max_batch/scale_factor");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return vido_set_error(nullptr, VIDO_E_NO_DEVICE, "vido_create: no HIP device visible (%s); this library has no CPU path",
                              e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return vido_set_error(nullptr, VIDO_E_INVALID, "vido_create: device %d out of range (%d visible)", cfg->device, ndev);
    vido_ctx* ctx = new vido_ctx();
    ctx->cfg = *cfg; ctx->device = cfg->device;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(ctx->device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, ctx->device)) != hipSuccess) {
        int rc = vido_set_error(nullptr, VIDO_E_HIP, "vido_create: hipSetDevice/GetDeviceProperties: %s", hipGetErrorString(e));
        delete ctx; return rc;
    }
    snprintf(ctx->dev_name, sizeof ctx->dev_name, "%s (%s)", prop.name, prop.gcnArchName);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        int rc = vido_set_error(nullptr, VIDO_E_NO_DEVICE, "vido_create: device %d is %s; kernels are built for gfx950 only", ctx->device, prop.gcnArchName);
        delete ctx; return rc;
    }
    // highest stream priority: the tracker's kernels are short and latency-bound; when the network nodes share the GPU (pipeline.py: the networks of
    // frame k+1 overlap the tracking of frame k) their long convolution kernels must not queue in front of them
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_greatest = 0;
    if (const char* pe = getenv("VIDO_CTX_PRIO")) {             // experiments: "normal" / "least" instead of the greatest priority; "skipN": N throw-away streams first (shifts the
        if (!strcmp(pe, "normal")) prio_greatest = 0;           // runtime's round-robin stream -> hardware-queue assignment)
        else if (!strcmp(pe, "least")) prio_greatest = prio_least;
        else if (!strncmp(pe, "skip", 4)) { for (int i = 0; i < atoi(pe + 4); i++) { hipStream_t t; if (hipStreamCreateWithPriority(&t, hipStreamNonBlocking, prio_greatest) != hipSuccess) break; } }
    }
    if ((e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_greatest)) != hipSuccess ||
        (e = hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_greatest)) != hipSuccess) {
        int rc = vido_set_error(nullptr, VIDO_E_HIP, "vido_create: hipStreamCreate: %s", hipGetErrorString(e));
        delete ctx; return rc;
    }
    // (conv1x1.hip) the range flag of the split-fp16 GEMM: a pinned host word — allocated here, not at the first launch, because that launch may sit inside a stream capture
    if (hipHostMalloc((void**)&ctx->c1_range_flag, sizeof(unsigned), hipHostMallocDefault) == hipSuccess) *ctx->c1_range_flag = 0u; else { ctx->c1_range_flag = nullptr; (void)hipGetLastError(); }
    int rc = orb_state_create(ctx);
    if (rc != VIDO_OK) { vido_set_error(nullptr, rc, "%s", ctx->err.c_str()); vido_destroy(ctx); return rc; }
    *out = ctx;
    return VIDO_OK;
}

void vido_destroy(vido_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    vido_rccl_destroy(ctx);
    orb_state_destroy(ctx);
    track_state_destroy(ctx);
    ham_state_destroy(ctx);
    pose_state_destroy(ctx);
    ba_state_destroy(ctx);
    net_state_destroy(ctx);
    pnp_state_destroy(ctx);
    bawin_state_destroy(ctx);
    if (ctx->detpost_buf) { hipFree(ctx->detpost_buf); ctx->detpost_buf = nullptr; }
    if (ctx->c1_range_flag) { hipHostFree(ctx->c1_range_flag); ctx->c1_range_flag = nullptr; }
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    if (ctx->stream2) hipStreamDestroy(ctx->stream2);
    delete ctx;
}

int vido_device_name(const vido_ctx* ctx, char* buf, int buflen)
{
    if (!ctx || !buf || buflen <= 0) return VIDO_E_INVALID;
    snprintf(buf, buflen, "%s", ctx->dev_name);
    return VIDO_OK;
}

void* vido_stream(vido_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int vido_set_stream(vido_ctx* ctx, void* hip_stream, int enable)
{
    if (!ctx) return VIDO_E_INVALID;
    ctx->ext_stream = (hipStream_t)hip_stream; ctx->has_ext_stream = enable != 0;
    return VIDO_OK;
}

int vido_stream_wait_event(vido_ctx* ctx, void* hip_event)
{
    if (!ctx || !hip_event) return VIDO_E_INVALID;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)hip_event, 0));
    return VIDO_OK;
}

int vido_synchronize(vido_ctx* ctx)
{
    if (!ctx) return VIDO_E_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VIDO_OK;
}

}  // extern "C"

// ---- VIDO_CALL_PROF: named host-side sections (common.hpp) -------------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
namespace {
struct SectionProf {
    bool on = getenv("VIDO_CALL_PROF") != nullptr; std::mutex m; std::map<std::string, std::pair<double, long> > acc;
    ~SectionProf() { if (!on) return; std::vector<std::pair<double, std::string> > v; for (auto& kv : acc) v.push_back({kv.second.first, kv.first});
                     std::sort(v.rbegin(), v.rend());
                     for (auto& e : v) fprintf(stderr, "[section prof] %-44s %8.3f ms total %7ld calls %8.3f ms each\n", e.second.c_str(), e.first, acc[e.second].second, e.first / std::max(1L, acc[e.second].second)); }
};
SectionProf g_sections;
}
bool vido_prof_on() { return g_sections.on; }
void vido_prof_add(const char* name, double ms) { std::lock_guard<std::mutex> g(g_sections.m); auto& a = g_sections.acc[name]; a.first += ms; a.second++; }
