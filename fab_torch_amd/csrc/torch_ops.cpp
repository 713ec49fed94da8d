// PyTorch-ROCm custom-op layer of fabhip: TORCH_LIBRARY(fabhip, ...) over the C ABI of libfabhip.so
// (include/fabhip.h).  This is how the Python mirror of fab-torch's plug-in interfaces reaches the HIP kernels
// (north_star: "called from Python through PyTorch-ROCm custom ops"; SURVEY.md section 8b, "Torch op layer").
//
//  * every op is registered for the CUDA (= HIP on ROCm) dispatch key ONLY: there is no CPU kernel, a CPU tensor
//    fails in the dispatcher ("could not run fabhip::... with arguments from the 'CPU' backend");
//  * ops only translate tensors into the raw-pointer argument structs of the C ABI, allocate outputs / scratch
//    through torch's caching allocator (stream-ordered, so scratch is never shared between streams) and enqueue on
//    the current HIP stream; no synchronisation, no host reads;
//  * a non-zero fabhip return code becomes a c10::Error (Python RuntimeError) via TORCH_CHECK;
//  * autograd for the training op (fabhip::realnvp_logprob_tape) is registered from Python with
//    torch.library.register_autograd (fab_torch_amd/_ops.py); its backward is fabhip::realnvp_param_grad.
//
// Flow parameters travel as `Tensor[] params` in the order of fabhip_flow_params: per layer
// {w1, b1, w2, b2, w3, b3, L, U, log_S, sign_S, P}, then {loc, log_scale}, then (act_norm flows only) one
// {ActNorm.s, ActNorm.t} pair per layer.  Targets travel as
// (int kind, float[] {a, b, c, log_norm}, Tensor? locs, Tensor? scales).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>
#include <torch/csrc/distributed/c10d/GroupRegistry.hpp>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fabhip.h"

namespace {

using at::Tensor;
using c10::optional;

fabhip_stream_t stream_of(const Tensor& t) {
    return (fabhip_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}

void chk(int rc, const char* what) {
    TORCH_CHECK(rc == FABHIP_OK, "fabhip ", what, " failed: ", fabhip_strerror(rc), " (code ", rc, ")");
}

void need(const Tensor& t, at::ScalarType st, const char* name) {
    TORCH_CHECK(t.is_cuda(), "fabhip: ", name, " must live on the GPU (no CPU path)");
    TORCH_CHECK(t.scalar_type() == st, "fabhip: ", name, " has dtype ", t.scalar_type(), ", expected ", st);
    TORCH_CHECK(t.is_contiguous(), "fabhip: ", name, " must be contiguous");
}
const float* fp(const Tensor& t, const char* name) { need(t, at::kFloat, name); return t.data_ptr<float>(); }
float* fpm(const Tensor& t, const char* name) { need(t, at::kFloat, name); return t.data_ptr<float>(); }
float* fpm_opt(const optional<Tensor>& t, const char* name) { return t.has_value() ? fpm(*t, name) : nullptr; }
const float* fp_opt(const optional<Tensor>& t, const char* name) { return t.has_value() ? fp(*t, name) : nullptr; }

// The kernels index raw pointers: every tensor whose pointer goes into an argument struct is checked for its element
// count and for living on the device of the tensor the launch is enqueued for (`ref`).  `at_least`: n is a lower bound.
void need_n(const Tensor& t, int64_t n, const Tensor& ref, const char* name, bool at_least = false) {
    TORCH_CHECK(t.device() == ref.device(), "fabhip: ", name, " is on ", t.device(), ", expected ", ref.device());
    TORCH_CHECK(at_least ? t.numel() >= n : t.numel() == n, "fabhip: ", name, " has ", t.numel(), " elements, expected ",
                at_least ? "at least " : "", n);
}
const float* fpn(const Tensor& t, int64_t n, const Tensor& ref, const char* name) { need_n(t, n, ref, name); return fp(t, name); }
float* fpmn(const Tensor& t, int64_t n, const Tensor& ref, const char* name, bool at_least = false) {
    need_n(t, n, ref, name, at_least);
    return fpm(t, name);
}
float* fpmn_opt(const optional<Tensor>& t, int64_t n, const Tensor& ref, const char* name, bool at_least = false) {
    return t.has_value() ? fpmn(*t, n, ref, name, at_least) : nullptr;
}

Tensor fempty(at::IntArrayRef shape, const Tensor& like) { return at::empty(shape, like.options().dtype(at::kFloat)); }
Tensor scratch(size_t bytes, const Tensor& like) {
    return at::empty({(int64_t)(bytes + 256)}, like.options().dtype(at::kByte));
}
void* aligned(const Tensor& ws) {
    auto p = (uintptr_t)ws.data_ptr();
    return (void*)((p + 255) & ~(uintptr_t)255);
}

fabhip_flow make_flow(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width, int64_t precision = 0) {
    fabhip_flow f;
    TORCH_CHECK(precision >= 0 && precision <= 2, "fabhip: precision must be 0 (process default), 1 (fp32) or 2 (fast)");
    f.dim = (int32_t)dim; f.n_layers = (int32_t)n_layers; f.width = (int32_t)width; f.precision = (int32_t)precision;
    f.packed = fp(packed, "packed flow image");
    const int64_t n = fabhip_flow_packed_floats(f.dim, f.n_layers, f.width);
    TORCH_CHECK(n > 0, "fabhip: flow shape not supported (dim ", dim, ", width ", width, ")");
    TORCH_CHECK(packed.numel() == n, "fabhip: packed image has ", packed.numel(), " floats, expected ", n);
    return f;
}

void fill_params(fabhip_flow_params& p, at::TensorList params, int64_t dim, int64_t n_layers, int64_t width) {
    TORCH_CHECK(n_layers >= 1 && n_layers <= FABHIP_MAX_LAYERS, "fabhip: n_layers out of range");
    const bool act_norm = (int64_t)params.size() == 13 * n_layers + 2;
    TORCH_CHECK(act_norm || (int64_t)params.size() == 11 * n_layers + 2, "fabhip: expected ", 11 * n_layers + 2,
                " parameter tensors (11 per layer + loc + log_scale), or ", 13 * n_layers + 2,
                " with one ActNorm {s, t} pair per layer appended, got ", params.size());
    p.dim = (int32_t)dim; p.n_layers = (int32_t)n_layers; p.width = (int32_t)width;
    const int64_t d = (dim + 1) / 2, DO = dim - d;     // make_normflow_model.py:17 `int(dim / 2 + 0.5)`
    const int64_t want[11] = {width * d, width, width * width, width, 2 * DO * width, 2 * DO, dim * dim, dim * dim, dim, dim,
                              dim * dim};
    static const char* names[11] = {"w1", "b1", "w2", "b2", "w3", "b3", "L", "U", "log_S", "sign_S", "P"};
    for (int64_t k = 0; k < n_layers; ++k) {
        const Tensor* t = &params[11 * k];
        for (int i = 0; i < 11; ++i) need_n(t[i], want[i], params[0], names[i]);
        p.w1[k] = fp(t[0], "w1"); p.b1[k] = fp(t[1], "b1"); p.w2[k] = fp(t[2], "w2"); p.b2[k] = fp(t[3], "b2");
        p.w3[k] = fp(t[4], "w3"); p.b3[k] = fp(t[5], "b3"); p.lu_L[k] = fp(t[6], "L"); p.lu_U[k] = fp(t[7], "U");
        p.log_S[k] = fp(t[8], "log_S"); p.sign_S[k] = fp(t[9], "sign_S"); p.perm_P[k] = fp(t[10], "P");
    }
    p.loc = fpn(params[11 * n_layers], dim, params[0], "loc");
    p.log_scale = fpn(params[11 * n_layers + 1], dim, params[0], "log_scale");
    for (int64_t k = 0; k < FABHIP_MAX_LAYERS; ++k) { p.an_s[k] = nullptr; p.an_t[k] = nullptr; }
    if (act_norm)
        for (int64_t k = 0; k < n_layers; ++k) {
            const Tensor &s = params[11 * n_layers + 2 + 2 * k], &t = params[11 * n_layers + 3 + 2 * k];
            need_n(s, dim, params[0], "ActNorm.s"); need_n(t, dim, params[0], "ActNorm.t");
            p.an_s[k] = fp(s, "ActNorm.s"); p.an_t[k] = fp(t, "ActNorm.t");
        }
}

fabhip_target make_target(int64_t kind, at::ArrayRef<double> prm, const optional<Tensor>& locs,
                          const optional<Tensor>& scales, int64_t dim) {
    TORCH_CHECK(prm.size() == 4, "fabhip: target parameters are {a, b, c, log_norm}");
    fabhip_target t;
    t.kind = (int32_t)kind; t.dim = (int32_t)dim;
    t.a = (float)prm[0]; t.b = (float)prm[1]; t.c = (float)prm[2]; t.log_norm = (float)prm[3];
    t.n_mix = 0; t.locs = nullptr; t.scales = nullptr;
    if (kind == FABHIP_TARGET_GMM) {
        TORCH_CHECK(locs.has_value() && scales.has_value(), "fabhip: GMM target needs locs and scales");
        t.locs = fp(*locs, "locs"); t.scales = fp(*scales, "scales");
        t.n_mix = (int32_t)locs->size(0);
        TORCH_CHECK(locs->dim() == 2 && locs->size(1) == dim && scales->sizes() == locs->sizes(), "fabhip: GMM shapes");
    }
    return t;
}

fabhip_anneal coefs(double beta, double alpha, bool p_target) {
    fabhip_anneal a;
    fabhip_anneal_coefs(beta, alpha, p_target ? 1 : 0, &a);
    return a;
}

// ------------------------------------------------------------------------------------------------------------------
// geometry queries (no tensors: catch-all kernels)
// ------------------------------------------------------------------------------------------------------------------
int64_t abi_version() { return fabhip_version(); }
int64_t flow_packed_floats(int64_t dim, int64_t n_layers, int64_t width) {
    return fabhip_flow_packed_floats((int32_t)dim, (int32_t)n_layers, (int32_t)width);
}
// Identity of a parameter set: two independent 64-bit mixes over (storage address, version counter) of every tensor, in
// order - what flow.native() compares to decide whether the packed image is current.  One call instead of two Python
// attribute reads per tensor (112 tensors for the headline flow: 90 us of host time per AIS call, during which the GPU idles).
std::vector<int64_t> tensors_key(at::TensorList ts) {
    uint64_t h1 = 0x9E3779B97F4A7C15ull, h2 = 0xC2B2AE3D27D4EB4Full;
    auto mix = [](uint64_t h, uint64_t v, uint64_t m) {
        h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= m;
        return h ^ (h >> 29);
    };
    for (const Tensor& t : ts) {
        const uint64_t p = t.defined() ? (uint64_t)(uintptr_t)t.data_ptr() : 0, v = t.defined() ? (uint64_t)t._version() : 0;
        h1 = mix(mix(h1, p, 0xBF58476D1CE4E5B9ull), v, 0x94D049BB133111EBull);
        h2 = mix(mix(h2, v ^ 0x5555555555555555ull, 0xD6E8FEB86659FD93ull), p, 0xFF51AFD7ED558CCDull);
    }
    return {(int64_t)h1, (int64_t)h2, (int64_t)ts.size()};
}
// The same for a parameter set registered once: the op layer keeps the tensor handles (the Python side keeps the objects alive
// and re-registers when the set itself changes), so the per-call check passes ONE integer through the dispatcher instead of a
// list of 112 tensors.  `param.data = ...`, `.to()`, load_state_dict and in-place updates all act on the registered TensorImpl.
static std::mutex g_keysets_mu;
static std::vector<std::vector<Tensor>> g_keysets;
static std::vector<int64_t> g_keysets_free;        // released slots (tensors_key_release)
int64_t tensors_key_register(at::TensorList ts, int64_t reuse) {
    std::lock_guard<std::mutex> lk(g_keysets_mu);
    std::vector<Tensor> v(ts.begin(), ts.end());
    // (a slot is re-used only while its owner still holds it: one that has been released - and may have been handed to another
    //  flow since - is not the caller's any more)
    if (reuse >= 0 && reuse < (int64_t)g_keysets.size() &&
        std::find(g_keysets_free.begin(), g_keysets_free.end(), reuse) == g_keysets_free.end()) {
        g_keysets[(size_t)reuse] = std::move(v);
        return reuse;
    }
    if (!g_keysets_free.empty()) {
        const int64_t h = g_keysets_free.back();
        g_keysets_free.pop_back();
        g_keysets[(size_t)h] = std::move(v);
        return h;
    }
    g_keysets.push_back(std::move(v));
    return (int64_t)g_keysets.size() - 1;
}
// a flow that is garbage-collected gives its slot back (the handles of dropped tensors are released; the slot is reused)
void tensors_key_release(int64_t handle) {
    std::lock_guard<std::mutex> lk(g_keysets_mu);
    if (handle < 0 || handle >= (int64_t)g_keysets.size()) return;
    if (std::find(g_keysets_free.begin(), g_keysets_free.end(), handle) != g_keysets_free.end()) return;   // already free: a second
                                                                                       // release must not hand the slot out twice
    g_keysets[(size_t)handle].clear();
    g_keysets[(size_t)handle].shrink_to_fit();
    g_keysets_free.push_back(handle);
}
std::vector<int64_t> tensors_key_of(int64_t handle) {
    std::lock_guard<std::mutex> lk(g_keysets_mu);
    TORCH_CHECK(handle >= 0 && handle < (int64_t)g_keysets.size(), "fabhip: unknown parameter-set handle ", handle);
    return tensors_key(g_keysets[(size_t)handle]);
}
int64_t flow_grad_floats(int64_t dim, int64_t n_layers, int64_t width) {
    return fabhip_flow_grad_floats((int32_t)dim, (int32_t)n_layers, (int32_t)width);
}
std::vector<int64_t> flow_grad_layout(int64_t dim, int64_t n_layers, int64_t width) {
    std::vector<int64_t> out(15);
    chk(fabhip_flow_grad_layout((int32_t)dim, (int32_t)n_layers, (int32_t)width, out.data()), "flow_grad_layout");
    return out;
}
int64_t set_fast_mode(bool on) { return fabhip_set_fast_mode(on ? 1 : 0); }
int64_t get_fast_mode() { return fabhip_get_fast_mode(); }
int64_t set_option(int64_t key, int64_t value) {
    const int prev = fabhip_set_option((int)key, (int)value);
    TORCH_CHECK(!(prev < 0 && (key < 0 || key >= FABHIP_OPT_COUNT)), "fabhip: unknown option key ", key);
    return prev;
}
int64_t get_option(int64_t key) {
    TORCH_CHECK(key >= 0 && key < FABHIP_OPT_COUNT, "fabhip: unknown option key ", key);
    return fabhip_get_option((int)key);
}
std::vector<int64_t> flow_tape_layout(int64_t dim, int64_t n_layers, int64_t width, int64_t B) {
    std::vector<int64_t> out(18);
    chk(fabhip_flow_tape_layout((int32_t)dim, (int32_t)n_layers, (int32_t)width, B, out.data()), "flow_tape_layout");
    return out;
}
int64_t spline_packed_floats(int64_t dim, int64_t n_layers, int64_t hidden) {
    return fabhip_spline_packed_floats((int32_t)dim, (int32_t)n_layers, (int32_t)hidden);
}
std::vector<double> anneal_coefs(double beta, double alpha, bool p_target) {
    const fabhip_anneal a = coefs(beta, alpha, p_target);
    return {a.c_q, a.c_p, a.g_q, a.g_p};
}

// ------------------------------------------------------------------------------------------------------------------
// RealNVP flow
// ------------------------------------------------------------------------------------------------------------------
void realnvp_pack(at::TensorList params, int64_t dim, int64_t n_layers, int64_t width, bool with_inverse,
                  Tensor packed) {
    c10::DeviceGuard g(packed.device());
    fabhip_flow_params p;
    fill_params(p, params, dim, n_layers, width);
    make_flow(packed, dim, n_layers, width);
    auto fn = with_inverse ? fabhip_flow_pack : fabhip_flow_pack_density;
    chk(fn(&p, packed.data_ptr<float>(), stream_of(packed)), "flow_pack");
}

std::tuple<Tensor, Tensor> realnvp_sample(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width,
                                          const Tensor& eps) {
    c10::DeviceGuard g(eps.device());
    const fabhip_flow f = make_flow(packed, dim, n_layers, width);
    TORCH_CHECK(eps.dim() == 2 && eps.size(1) == dim, "fabhip: eps must be [B, dim]");
    const int64_t B = eps.size(0);
    Tensor x = at::empty_like(eps), log_q = fempty({B}, eps);
    chk(fabhip_flow_sample(&f, fp(eps, "eps"), x.data_ptr<float>(), log_q.data_ptr<float>(), B, stream_of(eps)),
        "flow_sample");
    return {x, log_q};
}

std::tuple<Tensor, Tensor> realnvp_logprob_grad(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width,
                                                const Tensor& x, bool with_grad, int64_t precision) {
    c10::DeviceGuard g(x.device());
    const fabhip_flow f = make_flow(packed, dim, n_layers, width, precision);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == dim, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0);
    Tensor log_q = fempty({B}, x), grad = with_grad ? at::empty_like(x) : fempty({0}, x);
    chk(fabhip_flow_log_prob(&f, fp(x, "x"), log_q.data_ptr<float>(), with_grad ? grad.data_ptr<float>() : nullptr, B,
                             stream_of(x)),
        "flow_log_prob");
    return {log_q, grad};
}

// the differentiable training op: log q(x) with a tape; `theta` (the flat parameter image, or any tensor the
// parameters are a differentiable function of) is not read - it is the autograd handle of the parameters
std::tuple<Tensor, Tensor, Tensor> realnvp_logprob_tape(const Tensor& theta, const Tensor& x, const Tensor& packed,
                                                        at::TensorList params, int64_t dim, int64_t n_layers,
                                                        int64_t width, bool want_grad_x) {
    (void)theta; (void)params;
    c10::DeviceGuard g(x.device());
    const fabhip_flow f = make_flow(packed, dim, n_layers, width);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == dim, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0);
    const size_t nbytes = fabhip_flow_tape_bytes(f.dim, f.n_layers, f.width, B);
    Tensor log_q = fempty({B}, x), grad = want_grad_x ? at::empty_like(x) : fempty({0}, x);
    Tensor tape = fempty({(int64_t)(nbytes / 4 + 1)}, x);
    chk(fabhip_flow_log_prob_tape(&f, fp(x, "x"), log_q.data_ptr<float>(),
                                  want_grad_x ? grad.data_ptr<float>() : nullptr, B, tape.data_ptr<float>(), nbytes,
                                  stream_of(x)),
        "flow_log_prob_tape");
    return {log_q, grad, tape};
}

Tensor realnvp_param_grad(at::TensorList params, const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width,
                          const Tensor& tape, const Tensor& coef) {
    c10::DeviceGuard g(coef.device());
    fabhip_flow_params p;
    fill_params(p, params, dim, n_layers, width);
    const fabhip_flow f = make_flow(packed, dim, n_layers, width);
    const int64_t B = coef.size(0);
    const size_t nbytes = fabhip_flow_tape_bytes(f.dim, f.n_layers, f.width, B);
    TORCH_CHECK((size_t)tape.numel() * 4 >= nbytes, "fabhip: tape too small for ", B, " rows");
    int64_t lay[15];
    chk(fabhip_flow_grad_layout(f.dim, f.n_layers, f.width, lay), "flow_grad_layout");
    Tensor flat = fempty({p.an_s[0] ? lay[14] : lay[12]}, coef);     // ActNorm flows: + [an_s | an_t] per layer
    chk(fabhip_flow_param_grad(&p, &f, fp(tape, "tape"), nbytes, fp(coef, "coef"), B, flat.data_ptr<float>(),
                               stream_of(coef)),
        "flow_param_grad");
    return flat;
}

// the differentiable sampling op (reparameterised baseline losses, fab/core.py:130-152): forward = fabhip_flow_sample;
// autograd is registered from Python (_ops.py), its backward = realnvp_sample_grad_tape + realnvp_param_grad(coef = 1).
// `theta` is the autograd handle of the parameters, as in realnvp_logprob_tape.
std::tuple<Tensor, Tensor> realnvp_sample_tape(const Tensor& theta, const Tensor& eps, const Tensor& packed,
                                               at::TensorList params, int64_t dim, int64_t n_layers, int64_t width) {
    (void)theta; (void)params;
    return realnvp_sample(packed, dim, n_layers, width, eps);
}

std::tuple<Tensor, Tensor> realnvp_sample_grad_tape(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width,
                                                    const Tensor& x, const Tensor& grad_x, const Tensor& grad_log_q) {
    c10::DeviceGuard g(x.device());
    const fabhip_flow f = make_flow(packed, dim, n_layers, width);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == dim, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0);
    TORCH_CHECK(grad_x.dim() == 2 && grad_x.size(0) == B && grad_x.size(1) == dim && grad_log_q.numel() == B,
                "fabhip: grad_x must be [B, dim] and grad_log_q [B]");
    const size_t nbytes = fabhip_flow_tape_bytes(f.dim, f.n_layers, f.width, B);
    Tensor tape = fempty({(int64_t)(nbytes / 4 + 1)}, x), g_eps = at::empty_like(x);
    chk(fabhip_flow_sample_grad_tape(&f, fp(x, "x"), fp(grad_x, "grad_x"), fp(grad_log_q, "grad_log_q"),
                                     g_eps.data_ptr<float>(), B, tape.data_ptr<float>(), nbytes, stream_of(x)),
        "flow_sample_grad_tape");
    return {tape, g_eps};
}

void adam_clip_step(Tensor theta, const Tensor& grad, Tensor m, Tensor v, double lr, double beta1, double beta2,
                    double eps, Tensor step_count, double max_norm, Tensor grad_norm) {
    c10::DeviceGuard g(theta.device());
    const int64_t n = theta.numel();
    TORCH_CHECK(grad.numel() == n && m.numel() == n && v.numel() == n, "fabhip: adam image sizes differ");
    need(step_count, at::kInt, "step_count");
    const size_t nb = fabhip_adam_workspace_bytes(n);
    Tensor ws = scratch(nb, theta);
    chk(fabhip_adam_clip_step(fpm(theta, "theta"), fp(grad, "grad"), fpm(m, "m"), fpm(v, "v"), n, (float)lr,
                              (float)beta1, (float)beta2, (float)eps, step_count.data_ptr<int32_t>(), (float)max_norm,
                              fpm(grad_norm, "grad_norm"), aligned(ws), nb, stream_of(theta)),
        "adam_clip_step");
}

// One gradient step on one minibatch of the prioritised replay buffer (fab/train_with_prioritised_buffer.py:158-185) as ONE op:
// training pack, log q with the tape (the minibatch read in place from the buffer through `rows`), loss weights + buffer.adjust,
// parameter gradients, clipped Adam - nine launches, no host synchronisation, no autograd graph.  `pset`: the flow's registered
// parameter set (tensors_key_register: the order of fabhip_flow_params), whose tensors live inside `theta` (FlatAdam).
std::tuple<Tensor, Tensor, Tensor> buffer_train_step(int64_t pset, Tensor packed, int64_t dim, int64_t n_layers, int64_t width,
                                                     bool repack, const Tensor& x, const optional<Tensor>& rows,
                                                     const Tensor& log_q_old, bool log_q_old_rows, double alpha, double w_clip,
                                                     optional<Tensor> buf_log_w, optional<Tensor> buf_log_q_old, Tensor theta,
                                                     Tensor m, Tensor v, double lr, double beta1, double beta2, double eps,
                                                     Tensor step_count, double max_norm) {
    c10::DeviceGuard g(theta.device());
    fabhip_flow_params p;
    {
        std::lock_guard<std::mutex> lk(g_keysets_mu);
        TORCH_CHECK(pset >= 0 && pset < (int64_t)g_keysets.size() && !g_keysets[(size_t)pset].empty(),
                    "fabhip: unknown parameter-set handle ", pset);
        fill_params(p, g_keysets[(size_t)pset], dim, n_layers, width);
    }
    make_flow(packed, dim, n_layers, width);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == dim, "fabhip: x must be [rows, dim]");
    const int64_t B = rows.has_value() ? rows->numel() : x.size(0);
    TORCH_CHECK(B >= 1, "fabhip: empty minibatch");
    if (rows.has_value()) need(*rows, at::kLong, "rows");
    TORCH_CHECK(buf_log_w.has_value() == buf_log_q_old.has_value(), "fabhip: buf_log_w and buf_log_q_old go together");
    TORCH_CHECK(!(buf_log_w.has_value() || log_q_old_rows) || rows.has_value(), "fabhip: in-place buffer access needs `rows`");
    const int64_t n_rows = x.size(0);
    if (log_q_old_rows) need_n(log_q_old, n_rows, theta, "log_q_old (the buffer's)");
    else need_n(log_q_old, B, theta, "log_q_old");
    const int64_t n = theta.numel();
    TORCH_CHECK(m.numel() == n && v.numel() == n, "fabhip: adam image sizes differ");
    need(step_count, at::kInt, "step_count");
    int64_t lay[15];
    chk(fabhip_flow_grad_layout((int32_t)dim, (int32_t)n_layers, (int32_t)width, lay), "flow_grad_layout");
    TORCH_CHECK(n == (p.an_s[0] ? lay[14] : lay[12]), "fabhip: theta has ", n, " floats, the flow's flat layout ", lay[12]);
    // (the parameters must be views of theta: the gradient image and the Adam step use its layout)
    TORCH_CHECK(p.w1[0] == theta.data_ptr<float>() + lay[1], "fabhip: the flow's parameters do not live inside theta (FlatAdam)");
    Tensor log_q = fempty({B}, theta), adj = fempty({B}, theta), coef = fempty({B}, theta), grads = fempty({n}, theta),
           stats = fempty({8}, theta);                      // (every entry is written by the step's kernels)
    const size_t nb = fabhip_train_step_workspace_bytes((int32_t)dim, (int32_t)n_layers, (int32_t)width, B, n);
    TORCH_CHECK(nb > 0, "fabhip: flow shape not supported");
    Tensor ws = scratch(nb, theta);
    fabhip_train_step_args a;
    a.struct_bytes = sizeof(a);
    a.params = &p; a.packed = packed.data_ptr<float>(); a.repack = repack ? 1 : 0; a.log_q_old_rows = log_q_old_rows ? 1 : 0;
    a.x = fp(x, "x"); a.rows = rows.has_value() ? rows->data_ptr<int64_t>() : nullptr; a.log_q_old = fp(log_q_old, "log_q_old");
    a.B = B; a.alpha = (float)alpha; a.w_adjust_max_clip = (float)w_clip;
    a.buf_log_w = buf_log_w.has_value() ? fpmn(*buf_log_w, n_rows, theta, "buffer log_w") : nullptr;
    a.buf_log_q_old = buf_log_q_old.has_value() ? fpmn(*buf_log_q_old, n_rows, theta, "buffer log_q_old") : nullptr;
    a.log_q = log_q.data_ptr<float>(); a.log_w_adjust = adj.data_ptr<float>(); a.coef = coef.data_ptr<float>();
    a.grads = grads.data_ptr<float>(); a.stats = stats.data_ptr<float>();
    a.theta = fpm(theta, "theta"); a.m = fpm(m, "m"); a.v = fpm(v, "v"); a.n_params = n;
    a.lr = (float)lr; a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.max_grad_norm = (float)max_norm;
    a.step_count = step_count.data_ptr<int32_t>();
    a.workspace = aligned(ws); a.workspace_bytes = nb;
    chk(fabhip_buffer_train_step(&a, stream_of(theta)), "buffer_train_step");
    return {log_q, adj, stats};
}

// ------------------------------------------------------------------------------------------------------------------
// RQ-spline coupling flow.  `Tensor[] params`: per layer {meta, w0, b0, wa, ba, wb, bb, wf, bf, pfw, uw, uh, ud}
// (pfw may be an empty tensor), then {base_scale, base_circ}.
// ------------------------------------------------------------------------------------------------------------------
fabhip_spline_flow make_spline(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t hidden, int64_t precision = 0) {
    fabhip_spline_flow f;
    TORCH_CHECK(precision >= 0 && precision <= 2, "fabhip: precision must be 0 (process default), 1 (fp32) or 2 (fast)");
    f.dim = (int32_t)dim; f.n_layers = (int32_t)n_layers; f.hidden = (int32_t)hidden; f.precision = (int32_t)precision;
    f.packed = fp(packed, "packed spline image");
    const int64_t n = fabhip_spline_packed_floats(f.dim, f.n_layers, f.hidden);
    TORCH_CHECK(n > 0, "fabhip: spline flow shape not supported (dim ", dim, ", hidden ", hidden, ")");
    TORCH_CHECK(packed.numel() == n, "fabhip: packed spline image has ", packed.numel(), " floats, expected ", n);
    return f;
}

void spline_pack(at::TensorList params, int64_t dim, int64_t n_layers, int64_t hidden, Tensor packed) {
    c10::DeviceGuard g(packed.device());
    TORCH_CHECK(n_layers >= 1 && n_layers <= FABHIP_MAX_LAYERS, "fabhip: n_layers out of range");
    TORCH_CHECK((int64_t)params.size() == 13 * n_layers + 2, "fabhip: expected ", 13 * n_layers + 2, " spline tensors");
    make_spline(packed, dim, n_layers, hidden);
    fabhip_spline_params p;
    p.dim = (int32_t)dim; p.n_layers = (int32_t)n_layers; p.hidden = (int32_t)hidden;
    for (int64_t l = 0; l < n_layers; ++l) {
        const Tensor* t = &params[13 * l];
        p.meta[l] = fp(t[0], "meta"); p.w0[l] = fp(t[1], "w0"); p.b0[l] = fp(t[2], "b0"); p.wa[l] = fp(t[3], "wa");
        p.ba[l] = fp(t[4], "ba"); p.wb[l] = fp(t[5], "wb"); p.bb[l] = fp(t[6], "bb"); p.wf[l] = fp(t[7], "wf");
        p.bf[l] = fp(t[8], "bf"); p.pfw[l] = t[9].numel() ? fp(t[9], "pfw") : nullptr;
        p.uw[l] = fp(t[10], "uw"); p.uh[l] = fp(t[11], "uh"); p.ud[l] = fp(t[12], "ud");
        TORCH_CHECK(t[0].numel() == 12 * 64, "fabhip: spline meta must be [12, 64]");
    }
    p.base_scale = fp(params[13 * n_layers], "base_scale");
    p.base_circ = fp(params[13 * n_layers + 1], "base_circ");
    chk(fabhip_spline_pack(&p, packed.data_ptr<float>(), stream_of(packed)), "spline_pack");
}

std::tuple<Tensor, Tensor> spline_logprob_grad(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t hidden,
                                               const Tensor& x, bool with_grad, int64_t precision) {
    c10::DeviceGuard g(x.device());
    const fabhip_spline_flow f = make_spline(packed, dim, n_layers, hidden, precision);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == dim, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0);
    Tensor log_q = fempty({B}, x), grad = with_grad ? at::empty_like(x) : fempty({0}, x);
    const size_t nb = fabhip_spline_workspace_bytes(f.dim, f.n_layers, f.hidden, B, with_grad ? 1 : 0);
    Tensor ws = scratch(nb, x);
    chk(fabhip_spline_log_prob(&f, fp(x, "x"), log_q.data_ptr<float>(), with_grad ? grad.data_ptr<float>() : nullptr, B,
                               aligned(ws), nb, stream_of(x)),
        "spline_log_prob");
    return {log_q, grad};
}

std::vector<int64_t> spline_tape_layout(int64_t dim, int64_t n_layers, int64_t hidden, int64_t B) {
    std::vector<int64_t> out(16);
    chk(fabhip_spline_tape_layout((int32_t)dim, (int32_t)n_layers, (int32_t)hidden, B, out.data()), "spline_tape_layout");
    return out;
}

// log_q, d log_q / dx and the training tape (see fabhip_spline_tape_layout); the parameter gradients are GEMMs over
// tape slices done in fab_torch_amd/spline_flow.py
std::tuple<Tensor, Tensor, Tensor> spline_logprob_tape(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t hidden,
                                                       const Tensor& x) {
    c10::DeviceGuard g(x.device());
    const fabhip_spline_flow f = make_spline(packed, dim, n_layers, hidden);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == dim, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0);
    int64_t lay[16];
    chk(fabhip_spline_tape_layout(f.dim, f.n_layers, f.hidden, B, lay), "spline_tape_layout");
    Tensor log_q = fempty({B}, x), grad = at::empty_like(x), tape = fempty({lay[0]}, x);
    const size_t nb = fabhip_spline_workspace_bytes(f.dim, f.n_layers, f.hidden, B, 1);
    Tensor ws = scratch(nb, x);
    chk(fabhip_spline_log_prob_tape(&f, fp(x, "x"), log_q.data_ptr<float>(), grad.data_ptr<float>(), B,
                                    tape.data_ptr<float>(), lay[0], aligned(ws), nb, stream_of(x)),
        "spline_log_prob_tape");
    return {log_q, grad, tape};
}

// the backward of flow.sample_and_log_prob (fabhip_spline_sample_vjp_tape): the two tapes for the tape GEMMs (coefficients gl and
// 1) and v at the base side
std::tuple<Tensor, Tensor, Tensor> spline_sample_vjp_tape(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t hidden,
                                                          const Tensor& u, const Tensor& eps, const optional<Tensor>& gx,
                                                          const optional<Tensor>& gl) {
    c10::DeviceGuard g(u.device());
    const fabhip_spline_flow f = make_spline(packed, dim, n_layers, hidden);
    TORCH_CHECK(u.dim() == 2 && u.size(1) == dim && eps.sizes() == u.sizes(), "fabhip: u, eps must be [B, dim]");
    TORCH_CHECK(!gx.has_value() || gx->sizes() == u.sizes(), "fabhip: gx must be [B, dim]");
    const int64_t B = u.size(0);
    TORCH_CHECK(!gl.has_value() || (gl->dim() == 1 && gl->size(0) == B), "fabhip: gl must be [B]");
    int64_t lay[16];
    chk(fabhip_spline_tape_layout(f.dim, f.n_layers, f.hidden, B, lay), "spline_tape_layout");
    Tensor tape1 = fempty({lay[0]}, u), tape2 = fempty({lay[0]}, u), v_x = at::empty_like(u), v_base = at::empty_like(u);
    const size_t nb = fabhip_spline_workspace_bytes(f.dim, f.n_layers, f.hidden, B, 1);
    Tensor ws = scratch(nb, u);
    chk(fabhip_spline_sample_vjp_tape(&f, fp(u, "u"), fp(eps, "eps"), fp_opt(gx, "gx"), fp_opt(gl, "gl"), v_x.data_ptr<float>(),
                                      v_base.data_ptr<float>(), B, tape1.data_ptr<float>(), tape2.data_ptr<float>(), lay[0],
                                      aligned(ws), nb, stream_of(u)),
        "spline_sample_vjp_tape");
    return {tape1, tape2, v_base};
}

std::tuple<Tensor, Tensor> spline_sample(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t hidden,
                                         const Tensor& u, const Tensor& eps) {
    c10::DeviceGuard g(u.device());
    const fabhip_spline_flow f = make_spline(packed, dim, n_layers, hidden);
    TORCH_CHECK(u.dim() == 2 && u.size(1) == dim && eps.sizes() == u.sizes(), "fabhip: u, eps must be [B, dim]");
    const int64_t B = u.size(0);
    Tensor x = at::empty_like(u), log_q = fempty({B}, u);
    const size_t nb = fabhip_spline_workspace_bytes(f.dim, f.n_layers, f.hidden, B, 0);
    Tensor ws = scratch(nb, u);
    chk(fabhip_spline_sample(&f, fp(u, "u"), fp(eps, "eps"), x.data_ptr<float>(), log_q.data_ptr<float>(), B, aligned(ws),
                             nb, stream_of(u)),
        "spline_sample");
    return {x, log_q};
}

// ------------------------------------------------------------------------------------------------------------------
// targets, points
// ------------------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> target_logp_grad(int64_t kind, at::ArrayRef<double> prm, const optional<Tensor>& locs,
                                            const optional<Tensor>& scales, const Tensor& x, bool with_grad) {
    c10::DeviceGuard g(x.device());
    TORCH_CHECK(x.dim() == 2, "fabhip: x must be [B, dim]");
    const fabhip_target t = make_target(kind, prm, locs, scales, x.size(1));
    const int64_t B = x.size(0);
    Tensor lp = fempty({B}, x), grad = with_grad ? at::empty_like(x) : fempty({0}, x);
    chk(fabhip_target_log_prob(&t, fp(x, "x"), lp.data_ptr<float>(), with_grad ? grad.data_ptr<float>() : nullptr, B,
                               stream_of(x)),
        "target_log_prob");
    return {lp, grad};
}

std::tuple<Tensor, Tensor> manywell_logp_grad(const Tensor& x, double a, double b, double c, double log_norm) {
    const double prm[4] = {a, b, c, log_norm};
    return target_logp_grad(FABHIP_TARGET_MANYWELL, prm, c10::nullopt, c10::nullopt, x, true);
}

std::tuple<Tensor, Tensor> gmm_logp_grad(const Tensor& x, const Tensor& locs, const Tensor& scales) {
    const double prm[4] = {0, 0, 0, 0};
    return target_logp_grad(FABHIP_TARGET_GMM, prm, locs, scales, x, true);
}

std::tuple<Tensor, Tensor, Tensor, Tensor> create_point(const Tensor& packed, int64_t dim, int64_t n_layers,
                                                        int64_t width, int64_t kind, at::ArrayRef<double> prm,
                                                        const optional<Tensor>& locs, const optional<Tensor>& scales,
                                                        const Tensor& x, bool with_grad, int64_t precision) {
    c10::DeviceGuard g(x.device());
    const fabhip_flow f = make_flow(packed, dim, n_layers, width, precision);
    const fabhip_target t = make_target(kind, prm, locs, scales, dim);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == dim, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0);
    Tensor lq = fempty({B}, x), lp = fempty({B}, x);
    Tensor gq = with_grad ? at::empty_like(x) : fempty({0}, x), gp = with_grad ? at::empty_like(x) : fempty({0}, x);
    fabhip_point p{const_cast<float*>(fp(x, "x")), lq.data_ptr<float>(), lp.data_ptr<float>(),
                   with_grad ? gq.data_ptr<float>() : nullptr, with_grad ? gp.data_ptr<float>() : nullptr};
    chk(fabhip_create_point(&f, &t, &p, with_grad ? 1 : 0, B, stream_of(x)), "create_point");
    return {lq, lp, gq, gp};
}

// ------------------------------------------------------------------------------------------------------------------
// transitions (in place on the Point tensors, the step-size state and log_w)
// ------------------------------------------------------------------------------------------------------------------
void hmc_transition(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width, int64_t kind,
                    at::ArrayRef<double> prm, const optional<Tensor>& locs, const optional<Tensor>& scales, Tensor x,
                    Tensor log_q, Tensor log_p, Tensor grad_log_q, Tensor grad_log_p, optional<Tensor> log_w,
                    double beta, double beta_next, double alpha, bool p_target, const Tensor& noise_p,
                    const Tensor& noise_e, Tensor epsilons_row, Tensor common_epsilon, const Tensor& mass, int64_t L,
                    double max_grad, double target_p_accept, bool tune, optional<Tensor> p_accept,
                    optional<Tensor> avg_distance, int64_t precision) {
    c10::DeviceGuard g(x.device());
    fabhip_hmc_args a;
    a.flow = make_flow(packed, dim, n_layers, width, precision);
    a.target = make_target(kind, prm, locs, scales, dim);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == dim, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0), n_outer = epsilons_row.numel();
    TORCH_CHECK(noise_p.numel() == n_outer * B * dim && noise_e.numel() == n_outer * B, "fabhip: HMC noise shapes");
    TORCH_CHECK(n_outer >= 1, "fabhip: epsilons row must hold one step size per outer loop");
    a.point = fabhip_point{fpm(x, "x"), fpmn(log_q, B, x, "log_q"), fpmn(log_p, B, x, "log_p"),
                           fpmn(grad_log_q, B * dim, x, "grad_log_q"), fpmn(grad_log_p, B * dim, x, "grad_log_p")};
    a.B = B; a.n_valid = nullptr;
    a.cur = coefs(beta, alpha, p_target); a.next = coefs(beta_next, alpha, p_target);
    a.log_w = fpmn_opt(log_w, B, x, "log_w");
    a.noise_p = fpn(noise_p, n_outer * B * dim, x, "noise_p"); a.noise_e = fpn(noise_e, n_outer * B, x, "noise_e");
    a.epsilons = fpmn(epsilons_row, n_outer, x, "epsilons");
    a.common_epsilon = fpmn(common_epsilon, 1, x, "common_epsilon", true);
    a.mass = fpn(mass, dim, x, "mass");
    a.n_outer = (int32_t)n_outer; a.L = (int32_t)L; a.max_grad = (float)max_grad;
    a.target_p_accept = (float)target_p_accept; a.tune = tune ? 1 : 0;
    a.p_accept = fpmn_opt(p_accept, n_outer, x, "p_accept", true);
    a.avg_distance = fpmn_opt(avg_distance, 1, x, "avg_distance", true);
    const size_t nb = fabhip_hmc_workspace_bytes(B, (int32_t)dim, (int32_t)n_outer);
    Tensor ws = scratch(nb, x);
    a.workspace = aligned(ws); a.workspace_bytes = nb;
    a.partials = nullptr;
    chk(fabhip_hmc_transition(&a, stream_of(x)), "hmc_transition");
}

void metropolis_transition(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width, int64_t kind,
                           at::ArrayRef<double> prm, const optional<Tensor>& locs, const optional<Tensor>& scales,
                           Tensor x, Tensor log_q, Tensor log_p, optional<Tensor> log_w, double beta, double beta_next,
                           double alpha, bool p_target, const Tensor& noise_x, const Tensor& noise_u,
                           Tensor noise_scalings_row, double target_p_accept, bool tune) {
    c10::DeviceGuard g(x.device());
    fabhip_metropolis_args a;
    a.flow = make_flow(packed, dim, n_layers, width);
    a.target = make_target(kind, prm, locs, scales, dim);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == dim, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0), n_updates = noise_scalings_row.numel();
    TORCH_CHECK(noise_x.numel() == n_updates * B * dim && noise_u.numel() == n_updates * B,
                "fabhip: Metropolis noise shapes");
    TORCH_CHECK(n_updates >= 1, "fabhip: noise_scalings row must hold one scale per update");
    a.point = fabhip_point{fpm(x, "x"), fpmn(log_q, B, x, "log_q"), fpmn(log_p, B, x, "log_p"), nullptr, nullptr};
    a.B = B; a.n_valid = nullptr;
    a.cur = coefs(beta, alpha, p_target); a.next = coefs(beta_next, alpha, p_target);
    a.log_w = fpmn_opt(log_w, B, x, "log_w");
    a.noise_x = fpn(noise_x, n_updates * B * dim, x, "noise_x"); a.noise_u = fpn(noise_u, n_updates * B, x, "noise_u");
    a.noise_scalings = fpmn(noise_scalings_row, n_updates, x, "noise_scalings");
    a.n_updates = (int32_t)n_updates; a.target_p_accept = (float)target_p_accept; a.tune = tune ? 1 : 0;
    const size_t nb = fabhip_metropolis_workspace_bytes(B, (int32_t)dim, (int32_t)n_updates);
    Tensor ws = scratch(nb, x);
    a.workspace = aligned(ws); a.workspace_bytes = nb;
    chk(fabhip_metropolis_transition(&a, stream_of(x)), "metropolis_transition");
}

// ------------------------------------------------------------------------------------------------------------------
// generic plug-in path (any Distribution / LogProbFunc: densities + gradients evaluated by the caller)
// ------------------------------------------------------------------------------------------------------------------
Tensor generic_workspace(const Tensor& like, int64_t B, int64_t dim) {
    return at::empty({(int64_t)(fabhip_generic_workspace_bytes(B, (int32_t)dim) / 4 + 64)}, like.options().dtype(at::kFloat));
}
size_t ws_bytes(const Tensor& ws) { return (size_t)ws.numel() * 4; }

void hmc_generic_begin(const Tensor& start_x, const Tensor& start_gq, const Tensor& start_gp, const Tensor& cur_lq,
                       const Tensor& cur_lp, double beta, double alpha, bool p_target, const Tensor& noise_p,
                       const Tensor& mass, double max_grad, Tensor ws) {
    c10::DeviceGuard g(start_x.device());
    TORCH_CHECK(start_x.dim() == 2, "fabhip: x must be [B, dim]");
    const int64_t B = start_x.size(0), D = start_x.size(1);
    const Tensor& r = start_x;
    fabhip_point st{const_cast<float*>(fp(start_x, "x")), nullptr, nullptr,
                    const_cast<float*>(fpn(start_gq, B * D, r, "grad_log_q")),
                    const_cast<float*>(fpn(start_gp, B * D, r, "grad_log_p"))};
    fabhip_point cu{nullptr, const_cast<float*>(fpn(cur_lq, B, r, "log_q")), const_cast<float*>(fpn(cur_lp, B, r, "log_p")),
                    nullptr, nullptr};
    need_n(ws, 0, r, "workspace", true);
    chk(fabhip_hmc_generic_begin(&st, &cu, B, (int32_t)D, coefs(beta, alpha, p_target), fpn(noise_p, B * D, r, "noise_p"),
                                 fpn(mass, D, r, "mass"), (float)max_grad, fpm(ws, "workspace"), ws_bytes(ws), stream_of(start_x)),
        "hmc_generic_begin");
}

Tensor hmc_generic_leap_pre(int64_t B, int64_t dim, const Tensor& eps, const Tensor& ceps, const Tensor& mass,
                            Tensor ws) {
    c10::DeviceGuard g(ws.device());
    Tensor x = fempty({B, dim}, ws);
    need_n(eps, 1, ws, "epsilon", true); need_n(ceps, 1, ws, "common_epsilon", true); need_n(mass, dim, ws, "mass");
    chk(fabhip_hmc_generic_leap_pre(B, (int32_t)dim, fp(eps, "epsilon"), fp(ceps, "common_epsilon"), fp(mass, "mass"),
                                    x.data_ptr<float>(), fpm(ws, "workspace"), ws_bytes(ws), stream_of(ws)),
        "hmc_generic_leap_pre");
    return x;
}

void hmc_generic_leap_post(const Tensor& gq, const Tensor& gp, double beta, double alpha, bool p_target, double max_grad,
                           const Tensor& eps, const Tensor& ceps, Tensor ws) {
    c10::DeviceGuard g(ws.device());
    TORCH_CHECK(gq.dim() == 2, "fabhip: grad_log_q must be [B, dim]");
    const int64_t B = gq.size(0), D = gq.size(1);
    need_n(gq, B * D, ws, "grad_log_q"); need_n(gp, B * D, ws, "grad_log_p");
    need_n(eps, 1, ws, "epsilon", true); need_n(ceps, 1, ws, "common_epsilon", true);
    chk(fabhip_hmc_generic_leap_post(B, (int32_t)D, fp(gq, "grad_log_q"), fp(gp, "grad_log_p"),
                                     coefs(beta, alpha, p_target), (float)max_grad, fp(eps, "epsilon"),
                                     fp(ceps, "common_epsilon"), fpm(ws, "workspace"), ws_bytes(ws), stream_of(ws)),
        "hmc_generic_leap_post");
}

void hmc_generic_accept(const Tensor& prop_lq, const Tensor& prop_lp, const Tensor& prop_gq, const Tensor& prop_gp,
                        Tensor x, Tensor log_q, Tensor log_p, Tensor grad_log_q, Tensor grad_log_p,
                        optional<Tensor> log_w, double beta, double beta_next, double alpha, bool p_target,
                        const Tensor& noise_e, const Tensor& mass, Tensor eps, Tensor ceps, double target_p_accept,
                        bool tune, optional<Tensor> p_accept, optional<Tensor> avg_distance, Tensor ws) {
    c10::DeviceGuard g(x.device());
    TORCH_CHECK(x.dim() == 2, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0), D = x.size(1);
    fabhip_point pr{nullptr, const_cast<float*>(fpn(prop_lq, B, x, "log_q")), const_cast<float*>(fpn(prop_lp, B, x, "log_p")),
                    const_cast<float*>(fpn(prop_gq, B * D, x, "grad_log_q")),
                    const_cast<float*>(fpn(prop_gp, B * D, x, "grad_log_p"))};
    fabhip_point cu{fpm(x, "x"), fpmn(log_q, B, x, "log_q"), fpmn(log_p, B, x, "log_p"),
                    fpmn(grad_log_q, B * D, x, "grad_log_q"), fpmn(grad_log_p, B * D, x, "grad_log_p")};
    need_n(ws, 0, x, "workspace", true);
    chk(fabhip_hmc_generic_accept(&pr, &cu, B, (int32_t)D, coefs(beta, alpha, p_target), coefs(beta_next, alpha, p_target),
                                  fpmn_opt(log_w, B, x, "log_w"), fpn(noise_e, B, x, "noise_e"), fpn(mass, D, x, "mass"),
                                  fpmn(eps, 1, x, "epsilon", true), fpmn(ceps, 1, x, "common_epsilon", true),
                                  (float)target_p_accept, tune ? 1 : 0, fpmn_opt(p_accept, 1, x, "p_accept", true),
                                  fpmn_opt(avg_distance, 1, x, "avg_distance", true),
                                  fpm(ws, "workspace"), ws_bytes(ws), stream_of(x)),
        "hmc_generic_accept");
}

// One HMC transition (hmc.py:129-160, all n_outer x L leapfrogs) for the SPLINE flow + a native target, enqueued from
// C++: per leapfrog the generic element-wise kernels, the one-launch spline density + gradient kernel and the target
// kernel - what transition_operators.py::_transition_generic does step by step from Python, without the ~40 op calls
// per transition in between (they cost as much as the kernels at batch sizes <= 1024).
void spline_hmc_transition(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t hidden, int64_t kind,
                           at::ArrayRef<double> prm, const optional<Tensor>& locs, const optional<Tensor>& scales, Tensor x,
                           Tensor log_q, Tensor log_p, Tensor grad_log_q, Tensor grad_log_p, optional<Tensor> log_w,
                           double beta, double beta_next, double alpha, bool p_target, const Tensor& noise_p,
                           const Tensor& noise_e, Tensor eps_row, Tensor ceps, const Tensor& mass, int64_t n_outer,
                           int64_t n_leap, double max_grad, double target_p_accept, bool tune, optional<Tensor> p_accept,
                           optional<Tensor> avg_distance, int64_t precision) {
    c10::DeviceGuard g(x.device());
    const fabhip_spline_flow f = make_spline(packed, dim, n_layers, hidden, precision);
    const fabhip_target tg = make_target(kind, prm, locs, scales, dim);
    TORCH_CHECK(x.dim() == 2, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0), D = x.size(1);
    TORCH_CHECK(D == dim && n_outer >= 1 && n_leap >= 1, "fabhip: spline_hmc_transition shapes");
    need_n(log_q, B, x, "log_q"); need_n(log_p, B, x, "log_p");
    need_n(grad_log_q, B * D, x, "grad_log_q"); need_n(grad_log_p, B * D, x, "grad_log_p");
    if (log_w.has_value()) need_n(*log_w, B, x, "log_w");
    need_n(noise_p, n_outer * B * D, x, "noise_p"); need_n(noise_e, n_outer * B, x, "noise_e");
    need_n(eps_row, n_outer, x, "epsilons"); need_n(ceps, 1, x, "common_epsilon", true); need_n(mass, D, x, "mass");
    if (p_accept.has_value()) need_n(*p_accept, n_outer, x, "p_accept", true);
    if (avg_distance.has_value()) need_n(*avg_distance, 1, x, "avg_distance", true);
    if (B == 0) return;
    fabhip_spline_hmc_args a;
    a.flow = f; a.target = tg;
    a.point = fabhip_point{fpm(x, "x"), fpm(log_q, "log_q"), fpm(log_p, "log_p"), fpm(grad_log_q, "grad_log_q"),
                           fpm(grad_log_p, "grad_log_p")};
    a.B = B; a.n_valid = nullptr;
    a.cur = coefs(beta, alpha, p_target); a.next = coefs(beta_next, alpha, p_target);
    a.log_w = fpm_opt(log_w, "log_w");
    a.noise_p = fp(noise_p, "noise_p"); a.noise_e = fp(noise_e, "noise_e");
    a.epsilons = fpm(eps_row, "epsilons"); a.common_epsilon = fpm(ceps, "common_epsilon"); a.mass = fp(mass, "mass");
    a.n_outer = (int32_t)n_outer; a.L = (int32_t)n_leap; a.max_grad = (float)max_grad;
    a.target_p_accept = (float)target_p_accept; a.tune = tune ? 1 : 0;
    a.p_accept = fpm_opt(p_accept, "p_accept"); a.avg_distance = fpm_opt(avg_distance, "avg_distance");
    const size_t nb = fabhip_spline_hmc_workspace_bytes(f.dim, f.n_layers, f.hidden, B);
    Tensor ws = scratch(nb, x);
    a.workspace = aligned(ws); a.workspace_bytes = nb;
    chk(fabhip_spline_hmc_transition(&a, stream_of(x)), "spline_hmc_transition");
}

// The spline family's ais_run (fabhip_spline_ais_run): same outputs as ais_run.
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> spline_ais_run(
    const Tensor& packed, int64_t dim, int64_t n_layers, int64_t hidden, int64_t kind, at::ArrayRef<double> prm,
    const optional<Tensor>& locs, const optional<Tensor>& scales, at::ArrayRef<double> betas, double alpha, bool p_target,
    const Tensor& u0, const Tensor& eps0, const Tensor& noise_p, const Tensor& noise_e, Tensor epsilons, Tensor common_epsilon,
    const Tensor& mass, int64_t n_outer, int64_t L, double max_grad, double target_p_accept, bool tune,
    optional<Tensor> p_accept_first, optional<Tensor> p_accept_last, optional<Tensor> avg_distance_first,
    optional<Tensor> avg_distance_last, bool want_base, int64_t precision) {
    c10::DeviceGuard g(eps0.device());
    fabhip_spline_ais_args a;
    a.flow = make_spline(packed, dim, n_layers, hidden, precision);
    a.target = make_target(kind, prm, locs, scales, dim);
    TORCH_CHECK(eps0.dim() == 2 && eps0.size(1) == dim, "fabhip: eps0 must be [B, dim]");
    const int64_t B = eps0.size(0), M = (int64_t)betas.size() - 2;
    TORCH_CHECK(M >= 1 && n_outer >= 1 && L >= 1, "fabhip: spline_ais_run needs M, n_outer, L >= 1");
    a.B = B; a.M = (int32_t)M;
    std::vector<double> bt(betas.begin(), betas.end());
    a.betas = bt.data(); a.alpha = alpha; a.p_target = p_target ? 1 : 0;
    a.u0 = fpn(u0, B * dim, eps0, "u0"); a.eps0 = fp(eps0, "eps0");
    a.noise_p = fpn(noise_p, M * n_outer * B * dim, eps0, "noise_p"); a.noise_e = fpn(noise_e, M * n_outer * B, eps0, "noise_e");
    a.epsilons = fpmn(epsilons, M * n_outer, eps0, "epsilons");
    a.common_epsilon = fpmn(common_epsilon, 1, eps0, "common_epsilon", true);
    a.mass = fpn(mass, dim, eps0, "mass");
    a.n_outer = (int32_t)n_outer; a.L = (int32_t)L; a.max_grad = (float)max_grad;
    a.target_p_accept = (float)target_p_accept; a.tune = tune ? 1 : 0;
    Tensor x = fempty({B, dim}, eps0), lq = fempty({B}, eps0), lp = fempty({B}, eps0), log_w = fempty({B}, eps0);
    Tensor gq = fempty({B, dim}, eps0), gp = fempty({B, dim}, eps0);
    Tensor counts_stats = at::empty({18}, eps0.options());      // one device->host copy (_ops.read_counts_and_stats); every word is
                                                                // written by the call (the counts and statistics by the phase tails, stats[6..15] = 0)
    Tensor stats = counts_stats.narrow(0, 0, 16), n_valid = counts_stats.narrow(0, 16, 2).view(at::kInt);
    Tensor base_x = want_base ? fempty({B, dim}, eps0) : fempty({0}, eps0);
    Tensor base_lw = want_base ? fempty({B}, eps0) : fempty({0}, eps0);
    a.point = fabhip_point{x.data_ptr<float>(), lq.data_ptr<float>(), lp.data_ptr<float>(), gq.data_ptr<float>(),
                           gp.data_ptr<float>()};
    a.log_w = log_w.data_ptr<float>(); a.n_valid = n_valid.data_ptr<int32_t>(); a.stats = stats.data_ptr<float>();
    a.p_accept_first = fpmn_opt(p_accept_first, n_outer, eps0, "p_accept_first", true);
    a.p_accept_last = fpmn_opt(p_accept_last, n_outer, eps0, "p_accept_last", true);
    a.avg_distance_first = fpmn_opt(avg_distance_first, 1, eps0, "avg_distance_first", true);
    a.avg_distance_last = fpmn_opt(avg_distance_last, 1, eps0, "avg_distance_last", true);
    a.base_x = want_base ? base_x.data_ptr<float>() : nullptr;
    a.base_log_w = want_base ? base_lw.data_ptr<float>() : nullptr;
    const size_t nb = fabhip_spline_ais_workspace_bytes(a.flow.dim, a.flow.n_layers, a.flow.hidden, B);
    Tensor ws = scratch(nb, eps0);
    a.workspace = aligned(ws); a.workspace_bytes = nb;
    chk(fabhip_spline_ais_run(&a, stream_of(eps0)), "spline_ais_run");
    return {x, lq, lp, gq, gp, log_w, n_valid, stats, base_x, base_lw};
}

Tensor anneal_log_prob(const Tensor& log_q, const Tensor& log_p, double beta, double alpha, bool p_target) {
    c10::DeviceGuard g(log_q.device());
    Tensor out = at::empty_like(log_q);
    need_n(log_p, log_q.numel(), log_q, "log_p");
    chk(fabhip_anneal_log_prob(fp(log_q, "log_q"), fp(log_p, "log_p"), log_q.numel(), coefs(beta, alpha, p_target),
                               out.data_ptr<float>(), stream_of(log_q)),
        "anneal_log_prob");
    return out;
}

void log_w_update(const Tensor& log_q, const Tensor& log_p, double beta, double beta_next, double alpha, bool p_target,
                  Tensor log_w) {
    c10::DeviceGuard g(log_q.device());
    need_n(log_p, log_q.numel(), log_q, "log_p"); need_n(log_w, log_q.numel(), log_q, "log_w");
    chk(fabhip_log_w_update(fp(log_q, "log_q"), fp(log_p, "log_p"), log_q.numel(), coefs(beta, alpha, p_target),
                            coefs(beta_next, alpha, p_target), fpm(log_w, "log_w"), stream_of(log_q)),
        "log_w_update");
}

Tensor metropolis_generic_propose(const Tensor& x, const Tensor& noise_x, const Tensor& scale) {
    c10::DeviceGuard g(x.device());
    Tensor xn = at::empty_like(x);
    TORCH_CHECK(x.dim() == 2, "fabhip: x must be [B, dim]");
    need_n(noise_x, x.numel(), x, "noise_x"); need_n(scale, 1, x, "noise_scaling", true);
    chk(fabhip_metropolis_generic_propose(fp(x, "x"), fp(noise_x, "noise_x"), fp(scale, "noise_scaling"), x.size(0),
                                          (int32_t)x.size(1), xn.data_ptr<float>(), stream_of(x)),
        "metropolis_generic_propose");
    return xn;
}

void metropolis_generic_accept(const Tensor& x_new, const Tensor& new_lq, const Tensor& new_lp, Tensor x, Tensor log_q,
                               Tensor log_p, const Tensor& prev_log_prob, const Tensor& noise_u, double beta, double alpha,
                               bool p_target, Tensor scale, double target_p_accept, bool tune) {
    c10::DeviceGuard g(x.device());
    TORCH_CHECK(x.dim() == 2, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0), D = x.size(1);
    fabhip_point cu{fpm(x, "x"), fpmn(log_q, B, x, "log_q"), fpmn(log_p, B, x, "log_p"), nullptr, nullptr};
    need_n(x_new, B * D, x, "x_new"); need_n(new_lq, B, x, "new log_q"); need_n(new_lp, B, x, "new log_p");
    need_n(prev_log_prob, B, x, "prev_log_prob"); need_n(noise_u, B, x, "noise_u"); need_n(scale, 1, x, "noise_scaling", true);
    Tensor ws = generic_workspace(x, B, D);
    chk(fabhip_metropolis_generic_accept(fp(x_new, "x_new"), fp(new_lq, "log_q"), fp(new_lp, "log_p"), &cu,
                                         fp(prev_log_prob, "prev_log_prob"), fp(noise_u, "noise_u"), B, (int32_t)D,
                                         coefs(beta, alpha, p_target), fpm(scale, "noise_scaling"),
                                         (float)target_p_accept, tune ? 1 : 0, fpm(ws, "workspace"), ws_bytes(ws),
                                         stream_of(x)),
        "metropolis_generic_accept");
}

// ------------------------------------------------------------------------------------------------------------------
// the whole AIS call (ais.py:53-105): returns
//   (x, log_q, log_p, grad_log_q, grad_log_p, log_w, n_valid int32[2], stats float[16], base_x, base_log_w)
// grad_* are empty for Metropolis, base_* are empty unless want_base.
// ------------------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> ais_run(
    const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width, int64_t kind, at::ArrayRef<double> prm,
    const optional<Tensor>& locs, const optional<Tensor>& scales, at::ArrayRef<double> betas, double alpha,
    bool p_target, int64_t transition, const Tensor& eps0, const optional<Tensor>& noise_a_in,
    const optional<Tensor>& noise_b_in, Tensor step_state, optional<Tensor> common_epsilon, const optional<Tensor>& mass,
    int64_t n_inner, int64_t L, double max_grad, double target_p_accept, bool tune, optional<Tensor> p_accept_first,
    optional<Tensor> p_accept_last, optional<Tensor> avg_distance_first, optional<Tensor> avg_distance_last, bool want_base,
    int64_t precision) {
    c10::DeviceGuard g(eps0.device());
    // noise_a / noise_b absent: drawn HERE, from the default generator in the order the Python side draws them (normal
    // [M, n_inner, B, dim], then exponential(1) / uniform [M, n_inner, B]) - but AFTER the chain initialisation is enqueued, so the
    // device works while the host launches the draws (FABHIP_AIS_CONTINUE)
    const bool draw_inside = !noise_a_in.has_value() || !noise_b_in.has_value();
    TORCH_CHECK(noise_a_in.has_value() == noise_b_in.has_value(), "fabhip: pass both noise tensors or neither");
    Tensor noise_a = draw_inside ? Tensor() : *noise_a_in, noise_b = draw_inside ? Tensor() : *noise_b_in;
    fabhip_ais_args a;
    a.flow = make_flow(packed, dim, n_layers, width, precision);
    a.target = make_target(kind, prm, locs, scales, dim);
    TORCH_CHECK(eps0.dim() == 2 && eps0.size(1) == dim, "fabhip: eps0 must be [B, dim]");
    const int64_t B = eps0.size(0), M = (int64_t)betas.size() - 2;
    TORCH_CHECK(M >= 1, "fabhip: betas must hold M + 2 values");
    TORCH_CHECK(draw_inside || (noise_a.numel() == M * n_inner * B * dim && noise_b.numel() == M * n_inner * B),
                "fabhip: AIS noise shapes");
    TORCH_CHECK(step_state.numel() == M * n_inner, "fabhip: step-size state must be [M, n_inner]");
    const bool hmc = transition == FABHIP_TRANSITION_HMC;
    a.B = B; a.M = (int32_t)M;
    std::vector<double> bt(betas.begin(), betas.end());
    a.betas = bt.data();
    a.alpha = alpha; a.p_target = p_target ? 1 : 0; a.transition = (int32_t)transition;
    TORCH_CHECK(n_inner >= 1, "fabhip: n_inner (HMC outer loops / Metropolis updates per transition) must be >= 1");
    TORCH_CHECK(hmc || transition == FABHIP_TRANSITION_METROPOLIS, "fabhip: unknown transition kind ", transition);
    TORCH_CHECK(!hmc || (common_epsilon.has_value() && mass.has_value()),
                "fabhip: an HMC AIS run needs common_epsilon and the mass vector");
    a.eps0 = fp(eps0, "eps0");
    a.noise_a = a.noise_b = nullptr;
    if (!draw_inside) {
        a.noise_a = fpn(noise_a, M * n_inner * B * dim, eps0, "noise_a"); a.noise_b = fpn(noise_b, M * n_inner * B, eps0, "noise_b");
    }
    a.step_state = fpmn(step_state, M * n_inner, eps0, "step_state");
    a.common_epsilon = fpmn_opt(common_epsilon, 1, eps0, "common_epsilon", true);
    a.mass = mass.has_value() ? fpn(*mass, dim, eps0, "mass") : nullptr;
    a.n_inner = (int32_t)n_inner; a.L = (int32_t)L; a.max_grad = (float)max_grad;
    a.target_p_accept = (float)target_p_accept; a.tune = tune ? 1 : 0;
    // the outputs are views of ONE allocation (each padded to 256 bytes): one trip through the caching allocator instead of six
    // on the host path the GPU waits for
    auto pad64 = [](int64_t n) { return (n + 63) / 64 * 64; };
    Tensor pool = fempty({(hmc ? 3 : 1) * pad64(B * dim) + 3 * pad64(B)}, eps0);
    int64_t pool_off = 0;
    auto take = [&](int64_t rows, int64_t cols) {
        Tensor t = cols > 0 ? pool.narrow(0, pool_off, rows * cols).view({rows, cols}) : pool.narrow(0, pool_off, rows);
        pool_off += pad64(rows * (cols > 0 ? cols : 1));
        return t;
    };
    Tensor x = take(B, dim), lq = take(B, 0), lp = take(B, 0), log_w = take(B, 0);
    Tensor gq = hmc ? take(B, dim) : fempty({0}, eps0), gp = hmc ? take(B, dim) : fempty({0}, eps0);
    Tensor counts_stats = at::empty({18}, eps0.options());      // one device->host copy (_ops.read_counts_and_stats); every word is
                                                                // written by the call (the counts and statistics by the phase tails, stats[6..15] = 0)
    Tensor stats = counts_stats.narrow(0, 0, 16), n_valid = counts_stats.narrow(0, 16, 2).view(at::kInt);
    Tensor base_x = want_base ? fempty({B, dim}, eps0) : fempty({0}, eps0);
    Tensor base_lw = want_base ? fempty({B}, eps0) : fempty({0}, eps0);
    a.point = fabhip_point{x.data_ptr<float>(), lq.data_ptr<float>(), lp.data_ptr<float>(),
                           hmc ? gq.data_ptr<float>() : nullptr, hmc ? gp.data_ptr<float>() : nullptr};
    a.log_w = log_w.data_ptr<float>(); a.n_valid = n_valid.data_ptr<int32_t>(); a.stats = stats.data_ptr<float>();
    a.p_accept_first = fpmn_opt(p_accept_first, n_inner, eps0, "p_accept_first", true);
    a.p_accept_last = fpmn_opt(p_accept_last, n_inner, eps0, "p_accept_last", true);
    a.avg_distance_first = fpmn_opt(avg_distance_first, 1, eps0, "avg_distance_first", true);
    a.avg_distance_last = fpmn_opt(avg_distance_last, 1, eps0, "avg_distance_last", true);
    a.base_x = want_base ? base_x.data_ptr<float>() : nullptr;
    a.base_log_w = want_base ? base_lw.data_ptr<float>() : nullptr;
    const size_t nb = fabhip_ais_workspace_bytes(B, (int32_t)dim, (int32_t)n_inner);
    Tensor ws = scratch(nb, eps0);
    a.workspace = aligned(ws); a.workspace_bytes = nb;
    if (!draw_inside) {
        chk(fabhip_ais_run(&a, stream_of(eps0)), "ais_run");
    } else {
        chk(fabhip_ais_phase(&a, FABHIP_AIS_INIT, 1, 0, nullptr, stream_of(eps0)), "ais_run (chain initialisation)");
        noise_a = at::randn({M, n_inner, B, dim}, eps0.options());
        noise_b = hmc ? at::empty({M, n_inner, B}, eps0.options()).exponential_(1.0) : at::rand({M, n_inner, B}, eps0.options());
        a.noise_a = noise_a.data_ptr<float>(); a.noise_b = noise_b.data_ptr<float>();
        chk(fabhip_ais_phase(&a, FABHIP_AIS_CONTINUE | FABHIP_AIS_FINISH, 1, (int32_t)M, nullptr, stream_of(eps0)),
            "ais_run (transitions)");
    }
    return {x, lq, lp, gq, gp, log_w, n_valid, stats, base_x, base_lw};
}

// The same call in pieces (fabhip_ais_phase): the state tensors are the caller's, in/out across the phases of one AIS
// run.  Used when chains are sharded over ranks and the step sizes adapt on the acceptance of ALL chains: one transition
// per call with `partials`, the caller all-gathers the slabs and calls hmc_adapt_gathered (fab_torch_amd/parallel.py).
// `group_name` == nullptr: the phases / transitions the caller names, once.  Otherwise the WHOLE tuned call of one shard:
// INIT, then per transition {transition with the adaptation deferred, slab all-gather over the named c10d process group,
// step-size rule on the gathered slabs}, FINISH - see ais_sharded_tuned below.  Returns the number of collectives issued.
int64_t ais_phase_core(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width, int64_t kind, at::ArrayRef<double> prm,
               const optional<Tensor>& locs, const optional<Tensor>& scales, at::ArrayRef<double> betas, double alpha,
               bool p_target, int64_t transition, int64_t phases, int64_t j_begin, int64_t j_end,
               const optional<Tensor>& eps0, const Tensor& noise_a, const Tensor& noise_b, Tensor step_state,
               optional<Tensor> common_epsilon, const optional<Tensor>& mass, int64_t n_inner, int64_t L, double max_grad,
               double target_p_accept, bool tune, Tensor x, Tensor log_q, Tensor log_p, optional<Tensor> grad_log_q,
               optional<Tensor> grad_log_p, Tensor log_w, Tensor n_valid, Tensor stats, optional<Tensor> partials,
               optional<Tensor> p_accept_first, optional<Tensor> p_accept_last, optional<Tensor> avg_distance_first,
               optional<Tensor> avg_distance_last, optional<Tensor> base_x, optional<Tensor> base_log_w, int64_t precision,
               const std::string* group_name) {
    c10::DeviceGuard g(x.device());
    fabhip_ais_args a;
    a.flow = make_flow(packed, dim, n_layers, width, precision);
    a.target = make_target(kind, prm, locs, scales, dim);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == dim, "fabhip: x must be [B, dim]");
    const int64_t B = x.size(0), M = (int64_t)betas.size() - 2;
    TORCH_CHECK(M >= 1, "fabhip: betas must hold M + 2 values");
    TORCH_CHECK(n_inner >= 1, "fabhip: n_inner must be >= 1");
    const bool hmc = transition == FABHIP_TRANSITION_HMC;
    TORCH_CHECK(hmc || transition == FABHIP_TRANSITION_METROPOLIS, "fabhip: unknown transition kind ", transition);
    TORCH_CHECK(!hmc || (common_epsilon.has_value() && mass.has_value() && grad_log_q.has_value() && grad_log_p.has_value()),
                "fabhip: an HMC AIS run needs common_epsilon, the mass vector and the gradient fields of the Point");
    a.B = B; a.M = (int32_t)M;
    std::vector<double> bt(betas.begin(), betas.end());
    a.betas = bt.data();
    a.alpha = alpha; a.p_target = p_target ? 1 : 0; a.transition = (int32_t)transition;
    a.eps0 = eps0.has_value() ? fpn(*eps0, B * dim, x, "eps0") : nullptr;
    a.noise_a = fpn(noise_a, M * n_inner * B * dim, x, "noise_a"); a.noise_b = fpn(noise_b, M * n_inner * B, x, "noise_b");
    a.step_state = fpmn(step_state, M * n_inner, x, "step_state");
    a.common_epsilon = fpmn_opt(common_epsilon, 1, x, "common_epsilon", true);
    a.mass = mass.has_value() ? fpn(*mass, dim, x, "mass") : nullptr;
    a.n_inner = (int32_t)n_inner; a.L = (int32_t)L; a.max_grad = (float)max_grad;
    a.target_p_accept = (float)target_p_accept; a.tune = tune ? 1 : 0;
    a.point = fabhip_point{fpm(x, "x"), fpmn(log_q, B, x, "log_q"), fpmn(log_p, B, x, "log_p"),
                           hmc ? fpmn(*grad_log_q, B * dim, x, "grad_log_q") : nullptr,
                           hmc ? fpmn(*grad_log_p, B * dim, x, "grad_log_p") : nullptr};
    a.log_w = fpmn(log_w, B, x, "log_w");
    need(n_valid, at::kInt, "n_valid"); need_n(n_valid, 2, x, "n_valid");
    a.n_valid = n_valid.data_ptr<int32_t>();
    a.stats = fpmn(stats, 16, x, "stats");
    a.p_accept_first = fpmn_opt(p_accept_first, n_inner, x, "p_accept_first", true);
    a.p_accept_last = fpmn_opt(p_accept_last, n_inner, x, "p_accept_last", true);
    a.avg_distance_first = fpmn_opt(avg_distance_first, 1, x, "avg_distance_first", true);
    a.avg_distance_last = fpmn_opt(avg_distance_last, 1, x, "avg_distance_last", true);
    a.base_x = fpmn_opt(base_x, B * dim, x, "base_x");
    a.base_log_w = fpmn_opt(base_log_w, B, x, "base_log_w");
    const size_t nb = fabhip_ais_workspace_bytes(B, (int32_t)dim, (int32_t)n_inner);
    Tensor ws = scratch(nb, x);
    a.workspace = aligned(ws); a.workspace_bytes = nb;
    float* slab = fpmn_opt(partials, hmc ? fabhip_hmc_partials_floats(B)
                                         : fabhip_metropolis_partials_floats(B, (int32_t)M, (int32_t)n_inner), x, "partials");
    if (group_name == nullptr) {
        chk(fabhip_ais_phase(&a, (int32_t)phases, (int32_t)j_begin, (int32_t)j_end, slab, stream_of(x)), "ais_phase");
        return 0;
    }
    // ---- one shard's tuned call: the loop fab_torch_amd/parallel.py stepped from Python in rounds 2 - 4 ----
    TORCH_CHECK(hmc && n_inner == 1 && slab != nullptr,
                "fabhip: ais_sharded_tuned is HMC with n_outer == 1 and needs the acceptance slab (hmc_partials_floats(B) floats)");
    auto pg = c10d::resolve_process_group(*group_name);
    const int64_t world = pg->getSize(), nslab = fabhip_hmc_partials_floats(B);
    // RCCL ("nccl") takes the device slab as it is: the collective is ordered behind the transition on the current stream
    // and Work::wait() orders the stream behind the collective - the host never blocks.  A host-only backend (gloo: the
    // CPU-side tests of the N > 1 path, two ranks on a one-GPU box) is fed through the host.
    bool device_collective = false;
    try {
        device_collective = pg->getBackend(c10::DeviceType::CUDA)->getBackendName() == "nccl";
    } catch (const std::exception&) {                     // no backend registered for device tensors: through the host
    }
    Tensor gathered = fempty({world * nslab}, x), host_in, host_out;
    if (!device_collective) {
        host_in = at::empty({nslab}, at::TensorOptions().dtype(at::kFloat).device(at::kCPU));
        host_out = at::empty({world * nslab}, host_in.options());
    }
    const fabhip_stream_t st = stream_of(x);
    chk(fabhip_ais_phase(&a, FABHIP_AIS_INIT, 1, 0, nullptr, st), "ais_sharded_tuned (chain initialisation)");
    int64_t n_collectives = 0;
    for (int64_t j = 1; j <= M; ++j) {
        chk(fabhip_ais_phase(&a, 0, (int32_t)j, (int32_t)j, slab, st), "ais_sharded_tuned (transition)");
        if (world > 1 || device_collective) {          // (a one-rank RCCL group still runs the collective: the path a node takes)
            if (device_collective) {
                pg->_allgather_base(gathered, *partials)->wait();
            } else {
                host_in.copy_(*partials);
                pg->_allgather_base(host_out, host_in)->wait();
                gathered.copy_(host_out);
            }
            ++n_collectives;
        } else {
            gathered.copy_(*partials);
        }
        float* pa = j == 1 ? a.p_accept_first : (j == M ? a.p_accept_last : nullptr);        // hmc.py:173-183 (store_info)
        float* ad = j == 1 ? a.avg_distance_first : (j == M ? a.avg_distance_last : nullptr);
        chk(fabhip_hmc_adapt_gathered(gathered.data_ptr<float>(), (int32_t)world, B, a.step_state + (j - 1), a.common_epsilon,
                                      a.target_p_accept, 1, pa, ad, st),
            "ais_sharded_tuned (step-size rule on the gathered slabs)");
    }
    chk(fabhip_ais_phase(&a, FABHIP_AIS_FINISH, 1, 0, nullptr, st), "ais_sharded_tuned (chain end)");
    return n_collectives;
}

void ais_phase(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width, int64_t kind, at::ArrayRef<double> prm,
               const optional<Tensor>& locs, const optional<Tensor>& scales, at::ArrayRef<double> betas, double alpha,
               bool p_target, int64_t transition, int64_t phases, int64_t j_begin, int64_t j_end,
               const optional<Tensor>& eps0, const Tensor& noise_a, const Tensor& noise_b, Tensor step_state,
               optional<Tensor> common_epsilon, const optional<Tensor>& mass, int64_t n_inner, int64_t L, double max_grad,
               double target_p_accept, bool tune, Tensor x, Tensor log_q, Tensor log_p, optional<Tensor> grad_log_q,
               optional<Tensor> grad_log_p, Tensor log_w, Tensor n_valid, Tensor stats, optional<Tensor> partials,
               optional<Tensor> p_accept_first, optional<Tensor> p_accept_last, optional<Tensor> avg_distance_first,
               optional<Tensor> avg_distance_last, optional<Tensor> base_x, optional<Tensor> base_log_w, int64_t precision) {
    ais_phase_core(packed, dim, n_layers, width, kind, prm, locs, scales, betas, alpha, p_target, transition, phases, j_begin,
                   j_end, eps0, noise_a, noise_b, step_state, common_epsilon, mass, n_inner, L, max_grad, target_p_accept, tune,
                   x, log_q, log_p, grad_log_q, grad_log_p, log_w, n_valid, stats, partials, p_accept_first, p_accept_last,
                   avg_distance_first, avg_distance_last, base_x, base_log_w, precision, nullptr);
}

// One shard's WHOLE tuned AIS call in one op: chain initialisation, M x {HMC transition publishing its
// acceptance slab, ONE all-gather of the slabs through the c10d process group registered as `group_name` (RCCL over xGMI
// on a GPU node), the step-size rule of hmc.py:162-170 on the joined slabs in rank order (fabhip_hmc_adapt_gathered: every
// rank applies it to the same numbers)}, "chain end" filter + this shard's ESS / log Z.  Same kernels, same order and the
// same values as fab_torch_amd/parallel.py's Python-stepped loop (kept as the reference implementation the tests compare
// with); what goes is 2 M + 1 dispatcher round trips and M Python-level collectives per call.  Returns the collectives issued.
int64_t ais_sharded_tuned(const Tensor& packed, int64_t dim, int64_t n_layers, int64_t width, int64_t kind,
                          at::ArrayRef<double> prm, const optional<Tensor>& locs, const optional<Tensor>& scales,
                          at::ArrayRef<double> betas, double alpha, bool p_target, int64_t transition,
                          const optional<Tensor>& eps0, const Tensor& noise_a, const Tensor& noise_b, Tensor step_state,
                          optional<Tensor> common_epsilon, const optional<Tensor>& mass, int64_t L, double max_grad,
                          double target_p_accept, Tensor x, Tensor log_q, Tensor log_p, optional<Tensor> grad_log_q,
                          optional<Tensor> grad_log_p, Tensor log_w, Tensor n_valid, Tensor stats, Tensor partials,
                          optional<Tensor> p_accept_first, optional<Tensor> p_accept_last,
                          optional<Tensor> avg_distance_first, optional<Tensor> avg_distance_last, int64_t precision,
                          std::string group_name) {
    return ais_phase_core(packed, dim, n_layers, width, kind, prm, locs, scales, betas, alpha, p_target, transition, 0, 1, 0, eps0,
                          noise_a, noise_b, step_state, common_epsilon, mass, 1, L, max_grad, target_p_accept, true, x, log_q,
                          log_p, grad_log_q, grad_log_p, log_w, n_valid, stats, partials, p_accept_first, p_accept_last,
                          avg_distance_first, avg_distance_last, c10::nullopt, c10::nullopt, precision, &group_name);
}

// Linear backward over a tape (fabhip_tape_gemm): Y / X are views INTO `tape` given as float offsets of layer 0
std::tuple<Tensor, Tensor> tape_gemm(const Tensor& tape, int64_t layer_stride, int64_t L, int64_t off_y, int64_t ldy, int64_t P,
                                     int64_t off_x, int64_t ldx, int64_t Q, const Tensor& coef, bool want_colsum) {
    c10::DeviceGuard g(tape.device());
    const int64_t B = coef.numel();
    TORCH_CHECK(L >= 1 && P >= 1 && Q >= 1 && ldy >= P && ldx >= Q && off_y >= 0 && off_x >= 0, "fabhip: tape_gemm shapes");
    TORCH_CHECK((L - 1) * layer_stride + off_y + B * ldy <= tape.numel() && (L - 1) * layer_stride + off_x + B * ldx <= tape.numel(),
                "fabhip: tape_gemm operands leave the tape");
    need_n(coef, B, tape, "coef");
    Tensor C = fempty({L, P, Q}, tape), S = want_colsum ? fempty({L, P}, tape) : fempty({0}, tape);
    const float* t = fp(tape, "tape");
    chk(fabhip_tape_gemm(t + off_y, layer_stride, (int32_t)ldy, (int32_t)P, t + off_x, layer_stride, (int32_t)ldx, (int32_t)Q,
                         fp(coef, "coef"), B, (int32_t)L, C.data_ptr<float>(), want_colsum ? S.data_ptr<float>() : nullptr,
                         stream_of(tape)),
        "tape_gemm");
    return {C, S};
}

int64_t hmc_partials_floats(int64_t B) { return fabhip_hmc_partials_floats(B); }
int64_t metropolis_partials_floats(int64_t B, int64_t M, int64_t n_updates) {
    return fabhip_metropolis_partials_floats(B, (int32_t)M, (int32_t)n_updates);
}

// metropolis.py:68-73 on the acceptance of ALL chains of a sharded batch, for the M transitions of one AIS call at once
void metropolis_adapt_gathered(const Tensor& gathered, int64_t n_ranks, int64_t B_rank, Tensor noise_scalings,
                               double target_p_accept, bool tune) {
    c10::DeviceGuard g(gathered.device());
    TORCH_CHECK(n_ranks >= 1 && B_rank >= 1 && noise_scalings.dim() == 2, "fabhip: metropolis_adapt_gathered needs n_ranks, "
                "B_rank >= 1 and noise_scalings [M, n_updates]");
    const int64_t M = noise_scalings.size(0), nu = noise_scalings.size(1);
    need_n(gathered, n_ranks * fabhip_metropolis_partials_floats(B_rank, (int32_t)M, (int32_t)nu), gathered, "gathered partials");
    chk(fabhip_metropolis_adapt_gathered(fp(gathered, "gathered partials"), (int32_t)n_ranks, B_rank, (int32_t)M, (int32_t)nu,
                                         fpmn(noise_scalings, M * nu, gathered, "noise_scalings"), (float)target_p_accept,
                                         tune ? 1 : 0, stream_of(gathered)),
        "metropolis_adapt_gathered");
}

void hmc_adapt_gathered(const Tensor& gathered, int64_t n_ranks, int64_t B_rank, Tensor epsilon, Tensor common_epsilon,
                        double target_p_accept, bool tune, optional<Tensor> p_accept, optional<Tensor> avg_distance) {
    c10::DeviceGuard g(gathered.device());
    TORCH_CHECK(n_ranks >= 1 && B_rank >= 1, "fabhip: hmc_adapt_gathered needs n_ranks, B_rank >= 1");
    need_n(gathered, n_ranks * fabhip_hmc_partials_floats(B_rank), gathered, "gathered partials");
    chk(fabhip_hmc_adapt_gathered(fp(gathered, "gathered partials"), (int32_t)n_ranks, B_rank,
                                  fpmn(epsilon, 1, gathered, "epsilon"), fpmn(common_epsilon, 1, gathered, "common_epsilon", true),
                                  (float)target_p_accept, tune ? 1 : 0, fpmn_opt(p_accept, 1, gathered, "p_accept", true),
                                  fpmn_opt(avg_distance, 1, gathered, "avg_distance", true), stream_of(gathered)),
        "hmc_adapt_gathered");
}

// ------------------------------------------------------------------------------------------------------------------
// ESS / log Z, resampling, top-k
// ------------------------------------------------------------------------------------------------------------------
Tensor ess_logz(const Tensor& log_w, const optional<Tensor>& n_ptr, double n_norm) {
    c10::DeviceGuard g(log_w.device());
    const int64_t n = log_w.numel();
    Tensor out = fempty({3}, log_w);
    const size_t nb = fabhip_ess_workspace_bytes(n);
    Tensor ws = scratch(nb, log_w);
    const int32_t* np = nullptr;
    if (n_ptr.has_value()) { need(*n_ptr, at::kInt, "n_ptr"); np = n_ptr->data_ptr<int32_t>(); }
    chk(fabhip_ess_logz(fp(log_w, "log_w"), n, np, n_norm, out.data_ptr<float>(), aligned(ws), nb, stream_of(log_w)),
        "ess_logz");
    return out;
}

Tensor resample_multinomial(const Tensor& log_w, const Tensor& u) {
    c10::DeviceGuard g(log_w.device());
    need(u, at::kDouble, "u");
    const int64_t n = log_w.numel(), ns = u.numel();
    Tensor idx = at::empty({ns}, log_w.options().dtype(at::kLong));
    const size_t nb = fabhip_resample_workspace_bytes(n);
    Tensor ws = scratch(nb, log_w);
    chk(fabhip_resample_multinomial(fp(log_w, "log_w"), n, u.data_ptr<double>(), ns, idx.data_ptr<int64_t>(),
                                    aligned(ws), nb, stream_of(log_w)),
        "resample_multinomial");
    return idx;
}

// scan pass alone (after a first full call): returns the workspace so that a caller can time repeated scans
Tensor fixed_cdf(const Tensor& log_w, optional<Tensor> workspace) {
    c10::DeviceGuard g(log_w.device());
    const int64_t n = log_w.numel();
    const size_t nb = fabhip_resample_workspace_bytes(n);
    Tensor ws = workspace.has_value() ? *workspace : scratch(nb, log_w);
    chk(fabhip_fixed_cdf(fp(log_w, "log_w"), n, workspace.has_value() ? 1 : 0, nullptr, aligned(ws), nb, stream_of(log_w)),
        "fixed_cdf");
    return ws;
}

Tensor resample_systematic(const Tensor& log_w, double u0, int64_t n_samples) {
    c10::DeviceGuard g(log_w.device());
    const int64_t n = log_w.numel();
    Tensor idx = at::empty({n_samples}, log_w.options().dtype(at::kLong));
    const size_t nb = fabhip_resample_workspace_bytes(n);
    Tensor ws = scratch(nb, log_w);
    chk(fabhip_resample_systematic(fp(log_w, "log_w"), n, u0, n_samples, idx.data_ptr<int64_t>(), aligned(ws), nb,
                                   stream_of(log_w)),
        "resample_systematic");
    return idx;
}

Tensor multinomial_torch(const Tensor& probs, const Tensor& u) {
    c10::DeviceGuard g(probs.device());
    need(u, at::kDouble, "u");
    const int64_t n = probs.numel(), ns = u.numel();
    Tensor idx = at::empty({ns}, probs.options().dtype(at::kLong));
    const size_t nb = fabhip_multinomial_torch_workspace_bytes(n);
    Tensor ws = scratch(nb, probs);
    chk(fabhip_multinomial_torch(fp(probs, "probs"), n, u.data_ptr<double>(), ns, idx.data_ptr<int64_t>(), aligned(ws),
                                 nb, stream_of(probs)),
        "multinomial_torch");
    return idx;
}

Tensor gather_rows(const Tensor& src, const Tensor& idx) {
    c10::DeviceGuard g(src.device());
    need(idx, at::kLong, "idx");
    TORCH_CHECK(src.dim() >= 1, "fabhip: gather_rows needs at least one dimension");
    const int64_t rows = src.size(0), row_len = rows ? src.numel() / rows : 0, n_out = idx.numel();
    std::vector<int64_t> shape(src.sizes().begin(), src.sizes().end());
    shape[0] = n_out;
    Tensor out = at::empty(shape, src.options());
    if (n_out == 0 || row_len == 0) return out;
    chk(fabhip_gather_rows(fp(src, "src"), idx.data_ptr<int64_t>(), out.data_ptr<float>(), n_out, row_len,
                           stream_of(src)),
        "gather_rows");
    return out;
}

// PrioritisedReplayBuffer.add / the row selection of .sample as one op each (the draws come from torch's device generator)
void buffer_add(const Tensor& x, const Tensor& log_w, const Tensor& log_q_old, int64_t start, Tensor buf_x, Tensor buf_log_w,
                Tensor buf_log_q_old) {
    c10::DeviceGuard g(buf_x.device());
    TORCH_CHECK(x.dim() == 2 && buf_x.dim() == 2 && x.size(1) == buf_x.size(1), "fabhip: x must be [n, dim] like the buffer");
    const int64_t n = x.size(0), dim = x.size(1), L = buf_x.size(0);
    need_n(log_w, n, buf_x, "log_w"); need_n(log_q_old, n, buf_x, "log_q_old");
    need_n(buf_log_w, L, buf_x, "buffer log_w"); need_n(buf_log_q_old, L, buf_x, "buffer log_q_old");
    need_n(x, n * dim, buf_x, "x");
    chk(fabhip_buffer_add(fp(x, "x"), fp(log_w, "log_w"), fp(log_q_old, "log_q_old"), n, (int32_t)dim, start, L, fpm(buf_x, "buffer x"),
                          fpm(buf_log_w, "buffer log_w"), fpm(buf_log_q_old, "buffer log_q_old"), stream_of(buf_x)),
        "buffer_add");
}

Tensor buffer_sample_indices(const Tensor& log_w, int64_t k) {
    c10::DeviceGuard g(log_w.device());
    const int64_t n = log_w.numel();
    Tensor u = at::rand({n}, log_w.options().dtype(at::kFloat)), r = at::rand({4}, log_w.options().dtype(at::kFloat));
    Tensor idx = at::empty({k}, log_w.options().dtype(at::kLong));
    const size_t nb = fabhip_buffer_sample_workspace_bytes(n, k);
    TORCH_CHECK(nb > 0, "fabhip: buffer_sample_indices needs 1 <= k <= n");
    Tensor ws = scratch(nb, log_w);
    chk(fabhip_buffer_sample(fp(log_w, "log_w"), u.data_ptr<float>(), r.data_ptr<float>(), n, k, idx.data_ptr<int64_t>(), aligned(ws),
                             nb, stream_of(log_w)),
        "buffer_sample");
    return idx;
}

Tensor topk(const Tensor& keys, int64_t k, bool sorted) {
    c10::DeviceGuard g(keys.device());
    const int64_t n = keys.numel();
    Tensor idx = at::empty({k}, keys.options().dtype(at::kLong));
    const size_t nb = fabhip_topk_workspace_bytes(n, k);
    Tensor ws = scratch(nb, keys);
    chk(fabhip_topk(fp(keys, "keys"), n, k, sorted ? 1 : 0, idx.data_ptr<int64_t>(), nullptr, aligned(ws), nb,
                    stream_of(keys)),
        "topk");
    return idx;
}

}  // namespace

#define TGT "int target_kind, float[] target_params, Tensor? locs, Tensor? scales"
#define FLW "Tensor packed, int dim, int n_layers, int width"

TORCH_LIBRARY(fabhip, m) {
    m.def("abi_version() -> int", abi_version);
    m.def("flow_packed_floats(int dim, int n_layers, int width) -> int", flow_packed_floats);
    m.def("tensors_key(Tensor[] ts) -> int[]", tensors_key);
    m.def("tensors_key_register(Tensor[] ts, int reuse) -> int", tensors_key_register);
    m.def("tensors_key_of(int handle) -> int[]", tensors_key_of);
    m.def("tensors_key_release(int handle) -> ()", tensors_key_release);
    m.def("flow_grad_floats(int dim, int n_layers, int width) -> int", flow_grad_floats);
    m.def("flow_grad_layout(int dim, int n_layers, int width) -> int[]", flow_grad_layout);
    m.def("flow_tape_layout(int dim, int n_layers, int width, int B) -> int[]", flow_tape_layout);
    m.def("set_fast_mode(bool on) -> int", set_fast_mode);
    m.def("get_fast_mode() -> int", get_fast_mode);
    m.def("set_option(int key, int value) -> int", set_option);
    m.def("get_option(int key) -> int", get_option);
    m.def("anneal_coefs(float beta, float alpha, bool p_target) -> float[]", anneal_coefs);

    m.def("realnvp_pack(Tensor[] params, int dim, int n_layers, int width, bool with_inverse, Tensor(a!) packed) -> ()");
    m.def("realnvp_sample(" FLW ", Tensor eps) -> (Tensor, Tensor)");
    m.def("realnvp_logprob_grad(" FLW ", Tensor x, bool with_grad, int precision=0) -> (Tensor, Tensor)");
    m.def("realnvp_logprob_tape(Tensor theta, Tensor x, Tensor packed, Tensor[] params, int dim, int n_layers, "
          "int width, bool want_grad_x) -> (Tensor, Tensor, Tensor)");
    m.def("realnvp_param_grad(Tensor[] params, " FLW ", Tensor tape, Tensor coef) -> Tensor");
    m.def("realnvp_sample_tape(Tensor theta, Tensor eps, Tensor packed, Tensor[] params, int dim, int n_layers, "
          "int width) -> (Tensor, Tensor)");
    m.def("realnvp_sample_grad_tape(" FLW ", Tensor x, Tensor grad_x, Tensor grad_log_q) -> (Tensor, Tensor)");
    m.def("buffer_train_step(int pset, Tensor(a!) packed, int dim, int n_layers, int width, bool repack, Tensor x, Tensor? rows, "
          "Tensor log_q_old, bool log_q_old_rows, float alpha, float w_clip, Tensor(b!)? buf_log_w, Tensor(c!)? buf_log_q_old, "
          "Tensor(d!) theta, Tensor(e!) m, Tensor(f!) v, float lr, float beta1, float beta2, float eps, Tensor(g!) step_count, "
          "float max_norm) -> (Tensor, Tensor, Tensor)");
    m.def("adam_clip_step(Tensor(a!) theta, Tensor grad, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, "
          "float eps, Tensor(d!) step_count, float max_norm, Tensor(e!) grad_norm) -> ()");

    m.def("spline_packed_floats(int dim, int n_layers, int hidden) -> int", spline_packed_floats);
    m.def("spline_pack(Tensor[] params, int dim, int n_layers, int hidden, Tensor(a!) packed) -> ()");
    m.def("spline_logprob_grad(Tensor packed, int dim, int n_layers, int hidden, Tensor x, bool with_grad, int precision=0) -> (Tensor, Tensor)");
    m.def("spline_tape_layout(int dim, int n_layers, int hidden, int B) -> int[]", spline_tape_layout);
    m.def("spline_logprob_tape(Tensor packed, int dim, int n_layers, int hidden, Tensor x) -> (Tensor, Tensor, Tensor)");
    m.def("spline_sample(Tensor packed, int dim, int n_layers, int hidden, Tensor u, Tensor eps) -> (Tensor, Tensor)");
    m.def("spline_sample_vjp_tape(Tensor packed, int dim, int n_layers, int hidden, Tensor u, Tensor eps, Tensor? gx, Tensor? gl) -> (Tensor, Tensor, Tensor)");
    m.def("target_logp_grad(" TGT ", Tensor x, bool with_grad) -> (Tensor, Tensor)");
    m.def("manywell_logp_grad(Tensor x, float a, float b, float c, float log_norm) -> (Tensor, Tensor)");
    m.def("gmm_logp_grad(Tensor x, Tensor locs, Tensor scales) -> (Tensor, Tensor)");
    m.def("create_point(" FLW ", " TGT ", Tensor x, bool with_grad, int precision=0) -> (Tensor, Tensor, Tensor, Tensor)");

    m.def("hmc_transition(" FLW ", " TGT ", Tensor(a!) x, Tensor(b!) log_q, Tensor(c!) log_p, Tensor(d!) grad_log_q, "
          "Tensor(e!) grad_log_p, Tensor(f!)? log_w, float beta, float beta_next, float alpha, bool p_target, "
          "Tensor noise_p, Tensor noise_e, Tensor(g!) epsilons_row, Tensor(h!) common_epsilon, Tensor mass, int L, "
          "float max_grad, float target_p_accept, bool tune, Tensor(i!)? p_accept, Tensor(j!)? avg_distance, int precision=0) -> ()");
    m.def("metropolis_transition(" FLW ", " TGT ", Tensor(a!) x, Tensor(b!) log_q, Tensor(c!) log_p, Tensor(d!)? log_w, "
          "float beta, float beta_next, float alpha, bool p_target, Tensor noise_x, Tensor noise_u, "
          "Tensor(e!) noise_scalings_row, float target_p_accept, bool tune) -> ()");
    m.def("ais_run(" FLW ", " TGT ", float[] betas, float alpha, bool p_target, int transition, Tensor eps0, "
          "Tensor? noise_a, Tensor? noise_b, Tensor(a!) step_state, Tensor(b!)? common_epsilon, Tensor? mass, int n_inner, "
          "int L, float max_grad, float target_p_accept, bool tune, Tensor(c!)? p_accept_first, Tensor(d!)? p_accept_last, "
          "Tensor(e!)? avg_distance_first, Tensor(f!)? avg_distance_last, bool want_base, int precision=0) -> "
          "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");

    m.def("ais_phase(" FLW ", " TGT ", float[] betas, float alpha, bool p_target, int transition, int phases, int j_begin, "
          "int j_end, Tensor? eps0, Tensor noise_a, Tensor noise_b, Tensor(a!) step_state, Tensor(b!)? common_epsilon, "
          "Tensor? mass, int n_inner, int L, float max_grad, float target_p_accept, bool tune, Tensor(c!) x, "
          "Tensor(d!) log_q, Tensor(e!) log_p, Tensor(f!)? grad_log_q, Tensor(g!)? grad_log_p, Tensor(h!) log_w, "
          "Tensor(i!) n_valid, Tensor(j!) stats, Tensor(k!)? partials, Tensor(l!)? p_accept_first, "
          "Tensor(m!)? p_accept_last, Tensor(n!)? avg_distance_first, Tensor(o!)? avg_distance_last, Tensor(p!)? base_x, "
          "Tensor(q!)? base_log_w, int precision=0) -> ()");
    m.def("ais_sharded_tuned(" FLW ", " TGT ", float[] betas, float alpha, bool p_target, int transition, Tensor? eps0, "
          "Tensor noise_a, Tensor noise_b, Tensor(a!) step_state, Tensor(b!)? common_epsilon, Tensor? mass, int L, "
          "float max_grad, float target_p_accept, Tensor(c!) x, Tensor(d!) log_q, Tensor(e!) log_p, Tensor(f!)? grad_log_q, "
          "Tensor(g!)? grad_log_p, Tensor(h!) log_w, Tensor(i!) n_valid, Tensor(j!) stats, Tensor(k!) partials, "
          "Tensor(l!)? p_accept_first, Tensor(m!)? p_accept_last, Tensor(n!)? avg_distance_first, "
          "Tensor(o!)? avg_distance_last, int precision, str group_name) -> int");
    m.def("hmc_partials_floats(int B) -> int", hmc_partials_floats);
    m.def("metropolis_partials_floats(int B, int M, int n_updates) -> int", metropolis_partials_floats);
    m.def("metropolis_adapt_gathered(Tensor gathered, int n_ranks, int B_rank, Tensor(a!) noise_scalings, "
          "float target_p_accept, bool tune) -> ()");
    m.def("tape_gemm(Tensor tape, int layer_stride, int L, int off_y, int ldy, int P, int off_x, int ldx, int Q, Tensor coef, "
          "bool want_colsum) -> (Tensor, Tensor)");
    m.def("hmc_adapt_gathered(Tensor gathered, int n_ranks, int B_rank, Tensor(a!) epsilon, Tensor(b!) common_epsilon, "
          "float target_p_accept, bool tune, Tensor(c!)? p_accept, Tensor(d!)? avg_distance) -> ()");
    m.def("generic_workspace(Tensor like, int B, int dim) -> Tensor");
    m.def("hmc_generic_begin(Tensor start_x, Tensor start_gq, Tensor start_gp, Tensor cur_lq, Tensor cur_lp, float beta, "
          "float alpha, bool p_target, Tensor noise_p, Tensor mass, float max_grad, Tensor(a!) ws) -> ()");
    m.def("hmc_generic_leap_pre(int B, int dim, Tensor eps, Tensor ceps, Tensor mass, Tensor(a!) ws) -> Tensor");
    m.def("hmc_generic_leap_post(Tensor gq, Tensor gp, float beta, float alpha, bool p_target, float max_grad, Tensor eps, "
          "Tensor ceps, Tensor(a!) ws) -> ()");
    m.def("hmc_generic_accept(Tensor prop_lq, Tensor prop_lp, Tensor prop_gq, Tensor prop_gp, Tensor(a!) x, "
          "Tensor(b!) log_q, Tensor(c!) log_p, Tensor(d!) grad_log_q, Tensor(e!) grad_log_p, Tensor(f!)? log_w, float beta, "
          "float beta_next, float alpha, bool p_target, Tensor noise_e, Tensor mass, Tensor(g!) eps, Tensor(h!) ceps, "
          "float target_p_accept, bool tune, Tensor(i!)? p_accept, Tensor(j!)? avg_distance, Tensor(k!) ws) -> ()");
    m.def("spline_hmc_transition(Tensor packed, int dim, int n_layers, int hidden, int target_kind, float[] target_params, "
          "Tensor? locs, Tensor? scales, Tensor(a!) x, Tensor(b!) log_q, Tensor(c!) log_p, Tensor(d!) grad_log_q, "
          "Tensor(e!) grad_log_p, Tensor(f!)? log_w, float beta, float beta_next, float alpha, bool p_target, Tensor noise_p, "
          "Tensor noise_e, Tensor(g!) eps_row, Tensor(h!) ceps, Tensor mass, int n_outer, int n_leap, float max_grad, "
          "float target_p_accept, bool tune, Tensor(i!)? p_accept, Tensor(j!)? avg_distance, int precision=0) -> ()");
    m.def("spline_ais_run(Tensor packed, int dim, int n_layers, int hidden, " TGT ", float[] betas, float alpha, bool p_target, "
          "Tensor u0, Tensor eps0, Tensor noise_p, Tensor noise_e, Tensor(a!) epsilons, Tensor(b!) common_epsilon, Tensor mass, "
          "int n_outer, int L, float max_grad, float target_p_accept, bool tune, Tensor(c!)? p_accept_first, "
          "Tensor(d!)? p_accept_last, Tensor(e!)? avg_distance_first, Tensor(f!)? avg_distance_last, bool want_base, int precision=0) -> "
          "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("anneal_log_prob(Tensor log_q, Tensor log_p, float beta, float alpha, bool p_target) -> Tensor");
    m.def("log_w_update(Tensor log_q, Tensor log_p, float beta, float beta_next, float alpha, bool p_target, "
          "Tensor(a!) log_w) -> ()");
    m.def("metropolis_generic_propose(Tensor x, Tensor noise_x, Tensor scale) -> Tensor");
    m.def("metropolis_generic_accept(Tensor x_new, Tensor new_lq, Tensor new_lp, Tensor(a!) x, Tensor(b!) log_q, "
          "Tensor(c!) log_p, Tensor prev_log_prob, Tensor noise_u, float beta, float alpha, bool p_target, "
          "Tensor(d!) scale, float target_p_accept, bool tune) -> ()");

    m.def("ess_logz(Tensor log_w, Tensor? n_ptr, float n_norm) -> Tensor");
    m.def("fixed_cdf(Tensor log_w, Tensor(a!)? workspace) -> Tensor(a!)");
    m.def("resample_multinomial(Tensor log_w, Tensor u) -> Tensor");
    m.def("resample_systematic(Tensor log_w, float u0, int n_samples) -> Tensor");
    m.def("multinomial_torch(Tensor probs, Tensor u) -> Tensor");
    m.def("gather_rows(Tensor src, Tensor idx) -> Tensor");
    m.def("topk(Tensor keys, int k, bool sorted) -> Tensor");
    m.def("buffer_add(Tensor x, Tensor log_w, Tensor log_q_old, int start, Tensor(a!) buf_x, Tensor(b!) buf_log_w, "
          "Tensor(c!) buf_log_q_old) -> ()");
    m.def("buffer_sample_indices(Tensor log_w, int k) -> Tensor");
}

TORCH_LIBRARY_IMPL(fabhip, CUDA, m) {      // CUDA == HIP on PyTorch-ROCm; deliberately no CPU registration
    m.impl("realnvp_pack", realnvp_pack);
    m.impl("realnvp_sample", realnvp_sample);
    m.impl("realnvp_logprob_grad", realnvp_logprob_grad);
    m.impl("realnvp_logprob_tape", realnvp_logprob_tape);
    m.impl("realnvp_param_grad", realnvp_param_grad);
    m.impl("realnvp_sample_tape", realnvp_sample_tape);
    m.impl("realnvp_sample_grad_tape", realnvp_sample_grad_tape);
    m.impl("adam_clip_step", adam_clip_step);
    m.impl("buffer_train_step", buffer_train_step);
    m.impl("spline_pack", spline_pack);
    m.impl("spline_logprob_grad", spline_logprob_grad);
    m.impl("spline_logprob_tape", spline_logprob_tape);
    m.impl("spline_sample", spline_sample);
    m.impl("spline_sample_vjp_tape", spline_sample_vjp_tape);
    m.impl("target_logp_grad", target_logp_grad);
    m.impl("manywell_logp_grad", manywell_logp_grad);
    m.impl("gmm_logp_grad", gmm_logp_grad);
    m.impl("create_point", create_point);
    m.impl("hmc_transition", hmc_transition);
    m.impl("metropolis_transition", metropolis_transition);
    m.impl("ais_run", ais_run);
    m.impl("ais_phase", ais_phase);
    m.impl("ais_sharded_tuned", ais_sharded_tuned);
    m.impl("metropolis_adapt_gathered", metropolis_adapt_gathered);
    m.impl("tape_gemm", tape_gemm);
    m.impl("spline_ais_run", spline_ais_run);
    m.impl("hmc_adapt_gathered", hmc_adapt_gathered);
    m.impl("generic_workspace", generic_workspace);
    m.impl("hmc_generic_begin", hmc_generic_begin);
    m.impl("hmc_generic_leap_pre", hmc_generic_leap_pre);
    m.impl("hmc_generic_leap_post", hmc_generic_leap_post);
    m.impl("hmc_generic_accept", hmc_generic_accept);
    m.impl("spline_hmc_transition", spline_hmc_transition);
    m.impl("anneal_log_prob", anneal_log_prob);
    m.impl("log_w_update", log_w_update);
    m.impl("metropolis_generic_propose", metropolis_generic_propose);
    m.impl("metropolis_generic_accept", metropolis_generic_accept);
    m.impl("ess_logz", ess_logz);
    m.impl("fixed_cdf", fixed_cdf);
    m.impl("resample_multinomial", resample_multinomial);
    m.impl("resample_systematic", resample_systematic);
    m.impl("multinomial_torch", multinomial_torch);
    m.impl("gather_rows", gather_rows);
    m.impl("topk", topk);
    m.impl("buffer_add", buffer_add);
    m.impl("buffer_sample_indices", buffer_sample_indices);
}
