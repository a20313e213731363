// torch.ops.sesrq.forward, registered in C++ (TORCH_LIBRARY) and built with torch.utils.cpp_extension -- the binding SURVEY 8(b) and the
// north star name: "a thin C-ABI .so is called from Python via torch.utils.cpp_extension ... passes tensor.data_ptr() and the current HIP
// stream".  Replaces the reference's `gfake = model(inps)` (sim.py:205) as ONE operator of the lowered fx graph.  torch is plumbing here:
// output / workspace memory from the caching allocator, the current HIP stream, the dispatcher; the work is sesrq_forward (libsesrq.so).
//   (q, y) = torch.ops.sesrq.forward(x, engine_id)          q: int8 input.L after PixelShuffle, y: fp32 -- what the reference returns
//   torch.ops.sesrq.forward_into(x, engine_id, out_q, out_f, workspace, stream=0)   caller-owned buffers (either output may be None): no
//                                                             allocation; stream: a raw hipStream_t as an int, 0 = torch's current stream
// engine_id: an operator schema cannot carry a pointer, so the immutable device net travels as an integer registered through
// sesrq_torch_register (sesrq/torch_op.py calls it over ctypes when an Engine is registered, and sesrq_torch_unregister when it closes).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <mutex>
#include <unordered_map>

#include "sesrq.h"

namespace {
// a registered engine: its device net (NULL for a shape-only handle: tracing without a device) and its channel geometry
struct Entry { const sesrq_net *net; int cin, cout, r; };
std::mutex g_mu;
std::unordered_map<int64_t, Entry> g_nets;

Entry entry_of(int64_t id) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_nets.find(id);
    TORCH_CHECK(it != g_nets.end(), "sesrq::forward: engine handle ", id, " is not registered (or its Engine was destroyed)");
    return it->second;
}
const sesrq_net *net_of(const Entry &e) {
    TORCH_CHECK(e.net, "sesrq::forward: this handle carries shapes only (no device net): it can be traced, not run");
    return e.net;
}

struct Geometry { int dt; int64_t N, H, W, C, Ho, Wo; };
Geometry geometry(const at::Tensor &x, const Entry &e) {
    TORCH_CHECK_VALUE(x.dim() == 4, "Expect input tensor dimension: 4, but get ", x.dim());
    TORCH_CHECK_VALUE(x.scalar_type() == at::kFloat || x.scalar_type() == at::kChar, "input must be float32 (frame) or int8 (already quantised q0)");
    TORCH_CHECK_VALUE(x.size(1) == e.cin, "expected ", e.cin, " input channels, got ", x.size(1));
    TORCH_CHECK_VALUE(x.size(0) >= 1 && x.size(2) >= 1 && x.size(3) >= 1, "empty frame: N, H and W must be positive");
    return {x.scalar_type() == at::kFloat ? SESRQ_F32 : SESRQ_I8, x.size(0), x.size(2), x.size(3), e.cout / (e.r * e.r), x.size(2) * e.r, x.size(3) * e.r};
}

// stream: a raw hipStream_t handed over as an integer (torch.cuda.Stream.cuda_stream), or 0 = torch's current stream of the input's device
void run(const sesrq_net *net, const at::Tensor &x, const Geometry &g, void *q, void *y, const at::Tensor &ws, int64_t stream = 0) {
    const hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c10::hip::getCurrentHIPStream(x.device().index()).stream();
    const int rc = sesrq_forward(net, x.data_ptr(), g.dt, q, y, (int)g.N, (int)g.H, (int)g.W, ws.data_ptr(), (size_t)ws.numel(), (void *)st);
    TORCH_CHECK(rc == 0, "sesrq: ", sesrq_last_error());
}

std::tuple<at::Tensor, at::Tensor> forward_hip(const at::Tensor &x_, int64_t id) {
    const Entry en = entry_of(id);
    const sesrq_net *net = net_of(en);
    const Geometry g = geometry(x_, en);
    const c10::DeviceGuard guard(x_.device());
    const at::Tensor x = x_.contiguous();
    at::Tensor q = at::empty({g.N, g.C, g.Ho, g.Wo}, x.options().dtype(at::kChar));
    at::Tensor y = at::empty({g.N, g.C, g.Ho, g.Wo}, x.options().dtype(at::kFloat));
    // the workspace comes from the caching allocator and goes back to it on return: allocation and use are ordered on the current stream
    const at::Tensor ws = at::empty({(int64_t)sesrq_workspace_bytes(net, (int)g.N, (int)g.H, (int)g.W)}, x.options().dtype(at::kByte));
    run(net, x, g, q.data_ptr(), y.data_ptr(), ws);
    return {q, y};
}

std::tuple<at::Tensor, at::Tensor> forward_meta(const at::Tensor &x, int64_t id) {
    const Geometry g = geometry(x, entry_of(id));
    return {at::empty({g.N, g.C, g.Ho, g.Wo}, x.options().dtype(at::kChar)), at::empty({g.N, g.C, g.Ho, g.Wo}, x.options().dtype(at::kFloat))};
}

void forward_into_hip(const at::Tensor &x, int64_t id, const c10::optional<at::Tensor> &out_q, const c10::optional<at::Tensor> &out_f,
                      const at::Tensor &workspace, int64_t stream) {
    const Entry en = entry_of(id);
    const sesrq_net *net = net_of(en);
    const Geometry g = geometry(x, en);
    TORCH_CHECK_VALUE(x.is_contiguous() && workspace.is_contiguous() && workspace.scalar_type() == at::kByte, "forward_into: contiguous input, uint8 workspace");
    TORCH_CHECK_VALUE(out_q.has_value() || out_f.has_value(), "forward_into: both outputs are None");
    for (const auto *o : {&out_q, &out_f})
        if (o->has_value()) {
            const at::Tensor &t = o->value();
            TORCH_CHECK_VALUE(t.is_contiguous() && t.device() == x.device() && t.dim() == 4 && t.size(0) == g.N && t.size(1) == g.C && t.size(2) == g.Ho &&
                                  t.size(3) == g.Wo && t.scalar_type() == (o == &out_q ? at::kChar : at::kFloat),
                              "forward_into: outputs must be contiguous (N, C, H*r, W*r) tensors on the input's device, int8 / float32");
        }
    const c10::DeviceGuard guard(x.device());
    run(net, x, g, out_q.has_value() ? out_q->data_ptr() : nullptr, out_f.has_value() ? out_f->data_ptr() : nullptr, workspace, stream);
}
}  // namespace

TORCH_LIBRARY(sesrq, m) {
    m.def("forward(Tensor x, int engine_id) -> (Tensor, Tensor)");
    m.def("forward_into(Tensor x, int engine_id, Tensor(a!)? out_q, Tensor(b!)? out_f, Tensor(c!) workspace, int stream=0) -> ()");
}
TORCH_LIBRARY_IMPL(sesrq, CUDA, m) {      // HIP tensors dispatch on the CUDA key in PyTorch-ROCm
    m.impl("forward", &forward_hip);
    m.impl("forward_into", &forward_into_hip);
}
TORCH_LIBRARY_IMPL(sesrq, Meta, m) { m.impl("forward", &forward_meta); }

extern "C" {
// net == NULL: a shape-only handle (cin, cout = channels of the last conv, r = PixelShuffle factor); with a net the geometry is the net's own
int sesrq_torch_register(int64_t id, const void *net, int cin, int cout, int r) {
    Entry e = {static_cast<const sesrq_net *>(net), cin, cout, r};
    if (e.net && sesrq_net_shape(e.net, &e.cin, &e.cout, &e.r) != 0) return 1;
    if (e.cin < 1 || e.cout < 1 || e.r < 1 || e.cout % (e.r * e.r)) return 1;
    std::lock_guard<std::mutex> lk(g_mu);
    g_nets[id] = e;
    return 0;
}
int sesrq_torch_unregister(int64_t id) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_nets.erase(id) ? 0 : 1;
}
}
