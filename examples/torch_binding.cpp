// torch_binding.cpp -- the reference-side binding of INTEGRATION.md 2, as a buildable file: a pybind11 / torch extension
// with the reference module's one function
//
//     forward(kernel_cfg, q, k, v, o, benchmark=False) -> (Tensor, float milliseconds)
//
// (/root/reference/src/flash_attention.cu:34-37,137-140) whose body keeps the tensor-level checks of the reference and hands
// the launch to libfa_hip.so through the C ABI (include/fa_hip.h).  This repository's own Python path does the same through
// ctypes (flash_attention_from_scratch_amd/flash_attention_kernels.py); this file is what a maintainer of the reference would
// compile instead of src/flash_attention.cu.  Plain C++: no device code, no torch types cross the C ABI.
//
//   from torch.utils.cpp_extension import load
//   ext = load("fa_ref_binding", ["examples/torch_binding.cpp"], extra_include_paths=["include", "/opt/rocm/include"],
//              extra_cflags=["-D__HIP_PLATFORM_AMD__"], extra_ldflags=["-L<repo>/flash_attention_from_scratch_amd/lib", "-lfa_hip",
//              "-Wl,-rpath,<repo>/flash_attention_from_scratch_amd/lib", "-L<torch>/lib", "-lc10_hip"])
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>  // (PyTorch-ROCm calls its HIP devices "cuda": the stream API carries the long name)
#include <c10/core/DeviceGuard.h>
#include <torch/extension.h>

#include <optional>
#include <tuple>

#include "fa_hip.h"

namespace {

// kernel_cfg is any Python object with the reference's 13 attributes (kernel_configs.py:106-120); dtype is an
// enum whose integer value is torch's ScalarType code (5 = fp16, 15 = bf16).
fa_fwd_config config_from(const py::object &cfg) {
    auto i32 = [&](const char *name) { return (int32_t)py::cast<long>(cfg.attr(name)); };
    auto flag = [&](const char *name) { return (int32_t)(py::cast<bool>(cfg.attr(name)) ? 1 : 0); };
    fa_fwd_config c{};
    c.dtype = (int32_t)py::cast<long>(py::int_(cfg.attr("dtype")));
    c.d_head = i32("d_head");
    c.B_r = i32("B_r");
    c.B_c = i32("B_c");
    c.n_warps = i32("n_warps");
    c.async_copy = flag("async_copy");
    c.eager_load_blocks = flag("eager_load_blocks");
    c.swizzled = flag("swizzled");
    c.Q_mma_load_K_tiles = i32("Q_mma_load_K_tiles");
    c.K_mma_load_K_tiles = i32("K_mma_load_K_tiles");
    c.V_mma_load_K_tiles = i32("V_mma_load_K_tiles");
    c.mma_double_buffer_loads = flag("mma_double_buffer_loads");
    c.optimized_softmax = flag("optimized_softmax");
    return c;
}

void check_tensor(const torch::Tensor &t, const char *what) {
    TORCH_CHECK(t.is_cuda(), what, " must be a CUDA tensor");
    TORCH_CHECK(t.is_contiguous(), what, " must be contiguous");
}

std::tuple<torch::Tensor, float> forward(const py::object &kernel_cfg, const torch::Tensor &q, const torch::Tensor &k,
                                         const torch::Tensor &v, std::optional<torch::Tensor> o, bool benchmark) {
    check_tensor(q, "q");
    check_tensor(k, "k");
    check_tensor(v, "v");
    const c10::DeviceGuard on_q_device(q.device());  // the reference's CUDAGuard, flash_attention.cu:42

    const auto dtype = q.scalar_type();
    TORCH_CHECK(dtype == torch::kFloat16 || dtype == torch::kBFloat16, "Only fp16 and bf16 are supported");
    TORCH_CHECK(k.scalar_type() == dtype && v.scalar_type() == dtype, "Input tensors must have the same data type");
    TORCH_CHECK(q.dim() == 4, "q must have shape (batch, seq_len, n_heads, d_head)");

    fa_fwd_args a{};
    a.cfg = config_from(kernel_cfg);
    TORCH_CHECK(fa_fwd_supported(&a.cfg), "Kernel configuration was not found in flash_kernels (libfa_hip.so registry)");
    TORCH_CHECK(a.cfg.dtype == (int32_t)dtype, "Kernel configuration dtype does not match input dtype");
    TORCH_CHECK(q.sizes() == k.sizes(), "Query and key tensors have same shape");
    TORCH_CHECK(q.sizes() == v.sizes(), "Query and value tensors have same shape");

    torch::Tensor out;
    if (o.has_value()) {
        out = *o;
        check_tensor(out, "o");
        TORCH_CHECK(out.scalar_type() == dtype, "Output tensor must have the same dtype as inputs");
        TORCH_CHECK(out.sizes() == q.sizes(), "Query and output tensors have same shape");
    } else {
        out = torch::empty_like(q);
    }

    a.q = q.data_ptr();
    a.k = k.data_ptr();
    a.v = v.data_ptr();
    a.o = out.data_ptr();
    a.batch = q.size(0);
    a.seq_len = q.size(1);
    a.n_heads = q.size(2);
    a.d_head = q.size(3);
    a.batch_stride = q.stride(0);
    a.seq_stride = q.stride(1);
    a.head_stride = q.stride(2);

    // the seq_len % B_r / B_c rules, the d_head match and the alignment checks live behind the ABI (same messages)
    void *stream = (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();  // torch's current stream, :118
    float ms = 0.0f;
    const int rc = benchmark ? fa_fwd_launch_timed(&a, stream, &ms) : fa_fwd_launch(&a, stream);
    TORCH_CHECK(rc == FA_OK, fa_last_error());
    return {out, ms};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("forward", &forward, py::arg("kernel_cfg"), py::arg("q"), py::arg("k"), py::arg("v"), py::arg("o"),
          py::arg("benchmark") = false, "Flash Attention forward (libfa_hip.so, gfx950)");
    m.def("version", [] { return std::string(fa_version()); });
}
