// Binding recipe for oracle/_ref: exposes the reference's OWN native CPU attention kernels
// (compiled from /root/reference/python/sglang/kernels/aot/csrc/cpu/{decode,extend,kvcache}.cpp
// where they lie -- no reference source is copied) to Python.  The reference registers the same
// three functions through TORCH_LIBRARY in torch_extension_cpu.cpp:671-684,877-879; that file
// also registers ~60 unrelated ops, so this file declares just the three prototypes
// (torch_extension_cpu.cpp:189-226,535-541) and binds them with pybind11.
// TEST INFRASTRUCTURE ONLY (oracle pinning + optional cpu_baseline "reference").
#include <torch/extension.h>

#include <optional>

void decode_attention_cpu(at::Tensor& query, at::Tensor& k_cache, at::Tensor& v_cache,
                          at::Tensor& output, const std::optional<at::Tensor>& key,
                          const std::optional<at::Tensor>& value, at::Tensor& loc,
                          at::Tensor& attn_logits, at::Tensor& req_to_token,
                          at::Tensor& req_pool_indices, at::Tensor& seq_lens, double sm_scale,
                          double logit_cap, bool is_cross_attn, int64_t slidling_window_size,
                          std::optional<at::Tensor> encoder_lens, std::optional<at::Tensor> sinks);

void extend_attention_cpu(at::Tensor& q_extend, const std::optional<at::Tensor>& k_extend,
                          const std::optional<at::Tensor>& v_extend, at::Tensor& o_extend,
                          at::Tensor& k_buffer, at::Tensor& v_buffer, at::Tensor& req_to_token,
                          at::Tensor& req_pool_indices, at::Tensor& seq_lens,
                          at::Tensor& extend_seq_lens, at::Tensor& extend_start_loc,
                          int64_t max_len_extend, double sm_scale, double logit_cap,
                          bool is_cross_attn, int64_t sliding_window_size,
                          std::optional<at::Tensor> encoder_lens, std::optional<at::Tensor> sinks,
                          std::optional<at::Tensor> tree_mask);

void store_cache_cpu(const at::Tensor& k, const at::Tensor& v, const at::Tensor& k_cache,
                     const at::Tensor& v_cache, const at::Tensor& indices,
                     std::optional<int64_t> row_dim);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("decode_attention_cpu", &decode_attention_cpu);
  m.def("extend_attention_cpu", &extend_attention_cpu);
  m.def("store_cache_cpu", &store_cache_cpu);
}
