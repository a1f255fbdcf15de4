// split.cpp -- row-split (tensor-parallel) weight buffer type.
// Reference behaviour: R/ggml/src/ggml-cuda/ggml-cuda.cu:720-1046 (ggml_backend_cuda_split_buffer_*),
// one process drives all devices and gathers through the main GPU.  This framework scales as one
// process per GPU over RCCL instead (see host/ and DESIGN.md, "multi-GPU"), so inside a single process
// the split buffer type is declined: returning NULL makes the reference fall back to -sm layer
// (R/src/llama-model.cpp:310-322 checks the returned pointer).
#include "mi355x_common.h"
#include "ggml_mi355x.h"

extern "C" GGML_MI355X_API ggml_backend_buffer_type_t ggml_backend_mi355x_split_buffer_type(int main_device, const float * tensor_split) {
    (void) main_device; (void) tensor_split;
    static bool warned = false;
    if (!warned) { MI_LOG("in-process row split is not provided; tensor parallelism runs one process per GPU (RCCL)"); warned = true; }
    return nullptr;
}
