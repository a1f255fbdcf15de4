// Prints the layout facts that cross the plugin boundary as one JSON object.
// Built twice by tests/test_abi.py: against include/ggml_abi.h (ours) and, when the
// reference tree is present, against the reference's own headers (-DUSE_REFERENCE).
#ifdef USE_REFERENCE
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#include "ggml-impl.h"
#define GGML_COMMON_DECL_CPP
#include "ggml-common.h"
#else
#include "ggml_abi.h"
#endif
#include <cstdio>
#include <cstddef>

#define SZ(T)      printf("  \"sizeof(" #T ")\": %zu,\n", sizeof(T))
#define OFF(T, f)  printf("  \"offsetof(" #T "," #f ")\": %zu,\n", offsetof(T, f))
#define VAL(v)     printf("  \"" #v "\": %lld,\n", (long long)(v))

int main() {
    printf("{\n");
    SZ(struct ggml_tensor);
    OFF(struct ggml_tensor, type); OFF(struct ggml_tensor, buffer); OFF(struct ggml_tensor, ne);
    OFF(struct ggml_tensor, nb); OFF(struct ggml_tensor, op); OFF(struct ggml_tensor, op_params);
    OFF(struct ggml_tensor, flags); OFF(struct ggml_tensor, src); OFF(struct ggml_tensor, view_src);
    OFF(struct ggml_tensor, view_offs); OFF(struct ggml_tensor, data); OFF(struct ggml_tensor, name);
    OFF(struct ggml_tensor, extra);
    SZ(struct ggml_cgraph);
    OFF(struct ggml_cgraph, n_nodes); OFF(struct ggml_cgraph, n_leafs); OFF(struct ggml_cgraph, nodes);
    OFF(struct ggml_cgraph, leafs); OFF(struct ggml_cgraph, order);
    SZ(struct ggml_backend_buffer_type_i); SZ(struct ggml_backend_buffer_type);
    OFF(struct ggml_backend_buffer_type, device); OFF(struct ggml_backend_buffer_type, context);
    SZ(struct ggml_backend_buffer_i); SZ(struct ggml_backend_buffer);
    OFF(struct ggml_backend_buffer, buft); OFF(struct ggml_backend_buffer, context);
    OFF(struct ggml_backend_buffer, size); OFF(struct ggml_backend_buffer, usage);
    OFF(struct ggml_backend_buffer_i, get_base); OFF(struct ggml_backend_buffer_i, init_tensor);
    OFF(struct ggml_backend_buffer_i, set_tensor); OFF(struct ggml_backend_buffer_i, cpy_tensor);
    OFF(struct ggml_backend_buffer_i, reset);
    SZ(struct ggml_backend_i); SZ(struct ggml_backend);
    OFF(struct ggml_backend_i, synchronize); OFF(struct ggml_backend_i, graph_compute);
    OFF(struct ggml_backend_i, event_record); OFF(struct ggml_backend_i, event_wait);
    OFF(struct ggml_backend, iface); OFF(struct ggml_backend, device); OFF(struct ggml_backend, context);
    SZ(struct ggml_backend_event);
    SZ(struct ggml_backend_device_i); SZ(struct ggml_backend_device);
    OFF(struct ggml_backend_device_i, init_backend); OFF(struct ggml_backend_device_i, supports_op);
    OFF(struct ggml_backend_device_i, offload_op); OFF(struct ggml_backend_device_i, event_synchronize);
    OFF(struct ggml_backend_device, reg); OFF(struct ggml_backend_device, context);
    SZ(struct ggml_backend_reg_i); SZ(struct ggml_backend_reg);
    OFF(struct ggml_backend_reg, iface); OFF(struct ggml_backend_reg, context);
    SZ(struct ggml_backend_dev_props); OFF(struct ggml_backend_dev_props, type); OFF(struct ggml_backend_dev_props, caps);
    SZ(struct ggml_backend_dev_caps);
    SZ(block_q4_0); SZ(block_q8_0); SZ(block_q4_K); SZ(block_q5_K); SZ(block_q6_K);
    OFF(block_q4_K, scales); OFF(block_q4_K, qs); OFF(block_q5_K, qh); OFF(block_q5_K, qs);
    OFF(block_q6_K, qh); OFF(block_q6_K, scales); OFF(block_q6_K, d);
    VAL(GGML_MAX_DIMS); VAL(GGML_MAX_SRC); VAL(GGML_MAX_OP_PARAMS); VAL(GGML_MAX_NAME);
    VAL(GGML_KQ_MASK_PAD); VAL(GGML_BACKEND_API_VERSION);
    VAL(GGML_TYPE_F32); VAL(GGML_TYPE_F16); VAL(GGML_TYPE_Q4_0); VAL(GGML_TYPE_Q8_0); VAL(GGML_TYPE_Q8_1);
    VAL(GGML_TYPE_Q4_K); VAL(GGML_TYPE_Q5_K); VAL(GGML_TYPE_Q6_K); VAL(GGML_TYPE_Q8_K);
    VAL(GGML_TYPE_I32); VAL(GGML_TYPE_BF16); VAL(GGML_TYPE_COUNT);
    VAL(GGML_OP_NONE); VAL(GGML_OP_DUP); VAL(GGML_OP_ADD); VAL(GGML_OP_MUL); VAL(GGML_OP_CONCAT);
    VAL(GGML_OP_RMS_NORM); VAL(GGML_OP_MUL_MAT); VAL(GGML_OP_MUL_MAT_ID); VAL(GGML_OP_SCALE);
    VAL(GGML_OP_CPY); VAL(GGML_OP_CONT); VAL(GGML_OP_RESHAPE); VAL(GGML_OP_VIEW); VAL(GGML_OP_PERMUTE);
    VAL(GGML_OP_TRANSPOSE); VAL(GGML_OP_GET_ROWS); VAL(GGML_OP_SOFT_MAX); VAL(GGML_OP_ROPE);
    VAL(GGML_OP_FLASH_ATTN_EXT); VAL(GGML_OP_UNARY); VAL(GGML_OP_COUNT);
    VAL(GGML_UNARY_OP_RELU); VAL(GGML_UNARY_OP_SILU); VAL(GGML_UNARY_OP_GELU); VAL(GGML_UNARY_OP_COUNT);
    VAL(GGML_STATUS_ALLOC_FAILED); VAL(GGML_STATUS_FAILED); VAL(GGML_STATUS_SUCCESS); VAL(GGML_STATUS_ABORTED);
    VAL(GGML_BACKEND_BUFFER_USAGE_WEIGHTS); VAL(GGML_BACKEND_BUFFER_USAGE_COMPUTE);
    VAL(GGML_BACKEND_DEVICE_TYPE_CPU); VAL(GGML_BACKEND_DEVICE_TYPE_GPU); VAL(GGML_BACKEND_DEVICE_TYPE_ACCEL);
    VAL(GGML_TENSOR_FLAG_INPUT); VAL(GGML_TENSOR_FLAG_OUTPUT);
    VAL(GGML_ROPE_TYPE_NEOX); VAL(GGML_ROPE_TYPE_MROPE); VAL(GGML_ROPE_TYPE_VISION);
    printf("  \"_end\": 0\n}\n");
    return 0;
}
