/*
 * ggml_abi.h -- clean-room statement of the binary interface between a ggml host
 * (libggml-base / libllama of mkjsym/EAGLE-in-llama.cpp, ggml build 0.0.4779,
 * GGML_BACKEND_API_VERSION 1) and a dynamically loaded backend plugin.
 *
 * The plugin in this repository (libggml-mi355x.so) is compiled against THIS file
 * only, so it builds on a machine that holds nothing but this repository.  Nothing
 * here is code: it is the layout of the structs and the numeric values of the enums
 * that cross the boundary, written out from the reference's interface
 *
 *     R = /root/reference/llama.cpp
 *     R/ggml/include/ggml.h            (tensor :578-613, types :351-390, ops :430-526)
 *     R/ggml/include/ggml-backend.h    (usage :49-53, dev type :130-137, props :140-159)
 *     R/ggml/src/ggml-backend-impl.h   (the five vtables :17-201, objects :31-35,60-66,119-129,181-201)
 *     R/ggml/src/ggml-impl.h           (struct ggml_cgraph :287-300, hash set :184-188)
 *     R/ggml/src/ggml-common.h         (quant block layouts :160-328)
 *
 * tests/test_abi.py compiles a probe against the real reference headers (when
 * /root/reference is present) and checks every sizeof/offsetof/enum value stated
 * here; tests/golden/abi_layout.json holds the values the probe printed so the same
 * check runs where the reference is absent.
 */
#ifndef GGML_MI355X_ABI_H
#define GGML_MI355X_ABI_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- limits (ggml.h:218-226) ---- */
#define GGML_MAX_DIMS       4
#define GGML_MAX_SRC        10
#define GGML_MAX_OP_PARAMS  64
#define GGML_MAX_NAME       64
#define GGML_KQ_MASK_PAD    64          /* ggml.h:1778 */
#define GGML_BACKEND_API_VERSION 1      /* ggml-backend-impl.h:11 */

/* ---- status (ggml.h:320-325) ---- */
enum ggml_status {
    GGML_STATUS_ALLOC_FAILED = -2,
    GGML_STATUS_FAILED       = -1,
    GGML_STATUS_SUCCESS      =  0,
    GGML_STATUS_ABORTED      =  1,
};

/* ---- tensor element types: only the values this backend knows by name ---- */
enum ggml_type {
    GGML_TYPE_F32  = 0,  GGML_TYPE_F16  = 1,
    GGML_TYPE_Q4_0 = 2,  GGML_TYPE_Q4_1 = 3,
    GGML_TYPE_Q5_0 = 6,  GGML_TYPE_Q5_1 = 7,
    GGML_TYPE_Q8_0 = 8,  GGML_TYPE_Q8_1 = 9,
    GGML_TYPE_Q2_K = 10, GGML_TYPE_Q3_K = 11,
    GGML_TYPE_Q4_K = 12, GGML_TYPE_Q5_K = 13,
    GGML_TYPE_Q6_K = 14, GGML_TYPE_Q8_K = 15,
    GGML_TYPE_I8   = 24, GGML_TYPE_I16  = 25, GGML_TYPE_I32 = 26, GGML_TYPE_I64 = 27,
    GGML_TYPE_F64  = 28, GGML_TYPE_BF16 = 30,
    GGML_TYPE_COUNT = 39,
};

enum ggml_prec { GGML_PREC_DEFAULT = 0, GGML_PREC_F32 = 1 };

/* ---- operations.  The numeric values are part of the ABI (the host stores them in
 *      ggml_tensor::op); they are listed explicitly so that a reordering upstream is
 *      caught by tests/test_abi.py instead of silently mis-dispatching. ---- */
enum ggml_op {
    GGML_OP_NONE = 0,
    GGML_OP_DUP = 1, GGML_OP_ADD = 2, GGML_OP_ADD1 = 3, GGML_OP_ACC = 4, GGML_OP_SUB = 5,
    GGML_OP_MUL = 6, GGML_OP_DIV = 7, GGML_OP_SQR = 8, GGML_OP_SQRT = 9, GGML_OP_LOG = 10,
    GGML_OP_SIN = 11, GGML_OP_COS = 12, GGML_OP_SUM = 13, GGML_OP_SUM_ROWS = 14,
    GGML_OP_MEAN = 15, GGML_OP_ARGMAX = 16, GGML_OP_COUNT_EQUAL = 17, GGML_OP_REPEAT = 18,
    GGML_OP_REPEAT_BACK = 19, GGML_OP_CONCAT = 20, GGML_OP_SILU_BACK = 21, GGML_OP_NORM = 22,
    GGML_OP_RMS_NORM = 23, GGML_OP_RMS_NORM_BACK = 24, GGML_OP_GROUP_NORM = 25,
    GGML_OP_MUL_MAT = 26, GGML_OP_MUL_MAT_ID = 27, GGML_OP_OUT_PROD = 28,
    GGML_OP_SCALE = 29, GGML_OP_SET = 30, GGML_OP_CPY = 31, GGML_OP_CONT = 32,
    GGML_OP_RESHAPE = 33, GGML_OP_VIEW = 34, GGML_OP_PERMUTE = 35, GGML_OP_TRANSPOSE = 36,
    GGML_OP_GET_ROWS = 37, GGML_OP_GET_ROWS_BACK = 38, GGML_OP_DIAG = 39,
    GGML_OP_DIAG_MASK_INF = 40, GGML_OP_DIAG_MASK_ZERO = 41, GGML_OP_SOFT_MAX = 42,
    GGML_OP_SOFT_MAX_BACK = 43, GGML_OP_ROPE = 44, GGML_OP_ROPE_BACK = 45, GGML_OP_CLAMP = 46,
    GGML_OP_CONV_TRANSPOSE_1D = 47, GGML_OP_IM2COL = 48, GGML_OP_IM2COL_BACK = 49,
    GGML_OP_CONV_TRANSPOSE_2D = 50, GGML_OP_POOL_1D = 51, GGML_OP_POOL_2D = 52,
    GGML_OP_POOL_2D_BACK = 53, GGML_OP_UPSCALE = 54, GGML_OP_PAD = 55,
    GGML_OP_PAD_REFLECT_1D = 56, GGML_OP_ARANGE = 57, GGML_OP_TIMESTEP_EMBEDDING = 58,
    GGML_OP_ARGSORT = 59, GGML_OP_LEAKY_RELU = 60, GGML_OP_FLASH_ATTN_EXT = 61,
    GGML_OP_FLASH_ATTN_BACK = 62, GGML_OP_SSM_CONV = 63, GGML_OP_SSM_SCAN = 64,
    GGML_OP_WIN_PART = 65, GGML_OP_WIN_UNPART = 66, GGML_OP_GET_REL_POS = 67,
    GGML_OP_ADD_REL_POS = 68, GGML_OP_RWKV_WKV6 = 69, GGML_OP_GATED_LINEAR_ATTN = 70,
    GGML_OP_UNARY = 71,
    GGML_OP_MAP_UNARY = 72, GGML_OP_MAP_BINARY = 73,
    GGML_OP_MAP_CUSTOM1_F32 = 74, GGML_OP_MAP_CUSTOM2_F32 = 75, GGML_OP_MAP_CUSTOM3_F32 = 76,
    GGML_OP_MAP_CUSTOM1 = 77, GGML_OP_MAP_CUSTOM2 = 78, GGML_OP_MAP_CUSTOM3 = 79,
    GGML_OP_CROSS_ENTROPY_LOSS = 80, GGML_OP_CROSS_ENTROPY_LOSS_BACK = 81,
    GGML_OP_OPT_STEP_ADAMW = 82,
    GGML_OP_COUNT = 83,
};

enum ggml_unary_op {
    GGML_UNARY_OP_ABS = 0, GGML_UNARY_OP_SGN = 1, GGML_UNARY_OP_NEG = 2, GGML_UNARY_OP_STEP = 3,
    GGML_UNARY_OP_TANH = 4, GGML_UNARY_OP_ELU = 5, GGML_UNARY_OP_RELU = 6, GGML_UNARY_OP_SIGMOID = 7,
    GGML_UNARY_OP_GELU = 8, GGML_UNARY_OP_GELU_QUICK = 9, GGML_UNARY_OP_SILU = 10,
    GGML_UNARY_OP_HARDSWISH = 11, GGML_UNARY_OP_HARDSIGMOID = 12, GGML_UNARY_OP_EXP = 13,
    GGML_UNARY_OP_COUNT = 14,
};

enum ggml_tensor_flag {
    GGML_TENSOR_FLAG_INPUT = 1, GGML_TENSOR_FLAG_OUTPUT = 2,
    GGML_TENSOR_FLAG_PARAM = 4, GGML_TENSOR_FLAG_LOSS   = 8,
};

/* RoPE mode bits (ggml.h: GGML_ROPE_TYPE_*) */
#define GGML_ROPE_TYPE_NEOX   2
#define GGML_ROPE_TYPE_MROPE  8
#define GGML_ROPE_TYPE_VISION 24

/* ---- opaque handles ---- */
typedef struct ggml_backend_buffer_type * ggml_backend_buffer_type_t;
typedef struct ggml_backend_buffer      * ggml_backend_buffer_t;
typedef struct ggml_backend_event       * ggml_backend_event_t;
typedef struct ggml_backend             * ggml_backend_t;
typedef struct ggml_backend_reg         * ggml_backend_reg_t;
typedef struct ggml_backend_device      * ggml_backend_dev_t;
typedef void                            * ggml_backend_graph_plan_t;
typedef uint8_t   ggml_guid[16];
typedef ggml_guid * ggml_guid_t;
typedef uint16_t  ggml_fp16_t;

/* ---- the tensor (ggml.h:578-613); sizeof == 336 on LP64 ---- */
struct ggml_tensor {
    enum ggml_type type;
    struct ggml_backend_buffer * buffer;
    int64_t ne[GGML_MAX_DIMS];           /* extent per dim, ne[0] fastest               */
    size_t  nb[GGML_MAX_DIMS];           /* stride in BYTES per dim                      */
    enum ggml_op op;
    int32_t op_params[GGML_MAX_OP_PARAMS / sizeof(int32_t)];
    int32_t flags;
    struct ggml_tensor * src[GGML_MAX_SRC];
    struct ggml_tensor * view_src;       /* non-NULL => this tensor aliases view_src     */
    size_t               view_offs;
    void * data;                         /* device address inside buffer                 */
    char   name[GGML_MAX_NAME];
    void * extra;                        /* backend-private                              */
    char   padding[8];
};

/* ---- compute graph as handed to graph_compute (ggml-impl.h:184-188,281-300) ---- */
struct ggml_hash_set {
    size_t size;
    uint32_t * used;
    struct ggml_tensor ** keys;
};
enum ggml_cgraph_eval_order { GGML_CGRAPH_EVAL_ORDER_LEFT_TO_RIGHT = 0, GGML_CGRAPH_EVAL_ORDER_RIGHT_TO_LEFT = 1 };
struct ggml_cgraph {
    int size;
    int n_nodes;
    int n_leafs;
    struct ggml_tensor ** nodes;
    struct ggml_tensor ** grads;
    struct ggml_tensor ** grad_accs;
    struct ggml_tensor ** leafs;
    struct ggml_hash_set visited_hash_set;
    enum ggml_cgraph_eval_order order;
};

/* ---- device description (ggml-backend.h:49-53,130-159) ---- */
enum ggml_backend_buffer_usage {
    GGML_BACKEND_BUFFER_USAGE_ANY = 0,
    GGML_BACKEND_BUFFER_USAGE_WEIGHTS = 1,
    GGML_BACKEND_BUFFER_USAGE_COMPUTE = 2,
};
enum ggml_backend_dev_type {
    GGML_BACKEND_DEVICE_TYPE_CPU   = 0,
    GGML_BACKEND_DEVICE_TYPE_GPU   = 1,
    GGML_BACKEND_DEVICE_TYPE_ACCEL = 2,
};
struct ggml_backend_dev_caps { bool async; bool host_buffer; bool buffer_from_host_ptr; bool events; };
struct ggml_backend_dev_props {
    const char * name;
    const char * description;
    size_t memory_free;
    size_t memory_total;
    enum ggml_backend_dev_type type;
    struct ggml_backend_dev_caps caps;
};
struct ggml_backend_feature { const char * name; const char * value; };  /* ggml-backend.h:195-198 */

/* ---- vtable 1/5: buffer type (ggml-backend-impl.h:17-35) ---- */
struct ggml_backend_buffer_type_i {
    const char *          (*get_name)      (ggml_backend_buffer_type_t buft);
    ggml_backend_buffer_t (*alloc_buffer)  (ggml_backend_buffer_type_t buft, size_t size);   /* NULL on OOM */
    size_t                (*get_alignment) (ggml_backend_buffer_type_t buft);
    size_t                (*get_max_size)  (ggml_backend_buffer_type_t buft);                /* optional */
    size_t                (*get_alloc_size)(ggml_backend_buffer_type_t buft, const struct ggml_tensor * t); /* optional */
    bool                  (*is_host)       (ggml_backend_buffer_type_t buft);                /* optional */
};
struct ggml_backend_buffer_type {
    struct ggml_backend_buffer_type_i iface;
    ggml_backend_dev_t device;
    void * context;
};

/* ---- vtable 2/5: buffer (ggml-backend-impl.h:41-66) ---- */
struct ggml_backend_buffer_i {
    void   (*free_buffer)  (ggml_backend_buffer_t buffer);
    void * (*get_base)     (ggml_backend_buffer_t buffer);
    void   (*init_tensor)  (ggml_backend_buffer_t buffer, struct ggml_tensor * t);
    void   (*memset_tensor)(ggml_backend_buffer_t buffer,       struct ggml_tensor * t, uint8_t value, size_t offset, size_t size);
    void   (*set_tensor)   (ggml_backend_buffer_t buffer,       struct ggml_tensor * t, const void * data, size_t offset, size_t size);
    void   (*get_tensor)   (ggml_backend_buffer_t buffer, const struct ggml_tensor * t,       void * data, size_t offset, size_t size);
    bool   (*cpy_tensor)   (ggml_backend_buffer_t buffer, const struct ggml_tensor * src, struct ggml_tensor * dst);
    void   (*clear)        (ggml_backend_buffer_t buffer, uint8_t value);
    void   (*reset)        (ggml_backend_buffer_t buffer);
};
struct ggml_backend_buffer {
    struct ggml_backend_buffer_i iface;
    ggml_backend_buffer_type_t   buft;
    void * context;
    size_t size;
    enum ggml_backend_buffer_usage usage;
};

/* ---- vtable 3/5: backend == one stream (ggml-backend-impl.h:87-129) ---- */
struct ggml_backend_i {
    const char * (*get_name)(ggml_backend_t backend);
    void (*free)(ggml_backend_t backend);
    void (*set_tensor_async)(ggml_backend_t backend,       struct ggml_tensor * t, const void * data, size_t offset, size_t size);
    void (*get_tensor_async)(ggml_backend_t backend, const struct ggml_tensor * t,       void * data, size_t offset, size_t size);
    bool (*cpy_tensor_async)(ggml_backend_t backend_src, ggml_backend_t backend_dst, const struct ggml_tensor * src, struct ggml_tensor * dst);
    void (*synchronize)(ggml_backend_t backend);
    ggml_backend_graph_plan_t (*graph_plan_create) (ggml_backend_t backend, const struct ggml_cgraph * cgraph);
    void                      (*graph_plan_free)   (ggml_backend_t backend, ggml_backend_graph_plan_t plan);
    void                      (*graph_plan_update) (ggml_backend_t backend, ggml_backend_graph_plan_t plan, const struct ggml_cgraph * cgraph);
    enum ggml_status          (*graph_plan_compute)(ggml_backend_t backend, ggml_backend_graph_plan_t plan);
    enum ggml_status          (*graph_compute)     (ggml_backend_t backend, struct ggml_cgraph * cgraph);
    void (*event_record)(ggml_backend_t backend, ggml_backend_event_t event);
    void (*event_wait)  (ggml_backend_t backend, ggml_backend_event_t event);
};
struct ggml_backend {
    ggml_guid_t guid;
    struct ggml_backend_i iface;
    ggml_backend_dev_t device;
    void * context;
};
struct ggml_backend_event {
    struct ggml_backend_device * device;
    void * context;
};

/* ---- vtable 4/5: device (ggml-backend-impl.h:137-185) ---- */
struct ggml_backend_device_i {
    const char * (*get_name)(ggml_backend_dev_t dev);
    const char * (*get_description)(ggml_backend_dev_t dev);
    void         (*get_memory)(ggml_backend_dev_t dev, size_t * free, size_t * total);
    enum ggml_backend_dev_type (*get_type)(ggml_backend_dev_t dev);
    void         (*get_props)(ggml_backend_dev_t dev, struct ggml_backend_dev_props * props);
    ggml_backend_t (*init_backend)(ggml_backend_dev_t dev, const char * params);
    ggml_backend_buffer_type_t (*get_buffer_type)(ggml_backend_dev_t dev);
    ggml_backend_buffer_type_t (*get_host_buffer_type)(ggml_backend_dev_t dev);
    ggml_backend_buffer_t (*buffer_from_host_ptr)(ggml_backend_dev_t dev, void * ptr, size_t size, size_t max_tensor_size);
    bool (*supports_op)(ggml_backend_dev_t dev, const struct ggml_tensor * op);
    bool (*supports_buft)(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft);
    bool (*offload_op)(ggml_backend_dev_t dev, const struct ggml_tensor * op);
    ggml_backend_event_t (*event_new)        (ggml_backend_dev_t dev);
    void                 (*event_free)       (ggml_backend_dev_t dev, ggml_backend_event_t event);
    void                 (*event_synchronize)(ggml_backend_dev_t dev, ggml_backend_event_t event);
};
struct ggml_backend_device {
    struct ggml_backend_device_i iface;
    ggml_backend_reg_t reg;
    void * context;
};

/* ---- vtable 5/5: registry entry (ggml-backend-impl.h:191-207) ---- */
struct ggml_backend_reg_i {
    const char *       (*get_name)(ggml_backend_reg_t reg);
    size_t             (*get_device_count)(ggml_backend_reg_t reg);
    ggml_backend_dev_t (*get_device)(ggml_backend_reg_t reg, size_t index);
    void *             (*get_proc_address)(ggml_backend_reg_t reg, const char * name);
};
struct ggml_backend_reg {
    int api_version;
    struct ggml_backend_reg_i iface;
    void * context;
};

/* ---- quantised block layouts (ggml-common.h:160-328); on-disk == in-memory == in HBM ---- */
#define QK_K 256
#define QK4_0 32
#define QK8_0 32
#pragma pack(push, 1)
typedef struct { ggml_fp16_t d; uint8_t qs[16]; }                                        block_q4_0;  /* 18 B / 32 */
typedef struct { ggml_fp16_t d; int8_t  qs[32]; }                                        block_q8_0;  /* 34 B / 32 */
typedef struct { ggml_fp16_t d, dmin; uint8_t scales[12]; uint8_t qs[128]; }             block_q4_K;  /* 144 B / 256 */
typedef struct { ggml_fp16_t d, dmin; uint8_t scales[12]; uint8_t qh[32]; uint8_t qs[128]; } block_q5_K; /* 176 B / 256 */
typedef struct { uint8_t ql[128]; uint8_t qh[64]; int8_t scales[16]; ggml_fp16_t d; }    block_q6_K;  /* 210 B / 256 */
#pragma pack(pop)

#ifdef __cplusplus
}
#endif
#endif /* GGML_MI355X_ABI_H */
