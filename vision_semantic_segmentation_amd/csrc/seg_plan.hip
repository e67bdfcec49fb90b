// Segmentation plan: a validated, flat list of ops launched in order on one stream.
#include <vector>

#include "seg_types.h"

struct avl_seg_plan {
    std::vector<avl_seg_op> ops;
    std::vector<hipEvent_t> ev;   // 2 per op, created lazily by avl_seg_plan_profile
    hipGraph_t graph = nullptr;   // the op list captured once (avl_seg_plan_capture) ...
    hipGraphExec_t exec = nullptr;  // ... and replayed by avl_seg_plan_run with one launch
};

namespace avl {
namespace {

int validate(const avl_seg_op& op, int index) {
    int rc;
    if (op.kind == AVL_OP_GEMM) rc = validate_gemm(op);
    else if (op.kind == AVL_OP_DWPW) rc = validate_dwpw(op);
    else if (op.kind == AVL_OP_BOTTLENECK) rc = validate_bottleneck(op);
    else rc = validate_conv_op(op);
    if (rc != AVL_OK) {
        char msg[400];
        snprintf(msg, sizeof(msg), "%s", err_buf());
        return set_error(rc, "op %d (kind %d): %s", index, op.kind, msg);
    }
    return AVL_OK;
}

int launch(const avl_seg_op& op, hipStream_t s) {
    if (op.kind == AVL_OP_GEMM) return launch_gemm(op, s);
    if (op.kind == AVL_OP_DWPW) return launch_dwpw(op, s);
    if (op.kind == AVL_OP_BOTTLENECK) return launch_bottleneck(op, s);
    return launch_conv_op(op, s);
}

// algorithmic work of one op: flops (2 per MAC) and bytes (each tensor touched once)
// Bytes per element of an activation as the op touches it: the 16-bit (or fp32) plane, a second plane when it is split, and
// the FP4 copies of an MX bundle (half a byte per element and plane + one scale byte per 32 elements).
static double act_bytes(double es, bool lo_plane, bool mx, bool mx_lo) {
    double b = es + (lo_plane ? es : 0.0);
    if (mx) b += (mx_lo ? 2.0 : 1.0) * (0.5 + 1.0 / 32.0);
    return b;
}

void work(const avl_seg_op& op, double& flops, double& bytes) {
    const double es = elem_size(op.dtype);
    const double in_pix = (double)op.in_h * op.in_w, out_pix = (double)op.out_h * op.out_w;
    const double e_in = act_bytes(es, op.in_lo != nullptr, op.in_mx != nullptr, (op.mx_flags & AVL_MX_IN_LO) || op.in_lo);
    const double e_out = act_bytes(es, op.out_lo != nullptr, op.out_mx != nullptr, (op.mx_flags & AVL_MX_OUT_LO) || op.out_lo);
    flops = 0;
    bytes = in_pix * op.in_c * e_in + out_pix * op.out_c * e_out;
    switch (op.kind) {
        case AVL_OP_STEM:
            flops = 2.0 * out_pix * 64 * 147;
            bytes = (op.in2 ? (double)op.in_rows : in_pix) * 3 + out_pix * 64 * es;      // in2: the raw camera frame is what is read
            break;
        case AVL_OP_GEMM:
            flops = 2.0 * out_pix * op.out_c * (op.in_c + (op.in3 ? op.in3_c : 0));
            {
                const double k_all = op.in_c + (op.in3 ? op.in3_c : 0);
                const double e_w = op.w_split == 2 ? es + 2.0 * (0.5 + 1.0 / 32.0) : es * (op.w_split ? 2.0 : 1.0);      // hi + lo, or hi + two FP4 copies
                const double e_res = act_bytes(es, op.in2_lo != nullptr, (op.mx_flags & AVL_MX_RES_LO) != 0, false);
                bytes += (double)op.out_c * k_all * e_w + (op.in2 ? out_pix * op.out_c * e_res : 0.0) + (op.in3 ? in_pix * op.in3_c * e_in : 0.0);
            }
            if (op.out_f32) {
                bytes += out_pix * op.out_c * (4 - es);
                if (op.out_mx) bytes += out_pix - out_pix * op.out_c * (0.5 + 1.0 / 32.0);       // out_mx = the uint8 label map of the fused arg-max, not an MX bundle
            }
            break;
        case AVL_OP_GCONV:
            flops = 2.0 * out_pix * op.out_c * (op.in_c / op.groups) * 9;
            bytes += (double)op.out_c * (op.in_c / op.groups) * 9 * es * (op.w_split ? 2.0 : 1.0);
            break;
        case AVL_OP_DWCONV:
            flops = 2.0 * out_pix * op.out_c * 9;
            break;
        case AVL_OP_DWPW:
            flops = 2.0 * out_pix * op.in_c * 9 + 2.0 * out_pix * op.out_c * op.in_c;
            bytes += (double)op.out_c * op.in_c * es;
            if (op.out_f32) {      // the classifier + arg-max in the epilogue: the block's own result is never written; fp32 logits and uint8 labels are
                flops += 2.0 * out_pix * op.out_c * op.in3_c;
                bytes += out_pix * (op.in3_c * 4.0 + 1.0) - out_pix * op.out_c * e_out;
            }
            break;
        case AVL_OP_BILINEAR:
            flops = 8.0 * out_pix * op.out_c;
            break;
        case AVL_OP_BOTTLENECK: {
            // the three convolutions (+ the downsample 1x1) on the image's own pixels: halo recomputation is not algorithmic work;
            // bytes: input once, output once, weights once -- the intermediates never exist in memory
            const double width = op.in3_c, cg = width / op.groups;
            const double macs = op.in_c * width + width * cg * 9 + width * op.out_c + (op.w_layout ? (double)op.in_c * op.out_c : 0.0);
            flops = 2.0 * out_pix * macs;
            bytes += 2.0 * es * macs;
            break;
        }
        case AVL_OP_GAP:
            flops = in_pix * op.in_c;
            bytes = in_pix * op.in_c * es;
            break;
        case AVL_OP_GEMV:
            flops = 2.0 * op.in_c * op.out_c;
            bytes = 4.0 * op.in_c * op.out_c;
            break;
        case AVL_OP_ARGMAX:
            bytes = in_pix * op.in_c * 4 + in_pix;
            break;
        default:
            break;
    }
}

}  // namespace
}  // namespace avl

extern "C" int avl_seg_plan_create(const avl_seg_op* ops_host, int n_ops, avl_seg_plan** out_plan) {
    AVL_REQUIRE(ops_host && n_ops > 0 && out_plan, "avl_seg_plan_create: bad arguments");
    for (int i = 0; i < n_ops; ++i) {
        int rc = avl::validate(ops_host[i], i);
        if (rc) return rc;
    }
    avl_seg_plan* p = new avl_seg_plan();
    p->ops.assign(ops_host, ops_host + n_ops);
    *out_plan = p;
    return AVL_OK;
}

extern "C" void avl_seg_plan_destroy(avl_seg_plan* plan) {
    if (!plan) return;
    if (plan->exec) (void)hipGraphExecDestroy(plan->exec);
    if (plan->graph) (void)hipGraphDestroy(plan->graph);
    for (hipEvent_t e : plan->ev) (void)hipEventDestroy(e);
    delete plan;
}

extern "C" int avl_seg_plan_num_ops(const avl_seg_plan* plan) { return plan ? (int)plan->ops.size() : 0; }

// Captures the plan's ~90 launches into a hipGraph on `stream` (which must not be the legacy NULL stream).  The plan
// has to have run once before (kernel attributes are set on first launch, which is not allowed during capture).
// Every buffer is fixed at plan creation, so the graph stays valid for the plan's life; avl_seg_plan_run then
// replays it with a single hipGraphLaunch (host cost ~15 us instead of ~90 launches).
extern "C" int avl_seg_plan_capture(avl_seg_plan* plan, void* stream) {
    AVL_REQUIRE(plan && stream, "avl_seg_plan_capture needs a plan and a non-NULL stream");
    hipStream_t s = avl::as_stream(stream);
    if (plan->exec) { (void)hipGraphExecDestroy(plan->exec); plan->exec = nullptr; }
    if (plan->graph) { (void)hipGraphDestroy(plan->graph); plan->graph = nullptr; }
    AVL_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = AVL_OK;
    for (const avl_seg_op& op : plan->ops) {
        rc = avl::launch(op, s);
        if (rc) break;
    }
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return avl::set_error(AVL_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    plan->graph = g;
    AVL_HIP_CHECK(hipGraphInstantiate(&plan->exec, plan->graph, nullptr, nullptr, 0));
    return AVL_OK;
}

extern "C" int avl_seg_plan_run(avl_seg_plan* plan, void* stream) {
    AVL_REQUIRE(plan, "plan is NULL");
    hipStream_t s = avl::as_stream(stream);
    if (plan->exec) {
        AVL_HIP_CHECK(hipGraphLaunch(plan->exec, s));
        return AVL_OK;
    }
    for (const avl_seg_op& op : plan->ops) {
        int rc = avl::launch(op, s);
        if (rc) return rc;
    }
    return AVL_OK;
}

extern "C" int avl_seg_plan_profile(avl_seg_plan* plan, void* stream, float* ms_host, double* flops_host, double* bytes_host) {
    AVL_REQUIRE(plan && ms_host, "avl_seg_plan_profile: bad arguments");
    hipStream_t s = avl::as_stream(stream);
    const size_t n = plan->ops.size();
    while (plan->ev.size() < 2 * n) {
        hipEvent_t e;
        AVL_HIP_CHECK(hipEventCreate(&e));
        plan->ev.push_back(e);
    }
    for (size_t i = 0; i < n; ++i) {
        AVL_HIP_CHECK(hipEventRecord(plan->ev[2 * i], s));
        int rc = avl::launch(plan->ops[i], s);
        if (rc) return rc;
        AVL_HIP_CHECK(hipEventRecord(plan->ev[2 * i + 1], s));
    }
    AVL_HIP_CHECK(hipStreamSynchronize(s));
    for (size_t i = 0; i < n; ++i) {
        AVL_HIP_CHECK(hipEventElapsedTime(&ms_host[i], plan->ev[2 * i], plan->ev[2 * i + 1]));
        double f, b;
        avl::work(plan->ops[i], f, b);
        if (flops_host) flops_host[i] = f;
        if (bytes_host) bytes_host[i] = b;
    }
    return AVL_OK;
}

// ---- avl_seg_plan_nonfinite: the plan op by op, each op's output planes scanned for Inf / NaN where they are produced.
// The 16-bit modes convert fp32 accumulators to f16 / bf16 in every epilogue: a value beyond the type's range becomes Inf there, and
// a later ReLU can turn -Inf into a plausible 0 -- so an overflow is looked for at its source, not in the logits.  Diagnostic entry
// point (SemanticSegmentation's load-time self-check runs it once per checkpoint); never part of a frame.
namespace avl {
namespace {
template <typename T>
__global__ void k_count_nonfinite(const T* x, long long rows, int cols, int ld, unsigned long long* counter) {
    const long long n = rows * cols;
    unsigned long long bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = (float)x[(i / cols) * ld + (i % cols)];
        bad += (__builtin_isfinite(v) ? 0u : 1u);
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(counter, bad);
}
template <typename T>
void count_plane(const void* ptr, long long rows, int cols, int ld, unsigned long long* counter, hipStream_t s) {
    if (!ptr || rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL((k_count_nonfinite<T>), dim3(512), dim3(256), 0, s, static_cast<const T*>(ptr), rows, cols, ld, counter);
}
}  // namespace
}  // namespace avl

extern "C" int avl_seg_plan_nonfinite(avl_seg_plan* plan, void* stream, unsigned long long* counts_host) {
    AVL_REQUIRE(plan && counts_host, "avl_seg_plan_nonfinite: bad arguments");
    hipStream_t s = avl::as_stream(stream);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    AVL_HIP_CHECK(hipStreamIsCapturing(s, &cap));
    AVL_REQUIRE(cap == hipStreamCaptureStatusNone, "avl_seg_plan_nonfinite synchronises the stream: not while it is being captured");
    const size_t n = plan->ops.size();
    unsigned long long* dev = nullptr;
    AVL_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dev), n * sizeof(unsigned long long)));
    int rc = AVL_OK;
    hipError_t e = hipMemsetAsync(dev, 0, n * sizeof(unsigned long long), s);
    for (size_t i = 0; i < n && rc == AVL_OK && e == hipSuccess; ++i) {
        const avl_seg_op& op = plan->ops[i];
        rc = avl::launch(op, s);
        if (rc) break;
        if (op.kind == AVL_OP_ARGMAX) continue;                 // uint8 labels
        const long long rows = (op.kind == AVL_OP_GAP || op.kind == AVL_OP_GEMV) ? 1 : (long long)op.out_h * op.out_w;
        const bool f32 = op.dtype == AVL_F32 || op.out_f32 || op.kind == AVL_OP_GAP || op.kind == AVL_OP_GEMV;
        if (f32) avl::count_plane<float>(op.out, rows, (op.kind == AVL_OP_DWPW && op.out_f32) ? op.in3_c : op.out_c, op.out_ld, dev + i, s);
        else if (op.dtype == AVL_F16) {
            avl::count_plane<avl::f16>(op.out, rows, op.out_c, op.out_ld, dev + i, s);
            avl::count_plane<avl::f16>(op.out_lo, rows, op.out_c, op.out_ld, dev + i, s);
        } else avl::count_plane<avl::bf16>(op.out, rows, op.out_c, op.out_ld, dev + i, s);
        e = hipGetLastError();
    }
    if (rc == AVL_OK && e == hipSuccess) e = hipMemcpyAsync(counts_host, dev, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    if (rc == AVL_OK && e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dev);
    if (rc) return rc;
    if (e != hipSuccess) return avl::set_error(AVL_E_HIP, "avl_seg_plan_nonfinite: %s", hipGetErrorString(e));
    return AVL_OK;
}

namespace avl {
int launch_preprocess(const unsigned char*, int, int, const double*, const double*, int, unsigned char*, hipStream_t);
int launch_set_camera(void*, const double*, const double*, hipStream_t);
int launch_preprocess_area(const unsigned char*, int, int, const double*, const double*, int, int, unsigned char*, hipStream_t);
}
extern "C" int avl_preprocess_image_area(const uint8_t* bgr, int h, int w, const double* K_host, const double* dist_host, int out_h, int out_w,
                                         uint8_t* rgb_out, void* stream) {
    AVL_REQUIRE(bgr && rgb_out && h > 0 && w > 0, "bad image buffers");
    AVL_REQUIRE(out_h > 0 && out_w > 0 && out_h <= h && out_w <= w, "INTER_AREA shrinks: output %d x %d from %d x %d", out_h, out_w, h, w);
    AVL_REQUIRE((K_host == nullptr) == (dist_host == nullptr), "K_host and dist_host go together");
    return avl::launch_preprocess_area(bgr, h, w, K_host, dist_host, out_h, out_w, rgb_out, avl::as_stream(stream));
}
extern "C" int avl_stem_camera_set(void* camera_dev, const double* K_host, const double* dist_host, void* stream) {
    AVL_REQUIRE(camera_dev && reinterpret_cast<uintptr_t>(camera_dev) % 4 == 0, "bad camera block");
    AVL_REQUIRE((K_host == nullptr) == (dist_host == nullptr), "K_host and dist_host go together");
    return avl::launch_set_camera(camera_dev, K_host, dist_host, avl::as_stream(stream));
}
extern "C" int avl_preprocess_image(const uint8_t* bgr, int h, int w, const double* K_host, const double* dist_host, int factor,
                                    uint8_t* rgb_out, void* stream) {
    AVL_REQUIRE(bgr && rgb_out && h > 0 && w > 0, "bad image buffers");
    AVL_REQUIRE(factor >= 1 && h / factor > 0 && w / factor > 0, "downscale factor %d", factor);
    AVL_REQUIRE((K_host == nullptr) == (dist_host == nullptr), "K_host and dist_host go together");
    return avl::launch_preprocess(bgr, h, w, K_host, dist_host, factor, rgb_out, avl::as_stream(stream));
}
