/*
 * avl_hip.h -- C ABI of libavl_hip.so: the MI355X (gfx950) implementation of the per-frame hot
 * path of AutonomousVehicleLaboratory/vision_semantic_segmentation.
 *
 * The reference has no FFI layer (it is pure Python); each entry point below names the reference
 * function whose arithmetic it replaces (file:line under the reference root).  The Python shim in
 * vision_semantic_segmentation_amd/ binds these with ctypes and keeps the reference's class and
 * method names; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is
 *     asynchronous on it, nothing synchronises, nothing allocates;
 *   - the caller owns every buffer, including scratch; the library keeps no pointer after return
 *     (plan objects excepted: they keep the op list they were given);
 *   - return value: 0 on success, a negative AVL_E_* code otherwise; avl_last_error() then holds
 *     a message for the calling thread.  Nothing throws.
 */
#ifndef AVL_HIP_H
#define AVL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVL_OK 0
#define AVL_E_ARG (-1)      /* bad argument (null pointer, non-positive size, bad enum) */
#define AVL_E_HIP (-2)      /* a HIP runtime call failed                                */
#define AVL_E_UNSUPPORTED (-3)

/* point / grid element types */
#define AVL_F32 0
#define AVL_F64 1
#define AVL_BF16 2
#define AVL_F16 3

/* vote-mask layout (one uint32 per grid cell, 0 between frames):
 *   bit i        (i < 16) : some point of map class i fell into the cell this frame
 *   bit 16 + i            : ... and one of them qualified for the intensity bonus (+2 on channel i) */
#define AVL_MAX_MAP_CLASSES 16

/* ---- library ---------------------------------------------------------------------------- */
const char* avl_version(void);
/* copies the calling thread's last error message into buf (NUL terminated); returns its length */
int avl_last_error(char* buf, int len);

/* ---- a7: SemanticMapping.project_pcd (src/mapping.py:357-389) ---------------------------- */

/* Projection only -- lines :367-383.  Point k's component c (0=x,1=y,2=z,3=intensity) is read at
 * pts + k*point_stride + c*comp_stride (bytes) as `dtype` and widened to double.
 * T (row-major 4x4) = T_origin_to_velodyne of :369, or NULL for pcd_frame_id == "velodyne" (:373).
 * P (row-major 3x4) = Camera.P (src/camera.py:28).
 * out_ixy int32[2][n] = dehomogenize(P Xv).astype(int32) (:375, INT_MIN for nan/inf/overflow),
 * out_mask uint8[n]   = the mask of :378-383.  Either may be NULL. */
int avl_project_points(const void* pts, int n, int dtype, int64_t point_stride, int64_t comp_stride,
                       const double* P_host, const double* T_host, double range_max,
                       int img_w, int img_h, int32_t* out_ixy, uint8_t* out_mask, void* stream);

/* bytes of scratch avl_project_pcd needs for n points */
int64_t avl_project_pcd_scratch_bytes(int n);

/* Whole project_pcd: projection, mask, ORDER-PRESERVING compaction (:385) and label gather (:387).
 * image uint8[img_h][img_w][3].  Outputs have capacity n and leading dimension out_ld (>= n):
 * out_pcd double[4][out_ld] (the ORIGINAL-frame columns of pcd that passed, Q6),
 * out_label uint8[3][out_ld], out_count int32[1] = M. */
int avl_project_pcd(const void* pts, int n, int dtype, int64_t point_stride, int64_t comp_stride,
                    const double* P_host, const double* T_host, double range_max,
                    const uint8_t* image, int img_w, int img_h,
                    double* out_pcd, uint8_t* out_label, int64_t out_ld, int32_t* out_count,
                    void* scratch, void* stream);

/* ---- a8: SemanticMapping.update_map (src/mapping.py:391-444) ------------------------------ */

typedef struct avl_grid {
    void* map;            /* [Hm][Wm][C] of map_dtype (AVL_F64 = the reference's type, or AVL_F32) */
    int map_dtype;
    int Hm, Wm, C;        /* Hm indexes x, Wm indexes y (src/mapping.py:115-116); C <= 16        */
    double off_x, off_y;  /* pcd_origin_offset (:404)                                             */
    double b00, b10;      /* map_boundary[0][0], map_boundary[1][0] (:408)                        */
    double resolution;    /* (:409)                                                               */
    uint32_t* cell_mask;  /* scratch uint32[Hm*Wm]; all zero on entry, all zero again on return   */
    int32_t* touched;     /* scratch int32[touched_cap]: cells first touched this frame           */
    int32_t touched_cap;  /* >= min(n points, Hm*Wm)                                              */
    int32_t* counter;     /* scratch int32[counter_len] (a block of its own); ints [0,4) are zeroed by the
                             calls that use them, ints [4, counter_len) must be zero on entry and are zero
                             again on return (per-list cursors of the partitioned touched lists)      */
    int32_t counter_len;  /* >= 4; >= AVL_COUNTER_INTS enables the partitioned lists of avl_fused_frame   */
} avl_grid;
#define AVL_COUNTER_INTS 256

/* update_map for points that already carry an RGB label (the reference's own signature).
 * pcd double[4][ld] (rows x,y,z,intensity), label uint8[3][ld]; the number of points is m_host,
 * or *m_dev when m_dev != NULL (then m_host is the capacity the launch is sized for).
 * label_colors_host uint8[C][3]; only R and G are compared (Q2).
 * cm_host double[C][C]: column i is added to every cell holding a class-i point, once (Q1).
 * bonus_classes: bit i set => class i gets +2 on channel i when a point with intensity < 2 or > 14
 * is present (:431-437); 0 when MAPPING.PCD.USE_INTENSITY is false. */
int avl_update_map(const avl_grid* g, const double* pcd, const uint8_t* label, int64_t ld,
                   int m_host, const int32_t* m_dev,
                   const uint8_t* label_colors_host, const double* cm_host, uint32_t bonus_classes,
                   void* stream);

/* The two stages of avl_update_map, separately.
 * avl_vote_points: :403-437 up to the `+=` -- every point ORs its vote into cell_mask[cell]; the
 * cells it turned non-zero are listed in g->touched[0 .. g->counter[0]).  (g->map is not touched.)
 * avl_grid_apply: the buffered `+=` (:424,:437) for the listed cells, then clears their masks.
 * rows == NULL: applied to g->map.  rows != NULL: row k of rows ([count][C], rows_dtype) stands for
 * cell touched[k] -- for callers whose grid lives in host memory and who move only touched rows. */
int avl_vote_points(const avl_grid* g, const double* pcd, const uint8_t* label, int64_t ld,
                    int m_host, const int32_t* m_dev, const uint8_t* label_colors_host,
                    uint32_t bonus_classes, void* stream);
int avl_grid_apply(const avl_grid* g, const double* cm_host, void* rows, int rows_dtype, void* stream);

/* ---- a9: one fused frame of SemanticMapping.mapping (src/mapping.py:314-319) --------------- */

/* semantic source kinds */
#define AVL_SRC_RGB 0       /* uint8[src_h][src_w][3] colour image, matched on R,G (the ROS topic)     */
#define AVL_SRC_CLASSMAP 1  /* uint8[src_h][src_w] network class ids + lut: the un-colourised argmax   */

/* project_pcd + update_map without materialising the intermediate point list: every point is
 * projected, its label fetched, its grid cell computed from the ORIGINAL coordinates, and its vote
 * OR-ed into cell_mask; then the touched cells are updated once.  For AVL_SRC_CLASSMAP the source is
 * sampled as cv2.resize(..., (img_w,img_h), INTER_NEAREST) would have enlarged it
 * (vision_semantic_segmentation_node.py:109-110) and lut_host uint32[256] maps a network class to its
 * vote bits (= R,G match of its palette colour against LABEL_COLORS). For AVL_SRC_RGB src_w/src_h must
 * equal img_w/img_h and label_colors_host is used instead of the lut. */
int avl_fused_frame(const avl_grid* g, const void* pts, int n, int dtype, int64_t point_stride,
                    int64_t comp_stride, const double* P_host, const double* T_host, double range_max,
                    int src_kind, const uint8_t* src, int src_w, int src_h, int img_w, int img_h,
                    const uint32_t* lut_host, const uint8_t* label_colors_host,
                    const double* cm_host, uint32_t bonus_classes, void* stream);

/* a6: colourised full-resolution semantic image from the small argmax map
 * (vision_semantic_segmentation_node.py:102,109-116): nearest upscale + palette LUT.
 * labels uint8[lh][lw]; palette_host uint8[256][3]; out uint8[out_h][out_w][3]. */
int avl_colorize_labels(const uint8_t* labels, int lw, int lh, const uint8_t* palette_host,
                        uint8_t* out, int out_w, int out_h, void* stream);


/* ---- the node's image pre-processing (vision_semantic_segmentation_node.py:83-98) -----------
 * cv2.cvtColor(BGR2RGB) -> cv2.undistort(K, dist) -> cv2.resize(INTER_AREA) by an integer factor, fused.
 * bgr uint8[h][w][3]; K_host double[9] (row-major 3x3) and dist_host double[5] (k1,k2,p1,p2,k3), both NULL to
 * skip the undistortion; rgb_out uint8[h/factor][w/factor][3]. */
int avl_preprocess_image(const uint8_t* bgr, int h, int w, const double* K_host, const double* dist_host, int factor,
                         uint8_t* rgb_out, void* stream);
/* The same with cv2.resize(INTER_AREA) to ANY smaller size out_h x out_w (vision_semantic_segmentation_node.py:92-98 takes every
 * IMAGE_SCALE in (0, 1): width = int(W * scale), height = int(H * scale)): OpenCV's area decimation for a non-integer ratio -- partial first
 * and last source pixels weighted by their overlap, float accumulation row by row, round half to even.  Integer ratios should keep
 * avl_preprocess_image (OpenCV switches to integer box means there; for 2 x 2 it rounds half UP).  Parity unpinned (OpenCV absent). */
int avl_preprocess_image_area(const uint8_t* bgr, int h, int w, const double* K_host, const double* dist_host, int out_h, int out_w,
                              uint8_t* rgb_out, void* stream);

/* The same pre-processing INSIDE the network's first kernel: an AVL_OP_STEM op whose `in2` is set reads the RAW BGR camera
 * frame through `in` and applies BGR->RGB / undistort / INTER_AREA per pixel while it fills its LDS tile (the function
 * avl_preprocess_image applies, so the results are the same bytes), i.e. the RGB network input is never written or re-read.
 * Geometry of such an op: in_h x in_w = the NETWORK input (what avl_preprocess_image would have produced), in2_ld = src_w,
 * in_rows = src_h * src_w of the raw frame; the integer factor is src_w / in_w (in_h == src_h / factor is checked).
 * `in2` points at AVL_STEM_CAMERA_BYTES of DEVICE memory holding the camera model, written by avl_stem_camera_set()
 * (stream-ordered: a plan captured into a hipGraph serves camera1 and camera6 alike; K_host / dist_host as above, both NULL
 * = no undistortion). */
#define AVL_STEM_CAMERA_BYTES 64
int avl_stem_camera_set(void* camera_dev, const double* K_host, const double* dist_host, void* stream);

/* ---- the semantic point cloud mapping() publishes (src/mapping.py:316-317) ---------------
 * create_point_cloud (src/utils/utils_ros.py:31-59) without its per-point struct.pack loop: record k (16 bytes,
 * point_step 16) = float32 x,y,z of pcd[0:3][k] and uint32 rgba = r | g<<8 | b<<16 | 255<<24 of label[:,k].
 * pcd double[4][ld], label uint8[3][ld] as avl_project_pcd returns them; count = m_host or *m_dev. */
int avl_pack_semantic_cloud(const double* pcd, const uint8_t* label, int64_t ld, int m_host, const int32_t* m_dev,
                            void* out_records, void* stream);

/* ---- the planar (no-LiDAR) mode, update_map_planar (src/mapping.py:465-488) -----------------
 * The semantic image is warped onto the grid by a homography (generate_homography, src/homography.py:22-76:
 * cv2.warpPerspective(image, H, (Wm, Hm)), INTER_LINEAR, zero border), every cell whose warped colour matches class i and
 * whose column is >= sep gets map[cell][i] += 1, then negative cells are clamped to 0 (:481).
 *   Hinv_host double[9]: the INVERSE homography (grid cell (x = column, y = row) -> image pixel), row-major;
 *   match_colour 0: the reference as written -- it compares a uint8 channel with the label NAME (:474), which is never
 *                   true, so nothing is added and only the clamp acts;  1: R,G colour match as in update_map (Q2).
 * OpenCV is absent here: float64 bilinear weights, round half to even (parity unpinned; oracle/planar_oracle.py). */
int avl_planar_update(void* map, int map_dtype, int Hm, int Wm, int C, const uint8_t* image, int img_h, int img_w,
                      const double* Hinv_host, int sep, const uint8_t* label_colors_host, int match_colour, void* stream);

/* pcd_callback (src/mapping.py:172-183) without its per-point Python loop: a sensor_msgs/PointCloud2 payload
 * (`data`, n_points records of point_step bytes, FLOAT32 fields x / y / z / intensity at the given byte offsets, all
 * multiples of 4) -> out_xyzi float32[n_points][4], the layout avl_fused_frame reads.  read_points(skip_nans=True)
 * (:179) drops every point with a NaN in ANY of the four fields; here such a point keeps its slot but gets x = NaN, which
 * the projection kernels reject, so frame results are identical and no compaction pass is needed.
 * n_valid (device int32, or NULL) receives the number of points read_points would have yielded. */
int avl_unpack_pointcloud2(const uint8_t* data, int64_t n_points, int point_step, int off_x, int off_y, int off_z, int off_i,
                           float* out_xyzi, int32_t* n_valid, void* stream);

/* ---- end-of-run rendering (src/renderer.py; called at src/mapping.py:332-334) ------------
 * map [Hm][Wm][C] of map_dtype (AVL_F64 | AVL_F32); colors_host uint8[C][3]; out uint8[Hm][Wm][3]. */
/* render_bev_map (renderer.py:32-59): colour of the arg-max channel, black where the channel sum is 0 */
int avl_render_bev_map(const void* map, int map_dtype, int Hm, int Wm, int C, const uint8_t* colors_host,
                       uint8_t* out, void* stream);
/* render_bev_map_with_thresholds (renderer.py:131-172); priority_host int32[C] (NULL = 0..C-1, low to high),
 * thresholds_host double[C] (NULL = 0.01 each) */
int avl_render_bev_map_thresholds(const void* map, int map_dtype, int Hm, int Wm, int C, const uint8_t* colors_host,
                                  const int32_t* priority_host, const double* thresholds_host, uint8_t* out, void* stream);
/* apply_filter (renderer.py:175-189): 3x3 mean, kernel float32(1/9), BORDER_REFLECT_101; dst != src */
int avl_grid_box_filter(const void* src, void* dst, int map_dtype, int Hm, int Wm, int C, void* stream);
/* End-of-run evaluation (test/test_semantic_mapping.py, called at src/mapping.py:341-344): convert_labels (:6-19) and the
 * counting part of Test.iou (:127-161) in one pass over the rendered colour map.
 *   color_map uint8[H][W][3]; mask uint8[>=H][mask_ld] (0 = invalid) or NULL;
 *   labels_out uint8[H][W] or NULL: 1 road (128,64,128), 2 crosswalk (140,140,200), 3 lane (255,255,255),
 *                                   4 sidewalk (244,35,232), 5 vegetation (107,142,35), 0 anything else / masked;
 *   gt uint8[>=H][gt_ld] or NULL: ground-truth labels (pointer already at the shift_w/shift_h origin of :125-126);
 *   counts uint64[64] (device): counts[g * 8 + l] = pixels with ground truth g (values > 7 counted as 7) and label l.
 * IoU(c) = counts[c][c] / (sum_l counts[c][l] + sum_g counts[g][c] - counts[c][c]) etc. are formed on the host. */
int avl_eval_map(const uint8_t* color_map, int H, int W, const uint8_t* mask, int mask_ld, const uint8_t* gt, int gt_ld,
                 uint8_t* labels_out, unsigned long long* counts, void* stream);

/* ---- a1-a5: segmentation forward (DeepLabV3+ / ResNeXt-50 OS8, eval mode) -------------------
 *
 * The reference builds the network from torch modules (src/semantic_segmentation.py:21-57,
 * src/network/deeplab_v3_plus/models/{deeplab_v3_plus,aspp,decoder}.py, torchvision ResNet).  Here
 * the Python host folds BatchNorm into the convolutions, lays activations out as NHWC
 * ([H*W rows][channels], row stride `ld` elements, so channel slices of a wider buffer give
 * torch.cat for free) and hands the library a flat list of ops; avl_seg_plan_run() launches them in
 * order on one stream.  Activations are AVL_BF16 / AVL_F16 (16x16x32 MFMA, fp32 accumulate) or AVL_F32
 * (fp32-input MFMA, the reference's precision).  Biases are always fp32.
 *
 * Every buffer named by an op must stay allocated while the plan lives; `*_rows` is the number of
 * rows actually allocated (GEMM tiles read whole 128-row tiles, so M is padded up by the caller and
 * the plan checks it). */

#define AVL_OP_STEM 1        /* uint8 RGB [H][W][3] -> normalise (semantic_segmentation.py:35-39) -> 7x7 s2 p3 conv +bias+ReLU */
#define AVL_OP_MAXPOOL 2     /* 3x3 s2 p1 (torchvision ResNet.maxpool)                                          */
#define AVL_OP_GEMM 3        /* 1x1 conv: out[m][n] = act(sum_k in[row(m)][k] w[n][k] + bias[n] (+ in2[m][n])) */
#define AVL_OP_GCONV 4       /* grouped 3x3 conv, stride 1|2, dilation d, pad d, +bias+ReLU (Bottleneck.conv2)  */
#define AVL_OP_DWCONV 5      /* depthwise 3x3 conv, dilation d, pad p, +bias+ReLU (core/nn/modules/conv.py:131)  */
#define AVL_OP_BILINEAR 6    /* F.interpolate(mode='bilinear', align_corners=True) (aspp.py:88, decoder.py:47)   */
#define AVL_OP_GAP 7         /* AdaptiveAvgPool2d((1,1)) -> fp32 [C] (aspp.py:69)                                */
#define AVL_OP_GEMV 8        /* out[n] = act(sum_k w[n][k] in[k] + bias[n]) on fp32 vectors (pooled branch)      */
#define AVL_OP_ARGMAX 9      /* torch.argmax(dim=1) over fp32 logits [M][C] -> uint8 (semantic_segmentation.py:56) */
#define AVL_OP_SUBSAMPLE 10  /* rows of a stride-s 1x1 conv's input (Bottleneck.downsample in layer2.0)          */
#define AVL_OP_DWPW 11       /* DepthwiseSeparableConv2d in ONE kernel (conv.py:103-141; the dilated ASPP branches):
                                depthwise 3x3 (dil d, pad d) +bias+ReLU -> 1x1 conv +bias+ReLU, 16-bit types only.
                                weight/bias = the 1x1 conv's (as AVL_OP_GEMM); in2 = depthwise parameters packed
                                [K/64][8][6][8] dwords: five tap pairs (taps 2p | 2p+1 << 16 in the activation type)
                                and the fp32 bias of each 8-channel chunk; then int32[ceil(H*W/128)]: the order in
                                which the 128-pixel tiles are visited (a permutation; tiles a dilation apart adjacent).
                                w_split = 3 (AVL_F16; what the "mixed" network emits): the EXACT depthwise stage -- FP32 depthwise
                                weights and a split depthwise result (three MFMA passes); in2 = float32 [K/64][8][10][8] per
                                8-channel chunk: rows 0 .. 8 = tap t of the chunk's eight channels, row 9 = the bias; then the
                                tile order as above.  `weight` is the 1x1 conv's [n][K/64][hi 64 | lo 64] as for w_split = 1.
                                (w_split = 2 -- depthwise weights as f16 pairs, rounds 3-5 -- is no longer accepted.)
                                With out_f32 (split input, w_split = 3, out_c = 256): the network's LAST 1x1 conv (decoder.py:42-43:
                                256 -> in3_c <= 32 classes, bias, no BN / ReLU) and torch.argmax (semantic_segmentation.py:56) run in the
                                epilogue on the block's result, which is never written: in3 = classifier weights f16 [hi | lo][32][256]
                                (rows >= in3_c zero), in2_lo = fp32 bias[32], out = fp32 logits [rows][in3_c] (out_ld = in3_c),
                                out_mx = uint8 labels[rows].
                                With in_lo (w_split = 3): the input is two f16 planes (the "mixed" decoder's refine blocks,
                                decoder.py:33-43; the ASPP branches of the complete hi + lo plan).  w_layout = 1 (with in_lo):
                                a tile is an 8 x 16 block of output pixels instead of 128 consecutive ones; the order array
                                then has ceil(out_h / 8) * ceil(out_w / 16) entries (tile = block row * ceil(out_w / 16) + block column). */

#define AVL_OP_BOTTLENECK 12 /* one torchvision Bottleneck of layer1 (backbone/resnet.py:24-43; stride 1, dilation 1, width 128 -> 256
                                channels) in ONE kernel, AVL_F16 "mixed" precision: conv1 1x1 +b+ReLU -> grouped 3x3 (32 groups, pad 1)
                                +b+ReLU -> conv3 1x1 +b (+ identity | + downsample 1x1) -> ReLU; the two intermediates live in LDS.
                                in (+ in_lo: enters the residual sum only) = block input, in_c = 256 (identity residual, w_layout 0)
                                or 64 (the block's downsample 1x1 runs as extra K steps of conv3: w_layout 1); out (+ out_lo).
                                Weights are f16 pairs hi + lo in MFMA FRAGMENT order ([...][hi, lo][lane 64][8], network.pack_bottleneck):
                                weight = conv1 [n 8][ks in_c/32], in2 = the 3x3 as block-diagonal 16-channel windows [window 8][ks 5]
                                (K = 32 = two taps x 16 channels), in3 = conv3 [wave 8][ks 4 (+ in_c/32 downsample steps)][nj 2];
                                in3_c = 128 (the width); bias = fp32 [b1 128 | b2 128 | b3 256 (+ downsample bias)].
                                w_split = 1 (in_c = 64 only): conv1's result keeps a lo plane in LDS (conv2 runs a third pass). */

typedef struct avl_seg_op {
    int32_t kind;            /* AVL_OP_*                                                        */
    int32_t dtype;           /* activation type of in/in2/out: AVL_BF16, AVL_F16 or AVL_F32      */
    const void* in;          /* input activation (STEM: uint8 image; GEMV/GAP-out: fp32)       */
    const void* in2;         /* GEMM: residual added before the ReLU, or NULL; GAP: fp32 scratch [256][C];
                                DWCONV: 32 zero bytes (what a tap outside the image reads);
                                STEM: NULL, or the camera block of a pre-processing stem (avl_stem_camera_set) */
    void* out;
    const void* weight;      /* packed by the host, layout per kind (see network.py)            */
    const float* bias;       /* fp32 [out_c padded], or NULL                                    */
    int32_t in_h, in_w, in_c, in_ld, in_rows;
    int32_t out_h, out_w, out_c, out_ld, out_rows;
    int32_t in2_ld;
    int32_t ksize, stride, pad, dil, groups;
    int32_t relu;
    int32_t out_f32;         /* GEMM: write fp32 (the logits) instead of `dtype`.  With out_mx != NULL (N <= 32, no residual, no ReLU): out_mx is
                                uint8 labels[out_rows] and the epilogue also writes torch.argmax over each row's N logits there (first maximal
                                index wins, a NaN counts as maximal; semantic_segmentation.py:56) -- AVL_OP_ARGMAX without its launch */
    int32_t w_rows;          /* GEMM: rows of `weight` allocated (out_c padded to the N tile)   */
    int32_t w_layout;        /* DWPW:  0 = tiles of 128 consecutive pixels, 1 = 8 x 16-pixel blocks (split input only, see AVL_OP_DWPW)
                                GCONV: 0 = float [group][tap][ci][co] (direct kernel),
                                       1 = bf16 block-diagonal 32-channel windows [window][2][9][16][32] (MFMA kernel)
                                STEM:  0 = float [7][7][3][64] (direct kernel), 1 = bf16 [4][6][16][32] (MFMA kernel)
                                GEMM:  0 = the library picks the kernel; 1 .. 4 force one tile configuration of the 16-bit
                                       kernels (A/B experiments: 1 = 128 x 128 two-buffer kernel, 2 = 256 x 128 ring,
                                       3 = 256 x 256 ring, 4 = 256 x 128 on four waves); 5 = EXPERIMENT, not used by the
                                       network: the plain 16-bit GEMM on one wave per SIMD (k_gemm_w4: 128 x 128 wave tiles,
                                       accumulators in AGPRs; N % 256 == 0, no split planes) -- tools/bench_gemm.py --variants 0,5 */
    int32_t w_split;         /* "mixed" precision (AVL_F16 only): 1 = `weight` holds each folded weight as an f16 pair
                                hi = f16(w), lo = f16(w - hi), packed per 64-wide K block in the order the kernel
                                walks it (GEMM/DWPW: [n][K/64][hi 64 | lo 64], or [hi | lo | hi] when the input is
                                split too; GCONV: 18 taps = 9 hi + 9 lo).  hi + lo carries ~22 significant bits:
                                the MFMAs run on both parts and accumulate in fp32.                             */
    int32_t mx_flags;        /* w_split = 2 only, see below: AVL_MX_IN_LO | AVL_MX_RES_LO | AVL_MX_OUT_LO                     */
    /* split activations (w_split modes): a tensor may be stored as TWO f16 planes of identical shape and stride,
     * value = hi + lo.  NULL = the tensor is a single f16 plane.  in_lo: the GEMM / depthwise input's low plane;
     * in2_lo: the residual's; out_lo: where the low part of the result goes (the op then rounds nothing away). */
    const void* in_lo;
    const void* in2_lo;
    void* out_lo;
    /* w_split = 2 (GEMM on gfx950's block-scaled matrix cores): `weight` is the plain f16 hi part [w_rows][K]; the correction
     * products run on MX-FP4 copies (OCP e2m1 elements, element 2i in the low nibble of byte i, one E8M0 scale per 32 values
     * along K) at 4x the f16 rate: Q4(W lo) x Q4(in hi) and, when in_lo is set, Q4(W hi) x Q4(in lo).
     * An "MX bundle" of a [rows][C] tensor (C % 256 == 0, dense rows) is laid out
     *     [FP4 plane of the hi part: rows x C/2 bytes][its scales: C/256 x rows x 8 bytes][the same two for the lo part]
     * w_mx: bundle of the weights (rows = w_rows; first Q4(W lo), then Q4(W hi)); in_mx: bundle of the input (rows = in_rows);
     * the WEIGHT bundle's scale arrays are permuted inside every 16-row block (network.permute_w_scales): for a 256-wide K block
     * the 128 scale bytes of rows r0..r0+15 are stored [row & 3][k quarter kq 0..3][n-tile (row >> 2) & 3][k half kk 0..1], i.e.
     * byte ((row & 3) * 4 + kq) * 8 + 2 * ((row >> 2) & 3) + kk holds the scale of row r0 + (row) for K block 4 kk + kq of the
     * eight 32-wide blocks: the eight bytes a lane of the kernel needs in a sub-step are then one aligned 8-byte word.
     * Activation bundles keep the natural [C/256][rows][8] order.
     * out_mx (GEMM with w_split = 2, or GCONV with w_split = 1): the op also writes the bundle of its OUTPUT (rows = out_rows;
     * the lo half only if out_lo is set), which is what the next MX GEMM reads as in_mx.
     * A tensor may keep its lo part ONLY as the FP4 half of its bundle (no f16 lo plane: 3 instead of 5 bytes per element
     * for the residual trunk).  mx_flags then says so: AVL_MX_IN_LO = the input's lo part is in in_mx (in_lo NULL);
     * AVL_MX_RES_LO = the residual's lo part is the FP4 lo half of in2_mx (in2_lo NULL; the 10 % error of FP4 applies to a
     * term that is 2^-11 of the sum); AVL_MX_OUT_LO = write the lo half of out_mx although out_lo is NULL.
     * A SECOND input (GEMM, w_split = 2): in3 [rows][in3_c] (row stride in3_ld, bundle in3_mx, same rows as `in`) is appended
     * along K -- out = W[:, :in_c] . in + W[:, in_c:] . in3: a Bottleneck's conv3 and its downsample 1x1 (stride 1) in ONE
     * product, the identity tensor never exists.  `weight` / `w_mx` then hold the concatenated [w_rows][in_c + in3_c] matrix.
     * in3 must be a dense tensor of its own (in3_ld == in3_c: its bundle's planes are addressed with that pitch). */
    const void* w_mx;
    const void* in_mx;
    void* out_mx;
    const void* in2_mx;
    const void* in3;
    const void* in3_mx;
    int32_t in3_c, in3_ld;
} avl_seg_op;

#define AVL_MX_IN_LO 1
#define AVL_MX_RES_LO 2
#define AVL_MX_OUT_LO 4

typedef struct avl_seg_plan avl_seg_plan;

/* copies the op list, validates shapes/strides/allocated rows against what the kernels read */
int avl_seg_plan_create(const avl_seg_op* ops_host, int n_ops, avl_seg_plan** out_plan);
void avl_seg_plan_destroy(avl_seg_plan* plan);
/* launches every op on `stream` (no sync); after avl_seg_plan_capture: one hipGraphLaunch */
int avl_seg_plan_run(avl_seg_plan* plan, void* stream);
/* records the op list into a hipGraph (stream capture on `stream`, non-NULL; run the plan once before) */
int avl_seg_plan_capture(avl_seg_plan* plan, void* stream);
/* same, with a hipEvent pair around every op; blocks until done; ms_host[n_ops] = op durations.
 * flops_host / bytes_host (either may be NULL) receive each op's algorithmic flops and bytes. */
int avl_seg_plan_profile(avl_seg_plan* plan, void* stream, float* ms_host, double* flops_host, double* bytes_host);
int avl_seg_plan_num_ops(const avl_seg_plan* plan);
/* diagnostic: runs the ops one by one and counts the Inf / NaN values of every op's output planes where they are produced
 * (counts_host[n_ops]; blocks until done, never inside a capture).  The 16-bit precisions turn an fp32 accumulator beyond the
 * type's range into Inf in the producing op's epilogue; a later ReLU can hide that from the logits.  SemanticSegmentation's
 * load-time self-check (semantic_segmentation.py:28-32 loads real checkpoints) refuses a plan with a non-zero count. */
int avl_seg_plan_nonfinite(avl_seg_plan* plan, void* stream, unsigned long long* counts_host);

#ifdef __cplusplus
}
#endif
#endif /* AVL_HIP_H */
