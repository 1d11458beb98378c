/* libdif -- C ABI of the MI355X-native embedding + match hot path.
 *
 * The reference (sandyz1000/deep-insight-face) has no FFI: its seam is a duck-typed
 * Python protocol.  Each entry point below names the
 * reference interface it stands behind (paths relative to the reference root).  The
 * Python host side (deep-insight-face_amd/deep_insight_face/) binds these with ctypes
 * and keeps the reference's call signatures; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; dif_last_error()
 *     returns the message of the calling thread's last failure;
 *   - pointers named *_dev are DEVICE pointers borrowed for the duration of the call
 *     (stream-ordered: the work is enqueued on `stream`, a hipStream_t passed as
 *     void*; NULL = the default stream); host pointers are named *_host;
 *   - handles own their device memory (weights, gallery copy, workspaces) and are not
 *     re-entrant: one handle per process/GPU, calls on one handle from one thread;
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails.
 */
#ifndef DIF_H
#define DIF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIF_VERSION 110 /* 1.1: + dif_gallery_update / _reserve / _capacity, dif_*_option_name, options "sk2", "mt"; gallery option "frag" */

/* distance metrics: evaluation/utility.py:52-66 */
#define DIF_METRIC_SQL2 0   /* sum((a-b)^2, axis=1)                     utility.py:53-56 */
#define DIF_METRIC_COSINE 1 /* arccos(a.b / (|a||b|)) / pi              utility.py:57-62 */
#define DIF_METRIC_SIMILARITY 2 /* dif_pairwise only: the cosine similarity a.b / (|a||b|) itself, i.e.
                                 * utility.py:58-60 before the arccos (common/losses.py:39-40 precedent) */

/* input layouts / dtypes accepted by dif_net_embed */
#define DIF_LAYOUT_NHWC 0 /* the reference's layout (networks/inceptionv3.py:94,98) */
#define DIF_LAYOUT_NCHW 1 /* north_star's torch-side layout */
#define DIF_DTYPE_F32 0
#define DIF_DTYPE_U8 1
/* flags of dif_net_set_input_transform */
#define DIF_INPUT_BGR 1   /* swap R and B (keras vgg16 preprocess_input, predictions.py:95) */
#define DIF_INPUT_HFLIP 2 /* mirror every image left-right (scripts/insight_face.py:117-118 use_flipped_images) */

typedef struct dif_gallery dif_gallery;
typedef struct dif_net dif_net;
typedef struct dif_arcmargin dif_arcmargin;

int dif_version(void);
const char* dif_last_error(void);
/* number of visible HIP devices (0 without a GPU); never fails */
int dif_device_count(void);
/* measurement aid (no reference counterpart): the shader clock in GHz that the current device holds under ~20 ms of
 * back-to-back v_mfma_f32_32x32x2_f32 (pseudo-random operands) on every SIMD, and the TFLOP/s that loop sustained (may be null).  bench.py
 * records it next to its rooflines, which price against the 2.4 GHz peak. */
int dif_probe_mfma_clock(double* ghz_out, double* tflops_out, void* stream);

/* ------------------------------------------------------------------ distances
 * Row-paired distance, out_dev[i] = d(e1[i], e2[i]); n1 or n2 may be 1 (NumPy
 * broadcast of a single row, which is how a probe is compared with a whole gallery
 * in the reference's terms).  Replaces evaluation/utility.py:52-66 `distance`
 * (and its twin :174-188).  metric other than 0/1 fails like the reference's
 * RuntimeError('Undefined distance metric %d').  NaN where the reference gives NaN. */
int dif_pairwise(const float* e1_dev, int64_t n1, const float* e2_dev, int64_t n2, int d, int metric,
                 float* out_dev, void* stream);

/* LFW-protocol threshold sweep (evaluation/utility.py:36-49 calculate_accuracy and :69-77
 * calculate_val_far, as looped by calculate_roc :153-161 and calculate_val :104-107): for every
 * threshold t and test fold f, counts_dev[(f*T + t)*2 + 0/1] = number of pairs of fold f with
 * dist < thresholds[t] that are same / different.  fold_dev[i] = fold whose test split holds
 * pair i (KFold(shuffle=False): contiguous ranges). */
int dif_threshold_counts(const float* dist_dev, const uint8_t* issame_dev, const int32_t* fold_dev, int64_t n,
                         const double* thresholds_dev, int n_thresholds, int n_folds, int32_t* counts_dev,
                         void* stream);

/* ------------------------------------------------------------------ detector post-processing
 * YOLOv3-face box decode and suppression (detector/yolov3.py:36-172).
 * dif_yolo_decode: feats_dev = HOST array of n_layers DEVICE pointers, coarse grid first, each
 * [n_images][gh][gw][3*(5+n_classes)]; grid_hw_host [n_layers][2]; anchors_host [n_layers][3][2]
 * (pixels of the network input, already selected per layer); image_shape_dev [n_images][2] =
 * original (height, width).  Writes boxes_dev [n_images][n_boxes][4] (y_min, x_min, y_max, x_max in
 * image pixels; n_boxes = sum gh*gw*3) and scores_dev [n_images][n_boxes][n_classes] =
 * confidence * class probability (yolo_head :36-66, correct_boxes :69-93, boxes_and_scores :96-106).
 * dif_nms: per (image, class) keep boxes with score >= score_threshold and run the greedy
 * suppression of tf.image.non_max_suppression (get_yolo_output :149-160): keep_idx_dev
 * [n_images][n_classes][max_boxes] (box indices in pick order, -1 padded), keep_count_dev
 * [n_images][n_classes]; alive_ws_dev = n_images*n_classes*n_boxes bytes of scratch. */
int dif_yolo_decode(const float* const* feats_dev, const int32_t* grid_hw_host, const float* anchors_host,
                    int n_layers, int n_images, int n_classes, int input_h, int input_w,
                    const float* image_shape_dev, float* boxes_dev, float* scores_dev, void* stream);
int dif_nms(const float* boxes_dev, const float* scores_dev, int n_images, int n_boxes, int n_classes, int max_boxes,
            float score_threshold, float iou_threshold, uint8_t* alive_ws_dev, int32_t* keep_idx_dev,
            int32_t* keep_count_dev, void* stream);

/* Image resampling around the detector, uint8 NHWC in and out.
 * dif_letterbox: aspect-preserving BICUBIC resize (PIL semantics incl. antialiasing) onto a
 *   size x size canvas of (128,128,128)                       detector/yolov3.py:108-119
 * dif_crop_resize: per frame, box (left, top, right, bottom) + margin/2 per side, clamped
 *   (detector/run.py:63-87), resampled to size x size by area coverage (cv2 INTER_AREA, which is what
 *   predictions.py:93,154 selects); an empty / NaN box gives a black crop. */
int dif_letterbox(const uint8_t* frames_dev, int n, int h, int w, uint8_t* out_dev, int size, void* stream);
int dif_crop_resize(const uint8_t* frames_dev, int n, int h, int w, const float* boxes_ltrb_dev, float margin,
                    uint8_t* out_dev, int size, void* stream);
/* dif_area_resize: whole images [n][h][w][3] -> [n][out_h][out_w][3] by the same area coverage: the
 *   `cv2.resize(image, size, interpolation=Image.BICUBIC)` of predictions.py:93,154 (PIL's BICUBIC
 *   constant 3 is cv2.INTER_AREA) for crops that are not at the embedder's input size yet. */
int dif_area_resize(const uint8_t* images_dev, int n, int h, int w, uint8_t* out_dev, int out_h, int out_w,
                    void* stream);

/* ------------------------------------------------------------------ MTCNN cascade (BASELINE configs[4] as worded)
 * NOT IN THE REFERENCE (config.py:37, detector/run.py:124 name it in comments only; the detector it ships is YOLOv3-face,
 * above).  The three networks are dif_net archs "mtcnn_pnet" (any input >= 12 x 12; output map [H', W', 8] = logits 2 |
 * box regression 4 | 0 0), "mtcnn_rnet" (24 x 24 -> [8]) and "mtcnn_onet" (48 x 48 -> [16] = logits 2 | box 4 | landmarks
 * 10); inputs are (x - 127.5) / 128 (dif_net_set_input_transform).  The entry points below are the arithmetic between
 * them, on static shapes -- a fixed number of slots per frame and stage, an empty slot has score -1 -- so that a batch of
 * frames runs the whole cascade without a host round trip; dif_nms (above) does every suppression (scores >= 0 take part).
 * The calling convention kept from the reference is detector/run.py:120-173 (deep_insight_face.detector.mtcnn).
 * dif_mtcnn_propose: head map [n][gh][gw][ld] of P-Net at pyramid scale `scale` -> one proposal per cell:
 *   boxes_dev [n][gh*gw][4] = trunc((2 g + 1) / scale), trunc((2 g + 12) / scale) as (x1, y1, x2, y2) in frame pixels,
 *   scores_dev [n][gh*gw] = P(face) = softmax(logits)[1], or -1 below `threshold`.
 * dif_mtcnn_gather: slot (f, dst_offset + j) of the destination arrays ([n][n_dst] slots) <- source slot (f, keep[f][j])
 *   ([n][n_src] slots; src_reg rows are src_reg_ld floats apart -- the P-Net map itself serves, offset to its box
 *   channels), keep < 0 -> an empty slot; calibrate != 0: the box is regressed (x += reg * (side + 1)), squared around its
 *   centre and truncated first.  dst_reg_dev may be NULL.
 * dif_mtcnn_rescore: network outputs out_dev [slots][ld] -> scores (P(face) where the slot was alive and passes
 *   `threshold`, else -1) and reg_dev [slots][4]; plain_regression != 0 also regresses boxes_dev in place without
 *   squaring (the cascade's last step).
 * dif_crop_resize_multi: k boxes per frame, boxes [n][k][4] (left, top, right, bottom), valid_dev [n][k] (may be NULL;
 *   negative = empty slot -> black crop) -> out_dev [n*k][size][size][3]; arithmetic of dif_crop_resize. */
int dif_mtcnn_propose(const float* head_dev, int n, int gh, int gw, int ld, float scale, float threshold, float* boxes_dev,
                      float* scores_dev, void* stream);
int dif_mtcnn_gather(const int32_t* keep_dev, int n, int k, const float* src_boxes_dev, const float* src_scores_dev,
                     const float* src_reg_dev, int src_reg_ld, int n_src, float* dst_boxes_dev, float* dst_scores_dev,
                     float* dst_reg_dev, int n_dst, int dst_offset, int calibrate, void* stream);
int dif_mtcnn_rescore(const float* out_dev, int slots, int ld, float threshold, float* scores_dev, float* reg_dev,
                      float* boxes_dev, int plain_regression, void* stream);
int dif_crop_resize_multi(const uint8_t* frames_dev, int n, int h, int w, const float* boxes_ltrb_dev, const float* valid_dev,
                          int k, float margin, uint8_t* out_dev, int size, void* stream);

/* ------------------------------------------------------------------ gallery + 1:N match
 * The reference has no 1:N entry point; the semantics are utility.distance broadcast
 * over gallery rows + np.argmin (first minimum).  Housed Python-side under
 * deep_insight_face.oneshot (north_star). */
int dif_gallery_create(dif_gallery** out, int d);
int dif_gallery_destroy(dif_gallery* g);
/* copy `n` rows of `d` floats from device memory into the handle and precompute the
 * per-row norms; index_base = global index of row 0 (gallery row-sharded over ranks) */
int dif_gallery_set(dif_gallery* g, const float* rows_dev, int64_t n, int64_t index_base, void* stream);
int64_t dif_gallery_size(const dif_gallery* g);
/* incremental enrolment (no reference counterpart: the reference keeps its "database" as a Python dict of encodings,
 * predictions.py:98-126 `verify(image, identity, database, ...)`, and re-reads it per call).
 * dif_gallery_update: overwrite rows [first_row, first_row + n) with `n` rows from device memory, or append them
 *   (first_row == dif_gallery_size; first_row beyond it would leave a gap and fails); the rows must fit the
 *   capacity.  Costs O(n): only those rows' norms and filter copy are recomputed (dif_gallery_set is one pass over
 *   the whole gallery).  The match results are exactly those of a dif_gallery_set with the resulting rows.
 * dif_gallery_reserve: grow the capacity (rows are kept; never shrinks); dif_gallery_set sizes it to its `n`.
 * dif_gallery_capacity: rows the handle can hold without reallocating. */
int dif_gallery_update(dif_gallery* g, const float* rows_dev, int64_t n, int64_t first_row, void* stream);
int dif_gallery_reserve(dif_gallery* g, int64_t capacity, void* stream);
int64_t dif_gallery_capacity(const dif_gallery* g);
/* options.  "filter": what the MFMA stage of dif_match -- a candidate filter with a proven error bound; the winner
 * is chosen on the reference's own float32 arithmetic whatever it is -- runs on.  2 (default): the gallery rows and the
 * probes rounded to bf16 once (one bf16 MFMA per 16 k; bound ~0.008 |q|: a few rows per probe re-ranked; the copy
 * costs d * 2 bytes per row); 1: two-term split-bf16 copies (three MFMAs per 16 k; bound 1.6e-4 |q|; d * 4 bytes per
 * row); 0: the f32 MFMA on the rows themselves (no copy).  Results are identical.  Embedding sizes that are not
 * multiples of 64, or below 128, take the two-term form under 2 as well.  The copy is built by dif_gallery_set, or by
 * the first dif_match after the option changed; changing it frees the copy the new filter does not read (in either
 * order with dif_gallery_set); when a copy cannot be allocated the f32 filter serves and nothing fails
 * (dif_gallery_get_stat "split_copy" / "filter_terms" tell).
 * "frag": 1 (default) the one-term copy ("filter" = 2) is kept in MFMA-fragment order and the filter runs on
 * match_g1_kernel (the probes resident in LDS, the gallery streamed once from HBM straight into the MFMA operand
 * registers) where the embedding size is a multiple of 128 up to 512 and the gallery holds at least 2^18 rows (below,
 * that kernel's waves would not get a tile each); 2: whatever the row count; 0: row-major, match_b1_kernel.  Same
 * answers, same bytes per row; the copy is rewritten in the other layout by the next dif_match / dif_gallery_set
 * (also when dif_gallery_update takes the row count across the threshold).
 * "clamp_nan": 0 (default) dif_match reports NaN where the reference's distance is NaN; 1 reports the
 * distance of the similarity clamped to [-1, 1] instead (0 for a similarity rounded above 1, 1 below -1).  The
 * arg-min is the reference's either way.
 * "bd": 1 (default) the two-term filter ("filter" = 1) runs on match_bd_kernel from 65 probes up; 0 keeps it on
 * match_tile_kernel for every batch (tests, A/B).  Same answers.
 * "bd_fill": 1 (default) .. 16: blocks of match_bd_kernel per resident slot (development; no effect measured).
 * Every key the library accepts is listed here (dif_gallery_option_name; tests/test_cabi_symbols.py). */
int dif_gallery_set_option(dif_gallery* g, const char* key, int value);
/* read-outs (no reference counterpart; for capacity planning and tests).  "split_copy": 1 when the filter's bf16 copy
 * of the current rows exists; "filter_terms": bf16 terms per operand the next dif_match's filter runs on (1 or 2; 0 = the f32 rows);
 * "frag_copy": 1 when that copy is held in MFMA-fragment order (gallery option "frag");
 * "row_bytes": device bytes held per gallery row; "exact_probes": how many probes the
 * last dif_match on `stream` sent to the exact whole-gallery search (synchronises the stream). */
int dif_gallery_get_stat(dif_gallery* g, const char* key, int64_t* out, void* stream);
/* top-1 search of n probes [n][d]: idx_out_dev[n] = np.argmin over the reference's float32 distances
 * (int64 global index; first minimum; a row whose reference distance is NaN ranks first, as in
 * np.argmin: a similarity rounded beyond +-1, a zero-norm or non-finite gallery row or probe --
 * utility.py:58-62 guards none of them), dist_out_dev[n] = that row's distance (NaN where the
 * reference's is NaN unless "clamp_nan" is set), key_out_dev[n] (optional, may be NULL) = the ranking
 * key: the reference distance itself, -inf for NaN; comparable across gallery shards. */
int dif_match(dif_gallery* g, const float* probes_dev, int n, int metric, int64_t* idx_out_dev,
              float* dist_out_dev, float* key_out_dev, void* stream);
/* merge R per-shard results laid out [R][n] (after an all-gather): lowest key, then
 * lowest global index -- equals np.argmin over the concatenated gallery */
int dif_match_merge(const float* keys_dev, const int64_t* idx_dev, const float* dist_dev, int R, int n,
                    int64_t* idx_out_dev, float* dist_out_dev, void* stream);
/* the same merge over ONE all-gathered buffer of R per-rank records, each n*16 bytes:
 * { float key[n]; float dist[n]; int64 idx[n]; } -- dif_match can write its three outputs straight
 * into a rank's record (key_out = rec, dist_out = rec + n floats, idx_out = rec + 2n floats), so the
 * N>1 step needs one collective for the partial results */
int dif_match_merge_packed(const void* packed_dev, int R, int n, int64_t* idx_out_dev, float* dist_out_dev,
                           void* stream);

/* ------------------------------------------------------------------ embedding network
 * Stands behind the Keras model object of the reference:
 *   bottleneck_network(net, emd_size, input_shape)(default_model_ver)   networks/triplet.py:73-85
 *   emd_model.predict_on_batch(x[N,H,W,3]) -> [N,emd]                    predictions.py:96,156; evaluation/evals.py:56
 * arch: "resnet" (keras ResNet50V2, triplet.py:90-91), "iresnet50", "iresnet100" (also "vgg16", "mobilenet", "nn4",
 *       "yolov3", "mtcnn_pnet" / "mtcnn_rnet" / "mtcnn_onet")
 * head: "v1" (triplet.py:102-117), "v2" (GDC + L2-norm, triplet.py:119-141),
 *       "v3" (bare backbone, triplet.py:143-146); ignored for iresnet*. */
int dif_net_create(dif_net** out, const char* arch, const char* head, int emd_size, int in_h, int in_w);
int dif_net_destroy(dif_net* net);
/* parameter table, in model order.  Shapes are Keras conventions: conv kernels
 * [kh,kw,cin,cout], depthwise [kh,kw,c,1], dense [in,out], vectors [c]. */
int dif_net_param_count(const dif_net* net);
int dif_net_param_info(const dif_net* net, int i, const char** name, int* ndim, int64_t shape[4]);
/* copy one parameter from HOST memory (model.load_weights, api.py:87) */
int dif_net_set_param(dif_net* net, const char* name, const float* data_host, int64_t count);
/* read one parameter back to HOST memory (model.save_weights, networks/inceptionv3.py:86-88) */
int dif_net_get_param(const dif_net* net, const char* name, float* data_host, int64_t count);
/* input transform applied while converting to the internal NHWC4 f32 layout:
 * y[c] = x[bgr ? 2-c : c] * scale + bias[c]   (predictions.py:94,154 `* rescale`;
 * predictions.py:95 keras vgg16 preprocess_input = BGR swap + mean subtraction);
 * flags = DIF_INPUT_BGR | DIF_INPUT_HFLIP (1 keeps meaning "bgr") */
int dif_net_set_input_transform(dif_net* net, float scale, const float bias[3], int flags);
/* pack weights for the kernels, upload, and size the activation workspace */
int dif_net_finalize(dif_net* net, int max_batch);
/* execution options (no reference counterpart: Keras picks its kernels by itself).  EVERY key the library accepts is
 * listed here with its default (tests/test_cabi_symbols.py compares this list with dif_net_option_name); an unknown key
 * fails.  Unless a key says "before dif_net_finalize" it may be changed between forwards.  Most keys choose between
 * kernel families that compute the same products (the parity tests run both sides); none changes what is computed.
 *   "pipe"       1 (default): short-K convolutions may take the software-pipelined kernel (conv_pipe_kernel);
 *                0: every convolution stays on the plain implicit-GEMM kernel
 *   "bdp"        1 (default): 3x3 / stride 1 layers with several tiles per resident block may take the kernel that retires
 *                a tile's epilogue inside the next tile's K-steps (conv_bdp_kernel); 0: never; 2: wherever its
 *                restrictions allow (tests)
 *   "stem"       1 (default): 3-channel first layers run on their own kernels (stem.hip, elementwise.hip); 0: on the
 *                general implicit-GEMM kernel
 *   "patch"      1 (default): 3x3 / stride 1 / pad 1 layers keep the tile's input pixels + halo in LDS per 32-channel
 *                slice (a tap is an address offset); 0: the per-K-step gather
 *   "patch2d"    1 (default): the 8x8-tile form of that path on maps whose sides are multiples of 8; 0: off
 *   "bd"         1 (default): the patch kernels fetch the weights in MFMA-fragment order straight from L2 into registers
 *                (B-direct mainloop); 0: weights staged through LDS
 *   "t2"         1 (default): short-K 3x3 layers on 8-aligned maps run on conv_t2_kernel (128 pixels x 64 channels per
 *                block, two 8x8 sub-tiles); 0: the 64 x 64 kernels
 *   "tn"         1 (default): linear-patch 3x3 layers with whole 128-channel column blocks run on conv_tn_kernel
 *                (64 pixels x 128 channels per block, one whole tile per block); 0: conv_bdp_kernel / conv_igemm_kernel
 *   "sk2"        1 (default): layers with few tiles and a long K loop -- the reference's own call shapes, ONE image
 *                (predictions.py:152-156) or a batch of 12 (scripts/insight_face.py:112) -- run as split-K partials +
 *                a reduce / epilogue launch (conv_splitk.hpp); 0: round 4's persistent stream-K grid
 *   "mt"         1 (default): at ONE image per call (predictions.py:152-156) a layer runs in one launch on 16 x 16 tiles,
 *                operands straight from L2 into the MFMA registers, K split over the block's waves (conv_minitile.hpp);
 *                0: the split-K pair / the large-batch kernels
 *   "bf16x3"     0 (default): float32 MFMA, a bit-exact f32 fma chain -- the reference's arithmetic;
 *                1 (before dif_net_finalize): throughput mode -- every f32 operand split into bf16 terms, bf16 MFMA
 *                products accumulated in f32 (f32-level accuracy, same 1e-5 cosine gate, not bit-identical)
 *   "bf_terms"   3 (default) or 2: bf16 terms per operand in that mode (six / three MFMA products per multiply-add)
 *   "ysub"       1 (default; before dif_net_finalize): a first output read only by a 1x1 / stride 2 layer is written at
 *                even pixels only; 0: written whole
 *   "lane_split" -1 (default; before dif_net_finalize): the executor's lanes run half-chip persistent grids where the
 *                work per launch is small; 0: whole-chip grids; 1: half-chip grids always
 *   "lane_prio"  0 (default; before dif_net_finalize): all lanes of a multi-lane forward run on least-priority streams
 *                (own hardware queues, whatever else the process created); 1: lane 0 on the caller's stream
 *   "dbg"        0 (default): development aid, bit mask (256: block traces from dif_net_embed_clock; 1024: every layer on
 *                the general epilogue; other bits: ablations of the kernel under work)
 * env DIF_OPTIONS="key=value,..." applies keys to every net of the process at dif_net_finalize (A/B runs of the tools). */
int dif_net_set_option(dif_net* net, const char* key, int value);
/* the i-th key dif_net_set_option / dif_gallery_set_option accepts, NULL past the last one (so that the list above can be
 * checked against the library) */
const char* dif_net_option_name(int i);
const char* dif_gallery_option_name(int i);
int dif_net_output_dim(const dif_net* net, int64_t shape[3]); /* {emd,1,1} or {C,H,W} for v3 */
/* networks with several outputs (arch "yolov3": the three detection maps, coarse first; emd_size
 * carries the class count): out_dev then holds output 0 for all n images, then output 1, ... */
int dif_net_output_count(const dif_net* net);
int dif_net_output_info(const dif_net* net, int i, int64_t shape[3]); /* {C,H,W} of output i */
/* forward n <= max_batch images; x_dev is [n,H,W,3] (NHWC) or [n,3,H,W] (NCHW), f32 or u8;
 * out_dev is [n][emd] float32 (v3: [n,H,W,C] NHWC). */
int dif_net_embed(dif_net* net, const void* x_dev, int n, int layout, int dtype, float* out_dev, void* stream);
/* measurement aid: one forward of n images on a single lane whose convolution kernels record every block's life in
 * shader cycles and in 100 MHz ticks; ghz_out = the life-weighted mean shader clock held inside those kernels
 * (bench.py reports it beside rooflines priced at the 2.4 GHz peak) */
int dif_net_embed_clock(dif_net* net, const void* x_dev, int n, int layout, int dtype, float* out_dev, double* ghz_out,
                        void* stream);
/* algorithmic FLOPs of one forward per image (2 * MACs of every conv/dense), for rooflines */
double dif_net_flops_per_image(const dif_net* net);
/* profiling aids: number of layer ops (= kernel launches) per forward, their names and
 * algorithmic MACs per image, and a forward that brackets every launch with HIP events on
 * `stream` and returns the per-op milliseconds (ms_host[dif_net_launch_count]).  The
 * profiled forward synchronises the stream; it is a diagnostic, never the timed path. */
int dif_net_launch_count(const dif_net* net);
int dif_net_op_info(const dif_net* net, int i, const char** name, const char** kernel, double* macs_per_image);
/* compulsory HBM traffic of op i for the mixed roofline (bench.py: t_roof = max(flops / MFMA peak, bytes / HBM rate) per
 * launch): bytes of activations per image that the op must read (the input elements it uses, the shortcut) and write
 * (its one or two outputs), and the bytes of its parameters (read once per launch whatever the batch).  `kernel` of
 * dif_net_op_info is the instantiation the op's LAST launch ran (`family<tile, operand form>`), the family before. */
int dif_net_op_traffic(const dif_net* net, int i, double* act_bytes_per_image, double* param_bytes);
int dif_net_embed_profile(dif_net* net, const void* x_dev, int n, int layout, int dtype, float* out_dev,
                          void* stream, float* ms_host);

/* ------------------------------------------------------------------ ArcMargin logits
 * Not in the reference (north_star only; ArcFace, Deng et al. 2019):
 * logits = s * cos(theta + m * onehot(label)); labels_dev NULL -> s * cos(theta). */
int dif_arcmargin_create(dif_arcmargin** out, int d, int64_t n_classes, float s, float m);
int dif_arcmargin_destroy(dif_arcmargin* a);
int dif_arcmargin_set_weight(dif_arcmargin* a, const float* w_dev, void* stream); /* [C][d], rows normalised internally */
int dif_arcmargin_logits(dif_arcmargin* a, const float* emb_dev, const int64_t* labels_dev, int n,
                         float* logits_dev, void* stream); /* [n][C] */

#ifdef __cplusplus
}
#endif
#endif /* DIF_H */
