/* es_hip.h -- C ABI of libes_hip.so: the MI355X (gfx950) kernels of the EmbodiedScan
 * mv-3ddet train-step hot path.
 *
 * The reference (OpenRobotLab/EmbodiedScan) has NO FFI for this path: its native code lives in
 * un-vendored Python packages (MinkowskiEngine, mmcv._ext, pytorch3d._C, cuDNN via torch).  Each
 * entry point below therefore names the reference CALL SITE (file:line under /root/reference)
 * whose native dependency it replaces.  INTEGRATION.md shows the ctypes binding a maintainer
 * would add on the reference side.
 *
 * Conventions: plain pointers + sizes, no torch types.  All pointers are DEVICE pointers unless
 * the name ends in _host (or the comment says "host").  Row matrices are f32, row-major, with an
 * explicit leading dimension (ld*, in floats) so column slices of wider buffers can be passed.
 * `stream` is a hipStream_t.  Return value: 0 on success, a hipError_t (>0) or a negative
 * argument error otherwise; nothing is printed, nothing is allocated, inputs are only read.
 * Functions that must report a data-dependent size write it to *count_host after synchronising
 * `stream` (documented per function); every other function is purely stream-ordered.
 */
#ifndef ES_HIP_H
#define ES_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ES_MAX_SEG 32 /* max samples per batch for segmented (per-sample) operators */

/* ---- coordinate manager (replaces ME.SparseTensor / CoordinateManager) ------------------- */
/* A4: keys[i] = pack(batch, trunc(p_i / voxel_size)).  sparse_featfusion_single_stage.py:109-116 */
int es_voxel_keys(const float* points, int n, int ld, int batch, float voxel_size, int64_t* keys, void* stream);
/* A4: de-duplicate keeping the first occurrence; builds the key->row hash table (cap = power of two >= 2n).
 * scratch: 2n + n/2048 + 4 ints.  Synchronises; *count_host = number of unique keys.
 * sparse_featfusion_single_stage.py:118 (ME.SparseTensor) */
int es_unique_first(const int64_t* keys, int n, int64_t* tkeys, int* tvals, int cap, int* scratch,
                    int64_t* out_keys, int* out_src, int* count_host, void* stream);
/* Z-curve (Morton) ordering of a unique key list (row order of the root voxel set) */
size_t es_sort_scratch_bytes(int n);
int es_morton_sort(const int64_t* keys, const int* src, int n, void* scratch, size_t scratch_bytes, int64_t* out_keys,
                   int* out_src, void* stream);
/* plain ascending stable sort of non-negative keys < 2^62 with an int payload (same scratch as es_morton_sort) */
int es_sort_u64(const int64_t* keys, const int* src, int n, void* scratch, size_t scratch_bytes, int64_t* out_keys, int* out_src,
                void* stream);
/* N4: PointSample._points_random_sampling on the device (datasets/transforms/points.py:155-213: np.random.choice(range(n), k,
 * replace=False) per depth frame and over the aggregated cloud, multiview.py:139-169) with counter-based keys; see data.hip.
 * es_draw_keys: values (V*HW floats: key encodings, -1 for zero-depth pixels) for es_topk_mask_ws with one segment per view, and
 * 64-bit sort keys (view << 54 | (2^30 - 1 - key) << 24 | pixel); es_draw_keys_index: the same for elements 0..n-1 of a stream;
 * es_draw_unpack: sorted keys -> (view, pixel) lists (in_view / in_pix non-NULL: the key's low 24 bits index those lists). */
int es_draw_keys(const float* depth, int V, int HW, size_t seed, float* values, int64_t* keys, void* stream);
int es_draw_keys_index(int n, size_t seed, int stream_id, float* values, int64_t* keys, void* stream);
int es_draw_unpack(const int64_t* keys, int n, const int* in_view, const int* in_pix, int* out_view, int* out_pix, void* stream);
int es_build_table(const int64_t* keys, int n, int64_t* tkeys, int* tvals, int cap, void* stream);
/* strided output coordinates floor(c / ts) * ts.  mink_resnet.py:58-69,104-108 (stride-2 conv / pool) */
int es_stride_keys(const int64_t* in_keys, int n, int out_ts, int64_t* out_keys, void* stream);
int es_keys_to_coords(const int64_t* keys, int n, int* coords /* (n,4) b,x,y,z */, void* stream);
/* points = float(coords[:, 1:]) * voxel_size  (sparse_featfusion_single_stage.py:167-168, fcaf3d_head.py:1145-1147) */
int es_coords_to_points(const int* coords, int n, float voxel_size, float* points /* (n,3) */, void* stream);
int es_batch_offsets(const int64_t* keys, int n, int n_batch, int* offsets_dev /* n_batch+1 */, void* stream);
/* every strided set of the backbone (tensor strides ts[0 .. n_levels-1] of the root set) in one host round trip: each level is
 * the first-occurrence hash-unique of the ROOT keys floored to its stride -- the same rows, row order and key -> row tables as
 * the chain level l -> l+1 that MinkowskiEngine's strided convolutions induce (mink_resnet.py:131-143) -- with its row count and
 * per-sample offsets: res[l * (n_batch + 2)] = count, then n_batch + 1 offsets.  tkeys / tvals / out_keys: host arrays of
 * n_levels device pointers (caps[l] table slots, n rows of keys); tmp_keys n keys; scratch 2n + n/2048 + 8 ints. */
int es_strided_chain(const int64_t* root_keys, int n, int n_batch, int n_levels, const int* ts_host, int64_t* tmp_keys,
                     int* scratch, void** tkeys, void** tvals, const int* caps_host, void** out_keys, int* res_dev,
                     int* res_host, void* stream);
/* MinkowskiGenerativeConvolutionTranspose(k=2,s=2) output coordinates.  fcaf3d_head.py:937-941 */
int es_gen_children_keys(const int64_t* in_keys, int n, int half_ts, int64_t* out_keys /* 8n */, void* stream);
/* A6: nbr[j*K + k] = input row at out_j + offset_k * in_ts, or -1 (K = ksize^3, x fastest). */
int es_kernel_map(const int64_t* out_keys, int n_out, const int64_t* tkeys, const int* tvals, int cap, int ksize,
                  int in_ts, int* nbr, void* stream);
int es_inverse_map(const int* nbr, int n_out, int K, int n_in, int* inv /* (n_in,K) */, void* stream);
/* sparse `a + b` coordinate union, batch-major.  fcaf3d_head.py:1009.  scratch: 3nb + nb/2048 + 4 ints.
 * Synchronises; *count_host = rows of the union. */
int es_union_plan(const int64_t* keys_a, int na, const int64_t* tkeys_a, const int* tvals_a, int cap_a,
                  const int64_t* keys_b, int nb, const int* a_off_dev, const int* b_off_dev, int n_batch,
                  int* scratch, int* pos_a, int* pos_b, int64_t* out_keys, int* count_host, void* stream);
/* features_at_coordinates corner indices + trilinear weights.  fcaf3d_head.py:1102-1103 */
int es_interp_map(const int64_t* query_keys, int n, const int64_t* tkeys, const int* tvals, int cap, int table_ts,
                  int* idx /* (n,8) */, float* w /* (n,8) */, void* stream);
/* MinkowskiPruning row selection.  fcaf3d_head.py:1113.  scratch: n + n/2048 + 4 ints.  Synchronises to report the row count
 * unless count_host == NULL (callers that know it, e.g. top-k pruning: sum_b min(n_b, k), stay stream-ordered). */
int es_compact_mask(const int64_t* keys, int n, const int* mask, int* scratch, int64_t* out_keys, int* out_src,
                    int* count_host, void* stream);

/* ---- convolution engine (replaces MinkowskiConvolution*, and conv2d through image-grid maps) ------ */
/* Y[j] (+)= sum_k X[nbr[j,k]] . W[k] (+ bias).  W is [K][Cin][Cout]; trans_w: W is [K][Cout][Cin] (dgrad).
 * nbr == NULL means the identity map with K == 1 (plain row GEMM).  mink_resnet.py:58-62,88-120;
 * fcaf3d_head.py:907-984,1116-1136 */
int es_spconv_fwd(const float* X, int ldx, const float* W, const int* nbr, int n_out, int n_in, int K, int Cin,
                  int Cout, const float* bias, float* Y, int ldy, int trans_w, int accumulate, void* stream);
/* bf16-MFMA variant (f32 features rounded to bf16 while staged, f32 accumulate).  W_bf16 is [K][Cout][Cin]
 * (reduction index contiguous): the transposed copy for the forward pass, the natural copy for dgrad. */
int es_spconv_fwd_bf16(const void* X, int x_is_bf16, int ldx, const void* W_bf16, const int* nbr, int n_out, int n_in,
                       int K, int Cin, int Cout, const float* bias, float* Y, int ldy, int accumulate, void* stream);
/* the same with a caller-provided workspace of es_spconv_split_workspace_floats() floats: launches with too few tiles to
 * fill the chip split their tap list over several workgroups, which write partial tiles to the workspace; a SECOND launch
 * (k_sum_splits) adds them in slice order into Y (the shipped default, es_set_option key 16 = 0: bit-reproducible, no f32
 * atomics).  Key 16 = 1 folds that reduction into the launch (the last workgroup of a tile adds the slices; measured slower in
 * round 4: an agent-scope fence per workgroup).  The first 1024 floats of the workspace are the tiles' ticket counters of the
 * folded form (at most 1024 tiles): ZERO on entry, zero again on exit -- keep one zero-initialised workspace per stream. */
size_t es_spconv_split_workspace_floats(int n_out, int K, int Cin, int Cout);
int es_spconv_fwd_bf16_ws(const void* X, int x_is_bf16, int ldx, const void* W_bf16, const int* nbr, int n_out, int n_in,
                          int K, int Cin, int Cout, const float* bias, float* Y, int ldy, int accumulate, float* ws,
                          size_t ws_floats, void* stream);
/* X may also be a bf16 row matrix (x_is_bf16 = 1, ldx in bf16 elements): the shadow made by es_cast_rows_bf16; only for
 * shapes where es_spconv_bf16_is_fast() returns 1 */
int es_spconv_bf16_is_fast(int n_in, int ldx, int K, int Cin, int Cout);
/* MinkowskiGenerativeConvolutionTranspose(kernel 2, stride 2) of the head's up-blocks (fcaf3d_head.py:919-932) in one launch
 * per direction instead of eight K = 1 launches: forward Y (n, 8 Cout)[i, t Cout + c] = sum_k bf16(X[i, k]) Wt[t][c][k]; data
 * gradient dX[i] (+)= sum_t dY[i, t Cout ..] @ Wn[t]^T.  X / dY / Y / dX f32 rows, Wt / Wn the per-step bf16 copies
 * (es_cast_weight_bf16).  Returns 1 when the shape / alignment is not served (Cout % 32, 16-byte rows): the caller then issues
 * the per-tap es_spconv_fwd_bf16 launches.  Experimental in round 3 (host switch ES_GEN_FUSED, default off). */
int es_gen_transpose_fwd_bf16(const float* X, int ldx, const void* Wt_bf16, int n, int Cin, int Cout, float* Y, void* stream);
int es_gen_transpose_dgrad_bf16(const float* dY, const void* Wn_bf16, int n, int Cin, int Cout, float* dX, int ldx,
                                int accumulate, void* stream);
/* run-time tuning switches for A/B measurements: key 1 = ping-pong LDS buffers in the fast bf16 kernels (default 1),
 * key 2 = 256x256 weight-gradient tile for layers with C_in, C_out multiples of 256 and bf16-shadow operands (default 1),
 * key 3 = streaming row-GEMM kernel for K = 1 launches on the identity map (default 1),
 * keys 4-8 = weight-gradient slice targets / workspace cap / forward tap-split threshold (spconv.hip, options block),
 * key 10 = LDS-DMA staging (global_load_lds) in the fast bf16 kernels for bf16 input rows: 0 off, 1 = 32-channel chunks,
 * 2 = 64-channel chunks where C_in % 64 == 0 (default) -- bit-identical results in every mode (tests/test_gpu_dma.py);
 * 3 = EXPERIMENTAL three-buffer ring with two chunks in flight (32-channel chunks; not yet run on hardware);
 * key 11 = fewest input channels for which key 10 applies (default 768: the dense occupancy neck);
 * key 12 = fewest input channels for which the K = 1 row GEMM uses 128-column tiles (default 0);
 * key 13 = second-generation row GEMM (default 1); key 14 = weight-gradient tile with LDS-DMA staging and transposed LDS
 * reads (default 1 since round 4: bit-identical on the GPU); key 15 / 17 = one-launch norm forward / backward up to this many
 * rows; key 16 = fold the tap-split reduction into the launch (default 0, see es_spconv_fwd_bf16_ws);
 * key 18 = last-workgroup elections of the deterministic in-launch reductions (es_colsum, es_layernorm_bwd, es_contrastive_bwd,
 * es_topk_mask_ws): 0 (default) coherent (sc1) stores + drained ticket, no cache maintenance; 1 adds an agent-scope release
 * fence in every workgroup and an acquire fence in the winner (csrc/common.h es_last_block_sel; tests/test_gpu_elect.py);
 * round 6: key 9 = rows per norm-statistics chunk; 19 = fewest 128-column workgroups for the 128-column row-GEMM tile (0);
 * 20 = tap-split launches in a weight-sharing workgroup order (0: measured, no gain); 21 = K = 27, 3 -> 64 channel
 * convolutions (MinkResNet.conv1, es_spconv_fwd / es_spconv_wgrad) on the lane-per-output-channel kernels (1), 22 = their
 * weight-gradient row slices (768); 23 = 320 output columns of a K = 1 launch as one column tile (1, from 16 384 rows);
 * 24 = K = 1 launches with fewer 128-row workgroups than this on the 64 x 64 whole-stage kernels (256; 0: off; same bits);
 * 25 = C -> 4 C bf16 expansion layers from this many rows on the register-resident stream kernel (65 536; 0: off; same bits),
 * 26 = its persistent workgroups (1 024) */
int es_set_option(int key, int value);
/* Y = act((X*W) * scale[c] + shift[c] (+ res)): conv2d + frozen BatchNorm2d (+ residual) (+ ReLU) of mmdet.ResNet in one
 * launch (any shape; the tap-split of under-filled launches is not applied to fused calls).  act: 0 none, 1 ReLU,
 * 3 "gate": Y = (res > 0) ? (X*W) * scale[c] : 0 -- the data-gradient conv of layer i+1 fused with the ReLU / frozen-BN
 * backward of layer i (res = layer i's output, scale = its folded BN scale; shift may be NULL). */
int es_spconv_fwd_bf16_affine(const void* X, int ldx, const void* W_bf16, const int* nbr, int n_out, int n_in, int K,
                              int Cin, int Cout, const float* scale, const float* shift, const float* res, int ldr,
                              int act, float* Y, int ldy, void* stream);
/* Fused conv + frozen-BN (+ residual) (+ ReLU) / gated data gradient with per-operand storage kinds (round 3: the image
 * backbone keeps its ACTIVATIONS in bf16): x_half / res_half / y_half non-zero -> that row matrix is bf16 (ld in elements).
 * Replaces conv -> BatchNorm2d(eval) -> (+ identity) -> ReLU of mmdet.ResNet Bottleneck (configs/detection/...py:24-34). */
int es_spconv_fwd_bf16_io(const void* X, int x_half, int ldx, const void* W_bf16, const int* nbr, int n_out, int n_in, int K,
                          int Cin, int Cout, const float* scale, const float* shift, const void* res, int res_half, int ldr,
                          int act, void* Y, int y_half, int ldy, void* stream);
int es_cast_rows_bf16(const float* x, int ldx, int n, int C, void* h /* (n,C) bf16 */, void* stream);
/* per-step bf16 copies of an f32 [K][A][B] kernel: natural [K][A][B] and/or transposed [K][B][A] (either may be NULL) */
int es_cast_weight_bf16(const float* w, int K, int A, int B, void* natural, void* transposed, void* stream);
/* the same for every conv kernel of the model in one launch: table_dev = n_entries rows of 7 int64
 * {src f32 ptr, natural bf16 ptr, transposed bf16 ptr, K, A, B, first_tile}; one workgroup per 64x64 tile of a tap,
 * first_tile = exclusive prefix sum of K*ceil(A/64)*ceil(B/64) */
int es_cast_weights_table(const void* table_dev, int n_entries, int total_tiles, void* stream);
/* dW[k] += X[nbr[:,k]]^T . dY.  Deterministic (round 3): the rows are split into slices, every (tap, channel tile, slice)
 * workgroup owns its partial tile; with one slice the tile is added straight into dW, with several the partial tiles go to
 * the caller's workspace `ws` ([slice][K][Cin][Cout], es_spconv_wgrad_workspace_floats) and are added to dW in slice order
 * by a second launch.  No float atomics: bit-identical gradients run to run.  ws NULL: ONE slice (correct, under-filled).
 * accumulate 0: dW is OVERWRITTEN (the first gradient a weight receives in a step: no read of dW), 1: added to.
 * Replaces the backward of MinkowskiConvolution / nn.Conv2d / nn.Linear (mink_resnet.py:58-62, fcaf3d_head.py:907-984). */
int es_spconv_wgrad(const float* X, int ldx, const float* dY, int ldy, const int* nbr, int n_out, int n_in, int K,
                    int Cin, int Cout, float* dW, int accumulate, float* ws, size_t ws_floats, void* stream);
/* bf16-MFMA variant of es_spconv_wgrad (operands rounded to bf16 while staged, f32 accumulate into dW, which the caller
 * zeroes once per step).  Each (tap, row slice) workgroup compacts the valid (row, neighbour) pairs of its slice before
 * the GEMM, so absent neighbours cost one map read.  Same deterministic slice scheme as es_spconv_wgrad. */
int es_spconv_wgrad_bf16(const float* X, int ldx, const float* dY, int ldy, const int* nbr, int n_out, int n_in, int K,
                         int Cin, int Cout, float* dW, int accumulate, float* ws, size_t ws_floats, void* stream);
/* Same operator with either operand taken from a bf16 copy ("shadow", es_cast_rows_bf16) of the row matrix: x_half /
 * dy_half non-zero -> X / dY point at (n, ld) bf16 rows (ld in elements).  Results are bit-identical to
 * es_spconv_wgrad_bf16 on the f32 originals when both launches use the same tile and slices (the f32 path rounds to the
 * same bf16 values while staging); the gathered bytes halve. */
int es_spconv_wgrad_bf16_src(const void* X, int x_half, int ldx, const void* dY, int dy_half, int ldy, const int* nbr,
                             int n_out, int n_in, int K, int Cin, int Cout, float* dW, int accumulate, float* ws,
                             size_t ws_floats, void* stream);
/* floats of workspace the weight-gradient launch of this shape asks for (0: one slice).  bf16 = 0: es_spconv_wgrad;
 * 1: es_spconv_wgrad_bf16[_src] with these operand kinds, strides and pointers (their alignment selects the tile). */
size_t es_spconv_wgrad_workspace_floats(int bf16, const void* X, int x_half, int ldx, const void* dY, int dy_half, int ldy,
                                        int n_out, int n_in, int K, int Cin, int Cout);
int es_image_map(int n_img, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int* nbr, void* stream);

/* ---- row operators ----------------------------------------------------------------------------- */
/* MinkowskiBatchNorm (nseg = 1) / MinkowskiInstanceNorm (one segment per sample), train mode, fused
 * residual + activation (act: 0 none, 1 ReLU, 2 ELU).  seg_off is a HOST array of nseg+1 row offsets.
 * mink_resnet.py:64,109 ; fcaf3d_head.py:923,942,947 */
size_t es_norm_workspace_floats(int n, int C, const int* seg_off_host, int nseg);
int es_norm_fwd(const float* x, int ldx, int n, int C, const int* seg_off_host, int nseg, float eps,
                const float* weight, const float* bias, const float* res, int ldr, int act, float* running_mean,
                float* running_var, float momentum, float* mean, float* invstd, float* workspace, float* y, int ldy,
                void* y_bf16, void* stream);
/* y_bf16 / dx_bf16 (0 = none): a dense (n, C) bf16 copy of the rows the apply pass has just computed, written by the same
 * pass -- the gather source ("shadow") of the bf16 convolution that consumes them; bit-identical to es_cast_rows_bf16 of
 * y / dx.  dy is overwritten with the pre-activation gradient (== gradient of the residual input). */
int es_norm_bwd(float* dy, int ldd, const float* y, int ldy, const float* x, int ldx, int n, int C,
                const int* seg_off_host, int nseg, const float* mean, const float* invstd, const float* weight,
                int act, float* dweight, float* dbias, float* workspace, float* dx, int ldo, int accumulate,
                void* dx_bf16, void* stream);
/* MinkowskiMaxPooling(k=2,s=2) and the 2-D stem max-pool.  mink_resnet.py:66-69 */
int es_maxpool_fwd(const float* x, int ldx, const int* nbr, int n_out, int K, int C, float* y, int* arg, void* stream);
/* forward-only variant writing bf16 rows (the frozen stem pooling of the image backbone: nn.MaxPool2d(3, 2, 1)) */
int es_maxpool_fwd_h(const float* x, int ldx, const int* nbr, int n_out, int K, int C, void* y_bf16, void* stream);
int es_maxpool_bwd(const float* dy, const int* arg, int n_out, int C, float* dx, int ldo, void* stream);
/* mode 0: dst[i]=src[idx[i]]  1: dst[idx[i]]+=src[i]  2: dst[idx[i]]=src[i]  (idx NULL = identity) */
int es_row_move(float* dst, int ldd, const float* src, int lds, const int* idx, int n, int C, int mode, void* stream);
int es_axpy2d(float* dst, int ldd, const float* src, int lds, int n, int C, float alpha, int op, void* stream);
int es_interp_scores(const float* score, const int* idx, const float* w, int n, float* out, void* stream);
/* per-sample top-k keep mask (ties: lower row first).  fcaf3d_head.py:1105-1112 */
/* bias gradient of a Linear / convolution (the column sums torch autograd takes for `y = x W + b`): dst[c] (+)= sum_r g[r, c],
 * deterministic (per-chunk partial rows in `workspace`, added in chunk order by the last workgroup; ticket convention of
 * es_layernorm_bwd: one zero-initialised workspace per stream) */
size_t es_colsum_workspace_floats(int n, int C);
int es_colsum(const float* g, int ld, int n, int C, float* dst, int accumulate, float* workspace, size_t workspace_floats, void* stream);
int es_topk_mask(const float* values, const int* seg_off_host, int nseg, int k, int* mask, void* stream);
/* the same masks (bit-identical) with every segment spread over 32 workgroups: one launch per radix pass + tie count + mask
 * write (6 launches of a few microseconds instead of one 0.5 ms single-workgroup launch on the step's dependent chain).
 * workspace: es_topk_mask_workspace_ints(nseg) ints of its own (not shared with other kernels: the selection state stays in it);
 * zero-initialise it once -- tickets and histograms are left at zero by every call. */
size_t es_topk_mask_workspace_ints(int nseg);
int es_topk_mask_ws(const float* values, const int* seg_off_host, int nseg, int k, int* mask, int* workspace, size_t workspace_ints,
                    void* stream);
int es_row_max(const float* x, int ldx, int n, int C, float* out, void* stream);
/* frozen BatchNorm2d folded to scale/shift + residual + ReLU (mmdet.ResNet, norm_eval=True) */
int es_bn_fold(const float* w, const float* b, const float* rm, const float* rv, int C, float eps, float* scale,
               float* shift, void* stream);
int es_affine_act_fwd(const float* x, const float* scale, const float* shift, const float* res, size_t n, int C,
                      int act, float* y, void* stream);
int es_affine_act_bwd(const float* dy, const float* y, const float* scale, size_t n, int C, int act, float* dx,
                      int acc_x, float* dres, int acc_r, void* stream);
/* es_affine_act_bwd with the activation y stored in bf16 (only its sign is read); C % 4 == 0 */
int es_affine_act_bwd_yh(const float* dy, const void* y_bf16, const float* scale, size_t n, int C, int act, float* dx,
                         int acc_x, float* dres, int acc_r, void* stream);

/* ---- A8 projection fusion.  point_fusion.py:208-311 ----------------------------------------------- */
/* per-sample meta block layout (floats) */
#define ES_FUSE_NOPS 0    /* number of reverse-augmentation ops */
#define ES_FUSE_OPS 1     /* 8 op codes: 1 T, 2 S, 3 R, 4 HF, 5 VF (already reversed) */
#define ES_FUSE_ROTINV 9  /* 3x3 inverse of pcd_rotation, row-major */
#define ES_FUSE_ISCALE 18 /* 1 / pcd_scale_factor */
#define ES_FUSE_NTRANS 19 /* -pcd_trans (3) */
#define ES_FUSE_SFX 22
#define ES_FUSE_SFY 23
#define ES_FUSE_CROPX 24
#define ES_FUSE_CROPY 25
#define ES_FUSE_FLIP 26
#define ES_FUSE_ORIW 27
#define ES_FUSE_PADW 28
#define ES_FUSE_PADH 29
#define ES_FUSE_PROJ 32   /* V row-major 4x4 matrices intrinsic @ extrinsic */
int es_point_sample_fwd(const int* coords, int n, float voxel_size, const float* meta, int meta_stride, int V,
                        const float* feats /* (B,V,Hf,Wf,C) */, int Hf, int Wf, int C, float* out, int ldo,
                        int* pix /* (n,V) */, int* cnt /* (n) */, void* stream);
/* the same for explicit float locations (n,3) (coords[:,0] still gives the sample): the 40x40x16 prior points of the
 * occupancy detector, dense_fusion_occ.py:156-202 */
/* es_point_sample_fwd reading bf16 feature maps */
int es_point_sample_fwd_h(const int* coords, int n, float voxel_size, const float* meta, int meta_stride, int V,
                          const void* feats_bf16, int Hf, int Wf, int C, float* out, int ldo, int* pix, int* cnt, void* stream);
int es_point_sample_fwd_pts(const int* coords, const float* points, int n, const float* meta, int meta_stride, int V,
                            const float* feats, int Hf, int Wf, int C, float* out, int ldo, int* pix, int* cnt,
                            void* stream);
/* backward of the projection fusion (point_fusion.py:86-107 through autograd of grid_sample / the hit average):
 * dfeats ((n_img*Hf*Wf), C) (+)= for every feature-map pixel the sum over the voxels that sampled it of dout[i]/cnt[i], added
 * in ascending voxel order (deterministic gather: no float atomics; EVERY pixel is written, so with accumulate = 0 the
 * caller needs no memset).  n_img = samples * V; head: n_img*Hf*Wf ints, next: n*V ints of scratch. */
int es_point_sample_bwd(const int* coords, int n, int V, const float* dout, int ldo, const int* pix, const int* cnt,
                        int Hf, int Wf, int C, float* dfeats, int n_img, int* head, int* next, int accumulate, void* stream);

/* ---- A12 target assignment.  fcaf3d_head.py:1578-1664 --------------------------------------------- */
/* level_off: HOST array n_levels+1.  rot_neg: (G,9) row-major R(-euler) (ZXY) computed on the host.
 * scratch: G*N floats + (n_levels*G + G) ints + G floats. */
int es_get_targets(const float* points, int N, const int* level_off_host, int n_levels, const float* boxes,
                   const float* rot_neg, const int* labels, int G, int assign_thr, int center_thr, float* scratch,
                   float* center_t, float* bbox_t, int* cls_t, int* box_idx, int* n_pos_dev, void* stream);

/* ---- A13-A15 losses (value + gradient).  fcaf3d_head.py:1151-1294 ---------------------------------- */
int es_focal_loss(const float* logits, int ldl, const int* labels, int N, int C, float gamma, float alpha,
                  const float* avg_factor_dev, float grad_scale, float* grad, int ldg, double* partial /* 2048 */,
                  float* loss_out /* accumulated */, void* stream);
/* bbox[:, :6] = clamp(exp(scale * reg[:, :6]), 1e-3), bbox[:, 6:] = reg[:, 6:]  (fcaf3d_head.py:1135-1137) */
int es_reg_decode_fwd(const float* reg, int ldr, int n, const float* scale, float* bbox /* (n,12) */, void* stream);
/* partial: >= 512 floats of scratch (the Scale gradient is reduced in a fixed order, no atomics) */
int es_reg_decode_bwd(const float* reg, int ldr, const float* bbox, const float* dbbox, int n, const float* scale,
                      float* dreg, int ldg, float* dscale, float* partial, void* stream);
/* Per SAMPLE over its n locations of all levels (fine -> coarse, level_off_host = n_levels+1 row offsets inside the
 * per-sample arrays cls_t / points / center_t / bbox_t).  Rows with cls_t >= 0 are first compacted on the device (no
 * host-side nonzero()) into pos_ws (int[max_pos + 1], [0] = count); the loss kernel is launched over max_pos, the host's
 * upper bound on the positives (FCAF3D: pts_center_threshold * n_gt, every box keeps at most that many locations; n is
 * always valid; 0 = nothing to do).  Positives beyond max_pos would be dropped: the bound must hold.
 * ho_host[l] / dho_host[l]: this sample's first row of level l in the head output / its gradient (leading dim ldh,
 * column 0 = centerness logit); bbox_host[l] / dbbox_host[l]: decoded (.,12) boxes / their gradient.  All four are HOST
 * arrays of device pointers.  group_w: HOST array of the 4 decouple weights.
 * loss_acc[0] += sum BCE, loss_acc[1] += weighted corner loss (mean over n_pos*8); f64 accumulators: the positives are
 * compacted in slot-grab order and f64 sums of f32 terms do not depend on it after rounding back to f32. */
#define ES_MAX_LEVELS 8
int es_pos_losses(const int* cls_t, int n, const int* n_pos_dev, int max_pos, int* pos_ws, const float* points,
                  int n_levels, const int* level_off_host, const void* const* ho_host, const void* const* bbox_host,
                  void* const* dho_host, void* const* dbbox_host, int ldh, const float* center_t, const float* bbox_t,
                  const float* avg_factor_dev, float grad_scale, const float* group_w_host, double* loss_acc /* f64 sums */,
                  void* stream);

/* ---- N1 inference post-processing.  fcaf3d_head.py:1352-1399,1666-1725 + mmcv.ops.nms3d ------------------------- */
/* scores[i,c] = sigmoid(cls[i,c]) * sigmoid(centerness[i]) from the head output rows (col 0 centerness, 13.. classes) */
int es_predict_scores(const float* ho, int ldh, int n, int C, float* scores /* (n,C) */, float* max_scores, void* stream);
/* rows idx[0..m) (NULL = identity): 12-d prediction + location -> 9-DoF box (m,9) */
int es_decode_boxes(const float* points, const float* bbox, const int* idx, int m, float* out, void* stream);
/* per-class greedy NMS on the rotated BEV IoU of (x,y,z,dx,dy,dz,alpha); M <= 4096.  keep_idx (C,M), keep_cnt (C) */
int es_nms3d_multiclass(const float* boxes, const float* scores, int M, int C, float score_thr, float iou_thr,
                        int* keep_idx, int* keep_cnt, void* stream);

/* ---- optimiser.  configs/detection/mv-det3d_...py:219-223 ------------------------------------------ */
int es_grad_norm(const float* grad, size_t n, double* partial /* 2048 */, float* norm_out, void* stream);
/* grad_scale multiplies every gradient element after clipping (1 / world when grad still holds the SUM over the ranks:
 * the data-parallel mean is folded into this pass; grad_norm_dev must then be the norm of the MEAN gradient) */
int es_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float max_norm, const float* grad_norm_dev,
                  float grad_scale, void* stream);
/* The same update over a TABLE of arena pieces, writing the bf16 copies of the convolution kernels in the same pass (replaces
 * es_adamw_step + es_cast_weights_table after a step).  table_dev: n_entries rows of 9 int64 {offset into the arenas (elements),
 * K | length, A, B, natural bf16 copy | 0, transposed bf16 copy, first work item, double bits of lr_mult, of decay_mult};
 * a row with a natural-copy pointer is a kernel [K][A][B] (one work item per 64 x 64 tile of a tap), a row without a plain range
 * (one work item per 4096 elements); total_items = the work items of all rows; lr / weight_decay (doubles) are multiplied by the
 * row's lr_mult / decay_mult in double and rounded to float (mmengine paramwise_cfg; = the float(lr * lr_mult) es_adamw_step gets).  Element arithmetic identical to es_adamw_step, copies identical to
 * es_cast_weights_table. */
int es_adamw_table(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const void* table_dev, int n_entries,
                   int total_items, double lr, float beta1, float beta2, float eps, double weight_decay, int step, float max_norm,
                   const float* grad_norm_dev, float grad_scale, void* stream);
/* data parallel (parallel.py): sum of squares of one reduced gradient bucket -> 2048 doubles; the clip norm of the mean
 * gradient from all buckets' partials (scale = 1 / world) */
int es_sumsq_partial(const float* grad, size_t n, double* partial /* 2048 */, void* stream);
int es_norm_from_partials(const double* partial, int n_partials, float scale, float* norm_out, void* stream);

/* ---- A1-A3, A18 data side ---------------------------------------------------------------------------- */
int es_depth_to_points(const float* depth, int H, int W, const int* sel_view, const int* sel_pix, int n,
                       const float* mats /* (V,32) */, const float* aug /* 15 */, float* out, void* stream);
/* mmdet.ResNet stem: conv 7x7 s2 p3 (3 -> Cout in {16,32,64}) + frozen BN + ReLU, channels-last, forward only.
 * w is [49][3][Cout] (tap = ky*7+kx).  y: (n_img, Ho, Wo, Cout), Ho = (H-1)/2+1. */
int es_stem_conv_fwd(const float* x, const float* w, const float* scale, const float* shift, int n_img, int H, int W,
                     int Cout, float* y, void* stream);
/* ... followed by MaxPool2d(3, stride 2, padding 1) in the same launch (mmdet ResNet.forward: conv1 - norm1 - relu - maxpool;
 * configs/detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py:24-34, frozen_stages = 1: forward only): bf16 rows
 * (n_img, Hp, Wp, Cout), Hp = (Ho-1)/2+1, bit-identical to es_stem_conv_fwd + es_maxpool_fwd_h on the 3x3 s2 p1 image map. */
int es_stem_pool_fwd(const float* x, const float* w, const float* scale, const float* shift, int n_img, int H, int W,
                     int Cout, void* y_bf16, void* stream);
/* u8 (n_img,3,H,W) -> f32 channels-last (n_img,Hp,Wp,3): optional channel flip (bgr_to_rgb), (x-mean)/std, bottom/right
 * padding to (Hp,Wp) with pad_value.  data_preprocessor.py:249-264,286-305; data_preprocessors/utils.py:9-63 */
int es_preprocess_img(const unsigned char* img, int n_img, int H, int W, int Hp, int Wp, int flip,
                      const float* mean_host, const float* std_host, float pad_value,
                      float* out /* (n_img,Hp,Wp,3) */, void* stream);

/* ---- N4 real-data path: Resize((w,h), keep_ratio=False) of the decoded frames on the device (configs/detection/
 * mv-det3d_...py:143; mmcv.imresize -> cv2.resize INTER_LINEAR, 8-bit fixed point).  src (V,H,W,3) interleaved u8 ->
 * dst (V,3,h,w) planar u8.  xofs (w) / yofs (h): first source column / row; ialpha (w,2) / ibeta (h,2): the two 11-bit
 * weights (device arrays, built on the host once per size pair). */
int es_resize_u8(const unsigned char* src, int V, int H, int W, const int* xofs, const short* ialpha, const int* yofs,
                 const short* ibeta, int h, int w, unsigned char* dst, void* stream);

/* ---- A20 occupancy path (BASELINE config 5).  detectors/dense_fusion_occ.py:120-259, necks/imvoxel_neck.py:34-143,
 * dense_heads/imvoxel_occ_head.py:73-184, losses/occ_loss.py:7-141 ------------------------------------------------------ */
/* neighbour map of a dense (B,X,Y,Z) grid for nn.Conv3d(k, stride, pad): nbr (B*Xo*Yo*Zo, k^3), row = ((b*X+x)*Y+y)*Z+z,
 * tap = (kx*k+ky)*k+kz (torch (O,I,kD,kH,kW) order, D = x); feeds es_spconv_* (imvoxel_neck.py:84-86,121-137) */
int es_volume_map(int n_batch, int X, int Y, int Z, int Xo, int Yo, int Zo, int ksize, int stride, int pad, int* nbr,
                  void* stream);
/* nn.ConvTranspose3d(k=2,s=2) (imvoxel_neck.py:100-101) = 8 row GEMMs into rows 8*i+tap; idx (B*2X*2Y*2Z) = that row per
 * dense output voxel (gathered with es_row_move) */
int es_volume_up_index(int n_batch, int X, int Y, int Z, int* idx, void* stream);
/* keys = pack(batch, clamp(trunc((p - min) / voxel_size), 0, cmax)); rng_host = 9 HOST floats {min xyz, voxel size xyz,
 * clamp max xyz}.  dense_fusion_occ.py:227-245 */
int es_voxel_keys_range(const float* points, int n, int ld, int batch, const float* rng_host, int64_t* keys, void* stream);
/* SparseTensor.dense(): idx[i] = dense row ((b*X + x/ts)*Y + y/ts)*Z + z/ts of sparse row i (-1 outside).
 * dense_fusion_occ.py:252-255 */
int es_dense_index(const int* coords, int n, int ts, int X, int Y, int Z, int* idx, void* stream);
/* mmdet.FPN top-down: fine += nearest-upsampled coarse (channels-last, C % 4 == 0), and its gradient */
int es_upsample_nearest_add_fwd(float* fine, const float* coarse, int n_img, int Hf, int Wf, int Hc, int Wc, int C,
                                void* stream);
int es_upsample_nearest_add_bwd(const float* dfine, float* dcoarse, int n_img, int Hf, int Wf, int Hc, int Wc, int C,
                                int accumulate, void* stream);
int es_row_argmax(const float* x, int ldx, int n, int C, int* out, void* stream);
/* occ_multiscale_supervision for one sample: gt (X,Y,Z) int32 from gt_occ (n,4) int32 {x,y,z,label} at full resolution,
 * coords / ratio, last occurrence wins; mask (X*ratio, Y*ratio, Z*ratio) u8 or NULL: windows with no visible voxel
 * become 255.  winner_scratch: X*Y*Z ints.  occ_loss.py:7-36, imvoxel_occ_head.py:163-171 */
int es_occ_targets(const int* gt_occ, int n, int ratio, int X, int Y, int Z, const unsigned char* mask,
                   int* winner_scratch, int* gt, void* stream);
/* one level of ImVoxelOccHead.loss: weight * (CrossEntropy(ignore 255) + sem_scal_loss + geo_scal_loss) and its gradient
 * w.r.t. the (n,C) logits (dlogits may be NULL).  stats: 3C+2 doubles, coeff: 2C+1 floats (scratch).
 * loss_out[0..3] = CE, sem, geo, weighted total; total_acc[0] += weighted total (may be NULL).  C <= 256.
 * occ_loss.py:39-141, imvoxel_occ_head.py:156-181 */
int es_occ_loss(const float* logits, int ld, const int* gt, int n, int C, float weight, double* stats, float* coeff,
                float* dlogits, int ldg, float* loss_out, float* total_acc, void* stream);

/* ---- A19 grounding path (BASELINE config 4) + N2 device-side assignment --------------------------------------------------
 * layers/ground_transformer/decoder.py:37-297, dense_heads/grounding_head.py:20-824, detectors/sparse_featfusion_grounder.py:
 * 324-447, task_modules/assigners/hungarian_assigner.py:56-138, losses/match_cost.py:49-265 */
/* scaled dot-product attention core of nn.MultiheadAttention (mmcv MultiheadAttention, decoder.py:103-179), head_dim 32:
 * rows (b*L + i) of the projected Q/K/V matrices, head h in columns [32h, 32h+32); keys >= klen_dev[b] are masked
 * (key_padding_mask, always a prefix mask here; NULL = none).  O (B*Lq, H*32), lse (B,H,Lq).  bf16 != 0: bf16 matrix cores
 * with f32 softmax / accumulation; 0: exact-f32 matrix cores.  All leading dims multiples of 4 floats. */
int es_attn_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int B, int H, int Lq, int Lk,
                const int* klen_dev, float* O, int ldo, float* lse, int bf16, void* stream);
/* gradients of the same (recomputing the probabilities from lse); delta_scratch: B*H*Lq floats */
int es_attn_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                const float* dO, int ldd, const float* lse, int B, int H, int Lq, int Lk, const int* klen_dev,
                float* delta_scratch, float* dQ, int ldgq, float* dK, int ldgk, float* dV, int ldgv, int accumulate, int bf16,
                void* stream);
/* y = LayerNorm(x (+ res)) over the C columns of (n,C) rows (C <= 512); z = x + res is stored when z != NULL */
int es_layernorm_fwd(const float* x, const float* res, int n, int C, const float* w, const float* b, float eps, float* y,
                     float* z, float* mean, float* rstd, void* stream);
/* dw / db (+=) are reduced deterministically (round 4): partial sums per workgroup in `workspace` (>= es_layernorm_bwd_workspace_floats
 * floats; its first 4 floats hold a ticket counter that must be 0 on entry and is 0 again on exit -- keep one zero-initialised
 * workspace per stream), added in workgroup order by the last workgroup to arrive.  No float atomics. */
size_t es_layernorm_bwd_workspace_floats(int n, int C);
int es_layernorm_bwd(const float* dy, const float* z, int n, int C, const float* w, const float* mean, const float* rstd,
                     float* dz, int accumulate, float* dw, float* db, float* workspace, size_t workspace_floats, void* stream);
int es_relu_fwd(float* x, size_t n, void* stream);
int es_relu_bwd(float* dy, const float* y, size_t n, void* stream);
/* ContrastiveEmbed (grounding_head.py:62-99, log_scale='auto', bias): logits (B,L,Tout) = <v, text> / sqrt(C) + bias for
 * t < tlen[b] (and rows < vlen[b]), -inf elsewhere; rowmax (B,L) = max over tokens (either output may be NULL) */
int es_contrastive_fwd(const float* v, int B, int L, const float* text, int T, int C, const int* tlen_dev, const int* vlen_dev,
                       const float* bias_dev, float* logits, int Tout, float* rowmax, void* stream);
size_t es_contrastive_bwd_workspace_floats(int B, int T);
/* dtext: one workgroup per (sample, token) adds the rows in ascending order; dbias: per-workgroup partials in `workspace` (same
 * ticket convention as es_layernorm_bwd) summed in index order by the last workgroup.  Deterministic, no float atomics. */
int es_contrastive_bwd(const float* dlogits, int Tout, const float* v, int B, int L, const float* text, int T, int C,
                       const int* tlen_dev, float* dv, int acc_v, float* dtext /* += */, float* dbias /* += */, float* workspace,
                       size_t workspace_floats, void* stream);
/* GroundingHead._bbox_pred_to_bbox, box_coder 'baseline', 9 outputs: (pred[:3] + point, clamp(exp(pred[3:6]), 2e-2), pred[6:]) */
int es_ground_decode_fwd(const float* pred, int ldp, const float* points, int n, float* box /* (n,9) */, void* stream);
int es_ground_decode_bwd(const float* pred, int ldp, const float* dbox, int n, float* dpred, int ldg, int accumulate,
                         void* stream);
/* GroundingHead._bbox_pred_to_bbox, box_coder 'FCAF', 9 outputs (grounding_head.py:308-363, configs/grounding/..._fcaf-coder.py:64):
 * d = clamp(exp(pred[:6]), 2e-2); box = (point + R(pred[6:]) ((d1-d0)/2, (d3-d2)/2, (d5-d4)/2), (d0+d1, d2+d3, d4+d5), pred[6:]) */
int es_ground_decode_fcaf_fwd(const float* pred, int ldp, const float* points, int n, float* box /* (n,9) */, void* stream);
int es_ground_decode_fcaf_bwd(const float* pred, int ldp, const float* dbox, int n, float* dpred, int ldg, int accumulate,
                              void* stream);
/* N2: exact IoU of 9-DoF Euler (ZXY) boxes, (N,M) -- EulerInstance3DBoxes.overlaps / pytorch3d box3d_overlap */
int es_box3d_iou(const float* boxes1, int N, const float* boxes2, int M, float* iou, void* stream);
/* N2: HungarianAssigner3D for every sample of one decoder layer: costs (BinaryFocalLossCost w_cls, BBox3DL1Cost w_l1,
 * IoU3DCost w_iou) + scipy-compatible rectangular assignment.  logits (B,Q,Tout), boxes (B,Q,9), gt_boxes (sum G,9),
 * pos_map (sum G,T) u8, gt_off_dev (B+1).  cost: B*Gmax*Q doubles, work: B*(Gmax+2Q) doubles, iwork: B*(4Q+2Gmax) ints.
 * q2g (B,Q): matched ground-truth index (local to the sample) or -1.  Gmax <= Q. */
int es_ground_match(const float* logits, int Tout, const float* boxes, int B, int Q, const float* gt_boxes,
                    const unsigned char* pos_map, const int* gt_off_dev, int Gmax, const int* tlen_dev, int T, float w_cls,
                    float w_l1, float w_iou, double* cost, double* work, int* iwork, int* q2g, void* stream);
/* labels from the assignment + mmdet FocalLoss (sigmoid, py_sigmoid_focal_loss) over the un-padded tokens:
 * loss_sum[0] += sum; dlogits = d(sum / (avg_factor + eps)) * grad_scale (0 at padded tokens) */
int es_ground_focal(const float* logits, int Tout, int B, int Q, const int* q2g, const unsigned char* pos_map,
                    const int* gt_off_dev, const int* tlen_dev, int T, float alpha, float gamma, const float* avg_factor_dev,
                    float grad_scale, float* dlogits, double* loss_sum, void* stream);
/* 4-group decoupled corner-Chamfer loss on the matched (prediction, target) pairs of a batch: loss_acc[0] (f64: the sum must not depend on the arrival order of the workgroups) += weighted mean
 * over n_pairs*8 corners; dpred (B*Q,9) written at matched rows.  grounding_head.py:750-822, losses/chamfer_distance.py */
int es_box_cd_pairs(const float* pred, const int* q2g, int B, int Q, const float* gt_boxes, const int* gt_off_dev, int n_pairs,
                    float grad_scale, const float* group_w_host, float* dpred, double* loss_acc, void* stream);
/* per-sample indices of the k largest values, descending, ties by lower row; segment length <= 8192 */
int es_topk_sorted(const float* vals, int B, int L, const int* vlen_dev, int k, int* idx, void* stream);

/* ---- dense-volume convolution (occupancy neck; round 5, csrc/dconv.hip) ------------------------------------------------
 * nn.Conv3d(k, stride, pad) of IndoorImVoxelNeck on channels-last bf16 rows by ADDRESS ARITHMETIC (no neighbour map):
 * embodiedscan/models/necks/imvoxel_neck.py:78-143 (ResModule.conv1 / conv2, _make_block, _make_up_block).
 * geom_host: 7 host ints {B, X, Y, Z, ksize, stride, pad} of the operator's INPUT grid; rows of a grid are ((b*X + x)*Y + y)*Z + z.
 * mode 0 nn.Conv3d forward: Xh (B*X*Y*Z x ldx) bf16 rows, W_bf16 = the [K][Cout][Cin] copy, Y (B*Xo*Yo*Zo x ldy) f32.
 * mode 1 its data gradient: Xh = bf16 rows of the output gradient, W_bf16 = the natural [K][Cin][Cout] copy, Y = the input
 *   gradient (B*X*Y*Z x ldy); stride 1, or stride 2 with even sizes and k = 3 / pad = 1 or k = 1 / pad = 0 (the ResModule's 1x1x1
 *   down-sample, imvoxel_neck.py:126-129) by parity classes of the input voxels: no zero tap is multiplied; with k = 1 only class
 *   0 has a tap, the other seven classes write zeros (or leave an accumulated gradient alone).
 * mode 3 nn.ConvTranspose3d(k = 2, s = 2) forward (imvoxel_neck.py:98-106 _make_up_block): Xh = input rows on the coarse grid,
 *   W_bf16 = the [8][Cout][Cin] copy, Y = the (B*2X*2Y*2Z x ldy) output rows in dense order (no permutation pass);
 * mode 4 its data gradient: Xh = bf16 rows of the output gradient on the fine grid, W_bf16 = the natural [8][Cin][Cout] copy.
 * accumulate 1: Y += result.  Launches that would leave the chip under-filled split their reduction over several workgroups
 * per tile; the partial tiles go through `ws` (es_dconv_workspace_floats; 0 = not needed) and are added in slice order
 * (bit-reproducible).  -4: shape not taken (es_dconv_supported: reduction channels % 64, result channels % 128 -- 256-column tiles where they
 * divide, 128-column tiles for the neck's 128-channel out blocks; weight gradients (modes 2 / 5): both % 256). */
int es_dconv_supported(const int* geom_host, int mode, int Cin, int Cout);
size_t es_dconv_workspace_floats(const int* geom_host, int mode, int Cin, int Cout);
int es_dconv_fwd_bf16(const void* Xh, int ldx, const void* W_bf16, const int* geom_host, int mode, int Cin, int Cout, float* Y,
                      int ldy, int accumulate, float* ws, size_t ws_floats, void* stream);
/* FLAT grids: geom_host[3] = 0 is a 2-D operator (nn.Conv2d on (B, X, Y) images, rows (b*X + x)*Y + y, ksize x ksize taps, weights
 * [ksize*ksize][..][..]; the FPN's 3x3 output convolutions, mmdet/models/necks/fpn.py via
 * configs/occupancy/mv-occ_8xb1_embodiedscan-occ-80class.py:32-35); modes 0, 1 (stride 1) and 2 take them.  A bias is the caller's:
 * rows pre-filled with it and accumulate = 1 (the tile's accumulators fill the register file; an epilogue that also held the bias spilled). */
/* dW[K][Cin][Cout] (f32) = (accumulate ? dW : 0) + X^T . dY over the grid; Xh = bf16 rows of the operator's input, dYh = bf16 rows
 * of its output gradient; transposed 0: nn.Conv3d (X gathered under the tap), 1: nn.ConvTranspose3d(k = 2, s = 2) (dY gathered at
 * 2 r + p).  One workgroup per (tap, 256 x 256 tile): every element is written once, no atomics. */
int es_dconv_wgrad_bf16(const void* Xh, int ldx, const void* dYh, int ldy, const int* geom_host, int transposed, int Cin, int Cout,
                        float* dW, int accumulate, void* stream);
/* ... with a workspace: a launch of fewer than 64 tiles (the 2-D 3x3 layers: 9 tiles over 192 000 pixel rows) slices the ROWS over
 * several workgroups per tile; partial dW tensors ws[slice][K][Cin][Cout], added in slice order (bit-reproducible).
 * es_dconv_wgrad_workspace_floats: floats needed (0: one pass, ws may be NULL).  -5: workspace too small. */
size_t es_dconv_wgrad_workspace_floats(const int* geom_host, int transposed, int Cin, int Cout);
int es_dconv_wgrad_ws_bf16(const void* Xh, int ldx, const void* dYh, int ldy, const int* geom_host, int transposed, int Cin, int Cout,
                           float* dW, int accumulate, float* ws, size_t ws_floats, void* stream);
/* tuning switches of the dense engine (A/B runs): 20 row tile (0 auto, 256, 320), 21 loop order (1 chunk outer / tap inner),
 * 22 slices per tile (0 auto), 23 row slices of a weight gradient launched with a workspace (0 auto) */
int es_dconv_set_option(int key, int value);

/* ---- halo-tile sparse convolution (round 6, csrc/halo.hip) ---------------------------------------------------------------
 * K = 27 MinkowskiConvolution(kernel_size=3) forward / data gradient on the big sparse levels
 * (embodiedscan/models/backbones/mink_resnet.py:88-140, dense_heads/fcaf3d_head.py:907-1020): the distinct source rows of a
 * 256-row output tile (its halo: ~1.3 - 2.2 x the tile on Z-ordered sets) are staged in LDS once per 64-channel chunk and all 27
 * taps run out of LDS; only the weight tiles stream.
 * es_halo_plan: from a kernel map nbr[n_out][27] (es_kernel_map / es_inverse_map) -> loc[rows][27] uint16 (position of the
 *   neighbour in its tile's halo list, 0xFFFF absent), hrows[tiles][256*27] int32 (the tile's sorted distinct source rows),
 *   hcnt[tiles]; rows = es_halo_plan_rows(n_out) (whole tiles), tiles = rows / 256.  Cached with the map by the caller.
 * es_spconv_halo_bf16: Y (n_out x ldy, f32) (+)= conv of the bf16 rows Xh (n_in x ldx) with W_bf16 = the [27][Cout][Cin] copy
 *   (forward) or the natural [27][Cin][Cout] copy with the roles of Cin / Cout swapped and the plan of the inverse map (data
 *   gradient).  Cin % 64, Cout % 128, ldx % 8; -4 otherwise.  A tile whose halo exceeds the 704 resident rows runs in pages
 *   (slower, same result).  Fixed summation order: bit-reproducible.  mirror 1: tap k reads the plan's column 26 - k -- the data
 *   gradient of a stride-1 convolution on ONE coordinate set (its inverse map is the forward map with the taps mirrored:
 *   inv[i][k] == nbr[i][26 - k]) runs on the FORWARD map's plan, no second plan is built.
 * es_spconv_halo_supported: 1 when the shape is taken AND the launch fills the chip (>= `min workgroups`, es_halo_set_option 30;
 *   smaller launches belong to the tap-split gather kernels). */
size_t es_halo_plan_rows(int n_out);
int es_halo_plan(const int* nbr, int n_out, int K, void* loc, int* hrows, int* hcnt, void* stream);
int es_halo_set_option(int key, int value);
int es_spconv_halo_supported(int n_out, int n_in, int ldx, int K, int Cin, int Cout);
int es_spconv_halo_bf16(const void* Xh, int ldx, const void* W_bf16, const void* loc, const int* hrows, const int* hcnt, int n_out,
                        int n_in, int K, int Cin, int Cout, const float* bias, float* Y, int ldy, int accumulate, int mirror, void* stream);

/* ---- 3x3 image weight gradient on small channel counts (round 6, csrc/imgwgrad.hip) ---------------------------------------
 * dW[9][C][C] (f32) (+)= X^T . dY of Bottleneck.conv2 (mmdet.ResNet, stride 1 or 2, pad 1; (H, W) = the INPUT grid, even for stride 2,
 * dY on the (H / stride, W / stride) output grid) of the image backbone
 * (configs/detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py:24-34): Xh (n_img*H*W x ldx) bf16 activation rows, dY
 * (n_img*H*W x ldy) f32 rows of the output gradient (rounded to bf16 in the kernel), taps t = ty*3 + tx with the neighbour
 * (stride*y + ty - 1, stride*x + tx - 1) -- es_image_map's order.  Takes C = 32 with output width <= 128 (64 for stride 2) and C = 64 with <= 64 (the w16 backbone's
 * layer2 / layer3); es_img_wgrad9_workspace_floats returns 0 for any other shape (the caller keeps the map kernel) and -4 / -5
 * are returned for an unsupported shape / a workspace that is too small.  Partial tensors per workgroup are added in workgroup
 * order: bit-reproducible.  es_img_wgrad_set_option: 40 on / off, 41 / 42 workgroups aimed for at C = 32 / 64. */
size_t es_img_wgrad9_workspace_floats(int n_img, int H, int W, int C, int stride);
int es_img_wgrad9_bf16(const void* Xh, int ldx, const float* dY, int ldy, int n_img, int H, int W, int C, int stride, float* dW,
                       int accumulate, float* ws, size_t ws_floats, void* stream);
int es_img_wgrad_set_option(int key, int value);
/* dW[Cin][Cout] (f32) (+)= X^T . dY of a 1x1 convolution / Linear on CONTIGUOUS rows (identity map): Xh (n x ldx) bf16 activation rows, dY
 * (n x ldy) f32 gradient rows (rounded to bf16 in the kernel) -- Bottleneck.conv1 / conv3 of the image backbone.  Channel counts: multiples
 * of 64 up to 512, or 32 against >= 64; taken from 256 input channels or from 500 000 rows (es_img_wgrad_set_option 43), where it beats
 * the ring kernel.  es_rows_wgrad1_workspace_floats returns 0 for a shape it does not take (the caller keeps
 * es_spconv_wgrad_bf16_src); -4 / -5 as above.  Partial tensors per row slice, added in slice order: bit-reproducible. */
size_t es_rows_wgrad1_workspace_floats(int n, int Cin, int Cout);
int es_rows_wgrad1_bf16(const void* Xh, int ldx, const float* dY, int ldy, int n, int Cin, int Cout, float* dW, int accumulate, float* ws,
                        size_t ws_floats, void* stream);

/* ---- 3x3 image convolutions by address arithmetic (round 6, csrc/imgconv.hip) -------------------------------------------
 * Bottleneck.conv2 (3x3, pad 1) of the w16 image backbone (mmdet.ResNet, mv-det3d_...py:24-34) fused with its frozen BatchNorm2d
 * (+ ReLU) -- what es_spconv_fwd_bf16_io does through es_image_map's 9-wide map -- on C -> C channels (C = 16 / 32 / 64):
 * mode 0 forward: X (n_img*H*W x ldx) bf16 rows on the INPUT grid, W_bf16 = the [9][Cout][Cin] copy, Y = act((X*W)*scale[c] +
 *   shift[c]) on the (H/stride, W/stride) output grid, bf16 rows (y_half 1) or f32; stride 1 or 2 (even H, W); act 0 / 1 (ReLU);
 * mode 1 gated data gradient of a stride-1 layer: X = f32 rows of the output gradient, W_bf16 = the natural [9][Cin][Cout] copy,
 *   gate (n_img*H*W x ldg) = the layer input's bf16 activation rows, Y (f32) = (gate > 0) ? (X * W^T, taps mirrored) * scale[c] : 0.
 * es_img_conv3_supported: 1 when the shape is taken (C = 16: output width <= 128, stride 1; 32: <= 64; 64: <= 32); -4 otherwise.
 * es_img_conv_set_option: 50 on / off, 51 workgroups aimed for. */
int es_img_conv3_supported(int n_img, int H, int W, int C, int stride, int mode);
int es_img_conv3_bf16(const void* X, int ldx, const void* W_bf16, int n_img, int H, int W, int C, int stride, int mode,
                      const float* scale, const float* shift, const void* gate, int ldg, int act, void* Y, int y_half, int ldy,
                      void* stream);
int es_img_conv_set_option(int key, int value);

#ifdef __cplusplus
}
#endif
#endif
