/*
 * cloudct.h — C ABI of libcloudct.so, the MI355X (gfx950) implementation of the
 * Multi-Headed Cloud Transform hot path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes and a
 * HIP stream (passed as void* so that this header needs no HIP include), never
 * allocates, never synchronises the device, keeps no global state and returns
 * an int status: CT_OK (0) or a negative CT_E* code (ct_strerror() names it).
 * All tensors are fp32, contiguous, channels-first with the point index N
 * fastest — the reference's layout (layers/cloud_transform.py:140,156).  Work is
 * enqueued on the given stream (the caller's torch current stream), never on
 * the legacy default stream the reference's extensions use
 * (chamfer_extension/chamfer.cu:142, emd_linear/emd_cuda.cu:257).
 *
 * Notation: B batch, H heads, C features per head, N points, dim in {2,3},
 * W[dim] grid extents (x slowest), G = prod(W), V = 2^dim corners.
 *
 * Each function cites the reference interface it replaces (paths are into the
 * reference repository).
 */
#ifndef CLOUDCT_H
#define CLOUDCT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): ct_plane_sort / ct_plane_sort_bytes / ct_slice_bwd_ps / ct_bn_group_reduce_bwd_copy added, ct_debug_set_pw_kernel
 * removed, since version 1 — a stale library selected by CLOUDCT_LIB fails this check instead of failing at symbol binding. */
#define CT_ABI_VERSION 2

/* status codes */
#define CT_OK 0
#define CT_EINVAL (-1)    /* bad argument (null pointer, size, dim, reduce ...) */
#define CT_ELAUNCH (-2)   /* HIP reported an error at launch                     */
#define CT_EWORKSPACE (-3)/* workspace too small: see ct_*_workspace_bytes       */
#define CT_EPRECOND (-4)  /* reference precondition violated (e.g. EMD n%1024)   */

/* reduce modes of Splat */
#define CT_REDUCE_MAX0 0  /* max with a zero floor: what the reference executes
                             (torch_scatter.scatter_max into zeros,
                             layers/cloud_transform.py:164-173)                  */
#define CT_REDUCE_SUM 1   /* scatter-add: the variant BASELINE.json names        */

/* dtype of the optional pts_padding mask (B,N) */
#define CT_PAD_NONE 0
#define CT_PAD_F32 1
#define CT_PAD_I32 2      /* datasets/s3dis_closer.py:330,336 builds an int32 mask */

typedef void* ct_stream_t; /* hipStream_t */

int ct_abi_version(void);
const char* ct_strerror(int status);

/* ------------------------------------------------------------------------
 * DifferentiablePositions  (layers/cloud_transform.py:72-121,
 *                           layers/utils.py:100-186 bi/tri-linear corners)
 * keys f32[B,H*dim,N] -> local_coord f32[B,H,V,N], flat_idx i64[B,H,V,N]
 * ---------------------------------------------------------------------- */
int ct_positions_fwd(const float* keys, float* local_coord, int64_t* flat_idx,
                     int B, int H, int N, int dim, const int* W, ct_stream_t s);
/* g_keys[B,H*dim,N] = d(local_coord)/d(keys)^T g_local_coord, with the
 * GradientBalancing rule (cloud_transform.py:12-26: no (W-1)/2 factor) and the
 * clamp mask (:91).  Overwrites g_keys. */
int ct_positions_bwd(const float* keys, const float* g_local_coord, float* g_keys,
                     int B, int H, int N, int dim, const int* W, ct_stream_t s);

/* ------------------------------------------------------------------------
 * Fused hot path: corners are recomputed from `keys` inside the kernels, so
 * local_coord / flat_idx never touch HBM.  Used by the MultiHead* blocks.
 *
 * Splat.forward  (layers/cloud_transform.py:131-180)
 *   grid f32[B,H*C,G] (fully overwritten)
 * Splat backward (torch_scatter.scatter_max backward: the single arg-max
 *   contribution of a cell receives the cotangent; on EXACT ties exactly one of
 *   the tied contributions receives it — as in torch_scatter, whose CPU and CUDA
 *   paths differ in WHICH one.  Here: a single chance tie in a plane is awarded
 *   to the lowest point index; with more ties (duplicated points) the winner is
 *   unspecified.  HISTORY.md §2 has the rule per kernel family.)
 *   DEVIATION, deliberate: a candidate whose product is exactly +-0 (a point exactly on a cell boundary: one corner weight is 0;
 *   or a feature that is exactly 0, e.g. under a padding mask) aimed at a cell whose maximum stays at the zero floor receives
 *   NOTHING here.  torch_scatter.scatter_max records it as the arg-max (0 == 0) and routes the cell's cotangent to it.  Values
 *   and every cotangent not itself multiplied by that zero agree; what differs is g_keys of a point exactly on a boundary
 *   (through d(weight)/d(key): a few elements per million points) and g_feat of an exactly-zero feature.  Pinned by
 *   tests/test_tie_rule_gpu.py; INTEGRATION.md "Known deviations".
 *   needs `grid` = the forward output, g_grid f32[B,H*C,G];
 *   writes g_feat f32[B,H*C,N] and g_keys f32[B,H*dim,N] (both overwritten).
 * workspace: scratch of ct_splat_bwd_workspace_bytes(...) bytes (may be 0).
 * ---------------------------------------------------------------------- */
int ct_splat_fwd(const float* keys, const float* feat, const void* pad, int pad_dtype,
                 float* grid, int B, int H, int C, int N, int dim, const int* W,
                 int reduce, ct_stream_t s);
size_t ct_splat_bwd_workspace_bytes(int B, int H, int C, int N, int dim, const int* W, int reduce);
int ct_splat_bwd(const float* keys, const float* feat, const void* pad, int pad_dtype,
                 const float* grid, const float* g_grid, float* g_feat, float* g_keys,
                 void* workspace, size_t workspace_bytes,
                 int B, int H, int C, int N, int dim, const int* W, int reduce, ct_stream_t s);

/* ct_splat_bwd with options.  CT_BWD_ACCUMULATE_KEYS: g_keys += result instead of g_keys = result — the
 * keys of a block feed both Splat and Slice (layers/multihead_ct.py:99-107), so their two key cotangents are
 * summed by autograd; accumulating in the second backward saves that extra pass over (B,H*dim,N).
 * workspace: ct_splat_bwd_ex_workspace_bytes(..., flags) bytes. */
#define CT_BWD_ACCUMULATE_KEYS 1
size_t ct_splat_bwd_ex_workspace_bytes(int B, int H, int C, int N, int dim, const int* W, int reduce, int flags);
int ct_splat_bwd_ex(const float* keys, const float* feat, const void* pad, int pad_dtype,
                    const float* grid, const float* g_grid, float* g_feat, float* g_keys,
                    void* workspace, size_t workspace_bytes,
                    int B, int H, int C, int N, int dim, const int* W, int reduce, int flags, ct_stream_t s);

/* Test hooks (process-wide host state, never read by a kernel): flags select kernel families so that the
 * parity tests can compare them on identical inputs (atomic: set on one thread, read by entry points on any); the tag string
 * names the kernels the last Splat / Slice entry point launched — whichever thread called it, e.g. autograd's — and is
 * returned as a copy owned by the calling thread (writers and readers serialise on a mutex). */
#define CT_DEBUG_NO_HOT 1      /* keep the hot-shape kernels (csrc/ct_raster_hot.h) off */
#define CT_DEBUG_FORCE_HOT 2   /* use them for every eligible layout, however few (b,h) planes there are */
#define CT_DEBUG_NO_BAND 4     /* keep the banded four-channel kernels (csrc/ct_raster_band.h) off */
#define CT_DEBUG_FORCE_BAND 8  /* try them first, on every layout they can take (small grids in the tests) */
#define CT_DEBUG_NO_SORTED 16  /* keep the sorted-plane kernels (csrc/ct_raster_sorted.h) off */
#define CT_DEBUG_FORCE_SORTED 32 /* use them on every layout they can take, however few (b,h) planes there are */
#define CT_DEBUG_FORCE_SORTED_SEG 64 /* 2D: the sorted-SEGMENT kernel (csrc/ct_raster_sorted3d.h, DIM = 2) wherever it is legal; 3D grids
                                        take it under CT_DEBUG_FORCE_SORTED already */
void ct_debug_set_flags(unsigned flags);
const char* ct_debug_last_launch(void);
/* point segments of the hot Splat(max) backward (ct_splat_bwd_tk): 0 automatic, 1 never, n > 1: n wherever legal */
void ct_debug_set_nseg(int nseg);

/* Slice.forward  (layers/cloud_transform.py:190-227): out f32[B,H*C,N].
 * Slice backward (autograd of torch.gather = scatter-add, :216-221):
 *   g_grid f32[B,H*C,G] and g_keys f32[B,H*dim,N], both overwritten. */
int ct_slice_fwd(const float* keys, const float* grid, const void* pad, int pad_dtype,
                 float* out, int B, int H, int C, int N, int dim, const int* W, ct_stream_t s);
int ct_slice_bwd(const float* keys, const float* grid, const void* pad, int pad_dtype,
                 const float* g_out, float* g_grid, float* g_keys,
                 int B, int H, int C, int N, int dim, const int* W, ct_stream_t s);
/* ct_slice_bwd with scratch: with few (b,h) planes (the zoo's H = 16 blocks) the channel chunks of a plane are dealt to
 * several workgroups, whose partial g_keys go through `workspace` (ct_slice_bwd_workspace_bytes(...) bytes, may be 0).
 * Same results as ct_slice_bwd, which has to keep one workgroup per plane there. */
size_t ct_slice_bwd_workspace_bytes(int B, int H, int C, int N, int dim, const int* W);
int ct_slice_bwd_ws(const float* keys, const float* grid, const void* pad, int pad_dtype,
                    const float* g_out, float* g_grid, float* g_keys, void* workspace, size_t workspace_bytes,
                    int B, int H, int C, int N, int dim, const int* W, ct_stream_t s);
/* Arrival tickets: ct_slice_bwd_ws / ct_splat_bwd_ex with the sums over a plane's workgroups done INSIDE the kernel.
 * Where a (b,h) plane is shared by several workgroups (few planes: channel-chunk groups, point segments) each leaves a
 * partial g_keys / g_grid in `workspace`; without tickets a second small launch adds them.  With `tickets` — a device
 * buffer of CT_TICKETS_BYTES that is ZERO on entry (ct_tickets_init once after allocation) — every workgroup takes a
 * ticket of its plane when its partials are out and the holder of the last ticket adds them, in the same fixed order
 * (same bits as the two-launch form), and resets the ticket: the buffer is zero again when the kernel ends.  One
 * launch per pass instead of two or three.  The buffer carries state between launches: launches that share it must be
 * ordered (one buffer per stream), and it must be re-initialised after a launch that faulted.  tickets == NULL: exactly
 * ct_slice_bwd_ws / ct_splat_bwd_ex.  Same workspace sizes, same results. */
#define CT_TICKETS_BYTES 65536
int ct_tickets_init(void* tickets, ct_stream_t s);
int ct_slice_bwd_tk(const float* keys, const float* grid, const void* pad, int pad_dtype,
                    const float* g_out, float* g_grid, float* g_keys, void* workspace, size_t workspace_bytes,
                    void* tickets, int B, int H, int C, int N, int dim, const int* W, ct_stream_t s);
/* Sorted planes (csrc/ct_raster_sorted.h; no counterpart in the reference, whose Splat / Slice re-derive everything from
 * local_coordinate / flattened_index per call, layers/cloud_transform.py:72-124).  One diff_poss(lattice) feeds the Splat and
 * the Slice of a block (layers/multihead_ct.py:99-107), so the four raster passes of a step see the SAME keys: ct_plane_sort
 * counting-sorts every (b, h) plane's points by base cell once (the record is a function of the keys alone) and leaves
 * per plane the sorted positions, the fractional corner weights in sorted order and the plane's work items in `sorted`
 * (ct_plane_sort_bytes; 0: this layout has no sorted form — dim 3, N > 4096 ...).  The *_ps entry points take the record of
 * THEIR keys (NULL: they behave as the *_tk entry points and sort inside where they use the sorted form); a record made from
 * other keys gives wrong results, not an error. */
size_t ct_plane_sort_bytes(int B, int H, int N, int dim, const int* W);
int ct_plane_sort(const float* keys, void* sorted, size_t sorted_bytes, int B, int H, int N, int dim, const int* W, ct_stream_t s);
int ct_slice_bwd_ps(const float* keys, const float* grid, const void* pad, int pad_dtype,
                    const float* g_out, float* g_grid, float* g_keys, void* workspace, size_t workspace_bytes,
                    void* tickets, const void* sorted, int B, int H, int C, int N, int dim, const int* W, ct_stream_t s);
/* ct_splat_bwd_tk: g_keys = g_keys_add + (key cotangent of this Splat); g_keys_add NULL: g_keys = the cotangent;
 * g_keys_add == g_keys: in place, as CT_BWD_ACCUMULATE_KEYS.  With tickets AND a g_keys_add that is not g_keys (or NULL) a
 * plane's POINTS may be dealt to several workgroups (point segments: no partial sums at all, every workgroup walks all
 * channel chunks for its points); the plane's exact-tie test then runs across them through the tickets (a 64-bit word per
 * plane in the buffer's second half: the buffer must be 8-byte aligned for this form), and a plane with more than one tie
 * is redone by its last workgroup from g_keys_add — which is why it must not alias the output.
 * workspace: ct_splat_bwd_ex_workspace_bytes(..., CT_BWD_ACCUMULATE_KEYS if g_keys_add else 0). */
/* > 1: with tickets and a g_keys_add that is not g_keys, ct_splat_bwd_tk(reduce = max) deals the points of a plane of this
 * shape to that many workgroups — worth a second key-cotangent tensor; 1: it would not (adding in place is as good). */
int ct_splat_bwd_tk_segments(int B, int H, int C, int N, int dim, const int* W);
int ct_splat_bwd_tk(const float* keys, const float* feat, const void* pad, int pad_dtype,
                    const float* grid, const float* g_grid, float* g_feat, const float* g_keys_add, float* g_keys,
                    void* workspace, size_t workspace_bytes, void* tickets,
                    int B, int H, int C, int N, int dim, const int* W, int reduce, ct_stream_t s);
/* The two halves of ct_slice_bwd, for callers that need only one cotangent
 * (autograd's needs_input_grad) and for per-kernel timing: each is one launch. */
int ct_slice_bwd_grid(const float* keys, const void* pad, int pad_dtype, const float* g_out,
                      float* g_grid, int B, int H, int C, int N, int dim, const int* W, ct_stream_t s);
int ct_slice_bwd_keys(const float* keys, const float* grid, const void* pad, int pad_dtype,
                      const float* g_out, float* g_keys,
                      int B, int H, int C, int N, int dim, const int* W, ct_stream_t s);

/* ------------------------------------------------------------------------
 * The same four passes with EXPLICIT local_coord / flat_idx tensors — the
 * signature-compatible Splat / Slice nn.Modules call these when handed
 * arbitrary (local_coordinate, flattened_index) inputs.  g_local_coord
 * f32[B,H,V,N] is overwritten.  Every flat_idx must be in [0, G).
 * ---------------------------------------------------------------------- */
int ct_splat_lc_fwd(const float* local_coord, const int64_t* flat_idx, const float* feat,
                    const void* pad, int pad_dtype, float* grid,
                    int B, int H, int C, int N, int dim, const int* W, int reduce, ct_stream_t s);
int ct_splat_lc_bwd(const float* local_coord, const int64_t* flat_idx, const float* feat,
                    const void* pad, int pad_dtype, const float* grid, const float* g_grid,
                    float* g_feat, float* g_local_coord, void* workspace, size_t workspace_bytes,
                    int B, int H, int C, int N, int dim, const int* W, int reduce, ct_stream_t s);
int ct_slice_lc_fwd(const float* local_coord, const int64_t* flat_idx, const float* grid,
                    const void* pad, int pad_dtype, float* out,
                    int B, int H, int C, int N, int dim, const int* W, ct_stream_t s);
int ct_slice_lc_bwd(const float* local_coord, const int64_t* flat_idx, const float* grid,
                    const void* pad, int pad_dtype, const float* g_out,
                    float* g_grid, float* g_local_coord,
                    int B, int H, int C, int N, int dim, const int* W, ct_stream_t s);

/* Occupancy statistic of a rasterised grid (layers/multihead_ct.py:104-105):
 * count[0] = number of elements with |z| > 1e-9 (the caller divides by B*C*H).
 * count is a device int64 and is overwritten. */
int ct_grid_occupancy(const float* grid, int64_t n_elements, int64_t* count, ct_stream_t s);
/* The same statistic as the blocks report it, in ONE launch: out[0] = float(count) * inv_denominator (what torch computes for
 * `count.float() / (B*C*H)`, layers/multihead_ct.py:104-105), a device float that is overwritten.  `workspace`: CT_OCC_WORKSPACE_BYTES
 * of device memory, 8-byte aligned, zeroed ONCE by the caller (per-workgroup counts and an arrival ticket that the kernel hands
 * back as zero); launches that share a workspace must be ordered (one per stream).  n_elements <= 2^31 - 1. */
#define CT_OCC_WORKSPACE_BYTES 4096
int ct_grid_occupancy_ratio(const float* grid, int64_t n_elements, float inv_denominator, float* out, void* workspace, ct_stream_t s);

/* ------------------------------------------------------------------------
 * Lattice of an MHCT block: per-head rigid transform of (xyz + key residual) followed by tanh
 * (VolTransformer / PlaneTransformer: layers/utils.py:9-61; layers/multihead_ct.py:93-97,
 *  layers/multihead_ct_adain.py:112-116 with its learnable scalar on the residual):
 *   p = xyz + kscale*residual + shift_h;  keys_n = (sum_c p_c R_h[c][n]) * scales_h[n], n < dim;
 *   lattice = tanh(keys)
 * xyz f32[B,3,N], residual f32[B,H*3,N], R f32[H,3,3] (= so3_exponential_map(log_R), computed by the
 * caller), shift f32[H,3], scales f32[H,dim] | NULL, kscale device f32[1] | NULL (= 1).
 * Forward writes keys and lattice f32[B,H*dim,N].  Backward takes the cotangents of the lattice and /
 * or of the pre-tanh keys (either may be NULL, not both) and overwrites g_xyz, g_residual, g_R, g_shift (and g_scales / g_kscale iff scales / kscale given).
 * With a workspace of ct_lattice_bwd_workspace_bytes the per-head parameter cotangents are summed in a fixed order (reproducible);
 * with workspace NULL they are accumulated with one float atomic per workgroup and parameter.
 * ---------------------------------------------------------------------- */
size_t ct_lattice_fwd_workspace_bytes(int B, int H, int N);
int ct_lattice_fwd(const float* xyz, const float* residual, const float* R, const float* shift,
                   const float* scales, const float* kscale, float* keys, float* lattice,
                   float* key_stats /* NULL | f32[2]: mean and unbiased variance of the keys (the blocks' lattice statistics,
                                       layers/multihead_ct.py:109-112); needs the workspace */,
                   void* workspace, size_t workspace_bytes, int B, int H, int N, int dim, ct_stream_t s);
size_t ct_lattice_bwd_workspace_bytes(int B, int H, int N);
int ct_lattice_bwd(const float* xyz, const float* residual, const float* R, const float* shift,
                   const float* scales, const float* kscale, const float* lattice, const float* g_lattice,
                   const float* g_keys, float* g_xyz, float* g_residual, float* g_R, float* g_shift, float* g_scales,
                   float* g_kscale, void* workspace, size_t workspace_bytes, int B, int H, int N, int dim, ct_stream_t s);

/* The transformer of a block in its fewest launches (layers/utils.py:25-34,53-61 with the so3 map of :29,56 inside):
 *   ct_lattice_so3_fwd: R = so3_exponential_map(log_R, so3_eps) computed per head inside the lattice launch (R f32[H,3,3] is an
 *     OUTPUT, kept for the backward) and the key statistics finished by the launch's last workgroup — ONE launch where
 *     ct_so3_exp_fwd + ct_lattice_fwd are three.  `ticket`: one zero 32-bit word of device memory that the launch leaves zero
 *     (needed with key_stats; launches that share it must be ordered: e.g. the last word of a ct_tickets_init buffer).
 *   ct_lattice_so3_bwd: ct_lattice_bwd whose tail launch also turns every head's g_R into g_log_R f32[H,3] (the workgroup
 *     that sums a head's g_R finishes with the so3 map's backward) — two launches where ct_lattice_bwd + ct_so3_exp_bwd are
 *     three; g_R f32[H,3,3] is scratch the caller provides; the workspace is required. */
int ct_lattice_so3_fwd(const float* xyz, const float* residual, const float* log_R, float so3_eps, const float* shift,
                       const float* scales, const float* kscale, float* R, float* keys, float* lattice, float* key_stats,
                       void* workspace, size_t workspace_bytes, void* ticket, int B, int H, int N, int dim, ct_stream_t s);
int ct_lattice_so3_bwd(const float* xyz, const float* residual, const float* log_R, float so3_eps, const float* R,
                       const float* shift, const float* scales, const float* kscale, const float* lattice,
                       const float* g_lattice, const float* g_keys, float* g_xyz, float* g_residual, float* g_log_R, float* g_R,
                       float* g_shift, float* g_scales, float* g_kscale, void* workspace, size_t workspace_bytes, int B, int H,
                       int N, int dim, ct_stream_t s);

/* so3 exponential map of the per-head rotation parameters — the third-party call of the transformers
 * (pytorch3d.transforms.so3.so3_exponential_map, layers/utils.py:6,29,56; eps = 1e-4 there):
 *   theta = sqrt(max(|v|^2, eps));  R = I + (sin theta / theta) K + ((1 - cos theta) / theta^2) K^2,  K = hat(v)
 * log_R f32[H,3] -> R f32[H,3,3];  backward: g_R f32[H,3,3] -> g_log_R f32[H,3] (overwritten). */
int ct_so3_exp_fwd(const float* log_R, float* R, int H, float eps, ct_stream_t s);
int ct_so3_exp_bwd(const float* log_R, const float* g_R, float* g_log_R, int H, float eps, ct_stream_t s);

/* ------------------------------------------------------------------------
 * Training-mode BatchNorm1d fused with the ReLU that follows it in the blocks' `after` stacks
 * (nn.Sequential(nn.BatchNorm1d(C), nn.ReLU(inplace=True)): layers/multihead_ct.py:67-68,149-153), one launch
 * forward and one backward:
 *   y[b,c,n] = relu?( (x - mean_c) * rsqrt(var_c + eps) * weight[c] + bias[c] ) [+ residual[b,c,n]],
 *   mean / biased var over (b, n); `residual` (nullable) is the union block's skip connection
 *   (layers/multihead_ct.py:198: `residual + self.after(...)`), its cotangent is gy itself;
 *   num_batches_tracked (nullable, int64[1]) is incremented, as nn.BatchNorm1d.forward does
 * x, y, gy, gx f32[B,C,N], rows of N contiguous floats, 16-byte aligned; each has a batch stride in floats (0 = C*N,
 * contiguous; a multiple of 4 >= C*N for a channel slice of a wider tensor: key_bn / values_bn act on the two halves
 * of the keys_values_pred output, layers/multihead_ct.py:89-91, and their input cotangents are written straight into
 * the two halves of that tensor's cotangent).  weight, bias f32[C]; save_mean, save_rstd f32[C] are written by the
 * forward and read by the backward.  running_mean / running_var f32[C] (both or neither) are updated in place as
 * torch does: r = (1 - momentum) * r + momentum * (mean | unbiased var).  Backward overwrites gx, g_weight, g_bias;
 * with relu != 0 the mask is recomputed from x.  Shapes: 2 <= B*N < 2^31 (ct_bn_relu_supported(B, C, N) != 0); a channel
 * with N % 4 == 0 and B*N <= 32768 is held in registers between the statistics and the normalisation (one read, one
 * write), longer channels and rows that are not float4-addressable are re-read per pass.
 * ---------------------------------------------------------------------- */
int ct_bn_relu_supported(int B, int C, int N);
int ct_bn_relu_fwd(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                   float* running_mean, float* running_var, long long* num_batches_tracked, const float* residual,
                   long long residual_batch_stride, float* y, long long y_batch_stride, float* save_mean,
                   float* save_rstd, int B, int C, int N, float eps, float momentum, int relu, ct_stream_t s);
int ct_bn_relu_bwd(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                   const float* save_mean, const float* save_rstd, const float* gy, long long gy_batch_stride,
                   float* gx, long long gx_batch_stride, float* g_weight, float* g_bias, int B, int C, int N, int relu,
                   ct_stream_t s);
/* Several norms over the same (B, N) in ONE launch (n <= 8): the key / values norms of a block's heads on channel ranges of the
 * stacked projection, the heads' `after` norms on ranges of the concatenation (layers/multihead_ct.py:89-91,67-68) — one
 * workgroup per channel of every norm.  Fields as the arguments of ct_bn_relu_fwd_amax / ct_bn_relu_bwd_amax. */
typedef struct {
  const float* x; long long x_batch_stride; const float* weight; const float* bias; float* running_mean; float* running_var;
  long long* num_batches_tracked; const float* residual; long long residual_batch_stride; float* y; long long y_batch_stride;
  float* save_mean; float* save_rstd; float* amax_out; int C; float eps; float momentum; int relu;
} ct_bn_fwd_item;
typedef struct {
  const float* x; long long x_batch_stride; const float* weight; const float* bias; const float* save_mean; const float* save_rstd;
  const float* gy; long long gy_batch_stride; float* gx; long long gx_batch_stride; float* g_weight; float* g_bias; float* amax_out;
  int C; int relu;
} ct_bn_bwd_item;
int ct_bn_group_fwd(const ct_bn_fwd_item* items, int n, int B, int N, ct_stream_t s);
int ct_bn_group_bwd(const ct_bn_bwd_item* items, int n, int B, int N, ct_stream_t s);
/* The same group around ONE statistics exchange between ranks (SyncBatchNorm, train_segmentation.py:128): every phase of
 * all n norms in one launch — 2 launches + 1 collective per group and direction instead of 2 n + 1.  Ct = sum of the items' C,
 * item i's channels start at c0_i:
 *   ct_bn_group_stats_fwd : local f32[2 Ct + 1] = [mean | sum (x - mean)^2 | count] of THIS rank's batch (only x, C and
 *                           x_batch_stride of the items are read);
 *   ct_bn_group_apply_fwd : gathered f32[world][2 Ct + 1] (the ranks' `local` blocks, one all_gather) -> y, save_mean,
 *                           save_rstd, running statistics, amax_out of every item; count_total f32[1] = values per channel
 *                           of the whole job;
 *   ct_bn_group_reduce_bwd: sums f32[2 Ct] = [sum g' | sum g' xhat] of this rank (g_weight / g_bias / gx of the items unused);
 *   ct_bn_group_apply_bwd : sums all-reduced over the ranks + count -> gx, amax_out of every item. */
int ct_bn_group_stats_fwd(const ct_bn_fwd_item* items, int n, int B, int N, float* local, ct_stream_t s);
int ct_bn_group_apply_fwd(const ct_bn_fwd_item* items, int n, int B, int N, const float* gathered, int world,
                          float* count_total, ct_stream_t s);
int ct_bn_group_reduce_bwd(const ct_bn_bwd_item* items, int n, int B, int N, float* sums, ct_stream_t s);
/* the same, leaving a second copy of the sums in sums_copy f32[2 Ct] (same layout): the all_reduce runs in place on `sums`, the copy
 * stays this rank's g_bias / g_weight (torch: grad of the SyncBatchNorm's affine parameters is local, DDP averages it) — one launch
 * less per norm group than cloning the buffer */
int ct_bn_group_reduce_bwd_copy(const ct_bn_bwd_item* items, int n, int B, int N, float* sums, float* sums_copy, ct_stream_t s);
int ct_bn_group_apply_bwd(const ct_bn_bwd_item* items, int n, int B, int N, const float* sums, const float* count,
                          ct_stream_t s);
/* The same two with amax_out f32[C] (nullable): max |y| (after ReLU and skip) / max |g_x| per channel, a by-product of the
 * pass — the operand maxima ct_pw_gemm needs for the pointwise convolution that reads y / g_x next (n_amax = C). */
int ct_bn_relu_fwd_amax(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                        float* running_mean, float* running_var, long long* num_batches_tracked,
                        const float* residual, long long residual_batch_stride, float* y, long long y_batch_stride,
                        float* save_mean, float* save_rstd, float* amax_out, int B, int C, int N, float eps,
                        float momentum, int relu, ct_stream_t s);
int ct_bn_relu_bwd_amax(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                        const float* save_mean, const float* save_rstd, const float* gy, long long gy_batch_stride,
                        float* gx, long long gx_batch_stride, float* g_weight, float* g_bias, float* amax_out,
                        int B, int C, int N, int relu, ct_stream_t s);

/* The same norm split around an exchange of statistics between ranks — nn.SyncBatchNorm under data parallelism
 * (train_segmentation.py:128-130 converts every BatchNorm of the model).  Forward: ct_bn_stats_fwd on every norm of a
 * block into ONE buffer (mean[C], sum of squared deviations[C] per norm, this rank's count), one all_gather of that
 * buffer, then ct_bn_apply_fwd per norm, which merges the ranks' statistics itself (parallel-variance rule) and
 * normalises (+ ReLU, + skip).  Backward: ct_bn_reduce_bwd per norm into one buffer (these are also this rank's weight /
 * bias gradients), one all_reduce(sum), ct_bn_apply_bwd.  g_mean / g_m2 point at rank 0's entries of the norm's channels
 * inside the gathered buffer, g_count at rank 0's count; rank r's are g_stride floats further. */
int ct_bn_stats_fwd(const float* x, long long x_batch_stride, float* mean, float* m2, float* count /* nullable */,
                    int B, int C, int N, ct_stream_t s);
int ct_bn_apply_fwd(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                    const float* g_mean, const float* g_m2, const float* g_count, int world, long long g_stride,
                    float* running_mean, float* running_var, long long* num_batches_tracked, const float* residual,
                    long long residual_batch_stride, float* y, long long y_batch_stride, float* save_mean,
                    float* save_rstd, float* count_total /* nullable */, int B, int C, int N, float eps, float momentum,
                    int relu, ct_stream_t s);
int ct_bn_reduce_bwd(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                     const float* mean, const float* rstd, const float* gy, long long gy_batch_stride,
                     float* sum_g, float* sum_gxhat, int B, int C, int N, int relu, ct_stream_t s);
int ct_bn_apply_bwd(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                    const float* mean, const float* rstd, const float* gy, long long gy_batch_stride,
                    const float* sum_g, const float* sum_gxhat, const float* count, float* gx, long long gx_batch_stride,
                    int B, int C, int N, int relu, ct_stream_t s);
/* ct_bn_apply_fwd / _bwd with amax_out f32[C] (nullable), as ct_bn_relu_fwd_amax / _bwd_amax. */
int ct_bn_apply_fwd_amax(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                         const float* g_mean, const float* g_m2, const float* g_count, int world, long long g_stride,
                         float* running_mean, float* running_var, long long* num_batches_tracked, const float* residual,
                         long long residual_batch_stride, float* y, long long y_batch_stride, float* save_mean,
                         float* save_rstd, float* count_total, float* amax_out, int B, int C, int N, float eps,
                         float momentum, int relu, ct_stream_t s);
int ct_bn_apply_bwd_amax(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                         const float* mean, const float* rstd, const float* gy, long long gy_batch_stride,
                         const float* sum_g, const float* sum_gxhat, const float* count, float* gx,
                         long long gx_batch_stride, float* amax_out, int B, int C, int N, int relu, ct_stream_t s);

/* ------------------------------------------------------------------------
 * Adaptive instance normalisation of the AdaIN blocks (AdaIn1dUpd: layers/utils.py:82-97 =
 * InstanceNorm1d(affine=False, eps) -> * (gamma + 1) -> + beta; followed by ReLU in `after`,
 * layers/multihead_ct_adain.py:64-66,183-187), one launch forward and one backward:
 *   y[b,c,n] = relu?( (x - mean_n x) * rsqrt(var_n x + eps) * (gamma[b,c] + 1) + beta[b,c] )
 * x, y, gy, gx f32[B,C,N]; gamma_beta f32[B,2,C] (the Linear(style) output: [:,0] scale, [:,1] bias);
 * mean, rstd f32[B*C] are written by the forward and read by the backward (biased variance).
 * Backward overwrites gx and g_gamma_beta f32[B,2,C]; with relu != 0 the mask is recomputed from x.
 * Every [B,C,N] argument has a batch stride in floats (0 = C*N, contiguous; >= C*N for a channel slice of a wider
 * tensor — keys_bn / values_bn on the halves of keys_values_pred, multihead_ct_adain.py:108-111); `residual` (nullable)
 * is added after the ReLU (the union's skip connection, :216), its cotangent is gy itself.
 * ---------------------------------------------------------------------- */
int ct_adain_fwd(const float* x, long long x_batch_stride, const float* gamma_beta, const float* residual,
                 long long residual_batch_stride, float* y, long long y_batch_stride, float* mean, float* rstd,
                 int B, int C, int N, float eps, int relu, ct_stream_t s);
int ct_adain_bwd(const float* x, long long x_batch_stride, const float* gamma_beta, const float* mean, const float* rstd,
                 const float* gy, long long gy_batch_stride, float* gx, long long gx_batch_stride, float* g_gamma_beta,
                 int B, int C, int N, int relu, ct_stream_t s);
/* Several adaptive instance norms over the same (B, N) in ONE launch (n <= 8), as ct_bn_group_fwd / _bwd: fields as the
 * arguments of ct_adain_fwd_amax / ct_adain_bwd_amax. */
typedef struct {
  const float* x; long long x_batch_stride; const float* gamma_beta; const float* residual; long long residual_batch_stride;
  float* y; long long y_batch_stride; float* mean; float* rstd; float* amax_out; long long amax_batch_stride; int C; float eps;
  int relu;
  long long gamma_beta_batch_stride;   /* floats between clouds in gamma_beta (0 = 2*C: contiguous [B,2,C]); larger for a column
                                          range of a stacked style projection [B, sum 2*C_i] */
} ct_adain_fwd_item;
typedef struct {
  const float* x; long long x_batch_stride; const float* gamma_beta; const float* mean; const float* rstd; const float* gy;
  long long gy_batch_stride; float* gx; long long gx_batch_stride; float* g_gamma_beta; float* amax_out;
  long long amax_batch_stride; int C; int relu;
  long long gamma_beta_batch_stride;   /* as in ct_adain_fwd_item (g_gamma_beta is written contiguous [B,2,C]) */
} ct_adain_bwd_item;
int ct_adain_group_fwd(const ct_adain_fwd_item* items, int n, int B, int N, ct_stream_t s);
int ct_adain_group_bwd(const ct_adain_bwd_item* items, int n, int B, int N, ct_stream_t s);
/* The same two with amax_out (nullable): amax_out[b * amax_batch_stride + c] = max |y| / max |g_x| of row (b, c), a by-product of
 * the pass (amax_batch_stride 0 = C) — operand maxima for ct_pw_gemm, as ct_bn_relu_fwd_amax. */
int ct_adain_fwd_amax(const float* x, long long x_batch_stride, const float* gamma_beta, const float* residual,
                      long long residual_batch_stride, float* y, long long y_batch_stride, float* mean, float* rstd,
                      float* amax_out, long long amax_batch_stride, int B, int C, int N, float eps, int relu, ct_stream_t s);
int ct_adain_bwd_amax(const float* x, long long x_batch_stride, const float* gamma_beta, const float* mean,
                      const float* rstd, const float* gy, long long gy_batch_stride, float* gx, long long gx_batch_stride,
                      float* g_gamma_beta, float* amax_out, long long amax_batch_stride, int B, int C, int N, int relu,
                      ct_stream_t s);

/* ------------------------------------------------------------------------
 * Grouped 3^dim convolution over the rasterised planes / volumes, stride 1, padding 1
 * (MultiHead.conv: layers/multihead_ct.py:50-65 with bias; Res2DBlock / Res3DBlock:
 * unet2d/unet_parts.py:13-16, layers/v2v_groups.py:26-29 without), fp32 on the matrix
 * cores (v_mfma_f32_16x16x4_f32: exact fp32, one rounding per product, k-ordered).
 *   x f32[B, groups*Cin, *W]   w f32[groups*Cout, Cin, 3^dim]   bias f32[groups*Cout] | NULL
 *   y f32[B, groups*Cout, *W]
 * bwd_data : g_x from g_y and w.   bwd_weight: g_w (and g_bias unless NULL) from x and g_y;
 * both outputs are overwritten.  bwd_weight sums over (batch, positions) in workgroup-sized chunks:
 * with a workspace of ct_gconv_bwd_weight_workspace_bytes (0 = the shape does not use one) the chunk
 * sums are stored and added in a fixed order (bitwise reproducible); with workspace NULL they are
 * accumulated into g_w with float atomics (order not fixed, and several times slower).
 * ---------------------------------------------------------------------- */
int ct_gconv_fwd(const float* x, const float* w, const float* bias, float* y,
                 int B, int groups, int Cin, int Cout, int dim, const int* W, ct_stream_t s);
int ct_gconv_bwd_data(const float* g_y, const float* w, float* g_x,
                      int B, int groups, int Cin, int Cout, int dim, const int* W, ct_stream_t s);
int ct_gconv_supported(int B, int groups, int Cin, int Cout, int dim, const int* W);   /* 1: all three passes have a tile plan; 0: fall back to the library conv
                                                                                              (very wide rows with many channels per group) */
size_t ct_gconv_bwd_weight_workspace_bytes(int B, int groups, int Cin, int Cout, int dim, const int* W);
int ct_gconv_bwd_weight(const float* x, const float* g_y, float* g_w, float* g_bias,
                        void* workspace, size_t workspace_bytes,
                        int B, int groups, int Cin, int Cout, int dim, const int* W, ct_stream_t s);

/* ------------------------------------------------------------------------
 * Plane-resident MHCT core: the part of a MultiHead block between its norms,
 *   out = Slice(lattice, conv(Splat(lattice, values)))                (layers/multihead_ct.py:99-107:
 *   DifferentiablePositions -> Splat (max, zero floor) -> grouped 3^dim conv with bias -> Slice;
 *   layers/multihead_ct_adain.py:118-125 and layers/cloud_transform.py:72-227 likewise)
 * as ONE kernel per direction-independent plane: the rasterised grid z and the convolved grid y of a (batch, head)
 * plane stay in LDS.  Shapes: the grids whose two tiles fit a CU beside the filter bank — 2D 32x32 C16, 2D 16x16 C16,
 * 3D 8x8x8 C32 (ct_mhct_core_supported != 0), N % 4 == 0, 16-byte aligned tensors.
 *   keys f32[B,H*dim,N] (the lattice), feat f32[B,H*C,N], pad as for Splat, conv_w f32[H*C,C,3^dim], conv_b f32[H*C] | NULL
 *   out f32[B,H*C,N] (overwritten)
 *   z_save, y_save f32[B,H*C,G] | NULL: the two grids as tensors, for ct_mhct_core_bwd (training); with NULL they never
 *     leave the chip.  occ_count int64[1] | NULL: number of |z| > 1e-9 (layers/multihead_ct.py:104-105).
 * With few planes (B*H < CUs/2) a plane is shared by a cluster of 2..8 workgroups that exchange their partial grids
 * through `workspace` (ct_mhct_core_workspace_bytes).  The workspace also holds the clusters' arrival counters, which
 * must be ZERO when a launch starts: call ct_mhct_core_workspace_init once after allocating it (one small memset); every
 * launch leaves the counters zeroed again, so the workspace can be reused launch after launch (stream-ordered) without
 * further resets.  A cluster spins on its partners (bounded): do not run two of these launches concurrently on different
 * streams of one device, nor share one workspace between streams.
 * Backward (ct_mhct_core_bwd): from the saved grids, as Slice backward -> conv backward (data, weight) -> Splat backward
 * on this library's kernels; overwrites g_feat f32[B,H*C,N], g_keys f32[B,H*dim,N] (Slice's and Splat's key cotangents
 * summed), g_w f32[H*C,C,3^dim], g_b f32[H*C] | NULL; workspace of ct_mhct_core_bwd_workspace_bytes.
 * ---------------------------------------------------------------------- */
int ct_mhct_core_supported(int B, int H, int C, int N, int dim, const int* W);
size_t ct_mhct_core_workspace_bytes(int B, int H, int C, int N, int dim, const int* W);
int ct_mhct_core_workspace_init(void* workspace, size_t workspace_bytes, int B, int H, int C, int N, int dim, const int* W,
                                ct_stream_t s);
int ct_mhct_core_fwd(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* conv_w,
                     const float* conv_b, float* out, float* z_save, float* y_save, int64_t* occ_count,
                     void* workspace, size_t workspace_bytes, int B, int H, int C, int N, int dim, const int* W,
                     ct_stream_t s);
size_t ct_mhct_core_bwd_workspace_bytes(int B, int H, int C, int N, int dim, const int* W);
int ct_mhct_core_bwd(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* conv_w,
                     const float* z, const float* y, const float* g_out, float* g_feat, float* g_keys, float* g_w,
                     float* g_b, void* workspace, size_t workspace_bytes, int B, int H, int C, int N, int dim,
                     const int* W, ct_stream_t s);
/* the same with arrival tickets (ct_tickets_init; NULL = none) handed to its Slice / Splat backward passes */
int ct_mhct_core_bwd_tk(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* conv_w,
                        const float* z, const float* y, const float* g_out, float* g_feat, float* g_keys, float* g_w,
                        float* g_b, void* workspace, size_t workspace_bytes, void* tickets, int B, int H, int C, int N,
                        int dim, const int* W, ct_stream_t s);
/* The LDS-resident BACKWARD of the core, for the grid whose five tiles fit a CU (2D 16x16 with 16 features per head:
 * ct_mhct_core_bwd_fused_supported): nothing of the forward is read back — one workgroup per plane recomputes z and conv(z)
 * from the points and walks Slice backward -> conv^T / filter cotangent -> Splat backward in LDS (ct_mhct_core_fwd may then be
 * called with z_save = y_save = NULL in training too).  Same outputs as ct_mhct_core_bwd; the per-plane filter / bias
 * cotangents go through `workspace` (ct_mhct_core_bwd_fused_workspace_bytes) and are added over the batch in a fixed order. */
int ct_mhct_core_bwd_fused_supported(int B, int H, int C, int N, int dim, const int* W);
size_t ct_mhct_core_bwd_fused_workspace_bytes(int B, int H, int C, int N, int dim, const int* W);
int ct_mhct_core_bwd_fused(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* conv_w,
                           const float* conv_b, const float* g_out, float* g_feat, float* g_keys, float* g_w, float* g_b,
                           void* workspace, size_t workspace_bytes, int B, int H, int C, int N, int dim, const int* W,
                           ct_stream_t s);
/* A cluster whose workgroup gives up waiting for its partners (never in a correct run: co-residency of a plane's workgroups
 * rests on in-order dispatch and a grid of at most one workgroup per CU) sets the workspace's status word, writes NaN where
 * its results would have gone, and still takes part in the counters' bookkeeping — the workspace stays usable, the failure
 * cannot pass for a result; ct_mhct_core_status reads the word (callers check it at their flush points and re-initialise).
 * Test hooks: bit 0 = one workgroup per plane (no clusters); bit 1 = fault injection: the last workgroup of plane 0 arrives
 * late and its partners time out after a short spin; bits 8.. = force that many workgroups per plane (1, 2, 4, 8).
 * ct_mhct_core_status copies the workspace's status word to the host AFTER synchronising the stream (a test helper, the
 * only call of this library that waits for the device): 0 = no cluster gave up waiting for its partners. */
void ct_debug_set_core(unsigned flags);
/* Test hook of the grouped convolution: bit 0 = small-volume weight gradients take the vector-ALU kernel instead of the
 * matrix-core one (A/B measurements, tools/gconv64_bench.py). */
void ct_debug_set_gconv(unsigned flags);
/* Test hook of the EMD auction: bit 0 = the per-batch update (GetMax, Assign, next list) on one workgroup per batch in every
 * iteration (default: several workgroups per batch while the batch has more than 1024 unassigned points). */
void ct_debug_set_emd(unsigned flags);
int ct_mhct_core_status(const void* workspace, size_t workspace_bytes, int B, int H, int C, int N, int dim, const int* W,
                        int* host_status, ct_stream_t s);

/* ------------------------------------------------------------------------
 * Chamfer distance (chamfer_extension/chamfer_cuda.cpp:30-33 `forward`,
 * `backward`; kernels chamfer.cu:12-195).  xyz1 f32[B,n,3], xyz2 f32[B,m,3];
 * dist1 f32[B,n], idx1 i32[B,n] (nearest point of cloud 2, lowest index on
 * ties), dist2/idx2 symmetric.  Backward overwrites g_xyz1, g_xyz2.
 * ---------------------------------------------------------------------- */
int ct_chamfer_fwd(const float* xyz1, const float* xyz2, float* dist1, float* dist2,
                   int32_t* idx1, int32_t* idx2, int B, int n, int m, ct_stream_t s);
int ct_chamfer_bwd(const float* xyz1, const float* xyz2, const float* g_dist1, const float* g_dist2,
                   const int32_t* idx1, const int32_t* idx2, float* g_xyz1, float* g_xyz2,
                   int B, int n, int m, ct_stream_t s);

/* ------------------------------------------------------------------------
 * Approximate EMD by auction (emd_linear/emd.cpp:28-31 `forward`, `backward`;
 * kernels emd_cuda.cu:23-316).  xyz1, xyz2 f32[B,n,3] in [0,1]^3; outputs
 * dist f32[B,n] (squared distance to the assigned target) and assignment
 * i32[B,n].  Preconditions as the reference (emd_cuda.cu:236-249): n % 1024 == 0,
 * B <= 512 -> CT_EPRECOND otherwise; iters >= 1.  The reference's 12 caller-allocated
 * scratch tensors (emd_module.py:41-54) become one opaque workspace.
 * Results: the reference's own kernels (built for gfx950, oracle/Makefile `ref_emd`) return
 * the same assignment exactly and the same distances (bit for bit against their
 * -ffp-contract=off build, <= 2 ulp against the default-contraction one) wherever they are
 * deterministic; where several bidders are within 1e-6 of a target's best increment the
 * reference lets the last store win (emd_cuda.cu:181-194) — here: the highest bidder index,
 * always (tests/test_emd_gpu.py, tests/golden/emd_reference.npz).
 * ---------------------------------------------------------------------- */
size_t ct_emd_workspace_bytes(int B, int n);
int ct_emd_fwd(const float* xyz1, const float* xyz2, float* dist, int32_t* assignment,
               void* workspace, size_t workspace_bytes, int B, int n, float eps, int iters,
               ct_stream_t s);
/* g_xyz1 = 2 g_dist (x1 - x2[assignment]); no gradient to xyz2 (emd_module.py:66-70). */
int ct_emd_bwd(const float* xyz1, const float* xyz2, const float* g_dist, const int32_t* assignment,
               float* g_xyz1, int B, int n, ct_stream_t s);

/* ------------------------------------------------------------------------
 * Pointwise (kernel size 1) convolutions of the MHCT blocks and their gradients
 * (layers/multihead_ct.py:31-33,62-66,89-91: nn.Conv1d(Ci, Co, 1) on [B,Ci,N]), fp32 in /
 * fp32 out on the f16 matrix pipes: operands are scaled by a per-tensor power of two
 * and split into two f16 terms (22 bits), three MFMA terms per product, fp32 accumulation.
 *   CT_PW_FWD:   a = W f32[Co,Ci], b = x f32[B,Ci,N]   -> out = y f32[B,Co,N]   (y[b] = W x[b])
 *   CT_PW_DGRAD: a = W f32[Co,Ci], b = g_y f32[B,Co,N] -> out = g_x f32[B,Ci,N] (W^T g_y[b]; W^T goes
 *                through the workspace)
 *   CT_PW_WGRAD: a = g_y f32[B,Co,N], b = x f32[B,Ci,N] -> out = g_W f32[Co,Ci] (sum_b g_y[b] x[b]^T,
 *                partial sums added in a fixed order: deterministic)
 * amax_a / amax_b: device f32[n_amax_*] whose maximum is (an upper bound within ~2^10 of) max |.| of the
 * whole operand tensor; the GEMM folds them when it starts.  Either ct_amax_f32's ct_amax_len()
 * partial maxima, or what the kernel that produced the operand left behind (ct_bn_relu_fwd_amax /
 * _bwd_amax: one per channel; ct_adain_*_amax: one per (cloud, channel)); n_amax_* <= 32768.  NULL = scale 1 (the caller then guarantees
 * |values| < 65504).  Co, Ci, N multiples of 4,
 * 16-byte aligned pointers -> CT_EINVAL otherwise.  Workspace: ct_pw_gemm_workspace_bytes.
 * ---------------------------------------------------------------------- */
#define CT_PW_FWD 0
#define CT_PW_DGRAD 1
#define CT_PW_WGRAD 2
#define CT_PW_DGRAD_T 3 /* as CT_PW_DGRAD with a = W^T f32[Ci,Co] already (ct_pw_prep_weight): no workspace */
int ct_amax_f32(const float* x, int64_t n, float* amax, ct_stream_t s);
int ct_amax_len(void);
/* a layer's weight for all three products in one launch: its partial maxima (ct_pw_prep_weight_partials(Co, Ci) of them, 0 =
 * too many: use ct_amax_f32) and, when wt != NULL, W^T for CT_PW_DGRAD_T */
int ct_pw_prep_weight_partials(int Co, int Ci);
int ct_pw_prep_weight(const float* w, float* wt, float* amax, int Co, int Ci, ct_stream_t s);
size_t ct_pw_gemm_workspace_bytes(int mode, int B, int Co, int Ci, int N);
int ct_pw_gemm(int mode, const float* a, const float* b, float* out, const float* amax_a, int n_amax_a,
               const float* amax_b, int n_amax_b, void* workspace, size_t workspace_bytes, int B, int Co, int Ci,
               int N, ct_stream_t s);
/* Per-ROW scales.  The split keeps 22 bits of an element relative to its SCALE, and f16's exponent range ends the story below
 * ~2^-17 of it: with one scale per tensor an element 2^-17 .. 2^-24 below the tensor's maximum keeps 22 .. 15 bits, and below
 * 2^-24 it is flushed.  Error bound of one output, eps = 2^-21, M_a / M_b the maxima the scales were taken from:
 *     |err_ij| <= eps * sum_k |a_ik b_kj|  +  2^-38 * ( M_a * sum_k |b_kj|  +  M_b * sum_k |a_ik| ).
 * The second term is invisible while the rows of an operand are of one magnitude (post-norm activations) and takes over for
 * an output whose OWN operand row is tiny against its tensor: the weight gradient's row co sees only channel co of g_y — a
 * nearly-dead channel group 2^-24 below the tensor's maximum came out at 1e-5 relative, 2^-26 at 7e-5 (rocBLAS fp32: 1e-7).
 * ct_pw_gemm_rs takes the maxima PER ROW of an operand's k-contiguous arrangement — rows_a > 0: amax_a is
 * f32[n_amax_a / rows_a][rows_a], rows_a = Co (CT_PW_FWD: rows of W; CT_PW_WGRAD: channels of g_y) or Ci (CT_PW_DGRAD_T: rows
 * of W^T); rows_b = Ci for the channels of x in CT_PW_WGRAD — scales every row by its own power of two and takes the factor
 * out of that row's (column's) outputs, exactly: M_a, M_b in the bound become the ROW's maxima, i.e. the bound holds relative to
 * every output's own operands.  Both operands of the weight gradient can be row-scaled (its summed index is the point, not
 * the channel); in the forward and the data gradient the activation's summed index IS the channel, so it keeps one scale
 * (rows_b is ignored there) and its term of the bound stays M_b * sum_k |a_ik|: a contribution that is itself 2^-17 of the
 * row's largest.  Producers of per-row maxima: ct_bn_relu_*_amax / ct_bn_apply_*_amax (per channel), ct_adain_*_amax (per
 * (cloud, channel)), ct_amax_rows_f32 (x f32[B,C,N] -> f32[C]), ct_pw_prep_weight_rs (W -> rowmax f32[ceil(Ci/32)][Co] for
 * CT_PW_FWD, colmax f32[ceil(Co/32)][Ci] for CT_PW_DGRAD_T, and W^T).  n_amax_* <= 32768.  rows_* = 0: ct_pw_gemm. */
int ct_amax_rows_f32(const float* x, int B, int C, int N, float* amax, ct_stream_t s);
int ct_pw_prep_weight_rs(const float* w, float* wt, float* rowmax, float* colmax, int Co, int Ci, ct_stream_t s);
int ct_pw_gemm_rs(int mode, const float* a, const float* b, float* out, const float* amax_a, int n_amax_a, int rows_a,
                  const float* amax_b, int n_amax_b, int rows_b, void* workspace, size_t workspace_bytes, int B, int Co,
                  int Ci, int N, ct_stream_t s);
/* ct_pw_gemm_rs with out = product + addend, the addend read in the kernel's epilogue (f32 laid out as `out`, 16-byte aligned,
 * not `out` itself; NULL = ct_pw_gemm_rs).  For the data gradient of a projection whose INPUT also feeds a skip connection
 * (layers/multihead_ct.py:170-198: `residual = shortcut(x)` beside `keys_values_pred(x)`): autograd's sum of the two cotangents
 * of x — a pass over three tensors the size of x — becomes one more read inside the GEMM.  CT_PW_FWD / CT_PW_DGRAD /
 * CT_PW_DGRAD_T; CT_PW_WGRAD (its output is a fold of slabs) -> CT_EINVAL. */
int ct_pw_gemm_rs_add(int mode, const float* a, const float* b, float* out, const float* addend, const float* amax_a, int n_amax_a,
                      int rows_a, const float* amax_b, int n_amax_b, int rows_b, void* workspace, size_t workspace_bytes, int B,
                      int Co, int Ci, int N, ct_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDCT_H */
