/*
 * pnr.h -- C ABI of libpnr_hip.so, the MI355X (gfx950) native replacement for the four CUDA
 * extensions of zfkuang/PaletteNeRF (`_raymarching`, `_gridencoder`, `_shencoder`, `_palette_func`).
 *
 * Conventions (they mirror the reference's pybind layer):
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - the caller allocates every buffer (inputs, outputs, scratch); kernels never allocate;
 *   - zero-initialisation contracts are the reference's (xyzs/dirs/deltas of the march kernels,
 *     grad_embeddings, SH grad_inputs are caller-zeroed);
 *   - sizes are passed explicitly, exactly as the reference's Python passes them;
 *   - `stream` is a hipStream_t (the reference launches on the legacy default stream; here the
 *     caller passes torch.cuda.current_stream().cuda_stream);
 *   - return value: 0 on success, PNR_ERR_* (<0) otherwise; pnr_error_string() describes it.
 *     The Python shim turns a non-zero return into RuntimeError (the reference raises
 *     RuntimeError through TORCH_CHECK / std::runtime_error).
 *   - threads: no entry point keeps global mutable state (pnr_set_option's switches apart); the frame calls keep a pinned control
 *     block and the previous frame's iteration count per host thread and device and work inside the caller's workspace, so
 *     several frames may be in flight at once -- one host thread, one workspace and one stream each (palettenerf_amd/pipeline.py).
 *
 * Each entry point cites the reference interface it replaces (file:line under the reference).
 */
#ifndef PNR_H_
#define PNR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNR_OK 0
#define PNR_ERR_INVALID (-1)      /* null pointer / bad size                                  */
#define PNR_ERR_UNSUPPORTED (-2)  /* C not in {1,2,4,8}, D not in 1..5, n_channel > 128, ...  */
#define PNR_ERR_LAUNCH (-3)       /* hipGetLastError() != hipSuccess after a launch           */
#define PNR_ERR_ALIGNMENT (-4)    /* pnr_mlp_*: an activation array that does not start on a 16-byte boundary or holds fewer than four floats */

#define PNR_DTYPE_F32 0
#define PNR_DTYPE_F16 1

/* arithmetic of the fused field kernels: exact-fp32 matrix path (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain)
 * or split-fp16 (3 x v_mfma_f32_32x32x16_f16 per product, ~2^-22 relative, fp32 accumulate; ~5x fewer MFMA cycles) */
#define PNR_FIELD_FP32 0
#define PNR_FIELD_F16X3 1
/* opt-in: the COLOUR layers with weights split as above and activations rounded ONCE to fp16 (2 MFMAs per product, one conversion per activation
 * pair): ~1e-5 on a colour with unit-scale weights -- inside the 1e-4 colour contract, not fp32-class.  sigma_net keeps the split form: densities,
 * alphas, depth and the march are bit for bit those of PNR_FIELD_F16X3.  Same packed blobs.  Exists for the NeRF field and for the 4-basis PaletteNeRF
 * field without an edit head; the density-only call, other basis counts, edited and watched launches run it as PNR_FIELD_F16X3. */
#define PNR_FIELD_F16X2 2

#define PNR_CHANNEL_MAXIMUM 128   /* raymarching/src/raymarching.cu:13 */

typedef void* pnr_stream_t; /* hipStream_t */

const char* pnr_error_string(int code);
/* ABI version of this header; bumped on any signature change. */
int pnr_abi_version(void);
/* run-time switches that change speed only, for A/B measurements and tests: "block_skip" (exact jumps over empty blocks in the march),
 * "aux_fusion" (PaletteNeRF frame loop: aux composite inside the field kernel), "coop_march" (wave-cooperative march tail); these default to 1.
 * "composite_fusion" (frame loops; default 2): 2 = the field kernels do the whole compositing step of every iteration (three launches per
 * iteration), 1 = the NeRF field kernel composites the iterations with one sample per ray only, 0 = the composite launch does.  "adam_variant" (0..7, default 0): which multiply-adds of pnr_adam_step
 * are left uncontracted (kept for re-deriving the bit-exact form against a new torch build); "iteration_margin" (default 0): spare iterations
 * the frame loops enqueue beyond the previous frame's count before their first look at the control block; "hosted_tail" (default 1): the frame
 * loops' march launches hand the rays they have not finished within "march_budget" (default 2; "march_budget0" for a frame's first launch, default
 * 0 = that launch finishes every ray itself) sample-less probes to the first workgroups of the lookup launch that follows -- same rows, bit for bit;
 * "march_blocks" (default 0 = automatic): workgroup cap of such a budgeted march launch; "coarse_image" (default 1): pnr_grid_encode_backward_binned accumulates the
 * coarsest levels (tables of at most 16 384 rows) as LDS images instead of records; "cell_merge" (default 1): on its mid levels the samples of a ray that sit in
 * one cell are summed before their records are written; "scatter_staged" (default 1): its records are ordered by bucket in LDS and written coalesced
 * (0 = every lane writes its own records); "mlp_f16x3" (default 1): the training MLP launches use split-fp16 products; "train_coop" (default 1): pnr_march_rays_train*'s counting pass
 * marches four rays per wave cooperatively (same counts, same rows); "coop_march" also governs the cooperative tail of pnr_march_rays* */
int pnr_set_option(const char* name, int value);

/* ---------------------------------------------------------------- raymarching: utils ------- */

/* replaces near_far_from_aabb, raymarching/src/raymarching.h:7, raymarching.cu:151-159 */
int pnr_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                           float min_near, float* nears, float* fars, pnr_stream_t stream);
/* replaces sph_from_ray, raymarching.h:8, raymarching.cu:204-212 */
int pnr_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                     pnr_stream_t stream);
/* replaces morton3D / morton3D_invert, raymarching.h:9-10, raymarching.cu:232-263 */
int pnr_morton3d(const int32_t* coords, uint32_t N, int32_t* indices, pnr_stream_t stream);
int pnr_morton3d_invert(const int32_t* indices, uint32_t N, int32_t* coords, pnr_stream_t stream);
/* replaces packbits, raymarching.h:11, raymarching.cu:295-303 ; N = number of output bytes */
int pnr_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, pnr_stream_t stream);

/* ---------------------------------------------------------------- raymarching: training ---- */

/* Scratch size (bytes) pnr_march_rays_train / pnr_compact_alive need for N elements. */
uint64_t pnr_scan_scratch_bytes(uint32_t N);

/* replaces march_rays_train, raymarching.h:13, raymarching.cu:485-493.
 * Same arguments plus `scratch` (>= pnr_scan_scratch_bytes(N) bytes).  Row order of `rays` and the
 * sample offsets are DETERMINISTIC (ray n -> row n, offset = counter[0] + exclusive prefix sum of
 * the per-ray counts) instead of atomicAdd-ordered; counter[0] += total, counter[1] += N. */
int pnr_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                         float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                         const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                         int32_t* rays, int32_t* counter, const float* noises, void* scratch,
                         pnr_stream_t stream);

/* replaces composite_rays_train_forward / _backward, raymarching.h:14-15, raymarching.cu:647-655, 821-829 */
int pnr_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                     const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                     float* weights_sum, float* depth, float* image, pnr_stream_t stream);
int pnr_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                      const float* rgbs, const float* deltas, const int32_t* rays,
                                      const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                      float T_thresh, float* grad_sigmas, float* grad_rgbs, pnr_stream_t stream);
/* replaces composite_rays_flex_train_forward / _backward, raymarching.h:16-17, raymarching.cu:657-668, 831-844 */
int pnr_composite_rays_flex_train_forward(const float* sigmas, const float* input, const float* deltas,
                                          const int32_t* rays, uint32_t M, uint32_t N, uint32_t n_channel,
                                          float T_thresh, float* output, pnr_stream_t stream);
int pnr_composite_rays_flex_train_backward(const float* grad_output, const float* sigmas, const float* input,
                                           const float* deltas, const int32_t* rays, const float* output,
                                           uint32_t M, uint32_t N, uint32_t n_channel, float T_thresh,
                                           float* grad_input, pnr_stream_t stream);
/* replaces spread_ray_to_sample, raymarching.h:18, raymarching.cu:884-894 */
int pnr_spread_ray_to_sample(const float* input, const int32_t* rays, uint32_t M, uint32_t N, uint32_t n_channel,
                             float* output, pnr_stream_t stream);

/* ---------------------------------------------------------------- raymarching: inference --- */

/* replaces march_rays, raymarching.h:20, raymarching.cu:1014-1021 */
int pnr_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                   uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                   float* xyzs, float* dirs, float* deltas, const float* noises, pnr_stream_t stream);
/* replaces composite_rays, raymarching.h:21, raymarching.cu:1187-1193 (in place) */
int pnr_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                       const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                       float* depth, float* image, pnr_stream_t stream);
/* replaces composite_rays_flex, raymarching.h:22, raymarching.cu:1195-1205 (in place on output) */
int pnr_composite_rays_flex(uint32_t n_alive, uint32_t n_step, uint32_t n_channel, float T_thresh,
                            const int32_t* rays_alive, const float* rays_t, const float* sigmas,
                            const float* input, const float* deltas, const float* weights_sum, float* output,
                            pnr_stream_t stream);
/* SURVEY 8(b)'s "multi-map variant" (ABI 7): the six / seven composite_rays_flex calls one march iteration of PaletteRenderer.run_cuda issues over the SAME
 * sigmas / deltas / rays_alive / weights_sum (palette/renderer.py:508-516) as ONE launch -- a ray's weights are formed once and folded into every map.  Each
 * map's output is bit for bit what pnr_composite_rays_flex gives (same fmaf chain per channel).  Legal to batch because composite_rays_flex writes neither
 * rays_alive nor rays_t nor weights_sum (raymarching.cu:1114-1185) and the reference issues all of them BEFORE the iteration's composite_rays (:517-519).
 * maps: host array of n_maps <= PNR_FLEX_MAX_MAPS entries (n_channel <= 128 each; an entry with n_channel 0 is skipped); n_step > 8 runs the maps one by one. */
#define PNR_FLEX_MAX_MAPS 8
typedef struct {
    uint32_t n_channel;
    const float* input;     /* [n_alive * n_step, n_channel] */
    float* output;          /* [N, n_channel], accumulated in place */
} pnr_flex_map;
int pnr_composite_rays_flex_multi(uint32_t n_alive, uint32_t n_step, float T_thresh, const int32_t* rays_alive,
                                  const float* rays_t, const float* sigmas, const float* deltas,
                                  const float* weights_sum, const pnr_flex_map* maps, uint32_t n_maps,
                                  pnr_stream_t stream);

/* MI355X-first additions: occupancy mip.  The bitfield is Morton-ordered, so a 4x4x4 brick of cells is one
 * aligned 8-byte word; pnr_build_occupancy_mip reduces every brick to an any-bit and an all-bit
 * (pnr_occupancy_mip_bytes(C,H) bytes, device) and appends the world-space bounding box of all occupied bricks
 * (two-cell margin).  The *_mip march entry points stage it in LDS, answer probes in uniformly empty / full bricks
 * without a global load and stop a ray once it has left the occupied box for good (no sample can follow).  Results are bit-identical to pnr_march_rays /
 * pnr_march_rays_train (which are the same kernels with mip == NULL).  Requires H % 4 == 0 and an 8-byte aligned
 * bitfield; the mip must be rebuilt whenever the bitfield changes.  `noises` may be NULL (= no perturbation). */
uint64_t pnr_occupancy_mip_bytes(uint32_t C, uint32_t H);
int pnr_build_occupancy_mip(const uint8_t* grid, uint32_t C, uint32_t H, float bound, void* mip, pnr_stream_t stream);
int pnr_march_rays_mip(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                       const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                       uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                       float* xyzs, float* dirs, float* deltas, const float* noises, const void* mip,
                       pnr_stream_t stream);
/* The same march for buffers that arrive UNINITIALISED (the reference's Python zero-fills xyzs / dirs / deltas with three launches before every
 * call, raymarching.py:384-386): with fill_rows = the row count of the three buffers (>= n_alive * n_step; the alignment padding included) the
 * kernel itself zeroes every slot a ray leaves unfilled and the rows beyond n_alive * n_step.  fill_rows = 0: pnr_march_rays_mip. */
int pnr_march_rays_fill(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                       const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                       uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                       float* xyzs, float* dirs, float* deltas, const float* noises, const void* mip, uint32_t fill_rows,
                       pnr_stream_t stream);
int pnr_march_rays_train_mip(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                             float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                             const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                             int32_t* rays, int32_t* counter, const float* noises, void* scratch, const void* mip,
                             float* t_store /* optional [N * max_steps] floats: the counting pass keeps every sample's ray parameter there and the
                                               rows are then written 16 lanes per ray without a second walk; same outputs */,
                             pnr_stream_t stream);

/* ---------------------------------------------------------------- occupancy maintenance ----- */

/* Device-resident producer of the bitfield: replaces NeRFRenderer.update_extra_state, nerf/renderer.py:467-561
 * (= PaletteRenderer's, palette/renderer.py), whose reference form is Python over torch ops with a host read of the mean (`.item()`),
 * `nonzero`, boolean-mask scatters and raymarching.packbits (raymarching.h:11).  Per cascade c and visited cell: a jittered point
 *     p = (2 q / (H - 1) - 1) * (b_c - b_c / H) + (2 u - 1) * b_c / H,   b_c = min(2^c, bound), q = the cell's integer coordinates, u = noise
 * (fp32, operation by operation as the reference's torch expressions round on a GPU) -> sigma(p) * density_scale -> candidate of the cell;
 * then density_grid = max(density_grid * decay, candidate) where both are >= 0, mean = mean(max(density_grid, 0)),
 * bitfield = packbits(density_grid > min(mean, density_thresh)) and the brick mip of the march.  No step involves the host.
 *   mode 0 (the reference while iter_density < 16): every cell once; noise [C, H^3, 3], row = the cell's MORTON index (the order of density_grid).
 *   mode 1: per cascade n_partial uniformly drawn cells, coords int32 [C, n_partial, 3] in [0, H), followed by n_partial cells drawn from the
 *           currently occupied ones (density > 0), the k-th in ascending Morton order with k = occ_rand[c][j] % (number occupied) -- occ_rand
 *           int32 [C, n_partial] of non-negative random integers (what torch.randint reduces modulo the range); a cascade without an occupied
 *           cell skips that half (the reference raises there).  noise [C, 2 n_partial, 3] in sample order.
 * The random numbers are the CALLER's (the reference draws them with torch.rand_like / torch.randint), so two implementations can be compared.
 * Cells drawn more than once keep the LARGEST candidate (the reference keeps an unspecified one of them).
 *   pnr_occupancy_update   the whole sweep for the shipped field (16-level x 2 hash grid -> sigma_net 32 -> 64 -> 16): level-major lookup +
 *                          exact-fp32 matrix-core sigma_net (packed_sigma_net = a PNR_FIELD_FP32 blob of pnr_nerf_field_pack) + commit.
 *   any other field:       pnr_occupancy_begin; for slices of [0, pnr_occupancy_samples): pnr_occupancy_points -> caller's sigma(points)
 *                          -> pnr_occupancy_scatter; pnr_occupancy_commit.
 * points rows are float4 = world x, y, z and the global cell id c * H^3 + morton as int32 bits (-1: no sample).
 * workspace: pnr_occupancy_workspace_bytes(C, H, chunk) bytes, 256-byte aligned; `chunk` = samples in flight in pnr_occupancy_update (>= 256; the
 * other entry points need chunk 0 only).  H % 4 == 0, C <= 16, C * H^3 < 2^31.  state (optional): device float[2] = mean, threshold used. */
typedef struct pnr_occupancy_args {
    uint32_t C, H;
    float bound;
    float* density_grid;               /* [C, H^3] in/out */
    uint8_t* density_bitfield;         /* [C * H^3 / 8] out */
    void* mip;                         /* pnr_occupancy_mip_bytes(C, H) out, or NULL */
    float density_scale, decay, density_thresh;
    int mode;
    uint32_t n_partial;
    const float* noise;
    const int32_t* coords;
    const int32_t* occ_rand;
    const float* embeddings;           /* pnr_occupancy_update only: fp32 hash table [rows, 2] ... */
    const int32_t* offsets;
    uint32_t num_levels;
    float S;
    uint32_t base_resolution, gridtype;
    const float* packed_sigma_net;
    void* workspace;
    uint64_t workspace_bytes;
    float* state;
    float* points_out;                 /* pnr_occupancy_update only, optional: [samples, 4], every sample's point kept for inspection */
} pnr_occupancy_args;
uint64_t pnr_occupancy_workspace_bytes(uint32_t C, uint32_t H, uint32_t chunk);
uint32_t pnr_occupancy_samples(const pnr_occupancy_args* args);
int pnr_occupancy_update(const pnr_occupancy_args* args, pnr_stream_t stream);
int pnr_occupancy_begin(const pnr_occupancy_args* args, pnr_stream_t stream);
int pnr_occupancy_points(const pnr_occupancy_args* args, uint32_t first, uint32_t count, float* points, pnr_stream_t stream);
int pnr_occupancy_scatter(const pnr_occupancy_args* args, const float* points, const float* sigmas, uint32_t count, pnr_stream_t stream);
int pnr_occupancy_commit(const pnr_occupancy_args* args, pnr_stream_t stream);
/* replaces NeRFRenderer.mark_untrained_grid, nerf/renderer.py:395-465: cells whose centre no training camera has in its frustum (slack of one
 * cell), or that some camera sees closer than min_near (filter_close_point: or that lie within min_near of a camera), get density -1.
 * poses [B,4,4] row-major cam2world; n_marked (optional): device int32, number of cells marked. */
int pnr_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t C, uint32_t H, float bound, float min_near,
                            int filter_close_point, float* density_grid, int32_t* n_marked, pnr_stream_t stream);

/* Stable compaction replacing the host-side `rays_alive[rays_alive >= 0]` boolean mask
 * (nerf/renderer.py:376, palette/renderer.py:521).  Writes the surviving ids, in order, to
 * rays_alive_out and their number to n_alive_out[0].  scratch >= pnr_scan_scratch_bytes(n_alive). */
int pnr_compact_alive(uint32_t n_alive, const int32_t* rays_alive_in, int32_t* rays_alive_out,
                      int32_t* n_alive_out, void* scratch, pnr_stream_t stream);

/* ---------------------------------------------------------------- grid encoder ------------- */

/* replaces grid_encode_forward, gridencoder/src/gridencoder.h:12, gridencoder.cu:424-447.
 * dtype selects the table/outputs/dy_dx element type (PNR_DTYPE_F32 / PNR_DTYPE_F16).
 * outputs is [L,B,C] as in the reference; dy_dx may be NULL. offsets: device int32[L+1]. */
int pnr_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                            uint32_t gridtype, int align_corners, int dtype, pnr_stream_t stream);
/* MI355X-first addition: the same lookup with the output layout chosen by the caller.  PNR_LAYOUT_LEVELS = [L,B,C] (the reference kernel's,
 * gridencoder.cu:119); PNR_LAYOUT_ROWS = [B, L*C], what GridEncoder.forward returns after its permute-copy (gridencoder/grid.py:57) -- written
 * directly, same values.  PNR_LAYOUT_ROWS exists for D = 3, C = 2 without dy_dx (every shipped configuration); PNR_ERR_UNSUPPORTED otherwise. */
#define PNR_LAYOUT_LEVELS 0
#define PNR_LAYOUT_ROWS 1
int pnr_grid_encode_forward_layout(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                                   uint32_t gridtype, int align_corners, int dtype, int layout, pnr_stream_t stream);
/* Two tables of the same geometry (D = 3, C = 2, fp32; the same `offsets`) at the same points in one pass: `pair_embeddings` holds them interleaved row by row,
 * [row][table 0 | table 1][2] (16-byte rows, 16-byte aligned); out0 / out1 [L, B, 2] are bit for bit what pnr_grid_encode_forward returns for table 0 / table 1.
 * (palette/network.py:156-262 looks `encoder` and `encoder_palette` up at the same x.) */
int pnr_grid_encode_forward_pair(const float* inputs, const float* pair_embeddings, const int32_t* offsets, float* out0, float* out1, uint32_t B, uint32_t L, float S,
                                 uint32_t H, uint32_t gridtype, int align_corners, pnr_stream_t stream);
/* replaces grid_encode_backward, gridencoder.h:13, gridencoder.cu:449-479.
 * grad is [L,B,C]; grad_embeddings caller-zeroed; dy_dx/grad_inputs may both be NULL. */
int pnr_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                             void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                             uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                             int align_corners, int dtype, pnr_stream_t stream);

/* ---------------------------------------------------------------- fused field (MFMA) ------- */

/* MI355X-first addition, no single-call counterpart in the reference: evaluates everything
 * nerf/network.py:95-124 does after the hash-grid lookup (sigma_net 32->64->16, exp, SH degree 4, concat,
 * color_net 31->64->64->3, sigmoid) in one kernel on the fp32 matrix cores.
 *   pnr_nerf_field_pack: gathers the five bias-free nn.Linear weight matrices (row-major [out][in], device fp32)
 *                        into the MFMA-fragment-ordered blob `packed` (pnr_nerf_field_packed_bytes() bytes).  With all three colour
 *                        weights NULL only the sigma_net part is written (enough for pnr_nerf_density_forward / pnr_occupancy_update).
 *   pnr_nerf_field_forward: enc = raw [16,B,2] output of pnr_grid_encode_forward (fp32), dirs [B,3] -> sigmas [B]
 *                        (= exp(h0), NOT multiplied by density_scale), rgbs [B,3]. */
uint64_t pnr_nerf_field_packed_bytes(void);
int pnr_nerf_field_pack(const float* w_sigma0, const float* w_sigma1, const float* w_color0, const float* w_color1,
                        const float* w_color2, float* packed, int precision, pnr_stream_t stream);
int pnr_nerf_field_forward(const float* enc, const float* dirs, const float* packed, uint32_t B, float* sigmas,
                           float* rgbs, int precision, float enc_scale, pnr_stream_t stream);
/* enc_scale (here, in pnr_nerf_density_forward and in the frame / palette argument structs): a power of two the encoder features are
 * multiplied by before the split-fp16 matrix path splits them, undone exactly after the first (bias-free, positively homogeneous) stack;
 * 1 (or 0) = none.  For hash tables whose entries are tiny (the reference initialises them U(-1e-4, 1e-4), gridencoder/grid.py:107) it keeps
 * the low halves out of fp16's subnormal range.  Ignored by PNR_FIELD_FP32. */
/* sigma_net alone (NeRFNetwork.density / the frozen-geometry half of PaletteNetwork.forward: nerf/network.py:126-143,
 * palette/network.py:161-170): sigmas [B] = scale * exp(h0), geo_feat [B,15] = h[1:] (NULL: not wanted).  Same enc layout, packed blob
 * and precision modes as pnr_nerf_field_forward.  No gradient: for inference, the occupancy sweep and PaletteNeRF training (whose
 * geometry is detached). */
int pnr_nerf_density_forward(const float* enc, const float* packed, uint32_t B, float scale, float* sigmas, float* geo_feat, int precision,
                             float enc_scale, pnr_stream_t stream);

/* Device-driven inference frame of the NeRF path: the loop of nerf/renderer.py:344-380 (same n_step schedule,
 * same per-ray arithmetic, order-preserving compaction) with n_alive / n_step / step kept in a device control
 * block, so the host does not synchronise per iteration.  Outputs are the raw accumulations (weights_sum [N],
 * depth [N], image [N,3]) BEFORE the background mix / depth normalisation of nerf/renderer.py:382-383.
 * perturb is False (inference).  workspace: pnr_nerf_frame_workspace_bytes(N) bytes of device memory. */
typedef struct pnr_nerf_frame_args {
    uint32_t N;
    const float* rays_o;           /* [N,3] */
    const float* rays_d;           /* [N,3] */
    float* nears;                  /* [N]   in (pnr_near_far_from_aabb); with `aabb` set: OUT, written by the call */
    float* fars;                   /* [N]   likewise */
    const uint8_t* bitfield;       /* density bitfield, uint8[C*H^3/8] */
    const void* mip;               /* pnr_build_occupancy_mip output, or NULL */
    float bound;
    uint32_t C, H;
    float dt_gamma;
    uint32_t max_steps;
    float T_thresh;
    const float* embeddings;       /* fp32 hash table [sum T_l, 2] */
    const int32_t* offsets;        /* int32[num_levels+1] */
    uint32_t num_levels;           /* must be 16 (the fused field kernel's input width) */
    float S;                       /* log2(per_level_scale) */
    uint32_t base_resolution;
    uint32_t gridtype;             /* 0 hash, 1 tiled */
    const float* packed_weights;   /* pnr_nerf_field_pack output (packed with the same precision) */
    int field_precision;           /* PNR_FIELD_FP32 or PNR_FIELD_F16X3 */
    float density_scale;
    float* weights_sum;            /* [N]   out */
    float* depth;                  /* [N]   out */
    float* image;                  /* [N,3] out */
    void* workspace;
    uint64_t workspace_bytes;
    uint64_t* stats;               /* HOST, optional, 6 entries: [iterations, rendered samples, evaluated rows, enqueued iterations,
                                      host looks at the control block (= stream synchronisations of this frame),
                                      1 if watch_overflow saw an operand beyond fp16's range (the frame is then invalid: render it with PNR_FIELD_FP32)] */
    float* kernel_ms;              /* HOST, optional: [0] = summed HIP-event time (ms) of the grid-encode launches that did work,
                                      [1] = their number; events are recorded on `stream` around each launch */
    const int32_t* ray_order;      /* optional permutation of 0..N-1 (device): processing order of the rays, e.g. 8x8 pixel tiles per
                                      wave; NULL = as given.  The frame then runs on copies of the per-ray inputs gathered into that
                                      order; outputs are scattered back, indexed by ray id either way, and do not depend on the order */
    int finish;                    /* bit mask: apply the caller-side epilogue of run_cuda (nerf/renderer.py:382-384) before returning --
                                      bit 0: image += (1 - weights_sum) * bg_color;  bit 1: depth = max(depth - near, 0) / (far - near);
                                      bit 2 (pnr_palette_render_frame): aux_map[:, 0:3] (direct_rgb) += (1 - weights_sum) * bg_color (palette/renderer.py:529)
                                      (same fp32 operations, in the same order, as the reference's torch expressions) */
    float bg_color[3];             /* used when finish != 0 and bg_map == NULL (the reference's default is 1) */
    const float* bg_map;           /* optional per-ray background [N,3] (device) for finish */
    int table_dtype;               /* PNR_DTYPE_F32 (default) or PNR_DTYPE_F16: the reference's --fp16 tables (`embeddings` then points to halves;
                                      PaletteNeRF: embeddings_pair = both tables as interleaved halves, required, no clip head).  The lookup
                                      then reproduces the reference's half interpolation (as pnr_grid_encode_forward with dtype 1) */
    float enc_scale[3];            /* power-of-two prescale of the features of `embeddings` (PaletteNeRF: + embeddings_palette, embeddings_clip) in the
                                      split-fp16 field, see pnr_nerf_field_forward; 0 or 1 = none */
    int watch_overflow;            /* PNR_FIELD_F16X3 only: the field kernels watch the operands they split for magnitudes beyond fp16's range
                                      (65 504) and report through stats[5]; for weights whose activations the caller cannot bound (~1 VALU per operand) */
    const float* aabb;             /* optional (device, 6 floats: xmin ymin zmin xmax ymax zmax): the call computes `nears` / `fars` itself, inside its first
                                      launch, with pnr_near_far_from_aabb's arithmetic (raymarching.cu:95-148; same bits) and WRITES them to `nears` / `fars`
                                      (by ray id) -- run_cuda's near_far_from_aabb call (nerf/renderer.py:268) folded into the frame; NULL: `nears` / `fars` are inputs */
    float min_near;                /* used with `aabb` */
    float* depth_raw;              /* optional [N] out: the un-normalised depth by ray id when finish bit 1 rewrites `depth`
                                      (PaletteNeRF's depth_origin, palette/renderer.py:522) */
} pnr_nerf_frame_args;
uint64_t pnr_nerf_frame_workspace_bytes(uint32_t N);
int pnr_nerf_render_frame(const pnr_nerf_frame_args* args, pnr_stream_t stream);
/* The same frame in two calls (ABI 7).  _submit enqueues the frame's first launch, as many march iterations as the previous frame needed, the frame's last
 * launch and the 64-byte control-block read-back, and returns WITHOUT waiting: the host is free to prepare its next frame (rays, argument struct, outputs)
 * while this one runs -- the loop being replaced (nerf/renderer.py:344-380, palette/renderer.py:430-550) holds the host for the whole frame.  _finish waits for
 * the read-back, enqueues further iterations while the frame is not done (the iteration count is data) and fills `stats` / `kernel_ms`.  Rules: _finish follows
 * _submit on the same host thread, device and stream with the SAME argument struct (its address identifies the frame; PNR_ERR_INVALID otherwise); one
 * submitted frame per host thread and device (a second _submit is PNR_ERR_INVALID; a whole-frame call drops a submitted frame that was never finished);
 * nothing the frame reads or writes may be touched in between.  pnr_nerf_render_frame == _submit + _finish. */
int pnr_nerf_render_frame_submit(const pnr_nerf_frame_args* args, pnr_stream_t stream);
int pnr_nerf_render_frame_finish(const pnr_nerf_frame_args* args, pnr_stream_t stream);

/* Appearance editing heads of the PaletteNeRF inference loop, evaluated inside the fused field kernel's epilogue (HOST struct).
 *   mode 1  RegionEdit.forward (palette/renderer.py:121-147): per basis, final colour -> HSV (pnr_rgb_to_hsv's arithmetic), hue += delta_hsv[b][0]
 *           (+360, fmod 360), saturation / value *= delta_hsv[b][1..2] (clipped at 0), -> RGB, then lerp(original, edited, weight) with
 *           weight = exp(-|xyz - mean_xyz|^2 / std_xyz) * exp(-|clip_feat - mean_clip|^2 / std_clip) (each factor only when its mean is set);
 *           weight_mode != 0 returns the weight itself in every channel (the GUI's region preview).
 *   mode 2  Stylizer.forward (palette/renderer.py:166-183): rgb = sum_b omega_b clamp(max(softplus(r) + dI_b, 0) (P_b + dP_b + offsets_b . ddelta_b), 0, 1)
 *           + view_dep  (offsets_weight / view_dep_weight do not apply, as in the reference).
 *   mode 3  (pnr_palette_field_forward only) "network heads": no composite at all -- the aux row is what PaletteNetwork.forward returns per sample
 *           (palette/network.py:156-190): omega nb (normalised) | offsets_radiance 3 nb + 1 (raw, bias added) | view_dep 3 | diffuse 3 |
 *           clip_feat clip_dim | 0-pad; sigmas = density_scale * exp(logit), rgbs = 0.  For callers that keep the reference's own renderer
 *           arithmetic (palette/renderer.py:452-499) and only want the network as one launch (palettenerf_amd.dropin.fuse_field). */
#define PNR_MAX_BASIS 10
#define PNR_MAX_CLIP 32
typedef struct pnr_palette_edit {
    int mode;                                /* 0 none, 1 RegionEdit, 2 Stylizer, 3 network heads (stand-alone op only) */
    float delta_hsv[PNR_MAX_BASIS][3];
    int has_mean_xyz;  float mean_xyz[3];  float std_xyz;
    int has_mean_clip; float mean_clip[PNR_MAX_CLIP]; float std_clip;   /* has_mean_clip = number of entries (the model's clip_dim), 0 = not set */
    int weight_mode;
    float dI[PNR_MAX_BASIS]; float dP[PNR_MAX_BASIS][3]; float ddelta[PNR_MAX_BASIS][3][3];
} pnr_palette_edit;

/* The same device-driven loop for the PaletteNeRF model (palette/renderer.py:430-550, RegionEdit / Stylizer included through `edit`):
 * `base.embeddings` is the `encoder` table, `base.packed_weights` the pnr_palette_field_pack blob (packed with base.field_precision:
 * PNR_FIELD_F16X3 or the exact PNR_FIELD_FP32).  num_basis <= PNR_MAX_BASIS, clip_dim <= PNR_MAX_CLIP.
 * aux_map [N, pnr_palette_aux_channels(nb, clip_dim)] receives the composited packed row
 * direct_rgb 3 | view_dep 3 | basis_acc nb | basis_rgb 3nb | unscaled_basis_rgb 3nb | clip_feat clip_dim | pad
 * (raw accumulations: the background mix of direct_rgb is the caller's, palette/renderer.py:541). */
typedef struct pnr_palette_frame_args {
    pnr_nerf_frame_args base;
    const float* embeddings_palette;   /* `encoder_palette` table (same offsets / level parameters as `encoder`) */
    const float* embeddings_clip;      /* `encoder_clip` table (pred_clip only) */
    uint32_t num_basis, clip_dim;
    int pred_clip;
    float offsets_weight, view_dep_weight;
    float* aux_map;                    /* [N, aux channels] out */
    const float* embeddings_pair;      /* optional: `encoder` and `encoder_palette` interleaved row by row ([rows][4] floats, built by
                                          pnr_interleave_tables; rebuild when either table changes).  Used when pred_clip == 0: one 16-byte
                                          gather serves both tables, results bit-identical to the separate lookups */
    const float* embeddings_triple;    /* optional, pred_clip != 0: all three tables interleaved ([rows][8] floats: encoder, encoder_palette,
                                          encoder_clip, 2 pad; pnr_interleave_tables3): one 32-byte row per corner, bit-identical results */
    const pnr_palette_edit* edit;      /* HOST, optional: RegionEdit / Stylizer applied to every sample (NULL or mode 0: none) */
} pnr_palette_frame_args;
/* out[i] = (a[i].x, a[i].y, b[i].x, b[i].y) for two C = 2 fp32 tables of `rows` rows with the same level layout */
int pnr_interleave_tables(const float* a, const float* b, uint64_t rows, float* out, pnr_stream_t stream);
/* a, b, c: [rows][2] fp32 -> out [rows][8] = (a.x, a.y, b.x, b.y, c.x, c.y, 0, 0): the --pred_clip model's three lookups from one 32-byte row */
int pnr_interleave_tables3(const float* a, const float* b, const float* c, uint64_t rows, float* out, pnr_stream_t stream);
uint64_t pnr_palette_frame_workspace_bytes(uint32_t N, uint32_t num_basis, uint32_t clip_dim, int pred_clip);
int pnr_palette_render_frame(const pnr_palette_frame_args* args, pnr_stream_t stream);
int pnr_palette_render_frame_submit(const pnr_palette_frame_args* args, pnr_stream_t stream);   /* as pnr_nerf_render_frame_submit / _finish */
int pnr_palette_render_frame_finish(const pnr_palette_frame_args* args, pnr_stream_t stream);

/* Fused PaletteNeRF field + palette colour-basis composite (palette/network.py:156-280, palette/renderer.py:470-500, RegionEdit /
 * Stylizer included).  Split-fp16 or exact-fp32 matrix path (`precision`).  All weights are row-major [out][in]
 * device fp32 pointers of the bias-free nn.Linear layers; offsets_radiance has a bias; the palette (basis_color) and that bias are packed too. */
typedef struct pnr_palette_weights {
    const float *sigma0, *sigma1;              /* [64,32], [16,64]            */
    const float *diff0, *diff1, *diff2;        /* [64,15], [64,64], [3,64]    */
    const float *color0, *color1, *color2;     /* [64,31], [64,64], [3,64]    */
    const float *basis0, *basis1;              /* [64,35], [15,64]            */
    const float *offsets_radiance;             /* [3 nb + 1, 15]              */
    const float *omega;                        /* [nb, 15]                    */
    const float *clip0, *clip1;                /* [64,32], [clip_dim,64] (pred_clip only) */
    const float *basis_color;                  /* [nb,3] the palette (device; clamped to [0,1] at pack time, palette/renderer.py:480) */
    const float *or_bias;                      /* [3 nb + 1] offsets_radiance_net.bias (device) */
    uint32_t num_basis, clip_dim;              /* nb <= PNR_MAX_BASIS, clip_dim <= PNR_MAX_CLIP */
    int pred_clip;
    int precision;                             /* PNR_FIELD_F16X3 or PNR_FIELD_FP32: number format of the packed blob */
} pnr_palette_weights;
typedef struct pnr_palette_field_args {
    const void* ctl;               /* NULL for the stand-alone op (rows = B); internal frame control block otherwise */
    uint32_t B;                    /* rows */
    const float* enc;              /* [16, level_stride, 2] raw grid output of `encoder` */
    const float* enc_palette;      /* same for `encoder_palette` */
    const float* enc_clip;         /* same for `encoder_clip` (pred_clip only) */
    uint32_t level_stride;
    const float* dirs;             /* [B,3] */
    const float* deltas;           /* [B,2] or NULL; rows with deltas[:,0] == 0 are skipped */
    const void* packed;            /* pnr_palette_field_pack output (weights, palette and bias) */
    uint32_t num_basis, clip_dim;
    int pred_clip;
    float density_scale, offsets_weight, view_dep_weight;
    uint32_t aux_stride;           /* floats per aux row, >= pnr_palette_aux_channels(nb, clip_dim) semantics below */
    float* sigmas;                 /* [B]   = density_scale * exp(h0) */
    float* rgbs;                   /* [B,3] */
    float* aux;                    /* [B, aux_stride] = direct_rgb 3 | view_dep 3 | omega nb | basis_rgb 3nb | unscaled 3nb | clip | 0 pad */
    /* frame loop only (ctl != NULL), all three or none: in iterations with one sample per ray (ctl->n_step == 1) the aux row is
     * composited straight into aux_map (aux_map[ray] += weight * row, the arithmetic of composite_rays_flex) instead of being written
     * to `aux` and read back by the composite launch -- when the kernel can stage the tile in LDS (pnr_palette_field_stages_aux);
     * likewise with 2, 4 or 8 samples per ray (a ray's rows then sit in one 32-row tile) */
    const int32_t* rays_alive;     /* [n_alive] ray id of every slot */
    const float* weights_sum;      /* [N] of BEFORE this iteration */
    float* aux_map;                /* [N, aux_stride] */
    float T_thresh;                /* early-termination threshold of the composite (frame loop only) */
    int precision;                 /* PNR_FIELD_F16X3 / PNR_FIELD_FP32, as the blob was packed */
    const pnr_palette_edit* edit;  /* HOST, optional */
    const float* xyzs;             /* [B,3] world positions of the samples: needed by RegionEdit's spatial window (edit->has_mean_xyz) */
    const void* edit_device;       /* internal (frame loop): the edit parameters already on the device; NULL for callers */
    float enc_scale[3];            /* power-of-two prescales of enc / enc_palette / enc_clip in the split-fp16 path (0 or 1 = none) */
    int32_t* overflow_flag;        /* optional (device): set to 1 when a split-fp16 operand exceeds fp16's range (the kernel then watches its operands) */
    void* tile_counter;            /* internal (frame loop): device uint32, zero at launch -- waves fetch their 32-sample tiles from it; NULL = static schedule */
    /* internal (frame loop), all or none: with these the kernel also does the ray-state half of the iteration's compositing step (what the frame loop's
     * composite launch did): weights_sum / depth / image / rays_t of every ray, the alive list's holes, the per-chunk survivor counts */
    float* rays_t;                 /* [N] */
    float* weights_sum_rw;         /* [N] = weights_sum */
    float* depth;                  /* [N] */
    float* image;                  /* [N,3] */
    int32_t* rays_alive_rw;        /* [n_alive] = rays_alive */
    int32_t* counts_cur;           /* [chunks] survivors per 256-ray chunk of the alive list, cleared by the iteration's march launch */
} pnr_palette_field_args;
int pnr_palette_field_stages_aux(uint32_t num_basis, uint32_t clip_dim, int pred_clip);   /* 1 when the field kernel stages aux rows in LDS (then it can composite them) */
uint64_t pnr_palette_field_packed_bytes(uint32_t num_basis, uint32_t clip_dim, int pred_clip);
uint32_t pnr_palette_aux_channels(uint32_t num_basis, uint32_t clip_dim);   /* 6 + 7 nb + clip_dim rounded up to a multiple of 4 */
int pnr_palette_field_pack(const pnr_palette_weights* weights, void* packed, pnr_stream_t stream);
int pnr_palette_field_forward(const pnr_palette_field_args* args, pnr_stream_t stream);

/* ---------------------------------------------------------------- SH encoder --------------- */

/* replaces sh_encode_forward / sh_encode_backward, shencoder/src/shencoder.h:9-10, shencoder.cu:400-439.
 * fp32 only (the reference's Python forces fp32, sphere_harmonics.py:16). dy_dx may be NULL. */
int pnr_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx,
                          pnr_stream_t stream);
/* color_net's input rows [SH(dirs) | tail] = torch.cat([encoder_dir(d), geo_feat], dim=-1) (nerf/network.py:109-115, palette/network.py:248-249) in one
 * launch: inputs [B,3] unit directions, tail [B,tail_cols] -> outputs [B, C*C + tail_cols]; the SH columns are pnr_sh_encode_forward's bits.
 * C*C + tail_cols <= 64.  (No Jacobian: directions get no gradient on this path; the tail's gradient is the column slice of the output's.) */
int pnr_sh_encode_cat_forward(const float* inputs, const float* tail, uint32_t tail_cols, float* outputs, uint32_t B, uint32_t C,
                              pnr_stream_t stream);
/* The seam between sigma_net and color_net of the NeRF field in training (nerf/network.py:109-121: `sigma = trunc_exp(h[..., 0])`, `geo_feat = h[..., 1:]`,
 * `torch.cat([encoder_dir(d), geo_feat], -1)`) as one launch each way, so that autograd does not zero-fill, slice-copy and add two [B, hw] gradients:
 *   forward   h [B, hw] row-major, dirs [B, 3]  ->  sigma [B] = exp(h[:, 0]),  out [B, C*C + hw - 1] = [SH_C(dirs), h[:, 1:]]
 *   backward  grad_h [B, hw] = [dsigma * exp(clamp(h[:, 0], -15, 15)), dout[:, C*C:]]   (activation.py:14-17; dsigma / dout may be NULL = zeros) */
int pnr_sigma_geo_cat_forward(const float* h, uint32_t hw, const float* dirs, uint32_t C, uint32_t B, float* sigma, float* out, pnr_stream_t stream);
int pnr_sigma_geo_cat_backward(const float* h, uint32_t hw, const float* dsigma, const float* dout, uint32_t C, uint32_t B, float* grad_h, pnr_stream_t stream);
int pnr_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                           const float* dy_dx, float* grad_inputs, pnr_stream_t stream);

/* Table gradient of the hash grid for D = 3, C = 2, fp32 (the shipped fields) without global atomics in the inner loop: records are
 * binned by 8192-row table bucket, each bucket is accumulated in LDS and added to the table once (csrc/grid_binned.hip).  Same
 * contract as the embeddings part of pnr_grid_encode_backward (gridencoder.cu:216-286: grad [L,B,C] level-major, accumulates into
 * the caller-zeroed grad_embeddings); total_rows = rows of the table (offsets[L]); other shapes return PNR_ERR_UNSUPPORTED (use
 * pnr_grid_encode_backward).  workspace: pnr_grid_backward_binned_workspace_bytes(B, L, total_rows) bytes (10 B per record). */
uint64_t pnr_grid_backward_binned_workspace_bytes(uint32_t B, uint32_t L, uint64_t total_rows);
int pnr_grid_encode_backward_binned(const float* grad, const float* inputs, const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D,
                                    uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint64_t total_rows,
                                    void* workspace, uint64_t workspace_bytes, pnr_stream_t stream);

/* ---------------------------------------------------------------- training: dense-layer weight gradient */

/* dW[o][i] (+)= sum_b dY[b][o] * X[b][i]: the weight gradient autograd computes for every nn.Linear(in, out, bias=False) of the
 * fields (nerf/network.py:60-93, palette/network.py:60-153; torch sends it to the BLAS library in the reference).
 *   x [B, in_dim], dy [B, out_dim] row-major, fp32 or fp16 (PNR_DTYPE_*; autocast hands over halves), dw [out_dim, in_dim] fp32;
 *   in_dim, out_dim <= 64 (every layer of both models), else PNR_ERR_UNSUPPORTED; accumulate != 0 adds to dw;
 *   workspace: pnr_linear_wgrad_workspace_bytes(B, in_dim, out_dim) bytes of device memory (per-workgroup partials, reduced in a
 *   fixed order: the result is deterministic).  Products and sums are fp32 (v_mfma_f32_32x32x2_f32). */
uint64_t pnr_linear_wgrad_workspace_bytes(uint32_t B, uint32_t in_dim, uint32_t out_dim);
int pnr_linear_wgrad(const void* x, int x_dtype, const void* dy, int dy_dtype, uint32_t B, uint32_t in_dim, uint32_t out_dim, float* dw,
                     int accumulate, void* workspace, uint64_t workspace_bytes, pnr_stream_t stream);
/* The fields' small bias-free MLPs for TRAINING as one launch forward and one launch backward (replaces the nn.Linear(bias=False) stacks
 * of nerf/network.py:33-93 and palette/network.py:60-153 under autograd: per layer a GEMM + activation kernel forward, two GEMMs + an
 * activation kernel backward).  2 or 3 layers, every width <= 64, hidden activation 0 = ReLU, 1 = ELU(alpha 1), optionally torch.sigmoid on the output;
 * fp32 (exact fma chains on v_mfma_f32_32x32x2_f32).  weights w_l are [dims[l+1]][dims[l]] row-major as nn.Linear stores them.
 *   pnr_mlp_pack      weights -> MFMA-ordered blob of pnr_mlp_packed_bytes (W_l and W_l^T, once as fp32 and once as split-fp16 pairs scaled by one power
 *                     of two per layer); call again after every optimiser step
 *   The forward runs on the fp16 matrix pipe with split operands (22-bit products, fp32 accumulation, per-tile power-of-two scaling; outputs within 2e-6
 *   of the exact launch relative to the largest output); pnr_set_option("mlp_f16x3", 0) selects the exact fp32 instructions.  The backward runs in
 *   the same split-fp16 arithmetic (k_mlp_bwd_h) for the shapes whose tile it holds without scratch -- two layers unless both ends are 64 wide, three
 *   layers with ends <= 32 wide: every stack of both fields -- and on the exact fp32 instructions otherwise, so a wide stack's forward (split-fp16) and
 *   backward (fp32) differ in arithmetic; gradients of both agree with the exact launches to 2e-6 of the largest entry (profiles/r04_grad_tolerance.txt).
 *   pnr_mlp_forward   x [B, dims[0]] -> y [B, dims[n_layers]]
 *   pnr_mlp_backward  x, dy [B, dims[n_layers]] -> dx [B, dims[0]] (NULL: not wanted), dw_l [dims[l+1]][dims[l]] (NULL: not wanted);
 *                     hidden activations are recomputed from x; dw is reduced deterministically through `workspace`.
 *                     y: the forward's output, read only with PNR_MLP_OUT_SIGMOID (dZ = dY (1 - y) y; NULL otherwise).
 *   Every activation array (x, x_tail, y, dy, dx; the level-major forms' enc / denc) must start on a 16-byte boundary and hold at least four floats -- tiles move
 *   as 16-byte requests (round 5); PNR_ERR_ALIGNMENT otherwise (a contiguous view such as x[1:] of a 3-wide tensor starts inside an allocation: copy it). */
typedef struct {
    uint32_t n_layers;      /* 2 or 3 */
    uint32_t dims[4];       /* dims[0] = input width ... dims[n_layers] = output width, each 1..64 */
    int activation;         /* between layers: 0 ReLU, 1 ELU;  | PNR_MLP_OUT_SIGMOID: torch.sigmoid on the last layer's output as well */
} pnr_mlp_desc;
#define PNR_MLP_OUT_SIGMOID 0x100   /* colour heads: torch.sigmoid(h) behind color_net / diff_net (nerf/network.py:122, palette/network.py:245,254) */
uint64_t pnr_mlp_packed_bytes(const pnr_mlp_desc* desc);
int pnr_mlp_pack(const pnr_mlp_desc* desc, const float* w0, const float* w1, const float* w2, float* packed, pnr_stream_t stream);
int pnr_mlp_forward(const pnr_mlp_desc* desc, const float* packed, const float* x, uint32_t B, float* y, pnr_stream_t stream);
uint64_t pnr_mlp_backward_workspace_bytes(const pnr_mlp_desc* desc, uint32_t B);
int pnr_mlp_backward(const pnr_mlp_desc* desc, const float* packed, const float* x, const float* y, const float* dy, uint32_t B, float* dx, float* dw0,
                     float* dw1, float* dw2, void* workspace, uint64_t workspace_bytes, pnr_stream_t stream);

/* The same with the first 32 input columns taken straight from a hash-grid encoder output in its native level-major layout
 * enc [16][B][2] (pnr_grid_encode_forward's `outputs`) and the remaining dims[0] - 32 columns from a row-major x_tail [B, dims[0]-32]
 * (NULL when dims[0] == 32): replaces `torch.cat([encoder(x), tail])` + the [L,B,C] -> [B,L*C] copy of gridencoder/grid.py:51-52 in
 * front of sigma_net (nerf/network.py:99-101) and basis_net (palette/network.py:257-260).  The backward returns the encoder part of dX in
 * level-major layout too (what pnr_grid_encode_backward[_binned] takes as `grad`); the tail gets no gradient (the reference detaches it). */
int pnr_mlp_forward_lm(const pnr_mlp_desc* desc, const float* packed, const float* enc_level_major, uint32_t levels, const float* x_tail, uint32_t B, float* y,
                       pnr_stream_t stream);
int pnr_mlp_backward_lm(const pnr_mlp_desc* desc, const float* packed, const float* enc_level_major, uint32_t levels, const float* x_tail, const float* dy,
                        uint32_t B, float* denc_level_major, float* dw0, float* dw1, float* dw2, void* workspace, uint64_t workspace_bytes,
                        pnr_stream_t stream);

/* bias gradient of a dense layer with bias (palette/network.py:111 offsets_radiance_net, the only one): db[o] = sum_b dY[b][o];
 * workspace of pnr_linear_wgrad_workspace_bytes(B, 1, out_dim) */
int pnr_linear_bgrad(const void* dy, int dy_dtype, uint32_t B, uint32_t out_dim, float* db, int accumulate, void* workspace, uint64_t workspace_bytes,
                     pnr_stream_t stream);

/* ---------------------------------------------------------------- training: optimiser step -- */

/* torch.optim.Adam (the reference's optimiser: main_nerf.py:113, main_palette.py:223 -- lr 1e-2, betas (0.9, 0.99), eps 1e-15, no weight decay,
 * no amsgrad) for up to pnr_adam_max_tensors() fp32 tensors in ONE launch: read p, g, m, v -- write p, m, v, once.  Bit-identical to torch's
 * seven elementwise kernels per tensor (same operations, roundings and contractions; torch/optim/adam.py:_single_tensor_adam).  The scalars are
 * the ones torch forms on the host, narrowed to fp32 where its kernels narrow them:
 *   one_minus_beta1 = (float)(1 - beta1), one_minus_beta2 = (float)(1 - beta2), neg_step_size = (float)(-(lr / (1 - beta1^step))),
 *   bias_correction2_sqrt = (float)((1 - beta2^step) ** 0.5)   (torch's default foreach implementation divides by it; option "adam_variant" bit 0
 *   multiplies by its fp32 reciprocal instead, as torch's single-tensor kernels do for a division by a host scalar),
 *   inv_grad_scale: 1, or 1 / GradScaler's scale to fold `unscale_` into the same pass. */
typedef struct pnr_adam_tensor { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; uint64_t n; } pnr_adam_tensor;   /* HOST array of device pointers */
typedef struct pnr_adam_scalars { float one_minus_beta1, beta2, one_minus_beta2, bias_correction2_sqrt, eps, neg_step_size, inv_grad_scale; } pnr_adam_scalars;
uint32_t pnr_adam_max_tensors(void);
int pnr_adam_step(const pnr_adam_tensor* tensors, uint32_t count, const pnr_adam_scalars* scalars, pnr_stream_t stream);

/* ---------------------------------------------------------------- cache guard -------------- */

/* 64-bit checksums of `count` (<= 24) device buffers in one launch: buffers / nbytes / word_stride are HOST arrays (device pointers, sizes in
 * bytes -- multiples of 4 --, and the stride in 4-byte words at which a buffer is sampled: 1 or 0 = every word); out = device uint64[count].
 * Order-independent sums of position-dependent word hashes: deterministic.  Used by the frame loops to notice that the parameters a packed
 * blob was derived from were rewritten behind torch's version counters (`p.data` writes: torch_ema's copy_to / restore, nerf/utils.py:829-839). */
int pnr_checksum(const void* const* buffers, const uint64_t* nbytes, const uint32_t* word_stride, uint32_t count, uint64_t* out, pnr_stream_t stream);

/* ---------------------------------------------------------------- ray generation ----------- */

/* device counterpart of the deterministic core of get_rays (nerf/utils.py:53-149; torch ops on the GPU in the reference):
 * pinhole rays through pixel centres (+0.5), normalised, rotated by the camera-to-world pose.
 *   poses [B,4,4] row-major cam2world; intrinsics = fx, fy, cx, cy (utils.py:67); inds [B,N] int64 flat pixel ids (y*W + x) or
 *   NULL = all H*W pixels in row-major order (then N must be H*W); outputs rays_o, rays_d [B,N,3].
 * Scalar spec (same in the oracle): xs = ((x + 0.5) - cx) / fx, ys likewise, n = sqrt(fma(xs,xs, fma(ys,ys, 1))),
 * d = (xs/n, ys/n, 1/n), rays_d[k] = fma(d2,R[k][2], fma(d1,R[k][1], d0*R[k][0])), rays_o[k] = pose[k][3].
 * The index selection (random / patch / error-map sampling, utils.py:75-131) stays on the torch side. */
int pnr_get_rays(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, const int64_t* inds,
                 uint32_t N, float* rays_o, float* rays_d, pnr_stream_t stream);

/* The frame's last step on the way to the video / PNG writer (nerf/utils.py:719-723, 1013-1017: `(pred * 255).astype(np.uint8)` on the
 * host, after `linear_to_srgb` when the scene is in linear colour, utils.py:43-44, 1010-1011): fp32 values -> uint8 on the device, so that
 * a frame leaves the GPU as 3 bytes per pixel.  u8 = (uint8)(v * 255) (truncation, as numpy's astype for in-range values);
 * linear_to_srgb != 0: v = v < 0.0031308 ? 12.92 v : 1.055 v^0.41666 - 0.055 first. */
int pnr_image_to_uint8(const float* src, uint64_t n, int linear_to_srgb, uint8_t* dst, pnr_stream_t stream);

/* ---------------------------------------------------------------- palette ------------------ */

/* Training-mode palette colour-basis composite as one launch each way (replaces the ~40 torch launches of palette/renderer.py:344-386
 * between PaletteNetwork.forward and composite_rays_train / composite_rays_flex_train; same arithmetic):
 *   omega [M,nb], offsets_radiance [M,3nb+1] (offsets, then radiance), view_dep [M,3], diffuse [M,3], clip_feat [M,clip_dim] or NULL (zeros),
 *   smooth_norm [M] or NULL (zeros), basis_color [nb,3] (the unclamped parameter)
 *   -> rgbs [M,3] = sum_b omega_b softplus(radiance) (clamp(P_b,0,1) + offsets_b) + view_dep
 *      all_buffer [M, 13+clip_dim+nb] = omega_sparsity, view_dep_norm, offsets_norm, smooth_norm, view_dep, diffuse+view_dep, diffuse, clip_feat, omega
 * backward: grad_rgbs [M,3], grad_all [M,13+clip_dim+nb] -> gradients of every input (view_dep gets none through rgbs: the reference detaches
 * it there); grad_clip_feat / grad_smooth_norm / grad_basis_color may be NULL (not wanted).  grad_basis_color [nb,3] is reduced
 * deterministically through `workspace` (pnr_palette_train_shade_workspace_bytes).  nb <= 16. */
uint64_t pnr_palette_train_shade_workspace_bytes(uint32_t num_basis);
int pnr_palette_train_shade_forward(uint32_t M, uint32_t num_basis, uint32_t clip_dim, const float* omega, const float* offsets_radiance,
                                    const float* view_dep, const float* diffuse, const float* clip_feat, const float* smooth_norm,
                                    const float* basis_color, float* rgbs, float* all_buffer, pnr_stream_t stream);
int pnr_palette_train_shade_backward(uint32_t M, uint32_t num_basis, uint32_t clip_dim, const float* omega, const float* offsets_radiance,
                                     const float* view_dep, const float* basis_color, const float* grad_rgbs, const float* grad_all,
                                     float* grad_omega, float* grad_offsets_radiance, float* grad_view_dep, float* grad_diffuse,
                                     float* grad_clip_feat, float* grad_smooth_norm, float* grad_basis_color, void* workspace,
                                     uint64_t workspace_bytes, pnr_stream_t stream);

/* The two heads of PaletteNetwork.color under autograd as one launch each way (replaces, per training step, the two library GEMMs + bias add +
 * softplus + add + row sum + divide of palette/network.py:262-268 and their backward -- about twenty launches):
 *   h [M, in_dim] (basis_net's output, in_dim <= 16), w_offsets_radiance [3nb+1, in_dim], b_offsets_radiance [3nb+1], w_omega [nb, in_dim]
 *   -> offsets_radiance [M, 3nb+1] = h W_or^T + b,   omega [M, nb] = s / sum(s), s = softplus(h W_om^T) + 0.05      (fp32 fma chains)
 * backward: grad_offsets_radiance [M, 3nb+1], grad_omega [M, nb] -> grad_h [M, in_dim] (NULL: not wanted) and
 *   grad_pre [M, 3nb+1+nb] = [grad_offsets_radiance | dL/d(h W_om^T)], from which pnr_linear_wgrad(h, grad_pre) gives the stacked weight
 *   gradient [3nb+1+nb, in_dim] and pnr_linear_bgrad(grad_pre)[:3nb+1] the bias gradient.  nb <= PNR_MAX_BASIS. */
int pnr_palette_heads_forward(const float* h, const float* w_offsets_radiance, const float* b_offsets_radiance, const float* w_omega, uint32_t M,
                              uint32_t num_basis, uint32_t in_dim, float* offsets_radiance, float* omega, pnr_stream_t stream);
int pnr_palette_heads_backward(const float* h, const float* w_offsets_radiance, const float* w_omega, const float* grad_offsets_radiance,
                               const float* grad_omega, uint32_t M, uint32_t num_basis, uint32_t in_dim, float* grad_h, float* grad_pre,
                               pnr_stream_t stream);

/* The ray-level tail of a training step as ONE launch each way: the renderer's epilogue (palette/renderer.py:387-403 -- background blend of
 * image and direct_rgb, depth normalisation; nerf/renderer.py:331-336) and the trainer's loss on it (palette/utils.py:483-600 with the
 * MSELoss(reduction='none') criterion of main_palette.py:181,222; nerf/utils.py:534-556 when all_map is NULL).  In torch these are ~45
 * launches of 4096-element tensors per step, forward and backward.
 *   image      = image_raw + (1 - weights_sum) bg              depth = clamp(depth_raw - nears, 0) / (fars - nears)
 *   direct_rgb = all_map[:, 7:10] + (1 - weights_sum) bg
 *   loss = mean_n mean_c (image - gt_rgb)^2                                                                  terms[1]
 *        + lambda_sparsity mean all_map[:, 0] + lambda_offsets mean all_map[:, 2] + lambda_view_dep mean all_map[:, 1]   terms[2..4]
 *        + lambda_smooth mean all_map[:, 3]                                                                  terms[5]
 *        + lambda_palette mean_b sum_c (basis_color - basis_color_origin)^2     (both NULL: 0)               terms[6]
 *        + lambda_weight mean (gt_weights - all_map[:, 13+clip_dim:])^2         (gt_weights NULL: 0)         terms[7]
 *        + mean (direct_rgb - gt_rgb)^2                                                                      terms[8]
 *        + mean (all_map[:, 13:13+clip_dim] - gt_clip)^2                        (gt_clip NULL: 0)            terms[9]
 *   terms[0] = loss;  loss_ray [N] = mean_c (image - gt_rgb)^2 (what the reference's error map reads before the scalar terms are added)
 * all_map columns are those of pnr_palette_train_shade_forward's all_buffer after the flex composite; n_channel = 13 + clip_dim + num_basis,
 * or 0 with all_map NULL (the NeRF model: only the first term).  bg_mode 0: the constant bg_const; 1: bg_color [3]; 2: bg_color [N,3]
 * (no gradient reaches the background).  The sums are reduced in a fixed order (workgroup partials in `workspace`, combined by the last
 * workgroup to finish), so the value is reproducible.  The backward multiplies by the device scalar *grad_loss and writes EVERY element of
 * grad_weights_sum [N], grad_image_raw [N,3], grad_all_map [N,n_channel] and (if not NULL) grad_basis_color [num_basis,3].
 * image / depth / direct_rgb / loss_ray may be NULL (not wanted); depth needs depth_raw, nears and fars. */
typedef struct pnr_train_loss_args {
    uint32_t N, n_channel, num_basis, clip_dim;
    const float* weights_sum;
    const float* depth_raw;
    const float* image_raw;
    const float* all_map;
    const float* nears;
    const float* fars;
    const float* bg_color;
    float bg_const;
    int32_t bg_mode;
    const float* gt_rgb;
    const float* gt_clip;
    const float* gt_weights;
    const float* basis_color;
    const float* basis_color_origin;
    float lambda_sparsity, lambda_offsets, lambda_view_dep, lambda_smooth, lambda_weight, lambda_palette;
    /* forward outputs */
    float* image;
    float* depth;
    float* direct_rgb;
    float* loss_ray;
    float* terms;            /* [PNR_TRAIN_LOSS_TERMS] */
    /* backward */
    const float* grad_loss;  /* device scalar */
    float* grad_weights_sum;
    float* grad_image_raw;
    float* grad_all_map;
    float* grad_basis_color;
    void* workspace;         /* forward only; pnr_train_loss_workspace_bytes(N), zero-filled once by the caller before its first use */
    uint64_t workspace_bytes;
} pnr_train_loss_args;
#define PNR_TRAIN_LOSS_TERMS 10
uint64_t pnr_train_loss_workspace_bytes(uint32_t N);
int pnr_train_loss_forward(const pnr_train_loss_args* args, pnr_stream_t stream);
int pnr_train_loss_backward(const pnr_train_loss_args* args, pnr_stream_t stream);

/* replace rgb_to_hsv / hsv_to_rgb, palette/src/palette_func.h, palette.cu:135-149 */
int pnr_rgb_to_hsv(uint32_t n, const float* input, float* output, pnr_stream_t stream);
int pnr_hsv_to_rgb(uint32_t n, const float* input, float* output, pnr_stream_t stream);
/* device counterpart of compute_RGB_histogram (CPU C++ in the reference, palette/src/bindings.cpp:40-91):
 * colors_rgb [n,3], weights [n] -> bin_weights double[2^(3 bpc)] (zeroed by the callee), bin_centers float[2^(3 bpc), 3] */
int pnr_rgb_histogram(const float* colors_rgb, const float* weights, uint32_t n, int bits_per_channel, double* bin_weights,
                      float* bin_centers, pnr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PNR_H_ */
