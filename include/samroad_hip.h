/*
 * samroad_hip.h — C ABI of libsamroad_hip.so: the MI355X (gfx950) implementation of sam_road's
 * tiled-inference hot path.  Plain C types only (no torch, no C++ in the signatures).
 *
 * This is the drop-in boundary of SURVEY.md §8(b).  The reference has no FFI of its own for this
 * path (it is pure Python on torch.nn); each entry point below names the reference interface it
 * replaces (file:line under the reference repository) and INTEGRATION.md shows the ctypes binding a
 * sam_road maintainer would add behind `SAMRoad`.
 *
 * Conventions
 *   - every function returns 0 on success or a negative srh_status; srh_last_error(ctx) has text;
 *   - all data pointers are DEVICE pointers on the ctx's device unless documented as host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on
 *     that stream; nothing is retained past return except by srh_weights_pack (which copies);
 *   - the callee allocates nothing in hot calls except growing the ctx-owned workspace the first
 *     time a larger batch is seen;
 *   - image embeddings cross this ABI CHANNELS-LAST: [B, h, w, 256] f32 (the Python shim returns a
 *     permuted view with the reference's logical shape [B, 256, h, w]).
 */
#ifndef SAMROAD_HIP_H
#define SAMROAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRH_ABI_VERSION 8

typedef enum {
    SRH_OK = 0,
    SRH_ERR_BAD_ARG = -1,      /* null pointer, bad dtype code, bad shape */
    SRH_ERR_UNSUPPORTED = -2,  /* configuration not built (e.g. head_dim other than 64 / 80) */
    SRH_ERR_HIP = -3,          /* HIP runtime error (text in srh_last_error) */
    SRH_ERR_MISSING_WEIGHT = -4,
    SRH_ERR_NO_DEVICE = -5,
    SRH_ERR_NONFINITE = -6     /* ABI 8: a LayerNorm pass of an EARLIER call on this context read an Inf / NaN (fp16 overflow upstream) */
} srh_status;

typedef enum { SRH_F32 = 0, SRH_F16 = 1, SRH_U8 = 2, SRH_I32 = 3, SRH_I64 = 4 } srh_dtype;

typedef struct srh_ctx srh_ctx;          /* per (process, device): workspace + error state */
typedef struct srh_weights srh_weights;  /* packed, device-resident, immutable model weights */

/* Architecture, mirrors SAMRoad.__init__ (model.py:197-258, :283-300). */
typedef struct {
    int32_t embed_dim;          /* 768 / 1024 / 1280 */
    int32_t depth;              /* 12 / 24 / 32 */
    int32_t num_heads;          /* 12 / 16 / 16 */
    int32_t patch_size;         /* config.PATCH_SIZE: tile side in pixels (256 / 512) */
    int32_t n_global;           /* number of entries used in global_attn_indexes */
    int32_t global_attn_indexes[8];
    int32_t window_size;        /* 14 */
    int32_t toponet_version;    /* 0 'normal' (and 'no_tgt_features', App. D.7), 1 'no_offset', 2 'no_transformer' */
    int32_t use_sam_decoder;    /* config.USE_SAM_DECODER (model.py:260-282): 0 = naive map_decoder, 1 = SAM PromptEncoder + MaskDecoder */
} srh_model_cfg;

/* One state_dict entry (SURVEY.md Appendix A names), f32, contiguous. */
typedef struct {
    const char* name;
    const void* data;           /* host pointer, or device pointer if on_device != 0 */
    int32_t on_device;
    int32_t ndim;
    int64_t shape[4];
} srh_named_tensor;

int srh_abi_version(void);
/* 16 hex digits: sha256 over the sources (csrc, this header, compiler flags) the library was built from.  The Python shim,
 * __graft_entry__.build() / smoke() and bench.py compare it with the hash of the sources in the tree, so a stale prebuilt
 * library is rebuilt or refused instead of silently measured (the reference has no counterpart: it is interpreted). */
const char* srh_build_id(void);

/* lifetime ------------------------------------------------------------------------------------ */
int srh_ctx_create(int device, srh_ctx** out);
void srh_ctx_destroy(srh_ctx* ctx);
const char* srh_last_error(const srh_ctx* ctx);
/* ABI 8 — non-finite sentinel.  Inter-kernel activations are fp16; a checkpoint whose activations overflow would come out as silently
 * wrong masks (the reference's own guards, inferencer.py:206 NaN -> -100 and :219 assert 0 <= score <= 1, only see TopoNet's output).
 * Every LayerNorm pass of the encoder / neck / map_decoder reads each branch output anyway and flags a row whose variance is not finite
 * (flags live in host-mapped memory: nothing is copied or synchronised in the hot loop).  The condition is reported LAZILY as
 * SRH_ERR_NONFINITE — by the next srh_encode_decode / srh_scene_pass1 on the context, or by srh_ctx_check — with srh_last_error naming
 * the first stage that saw it ("encoder block 7, norm2 ..."); reporting clears it.  synchronize != 0: wait for `stream` first (the
 * answer then covers every call issued so far); 0: only what has already completed. */
int srh_ctx_check(srh_ctx* ctx, void* stream, int synchronize);

/* Packs `net.load_state_dict(ckpt["state_dict"])` + `net.to(device)` (inferencer.py:250-254):
 * fp16 MFMA operand copies of every matmul weight (conv kernels re-ordered to the GEMM K order),
 * f32 biases / LayerNorm affine / pos_embed, fp16 rel_pos tables. */
int srh_weights_pack(srh_ctx* ctx, const srh_model_cfg* cfg, const srh_named_tensor* tensors, int n,
                     srh_weights** out);
void srh_weights_free(srh_weights* w);

/* Multi-GPU weight distribution (north_star: "RCCL broadcast of weights over xGMI"; the reference is single-process,
 * inferencer.py:243-254).  The packed arena is position-independent and its layout depends on srh_model_cfg alone, so ONE
 * rank packs a checkpoint and the others receive the packed fp16/f32 bytes device-to-device — no state_dict travels, no rank
 * but the first reads the checkpoint or re-packs.  srh_weights_export: *bytes = size of the packed arena; with dst != NULL
 * (device pointer, `capacity` bytes) the arena is copied there.  srh_weights_import: builds a weights object for `cfg` from
 * such a copy (device pointer; SRH_ERR_BAD_ARG if `bytes` is not what `cfg` packs to). */
int srh_weights_export(srh_ctx* ctx, const srh_weights* w, void* dst, size_t capacity, size_t* bytes);
int srh_weights_import(srh_ctx* ctx, const srh_model_cfg* cfg, const void* src, size_t bytes, srh_weights** out);

/* model ----------------------------------------------------------------------------------------- */

/* SAMRoad.infer_masks_and_img_features (model.py:459-495; body shared with forward :420-446):
 * rgb [B,P,P,3] channels-last, values 0..255, dtype SRH_F32 or SRH_U8
 *   -> mask_logits (nullable) / mask_scores (nullable) [B,P,P,2] f32, embeddings [B,h,w,256] f32. */
int srh_encode_decode(srh_ctx* ctx, const srh_weights* w, const void* rgb, int rgb_dtype, int B,
                      float* mask_logits, float* mask_scores, float* embeddings, void* stream);

/* SAMRoad.infer_toponet (model.py:498-508) = BilinearSampler (:29-58) + TopoNet (:61-148).
 * embeddings [B,h,w,256] f32 channels-last; points [B,N,2] (x,y) SRH_I64 or SRH_F32;
 * pairs [B,Ns,K,2] SRH_I64 or SRH_I32; valid [B,Ns,K] u8 (bool); K must be 16.
 *   -> logits (nullable) / scores (nullable) [B,Ns,K] f32. */
int srh_toponet(srh_ctx* ctx, const srh_weights* w, const float* embeddings, const void* points,
                int points_dtype, const void* pairs, int pairs_dtype, const uint8_t* valid, int B, int N,
                int Ns, int K, float* logits, float* scores, void* stream);

/* infer_toponet over the query rows of MANY tiles at once, unpadded (pass 2 of infer_one_img, inferencer.py:179-207: the reference
 * pads every batch of INFER_BATCH_SIZE tiles to its longest tile, graph_collate_fn-style; every row is scored on its own, so the
 * padding rows are pure waste — 68 k padded rows for 48 k real ones on a CityScale scene).  Rows = the concatenated per-tile point
 * lists; embeddings [n_tiles,h,w,256]; points f32 [R,2] tile-local (x,y); point_tile i32 [R] = index of the row's tile in
 * `embeddings` (clamped into [0, n_tiles)); pairs i32 [R,K,2] = (row, target row) into the flat list; valid u8 [R,K]; scores f32
 * [R,K] out.  srh_pass2_pack_ragged builds these.
 * ABI 8: tile_offsets (HOST pointer, int64 [n_tiles + 1], nullable) = the first row of every tile, from 0 to R.  With it the rows are
 * scored in chunks of whole tiles of at most 16 384 rows (same bits: rows are independent and a pair only names rows of its own
 * tile), so the context's workspace is bounded like the reference's INFER_BATCH_SIZE batches instead of growing with the scene;
 * without it the call is one launch and refuses more than 65 536 rows. */
int srh_toponet_ragged(srh_ctx* ctx, const srh_weights* w, const float* embeddings, int n_tiles, const float* points,
                       const int32_t* point_tile, const int32_t* pairs, const uint8_t* valid, int64_t R, int K,
                       const int64_t* tile_offsets, float* scores, void* stream);

/* scene level (pass 1 of infer_one_img, inferencer.py:79-110) ------------------------------------ */

/* Tile batcher + model + mask fusion: crops n_tiles PxP tiles at tile_xy[(x0,y0)] (device int32)
 * out of a resident u8 scene [S,S,3], runs them in batches of B, accumulates mask scores into the
 * two f32 canvases [S,S] (caller zero-initialises) in the reference's sequential tile order, and
 * writes every tile's embeddings to embeddings_all [n_tiles,h,w,256]. */
int srh_scene_pass1(srh_ctx* ctx, const srh_weights* w, const uint8_t* scene, int S, const int32_t* tile_xy,
                    int n_tiles, int B, float* canvas_kp, float* canvas_road, float* embeddings_all,
                    void* stream);

/* canvas / coverage-count * 255 -> u8 (truncation; uncovered border -> 0), inferencer.py:106-110. */
int srh_scene_normalise(srh_ctx* ctx, const float* canvas_kp, const float* canvas_road, int S,
                        const int32_t* tile_xy, int n_tiles, int P, uint8_t* kp_u8, uint8_t* road_u8,
                        void* stream);

/* op level (used by the parity tests to localise a failure; same kernels as above) ----------------- */

/* out = act(A[M,K] W[N,K]^T + bias) (+resid); A,W fp16; N%128==0, K%64==0. act: 0/1 GELU/2 ReLU. */
int srh_op_gemm(srh_ctx* ctx, const void* A_f16, const void* W_f16, const float* bias, const float* resid,
                int M, int N, int K, int act, float* out_f32, void* out_f16, void* stream);
/* The same with the operand layouts the model uses between fc1 and fc2 of a ViT block (reference model.py:245-258 through the SAM
 * fork's MLPBlock: lin2(act(lin1(x)))): the MLP's hidden activation is kept in the BLOCKED-16 layout — 1 KiB blocks of 32 rows x 16
 * columns, element (m, n) at fp16 index ((m/32)*(cols/16) + n/16)*512 + ((n%16)/8)*256 + (m%32)*8 + n%8.  flags: SRH_GEMM_A_BLOCKED16
 * = A_f16 is in that layout, SRH_GEMM_OUT_BLOCKED16 = out_f16 is written in it.  Only the persistent 256x192 kernel understands the
 * layout: M % 256 == 0, N % 192 == 0, K % 768 == 0, >= 128 tiles, a bias, fp16 output only, (A blocked, act 0) or (out blocked, act 1
 * GELU); anything else returns SRH_ERR_UNSUPPORTED.  flags == 0 is srh_op_gemm. */
#define SRH_GEMM_A_BLOCKED16 1
#define SRH_GEMM_OUT_BLOCKED16 2
int srh_op_gemm_ex(srh_ctx* ctx, const void* A_f16, const void* W_f16, const float* bias, const float* resid,
                   int M, int N, int K, int act, float* out_f32, void* out_f16, int flags, void* stream);
/* Device bytes the context owns right now (activation / scene / TopoNet workspaces, the GEMM tile-table slab): grows only when a larger
 * batch is first seen, flat in steady state — what tests/test_gpu_ops.py::test_ctx_memory_is_flat_over_batch_sizes watches. */
size_t srh_ctx_device_bytes(const srh_ctx* ctx);
/* 3x3 pad-1 conv over channels-last [B,S,S,C] as implicit GEMM; W_f16 [N, 9*C] (k = tap*C + c). */
int srh_op_conv3x3(srh_ctx* ctx, const void* A_f16, const void* W_f16, int B, int S, int C, int N,
                   float* out_f32, void* stream);
int srh_op_layernorm(srh_ctx* ctx, const float* x, const float* gamma, const float* beta, float eps, int M,
                     int D, int gelu, float* out_f32, void* out_f16, void* stream);
/* SAM attention on a fused qkv tensor [B*S*S, 3*heads*64] fp16: rel-pos tables [2*win-1, 64] fp16,
 * qkv bias fp16 [3*heads*64] (pad-key rows), win = 14 (windowed) or S (global). out fp16 [B*S*S, heads*64]. */
int srh_op_attention(srh_ctx* ctx, const void* qkv_f16, const void* relpos_h_f16, const void* relpos_w_f16,
                     const void* bias_qkv_f16, int B, int S, int heads, int win, void* out_f16, void* stream);
/* The same with the head dim as an argument: 64 (ViT-B / ViT-L) or 80 (ViT-H, toponet_vith_256.yaml;
 * windows of 14 on any S, or the 16x16 global window).  Tables [2*win-1, head_dim], scale head_dim^-0.5
 * (segment_anything Attention.__init__, model.py:245-258 instantiates it with the checkpoint's dims). */
int srh_op_attention_hd(srh_ctx* ctx, const void* qkv_f16, const void* relpos_h_f16, const void* relpos_w_f16,
                        const void* bias_qkv_f16, int B, int S, int heads, int head_dim, int win,
                        void* out_f16, void* stream);

/* profiling --------------------------------------------------------------------------------------- */

/* When enabled, every kernel launch made by this ctx is bracketed by HIP events on the launch
 * stream.  srh_profile_read synchronises, then returns per-class totals since the last enable. */
typedef struct {
    char name[32];
    int64_t launches;
    double ms;          /* summed GPU time of the launches */
    double flops;       /* algorithmic FLOPs (2*M*N*K for GEMMs; q.k + p.v for attention), else 0 */
    double bytes;       /* algorithmic HBM bytes for bandwidth-bound classes, else 0 */
} srh_profile_row;
int srh_profile_enable(srh_ctx* ctx, int on);
int srh_profile_read(srh_ctx* ctx, srh_profile_row* rows, int max_rows, int* n_rows);
/* Calibration of the above: what an event pair around ONE launch adds beyond the time the kernel's waves run — median over 32
 * launches of a kernel that spins for exactly 50 us of the 100 MHz wall clock, minus those 50 us (dispatch, completion and the
 * marker packets; rocprofv3 counts most of it as kernel duration as well).  bench.py reports it next to its event-timed classes. */
int srh_profile_overhead(srh_ctx* ctx, void* stream, double* ms_per_launch);

/* ---- host-side geometry between the two GPU passes (no device work, callable without a GPU) ---------------
 * Greedy radius NMS of reference graph_utils.py:572-591 (nms_points), the step that turns the fused masks into
 * graph points (graph_extraction.py:130-139).  Candidates are given ALREADY in the reference's processing order
 * (np.argsort(scores)[::-1], computed by the caller so numpy's tie order is kept): xy int32 [n,2] (x,y),
 * force[i] = (score_i > 1.0).  Writes kept[i] in {0,1}.  Exact integer arithmetic, identical result to the
 * reference's KDTree loop. */
int srh_nms_points_host(const int32_t* xy, const uint8_t* force, int64_t n, int32_t radius, uint8_t* kept);

/* Pass-2 query builder for all tiles of a scene (reference inferencer.py:148-176: per-tile rtree box query + scipy
 * KDTree.query(k = K+1, distance_upper_bound = R)).  pts int64 [n,2] (x,y) graph points; boxes int32 [n_tiles,4]
 * (x0,y0,x1,y1), closed.  srh_pass2_count writes the number of points per tile; the caller builds
 * offsets = exclusive prefix sum and allocates ids [total] and knn [total,K]; srh_pass2_fill writes, per tile, the point
 * ids in ascending order and each point's K nearest other points of the tile (distance strictly < radius, ascending by
 * (distance, index); -1 = missing).  Where distances alone do not determine scipy's answer — the K-th and (K+1)-th neighbour
 * equidistant, or a point coinciding with the source — the row is decided exactly as `scipy.spatial.KDTree(tile points)
 * .query(p, k = K+1, distance_upper_bound = radius)[1:]` decides it (same kd-tree, traversal and heaps restated in
 * csrc/kdtree_emul.hpp; scipy 1.15) and comes in scipy's output order; ambiguous [total] marks those rows (information
 * only: nothing is left for the caller to recompute); local (nullable) int64 [total,2] receives every row's tile-local (x, y) =
 * point - (x0, y0), the coordinates TopoNet samples at (inferencer.py:151). */
int srh_pass2_count(const int64_t* pts, int64_t n, const int32_t* boxes, int32_t n_tiles, int64_t* counts);
int srh_pass2_fill(const int64_t* pts, int64_t n, const int32_t* boxes, int32_t n_tiles, int32_t K, int64_t radius,
                   const int64_t* offsets, int64_t* ids, int32_t* knn, uint8_t* ambiguous, int64_t* local, int32_t n_threads);

/* `scipy.spatial.KDTree(points, leafsize).query(queries, k, distance_upper_bound=r)` (the reference's kNN, inferencer.py:156-160)
 * restated for 2-D points, INCLUDING how scipy breaks ties between equidistant points: points f64 [n,2], queries f64 [nq,2] ->
 * out_idx i32 [nq,k] in scipy's output order (n where fewer than k points lie within r); tree_indices (nullable) i32 [n]
 * receives scipy's `tree.indices` permutation.  Host code; what srh_pass2_fill uses for tied cut-offs. */
int srh_kdtree_knn_host(const double* points, int64_t n, int32_t leafsize, const double* queries, int64_t nq, int32_t k,
                        double distance_upper_bound, int32_t* out_idx, int32_t* tree_indices);

/* Candidate pixels of a fused u8 mask: reference graph_extraction.py:24-28 (`np.where(mask > threshold)` and the scores
 * there), row-major order.  xy int64 [capacity,2] (x, y) and scores u8 [capacity] receive the candidates and *n their number;
 * if capacity is too small (or xy = scores = NULL) only *n is written (SRH_ERR_BAD_ARG in the former case): call again with
 * capacity >= *n.  n_threads worker threads scan bands of rows (count, prefix, write). */
int srh_mask_candidates(const uint8_t* mask, int32_t H, int32_t W, float threshold, int64_t* xy, uint8_t* scores,
                        int64_t capacity, int64_t* n, int32_t n_threads);

/* The last of the three nms_points calls of graph_extraction.py:130-139 together with the gathers in front of it: candidates =
 * [xy_a[ord_a]; xy_b[ord_b]] (keypoint, then road candidates, each in its np.argsort(scores)[::-1] order), visited in `order`
 * (np.argsort of the priorities [1]*na + [0]*nb, reversed — computed by the caller with numpy so that the tie order is the
 * reference's), no candidate forced; greedy radius suppression as srh_nms_points_host.  out_xy int64 [na+nb,2] receives the kept
 * points (x, y) in visiting order, *n_out their number. */
int srh_nms_merge_points(const int64_t* xy_a, const int64_t* ord_a, int64_t na, const int64_t* xy_b, const int64_t* ord_b, int64_t nb,
                         const int64_t* order, int32_t radius, int64_t* out_xy, int64_t* n_out);

/* Votes of one TopoNet batch in the reference's visiting order (inferencer.py:206-221: tile, source point, neighbour slot):
 * scores [nb, n_max, K] f32 host (NaN already replaced by -100), offsets [nb+1] rows of these tiles in ids / knn (the
 * srh_pass2_fill layout).  Appends keys[*count] = ids[src] * n_points + ids[tgt] and the float64 score, advances *count;
 * SRH_ERR_BAD_ARG if a valid pair's score is outside [0, 1] (the reference's assert, inferencer.py:219). */
int srh_pass2_votes(const float* scores, int32_t nb, int64_t n_max, int32_t K, const int64_t* offsets, const int64_t* ids,
                    const int32_t* knn, int64_t n_points, int64_t* keys, double* votes, int64_t capacity, int64_t* count);

/* Padded collate of one TopoNet batch (reference inferencer.py:179-185) from the flat arrays of srh_pass2_fill: tile b of the
 * batch owns rows offsets[b] .. offsets[b+1] of local [*,2] and knn [*,K]; writes points f32 [nb,n_max,2], pairs i32
 * [nb,n_max,K,2] (source row, target row — the source itself where invalid) and valid u8 [nb,n_max,K], zero beyond a tile's rows. */
int srh_pass2_pack(const int64_t* offsets, const int64_t* local, const int32_t* knn, int32_t nb, int64_t n_max, int32_t K,
                   float* points, int32_t* pairs, uint8_t* valid);
/* The unpadded collate for srh_toponet_ragged: rows keep their position in the flat query arrays (numbered from offsets[0]), pairs
 * index the flat list, point_tile[r] = the row's tile counted from the first tile of the range. */
int srh_pass2_pack_ragged(const int64_t* offsets, const int64_t* local, const int32_t* knn, int32_t n_tiles, int32_t K,
                          float* points, int32_t* pairs, uint8_t* valid, int32_t* point_tile);

/* Directed edge votes of pass 2 (reference inferencer.py:209-221: dict of score sums / counts keyed by (src, tgt), filled in
 * tile / point / slot order).  keys[i] = src * n_points + tgt, scores[i] in that visiting order.  Writes the unique keys in
 * ascending order with their float64 sums — accumulated in the reference's order, hence bit-identical to its loop — and
 * counts; out_first (nullable) receives the index i of each key's first vote, i.e. its insertion position in the reference's
 * dict, which fixes the order of the pred_edges list (inferencer.py:224-228); out arrays have capacity n. */
int srh_edge_vote_accumulate(const int64_t* keys, const double* scores, int64_t n, int64_t* out_keys, double* out_sums,
                             double* out_counts, int64_t* out_first, int64_t* n_unique);
/* The same with n_threads worker threads: the key space is cut into contiguous ranges of equal vote counts (one stable
 * partition pass), each sorted and accumulated by its own thread; identical outputs, bit for bit. */
int srh_edge_vote_accumulate_mt(const int64_t* keys, const double* scores, int64_t n, int64_t* out_keys, double* out_sums,
                                double* out_counts, int64_t* out_first, int64_t* n_unique, int32_t n_threads);

/* srh_pass2_votes + srh_edge_vote_accumulate in one pass over the query rows, without materialising the votes (reference
 * inferencer.py:206-228).  scores[b] = f32 [batch_nb[b], batch_n_max[b], K] host scores (NaN -> -100 done) of the tiles
 * batch_tile0[b] .. batch_tile0[b] + batch_nb[b] (indices into offsets [n_tiles+1]; ids / knn = the srh_pass2_fill layout; a tile
 * outside every batch must be empty).  Rows are grouped by source point and each point's votes are added, in the reference's
 * visiting order, into a table of its targets: same unique keys (ascending), float64 sums, counts and first-vote positions as the
 * two-call path, bit for bit.  capacity = number of non-negative knn entries always suffices.  SRH_ERR_BAD_ARG if a valid pair's
 * score is outside [0, 1] (the reference's assert, inferencer.py:219). */
int srh_pass2_vote_sums(const float* const* scores, const int32_t* batch_tile0, const int32_t* batch_nb, const int64_t* batch_n_max,
                        int32_t n_batches, int32_t K, const int64_t* offsets, int32_t n_tiles, const int64_t* ids, const int32_t* knn,
                        int64_t n_points, int64_t* out_keys, double* out_sums, double* out_counts, int64_t* out_first,
                        int64_t capacity, int64_t* n_unique, int32_t n_threads);

/* Edge list from the vote sums (reference inferencer.py:224-228): the (src, tgt) pairs whose mean score sums / max(counts, 1)
 * exceeds the threshold, in the insertion order of the reference's dict (ascending first-vote position; `first` values are
 * distinct).  out_edges int64 [n,2]; *n_edges their number. */
int srh_votes_to_edges(const int64_t* keys, const double* sums, const double* counts, const int64_t* first, int64_t n,
                       int64_t n_points, double threshold, int64_t* out_edges, int64_t* n_edges);

#ifdef __cplusplus
}
#endif
#endif /* SAMROAD_HIP_H */
