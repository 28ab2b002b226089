/*
 * uavgnn.h - C ABI of the MI355X-native hetero-GNN hot path (libuavgnn.so, gfx950 only).
 *
 * Drop-in boundary for the graph arithmetic that zhangxiaochen95/uav_bs_ctrl's MADRQN agent runs through DGL 0.9.0
 * (reference: algos/madrqn/agents/gnn_agents.py).  Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - All functions return 0 on success, a negative UAVGNN_E* code on argument errors, or the NEGATED hipError_t of a
 *     failed launch.  Nothing is thrown across this boundary.
 *   - All pointers are DEVICE pointers owned by the caller (PyTorch allocates every tensor incl. workspaces; the
 *     library allocates nothing and keeps no mutable global state; calls on distinct streams are independent).
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Launches are asynchronous and capturable
 *     into a hipGraph.
 *   - Matrices are row-major fp32; `ld*` is the leading dimension in elements; offsets/indices are int32.
 *   - Relation layout (the layout the reference's graph builder emits, algos/madrqn/utils/env_wrappers.py:69-89):
 *     edges of `seen`/`near` are grouped by destination and src id == edge id, so a relation is x_src[E,F] plus
 *     seg_off[N+1].  `talk` is given twice: CSC (in-edges per destination) and its transpose (out-edges per source,
 *     with the CSC position of each edge) so that the backward is a pure gather (deterministic, no float atomics).
 */
#ifndef UAVGNN_H
#define UAVGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVGNN_VERSION 100 /* 0.1.0 */

#define UAVGNN_EINVAL (-1000)      /* null pointer / non-positive size */
#define UAVGNN_EUNSUPPORTED (-1001) /* shape outside the compiled instantiations */
#define UAVGNN_EWORKSPACE (-1002)   /* workspace too small */

typedef void* uavgnn_stream_t;

int uavgnn_version(void);

/* Human-readable text for a return code of this library (static storage). */
const char* uavgnn_strerror(int code);

/* ------------------------------------------------------------------------------------------------------------------
 * K1  GATv2 relation, forward.
 * Replaces dglnn.GATv2Conv((F_src,F_dst), D, nh, residual=True, allow_zero_in_degree=True, activation=ReLU).forward
 * as constructed at gnn_agents.py:93-96 and called at :103-104 (and drqn/agents/gnn_agents.py:17-18,:27).
 *   el = W_s x_u + b_s ; er = W_d x_v + b_d ; e = attn . lrelu_slope(el+er) per head ; a = softmax over in-edges ;
 *   out[v] = ReLU( sum_u a el[u] + W_r x_v + b_r )            
 * x_src[E,F_src] (E = seg_off[N], also used to pick the schedule for sparse batches), x_dst[N,F_dst], seg_off[N+1]; W_s[H,F_src] b_s[H] W_d[H,F_dst] b_d[H] attn[H] W_r[H,F_dst]
 * b_r[H] (b_r may be NULL = zeros), H = nh*D.  out is written at out[v*ld_out + 0..H) (so two relations can share one
 * [N,2H] buffer = the th.cat of gnn_agents.py:106).  attn_save[E,nh] (may be NULL) receives the softmax weights for
 * the backward.  dst_order[N] (may be NULL = identity) is the order in which destinations are handed to the persistent
 * wavefronts; passing the destinations sorted by decreasing degree balances ragged batches (longest-first).  It only
 * affects scheduling, never results.  Supported: F_src in {2,4}, F_dst == 2, H <= 256, D in {8,16,32,64}.
 */
int uavgnn_gatv2_fwd(const float* x_src, int E, int F_src, const float* x_dst, int F_dst, const int32_t* seg_off,
                     const int32_t* dst_order, int N, const float* W_s, const float* b_s, const float* W_d, const float* b_d, const float* attn,
                     const float* W_r, const float* b_r, int nh, int D, float slope, float* out, int ld_out,
                     float* attn_save, uavgnn_stream_t stream);

/* uavgnn_gatv2_fwd picks the kernel: nh == 4, D in {16,32,64}: the low-degree kernel for two-feature relations whose
 * mean in-degree is <= 8 (`near`), else the fp32-MFMA row-tile kernel; any other (nh, D): the plain-VALU kernel.
 * Same contract, fixed kernel, exported as in-library A/B references: _valu = always the VALU kernel,
 * _mfma = the MFMA kernel whenever it is instantiated (never the low-degree kernel). */
int uavgnn_gatv2_fwd_valu(const float* x_src, int E, int F_src, const float* x_dst, int F_dst, const int32_t* seg_off,
                          const int32_t* dst_order, int N, const float* W_s, const float* b_s, const float* W_d, const float* b_d, const float* attn,
                          const float* W_r, const float* b_r, int nh, int D, float slope, float* out, int ld_out,
                          float* attn_save, uavgnn_stream_t stream);
int uavgnn_gatv2_fwd_mfma(const float* x_src, int E, int F_src, const float* x_dst, int F_dst, const int32_t* seg_off,
                          const int32_t* dst_order, int N, const float* W_s, const float* b_s, const float* W_d, const float* b_d, const float* attn,
                          const float* W_r, const float* b_r, int nh, int D, float slope, float* out, int ld_out,
                          float* attn_save, uavgnn_stream_t stream);

/* K1 forward, BOTH relations of GraphObservationEncoder.forward in ONE launch (reference: the two GATv2Conv calls and the
 * th.cat of algos/madrqn/agents/gnn_agents.py:103-106).  `seen` (x_gt [E_seen,4], seen_off, optional hand-out order
 * seen_order) fills out[:, 0:H), `near` (x_ubs [E_near,2], near_off) fills out[:, H:2H); ld_out >= 2H, 16-byte aligned.
 * seen_params / near_params: HOST arrays of 7 device pointers each, in DGL's GATv2Conv layout
 * {fc_src.weight, fc_src.bias, fc_dst.weight, fc_dst.bias, attn, res_fc.weight, res_fc.bias (may be NULL)}, 16-byte
 * aligned.  attn_save_* as in uavgnn_gatv2_fwd (NULL for inference).  `seen` runs on 16-edge MFMA row tiles per
 * destination, `near` packs two destinations per row tile with [x_u ; x_v] in the K = 4 contraction (any in-degree is
 * correct; <= 8 is one pass).  uavgnn_gatv2_hetero_supported: 1 when (F_seen, F_near, F_dst, nh, D) = (4, 2, 2, 4, 64),
 * else callers use uavgnn_gatv2_fwd per relation (same results up to summation order). */
int uavgnn_gatv2_hetero_supported(int F_seen, int F_near, int F_dst, int nh, int D);
int uavgnn_gatv2_hetero_fwd(const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order,
                            const float* x_ubs, int E_near, const int32_t* near_off, const float* x_dst, int N,
                            const float* const* seen_params, const float* const* near_params, int nh, int D, float slope,
                            float* out, int ld_out, float* attn_save_seen, float* attn_save_near,
                            uavgnn_stream_t stream);
/* Same contract with only some phases of the kernel executed (bit 0: `seen` row tiles of the non-isolated destinations,
 * bit 1: `near` + residual-only `seen` rows); phases = 3 is uavgnn_gatv2_hetero_fwd.  In-library ablation reference for
 * benchmarks - output rows of the skipped phase are left untouched.  Bit 8 (256): the score GEMM of both phases on fp32 MFMA
 * (csrc/gatv2_hetero_f32.hip) instead of the bf16 matrix cores with exact three-way operand splits - same results to fp32
 * rounding; the A/B reference and the strict-fp32 leg of bench.py. */
int uavgnn_gatv2_hetero_fwd_phases(const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order,
                                   const float* x_ubs, int E_near, const int32_t* near_off, const float* x_dst, int N,
                                   const float* const* seen_params, const float* const* near_params, int nh, int D,
                                   float slope, float* out, int ld_out, float* attn_save_seen, float* attn_save_near,
                                   int phases, uavgnn_stream_t stream);
/* Prepared parameters.  Every workgroup of the forward kernel first turns the parameters of the two modules into the image its
 * LDS holds (bf16 three-way splits of the weight tiles, scaled attention vectors, attn x W_s sums: one memory round trip + ~200
 * VALU instructions per wavefront = 2.9 us of a 20-us rollout launch).  A caller whose parameters stay put over many launches - a
 * rollout between two optimiser steps, the target network between two syncs - builds the image ONCE (uavgnn_gatv2_hetero_prepare,
 * one workgroup) into a buffer of uavgnn_gatv2_hetero_image_bytes() bytes (16-byte aligned) and launches with
 * uavgnn_gatv2_hetero_fwd_image, whose workgroups only copy it.  Results are bit-identical to uavgnn_gatv2_hetero_fwd_phases
 * with the same `phases`.  The image is a function of the parameter VALUES and `slope`: the caller must rebuild it after any
 * parameter change (the library cannot tell).  seen_params / near_params are still required (argument checks, fp32-MFMA route). */
size_t uavgnn_gatv2_hetero_image_bytes(void);
int uavgnn_gatv2_hetero_prepare(const float* const* seen_params, const float* const* near_params, int nh, int D, float slope,
                                void* image, uavgnn_stream_t stream);
int uavgnn_gatv2_hetero_fwd_image(const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order,
                                  const float* x_ubs, int E_near, const int32_t* near_off, const float* x_dst, int N,
                                  const float* const* seen_params, const float* const* near_params, int nh, int D,
                                  float slope, const void* image, float* out, int ld_out, float* attn_save_seen,
                                  float* attn_save_near, int phases, uavgnn_stream_t stream);
/* uavgnn_gatv2_hetero_fwd_image (image may be NULL: the in-kernel prologue) that ALSO writes rowmax_near / rowmax_seen [N]: the maximum
 * over the `near` / the `seen` half of every output row (rows are >= 0 behind the ReLU; every element has exactly one writer:
 * deterministic) - the two row bounds uavgnn_gemm_nt_h2 takes for the f_aggr product behind a TIME-BATCHED launch (gnn_agents.py:106).
 * Costs +9 % on a time-batched launch and +1.9 us on a 20-us rollout launch (why the rollout does not use it).  Not with the fp32-MFMA
 * build (phases bit 8): UAVGNN_EUNSUPPORTED. */
int uavgnn_gatv2_hetero_fwd_rowmax(const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order,
                                   const float* x_ubs, int E_near, const int32_t* near_off, const float* x_dst, int N,
                                   const float* const* seen_params, const float* const* near_params, int nh, int D, float slope,
                                   const void* image, float* out, int ld_out, float* attn_save_seen, float* attn_save_near,
                                   float* rowmax_near, float* rowmax_seen, int phases, uavgnn_stream_t stream);

/* K1 backward: parameter gradients only (observations are leaves: the reference never needs d/dx, Appendix A.4).
 * out / d_out are the forward output and its gradient (same ld).  Gradients are OVERWRITTEN.  Deterministic: per
 * workgroup partials in `workspace` are combined in a fixed order by a second launch (no float atomics).
 * workspace >= uavgnn_gatv2_bwd_workspace_bytes(F_src, nh*D). */
size_t uavgnn_gatv2_bwd_workspace_bytes(int F_src, int H);
int uavgnn_gatv2_bwd(const float* x_src, int E, int F_src, const float* x_dst, int F_dst, const int32_t* seg_off,
                     const int32_t* dst_order, int N, const float* W_s, const float* b_s, const float* W_d, const float* b_d, const float* attn,
                     int nh, int D, float slope, const float* out, const float* d_out, int ld_out,
                     const float* attn_save, float* dW_s, float* db_s, float* dW_d, float* db_d, float* dattn,
                     float* dW_r, float* db_r, void* workspace, size_t workspace_bytes, uavgnn_stream_t stream);
/* Same contract, generic kernel only (one destination per wavefront).  uavgnn_gatv2_bwd itself picks the
 * pair-of-destinations kernel for two-feature relations with mean in-degree <= 8 (`near`): in-library A/B reference. */
int uavgnn_gatv2_bwd_generic(const float* x_src, int E, int F_src, const float* x_dst, int F_dst, const int32_t* seg_off,
                     const int32_t* dst_order, int N, const float* W_s, const float* b_s, const float* W_d, const float* b_d, const float* attn,
                     int nh, int D, float slope, const float* out, const float* d_out, int ld_out,
                     const float* attn_save, float* dW_s, float* db_s, float* dW_d, float* db_d, float* dattn,
                     float* dW_r, float* db_r, void* workspace, size_t workspace_bytes, uavgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K3b  Targeted attention over the talk relation, forward.
 * Replaces g.apply_edges(fn.u_dot_v('s','q','e')); e/key_size; edge_softmax; update_all(u_mul_e('v','a'), sum)
 * (gnn_agents.py:261-267).   e_uv = <s_u, q_v> * scale ; a = softmax over in-edges of v ; c_v = sum_u a_uv v_u.
 * s,q: [N,K] rows with leading dims ld_s/ld_q, v: [N,M] ld_v (they may be column slices of one projection buffer).
 * talk_off[N+1], talk_src[E] = CSC.  c[N,M] (ld_c) is overwritten (zero for nodes without in-edges);
 * a_save[E] receives the attention weights in CSC order.  K <= 64, M <= 256.
 * With s == q == NULL the attention is uniform: c = mean of in-neighbour rows (the UDF reduce `mailbox.mean(1)` of
 * BaseComm/CommNet, gnn_agents.py:130-133,:214-216).
 * x_copy (may be NULL): when given, row d of x_copy[N, n_copy] (ld_x) is also copied to the n_copy floats IN FRONT of
 * c[d, :] (i.e. c - n_copy + d*ld_c), so that one pass leaves the GRU input [x || c] of gnn_agents.py:270 in place
 * of a separate th.cat.
 */
int uavgnn_talk_attn_fwd(const float* s, int ld_s, const float* q, int ld_q, const float* v, int ld_v, int K, int M,
                         const int32_t* talk_off, const int32_t* talk_src, int N, float scale, float* c, int ld_c,
                         float* a_save, const float* x_copy, int ld_x, int n_copy, uavgnn_stream_t stream);

/* K3b backward.  d_c[N,M] -> d_s, d_q [N,K], d_v [N,M] (overwritten; any of d_s/d_q may be NULL in uniform mode).
 * t_off[N+1], t_dst[E], t_pos[E]: transpose of the CSC (out-edges of each source, destination of each, and the CSC
 * position of the edge).  de_tmp[E] is scratch. */
int uavgnn_talk_attn_bwd(const float* s, int ld_s, const float* q, int ld_q, const float* v, int ld_v, int K, int M,
                         const int32_t* talk_off, const int32_t* talk_src, const int32_t* t_off, const int32_t* t_dst,
                         const int32_t* t_pos, int N, float scale, const float* a_save, const float* d_c, int ld_dc,
                         float* d_s, int ld_ds, float* d_q, int ld_dq, float* d_v, int ld_dv, float* de_tmp,
                         uavgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K4  GRU cell gate math (the pointwise half of nn.GRUCell: gnn_agents.py:29,:123,:164,:208,:246,:282).
 * gi = W_ih i + b_ih, gh = W_hh h + b_hh are [N,3H] (gate order r,z,n);  h' = (1-z) n + z h.
 */
int uavgnn_gru_gates_fwd(const float* gi, const float* gh, const float* h, int N, int H, float* h_out,
                         uavgnn_stream_t stream);
int uavgnn_gru_gates_bwd(const float* gi, const float* gh, const float* h, const float* d_hout, int N, int H,
                         float* d_gi, float* d_gh, float* d_h, uavgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K5  DiscreteComm message passing (gnn_agents.py:166-178,:189): per edge F.gumbel_softmax(logits[src].view(msg,2),
 * tau, hard=True), per destination the element-wise max over in-edges.  The Gumbel noise is either given explicitly
 * (gumbel [E, msg, 2], CSC order: fixtures carrying the reference's own draws) or, with gumbel == NULL, drawn inside the
 * kernel: Philox4x32-10 keyed by the 64-bit seed rng[0], counter (CSC position, channel, step rng[1]), g = -log(-log(u));
 * rng is a DEVICE array {seed, step} (a captured graph replays with the current step; the caller advances it).
 * uavgnn_gumbel_noise writes that same stream as an [E, msg, 2] tensor (tests).  logits[N, 2*msg] (ld) are the
 * per-SOURCE-node encoder outputs.  c[N, 2*msg] is overwritten (zeros for nodes without in-edges); y0_save[E, msg] and
 * sel[N, 2*msg] (CSC position that owns each channel's gradient: the first in-edge whose hard bit is set, else the first
 * in-edge) feed the backward.
 */
int uavgnn_disc_comm_fwd(const float* logits, int ld, const float* gumbel, const long long* rng, int msg,
                         const int32_t* talk_off, const int32_t* talk_src, int N, float inv_tau, float* c, int ld_c,
                         float* y0_save, int32_t* sel, uavgnn_stream_t stream);
int uavgnn_gumbel_noise(const long long* rng, long long E, int msg, float* out, uavgnn_stream_t stream);
/* d_c[N, 2*msg] -> d_logits[N, 2*msg] (overwritten), straight-through estimator; gather over the transposed CSC. */
int uavgnn_disc_comm_bwd(const float* d_c, int ld_dc, const float* y0_save, const int32_t* sel, int msg,
                         const int32_t* t_off, const int32_t* t_dst, const int32_t* t_pos, int N, float inv_tau,
                         float* d_logits, int ld_dl, uavgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * f1  Device-side observation -> graph construction for a batch of environments (replaces the Python loops of
 * env_wrappers.py:65-89 build_obs_graph, :139-154 build_comm_graph and dgl.batch/merge :67,:137).
 * Padded observations: gt[N, M, 1+Fg], ubs[N, U, 1+Fu] with column 0 the visibility flag (mubs_cov.py:215-242),
 * N = B*n agent rows; d_u2u[B, n, n] pairwise UBS distances.  Two passes with a prefix sum (caller) in between:
 *   uavgnn_obs_degrees  -> deg_seen[N], deg_near[N]                         (counts of kept rows)
 *   uavgnn_obs_compact  -> x_gt[E_seen, Fg], x_ubs[E_near, Fu] given seen_off / near_off = exclusive scans
 *   uavgnn_talk_degrees -> deg_in[N] (in-degree of every agent), env_edges[B] (edges per environment)
 *   uavgnn_talk_compact -> talk_src[E], talk_eid[E] (CSC; eid = the edge id the reference's i-major loop assigns)
 *                          given talk_off = scan(deg_in) and env_base = scan(env_edges).
 * Kept rows keep the reference's order (ascending m / ascending source i).  Fg == 4, Fu == 2, n <= 64.
 */
int uavgnn_obs_degrees(const float* gt, int M, int Fg, const float* ubs, int U, int Fu, int N, int32_t* deg_seen,
                       int32_t* deg_near, uavgnn_stream_t stream);
int uavgnn_obs_compact(const float* gt, int M, int Fg, const float* ubs, int U, int Fu, int N, const int32_t* seen_off,
                       const int32_t* near_off, float* x_gt, float* x_ubs, uavgnn_stream_t stream);
int uavgnn_talk_degrees(const float* d_u2u, int n, int B, float r_comm, int32_t* deg_in, int32_t* env_edges,
                        uavgnn_stream_t stream);
int uavgnn_talk_compact(const float* d_u2u, int n, int B, float r_comm, const int32_t* talk_off,
                        const int32_t* env_base, int32_t* talk_src, int32_t* talk_eid, uavgnn_stream_t stream);
/* The prefix sums between the two passes, up to four arrays in one launch: o_k[0] = 0, o_k[i + 1] = d_k[0] + ... + d_k[i]
 * (o_k has n_k + 1 entries; o_k == NULL skips the array).  Replaces torch.cumsum + fills in the device builder. */
int uavgnn_offsets_scan4(const int32_t* d0, int n0, int32_t* o0, const int32_t* d1, int n1, int32_t* o1, const int32_t* d2,
                         int n2, int32_t* o2, const int32_t* d3, int n3, int32_t* o3, uavgnn_stream_t stream);

/* ---- rollout: epsilon-greedy selection --------------------------------------------------------------------------
 * acts[a] = u_team[a / n_agents] <= eps ? min(floor(u_agent[a] * A), A - 1) : argmax_j q[a, j]   (first maximum).
 * MultiAgentQLearner.act, learner.py:73-80: one exploration draw per team of n_agents consecutive agents; u_team
 * [N / n_agents] and u_agent [N] are uniforms in [0, 1) supplied by the caller.  acts: int64 [N].
 */
int uavgnn_eps_greedy(const float* q, int ld_q, int N, int A, int n_agents, const float* u_team, const float* u_agent,
                      float eps, long long* acts, uavgnn_stream_t stream);
/* Same with the exploration rate read from DEVICE memory (*eps_dev) at execution time, so that a hipGraph captured around
 * the rollout step replays with the current epsilon of the schedule (algos/madrqn/run.py:60-61). */
int uavgnn_eps_greedy_dev(const float* q, int ld_q, int N, int A, int n_agents, const float* u_team, const float* u_agent,
                          const float* eps_dev, long long* acts, uavgnn_stream_t stream);

/* ---- bias gradients ---------------------------------------------------------------------------------------------
 * acc[s, :] += column sums of the rows [s*R, (s+1)*R) of x[N, C] (row stride ld, unit column stride), R = ceil(N / S).
 * The db = dY.sum(0) of the Linear / GRUCell backward (gnn_agents.py:43-46,:237-246 under learner.py:157), accumulated
 * across the BPTT steps in the caller's acc[S, C]; the caller folds the S partials once per update.  Deterministic.
 */
int uavgnn_colsum_acc(const float* x, long long ld, int N, int C, float* acc, int S, uavgnn_stream_t stream);
/* ---- dense layers on the f16 matrix cores, exactly scaled two-term splits (csrc/gemm_h2.hip, round 6) -----------------------------
 * Y = [X (K1 columns) || X2 (K - K1 columns; NULL: one source, K1 = K)] B^T (+ bias) (+ Y) (ReLU): the products of
 * uavgnn_gemm_nt_x3 / _cat with THREE f16 x f16 MFMA products per fp32 product instead of six bf16 ones (arithmetic, error and
 * non-finite behaviour: uavgnn_gru_cell_fwd_h2 above).  Replaces the input-gradient halves of the recurrent step and of f_aggr under
 * loss.backward() (learner.py:157 through gnn_agents.py:246, :99).
 *   rowmax / rowmax2  per row of the activation operand an upper bound of max |.| over the row (rowmax2 may be NULL; the larger of
 *                     the two is used), tight to within its power of two, from the kernels that produced the operand:
 *                     uavgnn_gru_gates_bwd_fused_sums_rowmax (d_gi / d_gh), uavgnn_relu_bwd_colsum_rowmax (masked gradient),
 *                     uavgnn_row_absmax (anything, as a pass of its own);
 *   planes            uavgnn_split_h2(W [R, C], transpose): f16 planes [2][n_out][K] + 2^-e per output row (uavgnn_split_h2_bytes(n_out,
 *                     K) bytes, 16-byte aligned), n_out / K = R / C (transpose = 0: y = x W^T) or C / R (transpose = 1: dx = dy W);
 *   epilogue          UAVGNN_GEMM_ACCUMULATE, UAVGNN_GEMM_RELU, UAVGNN_GEMM_STAGING_INTERLEAVED (the 256 x 128-tile kernel only:
 *                     UAVGNN_GEMM_TILE_* are UAVGNN_EUNSUPPORTED). */
int uavgnn_gemm_h2_supported(int M, int N, int K);
long long uavgnn_split_h2_bytes(int n_out, int K);
int uavgnn_split_h2(const float* W, int ld, int R, int C, int transpose, void* planes, uavgnn_stream_t stream);
int uavgnn_gemm_nt_h2(const float* X, int ldx, int K1, const float* X2, int ldx2, int M, int K, const float* rowmax, const float* rowmax2,
                      const void* planes, int N, const float* bias, float* Y, int ldy, int epilogue, uavgnn_stream_t stream);
/* uavgnn_gemm_nt_h2 for a second source WITHOUT a producer that bounds its rows (d_proj of the recurrent step: three kernels write its 96
 * columns): the launch takes the row maxima of X2 [M, K - K1] itself - every workgroup reads its 256 rows of X2 once more in front of its
 * first slice - uses max(rowmax, that) as the row's bound and writes the maxima to rowmax2_out [M] (Inf for a row that holds Inf / NaN:
 * what uavgnn_row_absmax(X2) returns, bit for bit; a maximum is order-independent).  X2 must not be NULL. */
int uavgnn_gemm_nt_h2_rm2(const float* X, int ldx, int K1, const float* X2, int ldx2, int M, int K, const float* rowmax, float* rowmax2_out,
                          const void* planes, int N, const float* bias, float* Y, int ldy, int epilogue, uavgnn_stream_t stream);
/* uavgnn_gru_gates_bwd_fused_sums / uavgnn_relu_bwd_colsum that ALSO write row_absmax [N] = max |.| over the rows of d_gi and d_gh /
 * of `out` (H = 256 / C = 256 only - one wavefront owns a row -, UAVGNN_EUNSUPPORTED otherwise; the second not in place). */
int uavgnn_gru_gates_bwd_fused_sums_rowmax(const float* pre, const float* h, const float* d_hout, const float* dq, int n_out,
                                           const float* W_out, int N, int H, float* d_gi, float* d_gh, float* d_h, float* col_sums,
                                           float* row_absmax, uavgnn_stream_t stream);
int uavgnn_relu_bwd_colsum_rowmax(const float* dy, long long ld, const float* y, long long ldy, float* out, long long ldo, int N, int C,
                                  float* acc, int S, float* row_absmax, uavgnn_stream_t stream);
/* The ReLU backward fused with the bias gradient of the Linear in front of it (reference: f_aggr = Sequential(Linear, ReLU),
 * gnn_agents.py:99-102, under learner.py:157): out [N, C] = dy where y > 0 else 0, acc[S, C] += row-blocked column sums of out
 * (as uavgnn_colsum_acc).  One pass over the gradient instead of autograd's threshold_backward + sum.  C % 4 == 0, strides % 4 == 0,
 * 16-byte aligned operands; `out` may be `dy` itself (same pointer and row stride: in place); any other
 * overlap of `out` with `dy` or `y` is UAVGNN_EINVAL. */
int uavgnn_relu_bwd_colsum(const float* dy, long long ld, const float* y, long long ldy, float* out, long long ldo, int N, int C,
                           float* acc, int S, uavgnn_stream_t stream);

/* ---- K3b, per-graph formulation ---------------------------------------------------------------------------------
 * Same contract and arithmetic as uavgnn_talk_attn_fwd / _bwd for a batch of B SMALL graphs (what dgl.batch of
 * per-environment graphs gives: common.py:45, env_wrappers.py:139-154): graph_off[B+1] = agent-node boundaries, every
 * graph has at most n_max <= 16 agents and at most n_max^2 talk edges, all of them inside the graph.  One wavefront
 * per graph, projections staged in LDS; the backward needs neither the transposed CSC nor a scratch array.  A graph
 * that violates the bounds gets NaN outputs.  uavgnn_talk_attn_env_supported(n_max, M, K) -> 1 when the shape fits
 * (K = 0 for uniform mode); otherwise use the per-destination entry points.
 */
int uavgnn_talk_attn_env_supported(int n_max, int M, int K);
int uavgnn_talk_attn_env_fwd(const float* s, int ld_s, const float* q, int ld_q, const float* v, int ld_v, int K, int M,
                             const int32_t* talk_off, const int32_t* talk_src, const int32_t* graph_off, int B,
                             int n_max, float scale, float* c, int ld_c, float* a_save, const float* x_copy, int ld_x,
                             int n_copy, uavgnn_stream_t stream);
int uavgnn_talk_attn_env_bwd(const float* s, int ld_s, const float* q, int ld_q, const float* v, int ld_v, int K, int M,
                             const int32_t* talk_off, const int32_t* talk_src, const int32_t* graph_off, int B,
                             int n_max, float scale, const float* a_save, const float* d_c, int ld_dc, float* d_s,
                             int ld_ds, float* d_q, int ld_dq, float* d_v, int ld_dv, uavgnn_stream_t stream);

/* ---- K3a + K3b in one launch (csrc/tarmac_msg.hip) ----------------------------------------------------------------
 * The TarMAC message of a batch of small graphs with a UNIFORM number of agents (gnn_agents.py:254-267):
 *   proj = [x || h] Wp^T + bp   (Wp = [f_val; f_sign; f_que].weight stacked: M value, K signature, K query columns),
 *   c_v = sum_{u -> v} softmax_u(<sign_u, query_v> * scale) value_u over `talk`.
 * Replaces the two projection GEMMs (ATen addmm behind gnn_agents.py:258-260) + uavgnn_talk_attn_env_fwd; the projections never
 * reach HBM on no-grad calls.  fp32 in / out; the projection GEMM runs on the bf16 matrix cores as six exact bf16 products per
 * fp32 product (csrc/bf16x3.h).
 *   uavgnn_tarmac_msg_supported  H % 32 == 0, M + 2K <= 128, K <= 64, n_ag in {1, 2, 4, 8, 16};
 *   uavgnn_tarmac_msg_prepare    the stacked weight [M + 2K, 2H] (row stride ld) -> bf16 plane tiles
 *                                (uavgnn_tarmac_msg_weight_bytes(H, M, K) bytes, 16-byte aligned); once per weight version;
 *   uavgnn_tarmac_msg_fwd        c_out [N, M] (row stride ld_c) always; training outputs, each optional (NULL): a_save [E]
 *                                attention weight per CSC position, proj_out [N, M + 2K], x_copy [N, H] (the x half of the GRU
 *                                input [x || c]); planes_out (NULL or uavgnn_tarmac_msg_planes_bytes(N, H, M) bytes): the GEMM
 *                                operand [x || c || h] of uavgnn_gru_cell_fwd_planes as bf16 planes in that kernel's tile order.
 * Preconditions: every graph has exactly n_ag agents (rows i n_ag .. (i + 1) n_ag - 1; N % n_ag == 0; 16 % n_ag == 0: no graph
 * straddles two 16-row tiles, the unit a wavefront works on); the rows of a tile have at most 256 in-edges, all from rows of the
 * same tile - a violating tile gets NaN messages, never a silent fallback. */
int uavgnn_tarmac_msg_supported(int H, int M, int K, int n_ag);
long long uavgnn_tarmac_msg_weight_bytes(int H, int M, int K);
long long uavgnn_tarmac_msg_planes_bytes(int N, int H, int M);
int uavgnn_tarmac_msg_prepare(const float* Wp, int ld, int H, int M, int K, void* tiles, uavgnn_stream_t stream);
int uavgnn_tarmac_msg_fwd(const float* x, int ld_x, const float* h, int ld_h, int N, int H, int n_ag, const void* tiles,
                          const float* bias, int M, int K, const int32_t* talk_off, const int32_t* talk_src, float scale,
                          float* c_out, int ld_c, float* a_save, float* proj_out, int ld_p, float* x_copy, int ld_xc,
                          void* planes_out, uavgnn_stream_t stream);
/* ... that also writes row_absmax [N] = max(|x_row|, |c_row|, |h_row|) per agent - the row scales of uavgnn_gru_cell_fwd_h2 behind it
 * (every element of x and h passes through this kernel anyway; a maximum is order-independent: deterministic).  Wavefront-pair
 * kernel only: M + 2K <= 96 (uavgnn_tarmac_msg_rowmax_supported), UAVGNN_EUNSUPPORTED otherwise. */
int uavgnn_tarmac_msg_rowmax_supported(int H, int M, int K, int n_ag);
int uavgnn_tarmac_msg_fwd_rowmax(const float* x, int ld_x, const float* h, int ld_h, int N, int H, int n_ag, const void* tiles,
                                 const float* bias, int M, int K, const int32_t* talk_off, const int32_t* talk_src, float scale,
                                 float* c_out, int ld_c, float* a_save, float* proj_out, int ld_p, float* x_copy, int ld_xc,
                                 float* row_absmax, uavgnn_stream_t stream);

/* ---- derived indexes of a batch --------------------------------------------------------------------------------
 * What the reference gets from DGL's lazy format materialisation (CSR/CSC created inside the first message-passing
 * call on every new graph - reached from gnn_agents.py:103-104, :264-267) is explicit here and deterministic.
 *
 * uavgnn_degree_order: order[N] = destinations of a segment relation sorted by decreasing in-degree
 *   (stable; degrees >= 255 share one bucket).  A scheduling hint for uavgnn_gatv2_fwd / _bwd (dst_order), never
 *   needed for correctness.  workspace: uavgnn_degree_order_workspace_bytes(N).
 * uavgnn_csc_transpose: out-edge lists of the talk CSC: t_off[N+1]; for slot k of source u, t_dst[k] = destination
 *   and t_pos[k] = position of that edge in the CSC, ascending within each source.  Consumed by
 *   uavgnn_talk_attn_bwd / uavgnn_disc_comm_bwd.  talk_src values must lie in [0, N).  Out-lists are sorted by one
 *   lane each (insertion sort): meant for the short lists of this domain (<= n_agents - 1 per source).
 *   workspace: uavgnn_csc_transpose_workspace_bytes(N).
 * uavgnn_csc_transpose_env: the same result for a batch of B small graphs (graph_off[B+1] = agent-node boundaries,
 *   graph_off[0] == 0, graph_off[B] == N) whose talk edges stay inside their graph - what dgl.batch of per-env
 *   graphs produces (common.py:45, env_wrappers.py:67).  One wavefront per graph, one launch, no workspace; cost per
 *   graph ~ ceil(agents / 64) x its edge count, so it is meant for graphs of up to a few hundred agents.
 */
size_t uavgnn_degree_order_workspace_bytes(int N);
int uavgnn_degree_order(const int32_t* seg_off, int N, int32_t* order, void* workspace, size_t workspace_bytes,
                        uavgnn_stream_t stream);
size_t uavgnn_csc_transpose_workspace_bytes(int N);
int uavgnn_csc_transpose(const int32_t* talk_off, const int32_t* talk_src, int N, int E, int32_t* t_off,
                         int32_t* t_dst, int32_t* t_pos, void* workspace, size_t workspace_bytes,
                         uavgnn_stream_t stream);
int uavgnn_csc_transpose_env(const int32_t* talk_off, const int32_t* talk_src, const int32_t* graph_off, int B, int N,
                             int32_t* t_off, int32_t* t_dst, int32_t* t_pos, uavgnn_stream_t stream);

/* Tail of the learner's update in one launch over FLAT fp32 buffers of n elements (reference:
 * algos/madrqn/learner.py:157-166): g[i] clamped to [-clip, clip] for i < n_clip (clip_grad_value_ on the policy network
 * only; clip <= 0: none; the clamped gradient is written back), torch.optim.AdamW step on (p, exp_avg, exp_avg_sq) with
 * decoupled weight decay, then p_targ <- polyak p_targ + (1 - polyak) p (p_targ may be NULL).  hyper: DEVICE array
 * {lr, t} (t = 1-based count of this step) so that a captured graph replays with current values.  The betas are
 * doubles and the bias corrections are formed in double, as torch.optim.AdamW does on the host.  16-byte aligned
 * buffers. */
int uavgnn_adamw_polyak(float* p, float* g, float* exp_avg, float* exp_avg_sq, float* p_targ, long long n,
                        long long n_clip, const float* hyper, double beta1, double beta2, float eps, float weight_decay,
                        float clip, float polyak, uavgnn_stream_t stream);

/* Batched simulator step of the multi-UBS coverage environment, B independent environments per launch (reference:
 * envs/mubs_cov/mubs_cov.py:104-129 step, :131-210 _transmit_data, :212-242 get_obs, :278-296 get_state, :324-341
 * _get_reward; envs/common.py:19-25 Jain index, :49-59 air-to-ground channel gain).  actions [B, n] int64 or NULL (the
 * reset-time transmission: no move, t stays as it is).  In/out per environment: pos_ubs [B,n,2] f64, prior [B,M] (GT
 * priorities used by this step -> priorities for the next one), avg_rate [B,M], t [B], run_f32 [B,4] = {total throughput,
 * average global utility, Jain index, global utility}, n_colls [B].  Outputs: d_u2g [B,n,M], d_u2u [B,n,n], gt_ubs /
 * gt_rb [B,M] (serving UBS / resource block per GT, -1 = unserved), rate_per_gt [B,M], rate_per_ubs [B,n] f64,
 * mask_collision [B,n], reward [B,n] f64, done [B], padded observations obs_gt [B,n,M,5|4], obs_ubs [B,n,n-1,3],
 * obs_agent [B,n,2] (column 0 = visibility flag: the input format of uavgnn_obs_degrees / _compact) and the global state
 * [B, uavgnn_env_state_dim] (may be NULL).  int_consts / f64_consts: HOST arrays, layout in csrc/env_sim.hip. */
int uavgnn_env_state_dim(int n_ubs, int n_gts, int fair_service);
int uavgnn_env_step(const int32_t* int_consts, const double* f64_consts, int B, const long long* actions,
                    const double* avail_moves, double* pos_ubs, const float* pos_gts, int32_t* prior, float* avg_rate,
                    int32_t* t, float* run_f32, double* n_colls, float* d_u2g, float* d_u2u, int32_t* gt_ubs,
                    int32_t* gt_rb, float* rate_per_gt, double* rate_per_ubs, int32_t* mask_collision, double* reward,
                    float* done, float* obs_gt, float* obs_ubs, float* obs_agent, float* state, uavgnn_stream_t stream);

/* The four construction passes above + the three prefix sums in ONE launch for small batches (B n <=
 * uavgnn_build_graph_small_max_agents() = 4096 agents; reference: env_wrappers.py:65-89,:122-154 for B environments): seen_off /
 * near_off / talk_off [B n + 1], graph_off [B + 1] and the compacted x_gt / x_ubs / talk_src / talk_eid (allocated by the
 * caller at capacity B n M / B n U / B n n rows; the first E rows are written).  d_u2u NULL: no talk relation (talk_off, when
 * given, is zero-filled).  Bit-identical to the multi-launch path. */
int uavgnn_build_graph_small_max_agents(void);
int uavgnn_build_graph_small(const float* gt, int M, int Fg, const float* ubs, int U, int Fu, const float* d_u2u, int n, int B,
                             float r_comm, int32_t* seen_off, int32_t* near_off, int32_t* talk_off, float* x_gt, float* x_ubs,
                             int32_t* talk_src, int32_t* talk_eid, int32_t* graph_off, uavgnn_stream_t stream);

/* K4, the whole GRU cell in one launch (reference: nn.GRUCell at algos/madrqn/agents/gnn_agents.py:29,:123,:164,:208,
 * :246,:282; SURVEY 2.2 "gru_cell_fused"): h_out = GRUCell(inp [N, K_in] (leading dimension ld_inp), h [N, H]) with PyTorch's
 * parameter layout W_ih [3H, K_in], W_hh [3H, H], b_ih / b_hh [3H] (gate order r, z, n).  Both GEMMs run on fp32 MFMA into
 * shared r / z accumulators and separate gi_n / gh_n accumulators, gates in the epilogue: the [N, 3H] pre-activations never
 * reach HBM.  pre_save (may be NULL): [N, 4H] = r_pre | z_pre | gi_n | gh_n (biases included) for
 * uavgnn_gru_gates_bwd_fused, which produces the same d_gi / d_gh [N, 3H] and d_h [N, H] as uavgnn_gru_gates_bwd.
 * uavgnn_gru_cell_supported: K_in and H multiples of 32 (else: vendor GEMMs + uavgnn_gru_gates_fwd). */
int uavgnn_gru_cell_supported(int K_in, int H);
int uavgnn_gru_cell_fwd(const float* inp, int ld_inp, int K_in, const float* h, int N, int H, const float* W_ih,
                        const float* b_ih, const float* W_hh, const float* b_hh, float* h_out, float* pre_save,
                        uavgnn_stream_t stream);
/* The same cell on the bf16 matrix cores (csrc/gru_x3.hip, csrc/bf16x3.h): every fp32 operand is split exactly into three
 * bf16 terms and each fp32 product is accumulated as six bf16 x bf16 MFMA products in fp32 (error below an fp32 GEMM's,
 * profiles/r02_ubench_gemm_bf16x3.txt; fp32 MFMA issues at 1/16 of the bf16 MFMA rate on gfx950).
 * uavgnn_gru_split_weights writes the bf16 planes of W_ih then W_hh ([3][3H][K_in] | [3][3H][H], ..._workspace_bytes bytes,
 * 16-byte aligned) - call it whenever the weights may have changed; uavgnn_gru_cell_fwd_x3 has the contract of
 * uavgnn_gru_cell_fwd with `planes` in place of the two weight matrices; 4 N max(ld_inp, H) must stay below 2^32 (32-bit byte
 * offsets inside the kernel: 3.3 M rows at K_in = 320), UAVGNN_EUNSUPPORTED otherwise. */
int uavgnn_gru_cell_x3_supported(int K_in, int H);   /* K_in % 32 == 0 and H % 64 == 0 */
long long uavgnn_gru_cell_x3_workspace_bytes(int K_in, int H);
int uavgnn_gru_split_weights(const float* W_ih, int K_in, const float* W_hh, int H, void* planes, uavgnn_stream_t stream);
int uavgnn_gru_cell_fwd_x3(const float* inp, int ld_inp, int K_in, const float* h, int N, int H, const void* planes,
                           const float* b_ih, const float* b_hh, float* h_out, float* pre_save, uavgnn_stream_t stream);
/* The same call with the cell's input given as TWO pieces [inp (K1 columns) || inp2 (K2 columns)] - the TarMAC step's
 * th.cat([x, c]) (gnn_agents.py:268-270) without the concatenated copy; K1, K2 multiples of 32 (K2 = 0: one piece), `planes`
 * built for K_in = K1 + K2. */
int uavgnn_gru_cell_fwd_x3_cat(const float* inp, int ld_inp, int K1, const float* inp2, int ld_inp2, int K2, const float* h,
                               int N, int H, const void* planes, const float* b_ih, const float* b_hh, float* h_out,
                               float* pre_save, uavgnn_stream_t stream);
/* ... with a per-call variant word (0 = uavgnn_gru_cell_fwd_x3_cat).  UAVGNN_GRU_STAGING_BLOCKS: the staging of a K slice as a
 * block in front of / behind its first MFMA group (the round-2 schedule; the A/B reference of tools/gru_probe.py) instead of
 * interleaved with it - bit-identical results.  A call argument, not process state: the library keeps no mutable globals. */
#define UAVGNN_GRU_STAGING_BLOCKS 1
int uavgnn_gru_cell_fwd_x3_opts(const float* inp, int ld_inp, int K1, const float* inp2, int ld_inp2, int K2, const float* h,
                                int N, int H, const void* planes, const float* b_ih, const float* b_hh, float* h_out,
                                float* pre_save, int flags, uavgnn_stream_t stream);
/* The same cell with HALF the matrix-core work (csrc/gru_h2.hip, round 6): "f16x2" - every row of the activation operand
 * [inp || inp2 || h] and every output unit's row of [W_ih | W_hh] is scaled by a power of two that puts its largest magnitude into
 * [2^14, 2^15) (exact), the scaled value is split as hi = rn_f16(v), lo = rn_f16(v - hi) (hi + lo = v to <= 2^-23 |v|), and an fp32
 * product is the fp32-accumulated sum of THREE f16 x f16 MFMA products (hi lo + lo hi + hi hi; the dropped lo lo <= 2^-22 |a b|),
 * un-scaled exactly in the epilogue.  fp32 in / out / accumulate; measured error against float64 at or below the vendor fp32 GEMM's
 * and the bf16x3 cell's (profiles/r06_h2_error_tables.txt).  Replaces nn.GRUCell at gnn_agents.py:246 exactly like
 * uavgnn_gru_cell_fwd_x3_cat.
 *   row_absmax [N]   an upper bound of max |.| over every row of [inp || inp2 || h], tight to within its power of two:
 *                    uavgnn_tarmac_msg_fwd_rowmax writes it on its way (the producer of `c`), uavgnn_row_absmax computes it in a pass
 *                    of its own.  A bound that is too SMALL overflows f16: the row's outputs are NaN, never a plausible number;
 *   planes           uavgnn_gru_split_weights_h2: f16 planes [2][3H][K_in] | [2][3H][H] | 2^-e per weight row (float [3H]),
 *                    uavgnn_gru_cell_h2_workspace_bytes bytes, 16-byte aligned; rebuilt whenever the weights may have changed.
 * Non-finite operands follow the bf16x3 kernels' contract (a row that holds Inf / NaN yields NaN outputs); rows whose largest
 * magnitude is below 2^-112 lose relative precision (the scale exponent is clamped to 126).  Same shape limits as the x3 cell. */
int uavgnn_gru_cell_h2_supported(int K_in, int H);   /* K_in % 32 == 0 and H % 64 == 0 */
long long uavgnn_gru_cell_h2_workspace_bytes(int K_in, int H);
int uavgnn_gru_split_weights_h2(const float* W_ih, int K_in, const float* W_hh, int H, void* planes, uavgnn_stream_t stream);
int uavgnn_gru_cell_fwd_h2(const float* inp, int ld_inp, int K1, const float* inp2, int ld_inp2, int K2, const float* h, int N, int H,
                           const float* row_absmax, const void* planes, const float* b_ih, const float* b_hh, float* h_out,
                           float* pre_save, uavgnn_stream_t stream);
/* out [N] = max |.| per row over up to three row-major pieces (a2 / a3 may be NULL); Inf when the row holds Inf or NaN. */
int uavgnn_row_absmax(const float* a1, int ld1, int K1, const float* a2, int ld2, int K2, const float* a3, int ld3, int K3, int N,
                      float* out, uavgnn_stream_t stream);
/* The Q head for n_actions <= 16 (csrc/head.hip; reference: nn.Linear(H, n_actions) at algos/madrqn/agents/gnn_agents.py:43-46,:56):
 * q [N, A] (row stride ld_q) = h [N, H] (row stride ld_h) W [A, H]^T (row stride ld_w) + b [A].  fp32 in / out, exact fp32 products
 * on the matrix cores (v_mfma_f32_16x16x4_f32), one pass over h.  H in {64, 128, 256}, A <= 16, ld_h % 4 == 0, ld_w % 4 == 0, h and W
 * 16-byte aligned (UAVGNN_EUNSUPPORTED otherwise). */
int uavgnn_head_supported(int H, int A);
int uavgnn_head_fwd(const float* h, int ld_h, int N, int H, const float* W, int ld_w, const float* b, int A, float* q, int ld_q,
                    uavgnn_stream_t stream);
/* Dense layers on the bf16 matrix cores (csrc/gemm_x3.hip; reference: the nn.Linear layers of
 * algos/madrqn/agents/gnn_agents.py - f_aggr :101-102, :106, TarMAC projections :227-236 - and the input-gradient GEMMs of
 * loss.backward(), learner.py:157): Y[M, N] = X[M, K] B[N, K]^T (+ bias[N]) (+ Y) (then ReLU), fp32 in / out, each fp32 product
 * as six exact bf16 products (bf16x3.h).  uavgnn_split_bf16x3 turns a weight matrix W [R, C] (row stride ld) into the bf16
 * planes the GEMM reads: [3][R][C] (transpose = 0: forward, B = W) or [3][C][R] (transpose = 1: input gradient, B = W^T);
 * 6 R C bytes, 16-byte aligned.  K % 32 == 0, ldx % 4 == 0, X 16-byte aligned; M, N arbitrary. */
#define UAVGNN_GEMM_ACCUMULATE 1
#define UAVGNN_GEMM_RELU 2
/* kernel variants, selected per call by further bits of `epilogue` (A/B references of tools/gemm_x3_probe.py; the two eight-wave
 * variants agree bit for bit, so do the two four-wave ones - the pairs differ in accumulation order, inside the same error
 * bound): default = 256 x 128 tiles, eight waves, double-buffered LDS, staging of a slice as a block;
 * STAGING_INTERLEAVED = the same with the staging interleaved with the MFMAs (faster per launch, not per power-limited cycle);
 * TILE_128 = 128 x 128 tiles, four waves; TILE_64 = 64 x 128 tiles, four waves (batches of a few thousand rows: twice the workgroups) */
#define UAVGNN_GEMM_STAGING_INTERLEAVED 4
#define UAVGNN_GEMM_TILE_128 8
#define UAVGNN_GEMM_TILE_64 16
int uavgnn_gemm_x3_supported(int M, int N, int K);
int uavgnn_split_bf16x3(const float* W, int ld, int R, int C, int transpose, void* planes, uavgnn_stream_t stream);
int uavgnn_gemm_nt_x3(const float* X, int ldx, int M, int K, const void* planes, int N, const float* bias, float* Y, int ldy,
                      int epilogue, uavgnn_stream_t stream);
/* ... with the contraction over TWO buffers, [X (K1 columns) || X2 (K - K1 columns)], without a concatenated copy (K1 % 32 == 0,
 * ldx2 % 4 == 0, X2 16-byte aligned; default tiles only - UAVGNN_EUNSUPPORTED with TILE_128 / TILE_64): the GRU backward's
 * d x = d_gi W_ih[:, :H] + d_proj Wp[:, :H] (learner.py:157 through gnn_agents.py:246,:258-260) as one product over K = 3H + M + 2K;
 * `planes` = uavgnn_split_bf16x3 of the stacked weight. */
int uavgnn_gemm_nt_x3_cat(const float* X, int ldx, int K1, const float* X2, int ldx2, int M, int K, const void* planes, int N,
                          const float* bias, float* Y, int ldy, int epilogue, uavgnn_stream_t stream);
/* uavgnn_gemm_nt_x3 (the 256 x 128-tile kernel only: UAVGNN_GEMM_TILE_* are UAVGNN_EUNSUPPORTED) that ALSO writes rowmax_out [M] = max |.|
 * over every row of X, a by-product of its staging: the bound of an f16x2 product that reads the same operand later (the layer's weight
 * gradient, uavgnn_gemm_tn_h2) at no extra traffic. */
int uavgnn_gemm_nt_x3_rowmax(const float* X, int ldx, int M, int K, const void* planes, int N, const float* bias, float* Y, int ldy,
                             int epilogue, float* rowmax_out, uavgnn_stream_t stream);

/* Weight gradient of a dense layer on the bf16x3 arithmetic (csrc/gemm_tn_x3.hip): partials[s][Mo, Ko] (+)= dY[rows_s, :Mo]^T
 * X[rows_s, :Ko] for the S contiguous row chunks rows_s of the n_rows rows (chunk = ceil(n_rows / S) rounded up to 32 rows);
 * the caller sums the S partial products in a fixed order.  dY [n_rows, >= Mo] (row stride ldy), X [n_rows, >= Ko] (row
 * stride ldx), fp32, unit inner stride, any 4-byte alignment; partials [S, Mo, Ko] fp32, fully overwritten unless
 * `accumulate`.  Replaces autograd's dW = dy^T x of nn.Linear / nn.GRUCell (gnn_agents.py:99, :243-246, :43-46 under
 * learner.py:157).  uavgnn_gemm_tn_x3_chunks: the S this library would pick (output tiles x S fills the chip). */
int uavgnn_gemm_tn_x3_chunks(long long n_rows, int Mo, int Ko);
int uavgnn_gemm_tn_x3(const float* dY, int ldy, int Mo, const float* X, int ldx, int Ko, long long n_rows, float* partials,
                      int S, int accumulate, uavgnn_stream_t stream);
/* The same weight gradient on the f16x2 arithmetic (csrc/gemm_tn_h2.hip, round 6; arithmetic and error: uavgnn_gru_cell_fwd_h2): the
 * contraction runs over the ROWS of both operands, so the power-of-two scales of the two-term split are per COLUMN.
 *   colmax_y [Mo] / colmax_x [Ko]  upper bounds of max |.| over every column of dY / X, tight to within their power of two:
 *                                  uavgnn_col_absmax (one pass over the operand; Inf for a column that holds Inf / NaN -> NaN outputs);
 *   partials [S][Mo][Ko]           one partial product per row chunk (the caller sums them in a fixed order), `accumulate` adds in place.
 * n_rows % 32 == 0, Mo % 4 == 0, Ko % 4 == 0, 16-byte aligned rows (uavgnn_gemm_tn_h2_supported; UAVGNN_EUNSUPPORTED otherwise).
 * Replaces the vendor's fp32 split-K GEMM for dW_ih / dW_hh of nn.GRUCell (gnn_agents.py:246) and csrc/gemm_tn_x3.hip for f_aggr
 * (gnn_agents.py:99) under loss.backward() (learner.py:157) over the time-batched rows of a BPTT sequence. */
int uavgnn_col_absmax(const float* x, long long ld, long long n, int C, float* out, uavgnn_stream_t stream);
int uavgnn_gemm_tn_h2_supported(long long n_rows, int Mo, int Ko);
int uavgnn_gemm_tn_h2_chunks(long long n_rows, int Mo, int Ko);
int uavgnn_gemm_tn_h2(const float* dY, long long ldy, int Mo, const float* X, long long ldx, int Ko, long long n_rows,
                      const float* colmax_y, const float* colmax_x, float* partials, int S, int accumulate, uavgnn_stream_t stream);
int uavgnn_gru_gates_bwd_fused(const float* pre, const float* h, const float* d_hout, int N, int H, float* d_gi, float* d_gh,
                               float* d_h, uavgnn_stream_t stream);
/* The same with the Q head's input gradient folded in: the gradient of h' is d_hout (NULL = 0) + dq W_out, dq [N, n_out] the
 * gradient of q = h' W_out^T + b_out (reference: algos/madrqn/agents/gnn_agents.py:56 `self.f_out(h)` on the GRU output), W_out
 * [n_out, H] row-major and 16-byte aligned, n_out <= 64.  Replaces `d_hout + dq @ W_out` followed by the call above: the [N, H]
 * sum never reaches HBM.  UAVGNN_EUNSUPPORTED for H % 4 != 0, n_out > 64 or an unaligned W_out. */
int uavgnn_gru_gates_bwd_fused_head(const float* pre, const float* h, const float* d_hout, const float* dq, int n_out,
                                    const float* W_out, int N, int H, float* d_gi, float* d_gh, float* d_h,
                                    uavgnn_stream_t stream);
/* ... which also writes the launch's share of the cell's BIAS gradients: col_sums [uavgnn_gru_gates_bwd_sum_rows(N, H)][4 H] (16-byte
 * aligned) holds per-workgroup column sums d_r | d_z | d_n (input side) | d_n (hidden side); db_ih = the sum over its rows of columns
 * [0 : 3H], db_hh = of columns [0 : 2H] and [3H : 4H] (fixed summation order: deterministic).  dq / W_out may both be NULL (no head
 * term; d_hout is then required).  Replaces autograd's db = sum over rows of d_gi / d_gh (nn.GRUCell under learner.py:157), i.e. a
 * second pass over the [N, 3H] gate gradients.  uavgnn_gru_gates_bwd_sum_rows returns 0 when H / 4 does not divide 256. */
int uavgnn_gru_gates_bwd_sum_rows(int N, int H);
int uavgnn_gru_gates_bwd_fused_sums(const float* pre, const float* h, const float* d_hout, const float* dq, int n_out,
                                    const float* W_out, int N, int H, float* d_gi, float* d_gh, float* d_h, float* col_sums,
                                    uavgnn_stream_t stream);

/* Backward of the fused GRU cell as ONE call (csrc/gru_bwd.hip; reference: autograd of nn.GRUCell at gnn_agents.py:246,:270
 * under learner.py:157): gate gradients from the pre-activation sets saved by the forward (pre [N, 4H]), d_inp [N, K_in]
 * (leading dimension ld_dinp) = d_gi W_ih, d_h [N, H] = d_hout * z + d_gh W_hh, both GEMMs on the bf16x3 arithmetic.
 * d_gi / d_gh [N, 3H] are OUTPUTS kept for the caller: the weight / bias gradients (dW_ih = d_gi^T inp, dW_hh = d_gh^T h, db_ih =
 * colsum d_gi, db_hh = [colsum d_gi[:, :2H] | colsum d_gh[:, 2H:]]) are sums over agents AND time steps, which a BPTT caller
 * forms once per sequence (uavgnn_gemm_tn_x3 / uavgnn_colsum_acc over the stacked operands).  planes_bwd: bf16 planes of
 * W_ih^T then W_hh^T written by uavgnn_gru_split_weights_bwd (uavgnn_gru_cell_bwd_workspace_bytes bytes, 16-byte aligned);
 * 3H % 32 == 0, pre / d_gi / d_gh / d_inp 16-byte aligned, ld_dinp % 4 == 0. */
long long uavgnn_gru_cell_bwd_workspace_bytes(int K_in, int H);
int uavgnn_gru_split_weights_bwd(const float* W_ih, int K_in, const float* W_hh, int H, void* planes, uavgnn_stream_t stream);
int uavgnn_gru_cell_bwd(const float* pre, const float* h, const float* d_hout, int N, int K_in, int H, const void* planes_bwd,
                        float* d_gi, float* d_gh, float* d_inp, int ld_dinp, float* d_h, uavgnn_stream_t stream);

/* ONE query for every caller-provided scratch / plane buffer (bytes; 0 = unknown kind or bad sizes).  Arguments per kind:
 *   GATV2_BWD (F_src, H)            partial rows of uavgnn_gatv2_bwd          DEGREE_ORDER (N)   CSC_TRANSPOSE (N)
 *   GRU_PLANES (K_in, H)            uavgnn_gru_split_weights                  GRU_BWD_PLANES (K_in, H)  uavgnn_gru_split_weights_bwd
 *   GEMM_PLANES (R, C)              uavgnn_split_bf16x3                       GEMM_TN_PARTIALS (n_rows, Mo, Ko)  uavgnn_gemm_tn_x3
 *   K1_IMAGE ()                     uavgnn_gatv2_hetero_prepare */
#define UAVGNN_WS_GATV2_BWD 1
#define UAVGNN_WS_DEGREE_ORDER 2
#define UAVGNN_WS_CSC_TRANSPOSE 3
#define UAVGNN_WS_GRU_PLANES 4
#define UAVGNN_WS_GRU_BWD_PLANES 5
#define UAVGNN_WS_GEMM_PLANES 6
#define UAVGNN_WS_GEMM_TN_PARTIALS 7
#define UAVGNN_WS_K1_IMAGE 8
long long uavgnn_workspace_bytes(int kind, long long a, long long b, long long c);

#ifdef __cplusplus
}
#endif
#endif /* UAVGNN_H */
