// Internal interface between the C ABI (mppi_capi.hip) and the gfx950 kernels (mppi_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mppi {

constexpr int MODEL_DIFF = 0, MODEL_RACE = 1;
constexpr int OBS_NONE = 0, OBS_CIRCLE = 1, OBS_OUTLINE = 2;
constexpr int FILTER_DIFF = 0, FILTER_RACE = 1, FILTER_NONE = 2, FILTER_TORCH = 3;
constexpr int NO_TRIGGER = 0x7fffffff;
constexpr int STATUS_DONE = 0, STATUS_NEED_ROUND = 1, STATUS_PATH_END = 2, STATUS_EXCHANGE_FAILED = 3;

// Peer-to-peer exchange buffer of one rank (mppi_comm_*), fine-grained device memory every rank maps:
// two slots (iteration parity), each {long long flag[XCHG_MAX_RANKS]; double rec[nranks][xchg_rec_len(T)]}.
// Rank r owns flag[r] / rec[r] of every buffer.
constexpr int XCHG_MAX_RANKS = 64;
constexpr int XCHG_LDS_RANKS = 16;  // up to this many ranks' records are staged in LDS before they are merged
__host__ __device__ inline int xchg_rec_len(int T) { return (3 + 2 * T + 1) & ~1; }
__host__ __device__ inline size_t xchg_slot_bytes(int T, int nranks) {
    return sizeof(long long) * XCHG_MAX_RANKS + sizeof(double) * (size_t)nranks * xchg_rec_len(T);
}

// Sequential waypoint index resolved in ONE launch (mppi_differential_drive.py:201-249: one `prev_way_point_idx`
// threaded through all K (T+1) cost calls in k-major order, each call `p <- first nearest waypoint in [p, p+W)`).
// Where a call's distances to the candidates behind c first fall strictly and then never fall again -- a position
// beside a path that does not fold back within the candidates -- its first minimum over ALL of them is m = the number
// of descents, and the search entered at any p returns max(p, m) (p <= m: m is the first minimum of [p, p+W) as long
// as m - p < W; p > m: nothing behind p is smaller).  The threaded index is then a running maximum of the calls' m in
// call order: a wave reduction per sample, a row reduction per workgroup, and across workgroups a decoupled look-back
// on one word per workgroup (fused_lookback) -- every sample is priced once, with its exact index, in the one rollout
// launch, and k_finalize only reads the largest offset.  LB_CAND candidates are examined: exact while every realised
// index p keeps [p, p+W) inside them (p <= LB_CAND - W, or the path ends inside them) and below W; a call that is not
// unimodal, an index beyond that reach or a look-back that times out sets the `bad` bit of the workgroup's word, and
// k_finalize hands the iteration to the speculation rounds (exact either way).
// Word of workgroup b: [31:8] the launch pair's tag, [7] bad, [6:0] largest offset of the workgroup's own calls.
constexpr int HYP_WINDOW = 20, HYP_WINDOW_CUDA = 10;  // the search windows of the reference's files
constexpr int LB_CAND = 32, LB_BAD = 0x80;
constexpr unsigned LB_TAG_MASK = 0xffffff00u;
constexpr unsigned long long LB_TIMEOUT_TICKS = 20000ull;  // 200 us at 100 MHz
constexpr int HYP_MAX_BLOCKS = 512;  // workgroups whose words one wave reads (four per lane and 16-byte load)
constexpr int LB_COPIES = 8, LB_COPY_STRIDE = 2048;  // copies of the words and the distance between them (in words: 8 KB)
// (the tag: a sequence number the host draws per rollout / finalize launch pair -- KParams::lb_seq, never 0 mod 2^24 --, so
// that no word of an earlier launch can be taken for this one's, whatever the caller does to the iteration counter)
__host__ __device__ inline unsigned lb_tag(unsigned seq) { return (seq & 0xffffffu) << 8; }

// Controller state that lives on the device (so closed loops need no host round trip).
struct DevState {
    double x0[4];      // observed state of the current iteration
    int p;             // prev_way_point_idx at the iteration boundary
    int c;             // waypoint index the pending rollouts start from
    int k_start;       // first sample whose cost is not final yet (sequential mode rounds)
    int first_k;       // atomicMin: first sample that moved the waypoint index this round
    int round;         // speculation rounds used by the current iteration
    int path_end;      // x0's nearest waypoint is the last one
    int idx_start;     // c right after the x0 call
    int pad;
    long long iter;    // completed iterations == sampler counter
};

// What the host reads back after a step (followed by 2*T doubles: the returned u).
struct StepResult {
    int status, k_next, c_next, idx_start;
    int idx_after, path_end, rounds;
    int costs_hyp;  // the iteration's sequential index was resolved in its one rollout launch (fused_lookback)
    int n_collided, pad_res;  // samples of this handle whose cost carries a collision penalty (-1: records of other ranks merged)
    long long iter;
    double rho, eta, ess;
    double u0[2];
    double x_next[4];
    long long seq;  // written LAST (after a system-scope fence) when the host polls mapped memory for completion
};

// Per-rank softmin partial of the split step (ABI layout): {rho, eta, eta2, W[T][2]} in doubles.
__host__ __device__ inline int partial_len(int T) { return 3 + 2 * T; }
// Per-block record (internal layout, handle precision): {rho, eta, eta2, pad, W[2T] padded to 16 bytes}.
__host__ __device__ inline int record_len(int T, int elem_bytes) {
    const int vw = 16 / elem_bytes;
    return 4 + ((2 * T + vw - 1) / vw) * vw;
}

// LDS of the merge code (k_merge / k_finalize), in elements of the handle's precision: the weighted noise in the
// filter's padded layout [2 (T + W + 1)], the updated controls [2T], 64 record scales per wave and window, block
// reductions, per-group partial sums.  Every region starts on a 16-byte boundary.
constexpr int MERGE_THREADS = 256;
constexpr int MERGE_MAX_RECORDS = 256;  // per window
constexpr int MERGE_MAX_WINDOWS = 2;    // k_finalize takes up to 512 records itself (K = 16384 in the dual layout)
__host__ __device__ inline size_t merge_lds_elems(int T, int W, size_t elem, int nt = MERGE_THREADS) {
    const size_t r4 = 3, nw = (2 * (size_t)(T + W + 1) + r4) & ~r4, nu = (2 * (size_t)T + r4) & ~r4;
    return nw + nu + (size_t)MERGE_MAX_WINDOWS * nt + 64 + (size_t)(nt / 32) * 32 * (16 / elem);
}

template <typename R> struct KParams {
    int K, T, k_offset, noise_stream;  // noise_stream: fourth Philox counter word (of agent 0) -- kept in the first
                                       // kernel-argument fetch: the draw is the first thing a wave does
    int n_exploit, n_ref, n_obs, window;
    int model, accumulate, sequential, obstacle_model;
    int clamp_rollout, wrap_stage, wrap_term, use_philox;
    int traj_per_block, slots;  // slots: several agents per launch, records per agent in the partials / heads buffers
    unsigned seed_lo, seed_hi;
    R dt, umax0, umax1, wheel_base;
    R beta, gamma, penalty, two_pi;
    R sinv[4], ws[4], wt[4];
    R shape_x[9], shape_y[9];
    float chol[3];
    const R *ref;      // [n_ref][4]  x, y, yaw, v
    const R *obs;      // [n_obs][4]  x, y, threshold^2, 0
    const R *u;        // [T][2] nominal controls
    const float *eps;  // [K][T][2] or nullptr (Philox); with eps_slots > 0: a ring [eps_slots][n_agents][K][T][2]
    int eps_slots;     // 0: `eps` is this call's tensor; 2^n: iteration i reads slot i mod eps_slots (mppi_set_noise_ring)
    int pad_eps;
    R *S;              // [K]
    int *pout;         // [K] waypoint index after sample k (sequential mode)
    DevState *st;
    R *heads;          // [n records][4] {rho, eta, eta2, 0}: compact copy of the record heads, so that the merge
                       // kernels read them coalesced (inside the records they sit 16 + 8T bytes apart)
    // synchronous step: the observed state and its nearest-waypoint index arrive as kernel arguments (the x0
    // call, mppi_differential_drive.py:96-99, was made by the host side of the ABI) instead of through *st
    int use_args, c_arg;
    double x0_arg[4];
    // several agents per launch (blockIdx.y): agent a's u / S / pout / state / records / heads follow agent a-1's
    int n_agents, layout;   // layout: rollout_layout() of the handle (host side only)
    // MPPI_WAYPOINT_PER_ROLLOUT: the index threads through a sample's own cost calls (as `sequential` threads it through
    // all samples' calls) and starts from the x0 call's index at every sample: samples stay independent
    int per_rollout, pad_pr;
    // one-launch resolution of the sequential index (see LB_CAND)
    int hyp;                // the handle qualifies (one pass per workgroup, T <= 64, window 20 / 10, `S[k] =`, one agent, <= 512 workgroups)
    unsigned lb_seq;        // this launch pair's tag (see lb_tag)
    unsigned *hyp_slots;    // [HYP_MAX_BLOCKS] one word per workgroup
};

struct FinalizeParams {
    int T, K, n_part, pad2;
    int filter_mode, filter_window, clamp_u, raise_at_path_end;
    int model, sequential, plant, n_ref;
    int window, is_f64, count_hits, pad1;  // count_hits: the block records' heads carry collision counts (a handle with obstacles)
    double beta, dt, wheel_base, umax0, umax1;
    const void *partials;    // [n_part][partial_len], n_part <= 256; element type: see launch_finalize
    const void *heads;       // compact heads of `partials` (KParams::heads); unused for the ABI layout
    void *u;                 // [T][2] in the kernel precision: the controls the finished rollouts used
    void *u_out;             // where the updated, shifted controls go (== u: in place)
    DevState *st_out;        // where the state after this call goes (== st: in place)
    void *u_before;          // copy of u before the update (for the viz rollouts)
    const void *ref;         // [n_ref][4] kernel precision
    const int *pout;
    DevState *st;
    StepResult *res;         // followed by 2*T doubles (device memory, or host memory mapped into the device)
    double *u0_trace;        // nullable: closed-loop trace [iter][2]
    long long seq;           // != 0: publish res->seq = seq last, behind a system-scope fence (host polls it)
    int use_args, c_arg;     // see KParams
    double x0_arg[4];
    // K sharded over GPUs with the records exchanged peer to peer: after merging its own block records the
    // block stores this rank's record into slot (x_seq & 1) of EVERY rank's exchange buffer, raises its
    // flag there to x_seq, waits for all flags in its own buffer and merges the x_nranks records
    int x_nranks, x_rank;    // x_nranks <= 1: no exchange
    long long x_seq;         // > 0, the same on every rank for the same iteration, increasing
    long long x_timeout;     // 100 MHz ticks the wait may take before the iteration is abandoned
    char *const *x_peers;    // device array [x_nranks]: every rank's exchange buffer (own one included)
    int *x_err;              // sticky device flag: an exchange timed out, later slots return at once
    // several agents per launch (blockIdx.y), see KParams
    int slots, n_agents;
    size_t res_stride;       // bytes between two agents' StepResult (+ returned u)
    // one-launch resolution of the sequential index (see LB_CAND / KParams)
    int hyp, hyp_blocks;
    const unsigned *hyp_slots;
    unsigned lb_seq, pad_lb;
};

// learned residual dynamics (mppi_mlp.hip): device pointers to fragment-packed weights
struct MlpParams {
    const float *w_in, *b_in;        // Linear(5 -> 512): packed [16][1][64][4], bias [512]
    const float *w_h[3], *b_h[3];    // Linear(512 -> 512) x 3: packed [16][64][64][4], bias [512]
    const float *w_out;              // Linear(512 -> 3): [3][512] as in the checkpoint
    float b_out[3];
    // the f16-split kernel (k_rollout_mlp_h3): per layer two f16 planes (hi, then lo) in its fragment order
    int use_h3;
    const unsigned short *h3_w_in;   // [2][16][1][64][8]
    const unsigned short *h3_w_h[3]; // [2][16][32][64][8]
    // |W_in z + b_in|_inf <= in_gain |z|_inf + in_bias: the split kernel derives the per-sample power-of-two scale of the
    // first layer's output from it, so that no f16 half overflows whatever the magnitude of the inputs
    float in_gain, in_bias;
    int n_hidden;  // hidden Linear(512, 512) + tanh layers: 3 (the architecture train/train_diff_mlp.py:13-36 builds) or 2
                   // (the reference's older checkpoints, saved_models/mlp_diff.pth, mlp_diff_300x100.pth, ..._v2.pth)
};

struct VizParams {
    int K, T, model, clamp_rollout;
    int k_offset, n_exploit, use_philox, pad;
    unsigned seed_lo, seed_hi;
    long long iter;
};

template <typename R> void launch_set_state(const KParams<R> &P, const double *x0_or_null, hipStream_t s);
template <typename R> void launch_rollout(const KParams<R> &P, hipStream_t s);
// softmin partial records are stored in the handle's precision R (block partials) or as double (the
// per-rank record of the split step, include/mppi_hip.h)
template <typename R> void launch_reduce(const KParams<R> &P, void *partials, int n_blocks, hipStream_t s);
// rollout + cost + per-block softmin partial in one launch (T <= 128); fused_blocks(K) records
template <typename R> void launch_rollout_fused(const KParams<R> &P, void *partials, hipStream_t s);
bool fused_supported(int T);
// the instantiation this thread's last rollout-class launch took, as rocprofv3 spells it ("k_rollout_dual<float, 1, 1, false, 2, true>")
const char *last_rollout_kernel();
// Which fused rollout kernel serves (K, T): decided ONCE per handle (it reads the MPPI_DUAL / MPPI_PAIR / MPPI_SEQ
// overrides) and carried in KParams::layout, so that a launch costs no environment lookups.
enum { LAYOUT_FUSED = 0, LAYOUT_DUAL = 1, LAYOUT_PAIR = 2, LAYOUT_TRI = 3, LAYOUT_KIND = 3, LAYOUT_TWICE = 4 };  // (KIND: mask)
// n_agents: problems batched in one launch; tri_ok: the handle is what k_rollout_tri serves (race car, f32, frozen index,
// `S[k] +=`, one agent) -- it takes horizons of 65 .. 96 steps then
int rollout_layout(int K, int T, int n_agents, int model, bool f64, bool per_rollout = false, bool tri_ok = false);
int fused_blocks(int K, int T, int layout);  // workgroups = block records of one launch
// records launch_rollout_fused(P) leaves (the streaming kernel, which serves tensors of noise, leaves fewer: see k_rollout_stream)
template <typename R> int fused_records(const KParams<R> &P);
// merges groups of `group` <= 256 records (precision R) of `recs[n]` into out[ceil(n/group)]
// (`heads` / `out_heads`: the compact head arrays of the input / internal-layout output records)
template <typename R>
void launch_merge(const void *recs, const void *heads, int n, int group, int T, double beta, void *out, void *out_heads,
                  bool out_f64, hipStream_t s);
// F.partials holds n_part <= 256 records of precision R (recs_f64 false) or double
// (with F.x_nranks > 1 and recs_f64 false: the peer-to-peer exchange variant)
template <typename R> void launch_finalize(const FinalizeParams &F, bool recs_f64, hipStream_t s);
// exchange self-test: one flag round over the peers, no records
void launch_exchange_probe(const FinalizeParams &F, int *ok_out, hipStream_t s);
// batched stage methods (mppi_eval_*): `what` of launch_eval
enum { EVAL_TRANSITION = 0, EVAL_COST_STAGE = 1, EVAL_COST_TERMINAL = 2, EVAL_COLLIDED = 3, EVAL_CLAMP = 4 };
template <typename R>
void launch_eval_index(const KParams<R> &P, const R *xy, int stride, int n, int p0, int sequential, int *idx_out, int *p_out,
                       hipStream_t s);
template <typename R>
void launch_eval(const KParams<R> &P, int what, const R *x, const R *v, const int *idx, int n, R *out, hipStream_t s);
template <typename R> void launch_eval_filter(const R *xx, R *out, int T, int W, int mode, hipStream_t s);
void launch_eval_weights(const double *S, int n, double beta, double *w, hipStream_t s);
template <typename R> void launch_set_state_dev(const KParams<R> &P, const double *x_dev, int nx, hipStream_t s);
template <typename R> void launch_weights(const KParams<R> &P, double rho, double eta, double *w_out, hipStream_t s);
void launch_sample(unsigned seed_lo, unsigned seed_hi, unsigned iter, int K, int T, int k_offset, const float *chol,
                   float *eps_out, hipStream_t s, unsigned stream_word = 0);
template <typename R>
void launch_viz(const KParams<R> &P, const R *u_before, const R *u_after_pre_shift, long long iter, float *opt,
                float *smp, hipStream_t s);
int reduce_blocks(int K, int traj_per_block);
// config 5: rollout through the residual MLP on MFMA, one record per 64-sample tile
void launch_rollout_mlp(const KParams<float> &P, const MlpParams &Q, void *partials, hipStream_t s);
// the visualisation rollouts (mppi_differential_drive.py:144-159) with the learned model: opt [T][3], smp [K][T][3] (either
// may be null); u_before / u_upd: the nominal controls before the update and the updated, unshifted ones
// `_state_transition` with the learned model for n (state, control) rows: x [n][3], v [n][2] -> out [n][3]
void launch_eval_mlp(const KParams<float> &P, const MlpParams &Q, const float *x, const float *v, int n, float *out, hipStream_t s);
void launch_viz_mlp(const KParams<float> &P, const MlpParams &Q, const float *u_before, const float *u_upd, long long iter,
                    float *opt, float *smp, hipStream_t s);
int mlp_blocks(int K, int tile);            // workgroups = softmin records of a launch over K samples
int mlp_tile(const MlpParams &Q);          // samples per workgroup of the rollout kernel that serves Q (32 or 64)
const char *mlp_kernel_name(const MlpParams &Q);  // as rocprofv3 spells the rollout kernel that serves Q
void pack_linear(const float *w, int n_in, float *packed);  // host: [512][n_in] -> fragment order
void pack_linear_h3(const float *w, int n_in, unsigned short *packed);  // host: -> two f16 planes in fragment order
constexpr int MODEL_DIFF_MLP = 2;

}  // namespace mppi
