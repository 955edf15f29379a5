// wbcqp_api.hip -- C ABI (include/wbcqp.h) over the one-wavefront-per-QP kernel.
//
// Replaces, for B robot instances at once, the two calls the reference makes per control tick
// (/root/reference/src/controllers/controller.cpp:244 computeProblemData [assembly half] and :247
// solver_->solve) plus the decode at :250-251.  There is no CPU path in this library: without a
// gfx950 device wbcqp_create fails with WBCQP_ERR_NO_DEVICE.
#include "wbcqp_device.hpp"
#include "wbcqp_terms.hpp"
#include "wbcqp_dense.hpp"
#include "wbcqp_small.hpp"

#include "../../include/wbcqp.h"

#include <dlfcn.h>
#include <limits>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace wbcqp;

namespace {

thread_local std::string g_create_error;

struct Slot {
    bool set = false;
    DevStruct host{};           // sizes, LDS layout and device pointers of the tables: travels by value with every launch
    DevStruct host_cp{};        // the same with the compact LDS layout (wbcqp_compact.hpp); host_cp.compact == 0: not eligible
    std::vector<void*> allocs;  // device arrays owned by this slot
    wbcqp_layout layout{};      // what wbcqp_layout_of reports: the compact layout where the structure is eligible
    int lds_full = 0, lds_cp = 0;
    bool small = false;         // eligible for the one-wavefront-per-QP kernel (wbcqp_small.hpp)
    int spec = 0;               // 1-based index of the compact kernel's instantiation for this very layout (kSpecDims), 0: the generic kernel
    std::vector<int> sel_col_h;       // host copies of what wbcqp_set_model needs of the structure: the posture task's columns
    std::vector<double> force_gen_h;  // and the contacts' force generators (the contact points sit in their skew blocks)
    double* ffc_dev = nullptr;  // the force blocks' factor for one weight (DevStruct::ffc; owned through `allocs`), null: none (no contacts, not compact, disabled)
    bool ffc_built = false;     // ... made from the first QP of the slot's first compact launch (solve_ragged)
    bool has_model = false;     // wbcqp_set_model: tree + task bindings for wbcqp_problem_data
    TermsDev terms{};
    std::vector<void*> model_allocs;
};

struct Staging {
    void* dev = nullptr;
    size_t bytes = 0;
};
// page-locked host memory for the host-pointer entry points at small batches: the input arrays of a QP record (eleven, twelve with a cop task; five output
// arrays) cross PCIe as ONE copy each way instead of eleven (five) -- at batch 1 the copies' fixed cost is most of the call
struct Pinned {
    void* host = nullptr;
    size_t bytes = 0;
};
constexpr size_t kPackedBytes = 1u << 20; // above this the arrays go up one by one (the extra host copy would cost more than it saves)

} // namespace

// launch-order state: the order left by one launch for the next one of the same shape on the same stream.  The handle has
// one PER STREAM it has launched on (two launches in flight on two streams share neither the order buffer -- the schedule
// kernel of one would overwrite what the solve kernel of the other is reading -- nor the queue counter); every captured tick
// (wbcqp_graph) and every sub-batch of a roll-out has its own, so that a replay never touches a buffer another launch may
// resize or overwrite (a graph bakes the buffer's address into its kernel nodes).
struct OrderState {
    int* order = nullptr;        // [2][cap]: longest-first, then the packed order (pack_order_kernel)
    bool packed = false;         // the second half is valid for total / sig / stream
    int age = 0;                 // launches that have used the order since it was computed
    int cap = 0;
    int total = 0;               // 0: no valid order
    unsigned long long sig = 0;  // shape of the launch the order belongs to
    hipStream_t stream = nullptr;
    int* queue = nullptr;        // the queue counter pair of solve_queue_kernel that goes with this order (allocated on first use)
};

constexpr size_t kMaxQueues = 16;
constexpr int kQueueMinLds = 48 * 1024; // workgroups at least this large take their QPs from the queue by default (measured: below it the dispatcher wins, launch())
constexpr int kQueue3MinLds = 40 * 1024; // ... and from 40 KB on where the launch runs three per CU through solve_queue3_kernel (the dispatcher's solve_kernel holds
                                         // two); a 40-48 KB stack that cannot take the twin (actuation bounds, warm start) stays with the dispatcher
constexpr int kLdsThree = 54592;        // the largest dynamic LDS block of which a CU holds three (tools/ubench/lds_granule.hip)
constexpr int kOrderRefresh = 4; // default period of the launch-order renewal (WBCQP_FLAG_REFRESH)

struct wbcqp_handle {
    int device = 0;
    int dtype = WBCQP_F64;
    std::string err;
    Slot slots[WBCQP_MAX_STRUCTURES];
    Staging stage_in, stage_out;
    Pinned pin_in, pin_out;
    // per kernel variant (0: full layout, 1: compact, 2 + i: the compact kernel specialised for kSpecDims[i]): largest dynamic LDS size set so far
    int max_lds[2 + kNumSpecs] = {};
    long long* dbg = nullptr; // diagnostic builds only (wbcqp_debug_set_stamp_buffer)
    // longest-first schedule (schedule_kernel): launch order for the next solve of the same shape on the same stream
    int flags = 0;
    // one order state (order buffer + queue counter pair) per stream this handle has launched on; a launch on a stream beyond
    // kMaxQueues distinct ones runs in index order on the hardware's dispatcher and leaves no state behind
    struct StreamState {
        hipStream_t stream;
        OrderState ord;
    };
    std::vector<StreamState> streams;
    int last_stream = -1;            // index of the stream state the most recent launch used (wbcqp_launch_order reports that one)
    OrderState* graph_ord = nullptr; // wbcqp_tick_graph_create: the launches of this tick use the graph's own order state
    bool capturing = false;          // ... and a captured tick always renews its order
    int n_cu = 0;
    int dense_max_lds = 0;
    // wbcqp_solve_dense_host: the reference's HQPOutput, owned by the solver and valid until the next call
    std::vector<double> dense_x, dense_obj;
    std::vector<int32_t> dense_status, dense_iters, dense_nact;
    wbcqp_dense_output dense_out{};
    int lds_pad = 0; // diagnostic (env WBCQP_DEBUG_LDS_PAD): extra dynamic LDS per workgroup, to force a lower residency
    int queue_lds[2 + kNumSpecs], queue_occ[2 + kNumSpecs] = {}; // occupancy of solve_queue_kernel<., CP, SPEC> at queue_lds bytes of LDS
    int queue_occ_warm[2 + kNumSpecs] = {};                      // ... of solve_queue_kernel_warm<., SPEC> (WBCQP_FLAG_WARM_START launches that kernel)
    bool debug_launch = false;                                   // env WBCQP_DEBUG_LAUNCH, read once at wbcqp_create (never on the per-tick path)
    bool no_ffcache = false;                                     // env WBCQP_DEBUG_NO_FFCACHE: every QP eliminates its force blocks itself (what tests compare the cache with)
    int queue_occ3[2 + kNumSpecs] = {};                          // ... of solve_queue3_kernel<., SPEC> where queue_three says it holds three
    bool warned_occupancy = false;                               // the one-time note of launch() when the runtime's occupancy answer is overruled
    bool queue_three[2 + kNumSpecs] = {};                        // ... and whether solve_queue3_kernel<., SPEC> holds three workgroups per CU at that size
    // wbcqp_rollout: sub-batches on streams of their own (each with its own launch-order state and queue counter), the record
    // arrays and the state ping-pong of the whole batch
    struct RollSub {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        OrderState ord;
    };
    std::vector<RollSub> roll_subs;
    hipEvent_t roll_start = nullptr;
    hipEvent_t roll_done = nullptr;  // end of the previous roll-out: the next one (on whatever stream) waits for it before it reuses the buffers
    // how many sub-batches a roll-out is cut into is MEASURED, per (slot, batch): the device time of every roll-out lies between
    // roll_start and roll_done (both timed events); the next call reads it without blocking (hipEventQuery) and keeps, per shape, a
    // running figure of microseconds per tick for one sub-batch (= what K calls of wbcqp_tick do) and for two
    struct RollStat {
        int slot = -1, batch = 0, calls = 0;
        double us[3] = {0.0, 0.0, 0.0}; // [S] running mean, 0: never measured
        int cold[3] = {1, 1, 1};        // [S] the next measurement of this form is its first: it paid the form's allocations and stream set-up, it is not kept
    };
    std::vector<RollStat> roll_stats;
    struct RollMeas { // one timed event pair around a roll-out, read by a later call once the device has passed it
        hipEvent_t t0 = nullptr, t1 = nullptr;
        int stat = -1, S = 0, ticks = 0;
        bool pending = false;
    };
    RollMeas roll_meas[4];
    Staging roll_rec, roll_state;
};

namespace {

int fail(wbcqp_handle* h, int code, const std::string& msg)
{
    if (h)
        h->err = msg;
    else
        g_create_error = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, WBCQP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

int odd(int v) { return v | 1; }
void set_lds(wbcqp_layout& L, int lds_bytes, bool compact = false, bool act_bounds = false);

// Validates a structure and derives sizes + LDS layout. Pure host code.
int derive(const wbcqp_structure* st, DevStruct& D, HostBlocks& HB, wbcqp_layout& L, std::string& why)
{
    if (!st) { why = "structure is NULL"; return WBCQP_ERR_INVALID; }
    if (st->nv <= 0 || st->na < 0 || st->na > st->nv || st->nc < 0) { why = "bad nv/na/nc"; return WBCQP_ERR_INVALID; }
    if (st->n_dense < 0 || st->n_sel < 0 || st->n_tasks <= 0 || st->n_bound < 0) { why = "bad level-1 sizes"; return WBCQP_ERR_INVALID; }
    if (st->n_ineq_blocks < 0 || st->n_ineq_blocks > WBCQP_MAX_INEQ_BLOCKS) { why = "too many inequality blocks"; return WBCQP_ERR_INVALID; }
    std::memset(&D, 0, sizeof(D));
    std::memset(&L, 0, sizeof(L));
    D.nv = st->nv; D.na = st->na; D.nc = st->nc; D.k = 12 * st->nc; D.n = D.nv + D.k; D.nu = D.nv - D.na;
    if (D.n > WBCQP_MAX_VARS) { why = "n = nv + 12 nc exceeds WBCQP_MAX_VARS"; return WBCQP_ERR_UNSUPPORTED; }
    if (D.nv > 64) { why = "nv exceeds 64 (the dv block is factorised on a 64 x 64 register grid)"; return WBCQP_ERR_UNSUPPORTED; }
    if (st->nc > 15) { why = "more than 15 contacts"; return WBCQP_ERR_UNSUPPORTED; }
    if (st->n_tasks > kSlot || st->n_dense > kSlot || st->n_bound > kSlot || 6 * st->nc > kSlot) { why = "a per-QP vector exceeds 128 entries"; return WBCQP_ERR_UNSUPPORTED; }
    D.n_dense = st->n_dense; D.n_tasks = st->n_tasks; D.n_sel = st->n_sel; D.n_bound = st->n_bound;
    D.act_bounds = st->act_bounds ? 1 : 0;
    D.neq = D.nu + 6 * D.nc;
    // level-1 tasks that couple the blocks of H ("torque", "cop"): H dense, full layout
    D.n_acteq = st->n_acteq;
    D.acteq_task = st->acteq_task;
    D.cop_task = st->cop_task;
    if (D.n_acteq < 0 || D.n_acteq > D.na) { why = "n_acteq outside [0, na]"; return WBCQP_ERR_INVALID; }
    if (D.n_acteq > 0) {
        if (!st->acteq_joint || !st->acteq_scale) { why = "acteq_joint / acteq_scale is NULL"; return WBCQP_ERR_INVALID; }
        if (D.acteq_task < 0 || D.acteq_task >= D.n_tasks) { why = "acteq_task out of range"; return WBCQP_ERR_INVALID; }
        for (int j = 0; j < D.n_acteq; ++j)
            if (st->acteq_joint[j] < 0 || st->acteq_joint[j] >= D.na || (j > 0 && st->acteq_joint[j] <= st->acteq_joint[j - 1])) { why = "acteq_joint must be ascending in [0, na)"; return WBCQP_ERR_INVALID; }
    }
    if (D.cop_task >= D.n_tasks) { why = "cop_task out of range"; return WBCQP_ERR_INVALID; }
    if (D.cop_task < 0) D.cop_task = -1;
    if (D.cop_task >= 0 && D.nc == 0) { why = "a cop task needs a contact"; return WBCQP_ERR_INVALID; }
    if (D.cop_task >= 0) {
        // A cop task is a task of its own: its index into w is shared with no other row.  This is also what catches the C idiom the field is a trap
        // for -- a zero-initialised (memset) structure says cop_task = 0, which is some dense or selection task's index in every real stack
        bool shared = (D.n_acteq > 0 && D.acteq_task == D.cop_task);
        for (int r = 0; r < st->n_dense && st->dense_row_task && !shared; ++r) shared = st->dense_row_task[r] == D.cop_task;
        for (int r = 0; r < st->n_sel && st->sel_task && !shared; ++r) shared = st->sel_task[r] == D.cop_task;
        for (int c2 = 0; c2 < st->nc && st->forcereg_task && !shared; ++c2) shared = st->forcereg_task[c2] == D.cop_task;
        if (shared) {
            why = "cop_task " + std::to_string(D.cop_task) + " is also the task of other level-1 rows: a cop task has a weight of its own (a structure WITHOUT a cop "
                  "task sets cop_task = -1; zero-initialisation alone declares one on task 0)";
            return WBCQP_ERR_INVALID;
        }
    }
    D.dense_h = (D.n_acteq > 0 || D.cop_task >= 0) ? 1 : 0;
    if (D.dense_h && D.n > 80) { why = "a torque / cop task makes H dense: supported for n <= 80 (the dense seam, wbcqp_solve_dense, carries n <= 96)"; return WBCQP_ERR_UNSUPPORTED; }
    if (D.cop_task >= 0 && 3 * D.k > kSlot) { why = "cop rows exceed one 128-entry slot"; return WBCQP_ERR_UNSUPPORTED; }
    D.r1 = D.n_dense + D.n_sel + 6 * D.nc + D.n_acteq + (D.cop_task >= 0 ? 3 : 0);
    HB.n_blocks = st->n_ineq_blocks;
    int off = 0;
    bool has_act = false;
    int act_off = -1, n_act_blocks = 0;
    for (int b = 0; b < HB.n_blocks; ++b) {
        const int kind = st->ineq_kind[b];
        int rows;
        if (kind == WBCQP_INEQ_BOUNDS) rows = D.n_bound;
        else if (kind == WBCQP_INEQ_ACTUATION) { rows = D.na; has_act = true; act_off = off; ++n_act_blocks; }
        else if (kind == WBCQP_INEQ_FORCE) {
            rows = 17;
            if (st->ineq_arg[b] < 0 || st->ineq_arg[b] >= D.nc) { why = "force block names a missing contact"; return WBCQP_ERR_INVALID; }
        }
        else { why = "unknown inequality kind"; return WBCQP_ERR_INVALID; }
        HB.blk_kind[b] = kind; HB.blk_arg[b] = st->ineq_arg[b]; HB.blk_off[b] = off; HB.blk_rows[b] = rows;
        off += 2 * rows;
    }
    if (has_act != (D.act_bounds != 0)) { why = "act_bounds flag and inequality blocks disagree"; return WBCQP_ERR_INVALID; }
    D.nin2 = off;
    D.act_off = (n_act_blocks == 1) ? act_off : -1;
    if (D.nin2 > 4 * kSlot) { why = "more than 512 one-sided inequality rows"; return WBCQP_ERR_UNSUPPORTED; }
    if (D.r1 > 2 * kSlot) { why = "more than 256 level-1 rows"; return WBCQP_ERR_UNSUPPORTED; }
    if (D.neq > D.n) { why = "more equalities than variables"; return WBCQP_ERR_INVALID; }
    for (int r = 0; r < D.n_dense; ++r)
        if (st->dense_row_task[r] < 0 || st->dense_row_task[r] >= D.n_tasks) { why = "dense_row_task out of range"; return WBCQP_ERR_INVALID; }
    for (int r = 0; r < D.n_sel; ++r)
        if (st->sel_col[r] < 0 || st->sel_col[r] >= D.nv || st->sel_task[r] < 0 || st->sel_task[r] >= D.n_tasks) { why = "selection row out of range"; return WBCQP_ERR_INVALID; }
    for (int r = 0; r < D.n_bound; ++r)
        if (st->bound_col[r] < 0 || st->bound_col[r] >= D.nv) { why = "bound_col out of range"; return WBCQP_ERR_INVALID; }
    for (int c = 0; c < D.nc; ++c)
        if (st->forcereg_task[c] < 0 || st->forcereg_task[c] >= D.n_tasks) { why = "forcereg_task out of range"; return WBCQP_ERR_INVALID; }
    D.max_iter = st->max_iter > 0 ? st->max_iter : 1000;
    D.hessian_reg = st->hessian_reg;

    // ---- LDS layout (doubles) ----
    const int n = D.n, nv = D.nv;
    D.ldj = odd(n); D.ldm = odd(nv); D.ldc = odd(nv); D.ldb = 2 * odd((4 * ((D.neq + 3) / 4) + 1) / 2); // twice an odd number: rows stay 16-byte aligned for the 4-wide column groups (which may read past m) and 16 rows still hit 16 distinct bank groups
    int o = 0;
    auto take = [&](int count) { int at = o; o += (count + 1) & ~1; return at; }; // keep 16-byte alignment
    D.o_J = take(n * D.ldj);
    int rsize = n * (n + 3) / 2 + 2;
    if (D.n_dense * 66 + 8 > rsize) rsize = D.n_dense * 66 + 8; // staged task rows: 64 columns (transposed groups) + (w, b) pairs
    if (D.neq > 0 && 256 + (n + 17) * D.ldb + 8 > rsize) rsize = 256 + (n + 17) * D.ldb + 8; // B of the blocked equality phase + 16 zero rows // + 64: the 4x4 H tiles may read past the last staged row
    D.o_R = take(rsize);
    D.o_M = take(nv * D.ldm);
    D.o_Jc = take(D.k * D.ldc);
    D.o_Ac = take(D.nc * 6 * nv);
    D.o_vec = take(V_COUNT * kSlot);
    D.o_eqw = take(D.neq > 0 ? (n + 1) * D.ldb + 8 : 0);
    D.o_eqt = take(D.neq > 0 ? D.neq * (D.neq + 1) + 4 * D.neq + 16 : 0);
    D.fric_lds = (D.nc > 0 && (o - D.o_eqw) >= 238 * D.nc) ? 1 : 0;
    D.o_int = o;
    o += kIntCount / 2 + 2;
    D.lds_doubles = o;
    D.compact = 0;

    std::memset(&L, 0, sizeof(L));
    L.n = n; L.neq = D.neq; L.nin = D.nin2 / 2; L.nin2 = D.nin2; L.r1 = D.r1;
    L.len_M = nv * (nv + 1) / 2; L.len_h = nv; L.len_A = D.n_dense * nv; L.len_b1 = D.r1;
    L.len_Ac = D.nc * 6 * nv; L.len_bc = D.nc * 6; L.len_blb = D.n_bound; L.len_bub = D.n_bound;
    L.len_tlb = D.act_bounds ? D.na : 0; L.len_tub = L.len_tlb; L.len_w = D.n_tasks;
    L.len_Acop = D.cop_task >= 0 ? 3 * D.k : 0;
    L.dense_h = D.dense_h;
    set_lds(L, o * 8);
    const int64_t n_in = (int64_t)L.len_M + L.len_h + L.len_A + L.len_b1 + L.len_Ac + L.len_bc + L.len_blb + L.len_bub +
                         L.len_tlb + L.len_tub + L.len_w + L.len_Acop;
    L.algorithmic_bytes = 8 * (n_in + n + D.na) + 8;
    if (L.lds_bytes > 160 * 1024) { why = "QP does not fit the 160 KiB LDS of one CU"; return WBCQP_ERR_UNSUPPORTED; }
    return WBCQP_OK;
}

// The compact LDS layout of an eligible structure (wbcqp_compact.hpp): J region | R region | vectors | ints; everything
// else is staged inside the first two while they are idle, or never enters LDS.  Returns false when not eligible.
// one wavefront per QP (wbcqp_small.hpp): fixed base, no contacts, n = nv <= 16, bounds as the only inequality rows
bool small_ok(const DevStruct& D, const HostBlocks& HB)
{
    if (D.dense_h) return false;
    if (!(D.nc == 0 && D.nu == 0 && D.neq == 0 && D.n == D.nv && D.n >= 1 && D.n <= 16 && D.na <= 16 && D.n_dense <= 16 && D.n_sel <= 16 &&
          D.n_tasks >= 1 && D.n_tasks <= 16 && D.n_bound <= 16 && D.nin2 <= 32 && D.r1 >= 1 && D.r1 <= 32 && !D.act_bounds))
        return false;
    for (int b = 0; b < HB.n_blocks; ++b)
        if (HB.blk_kind[b] != WBCQP_INEQ_BOUNDS) return false;
    return true;
}

bool derive_compact(const DevStruct& F, DevStruct& D)
{
    D = F;
    D.compact = 0;
    if (F.dense_h) return false; // a torque / cop task: H is one n x n matrix, the compact kernel factors a dv block and 12 x 12 blocks
    const int n = F.n, nv = F.nv;
    // (n <= 78: the 80-entry vector slots hold a zero pad pair behind column n for the loop's 16-byte row passes)
    if (!(n <= 78 && F.neq <= 22 && nv <= 52 && F.nc <= 2 && F.nu <= 8 && F.na <= 64 && F.n_bound <= 64 && F.nin2 <= 256 &&
          F.r1 <= 128 && F.n_tasks <= 64 && (!F.act_bounds || F.act_off >= 0) && n - F.neq <= 64))
        return false;
    int o = 0;
    auto take = [&](int count) { int at = o; o += (count + 1) & ~1; return at; };
    // rows of J: 16-byte aligned, a zero pad pair behind column n, and 2 x odd long -- sixteen rows then start in sixteen
    // different bank groups for 8-byte and for 16-byte accesses alike
    {
        int l = ((n + 1) & ~1) + 2;
        while ((l & 3) != 2) l += 2;
        // n itself is 2 x odd and there are equality columns to spare: the pad pair of a row is the next row's (dead, zeroed) columns 0-1 in the
        // loop, two more doubles end the last row
        if ((n & 3) == 2 && F.neq >= 2) l = n;
        D.ldj = l;
    }
    const int as_size = (F.n_dense * 66 + 8 + 1) & ~1;   // staged task rows: 64 columns + (w, b) pairs
    int jsize = n * D.ldj + 2;
    if (as_size + 1024 > jsize) jsize = as_size + 1024;  // + the elimination's panels (2 x 2 x 256)
    D.o_pan = as_size;
    D.o_J = take(jsize);
    int rs = 512;                                        // packed R of the equalities (neq <= 22 columns)
    if (F.neq > 0 && (n + 4) * F.ldb + 8 > rs) rs = (n + 4) * F.ldb + 8; // N = CE', then B = J0'N (from the region's start: the packed R follows it in time)
    {   // the inequality loop: Ri (row-packed, n - neq rows, one spare element per row; reads past its end land in what follows), four doubles per
        // rotation of a drop, the friction rows' table (one sign)
        const int mmax = n - F.neq;
        const int fric = cp::fric_in_j(n, F.neq, F.nc) != 0 ? 0 : F.nc * 17 * 12; // (with fourteen equalities the table lives in J's dead columns: wbcqp_compact.hpp)
        const int need = ((2 + mmax * (mmax + 3) / 2 + 1) & ~1) + 4 * (mmax + 2) + fric + 2; // (reads past Ri's last row reach at most mmax + 8 doubles into the 4 (mmax + 2) of the rotation table)
        if (need > rs) rs = need;
    }
    D.o_R = take(rs);
    D.o_vec = take(cp::vec_map(n, nv, F.act_bounds).COUNT);
    D.o_int = o;
    o += cp::ICOUNT / 2;
    D.o_M = D.o_Jc = D.o_Ac = D.o_eqw = D.o_eqt = 0;
    D.fric_lds = 0;
    D.lds_doubles = o;
    D.compact = 1;
    return true;
}

void set_lds(wbcqp_layout& L, int lds_bytes, bool compact, bool act_bounds)
{
    L.lds_bytes = lds_bytes;
    L.waves_per_cu = lds_bytes > 0 ? (160 * 1024) / lds_bytes : 0;
    // registers: the solve kernels allocate up to 256 VGPRs = two waves per SIMD = two workgroups per CU; the compact layout has a twin compiled
    // for three (solve_queue3_kernel), taken when three workgroups fit the CU's LDS: measured (tools/ubench/lds_granule.hip) the third one fits
    // up to 54 592 bytes of dynamic LDS beside the kernel's static word -- no coarser granule than 16 bytes (the launch asks the runtime itself)
    const int cap = (compact && !act_bounds && lds_bytes >= kQueue3MinLds && lds_bytes <= kLdsThree) ? 3 : 2; // (below kQueueMinLds: solve_kernel, two per CU;
                                                                                                           //  with actuation bounds three per CU measured slower: kThree)
    if (L.waves_per_cu > cap) L.waves_per_cu = cap;
}

template <typename T>
int upload(wbcqp_handle* h, Slot& s, const T* src, size_t count, const T** dst)
{
    void* p = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    HIP_TRY(h, hipMalloc(&p, bytes));
    s.allocs.push_back(p);
    if (count) HIP_TRY(h, hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return WBCQP_OK;
}

void release_model(Slot& s)
{
    for (void* p : s.model_allocs) (void)hipFree(p);
    s.model_allocs.clear();
    s.has_model = false;
}

void release(Slot& s)
{
    for (void* p : s.allocs) (void)hipFree(p);
    s.allocs.clear();
    s.set = false;
    release_model(s);
}

template <typename TI>
void fill_group(GroupArgs<TI>& g, const Slot& s, bool compact, int batch, const wbcqp_inputs* in, const wbcqp_outputs* out)
{
    g.st = compact ? s.host_cp : s.host;
    g.M = static_cast<const TI*>(in->M); g.h = static_cast<const TI*>(in->h); g.A = static_cast<const TI*>(in->A);
    g.b1 = static_cast<const TI*>(in->b1); g.Ac = static_cast<const TI*>(in->Ac); g.bc = static_cast<const TI*>(in->bc);
    g.blb = static_cast<const TI*>(in->blb); g.bub = static_cast<const TI*>(in->bub);
    g.tlb = static_cast<const TI*>(in->tlb); g.tub = static_cast<const TI*>(in->tub); g.w = static_cast<const TI*>(in->w);
    g.Acop = static_cast<const TI*>(in->Acop);
    g.x = static_cast<TI*>(out->x); g.tau = static_cast<TI*>(out->tau); g.objective = static_cast<TI*>(out->objective);
    g.status = out->status; g.iters = out->iters; g.n_active = out->n_active;
    g.amask = out->active_mask; // every kernel writes the mask; only the compact one takes it as the pick hint (g.warm)
    g.warm = 0;
    g.dbg = nullptr;
    g.count = batch;
}

int check_io(wbcqp_handle* h, const Slot& s, int batch, const wbcqp_inputs* in, const wbcqp_outputs* out)
{
    if (batch < 0) return fail(h, WBCQP_ERR_INVALID, "negative batch");
    if (batch == 0) return WBCQP_OK;
    if (!in || !out) return fail(h, WBCQP_ERR_INVALID, "inputs/outputs struct is NULL");
    const wbcqp_layout& L = s.layout;
    auto need = [&](const void* p, int len, const char* name) -> bool {
        if (len > 0 && !p) { h->err = std::string("input array ") + name + " is NULL"; return false; }
        return true;
    };
    if (!need(in->M, L.len_M, "M") || !need(in->h, L.len_h, "h") || !need(in->A, L.len_A, "A") || !need(in->b1, L.len_b1, "b1") ||
        !need(in->Ac, L.len_Ac, "Ac") || !need(in->bc, L.len_bc, "bc") || !need(in->blb, L.len_blb, "blb") ||
        !need(in->bub, L.len_bub, "bub") || !need(in->tlb, L.len_tlb, "tlb") || !need(in->tub, L.len_tub, "tub") ||
        !need(in->w, L.len_w, "w") || !need(in->Acop, L.len_Acop, "Acop"))
        return WBCQP_ERR_INVALID;
    if (!out->x || !out->status || !out->iters || (s.host.na > 0 && !out->tau))
        return fail(h, WBCQP_ERR_INVALID, "output arrays x, tau, status, iters are required");
    return WBCQP_OK;
}

// which instantiations have a three-per-CU twin: the compact kernel, generic or iCub's.  Not Talos's (two feet: 72 KB of LDS; one foot fits since its layout's
// last diet, 54 480 B, but LOSES there: 8.03 M QP/s at three per CU against 9.32 M at two -- with actuation bounds the loop keeps the actuation rows in 38
// registers, and at 168 they live in scratch, on the chain of every pick; tools/occ3_probe.py --stack talos_single_support).  The generic twin is likewise
// taken only for stacks WITHOUT actuation bounds (launch()).
template <bool CP, int SPEC> constexpr bool kThree = CP && (SPEC == 0 || SPEC == 2);
// which instantiations have a twin with the warm start's pick hint compiled in (WBCQP_FLAG_WARM_START): the generic compact kernel and Talos's; a handle
// with that flag runs every compact launch through one of the two (wbcqp_solve_ragged routes the other stacks to the generic one)
template <bool CP, int SPEC> constexpr bool kWarm = CP && (SPEC == 0 || SPEC == 1);

template <typename TI, bool CP, int SPEC = 0>
int launch(wbcqp_handle* h, GroupTable<TI>& tab, int total, int lds_bytes, hipStream_t stream)
{
    static_assert(SPEC == 0 || CP, "only the compact kernel is specialised");
    if (total == 0) return WBCQP_OK;
    constexpr int V = SPEC > 0 ? 1 + SPEC : (CP ? 1 : 0);
    if (lds_bytes > h->max_lds[V]) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<TI, CP, SPEC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_queue_kernel<TI, CP, SPEC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        if constexpr (kWarm<CP, SPEC>) {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel_warm<TI, SPEC>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_queue_kernel_warm<TI, SPEC>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        }
        if constexpr (kThree<CP, SPEC>)
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_queue3_kernel<TI, SPEC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        h->max_lds[V] = lds_bytes;
    }
    // schedule: the order left by the previous launch is used when it is of this very shape and was produced on this
    // stream (stream order then guarantees that it is complete); otherwise index order
    OrderState* osp = h->graph_ord;
    if (!osp) {
        for (size_t i = 0; i < h->streams.size() && !osp; ++i)
            if (h->streams[i].stream == stream) { osp = &h->streams[i].ord; h->last_stream = (int)i; }
        if (!osp && h->streams.size() < kMaxQueues) {
            h->streams.push_back({stream, OrderState{}});
            h->last_stream = (int)h->streams.size() - 1;
            osp = &h->streams.back().ord;
            osp->stream = stream;
        }
        else if (!osp)
            h->last_stream = -1;
    }
    OrderState none{};
    OrderState& os = osp ? *osp : none;
    const bool sched = osp && !(h->flags & WBCQP_FLAG_INDEX_ORDER) && total > 1;
    unsigned long long sig = 1469598103934665603ull;
    ScheduleArgs sa{};
    sa.n = tab.n;
    for (int g = 0; g < tab.n; ++g) {
        sig = (sig ^ (unsigned long long)(uintptr_t)tab.g[g].st.rowmeta) * 1099511628211ull;
        sig = (sig ^ (unsigned long long)tab.g[g].st.lds_doubles) * 1099511628211ull;
        sig = (sig ^ (unsigned long long)tab.g[g].count) * 1099511628211ull;
        sa.iters[g] = tab.g[g].iters;
        sa.count[g] = tab.g[g].count;
    }
    tab.order = (sched && os.total == total && os.sig == sig) ? os.order + (os.packed ? os.cap : 0) : nullptr;
    // The queue pays when a QP is long enough for a hand-over (1 us: atomic + order entry) to vanish and few enough workgroups
    // fit a CU for the dispatcher's binding of a workgroup to one shader engine to leave CUs idle: the humanoid stacks (one
    // or two workgroups per CU; measured on the compact layout, tools/dispatch_sweep.py: 1-2 % over the dispatcher at every
    // batch size).  Small QPs (Franka: 26 KB of LDS) give the dispatcher slack -- measured 27 M QP/s through the queue
    // against 36 M through the hardware.  WBCQP_FLAG_QUEUE forces the queue, WBCQP_FLAG_HW_DISPATCH the dispatcher.
    if (h->queue_lds[V] != lds_bytes) {
        // Resident workgroups per CU of the kernel that will be launched: the runtime's answer, checked against what THIS kernel's own resources admit on
        // the device the handle is bound to (wbcqp_create accepts gfx950 only): k workgroups while k (lds + 16) <= 160 KB (measured, tools/ubench/lds_granule.hip)
        // and k waves per SIMD while k x (its allocated VGPRs, 8-register granule) <= 512 -- both read from the kernel itself (hipFuncGetAttributes), not from a
        // constant.  A runtime that answers LESS than both admit is not believed: seen when a process holds TWO HIP runtimes (the library loaded before torch:
        // the first one then answers 1 for every kernel, tools/occ_state_probe.py); workgroups that do not fit wait their turn, results never depend on it.
        // An answer below the rule for any other reason (registers grown in a variant build, WBCQP_DEBUG_LDS_PAD) moves the rule with it and is kept.
        bool distrust = false;
        auto resident_of = [&](const void* kernel, const char* what, int cap, int& occ_out) -> int {
            int occ = 0;
            HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, kThreads, (size_t)lds_bytes));
            hipFuncAttributes fa{};
            HIP_TRY(h, hipFuncGetAttributes(&fa, kernel));
            const int regs = std::max(8, (fa.numRegs + 7) & ~7);
            const int admitted = std::min({(160 * 1024) / (lds_bytes + 16), 512 / regs, cap}); // (+ 16: the granule and the kernel's static word, as measured)
            if (occ < admitted && h->lds_pad == 0) {
                distrust = true;
                occ = admitted;
            }
            if (h->debug_launch)
                std::fprintf(stderr, "wbcqp occupancy: %s lds %d B (+ %d static) VGPRs %d -> %d per CU%s\n", what, lds_bytes, (int)fa.sharedSizeBytes, fa.numRegs, occ,
                             distrust ? " (runtime answered less)" : "");
            occ_out = occ;
            return WBCQP_OK;
        };
        int occ = 0;
        if (int rc = resident_of(reinterpret_cast<const void*>(&solve_queue_kernel<TI, CP, SPEC>), "solve_queue_kernel", 2, occ); rc != WBCQP_OK) return rc;
        if (occ < 1) return fail(h, WBCQP_ERR_HIP, "solve_queue_kernel: no workgroup fits a CU");
        h->queue_occ[V] = occ;
        h->queue_occ_warm[V] = occ;
        if constexpr (kWarm<CP, SPEC>) { // the warm start's twin is a register allocation of its own: its occupancy, not its sibling's
            int occw = 0;
            if (int rc = resident_of(reinterpret_cast<const void*>(&solve_queue_kernel_warm<TI, SPEC>), "solve_queue_kernel_warm", 2, occw); rc != WBCQP_OK) return rc;
            if (occw < 1) return fail(h, WBCQP_ERR_HIP, "solve_queue_kernel_warm: no workgroup fits a CU");
            h->queue_occ_warm[V] = occw;
        }
        h->queue_lds[V] = lds_bytes;
        h->queue_three[V] = false;
        // a workgroup small enough for three on a CU takes the kernel compiled for three waves per SIMD (wbcqp_device.hpp: solve_queue3_kernel)
        if constexpr (kThree<CP, SPEC>) {
            if (lds_bytes <= kLdsThree && h->lds_pad == 0) {
                int occ3 = 0;
                if (int rc = resident_of(reinterpret_cast<const void*>(&solve_queue3_kernel<TI, SPEC>), "solve_queue3_kernel", 3, occ3); rc != WBCQP_OK) return rc;
                if (occ3 >= 3) {
                    h->queue_three[V] = true;
                    h->queue_occ3[V] = occ3;
                }
            }
        }
        if (distrust && !h->warned_occupancy) {
            h->warned_occupancy = true;
            std::fprintf(stderr, "wbcqp: the HIP runtime reports fewer resident workgroups per CU than LDS (%d B) and the kernel's registers admit; launching %d per CU anyway. "
                                 "Two HIP runtimes in this process (libwbcqp.so loaded before torch)?  See INTEGRATION.md.\n", lds_bytes,
                         h->queue_three[V] ? h->queue_occ3[V] : h->queue_occ[V]);
        }
    }
    // three per CU: where the twin holds three AND no group of the launch has actuation bounds (kThree's comment says why)
    bool warm = false; // the handle asked for the warm start's pick hint: the kernels that carry its code
    if constexpr (kWarm<CP, SPEC>) warm = (h->flags & WBCQP_FLAG_WARM_START) != 0;
    bool three = false;
    if constexpr (kThree<CP, SPEC>) {
        three = h->queue_three[V] && !warm;
        for (int g = 0; g < tab.n; ++g) three = three && !tab.g[g].st.act_bounds;
    }
    const int queue_occ = three ? h->queue_occ3[V] : (warm ? h->queue_occ_warm[V] : h->queue_occ[V]);
    if (h->debug_launch)
        std::fprintf(stderr, "wbcqp launch: V %d spec %d total %d lds %d occupancy %d three %d n_cu %d flags 0x%x\n", V, SPEC, total, lds_bytes, queue_occ, (int)three,
                     h->n_cu, (unsigned)h->flags);
    int* queue = nullptr;
    if (osp && !(h->flags & WBCQP_FLAG_HW_DISPATCH) && (lds_bytes >= kQueueMinLds || (three && lds_bytes >= kQueue3MinLds) || (h->flags & WBCQP_FLAG_QUEUE))) {
        if (!os.queue && !h->graph_ord) {
            HIP_TRY(h, hipMalloc(&os.queue, 2 * sizeof(int)));
            HIP_TRY(h, hipMemset(os.queue, 0, 2 * sizeof(int)));
        }
        queue = os.queue;
    }
    if (queue) {
        const long long resident = (long long)queue_occ * h->n_cu;
        if constexpr (kThree<CP, SPEC>) {
            if (three)
                hipLaunchKernelGGL((solve_queue3_kernel<TI, SPEC>), dim3((unsigned)(total < resident ? total : resident)), dim3(kThreads), lds_bytes,
                                   stream, tab, queue, total);
        }
        if constexpr (kWarm<CP, SPEC>) {
            if (warm)
                hipLaunchKernelGGL((solve_queue_kernel_warm<TI, SPEC>), dim3((unsigned)(total < resident ? total : resident)), dim3(kThreads), lds_bytes,
                                   stream, tab, queue, total);
        }
        if (!three && !warm)
            hipLaunchKernelGGL((solve_queue_kernel<TI, CP, SPEC>), dim3((unsigned)(total < resident ? total : resident)), dim3(kThreads), lds_bytes,
                               stream, tab, queue, total);
    }
    else {
        if constexpr (kWarm<CP, SPEC>) {
            if (warm) hipLaunchKernelGGL((solve_kernel_warm<TI, SPEC>), dim3(total), dim3(kThreads), lds_bytes, stream, tab);
        }
        if (!warm) hipLaunchKernelGGL((solve_kernel<TI, CP, SPEC>), dim3(total), dim3(kThreads), lds_bytes, stream, tab);
    }
    HIP_TRY(h, hipGetLastError());
    // the order is renewed every `period` launches: iteration counts drift slowly from tick to tick, the queue absorbs what
    // drift there is, and the two order kernels (4.5 + 15 us) are then a fraction of a launch instead of a twentieth
    const int asked = (h->flags >> WBCQP_FLAG_REFRESH_SHIFT) & 0xff;
    const int period = h->capturing ? 1 : (asked ? asked : kOrderRefresh);
    if (sched && tab.order && os.age + 1 < period)
        ++os.age;
    else if (sched) {
        os.age = 0;
        if (total > os.cap) { // first launch of a larger shape (a graph's buffer has its final size from the start)
            if (h->graph_ord) return fail(h, WBCQP_ERR_INVALID, "captured tick: launch larger than the graph's order buffer");
            HIP_TRY(h, hipStreamSynchronize(stream));
            if (os.order) (void)hipFree(os.order);
            os.order = nullptr;
            os.cap = 0;
            os.total = 0;
            HIP_TRY(h, hipMalloc(&os.order, 2 * sizeof(int) * (size_t)total));
            os.cap = total;
        }
        hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(1024), 0, stream, sa, os.order, total);
        HIP_TRY(h, hipGetLastError());
        // a few QPs per resident workgroup, one structure, taken from the queue: pack the order (pack_order_kernel)
        const long long resident = queue ? (long long)queue_occ * h->n_cu : 0;
        // (with two workgroups per CU the packed order measured no better than plain longest-first: WBCQP_FLAG_QUEUE asks for it)
        os.packed = queue && !(h->flags & WBCQP_FLAG_NO_PACKING) && (queue_occ == 1 || (h->flags & WBCQP_FLAG_QUEUE)) && tab.n == 1 && resident % kPackSubs == 0 &&
                          total % kPackSubs == 0 && total > resident && total <= 8 * resident && total / kPackSubs <= kPackMaxItems;
        if (os.packed) {
            PackArgs pa{tab.g[0].iters, os.order, os.order + os.cap, total, (int)(resident / kPackSubs)};
            hipLaunchKernelGGL(pack_order_kernel, dim3(kPackSubs), dim3(256), 0, stream, pa);
            HIP_TRY(h, hipGetLastError());
        }
        os.total = total;
        os.sig = sig;
        os.stream = stream;
    }
    return WBCQP_OK;
}

// the small structures of a launch: one wavefront per QP, four per workgroup, in table order (no launch order: the QPs are
// short and alike, and 32 of them are resident per CU)
int ensure_pinned(wbcqp_handle* h, Pinned& p, size_t bytes)
{
    if (p.bytes >= bytes) return WBCQP_OK;
    if (p.host) (void)hipHostFree(p.host);
    p.host = nullptr;
    p.bytes = 0;
    HIP_TRY(h, hipHostMalloc(&p.host, bytes, hipHostMallocDefault));
    p.bytes = bytes;
    return WBCQP_OK;
}

template <typename TI>
int launch_small(wbcqp_handle* h, GroupTable<TI>& tab, int total, hipStream_t stream)
{
    if (total == 0) return WBCQP_OK;
    tab.order = nullptr;
    const int lds_bytes = kWaves * sm::COUNT * (int)sizeof(double);
    hipLaunchKernelGGL((solve_small_kernel<TI>), dim3((unsigned)((total + kWaves - 1) / kWaves)), dim3(kThreads), lds_bytes, stream, tab, total);
    HIP_TRY(h, hipGetLastError());
    return WBCQP_OK;
}

int ensure(wbcqp_handle* h, Staging& s, size_t bytes)
{
    if (s.bytes >= bytes) return WBCQP_OK;
    if (s.dev) (void)hipFree(s.dev);
    s.dev = nullptr;
    s.bytes = 0;
    HIP_TRY(h, hipMalloc(&s.dev, bytes));
    s.bytes = bytes;
    return WBCQP_OK;
}

} // namespace

extern "C" {

int wbcqp_version(void) { return WBCQP_VERSION; }

const char* wbcqp_last_error(const wbcqp_handle* handle) { return handle ? handle->err.c_str() : g_create_error.c_str(); }

int wbcqp_layout_of(const wbcqp_structure* st, wbcqp_layout* out)
{
    DevStruct D;
    HostBlocks HB;
    wbcqp_layout L;
    std::string why;
    int rc = derive(st, D, HB, L, why);
    if (rc != WBCQP_OK) return fail(nullptr, rc, why);
    DevStruct C;
    if (derive_compact(D, C)) {
        set_lds(L, C.lds_doubles * 8, true, C.act_bounds != 0);
        L.specialised = spec_of(C);
    }
    // how a row of kSpecDims (wbcqp_types.hpp) is made: the derived sizes and offsets of a stack, in the order of struct Dims
    if (std::getenv("WBCQP_DEBUG_DUMP_STRUCT") && C.compact)
        std::fprintf(stderr, "compact: nv %d na %d nc %d k %d n %d nu %d n_dense %d n_tasks %d n_sel %d n_bound %d act_bounds %d neq %d nin2 %d r1 %d max_iter %d "
                     "ldj %d ldb %d o_J %d o_R %d o_vec %d o_int %d o_pan %d act_off %d lds_doubles %d\n", C.nv, C.na, C.nc, C.k, C.n, C.nu, C.n_dense, C.n_tasks,
                     C.n_sel, C.n_bound, C.act_bounds, C.neq, C.nin2, C.r1, C.max_iter, C.ldj, C.ldb, C.o_J, C.o_R, C.o_vec, C.o_int, C.o_pan, C.act_off, C.lds_doubles);
    L.wave_per_qp = small_ok(D, HB) ? 1 : 0;
    if (out) *out = L;
    return WBCQP_OK;
}

int wbcqp_create(const wbcqp_desc* desc, wbcqp_handle** out)
{
    if (!desc || !out) return fail(nullptr, WBCQP_ERR_INVALID, "desc/out is NULL");
    if (desc->dtype != WBCQP_F64 && desc->dtype != WBCQP_F32) return fail(nullptr, WBCQP_ERR_INVALID, "unknown dtype");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, WBCQP_ERR_NO_DEVICE, "no HIP device visible: wbcqp has no CPU fallback");
    if (desc->device < 0 || desc->device >= count) return fail(nullptr, WBCQP_ERR_INVALID, "device ordinal out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, desc->device);
    if (e != hipSuccess) return fail(nullptr, WBCQP_ERR_HIP, hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, WBCQP_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    e = hipSetDevice(desc->device);
    if (e != hipSuccess) return fail(nullptr, WBCQP_ERR_HIP, hipGetErrorString(e));
    wbcqp_handle* h = new wbcqp_handle();
    h->device = desc->device;
    h->dtype = desc->dtype;
    h->flags = desc->flags;
    h->n_cu = prop.multiProcessorCount;
    for (int& q : h->queue_lds) q = -1;
    if (const char* pad = std::getenv("WBCQP_DEBUG_LDS_PAD")) h->lds_pad = std::atoi(pad);
    h->debug_launch = std::getenv("WBCQP_DEBUG_LAUNCH") != nullptr;
    h->no_ffcache = std::getenv("WBCQP_DEBUG_NO_FFCACHE") != nullptr;
    *out = h;
    return WBCQP_OK;
}

int wbcqp_destroy(wbcqp_handle* h)
{
    if (!h) return WBCQP_OK;
    (void)hipSetDevice(h->device);
    for (auto& s : h->slots) release(s);
    if (h->stage_in.dev) (void)hipFree(h->stage_in.dev);
    if (h->stage_out.dev) (void)hipFree(h->stage_out.dev);
    if (h->pin_in.host) (void)hipHostFree(h->pin_in.host);
    if (h->pin_out.host) (void)hipHostFree(h->pin_out.host);
    if (h->roll_rec.dev) (void)hipFree(h->roll_rec.dev);
    if (h->roll_state.dev) (void)hipFree(h->roll_state.dev);
    for (auto& r : h->roll_subs) {
        if (r.stream) (void)hipStreamDestroy(r.stream);
        if (r.done) (void)hipEventDestroy(r.done);
        if (r.ord.order) (void)hipFree(r.ord.order);
        if (r.ord.queue) (void)hipFree(r.ord.queue);
    }
    if (h->roll_start) (void)hipEventDestroy(h->roll_start);
    if (h->roll_done) (void)hipEventDestroy(h->roll_done);
    for (auto& mz : h->roll_meas) {
        if (mz.t0) (void)hipEventDestroy(mz.t0);
        if (mz.t1) (void)hipEventDestroy(mz.t1);
    }
    for (auto& ss : h->streams) {
        if (ss.ord.order) (void)hipFree(ss.ord.order);
        if (ss.ord.queue) (void)hipFree(ss.ord.queue);
    }
    delete h;
    return WBCQP_OK;
}

int wbcqp_set_structure(wbcqp_handle* h, int slot, const wbcqp_structure* st)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (slot < 0 || slot >= WBCQP_MAX_STRUCTURES) return fail(h, WBCQP_ERR_INVALID, "slot out of range");
    DevStruct D;
    HostBlocks HB;
    wbcqp_layout L;
    std::string why;
    int rc = derive(st, D, HB, L, why);
    if (rc != WBCQP_OK) return fail(h, rc, why);
    D.ffc = nullptr;
    HIP_TRY(h, hipSetDevice(h->device));
    Slot& s = h->slots[slot];
    release(s);
    s.ffc_dev = nullptr;
    s.ffc_built = false;
    const int nc = D.nc;
    // F'F and F' of the force-regularisation block, F = diag(w_f) T  (6 x 12)
    std::vector<double> ftf((size_t)nc * 144 + 1, 0.0), ft((size_t)nc * 72 + 1, 0.0);
    for (int c = 0; c < nc; ++c) {
        const double* F = st->forcereg_mat + (size_t)c * 72;
        for (int a = 0; a < 12; ++a) {
            for (int b = 0; b < 12; ++b) {
                double acc = 0.0;
                for (int q = 0; q < 6; ++q) acc += F[q * 12 + a] * F[q * 12 + b];
                ftf[(size_t)c * 144 + a * 12 + b] = acc;
            }
            for (int q = 0; q < 6; ++q) ft[(size_t)c * 72 + a * 6 + q] = F[q * 12 + a];
        }
    }
#define UP(field, src, count)                                                   \
    do {                                                                        \
        int rc_ = upload(h, s, src, (size_t)(count), &D.field);                 \
        if (rc_ != WBCQP_OK) { release(s); return rc_; }                        \
    } while (0)
    UP(dense_row_task, st->dense_row_task, D.n_dense);
    UP(sel_col, st->sel_col, D.n_sel);
    UP(sel_task, st->sel_task, D.n_sel);
    UP(forcereg_task, st->forcereg_task, nc);
    UP(bound_col, st->bound_col, D.n_bound);
    UP(acteq_joint, st->acteq_joint, D.n_acteq);
    UP(acteq_scale, st->acteq_scale, D.n_acteq);
    UP(force_gen, st->force_gen, nc * 72);
    UP(ftf, ftf.data(), nc * 144);
    UP(ft, ft.data(), nc * 72);
    UP(fric_mat, st->fric_mat, nc * 17 * 12);
    UP(fric_lb, st->fric_lb, nc * 17);
    UP(fric_ub, st->fric_ub, nc * 17);
    {
        std::vector<int> meta((size_t)D.nin2 + 1, 0);
        for (int b = 0; b < HB.n_blocks; ++b) {
            const int rows = HB.blk_rows[b], off = HB.blk_off[b], kind = HB.blk_kind[b];
            for (int r = 0; r < rows; ++r) {
                const int col = (kind == WBCQP_INEQ_BOUNDS) ? st->bound_col[r] : 0;
                const int ct = (kind == WBCQP_INEQ_FORCE) ? HB.blk_arg[b] : 0;
                meta[off + r] = row_meta_pack(kind, 0, r, ct, col);
                meta[off + rows + r] = row_meta_pack(kind, 1, r, ct, col);
            }
        }
        UP(rowmeta, meta.data(), D.nin2);
    }
    {
        std::vector<unsigned> mp((size_t)D.nv * (D.nv + 1) / 2 + 1, 0u);
        for (int i = 0; i < D.nv; ++i)
            for (int j = 0; j <= i; ++j)
                mp[(size_t)i * (i + 1) / 2 + j] = (unsigned)(i * D.ldm + j) | ((unsigned)(j * D.ldm + i) << 16);
        UP(mpack, mp.data(), D.nv * (D.nv + 1) / 2);
    }
    {
        std::vector<unsigned> ap((size_t)D.n_dense * D.nv + 1, 0u);
        for (int r = 0; r < D.n_dense; ++r)
            for (int col = 0; col < D.nv; ++col) ap[(size_t)r * D.nv + col] = (unsigned)(r * 64 + ((col >> 5) & 1) * 32 + (col & 15) * 2 + ((col >> 4) & 1)); // see wbcqp_types.hpp, apack
        UP(apack, ap.data(), D.n_dense * D.nv);
    }
    {
        std::vector<unsigned> cpk((size_t)nc * 6 * D.nv + 1, 0u);
        for (int rr = 0; rr < nc * 6; ++rr)
            for (int kk = 0; kk < D.nv; ++kk) cpk[(size_t)rr * D.nv + kk] = (unsigned)(kk * D.ldb + D.nu + rr);
        UP(acpack, cpk.data(), nc * 6 * D.nv);
    }
#undef UP
    s.host = D;
    s.sel_col_h.assign(st->sel_col, st->sel_col + D.n_sel);
    s.force_gen_h.assign(st->force_gen, st->force_gen + (size_t)nc * 72);
    s.lds_full = D.lds_doubles * 8;
    s.lds_cp = 0;
    s.host_cp = DevStruct{};
    if (!(h->flags & WBCQP_FLAG_FULL_LDS) && derive_compact(D, s.host_cp)) {
        s.lds_cp = s.host_cp.lds_doubles * 8;
        set_lds(L, s.lds_cp, true, s.host_cp.act_bounds != 0);
        if (nc > 0 && !h->no_ffcache) { // room for the force blocks' factor (DevStruct::ffc), invalid (weight NaN) until the slot's first launch makes it
            std::vector<double> init((size_t)nc * kFfcStride, 0.0);
            for (int c = 0; c < nc; ++c) init[(size_t)c * kFfcStride] = std::numeric_limits<double>::quiet_NaN();
            const double* dev = nullptr;
            rc = upload(h, s, init.data(), init.size(), &dev);
            if (rc != WBCQP_OK) { release(s); return rc; }
            s.ffc_dev = const_cast<double*>(dev);
            s.host_cp.ffc = dev;
        }
    }
    s.small = small_ok(D, HB);
    L.wave_per_qp = s.small ? 1 : 0;
    s.spec = s.host_cp.compact ? spec_of(s.host_cp) : 0;
    L.specialised = s.spec;
    s.layout = L;
    s.set = true;
    return WBCQP_OK;
}

int wbcqp_solve_ragged(wbcqp_handle* h, int n_groups, const wbcqp_group* groups, void* stream)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (n_groups < 0 || n_groups > kMaxGroups) return fail(h, WBCQP_ERR_INVALID, "n_groups must be in [0, 8]");
    if (n_groups > 0 && !groups) return fail(h, WBCQP_ERR_INVALID, "groups is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    int total = 0, lds = 0, used = 0, total_small = 0, used_small = 0;
    GroupTable<double> t64{}, s64{};
    GroupTable<float> t32{}, s32{};
    const bool wave_per_qp = !(h->flags & WBCQP_FLAG_WORKGROUP_PER_QP) && !h->dbg; // (the stamped diagnostic build profiles the four-wave kernels)
    // the compact kernel runs a launch whose groups are all eligible; one group that is not puts the launch on the full layout
    bool compact = true;
    for (int g = 0; g < n_groups; ++g) {
        const wbcqp_group& G = groups[g];
        if (G.slot < 0 || G.slot >= WBCQP_MAX_STRUCTURES || !h->slots[G.slot].set)
            return fail(h, WBCQP_ERR_INVALID, "group uses a slot with no structure");
        if (G.batch > 0 && !(wave_per_qp && h->slots[G.slot].small) && !h->slots[G.slot].host_cp.compact) compact = false;
    }
    for (int g = 0; g < n_groups; ++g) {
        const wbcqp_group& G = groups[g];
        Slot& s = h->slots[G.slot];
        int rc = check_io(h, s, G.batch, &G.in, &G.out);
        if (rc != WBCQP_OK) return rc;
        if (G.batch == 0) continue;
        bool user_capture = false; // (a caller capturing this very call into a graph of its own: no synchronisation there -- the cache waits for a plain launch)
        if (compact && s.ffc_dev && !s.ffc_built && !h->capturing) {
            hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cst) != hipSuccess) (void)hipGetLastError();
            user_capture = cst != hipStreamCaptureStatusNone;
        }
        if (compact && s.ffc_dev && !s.ffc_built && !h->capturing && !user_capture && !(wave_per_qp && s.small)) {
            // the slot's first compact launch: the force blocks' factor for the weights of its first QP, by the kernels' own code, ahead of the solve on its
            // stream; waited for once, so that a launch on another stream never meets a half-written entry
            if (h->dtype == WBCQP_F64) hipLaunchKernelGGL(ffcache_kernel<double>, dim3(1), dim3(128), 0, static_cast<hipStream_t>(stream), s.host_cp, static_cast<const double*>(G.in.w), s.ffc_dev);
            else hipLaunchKernelGGL(ffcache_kernel<float>, dim3(1), dim3(128), 0, static_cast<hipStream_t>(stream), s.host_cp, static_cast<const float*>(G.in.w), s.ffc_dev);
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
            s.ffc_built = true;
        }
        if (wave_per_qp && s.small) { // one wavefront per QP: a launch of their own (wbcqp_small.hpp)
            if (h->dtype == WBCQP_F64) fill_group(s64.g[used_small], s, false, G.batch, &G.in, &G.out);
            else fill_group(s32.g[used_small], s, false, G.batch, &G.in, &G.out);
            ++used_small;
            total_small += G.batch;
            continue;
        }
        if (h->dtype == WBCQP_F64) { fill_group(t64.g[used], s, compact, G.batch, &G.in, &G.out); t64.g[used].dbg = h->dbg; t64.g[used].warm = (h->flags & WBCQP_FLAG_WARM_START) ? 1 : 0; }
        else { fill_group(t32.g[used], s, compact, G.batch, &G.in, &G.out); t32.g[used].dbg = h->dbg; t32.g[used].warm = (h->flags & WBCQP_FLAG_WARM_START) ? 1 : 0; }
        ++used;
        total += G.batch;
        const int need = compact ? s.lds_cp : s.lds_full;
        if (need > lds) lds = need;
    }
    t64.n = used;
    t32.n = used;
    s64.n = used_small;
    s32.n = used_small;
    if (h->lds_pad > 0) lds = std::min(lds + h->lds_pad, 160 * 1024);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (total_small > 0) {
        const int rc = (h->dtype == WBCQP_F64) ? launch_small<double>(h, s64, total_small, hs) : launch_small<float>(h, s32, total_small, hs);
        if (rc != WBCQP_OK) return rc;
    }
    // a launch of ONE group whose structure is a shipped stack takes that stack's instantiation of the compact kernel (sizes and offsets as
    // literals: wbcqp_types.hpp); anything else -- ragged launches, other structures, WBCQP_FLAG_GENERIC_KERNEL -- the generic one.  Same bits.
    int spec = 0;
    if (compact && used == 1 && !(h->flags & WBCQP_FLAG_GENERIC_KERNEL) && h->lds_pad == 0)
        for (int g = 0; g < n_groups; ++g)
            if (groups[g].batch > 0 && !(wave_per_qp && h->slots[groups[g].slot].small)) spec = h->slots[groups[g].slot].spec;
    if ((h->flags & WBCQP_FLAG_WARM_START) && spec > 1) spec = 0; // the hint's code lives in the generic kernel and Talos's (kWarm)
    if (h->dtype == WBCQP_F64) {
        if (!compact) return launch<double, false>(h, t64, total, lds, hs);
        switch (spec) {
        case 1: return launch<double, true, 1>(h, t64, total, lds, hs);
        case 2: return launch<double, true, 2>(h, t64, total, lds, hs);
        case 3: return launch<double, true, 3>(h, t64, total, lds, hs);
        default: return launch<double, true>(h, t64, total, lds, hs);
        }
    }
    if (!compact) return launch<float, false>(h, t32, total, lds, hs);
    switch (spec) {
    case 1: return launch<float, true, 1>(h, t32, total, lds, hs);
    case 2: return launch<float, true, 2>(h, t32, total, lds, hs);
    case 3: return launch<float, true, 3>(h, t32, total, lds, hs);
    default: return launch<float, true>(h, t32, total, lds, hs);
    }
}

int wbcqp_solve_batch(wbcqp_handle* h, int slot, int batch, const wbcqp_inputs* in, const wbcqp_outputs* out, void* stream)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (batch == 0) return WBCQP_OK;
    if (!in || !out) return fail(h, WBCQP_ERR_INVALID, "inputs/outputs struct is NULL");
    wbcqp_group G;
    G.slot = slot;
    G.batch = batch;
    G.in = *in;
    G.out = *out;
    return wbcqp_solve_ragged(h, 1, &G, stream);
}

int wbcqp_solve_batch_host(wbcqp_handle* h, int slot, int batch, const wbcqp_inputs* in, const wbcqp_outputs* out)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (slot < 0 || slot >= WBCQP_MAX_STRUCTURES || !h->slots[slot].set) return fail(h, WBCQP_ERR_INVALID, "slot has no structure");
    const Slot& s = h->slots[slot];
    int rc = check_io(h, s, batch, in, out);
    if (rc != WBCQP_OK || batch == 0) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const wbcqp_layout& L = s.layout;
    const size_t es = (h->dtype == WBCQP_F64) ? 8 : 4;
    constexpr int NF = 12;
    const int lens[NF] = {L.len_M, L.len_h, L.len_A, L.len_b1, L.len_Ac, L.len_bc, L.len_blb, L.len_bub, L.len_tlb, L.len_tub, L.len_w, L.len_Acop};
    const void* src[NF] = {in->M, in->h, in->A, in->b1, in->Ac, in->bc, in->blb, in->bub, in->tlb, in->tub, in->w, in->Acop};
    size_t in_bytes = 0;
    size_t offs[NF];
    for (int f = 0; f < NF; ++f) {
        offs[f] = in_bytes;
        in_bytes += (((size_t)lens[f] * batch * es) + 255) & ~(size_t)255;
    }
    rc = ensure(h, h->stage_in, in_bytes + 256);
    if (rc != WBCQP_OK) return rc;
    const size_t o_x = 0;
    const size_t o_tau = (o_x + (size_t)L.n * batch * es + 255) & ~(size_t)255;
    const size_t o_obj = (o_tau + (size_t)s.host.na * batch * es + 255) & ~(size_t)255;
    const size_t o_st = (o_obj + (size_t)batch * es + 255) & ~(size_t)255;
    const size_t o_it = o_st + (((size_t)batch * 4 + 255) & ~(size_t)255);
    const size_t o_na = o_it + (((size_t)batch * 4 + 255) & ~(size_t)255);
    const size_t o_am = o_na + (((size_t)batch * 4 + 255) & ~(size_t)255);
    const size_t out_bytes = o_am + (size_t)batch * 32 + 256;
    rc = ensure(h, h->stage_out, out_bytes);
    if (rc != WBCQP_OK) return rc;
    char* din = static_cast<char*>(h->stage_in.dev);
    char* dout = static_cast<char*>(h->stage_out.dev);
    const bool packed = in_bytes + (size_t)batch * 32 <= kPackedBytes && out_bytes <= kPackedBytes;
    if (packed) {
        rc = ensure_pinned(h, h->pin_in, kPackedBytes);
        if (rc != WBCQP_OK) return rc;
        rc = ensure_pinned(h, h->pin_out, kPackedBytes);
        if (rc != WBCQP_OK) return rc;
        char* pin = static_cast<char*>(h->pin_in.host);
        for (int f = 0; f < NF; ++f)
            if (lens[f] > 0) std::memcpy(pin + offs[f], src[f], (size_t)lens[f] * batch * es);
        HIP_TRY(h, hipMemcpyAsync(din, pin, in_bytes, hipMemcpyHostToDevice, nullptr));
    }
    else
        for (int f = 0; f < NF; ++f)
            if (lens[f] > 0) HIP_TRY(h, hipMemcpyAsync(din + offs[f], src[f], (size_t)lens[f] * batch * es, hipMemcpyHostToDevice, nullptr));
    wbcqp_inputs di = {din + offs[0], din + offs[1], din + offs[2], din + offs[3], din + offs[4], din + offs[5],
                       din + offs[6], din + offs[7], din + offs[8], din + offs[9], din + offs[10], din + offs[11]};
    wbcqp_outputs dso{};
    dso.x = dout + o_x; dso.tau = dout + o_tau; dso.objective = dout + o_obj;
    dso.status = reinterpret_cast<int32_t*>(dout + o_st); dso.iters = reinterpret_cast<int32_t*>(dout + o_it);
    dso.n_active = reinterpret_cast<int32_t*>(dout + o_na);
    if (out->active_mask) { // in/out: the hint goes up (zeros where the caller has none), the solution's active set comes back
        dso.active_mask = reinterpret_cast<uint32_t*>(dout + o_am);
        HIP_TRY(h, hipMemcpyAsync(dso.active_mask, out->active_mask, (size_t)batch * 32, hipMemcpyHostToDevice, nullptr));
    }
    rc = wbcqp_solve_batch(h, slot, batch, &di, &dso, nullptr);
    if (rc != WBCQP_OK) return rc;
    if (packed) { // one copy down, then the arrays are taken apart on the host
        char* po = static_cast<char*>(h->pin_out.host);
        HIP_TRY(h, hipMemcpyAsync(po, dout, out_bytes, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(h, hipStreamSynchronize(nullptr));
        std::memcpy(out->x, po + o_x, (size_t)L.n * batch * es);
        if (s.host.na > 0) std::memcpy(out->tau, po + o_tau, (size_t)s.host.na * batch * es);
        std::memcpy(out->status, po + o_st, (size_t)batch * 4);
        std::memcpy(out->iters, po + o_it, (size_t)batch * 4);
        if (out->objective) std::memcpy(out->objective, po + o_obj, (size_t)batch * es);
        if (out->n_active) std::memcpy(out->n_active, po + o_na, (size_t)batch * 4);
        if (out->active_mask) std::memcpy(out->active_mask, po + o_am, (size_t)batch * 32);
        return WBCQP_OK;
    }
    if (out->active_mask) HIP_TRY(h, hipMemcpyAsync(out->active_mask, dso.active_mask, (size_t)batch * 32, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(h, hipMemcpyAsync(out->x, dso.x, (size_t)L.n * batch * es, hipMemcpyDeviceToHost, nullptr));
    if (s.host.na > 0) HIP_TRY(h, hipMemcpyAsync(out->tau, dso.tau, (size_t)s.host.na * batch * es, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(h, hipMemcpyAsync(out->status, dso.status, (size_t)batch * 4, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(h, hipMemcpyAsync(out->iters, dso.iters, (size_t)batch * 4, hipMemcpyDeviceToHost, nullptr));
    if (out->objective) HIP_TRY(h, hipMemcpyAsync(out->objective, dso.objective, (size_t)batch * es, hipMemcpyDeviceToHost, nullptr));
    if (out->n_active) HIP_TRY(h, hipMemcpyAsync(out->n_active, dso.n_active, (size_t)batch * 4, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(h, hipStreamSynchronize(nullptr));
    return WBCQP_OK;
}

int wbcqp_solve_dense(wbcqp_handle* h, int batch, int n, int neq, int nin, int max_iter, const wbcqp_dense_inputs* in,
                      const wbcqp_outputs* out, void* stream)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (batch < 0 || n <= 0 || neq < 0 || nin < 0) return fail(h, WBCQP_ERR_INVALID, "bad batch / n / neq / nin");
    if (batch == 0) return WBCQP_OK;
    if (!in || !out) return fail(h, WBCQP_ERR_INVALID, "inputs/outputs struct is NULL");
    if (!in->H || !in->g || (neq > 0 && (!in->CE || !in->ce0)) || (nin > 0 && (!in->CI || !in->ci0)))
        return fail(h, WBCQP_ERR_INVALID, "dense QP: H, g and the constraint arrays of non-empty blocks are required");
    if (!out->x || !out->status || !out->iters) return fail(h, WBCQP_ERR_INVALID, "output arrays x, status, iters are required");
    if (n > kDenseMaxVars || nin > kDenseMaxIneq || neq > n) return fail(h, WBCQP_ERR_UNSUPPORTED, "dense QP: n <= 126, nin <= 512, neq <= n");
    DenseArgs a{};
    a.n = n; a.neq = neq; a.nin = nin; a.ldj = odd(n);
    int o = 0;
    auto take = [&](int count) { int at = o; o += (count + 1) & ~1; return at; };
    a.blocked_eq = (neq >= 1 && neq <= 22 && n <= 80) ? 1 : 0;
    a.ldb = 2 * odd((4 * ((neq + 3) / 4) + 1) / 2); // as the structured layout's (derive): rows stay 16-byte aligned for the 4-wide column groups
    a.o_J = take(n * a.ldj);
    int rsize = n * (n + 3) / 2 + 2;
    if (a.blocked_eq && 256 + (n + 17) * a.ldb + 8 > rsize) rsize = 256 + (n + 17) * a.ldb + 8; // B of the blocked equality phase
    a.o_R = take(rsize);
    a.o_vec = take(V_COUNT * kSlot);
    a.o_eqw = take(a.blocked_eq ? (n + 1) * a.ldb + 8 : 0);
    a.o_eqt = take(a.blocked_eq ? neq * (neq + 1) + 4 * neq + 16 : 0);
    a.o_int = o;
    o += kIntCount / 2 + 2;
    const int lds_bytes = o * 8;
    if (lds_bytes > 160 * 1024) return fail(h, WBCQP_ERR_UNSUPPORTED, "dense QP does not fit the 160 KiB LDS of one CU (n <= 96)");
    a.max_iter = max_iter > 0 ? max_iter : 1000;
    a.count = batch;
    a.H = in->H; a.g = in->g; a.CE = in->CE; a.ce0 = in->ce0; a.CI = in->CI; a.ci0 = in->ci0;
    a.x = out->x; a.objective = out->objective; a.status = out->status; a.iters = out->iters; a.n_active = out->n_active;
    HIP_TRY(h, hipSetDevice(h->device));
    if (lds_bytes > h->dense_max_lds) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_dense_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_dense_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        h->dense_max_lds = lds_bytes;
    }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (h->dtype == WBCQP_F64) hipLaunchKernelGGL(solve_dense_kernel<double>, dim3(batch), dim3(kThreads), lds_bytes, hs, a);
    else hipLaunchKernelGGL(solve_dense_kernel<float>, dim3(batch), dim3(kThreads), lds_bytes, hs, a);
    HIP_TRY(h, hipGetLastError());
    return WBCQP_OK;
}

int wbcqp_solve_dense_host(wbcqp_handle* h, int batch, int n, int neq, int nin, int max_iter, const wbcqp_dense_inputs* in,
                           const wbcqp_dense_output** result)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (!in || !result) return fail(h, WBCQP_ERR_INVALID, "inputs / result is NULL");
    if (batch <= 0 || n <= 0 || neq < 0 || nin < 0) return fail(h, WBCQP_ERR_INVALID, "bad batch / n / neq / nin");
    *result = nullptr;
    HIP_TRY(h, hipSetDevice(h->device));
    const bool f32 = h->dtype != WBCQP_F64;
    const size_t es = f32 ? 4 : 8;
    const size_t B = (size_t)batch;
    const size_t lens[6] = {(size_t)n * n, (size_t)n, (size_t)neq * n, (size_t)neq, (size_t)nin * n, (size_t)nin};
    const void* src[6] = {in->H, in->g, in->CE, in->ce0, in->CI, in->ci0};
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t offs[6], in_bytes = 0;
    for (int f = 0; f < 6; ++f) { offs[f] = in_bytes; in_bytes += al(lens[f] * B * es); }
    int rc = ensure(h, h->stage_in, in_bytes + 256);
    if (rc != WBCQP_OK) return rc;
    const size_t o_x = 0, o_obj = al((size_t)n * B * es), o_st = o_obj + al(B * es), o_it = o_st + al(B * 4), o_na = o_it + al(B * 4);
    rc = ensure(h, h->stage_out, o_na + al(B * 4) + 256);
    if (rc != WBCQP_OK) return rc;
    char* din = static_cast<char*>(h->stage_in.dev);
    char* dout = static_cast<char*>(h->stage_out.dev);
    std::vector<float> tmp;
    for (int f = 0; f < 6; ++f)
        if (lens[f] != 0 && !src[f]) return fail(h, WBCQP_ERR_INVALID, "dense QP: a required input array is NULL");
    // a small batch (the reference's own use: one QP per call) crosses PCIe as ONE page-locked copy each way, as in
    // wbcqp_solve_batch_host: six pageable copies up and five synchronous ones down were a quarter of a single QP's wall time
    const size_t out_total = o_na + al(B * 4);
    const bool packed = !f32 && in_bytes <= kPackedBytes && out_total <= kPackedBytes;
    if (packed) {
        rc = ensure_pinned(h, h->pin_in, kPackedBytes);
        if (rc != WBCQP_OK) return rc;
        rc = ensure_pinned(h, h->pin_out, kPackedBytes);
        if (rc != WBCQP_OK) return rc;
        char* pin = static_cast<char*>(h->pin_in.host);
        for (int f = 0; f < 6; ++f)
            if (lens[f] != 0) std::memcpy(pin + offs[f], src[f], lens[f] * B * 8);
        HIP_TRY(h, hipMemcpyAsync(din, pin, in_bytes, hipMemcpyHostToDevice, nullptr));
    }
    for (int f = 0; f < 6 && !packed; ++f) {
        if (lens[f] == 0) continue;
        if (f32) { // the caller's arrays are double (Eigen); an F32 handle carries float at the device boundary
            tmp.resize(lens[f] * B);
            const double* sd = static_cast<const double*>(src[f]);
            for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = (float)sd[i];
            HIP_TRY(h, hipMemcpy(din + offs[f], tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
        }
        else HIP_TRY(h, hipMemcpyAsync(din + offs[f], src[f], lens[f] * B * 8, hipMemcpyHostToDevice, nullptr));
    }
    wbcqp_dense_inputs di = {din + offs[0], din + offs[1], neq ? din + offs[2] : nullptr, neq ? din + offs[3] : nullptr,
                             nin ? din + offs[4] : nullptr, nin ? din + offs[5] : nullptr};
    wbcqp_outputs dso{};
    dso.x = dout + o_x; dso.objective = dout + o_obj;
    dso.status = reinterpret_cast<int32_t*>(dout + o_st); dso.iters = reinterpret_cast<int32_t*>(dout + o_it);
    dso.n_active = reinterpret_cast<int32_t*>(dout + o_na);
    rc = wbcqp_solve_dense(h, batch, n, neq, nin, max_iter, &di, &dso, nullptr);
    if (rc != WBCQP_OK) return rc;
    h->dense_x.assign((size_t)n * B, 0.0);
    h->dense_obj.assign(B, 0.0);
    h->dense_status.assign(B, WBCQP_HQP_UNKNOWN);
    h->dense_iters.assign(B, 0);
    h->dense_nact.assign(B, 0);
    if (packed) {
        char* po = static_cast<char*>(h->pin_out.host);
        HIP_TRY(h, hipMemcpyAsync(po, dout, out_total, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(h, hipStreamSynchronize(nullptr));
        std::memcpy(h->dense_x.data(), po + o_x, (size_t)n * B * 8);
        std::memcpy(h->dense_obj.data(), po + o_obj, B * 8);
        std::memcpy(h->dense_status.data(), po + o_st, B * 4);
        std::memcpy(h->dense_iters.data(), po + o_it, B * 4);
        std::memcpy(h->dense_nact.data(), po + o_na, B * 4);
        h->dense_out = {batch, n, h->dense_x.data(), h->dense_status.data(), h->dense_iters.data(), h->dense_obj.data(), h->dense_nact.data()};
        *result = &h->dense_out;
        return WBCQP_OK;
    }
    if (f32) {
        std::vector<float> xf((size_t)n * B), of(B);
        HIP_TRY(h, hipMemcpy(xf.data(), dso.x, xf.size() * 4, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(of.data(), dso.objective, B * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < xf.size(); ++i) h->dense_x[i] = xf[i];
        for (size_t i = 0; i < B; ++i) h->dense_obj[i] = of[i];
    }
    else {
        HIP_TRY(h, hipMemcpy(h->dense_x.data(), dso.x, (size_t)n * B * 8, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(h->dense_obj.data(), dso.objective, B * 8, hipMemcpyDeviceToHost));
    }
    HIP_TRY(h, hipMemcpy(h->dense_status.data(), dso.status, B * 4, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(h->dense_iters.data(), dso.iters, B * 4, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(h->dense_nact.data(), dso.n_active, B * 4, hipMemcpyDeviceToHost));
    h->dense_out = {batch, n, h->dense_x.data(), h->dense_status.data(), h->dense_iters.data(), h->dense_obj.data(), h->dense_nact.data()};
    *result = &h->dense_out;
    return WBCQP_OK;
}

// RCCL is resolved at run time from whatever librccl the process already has (PyTorch's, or the
// system one): the library itself carries no link-time dependency on it.
int wbcqp_allgather_tau(wbcqp_handle* h, void* comm, const void* send, void* recv, size_t count, void* stream)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (!comm || !send || !recv) return fail(h, WBCQP_ERR_INVALID, "comm/send/recv is NULL");
    typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, void*);
    static allgather_fn fn = nullptr;
    if (!fn) {
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (!sym) {
            void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (lib) sym = dlsym(lib, "ncclAllGather");
        }
        if (!sym) return fail(h, WBCQP_ERR_RCCL, "ncclAllGather not found (librccl not loadable)");
        fn = reinterpret_cast<allgather_fn>(sym);
    }
    const int nccl_dtype = (h->dtype == WBCQP_F64) ? 8 /* ncclFloat64 */ : 7 /* ncclFloat32 */;
    int rc = fn(send, recv, count, nccl_dtype, comm, stream);
    if (rc != 0) return fail(h, WBCQP_ERR_RCCL, "ncclAllGather failed with code " + std::to_string(rc));
    return WBCQP_OK;
}

// Diagnostic hook, not part of include/wbcqp.h: per-QP phase cycle counters ([batch][20] int64, device memory)
// are written only by a library built with -DWBCQP_STAMPS (inria_wbc_amd/build.py --stamps); single group only.
int wbcqp_launch_order(wbcqp_handle* h, int32_t* order, int32_t capacity, int32_t* packed)
{
    if (!h || !order || capacity < 0) return WBCQP_ERR_INVALID;
    if (packed) *packed = 0;
    if (h->last_stream < 0 || h->last_stream >= (int)h->streams.size()) return 0;
    const OrderState& o = h->streams[h->last_stream].ord; // the stream of the most recent launch
    if (!o.order || o.total <= 0) return 0;
    if (capacity < o.total) return fail(h, WBCQP_ERR_INVALID, "wbcqp_launch_order: capacity below the order's length");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(order, o.order + (o.packed ? o.cap : 0), sizeof(int) * (size_t)o.total, hipMemcpyDeviceToHost));
    if (packed) *packed = o.packed ? 1 : 0;
    return o.total;
}

#ifdef WBCQP_STAMPS
// exported by the DIAGNOSTIC library only (libwbcqp_stamps.so, inria_wbc_amd/build.py --stamps): the product library exports exactly
// what include/wbcqp.h declares
int wbcqp_debug_set_stamp_buffer(wbcqp_handle* h, void* dev_ptr)
{
    if (!h) return WBCQP_ERR_INVALID;
    h->dbg = static_cast<long long*>(dev_ptr);
    return WBCQP_OK;
}
#endif

static int integrate_impl(wbcqp_handle* h, int batch, int nv, int floating_base, double dt, const void* q, const void* dq, const void* x,
                          int ldx, const int32_t* status, void* q_next, void* v_next, void* q_solver, void* stream, const RollAcc& acc);

int wbcqp_integrate(wbcqp_handle* h, int batch, int nv, int floating_base, double dt, const void* q, const void* dq, const void* x,
                    int ldx, const int32_t* status, void* q_next, void* v_next, void* q_solver, void* stream)
{
    return integrate_impl(h, batch, nv, floating_base, dt, q, dq, x, ldx, status, q_next, v_next, q_solver, stream, RollAcc{});
}

static int integrate_impl(wbcqp_handle* h, int batch, int nv, int floating_base, double dt, const void* q, const void* dq, const void* x,
                          int ldx, const int32_t* status, void* q_next, void* v_next, void* q_solver, void* stream, const RollAcc& acc)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (batch < 0 || nv <= 0 || ldx < nv) return fail(h, WBCQP_ERR_INVALID, "bad batch / nv / ldx");
    if (floating_base && nv < 6) return fail(h, WBCQP_ERR_INVALID, "a floating base needs nv >= 6");
    if (batch == 0) return WBCQP_OK;
    if (!q || !dq || !x || !q_next || !v_next) return fail(h, WBCQP_ERR_INVALID, "q / dq / x / q_next / v_next is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    const dim3 grid((batch + 3) / 4), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (h->dtype == WBCQP_F64)
        hipLaunchKernelGGL(integrate_kernel<double>, grid, block, 0, st, batch, nv, floating_base ? 1 : 0, dt,
                           static_cast<const double*>(q), static_cast<const double*>(dq), static_cast<const double*>(x), ldx, status,
                           static_cast<double*>(q_next), static_cast<double*>(v_next), static_cast<double*>(q_solver), acc);
    else
        hipLaunchKernelGGL(integrate_kernel<float>, grid, block, 0, st, batch, nv, floating_base ? 1 : 0, dt,
                           static_cast<const float*>(q), static_cast<const float*>(dq), static_cast<const float*>(x), ldx, status,
                           static_cast<float*>(q_next), static_cast<float*>(v_next), static_cast<float*>(q_solver), acc);
    HIP_TRY(h, hipGetLastError());
    return WBCQP_OK;
}

int wbcqp_integrate_host(wbcqp_handle* h, int batch, int nv, int floating_base, double dt, const void* q, const void* dq,
                         const void* x, int ldx, const int32_t* status, void* q_next, void* v_next, void* q_solver)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (batch < 0 || nv <= 0 || ldx < nv) return fail(h, WBCQP_ERR_INVALID, "bad batch / nv / ldx");
    if (batch == 0) return WBCQP_OK;
    if (!q || !dq || !x || !q_next || !v_next) return fail(h, WBCQP_ERR_INVALID, "q / dq / x / q_next / v_next is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t es = (h->dtype == WBCQP_F64) ? 8 : 4;
    const int nq = floating_base ? nv + 1 : nv;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t bq = al((size_t)batch * nq * es), bv = al((size_t)batch * nv * es), bx = al((size_t)batch * ldx * es),
                 bs = al((size_t)batch * 4);
    int rc = ensure(h, h->stage_in, bq + bv + bx + bs + 256);
    if (rc != WBCQP_OK) return rc;
    rc = ensure(h, h->stage_out, bq + 2 * bv + 256);
    if (rc != WBCQP_OK) return rc;
    char* din = static_cast<char*>(h->stage_in.dev);
    char* dout = static_cast<char*>(h->stage_out.dev);
    HIP_TRY(h, hipMemcpy(din, q, (size_t)batch * nq * es, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(din + bq, dq, (size_t)batch * nv * es, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(din + bq + bv, x, (size_t)batch * ldx * es, hipMemcpyHostToDevice));
    if (status) HIP_TRY(h, hipMemcpy(din + bq + bv + bx, status, (size_t)batch * 4, hipMemcpyHostToDevice));
    rc = wbcqp_integrate(h, batch, nv, floating_base, dt, din, din + bq, din + bq + bv, ldx,
                         status ? reinterpret_cast<const int32_t*>(din + bq + bv + bx) : nullptr, dout, dout + bq,
                         q_solver ? dout + bq + bv : nullptr, nullptr);
    if (rc != WBCQP_OK) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(q_next, dout, (size_t)batch * nq * es, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(v_next, dout + bq, (size_t)batch * nv * es, hipMemcpyDeviceToHost));
    if (q_solver) HIP_TRY(h, hipMemcpy(q_solver, dout + bq + bv, (size_t)batch * nv * es, hipMemcpyDeviceToHost));
    return WBCQP_OK;
}

// Validates a tree + task bindings against a structure and derives the rows kernel's tables and LDS layout.  Pure host code
// (wbcqp_check_model runs it without a device; wbcqp_set_model uploads what it returns).
// sel_host [D.n_sel]: the posture task's columns; force_gen_host [D.nc][6][12]: the contacts' force generators (host copies)
static int derive_terms(wbcqp_handle* h, const DevStruct& D, const int* sel_host, const double* force_gen_host, const wbcqp_model* md,
                        const wbcqp_taskmap* tm, TermsDev& T, std::vector<int>& ipool, std::vector<double>& dpool)
{
    if (!md || !tm) return fail(h, WBCQP_ERR_INVALID, "model / taskmap is NULL");
    const int nb = md->nbody, fb = md->floating_base ? 1 : 0;
    if (nb <= 0 || !md->parent || !md->jtype || !md->placement || !md->inertia) return fail(h, WBCQP_ERR_INVALID, "empty model");
    if (nb > kWave) return fail(h, WBCQP_ERR_UNSUPPORTED, "more than 64 bodies (one lane per body)");
    const int nv = nb + (fb ? 5 : 0), nq = nb + (fb ? 6 : 0), na = nv - (fb ? 6 : 0);
    if (nv != D.nv || na != D.na) return fail(h, WBCQP_ERR_INVALID, "model and structure disagree on nv / na");
    if (tm->n_contact != D.nc) return fail(h, WBCQP_ERR_INVALID, "taskmap and structure disagree on the number of contacts");
    if ((tm->bounds ? na : 0) != D.n_bound) return fail(h, WBCQP_ERR_INVALID, "taskmap and structure disagree on the bounds rows");
    if (tm->n_task < 0 || (tm->n_task > 0 && !tm->task) || tm->nref < 0 || !(tm->dt > 0.0)) return fail(h, WBCQP_ERR_INVALID, "bad taskmap");
    // tree: parents first, depth-first numbering (a subtree is a contiguous range)
    std::vector<int> depth(nb, 0), last(nb), idxq(nb), idxv(nb), bodyof(nv), kof(nv);
    for (int i = 0; i < nb; ++i) {
        last[i] = i;
        if (i == 0 ? md->parent[0] != -1 : (md->parent[i] < 0 || md->parent[i] >= i)) return fail(h, WBCQP_ERR_INVALID, "parent[i] must be in [0, i), -1 for body 0");
        const int jt = md->jtype[i];
        if (jt < WBCQP_J_FREEFLYER || jt > WBCQP_J_PZ || ((jt == WBCQP_J_FREEFLYER) != (fb && i == 0)))
            return fail(h, WBCQP_ERR_INVALID, "joint type out of range, or a free-flyer that is not body 0 of a floating-base model");
        if (i) depth[i] = depth[md->parent[i]] + 1;
        idxq[i] = fb ? (i == 0 ? 0 : 6 + i) : i;
        idxv[i] = fb ? (i == 0 ? 0 : 5 + i) : i;
    }
    for (int i = nb - 1; i > 0; --i) last[md->parent[i]] = std::max(last[md->parent[i]], last[i]);
    for (int i = 1; i < nb; ++i) {
        // depth-first: the parent of i is the body just before it or one of that body's ancestors
        bool on_path = false;
        for (int b = i - 1; b >= 0 && !on_path; b = md->parent[b]) on_path = (b == md->parent[i]);
        if (!on_path) return fail(h, WBCQP_ERR_INVALID, "bodies are not numbered depth-first");
    }
    int maxdepth = 0;
    for (int i = 0; i < nb; ++i) {
        maxdepth = std::max(maxdepth, depth[i]);
        const int cnt = (md->jtype[i] == WBCQP_J_FREEFLYER) ? 6 : 1;
        for (int k = 0; k < cnt; ++k) { bodyof[idxv[i] + k] = i; kof[idxv[i] + k] = k; }
    }
    if (md->nframe < 0 || (md->nframe > 0 && (!md->frame_body || !md->frame_placement))) return fail(h, WBCQP_ERR_INVALID, "bad frame tables");
    for (int f = 0; f < md->nframe; ++f)
        if (md->frame_body[f] < 0 || md->frame_body[f] >= nb) return fail(h, WBCQP_ERR_INVALID, "a frame hangs on a body that does not exist");
    auto frame_ok = [&](int f) { return f >= 0 && f < md->nframe; };
    // tasks -> law lanes (SE3 blocks then contacts), self-collision pairs, blocks
    std::vector<int> law_body, law_mask, law_row, law_ref, law_va, law_contact, pair_bt, pair_ba;
    std::vector<int> blk_kind, blk_mask, blk_row, blk_ref, blk_pair0, blk_npair;
    std::vector<double> law_place, law_kp, law_kd, scf_place, pair_par, blk_kp, blk_kd;
    std::vector<int> scf_frame, scf_body, pair_ft, pair_fa;
    auto scf_index = [&](int f) {
        for (size_t k = 0; k < scf_frame.size(); ++k)
            if (scf_frame[k] == f) return (int)k;
        scf_frame.push_back(f); scf_body.push_back(md->frame_body[f]);
        scf_place.insert(scf_place.end(), md->frame_placement + 12 * f, md->frame_placement + 12 * f + 12);
        return (int)scf_frame.size() - 1;
    };
    auto popc = [](int m, int bits) { int c = 0; for (int i = 0; i < bits; ++i) c += (m >> i) & 1; return c; };
    int row = 0;
    for (int t = 0; t < tm->n_task; ++t) {
        const wbcqp_task& K = tm->task[t];
        blk_kind.push_back(K.kind); blk_mask.push_back(K.mask); blk_row.push_back(row); blk_ref.push_back(K.ref);
        blk_kp.push_back(K.kp); blk_kd.push_back(K.kd);
        blk_pair0.push_back((int)pair_bt.size()); blk_npair.push_back(0);
        int need = 0;
        if (K.kind == WBCQP_T_SE3) {
            if (!frame_ok(K.frame)) return fail(h, WBCQP_ERR_INVALID, "SE3 task tracks a frame that does not exist");
            law_body.push_back(md->frame_body[K.frame]); law_mask.push_back(K.mask & 63); law_row.push_back(row);
            law_ref.push_back(K.ref); law_va.push_back(1); law_contact.push_back(-1);
            law_kp.push_back(K.kp); law_kd.push_back(K.kd);
            law_place.insert(law_place.end(), md->frame_placement + 12 * K.frame, md->frame_placement + 12 * K.frame + 12);
            row += popc(K.mask, 6); need = 24;
        }
        else if (K.kind == WBCQP_T_COM) { row += popc(K.mask, 3); need = 9; blk_mask.back() = K.mask & 7; }
        else if (K.kind == WBCQP_T_MOMENTUM) { row += popc(K.mask, 6); need = 12; blk_mask.back() = K.mask & 63; }
        else if (K.kind == WBCQP_T_SELFCOLLISION) {
            if (!frame_ok(K.frame) || K.n_avoided < 0 || (K.n_avoided > 0 && (!K.avoided_frame || !K.avoided_r0)))
                return fail(h, WBCQP_ERR_INVALID, "bad self-collision task");
            if (!(K.m > 0.0) || !(K.margin > 0.0)) return fail(h, WBCQP_ERR_INVALID, "self-collision needs m > 0 and margin > 0");
            // constants of the 5PL repulsor (task-self-collision.cpp:147-149)
            const double k5 = -std::log(std::pow(-1e-5 + 1., -1. / K.m) - 1.) / K.margin;
            const double s_p = -1. / k5 * std::log(-1 + std::pow(2, 1. / K.m));
            for (int a = 0; a < K.n_avoided; ++a) {
                const int fa = K.avoided_frame[a];
                if (!frame_ok(fa)) return fail(h, WBCQP_ERR_INVALID, "self-collision task avoids a frame that does not exist");
                pair_bt.push_back(md->frame_body[K.frame]); pair_ba.push_back(md->frame_body[fa]);
                pair_ft.push_back(scf_index(K.frame)); pair_fa.push_back(scf_index(fa));
                const double par[6] = {K.avoided_r0[a] + K.radius, k5, s_p, K.m, K.kp, K.kd};
                pair_par.insert(pair_par.end(), par, par + 6);
            }
            blk_npair.back() = K.n_avoided;
            row += 1;
        }
        else return fail(h, WBCQP_ERR_INVALID, "unknown task kind");
        if (need && (K.ref < 0 || K.ref + need > tm->nref)) return fail(h, WBCQP_ERR_INVALID, "a task reference lies outside the reference vector");
    }
    if (row != D.n_dense) return fail(h, WBCQP_ERR_INVALID, "the tasks' rows do not add up to the structure's n_dense");
    for (int c = 0; c < tm->n_contact; ++c) {
        const int f = tm->contact_frame[c];
        if (!frame_ok(f)) return fail(h, WBCQP_ERR_INVALID, "contact frame does not exist");
        if (tm->contact_ref[c] < 0 || tm->contact_ref[c] + 24 > tm->nref) return fail(h, WBCQP_ERR_INVALID, "a contact reference lies outside the reference vector");
        law_body.push_back(md->frame_body[f]); law_mask.push_back(63); law_row.push_back(0); law_ref.push_back(tm->contact_ref[c]);
        law_va.push_back(1); law_contact.push_back(c); law_kp.push_back(tm->contact_kp[c]); law_kd.push_back(tm->contact_kd[c]);
        law_place.insert(law_place.end(), md->frame_placement + 12 * f, md->frame_placement + 12 * f + 12);
    }
    if ((int)law_body.size() > kWave || (int)blk_kind.size() > kWave || (int)scf_frame.size() > kWave)
        return fail(h, WBCQP_ERR_UNSUPPORTED, "more than 64 framed tasks or self-collision frames (one lane each)");
    if (D.n_sel > 0 && (tm->posture_ref < 0 || tm->posture_ref + na > tm->nref)) return fail(h, WBCQP_ERR_INVALID, "the posture reference lies outside the reference vector");
    if (D.n_bound > 0 && (!md->q_lb || !md->q_ub || !md->dq_max)) return fail(h, WBCQP_ERR_INVALID, "bounds need q_lb / q_ub / dq_max");

    T = TermsDev{};
    T.nb = nb; T.nq = nq; T.nv = nv; T.na = na; T.floating_base = fb;
    int nrounds = 0;
    while ((1 << nrounds) < maxdepth + 1) ++nrounds;
    T.nrounds = nrounds; // <= 6 for 64 bodies
    std::vector<int> anc((size_t)std::max(nrounds, 1) * nb, -1);
    for (int i = 0; i < nb; ++i) anc[i] = md->parent[i];
    for (int r = 1; r < nrounds; ++r)
        for (int i = 0; i < nb; ++i) {
            const int a = anc[(size_t)(r - 1) * nb + i];
            anc[(size_t)r * nb + i] = (a >= 0) ? anc[(size_t)(r - 1) * nb + a] : -1;
        }
    T.nlaw = (int)law_body.size(); T.npair = (int)pair_bt.size(); T.nscf = (int)scf_frame.size(); T.nblock = (int)blk_kind.size(); T.nc = D.nc;
    T.n_dense = D.n_dense; T.n_sel = D.n_sel; T.n_bound = D.n_bound; T.r1 = D.r1; T.nref = tm->nref;
    T.posture_ref = tm->posture_ref; T.posture_kp = tm->posture_kp; T.posture_kd = tm->posture_kd; T.dt = tm->dt;
    for (int k = 0; k < 3; ++k) T.g[k] = md->gravity[k];
    ipool.clear();
    dpool.clear();
    auto puti = [&](const int* a, size_t n) { int at = (int)ipool.size(); ipool.insert(ipool.end(), a, a + n); ipool.push_back(0); return at; };
    auto putd = [&](const double* a, size_t n) { int at = (int)dpool.size(); dpool.insert(dpool.end(), a, a + n); dpool.push_back(0.0); return at; };
    // the posture task's columns: the actuated joints its mask keeps (tasks.cpp:197-217), from the structure
    std::vector<int> sel(D.n_sel);
    for (int r = 0; r < D.n_sel; ++r) {
        sel[r] = sel_host[r];
        if (sel[r] < nv - na || sel[r] >= nv) return fail(h, WBCQP_ERR_INVALID, "a posture row selects a column that is not an actuated joint");
    }
    // cop task (tasks.cpp:156-178): the rows kernel forms its three rows from the contact frames; the contact points are the skew
    // blocks of the force generators, T(3.., 3 p ..) = skew(p): x = T(5, 3p + 1), y = T(3, 3p + 2), z = T(4, 3p)
    std::vector<double> cop_pts;
    T.cop = D.cop_task >= 0 ? 1 : 0;
    if (T.cop)
        for (int c = 0; c < D.nc; ++c)
            for (int p = 0; p < 4; ++p) {
                const double* Tg = force_gen_host + (size_t)c * 72;
                cop_pts.push_back(Tg[5 * 12 + 3 * p + 1]);
                cop_pts.push_back(Tg[3 * 12 + 3 * p + 2]);
                cop_pts.push_back(Tg[4 * 12 + 3 * p]);
            }
    T.i_jtype = puti(md->jtype, nb); T.i_last = puti(last.data(), nb);
    T.i_anc = puti(anc.data(), (size_t)nrounds * nb);
    T.i_idxq = puti(idxq.data(), nb); T.i_idxv = puti(idxv.data(), nb); T.i_bodyof = puti(bodyof.data(), nv); T.i_kof = puti(kof.data(), nv);
    T.i_law_body = puti(law_body.data(), law_body.size()); T.i_law_mask = puti(law_mask.data(), law_mask.size());
    T.i_law_row = puti(law_row.data(), law_row.size()); T.i_law_ref = puti(law_ref.data(), law_ref.size());
    T.i_law_va = puti(law_va.data(), law_va.size()); T.i_law_contact = puti(law_contact.data(), law_contact.size());
    T.i_pair_bt = puti(pair_bt.data(), pair_bt.size());
    T.i_pair_ba = puti(pair_ba.data(), pair_ba.size());
    T.i_pair_ft = puti(pair_ft.data(), pair_ft.size()); T.i_pair_fa = puti(pair_fa.data(), pair_fa.size());
    T.i_scf_body = puti(scf_body.data(), scf_body.size());
    T.i_blk_kind = puti(blk_kind.data(), blk_kind.size()); T.i_blk_mask = puti(blk_mask.data(), blk_mask.size());
    T.i_blk_row = puti(blk_row.data(), blk_row.size()); T.i_blk_ref = puti(blk_ref.data(), blk_ref.size());
    T.i_blk_pair0 = puti(blk_pair0.data(), blk_pair0.size());
    T.i_blk_npair = puti(blk_npair.data(), blk_npair.size()); T.i_sel_col = puti(sel.data(), sel.size());
    T.d_place = putd(md->placement, (size_t)nb * 12); T.d_inertia = putd(md->inertia, (size_t)nb * 10);
    T.d_law_place = putd(law_place.data(), law_place.size()); T.d_law_kp = putd(law_kp.data(), law_kp.size());
    T.d_law_kd = putd(law_kd.data(), law_kd.size());
    T.d_scf_place = putd(scf_place.data(), scf_place.size());
    T.d_pair_par = putd(pair_par.data(), pair_par.size());
    T.d_blk_kp = putd(blk_kp.data(), blk_kp.size()); T.d_blk_kd = putd(blk_kd.data(), blk_kd.size());
    T.d_cop_pts = putd(cop_pts.data(), cop_pts.size());
    T.d_qlb = putd(md->q_lb, D.n_bound ? na : 0); T.d_qub = putd(md->q_ub, D.n_bound ? na : 0); T.d_dqmax = putd(md->dq_max, D.n_bound ? na : 0);
    int o = 0;
    auto take = [&](int count) { int at = o; o += (count + 1) & ~1; return at; };
    T.o_state = take(nq + nv + tm->nref);
    T.o_kin = take(nb * kKinStride);
    T.o_scan = take((nb + 1) * kScanStride);
    T.o_tot = take(8); // momentum totals (6), the frames-published flag
    T.o_sf = take(nv * kSFStride);
    T.o_law = take(T.nlaw * kLawStride);
    T.o_pair = take(T.npair * kPairStride);
    T.o_scf = take(T.nscf * kScfStride);
    T.o_b1 = take(D.r1);
    T.o_bc = take(6 * D.nc);
    T.lds_doubles = o;
    if ((size_t)o * 8 > 160 * 1024) return fail(h, WBCQP_ERR_UNSUPPORTED, "the working set of one instance exceeds the LDS");
    return WBCQP_OK;
}

int wbcqp_check_model(const wbcqp_structure* st, const wbcqp_model* md, const wbcqp_taskmap* tm, int32_t* lds_bytes)
{
    DevStruct D;
    HostBlocks HB;
    wbcqp_layout L;
    std::string why;
    int rc = derive(st, D, HB, L, why);
    if (rc != WBCQP_OK) return fail(nullptr, rc, why);
    TermsDev T{};
    std::vector<int> ipool;
    std::vector<double> dpool;
    if ((D.n_sel > 0 && !st->sel_col) || (D.nc > 0 && !st->force_gen)) return fail(nullptr, WBCQP_ERR_INVALID, "sel_col / force_gen is NULL");
    rc = derive_terms(nullptr, D, st->sel_col, st->force_gen, md, tm, T, ipool, dpool);
    if (rc == WBCQP_OK && lds_bytes) *lds_bytes = T.lds_doubles * 8;
    return rc;
}

int wbcqp_set_model(wbcqp_handle* h, int slot, const wbcqp_model* md, const wbcqp_taskmap* tm)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (slot < 0 || slot >= WBCQP_MAX_STRUCTURES || !h->slots[slot].set) return fail(h, WBCQP_ERR_INVALID, "slot has no structure");
    Slot& s = h->slots[slot];
    TermsDev T{};
    std::vector<int> ipool;
    std::vector<double> dpool;
    int rc = derive_terms(h, s.host, s.sel_col_h.data(), s.force_gen_h.data(), md, tm, T, ipool, dpool);
    if (rc != WBCQP_OK) return rc;
    const int o = T.lds_doubles;
    HIP_TRY(h, hipSetDevice(h->device));
    release_model(s);
    void *di = nullptr, *dd = nullptr;
    HIP_TRY(h, hipMalloc(&di, ipool.size() * sizeof(int)));
    s.model_allocs.push_back(di);
    HIP_TRY(h, hipMalloc(&dd, dpool.size() * sizeof(double)));
    s.model_allocs.push_back(dd);
    HIP_TRY(h, hipMemcpy(di, ipool.data(), ipool.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(dd, dpool.data(), dpool.size() * sizeof(double), hipMemcpyHostToDevice));
    T.ipool = static_cast<const int*>(di);
    T.dpool = static_cast<const double*>(dd);
    if (o * 8 > 64 * 1024) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&terms_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, o * 8));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&terms_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, o * 8));
    }
    s.terms = T;
    s.has_model = true;
    return WBCQP_OK;
}

int wbcqp_problem_data(wbcqp_handle* h, int slot, int batch, const wbcqp_state* st, const wbcqp_inputs* rows, void* stream)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (slot < 0 || slot >= WBCQP_MAX_STRUCTURES || !h->slots[slot].set || !h->slots[slot].has_model)
        return fail(h, WBCQP_ERR_INVALID, "slot has no model (wbcqp_set_model)");
    if (batch < 0) return fail(h, WBCQP_ERR_INVALID, "negative batch");
    if (batch == 0) return WBCQP_OK;
    const Slot& s = h->slots[slot];
    const wbcqp_layout& L = s.layout;
    if (!st || !rows || !st->q || !st->v || (s.terms.nref > 0 && !st->ref)) return fail(h, WBCQP_ERR_INVALID, "state arrays q / v / ref are required");
    if (!rows->M || !rows->h || (L.len_A && !rows->A) || (L.len_b1 && !rows->b1) || (L.len_Ac && !rows->Ac) || (L.len_bc && !rows->bc) ||
        (L.len_blb && (!rows->blb || !rows->bub)) || (L.len_Acop && !rows->Acop))
        return fail(h, WBCQP_ERR_INVALID, "row arrays M, h, A, b1, Ac, bc, blb, bub (Acop with a cop task) are required");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t sm = static_cast<hipStream_t>(stream);
    const int lds = s.terms.lds_doubles * 8;
    if (h->dtype == WBCQP_F64) {
        TermsArgs<double> a{};
        a.T = s.terms; a.batch = batch; a.dbg = h->dbg;
        a.q = static_cast<const double*>(st->q); a.v = static_cast<const double*>(st->v); a.ref = static_cast<const double*>(st->ref);
        a.M = (double*)rows->M; a.h = (double*)rows->h; a.A = (double*)rows->A; a.b1 = (double*)rows->b1; a.Ac = (double*)rows->Ac;
        a.bc = (double*)rows->bc; a.blb = (double*)rows->blb; a.bub = (double*)rows->bub; a.Acop = (double*)rows->Acop;
        a.momentum = static_cast<double*>(st->momentum);
        hipLaunchKernelGGL(terms_kernel<double>, dim3(batch), dim3(kTermsThreads), lds, sm, a);
    }
    else {
        TermsArgs<float> a{};
        a.T = s.terms; a.batch = batch; a.dbg = h->dbg;
        a.q = static_cast<const float*>(st->q); a.v = static_cast<const float*>(st->v); a.ref = static_cast<const float*>(st->ref);
        a.M = (float*)rows->M; a.h = (float*)rows->h; a.A = (float*)rows->A; a.b1 = (float*)rows->b1; a.Ac = (float*)rows->Ac;
        a.bc = (float*)rows->bc; a.blb = (float*)rows->blb; a.bub = (float*)rows->bub; a.Acop = (float*)rows->Acop;
        a.momentum = static_cast<float*>(st->momentum);
        hipLaunchKernelGGL(terms_kernel<float>, dim3(batch), dim3(kTermsThreads), lds, sm, a);
    }
    HIP_TRY(h, hipGetLastError());
    return WBCQP_OK;
}

int wbcqp_problem_data_host(wbcqp_handle* h, int slot, int batch, const wbcqp_state* st, const wbcqp_inputs* rows)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (slot < 0 || slot >= WBCQP_MAX_STRUCTURES || !h->slots[slot].set || !h->slots[slot].has_model)
        return fail(h, WBCQP_ERR_INVALID, "slot has no model (wbcqp_set_model)");
    if (batch < 0) return fail(h, WBCQP_ERR_INVALID, "negative batch");
    if (batch == 0) return WBCQP_OK;
    if (!st || !rows) return fail(h, WBCQP_ERR_INVALID, "state / rows is NULL");
    const Slot& s = h->slots[slot];
    const wbcqp_layout& L = s.layout;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t es = (h->dtype == WBCQP_F64) ? 8 : 4;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int ilen[3] = {s.terms.nq, s.terms.nv, s.terms.nref};
    const void* isrc[3] = {st->q, st->v, st->ref};
    size_t ioff[3], in_bytes = 0;
    for (int f = 0; f < 3; ++f) { ioff[f] = in_bytes; in_bytes += al((size_t)ilen[f] * batch * es); }
    constexpr int NR = 9; // row arrays of the record: the eight every stack has, and the cop rows
    const int olen[NR] = {L.len_M, L.len_h, L.len_A, L.len_b1, L.len_Ac, L.len_bc, L.len_blb, L.len_bub, L.len_Acop};
    void* odst[NR] = {(void*)rows->M, (void*)rows->h, (void*)rows->A, (void*)rows->b1, (void*)rows->Ac, (void*)rows->bc, (void*)rows->blb, (void*)rows->bub,
                      (void*)rows->Acop};
    size_t ooff[NR], out_bytes = 0;
    for (int f = 0; f < NR; ++f) { ooff[f] = out_bytes; out_bytes += al((size_t)olen[f] * batch * es); }
    const size_t omom = out_bytes;
    out_bytes += al((size_t)6 * batch * es);
    int rc = ensure(h, h->stage_in, in_bytes + 256);
    if (rc != WBCQP_OK) return rc;
    rc = ensure(h, h->stage_out, out_bytes + 256);
    if (rc != WBCQP_OK) return rc;
    char* din = static_cast<char*>(h->stage_in.dev);
    char* dout = static_cast<char*>(h->stage_out.dev);
    for (int f = 0; f < 3; ++f) {
        if (ilen[f] > 0 && !isrc[f]) return fail(h, WBCQP_ERR_INVALID, "state arrays q / v / ref are required");
        if (ilen[f] > 0) HIP_TRY(h, hipMemcpy(din + ioff[f], isrc[f], (size_t)ilen[f] * batch * es, hipMemcpyHostToDevice));
    }
    wbcqp_state ds = {din + ioff[0], din + ioff[1], din + ioff[2], st->momentum ? dout + omom : nullptr};
    wbcqp_inputs dr{};
    dr.M = dout + ooff[0]; dr.h = dout + ooff[1]; dr.A = dout + ooff[2]; dr.b1 = dout + ooff[3]; dr.Ac = dout + ooff[4];
    dr.bc = dout + ooff[5]; dr.blb = dout + ooff[6]; dr.bub = dout + ooff[7]; dr.Acop = dout + ooff[8];
    rc = wbcqp_problem_data(h, slot, batch, &ds, &dr, nullptr);
    if (rc != WBCQP_OK) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    for (int f = 0; f < NR; ++f) {
        if (olen[f] > 0 && !odst[f]) return fail(h, WBCQP_ERR_INVALID, "row arrays M, h, A, b1, Ac, bc, blb, bub (Acop with a cop task) are required");
        if (olen[f] > 0) HIP_TRY(h, hipMemcpy(odst[f], dout + ooff[f], (size_t)olen[f] * batch * es, hipMemcpyDeviceToHost));
    }
    if (st->momentum) HIP_TRY(h, hipMemcpy(st->momentum, dout + omom, (size_t)6 * batch * es, hipMemcpyDeviceToHost));
    return WBCQP_OK;
}

static int tick_impl(wbcqp_handle* h, int slot, int batch, const wbcqp_tick_io* io, void* stream, const RollAcc& acc);

int wbcqp_tick(wbcqp_handle* h, int slot, int batch, const wbcqp_tick_io* io, void* stream) { return tick_impl(h, slot, batch, io, stream, RollAcc{}); }

static int tick_impl(wbcqp_handle* h, int slot, int batch, const wbcqp_tick_io* io, void* stream, const RollAcc& acc)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (!io) return fail(h, WBCQP_ERR_INVALID, "io is NULL");
    if (slot < 0 || slot >= WBCQP_MAX_STRUCTURES || !h->slots[slot].set || !h->slots[slot].has_model)
        return fail(h, WBCQP_ERR_INVALID, "slot has no model (wbcqp_set_model)");
    if (batch < 0) return fail(h, WBCQP_ERR_INVALID, "negative batch");
    if (batch == 0) return WBCQP_OK;
    if (!io->q_next || !io->v_next) return fail(h, WBCQP_ERR_INVALID, "q_next / v_next is NULL");
    const Slot& s = h->slots[slot];
    int rc = wbcqp_problem_data(h, slot, batch, &io->state, &io->rows, stream);
    if (rc != WBCQP_OK) return rc;
    rc = wbcqp_solve_batch(h, slot, batch, &io->rows, &io->out, stream);
    if (rc != WBCQP_OK) return rc;
    return integrate_impl(h, batch, s.terms.nv, s.terms.floating_base, io->dt, io->state.q, io->state.v, io->out.x, s.host.n,
                          io->out.status, io->q_next, io->v_next, io->q_solver, stream, acc);
}

int wbcqp_rollout(wbcqp_handle* h, int slot, int batch, int n_ticks, const wbcqp_rollout_io* io, void* stream)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (!io) return fail(h, WBCQP_ERR_INVALID, "io is NULL");
    if (slot < 0 || slot >= WBCQP_MAX_STRUCTURES || !h->slots[slot].set || !h->slots[slot].has_model)
        return fail(h, WBCQP_ERR_INVALID, "slot has no model (wbcqp_set_model)");
    if (batch < 0 || n_ticks < 0) return fail(h, WBCQP_ERR_INVALID, "negative batch / n_ticks");
    if (batch == 0 || n_ticks == 0) return WBCQP_OK;
    const Slot& s = h->slots[slot];
    const wbcqp_layout& L = s.layout;
    const TermsDev& T = s.terms;
    if (!io->state.q || !io->state.v || (T.nref > 0 && !io->state.ref)) return fail(h, WBCQP_ERR_INVALID, "state arrays q / v / ref are required");
    if ((L.len_tlb && (!io->tlb || !io->tub)) || !io->w) return fail(h, WBCQP_ERR_INVALID, "tlb / tub / w are required");
    if (!io->out.x || !io->out.status || !io->out.iters || (s.host.na > 0 && !io->out.tau) || !io->q_next || !io->v_next)
        return fail(h, WBCQP_ERR_INVALID, "x, tau, status, iters, q_next, v_next are required");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t sm = static_cast<hipStream_t>(stream);
    const size_t es = (h->dtype == WBCQP_F64) ? 8 : 4;
    const size_t B = (size_t)batch;
    // One sub-batch (what K calls of wbcqp_tick do) or two.  Two pay where a tick's solve has a long tail -- one instance far
    // above the rest, B = 1024: 1.04x -- and cost where it has none (every instance heavy: 0.76x) or where the launch is large
    // enough to hide its tail by itself (B = 4096: 0.93x); three gain less (1.02x), four queue behind one another (0.57x)
    // [tools/rollout_bench.py].  Which regime a caller is in is not knowable from the arguments, so it is measured: the first
    // roll-outs of a (slot, batch) run one sub-batch, then two, each timed on the device by an event pair a LATER call reads
    // without blocking -- the first sample of either form is discarded (it pays that form's allocations, stream creation and a
    // device synchronisation) --; from then on the faster of the two, the other one tried again every 64th call (a workload drifts).  WBCQP_ROLLOUT_STREAMS overrides (1 .. 8).  The result does not depend on the choice, bit for bit.
    for (auto& mz : h->roll_meas) {
        if (!mz.pending || hipEventQuery(mz.t1) != hipSuccess) continue;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, mz.t0, mz.t1) == hipSuccess && mz.ticks > 0 && mz.S >= 1 && mz.S <= 2 && mz.stat >= 0 &&
            mz.stat < (int)h->roll_stats.size()) {
            auto& stt = h->roll_stats[mz.stat];
            double& slot_us = stt.us[mz.S];
            const double us = (double)ms * 1e3 / mz.ticks;
            if (stt.cold[mz.S]) stt.cold[mz.S] = 0; // (the form runs once more before it is compared)
            else slot_us = slot_us > 0.0 ? 0.5 * (slot_us + us) : us;
        }
        mz.pending = false;
    }
    (void)hipGetLastError(); // (hipEventQuery's hipErrorNotReady is not an error of this call)
    int stat_i = -1;
    for (size_t i = 0; i < h->roll_stats.size(); ++i)
        if (h->roll_stats[i].slot == slot && h->roll_stats[i].batch == batch) stat_i = (int)i;
    if (stat_i < 0) {
        if (h->roll_stats.size() >= 64) { // (an index into the table is kept by the pending measurements: start over)
            h->roll_stats.clear();
            for (auto& mz : h->roll_meas) mz.stat = -1;
        }
        h->roll_stats.emplace_back();
        stat_i = (int)h->roll_stats.size() - 1;
        h->roll_stats[stat_i].slot = slot;
        h->roll_stats[stat_i].batch = batch;
    }
    int S = 1;
    {
        auto& stt = h->roll_stats[stat_i];
        if (batch >= 512) {
            bool two_in_flight = false; // a roll-out with two sub-batches is on the device and not measured yet
            for (const auto& mz : h->roll_meas) two_in_flight = two_in_flight || (mz.pending && mz.stat == stat_i && mz.S == 2);
            if (stt.us[1] <= 0.0) S = 1;
            else if (stt.us[2] <= 0.0) S = two_in_flight ? 1 : 2;
            else {
                S = stt.us[2] < stt.us[1] ? 2 : 1;
                const int other = 3 - S;
                if (stt.calls % 64 == 63) S = other; // the other form again, whatever its last figure: a workload drifts, and one inflated sample must not pin the choice
            }
        }
        ++stt.calls;
    }
    if (const char* ev = std::getenv("WBCQP_ROLLOUT_STREAMS")) S = std::max(1, std::min({std::atoi(ev), 8, batch}));
    if (std::getenv("WBCQP_ROLLOUT_DEBUG"))
        std::fprintf(stderr, "wbcqp_rollout: slot %d batch %d ticks %d -> %d sub-batch(es); measured us per tick: one %.1f, two %.1f\n", slot, batch,
                     n_ticks, S, h->roll_stats[stat_i].us[1], h->roll_stats[stat_i].us[2]);
    for (int k = 0; k < S; ++k) { // the handle owns a sub-batch's stream, event and counters from the moment they exist (a failure half
        // way leaves them to wbcqp_destroy)
        if ((int)h->roll_subs.size() <= k) h->roll_subs.emplace_back();
        wbcqp_handle::RollSub& r = h->roll_subs[k];
        if (!r.stream) HIP_TRY(h, hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
        if (!r.done) HIP_TRY(h, hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
        if (!r.ord.queue) {
            HIP_TRY(h, hipMalloc(&r.ord.queue, 2 * sizeof(int)));
            HIP_TRY(h, hipMemset(r.ord.queue, 0, 2 * sizeof(int)));
        }
    }
    if (!h->roll_start) HIP_TRY(h, hipEventCreateWithFlags(&h->roll_start, hipEventDisableTiming));
    const bool had_roll = h->roll_done != nullptr;
    if (!h->roll_done) HIP_TRY(h, hipEventCreateWithFlags(&h->roll_done, hipEventDisableTiming));
    // the record of every instance (the rows kernel's output, the solve's input) and the state ping-pong
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int rlen[9] = {L.len_M, L.len_h, L.len_A, L.len_b1, L.len_Ac, L.len_bc, L.len_blb, L.len_bub, L.len_Acop};
    size_t roff[9], rec_bytes = 0;
    for (int f = 0; f < 9; ++f) { roff[f] = rec_bytes; rec_bytes += al((size_t)rlen[f] * B * es); }
    const size_t qb = al((size_t)T.nq * B * es), vb = al((size_t)T.nv * B * es);
    const int sub_cap = (batch + S - 1) / S;
    bool grow = h->roll_rec.bytes < rec_bytes || h->roll_state.bytes < 2 * (qb + vb);
    for (int k = 0; k < S; ++k) grow = grow || h->roll_subs[k].ord.cap < sub_cap;
    if (grow) { // first call of a larger shape: nothing of an earlier call may still be running on what is replaced
        HIP_TRY(h, hipDeviceSynchronize());
        int rc = ensure(h, h->roll_rec, rec_bytes);
        if (rc != WBCQP_OK) return rc;
        rc = ensure(h, h->roll_state, 2 * (qb + vb));
        if (rc != WBCQP_OK) return rc;
        for (int k = 0; k < S; ++k) {
            OrderState& os = h->roll_subs[k].ord;
            if (os.cap < sub_cap) {
                if (os.order) (void)hipFree(os.order);
                os.order = nullptr;
                os.cap = os.total = 0;
                HIP_TRY(h, hipMalloc(&os.order, 2 * sizeof(int) * (size_t)sub_cap));
                os.cap = sub_cap;
            }
        }
    }
    char* rec = static_cast<char*>(h->roll_rec.dev);
    char* stt = static_cast<char*>(h->roll_state.dev);
    char* qbuf[2] = {stt, stt + qb};
    char* vbuf[2] = {stt + 2 * qb, stt + 2 * qb + vb};
    // the sub-streams of the previous roll-out may still be on the ping-pong buffers and the record when this one comes in on
    // another stream: wait for that roll-out's end first (on the same stream the wait is already implied)
    if (had_roll) HIP_TRY(h, hipStreamWaitEvent(sm, h->roll_done, 0));
    // (the first tick reads the caller's q / v in place: no copy into the ping-pong buffers)
    wbcqp_handle::RollMeas* meas = nullptr; // a free event pair: this roll-out is measured
    for (auto& mz : h->roll_meas)
        if (!mz.pending && !meas) meas = &mz;
    if (meas) {
        if (!meas->t0) HIP_TRY(h, hipEventCreate(&meas->t0));
        if (!meas->t1) HIP_TRY(h, hipEventCreate(&meas->t1));
        HIP_TRY(h, hipEventRecord(meas->t0, sm));
    }
    HIP_TRY(h, hipEventRecord(h->roll_start, sm));
    auto at = [es](const void* p, size_t elems) -> const void* { return p ? static_cast<const char*>(p) + elems * es : nullptr; };
    auto atw = [es](void* p, size_t elems) -> void* { return p ? static_cast<char*>(p) + elems * es : nullptr; };
    int rc_all = WBCQP_OK;
    // one sub-batch: the ticks go out on the caller's own stream with that stream's launch-order state -- K calls of wbcqp_tick, minus
    // the caller's loop (on a stream of its own the same sequence measured 1.5 % slower than the tick loop: fork, join, a second queue)
    const bool own_stream = S == 1;
    if (!own_stream)
        for (int k = 0; k < S; ++k) HIP_TRY(h, hipStreamWaitEvent(h->roll_subs[k].stream, h->roll_start, 0));
    // tick t of every sub-batch is enqueued before tick t + 1 of any: the streams then advance together on the device (enqueued one
    // sub-batch after the other, the last stream's first tick would reach the device when the first stream is almost through), and the
    // tail of one sub-batch's solve (its longest QP) runs beside the bulk of another's
    for (int t = 0; t < n_ticks && rc_all == WBCQP_OK; ++t) {
        const bool last = t + 1 == n_ticks;
        for (int k = 0; k < S && rc_all == WBCQP_OK; ++k) {
            const size_t b0 = (size_t)k * batch / S, b1 = (size_t)(k + 1) * batch / S;
            const int nb = (int)(b1 - b0);
            if (nb == 0) continue;
            wbcqp_handle::RollSub& sub = h->roll_subs[k];
            wbcqp_tick_io d{};
            d.rows.M = rec + roff[0] + b0 * rlen[0] * es; d.rows.h = rec + roff[1] + b0 * rlen[1] * es; d.rows.A = rec + roff[2] + b0 * rlen[2] * es;
            d.rows.b1 = rec + roff[3] + b0 * rlen[3] * es; d.rows.Ac = rec + roff[4] + b0 * rlen[4] * es; d.rows.bc = rec + roff[5] + b0 * rlen[5] * es;
            d.rows.blb = rec + roff[6] + b0 * rlen[6] * es; d.rows.bub = rec + roff[7] + b0 * rlen[7] * es;
            d.rows.Acop = rec + roff[8] + b0 * rlen[8] * es;
            d.rows.tlb = at(io->tlb, b0 * L.len_tlb); d.rows.tub = at(io->tub, b0 * L.len_tub); d.rows.w = at(io->w, b0 * L.len_w);
            d.out.x = atw(io->out.x, b0 * L.n); d.out.tau = atw(io->out.tau, b0 * s.host.na); d.out.objective = atw(io->out.objective, b0);
            d.out.status = io->out.status + b0; d.out.iters = io->out.iters + b0;
            d.out.n_active = io->out.n_active ? io->out.n_active + b0 : nullptr;
            d.out.active_mask = io->out.active_mask ? io->out.active_mask + b0 * 8 : nullptr;
            d.state.q = (t == 0) ? at(io->state.q, b0 * T.nq) : (const void*)(qbuf[t & 1] + b0 * T.nq * es);
            d.state.v = (t == 0) ? at(io->state.v, b0 * T.nv) : (const void*)(vbuf[t & 1] + b0 * T.nv * es);
            d.state.ref = at(io->state.ref, ((size_t)t * B + b0) * T.nref);
            d.state.momentum = last ? atw(io->state.momentum, b0 * 6) : nullptr;
            d.q_next = last ? atw(io->q_next, b0 * T.nq) : (void*)(qbuf[(t + 1) & 1] + b0 * T.nq * es);
            d.v_next = last ? atw(io->v_next, b0 * T.nv) : (void*)(vbuf[(t + 1) & 1] + b0 * T.nv * es);
            d.q_solver = last ? atw(io->q_solver, b0 * T.nv) : nullptr;
            d.dt = io->dt;
            h->graph_ord = own_stream ? nullptr : &sub.ord; // a sub-batch's own launch-order state and queue counter (as a captured tick has)
            const RollAcc acc = {io->out.iters + b0, io->iters_sum ? io->iters_sum + b0 : nullptr, io->ticks_ok ? io->ticks_ok + b0 : nullptr, t == 0 ? 1 : 0};
            rc_all = tick_impl(h, slot, nb, &d, own_stream ? sm : sub.stream, acc); // (the per-instance totals ride along with the integration)
            h->graph_ord = nullptr;
        }
    }
    for (int k = 0; k < S && !own_stream; ++k) {
        HIP_TRY(h, hipEventRecord(h->roll_subs[k].done, h->roll_subs[k].stream));
        HIP_TRY(h, hipStreamWaitEvent(sm, h->roll_subs[k].done, 0));
    }
    HIP_TRY(h, hipEventRecord(h->roll_done, sm));
    if (meas && rc_all == WBCQP_OK) {
        HIP_TRY(h, hipEventRecord(meas->t1, sm));
        meas->stat = stat_i; meas->S = S; meas->ticks = n_ticks; meas->pending = true;
    }
    return rc_all;
}

int wbcqp_tick_host(wbcqp_handle* h, int slot, int batch, const wbcqp_tick_io* io)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (!io) return fail(h, WBCQP_ERR_INVALID, "io is NULL");
    if (slot < 0 || slot >= WBCQP_MAX_STRUCTURES || !h->slots[slot].set || !h->slots[slot].has_model)
        return fail(h, WBCQP_ERR_INVALID, "slot has no model (wbcqp_set_model)");
    if (batch < 0) return fail(h, WBCQP_ERR_INVALID, "negative batch");
    if (batch == 0) return WBCQP_OK;
    const Slot& s = h->slots[slot];
    const wbcqp_layout& L = s.layout;
    const TermsDev& T = s.terms;
    if (!io->state.q || !io->state.v || (T.nref > 0 && !io->state.ref)) return fail(h, WBCQP_ERR_INVALID, "state arrays q / v / ref are required");
    if ((L.len_tlb && (!io->rows.tlb || !io->rows.tub)) || !io->rows.w) return fail(h, WBCQP_ERR_INVALID, "tlb / tub / w are required");
    if (!io->out.x || !io->out.status || !io->out.iters || (s.host.na > 0 && !io->out.tau) || !io->q_next || !io->v_next)
        return fail(h, WBCQP_ERR_INVALID, "x, tau, status, iters, q_next, v_next are required");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t es = (h->dtype == WBCQP_F64) ? 8 : 4;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t B = (size_t)batch;
    // device staging: inputs | the record | outputs
    const int ilen[6] = {T.nq, T.nv, T.nref, L.len_tlb, L.len_tub, L.len_w};
    const void* isrc[6] = {io->state.q, io->state.v, io->state.ref, io->rows.tlb, io->rows.tub, io->rows.w};
    size_t ioff[6], in_bytes = 0;
    for (int f = 0; f < 6; ++f) { ioff[f] = in_bytes; in_bytes += al((size_t)ilen[f] * B * es); }
    constexpr int NR = 9;
    const int rlen[NR] = {L.len_M, L.len_h, L.len_A, L.len_b1, L.len_Ac, L.len_bc, L.len_blb, L.len_bub, L.len_Acop};
    void* rdst[NR] = {(void*)io->rows.M, (void*)io->rows.h, (void*)io->rows.A, (void*)io->rows.b1, (void*)io->rows.Ac, (void*)io->rows.bc,
                      (void*)io->rows.blb, (void*)io->rows.bub, (void*)io->rows.Acop};
    size_t roff[NR];
    for (int f = 0; f < NR; ++f) { roff[f] = in_bytes; in_bytes += al((size_t)rlen[f] * B * es); }
    const size_t o_x = 0, o_tau = o_x + al((size_t)L.n * B * es), o_obj = o_tau + al((size_t)s.host.na * B * es), o_st = o_obj + al(B * es),
                 o_it = o_st + al(B * 4), o_na = o_it + al(B * 4), o_qn = o_na + al(B * 4), o_vn = o_qn + al((size_t)T.nq * B * es),
                 o_qs = o_vn + al((size_t)T.nv * B * es), o_mom = o_qs + al((size_t)T.nv * B * es), o_am = o_mom + al((size_t)6 * B * es),
                 out_bytes = o_am + (io->out.active_mask ? al(B * 32) : 0);
    int rc = ensure(h, h->stage_in, in_bytes + 256);
    if (rc != WBCQP_OK) return rc;
    rc = ensure(h, h->stage_out, out_bytes + 256);
    if (rc != WBCQP_OK) return rc;
    char* din = static_cast<char*>(h->stage_in.dev);
    char* dout = static_cast<char*>(h->stage_out.dev);
    const size_t in_only = roff[0]; // the inputs lie in front of the record: [0, roff[0])
    const bool packed = in_only <= kPackedBytes && out_bytes <= kPackedBytes;
    if (packed) {
        rc = ensure_pinned(h, h->pin_in, kPackedBytes);
        if (rc != WBCQP_OK) return rc;
        rc = ensure_pinned(h, h->pin_out, kPackedBytes);
        if (rc != WBCQP_OK) return rc;
        char* pin = static_cast<char*>(h->pin_in.host);
        for (int f = 0; f < 6; ++f)
            if (ilen[f] > 0) std::memcpy(pin + ioff[f], isrc[f], (size_t)ilen[f] * B * es);
        HIP_TRY(h, hipMemcpyAsync(din, pin, in_only, hipMemcpyHostToDevice, nullptr));
    }
    else
        for (int f = 0; f < 6; ++f)
            if (ilen[f] > 0) HIP_TRY(h, hipMemcpyAsync(din + ioff[f], isrc[f], (size_t)ilen[f] * B * es, hipMemcpyHostToDevice, nullptr));
    wbcqp_tick_io d{};
    d.state = {din + ioff[0], din + ioff[1], din + ioff[2], io->state.momentum ? dout + o_mom : nullptr};
    d.rows.M = din + roff[0]; d.rows.h = din + roff[1]; d.rows.A = din + roff[2]; d.rows.b1 = din + roff[3]; d.rows.Ac = din + roff[4];
    d.rows.bc = din + roff[5]; d.rows.blb = din + roff[6]; d.rows.bub = din + roff[7]; d.rows.Acop = din + roff[8];
    d.rows.tlb = din + ioff[3]; d.rows.tub = din + ioff[4]; d.rows.w = din + ioff[5];
    d.out.x = dout + o_x; d.out.tau = dout + o_tau; d.out.objective = dout + o_obj;
    d.out.status = reinterpret_cast<int32_t*>(dout + o_st); d.out.iters = reinterpret_cast<int32_t*>(dout + o_it);
    d.out.n_active = reinterpret_cast<int32_t*>(dout + o_na);
    if (io->out.active_mask) { // in/out like wbcqp_solve_batch_host: the hint goes up (zeros: none), the solution's active rows come back
        d.out.active_mask = reinterpret_cast<uint32_t*>(dout + o_am);
        HIP_TRY(h, hipMemcpyAsync(d.out.active_mask, io->out.active_mask, B * 32, hipMemcpyHostToDevice, nullptr));
    }
    d.q_next = dout + o_qn; d.v_next = dout + o_vn; d.q_solver = io->q_solver ? dout + o_qs : nullptr;
    d.dt = io->dt;
    rc = wbcqp_tick(h, slot, batch, &d, nullptr);
    if (rc != WBCQP_OK) return rc;
    if (packed) { // one copy down (the rows, when asked for, follow one by one), then taken apart on the host
        char* po = static_cast<char*>(h->pin_out.host);
        HIP_TRY(h, hipMemcpyAsync(po, dout, out_bytes, hipMemcpyDeviceToHost, nullptr));
        for (int f = 0; f < NR; ++f)
            if (rlen[f] > 0 && rdst[f]) HIP_TRY(h, hipMemcpyAsync(rdst[f], din + roff[f], (size_t)rlen[f] * B * es, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(h, hipStreamSynchronize(nullptr));
        std::memcpy(io->out.x, po + o_x, (size_t)L.n * B * es);
        if (s.host.na > 0) std::memcpy(io->out.tau, po + o_tau, (size_t)s.host.na * B * es);
        std::memcpy(io->out.status, po + o_st, B * 4);
        std::memcpy(io->out.iters, po + o_it, B * 4);
        if (io->out.objective) std::memcpy(io->out.objective, po + o_obj, B * es);
        if (io->out.n_active) std::memcpy(io->out.n_active, po + o_na, B * 4);
        if (io->out.active_mask) std::memcpy(io->out.active_mask, po + o_am, B * 32);
        std::memcpy(io->q_next, po + o_qn, (size_t)T.nq * B * es);
        std::memcpy(io->v_next, po + o_vn, (size_t)T.nv * B * es);
        if (io->q_solver) std::memcpy(io->q_solver, po + o_qs, (size_t)T.nv * B * es);
        if (io->state.momentum) std::memcpy(io->state.momentum, po + o_mom, (size_t)6 * B * es);
        return WBCQP_OK;
    }
    HIP_TRY(h, hipMemcpyAsync(io->out.x, d.out.x, (size_t)L.n * B * es, hipMemcpyDeviceToHost, nullptr));
    if (s.host.na > 0) HIP_TRY(h, hipMemcpyAsync(io->out.tau, d.out.tau, (size_t)s.host.na * B * es, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(h, hipMemcpyAsync(io->out.status, d.out.status, B * 4, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(h, hipMemcpyAsync(io->out.iters, d.out.iters, B * 4, hipMemcpyDeviceToHost, nullptr));
    if (io->out.objective) HIP_TRY(h, hipMemcpyAsync(io->out.objective, d.out.objective, B * es, hipMemcpyDeviceToHost, nullptr));
    if (io->out.n_active) HIP_TRY(h, hipMemcpyAsync(io->out.n_active, d.out.n_active, B * 4, hipMemcpyDeviceToHost, nullptr));
    if (io->out.active_mask) HIP_TRY(h, hipMemcpyAsync(io->out.active_mask, d.out.active_mask, B * 32, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(h, hipMemcpyAsync(io->q_next, d.q_next, (size_t)T.nq * B * es, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(h, hipMemcpyAsync(io->v_next, d.v_next, (size_t)T.nv * B * es, hipMemcpyDeviceToHost, nullptr));
    if (io->q_solver) HIP_TRY(h, hipMemcpyAsync(io->q_solver, d.q_solver, (size_t)T.nv * B * es, hipMemcpyDeviceToHost, nullptr));
    if (io->state.momentum) HIP_TRY(h, hipMemcpyAsync(io->state.momentum, d.state.momentum, (size_t)6 * B * es, hipMemcpyDeviceToHost, nullptr));
    for (int f = 0; f < NR; ++f)
        if (rlen[f] > 0 && rdst[f]) HIP_TRY(h, hipMemcpyAsync(rdst[f], din + roff[f], (size_t)rlen[f] * B * es, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(h, hipStreamSynchronize(nullptr));
    return WBCQP_OK;
}

struct wbcqp_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t capture = nullptr;
    OrderState ord; // the captured kernels read and write THESE buffers on every replay: nothing else ever touches them
};

int wbcqp_tick_graph_destroy(wbcqp_handle* h, wbcqp_graph* g)
{
    if (!g) return WBCQP_OK;
    if (h) (void)hipSetDevice(h->device);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->capture) (void)hipStreamDestroy(g->capture);
    if (g->ord.order) (void)hipFree(g->ord.order);
    if (g->ord.queue) (void)hipFree(g->ord.queue);
    delete g;
    return WBCQP_OK;
}

int wbcqp_tick_graph_create(wbcqp_handle* h, int slot, int batch, const wbcqp_tick_io* io, wbcqp_graph** out)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (!out || !io) return fail(h, WBCQP_ERR_INVALID, "io / out is NULL");
    if (batch <= 0) return fail(h, WBCQP_ERR_INVALID, "a graph needs a positive batch");
    *out = nullptr;
    HIP_TRY(h, hipSetDevice(h->device));
    wbcqp_graph* g = new wbcqp_graph();
    hipError_t e = hipStreamCreateWithFlags(&g->capture, hipStreamNonBlocking);
    if (e != hipSuccess) { delete g; return fail(h, WBCQP_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    // the graph's own launch-order buffer and queue counter, at their final size: a replay reads and rewrites them and
    // nothing else does (the handle's own buffer may be resized or overwritten by any other launch)
    e = hipMalloc(&g->ord.order, 2 * sizeof(int) * (size_t)batch);
    if (e == hipSuccess) e = hipMalloc(&g->ord.queue, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(g->ord.queue, 0, 2 * sizeof(int));
    if (e != hipSuccess) { wbcqp_tick_graph_destroy(h, g); return fail(h, WBCQP_ERR_HIP, std::string("graph buffers: ") + hipGetErrorString(e)); }
    g->ord.cap = batch;
    // an ordinary tick on the capture stream first: function attributes are then settled, the graph's order buffer holds a
    // valid order for this stream and shape, and the capture below contains kernel launches only
    h->graph_ord = &g->ord;
    int rc = wbcqp_tick(h, slot, batch, io, g->capture);
    if (rc == WBCQP_OK && hipStreamSynchronize(g->capture) != hipSuccess) rc = fail(h, WBCQP_ERR_HIP, "warm-up tick failed");
    if (rc != WBCQP_OK) { h->graph_ord = nullptr; wbcqp_tick_graph_destroy(h, g); return rc; }
    e = hipStreamBeginCapture(g->capture, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { h->graph_ord = nullptr; wbcqp_tick_graph_destroy(h, g); return fail(h, WBCQP_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e)); }
    h->capturing = true;
    rc = wbcqp_tick(h, slot, batch, io, g->capture);
    h->capturing = false;
    h->graph_ord = nullptr;
    e = hipStreamEndCapture(g->capture, &g->graph);
    if (rc != WBCQP_OK || e != hipSuccess || !g->graph) {
        wbcqp_tick_graph_destroy(h, g);
        return rc != WBCQP_OK ? rc : fail(h, WBCQP_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { wbcqp_tick_graph_destroy(h, g); return fail(h, WBCQP_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
    *out = g;
    return WBCQP_OK;
}

int wbcqp_tick_graph_launch(wbcqp_handle* h, wbcqp_graph* g, void* stream)
{
    if (!h) return WBCQP_ERR_INVALID;
    if (!g || !g->exec) return fail(h, WBCQP_ERR_INVALID, "graph is NULL");
    HIP_TRY(h, hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream)));
    return WBCQP_OK;
}

int wbcqp_sync(wbcqp_handle* h, void* stream)
{
    if (!h) return WBCQP_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return WBCQP_OK;
}

} // extern "C"
