/*
 * msd_capi.cpp -- the C-ABI of include/modes_hip.h: context, device memory, launch order, the
 * two-deep batch pipeline, and the hand-off to the ordered resolve stage (msd_resolve.c).
 *
 * Stream layout per context:
 *   compute stream : memset sums -> scan kernel -> offsets+gather [-> float means] (per batch)
 *   copy stream    : waits on the batch's "kernels done" event, then D2H totals / lists / sums,
 *                    so a batch's download overlaps the next batch's kernels.
 *   aux stream     : the signal-power round trip of the batch being resolved.
 * With three batches in flight (msd_launch_device / msd_collect), while the host resolves batch k
 * the lists of batch k+1 come down and the GPU scans batch k+2; msd_submit_* is the depth-1
 * synchronous form.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "modes_hip.h"
#include "msd_internal.h"
#include "msd_kernels.h"

extern "C" int msd_tables_selftest(const msd_tables *t);

namespace {

constexpr uint64_t MIN_HIT_ARENA = 131072;      /* every position of one buffer */
constexpr uint64_t MIN_TRY_ARENA = 131072 * 5;  /* every phase of every position of one buffer */
constexpr int TAIL_SAMPLES = MSD_HALO_FRONT;

/* One helper thread per context for the per-message part of finishing a batch (signal level, power
 * statistics, the copy into the caller's arrays), so that it overlaps with the calling thread queueing
 * the next batch's resolve.  At most one job at a time; run() returns at once, wait() joins it. */
struct Helper {
    /* one worker thread, jobs in order.  A job may call mark_delivered() when the part its poster waits for is
     * done; what it does after that is background work that the next job queues up behind.  Both sides spin for
     * a few hundred microseconds before they sleep: in a running stream the next event is never further away,
     * and a sleeping thread on a busy host comes back late. */
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    std::atomic<uint64_t> posted{0}, delivered{0}, finished{0}; /* jobs posted / past their delivery point / complete */
    bool stop = false;
    int device = 0;
    static void relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    template <typename Pred>
    static bool spin_for(Pred pred, int microseconds)
    {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(microseconds);
        for (;;) {
            for (int i = 0; i < 64; ++i) {
                if (pred())
                    return true;
                relax();
            }
            if (std::chrono::steady_clock::now() >= until)
                return false;
        }
    }
    void loop()
    {
        (void)hipSetDevice(device);
        uint64_t taken = 0;
        for (;;) {
            (void)spin_for([&] { return posted.load(std::memory_order_acquire) > taken; }, 2000);
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !jobs.empty() || stop; });
            if (jobs.empty())
                return; /* stop, and nothing left to do */
            std::function<void()> job = std::move(jobs.front());
            jobs.pop_front();
            ++taken;
            lk.unlock();
            job();
            lk.lock();
            finished.store(taken, std::memory_order_release);
            if (delivered.load(std::memory_order_relaxed) < taken)
                delivered.store(taken, std::memory_order_release);
            cv.notify_all();
        }
    }
    void run(std::function<void()> f) /* does not wait: the job starts when the ones before it are complete */
    {
        std::unique_lock<std::mutex> lk(mu);
        if (!th.joinable())
            th = std::thread([this] { loop(); });
        jobs.push_back(std::move(f));
        posted.fetch_add(1, std::memory_order_release);
        cv.notify_all();
    }
    void mark_delivered() /* from the running job */
    {
        std::unique_lock<std::mutex> lk(mu);
        delivered.store(finished.load(std::memory_order_relaxed) + 1, std::memory_order_release);
        cv.notify_all();
    }
    void wait_delivered() /* the last job posted has passed its delivery point */
    {
        const uint64_t want = posted.load(std::memory_order_acquire);
        if (spin_for([&] { return delivered.load(std::memory_order_acquire) >= want; }, 2000))
            return;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return delivered.load(std::memory_order_acquire) >= want; });
    }
    void wait() /* everything posted is complete */
    {
        const uint64_t want = posted.load(std::memory_order_acquire);
        if (spin_for([&] { return finished.load(std::memory_order_acquire) >= want; }, 200))
            return;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return finished.load(std::memory_order_acquire) >= want; });
    }
    void shutdown()
    {
        wait();
        {
            std::unique_lock<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        if (th.joinable())
            th.join();
    }
};

/* hipEventSynchronize for events that are about to fire: poll for two milliseconds first -- eight batch periods; the
 * runtime's wait may put the thread to sleep, and on a busy host it then comes back late, which the in-order chain
 * feels at once (the next resolve pass can only be queued when this one has reported) */
static hipError_t event_wait(hipEvent_t ev)
{
    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(2000);
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady)
            return e;
        if (std::chrono::steady_clock::now() >= until)
            return hipEventSynchronize(ev);
        Helper::relax();
    }
}

struct Slot {
    bool busy = false;
    bool download_started = false;
    bool gpu_resolve = false; /* the candidate lists stay in HBM: resolved there (msd_resolve_kernels.hip) */
    bool resolve_inflight = false; /* its first resolve pass (and the speculative message records) are queued */
    int threshold = 0;             /* Modes.preambleThreshold when the batch was launched */
    bool timed = false;            /* ev_start / ev_scan / ev_kernels were recorded for this batch */
    bool state_reset_done = false; /* ... and filter and clocks have been reset already (its chain was queued early) */
    bool reset_before = false;     /* msd_restart(): first batch of a new capture -- filter, clock and counters start
                                      over when its turn comes */
    bool dc = false;               /* --dcfilter: d_iq points at d_dcmag, the float sums come from d_magsq */
    uint16_t *d_dcmag = nullptr;   /* DC-blocked magnitudes of the batch (what the scan kernel reads) */
    float *d_magsq = nullptr;      /* their clamped squares, for the per-buffer float sums */
    uint64_t dropped_before = 0;   /* msd_note_dropped(): samples missing in front of this batch, not yet on the clock */
    uint32_t resolve_ntodo = 0;
    msd_rbuf *h_rbuf = nullptr;    /* pinned; the resolve kernel reports straight into it */
    msd_acc *d_acc = nullptr;
    uint32_t *d_adds = nullptr, *d_nmsgs = nullptr, *d_acc_ac = nullptr, *d_nac = nullptr;
    uint32_t *d_pred = nullptr; /* the batch's prediction table (msd_internal.h: MSD_PRED_WORDS), filled by its scan */
    uint32_t pred_gen = 0, pred_uses = 0; /* its generation for the batch in the slot; batches it has served */
    /* lean layout (UC8 / magnitudes, Mode S only, chain in order, resolve on the GPU): no gather kernel -- the
     * candidate lists stay in this slot's own region arenas until the batch's records are out, the resolve
     * workgroups read their buffer's region slices, the first resolve pass publishes sums and totals */
    bool lean = false;
    bool power_done = false; /* the batch's signal power kernel has been queued (d_powr, d_rec_off) */
    bool ahead_done = false; /* its resolve passes are through and its filter changes committed (by the msd_collect of
                                the batch before it); counters and delivery wait for its own msd_collect */
    int ahead_verdict = 0;   /* 1 / 2: the msd_collect before this batch's already found that the host resolver has to take
                                it / that its arenas overflowed (resolve_passes); nothing was committed */
    bool records_current = true; /* no further resolve pass ran after the one whose records were written */
    uint32_t npass = 0;
    uint64_t sample_counter0 = 0; /* the sample clock at the batch's first sample (gpu_begin) */
    msd_hit *d_rhits = nullptr;
    msd_try *d_rtries = nullptr;
    msd_region_counts *d_rcounts = nullptr;
    msd_wg_totals *d_rwgt = nullptr;
    uint32_t lean_k = 0, lean_hcap = 0, lean_tcap = 0, lean_nreg = 0; /* regions per buffer, slice capacities, regions */
    /* the slot's own arena sizes: the context's (msd_config) to begin with; grow_and_rescan() enlarges the region slices of
     * a slot whose batch overflowed them, lean_gather_now() the dense lists if somebody on the host wants such a batch */
    uint64_t rhit_arena = 0, rtry_arena = 0, dense_hits = 0, dense_tries = 0;
    uint64_t *d_powr = nullptr; /* [buffer][MSD_RB_MSG_CAP] signal power of the accepted messages */
    uint8_t *h_ctl = nullptr; /* pinned, read by the kernels in place: ts[2n] u64 | valid[n] | snap_idx[n] | todo[n] */
    msd_wire *d_wire = nullptr, *h_wire = nullptr; /* message records of the emit kernel and their pinned copy */
    unsigned long long *h_side = nullptr;          /* per record: power sum | signal_len << 48, for the statistics */
    msd_fields *h_fields = nullptr; /* pinned: header fields next to the records (MSD_CFG_DECODE_FIELDS) */
    msd_fields *d_fields = nullptr; /* their device copy when the records travel by DMA (records_dma) */
    hipEvent_t ev_resolve = nullptr, ev_records = nullptr, ev_power = nullptr;
    hipEvent_t ev_scanned = nullptr; /* side-stream layout: this batch's scan + gather are done (its float sums / Mode A/C kernels follow) */
    uint64_t launch_seq = 0;         /* running number of the launch that filled the slot */
    /* batch description */
    const uint8_t *d_iq = nullptr;
    const uint8_t *d_prev = nullptr;
    int have_prev = 0;
    uint64_t batch_first = 0; /* absolute sample index */
    uint64_t nsamples = 0;
    uint32_t nbuffers = 0;
    int last = 0;
    /* device */
    msd_hit *d_hits = nullptr;
    msd_try *d_tries = nullptr;
    uint64_t *d_totals = nullptr;
    float *d_tile_sums = nullptr;    /* SC16 / SC16Q11: the scan's per-tile float sums, for the float-sum kernel's predictions */
    void *d_fm_work = nullptr;       /* 16-bit IQ, --dcfilter: the float-sum kernels' hand-over (msd_fm_work_bytes) */
    msd_ac_hit *d_ac_regions = nullptr; /* Mode A/C: the candidate kernel's region slices and counts, gathered into d_ac */
    msd_wg_counts *d_ac_counts = nullptr;
    uint32_t *d_rec_off = nullptr;   /* [max_buffers + 2] records in front of each buffer's (power kernel) */
    uint32_t *d_buf_first = nullptr; /* [max_buffers + 2] start of each buffer's hits in d_hits (gather kernel) */
    bool buf_first_valid = false;
    uint64_t *d_sums = nullptr;
    uint16_t *d_mag = nullptr;        /* Mode A/C: the batch's magnitudes as the scan computed them (MsdScanParams.mag_out) */
    const uint16_t *d_mag_prev = nullptr; /* ... and the last MSD_HALO_FRONT of the batch before, in that batch's own array */
    bool mag_pass = false;            /* this batch's Mode A/C candidate kernel reads d_mag */
    float *d_fmeans = nullptr;
    /* pinned host */
    uint64_t *h_totals = nullptr;
    uint64_t *h_sums = nullptr;
    float *h_fmeans = nullptr;
    msd_hit *h_hits = nullptr;
    size_t h_hits_cap = 0;
    msd_try *h_tries = nullptr;
    size_t h_tries_cap = 0;
    uint8_t *d_ragged = nullptr; /* zero-padded copy of a partially filled last 8-sample group */
    uint8_t *tail_dst = nullptr; /* where the gather kernel leaves the batch's last samples for its successor */
    uint8_t *d_upload = nullptr; /* msd_launch_host: this slot's copy of the batch in HBM */
    hipEvent_t ev_upload = nullptr;
    /* Mode A/C candidates */
    msd_ac_hit *d_ac = nullptr;
    uint64_t *d_ac_totals = nullptr, *h_ac_totals = nullptr;
    msd_ac_hit *h_ac = nullptr;
    size_t h_ac_cap = 0;
    /* deferred signal power of the accepted messages */
    uint64_t *d_req = nullptr, *d_pow = nullptr, *h_req = nullptr, *h_pow = nullptr;
    size_t req_cap = 0;
    hipEvent_t ev_start = nullptr, ev_scan = nullptr, ev_kernels = nullptr, ev_totals = nullptr,
               ev_copy0 = nullptr, ev_copy1 = nullptr;
};

} /* namespace */

struct msd_ctx {
    msd_config cfg{};
    hipStream_t stream = nullptr, copy_stream = nullptr, aux_stream = nullptr, emit_stream = nullptr;
    bool own_stream = false;
    int bps = 2;
    msd_tables *tables = nullptr;
    uint16_t *d_lut = nullptr;
    uint32_t *d_crc = nullptr, *d_syn56 = nullptr, *d_syn112 = nullptr, *d_slicer = nullptr, *d_synhash = nullptr;
    uint64_t *d_fix2[2] = {nullptr, nullptr}; /* two-bit correction tables for 56 / 112 bits (nfix_crc == 2) */
    uint32_t fix2_lg[2] = {0, 0};
    /* per-workgroup candidate regions (shared by all batches: stream order serialises them) */
    msd_hit *d_region_hits = nullptr;
    msd_try *d_region_tries = nullptr;
    uint64_t hit_arena = 0, try_arena = 0;
    uint64_t *h_conv = nullptr; /* msd_convert_begin / _end: the sums of the conversion in flight (page-locked) */
    bool conv_pending = false;
    unsigned conv_n = 0;
    const msd_magbuf_view *magbuf_views = nullptr; /* msd_demodulate_magbufs: the caller's buffers while its finish() runs */
    const uint32_t *magbuf_noise = nullptr;        /* ... and their Mode A/C noise levels (demod_2400.c:530-531 from the caller's
                                                      means), for a batch that has to be scanned again in pieces */
    unsigned magbuf_nviews = 0;
    double want_hits_per_sample = 0, want_tries_per_sample = 0; /* region slices a slot should have at its next launch (grow_and_rescan) */
    msd_region_counts *d_counts = nullptr; /* per region (wavefront) of the scan kernel */
    msd_wg_totals *d_wg_totals = nullptr;  /* per workgroup of the scan kernel */
    uint32_t max_wg = 0, max_buffers = 0;  /* max_wg: most regions a scan is split into */
    /* Mode A/C candidate regions */
    uint64_t ac_arena = 0;
    uint64_t *d_ac_offsets = nullptr;
    uint32_t *d_noise = nullptr;
    void *d_fm_work = nullptr;       /* 16-bit IQ, --dcfilter: the float-sum kernels' hand-over (msd_fm_work_bytes) */
    uint32_t ac_max_wg = 0;
    unsigned long long *d_timers = nullptr; /* MSD_KERNEL_TIMING experiments */
    /* GPU resolve stage: per-buffer reports, accepted-message records, filter snapshots, control arrays */
    bool gpu_resolve = false;
    uint32_t *d_snaps = nullptr, *h_snaps = nullptr;
    uint32_t snaps_uploaded = 0;
    uint32_t inline_adds = MSD_RB_ADD_INLINE; /* msd_config.test_inline_adds lowers it */
    bool want_fields = false;       /* MSD_CFG_DECODE_FIELDS */
    msd_fields_fn fsink = nullptr;  /* set while msd_collect_fields runs: messages go here with their fields */
    void *fuser = nullptr;
    std::vector<msd_fields> out_fields; /* host-resolve path */
    bool records_dma = false; /* MSD_RECORDS_DMA=1: fetch the message records with a DMA instead of kernel stores */
    hipEvent_t ev_aux = nullptr, ev_inputs = nullptr;
    msd_pred_entry *h_pred = nullptr;
    uint32_t *h_pred_count = nullptr;
    msd_pred_patch *h_patches = nullptr;
    uint32_t npatches = 0;
    /* the last MSD_HALO_FRONT samples of the previous batch, one buffer per pipeline stage + 1 */
    uint8_t *d_tail[MSD_PIPELINE_DEPTH + 1] = {};
    int tail_cur = 0;
    bool have_prev = false;
    uint8_t *d_stage = nullptr; /* msd_submit_host / msd_convert / msd_demodulate_magbuf staging */
    uint16_t *d_mag = nullptr;
    Slot slots[MSD_PIPELINE_DEPTH];
    int head = 0, outstanding = 0;
    uint64_t next_sample = 0;
    bool finished = false;
    uint64_t pending_dropped = 0; /* msd_note_dropped() since the last launch */
    bool restart_pending = false; /* msd_restart() since the last launch */
    const uint16_t *mag_prev = nullptr; /* Mode A/C: where the previous batch's last magnitudes are (its slot's d_mag) */
    uint32_t timing_interval = 1; /* msd_set_timing_interval() */
    /* experiment knobs, read from the environment once in msd_create */
    bool trace = false;      /* MSD_RESOLVE_TRACE */
    bool repass_aux = false; /* MSD_REPASS_AUX */
    /* In-order layout without field decoding: the record kernel of a batch is not launched; the wavefronts of the
     * next scan write the records on their way in (MsdScanParams.emit).  pending_emit: resolve chain and signal
     * power queued, records not yet.  MSD_EMIT_FUSED=0 turns it off. */
    std::vector<uint32_t> bg_valid, bg_buf; /* the statistics half of finishing a batch, on the helper thread */
    std::vector<double> bg_means;
    std::vector<uint64_t> bg_scaled; /* per message: power sum | signal_len << 48 (msd_emit_impl.h) */
    bool emit_fused = false;
    bool power_fused = true; /* no signal power kernel: the resolve workgroups sum it (MSD_POWER_FUSED=0 keeps the kernel) */
    struct Slot *pending_emit = nullptr;
    bool chain_inline = true; /* MSD_CHAIN_INLINE=0: resolve chain on side streams instead of in order on the scan stream */
    bool lean_ok = false;     /* the configuration allows the lean layout (Slot::lean; MSD_LEAN=0 turns it off) */
    bool wait_inputs_on_stream = false; /* MSD_WAIT_INPUTS_ON_STREAM=1: the resolve kernel's stream waits for the snapshot upload */
    bool resolve_ahead = true; /* msd_collect also takes the next batch through its resolve passes (MSD_RESOLVE_AHEAD=0: no) */
    int debug_flags = 0;     /* MSD_DEBUG_FLAGS */
    uint64_t enqueue_seq = 0;
    uint64_t launch_count = 0;
    bool dc = false;              /* MSD_CFG_DC_FILTER */
    int q11_bits = 0;             /* msd_config.sc16q11_table_bits in effect: the batches' IQ goes through d_q11_table first */
    uint16_t *d_q11_table = nullptr;
    float *d_conv_magsq = nullptr; /* msd_convert of a MSD_CFG_DC_FILTER context: the clamped squares of the call's samples */
    float dc_a = 0, dc_b = 1;     /* struct converter_state, convert.c:28-33,479-482 */
    float *d_dcstate = nullptr;   /* z1_I, z1_Q on the device, carried from batch to batch */
    void *d_dc_work = nullptr;    /* the parallel-in-time DC filter's blocks, tables and control word (msd_dcp_work_bytes) */
    bool dc_last_parallel = false; /* the most recent DC block went through the parallel kernels (msd_dc_filter_status) */
    uint32_t dc_last_blocks = 0;
    bool dc_fused = false;        /* MSD_CFG_DC_FUSED_LAUNCH: the passes in one cooperative launch (measured slower) */
    int dc_passes = 24;           /* passes queued per batch (MSD_CFG_DC_ONE_PASS: 1, so that the in-order kernel behind them runs) */
    int scan_format = 0;          /* what the scan and its follow-up kernels read: cfg.format, or MAG16 behind the DC filter */
    size_t scan_bps = 2;
    msd_resolver resolver{};
    msd_stats stats{};
    msd_timing timing{};
    std::vector<double> means;
    std::vector<uint32_t> valid;
    std::vector<msd_message> out_msgs;
    std::vector<uint64_t> out_req;
    std::vector<uint32_t> out_buf;
    int cu_count = 256;
    Helper helper;
    bool failed = false;    /* a batch could not be finished: only msd_reset() / msd_destroy() are accepted */
    bool scan_queued = false; /* enqueue(): its scan kernel is on the stream (a later failure cannot be undone) */
    bool no_helper = false; /* MSD_NO_HELPER: everything on the calling thread */
    char err[256] = {0};
};

namespace {

/* why the calling thread's last msd_create failed (there is no context to hold the text yet);
 * msd_last_error(NULL) returns it */
thread_local char g_create_err[256] = {0};

int fail(msd_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((c), -EIO, "%s failed: %s", #call, hipGetErrorString(e_));              \
    } while (0)

void emit_thunk(const msd_message *mm, const uint64_t *power_req, uint32_t count, uint32_t buffer, void *user)
{
    msd_ctx *c = static_cast<msd_ctx *>(user);
    c->out_msgs.insert(c->out_msgs.end(), mm, mm + count);
    c->out_req.insert(c->out_req.end(), power_req, power_req + count);
    c->out_buf.insert(c->out_buf.end(), count, buffer);
}

int ensure_req(msd_ctx *c, Slot &s, size_t n)
{
    if (n <= s.req_cap)
        return 0;
    size_t cap = s.req_cap ? s.req_cap : (size_t)1 << 14;
    while (cap < n)
        cap *= 2;
    (void)hipFree(s.d_req); (void)hipFree(s.d_pow);
    if (s.h_req) (void)hipHostFree(s.h_req);
    if (s.h_pow) (void)hipHostFree(s.h_pow);
    if (s.h_wire) (void)hipHostFree(s.h_wire);
    if (s.h_side) (void)hipHostFree(s.h_side);
    if (s.h_fields) (void)hipHostFree(s.h_fields);
    (void)hipFree(s.d_wire);
    (void)hipFree(s.d_fields);
    s.d_fields = nullptr;
    s.d_req = s.d_pow = s.h_req = s.h_pow = nullptr;
    s.h_wire = s.d_wire = nullptr;
    s.h_side = nullptr;
    s.h_fields = nullptr;
    s.req_cap = 0;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&s.d_req), cap * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&s.d_pow), cap * sizeof(uint64_t)));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&s.h_req), cap * sizeof(uint64_t)));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&s.h_pow), cap * sizeof(uint64_t)));
    if (c->gpu_resolve) {
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&s.h_wire), cap * sizeof(msd_wire)));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&s.h_side), cap * sizeof(unsigned long long)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&s.d_wire), cap * sizeof(msd_wire)));
        if (c->want_fields)
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&s.h_fields), cap * sizeof(msd_fields)));
        if (c->want_fields && c->records_dma)
            HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&s.d_fields), cap * sizeof(msd_fields)));
    }
    s.req_cap = cap;
    return 0;
}

void fill_params(const msd_ctx *c, const Slot &s, MsdScanParams &p)
{
    p.iq = s.d_iq;
    p.prev_tail = s.d_prev;
    p.ragged = s.d_ragged;
    p.have_prev = s.have_prev;
    p.threshold = s.threshold;
    p.batch_first = s.batch_first;
    p.nsamples = s.nsamples;
    p.lut = c->d_lut;
    p.crc_tab = c->d_crc;
    p.syn56 = c->d_syn56;
    p.slicer = c->d_slicer;
    p.syn112 = c->d_syn112;
    p.nsyn56 = c->tables->nsyn56;
    p.nsyn112 = c->tables->nsyn112;
    p.synhash = c->d_synhash;
    p.synh_mul56 = c->tables->synhash_mul[0];
    p.synh_mul112 = c->tables->synhash_mul[1];
    p.fix2_56 = c->d_fix2[0];
    p.fix2_112 = c->d_fix2[1];
    p.fix2_lg56 = c->fix2_lg[0];
    p.fix2_lg112 = c->fix2_lg[1];
}

int ensure_host(msd_ctx *c, Slot &s, size_t nh, size_t nt)
{
    if (nh > s.h_hits_cap) {
        size_t cap = s.h_hits_cap ? s.h_hits_cap : (size_t)1 << 16;
        while (cap < nh)
            cap *= 2;
        if (s.h_hits)
            (void)hipHostFree(s.h_hits);
        s.h_hits = nullptr;
        s.h_hits_cap = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&s.h_hits), cap * sizeof(msd_hit)));
        s.h_hits_cap = cap;
    }
    if (nt > s.h_tries_cap) {
        size_t cap = s.h_tries_cap ? s.h_tries_cap : (size_t)1 << 15;
        while (cap < nt)
            cap *= 2;
        if (s.h_tries)
            (void)hipHostFree(s.h_tries);
        s.h_tries = nullptr;
        s.h_tries_cap = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&s.h_tries), cap * sizeof(msd_try)));
        s.h_tries_cap = cap;
    }
    return 0;
}

size_t bps_of(int format)
{
    return (format == MSD_FMT_UC8 || format == MSD_FMT_MAG16) ? 2 : 4;
}

/* Enqueue the GPU stage for `nsamples` samples at d_iq (absolute index batch_first). */
int flush_pending_emit(msd_ctx *c);
struct GpuCtl;
void gpu_params(const msd_ctx *c, const Slot &s, MsdResolveParams &rp);

bool gpu_eligible(const msd_ctx *c, const Slot &s);

/* The arenas of the layouts without region slices (a batch that is not lean: msd_submit_*-sized batches of fewer than four
 * buffers, the mag_buf entry, MSD_CFG_NO_LEAN / MSD_CFG_HOST_RESOLVE contexts, the pieces of rerun_in_pieces), made at
 * first use. */
int ensure_dense(msd_ctx *c, Slot &s)
{
    if (!c->d_region_hits)
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_region_hits), c->hit_arena * sizeof(msd_hit)));
    if (!c->d_region_tries)
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_region_tries), c->try_arena * sizeof(msd_try)));
    if (!s.d_hits || s.dense_hits < c->hit_arena) {
        if (s.d_hits) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            (void)hipFree(s.d_hits);
            s.d_hits = nullptr;
        }
        s.dense_hits = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&s.d_hits), c->hit_arena * sizeof(msd_hit)));
        s.dense_hits = c->hit_arena;
    }
    if (!s.d_tries || s.dense_tries < c->try_arena) {
        if (s.d_tries) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            (void)hipFree(s.d_tries);
            s.d_tries = nullptr;
        }
        s.dense_tries = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&s.d_tries), c->try_arena * sizeof(msd_try)));
        s.dense_tries = c->try_arena;
    }
    return 0;
}

int enqueue(msd_ctx *c, Slot &s, int format, const uint32_t *host_noise, bool pipelined = false)
{
    const uint64_t tile = msd_scan_tile(format);
    const uint64_t ntiles64 = (s.nsamples + tile - 1) / tile;
    const uint32_t ntiles = (uint32_t)ntiles64;
    /* one region per wavefront: MSD_SCAN_WAVES per CU */
    uint32_t target_wg = c->max_wg;
    uint32_t tpw = ntiles ? (ntiles + target_wg - 1) / target_wg : 1;
    if (tpw == 0)
        tpw = 1;
    uint32_t nwg = ntiles ? (ntiles + tpw - 1) / tpw : 0;
    /* Lean layout: k regions per buffer, none across a buffer boundary.  Only for batches that come through
     * msd_launch_* and will be resolved on the GPU in order on this stream. */
    const uint32_t tiles_per_buffer = (uint32_t)(MSD_CHUNK_SAMPLES / tile);
    uint32_t lean_k = 0, lean_tpr = 0;
    s.lean = false;
    s.mag_pass = false; /* set by the scan block below; an empty batch has none and must not inherit the slot's last one */
    if (pipelined && c->lean_ok && nwg && gpu_eligible(c, s) && s.nbuffers <= c->max_wg && !(c->debug_flags & 0x1f)) {
        /* (the buffers that hold samples: a capture's last batch ends with one more, empty or short, buffer --
         * counting it would cost a full batch of 512 buffers an eighth of its regions, 4096 / 513 = 7) */
        const uint32_t nb_data = (uint32_t)((s.nsamples + MSD_CHUNK_SAMPLES - 1) / MSD_CHUNK_SAMPLES);
        uint32_t k = c->max_wg / (nb_data ? nb_data : 1);
        if (k > 64)
            k = 64;
        if (k > tiles_per_buffer)
            k = tiles_per_buffer;
        lean_tpr = (tiles_per_buffer + k - 1) / k;
        lean_k = (tiles_per_buffer + lean_tpr - 1) / lean_tpr; /* no empty pieces */
        nwg = (nb_data ? nb_data : 1) * lean_k;
        tpw = lean_tpr;
        s.lean = true;
        /* another slot's batch overflowed its region slices and got bigger ones (grow_and_rescan): this slot follows before
         * it meets the same traffic -- it is idle now, its last batch has been collected; a failed allocation leaves it as it is */
        const uint64_t want_h = (uint64_t)(c->want_hits_per_sample * (double)s.nsamples), want_t = (uint64_t)(c->want_tries_per_sample * (double)s.nsamples);
        if ((want_h > s.rhit_arena + s.rhit_arena / 64 || want_t > s.rtry_arena + s.rtry_arena / 64) && want_t < (1ull << 30)) { /* (not for a rounding's worth) */
            size_t free_b = 0, total_b = 0;
            const uint64_t grow_b = (want_h > s.rhit_arena ? (want_h - s.rhit_arena) * sizeof(msd_hit) : 0) +
                                    (want_t > s.rtry_arena ? (want_t - s.rtry_arena) * sizeof(msd_try) : 0);
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && grow_b + (1ull << 30) <= free_b) {
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (want_h > s.rhit_arena) {
                    msd_hit *nh = nullptr;
                    if (hipMalloc(reinterpret_cast<void **>(&nh), want_h * sizeof(msd_hit)) == hipSuccess) {
                        (void)hipFree(s.d_rhits);
                        s.d_rhits = nh;
                        s.rhit_arena = want_h;
                    } else {
                        (void)hipGetLastError();
                    }
                }
                if (want_t > s.rtry_arena) {
                    msd_try *nt = nullptr;
                    if (hipMalloc(reinterpret_cast<void **>(&nt), want_t * sizeof(msd_try)) == hipSuccess) {
                        (void)hipFree(s.d_rtries);
                        s.d_rtries = nt;
                        s.rtry_arena = want_t;
                    } else {
                        (void)hipGetLastError();
                    }
                }
            }
        }
    }

    if (!s.lean && nwg) {
        const int erc = ensure_dense(c, s);
        if (erc)
            return erc;
    }
    if (s.nsamples & 7u) { /* the last, partially filled 8-sample group: a zero-padded private copy */
        const size_t bps = (format == MSD_FMT_UC8 || format == MSD_FMT_MAG16) ? 2 : 4;
        HIPCHK(c, hipMemsetAsync(s.d_ragged, 0, 64, c->stream));
        HIPCHK(c, hipMemcpyAsync(s.d_ragged, s.d_iq + (s.nsamples & ~7ull) * bps, (s.nsamples & 7u) * bps,
                                 hipMemcpyDeviceToDevice, c->stream));
    }
    /* d_sums is zero: whoever published the slot's previous batch left it so.  The offsets kernel
     * overwrites the totals. */
    const bool fm = format == MSD_FMT_SC16 || format == MSD_FMT_SC16Q11 || s.dc;
    const bool gather_publishes = nwg && !c->cfg.mode_ac && !fm; /* totals and sums are published by the gather kernel */
    if (!nwg) {
        HIPCHK(c, hipMemsetAsync(s.d_totals, 0, sizeof(uint64_t) * 4, c->stream));
        s.buf_first_valid = false;
    }
    if (c->cfg.mode_ac && !(s.nbuffers && s.nsamples)) /* (msd_launch_ac's offsets kernel writes them otherwise) */
        HIPCHK(c, hipMemsetAsync(s.d_ac_totals, 0, sizeof(uint64_t) * 4, c->stream));
    /* the three timing events cost about 5 us of stream time each (a barrier packet per record): they are
     * recorded for one batch in every c->timing_interval */
    s.timed = c->timing_interval && (c->enqueue_seq++ % c->timing_interval) == 0;
    if (s.timed)
        HIPCHK(c, hipEventRecord(s.ev_start, c->stream));
    if (nwg) {
        MsdScanParams p{};
        fill_params(c, s, p);
        p.ntiles = ntiles;
        p.tiles_per_wg = tpw;
        p.hits = s.lean ? s.d_rhits : c->d_region_hits;
        p.tries = s.lean ? s.d_rtries : c->d_region_tries;
        /* a tile can never produce more than one hit per position and five tries per hit */
        uint64_t hcap = (s.lean ? s.rhit_arena : c->hit_arena) / nwg, tcap = (s.lean ? s.rtry_arena : c->try_arena) / nwg;
        if (hcap > (uint64_t)tpw * tile)
            hcap = (uint64_t)tpw * tile;
        if (tcap > (uint64_t)tpw * tile * 5)
            tcap = (uint64_t)tpw * tile * 5;
        p.hcap = (uint32_t)hcap;
        p.tcap = (uint32_t)tcap;
        s.lean_hcap = p.hcap;
        s.lean_tcap = p.tcap;
        p.counts = s.lean ? s.d_rcounts : c->d_counts;
        p.wg_totals = s.lean ? s.d_rwgt : c->d_wg_totals;
        if (s.d_pred && s.gpu_resolve) { /* the batch's prediction table: a generation of its own, no wipe */
            s.pred_gen = s.pred_uses % MSD_PRED_GENS;
            if (s.pred_uses && s.pred_gen == 0) /* the 8-bit generations have come round: old entries must go */
                HIPCHK(c, hipMemsetAsync(s.d_pred, 0xFF, sizeof(uint32_t) * MSD_PRED_WORDS, c->stream));
            s.pred_uses++;
            p.pred = reinterpret_cast<unsigned long long *>(s.d_pred);
            p.pred_gen = s.pred_gen;
        }
        const bool tail_here = s.tail_dst && s.nsamples >= (uint64_t)TAIL_SAMPLES &&
                               (((s.nsamples - TAIL_SAMPLES) * bps_of(format)) & 3u) == 0; /* copied as dwords */
        if (s.lean) {
            p.regions_per_buffer = lean_k;
            p.tiles_per_region = lean_tpr;
            p.overflow = reinterpret_cast<unsigned long long *>(s.d_totals + 2);
            if (tail_here) {
                p.tail_src = reinterpret_cast<const uint32_t *>(s.d_iq + (s.nsamples - TAIL_SAMPLES) * bps_of(format));
                p.tail_dst = reinterpret_cast<uint32_t *>(s.tail_dst);
                p.tail_words = (uint32_t)(TAIL_SAMPLES * bps_of(format) / 4);
            }
            s.lean_k = lean_k;
            s.lean_nreg = nwg;
        }
        p.chunk_sums = s.d_sums;
        s.mag_pass = c->cfg.mode_ac && s.d_mag && pipelined && nwg && !host_noise;
        p.mag_out = s.mag_pass ? s.d_mag : nullptr;
        p.tile_sums = (fm && !s.dc && tile == 1024) ? s.d_tile_sums : nullptr;
        p.timers = c->d_timers;
        p.debug_flags = c->debug_flags;
        Slot *carried = nullptr;
        if (c->pending_emit && c->pending_emit != &s) {
            Slot &a = *c->pending_emit;
            if (nwg >= a.nbuffers && a.nbuffers) { /* this scan's wavefronts write that batch's records */
                MsdResolveParams rp{};
                gpu_params(c, a, rp);
                p.emit.nbuffers = a.nbuffers;
                p.emit.stride = nwg / a.nbuffers;
                p.emit.cap = (uint32_t)a.req_cap;
                p.emit.totals = rp.totals;
                p.emit.nmsgs = rp.nmsgs;
                p.emit.rec_off = a.power_done ? a.d_rec_off : nullptr;
                p.emit.acc = rp.acc;
                p.emit.tries = rp.tries;
                p.emit.ts = rp.ts;
                p.emit.power = reinterpret_cast<const unsigned long long *>(a.d_powr);
                p.emit.dense = c->records_dma ? a.d_wire : a.h_wire;
                p.emit.side = a.h_side;
                p.emit.ac = rp.ac;
                p.emit.ac_totals = rp.ac_totals;
                p.emit.acc_ac = rp.acc_ac;
                p.emit.nac = rp.nac;
                carried = &a;
                c->pending_emit = nullptr;
            } else {
                int rc = flush_pending_emit(c);
                if (rc)
                    return rc;
            }
        }
        c->scan_queued = false;
        int rc = msd_launch_scan(&p, format, nwg, c->stream);
        if (rc) {
            /* nothing was queued: the older batch's records are still owed (finish_gpu / flush_pending_emit write
             * them), as the header promises for a failed launch */
            if (carried)
                c->pending_emit = carried;
            return fail(c, rc, "scan kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        }
        c->scan_queued = true;
        if (s.timed)
            HIPCHK(c, hipEventRecord(s.ev_scan, c->stream));
        if (carried) {
            /* the records are on their way with this scan; without the event nobody could tell when they are
             * complete, so a failure here poisons the context instead of delivering stale records */
            const hipError_t e = hipEventRecord(carried->ev_records, c->stream);
            if (e != hipSuccess) {
                c->failed = true;
                return fail(c, -EIO, "hipEventRecord(ev_records) failed: %s", hipGetErrorString(e));
            }
        }
        if (s.lean) {
            /* no gather: the resolve workgroups read the region slices, the first resolve pass publishes */
            s.buf_first_valid = false;
            if (tail_here)
                s.tail_dst = nullptr; /* the scan's first wavefront copies it */
        } else {
            rc = msd_launch_gather(c->d_counts, c->d_wg_totals, nwg, s.d_totals, c->d_region_hits, c->d_region_tries, p.hcap, p.tcap,
                                   s.d_hits, s.dense_hits, s.d_tries, s.dense_tries, s.d_sums, s.nbuffers,
                                   gather_publishes ? s.h_totals : nullptr, gather_publishes ? s.h_sums : nullptr, nullptr, 0,
                                   tail_here ? s.d_iq + (s.nsamples - TAIL_SAMPLES) * bps_of(format) : nullptr,
                                   tail_here ? s.tail_dst : nullptr, tail_here ? (uint32_t)(TAIL_SAMPLES * bps_of(format)) : 0,
                                   tpw * tile, s.d_buf_first, 0, c->stream);
            s.buf_first_valid = true;
            if (tail_here)
                s.tail_dst = nullptr; /* done */
            if (rc)
                return fail(c, rc, "gather kernel launch failed");
        }
        if (!c->chain_inline && s.ev_scanned) /* the previous batch's chain is queued to start here, beside what follows */
            HIPCHK(c, hipEventRecord(s.ev_scanned, c->stream));
    } else if (s.timed) {
        HIPCHK(c, hipEventRecord(s.ev_scan, c->stream));
    }
    if (fm && s.nbuffers) {
        /* (the apply walk and the Mode A/C gather stay on the scan stream: deferred to the head of the batch's resolve chain
         * they were 2 % slower, LABLOG R4.5; the variant is scripts/experiments/r4_defer_tails.patch) */
        int rc = s.dc ? msd_launch_dc_sums(s.d_magsq, s.nsamples, MSD_CHUNK_SAMPLES, s.nbuffers, s.d_fmeans, s.d_fm_work, 0, c->stream)
                      : msd_launch_float_means(format, s.d_iq, s.nsamples, MSD_CHUNK_SAMPLES, s.nbuffers, s.d_fmeans,
                                               nwg && msd_scan_tile(format) == 1024 ? s.d_tile_sums : nullptr,
                                               s.d_fm_work, 0, c->stream);
        if (rc)
            return fail(c, rc, "float means kernel launch failed");
    }
    if (c->cfg.mode_ac && s.nbuffers) {
        if (host_noise)
            HIPCHK(c, hipMemcpyAsync(c->d_noise, host_noise, sizeof(uint32_t) * s.nbuffers, hipMemcpyHostToDevice,
                                     c->stream));
        MsdScanParams p{};
        fill_params(c, s, p);
        p.debug_flags = c->debug_flags;
        int ac_format = format;
        if (s.mag_pass) { /* the scan in front left the magnitudes: nothing is converted twice */
            ac_format = MSD_FMT_MAG16;
            p.iq = reinterpret_cast<const uint8_t *>(s.d_mag);
            p.prev_tail = reinterpret_cast<const uint8_t *>(s.d_mag_prev);
            p.have_prev = s.have_prev && s.d_mag_prev;
            p.ragged = reinterpret_cast<const uint8_t *>(s.d_mag + (s.nsamples & ~7ull)); /* zeros behind the last sample */
        }
        int rc = msd_launch_ac(&p, ac_format, s.d_sums, s.d_fmeans, s.nbuffers, c->d_noise,
                               host_noise != nullptr ? 1 : ((s.dc || (s.mag_pass && fm)) ? 2 : 0), /* (16-bit IQ: the float sums, whatever the pass reads) */
                               s.d_ac_regions, c->ac_arena, s.d_ac_counts, c->d_ac_offsets, s.d_ac_totals, s.d_ac,
                               c->ac_arena, c->ac_max_wg, 0, c->stream);
        if (rc)
            return fail(c, rc, "Mode A/C kernel launch failed");
    }
    if (s.timed)
        HIPCHK(c, hipEventRecord(s.ev_kernels, c->stream));

    /* totals and per-buffer sums go to pinned host memory from this stream, right behind the kernels */
    if (!gather_publishes && !s.lean) {
        int rc = msd_launch_publish(s.d_totals, c->cfg.mode_ac ? s.d_ac_totals : nullptr, s.d_sums,
                                    fm ? s.d_fmeans : nullptr, s.nbuffers, s.h_totals, s.h_ac_totals, s.h_sums,
                                    s.h_fmeans, c->stream);
        if (rc)
            return fail(c, rc, "publish kernel launch failed");
    }
    if (!s.lean || !c->chain_inline) /* in order, a lean batch's totals and sums come with its first resolve pass (ev_resolve)
                                        and nobody waits for this event; on side streams the chain does */
        HIPCHK(c, hipEventRecord(s.ev_totals, c->stream));
    return 0;
}

/* Stage 1 of finishing a batch: once its totals are known, start the download of its candidate
 * lists on the copy stream.  Idempotent; blocks only until the batch's kernels are done. */
int ensure_ac_host(msd_ctx *c, Slot &s, size_t nac)
{
    if (nac <= s.h_ac_cap)
        return 0;
    size_t cap = s.h_ac_cap ? s.h_ac_cap : (size_t)1 << 14;
    while (cap < nac)
        cap *= 2;
    if (s.h_ac)
        (void)hipHostFree(s.h_ac);
    s.h_ac = nullptr;
    s.h_ac_cap = 0;
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&s.h_ac), cap * sizeof(msd_ac_hit)));
    s.h_ac_cap = cap;
    return 0;
}

/* A batch whose candidate lists did not fit its arenas (an interference storm: a large share of
 * all positions look like preambles) is scanned again in pieces -- halves, quarters, ... down to
 * single buffers, which always fit -- and the pieces' lists are stitched together on the host.
 * Synchronous and slow on purpose; nothing is ever dropped. */
int rerun_in_pieces(msd_ctx *c, Slot &s, int format)
{
    {
        const int erc = ensure_dense(c, s); /* before the pieces copy the slot: they share its dense lists */
        if (erc)
            return erc;
    }
    const uint64_t total_buffers = s.nbuffers;
    for (uint64_t pieces = 2;; pieces *= 2) {
        uint64_t piece = ((s.nsamples + pieces - 1) / pieces + MSD_CHUNK_SAMPLES - 1) / MSD_CHUNK_SAMPLES *
                         MSD_CHUNK_SAMPLES;
        if (piece < MSD_CHUNK_SAMPLES)
            piece = MSD_CHUNK_SAMPLES;
        std::vector<msd_hit> hits;
        std::vector<msd_try> tries;
        std::vector<msd_ac_hit> acs;
        bool again = false;
        for (uint64_t off = 0; off < s.nsamples || (off == 0 && s.nsamples == 0); off += piece) {
            Slot t = s; /* shares the device buffers and events of s */
            t.gpu_resolve = false; /* the host resolver takes the pieces: no predictions wanted (they would land in the
                                      slot's table under a generation its next batch uses) */
            const uint64_t n = s.nsamples - off < piece ? s.nsamples - off : piece;
            const bool is_last = off + n >= s.nsamples;
            const uint64_t b0 = off / MSD_CHUNK_SAMPLES;
            t.d_iq = s.d_iq + off * bps_of(format);
            if (s.d_magsq)
                t.d_magsq = s.d_magsq + off;
            if (off) {
                t.d_prev = s.d_iq + (off - TAIL_SAMPLES) * bps_of(format);
                t.have_prev = 1;
            }
            t.batch_first = s.batch_first + off;
            t.nsamples = n;
            t.nbuffers = (uint32_t)((is_last ? total_buffers : (off + n) / MSD_CHUNK_SAMPLES) - b0);
            t.d_sums = s.d_sums + 2 * b0;
            t.d_fmeans = s.d_fmeans + 2 * b0;
            t.h_sums = s.h_sums + 2 * b0;
            t.h_fmeans = s.h_fmeans + 2 * b0;
            t.mag_pass = false; /* the pieces' Mode A/C passes convert the IQ themselves */
            t.d_mag = nullptr;
            /* mag_buf batches: the pieces keep the noise levels the first pass was given -- from the caller's per-buffer
             * mean_level / mean_power, which for the float converters are not what the integer sums of a MAG16 scan give */
            int rc = enqueue(c, t, format, c->magbuf_noise ? c->magbuf_noise + b0 : nullptr);
            if (rc)
                return rc;
            HIPCHK(c, hipEventSynchronize(t.ev_totals));
            const uint64_t H = s.h_totals[0], Tn = s.h_totals[1];
            const uint64_t nac = c->cfg.mode_ac ? s.h_ac_totals[0] : 0;
            if (s.h_totals[2] || (c->cfg.mode_ac && s.h_ac_totals[2])) {
                if (piece == MSD_CHUNK_SAMPLES)
                    return fail(c, -EOVERFLOW, "candidate arena overflow on a single buffer");
                again = true;
                break;
            }
            rc = ensure_host(c, s, H, Tn);
            if (!rc)
                rc = ensure_ac_host(c, s, nac);
            if (rc)
                return rc;
            if (H)
                HIPCHK(c, hipMemcpyAsync(s.h_hits, s.d_hits, H * sizeof(msd_hit), hipMemcpyDeviceToHost, c->copy_stream));
            if (Tn)
                HIPCHK(c, hipMemcpyAsync(s.h_tries, s.d_tries, Tn * sizeof(msd_try), hipMemcpyDeviceToHost, c->copy_stream));
            if (nac)
                HIPCHK(c, hipMemcpyAsync(s.h_ac, s.d_ac, nac * sizeof(msd_ac_hit), hipMemcpyDeviceToHost, c->copy_stream));
            HIPCHK(c, hipStreamSynchronize(c->copy_stream));
            const uint64_t try0 = tries.size();
            for (uint64_t i = 0; i < H; ++i) {
                msd_hit h = s.h_hits[i] + off; /* position is the low field: no carry into the mask */
                if (MSD_HIT_NLIVE(h))
                    h += (msd_hit)try0 << 34;
                hits.push_back(h);
            }
            for (uint64_t i = 0; i < Tn; ++i) {
                msd_try tr = s.h_tries[i];
                tr.pos += (uint32_t)off;
                tries.push_back(tr);
            }
            for (uint64_t i = 0; i < nac; ++i) {
                msd_ac_hit a = s.h_ac[i];
                a.pos += off;
                acs.push_back(a);
            }
            if (s.nsamples == 0)
                break;
        }
        if (again)
            continue;
        int rc = ensure_host(c, s, hits.size(), tries.size());
        if (!rc)
            rc = ensure_ac_host(c, s, acs.size());
        if (rc)
            return rc;
        if (!hits.empty())
            memcpy(s.h_hits, hits.data(), hits.size() * sizeof(msd_hit));
        if (!tries.empty())
            memcpy(s.h_tries, tries.data(), tries.size() * sizeof(msd_try));
        if (!acs.empty())
            memcpy(s.h_ac, acs.data(), acs.size() * sizeof(msd_ac_hit));
        s.h_totals[0] = hits.size();
        s.h_totals[1] = tries.size();
        s.h_totals[2] = 0;
        if (c->cfg.mode_ac) {
            s.h_ac_totals[0] = acs.size();
            s.h_ac_totals[2] = 0;
        }
        c->timing.reruns++;
        return 0;
    }
}

/* Stage 1 of finishing a batch: once its totals are known, start the download of its candidate
 * lists on the copy stream.  Idempotent; blocks only until the batch's kernels are done. */
int start_download(msd_ctx *c, Slot &s, int format)
{
    if (s.download_started)
        return 0;
    if (s.lean) { /* totals, sums and the overflow flag arrive with the first resolve pass (finish_gpu) */
        HIPCHK(c, hipEventRecord(s.ev_copy0, c->copy_stream));
        HIPCHK(c, hipEventRecord(s.ev_copy1, c->copy_stream));
        s.download_started = true;
        return 0;
    }
    const bool trace = c->trace;
    auto td0 = std::chrono::steady_clock::now();
    HIPCHK(c, event_wait(s.ev_totals));
    if (trace && s.timed) {
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, s.ev_start, s.ev_scan);
        (void)hipEventElapsedTime(&b, s.ev_start, s.ev_kernels);
        fprintf(stderr, "start_download: waited %.3f ms for the totals; scan %.3f ms, all kernels %.3f ms after its start\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td0).count(), a, b);
    }
    const bool overflow = s.h_totals[2] || (c->cfg.mode_ac && s.h_ac_totals[2]);
    if (overflow) {
        int rc = rerun_in_pieces(c, s, format);
        if (rc)
            return rc;
    }
    const uint64_t H = s.h_totals[0], Tn = s.h_totals[1];
    HIPCHK(c, hipEventRecord(s.ev_copy0, c->copy_stream));
    if (overflow) { /* rescanned in pieces and stitched on the host: the host resolver takes it */
        s.gpu_resolve = false;
        s.resolve_inflight = false;
    }
    if (!overflow && !s.gpu_resolve) {
        int rc = ensure_host(c, s, H, Tn);
        if (rc)
            return rc;
        if (H)
            HIPCHK(c, hipMemcpyAsync(s.h_hits, s.d_hits, H * sizeof(msd_hit), hipMemcpyDeviceToHost, c->copy_stream));
        if (Tn)
            HIPCHK(c, hipMemcpyAsync(s.h_tries, s.d_tries, Tn * sizeof(msd_try), hipMemcpyDeviceToHost, c->copy_stream));
        if (c->cfg.mode_ac) {
            const uint64_t nac = s.h_ac_totals[0];
            rc = ensure_ac_host(c, s, nac);
            if (rc)
                return rc;
            if (nac)
                HIPCHK(c, hipMemcpyAsync(s.h_ac, s.d_ac, nac * sizeof(msd_ac_hit), hipMemcpyDeviceToHost, c->copy_stream));
        }
    }
    HIPCHK(c, hipEventRecord(s.ev_copy1, c->copy_stream));
    s.download_started = true;
    return 0;
}

constexpr uint32_t SNAP_CAP = 64; /* filter membership versions of one batch kept on the device */

/* The resolve stage with the candidate lists left in HBM: one workgroup per buffer against a
 * snapshot of the ICAO filter; the host only replays the buffers' add lists to find the snapshot
 * every buffer has to see (msd_resolve.c) and re-launches the ones that saw another.
 *
 * Queueing: the first pass over a batch, the message records and their signal power are queued on
 * the scan stream as soon as the previous batch is committed (often long before anybody waits for
 * them), so they never share compute units with a scan kernel; everything they report lands in
 * pinned host memory, the host waits for one event.  Only the rare further passes use the
 * high-priority aux stream. */
struct GpuCtl {
    uint64_t *h_ts;
    uint32_t *h_valid, *h_snap, *h_todo;
};

GpuCtl gpu_ctl(const msd_ctx *c, const Slot &s)
{
    const size_t N = c->max_buffers;
    GpuCtl g;
    g.h_ts = reinterpret_cast<uint64_t *>(s.h_ctl);
    g.h_valid = reinterpret_cast<uint32_t *>(s.h_ctl + 16 * N);
    g.h_snap = g.h_valid + N;
    g.h_todo = g.h_snap + N;
    return g;
}

void gpu_params(const msd_ctx *c, const Slot &s, MsdResolveParams &rp)
{
    const GpuCtl g = gpu_ctl(c, s);
    rp.hits = s.d_hits;
    rp.tries = s.d_tries;
    rp.totals = s.d_totals;
    rp.buf_first = s.buf_first_valid ? s.d_buf_first : nullptr;
    /* the control arrays are read where they are, in pinned host memory: a few words per workgroup,
     * and an upload of 14 KiB would run as a blit kernel that fights the scan for compute units */
    rp.valid = g.h_valid;
    rp.ts = g.h_ts;
    rp.snaps = c->d_snaps;
    rp.snap_idx = g.h_snap;
    rp.todo = g.h_todo;
    rp.rbuf = s.h_rbuf;
    rp.nmsgs = s.d_nmsgs;
    rp.acc = s.d_acc;
    rp.adds = s.d_adds;
    if (c->cfg.mode_ac) {
        rp.ac = s.d_ac;
        rp.ac_totals = s.d_ac_totals;
        rp.acc_ac = s.d_acc_ac;
        rp.nac = s.d_nac;
    }
    rp.pred = reinterpret_cast<const unsigned long long *>(s.d_pred);
    rp.pred_gen = s.pred_gen;
    if (c->power_fused) { /* the signal power of the accepted messages at the end of every resolve workgroup */
        MsdScanParams sp{};
        fill_params(c, s, sp);
        rp.power = reinterpret_cast<unsigned long long *>(s.d_powr);
        rp.iq = sp.iq;
        rp.prev_tail = sp.prev_tail;
        rp.have_prev = sp.have_prev;
        rp.batch_first = sp.batch_first;
        rp.nsamples = sp.nsamples;
        rp.lut = sp.lut;
        rp.format = c->scan_format;
    }
    if (s.lean) {
        rp.hits = s.d_rhits;
        rp.tries = s.d_rtries;
        rp.buf_first = nullptr;
        rp.region_counts = s.d_rcounts;
        rp.wg_totals = s.d_rwgt;
        rp.regions_per_buffer = s.lean_k;
        rp.hcap = s.lean_hcap;
        rp.nscan_wg = (s.lean_nreg + MSD_SCAN_WAVES - 1) / MSD_SCAN_WAVES;
        rp.nregions = s.lean_nreg;
        rp.sums = s.d_sums;
        rp.h_sums = s.h_sums;
        rp.h_totals = s.h_totals;
        if (c->scan_format == MSD_FMT_SC16 || c->scan_format == MSD_FMT_SC16Q11) {
            rp.fmeans = s.d_fmeans;
            rp.h_fmeans = s.h_fmeans;
        }
        rp.h_ac_totals = s.h_ac_totals;
    }
}

uint32_t slot_valid(const Slot &s, uint32_t b)
{
    const uint64_t first = (uint64_t)b * MSD_CHUNK_SAMPLES;
    uint64_t n = s.nsamples > first ? s.nsamples - first : 0;
    return (uint32_t)(n > MSD_CHUNK_SAMPLES ? MSD_CHUNK_SAMPLES : n);
}

/* one resolve pass over the s.resolve_ntodo buffers of the to-do list, on `ks`.  Its inputs (new
 * filter snapshots, control arrays) go up on the aux stream right away -- `ks` is usually still busy
 * with a scan -- and the kernel waits for them through an event. */
int gpu_queue_pass(msd_ctx *c, Slot &s, hipStream_t ks, bool first_pass)
{
    const uint32_t nsn = msd_gpu_resolve_nsnaps(&c->resolver);
    for (uint32_t i = c->snaps_uploaded; i < nsn; ++i) {
        uint32_t *stage = c->h_snaps + (size_t)i * MSD_SNAP_WORDS;
        uint32_t active = 0;
        const uint32_t *two = msd_gpu_resolve_snapshot(&c->resolver, i, &active); /* slot[2][8192] */
        for (uint32_t h = 0; h < 8192; ++h) { /* interleaved on the device: a probe's two first slots are one load */
            stage[2 * h] = two[h];
            stage[2 * h + 1] = two[8192 + h];
        }
        stage[16384] = active;
        HIPCHK(c, hipMemcpyAsync(c->d_snaps + (size_t)i * MSD_SNAP_WORDS, stage, sizeof(uint32_t) * MSD_SNAP_WORDS,
                                 hipMemcpyHostToDevice, c->aux_stream));
    }
    c->snaps_uploaded = nsn;
    if (ks != c->aux_stream) {
        HIPCHK(c, hipEventRecord(c->ev_inputs, c->aux_stream));
        if (ks == c->stream && c->chain_inline && !c->wait_inputs_on_stream) {
            /* In order on the scan stream: the caller waits the few microseconds the 64 KB take (the copy engine is
             * idle) instead of the stream -- a wait packet in front of the resolve kernel holds the stream for 10 us
             * behind every scan, however long ago the event fired. */
            HIPCHK(c, event_wait(c->ev_inputs));
        } else {
            HIPCHK(c, hipStreamWaitEvent(ks, c->ev_inputs, 0));
        }
    }
    MsdResolveParams rp{};
    gpu_params(c, s, rp);
    int rc = 0;
    rp.first_pass = first_pass ? 1 : 0;
    rp.ctl_implicit = 1;
    rp.sample_counter0 = s.sample_counter0;
    rp.batch_samples = s.nsamples;
    rp.h_pred = c->h_pred;
    rp.h_pred_count = c->h_pred_count;
    if (!first_pass) /* (the first pass finds the table as the batch's scan kernel left it) */
        rc = msd_launch_pred_patch(reinterpret_cast<unsigned long long *>(s.d_pred), c->h_patches, c->npatches, ks);
    if (rc)
        return fail(c, rc, "prediction patch kernel launch failed");
    rc = msd_launch_resolve(&rp, s.resolve_ntodo, ks);
    if (rc)
        return fail(c, rc, "resolve kernel launch failed");
    return 0;
}

/* message records and signal power of every buffer on `ks` (device memory); ev_records marks the
 * end.  The host fetches them with one DMA once it knows how many there are (fetch_records). */
int gpu_queue_emit(msd_ctx *c, Slot &s, int format, hipStream_t ps, hipStream_t ks, bool do_power = true,
                   bool do_emit = true)
{
    MsdResolveParams rp{};
    gpu_params(c, s, rp);
    MsdScanParams p{};
    fill_params(c, s, p);
    /* the signal power on `ps`, the records on `ks` behind it */
    if (c->power_fused && do_power) { /* the resolve workgroups left the sums; the emit kernel adds up its own offsets */
        do_power = false;
        s.power_done = false;
    }
    int rc = do_power ? msd_launch_power_buffers(&p, format, s.d_acc, s.d_tries, s.d_nmsgs, s.nbuffers, s.d_totals,
                                                 reinterpret_cast<unsigned long long *>(s.d_powr),
                                                 c->cfg.mode_ac ? s.d_nac : nullptr, s.d_rec_off, ps)
                      : 0;
    if (rc)
        return fail(c, rc, "power kernel launch failed");
    if (do_power)
        s.power_done = true;
    if (!do_emit)
        return 0;
    if (ps != ks) {
        HIPCHK(c, hipEventRecord(s.ev_power, ps));
        HIPCHK(c, hipStreamWaitEvent(ks, s.ev_power, 0));
    }
    rc = msd_launch_emit(&rp, s.nbuffers, reinterpret_cast<const unsigned long long *>(s.d_powr), s.h_side,
                         c->records_dma ? s.d_wire : s.h_wire,
                         c->want_fields ? (c->records_dma ? s.d_fields : s.h_fields) : nullptr,
                         (uint32_t)s.req_cap, ks);
    if (rc)
        return fail(c, rc, "emit kernel launch failed");
    HIPCHK(c, hipEventRecord(s.ev_records, ks));
    return 0;
}

/* The records in pinned memory.  Default: the emit kernel wrote them there itself (PCIe-bound, ~45 us
 * per 35 000 messages, in order behind the batch's other kernels, nothing else shares the GPU with a
 * scan).  MSD_RECORDS_DMA=1: the emit kernel wrote to HBM and one hipMemcpyAsync fetches the records
 * while the next scan runs -- faster when the runtime gives the copy a DMA engine (it does when the
 * copy is issued on an idle stream with no event to wait for), but under rocprofv3, and whenever the
 * runtime picks a blit kernel instead, that copy takes compute units from the scan. */
int fetch_records(msd_ctx *c, Slot &s, uint32_t total)
{
    HIPCHK(c, event_wait(s.ev_records));
    if (total && c->records_dma) {
        HIPCHK(c, hipMemcpyAsync(s.h_wire, s.d_wire, (size_t)total * sizeof(msd_wire), hipMemcpyDeviceToHost,
                                 c->copy_stream));
        if (c->want_fields)
            HIPCHK(c, hipMemcpyAsync(s.h_fields, s.d_fields, (size_t)total * sizeof(msd_fields), hipMemcpyDeviceToHost,
                                     c->copy_stream));
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    }
    return 0;
}

/* The records of the batch whose chain was queued last and whose records are still owed, by the stand-alone
 * kernel on the scan stream (no scan came along to carry them). */
int flush_pending_emit(msd_ctx *c)
{
    Slot *p = c->pending_emit;
    if (!p)
        return 0;
    c->pending_emit = nullptr;
    return gpu_queue_emit(c, *p, c->scan_format, c->stream, c->stream, !p->power_done, true);
}

/* Clocks, snapshot 0 = the live filter, first pass over every buffer and the (speculative) message
 * records, all behind the batch's own kernels on the scan stream.  Every earlier batch must have
 * been committed: this is the earliest moment its successor can start. */
/* the samples a live receiver dropped in front of this batch go onto the sample clock when the batch's
 * turn comes (every earlier batch has been committed by then), sdr_rtlsdr.c:284,299 */
void apply_dropped(msd_ctx *c, Slot &s)
{
    c->resolver.sample_counter += s.dropped_before;
    c->stats.samples_dropped += s.dropped_before; /* readsb.c:836 */
    s.dropped_before = 0;
}

int gpu_begin(msd_ctx *c, Slot &s, int format)
{
    apply_dropped(c, s);
    s.power_done = false;
    const GpuCtl g = gpu_ctl(c, s);
    for (uint32_t b = 0; b < s.nbuffers; ++b)
        g.h_valid[b] = slot_valid(s, b);
    s.sample_counter0 = c->resolver.sample_counter;
    msd_gpu_resolve_begin(&c->resolver, s.nbuffers, g.h_valid, g.h_ts, g.h_snap, g.h_todo, &s.resolve_ntodo);
    c->snaps_uploaded = 0;
    /* The scan stream carries scans (and their gathers) only, back to back.  Prediction + resolve run on
     * the high-priority chain stream behind the batch's own scan (ev_totals), power + records on a third
     * one behind the resolve: they share the GPU with the next batch's scan instead of delaying it. */
    hipStream_t ks = c->chain_inline ? c->stream : c->aux_stream;
    hipStream_t es = c->chain_inline ? c->stream : c->emit_stream;
    hipStream_t pws = es; /* the signal power kernel */
    int rc = ensure_req(c, s, (size_t)s.nbuffers * 96 + 4096);
    if (rc)
        return rc;
    if (ks != c->stream) {
        HIPCHK(c, hipStreamWaitEvent(ks, s.ev_totals, 0));
        /* Side streams: a scan workgroup fills its compute unit (all registers, all LDS), so a chain kernel that
         * meets a scan waits for it; the latency-bound kernels behind the scan (float sums, Mode A/C) leave
         * room.  If the next batch is queued already, the chain starts when that batch's scan has retired. */
        Slot &nx = c->slots[((&s - c->slots) + 1) % MSD_PIPELINE_DEPTH];
        if (&nx != &s && nx.busy && nx.launch_seq == s.launch_seq + 1 && nx.ev_scanned && nx.nsamples >= MSD_CHUNK_SAMPLES)
            HIPCHK(c, hipStreamWaitEvent(ks, nx.ev_scanned, 0));
    }
    rc = gpu_queue_pass(c, s, ks, true);
    if (!rc) {
        HIPCHK(c, hipEventRecord(s.ev_resolve, ks));
        if (c->emit_fused) {
            rc = flush_pending_emit(c); /* an older one no scan came after */
            if (!rc && !c->power_fused)
                rc = gpu_queue_emit(c, s, format, ks, ks, true, false); /* signal power now, records with the next scan */
            if (!rc)
                c->pending_emit = &s; /* (power_fused: the next scan's wavefronts sum the signal power as well) */
        } else {
            if (pws != ks)
                HIPCHK(c, hipStreamWaitEvent(pws, s.ev_resolve, 0));
            rc = gpu_queue_emit(c, s, format, pws, es);
        }
    }
    if (rc)
        return rc;
    s.resolve_inflight = true;
    return 0;
}

bool gpu_eligible(const msd_ctx *c, const Slot &s)
{
    return c->gpu_resolve && s.nbuffers >= 4;
}

/* per-buffer sample counts and means (mag_buf.validLength - overlap, .mean_level, .mean_power) of an integer format */
void means_from_sums(msd_ctx *c, const Slot &s)
{
    c->valid.assign(s.nbuffers, 0);
    c->means.assign(2 * (size_t)s.nbuffers, 0.0);
    const bool fm = c->scan_format == MSD_FMT_SC16 || c->scan_format == MSD_FMT_SC16Q11;
    for (uint32_t b = 0; b < s.nbuffers; ++b) {
        const uint32_t n = slot_valid(s, b);
        c->valid[b] = n;
        if (fm) { /* convert.c:245-251: float sum / unsigned -> float division, widened to double */
            c->means[2 * b] = (double)(s.h_fmeans[2 * b] / (float)n);
            c->means[2 * b + 1] = (double)(s.h_fmeans[2 * b + 1] / (float)n);
            continue;
        }
        /* convert.c:104-110 (note 65536 for the level, 65535^2 for the power) */
        c->means[2 * b] = (double)s.h_sums[2 * b] / 65536.0 / (double)n;
        c->means[2 * b + 1] = (double)s.h_sums[2 * b + 1] / 65535.0 / 65535.0 / (double)n;
    }
}

/* Lean layout: the dense, ordered candidate lists after all (somebody on the host wants them): the gather kernel
 * over the slot's region slices, synchronously.  Totals land in h_totals; the sums were published already. */
int lean_gather_now(msd_ctx *c, Slot &s)
{
    /* the first resolve pass has published the batch's totals: a slot whose region slices were enlarged may hold more
     * than the dense lists were made for */
    const uint64_t H = s.h_totals[0] ? s.h_totals[0] : 1, Tn = s.h_totals[1] ? s.h_totals[1] : 1; /* (never a null list) */
    if (H > s.dense_hits || Tn > s.dense_tries) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (H > s.dense_hits) {
            (void)hipFree(s.d_hits);
            s.d_hits = nullptr;
            s.dense_hits = 0;
            HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&s.d_hits), (H + H / 8) * sizeof(msd_hit)));
            s.dense_hits = H + H / 8;
        }
        if (Tn > s.dense_tries) {
            (void)hipFree(s.d_tries);
            s.d_tries = nullptr;
            s.dense_tries = 0;
            HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&s.d_tries), (Tn + Tn / 8) * sizeof(msd_try)));
            s.dense_tries = Tn + Tn / 8;
        }
    }
    int rc = msd_launch_gather(s.d_rcounts, s.d_rwgt, s.lean_nreg, s.d_totals, s.d_rhits, s.d_rtries, s.lean_hcap, s.lean_tcap,
                               s.d_hits, s.dense_hits, s.d_tries, s.dense_tries, s.d_sums, s.nbuffers, s.h_totals, nullptr,
                               nullptr, 0, nullptr, nullptr, 0, 0, nullptr, 1, c->stream);
    if (rc)
        return fail(c, rc, "gather kernel launch failed");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

/* A lean batch overflowed the region slices of its slot (an interference storm, a pulse train: a large share of all
 * positions look like preambles).  The scan counts on past the end of a slice, so the region counts say exactly what
 * every region needed: the slot gets slices the densest region fits (plus an eighth), the batch is scanned again into
 * them and stays on the GPU resolve -- one extra scan instead of the whole batch in pieces through the host resolver
 * (rerun_in_pieces: 0.09-0.7 GS/s on such input).  The slot keeps the bigger slices; the other slots follow at their
 * next launch (enqueue: c->want_*).  0: rescanned, the first resolve pass has to be begun again; 1: not possible (no
 * device memory for it, or it was the Mode A/C arena): the old way; < 0: error. */
int grow_and_rescan(msd_ctx *c, Slot &s, int format)
{
    if (!s.lean || !s.lean_nreg || !s.d_rcounts || (c->cfg.flags & MSD_CFG_NO_ARENA_GROWTH))
        return 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<msd_region_counts> rc(s.lean_nreg);
    HIPCHK(c, hipMemcpy(rc.data(), s.d_rcounts, rc.size() * sizeof(msd_region_counts), hipMemcpyDeviceToHost));
    uint64_t mh = 0, mt = 0;
    for (const msd_region_counts &r : rc) {
        mh = r.nhits > mh ? r.nhits : mh;
        mt = r.ntries > mt ? r.ntries : mt;
    }
    if (mh <= s.lean_hcap && mt <= s.lean_tcap)
        return 1; /* the Mode S slices held: it was the Mode A/C arena */
    uint64_t hcap = mh + mh / 8 + 64, tcap = mt + mt / 8 + 64;
    hcap = hcap > s.lean_hcap ? hcap : s.lean_hcap;
    tcap = tcap > s.lean_tcap ? tcap : s.lean_tcap;
    const uint64_t need_h = hcap * s.lean_nreg, need_t = tcap * s.lean_nreg;
    if (need_t >= (1ull << 30) || hcap >= (1ull << 32) || tcap >= (1ull << 32))
        return 1; /* try indices are 30 bits of the hit record */
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
        return 1;
    const uint64_t grow_b = (need_h > s.rhit_arena ? (need_h - s.rhit_arena) * sizeof(msd_hit) : 0) +
                            (need_t > s.rtry_arena ? (need_t - s.rtry_arena) * sizeof(msd_try) : 0);
    if (grow_b + (1ull << 30) > free_b)
        return 1;
    const bool grew_h = need_h > s.rhit_arena, grew_t = need_t > s.rtry_arena;
    if (need_h > s.rhit_arena) {
        msd_hit *nh = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&nh), need_h * sizeof(msd_hit)) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        (void)hipFree(s.d_rhits);
        s.d_rhits = nh;
        s.rhit_arena = need_h;
    }
    if (need_t > s.rtry_arena) {
        msd_try *nt = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&nt), need_t * sizeof(msd_try)) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        (void)hipFree(s.d_rtries);
        s.d_rtries = nt;
        s.rtry_arena = need_t;
    }
    /* what the other slots should have before they meet the same traffic, per sample of a batch: from what THIS batch
     * needed (need_h / need_t entries for its nsamples -- not the slot's whole arena, which is sized for max_batch_samples
     * and would turn a short batch's overflow into several hits per sample for everybody), only for the arena that grew,
     * and never more than the kernels can produce -- one hit per position, five tries per hit -- plus the eighth of
     * head-room the slices are given above */
    const double ns = (double)(s.nsamples ? s.nsamples : 1);
    if (grew_h) {
        const double per_h = std::min(1.25, (double)need_h / ns);
        c->want_hits_per_sample = per_h > c->want_hits_per_sample ? per_h : c->want_hits_per_sample;
    }
    if (grew_t) {
        const double per_t = std::min(6.25, (double)need_t / ns);
        c->want_tries_per_sample = per_t > c->want_tries_per_sample ? per_t : c->want_tries_per_sample;
    }
    HIPCHK(c, hipMemsetAsync(s.d_totals, 0, 4 * sizeof(uint64_t), c->stream)); /* the overflow flag the first scan raised */
    s.h_totals[2] = 0;
    if (c->pending_emit == &s)
        c->pending_emit = nullptr;
    const int erc = enqueue(c, s, format, nullptr, true);
    if (erc)
        return erc < 0 ? erc : -EIO;
    if (!s.lean)
        return 1;
    c->timing.reruns++;
    return 0;
}

/* Returns 1 when the batch has to go through the host resolver instead (nothing committed); 2 when its candidate
 * arenas overflowed (lean layout: only the first resolve pass tells). */
/* The resolve passes of a batch whose first pass is queued: wait, replay the filter changes on the host, run the
 * buffers that saw the wrong filter again, until the replay agrees with what every buffer assumed.  0: done (nothing
 * is committed yet); 1: the host resolver has to take the batch; 2: its candidate arenas overflowed (lean layout:
 * only the first pass tells); < 0: error. */
int resolve_passes(msd_ctx *c, Slot &s, double &t_wait, double &t_replay)
{
    const uint32_t n = s.nbuffers;
    const GpuCtl g = gpu_ctl(c, s);
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    hipEvent_t wait_for = s.ev_resolve;
    s.records_current = true; /* the message records in host memory belong to the latest pass */
    s.npass = 0;
    for (uint32_t pass = 0;; ++pass) {
        auto k0 = tnow();
        ++s.npass;
        HIPCHK(c, event_wait(wait_for));
        auto k1 = tnow();
        if (s.lean && pass == 0 && (s.h_totals[2] || (c->cfg.mode_ac && s.h_ac_totals[2]))) /* what the gather
                                                                                                   kernels' totals used to say */
            return 2;
        int rc = msd_gpu_resolve_replay(&c->resolver, n, s.h_rbuf, nullptr, c->inline_adds, pass, SNAP_CAP, c->h_pred,
                                        *c->h_pred_count, c->h_patches, &c->npatches, g.h_snap, g.h_todo,
                                        &s.resolve_ntodo);
        if (rc == -2) { /* a flip, or very many new addresses in one buffer: the complete add lists are needed.
                           They are in pinned host memory already (the resolve kernel writes a buffer's ~60 addresses
                           there itself; fetching the [buffer][1024] array cost a 2 MB copy per flip) */
            c->timing.resolve_long_lists++;
            rc = msd_gpu_resolve_replay(&c->resolver, n, s.h_rbuf, s.d_adds, c->inline_adds, pass, SNAP_CAP, c->h_pred,
                                        *c->h_pred_count, c->h_patches, &c->npatches, g.h_snap, g.h_todo,
                                        &s.resolve_ntodo);
        }
        t_wait += tms(k0, k1);
        t_replay += tms(k1, tnow());
        if (rc == 0)
            return 0;
        if (rc < 0)
            return 1;
        /* some buffers saw the wrong filter: once more for those, ahead of the queued scans */
        hipStream_t ps = (c->repass_aux || !c->chain_inline) ? c->aux_stream : c->stream;
        rc = gpu_queue_pass(c, s, ps, false);
        if (rc)
            return rc;
        HIPCHK(c, hipEventRecord(c->ev_aux, ps));
        wait_for = c->ev_aux;
        s.records_current = false;
    }
}

/* The successor of a batch whose filter changes have just been committed: its first resolve pass can be queued. */
int begin_successor(msd_ctx *c, Slot &s)
{
    Slot &nx = c->slots[((&s - c->slots) + 1) % MSD_PIPELINE_DEPTH];
    /* Across a capture boundary too: the filter and the clocks start over now (this batch was the old capture's
     * last one), the counters when the new capture's first batch is collected -- the caller may still want
     * the old ones.  (Not if samples were dropped in front of the new capture: they count on its counters.) */
    if (&nx != &s && nx.busy && nx.launch_seq == s.launch_seq + 1 && nx.gpu_resolve && !nx.resolve_inflight && !nx.ahead_done &&
        (!nx.reset_before || nx.dropped_before == 0)) {
        if (nx.reset_before) {
            msd_resolver_reset_state(&c->resolver);
            nx.state_reset_done = true;
        }
        return gpu_begin(c, nx, c->scan_format);
    }
    return 0;
}

/* Returns 1 when the batch has to go through the host resolver instead (nothing committed); 2 when its candidate
 * arenas overflowed (lean layout: only the first resolve pass tells). */
int finish_gpu(msd_ctx *c, Slot &s, int format, msd_message_fn sink, void *user)
{
    const uint32_t n = s.nbuffers;
    const GpuCtl g = gpu_ctl(c, s);
    const bool trace = c->trace;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    double t_wait = 0, t_replay = 0;
    const bool early = s.resolve_inflight || s.ahead_done;
    int begin_rc = 0;
    if (s.ahead_verdict) { /* the msd_collect before this one has been through the passes already */
        const int v = s.ahead_verdict;
        s.ahead_verdict = 0;
        s.resolve_inflight = false;
        if (c->pending_emit == &s)
            c->pending_emit = nullptr; /* its speculative records are void */
        return v;
    }
    if (!s.ahead_done) { /* (an earlier msd_collect may have done this half already: see below) */
        if (!s.resolve_inflight) {
            int rc = gpu_begin(c, s, format);
            if (rc)
                return rc;
        }
        s.resolve_inflight = false;
        if (c->pending_emit == &s) { /* no scan was launched since */
            int rc = flush_pending_emit(c);
            if (rc)
                return rc;
        }
        int rc = resolve_passes(c, s, t_wait, t_replay);
        if (rc)
            return rc;
        msd_gpu_resolve_commit_state(&c->resolver, n, g.h_valid, s.h_rbuf);
        /* the filter is final for this batch: its successor can start */
        if (c->outstanding > 1)
            begin_rc = begin_successor(c, s);
    }
    s.ahead_done = false;
    s.resolve_inflight = false;
    if (c->pending_emit == &s) { /* no scan was launched since its chain was queued: nobody carries its records */
        int rc = flush_pending_emit(c);
        if (rc)
            return rc;
    }
    const uint32_t npass = s.npass;
    bool records_current = s.records_current;
    hipEvent_t wait_for = c->ev_aux; /* (only looked at after a further pass, which recorded it) */
    c->timing.resolve_passes = npass;
    auto e0 = tnow();
    if (s.lean)
        means_from_sums(c, s);
    msd_gpu_resolve_commit_stats(&c->resolver, n, g.h_valid, s.h_rbuf);

    uint32_t total = 0;
    c->out_buf.clear();
    for (uint32_t b = 0; b < n; ++b) {
        const uint32_t k = s.h_rbuf[b].nmsgs + (c->cfg.mode_ac ? s.h_rbuf[b].nac : 0u);
        total += k;
        c->out_buf.insert(c->out_buf.end(), k, b);
    }
    if (total > s.req_cap) { /* more messages than the arrays of the speculative records hold */
        HIPCHK(c, hipEventSynchronize(s.ev_records));
        int rc = ensure_req(c, s, total);
        if (rc)
            return rc;
        records_current = false;
    }
    hipStream_t rs = c->chain_inline ? (c->repass_aux ? c->aux_stream : c->stream) : c->emit_stream;
    if (!records_current && total) {
        if (!c->chain_inline) /* behind the last pass (or, if only the arrays grew, behind nothing new) */
            HIPCHK(c, hipStreamWaitEvent(rs, wait_for, 0));
        int rc = gpu_queue_emit(c, s, format, rs, rs);
        if (rc)
            return rc;
    }
    /* The per-message half -- wait for the records, signal level and the order-sensitive power
     * statistics (demod_2400.c:386-408,422-427), the copy into the caller's arrays -- goes to the helper
     * thread; it touches this batch's records and the power fields of the statistics only. */
    int fetch_rc = 0;
    double t_power = 0;
    static_assert(sizeof(msd_wire) == sizeof(msd_message), "the records are msd_message arrays");
    auto deliver = [&, total, n, cc = c]() {
        fetch_rc = fetch_records(c, s, total);
        if (fetch_rc)
            return;
        auto p0 = tnow();
        /* what the statistics half below needs, in the context's own storage: the caller's next batch reuses
         * c->valid / c->means / c->out_buf and this batch's slot while it runs */
        cc->bg_valid = cc->valid;
        cc->bg_means = cc->means;
        cc->bg_buf.swap(cc->out_buf);
        cc->bg_scaled.resize(total ? total : 1);
        memcpy(cc->bg_scaled.data(), s.h_side, (size_t)total * sizeof(uint64_t));
        t_power = tms(p0, tnow());
        /* the library's own array sinks take the whole batch with one copy instead of 35 000 calls */
        if (c->fsink == msd_array_fields_sink) {
            msd_array_fields_sink_state *st = static_cast<msd_array_fields_sink_state *>(c->fuser);
            const size_t room = st->count < st->cap ? st->cap - st->count : 0, k = total < room ? total : room;
            memcpy(st->out + st->count, s.h_wire, k * sizeof(msd_message));
            memcpy(st->fields + st->count, s.h_fields, k * sizeof(msd_fields));
            st->count += total;
        } else if (!c->fsink && sink == msd_array_sink) {
            msd_array_sink_state *st = static_cast<msd_array_sink_state *>(user);
            const size_t room = st->count < st->cap ? st->cap - st->count : 0, k = total < room ? total : room;
            memcpy(st->out + st->count, s.h_wire, k * sizeof(msd_message));
            st->count += total;
        }
        /* ---- the caller has its messages; from here on nothing of finish_gpu's frame or of the slot is touched ---- */
        msd_ctx *const ctx = cc;
        const uint32_t nb = n;
        const uint64_t nm = total;
        ctx->helper.mark_delivered();
        msd_resolve_power_stats(&ctx->resolver, nb, ctx->bg_valid.data(), ctx->bg_means.data(), ctx->bg_buf.data(),
                                ctx->bg_scaled.data(), nm);
    };
    const bool threaded = !c->no_helper;
    if (threaded)
        c->helper.run(deliver);
    /* While this batch's records are on their way: the resolve half of the NEXT batch -- wait for its first pass
     * (queued when this batch's filter changes were committed, possibly by the msd_collect before this one), replay,
     * commit its filter changes and queue the first pass of the batch behind it.  The chain of resolve passes then
     * runs one batch ahead of the delivery: the caller, who can only launch the next scan once this call returns,
     * never finds the GPU waiting for a resolve pass it has not been able to queue yet.  The counters of the next
     * batch are added when it is collected, as before. */
    if (c->outstanding > 1 && !begin_rc && c->resolve_ahead) {
        Slot &nx = c->slots[((&s - c->slots) + 1) % MSD_PIPELINE_DEPTH];
        if (&nx != &s && nx.busy && nx.launch_seq == s.launch_seq + 1 && nx.gpu_resolve && nx.resolve_inflight && !nx.ahead_done &&
            (!nx.reset_before || nx.state_reset_done)) { /* (a new capture's first batch: only once filter and clocks have started over) */
            double tw = 0, tr = 0;
            const int arc = resolve_passes(c, nx, tw, tr);
            if (trace)
                fprintf(stderr, "ahead: next batch's passes: waits %.3f ms, replay %.3f ms, verdict %d\n", tw, tr, arc);
            if (arc < 0) {
                begin_rc = arc;
            } else if (arc == 0) {
                const GpuCtl gn = gpu_ctl(c, nx);
                const auto a0 = tnow();
                msd_gpu_resolve_commit_state(&c->resolver, nx.nbuffers, gn.h_valid, nx.h_rbuf);
                nx.ahead_done = true;
                nx.resolve_inflight = false;
                const auto a1 = tnow();
                if (c->outstanding > 2)
                    begin_rc = begin_successor(c, nx);
                if (trace)
                    fprintf(stderr, "ahead: commit %.3f ms, the batch behind it begun in %.3f ms\n", tms(a0, a1), tms(a1, tnow()));
            } else { /* the host resolver's case or an overflow: nothing is committed, that batch's msd_collect acts on it */
                nx.ahead_verdict = arc;
            }
        }
    }
    auto e1 = tnow();
    if (threaded)
        c->helper.wait_delivered();
    else
        deliver();
    if (begin_rc)
        return begin_rc;
    if (fetch_rc)
        return fetch_rc;
    if (trace) {
        double cyc[8] = {0};
        for (uint32_t b = 0; b < n; ++b)
            for (int k = 0; k < 8; ++k)
                cyc[k] += s.h_rbuf[b].cyc[k];
        if (cyc[0] > 0) /* built with -DMSD_RESOLVE_TIMING=1 */
            fprintf(stderr, "resolve kernel, mean us per buffer: setup %.1f segment %.1f stage %.1f eval %.1f walk %.1f count %.1f no-try hits %.1f power %.1f\n",
                    cyc[0] / n / 100, cyc[5] / n / 100, cyc[1] / n / 100, cyc[2] / n / 100, cyc[3] / n / 100, cyc[4] / n / 100, cyc[7] / n / 100,
                    cyc[6] / n / 100);
        fprintf(stderr, "gpu resolve: %u passes%s, waits %.3f ms, replay %.3f ms, commit + next batch's first pass %.3f ms, "
                "power stats %.3f ms (helper), then waited %.3f ms for it\n", npass, early ? " (first one queued early)" : "",
                t_wait, t_replay, tms(e0, e1), t_power, tms(e1, tnow()));
    }
    /* callback sinks run on the calling thread, in order */
    if (c->fsink && c->fsink != msd_array_fields_sink) {
        for (uint32_t i = 0; i < total; ++i)
            c->fsink(&s.h_wire[i].mm, &s.h_fields[i], c->fuser);
    } else if (!c->fsink && sink && sink != msd_array_sink) {
        for (uint32_t i = 0; i < total; ++i)
            sink(&s.h_wire[i].mm, user);
    }
    return 0;
}

/* Wait for a batch's lists, resolve in order, deliver messages. */
int finish(msd_ctx *c, Slot &s, int format, msd_message_fn sink, void *user,
           const uint64_t *ts_override, const double *means_override, uint64_t resolver_first_chunk)
{
    auto ta = std::chrono::steady_clock::now();
    if (s.reset_before) { /* msd_restart(): every batch of the previous capture has been delivered */
        c->helper.wait(); /* ... and its statistics are complete */
        if (s.state_reset_done)
            msd_resolver_reset_stats(&c->resolver);
        else
            msd_resolver_reset(&c->resolver);
        s.state_reset_done = false;
        { /* the counters start over with the capture; the kernel-time sampling (one batch in timing_interval) runs on */
            const uint64_t timed = c->timing.timed_batches;
            const float scan_ms = c->timing.scan_kernel_ms, other_ms = c->timing.other_kernels_ms;
            memset(&c->timing, 0, sizeof c->timing);
            c->timing.timed_batches = timed;
            c->timing.scan_kernel_ms = scan_ms;
            c->timing.other_kernels_ms = other_ms;
        }
        s.reset_before = false;
    }
    int rc = start_download(c, s, format);
    if (rc)
        return rc;
    uint64_t H = s.lean ? 0 : s.h_totals[0], Tn = s.lean ? 0 : s.h_totals[1];
    HIPCHK(c, hipEventSynchronize(s.ev_copy1));
    auto tb = std::chrono::steady_clock::now();
    s.download_started = false;
    /* the following batch's lists can come down while this one is resolved on the host */
    if (c->outstanding > 1) {
        Slot &nx = c->slots[(c->head + 1) % MSD_PIPELINE_DEPTH];
        if (&nx != &s && nx.busy && !nx.lean && hipEventQuery(nx.ev_totals) == hipSuccess) {
            rc = start_download(c, nx, c->scan_format); /* its kernels are done: does not block */
            if (rc)
                return rc;
        }
    }

    /* per-buffer sample counts and means (mag_buf.validLength-overlap, .mean_level, .mean_power); a lean batch's
     * sums arrive with its first resolve pass (finish_gpu) */
    c->valid.assign(s.nbuffers, 0);
    c->means.assign(2 * (size_t)s.nbuffers, 0.0);
    for (uint32_t b = 0; b < s.nbuffers && !s.lean; ++b) {
        const uint64_t first = (uint64_t)b * MSD_CHUNK_SAMPLES;
        uint64_t n = s.nsamples > first ? s.nsamples - first : 0;
        if (n > MSD_CHUNK_SAMPLES)
            n = MSD_CHUNK_SAMPLES;
        c->valid[b] = (uint32_t)n;
        if (means_override) {
            c->means[2 * b] = means_override[2 * b];
            c->means[2 * b + 1] = means_override[2 * b + 1];
        } else if (format == MSD_FMT_SC16 || format == MSD_FMT_SC16Q11 || s.dc) {
            /* convert.c:245-251: float sum / unsigned -> float division, widened to double */
            c->means[2 * b] = (double)(s.h_fmeans[2 * b] / (float)(unsigned)n);
            c->means[2 * b + 1] = (double)(s.h_fmeans[2 * b + 1] / (float)(unsigned)n);
        } else {
            /* convert.c:104-110 (note 65536 for the level, 65535^2 for the power) */
            c->means[2 * b] = (double)s.h_sums[2 * b] / 65536.0 / (double)(unsigned)n;
            c->means[2 * b + 1] = (double)s.h_sums[2 * b + 1] / 65535.0 / 65535.0 / (double)(unsigned)n;
        }
    }

    auto t0 = std::chrono::steady_clock::now();
    const bool skip_resolve = (c->debug_flags & 0x1c) != 0; /* perf experiments with incomplete candidates */
    if (s.gpu_resolve) {
        rc = (ts_override || skip_resolve) ? 1 : finish_gpu(c, s, format, sink, user);
        s.resolve_inflight = false;
        if (rc == 2 && s.lean) { /* the region slices overflowed: bigger ones, one more scan, and the GPU resolve again */
            const int g = grow_and_rescan(c, s, format);
            if (g < 0)
                return g;
            if (g == 0) {
                s.ahead_done = false;
                s.ahead_verdict = 0;
                rc = finish_gpu(c, s, format, sink, user);
                s.resolve_inflight = false;
            }
        }
        if (rc < 0)
            return rc;
        if (s.lean) {
            H = s.h_totals[0];
            Tn = s.h_totals[1];
        }
        if (rc == 0) {
            auto t1 = std::chrono::steady_clock::now();
            if (c->outstanding > 1) {
                Slot &nx = c->slots[(c->head + 1) % MSD_PIPELINE_DEPTH];
                if (&nx != &s && nx.busy && !nx.download_started) {
                    rc = start_download(c, nx, c->scan_format);
                    if (rc)
                        return rc;
                }
            }
            if (c->trace) {
                auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                fprintf(stderr, "finish: wait-download %.3f  means %.3f  gpu resolve+power+sink %.3f ms\n", ms(ta, tb),
                        ms(tb, t0), ms(t0, t1));
            }
            float ms = 0;
            c->timing.hits = H;
            c->timing.tries = Tn;
            if (s.timed && hipEventElapsedTime(&ms, s.ev_start, s.ev_scan) == hipSuccess)
                c->timing.scan_kernel_ms = ms, c->timing.timed_batches++;
            if (s.timed && hipEventElapsedTime(&ms, s.ev_scan, s.ev_kernels) == hipSuccess)
                c->timing.other_kernels_ms = ms;
            if (hipEventElapsedTime(&ms, s.ev_copy0, s.ev_copy1) == hipSuccess)
                c->timing.d2h_ms = ms;
            c->timing.resolve_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
            s.busy = false;
            return 0;
        }
        /* the host resolver takes the batch: it needs the lists after all */
        const bool overflowed = rc == 2;
        if (overflowed) { /* lean layout, arenas overflowed: scanned again in pieces, stitched on the host */
            if (c->pending_emit == &s)
                c->pending_emit = nullptr; /* its speculative records are void */
            HIPCHK(c, hipStreamSynchronize(c->stream));
            s.lean = false;
            rc = rerun_in_pieces(c, s, format);
            if (rc)
                return rc;
            means_from_sums(c, s); /* the pieces' gather / publish kernels published the sums again */
            s.gpu_resolve = false;
        } else if (s.lean) {
            means_from_sums(c, s); /* (published by the first resolve pass, which did run) */
            rc = lean_gather_now(c, s);
            if (rc)
                return rc;
        }
        H = s.h_totals[0];
        Tn = s.h_totals[1];
        rc = overflowed ? 0 : ensure_host(c, s, H, Tn);
        if (rc)
            return rc;
        if (H && !overflowed)
            HIPCHK(c, hipMemcpyAsync(s.h_hits, s.d_hits, H * sizeof(msd_hit), hipMemcpyDeviceToHost, c->aux_stream));
        if (Tn && !overflowed)
            HIPCHK(c, hipMemcpyAsync(s.h_tries, s.d_tries, Tn * sizeof(msd_try), hipMemcpyDeviceToHost, c->aux_stream));
        if (c->cfg.mode_ac && !overflowed) { /* (rerun_in_pieces has stitched the pieces' Mode A/C lists on the host already; the
                                                device holds the last piece's only -- round 5's fuzzer, drawing arena sizes,
                                                found a reply six buffers early: this copy used to run in both cases) */
            const uint64_t nac = s.h_ac_totals[0];
            rc = ensure_ac_host(c, s, nac);
            if (rc)
                return rc;
            if (nac)
                HIPCHK(c, hipMemcpyAsync(s.h_ac, s.d_ac, nac * sizeof(msd_ac_hit), hipMemcpyDeviceToHost, c->aux_stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->aux_stream));
        c->timing.resolve_fallback++;
    }
    c->helper.wait(); /* the statistics of the previous batch, if they are still being summed */
    c->timing.resolve_passes = 0;
    c->out_msgs.clear();
    c->out_req.clear();
    c->out_buf.clear();
    apply_dropped(c, s);
    if (!skip_resolve)
        msd_resolve_batch(&c->resolver, resolver_first_chunk, s.nbuffers, c->valid.data(), s.h_hits, H, s.h_tries, Tn,
                      c->cfg.mode_ac ? s.h_ac : nullptr, c->cfg.mode_ac ? s.h_ac_totals[0] : 0, ts_override,
                      emit_thunk, c);
    auto t1 = std::chrono::steady_clock::now();
    if (c->outstanding > 1) { /* if the next batch was still running before the resolve, fetch it now */
        Slot &nx = c->slots[(c->head + 1) % MSD_PIPELINE_DEPTH];
        if (&nx != &s && nx.busy && !nx.download_started) {
            rc = start_download(c, nx, c->scan_format);
            if (rc)
                return rc;
        }
    }

    /* signal power of the accepted messages: a small follow-up kernel on the copy stream */
    const size_t nm = c->out_msgs.size();
    if (nm) {
        rc = ensure_req(c, s, nm);
        if (rc)
            return rc;
        memcpy(s.h_req, c->out_req.data(), nm * sizeof(uint64_t));
        if (c->magbuf_views) {
            /* msd_demodulate_magbuf[s]: the caller's magnitudes are in host memory already -- the sums of squares of the few
             * accepted messages (demod_2400.c:386-399: m[j + 19 + k], k < msglen * 12 / 5) cost less here than a request
             * upload, a kernel, a download and a synchronisation (35 us of a 165 us call).  Position -> buffer: the batch
             * is the buffers' new samples one after the other; a message may run on into the next buffer's. */
            for (size_t i = 0; i < nm; ++i) {
                const uint64_t rq = c->out_req[i];
                const int64_t first = (int64_t)(rq >> 16) - (int64_t)MSD_OVERLAP + 19; /* index into the batch's new samples */
                const uint64_t len = rq & 0xffffu;
                uint64_t acc = 0;
                if (first >= 0 && len) { /* nearly always the message lies inside one buffer's new samples: a plain sum of squares over
                                            consecutive u16 (the general walk below cost 0.9 ns a sample, a quarter of a twelve-buffer call) */
                    const uint64_t b = (uint64_t)first / MSD_CHUNK_SAMPLES, o = (uint64_t)first % MSD_CHUNK_SAMPLES;
                    if (b < c->magbuf_nviews && o + len <= MSD_CHUNK_SAMPLES && o + len + MSD_OVERLAP <= c->magbuf_views[b].validLength) {
                        const uint16_t *m = c->magbuf_views[b].data + MSD_OVERLAP + o;
                        for (uint64_t k = 0; k < len; ++k)
                            acc += (uint64_t)((uint32_t)m[k] * (uint32_t)m[k]);
                        s.h_pow[i] = acc;
                        continue;
                    }
                }
                for (uint64_t k = 0; k < len; ++k) {
                    const int64_t idx = first + (int64_t)k;
                    uint64_t x = 0;
                    if (idx < 0) {
                        if (idx >= -(int64_t)MSD_OVERLAP)
                            x = c->magbuf_views[0].data[(int64_t)MSD_OVERLAP + idx];
                    } else {
                        const uint64_t b = (uint64_t)idx / MSD_CHUNK_SAMPLES, o = (uint64_t)idx % MSD_CHUNK_SAMPLES;
                        if (b < c->magbuf_nviews && o + MSD_OVERLAP < c->magbuf_views[b].validLength)
                            x = c->magbuf_views[b].data[MSD_OVERLAP + o];
                    }
                    acc += x * x;
                }
                s.h_pow[i] = acc;
            }
        } else {
        /* its own stream: the copy stream may already be busy downloading the next batch's lists */
        HIPCHK(c, hipMemcpyAsync(s.d_req, s.h_req, nm * sizeof(uint64_t), hipMemcpyHostToDevice, c->aux_stream));
        MsdScanParams p{};
        fill_params(c, s, p);
        rc = msd_launch_power(&p, format, s.d_req, (uint32_t)nm, reinterpret_cast<unsigned long long *>(s.d_pow),
                              c->aux_stream);
        if (rc)
            return fail(c, rc, "power kernel launch failed");
        HIPCHK(c, hipMemcpyAsync(s.h_pow, s.d_pow, nm * sizeof(uint64_t), hipMemcpyDeviceToHost, c->aux_stream));
        HIPCHK(c, hipStreamSynchronize(c->aux_stream));
        }
    }
    msd_resolve_power(&c->resolver, s.nbuffers, c->valid.data(), c->means.data(), c->out_msgs.data(), sizeof(msd_message),
                      c->out_req.data(), c->out_buf.data(), s.h_pow, sizeof(uint64_t), nm);
    auto t2 = std::chrono::steady_clock::now();
    if (c->fsink) { /* header fields on the host for the batches resolved here */
        c->out_fields.resize(nm ? nm : 1);
        msd_fields_batch(c->out_msgs.data(), sizeof(msd_message), c->out_buf.data(), nm, c->out_fields.data());
        for (size_t i = 0; i < nm; ++i)
            c->fsink(&c->out_msgs[i], &c->out_fields[i], c->fuser);
    } else if (sink)
        for (size_t i = 0; i < nm; ++i)
            sink(&c->out_msgs[i], user);
    if (c->trace) {
        auto t3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "finish: wait-download %.3f  next-download+means %.3f  resolve %.3f  power %.3f  sink %.3f ms\n",
                ms(ta, tb), ms(tb, t0), ms(t0, t1), ms(t1, t2), ms(t2, t3));
    }

    float ms = 0;
    c->timing.hits = H;
    c->timing.tries = Tn;
    if (s.timed && hipEventElapsedTime(&ms, s.ev_start, s.ev_scan) == hipSuccess)
        c->timing.scan_kernel_ms = ms, c->timing.timed_batches++;
    if (s.timed && hipEventElapsedTime(&ms, s.ev_scan, s.ev_kernels) == hipSuccess)
        c->timing.other_kernels_ms = ms;
    if (hipEventElapsedTime(&ms, s.ev_copy0, s.ev_copy1) == hipSuccess)
        c->timing.d2h_ms = ms;
    c->timing.resolve_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    s.busy = false;
    return 0;
}

int check_batch(msd_ctx *c, const void *p, uint64_t nsamples, int last)
{
    if (!c)
        return -EINVAL;
    if (c->failed)
        return -EIO;
    if (c->finished)
        return fail(c, -EINVAL, "capture already finished; call msd_reset()");
    if (nsamples > c->cfg.max_batch_samples || nsamples > MSD_MAX_BATCH_SAMPLES)
        return fail(c, -E2BIG, "batch of %llu samples exceeds max_batch_samples %llu",
                    (unsigned long long)nsamples, (unsigned long long)c->cfg.max_batch_samples);
    if (!last && (nsamples == 0 || nsamples % MSD_CHUNK_SAMPLES != 0))
        return fail(c, -EINVAL, "only the last batch may be a partial buffer");
    if (nsamples && (!p || (reinterpret_cast<uintptr_t>(p) & 15u)))
        return fail(c, -EINVAL, "IQ pointer must be non-null and 16-byte aligned");
    return 0;
}

/* --dcfilter: the batch's IQ -> DC-blocked magnitudes and squares, the converter state advanced; strictly in stream
 * order on `stream`.  The parallel-in-time kernels first, the in-order kernel behind them (it returns at once when they
 * came out exact -- they always have so far; MSD_CFG_DC_SEQUENTIAL: the in-order kernel alone). */
int launch_dc_block(msd_ctx *c, const void *d_iq, uint64_t nsamples, uint16_t *d_mag, float *d_magsq, hipStream_t stream)
{
    const void *skip_if = nullptr;
    c->dc_last_parallel = false;
    if (c->d_dc_work && nsamples && (reinterpret_cast<uintptr_t>(d_iq) & 15u) == 0) {
        const uint32_t L = msd_dcp_block_len(nsamples);
        /* a pass that is not needed is two launches that return at once, 12 us: a batch of a buffer or two (at most 128 blocks,
         * 5-10 passes measured, profiles/r06_dc_passes.txt) gets 12 queued, not 24 -- what the in-order kernel behind them
         * would cost such a batch if they ever ran out is a millisecond */
        const int passes = (nsamples + L - 1) / L <= 128u && c->dc_passes > 12 ? 12 : c->dc_passes;
        const int rc = msd_launch_dcfilter_parallel(c->cfg.format, d_iq, nsamples, c->dc_a, c->dc_b, c->d_dcstate, d_mag, d_magsq,
                                                    c->d_dc_work, L, passes, c->dc_fused ? 1 : 0, stream);
        if (rc)
            return rc;
        skip_if = c->d_dc_work;
        c->dc_last_parallel = true;
        c->dc_last_blocks = (uint32_t)((nsamples + L - 1) / L);
    }
    return msd_launch_dcfilter(c->cfg.format, d_iq, nsamples, c->dc_a, c->dc_b, c->d_dcstate, d_mag, d_magsq, skip_if, stream);
}

int launch(msd_ctx *c, const void *d_iq, uint64_t nsamples, int last)
{
    int rc = check_batch(c, d_iq, nsamples, last);
    if (rc)
        return rc;
    if (c->outstanding >= MSD_PIPELINE_DEPTH)
        return fail(c, -EBUSY, "pipeline full: call msd_collect() first");
    Slot &s = c->slots[(c->head + c->outstanding) % MSD_PIPELINE_DEPTH];
    s.busy = true;
    s.launch_seq = ++c->launch_count;
    s.d_iq = static_cast<const uint8_t *>(d_iq);
    s.d_prev = c->d_tail[c->tail_cur];
    s.have_prev = c->have_prev ? 1 : 0;
    s.d_mag_prev = c->mag_prev;
    s.batch_first = c->next_sample;
    s.nsamples = nsamples;
    s.last = last;
    s.dropped_before = c->pending_dropped;
    c->pending_dropped = 0;
    s.reset_before = c->restart_pending;
    c->restart_pending = false;
    s.threshold = c->cfg.preamble_threshold;
    s.dc = c->dc;
    if (c->dc) { /* the converter proper: IQ -> DC-blocked magnitudes, strictly in stream order */
        rc = launch_dc_block(c, d_iq, nsamples, s.d_dcmag, s.d_magsq, c->stream);
        if (rc) {
            s.busy = false;
            return fail(c, rc, "DC filter kernel launch failed");
        }
        s.d_iq = reinterpret_cast<const uint8_t *>(s.d_dcmag);
    } else if (c->q11_bits) { /* convert_sc16q11_table: the scan runs on its magnitudes and its integer sums are the converter's */
        rc = msd_launch_q11_table(d_iq, nsamples, c->d_q11_table, c->q11_bits, s.d_dcmag, nullptr, c->cu_count, c->stream);
        if (rc) {
            s.busy = false;
            return fail(c, rc, "SC16Q11 table converter launch failed");
        }
        s.d_iq = reinterpret_cast<const uint8_t *>(s.d_dcmag);
    }
    /* a capture of N samples is floor(N/131072)+1 buffers, the last possibly empty
     * (sdr_ifile.c:192-216: EOF is only noticed by a short read) */
    s.nbuffers = (uint32_t)(nsamples / MSD_CHUNK_SAMPLES) + (last ? 1u : 0u);
    const int tail_nxt = (c->tail_cur + 1) % (MSD_PIPELINE_DEPTH + 1);
    s.tail_dst = nsamples >= (uint64_t)TAIL_SAMPLES ? c->d_tail[tail_nxt] : nullptr;
    auto tl0 = std::chrono::steady_clock::now();
    c->scan_queued = false;
    s.gpu_resolve = gpu_eligible(c, s);
    rc = enqueue(c, s, c->scan_format, nullptr, true);
    if (rc) { /* nothing was consumed: the dropped samples and a pending restart wait for the next launch */
        if (c->scan_queued) /* ... unless kernels of this batch are on the stream already: their follow-ups are missing,
                               the slot's lists and sums are half written -- only msd_reset() starts over */
            c->failed = true;
        s.busy = false;
        c->pending_dropped += s.dropped_before;
        c->restart_pending = c->restart_pending || s.reset_before;
        s.dropped_before = 0;
        s.reset_before = false;
        return rc;
    }
    if (c->trace) {
        static const auto t_origin = std::chrono::steady_clock::now();
        fprintf(stderr, "launch: at %.3f ms, enqueue %.3f ms\n",
                std::chrono::duration<double, std::milli>(tl0 - t_origin).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count());
    }
    s.resolve_inflight = false;
    s.ahead_done = false;
    s.ahead_verdict = 0;
    if (s.gpu_resolve && c->outstanding == 0) { /* no earlier batch to wait for: resolve right behind the scan */
        rc = gpu_begin(c, s, c->scan_format);
        if (rc) {
            s.busy = false;
            return rc;
        }
    }
    if (nsamples >= (uint64_t)TAIL_SAMPLES) {
        if (s.tail_dst) /* no gather kernel ran (cannot happen with that many samples) */
            HIPCHK(c, hipMemcpyAsync(s.tail_dst, s.d_iq + (nsamples - TAIL_SAMPLES) * c->scan_bps,
                                     (size_t)TAIL_SAMPLES * c->scan_bps, hipMemcpyDeviceToDevice, c->stream));
        s.tail_dst = nullptr;
        c->tail_cur = tail_nxt;
        c->have_prev = true;
        c->mag_prev = s.mag_pass ? s.d_mag + (nsamples - TAIL_SAMPLES) : nullptr;
    }
    c->next_sample += nsamples;
    c->outstanding++;
    if (last)
        c->finished = true;
    return 0;
}

int collect(msd_ctx *c, msd_message_fn sink, void *user)
{
    if (!c)
        return -EINVAL;
    if (c->failed)
        return -EIO; /* msd_last_error() still says why; msd_reset() starts over */
    if (c->outstanding == 0)
        return fail(c, -ENODATA, "no batch outstanding");
    Slot &s = c->slots[c->head];
    int rc = finish(c, s, c->scan_format, sink, user, nullptr, nullptr, s.batch_first / MSD_CHUNK_SAMPLES);
    if (rc < 0) { /* the batch is lost and the filter / clocks are in an unknown state: the context refuses further
                     batches until msd_reset() */
        c->failed = true;
        s.busy = false;
    }
    c->head = (c->head + 1) % MSD_PIPELINE_DEPTH;
    c->outstanding--;
    return rc;
}

void destroy(msd_ctx *c)
{
    if (!c)
        return;
    c->helper.shutdown();
    (void)hipSetDevice(c->cfg.device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream)
        (void)hipStreamSynchronize(c->copy_stream);
    if (c->aux_stream)
        (void)hipStreamSynchronize(c->aux_stream);
    if (c->emit_stream)
        (void)hipStreamSynchronize(c->emit_stream);
    if (c->d_timers) {
        unsigned long long t[16];
        if (hipMemcpy(t, c->d_timers, sizeof t, hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "kernel section cycles (sum over wavefronts):");
            for (int k = 0; k < 12; ++k)
                fprintf(stderr, " [%d]=%llu", k, t[k]);
            fprintf(stderr, "\n");
        }
        (void)hipFree(c->d_timers);
    }
    for (Slot &s : c->slots) {
        (void)hipFree(s.d_hits); (void)hipFree(s.d_tries); (void)hipFree(s.d_totals); (void)hipFree(s.d_buf_first); (void)hipFree(s.d_rec_off); (void)hipFree(s.d_tile_sums); (void)hipFree(s.d_sums); (void)hipFree(s.d_fmeans);
        (void)hipFree(s.d_fm_work); (void)hipFree(s.d_ac_regions); (void)hipFree(s.d_ac_counts);
        if (s.h_totals) (void)hipHostFree(s.h_totals);
        if (s.h_sums) (void)hipHostFree(s.h_sums);
        if (s.h_fmeans) (void)hipHostFree(s.h_fmeans);
        if (s.h_hits) (void)hipHostFree(s.h_hits);
        if (s.h_tries) (void)hipHostFree(s.h_tries);
        (void)hipFree(s.d_req); (void)hipFree(s.d_pow);
        if (s.h_req) (void)hipHostFree(s.h_req);
        if (s.h_pow) (void)hipHostFree(s.h_pow);
        (void)hipFree(s.d_ac); (void)hipFree(s.d_ac_totals); (void)hipFree(s.d_mag); (void)hipFree(s.d_ragged);
        (void)hipFree(s.d_acc); if (s.d_adds) (void)hipHostFree(s.d_adds); (void)hipFree(s.d_nmsgs); (void)hipFree(s.d_powr); (void)hipFree(s.d_pred); (void)hipFree(s.d_rhits); (void)hipFree(s.d_rtries); (void)hipFree(s.d_rcounts); (void)hipFree(s.d_rwgt); (void)hipFree(s.d_acc_ac); (void)hipFree(s.d_nac);
        if (s.h_rbuf) (void)hipHostFree(s.h_rbuf);
        if (s.h_ctl) (void)hipHostFree(s.h_ctl);
        if (s.h_side) (void)hipHostFree(s.h_side);
        if (s.h_wire) (void)hipHostFree(s.h_wire);
        if (s.h_fields) (void)hipHostFree(s.h_fields);
        (void)hipFree(s.d_dcmag); (void)hipFree(s.d_magsq);
        if (s.ev_resolve) (void)hipEventDestroy(s.ev_resolve);
        if (s.ev_records) (void)hipEventDestroy(s.ev_records);
        if (s.ev_power) (void)hipEventDestroy(s.ev_power);
        if (s.ev_scanned) (void)hipEventDestroy(s.ev_scanned);
        if (s.ev_upload) (void)hipEventDestroy(s.ev_upload);
        (void)hipFree(s.d_upload);
        (void)hipFree(s.d_wire);
        (void)hipFree(s.d_fields);
        if (s.h_ac_totals) (void)hipHostFree(s.h_ac_totals);
        if (s.h_ac) (void)hipHostFree(s.h_ac);
        hipEvent_t *evs[] = {&s.ev_start, &s.ev_scan, &s.ev_kernels, &s.ev_totals, &s.ev_copy0, &s.ev_copy1};
        for (hipEvent_t *e : evs)
            if (*e)
                (void)hipEventDestroy(*e);
    }
    (void)hipFree(c->d_lut); (void)hipFree(c->d_crc); (void)hipFree(c->d_syn56); (void)hipFree(c->d_syn112); (void)hipFree(c->d_slicer); (void)hipFree(c->d_synhash);
    (void)hipFree(c->d_fix2[0]); (void)hipFree(c->d_fix2[1]);
    (void)hipFree(c->d_dcstate); (void)hipFree(c->d_dc_work); (void)hipFree(c->d_fm_work); (void)hipFree(c->d_q11_table); (void)hipFree(c->d_conv_magsq);
    if (c->h_conv)
        (void)hipHostFree(c->h_conv);
    (void)hipFree(c->d_region_hits); (void)hipFree(c->d_region_tries); (void)hipFree(c->d_counts); (void)hipFree(c->d_wg_totals);
    (void)hipFree(c->d_ac_offsets);
    (void)hipFree(c->d_noise);
    (void)hipFree(c->d_snaps);
    if (c->h_snaps) (void)hipHostFree(c->h_snaps);
    if (c->ev_aux) (void)hipEventDestroy(c->ev_aux);
    if (c->ev_inputs) (void)hipEventDestroy(c->ev_inputs);
    if (c->h_pred) (void)hipHostFree(c->h_pred);
    if (c->h_pred_count) (void)hipHostFree(c->h_pred_count);
    if (c->h_patches) (void)hipHostFree(c->h_patches);
    for (uint8_t *t : c->d_tail)
        (void)hipFree(t);
    (void)hipFree(c->d_stage);
    (void)hipFree(c->d_mag);
    if (c->copy_stream)
        (void)hipStreamDestroy(c->copy_stream);
    if (c->aux_stream)
        (void)hipStreamDestroy(c->aux_stream);
    if (c->emit_stream)
        (void)hipStreamDestroy(c->emit_stream);
    if (c->own_stream && c->stream)
        (void)hipStreamDestroy(c->stream);
    msd_resolver_free(&c->resolver);
    free(c->tables);
    delete c;
}

} /* namespace */

extern "C" {

void msd_array_sink(const msd_message *mm, void *state)
{
    msd_array_sink_state *st = static_cast<msd_array_sink_state *>(state);
    if (st->count < st->cap)
        st->out[st->count] = *mm;
    st->count++;
}

static int create_context(const msd_config *cfg, msd_ctx **out, bool *out_of_memory)
{
    if (!cfg || !out)
        return -EINVAL;
    *out = nullptr;
    if (cfg->format < MSD_FMT_UC8 || cfg->format > MSD_FMT_MAG16 || cfg->nfix_crc < 0 || cfg->nfix_crc > 2 ||
        cfg->preamble_threshold < 1 || cfg->preamble_threshold > MSD_MAX_PREAMBLE_THRESHOLD ||
        ((cfg->flags & MSD_CFG_DC_FILTER) && cfg->format == MSD_FMT_MAG16) || cfg->sc16q11_table_bits < 0 ||
        cfg->sc16q11_table_bits > 11 || (cfg->sc16q11_table_bits && cfg->format != MSD_FMT_SC16Q11) ||
        !(cfg->sample_rate >= 0.0) || (cfg->sample_rate > 0.0 && cfg->sample_rate < 1.0)) {
        snprintf(g_create_err, sizeof g_create_err, "msd_create: invalid configuration");
        return -EINVAL;
    }
    g_create_err[0] = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
    {   /* no GPU: there is deliberately no CPU fallback */
        snprintf(g_create_err, sizeof g_create_err, "no usable HIP device (found %d, asked for device %d): this library has no CPU path", ndev, cfg->device);
        return -ENODEV;
    }
    msd_ctx *c = new (std::nothrow) msd_ctx;
    if (!c)
        return -ENOMEM;
    c->cfg = *cfg;
    if (c->cfg.max_batch_samples == 0)
        c->cfg.max_batch_samples = MSD_CHUNK_SAMPLES;
    if (c->cfg.max_batch_samples > MSD_MAX_BATCH_SAMPLES)
        c->cfg.max_batch_samples = MSD_MAX_BATCH_SAMPLES;
    c->bps = (cfg->format == MSD_FMT_UC8 || cfg->format == MSD_FMT_MAG16) ? 2 : 4;
    c->dc = (cfg->flags & MSD_CFG_DC_FILTER) != 0;
    c->q11_bits = (!c->dc && cfg->format == MSD_FMT_SC16Q11) ? cfg->sc16q11_table_bits : 0;
    c->scan_format = (c->dc || c->q11_bits) ? (int)MSD_FMT_MAG16 : cfg->format;
    c->scan_bps = bps_of(c->scan_format);
    if (c->dc) { /* init_converter's "DC block @ 1Hz", convert.c:479-482, at its sample_rate argument (Modes.sample_rate = 2.4 MHz) */
        c->dc_b = (float)exp(-2.0 * M_PI * 1.0 / (cfg->sample_rate > 0.0 ? cfg->sample_rate : 2400000.0));
        c->dc_a = (float)(1.0 - c->dc_b);
    }

#define CK(call)                                                              \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            snprintf(g_create_err, sizeof g_create_err, "msd_create: %s: %s", #call, hipGetErrorString(e_)); \
            if (e_ == hipErrorOutOfMemory)                                    \
                *out_of_memory = true;                                        \
            (void)hipGetLastError();                                          \
            destroy(c);                                                       \
            return -EIO;                                                      \
        }                                                                     \
    } while (0)

    CK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, cfg->device));
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (cfg->stream) {
        c->stream = static_cast<hipStream_t>(cfg->stream);
    } else {
        CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    CK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    { /* the resolve/power follow-ups are short and on the critical path: let them jump the queued scans */
        int least = 0, greatest = 0;
        CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        CK(hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, greatest));
        CK(hipStreamCreateWithPriority(&c->emit_stream, hipStreamNonBlocking, greatest));
    }

    c->tables = static_cast<msd_tables *>(malloc(sizeof(msd_tables)));
    if (!c->tables) {
        destroy(c);
        return -ENOMEM;
    }
    msd_tables_build(c->tables, cfg->nfix_crc);
    if (msd_tables_selftest(c->tables) != 0) {
        destroy(c);
        return -EDOM; /* the folded UC8 table would not reproduce the reference's */
    }
    static_assert(offsetof(msd_tables, uc8_scan) == sizeof(((msd_tables *)nullptr)->uc8_folded) &&
                  sizeof(((msd_tables *)nullptr)->uc8_folded) == MSD_LUT_SCAN_OFFSET * sizeof(uint16_t),
                  "the scan kernel's table lies directly behind the folded one");
    CK(hipMalloc(reinterpret_cast<void **>(&c->d_lut), sizeof c->tables->uc8_folded + sizeof c->tables->uc8_scan));
    CK(hipMalloc(reinterpret_cast<void **>(&c->d_crc), sizeof c->tables->crc_byte));
    CK(hipMalloc(reinterpret_cast<void **>(&c->d_syn56), sizeof c->tables->syn56 + 16));
    CK(hipMalloc(reinterpret_cast<void **>(&c->d_syn112), sizeof c->tables->syn112 + 16));
    CK(hipMemcpy(c->d_lut, c->tables->uc8_folded, sizeof c->tables->uc8_folded + sizeof c->tables->uc8_scan, hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_crc, c->tables->crc_byte, sizeof c->tables->crc_byte, hipMemcpyHostToDevice));
    CK(hipMalloc(reinterpret_cast<void **>(&c->d_slicer), sizeof c->tables->slicer));
    CK(hipMemcpy(c->d_slicer, c->tables->slicer, sizeof c->tables->slicer, hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_syn56, c->tables->syn56, sizeof c->tables->syn56, hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_syn112, c->tables->syn112, sizeof c->tables->syn112, hipMemcpyHostToDevice));
    CK(hipMalloc(reinterpret_cast<void **>(&c->d_synhash), sizeof c->tables->synhash));
    CK(hipMemcpy(c->d_synhash, c->tables->synhash, sizeof c->tables->synhash, hipMemcpyHostToDevice));
    if (cfg->nfix_crc == 2) { /* --aggressive, crc.c:374-379 */
        for (int k = 0; k < 2; ++k) {
            uint64_t *tab = msd_fix2_table(c->tables, k ? 112 : 56, &c->fix2_lg[k]);
            if (!tab) {
                destroy(c);
                return -ENOMEM;
            }
            const size_t bytes = sizeof(uint64_t) << c->fix2_lg[k];
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->d_fix2[k]), bytes);
            if (e == hipSuccess)
                e = hipMemcpy(c->d_fix2[k], tab, bytes, hipMemcpyHostToDevice);
            free(tab);
            CK(e);
        }
    }

    const uint64_t B = c->cfg.max_batch_samples;
    /* Candidate arenas.  Base size: one hit per 8 samples, one live try per 16 -- eight times the benchmark capture's
     * density, and all that round 1's 8 GB budget allowed.  Default since round 4: four times the base, one hit per 2
     * samples and one try per 4 (14.5 GB per context at 128 Mi-sample batches, of 288): a burst of pulse trains that fills a
     * tenth of every buffer with preambles of five trial phases each stays on the fast path instead of sending the
     * whole batch through rerun_in_pieces (profiles/r04_density.txt: 74 against 1.2 GS/s).  What overflows even these is
     * rescanned in pieces as before -- nothing is ever truncated. */
    uint64_t hit_want = B / 8, try_want = B / 16;
    if (cfg->test_arena_permille > 0) { /* explicit size in thousandths of the base (tests: provoke the overflow path) */
        const uint64_t pm = (uint64_t)cfg->test_arena_permille;
        hit_want = hit_want * pm / 1000;
        try_want = try_want * pm / 1000;
    } else {
        hit_want *= 4;
        try_want *= 4;
    }
    c->hit_arena = hit_want > MIN_HIT_ARENA ? hit_want : MIN_HIT_ARENA;
    c->try_arena = try_want > MIN_TRY_ARENA ? try_want : MIN_TRY_ARENA;
    c->max_wg = (uint32_t)c->cu_count * MSD_SCAN_WAVES * MSD_SCAN_WGS_PER_CU;
    c->max_buffers = (uint32_t)(B / MSD_CHUNK_SAMPLES) + 2u;
    /* (the arenas of the layouts without region slices -- c->d_region_*, the slots' dense lists d_hits / d_tries: 60 of the
     * 108 bytes per sample -- are made when a batch first takes such a layout: ensure_dense(); a context that only ever
     * runs the lean pipelined path, the default, never allocates them) */
    CK(hipMalloc(reinterpret_cast<void **>(&c->d_counts), c->max_wg * sizeof(msd_region_counts)));
    CK(hipMalloc(reinterpret_cast<void **>(&c->d_wg_totals), (size_t)c->cu_count * MSD_SCAN_WGS_PER_CU * sizeof(msd_wg_totals)));
    if (cfg->mode_ac) {
        c->ac_arena = B / 32 > MIN_HIT_ARENA ? B / 32 : MIN_HIT_ARENA;
        c->ac_max_wg = (uint32_t)c->cu_count * 28u; /* regions of the Mode A/C candidate kernel: one per wavefront, 28 resident per CU (54 registers, 5 KB of LDS each) */
        CK(hipMalloc(reinterpret_cast<void **>(&c->d_ac_offsets), c->ac_max_wg * 2 * sizeof(uint64_t)));
        CK(hipMalloc(reinterpret_cast<void **>(&c->d_noise), c->max_buffers * sizeof(uint32_t)));
    }
    if (cfg->format != MSD_FMT_UC8 || (cfg->flags & MSD_CFG_DC_FILTER))
        CK(hipMalloc(&c->d_fm_work, msd_fm_work_bytes(1))); /* the converter entry's one buffer; the batches have their slots' */
    for (uint8_t *&t : c->d_tail) {
        CK(hipMalloc(reinterpret_cast<void **>(&t), (size_t)TAIL_SAMPLES * 4));
        CK(hipMemset(t, 0, (size_t)TAIL_SAMPLES * 4));
    }
    for (Slot &s : c->slots) {
        CK(hipMalloc(reinterpret_cast<void **>(&s.d_totals), 4 * sizeof(uint64_t)));
        CK(hipMalloc(reinterpret_cast<void **>(&s.d_buf_first), (c->max_buffers + 2) * sizeof(uint32_t)));
        if (cfg->format != MSD_FMT_UC8 || (cfg->flags & MSD_CFG_DC_FILTER))
            CK(hipMalloc(&s.d_fm_work, msd_fm_work_bytes(c->max_buffers)));
        if (cfg->mode_ac) {
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_ac_regions), c->ac_arena * sizeof(msd_ac_hit)));
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_ac_counts), c->ac_max_wg * sizeof(msd_wg_counts)));
        }
        if (cfg->format == MSD_FMT_SC16 || cfg->format == MSD_FMT_SC16Q11)
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_tile_sums), ((size_t)c->max_buffers * (MSD_CHUNK_SAMPLES / 1024) + 2) * 2 * sizeof(float)));
        CK(hipMalloc(reinterpret_cast<void **>(&s.d_rec_off), (c->max_buffers + 2) * sizeof(uint32_t)));
        CK(hipMalloc(reinterpret_cast<void **>(&s.d_ragged), 64));
        CK(hipMemset(s.d_ragged, 0, 64));
        CK(hipMalloc(reinterpret_cast<void **>(&s.d_sums), 2 * sizeof(uint64_t) * c->max_buffers));
        CK(hipMemset(s.d_sums, 0, 2 * sizeof(uint64_t) * c->max_buffers));
        CK(hipMalloc(reinterpret_cast<void **>(&s.d_fmeans), 2 * sizeof(float) * c->max_buffers));
        CK(hipHostMalloc(reinterpret_cast<void **>(&s.h_totals), 4 * sizeof(uint64_t)));
        if (cfg->mode_ac) {
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_ac), c->ac_arena * sizeof(msd_ac_hit)));
            if (!c->dc && !c->q11_bits && cfg->format != MSD_FMT_MAG16) /* (those scan magnitudes already) */
                CK(hipMalloc(reinterpret_cast<void **>(&s.d_mag), (cfg->max_batch_samples + 4096) * sizeof(uint16_t)));
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_ac_totals), 4 * sizeof(uint64_t)));
            CK(hipHostMalloc(reinterpret_cast<void **>(&s.h_ac_totals), 4 * sizeof(uint64_t)));
        }
        CK(hipHostMalloc(reinterpret_cast<void **>(&s.h_sums), 2 * sizeof(uint64_t) * c->max_buffers));
        CK(hipHostMalloc(reinterpret_cast<void **>(&s.h_fmeans), 2 * sizeof(float) * c->max_buffers));
        if (c->dc) {
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_dcmag), c->cfg.max_batch_samples * sizeof(uint16_t) + 64));
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_magsq), c->cfg.max_batch_samples * sizeof(float) + 64));
        } else if (c->q11_bits) { /* the table converter's magnitudes: what the scan kernel reads */
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_dcmag), c->cfg.max_batch_samples * sizeof(uint16_t) + 64));
            CK(hipMemset(s.d_dcmag, 0, c->cfg.max_batch_samples * sizeof(uint16_t) + 64));
        }
        hipEvent_t *evs[] = {&s.ev_start, &s.ev_scan, &s.ev_kernels, &s.ev_totals, &s.ev_copy0, &s.ev_copy1};
        for (hipEvent_t *e : evs)
            CK(hipEventCreate(e));
    }
    if (c->q11_bits) {
        std::vector<uint16_t> tab((size_t)1 << (2 * c->q11_bits));
        msd_sc16q11_table_build(c->q11_bits, tab.data());
        CK(hipMalloc(reinterpret_cast<void **>(&c->d_q11_table), tab.size() * sizeof(uint16_t)));
        CK(hipMemcpy(c->d_q11_table, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    if (c->dc) {
        CK(hipMalloc(reinterpret_cast<void **>(&c->d_dcstate), 2 * sizeof(float)));
        CK(hipMemset(c->d_dcstate, 0, 2 * sizeof(float))); /* convert.c:476-477 */
        if (!(cfg->flags & MSD_CFG_DC_SEQUENTIAL)) { /* 0.3 MB + 8 bytes per 64 samples of a batch */
            CK(hipMalloc(&c->d_dc_work, msd_dcp_work_bytes(c->cfg.max_batch_samples, 0)));
            c->dc_passes = (cfg->flags & MSD_CFG_DC_ONE_PASS) ? 1 : 24;
            c->dc_fused = (cfg->flags & MSD_CFG_DC_FUSED_LAUNCH) != 0;
        }
    }
    {
        c->trace = (cfg->flags & MSD_CFG_TRACE) != 0;
        c->resolver.trace = c->trace;
        c->resolver.threads = cfg->resolve_threads > 0 ? cfg->resolve_threads : 0;
        c->repass_aux = (cfg->flags & MSD_CFG_REPASS_AUX) != 0;
        /* Where the resolve chain (prediction, resolve, power, records) runs.  In order on the
         * scan stream when that stream carries nothing but scans (UC8 / magnitudes, Mode S only): a kernel that
         * shares the GPU with a scan slows it by about its own duration, so side streams buy 2 % there and make
         * the scan launches 15 % longer.  On side streams (prediction + resolve on a high-priority one, power +
         * records on a third) when the scan stream also carries the latency-bound float-sum or Mode A/C
         * kernels, which the chain overlaps well: +13..17 % whole-job rate, measured.  MSD_CFG_CHAIN_IN_ORDER /
         * MSD_CFG_CHAIN_SIDE_STREAMS override. */
        {
            const bool follow_ups = cfg->mode_ac || cfg->format == MSD_FMT_SC16 || cfg->format == MSD_FMT_SC16Q11 ||
                                    (cfg->flags & MSD_CFG_DC_FILTER);
            c->chain_inline = (cfg->flags & MSD_CFG_CHAIN_IN_ORDER) ? true : (cfg->flags & MSD_CFG_CHAIN_SIDE_STREAMS) ? false : !follow_ups;
        }
        c->no_helper = (cfg->flags & MSD_CFG_NO_HELPER) != 0;
        c->emit_fused = c->chain_inline && !c->repass_aux /* a re-pass on another stream would race the scan that carries the records */ &&
                        !(cfg->flags & MSD_CFG_DECODE_FIELDS) && !(cfg->flags & MSD_CFG_EMIT_KERNEL);
        /* the signal power in the resolve workgroups (no kernel of its own): in the in-order layout it takes a kernel
         * and a gap off the stream; on side streams, where the resolve kernel shares the GPU with a scan, a longer
         * resolve kernel costs more than the small power kernel behind it (measured: SC16 121 -> 124.5, Mode A/C 149 ->
         * 156 GS/s with the kernel) */
        c->power_fused = (cfg->flags & MSD_CFG_POWER_IN_RESOLVE) ? true : (cfg->flags & MSD_CFG_POWER_KERNEL) ? false : c->chain_inline;
        c->resolve_ahead = !(cfg->flags & MSD_CFG_NO_RESOLVE_AHEAD);
        c->wait_inputs_on_stream = (cfg->flags & MSD_CFG_WAIT_INPUTS_ON_STREAM) != 0;
        c->helper.device = cfg->device;
        c->debug_flags = cfg->debug_flags;
        c->gpu_resolve = !(cfg->flags & MSD_CFG_HOST_RESOLVE);
        c->want_fields = (cfg->flags & MSD_CFG_DECODE_FIELDS) != 0;
        c->records_dma = (cfg->flags & MSD_CFG_RECORDS_DMA) != 0;
        if (cfg->test_inline_adds > 0 && (uint32_t)cfg->test_inline_adds < MSD_RB_ADD_INLINE)
            c->inline_adds = (uint32_t)cfg->test_inline_adds;
        else if (cfg->test_inline_adds < 0)
            c->inline_adds = 0; /* every add through the long list */
    }
    /* (every layout: UC8 / magnitudes in order, 16-bit IQ and Mode A/C with the chain on side streams) */
    c->lean_ok = c->gpu_resolve && !c->dc && !(cfg->flags & MSD_CFG_NO_LEAN);
    if (c->lean_ok)
        for (Slot &s : c->slots) {
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_rhits), c->hit_arena * sizeof(msd_hit)));
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_rtries), c->try_arena * sizeof(msd_try)));
            s.rhit_arena = c->hit_arena;
            s.rtry_arena = c->try_arena;
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_rcounts), c->max_wg * sizeof(msd_region_counts)));
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_rwgt), (size_t)c->cu_count * MSD_SCAN_WGS_PER_CU * sizeof(msd_wg_totals)));
            CK(hipMemset(s.d_totals, 0, 4 * sizeof(uint64_t)));
        }
    if (c->gpu_resolve) {
        const size_t ctl_bytes = (size_t)28 * c->max_buffers;
        for (Slot &s : c->slots) {
            CK(hipHostMalloc(reinterpret_cast<void **>(&s.h_rbuf), sizeof(msd_rbuf) * c->max_buffers));
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_acc), sizeof(msd_acc) * MSD_RB_MSG_CAP * c->max_buffers));
            /* (page-locked host memory the kernels write into: see finish_gpu) */
            CK(hipHostMalloc(reinterpret_cast<void **>(&s.d_adds), sizeof(uint32_t) * MSD_RB_MSG_CAP * c->max_buffers));
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_nmsgs), sizeof(uint32_t) * c->max_buffers));
            if (cfg->mode_ac) {
                CK(hipMalloc(reinterpret_cast<void **>(&s.d_acc_ac), sizeof(uint32_t) * MSD_RB_AC_CAP * c->max_buffers));
                CK(hipMalloc(reinterpret_cast<void **>(&s.d_nac), sizeof(uint32_t) * c->max_buffers));
            }
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_pred), sizeof(uint32_t) * MSD_PRED_WORDS));
            CK(hipMemset(s.d_pred, 0xFF, sizeof(uint32_t) * MSD_PRED_WORDS)); /* every slot vacant (generation 0xff) */
            CK(hipMalloc(reinterpret_cast<void **>(&s.d_powr), sizeof(uint64_t) * MSD_RB_MSG_CAP * c->max_buffers));
            CK(hipHostMalloc(reinterpret_cast<void **>(&s.h_ctl), ctl_bytes));
            memset(s.h_ctl, 0, ctl_bytes);
            CK(hipEventCreateWithFlags(&s.ev_resolve, hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&s.ev_records, hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&s.ev_power, hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&s.ev_scanned, hipEventDisableTiming));
        }
        CK(hipMalloc(reinterpret_cast<void **>(&c->d_snaps), sizeof(uint32_t) * MSD_SNAP_WORDS * (SNAP_CAP + 1)));
        CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_snaps), sizeof(uint32_t) * MSD_SNAP_WORDS * (SNAP_CAP + 1)));
        CK(hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&c->ev_inputs, hipEventDisableTiming));
        CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_pred), sizeof(msd_pred_entry) * MSD_PRED_LIST));
        CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_pred_count), 64));
        CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_patches), sizeof(msd_pred_patch) * 2 * MSD_PRED_LIST));
        *c->h_pred_count = 0;
    } else {
        c->gpu_resolve = false;
    }
#undef CK
#ifdef MSD_KERNEL_TIMING /* -DMSD_KERNEL_TIMING builds only: the scan kernel's section clocks, printed by msd_destroy */
    {
        if (hipMalloc(reinterpret_cast<void **>(&c->d_timers), 16 * sizeof(unsigned long long)) == hipSuccess)
            (void)hipMemset(c->d_timers, 0, 16 * sizeof(unsigned long long));
    }
#endif
    c->resolver.stats = &c->stats;
    c->resolver.mode_ac = cfg->mode_ac;
    msd_resolver_reset(&c->resolver);
    *out = c;
    return 0;
}

int msd_create(const msd_config *cfg, msd_ctx **out)
{
    bool oom = false;
    int rc = create_context(cfg, out, &oom);
    if (rc && oom && cfg->test_arena_permille == 0) {
        /* The default candidate arenas are four times the base size (108 bytes per sample of max_batch_samples over
         * the four pipeline slots).  On a smaller or a shared GPU that may not fit where the base size does: one more
         * try at the base size -- the only price is an earlier arena overflow (a batch rescanned in pieces) on captures
         * that are mostly preamble -- and the context says so (msd_arena_permille). */
        msd_config smaller = *cfg;
        smaller.test_arena_permille = 1000;
        oom = false;
        rc = create_context(&smaller, out, &oom);
    }
    return rc;
}

int msd_arena_permille(const msd_ctx *ctx)
{
    if (!ctx)
        return -EINVAL;
    return ctx->cfg.test_arena_permille > 0 ? ctx->cfg.test_arena_permille : 4000;
}

void msd_destroy(msd_ctx *ctx)
{
    destroy(ctx);
}

const char *msd_last_error(const msd_ctx *ctx)
{
    return ctx ? ctx->err : g_create_err; /* NULL: why this thread's last msd_create failed */
}

int msd_reset(msd_ctx *c)
{
    if (!c)
        return -EINVAL;
    if (c->outstanding && !c->failed)
        return fail(c, -EBUSY, "batches outstanding");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->failed) { /* drop whatever was in flight when a batch failed */
        HIPCHK(c, hipDeviceSynchronize());
        for (Slot &s : c->slots) {
            s.busy = false;
            s.resolve_inflight = false;
            s.ahead_done = false;
            s.ahead_verdict = 0;
        }
        c->head = 0;
        c->outstanding = 0;
        c->pending_emit = nullptr;
        c->failed = false;
    }
    c->next_sample = 0;
    c->have_prev = false;
    c->finished = false;
    c->pending_dropped = 0;
    if (c->d_dcstate)
        HIPCHK(c, hipMemset(c->d_dcstate, 0, 2 * sizeof(float)));
    c->helper.wait();
    msd_resolver_reset(&c->resolver);
    memset(&c->timing, 0, sizeof c->timing);
    return 0;
}

int msd_decode_fields_device(msd_ctx *c, const msd_message *msgs, size_t n, msd_fields *out)
{
    if (!c || (n && (!msgs || !out)) || n > (1u << 24))
        return -EINVAL;
    if (n == 0)
        return 0;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    msd_message *d_in = nullptr;
    msd_fields *d_out = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_in), n * sizeof *d_in);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void **>(&d_out), n * sizeof *d_out);
    int rc = 0;
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_in, msgs, n * sizeof *d_in, hipMemcpyHostToDevice, c->aux_stream);
    if (e == hipSuccess)
        rc = msd_launch_fields(d_in, d_out, (uint32_t)n, c->aux_stream);
    if (e == hipSuccess && !rc)
        e = hipMemcpyAsync(out, d_out, n * sizeof *d_out, hipMemcpyDeviceToHost, c->aux_stream);
    if (e == hipSuccess && !rc)
        e = hipStreamSynchronize(c->aux_stream);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess)
        return fail(c, -EIO, "msd_decode_fields_device: %s", hipGetErrorString(e));
    return rc ? fail(c, rc, "field kernel launch failed") : 0;
}

int msd_restart(msd_ctx *c)
{
    if (!c)
        return -EINVAL;
    if (!c->outstanding)
        return msd_reset(c);
    if (!c->finished)
        return fail(c, -EINVAL, "msd_restart: the running capture has not been closed (last != 0) yet");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->next_sample = 0;
    c->have_prev = false;
    c->finished = false;
    c->pending_dropped = 0;
    if (c->d_dcstate) /* behind the converter kernels of the capture that is still draining */
        HIPCHK(c, hipMemsetAsync(c->d_dcstate, 0, 2 * sizeof(float), c->stream));
    c->restart_pending = true;
    return 0;
}

int msd_note_dropped(msd_ctx *c, uint64_t nsamples)
{
    if (!c)
        return -EINVAL;
    if (c->failed)
        return -EIO;
    if (c->finished)
        return fail(c, -EINVAL, "capture already finished; call msd_reset()");
    if (nsamples) {
        c->pending_dropped += nsamples;
        c->have_prev = false; /* MAGBUF_DISCONTINUOUS: fifo.c:178-181 zeroes the overlap */
    }
    return 0;
}

int msd_set_timing_interval(msd_ctx *c, uint32_t every)
{
    if (!c)
        return -EINVAL;
    c->timing_interval = every;
    c->enqueue_seq = 0;
    return 0;
}

int msd_set_preamble_threshold(msd_ctx *c, int threshold)
{
    if (!c)
        return -EINVAL;
    if (threshold < 1 || threshold > MSD_MAX_PREAMBLE_THRESHOLD)
        return fail(c, -EINVAL, "preamble threshold %d outside 1..%d", threshold, MSD_MAX_PREAMBLE_THRESHOLD);
    c->cfg.preamble_threshold = threshold;
    return 0;
}

int msd_launch_device(msd_ctx *c, const void *d_iq, uint64_t nsamples, int last)
{
    if (!c)
        return -EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return launch(c, d_iq, nsamples, last);
}

int msd_collect(msd_ctx *c, msd_message_fn sink, void *user)
{
    if (!c)
        return -EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return collect(c, sink, user);
}

int msd_collect_fields(msd_ctx *c, msd_fields_fn sink, void *user)
{
    if (!c)
        return -EINVAL;
    if (!c->want_fields)
        return fail(c, -EINVAL, "the context was created without MSD_CFG_DECODE_FIELDS");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->fsink = sink;
    c->fuser = user;
    const int rc = collect(c, nullptr, nullptr);
    c->fsink = nullptr;
    c->fuser = nullptr;
    return rc;
}

void msd_array_fields_sink(const msd_message *mm, const msd_fields *fields, void *state)
{
    msd_array_fields_sink_state *st = static_cast<msd_array_fields_sink_state *>(state);
    if (st->count < st->cap) {
        st->out[st->count] = *mm;
        st->fields[st->count] = *fields;
    }
    st->count++;
}

int msd_submit_device(msd_ctx *c, const void *d_iq, uint64_t nsamples, int last, msd_message_fn sink,
                      void *user)
{
    if (!c)
        return -EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    while (c->outstanding) {
        int rc = collect(c, sink, user);
        if (rc)
            return rc;
    }
    int rc = launch(c, d_iq, nsamples, last);
    if (rc)
        return rc;
    return collect(c, sink, user);
}

int msd_launch_host(msd_ctx *c, const void *h_iq, uint64_t nsamples, int last)
{
    if (!c)
        return -EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (nsamples > c->cfg.max_batch_samples)
        return fail(c, -E2BIG, "batch exceeds max_batch_samples");
    if (c->outstanding >= MSD_PIPELINE_DEPTH)
        return fail(c, -EBUSY, "pipeline full: call msd_collect() first");
    if (nsamples && !h_iq)
        return fail(c, -EINVAL, "IQ pointer must be non-null");
    Slot &s = c->slots[(c->head + c->outstanding) % MSD_PIPELINE_DEPTH];
    if (!s.d_upload) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&s.d_upload), c->cfg.max_batch_samples * c->bps + 64));
        HIPCHK(c, hipEventCreateWithFlags(&s.ev_upload, hipEventDisableTiming));
    }
    /* upload on the copy stream (it runs ahead of the kernels of the batches in front), the batch's
     * kernels wait for it through an event */
    if (nsamples)
        HIPCHK(c, hipMemcpyAsync(s.d_upload, h_iq, nsamples * c->bps, hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(c, hipEventRecord(s.ev_upload, c->copy_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, s.ev_upload, 0));
    return launch(c, s.d_upload, nsamples, last);
}

int msd_host_alloc(msd_ctx *c, size_t bytes, void **out)
{
    if (!c || !out)
        return -EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    *out = nullptr;
    HIPCHK(c, hipHostMalloc(out, bytes ? bytes : 1));
    return 0;
}

void msd_host_free(msd_ctx *c, void *p)
{
    if (c && p) {
        (void)hipSetDevice(c->cfg.device);
        (void)hipHostFree(p);
    }
}

int msd_thread_attach(msd_ctx *c)
{
    if (!c)
        return -EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    (void)hipStreamQuery(c->stream);
    (void)hipGetLastError();
    return 0;
}

int msd_host_register(msd_ctx *c, void *p, size_t bytes)
{
    if (!c || !p || !bytes)
        return -EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipHostRegister(p, bytes, hipHostRegisterDefault));
    return 0;
}

void msd_host_unregister(msd_ctx *c, void *p)
{
    if (c && p) {
        (void)hipSetDevice(c->cfg.device);
        (void)hipHostUnregister(p);
    }
}

int msd_submit_host(msd_ctx *c, const void *h_iq, uint64_t nsamples, int last, msd_message_fn sink,
                    void *user)
{
    if (!c)
        return -EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (nsamples > c->cfg.max_batch_samples)
        return fail(c, -E2BIG, "batch exceeds max_batch_samples");
    if (!c->d_stage)
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_stage), c->cfg.max_batch_samples * 4 + 64));
    while (c->outstanding) {
        int rc = collect(c, sink, user);
        if (rc)
            return rc;
    }
    if (nsamples)
        HIPCHK(c, hipMemcpyAsync(c->d_stage, h_iq, nsamples * c->bps, hipMemcpyHostToDevice, c->stream));
    return msd_submit_device(c, c->d_stage, nsamples, last, sink, user);
}

int msd_get_stats(const msd_ctx *c, msd_stats *st)
{
    if (!c || !st)
        return -EINVAL;
    const_cast<msd_ctx *>(c)->helper.wait(); /* the power statistics of the last batch are summed on the helper thread */
    *st = c->stats;
    return 0;
}

int msd_get_timing(const msd_ctx *c, msd_timing *t)
{
    if (!c || !t)
        return -EINVAL;
    *t = c->timing;
    return 0;
}

int msd_dc_filter_status(msd_ctx *c, uint32_t out[4])
{
    if (!c || !out || !c->dc)
        return -EINVAL;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!c->d_dc_work || !c->dc_last_parallel)
        return 0;
    uint32_t ctl[12] = {0};
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(ctl, c->d_dc_work, sizeof ctl, hipMemcpyDeviceToHost));
    out[0] = ctl[0];  /* DcpCtl.done */
    out[1] = ctl[9] > ctl[10] ? ctl[9] : ctl[10]; /* .passes_ch */
    out[2] = ctl[11]; /* .guessed */
    out[3] = c->dc_last_blocks;
    return 0;
}

int msd_get_buffer_means(const msd_ctx *c, double *means, size_t cap)
{
    if (!c || !means)
        return -EINVAL;
    size_t n = c->means.size() / 2;
    for (size_t i = 0; i < n && i < cap; ++i) {
        means[2 * i] = c->means[2 * i];
        means[2 * i + 1] = c->means[2 * i + 1];
    }
    return (int)n;
}

int msd_convert_begin(msd_ctx *c, const void *iq_data, uint16_t *mag_data, unsigned nsamples)
{
    if (!c || c->cfg.format == MSD_FMT_MAG16)
        return -EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (nsamples > c->cfg.max_batch_samples)
        return fail(c, -E2BIG, "nsamples exceeds max_batch_samples");
    if (c->conv_pending)
        return fail(c, -EBUSY, "a conversion is in flight: msd_convert_end() first");
    if (!c->d_stage)
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_stage), c->cfg.max_batch_samples * 4 + 64));
    if (!c->d_mag)
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_mag), c->cfg.max_batch_samples * 2 + 64));
    if (!c->h_conv) { /* where the sums come home: page-locked, so that nothing of this call waits for the device */
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_conv), 4 * sizeof(uint64_t)));
    }
    Slot &s = c->slots[0];
    if (c->outstanding)
        return fail(c, -EBUSY, "batches outstanding");
    memset(c->h_conv, 0, 4 * sizeof(uint64_t));
    float *h_fm = reinterpret_cast<float *>(c->h_conv + 2);
    HIPCHK(c, hipMemsetAsync(s.d_sums, 0, 2 * sizeof(uint64_t), c->stream));
    if (c->dc) {
        /* convert_*_generic (convert.c:113-213, 374-423): the DC estimate of the two channels lives in the context, as in
         * the reference's struct converter_state, and runs on from call to call -- a context that converts this way is a
         * converter and nothing else (msd_launch_* of the same context would advance the same state) */
        if (nsamples) {
            if (!c->d_conv_magsq)
                HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_conv_magsq), c->cfg.max_batch_samples * sizeof(float) + 64));
            HIPCHK(c, hipMemcpyAsync(c->d_stage, iq_data, (size_t)nsamples * c->bps, hipMemcpyHostToDevice, c->stream));
            int rc = launch_dc_block(c, c->d_stage, nsamples, c->d_mag, c->d_conv_magsq, c->stream);
            if (!rc)
                rc = msd_launch_dc_sums(c->d_conv_magsq, nsamples, nsamples, 1, s.d_fmeans, c->d_fm_work, 0, c->stream);
            if (rc)
                return fail(c, rc, "DC filter converter launch failed");
            HIPCHK(c, hipMemcpyAsync(mag_data, c->d_mag, (size_t)nsamples * 2, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(h_fm, s.d_fmeans, 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        }
        c->conv_pending = true;
        c->conv_n = nsamples;
        return 0;
    }
    if (nsamples) {
        HIPCHK(c, hipMemcpyAsync(c->d_stage, iq_data, (size_t)nsamples * c->bps, hipMemcpyHostToDevice, c->stream));
        int rc = c->q11_bits ? msd_launch_q11_table(c->d_stage, nsamples, c->d_q11_table, c->q11_bits, c->d_mag,
                                                    reinterpret_cast<unsigned long long *>(s.d_sums), c->cu_count, c->stream)
                             : msd_launch_convert(c->cfg.format, c->d_stage, nsamples, c->d_lut, c->d_mag,
                                                  reinterpret_cast<unsigned long long *>(s.d_sums), c->stream);
        if (rc)
            return fail(c, rc, "convert kernel launch failed");
        if (c->cfg.format != MSD_FMT_UC8 && !c->q11_bits) {
            rc = msd_launch_float_means(c->cfg.format, c->d_stage, nsamples, nsamples, 1, s.d_fmeans, nullptr, c->d_fm_work, 0, c->stream);
            if (rc)
                return fail(c, rc, "float means kernel launch failed");
        }
        HIPCHK(c, hipMemcpyAsync(mag_data, c->d_mag, (size_t)nsamples * 2, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->h_conv, s.d_sums, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    if (c->cfg.format != MSD_FMT_UC8 && !c->q11_bits && nsamples)
        HIPCHK(c, hipMemcpyAsync(h_fm, s.d_fmeans, 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    c->conv_pending = true;
    c->conv_n = nsamples;
    return 0;
}

int msd_convert_end(msd_ctx *c, double *out_mean_level, double *out_mean_power)
{
    if (!c)
        return -EINVAL;
    if (!c->conv_pending)
        return fail(c, -EINVAL, "no conversion in flight");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->conv_pending = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const unsigned nsamples = c->conv_n;
    const uint64_t *sums = c->h_conv;
    const float *fm = reinterpret_cast<const float *>(c->h_conv + 2);
    if (!c->dc && (c->cfg.format == MSD_FMT_UC8 || c->q11_bits)) { /* integer sums: convert.c:104-110 and :318-326 */
        if (out_mean_level)
            *out_mean_level = (double)sums[0] / 65536.0 / (double)nsamples;
        if (out_mean_power)
            *out_mean_power = (double)sums[1] / 65535.0 / 65535.0 / (double)nsamples;
    } else {
        if (out_mean_level)
            *out_mean_level = (double)(fm[0] / (float)nsamples);
        if (out_mean_power)
            *out_mean_power = (double)(fm[1] / (float)nsamples);
    }
    return 0;
}

int msd_convert(msd_ctx *c, const void *iq_data, uint16_t *mag_data, unsigned nsamples,
                double *out_mean_level, double *out_mean_power)
{
    int rc = msd_convert_begin(c, iq_data, mag_data, nsamples);
    if (!rc)
        rc = msd_convert_end(c, out_mean_level, out_mean_power);
    return rc;
}

int msd_demodulate_magbufs(msd_ctx *c, const msd_magbuf_view *bufs, unsigned n, msd_message_fn sink, void *user)
{
    if (!c || !bufs || !n)
        return -EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if ((uint64_t)n * MSD_CHUNK_SAMPLES > c->cfg.max_batch_samples || n > c->max_buffers)
        return fail(c, -E2BIG, "more mag_bufs than max_batch_samples holds");
    for (unsigned k = 0; k < n; ++k) {
        const msd_magbuf_view &b = bufs[k];
        if (!b.data || b.overlap != MSD_OVERLAP || b.validLength < b.overlap || b.validLength - b.overlap > MSD_CHUNK_SAMPLES ||
            (k + 1 < n && b.validLength - b.overlap != MSD_CHUNK_SAMPLES))
            return fail(c, -EINVAL, "mag_buf geometry must be overlap=326, 131072 new samples (the last one: at most)");
    }
    if (c->outstanding)
        return fail(c, -EBUSY, "batches outstanding");
    if (!c->d_stage)
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_stage), c->cfg.max_batch_samples * 4 + 64));
    Slot &s = c->slots[0];
    /* the first buffer's data[0..326) -> the two-sample-padded "previous tail"; every buffer's data[326..) -> the batch */
    uint8_t *tail = c->d_tail[0];
    HIPCHK(c, hipMemsetAsync(tail, 0, (size_t)(TAIL_SAMPLES - MSD_OVERLAP) * 2, c->stream));
    HIPCHK(c, hipMemcpyAsync(tail + (size_t)(TAIL_SAMPLES - MSD_OVERLAP) * 2, bufs[0].data, (size_t)MSD_OVERLAP * 2,
                             hipMemcpyHostToDevice, c->stream));
    uint64_t total = 0;
    std::vector<uint64_t> ts(2 * (size_t)n);
    std::vector<double> means(2 * (size_t)n);
    std::vector<uint32_t> noise(n);
    for (unsigned k = 0; k < n; ++k) {
        const msd_magbuf_view &b = bufs[k];
        const unsigned mlen = b.validLength - b.overlap;
        if (mlen)
            HIPCHK(c, hipMemcpyAsync(c->d_stage + (size_t)k * MSD_CHUNK_SAMPLES * 2, b.data + b.overlap, (size_t)mlen * 2,
                                     hipMemcpyHostToDevice, c->stream));
        total += mlen;
        ts[2 * k] = b.sampleTimestamp;
        ts[2 * k + 1] = b.sysTimestamp;
        means[2 * k] = b.mean_level;
        means[2 * k + 1] = b.mean_power;
        /* demod_2400.c:530-531 from the caller's mag_buf.mean_level / .mean_power */
        const double noise_stddev = sqrt(b.mean_power - b.mean_level * b.mean_level);
        noise[k] = mlen ? (uint32_t)((b.mean_power + noise_stddev) * 65535 + 0.5) : 0u;
    }
    s.busy = true;
    s.d_iq = c->d_stage;
    s.d_prev = tail;
    s.have_prev = 1;
    s.threshold = c->cfg.preamble_threshold;
    s.dropped_before = 0;
    s.gpu_resolve = false; /* the caller's clocks and means: the host resolver */
    s.resolve_inflight = false;
    s.dc = false;
    s.batch_first = 0;
    s.nsamples = total;
    s.nbuffers = n;
    s.last = 1;
    int rc = enqueue(c, s, MSD_FMT_MAG16, c->cfg.mode_ac ? noise.data() : nullptr);
    if (rc) {
        s.busy = false;
        return rc;
    }
    c->magbuf_views = bufs;
    c->magbuf_nviews = n;
    c->magbuf_noise = c->cfg.mode_ac ? noise.data() : nullptr;
    rc = finish(c, s, MSD_FMT_MAG16, sink, user, ts.data(), means.data(), 0);
    c->magbuf_views = nullptr;
    c->magbuf_nviews = 0;
    c->magbuf_noise = nullptr;
    /* the stream interface's tail ring was borrowed: a following msd_submit_* starts afresh */
    c->have_prev = false;
    c->tail_cur = 0;
    return rc;
}

int msd_demodulate_magbuf(msd_ctx *c, const uint16_t *data, unsigned validLength, unsigned overlap,
                          uint64_t sampleTimestamp, uint64_t sysTimestamp, double mean_level,
                          double mean_power, msd_message_fn sink, void *user)
{
    const msd_magbuf_view one = {data, validLength, overlap, sampleTimestamp, sysTimestamp, mean_level, mean_power};
    return msd_demodulate_magbufs(c, &one, 1, sink, user);
}

} /* extern "C" */
